"""Test infrastructure: the flat optimizer passes of csrc/nm_optim.hip (nm_optim_partials / nm_optim_segments /
nm_optim_apply) restated with torch-CPU float32 arithmetic over the SAME chunk table (ops.optimizer_chunk_table), for
the gloo world-size-2 tests of the sharded optimizer (distributed.DataParallel.optimizer_step takes the tables as an
argument: libnmhip's kernels in the product path, this stand-in on CPU).  Reference arithmetic:
trainers/generic_trainer.py:84-195 (L1 / L2 over the non-bias variables, tf.clip_by_norm per tensor, Adam)."""
import torch

from neuralmonkey_amd import ops


class CpuOptimizerTables:
    def __init__(self, store, regularizable, trainable, cuts=()):
        (self.starts, self.lens, self.segs, self.seg_first, self.seg_count,
         self.seg_flags) = ops.optimizer_chunk_table(store, regularizable, trainable, cuts, chunk=256)
        self.nchunk, self.nseg = len(self.starts), len(self.seg_first)
        self.workspace = torch.zeros(3 * self.nchunk + self.nseg, dtype=torch.float32)
        self.l1l2 = torch.zeros(2, dtype=torch.float32)

    def chunk_range(self, lo, hi):
        return ops.chunk_range_of(self.starts, self.lens, lo, hi)

    def partial_vector(self):
        return self.workspace[:3 * self.nchunk]

    def chunk_list(self, ranges):
        """The chunk indices of the (begin, end) ranges (the product keeps this list on the device)."""
        return [c for b, e in ranges for c in range(int(b), int(e))]

    def partials(self, theta, grad, l1_weight, l2_weight, chunks, chunk_list=None):
        part = self.partial_vector()
        for c in (chunk_list if chunk_list is not None else range(*chunks)):
            lo, n = self.starts[c], self.lens[c]
            th, g = theta[lo:lo + n], grad[lo:lo + n]
            a1 = a2 = torch.zeros((), dtype=torch.float32)
            if self.seg_flags[self.segs[c]] & 1:
                a1, a2 = th.abs().sum(), (th * th).sum()
                g += torch.tensor(l1_weight, dtype=torch.float32) * torch.sign(th) \
                    + torch.tensor(2.0 * l2_weight, dtype=torch.float32) * th
            part[3 * c], part[3 * c + 1], part[3 * c + 2] = (g * g).sum(), a1, a2

    def segments(self):
        part = self.partial_vector()
        norm2 = self.workspace[3 * self.nchunk:]
        l1 = l2 = torch.zeros((), dtype=torch.float32)
        for s in range(self.nseg):
            gs = a1 = a2 = torch.zeros((), dtype=torch.float32)
            for c in range(self.seg_first[s], self.seg_first[s] + self.seg_count[s]):     # fixed chunk order
                gs, a1, a2 = gs + part[3 * c], a1 + part[3 * c + 1], a2 + part[3 * c + 2]
            norm2[s] = gs
            l1, l2 = l1 + a1, l2 + a2
        self.l1l2[0], self.l1l2[1] = l1, l2
        return self.l1l2

    def apply(self, kind, theta, grad, slot0, slot1, clip_norm, params, skip=None, chunks=None, chunk_list=None):
        assert kind == 0, "the stand-in restates Adam"
        if skip is not None and int(skip.item()) != 0:
            return
        lr_t, b1, b2, eps = (torch.tensor(float(p), dtype=torch.float32) for p in params)
        norm2 = self.workspace[3 * self.nchunk:]
        one = torch.tensor(1.0, dtype=torch.float32)
        for c in (chunk_list if chunk_list is not None else range(*(chunks if chunks is not None else (0, self.nchunk)))):
            seg = self.segs[c]
            if not self.seg_flags[seg] & 2:
                continue
            lo, n = self.starts[c], self.lens[c]
            scale = one
            if clip_norm:
                clip = torch.tensor(float(clip_norm), dtype=torch.float32)
                scale = clip / torch.maximum(torch.sqrt(norm2[seg]), clip)
            g = grad[lo:lo + n] * scale
            m = b1 * slot0[lo:lo + n] + (one - b1) * g
            v = b2 * slot1[lo:lo + n] + (one - b2) * g * g
            slot0[lo:lo + n], slot1[lo:lo + n] = m, v
            theta[lo:lo + n] -= lr_t * m / (torch.sqrt(v) + eps)
