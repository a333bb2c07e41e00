"""Single-node data parallelism: one process per GPU, RCCL over xGMI.

New capability (the reference has no multi-GPU code).  Each rank
holds a full replica and a B/N slice of the minibatch; the only exchange per
optimizer step is the all-reduce (sum) of the flat gradient buffer plus one
scalar all-reduce of the target-token count so that every rank back-propagates
sum_local(xent) / sum_global(mask).  L1/L2 terms, per-tensor
clip_by_norm and Adam run after the reduction on identical data on every rank.

The gradient lives in ONE contiguous buffer (variables.py), so the exchange is
a handful of large collectives (``bucket_bytes`` each) instead of one per
tensor; on the fully connected 8-GPU xGMI mesh large messages are what lets
RCCL drive all seven links per GPU.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist

_CURRENT: Optional["DataParallel"] = None


class DataParallel:
    def __init__(self, bucket_bytes: int = 64 << 20) -> None:
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.rank = dist.get_rank()
        self.world_size = dist.get_world_size()
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = os.environ.get("NM_DP_OVERLAP", "1") != "0"
        self._handles: list = []          # collectives in flight this step
        self._early: list = []            # [lo, hi) spans of the flat gradient already being reduced
        # Host scalars (the global target-token count of a step) travel over a gloo side group:
        # reading an RCCL result back would synchronise the device every step and stop the host
        # from enqueueing ahead of the GPU.
        self._host_group = None
        if self.world_size > 1 and dist.get_backend() == "nccl":
            try:
                self._host_group = dist.new_group(backend="gloo")
            except Exception:           # pylint: disable=broad-except
                self._host_group = None  # fall back to a device all-reduce + readback

    def all_reduce_scalar(self, value: float) -> float:
        """Sum of a host scalar over ranks (global target-token count)."""
        if self.world_size == 1:
            return value
        if dist.get_backend() != "nccl" or self._host_group is not None:
            t = torch.tensor([value], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._host_group)
            return float(t.item())
        t = torch.tensor([value], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def begin_step(self) -> None:
        """Forget the bookkeeping of a step that did not reach ``all_reduce_gradients`` (an exception)."""
        for hnd in self._handles:
            hnd.wait()
        self._handles, self._early = [], []

    def _reduce_span(self, grad, lo: int, hi: int) -> None:
        for start in range(lo, hi, self.bucket_elems):
            chunk = grad[start:min(hi, start + self.bucket_elems)]
            self._handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))

    def all_reduce_early(self, store, names) -> None:
        """Start summing the gradient slices of ``names`` NOW, while the rest of the backward pass still
        runs: the vocabulary projection (65 MB at V=32k) is final after the first GEMMs of the backward
        pass, the decoder embeddings after the decoder's BPTT -- together more than half of the 228 MB
        exchanged per step.  The collective is ordered after the *current* stream (call it
        on the stream that produced the slices); ``all_reduce_gradients`` later skips these spans and
        waits for them.  A caller must only name variables that receive no further contributions."""
        if self.world_size == 1 or not self.overlap:
            return
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
        grad = store.ensure_grad()
        spans = sorted((store.offset(n), store.offset(n) + store[n].numel()) for n in names)
        merged = []
        for lo, hi in spans:
            if merged and lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
            else:
                merged.append([lo, hi])
        for lo, hi in merged:
            if any(lo < dhi and dlo < hi for dlo, dhi in self._early):
                raise RuntimeError("gradient span [{}, {}) was already reduced this step".format(lo, hi))
            self._early.append((lo, hi))
            self._reduce_span(grad, lo, hi)

    def all_reduce_gradients(self, store) -> None:
        """In-place sum of the flat gradient buffer over ranks, in large buckets (minus the spans
        ``all_reduce_early`` already started); returns with every collective of the step ordered
        before the current stream."""
        if self.world_size == 1:
            return
        grad = store.ensure_grad()
        pos = 0
        for lo, hi in sorted(self._early):
            if lo > pos:
                self._reduce_span(grad, pos, lo)
            pos = max(pos, hi)
        if pos < grad.numel():
            self._reduce_span(grad, pos, grad.numel())
        for hnd in self._handles:
            hnd.wait()
        self._handles, self._early = [], []

    def broadcast_parameters(self, store, src: int = 0) -> None:
        """Make every replica start from rank ``src``'s variables."""
        if self.world_size > 1:
            dist.broadcast(store.theta, src=src)

    def shard(self, dataset):
        """This rank's contiguous share of a batch (SURVEY 8e partitioning): the rows are dealt as evenly as
        possible (sizes differ by at most one), so no rank is left without work while another holds two rows
        more.  A batch with fewer rows than ranks cannot be sharded: every rank sees the same batch and raises
        the same error -- nobody is left waiting in a collective."""
        n = len(dataset)
        if n < self.world_size:
            raise ValueError("a batch of {} sentences cannot be sharded over {} ranks: drop or pad the last batch "
                             "(dataset.BatchingScheme(drop_remainder=True))".format(n, self.world_size))
        base, rem = divmod(n, self.world_size)
        start = self.rank * base + min(self.rank, rem)
        return dataset.subset(start, base + (1 if self.rank < rem else 0))


def init_from_env(backend: Optional[str] = None) -> Optional[DataParallel]:
    """Initialise from torchrun's RANK / WORLD_SIZE / MASTER_* variables."""
    global _CURRENT
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # a world of one trains without a process group -- unless NM_DIST_FORCE is set: the single-GPU smoke test of
    # the RCCL code path (process-group set-up, bucketed / early all-reduce on its streams) with nobody to talk to
    if world <= 1 and not dist.is_initialized() and not os.environ.get("NM_DIST_FORCE"):
        _CURRENT = None
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("NM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    _CURRENT = DataParallel()
    return _CURRENT


def current() -> Optional[DataParallel]:
    return _CURRENT


def shutdown() -> None:
    global _CURRENT
    _CURRENT = None
    if dist.is_initialized():
        dist.destroy_process_group()
