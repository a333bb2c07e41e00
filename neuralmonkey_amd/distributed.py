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
        # NM_DIST_FORCE=1: a world of one issues its collectives all the same (every one of them is the identity):
        # the single-GPU check of the exchange code -- ordering against the streams, early spans, buckets
        self.forced = os.environ.get("NM_DIST_FORCE") == "1"
        # NM_DP_SPARSE_EMB=1: an embedding matrix whose gradient touches only the rows of this rank's tokens travels
        # as (row ids, rows) instead of as a dense [V, E] slice of the flat buffer (exchange_sparse_rows)
        self.sparse_embeddings = os.environ.get("NM_DP_SPARSE_EMB", "0") == "1"
        self._sparse_bufs: dict = {}
        self.sparse_bytes_per_step = 0
        self._handles: list = []          # collectives in flight this step
        self._early: list = []            # [lo, hi) spans of the flat gradient already being reduced
        # Early spans are ORDERED on the device where their producer stands (an event), but their collectives are
        # ENQUEUED by the host only at the end of the backward pass (all_reduce_gradients): enqueueing an RCCL
        # collective behind unfinished work holds the host thread until that work is done -- 1 ms of every step at
        # the headline shape, the main stream idle meanwhile (profiles/r05_dp_early_issue.txt).  The host runs
        # milliseconds ahead of the device, so the collectives are still in the queue long before their events
        # fire.  NM_DP_EARLY_ISSUE=now: the round-4 behaviour.
        self.defer_issue = os.environ.get("NM_DP_EARLY_ISSUE", "deferred") != "now"
        self._early_pending: list = []    # (event, lo, hi) not yet enqueued
        self._issue_stream = None
        self._loops_done = None           # event behind the step's last cluster time loop (after_time_loops)
        # optional accounting of the exchange (bench.py --gpus N): event pairs on the compute stream around the
        # point where it has to wait for the collectives = the part of the all-reduce that is NOT hidden
        self.timing = False
        self._timed: list = []
        self.bytes_per_step = 0
        self.early_bytes_per_step = 0
        # Host scalars (the global target-token count of a step) travel over a gloo side group:
        # reading an RCCL result back would synchronise the device every step and stop the host
        # from enqueueing ahead of the GPU.
        # NM_DIST_ALLREDUCE=nmhip: the gradient buckets go through the library's own RCCL communicator
        # (nm_allreduce_*, csrc/nm_comm.hip) instead of torch.distributed's; torch's process group stays for what is
        # not on the step's critical path (parameter broadcast, the unique id, host scalars).  Opt-in: the default
        # path is the one the multi-rank tests have run on.
        self._comm = None
        if os.environ.get("NM_DIST_ALLREDUCE", "torch") == "nmhip":
            self._init_library_communicator()
        self._host_group = None
        if self.world_size > 1 and dist.get_backend() == "nccl":
            try:
                self._host_group = dist.new_group(backend="gloo")
            except Exception:           # pylint: disable=broad-except
                self._host_group = None  # fall back to a device all-reduce + readback

    def _init_library_communicator(self) -> None:
        import ctypes
        from . import _lib
        if dist.get_backend() != "nccl" or not torch.cuda.is_available():
            raise RuntimeError("NM_DIST_ALLREDUCE=nmhip needs the nccl (RCCL) backend on GPUs")
        lib = _lib.load()
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.nm_allreduce_unique_id(uid, 128), "nm_allreduce_unique_id")
        box = [uid.raw]
        if self.world_size > 1:
            dist.broadcast_object_list(box, src=0)            # any side channel does: here torch's own group
        handle = ctypes.c_void_p()
        _lib.check(lib.nm_allreduce_init(self.rank, self.world_size, ctypes.create_string_buffer(box[0], 128),
                                         ctypes.byref(handle)), "nm_allreduce_init")
        self._comm = handle

    def close(self) -> None:
        if self._comm is not None:
            from . import _lib
            _lib.check(_lib.load().nm_allreduce_destroy(self._comm), "nm_allreduce_destroy")
            self._comm = None

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

    def scale_by_global_count(self, count: float, weight: float, scale: torch.Tensor) -> None:
        """``scale[0] = weight / sum over ranks of count`` (0 when nobody has a target token): the factor every rank
        back-propagates with, sum_local(xent) / sum_global(mask) (decoders/autoregressive.py:313-316).

        With RCCL the sum never visits the host: the local count goes into a device scalar, the all-reduce is
        enqueued behind the current stream and the division runs on the device, so the host thread goes straight on
        to enqueueing the forward pass -- the first reader of ``scale`` is the cross-entropy backward, a whole
        forward pass later.  (The host-side gloo exchange this replaces blocked every rank at the top of every
        step until the slowest rank had arrived.)  Other backends (gloo on CPU / in tests) keep the host sum."""
        if (self.world_size == 1 and not self.forced) or dist.get_backend() != "nccl" or scale.device.type != "cuda":
            total = self.all_reduce_scalar(count)
            scale.fill_(weight / total if total else 0.0)
            return
        t = torch.full((1,), float(count), dtype=torch.float64, device=scale.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                  # the stream waits for it, the host does not
        scale.copy_(torch.where(t > 0, float(weight) / t.clamp_min(1.0), torch.zeros_like(t)).to(scale.dtype))

    def _wait_handles(self) -> None:
        """The current stream waits for every collective of the step enqueued so far."""
        if self._comm is not None:
            if self._handles:
                from . import _lib
                _lib.check(_lib.load().nm_allreduce_wait(self._comm, torch.cuda.current_stream().cuda_stream),
                           "nm_allreduce_wait")
            return
        for hnd in self._handles:
            hnd.wait()

    def begin_step(self) -> None:
        """Forget the bookkeeping of a step that did not reach ``all_reduce_gradients`` (an exception)."""
        self._wait_handles()
        self._handles, self._early, self._early_pending = [], [], []
        self._loops_done = None
        self.sparse_bytes_per_step = 0

    def _reduce_span(self, grad, lo: int, hi: int) -> None:
        for start in range(lo, hi, self.bucket_elems):
            chunk = grad[start:min(hi, start + self.bucket_elems)]
            if self._comm is not None:
                from . import _lib
                _lib.check(_lib.load().nm_allreduce_bucket(self._comm, torch.cuda.current_stream().cuda_stream,
                                                           chunk.data_ptr(), chunk.numel()), "nm_allreduce_bucket")
                self._handles.append(None)
            else:
                self._handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))

    def all_reduce_early(self, store, names) -> None:
        """Start summing the gradient slices of ``names`` NOW, while the rest of the backward pass still
        runs: the vocabulary projection (65 MB at V=32k) is final after the first GEMMs of the backward
        pass, the decoder embeddings after the decoder's BPTT -- together more than half of the 228 MB
        exchanged per step.  The collective is ordered after the *current* stream (call it
        on the stream that produced the slices); ``all_reduce_gradients`` later skips these spans and
        waits for them.  A caller must only name variables that receive no further contributions."""
        if (self.world_size == 1 and not self.forced) or not self.overlap:
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
            if self.defer_issue and grad.is_cuda:
                done = torch.cuda.Event()
                done.record()                  # on the producer's stream
                self._early_pending.append((done, lo, hi))
            else:
                self._reduce_span(grad, lo, hi)

    def after_time_loops(self) -> None:
        """Called on the stream that just launched a cluster time loop (ops.gru_seq_bwd): early collectives start
        only once the LAST such loop of the step is done.  A cluster loop needs every CU to take one of its
        workgroups (320 of a SIMD's 512 registers beside a capped GEMM workgroup's 160); an RCCL workgroup that sits
        on a CU when the loop is launched would keep the whole loop waiting for the collective to finish.  The
        leaf GEMMs behind the loops (~2 ms at the headline shape) still hide the early spans' exchange."""
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            self._loops_done = torch.cuda.Event()
            self._loops_done.record()

    def _issue_early(self, grad) -> None:
        """Enqueue the collectives of the early spans, each ordered after the event its producer left behind (and
        after the step's last cluster time loop)."""
        if not self._early_pending:
            return
        if self._issue_stream is None:
            self._issue_stream = torch.cuda.Stream(device=grad.device)
        pending, self._early_pending = self._early_pending, []
        with torch.cuda.stream(self._issue_stream):
            if self._loops_done is not None:
                self._issue_stream.wait_event(self._loops_done)
            for done, lo, hi in pending:
                self._issue_stream.wait_event(done)
                self._reduce_span(grad, lo, hi)

    def _host_all_gather_int(self, value: int):
        """``value`` of every rank, in rank order (host side: the gloo group next to RCCL, or the gloo world)."""
        if dist.get_backend() != "nccl" or self._host_group is not None:
            mine = torch.tensor([int(value)], dtype=torch.int64)
            out = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world_size)]
            dist.all_gather(out, mine, group=self._host_group)
            return [int(t.item()) for t in out]
        dev = torch.device("cuda", torch.cuda.current_device())
        mine = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        out = torch.zeros(self.world_size, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, mine)
        return [int(x) for x in out.cpu().tolist()]

    def exchange_sparse_rows(self, store, name: str, host_ids, gather_rows, scatter_add, negate) -> bool:
        """Sum the gradient of the embedding matrix ``name`` over ranks by exchanging only the rows that are not
        zero: every rank contributes the rows of ITS tokens (``host_ids``: the ids the rank embedded this step,
        pad id 0 excluded), B*S rows of E floats instead of V rows -- 13 MB instead of 65 MB per rank at the
        benchmark shape.  Protocol: (1) the local dense gradient is complete (the caller's
        scatter-add ran); (2) unique sorted ids -> the dense rows at those ids are gathered into a [cap, E] block,
        cap = the largest count of any rank; (3) all-gather of the id and row blocks; (4) the local rows are
        cancelled (x + (-x) = +0 exactly) and every rank's block is added in RANK ORDER, one launch per rank, no id
        twice within a launch -- so every element sees the same additions in the same order on every rank and
        the replicas stay bit-identical, which atomics over raw token rows would not guarantee; (5) the span is
        marked reduced: ``all_reduce_gradients`` skips it.  ``gather_rows(src, idx, dst)``,
        ``scatter_add(table, ids, rows)`` and ``negate(src, dst)`` are the device primitives (libnmhip kernels in the
        product path).  Returns False when the exchange did not happen (world of one, switch off, graph capture):
        the dense all-reduce then covers the span as usual."""
        import numpy as np
        if self.world_size == 1 or not self.sparse_embeddings:
            return False
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return False
        table = store.g(name)                                        # [V, E] view of the flat gradient
        vsz, esz = table.shape
        lo = store.offset(name)
        hi = lo + table.numel()
        if any(lo < dhi and dlo < hi for dlo, dhi in self._early):
            raise RuntimeError("gradient span [{}, {}) was already reduced this step".format(lo, hi))
        ids = np.unique(np.asarray(host_ids).reshape(-1))
        ids = ids[ids != 0].astype(np.int32)
        counts = self._host_all_gather_int(len(ids))
        cap = max(256, -(-max(counts) // 256) * 256)
        dev = table.device
        bufs = self._sparse_bufs.get(name)
        if bufs is None or bufs["cap"] < cap:
            bufs = {"cap": cap,
                    "ids": torch.zeros(cap, dtype=torch.int32, device=dev),
                    "rows": torch.zeros(cap * esz, dtype=torch.float32, device=dev),
                    "neg": torch.zeros(cap * esz, dtype=torch.float32, device=dev),
                    "all_ids": torch.zeros(self.world_size * cap, dtype=torch.int32, device=dev),
                    "all_rows": torch.zeros(self.world_size * cap * esz, dtype=torch.float32, device=dev)}
            self._sparse_bufs[name] = bufs
        n = len(ids)
        padded = np.zeros(cap, np.int32)
        padded[:n] = ids
        my_ids = bufs["ids"][:cap]
        my_ids.copy_(torch.from_numpy(padded))
        rows = bufs["rows"][:cap * esz].view(cap, esz)
        neg = bufs["neg"][:cap * esz].view(cap, esz)
        if n:
            gather_rows(table, my_ids[:n], rows[:n])
        all_ids = bufs["all_ids"][:self.world_size * cap]
        all_rows = bufs["all_rows"][:self.world_size * cap * esz]
        dist.all_gather_into_tensor(all_ids, my_ids)
        dist.all_gather_into_tensor(all_rows, rows.view(-1))
        if n:
            negate(rows[:n], neg[:n])
            scatter_add(table, my_ids[:n], neg[:n])                  # exactly zero again
        all_ids = all_ids.view(self.world_size, cap)
        all_rows = all_rows.view(self.world_size, cap, esz)
        for r, cnt in enumerate(counts):
            if cnt:
                scatter_add(table, all_ids[r, :cnt], all_rows[r, :cnt])
        self._early.append((lo, hi))
        self.sparse_bytes_per_step += 4 * cap * (esz + 1)
        return True

    def all_reduce_gradients(self, store) -> None:
        """In-place sum of the flat gradient buffer over ranks, in large buckets (minus the spans
        ``all_reduce_early`` already started); returns with every collective of the step ordered
        before the current stream."""
        if self.world_size == 1 and not self.forced:
            return
        grad = store.ensure_grad()
        self._issue_early(grad)
        pos = 0
        for lo, hi in sorted(self._early):
            if lo > pos:
                self._reduce_span(grad, pos, lo)
            pos = max(pos, hi)
        if pos < grad.numel():
            self._reduce_span(grad, pos, grad.numel())
        timed = self.timing and grad.is_cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self._wait_handles()
        if timed:
            ev1.record()
            self._timed.append((ev0, ev1))
        self.bytes_per_step = 4 * grad.numel()
        self.early_bytes_per_step = 4 * sum(hi - lo for lo, hi in self._early)
        self.sparse_bytes_last, self.sparse_bytes_per_step = self.sparse_bytes_per_step, 0
        self._handles, self._early = [], []

    def exchange_report(self) -> dict:
        """Mean exposed wait per step since the last call (``timing`` on), bytes exchanged per step and how many of
        them were started early, from inside the backward pass."""
        waits = []
        for first, second in self._timed:
            second.synchronize()
            waits.append(first.elapsed_time(second))
        self._timed = []
        return {"ranks_seen": self.ranks_seen(),
                "allreduce_exposed_ms": (sum(waits) / len(waits)) if waits else None, "steps": len(waits),
                "bytes": self.bytes_per_step, "early_bytes": self.early_bytes_per_step,
                "sparse_rows_bytes": getattr(self, "sparse_bytes_last", 0),
                "buckets_mb": self.bucket_elems * 4 / 2 ** 20}

    def ranks_seen(self) -> int:
        """How many distinct ranks answer an all-gather over the process group (RCCL on GPUs): the proof, inside a
        bench line, that N processes really exchanged data (tools/scale.sh asserts it equals --gpus)."""
        if getattr(self, "_ranks_seen", None) is None:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            mine = torch.tensor([self.rank], dtype=torch.int64, device=dev)
            everyone = [torch.zeros_like(mine) for _ in range(self.world_size)]
            dist.all_gather(everyone, mine)
            self._ranks_seen = len({int(t.item()) for t in everyone})
        return self._ranks_seen

    def broadcast_parameters(self, store, src: int = 0) -> None:
        """Make every replica start from rank ``src``'s variables."""
        if self.world_size > 1:
            dist.broadcast(store.theta, src=src)
            store.epoch += 1              # (a collective wrote the variables: Session.variables_signature)

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
    if _CURRENT is not None:
        _CURRENT.close()
    _CURRENT = None
    if dist.is_initialized():
        dist.destroy_process_group()
