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

Sharded optimizer (default with more than one rank; NM_DP_SHARDED=0:
the replicated update above).  The flat buffers are cut into buckets that follow the
variable layout (``ShardPlan``); every bucket is split evenly over the ranks.  Per step:
reduce-scatter of every bucket's gradient (each rank ends up with the SUM of its slice
only) -> regulariser terms + squared norms of the rank's own chunks, the per-chunk
partial sums all-reduced (3 floats per 64 K elements: every entry has ONE non-zero
contributor, so the sum is exact) -> per-tensor clip + Adam / Adadelta on the rank's
slices only (1/N of the optimizer's 7 streams over the parameters) -> all-gather of the
updated parameter slices.  The update every replica ends up with is the reference's
(trainers/generic_trainer.py:179-195: clip by the norm of the SUMMED gradient of a whole
tensor, one identical step everywhere); replicas are bit-identical by construction -- each
element is updated once, by its owner, and copied.  The optimizer slots are valid on
their owners only (``gather_optimizer_slots`` before a checkpoint).
"""
import os
from typing import Optional

import torch
import torch.distributed as dist

_CURRENT: Optional["DataParallel"] = None


class ShardPlan:
    """Who owns which elements of a store's flat buffers.

    Buckets follow the VARIABLE LAYOUT, never a step's run-time state, so ownership is stable for the life of the
    store (the optimizer slots of an element live on its owner): a variable of ``big`` elements or more is a run of
    buckets of its own (at most ``bucket_elems`` each) -- such a variable can be exchanged on its own, early in the
    backward pass or as rows -- and smaller variables are packed, in layout order, into shared buckets.  A bucket
    [lo, hi) is split into ``world`` slices of q = 4 * floor((hi - lo) / (4 * world)) elements (rank r owns
    [lo + r q, lo + (r + 1) q): what reduce_scatter_tensor / all_gather_into_tensor move in place) and a tail of fewer
    than 4 * world elements that is all-reduced and updated by every rank alike."""

    def __init__(self, store, world: int, bucket_elems: int, big: int) -> None:
        self.world, self.total = world, int(store.total)
        self.buckets = []                  # (lo, hi, q, variable name or None)
        self.of_variable = {}              # name -> indices of the buckets that hold nothing but this variable
        specs = sorted(store.specs.items(), key=lambda kv: kv[1].offset)
        group_lo = None

        def close(lo, hi, name):
            if hi > lo:
                self.buckets.append((lo, hi, ((hi - lo) // (4 * world)) * 4, name))
                if name is not None:
                    self.of_variable.setdefault(name, []).append(len(self.buckets) - 1)

        pos = 0
        for name, spec in specs:
            lo, hi = spec.offset, spec.offset + spec.size
            if spec.size >= big:
                if group_lo is not None:
                    close(group_lo, lo, None)
                    group_lo = None
                elif lo > pos:
                    close(pos, lo, None)           # (padding between variables, if the layout has any)
                for start in range(lo, hi, bucket_elems):
                    close(start, min(hi, start + bucket_elems), name)
            else:
                if group_lo is None:
                    group_lo = pos
                if hi - group_lo >= bucket_elems:
                    close(group_lo, hi, None)
                    group_lo = None
            pos = hi
        if group_lo is not None:
            close(group_lo, self.total, None)
        elif pos < self.total:
            close(pos, self.total, None)
        assert sum(hi - lo for lo, hi, _, _ in self.buckets) == self.total and self.buckets[0][0] == 0
        cuts = set()
        for lo, hi, q, _ in self.buckets:
            cuts.update(lo + r * q for r in range(world + 1))
            cuts.add(hi)
        self.cuts = sorted(cuts)

    def owned(self, rank: int):
        """[lo, hi) ranges of the flat buffers that ``rank`` reduces, updates and publishes."""
        return [(lo + rank * q, lo + (rank + 1) * q) for lo, _, q, _ in self.buckets if q]

    def tails(self):
        """The buckets' indivisible remainders: all-reduced, updated by every rank."""
        return [(lo + self.world * q, hi) for lo, hi, q, _ in self.buckets if lo + self.world * q < hi]

    def owned_elements(self, rank: int) -> int:
        return sum(hi - lo for lo, hi in self.owned(rank)) + sum(hi - lo for lo, hi in self.tails())


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
        # NM_DP_SHARDED=0: every rank all-reduces the whole gradient and applies the whole update (the round-1..5 path)
        self.sharded = os.environ.get("NM_DP_SHARDED", "1") != "0"
        self.big_variable = int(os.environ.get("NM_DP_BIG_VARIABLE", str(1 << 20)))      # elements
        self._plans: dict = {}
        self._reduced_buckets: set = set()     # buckets of this step whose gradient is already reduced / scattered
        self._bucket_events: dict = {}         # bucket -> event of its producer (early, deferred issue)
        self.poison_foreign = False            # tests: after a reduction, what this rank does not own reads NaN
        self.optimizer_ms: list = []
        self.gather_bytes_per_step = 0
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
        self._reduced_buckets, self._bucket_events = set(), {}
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
        if self.sharded_active():
            # whole buckets only: a variable big enough to own its buckets starts its reduce-scatter now (ordered
            # behind its producer), small ones wait for the end of the backward pass with their bucket mates
            plan = self.plan(store)
            for name in names:
                for idx in plan.of_variable.get(name, ()):
                    if idx in self._reduced_buckets or idx in self._bucket_events:
                        raise RuntimeError("bucket {} of {} was already reduced this step".format(idx, name))
                    if self.defer_issue and grad.is_cuda:
                        done = torch.cuda.Event()
                        done.record()
                        self._bucket_events[idx] = done
                    else:
                        self._reduce_bucket(grad, plan, idx)
            return
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

    def exchange_sparse_rows(self, store, name: str, host_ids, gather_rows, scatter_add, negate,
                             skip_pad: bool = True) -> bool:
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
        own_buckets = None
        if self.sharded_active():
            own_buckets = self.plan(store).of_variable.get(name)
            if not own_buckets:                 # shares its buckets with other variables: dense, with them
                return False
            if any(i in self._reduced_buckets or i in self._bucket_events for i in own_buckets):
                raise RuntimeError("gradient of {} was already reduced this step".format(name))
        if any(lo < dhi and dlo < hi for dlo, dhi in self._early):
            raise RuntimeError("gradient span [{}, {}) was already reduced this step".format(lo, hi))
        ids = np.unique(np.asarray(host_ids).reshape(-1))
        if skip_pad:                              # (a sequence whose pad rows are masked out of the gradient)
            ids = ids[ids != 0]
        ids = ids.astype(np.int32)
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
        if own_buckets is not None:
            self._reduced_buckets.update(own_buckets)      # summed on EVERY rank: the owners find their slices complete
        else:
            self._early.append((lo, hi))
        self.sparse_bytes_per_step += 4 * cap * (esz + 1)
        return True

    def all_reduce_gradients(self, store, error_word=None) -> None:
        """In-place sum of the flat gradient buffer over ranks, in large buckets (minus the spans
        ``all_reduce_early`` already started); returns with every collective of the step ordered
        before the current stream.  Sharded optimizer: every bucket is reduce-SCATTERED instead -- afterwards a rank
        holds the sum of the slices it owns (and of the buckets' tails), the rest of its gradient buffer is stale.
        ``error_word``: the session's device error word (int32 [1]): its maximum over ranks travels with the
        gradients, so that a step one rank must run again is skipped -- and run again -- by all of them."""
        if self.world_size == 1 and not self.forced:
            return
        grad = store.ensure_grad()
        if self.sharded_active():
            self._reduce_scatter_gradients(store, grad, error_word)
            return
        if error_word is not None and self.world_size > 1:
            if self._comm is None:
                self._handles.append(dist.all_reduce(error_word, op=dist.ReduceOp.MAX, async_op=True))
            else:
                dist.all_reduce(error_word, op=dist.ReduceOp.MAX)
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

    # -- sharded optimizer ------------------------------------------------------------------------------------
    def sharded_active(self) -> bool:
        """Reduce-scatter -> update of this rank's slices -> all-gather?  (Needs torch.distributed's collectives: the
        library's own communicator only knows the all-reduce.)"""
        return self.sharded and (self.world_size > 1 or self.forced) and self._comm is None

    def plan(self, store) -> ShardPlan:
        key = id(store)
        if key not in self._plans:
            self._plans[key] = ShardPlan(store, self.world_size, self.bucket_elems, self.big_variable)
        return self._plans[key]

    def optimizer_cuts(self, store):
        """Flat offsets at which the optimizer's chunk table must be cut (ops.OptimizerTables(cuts=...)).  The same for
        the sharded and the replicated update: both then add the same partial sums in the same order."""
        return self.plan(store).cuts

    def _in_place(self) -> bool:
        return dist.get_backend() == "nccl"

    def _reduce_bucket(self, grad, plan: ShardPlan, idx: int) -> None:
        lo, hi, q, _ = plan.buckets[idx]
        n, r = self.world_size, self.rank
        if q:
            whole, mine = grad[lo:lo + n * q], grad[lo + r * q:lo + (r + 1) * q]
            if self._in_place():             # RCCL reduces in place when the output is the rank's slice of the input
                self._handles.append(dist.reduce_scatter_tensor(mine, whole, op=dist.ReduceOp.SUM, async_op=True))
            else:
                tmp = torch.empty_like(mine)
                dist.reduce_scatter_tensor(tmp, whole, op=dist.ReduceOp.SUM)
                mine.copy_(tmp)
        if lo + n * q < hi:
            self._handles.append(dist.all_reduce(grad[lo + n * q:hi], op=dist.ReduceOp.SUM, async_op=True))
        self._reduced_buckets.add(idx)

    def _reduce_scatter_gradients(self, store, grad, error_word) -> None:
        plan = self.plan(store)
        if self._bucket_events:
            if self._issue_stream is None and grad.is_cuda:
                self._issue_stream = torch.cuda.Stream(device=grad.device)
            pending, self._bucket_events = self._bucket_events, {}
            with torch.cuda.stream(self._issue_stream):
                if self._loops_done is not None:
                    self._issue_stream.wait_event(self._loops_done)
                for idx, done in sorted(pending.items()):
                    self._issue_stream.wait_event(done)
                    self._reduce_bucket(grad, plan, idx)
        early = len(self._reduced_buckets)
        early_bytes = 4 * sum(plan.buckets[i][1] - plan.buckets[i][0] for i in self._reduced_buckets)
        if error_word is not None and self.world_size > 1:
            self._handles.append(dist.all_reduce(error_word, op=dist.ReduceOp.MAX, async_op=True))
        for idx in range(len(plan.buckets)):
            if idx not in self._reduced_buckets:
                self._reduce_bucket(grad, plan, idx)
        timed = self.timing and grad.is_cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self._wait_handles()
        if timed:
            ev1.record()
            self._timed.append((ev0, ev1))
        if self.poison_foreign:                  # (tests) nothing may read what this rank does not own
            keep = torch.zeros(grad.numel(), dtype=torch.bool, device=grad.device)
            for lo, hi in plan.owned(self.rank) + plan.tails():
                keep[lo:hi] = True
            grad[~keep] = float("nan")
        self.bytes_per_step = 4 * grad.numel()
        self.early_bytes_per_step = early_bytes if early else 0
        self.sparse_bytes_last, self.sparse_bytes_per_step = self.sparse_bytes_per_step, 0
        self._handles, self._early, self._reduced_buckets = [], [], set()

    def _gather_buckets(self, flat, plan: ShardPlan) -> int:
        """In-place all-gather of every bucket's slices of ``flat`` (parameters after the update; optimizer slots before
        a checkpoint).  Returns the bytes a rank receives."""
        n, r = self.world_size, self.rank
        handles, moved = [], 0
        for lo, _, q, _ in plan.buckets:
            if not q:
                continue
            whole, mine = flat[lo:lo + n * q], flat[lo + r * q:lo + (r + 1) * q]
            if self._in_place():
                handles.append(dist.all_gather_into_tensor(whole, mine, async_op=True))
            else:
                tmp = torch.empty_like(whole)
                dist.all_gather_into_tensor(tmp, mine.clone())
                whole.copy_(tmp)
            moved += 4 * (n - 1) * q
        for hnd in handles:
            hnd.wait()
        return moved

    def optimizer_step(self, store, tables, kind, slot0, slot1, l1_weight, l2_weight, clip_norm, params, skip=None):
        """Everything between the backward pass and the next forward pass: gradient exchange, regulariser terms and
        per-tensor norms, clip + update, and -- sharded -- the all-gather of the updated parameters.  ``tables``: the
        optimizer's chunk tables (ops.OptimizerTables built with ``optimizer_cuts``: ``partials`` / ``partial_vector`` /
        ``segments`` / ``apply`` / ``chunk_range``); kind 0 Adam, 1 Adadelta, ``params`` their four scalars; ``skip``:
        the device word that voids the update (its maximum over ranks is taken here).  Returns the device [L1, L2]."""
        grad = store.ensure_grad()
        self.all_reduce_gradients(store, error_word=skip)
        timed = self.timing and grad.is_cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if not self.sharded_active():
            tables.partials(store.theta, grad, l1_weight, l2_weight, (0, tables.nchunk))
            l1l2 = tables.segments()
            tables.apply(kind, store.theta, grad, slot0, slot1, clip_norm, params, skip=skip)
        else:
            plan = self.plan(store)
            ranges = [tables.chunk_range(lo, hi) for lo, hi in plan.owned(self.rank)]
            tails = [tables.chunk_range(lo, hi) for lo, hi in plan.tails()]
            # ONE launch per pass over everything this rank owns (a device list of chunk indices): range by range the
            # 14 launches of the headline model took 2.2 instead of 0.5 ms (profiles/r06_dp_force_kernel_stats_*.csv).
            # The tails are everybody's -- every rank regularises and updates them -- but their partial sums count once:
            # rank 0's (the other ranks multiply theirs by zero).
            from . import ops
            partial = tables.partial_vector()
            ops.zero(partial)
            everything = tables.chunk_list(ranges + tails)
            tables.partials(store.theta, grad, l1_weight, l2_weight, None, chunk_list=everything)
            if self.rank != 0 and tails:         # (pass 1 also adds the regulariser terms to the tails' gradient)
                mask = tables.__dict__.get("_tail_mask")          # (kept on the tables: it lives as long as they do)
                if mask is None:
                    mask = torch.ones_like(partial)
                    for b, e in tails:
                        mask[3 * b:3 * e] = 0.0
                    tables.__dict__["_tail_mask"] = mask
                if partial.is_cuda:
                    ops.ew("mul", partial, mask, partial)
                else:
                    partial.mul_(mask)
            if self.world_size > 1:
                dist.all_reduce(partial, op=dist.ReduceOp.SUM)
            l1l2 = tables.segments()
            tables.apply(kind, store.theta, grad, slot0, slot1, clip_norm, params, skip=skip,
                         chunk_list=everything)
            self.gather_bytes_per_step = self._gather_buckets(store.theta, plan)
            store.epoch += 1                     # (a collective wrote the variables: Session.variables_signature)
        if timed:
            ev1.record()
            self.optimizer_ms.append((ev0, ev1))
        return l1l2

    def gather_optimizer_slots(self, store, slot0, slot1) -> None:
        """Sharded optimizer: a rank holds the slots of its own slices only.  Before they are written to a checkpoint
        (or compared in a test) every rank collects the others'."""
        if self.sharded_active() and self.world_size > 1:
            plan = self.plan(store)
            self._gather_buckets(slot0, plan)
            self._gather_buckets(slot1, plan)

    def exchange_report(self) -> dict:
        """Mean exposed wait per step since the last call (``timing`` on), bytes exchanged per step and how many of
        them were started early, from inside the backward pass."""
        waits = []
        for first, second in self._timed:
            second.synchronize()
            waits.append(first.elapsed_time(second))
        self._timed = []
        opt = []
        for first, second in self.optimizer_ms:
            second.synchronize()
            opt.append(first.elapsed_time(second))
        self.optimizer_ms = []
        sharded = self.sharded_active()
        plan = next(iter(self._plans.values()), None)
        # bytes a rank SENDS per step with ring collectives: reduce-scatter and all-gather move (N-1)/N of the buffer
        # each, an all-reduce both; embedding rows exchanged as rows replace their dense share
        n = max(1, self.world_size)
        dense = self.bytes_per_step
        return {"ranks_seen": self.ranks_seen(),
                "allreduce_exposed_ms": (sum(waits) / len(waits)) if waits else None, "steps": len(waits),
                "bytes": self.bytes_per_step, "early_bytes": self.early_bytes_per_step,
                "sparse_rows_bytes": getattr(self, "sparse_bytes_last", 0),
                "buckets_mb": self.bucket_elems * 4 / 2 ** 20,
                "optimizer": "sharded" if sharded else "replicated",
                "optimizer_ms": (sum(opt) / len(opt)) if opt else None,
                "optimizer_elements_per_rank": plan.owned_elements(self.rank) if (sharded and plan) else None,
                "exchanged_bytes_per_rank": {"gradients": int(dense * (n - 1) / n) * (1 if sharded else 2),
                                             "parameters": self.gather_bytes_per_step if sharded else 0,
                                             "rows": getattr(self, "sparse_bytes_last", 0)}}

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
