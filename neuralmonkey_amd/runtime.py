"""Eager replacement of the TF-1 graph/session machinery.

The reference builds a lazy TF graph through ``@tensor`` properties
(decorators.py:9-27) and evaluates fetches with ``Session.run``
(tf_manager.py:158-185).  Here a ``@tensor`` property returns a ``Fetch``
handle; ``Session.run(fetches, feed_dict)`` evaluates the handles eagerly,
memoised per run, launching libnmhip kernels on the device.  The call shape
(fetch dictionaries in, numpy structures out) is unchanged.
"""
import atexit
import gc
import os
import threading
import weakref
from contextlib import contextmanager
from typing import Any, Dict, List, Optional

import numpy as np
import torch

_REGISTRY: List[Any] = []          # every Parameterized part, in creation order

# HIP-graph captures are thread-local: other threads of the process keep making runtime calls while this
# thread captures -- the RCCL watchdog polls the events of in-flight collectives, the input-pipeline
# worker uploads the next batch on its copy stream -- and under the default ("global") mode any such
# call would invalidate the capture.
CAPTURE_MODE = "thread_local"


def register_part(part) -> None:
    _REGISTRY.append(part)


def registered_parts() -> List[Any]:
    return list(_REGISTRY)


def reset_registry() -> None:
    """Forget all model parts (== tf.reset_default_graph())."""
    _REGISTRY.clear()


class Placeholder:
    """Stands for tf.placeholder: a key of the feed dictionary."""
    __slots__ = ("name", "default")

    def __init__(self, name: str, default: Any = None):
        self.name = name
        self.default = default

    def __repr__(self):
        return f"<Placeholder {self.name}>"


class Fetch:
    """Handle of a value computable inside a run (stands for a tf.Tensor)."""
    __slots__ = ("owner", "fn")

    def __init__(self, owner, fn):
        self.owner, self.fn = owner, fn

    @property
    def key(self):
        return (id(self.owner), self.fn.__name__)

    def __call__(self, ctx: "RunContext"):
        memo = ctx.memo
        k = self.key
        if k not in memo:
            memo[k] = self.fn(self.owner, ctx)
        return memo[k]

    def __repr__(self):
        return f"<Fetch {getattr(self.owner, 'name', self.owner)}.{self.fn.__name__}>"


def tensor(fn):
    """``@tensor def x(self, ctx)`` -> property returning a Fetch handle."""
    return property(lambda self: Fetch(self, fn), doc=fn.__doc__)


class RunContext:
    def __init__(self, session: "Session", feed: Dict[Placeholder, Any]):
        self.session = session
        self.feed = feed
        self.memo: Dict[Any, Any] = {}

    @property
    def store(self):
        return self.session.store

    @property
    def device(self):
        return self.session.device

    def fed(self, placeholder: Placeholder):
        if placeholder in self.feed:
            return self.feed[placeholder]
        if placeholder.default is not None:
            return placeholder.default
        raise KeyError(f"placeholder {placeholder.name} was not fed")

    def is_fed(self, placeholder: Placeholder) -> bool:
        return placeholder in self.feed

    def wants_backward(self, train_mode: bool) -> bool:
        """Must a forward pass keep what its backward pass reads?  Yes in training mode, and whenever a trainer
        runs this context -- the reference's train_op differentiates whatever ``train_mode`` is fed (the
        placeholder only switches dropout, model/model_part.py)."""
        return bool(train_mode) or bool(self.memo.get("want_backward", False))

    def buffer(self, key, shape, dtype=torch.float32, zero=False, zero_init=False):
        """Persistent scratch buffer owned by the session (no per-step malloc).  ``zero``: cleared on every
        request; ``zero_init``: cleared when it is created only (state the kernels themselves keep at zero)."""
        return self.session.buffer(key, shape, dtype, zero, zero_init)

    # -- encoder gradients shared by several decoders ---------------------------------------------------
    def defer_backward(self, encoder, d_states, d_final) -> None:
        """A decoder hands the gradient of an encoder's states / final output over here instead of
        calling ``encoder.backward`` itself: an encoder read by several decoders (multi-task trainers,
        tests/flat-multiattention.ini trains four decoders over two encoders) must run its backward pass
        ONCE, on the sum of what its readers sent (what tf.gradients does with a fan-out node)."""
        if not self.memo.get("backward_deferred", False):
            if hasattr(encoder, "backward"):
                encoder.backward(self, d_states, d_final)
            return
        from . import ops
        pending = self.memo.setdefault("pending_backward", {})
        slot = pending.setdefault(encoder, [None, None, False, False])      # grads, "is my own accumulator"
        for i, g in enumerate((d_states, d_final)):
            if g is None:
                continue
            if slot[i] is None:
                slot[i] = g
                continue
            if not slot[2 + i]:                       # second reader: start a private accumulator
                acc = self.buffer((id(encoder), "deferred_grad", i, tuple(g.shape)), tuple(g.shape))
                ops.ew("copy", slot[i].reshape(-1, g.shape[-1]), None, acc.view(-1, g.shape[-1]))
                slot[i], slot[2 + i] = acc, True
            ops.ew("copy", g.reshape(-1, g.shape[-1]), None, slot[i].view(-1, g.shape[-1]), accumulate=True)

    def flush_backward(self) -> None:
        """Run the deferred encoder backward passes, readers before the encoders they read."""
        pending = self.memo.get("pending_backward", {})
        while pending:
            order = list(pending)
            # an encoder that another pending encoder cross-attends to waits for that one's gradient
            ready = [e for e in order if not any(getattr(o, "input_for_cross_attention", None) is e
                                                 for o in order if o is not e)]
            enc = (ready or order)[0]
            d_states, d_final = pending.pop(enc)[:2]
            if hasattr(enc, "backward"):
                enc.backward(self, d_states, d_final)

    def salt(self, *site) -> int:
        """32-bit salt of a dropout call site: crc32 of the site path.  The kernels add
        global_step * 0x9E3779B9 on the device (``Session.step_tensor``), so every training step draws
        fresh masks -- also when the step is a replayed HIP graph."""
        import zlib
        return zlib.crc32("/".join(str(s) for s in site).encode()) & 0xFFFFFFFF


_PINNED_FETCH: Dict[Any, list] = {}       # pinned staging buffers of _to_host, by (capacity in elements, dtype)


def _map_tensors(val, fn):
    if isinstance(val, torch.Tensor):
        return fn(val)
    if isinstance(val, tuple) and hasattr(val, "_fields"):
        return type(val)(*[_map_tensors(v, fn) for v in val])
    if isinstance(val, (list, tuple)):
        return type(val)(_map_tensors(v, fn) for v in val)
    if isinstance(val, dict):
        return {k: _map_tensors(v, fn) for k, v in val.items()}
    return val


def _to_host(val):
    """Fetched tensors -> NumPy arrays.  All device tensors of a fetch structure (a beam search returns seven) are
    copied into pinned staging buffers asynchronously and the stream is waited for ONCE, instead of one blocking
    pageable copy per tensor (~35 us each)."""
    devs = []
    _map_tensors(val, lambda t: devs.append(t) if t.is_cuda else None)
    if not devs:
        return _map_tensors(val, lambda t: t.detach().cpu().numpy())
    staged = {}
    for t in devs:
        if id(t) in staged or t.numel() == 0:
            continue
        src, shaped = t.detach(), None
        if not src.is_contiguous():
            # a strided view (a column block, a transposed history): its whole storage span travels as ONE plain copy
            # and the view is taken on the host, instead of a gather kernel that packs it on the device first
            span = 1 + sum((n - 1) * st for n, st in zip(src.shape, src.stride()))
            if all(st >= 0 for st in src.stride()) and span <= 4 * src.numel() + 1024:
                shaped = (tuple(src.shape), tuple(src.stride()))
                src = src.as_strided((span,), (1,))
        count = src.numel()
        cap = 1 << max(8, (count - 1).bit_length())              # few distinct buffers for the varying decode lengths
        pool = _PINNED_FETCH.setdefault((cap, t.dtype), [])
        host = pool.pop() if pool else torch.empty(cap, dtype=t.dtype).pin_memory()
        flat = host[:count]
        flat.copy_(src.reshape(-1), non_blocking=True)
        if shaped is not None:
            flat = flat.as_strided(*shaped)
        staged[id(t)] = (host, flat, (cap, t.dtype))
    for device in {t.device for t in devs}:
        torch.cuda.current_stream(device).synchronize()

    def fetch(t):
        if not t.is_cuda:
            return t.detach().cpu().numpy()
        if t.numel() == 0:
            return np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
        return staged[id(t)][1].numpy().reshape(tuple(t.shape)).copy()
    out = _map_tensors(val, fetch)
    for host, _, key in staged.values():
        _PINNED_FETCH[key].append(host)
    return out


class HostPending:
    """A few device words on their way to the host: an asynchronous copy into pinned memory plus an event.
    ``get()`` waits for the event.  A training step hands its loss scalars back like this, so the host thread is
    free to enqueue the next step while the GPU still runs this one (a blocking read-back at the end of every
    step left the GPU idle for ~0.3 ms at the start of the next: `profiles/r03_train_step_timeline.txt`)."""

    def __init__(self, session=None, key=None, host=None, event=None, shape=(), value=None):
        self._session, self._key, self._host, self._event, self._shape, self._value = (
            session, key, host, event, tuple(shape), value)
        self._redirect = None

    def get(self) -> np.ndarray:
        if self._redirect is not None:            # the step was run again (Session.recover_training)
            return self._redirect.get()
        if self._value is None:
            self._event.synchronize()
            self._value = self._host.numpy().copy().reshape(self._shape)
            self._release()
        return self._value

    def redirect(self, other: "HostPending") -> None:
        """The values this copy carries are void (the step that produced them was run again): ``get`` hands out
        ``other``'s from now on."""
        self._redirect = other

    @property
    def session(self):
        return self._session

    def _release(self):
        host, self._host = self._host, None
        if host is not None and self._session is not None:
            # re-use is ordered by the stream: the next copy into this buffer is enqueued behind this one
            self._session._pinned_pool.setdefault(self._key, []).append(host)     # pylint: disable=protected-access

    def __del__(self):
        try:
            self._release()
        except Exception:        # pylint: disable=broad-except
            pass


# A/B switch: the finished-flag copies of the decoding loops in the launch stream (0, default) or on the session's
# copy stream behind an event (1).  Measured (tools/batch_boundary_probe.py, same box): the cross-stream event costs
# more than the ~10 us the in-stream copy holds the launch stream -- greedy 5.69 -> 5.97 ms per batch, beam 17.65 ->
# 17.95, Transformer greedy 32.6 -> 36.4.
FLAG_COPY_STREAM = os.environ.get("NM_FLAG_COPY_STREAM", "0") != "0"

_LIVE_SESSIONS: "weakref.WeakSet" = weakref.WeakSet()


@contextmanager
def _capture(graph):
    """Stream capture with Python's cyclic garbage collector held off.  A collection that happens to run
    inside a capture may finalise objects of models built earlier -- captured graphs with their private
    memory pools, events -- and releasing those calls hipFree / hipGraphExecDestroy, which is illegal on a
    capturing thread and aborts the process from inside a destructor (seen once in ~10 runs of the test
    suite, always under "Garbage-collecting").  Reference-counted frees of tensors are fine: the caching
    allocator defers them."""
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
            yield
    finally:
        if was_enabled:
            gc.enable()


def _release_device_objects() -> None:
    """Interpreter exit: drain the device and drop captured graphs / streams / events while the HIP
    runtime is still up, in a fixed order (module teardown order is arbitrary otherwise)."""
    try:
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
    except Exception:        # pylint: disable=broad-except
        return
    for sess in list(_LIVE_SESSIONS):
        sess.__dict__.get("_step_graphs", {}).clear()
        sess._graphs.clear()                      # pylint: disable=protected-access
        sess._h2d.clear()                         # pylint: disable=protected-access


atexit.register(_release_device_objects)


class Session:
    """One set of variables on one device (stands for a tf.Session)."""

    def __init__(self, device, seed: Optional[int] = None):
        from .variables import VariableStore
        self.device = torch.device(device)
        self.seed = seed                 # also folded into the salts of sampling loops (decoders/autoregressive.py)
        self.store = VariableStore(self.device, seed)
        _LIVE_SESSIONS.add(self)
        self._buffers: Dict[Any, torch.Tensor] = {}
        self._h2d: Dict[Any, Any] = {}
        self._graphs: Dict[Any, Any] = {}
        self.use_graphs = os.environ.get("NM_GRAPHS", "1") != "0"
        self.use_side_stream = os.environ.get("NM_SIDE_STREAM", "1") != "0"
        # whole training steps of taped (general-path) models as one HIP graph per batch shape
        self.use_step_graphs = os.environ.get("NM_STEP_GRAPHS", "1") != "0"
        # GRU time loops as ONE launch each (nm_gru_seq_fwd / nm_gru_seq_bwd: workgroup clusters that keep the recurrent
        # kernels in registers and hand stage outputs over as tagged granules, csrc/nm_gru_cluster.hip) wherever the
        # shape allows (ops.gru_seq_supported); NM_CLUSTER_LOOPS=0: two graph-replayed launches per step everywhere.
        # Per step at 128 rows x 512 units (tools/gru_loop_bench.py): forward 11.1 -> 5.0 us, BPTT 12.1 -> 5.2 us.
        self.use_cluster_loops = os.environ.get("NM_CLUSTER_LOOPS", "1") != "0"
        self._error_word = None
        self._error_pending = None
        self.background_leaves = os.environ.get("NM_LEAF_BACKGROUND", "1") != "0"
        # Measured and left off: greedy batches are unchanged (5.72 vs 5.74 ms), beam batches go from 17.8 to 23 ms --
        # the beam step's kernels need up to 128 KB of LDS, a CU that holds a capped (82 KB) workgroup cannot take
        # them, and a capped launch keeps every CU occupied four times longer (profiles/r04_lookahead_background.txt)
        self.ahead_in_background = os.environ.get("NM_AHEAD_BACKGROUND", "0") != "0"
        self._background = False         # inside _run_ahead: launches (and captures) are in the library's background mode
        self._deferred_side = []
        self._side_streams = {}          # lane -> HIP stream
        self._side_dirty = set()
        self._copy_stream = None
        self._tls = threading.local()
        self.global_step = 0
        # Look-ahead (``run(..., ahead=...)``): the encoder side of the NEXT batch is evaluated on a second stream
        # while this batch decodes.  Every persistent buffer and captured graph belongs to a SLOT; consecutive
        # batches alternate between slot 0 and slot 1, so the batch that is being encoded ahead never touches what
        # the running batch reads.
        self._pending_ahead = None
        self._pinned_pool: Dict[Any, list] = {}      # pinned host buffers of to_host_async, by (numel, dtype)
        self.slot = 0
        self._ahead: list = []                # [(feed signature, slot, memo, event, feed)]
        self._ahead_stream = None

    def to_device(self, array, dtype, tag=None, derive=None):
        """Host array (or ``derive(array)``) -> device tensor.  Arrays that are
        fed again (same object, e.g. a benchmark batch kept by the caller) stay
        resident in HBM and are not re-derived."""
        if isinstance(array, torch.Tensor):
            return array.to(self.device, dtype)
        key = (id(array), tag, dtype)
        hit = self._h2d.get(key)
        if hit is not None and hit[0] is array:
            if len(hit) > 2 and hit[2] is not None and not self.prefetching:
                # uploaded ahead of time on the copy stream (input_pipeline.Prefetcher): order this stream
                # after the copy, once, and tell the allocator the tensor is in use here
                torch.cuda.current_stream(self.device).wait_event(hit[2])
                hit[1].record_stream(torch.cuda.current_stream(self.device))
                self._h2d[key] = (hit[0], hit[1], None)
            return hit[1]
        src = derive(array) if derive is not None else array
        host = torch.as_tensor(np.ascontiguousarray(src))
        event = None
        if self.prefetching and self.device.type == "cuda":
            # pinned staging buffer + asynchronous copy on the session's copy stream: the DMA engine moves the
            # next batch while the compute stream runs the current step
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._copy_stream):
                ten = host.to(dtype).pin_memory().to(self.device, non_blocking=True)
                event = torch.cuda.Event()
                event.record(self._copy_stream)
        else:
            ten = host.to(self.device, dtype)
        if len(self._h2d) > 256:
            for old in list(self._h2d)[:128]:          # oldest half (dicts keep insertion order)
                self._h2d.pop(old, None)
        self._h2d[key] = (array, ten, event)
        return ten

    @property
    def prefetching(self) -> bool:
        """True on a thread that is uploading a FUTURE batch (input_pipeline.Prefetcher)."""
        return getattr(self._tls, "prefetching", False)

    @contextmanager
    def prefetch_scope(self):
        self._tls.prefetching = True
        try:
            yield
        finally:
            self._tls.prefetching = False

    def buffer(self, key, shape, dtype=torch.float32, zero=False, zero_init=False):
        """Persistent scratch tensor.  Keyed by (key, shape, dtype) and never
        re-allocated, so device pointers baked into captured HIP graphs stay valid."""
        shape = tuple(int(s) for s in shape)
        full = (key, shape, dtype) if not self.slot else (("slot", self.slot), key, shape, dtype)
        buf = self._buffers.get(full)
        if buf is None:
            buf = torch.empty(shape, dtype=dtype, device=self.device)
            self._buffers[full] = buf
            zero = zero or zero_init
        if zero:
            from . import ops
            ops.zero(buf)                # (the library's fill: buffers are also created -- and zeroed -- inside captures)
        return buf

    def read_small(self, dev_tensor: torch.Tensor):
        """A few device words (loop flags) -> host NumPy array, through a persistent PINNED buffer: one
        asynchronous copy + one event wait instead of a pageable ``.cpu()`` (staging allocation, blocking copy).
        The decoding loops do this between chunks of steps while the GPU idles, so its latency is step time."""
        if self.device.type != "cuda":
            return dev_tensor.cpu().numpy()
        key = (dev_tensor.numel(), dev_tensor.dtype)
        slot = self.__dict__.setdefault("_pinned_small", {}).get(key)
        if slot is None:
            slot = (torch.empty(dev_tensor.numel(), dtype=dev_tensor.dtype).pin_memory(), torch.cuda.Event())
            self._pinned_small[key] = slot
        host, event = slot
        host.copy_(dev_tensor.reshape(-1), non_blocking=True)
        event.record()
        event.synchronize()
        return host.numpy().reshape(tuple(dev_tensor.shape))

    def decode_chunks(self, total: int, every: int, launch, allfin: torch.Tensor, run_ahead: bool = True):
        """The host side of ``tf.while_loop`` around a decoding body (decoders/autoregressive.py:425-437,554;
        beam_search_decoder.py:330-355,371): ``launch(t0, n)`` enqueues steps t0 .. t0+n-1, every step leaves
        ``allfin[t]`` != 0 when all rows were finished after it.  The loop's criterion lives on the device; what the
        host needs is only WHEN TO STOP ENQUEUEING, so with ``run_ahead`` it reads the flags of chunk i while chunk
        i+1 already runs (a copy enqueued behind chunk i, waited for after chunk i+1 is enqueued): the GPU never
        waits for a flag read-back (7 of them were ~1.1 ms of a 5.8 ms greedy batch,
        profiles/r03_decode_batch_boundary.txt).  Steps past the first all-finished one leave the loop state
        unchanged and append <pad> rows, which the callers crop, so a batch that finishes early costs at most one
        chunk of wasted steps.  Returns (steps of the reference's loop, steps enqueued)."""
        steps, pending = 0, None
        if os.environ.get("NM_DECODE_RUN_AHEAD", "1") == "0":      # A/B switch: read every chunk's own flags
            run_ahead = False
        while steps < total:
            n = min(every, total - steps)
            launch(steps, n)
            steps += n
            self.kick_ahead()                       # the next batch's encoder is launched while this chunk runs
            probe = self.to_host_async(allfin[:steps], off_stream=FLAG_COPY_STREAM)
            if not run_ahead:
                pending, probe = probe, None
            if pending is not None:
                done = np.nonzero(pending.get())[0]
                if done.size:                       # the loop ends after the first all-finished step
                    return int(done[0]) + 1, steps
            pending = probe
        if pending is not None:
            done = np.nonzero(pending.get())[0]
            if done.size:
                return int(done[0]) + 1, steps
        return steps, steps

    # -- errors that only the device can see ------------------------------------------------------------------
    def error_word(self) -> torch.Tensor:
        """One int32 on the device that kernels set (and never clear) when they gave up: today the cluster time loops
        whose hand-offs timed out (ops.gru_seq_fwd / gru_seq_bwd, ``sticky``).  Their results are garbage, so whoever
        hands results to the caller looks at the word first: the trainer reads it with the step's losses
        (GenericTrainer.objective_values), inference polls it one batch late while batches are announced ahead, at once
        otherwise (``poll_device_errors``).  A set word is not the end of the run: ``recover_training`` /
        ``TensorFlowManager.execute`` run the affected work again on the per-step path."""
        if self._error_word is None:
            self._error_word = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._error_word

    def raise_device_error(self) -> None:
        raise RuntimeError("a GRU time loop gave up waiting for a hand-off between workgroups (0.2 s without progress) "
                           "and so did the stepwise fallback: the results of this and later steps are garbage")

    # -- recovery: a cluster loop that gives up costs a slow step, not the run ----------------------------------
    # A cluster loop needs every one of its workgroups resident at once (csrc/nm_gru_cluster.hip); anything that holds
    # CUs for long -- another process on the GPU, a communication kernel, a second loop on another stream -- makes its
    # hand-offs time out.  The launch then raises the session's error word and its results are garbage.  Nothing of
    # that reaches the variables: the optimizer kernels skip their update while the word is set (nm_optim_apply,
    # ``skip_word``), and the host runs the affected steps / batches again on the per-step path, which has no
    # residency requirement, with the cluster loops off for the rest of the session.
    def cluster_failure(self) -> bool:
        """Blocking read of the error word (a few microseconds once the stream is idle)."""
        if self._error_word is None or self.device.type != "cuda":
            return False
        return int(self._error_word.item()) != 0

    def demote_cluster_loops(self) -> None:
        """A time loop gave up: from now on this session steps its loops with two launches per step.  ONE warning;
        raises when the loops were already off (the error word was raised by the fallback path itself)."""
        import warnings
        torch.cuda.synchronize(self.device)
        if not self.use_cluster_loops:
            self.raise_device_error()
        self.use_cluster_loops = False
        self.cluster_demotions = getattr(self, "cluster_demotions", 0) + 1
        warnings.warn("a GRU time loop launched as one cluster kernel gave up waiting for a hand-off between "
                      "workgroups (something else held compute units for 0.2 s): the affected work is run again and "
                      "this session continues with two launches per recurrent step (what NM_CLUSTER_LOOPS=0 selects)")
        if self._error_word is not None:
            self._error_word.zero_()
        self._error_pending = None
        # graphs that were captured with a cluster launch inside must not be replayed; anything evaluated ahead of
        # time may hold a given-up loop's results
        self._graphs.clear()
        self.__dict__.get("_step_graphs", {}).clear()
        self._ahead = []
        self._pending_ahead = None
        torch.cuda.synchronize(self.device)

    def begin_guarded_step(self, trainer, feed, lookback: int = 2) -> None:
        """Called by a trainer before it enqueues anything of a step.  Reads the error flags of the steps that are
        more than ``lookback`` steps old (their copies to the host finished long ago: the host stays at most that
        many steps ahead of the device) and runs everything since a failed one again; then opens the record of the
        new step: its feed and the host-side counters an update advances."""
        guard = self.__dict__.setdefault("_train_guard", [])
        if self.device.type != "cuda":
            return
        self.__dict__.setdefault("_guarded_trainers", weakref.WeakSet()).add(trainer)
        if not getattr(self, "_recovering", False):
            while len(guard) > lookback:
                rec = guard[0]
                if rec["pending"] is not None and float(rec["pending"].get().reshape(-1)[-1]) != 0.0:
                    self.recover_training(0)
                    break
                guard.pop(0)
        guard.append({"trainer": trainer, "feed": dict(feed), "pending": None,
                      "snap": (self.global_step, [(t, t.snapshot_counters(self)) for t in self._guarded_trainers])})

    def settle_training(self) -> None:
        """Read every outstanding error flag of the training steps enqueued so far (blocking) and run the steps since a
        failed one again: afterwards variables, optimizer slots and ``global_step`` are those of clean steps only.
        Called before variables are saved or exported."""
        guard = self.__dict__.get("_train_guard") or []
        if self.device.type != "cuda" or getattr(self, "_recovering", False):
            return
        for i, rec in enumerate(list(guard)):
            if rec["pending"] is not None and float(rec["pending"].get().reshape(-1)[-1]) != 0.0:
                self.recover_training(i)
                self.settle_training()              # (the steps that were run again left records of their own)
                return
        if guard and any(rec["pending"] is None for rec in guard) and self.cluster_failure():
            hits = [i for i, rec in enumerate(guard) if rec["pending"] is None]
            self.recover_training(hits[0])
            self.settle_training()
            return
        del guard[:]

    def attach_guarded_losses(self, trainer, pending) -> None:
        guard = self.__dict__.get("_train_guard") or []
        if guard and guard[-1]["trainer"] is trainer and guard[-1]["pending"] is None:
            guard[-1]["pending"] = pending

    def recover_training(self, index: int = 0, pending=None) -> None:
        """The step of record ``index`` (or the one whose losses ``pending`` carries) ran a time loop that gave up.
        Its update and those of the steps enqueued since were skipped on the device (the error word is sticky), so the
        variables and optimizer slots are those of the moment before that step: switch to the per-step path, put the
        host counters back, run those steps again from their saved feeds and point the losses their callers hold at
        the new values."""
        guard = self.__dict__.setdefault("_train_guard", [])
        if pending is not None:
            hits = [i for i, rec in enumerate(guard) if rec["pending"] is pending]
            if not hits:
                self.raise_device_error()          # too old to run again: nothing was saved for it
            index = hits[0]
        # earlier records may have failed too (the word is sticky, but a step can finish before anybody looks)
        for i in range(index):
            rec = guard[i]
            if rec["pending"] is not None and float(rec["pending"].get().reshape(-1)[-1]) != 0.0:
                index = i
                break
        bad = guard[index:]
        del guard[index:]
        self.demote_cluster_loops()
        step, counters = bad[0]["snap"]
        self.global_step = step
        for trainer, snap in counters:
            trainer.restore_counters(self, snap)
        self._recovering = True
        try:
            for rec in bad:
                out = self.run({"again": rec["trainer"].fetches}, feed_dict=rec["feed"])["again"]
                new = out.get("losses") if isinstance(out, dict) else None
                if rec["pending"] is not None and isinstance(new, HostPending):
                    rec["pending"].redirect(new)
        finally:
            self._recovering = False

    def poll_device_errors(self, last: bool = False) -> None:
        """Kept for callers that drive a session without TensorFlowManager.execute: raise if the error word is set."""
        if self.cluster_failure():
            self.raise_device_error()

    def to_host_async(self, dev_tensor: torch.Tensor, off_stream: bool = False) -> HostPending:
        """Start copying a small device tensor to pinned host memory; the caller reads it with ``get()`` when (if)
        it wants the values.  ``off_stream``: the copy is ordered after the work enqueued so far but runs on the
        session's copy stream, so the launch stream goes straight on to what is enqueued next (a copy between two
        chunk graphs of a decoding loop left the launch stream idle for ~10 us per chunk) -- only for PERSISTENT
        buffers that the following work does not overwrite at the copied positions (the loops' finished flags)."""
        if self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return HostPending(value=dev_tensor.detach().cpu().numpy())
        # (buffers of the two kinds never mix: re-use is ordered by the stream the copies run on)
        key = (dev_tensor.numel(), dev_tensor.dtype, bool(off_stream))
        pool = self._pinned_pool.setdefault(key, [])
        host = pool.pop() if pool else torch.empty(dev_tensor.numel(), dtype=dev_tensor.dtype).pin_memory()
        if off_stream:
            ready = torch.cuda.Event()
            ready.record()
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(ready)
                host.copy_(dev_tensor.detach().reshape(-1), non_blocking=True)
                event = torch.cuda.Event()
                event.record(self._copy_stream)
            return HostPending(self, key, host, event, tuple(dev_tensor.shape))
        host.copy_(dev_tensor.detach().reshape(-1), non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        return HostPending(self, key, host, event, tuple(dev_tensor.shape))

    def staged(self, key, src: torch.Tensor) -> torch.Tensor:
        """Copy a per-batch device tensor into a persistent buffer (same pointer
        every run), so time loops that read it can be replayed as HIP graphs."""
        if self.prefetching:          # a future batch: the persistent buffer still belongs to the running step
            return src
        buf = self.buffer(("staged", key), tuple(src.shape), src.dtype)
        buf.copy_(src)
        return buf

    @contextmanager
    def side(self, lane: int = 0, after=()):
        """Run the enclosed launches on one of the session's secondary HIP streams
        (``lane``), ordered after everything already enqueued on the main stream
        -- and after what the lanes ``after`` hold so far.
        Used for "leaf" work of the backward pass (weight-gradient GEMMs, bias
        column sums) so that it fills the CUs the latency-bound BPTT loops leave
        idle.  Lane 1 is for the one long leaf GEMM of a step (the vocabulary
        projection's weight gradient): on its own lane the short leaf launches of
        lane 0 do not queue behind it.  ``join_side`` orders the main stream after
        every lane."""
        if not self.use_side_stream or self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            yield
            return
        stream = self._side_streams.get(lane)
        if stream is None:
            stream = self._side_streams[lane] = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(main)
        stream.wait_event(ev)
        for other in after:
            if other != lane and other in self._side_dirty:
                done = torch.cuda.Event()
                done.record(self._side_streams[other])
                stream.wait_event(done)
        from . import ops
        was = getattr(ops._TLS, "on_side", False)         # pylint: disable=protected-access
        ops._TLS.on_side = True                           # (ops.colsum: the kernel that leaves the time loops alone)
        try:
            with torch.cuda.stream(stream):
                yield
        finally:
            ops._TLS.on_side = was
        self._side_dirty.add(lane)

    def defer_side(self, fn) -> None:
        """``fn`` (launches that depend on nothing still to come) is to run on side lane 0 beside the NEXT time loop
        the main stream launches -- ``start_deferred_side`` is called by the loops' owners right before they launch,
        and by whoever deferred the work once the loop it hoped for can no longer come."""
        self._deferred_side.append(fn)

    def start_deferred_side(self) -> None:
        if self._deferred_side:
            todo, self._deferred_side = self._deferred_side, []
            with self.side(0):
                for fn in todo:
                    fn()

    def side_active(self) -> bool:
        """Would ``side()`` move launches to another stream right now?"""
        return bool(self.use_side_stream and self.device.type == "cuda"
                    and not torch.cuda.is_current_stream_capturing())

    def leaf_algo(self) -> int:
        """``algo`` of a leaf GEMM enqueued under ``side()``: the residency-capped background kernels when the
        launches really go to another stream (they then run beside the main stream's time loops)."""
        from . import ops
        return ops.GEMM_BACKGROUND if self.side_active() and self.background_leaves else 0

    def join_side(self, lane: Optional[int] = None) -> None:
        """Order the main stream after one lane (or, by default, after every lane) of ``side()`` work."""
        lanes = sorted(self._side_dirty) if lane is None else [lane] if lane in self._side_dirty else []
        for ln in lanes:
            ev = torch.cuda.Event()
            ev.record(self._side_streams[ln])
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._side_dirty.discard(ln)

    def graphed(self, key, fn) -> None:
        """Run ``fn`` (kernel launches on persistent buffers only, no host
        synchronisation).  First call runs eagerly (allocations), second call
        captures a HIP graph, later calls replay it: the launch-bound time loops
        stop paying Python / launch overhead per kernel."""
        if not self.use_graphs or self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            fn()                      # (inside an enclosing capture the launches simply join that graph)
            return
        if self.slot:
            key = (("slot", self.slot), key)
        if self._background:             # a capture keeps the mode it was made in
            key = ("background", key)
        state = self._graphs.get(key)
        if state is None:
            fn()
            self._graphs[key] = 1
        elif state == 1:
            graph = torch.cuda.CUDAGraph()
            with _capture(graph):
                fn()
            self._graphs[key] = graph
            graph.replay()
        else:
            state.replay()

    MAX_STEP_GRAPHS = 8

    def graphed_call(self, key, fn):
        """Like ``graphed`` for a whole training step of the taped path: ``fn`` returns a value made of
        persistent buffers (a TrainResult); it is kept with the graph and handed back on replay, when
        the Python body does not run.  At most ``MAX_STEP_GRAPHS`` shapes are kept (least recently
        used first out): length-bucketed training sees many shapes, each capture pins its buffers."""
        if not self.use_graphs or self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return fn()
        store = self.__dict__.setdefault("_step_graphs", {})
        if self.slot:
            key = (("slot", self.slot), key)
        state = store.pop(key, None)
        if state is None:
            result = fn()
            state = (1, None, result)
        elif state[0] == 1:
            graph = torch.cuda.CUDAGraph()
            with _capture(graph):
                result = fn()
            graph.replay()
            state = (2, graph, result)
        else:
            state[1].replay()
        store[key] = state               # re-inserted last = most recently used
        while len(store) > self.MAX_STEP_GRAPHS:
            torch.cuda.synchronize(self.device)       # never destroy a graph that may still be executing
            store.pop(next(iter(store)))
        return state[2]

    def step_tensor(self) -> torch.Tensor:
        """Device copy of ``global_step`` (int32 [1]): dropout kernels read it to advance their salts,
        so a captured training step draws fresh masks on every replay."""
        if getattr(self, "_step_dev", None) is None:
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._step_dev_value = 0
        if self._step_dev_value != self.global_step:
            from . import ops
            ops.fill(self._step_dev, self.global_step)
            self._step_dev_value = self.global_step
        return self._step_dev

    def _eval(self, fetch, ctx):
        if isinstance(fetch, Fetch):
            return fetch(ctx)
        if isinstance(fetch, dict):
            return {k: self._eval(v, ctx) for k, v in fetch.items()}
        if isinstance(fetch, tuple) and hasattr(fetch, "_fields"):
            return type(fetch)(*[self._eval(v, ctx) for v in fetch])
        if isinstance(fetch, (list, tuple)):
            return type(fetch)(self._eval(v, ctx) for v in fetch)
        return fetch                      # constants / None pass through

    # -- look-ahead ---------------------------------------------------------------------------------------
    @staticmethod
    def _feed_signature(feed) -> frozenset:
        """Which batch a feed dictionary belongs to: the identities of the host arrays it feeds (a batch hands
        out the SAME id arrays every time it is fed, model/sequence.py:cached_index)."""
        return frozenset((ph.name, id(val)) for ph, val in feed.items() if isinstance(val, np.ndarray))

    def _claim_ahead(self, feed) -> Optional[Dict[Any, Any]]:
        """If this feed was evaluated ahead: switch to its buffer slot, order the current stream after the
        look-ahead stream's work and hand back what was computed.  Stale entries (batches that never came) go."""
        if not self._ahead:
            return None
        now = self._feed_signature(feed)
        # (the variables must be the ones the look-ahead saw: torch-side writes bump the version counter of the
        # flat parameter tensor, the optimizer kernels -- raw pointers -- announce themselves, variables_changed)
        current = self.variables_signature()
        live = [entry for entry in self._ahead if entry[5] == current]
        hit = next((entry for entry in reversed(live) if entry[0] and entry[0] <= now), None)
        if hit is None:
            # not this batch's: a second run() for the batch that is being decoded (tf_manager.execute loops until
            # every executable has its result) must not throw away what was just evaluated for the NEXT batch.  An
            # entry whose batch never comes is dropped after two misses.
            self._ahead_misses = getattr(self, "_ahead_misses", 0) + 1
            self._ahead = live[-1:] if self._ahead_misses < 2 else []
            return None
        self._ahead, self._ahead_misses = [], 0
        _, slot, memo, event, _, _ = hit
        self.slot = slot
        torch.cuda.current_stream(self.device).wait_event(event)
        return memo

    def variables_changed(self) -> None:
        """Called by whoever rewrites the variables behind torch's back (optimizer kernels, collectives): anything
        evaluated ahead of time or tabulated from the old values is dropped."""
        self._ahead = []
        self.store.epoch += 1

    def variables_signature(self):
        """Changes whenever the variables may have: torch-side writes (``store[name].copy_``, a restored
        checkpoint, a test poking a row) bump the version counter the views of the flat parameter tensor share,
        kernels that write through raw pointers announce themselves with ``variables_changed``.  Things derived
        from the variables alone (transposed step weights, input tables) are cached under it."""
        theta = self.store.theta
        return (theta.data_ptr(), theta._version, self.store.epoch)     # pylint: disable=protected-access

    def kick_ahead(self) -> None:
        """Start the look-ahead evaluation ``run`` was asked for.  The decoding loops call this right after they
        have enqueued their first chunk of steps: the host then launches the next batch's encoder while the GPU is
        busy instead of in front of the first step."""
        pending, self._pending_ahead = self._pending_ahead, None
        if pending is not None:
            try:
                self._run_ahead(*pending)
            except Exception as exc:        # pylint: disable=broad-except
                # the look-ahead is an optimisation of the NEXT batch; the batch that is being decoded must not
                # fail because of it (the next batch then simply computes its encoder itself -- and reports the
                # error there, if it is one of the model's)
                import warnings
                warnings.warn("look-ahead evaluation dropped: {!r}".format(exc))
                self._ahead = []

    def _run_ahead(self, fetches, feed) -> None:
        """Evaluate ``fetches`` (the encoder side of a FUTURE batch) on the look-ahead stream, into the buffer slot
        the running batch does not use.  Nothing of the running batch is touched: the other slot's buffers were last
        read by the batch before it, whose results the host has already collected."""
        if self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return
        if self._ahead_stream is None:
            self._ahead_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("NM_AHEAD_PRIO", "0")))
        main = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(main)                                   # after whatever is already queued (nothing, normally)
        mine = self.slot
        self.slot = mine ^ 1
        from . import _lib, ops
        tag = ops.set_workspace_tag(("ahead", self.slot))     # scratch of its own (ops.workspace_tag: per thread)
        # The look-ahead work runs under the decoding loop of the running batch.  Left alone, every launch of its
        # encoder time loops (1024 workgroups of 16 waves) fills the chip for a few microseconds and the decoding
        # steps queue behind them: the two loops add up instead of overlapping (4.45 -> 5.4 ms per greedy batch).
        # In background mode they take half a CU each and leave the issue priority to the decoding steps.
        self._background = self.ahead_in_background
        if self._background:
            _lib.check(_lib.load().nm_ctx_set_background(None, 1), "nm_ctx_set_background")
        try:
            self._ahead_stream.wait_event(start)
            with torch.cuda.stream(self._ahead_stream), torch.no_grad():
                ctx = RunContext(self, dict(feed))
                self._eval(fetches, ctx)
                done = torch.cuda.Event()
                done.record(self._ahead_stream)
            self._ahead = [(self._feed_signature(feed), self.slot, ctx.memo, done, feed,
                            self.variables_signature())]          # one batch ahead, never a backlog
        finally:
            if self._background:
                _lib.check(_lib.load().nm_ctx_set_background(None, 0), "nm_ctx_set_background")
            self._background = False
            self.slot = mine
            ops.set_workspace_tag(tag)

    def run(self, fetches, feed_dict: Optional[Dict[Placeholder, Any]] = None, ahead=None):
        """``ahead`` = (fetches, feed_dict) of the NEXT batch: tensors that do not depend on this run (its encoder
        states, attention keys, initial decoder state) are evaluated on a second stream while this run decodes;
        the run that later feeds that batch finds them computed."""
        feed = dict(feed_dict or {})
        claimed = self._claim_ahead(feed)
        if claimed is None:
            if self._ahead:                # a batch evaluated ahead is waiting in its slot: stay out of it
                self.slot = self._ahead[-1][1] ^ 1
            elif ahead is None:            # no look-ahead in play (training steps, plain runs): the default buffers --
                self.slot = 0              # the slot is not sticky, or every shape would be allocated and captured twice
        self._deferred_side = []          # (left over only if a previous run raised)
        self._pending_ahead = ahead       # started by the first decoding loop (kick_ahead) or right after the fetches
        ctx = RunContext(self, feed)
        if claimed:
            ctx.memo.update(claimed)
        with torch.no_grad():
            out = self._eval(fetches, ctx)
        self.kick_ahead()
        # cached steppers (decoders/decoder_general.py) are re-used by the next batch with ITS context: between
        # runs they must not keep this run's memo -- every tensor evaluated for this batch -- alive
        for per_decoder in self.__dict__.get("_fused_steppers", {}).values():
            for _, stepper in per_decoder.values():
                stepper.ctx = None
        return _to_host(out)
