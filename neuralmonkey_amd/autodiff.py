"""Reverse-mode tape over libnmhip kernels: the general model path.

The reference gets its gradients from ``tf.gradients`` over the TF graph
(trainers/generic_trainer.py:160-170).  The headline configuration (TF GRU
cells, Bahdanau attention that does not feed the recurrence, no dropout) has a
hand-scheduled forward/backward in ``decoders/decoder.py`` and
``encoders/recurrent.py``.  Every other configuration the reference accepts --
NematusGRU / LSTM cells, conditional GRU, attention on input, dropout, the
output-projection variants, the Transformer -- runs through this tape: a model
part expresses one step with the functions below, each of which launches its
forward kernel(s) and, when the tape is recording, appends a closure that
launches the matching gradient kernels.  ``Tape.backward`` replays the closures
in reverse.  Gradients always *accumulate* into zero-initialised buffers, so
fan-out needs no special casing and column / row views alias their parent.

Buffers come from the session's persistent scratch pool keyed by
(tape key, slot, creation index): the same model on the same shapes re-uses
the same device memory every run.  During inference the tape does not record
and ``rewind(slot)`` re-uses one step's buffers for every step (two slots, so
that the state of step t-1 survives while step t is computed).
"""
import os
from typing import Callable, List, Optional, Sequence

import torch

from . import ops

# layer norm backward with its parameter gradients in one call (nm_layer_norm_bwd_params); NM_LN_BWD_FUSED=0: the row
# pass + two column sums of rounds 1-5
FUSED_LN_BWD = os.environ.get("NM_LN_BWD_FUSED", "1") != "0"
# the weight gradients of one shape launched together when a backward pass ends (Tape.defer_wgrad)
GROUP_WGRADS = os.environ.get("NM_WGRAD_GROUPS", "1") != "0"
# the weight / bias gradients of a taped time loop (one small product per step and kernel) as one chained product per
# kernel when the backward pass ends
CHAIN_WGRADS = os.environ.get("NM_WGRAD_CHAINS", "1") != "0"
# zeroed tape buffers as slices of a few chunks cleared by one fill each (_ZeroArena)
ZERO_ARENA = os.environ.get("NM_TAPE_ZERO_ARENA", "1") != "0"
# the gradient buffer of a sum handed to one of its operands instead of copied (autodiff.add)
ALIAS_ADD_GRADS = os.environ.get("NM_ADD_GRAD_ALIAS", "1") != "0"
# a sum (residual connection) computed by its first reader: a layer norm adds and norms in one pass (autodiff.add)
LAZY_ADD = os.environ.get("NM_LAZY_ADD", "1") != "0"
# a NematusGRUCell step's state product and point-wise part in one launch (nm_nematus_state_step)
FUSED_STATE_STEP = os.environ.get("NM_NEMATUS_STATE_STEP", "1") != "0"
# ... and the step's input product as well when nothing projected it ahead of the loop (nm_nematus_full_step)
FUSED_FULL_STEP = os.environ.get("NM_NEMATUS_FULL_STEP", "1") != "0"


class Var:
    """A tensor on the tape plus (lazily) its gradient."""
    __slots__ = ("_data", "grad", "needs_grad", "fresh", "pending", "is_leaf")

    def __init__(self, data: torch.Tensor, grad: Optional[torch.Tensor] = None, needs_grad: bool = True):
        self._data = data
        self.grad = grad
        self.needs_grad = needs_grad
        self.fresh = False          # ``grad`` was handed out unwritten: the first contribution overwrites it
        self.pending = None         # (a, b): ``data`` is the sum a + b that nobody has computed yet (autodiff.add)
        self.is_leaf = False        # no closure of this tape reads ``grad`` (Tape.leaf / param): contributions may be deferred

    @property
    def data(self) -> torch.Tensor:
        """The tensor.  A pending sum (a residual connection whose first reader may be a layer norm that computes it
        on the way: ``layer_norm``) is computed when somebody else looks first."""
        if self.pending is not None:
            a, b = self.pending
            self.pending = None
            ops.ew("add", a, b, self._data)
        return self._data

    @data.setter
    def data(self, value: torch.Tensor) -> None:
        self._data = value
        self.pending = None

    @property
    def shape(self):
        return self._data.shape


class _ZeroArena:
    """Zero-initialised scratch of one recording tape: the gradient buffers ``Tape.grad`` hands out (and every other
    ``buf(zero=True)``) are slices of a few large chunks that ONE fill per chunk clears when the step's tape is created
    -- the taped general-path model at the headline size asked for 466 zeroed buffers per training step, a 4.5 us fill
    launch each.  Chunks are never moved or freed (their addresses are baked into captured graphs); the order of the
    requests, and with it every address, repeats from step to step."""
    CHUNK = 16 << 20                   # floats

    def __init__(self, device):
        self.device = device
        self.chunks: List[torch.Tensor] = []
        self.used: List[int] = []
        self.cur = 0

    def begin(self) -> None:
        for chunk, used in zip(self.chunks, self.used):
            if used:
                ops.zero(chunk[:used])
        self.cur = 0
        self.pos = 0

    def take(self, shape) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        size = (n + 63) // 64 * 64     # 256-byte granules
        if size > self.CHUNK:
            return None
        while True:
            if self.cur == len(self.chunks):
                self.chunks.append(ops.zero(torch.empty(self.CHUNK, dtype=torch.float32, device=self.device)))
                self.used.append(0)
                self.pos = 0
            if self.pos + size <= self.CHUNK:
                break
            self.cur += 1
            self.pos = 0
        out = self.chunks[self.cur][self.pos:self.pos + n].view(tuple(int(d) for d in shape))
        self.pos += size
        self.used[self.cur] = max(self.used[self.cur], self.pos)
        return out


class Tape:
    def __init__(self, ctx, key, recording: bool = True):
        self.ctx = ctx
        self.key = key
        self.recording = recording
        self._arena = None
        if recording and ZERO_ARENA and getattr(ctx, "device", None) is not None and ctx.device.type == "cuda":
            arenas = ctx.session.__dict__.setdefault("_tape_arenas", {})
            self._arena = arenas.get(key)
            if self._arena is None:
                self._arena = arenas[key] = _ZeroArena(ctx.device)
            self._arena.begin()
        self._ops: List[Callable[[], None]] = []
        self._wgrads = {}
        self._chains = {}
        self._n = 0
        self._slot = 0

    # -- buffers ------------------------------------------------------------------------------
    def rewind(self, slot: int = 0) -> None:
        assert not self.recording, "a recording tape must keep every intermediate"
        self._n = 0
        self._slot = slot

    def buf(self, shape, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
        if zero and self._arena is not None and dtype == torch.float32:
            out = self._arena.take(shape)
            if out is not None:
                return out
        key = ("tape", self.key, self._slot, self._n)
        self._n += 1
        return self.ctx.buffer(key, shape, dtype, zero)

    def new(self, shape) -> Var:
        return Var(self.buf(shape), None, self.recording)

    def leaf(self, data: torch.Tensor, needs_grad: bool = False) -> Var:
        """Wrap an existing tensor (an encoder output, an embedded input)."""
        v = Var(data, None, needs_grad and self.recording)
        v.is_leaf = True
        return v

    def param(self, part, name: str) -> Var:
        """A trainable variable of ``part``; its gradient is its slice of the flat gradient buffer."""
        ctx = self.ctx
        grad = ctx.store.g(part.var_name(name)) if self.recording else None
        return Var(part.var(ctx, name), grad, self.recording)

    def named_param(self, full_name: str) -> Var:
        """A variable addressed by its full store name (e.g. a shared embedding matrix)."""
        store = self.ctx.store
        return Var(store[full_name], store.g(full_name) if self.recording else None, self.recording)

    def grad(self, v: Var) -> Optional[torch.Tensor]:
        if not v.needs_grad:
            return None
        if v.grad is None:
            v.grad = self.buf(tuple(v.shape), zero=True)
        elif v.fresh:
            ops.zero(v.grad)
            v.fresh = False
        return v.grad

    def grad_slot(self, v: Var):
        """(gradient buffer of ``v``, accumulate?) for a backward kernel that can either overwrite or add: the first
        contribution to a fresh buffer OVERWRITES it (accumulate False), which saves the zero-fill ``grad`` would
        have issued -- 195 fills of 13 MB per step of the Transformer-base model, 1.3 of 33 ms.  Buffers that were
        handed out before (``grad``, views, parameter slices of the flat gradient) accumulate as always."""
        if not v.needs_grad:
            return None, False
        if v.grad is None:
            v.grad = self.buf(tuple(v.shape), zero=False)
            return v.grad, False
        if v.fresh:                 # a pre-assigned slice of a shared buffer (linear_multi) that nobody wrote yet
            v.fresh = False
            return v.grad, False
        return v.grad, True

    def view(self, v: Var, fn: Callable[[torch.Tensor], torch.Tensor]) -> Var:
        """A strided view (column / row block) of ``v``; gradient flows by aliasing."""
        g = self.grad(v) if (self.recording and v.needs_grad) else None
        return Var(fn(v.data), None if g is None else fn(g), g is not None)

    def cols(self, v: Var, lo: int, hi: int) -> Var:
        return self.view(v, lambda t: t[:, lo:hi])

    def rows(self, v: Var, lo: int, hi: int) -> Var:
        return self.view(v, lambda t: t[lo:hi])

    # -- recording ----------------------------------------------------------------------------
    def record(self, fn: Callable[[], None]) -> None:
        if self.recording:
            self._ops.append(fn)

    def backward(self) -> None:
        for fn in reversed(self._ops):
            fn()
        self._ops = []
        self.flush_wgrads()

    # -- weight gradients, grouped -------------------------------------------------------------------------
    def defer_wgrad(self, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, trans_a: bool) -> None:
        """``out += op(a) @ b`` is a weight gradient: nothing of this backward pass reads it, its operands (an
        activation and the gradient of a product's output: tape buffers, one per creation index) stay as they are
        until the pass ends.  Alone such a product is deep (K = the rows of the batch) with few output tiles, so it
        splits K and reduces slabs in a second launch; the weight gradients of ONE shape are launched together when
        the pass ends (``flush_wgrads``: nm_gemm_f32_group -- 72 of the 512 x 512 kernels of Transformer-base in one
        grid).  NM_WGRAD_GROUPS=0, shapes the grouped kernel does not take, and lone products: launched at once."""
        ok = (GROUP_WGRADS and a.is_cuda and a.dim() == 2 and b.dim() == 2 and out.dim() == 2
              and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
              and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0 and out.stride(0) % 4 == 0
              and a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0 and out.shape[1] % 4 == 0
              and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0
              and a.shape[0] >= 1024 and out.shape[0] * out.shape[1] <= (1 << 21))
        if not ok:
            # a product of a few rows (one step of a taped time loop): the steps' products for one kernel are ONE
            # product over the chain of their rows when the pass ends (``flush_wgrads``: nm_gemm_f32_chain)
            small = (CHAIN_WGRADS and trans_a and a.is_cuda and a.dim() == 2 and b.dim() == 2 and out.dim() == 2
                     and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
                     and a.shape[0] % 16 == 0 and 16 <= a.shape[0] < 1024
                     and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0 and a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0
                     and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)
            if not small:
                ops.gemm(a, b, out=out, trans_a=trans_a, accumulate=True)
                return
            key = ("chain", out.data_ptr(), tuple(a.shape), tuple(b.shape), a.stride(0), b.stride(0))
            self._chains.setdefault(key, [out, []])[1].append((a, b))
            return
        key = (trans_a, tuple(a.shape), tuple(b.shape), a.stride(0), b.stride(0), out.stride(0))
        self._wgrads.setdefault(key, []).append((a, b, out))

    def defer_bias(self, dy: torch.Tensor, out: torch.Tensor) -> None:
        """``out += column sums of dy``: a bias gradient.  Those of a taped time loop (one per step and bias) are summed
        over the chain of the steps' rows when the pass ends (nm_colsum_chain); others at once."""
        small = (CHAIN_WGRADS and dy.is_cuda and dy.dim() == 2 and dy.stride(1) == 1 and dy.shape[0] < 1024
                 and dy.stride(0) % 4 == 0 and dy.shape[1] % 4 == 0 and dy.data_ptr() % 16 == 0)
        if not small:
            ops.colsum(dy, out, accumulate=True)
            return
        key = ("bias", out.data_ptr(), tuple(dy.shape), dy.stride(0))
        self._chains.setdefault(key, [out, []])[1].append(dy)

    def flush_wgrads(self) -> None:
        chains, self._chains = self._chains, {}
        for key, (out, members) in chains.items():
            if key[0] == "outer":
                for i in range(0, len(members), ops.OUTER_CHAIN_MAX):
                    ops.outer_chain(members[i:i + ops.OUTER_CHAIN_MAX], out, accumulate=True)
            elif key[0] == "bias":
                if len(members) == 1:
                    ops.colsum(members[0], out, accumulate=True)
                else:
                    ops.colsum_chain(members, out, accumulate=True)
            elif len(members) == 1:
                ops.gemm(members[0][0], members[0][1], out=out, trans_a=True, accumulate=True)
            else:
                ops.gemm_chain(members, out, accumulate=True)
        pending, self._wgrads = self._wgrads, {}
        for (trans_a, *_), items in pending.items():
            while items:
                # (a kernel that collects several contributions -- tied / shared weights -- gets them one launch
                # after the other: no two products of a launch write the same output)
                seen, now, later = set(), [], []
                for it in items:
                    (later if it[2].data_ptr() in seen else now).append(it)
                    seen.add(it[2].data_ptr())
                if len(now) == 1:
                    ops.gemm(now[0][0], now[0][1], out=now[0][2], trans_a=trans_a, accumulate=True)
                else:
                    ops.gemm_group(now, trans_a=trans_a, accumulate=True)
                items = later


# ------------------------------------------------------------------------------------------------
# functions (forward kernel now, gradient closure on the tape)
# ------------------------------------------------------------------------------------------------
def linear(tape: Tape, x: Var, w: Var, b: Optional[Var] = None, out: Optional[Var] = None,
           accumulate: bool = False, trans_b: bool = False, act: Optional[str] = None) -> Var:
    """out (+)= x . op(w) + b   -- tf.layers.dense / tf.matmul on MFMA.  ``act`` (inference tapes only): the
    activation in the product's epilogue instead of a launch of its own."""
    n = w.shape[0] if trans_b else w.shape[1]
    if out is None:
        assert not accumulate
        out = tape.new((x.shape[0], n))
    assert act in (None, "relu") or not tape.recording, "of the fused activations only relu has a backward closure"
    assert act is None or not accumulate
    ops.gemm(x.data, w.data, out=out.data, bias=None if b is None else b.data, accumulate=accumulate,
             trans_b=trans_b, act=act)

    def bwd():
        dy = out.grad
        if dy is None:
            return
        if act == "relu":       # relu in the product's epilogue: its gradient from the OUTPUT, in place (dy is dead after)
            ops.ew("relu_bwd", out.data, dy, dy)
        if x.needs_grad:
            gx, acc = tape.grad_slot(x)
            ops.gemm(dy, w.data, out=gx, trans_b=not trans_b, accumulate=acc)
        if w.needs_grad:
            if trans_b:
                tape.defer_wgrad(dy, x.data, tape.grad(w), True)
            else:
                tape.defer_wgrad(x.data, dy, tape.grad(w), True)
        if b is not None and b.needs_grad:
            tape.defer_bias(dy, tape.grad(b))
    tape.record(bwd)
    return out


def lstm_cell(tape: Tape, z: Var, c_prev: Var, forget_bias: float = 1.0):
    """(h', c') of one LSTMCell step from its pre-activations z = [x, h].W + b [R, 4H] (gate order i, j, f, o) in
    ONE launch forward and one backward (nm_lstm_cell_fwd / _bwd) -- was four activation and four element-wise
    launches each way (decoders/decoder.py:309-325, encoders/recurrent.py:21 around tf's LSTMCell)."""
    rows, h = c_prev.shape
    c_new, h_new = tape.new((rows, h)), tape.new((rows, h))
    gates = tape.buf((rows, 4 * h)) if tape.recording else None
    ops.lstm_cell_fwd(z.data, c_prev.data, c_new.data, h_new.data, gates, forget_bias)

    def bwd():
        if h_new.grad is None and c_new.grad is None:
            return
        dz, acc_z = tape.grad_slot(z)
        dcp, acc_c = tape.grad_slot(c_prev)
        if dz is None:
            return
        ops.lstm_cell_bwd(h_new.grad, c_new.grad, gates, c_prev.data, c_new.data, dz, dcp, acc_z, acc_c)
    tape.record(bwd)
    return h_new, c_new


def nematus_cell(tape: Tape, g_pre: Var, sc: Var, ci: Var, h_prev: Var) -> Var:
    """h' of one NematusGRUCell step from its four products (nn/ortho_gru_cell.py:73-105) in ONE launch forward and one
    backward (nm_nematus_cell_fwd / _bwd) -- was sigmoid, mul, tanh, blend and their five backward launches."""
    rows, h = h_prev.shape
    h_new = tape.new((rows, h))
    ru = tape.buf((rows, 2 * h)) if tape.recording else None
    c = tape.buf((rows, h)) if tape.recording else None
    ops.nematus_cell_fwd(g_pre.data, sc.data, ci.data, h_prev.data, h_new.data, ru, c)

    def bwd():
        if h_new.grad is None:
            return
        dg, acc_g = tape.grad_slot(g_pre)
        dci, acc_ci = tape.grad_slot(ci)
        dsc, acc_sc = tape.grad_slot(sc)
        dhp, acc_hp = tape.grad_slot(h_prev)
        if dg is None:
            return
        ops.nematus_cell_bwd(h_new.grad, ru, c, sc.data, h_prev.data, dg, dci, dsc, dhp, acc_g, acc_ci, acc_sc, acc_hp)
    tape.record(bwd)
    return h_new


def nematus_input_projection(tape: Tape, x_all: Var, w_in: torch.Tensor, b_in: Optional[torch.Tensor], params) -> Var:
    """x . [W_g | W_c] + [b_g | b_c] for the inputs of ALL steps of a loop whose inputs are known beforehand (the target
    embeddings of a teacher-forced decoder): one product instead of one per step; the steps read their rows
    (``nematus_cell_merged(..., x_proj=)``) and leave their gradients in the rows of this output's gradient."""
    rows, h3 = x_all.shape[0], w_in.shape[1]
    h = h3 // 3
    out = tape.new((rows, h3))
    ops.gemm(x_all.data, w_in, out=out.data, bias=b_in)

    def bwd():
        d = out.grad
        if d is None:
            return
        if x_all.needs_grad:
            gx, acc = tape.grad_slot(x_all)
            ops.gemm(d, w_in, out=gx, trans_b=True, accumulate=acc)
        for key, lo, hi in (("gi", 0, 2 * h), ("ci", 2 * h, 3 * h)):
            w, b = params[key]
            if w.needs_grad:
                tape.defer_wgrad(x_all.data, d[:, lo:hi], tape.grad(w), True)
            if b is not None and b.needs_grad:
                tape.defer_bias(d[:, lo:hi], tape.grad(b))
    tape.record(bwd)
    return out


def nematus_cell_merged(tape: Tape, x: Optional[Var], h_prev: Var, w_in: torch.Tensor, b_in: Optional[torch.Tensor],
                        w_st: torch.Tensor, b_st: Optional[torch.Tensor], params, x_proj: Optional[Var] = None,
                        out: Optional[Var] = None) -> Var:
    """One NematusGRUCell step as TWO products and one point-wise launch each way: x . [W_g | W_c] and h . [U_g | U_c]
    against column-concatenated copies of the four kernels (``w_in`` [D, 3H], ``w_st`` [H, 3H]: nn/cells.py refreshes them
    once per step), then nm_nematus_cell_fwd on the halves.  ``params``: the Vars of the four kernels and four biases
    ({"gi", "ci", "gs", "cs"} -> (kernel, bias or None)) -- their gradients are products against column slices of the
    two gradient buffers, chained over the steps when the pass ends (Tape.defer_wgrad)."""
    rows, h = h_prev.shape
    s_all = tape.buf((rows, 3 * h))
    h_new = out if out is not None else tape.new((rows, h))      # (``out``: the step's rows of a buffer of all steps)
    # the state product and the point-wise part in one launch where that pays (ops.nematus_state_step_ok)
    fused = FUSED_STATE_STEP and ops.nematus_state_step_ok(h_prev.data, w_st, h_new.data)
    if not fused:
        ops.gemm(h_prev.data, w_st, out=s_all, bias=b_st)
    # ... and the input half too when it was not projected ahead of the loop (a conditional decoder's second cell)
    full = fused and FUSED_FULL_STEP and x_proj is None and ops.nematus_full_step_ok(x.data, w_in)
    if x_proj is not None:            # the input half was projected for all steps at once (nematus_input_projection)
        x_all = x_proj.data
    elif not full:
        x_all = tape.buf((rows, 3 * h))
        ops.gemm(x.data, w_in, out=x_all, bias=b_in)
    ru = tape.buf((rows, 2 * h)) if tape.recording else None
    c = tape.buf((rows, h)) if tape.recording else None
    keep_sc = s_all[:, 2 * h:] if tape.recording else None      # (of s_all only the candidate's columns are kept: the
    if full:                                                    #  backward pass reads them)
        ops.nematus_full_step(h_prev.data, w_st, b_st, x.data, w_in, b_in, h_new.data, ru, c, keep_sc)
    elif fused:
        ops.nematus_state_step(h_prev.data, w_st, b_st, x_all, h_new.data, ru, c, keep_sc)
    else:
        ops.nematus_cell_fwd(s_all[:, :2 * h], s_all[:, 2 * h:], x_all[:, 2 * h:], h_prev.data, h_new.data, ru, c,
                             g2=x_all[:, :2 * h])

    def bwd():
        if h_new.grad is None:
            return
        d_st = tape.buf((rows, 3 * h))
        d_in = tape.buf((rows, 3 * h)) if x_proj is None else tape.grad(x_proj)       # (rows of the projection's gradient)
        dhp, acc_hp = tape.grad_slot(h_prev)
        ops.nematus_cell_bwd(h_new.grad, ru, c, s_all[:, 2 * h:], h_prev.data, d_st[:, :2 * h], d_in[:, 2 * h:],
                             d_st[:, 2 * h:], dhp, False, False, False, acc_hp, dg2=d_in[:, :2 * h])
        if dhp is not None:
            ops.gemm(d_st, w_st, out=dhp, trans_b=True, accumulate=True)
        blocks = [("gs", d_st, h_prev, 0, 2 * h), ("cs", d_st, h_prev, 2 * h, 3 * h)]
        if x_proj is None:
            if x.needs_grad:
                gx, acc = tape.grad_slot(x)
                ops.gemm(d_in, w_in, out=gx, trans_b=True, accumulate=acc)
            blocks += [("gi", d_in, x, 0, 2 * h), ("ci", d_in, x, 2 * h, 3 * h)]
        for key, src, act, lo, hi in blocks:
            w, b = params[key]
            if w.needs_grad:
                tape.defer_wgrad(act.data, src[:, lo:hi], tape.grad(w), True)
            if b is not None and b.needs_grad:
                tape.defer_bias(src[:, lo:hi], tape.grad(b))
    tape.record(bwd)
    return h_new


def linear_multi(tape: Tape, x: Var, ws: List[Var]) -> List[Var]:
    """``[x . w for w in ws]`` -- the query / key / value projections of one attention block read the same rows
    (attention/scaled_dot_product.py:170-176) -- as ONE batched product when the kernels lie back to back in the flat
    parameter buffer: grid z = len(ws) fills the chip with 128x128 tiles (600 workgroups for three 6400 x 512 x 512
    products) where each product alone runs on 64x64 tiles at 65-73 TF.  The outputs (and their gradients) are the
    slices of one [n, rows, N] buffer.  Backward: the weight gradients stay separate products (they split K over
    workgroups, which the batched launch cannot), the input gradient is the sum over the projections as before.
    Falls back to separate products when the kernels are not adjacent."""
    n = len(ws)
    w0 = ws[0].data
    adjacent = n > 1 and all(
        w.data.shape == w0.shape and w.data.is_contiguous()
        and w.data.data_ptr() == w0.data_ptr() + i * w0.numel() * w0.element_size() for i, w in enumerate(ws))
    if not adjacent:
        return [linear(tape, x, w) for w in ws]
    rows, k = x.shape
    nout = w0.shape[1]
    w3 = torch.as_strided(w0, (n, k, nout), (k * nout, nout, 1))
    out3 = tape.buf((n, rows, nout))
    ops.gemm(x.data.unsqueeze(0).expand(n, rows, k), w3, out=out3)
    outs = [Var(out3[i], None, tape.recording) for i in range(n)]
    if tape.recording:
        g3 = tape.buf((n, rows, nout))
        for i, o in enumerate(outs):
            o.grad, o.fresh = g3[i], True

    def bwd():
        for o, w in zip(outs, ws):
            if o.fresh:                      # nobody consumed this projection
                continue
            if x.needs_grad:
                gx, acc = tape.grad_slot(x)
                ops.gemm(o.grad, w.data, out=gx, trans_b=True, accumulate=acc)
            if w.needs_grad:
                tape.defer_wgrad(x.data, o.grad, tape.grad(w), True)
    tape.record(bwd)
    return outs


def _unary(tape: Tape, op: str, bwd_op: Optional[str], x: Var, alpha: float = 0.0,
           out: Optional[Var] = None) -> Var:
    if out is None:
        out = tape.new(tuple(x.shape))
    ops.ew(op, x.data, None, out.data, alpha=alpha)

    def bwd():
        if out.grad is None or not x.needs_grad:
            return
        gx, acc = tape.grad_slot(x)
        if bwd_op is None:           # copy / scale
            ops.ew("scale" if op == "scale" else "copy", out.grad, None, gx, alpha=alpha, accumulate=acc)
        else:
            ops.ew(bwd_op, out.data, out.grad, gx, accumulate=acc)
    tape.record(bwd)
    return out


def sigmoid(tape: Tape, x: Var, shift: float = 0.0) -> Var:
    """sigmoid(x + shift) (shift = the LSTM forget bias)."""
    return _unary(tape, "sigmoid", "sigmoid_bwd", x, alpha=shift)


def tanh(tape: Tape, x: Var) -> Var:
    return _unary(tape, "tanh", "tanh_bwd", x)


def relu(tape: Tape, x: Var) -> Var:
    return _unary(tape, "relu", "relu_bwd", x)


def scale(tape: Tape, x: Var, alpha: float) -> Var:
    return _unary(tape, "scale", None, x, alpha=alpha)


def copy(tape: Tape, x: Var, out: Optional[Var] = None) -> Var:
    return _unary(tape, "copy", None, x, out=out)


ACTIVATIONS = {"tanh": tanh, "relu": relu, "sigmoid": sigmoid, "identity": lambda tape, x: x}


def add(tape: Tape, a: Var, b: Var) -> Var:
    out = tape.new(tuple(a.shape))
    if LAZY_ADD and tape.recording and a.data.is_cuda and a.data.dim() == 2 and a.shape == b.shape:
        out.pending = (a.data, b.data)      # computed by whoever reads it first: a layer norm does it on the way
    else:
        ops.ew("add", a.data, b.data, out.data)

    def bwd():
        if out.grad is None:
            return
        # the sum's own gradient buffer is dead once this closure has run: the first operand that has no gradient
        # buffer yet simply takes it over (whatever else flows into that operand is added to it in place), the other
        # gets a copy -- one launch per residual connection instead of two
        handed = a is b or not ALIAS_ADD_GRADS
        for v in (a, b):
            if v.needs_grad:
                if not handed and v.grad is None and out.grad.is_contiguous():
                    v.grad, v.fresh, handed = out.grad, False, True
                    continue
                gv, acc = tape.grad_slot(v)
                ops.ew("copy", out.grad, None, gv, accumulate=acc)
    tape.record(bwd)
    return out


def add_(tape: Tape, acc: Var, x: Var) -> Var:
    """acc += x in place (a sum needs none of its inputs in the backward pass)."""
    ops.ew("copy", x.data, None, acc.data, accumulate=True)

    def bwd():
        if acc.grad is not None and x.needs_grad:
            ops.ew("copy", acc.grad, None, tape.grad(x), accumulate=True)
    tape.record(bwd)
    return acc


def mul(tape: Tape, a: Var, b: Var) -> Var:
    out = tape.new(tuple(a.shape))
    ops.ew("mul", a.data, b.data, out.data)

    def bwd():
        if out.grad is None:
            return
        if a.needs_grad:
            ops.ew("mul", out.grad, b.data, tape.grad(a), accumulate=True)
        if b.needs_grad:
            ops.ew("mul", out.grad, a.data, tape.grad(b), accumulate=True)
    tape.record(bwd)
    return out


def div(tape: Tape, a: Var, b: Var) -> Var:
    """a / b element-wise."""
    out = tape.new(tuple(a.shape))
    ops.ew("div", a.data, b.data, out.data)

    def bwd():
        if out.grad is None:
            return
        if a.needs_grad:
            ops.ew("div", out.grad, b.data, tape.grad(a), accumulate=True)
        if b.needs_grad:                                   # d(a/b)/db = -(a/b)/b
            tmp = tape.buf(tuple(a.shape))
            ops.ew("mul", out.grad, out.data, tmp)
            ops.ew("div", tmp, b.data, tmp)
            ops.ew("scale", tmp, None, tape.grad(b), alpha=-1.0, accumulate=True)
    tape.record(bwd)
    return out


def add_scalar(tape: Tape, x: Var, alpha: float) -> Var:
    return _unary(tape, "add_scalar", None, x, alpha=alpha)


def blend(tape: Tape, u: Var, h: Var, c: Var) -> Var:
    """u*h + (1-u)*c."""
    out = tape.new(tuple(h.shape))
    ops.blend_fwd(u.data, h.data, c.data, out.data)

    def bwd():
        if out.grad is None:
            return
        ops.blend_bwd(out.grad, u.data, h.data, c.data, tape.grad(u), tape.grad(h), tape.grad(c))
    tape.record(bwd)
    return out


def dropout(tape: Tape, x: Var, keep_prob: float, train_mode: bool, salt: int) -> Var:
    """nn/utils.py:6-22.  Identity at keep_prob 1 or outside training."""
    if keep_prob <= 0.0 or keep_prob > 1.0:
        raise ValueError("keep_prob must be a scalar tensor or a float in the range (0, 1], got {}"
                         .format(keep_prob))
    if keep_prob == 1.0 or not train_mode:
        return x
    out = tape.new(tuple(x.shape))
    step = tape.ctx.session.step_tensor()          # device-side global step: fresh masks on graph replays
    ops.dropout(x.data, out.data, keep_prob, salt, step=step)

    def bwd():
        if out.grad is not None and x.needs_grad:
            gx, acc = tape.grad_slot(x)
            ops.dropout(out.grad, gx, keep_prob, salt, accumulate=acc, step=step)
    tape.record(bwd)
    return out


def concat(tape: Tape, parts: Sequence[Var]) -> Var:
    """tf.concat(parts, 1)."""
    if len(parts) == 1:
        return parts[0]
    rows = parts[0].shape[0]
    widths = [p.shape[1] for p in parts]
    out = tape.new((rows, sum(widths)))
    col = 0
    for p, w in zip(parts, widths):
        ops.ew("copy", p.data, None, out.data[:, col:col + w])
        col += w

    def bwd():
        if out.grad is None:
            return
        c = 0
        for p, w in zip(parts, widths):
            if p.needs_grad:
                ops.ew("copy", out.grad[:, c:c + w], None, tape.grad(p), accumulate=True)
            c += w
    tape.record(bwd)
    return out


def embedding(tape: Tape, table: Var, ids: torch.Tensor, out: Optional[Var] = None, mask_pad: bool = False,
              scale_by: float = 1.0) -> Var:
    """tf.nn.embedding_lookup (model/sequence.py:170-194, autoregressive.py:258-272)."""
    if out is None:
        out = tape.new((ids.numel(), table.shape[1]))
    ops.embedding_gather(table.data, ids, out=out.data, mask_pad=mask_pad, scale=scale_by)

    def bwd():
        if out.grad is None or not table.needs_grad:
            return
        d = out.grad
        if scale_by != 1.0:
            d = ops.ew("scale", out.grad, None, tape.buf(tuple(out.grad.shape)), alpha=scale_by)
        ops.embedding_scatter_add(table.grad, ids, d, skip_pad=mask_pad)
    tape.record(bwd)
    return out


def layer_norm(tape: Tape, x: Var, gamma: Var, beta: Var, eps: float = 1e-6) -> Var:
    """tf_utils.py:189-219."""
    rows = x._data.numel() // x.shape[-1]
    out = tape.new(tuple(x.shape))
    mean, rstd = tape.buf((rows,)), tape.buf((rows,))
    pend = x.pending
    if pend is not None and ops.add_layer_norm_stats_ok(pend[0], pend[1], gamma.data, beta.data):
        # x is a residual sum nobody has read yet: sum and norm in one pass over the rows
        x.pending = None
        ops.add_layer_norm_stats_fwd(pend[0], pend[1], gamma.data, beta.data, x._data, out.data, mean, rstd, eps)
    else:
        ops.layer_norm_fwd(x.data, gamma.data, beta.data, out=out.data, mean=mean, rstd=rstd, eps=eps)

    def bwd():
        if out.grad is None:
            return
        d = x.shape[-1]
        gx, acc = tape.grad_slot(x)
        # a fresh gradient buffer takes dx directly (no scratch + copy); otherwise dx is added to what is there
        fused = (FUSED_LN_BWD and gamma.needs_grad and d <= 2048 and d % 4 == 0 and out.grad.is_contiguous()
                 and x.data.is_contiguous() and out.grad.data_ptr() % 16 == 0 and x.data.data_ptr() % 16 == 0
                 and gamma.data.data_ptr() % 16 == 0)
        if fused and gx is not None and gx.is_contiguous() and gx.data_ptr() % 16 == 0:
            # dx, dgamma and dbeta in one call: no [rows, D] buffer of dy * xhat, no separate column sums; dx written
            # to -- or, when a residual connection has left its gradient there already, added to -- x's gradient
            ops.layer_norm_bwd_params(out.grad, x.data, mean, rstd, gamma.data, gx, gamma.grad, beta.grad,
                                      accumulate_dx=acc)
            return
        dx = gx if (gx is not None and not acc and gx.is_contiguous()) else tape.buf(tuple(x.shape))
        if fused and dx.data_ptr() % 16 == 0:
            ops.layer_norm_bwd_params(out.grad, x.data, mean, rstd, gamma.data, dx, gamma.grad, beta.grad)
        else:
            dyx = tape.buf(tuple(x.shape))
            ops.layer_norm_bwd(out.grad, x.data, mean, rstd, gamma.data, dx, dyx)
            if gamma.needs_grad:
                ops.colsum(dyx.view(rows, d), gamma.grad, accumulate=True)
                ops.colsum(out.grad.view(rows, d), beta.grad, accumulate=True)
        if gx is not None and dx is not gx:
            ops.ew("copy", dx, None, gx, accumulate=True)
    tape.record(bwd)
    return out


def add_layer_norm(tape: Tape, a: Var, x: Var, gamma: Var, beta: Var, eps: float = 1e-6):
    """(a + x, layer_norm(a + x)): a residual connection and the next sub-layer's pre-norm.  One launch on an
    inference tape (nm_add_layer_norm_fwd); ``add`` + ``layer_norm`` when the tape records."""
    if tape.recording or not (a.data.is_contiguous() and x.data.is_contiguous()):
        total = add(tape, a, x)
        return total, layer_norm(tape, total, gamma, beta, eps)
    total, normed = tape.new(tuple(x.shape)), tape.new(tuple(x.shape))
    ops.add_layer_norm_fwd(a.data, x.data, gamma.data, beta.data, total.data, normed.data, eps)
    return total, normed


def rnn_select(tape: Tape, h_new: Var, h_prev: Var, lengths: Optional[torch.Tensor], t: int,
               y_out: Optional[Var]) -> Var:
    """dynamic_rnn length masking of step t: returns the carried state, writes the emitted row."""
    h_out = tape.new(tuple(h_new.shape))
    ops.rnn_select_fwd(h_new.data, h_prev.data, lengths, t, h_out.data, None if y_out is None else y_out.data)

    def bwd():
        dh = h_out.grad
        dy = None if y_out is None else y_out.grad
        if dh is None and dy is None:
            return
        ops.rnn_select_bwd(dh, dy, lengths, t, tape.grad(h_new), tape.grad(h_prev) if dh is not None else None)
    tape.record(bwd)
    return h_out


def reverse_sequence(tape: Tape, x: Var, lengths: torch.Tensor) -> Var:
    out = tape.new(tuple(x.shape))
    ops.reverse_sequence(x.data, out.data, lengths)

    def bwd():
        if out.grad is not None and x.needs_grad:
            ops.reverse_sequence(out.grad, tape.grad(x), lengths, accumulate=True)
    tape.record(bwd)
    return out


def maxout(tape: Tape, x: Var, pool: int = 2) -> Var:
    rows, cols = x.shape
    out = tape.new((rows, cols // pool))
    arg = tape.buf((rows, cols // pool), torch.int32)
    ops.maxout_fwd(x.data, out.data, arg, pool)

    def bwd():
        if out.grad is not None and x.needs_grad:
            ops.maxout_bwd(out.grad, arg, tape.grad(x), pool)
    tape.record(bwd)
    return out


def sdp_attention(tape: Tape, q: Var, k: Var, v: Var, key_mask: Optional[torch.Tensor], heads: int, bq: int,
                  tq: int, bk: int, tk: int, causal: bool = False, keep_prob: float = 1.0, salt: int = 0,
                  k_data: Optional[torch.Tensor] = None, v_data: Optional[torch.Tensor] = None,
                  w_out: Optional[torch.Tensor] = None, ancestors: Optional[torch.Tensor] = None) -> Var:
    """Multi-head scaled dot-product attention (attention/scaled_dot_product.py:98-226) on
    [B*T, D] rows.  ``k_data`` / ``v_data`` override the key / value storage (a [R,Tmax,D] cache
    view during decoding); gradients are defined for bq == bk.  ``w_out`` [bq, heads, tq, tk] receives
    the (post-dropout) weights, e.g. a decoder's attention history."""
    d = q.shape[1]
    out = tape.new((bq * tq, d))
    w = w_out if w_out is not None else (tape.buf((bq, heads, tq, tk)) if tape.recording else None)
    k3 = k_data if k_data is not None else k.data.view(bk, tk, d)
    v3 = v_data if v_data is not None else v.data.view(bk, tk, d)
    step = tape.ctx.session.step_tensor() if keep_prob < 1.0 else None
    if ancestors is not None:            # a decoding step whose cache rows are addressed through an ancestor table
        assert tq == 1 and bq == bk and keep_prob >= 1.0 and not tape.recording
        ops.sdp_attn_step(q.data.view(bq, 1, d), k3, v3, key_mask, heads, ancestors, out.data.view(bq, 1, d), w)
        return out
    ops.sdp_attn_fwd(q.data.view(bq, tq, d), k3, v3, key_mask, heads, out.data.view(bq, tq, d), w, causal,
                     bq // bk, keep_prob, salt, step)

    def bwd():
        if out.grad is None:
            return
        assert bq == bk and k_data is None and v_data is None
        de = tape.buf((bq, heads, tq, tk))
        (gq, aq), (gk, ak), (gv, av) = tape.grad_slot(q), tape.grad_slot(k), tape.grad_slot(v)
        if not (aq == ak == av):         # one accumulate flag for the three outputs: zero whichever buffer is fresh
            for g, a in ((gq, aq), (gk, ak), (gv, av)):
                if not a:
                    ops.zero(g)
            aq = True
        ops.sdp_attn_bwd(q.data.view(bq, tq, d), k3, v3, key_mask, w, out.grad.view(bq, tq, d), heads,
                         gq.view(bq, tq, d), gk.view(bk, tk, d), gv.view(bk, tk, d),
                         de, causal, keep_prob, salt, accumulate=aq, step=step)
    tape.record(bwd)
    return out


def rowscale(tape: Tape, x: Var, s: Var, out: Optional[Var] = None, accumulate: bool = False) -> Var:
    """out (+)= x[r,:] * s[r,0]: one attention weight per row times that row's vector."""
    if out is None:
        assert not accumulate
        out = tape.new(tuple(x.shape))
    ops.ew("rowscale", x.data, s.data, out.data, accumulate=accumulate)

    def bwd():
        if out.grad is None:
            return
        if x.needs_grad:
            ops.ew("rowscale", out.grad, s.data, tape.grad(x), accumulate=True)
        if s.needs_grad:                      # ds[r] = <dout[r,:], x[r,:]> as a [R,A].[A,1] product of the products
            prod = ops.ew("mul", out.grad, x.data, tape.buf(tuple(x.shape)))
            ones = tape.buf((x.shape[1], 1))
            ops.fill(ones, 1.0)
            ops.gemm(prod, ones, out=tape.grad(s), accumulate=True)
    tape.record(bwd)
    return out


def attn_energies(tape: Tape, y: Var, hf: Var, v: Var, bsz: int, slen: int, rows_per_key: int = 1) -> Var:
    """e[r,s] = sum_a v[a] * tanh(y[r,a] + hf[r // k, s, a])  -- the Bahdanau energies alone, for
    attentions that assemble their distribution from several sources (combination.py:262-266).
    Forward: the fused step kernel with its context output discarded; backward: nm_attn_energy_bwd."""
    ctx = tape.ctx
    rows, a = y.shape
    out = tape.new((rows, slen))
    hf3 = hf.data.view(bsz, slen, a)
    ws = ctx.buffer(("attn_energies_ws", rows, slen, a), ((ops._lib.load().nm_attn_workspace_bytes(rows, slen, a)
                                                            + 3) // 4,), zero_init=True)
    scratch_ctx = ctx.buffer(("attn_energies_ctx", rows, a), (rows, a))
    ops.attn_fwd(y.data, hf3, hf3, None, v.data, None, rows_per_key, scratch_ctx, None, ws, out.data)

    def bwd():
        if out.grad is None:
            return
        assert rows == bsz, "the attention gradient is defined for one query per sentence"
        dhf, dvp, dy = tape.buf((bsz * slen, a)), tape.buf((bsz * slen, a)), tape.buf((rows, a))
        ops.attn_energy_bwd(out.grad.view(1, bsz, slen), hf3, y.data.view(1, bsz, a), v.data, dhf.view(bsz, slen, a),
                            dvp, dy.view(1, bsz, a))
        if hf.needs_grad:
            ops.ew("copy", dhf, None, tape.grad(hf), accumulate=True)
        if y.needs_grad:
            ops.ew("copy", dy, None, tape.grad(y), accumulate=True)
        if v.needs_grad:
            ops.colsum(dvp, v.grad.view(-1), accumulate=True)
    tape.record(bwd)
    return out


def attn_softmax(tape: Tape, e: Var, mask: Optional[torch.Tensor], bsz: int, rows_per_key: int = 1,
                 w_out: Optional[torch.Tensor] = None) -> Var:
    """softmax over all positions, mask, renormalise with +1e-8 (feed_forward.py:139-144,
    combination.py:301-307); plain softmax when ``mask`` is None."""
    w = Var(w_out, None, tape.recording) if w_out is not None else tape.new(tuple(e.shape))
    ops.attn_softmax_fwd(e.data, mask, w.data, bsz, rows_per_key)

    def bwd():
        if w.grad is None or not e.needs_grad:
            return
        assert e.shape[0] == bsz
        de = tape.buf(tuple(e.shape))
        ops.attn_softmax_bwd(w.grad, e.data, mask, de, bsz)
        ops.ew("copy", de, None, tape.grad(e), accumulate=True)
    tape.record(bwd)
    return w


def weighted_sum(tape: Tape, w: Var, vals: Var, bsz: int, slen: int, rows_per_key: int = 1) -> Var:
    """ctx[r,:] = sum_s w[r,s] * vals[r // k, s, :] over the first ``slen`` columns of ``w``."""
    rows, width = w.shape
    a = vals.shape[1]
    k = rows_per_key
    out = tape.new((rows, a))
    w3 = w.data.view(bsz, k, width)[:, :, :slen]
    v3 = vals.data.view(bsz, slen, a)
    ops.gemm(w3, v3, out=out.data.view(bsz, k, a))

    def bwd():
        if out.grad is None:
            return
        d3 = out.grad.view(bsz, k, a)
        if w.needs_grad:
            ops.gemm(d3, v3, out=tape.grad(w).view(bsz, k, width)[:, :, :slen], trans_b=True, accumulate=True)
        if vals.needs_grad:
            gv = tape.grad(vals).view(bsz, slen, a)
            if (CHAIN_WGRADS and vals.is_leaf and k == 1 and w.data.is_cuda and gv.is_contiguous()
                    and slen * ops.OUTER_CHAIN_MAX * 4 <= 65536 and bsz < 65536):
                # one query per sentence: a rank-1 update per step -- the steps of a taped loop are summed in ONE launch
                # when the pass ends (flush_wgrads: nm_outer_chain), their K dimension.  (Only for leaves: nothing on
                # this tape reads their gradient before the pass is over.)
                key = ("outer", gv.data_ptr(), bsz, slen, a, w.data.stride(0), out.grad.stride(0))
                tape._chains.setdefault(key, [gv, []])[1].append((w.data, out.grad))     # pylint: disable=protected-access
            else:
                ops.gemm(w3, d3, out=gv, trans_a=True, accumulate=True)
    tape.record(bwd)
    return out


def add_position(tape: Tape, x: Var, signal: torch.Tensor, bsz: int, steps: int, t0: int = 0) -> Var:
    """x [B*T, D] + position_signal[t0:t0+T] (encoders/transformer.py:23-45, 187-189)."""
    d = x.shape[1]
    out = tape.new((bsz * steps, d))
    ops.add_position(x.data.view(bsz, steps, d), signal, out.data.view(bsz, steps, d), t0)

    def bwd():
        if out.grad is not None and x.needs_grad:
            ops.ew("copy", out.grad, None, tape.grad(x), accumulate=True)
    tape.record(bwd)
    return out


def add_row(tape: Tape, x: Var, row: Var) -> Var:
    """x [N, D] + row [1, D] on every row (the target-modality embedding of encoders/transformer.py:202-203);
    the broadcast add is the position-signal kernel with one time step per row."""
    n, d = x.shape
    out = tape.new((n, d))
    ops.add_position(x.data.view(n, 1, d), row.data.reshape(1, d), out.data.view(n, 1, d), 0)

    def bwd():
        if out.grad is None:
            return
        if x.needs_grad:
            ops.ew("copy", out.grad, None, tape.grad(x), accumulate=True)
        if row.needs_grad:
            ops.colsum(out.grad, tape.grad(row).view(d), accumulate=True)
    tape.record(bwd)
    return out


def time_sum(tape: Tape, x: Var, bsz: int, steps: int) -> Var:
    """[B*T, D] -> [B, D] sum over time (encoders/transformer.py:170-172)."""
    d = x.shape[1]
    out = tape.new((bsz, d))
    ops.time_sum(x.data.view(bsz, steps, d), out.data)

    def bwd():
        if out.grad is not None and x.needs_grad:
            ops.time_bcast_add(out.grad, tape.grad(x).view(bsz, steps, d))
    tape.record(bwd)
    return out


def xent(tape: Tape, logits: Var, targets: torch.Tensor, weights: Optional[torch.Tensor],
         grad_scale: Optional[torch.Tensor], label_smoothing: float = 0.0) -> torch.Tensor:
    """Masked sparse softmax cross entropy per row (autoregressive.py:289-316).  When recording,
    the kernel overwrites the logits with their gradient (scaled by ``grad_scale``), which then
    *is* the gradient buffer of ``logits``."""
    loss_rows = tape.buf((logits.shape[0],))
    ops.xent(logits.data, targets, weights, loss_rows, grad_scale, tape.recording, label_smoothing)
    if tape.recording:
        logits.grad = logits.data
    return loss_rows
