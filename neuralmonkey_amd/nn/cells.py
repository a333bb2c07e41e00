"""RNN cells of the general (taped) path.

    GRU         tf.contrib.rnn.GRUCell via OrthoGRUCell      nn/ortho_gru_cell.py:44-53
    NematusGRU  NematusGRUCell                               nn/ortho_gru_cell.py:57-105
    LSTM        tf.contrib.rnn.LSTMCell (forget_bias 1.0)    encoders/recurrent.py:21, decoders/decoder.py:29

Each cell declares the variables TF would create under ``<scope>/`` (names as
in scripts/import_nematus.py:88-130) and evaluates one step as MFMA GEMMs plus
the element-wise kernels of ``autodiff``.  The plain-GRU fast path
(``nn/gru.py``: fused GEMM epilogues, HIP-graph loops) uses the same variables.
"""
import os
from typing import Tuple

from .. import autodiff as F
from ..variables import constant_initializer, orthogonal_initializer, zeros_initializer

RNN_CELL_TYPES = ("NematusGRU", "GRU", "LSTM")
# gates, candidate and blend of a NematusGRUCell step in one launch each way (autodiff.nematus_cell); 0: one launch per
# element-wise operation as in rounds 2-5
FUSED_NEMATUS_CELL = os.environ.get("NM_NEMATUS_CELL_FUSED", "1") != "0"
# ... and its four products as two against column-concatenated copies of the kernels (autodiff.nematus_cell_merged)
MERGED_NEMATUS_CELL = os.environ.get("NM_NEMATUS_CELL_MERGED", "1") != "0"
# ... and the input half of a loop whose inputs are known beforehand (teacher-forced decoder) for all steps at once
HOIST_INPUTS = os.environ.get("NM_HOIST_INPUTS", "1") != "0"


class Cell:
    """One recurrent cell bound to ``part`` (the ModelPart owning the variables)."""
    state_count = 1                  # tensors carried between steps

    def __init__(self, part, scope: str, input_size: int, num_units: int):
        self.part, self.scope, self.input_size, self.num_units = part, scope, input_size, num_units

    def _n(self, suffix: str) -> str:
        return "{}/{}".format(self.scope, suffix)

    def declare_variables(self, store) -> None:
        raise NotImplementedError

    def step(self, tape: F.Tape, x: F.Var, state: Tuple[F.Var, ...]) -> Tuple[F.Var, Tuple[F.Var, ...]]:
        """(output, new_state)."""
        raise NotImplementedError


class GRUCell(Cell):
    """r,u = sigmoid([x,h].Wg + bg) ; c = tanh([x, r*h].Wc + bc) ; h' = u*h + (1-u)*c."""

    def __init__(self, part, scope, input_size, num_units, cell_scope: str = "OrthoGRUCell"):
        Cell.__init__(self, part, "{}/{}".format(scope, cell_scope) if scope else cell_scope, input_size,
                      num_units)

    def declare_variables(self, store) -> None:
        d, h = self.input_size, self.num_units
        self.part.declare(store, self._n("gates/kernel"), (d + h, 2 * h), orthogonal_initializer())
        self.part.declare(store, self._n("gates/bias"), (2 * h,), constant_initializer(1.0))
        self.part.declare(store, self._n("candidate/kernel"), (d + h, h), orthogonal_initializer())
        self.part.declare(store, self._n("candidate/bias"), (h,), zeros_initializer())

    def step(self, tape, x, state):
        (h_prev,) = state
        d, h = self.input_size, self.num_units
        wg, bg = tape.param(self.part, self._n("gates/kernel")), tape.param(self.part, self._n("gates/bias"))
        wc, bc = tape.param(self.part, self._n("candidate/kernel")), tape.param(self.part,
                                                                                 self._n("candidate/bias"))
        g_pre = F.linear(tape, x, tape.rows(wg, 0, d), bg)
        F.linear(tape, h_prev, tape.rows(wg, d, d + h), out=g_pre, accumulate=True)
        g = F.sigmoid(tape, g_pre)
        r, u = tape.cols(g, 0, h), tape.cols(g, h, 2 * h)
        rh = F.mul(tape, r, h_prev)
        c_pre = F.linear(tape, x, tape.rows(wc, 0, d), bc)
        F.linear(tape, rh, tape.rows(wc, d, d + h), out=c_pre, accumulate=True)
        c = F.tanh(tape, c_pre)
        h_new = F.blend(tape, u, h_prev, c)
        return h_new, (h_new,)


class NematusGRUCell(Cell):
    """Reset gate applied after the state projection: c = tanh(x.Wc + r*(h.Uc + bcs) + bci)."""

    def __init__(self, part, scope, input_size, num_units, use_state_bias: bool = False,
                 use_input_bias: bool = True, cell_scope: str = "nematus_gru_cell"):
        Cell.__init__(self, part, "{}/{}".format(scope, cell_scope) if scope else cell_scope, input_size,
                      num_units)
        self.use_state_bias, self.use_input_bias = use_state_bias, use_input_bias

    def declare_variables(self, store) -> None:
        d, h = self.input_size, self.num_units
        for block, width in (("gates", 2 * h), ("candidate", h)):
            self.part.declare(store, self._n(block + "/input_proj/kernel"), (d, width))
            self.part.declare(store, self._n(block + "/state_proj/kernel"), (h, width), orthogonal_initializer())
            if self.use_input_bias:
                self.part.declare(store, self._n(block + "/input_proj/bias"), (width,), zeros_initializer())
            if self.use_state_bias:
                self.part.declare(store, self._n(block + "/state_proj/bias"), (width,), zeros_initializer())
        # NematusGRUCell overrides GRUCell.call and inherits GRUCell.build (nn/ortho_gru_cell.py:57-72): TensorFlow
        # creates the plain cell's four variables in the cell's scope as well, no computation reads them, and every
        # checkpoint of the reference holds them (kernels from the scope's initializer: no kernel_initializer is
        # passed; gates bias 1, candidate bias 0)
        self.part.declare_checkpoint_only(store, self._n("gates/kernel"), (d + h, 2 * h))
        self.part.declare_checkpoint_only(store, self._n("gates/bias"), (2 * h,), constant_initializer(1.0))
        self.part.declare_checkpoint_only(store, self._n("candidate/kernel"), (d + h, h))
        self.part.declare_checkpoint_only(store, self._n("candidate/bias"), (h,), zeros_initializer())

    def _merged(self, tape):
        """[W_g | W_c] and [U_g | U_c] (and their biases) as column-concatenated copies, refreshed once per run context
        (the variables change with every optimizer step; a decoding batch keeps its context)."""
        ctx = tape.ctx
        key = (id(self), "merged_kernels", bool(tape.recording))
        hit = ctx.memo.get(key)
        if hit is not None:
            return hit
        from .. import ops
        d, h = self.input_size, self.num_units

        def var(block, which, kind):
            return tape.param(self.part, self._n("{}/{}_proj/{}".format(block, which, kind)))
        params = {"gi": (var("gates", "input", "kernel"), var("gates", "input", "bias") if self.use_input_bias else None),
                  "ci": (var("candidate", "input", "kernel"),
                         var("candidate", "input", "bias") if self.use_input_bias else None),
                  "gs": (var("gates", "state", "kernel"), var("gates", "state", "bias") if self.use_state_bias else None),
                  "cs": (var("candidate", "state", "kernel"),
                         var("candidate", "state", "bias") if self.use_state_bias else None)}
        w_in = ctx.buffer((id(self), "w_in_cat"), (d, 3 * h))
        w_st = ctx.buffer((id(self), "w_st_cat"), (h, 3 * h))
        ops.copy_cols(params["gi"][0].data, w_in[:, :2 * h])
        ops.copy_cols(params["ci"][0].data, w_in[:, 2 * h:])
        ops.copy_cols(params["gs"][0].data, w_st[:, :2 * h])
        ops.copy_cols(params["cs"][0].data, w_st[:, 2 * h:])
        b_in = b_st = None
        if self.use_input_bias:
            b_in = ctx.buffer((id(self), "b_in_cat"), (3 * h,))
            ops.copy_cols(params["gi"][1].data.view(1, -1), b_in[:2 * h].view(1, -1))
            ops.copy_cols(params["ci"][1].data.view(1, -1), b_in[2 * h:].view(1, -1))
        if self.use_state_bias:
            b_st = ctx.buffer((id(self), "b_st_cat"), (3 * h,))
            ops.copy_cols(params["gs"][1].data.view(1, -1), b_st[:2 * h].view(1, -1))
            ops.copy_cols(params["cs"][1].data.view(1, -1), b_st[2 * h:].view(1, -1))
        hit = ctx.memo[key] = (w_in, b_in, w_st, b_st, params)
        return hit

    def _proj(self, tape, block, which, inp, use_bias, out=None, accumulate=False):
        w = tape.param(self.part, self._n("{}/{}_proj/kernel".format(block, which)))
        b = tape.param(self.part, self._n("{}/{}_proj/bias".format(block, which))) if use_bias else None
        return F.linear(tape, inp, w, b, out=out, accumulate=accumulate)

    def project_inputs(self, tape, x_all):
        """The input half of the cell for the inputs of ALL steps at once ([T*B, D] -> Var [T*B, 3H], rows to be handed to
        ``step(..., x_proj=)``), or None where the merged cell step does not apply."""
        h = self.num_units
        if not (MERGED_NEMATUS_CELL and FUSED_NEMATUS_CELL and HOIST_INPUTS and x_all.data.is_cuda and h % 4 == 0
                and self.input_size % 4 == 0):
            return None
        w_in, b_in, _, _, params = self._merged(tape)
        return F.nematus_input_projection(tape, x_all, w_in, b_in, params)

    def step(self, tape, x, state, x_proj=None, out=None):
        """``out`` (optional Var): where the new state is to be written (rows of a buffer of all steps); the caller
        checks what it got back -- only the merged step writes there."""
        (h_prev,) = state
        h = self.num_units
        if MERGED_NEMATUS_CELL and FUSED_NEMATUS_CELL and h_prev.data.is_cuda and h % 4 == 0 and self.input_size % 4 == 0:
            w_in, b_in, w_st, b_st, params = self._merged(tape)
            h_new = F.nematus_cell_merged(tape, x, h_prev, w_in, b_in, w_st, b_st, params, x_proj=x_proj, out=out)
            return h_new, (h_new,)
        assert x_proj is None, "project_inputs() answered for a cell that does not take the merged step"
        g_pre = self._proj(tape, "gates", "state", h_prev, self.use_state_bias)
        self._proj(tape, "gates", "input", x, self.use_input_bias, out=g_pre, accumulate=True)
        sc = self._proj(tape, "candidate", "state", h_prev, self.use_state_bias)
        if FUSED_NEMATUS_CELL and g_pre.data.is_cuda:
            ci = self._proj(tape, "candidate", "input", x, self.use_input_bias)
            h_new = F.nematus_cell(tape, g_pre, sc, ci, h_prev)           # gates, candidate and blend: one launch
            return h_new, (h_new,)
        g = F.sigmoid(tape, g_pre)
        r, u = tape.cols(g, 0, h), tape.cols(g, h, 2 * h)
        c_pre = F.mul(tape, sc, r)
        self._proj(tape, "candidate", "input", x, self.use_input_bias, out=c_pre, accumulate=True)
        c = F.tanh(tape, c_pre)
        h_new = F.blend(tape, u, h_prev, c)
        return h_new, (h_new,)


class LSTMCell(Cell):
    """z = [x,h].W + b ; i,j,f,o = split(z) ; c' = sigmoid(f+1)*c + sigmoid(i)*tanh(j) ;
    h' = sigmoid(o)*tanh(c').  State = (c, h)."""
    state_count = 2

    def __init__(self, part, scope, input_size, num_units, cell_scope: str = "lstm_cell"):
        Cell.__init__(self, part, "{}/{}".format(scope, cell_scope) if scope else cell_scope, input_size,
                      num_units)

    def declare_variables(self, store) -> None:
        d, h = self.input_size, self.num_units
        self.part.declare(store, self._n("kernel"), (d + h, 4 * h))
        self.part.declare(store, self._n("bias"), (4 * h,), zeros_initializer())

    def project_inputs(self, tape, x_all):
        """x . W_x + b for the inputs of ALL steps at once ([T*B, D] -> Var [T*B, 4H]; the steps add their state halves
        to their rows: ``step(..., x_proj=)``), or None."""
        if not (HOIST_INPUTS and x_all.data.is_cuda):
            return None
        w, b = tape.param(self.part, self._n("kernel")), tape.param(self.part, self._n("bias"))
        return F.linear(tape, x_all, tape.rows(w, 0, self.input_size), b)

    def step(self, tape, x, state, x_proj=None):
        c_prev, h_prev = state
        d, h = self.input_size, self.num_units
        w, b = tape.param(self.part, self._n("kernel")), tape.param(self.part, self._n("bias"))
        z = x_proj if x_proj is not None else F.linear(tape, x, tape.rows(w, 0, d), b)
        F.linear(tape, h_prev, tape.rows(w, d, d + h), out=z, accumulate=True)
        h_new, c_new = F.lstm_cell(tape, z, c_prev, forget_bias=1.0)      # gates + blend: one launch each way
        return h_new, (c_new, h_new)


def make_cell(kind: str, part, scope: str, input_size: int, num_units: int, **kw) -> Cell:
    if kind == "GRU":
        return GRUCell(part, scope, input_size, num_units, **kw)
    if kind == "NematusGRU":
        return NematusGRUCell(part, scope, input_size, num_units, **kw)
    if kind == "LSTM":
        return LSTMCell(part, scope, input_size, num_units, **kw)
    raise ValueError("RNN cell must be a either 'GRU', 'LSTM', or 'NematusGRU'. Not {}".format(kind))
