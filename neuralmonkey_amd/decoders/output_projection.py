"""Output projections (mirror of neuralmonkey/decoders/output_projection.py).

The reference returns ``(callable, output_size)`` tuples whose callable builds
``tf.layers.dense`` ops; here the first element is a small object that
declares the same variables (``<decoder>/attention_decoder/<scope>/{kernel,bias}``)
and applies the projection with MFMA GEMMs over any number of rows.

Two entry points per projection:
  ``apply``      plain tensors, no gradient bookkeeping -- the hand-scheduled
                 fast path of the decoder (NonlinearOutput only) hoists it out
                 of the time loop and back-propagates it itself;
  ``apply_var``  the same arithmetic on an autodiff tape (general path).
Concatenations are never materialised: ``dense(concat(parts))`` runs as
accumulating GEMMs over the row blocks of the kernel.
"""
from typing import Callable, List, Tuple, Union

from .. import autodiff as F
from .. import ops
from ..checking import check_argument_types
from ..variables import zeros_initializer


def _act_name(fn) -> str:
    name = getattr(fn, "nm_name", None) or getattr(fn, "__name__", None) or str(fn)
    if name not in ("tanh", "relu", "identity", "sigmoid"):
        raise ValueError("Unsupported activation for the HIP engine: {}".format(fn))
    return name


def _dense_blocks(tape, decoder, scope: str, parts, sizes):
    """dense(concat(parts)) with kernel ``attention_decoder/<scope>/kernel`` [sum(sizes), O]."""
    w = tape.param(decoder, "attention_decoder/{}/kernel".format(scope))
    b = tape.param(decoder, "attention_decoder/{}/bias".format(scope))
    out, row = None, 0
    for x, sz in zip(parts, sizes):
        out = F.linear(tape, x, tape.rows(w, row, row + sz), b if out is None else None, out=out,
                       accumulate=out is not None)
        row += sz
    return out


class OutputProjection:
    """Subclasses set ``scope`` names and ``dropout_keep_prob``."""
    dropout_keep_prob = 1.0

    def declare_variables(self, decoder, store, state_size: int, emb_size: int, ctx_sizes: List[int]):
        raise NotImplementedError

    def apply(self, ctx, decoder, state, prev_output, ctx_tensors, out):
        raise NotImplementedError("{} runs on the general (taped) path only".format(type(self).__name__))

    def apply_var(self, tape, decoder, state, prev_output, ctx_vars, train_mode: bool, salt: int):
        raise NotImplementedError

    def _dropout(self, tape, x, train_mode, salt):
        return F.dropout(tape, x, self.dropout_keep_prob, train_mode, salt)


class NonlinearOutput(OutputProjection):
    """nonlinear_output (output_projection.py:115-130):
    dropout(act(dense(concat[state, prev_output, *ctx])))."""

    def __init__(self, output_size: int, activation: str, dropout_keep_prob: float, scope: str = "dense"):
        self.output_size = output_size
        self.activation = activation
        self.dropout_keep_prob = dropout_keep_prob
        self.scope = scope

    def declare_variables(self, decoder, store, state_size, emb_size, ctx_sizes):
        self.sizes = [state_size, emb_size] + list(ctx_sizes)
        decoder.declare(store, "attention_decoder/{}/kernel".format(self.scope),
                        (sum(self.sizes), self.output_size))
        decoder.declare(store, "attention_decoder/{}/bias".format(self.scope), (self.output_size,),
                        zeros_initializer())

    def kernel(self, ctx, decoder):
        return decoder.var(ctx, "attention_decoder/{}/kernel".format(self.scope))

    def bias(self, ctx, decoder):
        return decoder.var(ctx, "attention_decoder/{}/bias".format(self.scope))

    def apply(self, ctx, decoder, state, prev_output, ctx_tensors, out):
        """out[R,O] = act([state | prev_output | ctx...] . W + b) as accumulating
        GEMMs over the row blocks of W (no concat copy)."""
        w, b = self.kernel(ctx, decoder), self.bias(ctx, decoder)
        parts = [state, prev_output] + list(ctx_tensors)
        row = 0
        for i, (x, sz) in enumerate(zip(parts, self.sizes)):
            last = i == len(parts) - 1
            ops.gemm(x, w[row:row + sz], out=out, accumulate=i > 0,
                     bias=b if last else None,
                     act=(self.activation if self.activation != "identity" else None) if last else None)
            row += sz
        return out

    def apply_concat(self, ctx, decoder, concat, out):
        """The same projection when the producers already wrote [state | prev_output | ctx...]
        side by side into ``concat`` [R, sum(sizes)]: one GEMM with the activation in its epilogue."""
        return ops.gemm(concat, self.kernel(ctx, decoder), out=out, bias=self.bias(ctx, decoder),
                        act=self.activation if self.activation != "identity" else None)

    def apply_var(self, tape, decoder, state, prev_output, ctx_vars, train_mode, salt):
        pre = _dense_blocks(tape, decoder, self.scope, [state, prev_output] + list(ctx_vars), self.sizes)
        return self._dropout(tape, F.ACTIVATIONS[self.activation](tape, pre), train_mode, salt)


class NematusOutput(OutputProjection):
    """nematus_output (output_projection.py:76-112): three dense layers ``rnn_state``,
    ``prev_out``, ``context`` (each with a bias) summed, activation, dropout."""

    def __init__(self, output_size: int, activation: str, dropout_keep_prob: float):
        self.output_size, self.activation, self.dropout_keep_prob = output_size, activation, dropout_keep_prob

    def declare_variables(self, decoder, store, state_size, emb_size, ctx_sizes):
        self.ctx_sizes = list(ctx_sizes)
        for scope, size in (("rnn_state", state_size), ("prev_out", emb_size), ("context", sum(ctx_sizes))):
            decoder.declare(store, "attention_decoder/{}/kernel".format(scope), (size, self.output_size))
            decoder.declare(store, "attention_decoder/{}/bias".format(scope), (self.output_size,),
                            zeros_initializer())

    def apply_var(self, tape, decoder, state, prev_output, ctx_vars, train_mode, salt):
        acc = _dense_blocks(tape, decoder, "rnn_state", [state], [state.shape[1]])
        F.add_(tape, acc, _dense_blocks(tape, decoder, "prev_out", [prev_output], [prev_output.shape[1]]))
        F.add_(tape, acc, _dense_blocks(tape, decoder, "context", list(ctx_vars), self.ctx_sizes))
        return self._dropout(tape, F.ACTIVATIONS[self.activation](tape, acc), train_mode, salt)


class MaxoutOutput(OutputProjection):
    """maxout_output (output_projection.py:133-160) over nn/projection.py:7-35: a dense layer of
    2*size units (scope ``MaxoutProjection/MaxoutProjection``), max over the two halves, dropout."""

    def __init__(self, maxout_size: int, dropout_keep_prob: float):
        self.output_size, self.dropout_keep_prob = maxout_size, dropout_keep_prob
        self.scope = "MaxoutProjection/MaxoutProjection"

    def declare_variables(self, decoder, store, state_size, emb_size, ctx_sizes):
        self.sizes = [state_size, emb_size] + list(ctx_sizes)
        decoder.declare(store, "attention_decoder/{}/kernel".format(self.scope),
                        (sum(self.sizes), 2 * self.output_size))
        decoder.declare(store, "attention_decoder/{}/bias".format(self.scope), (2 * self.output_size,),
                        zeros_initializer())

    def apply_var(self, tape, decoder, state, prev_output, ctx_vars, train_mode, salt):
        pre = _dense_blocks(tape, decoder, self.scope, [state, prev_output] + list(ctx_vars), self.sizes)
        return self._dropout(tape, F.maxout(tape, pre, 2), train_mode, salt)


class MlpOutput(OutputProjection):
    """mlp_output (output_projection.py:163-188) over multilayer_projection
    (nn/projection.py:38-58): dense+activation+dropout per layer, scope ``deep_output_mlp``."""

    def __init__(self, layer_sizes: List[int], activation: str, dropout_keep_prob: float):
        self.layer_sizes, self.activation, self.dropout_keep_prob = list(layer_sizes), activation, dropout_keep_prob
        self.output_size = self.layer_sizes[-1]

    def declare_variables(self, decoder, store, state_size, emb_size, ctx_sizes):
        self.sizes = [state_size, emb_size] + list(ctx_sizes)
        width = sum(self.sizes)
        for i, size in enumerate(self.layer_sizes):
            scope = "deep_output_mlp/mlp_layer_{}".format(i)
            decoder.declare(store, "attention_decoder/{}/kernel".format(scope), (width, size))
            decoder.declare(store, "attention_decoder/{}/bias".format(scope), (size,), zeros_initializer())
            width = size

    def apply_var(self, tape, decoder, state, prev_output, ctx_vars, train_mode, salt):
        parts, sizes = [state, prev_output] + list(ctx_vars), self.sizes
        x = None
        for i, size in enumerate(self.layer_sizes):
            scope = "deep_output_mlp/mlp_layer_{}".format(i)
            pre = _dense_blocks(tape, decoder, scope, parts, sizes)
            x = F.dropout(tape, F.ACTIVATIONS[self.activation](tape, pre), self.dropout_keep_prob, train_mode,
                          (salt + 0x632BE5AB * (i + 1)) & 0xFFFFFFFF)
            parts, sizes = [x], [size]
        return x


OutputProjectionSpec = Union[Tuple[OutputProjection, int], OutputProjection]


def nonlinear_output(output_size: int, activation_fn: Callable = None,
                     dropout_keep_prob: float = 1.0) -> Tuple[OutputProjection, int]:
    act = "tanh" if activation_fn is None else _act_name(activation_fn)
    return NonlinearOutput(output_size, act, dropout_keep_prob), output_size


def nematus_output(output_size: int, activation_fn: Callable = None,
                   dropout_keep_prob: float = 1.0) -> Tuple[OutputProjection, int]:
    act = "tanh" if activation_fn is None else _act_name(activation_fn)
    return NematusOutput(output_size, act, dropout_keep_prob), output_size


def maxout_output(maxout_size: int, dropout_keep_prob: float = 1.0) -> Tuple[OutputProjection, int]:
    return MaxoutOutput(maxout_size, dropout_keep_prob), maxout_size


def mlp_output(layer_sizes: List[int], activation: Callable = None,
               dropout_keep_prob: float = 1.0) -> Tuple[OutputProjection, int]:
    act = "tanh" if activation is None else _act_name(activation)
    return MlpOutput(layer_sizes, act, dropout_keep_prob), layer_sizes[-1]


class LegacyOutput(OutputProjection):
    """_legacy_linear / _legacy_relu (output_projection.py:33-72): ``[relu](dense(concat[state, *ctx]))`` in the
    scope ``AttnOutputProjection``; the previous output is not an input and there is no dropout."""

    def __init__(self, output_size: int, activation: str):
        self.output_size, self.activation = output_size, activation

    def declare_variables(self, decoder, store, state_size, emb_size, ctx_sizes):
        self.sizes = [state_size] + list(ctx_sizes)
        decoder.declare(store, "attention_decoder/AttnOutputProjection/kernel", (sum(self.sizes), self.output_size))
        decoder.declare(store, "attention_decoder/AttnOutputProjection/bias", (self.output_size,),
                        zeros_initializer())

    def apply_var(self, tape, decoder, state, prev_output, ctx_vars, train_mode, salt):
        pre = _dense_blocks(tape, decoder, "AttnOutputProjection", [state] + list(ctx_vars), self.sizes)
        return F.ACTIVATIONS[self.activation](tape, pre)


def _legacy_linear(output_size: int) -> Tuple[OutputProjection, int]:
    check_argument_types()
    return LegacyOutput(output_size, "identity"), output_size


def _legacy_relu(output_size: int) -> Tuple[OutputProjection, int]:
    check_argument_types()
    return LegacyOutput(output_size, "relu"), output_size
