"""Output projections (mirror of neuralmonkey/decoders/output_projection.py).

The reference returns ``(callable, output_size)`` tuples whose callable builds
``tf.layers.dense`` ops; here the first element is a small object that
declares the same variables (``<decoder>/attention_decoder/dense/{kernel,bias}``)
and applies the projection with MFMA GEMMs over any number of rows, so the
training path can hoist it out of the time loop."""
from typing import Callable, List, Tuple, Union

import torch

from .. import ops
from ..variables import zeros_initializer


def _act_name(fn) -> str:
    name = getattr(fn, "nm_name", None) or getattr(fn, "__name__", None) or str(fn)
    if name not in ("tanh", "relu", "identity"):
        raise ValueError("Unsupported activation for the HIP engine: {}".format(fn))
    return name


class OutputProjection:
    def declare_variables(self, decoder, store, state_size: int, emb_size: int, ctx_sizes: List[int]):
        raise NotImplementedError

    def apply(self, ctx, decoder, state, prev_output, ctx_tensors, out):
        raise NotImplementedError


class NonlinearOutput(OutputProjection):
    """nonlinear_output (output_projection.py:115-130):
    act(dense(concat[state, prev_output, *ctx]))."""

    def __init__(self, output_size: int, activation: str, dropout_keep_prob: float, scope: str = "dense"):
        self.output_size = output_size
        self.activation = activation
        self.dropout_keep_prob = dropout_keep_prob
        self.scope = scope
        if dropout_keep_prob != 1.0:
            raise NotImplementedError("output projection dropout is not implemented in the HIP engine")

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


OutputProjectionSpec = Union[Tuple[OutputProjection, int], OutputProjection]


def nonlinear_output(output_size: int, activation_fn: Callable = None,
                     dropout_keep_prob: float = 1.0) -> Tuple[OutputProjection, int]:
    act = "tanh" if activation_fn is None else _act_name(activation_fn)
    return NonlinearOutput(output_size, act, dropout_keep_prob), output_size


def _legacy_linear(output_size: int) -> Tuple[OutputProjection, int]:
    raise NotImplementedError("_legacy_linear output projection is not implemented in the HIP engine")


def maxout_output(maxout_size: int, dropout_keep_prob: float = 1.0):
    raise NotImplementedError("maxout_output is not implemented in the HIP engine yet")


def nematus_output(output_size: int, activation_fn: Callable = None, dropout_keep_prob: float = 1.0):
    raise NotImplementedError("nematus_output is not implemented in the HIP engine yet")


def mlp_output(layer_sizes: List[int], activation: Callable = None, dropout_keep_prob: float = 1.0):
    raise NotImplementedError("mlp_output is not implemented in the HIP engine yet")
