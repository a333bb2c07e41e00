"""A Stateful encoder's output as a "static" context vector (mirror of
neuralmonkey/attention/stateful_context.py:14-94: ``StatefulContext``).

Not an attention: the context is ``encoder.output`` at every step, whatever the query; the weights are
ones of width 1.  It exists so that a decoder cell / output projection can be conditioned on a
sentence vector through the ``attentions`` parameter.  No variables.  Under beam search every
hypothesis of a sentence gets the sentence's vector (row // k) -- the reference hands the decoder B
context rows for B*k hypotheses there and cannot run.
"""
from typing import Any, Optional

import torch

from .. import autodiff as F
from .. import ops
from ..model.model_part import InitializerSpecs, ModelPart
from ..model.stateful import Stateful
from .base_attention import AttentionLoopState, BaseAttention


class StatefulContext(BaseAttention):
    tape_only = True

    # pylint: disable=too-many-arguments
    def __init__(self, name: str, encoder: Stateful, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        BaseAttention.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.encoder = encoder
        self.rows_per_key = 1

    @property
    def dependencies(self):
        return ModelPart.dependencies.fget(self) + ["encoder"]

    @property
    def context_vector_size(self) -> int:
        return self.encoder.output_size

    @property
    def state_size(self) -> int:
        return self.context_vector_size

    def bind_query_size(self, size: int) -> None:
        self.query_state_size = size                      # the query is not looked at

    def initial_loop_state(self, ctx, rows: int, max_steps: int, precompute: bool = True) -> AttentionLoopState:
        weights = ctx.buffer((id(self), "weights", rows, max_steps), (max_steps, rows, 1))
        ops.fill(weights, 1.0)                                # :70
        return AttentionLoopState(
            contexts=ctx.buffer((id(self), "contexts", rows, max_steps), (max_steps, rows, self.context_vector_size)),
            weights=weights, step=0)

    def finalize_loop(self, key: str, last_loop_state: Any) -> None:
        pass

    def attention(self, ctx, query, decoder_prev_state, decoder_input, loop_state):
        raise NotImplementedError("'StatefulContext' runs through tape_session().step (general decoder path)")

    def tape_session(self, tape: F.Tape, train_mode: bool) -> "StatefulContextSession":
        return StatefulContextSession(self, tape)


class StatefulContextSession:
    """One decoding run on a tape: the encoder output as a leaf, tiled once for a beam."""

    def __init__(self, att: StatefulContext, tape: F.Tape):
        ctx = tape.ctx
        self.att, self.tape = att, tape
        out = att.encoder.output(ctx)                                        # [B, C]
        self.bsz, k = out.shape[0], att.rows_per_key
        self.leaf = tape.leaf(out, needs_grad=True)
        self.var = self.leaf
        if k > 1:                                                             # beam search: inference only
            rows = torch.arange(self.bsz * k, dtype=torch.int32, device=out.device) // k
            tiled = ctx.buffer((id(att), "tiled", self.bsz, k), (self.bsz * k, out.shape[1]))
            ops.gather_rows(out, rows, tiled)
            self.var = tape.leaf(tiled)

    @property
    def shape_key(self):
        return (self.bsz, self.att.rows_per_key)

    def encoder_grads(self):
        """The gradient goes to the encoder's OUTPUT, not to its states."""
        if self.leaf.grad is None:
            return []
        return [(self.att.encoder, self.leaf.grad, "output")]

    def step(self, query: F.Var, w_out: Optional[torch.Tensor] = None, prev_state=None, rnn_input=None) -> F.Var:
        if query.shape[0] != self.var.shape[0]:
            raise ValueError("StatefulContext '{}': {} query rows for {} context rows"
                             .format(self.att.name, query.shape[0], self.var.shape[0]))
        return self.var
