"""Scaled dot-product attention as an RNN-decoder attention (mirror of the classes of
neuralmonkey/attention/scaled_dot_product.py:247-400: ``MultiHeadAttention``,
``ScaledDotProdAttention``; the Transformer layers use the same kernel through
``nn/transformer_blocks.py``).

One decoder step is K8 with Tq = 1 (tests/factored.ini, tests/post-edit.ini): the
query is the cell output [R,Q], keys / values the states of one or two encoders.  For
``n_heads > 1`` the reference's ``attention()`` (:98-226) adds four bias-free dense layers; the key
and value projections do not depend on the decoder step and are computed once per batch here
(the reference re-computes them inside the while-loop body), query / output projections per step.
The fused kernel ``nm_sdp_attn_fwd`` scores, masks (``e*m + (1-m)*-1e9``, :45-69), soft-maxes,
drops out and sums a head's keys from LDS; k hypotheses of a sentence share its keys (row // k).

Constraints the reference imposes through tensor shapes (:151-166) are checked up front: the key
dimension equals the query (decoder state) size, and so does the value dimension when there are
no projections (n_heads == 1).  The context therefore always has the decoder's state size.
"""
from typing import Any, Optional

import torch

from .. import autodiff as F
from ..model.model_part import InitializerSpecs, ModelPart
from ..variables import glorot_uniform_initializer
from .base_attention import Attendable, AttentionLoopState, BaseAttention, get_attention_mask, get_attention_states

PROJECTIONS = ("query_proj", "keys_proj", "vals_proj", "output_proj")


class MultiHeadAttention(BaseAttention):
    tape_only = True

    # pylint: disable=too-many-arguments
    def __init__(self, name: str, n_heads: int, keys_encoder: Attendable, values_encoder: Attendable = None,
                 dropout_keep_prob: float = 1.0, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        BaseAttention.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.n_heads = n_heads
        self.dropout_keep_prob = dropout_keep_prob
        self.keys_encoder = keys_encoder
        self.values_encoder = values_encoder if values_encoder is not None else keys_encoder
        if self.n_heads <= 0:
            raise ValueError("Number of heads must be greater than zero.")
        if self.dropout_keep_prob <= 0.0 or self.dropout_keep_prob > 1.0:
            raise ValueError("Dropout keep prob must be inside (0,1].")
        self.set_default_initializer(glorot_uniform_initializer())      # variance_scaling(fan_avg, uniform) :283-284
        self.rows_per_key = 1
        self._decoder = None

    @property
    def dependencies(self):
        return ModelPart.dependencies.fget(self) + ["keys_encoder", "values_encoder"]

    @property
    def context_vector_size(self) -> int:
        """The reference reads the value dimension (:369-371); the shape checks of ``attention()`` make it
        the query size in every configuration that builds."""
        if self.query_state_size is None:
            return self.values_encoder.dimension
        return self.query_state_size

    @property
    def state_size(self) -> int:
        return self.context_vector_size

    def bind_query_size(self, size: int) -> None:
        if self.query_state_size is not None and self.query_state_size != size:
            raise ValueError("Attention '{}' is queried with two different state sizes ({} vs {})"
                             .format(self.name, self.query_state_size, size))
        kdim, vdim = self.keys_encoder.dimension, self.values_encoder.dimension
        if size != kdim:                                                  # :155-158
            raise ValueError("Queries and keys do not match in the last dimension. Queries: {}, Keys: {}"
                             .format(size, kdim))
        if size % self.n_heads != 0:                                      # :165-168
            raise ValueError("Last dimension of the query ({}) should be divisible by the number of heads ({})"
                             .format(size, self.n_heads))
        if self.n_heads == 1 and vdim != size:
            raise ValueError("Without head projections the values ({}) must have the dimension of the "
                             "queries ({})".format(vdim, size))
        self.query_state_size = size

    def bind_decoder(self, decoder) -> None:
        if self._decoder is not None and self._decoder is not decoder:
            raise ValueError("Attention '{}' is used by two decoders".format(self.name))
        others = [a for a in decoder.attentions if a is not self and isinstance(a, MultiHeadAttention)
                  and a.n_heads > 1]
        if self.n_heads > 1 and others:
            raise ValueError("two multi-head attentions in decoder '{}' would share the dense layers "
                             "query_proj / keys_proj / vals_proj / output_proj of its step scope".format(decoder.name))
        self._decoder = decoder

    def declare_decoder_variables(self, dec, store) -> None:
        """tf.layers.dense(..., name=...) inside the decoder's step scope (:170-176, 217-219)."""
        if self.n_heads == 1:
            return
        q = self.query_state_size
        dims = {"query_proj": q, "keys_proj": self.keys_encoder.dimension, "vals_proj": self.values_encoder.dimension,
                "output_proj": q}
        init = glorot_uniform_initializer()
        for proj in PROJECTIONS:
            dec.declare(store, "attention_decoder/{}/kernel".format(proj), (dims[proj], q), init)

    def initial_loop_state(self, ctx, rows: int, max_steps: int, precompute: bool = True) -> AttentionLoopState:
        slen = get_attention_states(self.keys_encoder, ctx).shape[1]
        return AttentionLoopState(
            contexts=ctx.buffer((id(self), "contexts", rows, max_steps), (max_steps, rows, self.context_vector_size)),
            weights=ctx.buffer((id(self), "weights", rows, max_steps, slen), (max_steps, rows, self.n_heads * slen)),
            step=0)

    def finalize_loop(self, key: str, last_loop_state: Any) -> None:
        steps = last_loop_state.step
        w = last_loop_state.weights[:steps]
        w4 = w.view(steps, w.shape[1], self.n_heads, -1)
        for i in range(self.n_heads):                                     # :362-366
            self.histories["{}_head{}".format(key, i)] = w4[:, :, i]

    def attention(self, ctx, query, decoder_prev_state, decoder_input, loop_state):
        raise NotImplementedError("'{}' runs through tape_session().step (general decoder path)"
                                  .format(type(self).__name__))

    def tape_session(self, tape, train_mode: bool) -> "DotProductSession":
        return DotProductSession(self, tape, train_mode)


class ScaledDotProdAttention(MultiHeadAttention):
    # pylint: disable=too-many-arguments
    def __init__(self, name: str, keys_encoder: Attendable, values_encoder: Attendable = None,
                 dropout_keep_prob: float = 1.0, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        MultiHeadAttention.__init__(self, name, 1, keys_encoder, values_encoder, dropout_keep_prob, reuse,
                                    save_checkpoint, load_checkpoint, initializers)


class DotProductSession:
    """One decoding run on a tape: encoder states (and their head projections) once, K8 per step."""

    def __init__(self, att: MultiHeadAttention, tape: F.Tape, train_mode: bool):
        if att._decoder is None:                                          # pylint: disable=protected-access
            raise RuntimeError("Attention '{}' is not attached to a decoder".format(att.name))
        ctx = tape.ctx
        self.att, self.tape, self.train = att, tape, train_mode
        self.dec = att._decoder                                           # pylint: disable=protected-access
        keys = get_attention_states(att.keys_encoder, ctx)
        self.bsz, self.slen, kdim = keys.shape
        self.keys_in = tape.leaf(keys.reshape(self.bsz * self.slen, kdim), needs_grad=True)
        if att.values_encoder is att.keys_encoder:
            self.vals_in = self.keys_in
        else:
            vals = get_attention_states(att.values_encoder, ctx)
            if vals.shape[1] != self.slen:                                # :160-163
                raise ValueError("Keys and values 'time' dimension does not match. Keys: {}, Values: {}"
                                 .format(self.slen, vals.shape[1]))
            self.vals_in = tape.leaf(vals.reshape(self.bsz * self.slen, vals.shape[2]), needs_grad=True)
        self.mask = get_attention_mask(att.keys_encoder, ctx)
        self.k, self.v = self.keys_in, self.vals_in
        if att.n_heads > 1:
            self.k = F.linear(tape, self.keys_in, self._w("keys_proj"))
            self.v = F.linear(tape, self.vals_in, self._w("vals_proj"))
        self.t = 0

    def _w(self, proj: str) -> F.Var:
        return self.tape.param(self.dec, "attention_decoder/{}/kernel".format(proj))

    @property
    def shape_key(self):
        return (self.bsz, self.slen, self.att.n_heads)

    def encoder_grads(self):
        att, out = self.att, []
        if self.keys_in.grad is not None:
            out.append((att.keys_encoder, self.keys_in.grad.view(self.bsz, self.slen, -1)))
        if self.vals_in is not self.keys_in and self.vals_in.grad is not None:
            out.append((att.values_encoder, self.vals_in.grad.view(self.bsz, self.slen, -1)))
        return out

    def step(self, query: F.Var, w_out: Optional[torch.Tensor] = None, prev_state=None, rnn_input=None) -> F.Var:
        """MultiHeadAttention.attention (:297-352)."""
        tape, att = self.tape, self.att
        ctx = tape.ctx
        rows = query.shape[0]
        heads = att.n_heads
        q = F.linear(tape, query, self._w("query_proj")) if heads > 1 else query
        w4 = None if w_out is None else w_out.view(rows, heads, 1, self.slen)
        out = F.sdp_attention(tape, q, self.k, self.v, self.mask, heads, rows, 1, self.bsz, self.slen, False,
                              att.dropout_keep_prob if self.train else 1.0, ctx.salt(att.name, "weights", self.t),
                              w_out=w4)
        self.t += 1
        return F.linear(tape, out, self._w("output_proj")) if heads > 1 else out
