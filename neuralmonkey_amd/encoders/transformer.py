"""Transformer encoder (mirror of neuralmonkey/encoders/transformer.py).

layer(level) (transformer.py:290-314): inputs = embeddings + position signal, dropout; per layer a
pre-LN self-attention sublayer with residual (:199-222) and a pre-LN ReLU feed-forward sublayer with
residual (:262-288); a final layer norm after the last layer; ``output`` is the sum of the states
over time (:170-172).

MI355X mapping: all B*S positions of a batch go through every dense layer as one MFMA GEMM, the
attention core is one fused kernel launch per layer (``nm_sdp_attn_fwd``: K/V tiles of a head in
LDS, wave-level softmax).  The layer stack is recorded on an autodiff tape; ``backward`` replays it.
"""
from typing import List, Optional

import torch

from .. import autodiff as F
from ..attention.base_attention import Attendable, get_attention_mask, get_attention_states
from ..model.model_part import InitializerSpecs, ModelPart
from ..model.stateful import TemporalStateful, TemporalStatefulWithOutput
from ..nn import transformer_blocks as TB
from ..runtime import tensor
from ..variables import glorot_uniform_initializer, ones_initializer, zeros_initializer


# pylint: disable=too-many-instance-attributes
class TransformerEncoder(ModelPart, TemporalStatefulWithOutput):
    # pylint: disable=too-many-arguments,too-many-locals
    def __init__(self, name: str, input_sequence: TemporalStateful, ff_hidden_size: int, depth: int, n_heads: int,
                 dropout_keep_prob: float = 1.0, attention_dropout_keep_prob: float = 1.0,
                 target_space_id: int = None, use_att_transform_bias: bool = False,
                 use_positional_encoding: bool = True, input_for_cross_attention: Attendable = None,
                 n_cross_att_heads: int = None, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        ModelPart.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.input_sequence = input_sequence
        self.ff_hidden_size = ff_hidden_size
        self.depth = depth
        self.n_heads = n_heads
        self.dropout_keep_prob = dropout_keep_prob
        self.attention_dropout_keep_prob = attention_dropout_keep_prob
        self.target_space_id = target_space_id
        self.use_att_transform_bias = use_att_transform_bias
        self.use_positional_encoding = use_positional_encoding
        self.input_for_cross_attention = input_for_cross_attention
        self.n_cross_att_heads = n_cross_att_heads
        if self.depth <= 0:
            raise ValueError("Depth must be a positive integer.")
        if self.ff_hidden_size <= 0:
            raise ValueError("Feed forward hidden size must be a positive integer.")
        if self.n_heads <= 0:                              # the reference fails on this when it splits the heads
            raise ValueError("Number of heads must be a positive integer.")
        if self.dropout_keep_prob <= 0.0 or self.dropout_keep_prob > 1.0:
            raise ValueError("Dropout keep prob must be inside (0,1].")
        if self.attention_dropout_keep_prob <= 0.0 or self.attention_dropout_keep_prob > 1.0:
            raise ValueError("Dropout keep prob for attn must be in (0,1].")
        if self.target_space_id is not None and (self.target_space_id >= 32 or self.target_space_id < 0):
            raise ValueError("If provided, the target space ID should be between 0 and 31. Was: {}"
                             .format(self.target_space_id))
        if (input_for_cross_attention is None) != (n_cross_att_heads is None):
            raise ValueError("Either both input_for_cross_attention and n_cross_att_heads must be provided "
                             "or none of them.")
        # variance_scaling_initializer(mode="fan_avg", distribution="uniform") (transformer.py:152-153)
        self.set_default_initializer(glorot_uniform_initializer())

    @property
    def dependencies(self) -> List[str]:
        deps = ModelPart.dependencies.fget(self)
        if self.input_for_cross_attention is not None:
            return deps + ["input_for_cross_attention"]
        return deps

    @property
    def model_dimension(self) -> int:
        dim = self.input_sequence.dimension
        if self.input_for_cross_attention is not None and self.input_for_cross_attention.dimension != dim:
            raise ValueError("The input for cross-attention must be of the same dimension as the model, was {}."
                             .format(self.input_for_cross_attention.dimension))
        return dim

    @property
    def dimension(self) -> int:
        return self.model_dimension

    @property
    def output_size(self) -> int:
        return self.model_dimension

    def stage_inputs(self, ctx) -> None:
        """The position-signal table is built on the host: make sure it covers this batch before a
        training step is (re)played as a graph."""
        if self.use_positional_encoding:
            mask = self.input_sequence.temporal_mask(ctx)
            TB.signal_table(ctx, self.model_dimension, mask.shape[1])

    def graph_safe_training(self, train_mode: bool) -> bool:
        return self.input_for_cross_attention is None or getattr(
            self.input_for_cross_attention, "graph_safe_training", lambda t: False)(train_mode)

    def declare_variables(self, store) -> None:
        d = self.model_dimension
        for i in range(self.depth):
            pre = "layer_{}".format(i)
            TB.declare_layer_norm(self, store, pre + "/self_attention", d)
            TB.declare_attention(self, store, pre + "/self_attention", d, self.n_heads, self.use_att_transform_bias)
            if self.input_for_cross_attention is not None:
                TB.declare_layer_norm(self, store, pre + "/cross_attention", d)
                TB.declare_attention(self, store, pre + "/cross_attention", d, self.n_cross_att_heads,
                                     self.use_att_transform_bias)
            TB.declare_feedforward(self, store, pre + "/feedforward", d, self.ff_hidden_size)
        if self.target_space_id is not None:                                   # :175-188, all 32 modalities
            self.declare(store, "target_modality_embedding_matrix", (32, d))
        self.declare(store, "LayerNorm/gamma", (d,), ones_initializer())      # after the last layer (:309-310)
        self.declare(store, "LayerNorm/beta", (d,), zeros_initializer())

    # -- forward -------------------------------------------------------------------------------------
    @tensor
    def _activations(self, ctx):
        train = bool(ctx.fed(self.train_mode))
        keep, att_keep = self.dropout_keep_prob, self.attention_dropout_keep_prob
        x_raw = self.input_sequence.temporal_states(ctx)                     # [B,S,D]
        mask = self.input_sequence.temporal_mask(ctx)                        # [B,S] float
        bsz, slen, d = x_raw.shape
        tape = F.Tape(ctx, (id(self), "tenc"), recording=ctx.wants_backward(train))
        x_in = tape.leaf(x_raw.reshape(bsz * slen, d), needs_grad=True)
        x = x_in
        if self.target_space_id is not None:                                  # :202-203
            table = tape.param(self, "target_modality_embedding_matrix")
            x = F.add_row(tape, x, tape.rows(table, self.target_space_id, self.target_space_id + 1))
        if self.use_positional_encoding:                                      # :205-209
            x = F.add_position(tape, x, TB.signal_table(ctx, d, slen), bsz, slen)
        x = F.dropout(tape, x, keep, train, ctx.salt(self.name, "encoder_inputs"))
        cross = None
        if self.input_for_cross_attention is not None:
            cs = get_attention_states(self.input_for_cross_attention, ctx)
            cross = (tape.leaf(cs.reshape(cs.shape[0] * cs.shape[1], cs.shape[2]), needs_grad=True),
                     get_attention_mask(self.input_for_cross_attention, ctx), cs.shape[1])
        for i in range(self.depth):
            pre = "layer_{}".format(i)
            site = (self.name, pre)
            # self-attention sublayer (:199-222)
            normed = TB.layer_norm(tape, self, pre + "/self_attention", x)
            att = TB.multihead_attention(tape, self, pre + "/self_attention", normed, normed, mask, self.n_heads,
                                         bsz, slen, bsz, slen, False, att_keep, train,
                                         ctx.salt(*site, "self_attention_weights"), self.use_att_transform_bias)
            att = F.dropout(tape, att, keep, train, ctx.salt(*site, "self_attention"))
            x = F.add(tape, att, x)
            if cross is not None:                                             # :224-260
                cvar, cmask, clen = cross
                normed = TB.layer_norm(tape, self, pre + "/cross_attention", x)
                att = TB.multihead_attention(tape, self, pre + "/cross_attention", normed, cvar, cmask,
                                             self.n_cross_att_heads, bsz, slen, bsz, clen, False, att_keep, train,
                                             ctx.salt(*site, "cross_attention_weights"),
                                             self.use_att_transform_bias)
                att = F.dropout(tape, att, keep, train, ctx.salt(*site, "cross_attention"))
                x = F.add(tape, att, x)
            x = TB.feedforward_sublayer(tape, self, pre + "/feedforward", x, keep, train, site)
        x = F.layer_norm(tape, x, tape.param(self, "LayerNorm/gamma"), tape.param(self, "LayerNorm/beta"))  # :309-310
        out = F.time_sum(tape, x, bsz, slen)                                  # :170-172
        return {"tape": tape, "x_in": x_in, "states": x, "output": out, "cross": cross, "shape": (bsz, slen, d)}

    @tensor
    def temporal_states(self, ctx) -> torch.Tensor:
        act = self._activations(ctx)
        bsz, slen, d = act["shape"]
        return act["states"].data.view(bsz, slen, d)

    @tensor
    def temporal_mask(self, ctx) -> torch.Tensor:
        return self.input_sequence.temporal_mask(ctx)

    @tensor
    def output(self, ctx) -> torch.Tensor:
        return self._activations(ctx)["output"].data

    def backward(self, ctx, d_states: Optional[torch.Tensor], d_final: Optional[torch.Tensor]) -> None:
        act = self._activations(ctx)
        tape = act["tape"]
        if not tape.recording:
            raise RuntimeError("TransformerEncoder.backward needs a run with train_mode=True")
        bsz, slen, d = act["shape"]
        if d_states is not None:
            act["states"].grad = None
            F.ops.ew("copy", d_states.reshape(bsz * slen, d), None, tape.grad(act["states"]), accumulate=True)
        if d_final is not None:
            F.ops.ew("copy", d_final, None, tape.grad(act["output"]), accumulate=True)
        tape.backward()
        if act["x_in"].grad is not None and hasattr(self.input_sequence, "backward"):
            self.input_sequence.backward(ctx, act["x_in"].grad.view(bsz, slen, d))
        cross = act["cross"]
        if cross is not None and cross[0].grad is not None:
            cs = cross[0].grad
            ctx.defer_backward(self.input_for_cross_attention, cs.view(bsz, cross[2], -1), None)
