"""Attention combination strategies for multi-source decoders (mirror of
neuralmonkey/attention/combination.py; Libovicky & Helcl 2017).

``FlatMultiAttention`` (combination.py:106-323): every encoder's states are projected into one
space, one distribution is taken over *all* positions of all encoders (+ an optional sentinel
position), the context is the weighted sum of a second projection of the same positions.
``HierarchicalMultiAttention`` (:345-475): each child attention yields its own context, a second
attention over those few vectors (+ sentinel) mixes them.

MI355X mapping.  Both run on the autodiff tape (the contexts feed conditional-GRU / output
projections, gradients reach several encoders).  The per-position work re-uses the Bahdanau
kernels: the projections of all encoders are laid side by side in one [B, S_total, A] key buffer
and one value buffer *once per batch*, so one step is the fused energies kernel over S_total
positions, one ``nm_attn_softmax_fwd`` over the assembled energies, one batched GEMM for the context.
Under beam search a sentence's k hypotheses share its keys (row // k), which -- unlike the
reference's ``tf.tile`` broadcast trick (:289-299, correct only for batch size 1) -- is right for
any batch size.

Variables created inside ``attention()`` live in the *decoder's* step scope in the reference
(``tf.variable_scope(self.att_scope_name)`` under ``attention_decoder``, :254,396), the projections
and ``attn_v`` in the attention's own scope; the names below follow that split so TF checkpoints
map one to one.
"""
from typing import Any, List, Optional

import torch

from .. import autodiff as F
from .. import ops
from ..model.model_part import InitializerSpecs, ModelPart
from ..variables import zeros_initializer
from .base_attention import (Attendable, AttentionLoopState, BaseAttention, get_attention_mask,
                             get_attention_states)


def _encoder_dim(enc) -> int:
    return enc.dimension


class MultiAttention(BaseAttention):
    """combination.py:33-103."""
    tape_only = True                 # no hand-scheduled fast path: the decoder runs on the tape

    # pylint: disable=too-many-arguments
    def __init__(self, name: str, attention_state_size: int, share_attn_projections: bool = False,
                 use_sentinels: bool = False, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        BaseAttention.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.attention_state_size = attention_state_size
        self._share_projections = share_attn_projections
        self._use_sentinels = use_sentinels
        self.att_scope_name = "attention_{}".format(name)
        self.dropout_keep_prob = 1.0
        self._decoder = None
        self._rows_per_key = 1

    @property
    def attn_size(self) -> int:
        return self.attention_state_size

    @property
    def context_vector_size(self) -> int:
        return self.attention_state_size

    @property
    def state_size(self) -> int:
        return self.attention_state_size

    @property
    def rows_per_key(self) -> int:
        return self._rows_per_key

    @rows_per_key.setter
    def rows_per_key(self, k: int) -> None:
        self._rows_per_key = k
        for child in getattr(self, "attentions", []):
            child.rows_per_key = k

    def bind_query_size(self, size: int) -> None:
        if self.query_state_size is not None and self.query_state_size != size:
            raise ValueError("Attention '{}' is queried with two different state sizes ({} vs {})"
                             .format(self.name, self.query_state_size, size))
        self.query_state_size = size
        for child in getattr(self, "attentions", []):
            child.bind_query_size(size)

    def bind_decoder(self, decoder) -> None:
        """The decoder whose step scope owns the variables created inside ``attention()``."""
        if self._decoder is not None and self._decoder is not decoder:
            raise ValueError("Attention '{}' is used by two decoders ('{}' and '{}'): its step variables "
                             "belong to one decoder scope".format(self.name, self._decoder.name, decoder.name))
        self._decoder = decoder

    # -- names in the decoder's step scope --------------------------------------------------------
    def _step(self, local: str) -> str:
        return "attention_decoder/{}/{}".format(self.att_scope_name, local)

    def declare_variables(self, store) -> None:
        self.declare(store, "attn_v", (self.attention_state_size,))           # [1,1,A] in TF

    def _declare_vector_logit(self, dec, store, scope: str, vec_size: int) -> None:
        """_vector_logit (combination.py:74-103)."""
        a = self.attention_state_size
        pre = "{}_logit".format(scope)
        dec.declare(store, self._step(pre + "/vector_bias"), (1,), zeros_initializer())
        dec.declare(store, self._step(pre + "/vector_projection/kernel"), (vec_size, a))
        dec.declare(store, self._step(pre + "/vector_projection/bias"), (a,), zeros_initializer())
        if not self._share_projections:
            dec.declare(store, self._step(pre + "/vector_ctx_proj/kernel"), (vec_size, a))
            dec.declare(store, self._step(pre + "/vector_ctx_proj/bias"), (a,), zeros_initializer())

    def _declare_sentinel(self, dec, store) -> None:
        """_sentinel (:326-342): gate over [prev_state ; rnn_input]."""
        h = self.query_state_size
        dec.declare(store, self._step("sentinel/dense/kernel"), (h + dec.embedding_size, h))
        dec.declare(store, self._step("sentinel/dense/bias"), (h,), zeros_initializer())

    def declare_decoder_variables(self, dec, store) -> None:
        a = self.attention_state_size
        dec.declare(store, self._step("dense/kernel"), (self.query_state_size, a))
        dec.declare(store, self._step("dense/bias"), (a,), zeros_initializer())

    def tape_session(self, tape, train_mode: bool):
        raise NotImplementedError("Abstract method")

    def attention(self, ctx, query, decoder_prev_state, decoder_input, loop_state):
        raise NotImplementedError("'{}' runs through tape_session().step (general decoder path)"
                                  .format(type(self).__name__))


class _MultiSession:
    """Shared pieces of one decoding run of a combination attention on a tape."""

    def __init__(self, att: MultiAttention, tape: F.Tape):
        if att._decoder is None:                                      # pylint: disable=protected-access
            raise RuntimeError("Attention '{}' is not attached to a decoder".format(att.name))
        self.att, self.tape = att, tape
        self.dec = att._decoder                                       # pylint: disable=protected-access
        self.asz = att.attention_state_size
        self.v = tape.param(att, "attn_v")
        self.v_col = tape.view(self.v, lambda t: t.view(-1, 1))
        self.wd = self.sparam("dense/kernel")
        self.bd = self.sparam("dense/bias")

    def sparam(self, local: str) -> F.Var:
        return self.tape.param(self.dec, self.att._step(local))       # pylint: disable=protected-access

    def vector_logit(self, projected_state: F.Var, vector: F.Var, scope: str):
        """_vector_logit (combination.py:74-103) -> (projection for the context [R,A], logit [R,1])."""
        tape, att = self.tape, self.att
        pre = "{}_logit".format(scope)
        proj_logit = F.linear(tape, vector, self.sparam(pre + "/vector_projection/kernel"),
                              self.sparam(pre + "/vector_projection/bias"))
        if att._share_projections:                                    # pylint: disable=protected-access
            proj_ctx = proj_logit
        else:
            proj_ctx = F.linear(tape, vector, self.sparam(pre + "/vector_ctx_proj/kernel"),
                                self.sparam(pre + "/vector_ctx_proj/bias"))
        act = F.tanh(tape, F.add(tape, projected_state, proj_logit))
        logit = F.linear(tape, act, self.v_col, self.sparam(pre + "/vector_bias"))
        return proj_ctx, logit

    def sentinel(self, query: F.Var, prev_state: F.Var, rnn_input: F.Var) -> F.Var:
        """_sentinel (:326-342): sigmoid(dense([prev_state ; input])) * state."""
        tape = self.tape
        gate = F.sigmoid(tape, F.linear(tape, F.concat(tape, [prev_state, rnn_input]),
                                        self.sparam("sentinel/dense/kernel"), self.sparam("sentinel/dense/bias")))
        return F.mul(tape, gate, query)


class FlatMultiAttention(MultiAttention):
    """combination.py:106-323."""

    # pylint: disable=too-many-arguments
    def __init__(self, name: str, encoders: List[Attendable], attention_state_size: int,
                 share_attn_projections: bool = False, use_sentinels: bool = False, reuse: ModelPart = None,
                 save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        MultiAttention.__init__(self, name=name, attention_state_size=attention_state_size,
                                share_attn_projections=share_attn_projections, use_sentinels=use_sentinels,
                                reuse=reuse, save_checkpoint=save_checkpoint, load_checkpoint=load_checkpoint,
                                initializers=initializers)
        self._encoders = encoders

    def declare_variables(self, store) -> None:
        MultiAttention.declare_variables(self, store)
        a = self.attention_state_size
        scopes = ["logits_projections"] + ([] if self._share_projections else ["context_projections"])
        for scope in scopes:                                          # get_encoder_projections (:199-232)
            for i, enc in enumerate(self._encoders):
                self.declare(store, "{}/proj_matrix_{}".format(scope, i), (_encoder_dim(enc), a))
                self.declare(store, "{}/proj_bias_{}".format(scope, i), (a,), zeros_initializer())
        for i in range(len(self._encoders)):                          # encoder_attn_biases (:159-163)
            self.declare(store, "attn_bias_{}".format(i), (1,), zeros_initializer())

    def declare_decoder_variables(self, dec, store) -> None:
        MultiAttention.declare_decoder_variables(self, dec, store)
        if self._use_sentinels:
            self._declare_sentinel(dec, store)
            self._declare_vector_logit(dec, store, "sentinel", self.query_state_size)

    def _lengths(self, ctx) -> List[int]:
        return [get_attention_states(e, ctx).shape[1] for e in self._encoders]

    def initial_loop_state(self, ctx, rows: int, max_steps: int, precompute: bool = True) -> AttentionLoopState:
        length = sum(self._lengths(ctx)) + (1 if self._use_sentinels else 0)
        return AttentionLoopState(
            contexts=ctx.buffer((id(self), "contexts", rows, max_steps), (max_steps, rows, self.context_vector_size)),
            weights=ctx.buffer((id(self), "weights", rows, max_steps, length), (max_steps, rows, length)),
            step=0)

    def finalize_loop(self, key: str, last_loop_state: AttentionLoopState) -> None:
        self.histories[key] = last_loop_state.weights[:last_loop_state.step]

    def tape_session(self, tape, train_mode: bool) -> "FlatSession":
        return FlatSession(self, tape)


class FlatSession(_MultiSession):
    """Setup once per run: both projections of every encoder, laid out over one position axis."""

    def __init__(self, att: FlatMultiAttention, tape: F.Tape):
        _MultiSession.__init__(self, att, tape)
        ctx = tape.ctx
        encs = att._encoders                                           # pylint: disable=protected-access
        share = att._share_projections                                 # pylint: disable=protected-access
        a = self.asz
        raws = [get_attention_states(e, ctx) for e in encs]
        self.bsz = raws[0].shape[0]
        self.lens = [r.shape[1] for r in raws]
        self.stot = sum(self.lens)
        self.width = self.stot + (1 if att._use_sentinels else 0)      # pylint: disable=protected-access
        b = self.bsz
        self.states_in = [tape.leaf(r.reshape(b * r.shape[1], r.shape[2]), needs_grad=True) for r in raws]
        self.encoders = list(encs)
        # keys / values of all encoders side by side: [B, S_total, A] as [B, S_total*A] column blocks
        self.keys = tape.new((b * self.stot, a))
        self.vals = self.keys if share else tape.new((b * self.stot, a))
        keys2 = tape.view(self.keys, lambda t: t.view(b, self.stot * a))
        vals2 = keys2 if share else tape.view(self.vals, lambda t: t.view(b, self.stot * a))
        mask = ctx.buffer((id(att), "mask", b, self.width), (b, self.width))
        ops.fill(mask, 1.0)
        off = 0
        for i, (enc, st, slen) in enumerate(zip(encs, self.states_in, self.lens)):
            for scope, dst in [("logits_projections", keys2)] + ([] if share else [("context_projections", vals2)]):
                proj = F.linear(tape, st, tape.param(att, "{}/proj_matrix_{}".format(scope, i)),
                                tape.param(att, "{}/proj_bias_{}".format(scope, i)))
                F.copy(tape, tape.view(proj, lambda t, n=slen: t.view(b, n * a)),
                       out=tape.cols(dst, off * a, (off + slen) * a))
            m = get_attention_mask(enc, ctx)
            if m is not None:
                ops.ew("copy", m.reshape(b, slen), None, mask[:, off:off + slen])
            off += slen
        self.mask = mask
        # per-position bias row p[1, W] = [bias_0]*S_0 ++ [bias_1]*S_1 ++ ... (++ 0 for the sentinel, whose
        # own vector_bias is part of its logit) as biases[1,n] . segments[n,W]
        n = len(encs)
        seg = ctx.buffer((id(att), "segments", tuple(self.lens), self.width), (n, self.width), zero=True)
        off = 0
        for i, slen in enumerate(self.lens):
            ops.fill(seg[i, off:off + slen], 1.0)
            off += slen
        biases = F.concat(tape, [tape.view(tape.param(att, "attn_bias_{}".format(i)), lambda t: t.view(1, 1))
                                 for i in range(n)])
        self.pos_bias = F.linear(tape, biases, tape.leaf(seg))
        self._ones = {}

    @property
    def shape_key(self):
        return (self.bsz, tuple(self.lens), self.width)

    def encoder_grads(self):
        out = []
        for enc, st, slen in zip(self.encoders, self.states_in, self.lens):
            if st.grad is not None:
                out.append((enc, st.grad.view(self.bsz, slen, -1)))
        return out

    def _ones_col(self, rows: int) -> F.Var:
        if rows not in self._ones:
            buf = self.tape.ctx.buffer((id(self.att), "ones", rows), (rows, 1))
            ops.fill(buf, 1.0)
            self._ones[rows] = self.tape.leaf(buf)
        return self._ones[rows]

    def step(self, query: F.Var, w_out: Optional[torch.Tensor] = None, prev_state: F.Var = None,
             rnn_input: F.Var = None) -> F.Var:
        """combination.py:245-296."""
        tape, att = self.tape, self.att
        rows = query.shape[0]
        k = att.rows_per_key
        projected = F.linear(tape, query, self.wd, self.bd)
        e_enc = F.attn_energies(tape, projected, self.keys, self.v, self.bsz, self.stot, k)
        e = tape.new((rows, self.width))
        F.copy(tape, e_enc, out=tape.cols(e, 0, self.stot))
        sent_ctx = None
        if att._use_sentinels:                                         # pylint: disable=protected-access
            value = self.sentinel(query, prev_state, rnn_input)
            sent_ctx, sent_logit = self.vector_logit(projected, value, "sentinel")
            F.copy(tape, sent_logit, out=tape.cols(e, self.stot, self.width))
        F.linear(tape, self._ones_col(rows), self.pos_bias, out=e, accumulate=True)
        w = F.attn_softmax(tape, e, self.mask, self.bsz, k, w_out)
        out = F.weighted_sum(tape, w, self.vals, self.bsz, self.stot, k)
        if sent_ctx is not None:
            F.rowscale(tape, sent_ctx, tape.cols(w, self.stot, self.width), out=out, accumulate=True)
        return out


class HierarchicalMultiAttention(MultiAttention):
    """combination.py:345-475."""

    # pylint: disable=too-many-arguments
    def __init__(self, name: str, attentions: List[BaseAttention], attention_state_size: int,
                 use_sentinels: bool, share_attn_projections: bool, reuse: ModelPart = None,
                 save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        MultiAttention.__init__(self, name=name, attention_state_size=attention_state_size,
                                use_sentinels=use_sentinels, share_attn_projections=share_attn_projections,
                                reuse=reuse, save_checkpoint=save_checkpoint, load_checkpoint=load_checkpoint,
                                initializers=initializers)
        self.attentions = attentions

    def bind_decoder(self, decoder) -> None:
        MultiAttention.bind_decoder(self, decoder)
        for child in self.attentions:
            if hasattr(child, "bind_decoder"):
                child.bind_decoder(decoder)

    def declare_decoder_variables(self, dec, store) -> None:
        MultiAttention.declare_decoder_variables(self, dec, store)
        a = self.attention_state_size
        for child in self.attentions:
            if hasattr(child, "declare_decoder_variables"):
                child.declare_decoder_variables(dec, store)
            self._declare_vector_logit(dec, store, child.name, child.context_vector_size)
            if not self._share_projections:                            # proj_attn_<name> (:426-432)
                dec.declare(store, self._step("proj_attn_{}/kernel".format(child.name)),
                            (child.context_vector_size, a))
                dec.declare(store, self._step("proj_attn_{}/bias".format(child.name)), (a,), zeros_initializer())
        if self._use_sentinels:
            self._declare_sentinel(dec, store)
            self._declare_vector_logit(dec, store, "sentinel", self.query_state_size)
            if not self._share_projections:
                dec.declare(store, self._step("proj_sentinel/kernel"), (self.query_state_size, a))
                dec.declare(store, self._step("proj_sentinel/bias"), (a,), zeros_initializer())

    def initial_loop_state(self, ctx, rows: int, max_steps: int, precompute: bool = True):
        length = len(self.attentions) + (1 if self._use_sentinels else 0)
        own = AttentionLoopState(
            contexts=ctx.buffer((id(self), "contexts", rows, max_steps), (max_steps, rows, self.context_vector_size)),
            weights=ctx.buffer((id(self), "weights", rows, max_steps, length), (max_steps, rows, length)),
            step=0)
        self._child_states = [c.initial_loop_state(ctx, rows, max_steps, precompute=False)
                              for c in self.attentions]
        return own

    def finalize_loop(self, key: str, last_loop_state: Any) -> None:
        for child, st in zip(self.attentions, getattr(self, "_child_states", [])):
            child.finalize_loop(key, AttentionLoopState(st.contexts, st.weights, last_loop_state.step))
        self.histories[key] = last_loop_state.weights[:last_loop_state.step]

    def tape_session(self, tape, train_mode: bool) -> "HierarchicalSession":
        return HierarchicalSession(self, tape, train_mode)


class HierarchicalSession(_MultiSession):
    def __init__(self, att: HierarchicalMultiAttention, tape: F.Tape, train_mode: bool):
        _MultiSession.__init__(self, att, tape)
        self.children = [c.tape_session(tape, train_mode) for c in att.attentions]
        self.t = 0

    @property
    def shape_key(self):
        return tuple(child.shape_key for child in self.children)

    def encoder_grads(self):
        out = []
        for child in self.children:
            out.extend(child.encoder_grads())
        return out

    def step(self, query: F.Var, w_out: Optional[torch.Tensor] = None, prev_state: F.Var = None,
             rnn_input: F.Var = None) -> F.Var:
        """combination.py:389-457."""
        tape, att = self.tape, self.att
        rows = query.shape[0]
        share = att._share_projections                                 # pylint: disable=protected-access
        projected = F.linear(tape, query, self.wd, self.bd)
        child_states = getattr(att, "_child_states", None)
        vectors, names = [], []
        for i, (child, sess) in enumerate(zip(att.attentions, self.children)):
            cw = None
            if child_states is not None and child_states[i].weights.shape[1] == rows \
                    and self.t < child_states[i].weights.shape[0]:
                cw = child_states[i].weights[self.t]
            vectors.append(sess.step(query, cw, prev_state=prev_state, rnn_input=rnn_input))
            names.append(child.name)
        if att._use_sentinels:                                         # pylint: disable=protected-access
            vectors.append(self.sentinel(query, prev_state, rnn_input))
            names.append("sentinel")
        n = len(vectors)
        logits = tape.new((rows, n))
        proj_ctxs = []
        for i, (vec, name) in enumerate(zip(vectors, names)):
            pc, logit = self.vector_logit(projected, vec, name)
            proj_ctxs.append(pc)
            F.copy(tape, logit, out=tape.cols(logits, i, i + 1))
        distr = F.attn_softmax(tape, logits, None, rows, 1, w_out)     # plain softmax (:421)
        if share:
            outputs = proj_ctxs
        else:
            outputs = []
            for vec, name in zip(vectors, names):
                scope = "proj_sentinel" if name == "sentinel" else "proj_attn_{}".format(name)
                outputs.append(F.linear(tape, vec, self.sparam(scope + "/kernel"), self.sparam(scope + "/bias")))
        out = None
        for i, vec in enumerate(outputs):
            out = F.rowscale(tape, vec, tape.cols(distr, i, i + 1), out=out, accumulate=out is not None)
        self.t += 1
        return out
