"""Coverage attention, Tu et al. 2016 (mirror of neuralmonkey/attention/coverage.py:19-66).

The Bahdanau attention of ``feed_forward.Attention`` whose energies also see how much attention every source
position has received so far, scaled by a learned per-position fertility:

    fertility[b,s] = 1e-8 + max_fertility * sigmoid(sum_c fertility_matrix[c] * states[b,s,c])        (:47-50)
    coverage[b,s]  = (sum_{t' < t} weights_t'[b,s]) / fertility[b,s] * mask[b,s]                       (:53-57)
    e[b,s]         = sum_a v[a] * tanh(hf[b,s,a] + y[b,a] + coverage_matrix[a] * coverage[b,s])        (:58-64)

(no ``attn_bias``: the reference's ``get_energies`` override never touches ``bias_term``, so the variable does
not exist in its graph; softmax is shift invariant anyway).  Upstream line 52 asks a ``tf.Tensor`` for
``.size()`` and cannot build; the intent -- the sum over the weights of the previous steps, zero at the first
step -- is what the loop state it is handed (``loop_state.weights`` [t,B,S], feed_forward.py:158-159) holds.

MI355X mapping.  The context of step t depends on the weights of all earlier steps, so the gradient is taken
step by step on the autodiff tape (``tape_only``): per step the position-dependent term is added to the keys
(a rank-1 update of the [R*S, A] key block: one K = 1 GEMM accumulated onto a copy), then the same fused
energies kernel, softmax / mask / renormalise kernel and batched context GEMM as the multi-source attentions
(attention/combination.py).  The running sum lives in ONE persistent buffer that every step adds its weights
to (a sum needs none of its inputs in the backward pass), so a chunk of steps can be replayed from a HIP graph.
Under beam search the k hypotheses of a sentence share keys and fertility (row // k) but each carries its own
coverage, which ``reorder`` gathers together with the decoder state -- the reference's loop state cannot do
that (its weights history is not part of the beam's gather), so coverage + beam search is only right there for
beam size 1.
"""
from typing import Optional

import torch

from .. import autodiff as F
from .. import ops
from ..model.model_part import InitializerSpecs, ModelPart
from ..variables import zeros_initializer
from .base_attention import Attendable, AttentionLoopState, get_attention_states
from .feed_forward import Attention


class CoverageAttention(Attention):
    """coverage.py:19-66."""
    tape_only = True                 # the decoder runs on the tape (decoders/decoder_general.py)

    # pylint: disable=too-many-arguments
    def __init__(self, name: str, encoder: Attendable, dropout_keep_prob: float = 1.0, state_size: int = None,
                 max_fertility: int = 5, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        Attention.__init__(self, name, encoder, dropout_keep_prob, state_size, reuse, save_checkpoint,
                           load_checkpoint, initializers)
        self.max_fertility = max_fertility

    def declare_variables(self, store) -> None:
        if self.query_state_size is None:
            raise RuntimeError("Attention '{}' is not attached to a decoder".format(self.name))
        a, c = self.state_size, self.context_vector_size
        self.declare(store, "Attention/attn_query_projection", (self.query_state_size, a))
        self.declare(store, "attn_key_projection", (c, a))
        self.declare(store, "attn_similarity_v", (a,))
        self.declare(store, "attn_projection_bias", (a,), zeros_initializer())
        self.declare(store, "coverage_matrix", (1, 1, 1, a))          # coverage.py:40-41
        self.declare(store, "fertility_matrix", (1, 1, c))            # :44-46

    def initial_loop_state(self, ctx, rows: int, max_steps: int, precompute: bool = True) -> AttentionLoopState:
        return Attention.initial_loop_state(self, ctx, rows, max_steps, precompute=False)

    def attention(self, ctx, query, decoder_prev_state, decoder_input, loop_state):
        raise NotImplementedError("CoverageAttention runs through tape_session().step (general decoder path)")

    def tape_session(self, tape, train_mode: bool) -> "CoverageSession":
        return CoverageSession(self, tape, train_mode)


class CoverageSession:
    """One decoding run on a tape.  Setup once per run: attention_states = dropout(states), keys = states . Wk
    (feed_forward.py:47-51,105-118), fertility (coverage.py:47-50); ``step`` is coverage.py:52-64 followed by
    feed_forward.py:139-154."""

    def __init__(self, att: CoverageAttention, tape: F.Tape, train_mode: bool):
        ctx = tape.ctx
        self.att, self.tape = att, tape
        raw = get_attention_states(att.encoder, ctx)
        b, s, c = raw.shape
        a = att.state_size
        k = att.rows_per_key
        self.bsz, self.slen, self.csz, self.asz, self.k = b, s, c, a, k
        self.rows = rows = b * k
        if k > 1 and tape.recording:
            raise RuntimeError("CoverageAttention: several queries per sentence exist at inference only")
        self.states_in = tape.leaf(raw.reshape(b * s, c), needs_grad=True)
        self.states = F.dropout(tape, self.states_in, att.dropout_keep_prob, train_mode,
                                ctx.salt(att.name, "attention_states"))
        hf = F.linear(tape, self.states, tape.param(att, "attn_key_projection"))
        self.mask = att.attention_mask(ctx)
        self.wq = tape.param(att, "Attention/attn_query_projection")
        self.bq = tape.param(att, "attn_projection_bias")
        self.v = tape.param(att, "attn_similarity_v")
        self.covw = tape.view(tape.param(att, "coverage_matrix"), lambda t: t.view(1, a))
        fert_w = tape.view(tape.param(att, "fertility_matrix"), lambda t: t.view(c, 1))
        logit = F.linear(tape, self.states, fert_w)                                     # [B*S, 1]
        fert = F.add_scalar(tape, F.scale(tape, F.sigmoid(tape, logit), float(att.max_fertility)), 1e-8)
        fert = tape.view(fert, lambda t: t.view(b, s))
        mask = None if self.mask is None else tape.leaf(self.mask)
        if k > 1:                   # every hypothesis row gets the keys / fertility / mask of its sentence
            src = torch.arange(rows, dtype=torch.int32, device=raw.device) // k

            def take(t2d):
                dst = tape.buf((rows, t2d.shape[1]))
                ops.gather_rows(t2d, src, dst)
                return tape.leaf(dst)
            hf = tape.view(take(hf.data.view(b, s * a)), lambda t: t.view(rows * s, a))
            fert = take(fert.data)
            mask = None if mask is None else take(mask.data)
        self.hf, self.fert, self.mask_rows = hf, fert, mask
        self.wsum = tape.new((rows, s))              # sum of the weights of the steps so far
        self._gathered = tape.buf((rows, s))
        self.t = 0

    @property
    def shape_key(self):
        return (self.bsz, self.slen, self.csz, self.asz, self.k)

    def encoder_grads(self):
        g = self.states_in.grad
        return [] if g is None else [(self.att.encoder, g.view(self.bsz, self.slen, self.csz))]

    def reorder(self, src_rows: torch.Tensor) -> None:
        """Beam search: hypothesis r continues hypothesis src_rows[r] -- its coverage comes along."""
        ops.gather_rows(self.wsum.data, src_rows, self._gathered)
        ops.ew("copy", self._gathered, None, self.wsum.data)

    def step(self, query: F.Var, w_out: Optional[torch.Tensor] = None, prev_state=None, rnn_input=None) -> F.Var:
        tape = self.tape
        rows, s, a = self.rows, self.slen, self.asz
        assert query.shape[0] == rows
        if self.t == 0:
            ops.zero(self.wsum.data)                   # weights_in_time is empty at the first step (:53-56)
        y = F.linear(tape, query, self.wq, self.bq)
        cov = F.div(tape, self.wsum, self.fert)                                          # :57
        if self.mask_rows is not None:
            cov = F.mul(tape, cov, self.mask_rows)
        # hidden_features + coverage_weights * coverage (:60-62): a rank-1 update of the key block
        keys = F.copy(tape, self.hf)
        F.linear(tape, tape.view(cov, lambda t: t.view(rows * s, 1)), self.covw, out=keys, accumulate=True)
        e = F.attn_energies(tape, y, keys, self.v, rows, s, 1)
        w = F.attn_softmax(tape, e, self.mask, self.bsz, self.k, w_out)                 # feed_forward.py:136-144
        out = F.weighted_sum(tape, w, self.states, self.bsz, s, self.k)                 # :151-154
        F.add_(tape, self.wsum, w)
        self.t += 1
        return out
