"""Bahdanau feed-forward attention (mirror of neuralmonkey/attention/feed_forward.py).

hidden_features = states . Wk          once per batch  (feed_forward.py:105-118)  -> MFMA GEMM
attention()     = q . Wq + b, then the fused HIP step kernel nm_attn_fwd
                  (energies, softmax, mask-renorm, context; feed_forward.py:120-166).
Keys are indexed by ``row // rows_per_key`` so a beam of k hypotheses per
sentence shares one copy of the keys."""
import contextlib
import os
from typing import Optional, Tuple

import torch

from .. import ops
from ..model.model_part import InitializerSpecs, ModelPart
from ..nn.dropout import dropout
from ..runtime import tensor
from ..variables import zeros_initializer
from .base_attention import (Attendable, AttentionLoopState, BaseAttention, get_attention_mask,
                             get_attention_states)


ENERGY_STACK = 64        # steps of a taped loop whose energies' gradients are stacked for one key-side backward call
STEP_BWD_FUSED = os.environ.get("NM_ATTN_STEP_BWD", "1") != "0"        # a taped step's backward up to the query: one launch


class Attention(BaseAttention):
    # pylint: disable=too-many-arguments
    def __init__(self, name: str, encoder: Attendable, dropout_keep_prob: float = 1.0,
                 state_size: int = None, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        BaseAttention.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.encoder = encoder
        self.dropout_keep_prob = dropout_keep_prob
        self._state_size = state_size
        self.rows_per_key = 1         # set by a beam-search decoder for the duration of its run

    @property
    def context_vector_size(self) -> int:
        return self.encoder.dimension

    @property
    def state_size(self) -> int:
        return self._state_size if self._state_size is not None else self.context_vector_size

    def bind_query_size(self, size: int) -> None:
        """The reference learns the query size when the decoder first calls
        ``attention`` (feed_forward.py:130); sizes must be static here."""
        if self.query_state_size is not None and self.query_state_size != size:
            raise ValueError("Attention '{}' is queried with two different state sizes ({} vs {})"
                             .format(self.name, self.query_state_size, size))
        self.query_state_size = size

    def declare_variables(self, store) -> None:
        if self.query_state_size is None:
            raise RuntimeError("Attention '{}' is not attached to a decoder".format(self.name))
        self.declare(store, "Attention/attn_query_projection", (self.query_state_size, self.state_size))
        self.declare(store, "attn_key_projection", (self.context_vector_size, self.state_size))
        self.declare(store, "attn_similarity_v", (self.state_size,))
        self.declare(store, "attn_projection_bias", (self.state_size,), zeros_initializer())
        self.declare(store, "attn_bias", (1,), zeros_initializer())      # scalar, kept as [1] on device

    @tensor
    def attention_states(self, ctx) -> torch.Tensor:
        return dropout(ctx, get_attention_states(self.encoder, ctx), self.dropout_keep_prob,
                       ctx.fed(self.train_mode))

    @tensor
    def attention_mask(self, ctx) -> Optional[torch.Tensor]:
        return get_attention_mask(self.encoder, ctx)

    @tensor
    def hidden_features(self, ctx) -> torch.Tensor:
        """[B,S,A] = states . Wk, no bias (the reference's 1x1 conv)."""
        states = self.attention_states(ctx)
        bsz, slen, c = states.shape
        hf = ctx.buffer((id(self), "hf"), (bsz, slen, self.state_size))
        ops.gemm(states.reshape(bsz * slen, c), self.var(ctx, "attn_key_projection"),
                 out=hf.view(bsz * slen, self.state_size))
        return hf

    def project_query(self, ctx, query: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        return ops.gemm(query, self.var(ctx, "Attention/attn_query_projection"), out=out,
                        bias=self.var(ctx, "attn_projection_bias"))

    def attention_into(self, ctx, query: torch.Tensor, y: torch.Tensor, ctx_out: torch.Tensor,
                       w_out: Optional[torch.Tensor], energies_out: Optional[torch.Tensor] = None) -> None:
        """y = q.Wq + b (kept by the caller for the backward pass), then the fused step kernel."""
        rows = query.shape[0]
        states = self.attention_states(ctx)
        hf = self.hidden_features(ctx)
        self.project_query(ctx, query, y)
        ws = ctx.buffer((id(self), "ws", rows), ((ops._lib.load().nm_attn_workspace_bytes(
            rows, states.shape[1], states.shape[2]) + 3) // 4,), zero_init=True)
        ops.attn_fwd(y, hf, states, self.attention_mask(ctx), self.var(ctx, "attn_similarity_v"),
                     self.var(ctx, "attn_bias"), self.rows_per_key, ctx_out, w_out, ws, energies_out)

    def partials_plan(self, ctx, rows: int):
        """Can a step of ``rows`` queries leave its split-S partials for a fused consumer (``nm_step_group``)?
        Returns None or a dict with the workspace and the views a consumer needs."""
        if self.dropout_keep_prob != 1.0 and bool(ctx.fed(self.train_mode)):
            return None
        states = self.attention_states(ctx)
        bk, slen, c = states.shape
        a = self.state_size
        lay = ops.attn_partials_layout(rows, slen, a, c)
        if lay is None or lay[0] > 8 or c % 16 or a % 4 or rows % self.rows_per_key or rows // self.rows_per_key != bk:
            return None
        nchunk, pctx_off, pstat_off = lay
        ws = ctx.buffer((id(self), "ws", rows), ((ops._lib.load().nm_attn_workspace_bytes(rows, slen, c) + 3) // 4,), zero_init=True)
        # the arrival counters of the in-kernel merge (workspace tail) must be zero when a step is launched; the
        # merging workgroup restores the zero, but an aborted launch or a failed capture would leave them counting
        # and every later merge silently skipped -- one small memset per decoding run makes that self-healing
        ops.zero(ws[-(((rows + 3) // 4) * 4):])
        return {"ws": ws, "nchunk": nchunk, "energies": ws[:rows * slen].view(rows, slen),
                "pctx": ws[pctx_off:pctx_off + rows * nchunk * c], "pstat": ws[pstat_off:pstat_off + rows * nchunk * 4],
                "S": slen, "C": c, "Bk": bk}

    def attention_partials(self, ctx, y: torch.Tensor, plan) -> None:
        """Energies and split-S partials of one step for projected queries ``y`` [R,A] (no combine launch)."""
        ops.attn_fwd_partials(y, self.hidden_features(ctx), self.attention_states(ctx), self.attention_mask(ctx),
                              self.var(ctx, "attn_similarity_v"), self.var(ctx, "attn_bias"), self.rows_per_key,
                              plan["ws"])

    def attention_all_steps(self, ctx, queries: torch.Tensor, y_all: torch.Tensor, ctx_all: torch.Tensor,
                            w_all: torch.Tensor, e_all: torch.Tensor) -> None:
        """Teacher-forced training: the attention of step t only needs the decoder
        state s_t (nothing of it feeds the recurrence), so all T steps run as one
        projection GEMM + one fused attention launch in which every sentence's
        keys are read from HBM once for all T queries.  queries [T,B,Q]."""
        steps, bsz, qdim = queries.shape
        states = self.attention_states(ctx)
        hf = self.hidden_features(ctx)
        self.project_query(ctx, queries.reshape(steps * bsz, qdim), y_all.view(steps * bsz, -1))
        ws = ctx.buffer((id(self), "ws_all", steps, bsz), ((ops._lib.load().nm_attn_workspace_bytes(
            steps * bsz, states.shape[1], states.shape[2]) + 3) // 4,), zero_init=True)
        ops.attn_fwd_time_major(y_all, hf, states, self.attention_mask(ctx), self.var(ctx, "attn_similarity_v"),
                                self.var(ctx, "attn_bias"), ctx_all, w_all, ws, e_all)

    def backward(self, ctx, dctx_all: torch.Tensor, queries: torch.Tensor, y_all: torch.Tensor,
                 w_all: torch.Tensor, e_all: torch.Tensor, dquery_accum: torch.Tensor) -> torch.Tensor:
        """Gradient of T attention steps at once (nothing here feeds the recurrence).

        dctx_all [T,B,C], queries [T,B,Q], y_all [T,B,A], w_all / e_all [T,B,S].
        Adds dL/dquery into ``dquery_accum`` [T*B,Q], accumulates this part's
        variable gradients and returns dL/d(attention_states) [B,S,C]."""
        store = ctx.store
        states, hf, mask = self.attention_states(ctx), self.hidden_features(ctx), self.attention_mask(ctx)
        bsz, slen, c = states.shape
        steps = dctx_all.shape[0]
        a = self.state_size
        rows = steps * bsz
        key = (id(self), "bwd")
        dw = ctx.buffer(key + ("dw",), (steps, bsz, slen))
        dstates = ctx.buffer(key + ("dstates",), (bsz, slen, c))
        dctx_b = dctx_all.permute(1, 0, 2)                                   # [B,T,C] strided view
        ops.gemm(dctx_b, states, out=dw.permute(1, 0, 2), trans_b=True)      # dw[t,b,:] = dctx[t,b,:].states[b]^T
        ops.gemm(w_all.permute(1, 0, 2), dctx_b, out=dstates, trans_a=True)  # dstates[b] = w[:,b,:]^T.dctx[:,b,:]
        de = ctx.buffer(key + ("de",), (steps, bsz, slen))
        ops.attn_softmax_bwd(dw, e_all, mask, de, bsz)
        ops.reduce_sum(de.view(-1), ctx.buffer(key + ("dbias",), (1,)))
        g_bias = store.g(self.var_name("attn_bias")).view(1)
        ops.ew("add", g_bias, ctx.buffer(key + ("dbias",), (1,)), g_bias)
        dhf = ctx.buffer(key + ("dhf",), (bsz, slen, a))
        dvp = ctx.buffer(key + ("dvp",), (bsz * slen, a))
        dy = ctx.buffer(key + ("dy",), (steps, bsz, a))
        ops.attn_energy_bwd(de, hf, y_all, self.var(ctx, "attn_similarity_v"), dhf, dvp, dy)
        dy2 = dy.view(rows, a)
        wq = self.var(ctx, "Attention/attn_query_projection")
        ops.gemm(dy2, wq, out=dquery_accum, trans_b=True, accumulate=True)
        wk = self.var(ctx, "attn_key_projection")
        ops.gemm(dhf.view(bsz * slen, a), wk, out=dstates.view(bsz * slen, c), trans_b=True, accumulate=True)
        loops = getattr(self.encoder, "has_time_loop", False)      # see Decoder.backward: side streams pay beside loops
        bg = ctx.session.leaf_algo() if loops else 0
        with (ctx.session.side() if loops else contextlib.nullcontext()):   # leaf gradients of this part's variables
            ops.colsum(dvp, store.g(self.var_name("attn_similarity_v")), accumulate=True)
            ops.colsum(dy2, store.g(self.var_name("attn_projection_bias")), accumulate=True)
            ops.gemm(queries.reshape(rows, -1), dy2,
                     out=store.g(self.var_name("Attention/attn_query_projection")), trans_a=True,
                     accumulate=True, algo=bg)
            ops.gemm(states.reshape(bsz * slen, c), dhf.view(bsz * slen, a),
                     out=store.g(self.var_name("attn_key_projection")), trans_a=True, accumulate=True, algo=bg)
        return dstates

    def tape_session(self, tape, train_mode: bool) -> "AttentionTapeSession":
        """The general (taped) path: keys and per-step attention as differentiable tape ops."""
        return AttentionTapeSession(self, tape, train_mode)

    def attention(self, ctx, query: torch.Tensor, decoder_prev_state, decoder_input,
                  loop_state: AttentionLoopState) -> Tuple[torch.Tensor, AttentionLoopState]:
        rows = query.shape[0]
        step = loop_state.step
        y = ctx.buffer((id(self), "y", rows), (rows, self.state_size))
        self.attention_into(ctx, query, y, loop_state.contexts[step], loop_state.weights[step])
        return loop_state.contexts[step], AttentionLoopState(loop_state.contexts, loop_state.weights,
                                                             step + 1)

    def initial_loop_state(self, ctx, rows: int, max_steps: int, precompute: bool = True) -> AttentionLoopState:
        states = get_attention_states(self.encoder, ctx)
        if precompute:
            self.hidden_features(ctx)    # pre-compute outside the loop (feed_forward.py:168-186)
        return AttentionLoopState(
            contexts=ctx.buffer((id(self), "contexts", rows, max_steps),
                                (max_steps, rows, self.context_vector_size)),
            weights=ctx.buffer((id(self), "weights", rows, max_steps), (max_steps, rows, states.shape[1])),
            step=0)

    def finalize_loop(self, key: str, last_loop_state: AttentionLoopState) -> None:
        self.histories[key] = last_loop_state.weights[:last_loop_state.step]


class AttentionTapeSession:
    """One decoding run of an ``Attention`` on an autodiff tape (general path: conditional GRU,
    attention on input, dropout, NematusGRU / LSTM decoders -- anything where the context feeds
    the recurrence, so the attention gradient has to be taken step by step).

    Setup (once per run): attention_states = dropout(states) (feed_forward.py:47-51) and the
    keys states.Wk (:105-118) as tape ops, so their gradients -- dL/d(encoder states) and the
    key-projection gradient -- fall out of ``Tape.backward``.  ``step`` is the fused kernel
    ``nm_attn_fwd``; its gradient closure runs the same kernels as ``Attention.backward`` with T=1
    and accumulates into the keys' gradient."""

    def __init__(self, att: Attention, tape, train_mode: bool):
        from .. import autodiff as F
        ctx = tape.ctx
        self.att, self.tape = att, tape
        raw = get_attention_states(att.encoder, ctx)
        self.bsz, self.slen, self.csz = raw.shape
        self.asz = att.state_size
        self.states_in = tape.leaf(raw.reshape(self.bsz * self.slen, self.csz), needs_grad=True)
        self.states = F.dropout(tape, self.states_in, att.dropout_keep_prob, train_mode,
                                ctx.salt(att.name, "attention_states"))
        self.hf = F.linear(tape, self.states, tape.param(att, "attn_key_projection"))
        # the states' gradient through the context sums: one rank-1 update per step (w_t ^T dctx_t, a batched product on
        # 128x128 tiles with M = S, K = 1: 40 us) -- collected by the steps' closures and summed in ONE launch by the
        # closure recorded here, which runs after all of them and before the closures of the two operations above read
        # that gradient (nm_outer_chain: the steps are its K dimension)
        self._outer = []

        def flush_outer():
            pairs, self._outer = self._outer, []
            if pairs:
                gs = tape.grad(self.states).view(self.bsz, self.slen, self.csz)
                for i in range(0, len(pairs), ops.OUTER_CHAIN_MAX):
                    ops.outer_chain(pairs[i:i + ops.OUTER_CHAIN_MAX], gs, accumulate=True)
        tape.record(flush_outer)
        # ... and the key-side sums of the energies' backward (dhf, the partial sums of dv): a step needs the QUERY
        # gradient at once (it flows back into the recurrence), the sums over the steps only when the pass has been
        # through all of them -- one call over the stacked energies' gradients and queries of all steps (the kernel keeps
        # a sentence's keys in registers while it walks the T queries) instead of 156 MB of read-modify-write per step
        self._steps = 0
        self._stack = None          # (de [cap, B, S], y [cap, B, A]) of the steps, allocated by the first one
        self._deferred_steps = []

        def flush_energies():
            idx, self._deferred_steps = self._deferred_steps, []
            if idx:
                n = max(idx) + 1
                de_all, y_all = self._stack
                for t in sorted(set(range(n)) - set(idx)):       # a step whose context nobody differentiated: no term
                    ops.zero(de_all[t])
                dbias = tape.buf((1,))                           # the scalar bias: the sum of all the energies' gradients
                ops.reduce_sum(de_all[:n].reshape(-1), dbias)
                ops.ew("copy", dbias, None, self.bias.grad, accumulate=True)
                # (``finish`` below was recorded later and has run by now: these steps' partial sums of dv are summed here)
                dvp = tape.buf((self.bsz * self.slen, self.asz), zero=True)
                scratch = tape.buf((n, self.bsz, self.asz))
                ops.attn_energy_bwd(de_all[:n], self.hf.data.view(self.bsz, self.slen, self.asz), y_all[:n], self.v.data,
                                    tape.grad(self.hf).view(self.bsz, self.slen, self.asz), dvp, scratch, accumulate=True)
                ops.colsum(dvp, self.v.grad, accumulate=True)
        tape.record(flush_energies)
        self.mask = att.attention_mask(ctx)
        self.wq = tape.param(att, "Attention/attn_query_projection")
        self.bq = tape.param(att, "attn_projection_bias")
        self.v = tape.param(att, "attn_similarity_v")
        self.bias = tape.param(att, "attn_bias")
        self._dvp = None

        def finish():           # recorded first => runs after every step's gradient closure
            if self._dvp is not None:
                ops.colsum(self._dvp, self.v.grad, accumulate=True)
        tape.record(finish)

    @property
    def d_states(self) -> Optional[torch.Tensor]:
        """dL/d(encoder states) [B,S,C] after ``Tape.backward``."""
        g = self.states_in.grad
        return None if g is None else g.view(self.bsz, self.slen, self.csz)

    @property
    def shape_key(self):
        return (self.bsz, self.slen, self.csz, self.asz)

    def encoder_grads(self):
        """[(encoder, dL/d states [B,S,C])] after ``Tape.backward``."""
        g = self.d_states
        return [] if g is None else [(self.att.encoder, g)]

    def step(self, query, w_out: Optional[torch.Tensor] = None, prev_state=None, rnn_input=None, out=None):
        """query Var [R,Q] -> context Var [R,C]; ``w_out`` [R,S] receives the weights.  (The previous
        decoder state and the RNN input of the reference's signature only matter to sentinels.)"""
        from .. import autodiff as F
        tape, att = self.tape, self.att
        ctx = tape.ctx
        b, s, c, a = self.bsz, self.slen, self.csz, self.asz
        rows = query.shape[0]
        slot = None                                      # this step's row of the stacks (training, one query per sentence)
        if tape.recording and F.CHAIN_WGRADS and rows == b and query.data.is_cuda and self._steps < ENERGY_STACK:
            if self._stack is None:
                self._stack = (tape.buf((ENERGY_STACK, b, s)), tape.buf((ENERGY_STACK, b, a)))
            slot = self._steps
        self._steps += 1
        y = F.linear(tape, query, self.wq, self.bq,
                     out=None if slot is None else F.Var(self._stack[1][slot], None, True))
        if out is None:                                  # (else: the step's rows of a buffer of all steps' contexts)
            out = tape.new((rows, c))
        w = w_out if w_out is not None else tape.buf((rows, s))
        e = tape.buf((rows, s)) if tape.recording else None
        ws = ctx.buffer((id(att), "ws", rows), ((ops._lib.load().nm_attn_workspace_bytes(rows, s, c) + 3) // 4,), zero_init=True)
        hf3, st3 = self.hf.data.view(b, s, a), self.states.data.view(b, s, c)
        ops.attn_fwd(y.data, hf3, st3, self.mask, self.v.data, self.bias.data, att.rows_per_key, out.data, w, ws, e)

        def bwd():
            dctx = out.grad
            if dctx is None:
                return
            assert rows == b, "the attention gradient is defined for one query per sentence"
            fused = (STEP_BWD_FUSED and slot is not None and ops.attn_step_bwd_ok(dctx, c))
            if not fused:
                dw = tape.buf((b, 1, s))
                ops.gemm(dctx.view(b, 1, c), st3, out=dw, trans_b=True)
            if self.states.needs_grad:
                if F.CHAIN_WGRADS and dctx.is_cuda and s * ops.OUTER_CHAIN_MAX * 4 <= 65536 and b < 65536:
                    self._outer.append((w, dctx))
                else:
                    ops.gemm(w.view(b, 1, s), dctx.view(b, 1, c), out=tape.grad(self.states).view(b, s, c),
                             trans_a=True, accumulate=True)
            de = tape.buf((1, b, s)) if slot is None else self._stack[0][slot:slot + 1]
            if fused:                    # the context sum's weights, the softmax part and the query gradient: one launch
                ops.attn_step_bwd(dctx, st3, e, self.mask, hf3, y.data, self.v.data, de[0], tape.grad(y))
                self._deferred_steps.append(slot)
                return
            ops.attn_softmax_bwd(dw.view(1, b, s), e.view(1, b, s), self.mask, de, b)
            if slot is None:             # (a stacked step's share of the bias gradient: one sum in flush_energies)
                dbias = tape.buf((1,))
                ops.reduce_sum(de.view(-1), dbias)
                ops.ew("copy", dbias, None, self.bias.grad, accumulate=True)
            if slot is not None:         # the query gradient now, the key-side sums in flush_energies
                ops.attn_energy_bwd(de, hf3, y.data.view(1, b, a), self.v.data, None, None, tape.grad(y).view(1, b, a))
                self._deferred_steps.append(slot)
                return
            if self._dvp is None:
                self._dvp = tape.buf((b * s, a), zero=True)
            ops.attn_energy_bwd(de, hf3, y.data.view(1, b, a), self.v.data, tape.grad(self.hf).view(b, s, a),
                                self._dvp, tape.grad(y).view(1, b, a), accumulate=True)
        tape.record(bwd)
        return out
