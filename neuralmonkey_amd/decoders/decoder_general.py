"""General (taped) path of the RNN decoder: every ``Decoder`` configuration outside the
hand-scheduled plain-GRU fast path of ``decoder.py``.

    rnn_cell = "NematusGRU" | "LSTM"                 decoders/decoder.py:253-254, 309-325
    conditional_gru (second cell on the contexts)    decoders/decoder.py:256-261, 303-307
    attention_on_input                               decoders/decoder.py:264-277
    dropout_keep_prob < 1 in training                decoders/decoder.py:235-240, 331-334,
                                                     autoregressive.py:269-272
    nematus / maxout / mlp output projections        decoders/output_projection.py:76-188

One step is written once (``general_step``) against ``autodiff`` and used by training (tape
recording, gradients by ``Tape.backward``) and by greedy / beam decoding (non-recording tape whose
step buffers are recycled).  The context vectors feed the recurrence in most of these
configurations, so nothing is hoisted out of the time loop except the embedding gather and the
vocabulary projection + cross entropy.
"""
import os
import weakref
from typing import List, Optional

import torch

from .. import autodiff as F
from .. import ops
from ..nn.cells import GRUCell, LSTMCell, NematusGRUCell, make_cell
from ..variables import zeros_initializer

HOIST_OUTPUT = os.environ.get("NM_HOIST_OUTPUT", "1") != "0"


class GeneralDecoderMixin:
    """Mixed into ``Decoder``; relies on its configuration attributes."""

    # -- configuration -------------------------------------------------------------------------
    def _build_cells(self) -> None:
        e, h = self.embedding_size, self.rnn_size
        self._cell_obj = make_cell(self._rnn_cell_str, self, "attention_decoder", e, h)
        self._cond_cell = None
        if self._conditional_gru and self._rnn_cell_str in ("GRU", "NematusGRU"):
            ctx_total = sum(a.context_vector_size for a in self.attentions)
            if ctx_total == 0:
                raise ValueError("conditional_gru needs at least one attention")
            if self._rnn_cell_str == "NematusGRU":      # decoders/decoder.py:256-261
                self._cond_cell = NematusGRUCell(self, "attention_decoder", ctx_total, h, use_state_bias=True,
                                                 use_input_bias=False, cell_scope="cond_gru_2_cell")
            else:
                self._cond_cell = GRUCell(self, "attention_decoder", ctx_total, h, cell_scope="cond_gru_2_cell")

    def _declare_general_variables(self, store) -> None:
        self._cell_obj.declare_variables(store)
        if self._cond_cell is not None:
            self._cond_cell.declare_variables(store)
        if self._attention_on_input:
            e = self.embedding_size
            width = e + sum(a.context_vector_size for a in self.attentions)
            self.declare(store, "attention_decoder/input_projection/kernel", (width, e))
            self.declare(store, "attention_decoder/input_projection/bias", (e,), zeros_initializer())
        for att in self.attentions:                  # variables the attention creates in the decoder's step scope
            if hasattr(att, "declare_decoder_variables"):
                att.declare_decoder_variables(self, store)

    def uses_general_path(self, train_mode: bool) -> bool:
        from .output_projection import NonlinearOutput
        if self._rnn_cell_str != "GRU" or self._cond_cell is not None or self._attention_on_input:
            return True
        if any(getattr(a, "tape_only", False) for a in self.attentions):
            return True                  # combination / dot-product attentions exist on the tape only
        if self.rnn_size % 4 != 0 or self.embedding_size % 4 != 0:
            return True                  # the fused gate kernels move float4s
        proj = self.output_projection
        if not isinstance(proj, NonlinearOutput) or proj.activation not in ("tanh", "identity"):
            return True
        if train_mode and not getattr(self.encoder_projection, "fast_path", True):
            return True                  # the projection has no hand-scheduled backward (nematus_projection)
        if train_mode:
            keeps = [self.dropout_keep_prob, proj.dropout_keep_prob, self.encoder_projection.dropout_keep_prob]
            keeps += [getattr(a, "dropout_keep_prob", 1.0) for a in self.attentions]
            if any(k != 1.0 for k in keeps):
                return True
        return False

    def graph_safe_training(self, train_mode: bool) -> bool:
        """May forward + backward of a training step be captured as one HIP graph?  Yes on the tape
        (kernel launches on persistent buffers, no host round trips); the hand-scheduled fast path
        keeps its own loop graphs and its side-stream overlap instead."""
        return self.uses_general_path(train_mode) and all(
            getattr(e, "graph_safe_training", lambda t: False)(train_mode) for e in self.encoders)

    def make_stepper(self, ctx, rows: int, tag: str, rows_per_key: int = 1, max_positions: int = 0):
        """Stepwise inference driver (greedy loop, beam search)."""
        return make_stepper(self, ctx, rows, tag)

    def state_sizes(self) -> List[int]:
        """Widths of the tensors carried between steps: RNNFeedables (decoders/decoder.py:34-50)."""
        return [self.rnn_size, self.rnn_size] + [a.context_vector_size for a in self.attentions]

    # -- one step --------------------------------------------------------------------------------
    def general_step(self, tape: F.Tape, emb_in: F.Var, state: List[F.Var], sessions, w_outs, train: bool,
                     t: int, project: bool = True, x_proj=None, out_views=None):
        """Decoder.next_state (decoders/decoder.py:279-358).  ``state`` = [prev_rnn_state,
        prev_rnn_output, *prev_contexts]; returns (output, new_state)."""
        ctx = tape.ctx
        keep = self.dropout_keep_prob
        prev_state, prev_out, prev_ctxs = state[0], state[1], state[2:]
        if self._attention_on_input:                                   # :264-277
            w = tape.param(self, "attention_decoder/input_projection/kernel")
            b = tape.param(self, "attention_decoder/input_projection/bias")
            x, row = None, 0
            for part in [emb_in] + list(prev_ctxs):
                sz = part.shape[1]
                x = F.linear(tape, part, tape.rows(w, row, row + sz), b if x is None else None, out=x,
                             accumulate=x is not None)
                row += sz
            rnn_input = F.dropout(tape, x, keep, train, ctx.salt(self.name, "input_projection", t))
        else:
            rnn_input = emb_in
        if isinstance(self._cell_obj, LSTMCell):                       # :309-325
            cell_output, (next_state, _) = self._cell_obj.step(tape, rnn_input, (prev_state, prev_out), x_proj=x_proj)
            contexts = [s.step(cell_output, w, prev_state=prev_out, rnn_input=rnn_input)
                        for s, w in zip(sessions, w_outs)]
        else:                                                          # :288-307
            # ``out_views`` (training without dropout): the step's rows of the buffers the output projection reads after
            # the loop -- the last cell and the attentions write there where they can (no copy launches)
            s_view, c_views = out_views if out_views is not None else (None, [None] * len(sessions))
            nem = isinstance(self._cell_obj, NematusGRUCell)
            first_out = s_view if (nem and self._cond_cell is None) else None
            kw = {"out": first_out} if nem else {}
            if x_proj is not None:       # the cell's input half of this step: rows of a product over all steps
                cell_output, (next_state,) = self._cell_obj.step(tape, None, (prev_out,), x_proj=x_proj, **kw)
            else:
                cell_output, (next_state,) = self._cell_obj.step(tape, rnn_input, (prev_out,), **kw)
            contexts = [s.step(cell_output, w, prev_state=prev_out, rnn_input=rnn_input,
                               **({"out": cv} if (cv is not None and hasattr(s, "_stack")) else {}))
                        for s, w, cv in zip(sessions, w_outs, c_views)]
            if self._cond_cell is not None:
                kw2 = {"out": s_view} if isinstance(self._cond_cell, NematusGRUCell) else {}
                cell_output, (next_state,) = self._cond_cell.step(tape, F.concat(tape, contexts), (next_state,), **kw2)
        contexts = [F.dropout(tape, c, keep, train, ctx.salt(self.name, "context", i, t))
                    for i, c in enumerate(contexts)]                   # :331-332
        cell_output = F.dropout(tape, cell_output, keep, train, ctx.salt(self.name, "cell_output", t))
        if not project:      # the caller projects the outputs of all steps at once (nothing of it feeds the recurrence)
            return None, [next_state, cell_output] + contexts
        output = self.output_projection.apply_var(tape, self, cell_output, emb_in, contexts, train,
                                                  ctx.salt(self.name, "output_projection", t))
        return output, [next_state, cell_output] + contexts

    def _logit_params(self, tape: F.Tape):
        """(W Var, trans_b, bias Var) of state_to_logits (autoregressive.py:450-459)."""
        ctx = tape.ctx
        bias_data = self.decoding_bias(ctx)
        if self.tie_embeddings:
            return tape.named_param(self.embedding_matrix_name), True, tape.leaf(bias_data)
        b = tape.param(self, "state_to_word_b")
        return tape.param(self, "state_to_word_W"), False, F.Var(bias_data, b.grad, b.needs_grad)

    # -- training --------------------------------------------------------------------------------
    def _general_train_loop(self, ctx, want_grad: bool, grad_scale: Optional[torch.Tensor]):
        from .decoder import TrainResult
        train = bool(ctx.fed(self.train_mode))
        tape = F.Tape(ctx, (id(self), "gtrain"), recording=want_grad)
        tgt, tmask = self.train_inputs(ctx), self.train_mask(ctx)
        steps, bsz = tgt.shape
        rows = steps * bsz
        keep = self.dropout_keep_prob

        table = tape.named_param(self.embedding_matrix_name)
        emb_all = F.embedding(tape, table, self._dec_input_ids(ctx).reshape(-1))
        emb_all = F.dropout(tape, emb_all, keep, train, ctx.salt(self.name, "embedded_input"))

        enc_outs = [tape.leaf(enc.output(ctx), needs_grad=True) for enc in self.encoders]
        s0 = self.encoder_projection.apply_var(tape, self, self.rnn_size, enc_outs, bsz, train)
        s0 = F.dropout(tape, s0, keep, train, ctx.salt(self.name, "initial_state"))      # :235-240
        sessions = [a.tape_session(tape, train) for a in self.attentions]
        att_states = [a.initial_loop_state(ctx, bsz, steps, precompute=False) for a in self.attentions]
        state = [s0, s0] + [tape.leaf(tape.buf((bsz, a.context_vector_size), zero=True))
                            for a in self.attentions]
        # The output projection reads the step's cell output, input embedding and contexts and feeds nothing back into
        # the recurrence: without dropout (whose masks are drawn per step) it is ONE pass over the rows of all steps
        # after the loop -- three products and an activation per step otherwise (3 + 3 of the 16 skinny products of a
        # step of the general-path model at the headline size).  NM_HOIST_OUTPUT=0: inside the loop as in rounds 2-5.
        op_keep = getattr(self.output_projection, "dropout_keep_prob", 1.0)
        hoist = HOIST_OUTPUT and (not train or (keep == 1.0 and op_keep == 1.0))
        out_all = None if hoist else tape.new((rows, self.output_dimension))
        s_all = c_alls = None
        # the first cell's input half: the embedded target words are all known (teacher forcing) -- one product over
        # the rows of all steps where the cell offers it (NematusGRUCell) and nothing else enters the cell's input
        xproj_all = None
        if not self._attention_on_input and hasattr(self._cell_obj, "project_inputs"):
            xproj_all = self._cell_obj.project_inputs(tape, emb_all)
        for t in range(steps):
            emb_t = tape.rows(emb_all, t * bsz, (t + 1) * bsz)
            views = None
            if hoist:
                if s_all is None:
                    s_all = tape.new((rows, self.rnn_size))
                    c_alls = [tape.new((rows, a.context_vector_size)) for a in self.attentions]
                views = (tape.rows(s_all, t * bsz, (t + 1) * bsz),
                         [tape.rows(c_all, t * bsz, (t + 1) * bsz) for c_all in c_alls])
            out_t, state = self.general_step(tape, emb_t, state, sessions, [st.weights[t] for st in att_states],
                                             train, t, project=not hoist,
                                             x_proj=None if xproj_all is None else tape.rows(xproj_all, t * bsz, (t + 1) * bsz),
                                             out_views=views)
            if hoist:                # whatever did not write its rows itself is copied there
                if state[1] is not views[0]:
                    F.copy(tape, state[1], out=views[0])
                for cv, c in zip(views[1], state[2:]):
                    if c is not cv:
                        F.copy(tape, c, out=cv)
            else:
                F.copy(tape, out_t, out=tape.rows(out_all, t * bsz, (t + 1) * bsz))
        if hoist:
            out_all = self.output_projection.apply_var(tape, self, s_all, emb_all, c_alls, train,
                                                       ctx.salt(self.name, "output_projection", 0))
        w, trans_b, bias = self._logit_params(tape)
        logits = F.linear(tape, out_all, w, bias, trans_b=trans_b)
        loss_rows = F.xent(tape, logits, tgt.reshape(-1), self.xent_weights(tmask.reshape(-1)), grad_scale,
                           self.label_smoothing or 0.0)
        loss_sum = ctx.buffer((id(self), "gtrain", "loss_sum"), (1,))
        ops.reduce_sum(loss_rows, loss_sum)
        from ..attention.base_attention import AttentionLoopState
        for att, st in zip(self.attentions, att_states):
            att.finalize_loop("{}_train".format(self.name), AttentionLoopState(st.contexts, st.weights, steps))
        saved = {"tape": tape, "enc_outs": enc_outs, "sessions": sessions, "steps": steps, "bsz": bsz,
                 "dlogits": logits.data if want_grad else None, "logits": logits.data,
                 "loss_rows": loss_rows, "loss_layout": "tb"}
        return TrainResult(loss_sum, self.train_token_count(ctx), steps, saved)

    def _general_backward(self, ctx, res) -> None:
        sv = res.saved
        sv["tape"].backward()
        enc_grads = {}

        def add_states_grad(enc, g):
            slot = enc_grads.setdefault(enc, [None, None])
            if slot[0] is None:
                slot[0] = g
            else:                        # several attentions over one encoder
                ops.ew("copy", g.reshape(-1, g.shape[-1]), None, slot[0].reshape(-1, g.shape[-1]), accumulate=True)
        def add_output_grad(enc, g):
            slot = enc_grads.setdefault(enc, [None, None])
            if slot[1] is None or g is None:
                slot[1] = g if slot[1] is None else slot[1]
            else:                        # initial state and a StatefulContext over one encoder
                ops.ew("copy", g, None, slot[1], accumulate=True)
        for sess in sv["sessions"]:
            for enc, g, *kind in sess.encoder_grads():
                (add_output_grad if kind == ["output"] else add_states_grad)(enc, g)
        for enc, var in zip(self.encoders, sv["enc_outs"]):
            add_output_grad(enc, var.grad)
        proj_states = getattr(self.encoder_projection, "states_var", None)
        if proj_states is not None and proj_states.grad is not None:     # nematus_projection reads the states
            add_states_grad(self.encoders[0], proj_states.grad.view(sv["bsz"], -1, proj_states.shape[1]))
            self.encoder_projection.states_var = None
        for enc, (dst, dfin) in enc_grads.items():
            ctx.defer_backward(enc, dst, dfin)


class FastStepper:
    """Inference steps of the plain TF-GRU decoder, 6 GEMM launches per step:

        xp     = emb . [Wg_x | Wc_x] + [bg | bc]      one GEMM over a fused copy of the input halves
        h      = GRU(xp, h_prev)                      two skinny MFMA GEMMs with gate / blend epilogues
        ctx    = attention(h)                         query GEMM + fused Bahdanau kernel
        out    = tanh([h | emb | ctx] . Wo + bo)      one GEMM: the three producers write side by side
        logits = out . W + b

    ``emb_view`` is where the caller embeds the step's input symbols.  A step touches persistent
    buffers only and keeps no Python-side state when ``h_prev`` / ``h_out`` are passed explicitly, so
    runs of steps can be captured into HIP graphs (``graph_safe``)."""
    graph_safe = True

    def __init__(self, dec, ctx, rows: int, tag: str):
        self.dec, self.ctx, self.rows = dec, ctx, rows
        self.cell = dec._cell(ctx)                        # pylint: disable=protected-access
        self.bufs = dec._step_bufs(ctx, rows)             # pylint: disable=protected-access
        e, h = dec.embedding_size, dec.rnn_size
        csz = [a.context_vector_size for a in dec.attentions]
        key = (id(dec), tag, rows)
        self.hbuf = ctx.buffer(key + ("h",), (2, rows, h))
        self.sel = ctx.buffer(key + ("hsel",), (rows, h))
        self.cat = ctx.buffer(key + ("cat",), (rows, h + e + sum(csz)))      # [h | emb | ctx...]
        self.emb_view = self.cat[:, h:h + e]
        self.ctx_views, col = [], h + e
        for c in csz:
            self.ctx_views.append(self.cat[:, col:col + c])
            col += c
        self.y = [ctx.buffer(key + ("y", i), (rows, a.state_size)) for i, a in enumerate(dec.attentions)]
        # fused input halves of the two GRU kernels (weights are constant during a decoding run)
        cell = self.cell
        self.w_in = ctx.buffer(key + ("w_in",), (e, 3 * h))
        self.b_in = ctx.buffer(key + ("b_in",), (3 * h,))
        ops.copy_cols(cell["wg_x"], self.w_in[:, :2 * h])
        ops.copy_cols(cell["wc_x"], self.w_in[:, 2 * h:])
        ops.copy(self.b_in[:2 * h], cell["bg"])
        ops.copy(self.b_in[2 * h:], cell["bc"])
        self.cur = 0
        self.src = None

    def start(self, s0: torch.Tensor) -> None:
        self.src = s0

    def step(self, emb, att_states, out_state, logits, h_out: Optional[torch.Tensor] = None, finished=None,
             h_prev: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None):
        """``stats``: run the vocabulary projection with its row statistics in the GEMM epilogue
        (``ops.logits_stats_gemm``); ``logits`` may then be None (nobody reads the logits)."""
        from ..attention.base_attention import AttentionLoopState
        from ..nn import gru
        dec, ctx, rows = self.dec, self.ctx, self.rows
        h = dec.rnn_size
        nxt = self.cur ^ 1
        dst = h_out if h_out is not None else self.hbuf[nxt]
        if h_prev is not None:
            self.src = h_prev
        if emb.data_ptr() != self.emb_view.data_ptr():           # caller used its own buffer
            ops.copy_cols(emb, self.emb_view)
        bufs = self.bufs
        xp = ops.gemm(self.emb_view, self.w_in, out=bufs["xp"], bias=self.b_in)
        ld = self.cat.stride(0)
        gru.step_fwd(xp, (0, 3 * h, 0), self.src, dst, self.cell["wg_h"], self.cell["wc_h"], bufs["ru"], bufs["rh"],
                     None, self.cat, (0, ld, 0), None, 0, 1, rows, h, False, bufs["hg"], bufs["hc"])
        new_states = []
        for att, st, y, cview in zip(dec.attentions, att_states, self.y, self.ctx_views):
            att.attention_into(ctx, dst, y, cview, st.weights[st.step])
            new_states.append(AttentionLoopState(st.contexts, st.weights, st.step + 1))
        dec.output_projection.apply_concat(ctx, dec, self.cat, out_state)
        if stats is not None:
            dec.state_to_logits_stats(ctx, out_state, stats, out=logits)
        else:
            dec.state_to_logits(ctx, out_state, logits)
        self.cur, self.src = nxt, dst
        return new_states

    def reorder(self, src_rows: torch.Tensor) -> None:
        ops.gather_rows(self.src, src_rows, self.sel)
        self.src = self.sel


class GeneralStepper:
    """Inference steps of any other decoder configuration (non-recording tape)."""
    graph_safe = False          # the carried state is a chain of Python-side Vars (see ``indexed`` below)

    def __init__(self, dec, ctx, rows: int, tag: str):
        self.dec, self.ctx, self.rows = dec, ctx, rows
        self.tape = F.Tape(ctx, (id(dec), tag, rows), recording=False)
        self.sessions = [a.tape_session(self.tape, False) for a in dec.attentions]
        self.sel = [ctx.buffer((id(dec), tag, "sel", rows, i), (rows, sz))
                    for i, sz in enumerate(dec.state_sizes())]
        self.zero_ctx = [ctx.buffer((id(dec), tag, "ctx0", rows, i), (rows, a.context_vector_size), zero=True)
                         for i, a in enumerate(dec.attentions)]
        self.emb_view = ctx.buffer((id(dec), tag, "emb", rows), (rows, dec.embedding_size))
        self.base = self.tape._n                          # pylint: disable=protected-access
        self.t = 0
        self.state = None
        self._initial = None
        # The state a step of parity p produces lives in the same persistent buffers on every batch: the
        # handles are kept with the session so that a chunk of steps launched from Python can follow a chunk
        # that was replayed from a captured graph (whose Python body did not run).
        self._produced = ctx.session.__dict__.setdefault("_stepper_state", {}).setdefault(
            (id(dec), tag, rows, ctx.session.slot), {})

    indexed = True       # set_position(t, cur) makes a step a function of its index: HIP-graph capturable

    @property
    def shape_key(self):
        """What a captured step depends on besides its index: the shapes its attentions read."""
        return tuple(getattr(s, "shape_key", ()) for s in self.sessions)

    def set_position(self, t: int, cur: int = 0) -> None:
        """Run step ``t`` next, from the state step ``t - 1`` left in its parity's buffers."""
        self.t = t
        for sess in self.sessions:
            if hasattr(sess, "t"):
                sess.t = t
        self.state = self._initial if t == 0 else self._produced[(t - 1) & 1]

    def start(self, s0: torch.Tensor) -> None:
        tape = self.tape
        self.state = self._initial = [tape.leaf(s0), tape.leaf(s0)] + [tape.leaf(z) for z in self.zero_ctx]
        self.t = 0

    def step(self, emb, att_states, out_state, logits, h_out: Optional[torch.Tensor] = None, finished=None,
             h_prev=None):
        from ..attention.base_attention import AttentionLoopState
        tape, dec = self.tape, self.dec
        tape._n, tape._slot = self.base, self.t & 1       # recycle this parity's step buffers
        w_outs = [st.weights[st.step] for st in att_states]
        out, self.state = dec.general_step(tape, tape.leaf(emb), self.state, self.sessions, w_outs, False,
                                           self.t)
        self._produced[self.t & 1] = self.state
        ops.copy(out_state, out.data)
        dec.state_to_logits(self.ctx, out_state, logits)
        if h_out is not None:
            ops.copy(h_out, self.state[1].data)
        self.t += 1
        return [AttentionLoopState(st.contexts, st.weights, st.step + 1) for st in att_states]

    def reorder(self, src_rows: torch.Tensor) -> None:
        for var, sel in zip(self.state, self.sel):
            ops.gather_rows(var.data, src_rows, sel)
        self.state = [self.tape.leaf(sel) for sel in self.sel]
        for sess in self.sessions:                        # per-hypothesis attention state (coverage)
            if hasattr(sess, "reorder"):
                sess.reorder(src_rows)


def input_table(dec, ctx) -> Optional[torch.Tensor]:
    """[V, 2H + H + O] = [E.Wg_x | E.Wc_x + bc | E.Wo_e]: everything an inference step of the plain-GRU decoder
    computes from the embedded input symbol alone (the input rows of the GRU kernels, nn/ortho_gru_cell.py:44-53,
    and the embedding rows of the output projection, decoders/output_projection.py:115-130) is a function of the
    symbol, so it is tabulated once per set of weights -- one [V, E] x [E, 2048] GEMM, 0.5 ms and 262 MB at the
    benchmark shape -- and the step's group 1 shrinks to the state half of the gates product
    (``nm_decoder_step.in_table``).  The table is tied to the session's variable signature
    (``Session.variables_signature``: torch's version counter of the flat parameter tensor + the counter that
    kernels writing through raw pointers bump) and rebuilt when that moved -- an optimizer step, a checkpoint, a
    test poking a row.  (Round 3 first compared the SUMS of the five tensors at the start of every decoding run:
    five reductions + a blocking 20-byte read-back, 0.15 ms of every batch.)  NM_STEP_TABLES=0: off."""
    if os.environ.get("NM_STEP_TABLES", "1") == "0" or ctx.device.type != "cuda":
        return None
    if torch.cuda.is_current_stream_capturing():
        return None
    e, h, o = dec.embedding_size, dec.rnn_size, dec.output_dimension
    pre = "attention_decoder/OrthoGRUCell"
    emb = dec.embedding_matrix(ctx)
    wg, wc = dec.var(ctx, pre + "/gates/kernel"), dec.var(ctx, pre + "/candidate/kernel")
    bc = dec.var(ctx, pre + "/candidate/bias")
    wo = dec.output_projection.kernel(ctx, dec)                  # rows [h | emb | ctx]
    parts = [emb, wg[:e], wc[:e], bc, wo[h:h + e]]
    if not all(p.is_contiguous() for p in parts):
        return None
    # keyed by the decoder OBJECT (weakly): an entry dies with its model part, and a new part that happens to get
    # the address of a collected one can never hit its table
    cache = ctx.session.__dict__.setdefault("_input_tables", weakref.WeakKeyDictionary())
    entry = cache.get(dec)
    print_ = ctx.session.variables_signature() + tuple(p.data_ptr() for p in parts)
    if entry is not None and entry[0] == print_:
        return entry[1]
    vsz = emb.shape[0]
    table = entry[1] if entry is not None and tuple(entry[1].shape) == (vsz, 3 * h + o) else \
        torch.empty((vsz, 3 * h + o), dtype=torch.float32, device=ctx.device)
    ops.gemm(emb, wg[:e], out=table[:, :2 * h])
    ops.gemm(emb, wc[:e], out=table[:, 2 * h:3 * h], bias=bc)
    ops.gemm(emb, wo[h:h + e], out=table[:, 3 * h:])
    cache[dec] = (print_, table)
    return table


class FusedStepper:
    """One inference step of the plain-GRU decoder with ONE Bahdanau attention in four GEMM-group launches
    and the split-S attention kernel (``nm_step_group``, csrc/nm_step.hip) -- Decoder.next_state,
    decoders/decoder.py:279-358:

        group 1   [emb | h] . Wg + bg -> r, u, r*h     emb . Wc_x + bc -> xc     emb . Wo[emb rows] -> Pe
        group 2   (r*h) . Wc_h + xc -> c ; h' = u*h + (1-u)*c     (in place in ``cat`` and into ``h_out``)
        group 3   y = h' . Wq + bq                      P = h' . Wo[h rows] + Pe + bo
        attention (one launch: energies, softmax, mask renormalisation, context; the split-S partials are merged
                  by the last-arriving chunk workgroup of every sentence)
        group 4   out = act(ctx . Wo[ctx rows] + P)
        logits    (+ per-tile row statistics when the caller passes ``stats``)

    ``cat`` = [emb | h] is the persistent input row of the step: the caller embeds the next input symbols
    into ``emb_view`` and -- beam search -- gathers the surviving states into ``sel``; group 2 leaves h' in
    ``sel`` (``state_resident``: a greedy loop passes ``h_prev`` only for its first step).  The weights are
    transposed once per decoding run ([N,K]: both MFMA fragments of a wave are 16-byte loads)."""
    graph_safe = True
    state_resident = True

    @staticmethod
    def supported(dec, ctx, rows: int):
        from ..attention.feed_forward import Attention
        if len(dec.attentions) != 1 or type(dec.attentions[0]) is not Attention:      # pylint: disable=unidiomatic-typecheck
            return None
        e, h = dec.embedding_size, dec.rnn_size
        # up to 256 rows the groups run on 16-row tiles, above (beam search: batch x beam hypotheses) on 32x32 tiles
        # (step_group_medium_kernel, csrc/nm_step.hip); NM_STEP_MEDIUM=0 sends those to the six-GEMM stepper
        limit = 4096 if os.environ.get("NM_STEP_MEDIUM", "1") != "0" else 256
        if e % 16 or h % 16 or dec.output_dimension % 4 or rows > limit:
            return None
        return dec.attentions[0].partials_plan(ctx, rows)

    def __init__(self, dec, ctx, rows: int, tag: str, plan):
        self.dec, self.ctx, self.rows, self.plan = dec, ctx, rows, plan
        att = self.att = dec.attentions[0]
        e, h, o = dec.embedding_size, dec.rnn_size, dec.output_dimension
        a, c = att.state_size, att.context_vector_size
        key = (id(dec), tag, rows, "fused")
        buf = lambda name, shape: ctx.buffer(key + (name,), shape)
        # Operand rows are padded by 128 bytes: with power-of-two row strides (E + H = 1024 floats = 4 KB) every row
        # of a 16- or 32-row operand tile starts on the SAME L2 channel, and all workgroups of a group walk K in
        # step.  Measured at 640 and 128 rows with 32 floats of padding: no difference (group 1: 39.4 us either way,
        # profiles/r03_decode_beam_kernels_v2.csv), so the default stays dense; NM_STEP_PAD=32 pads.
        pad = int(os.environ.get("NM_STEP_PAD", "0"))
        padded = lambda name, r, c: buf(name, (r, c + pad))[:, :c]
        self.cat = padded("cat", rows, e + h)
        self.emb_view, self.sel = self.cat[:, :e], self.cat[:, e:]
        self.hbuf = buf("h", (2, rows, h))
        self.ru, self.rh, self.xc = buf("ru", (rows, 2 * h)), buf("rh", (rows, h)), buf("xc", (rows, h))
        self.y, self.pre, self.pre_e = buf("y", (rows, a)), buf("pre", (rows, o)), buf("pre_e", (rows, o))
        # transposed weights of this run
        pre = "attention_decoder/OrthoGRUCell"
        wg, wc = dec.var(ctx, pre + "/gates/kernel"), dec.var(ctx, pre + "/candidate/kernel")
        proj = dec.output_projection
        wo = proj.kernel(ctx, dec)                                    # rows [h | emb | ctx]
        tr = lambda name, w: padded(name, w.shape[1], w.shape[0]).copy_(w.t())
        self.wg_t = tr("wgT", wg)                                     # [2H, E+H]
        self.wcx_t, self.wch_t = tr("wcxT", wc[:e]), tr("wchT", wc[e:])
        self.wq_t = tr("wqT", att.var(ctx, "Attention/attn_query_projection"))
        self.wo_h_t, self.wo_e_t, self.wo_c_t = tr("wo_hT", wo[:h]), tr("wo_eT", wo[h:h + e]), tr("wo_cT", wo[h + e:])
        bg, bc = dec.var(ctx, pre + "/gates/bias"), dec.var(ctx, pre + "/candidate/bias")
        ld = self.cat.stride(0)
        lds = lambda w: w.stride(0)
        rpk, bk = att.rows_per_key, plan["Bk"]
        self.g1 = ops.StepGroup(rows, [
            dict(A=self.cat, lda=ld, Bt=self.wg_t, ldb=lds(self.wg_t), N=2 * h, K=e + h, epilogue=1, bias=bg, h=self.sel,
                 ldh=ld, ru=self.ru, rh=self.rh),
            dict(A=self.cat, lda=ld, Bt=self.wcx_t, ldb=lds(self.wcx_t), N=h, K=e, epilogue=0, bias=bc, C=self.xc, ldc=h),
            dict(A=self.cat, lda=ld, Bt=self.wo_e_t, ldb=lds(self.wo_e_t), N=o, K=e, epilogue=0, C=self.pre_e, ldc=o)])
        self.g2 = ops.StepGroup(rows, [
            dict(A=self.rh, lda=h, Bt=self.wch_t, ldb=lds(self.wch_t), N=h, K=h, epilogue=2, xc=self.xc, ldxc=h, ru=self.ru,
                 h=self.sel, ldh=ld, h_out=self.sel, ldho=ld, h_out2=self.hbuf[0], ldho2=h)])
        self.g3 = ops.StepGroup(rows, [
            dict(A=self.sel, lda=ld, Bt=self.wq_t, ldb=lds(self.wq_t), N=a, K=h, epilogue=0,
                 bias=att.var(ctx, "attn_projection_bias"), C=self.y, ldc=a),
            dict(A=self.sel, lda=ld, Bt=self.wo_h_t, ldb=lds(self.wo_h_t), N=o, K=h, epilogue=0, bias=proj.bias(ctx, dec),
                 add=self.pre_e, ldadd=o, C=self.pre, ldc=o)])
        self.ctxbuf = padded("ctx", rows, c)
        ldc_ = self.ctxbuf.stride(0)
        self.g4 = ops.StepGroup(rows, [
            dict(A=self.ctxbuf, lda=ldc_, Bt=self.wo_c_t, ldb=lds(self.wo_c_t), N=o, K=c, epilogue=0,
                 act=1 if proj.activation == "tanh" else 0, add=self.pre, ldadd=o, C=self.pre, ldc=o)])
        att.hidden_features(ctx)
        dec.ensure_split_projection(ctx)          # (opt-in NM_PROJ_SPLIT=1; a no-op otherwise)
        self._pending, self._cur = None, 0
        # the same step behind the single C entry (nm_decoder_step_fused): descriptor built once, per-step
        # pointers patched in step()
        tied = dec.tie_embeddings
        wv = dec.embedding_matrix(ctx) if tied else dec.var(ctx, "state_to_word_W")
        self.whole = ops.DecoderStepCall(dict(
            rows=rows, emb=e, rnn=h, attn_state=a, ctx_width=c, out=o, vocab=wv.shape[0] if tied else wv.shape[1],
            src_len=att.attention_states(ctx).shape[1], rows_per_key=rpk, cat=self.cat,
            ru=self.ru, rh=self.rh, xc=self.xc, y=self.y, pre_e=self.pre_e, pre=self.pre, ctx=self.ctxbuf,
            attn_workspace=plan["ws"], attn_workspace_bytes=plan["ws"].numel() * 4,
            wg_t=self.wg_t, bg=bg, wcx_t=self.wcx_t, wch_t=self.wch_t, bc=bc, wq_t=self.wq_t,
            bq=att.var(ctx, "attn_projection_bias"), keys=att.hidden_features(ctx), values=att.attention_states(ctx),
            mask=att.attention_mask(ctx), v=att.var(ctx, "attn_similarity_v"), attn_bias=att.var(ctx, "attn_bias"),
            wo_h_t=self.wo_h_t, wo_e_t=self.wo_e_t, wo_c_t=self.wo_c_t, bo=proj.bias(ctx, dec),
            out_act=1 if proj.activation == "tanh" else 0, w_vocab=wv, ld_w_vocab=wv.stride(0),
            b_vocab=dec.decoding_bias(ctx), vocab_trans_b=int(tied),
            ld_cat=ld, ld_ctx=ldc_, ld_wg=lds(self.wg_t), ld_wcx=lds(self.wcx_t), ld_wch=lds(self.wch_t),
            ld_wq=lds(self.wq_t), ld_wo_h=lds(self.wo_h_t), ld_wo_e=lds(self.wo_e_t), ld_wo_c=lds(self.wo_c_t)))
        self.single_call = not os.environ.get("NM_STEP_GROUPS")
        # the same step with the embedding half of group 1 read from the input tables (``step(..., ids=)``)
        self.table = input_table(dec, ctx) if self.single_call else None
        self.whole_tab = None
        if self.table is not None:
            fields = {name: getattr(self.whole.desc, name) for name, _ in self.whole.desc._fields_}
            self.whole_tab = ops.DecoderStepCall(fields)
            self.whole_tab.keep = dict(self.whole.keep)
            self.whole_tab.launch_fields = dict(in_table=self.table, ld_table=self.table.stride(0))
            # greedy-sized steps: the three recurrent step groups as ONE launch of workgroup clusters
            # (dec_step_cluster_kernel, csrc/nm_gru_cluster.hip).  The workspace -- counters, epoch, tagged granules --
            # is this stepper's own and starts as zeros.  OPT-IN (NM_STEP_CLUSTER=1): measured SLOWER than the three
            # launches, 37.1 against 23.8 us per greedy step (profiles/r06_decode_greedy_kernel_stats_v1.csv) -- a launch
            # is one step, so the weights are not stationary as in the time loops: every workgroup re-fetches its 192 KB
            # of weight slices per step and gathers 128 KB of granules, the same L1-fill traffic as the three groups,
            # plus the role agreement; kept as a verified negative result (tests/test_step_cluster_gpu.py).
            self.cluster_ws = None
            lib = ops._lib.load()               # pylint: disable=protected-access
            if (os.environ.get("NM_STEP_CLUSTER", "0") == "1" and ctx.session.use_cluster_loops
                    and lib.nm_dec_step_cluster_supported(rows, h, a, o)):
                self.cluster_ws = torch.zeros(lib.nm_dec_step_cluster_workspace_bytes(rows, h) // 4,
                                              dtype=torch.float32, device=self.cat.device)

    def start(self, s0: torch.Tensor) -> None:
        self._pending, self._cur = s0, 0

    def step(self, emb, att_states, out_state, logits, h_out: Optional[torch.Tensor] = None, finished=None,
             h_prev: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None,
             ids: Optional[torch.Tensor] = None):
        """``ids`` (int32 [rows], the step's input symbols): with input tables the embedded input is not read at
        all -- the caller need not embed."""
        from ..attention.base_attention import AttentionLoopState
        dec, ctx = self.dec, self.ctx
        tabled = ids is not None and self.whole_tab is not None
        if not tabled and emb.data_ptr() != self.emb_view.data_ptr():
            ops.copy_cols(emb, self.emb_view)
        if h_prev is None and self._pending is not None:          # stateful use (ensembles): start() then step()s
            h_prev = self._pending
        self._pending = None
        if h_prev is not None and h_prev.data_ptr() != self.sel.data_ptr():
            ops.copy_cols(h_prev, self.sel)                       # first step of a loop: the initial state
        if h_out is None:                                         # stateful use: keep a copy for reorder()
            self._cur ^= 1
            h_out = self.hbuf[self._cur]
        st = att_states[0]
        if self.single_call:
            call = self.whole_tab if tabled else self.whole
            extra = dict(call.launch_fields, in_ids=ids) if tabled else {}
            if tabled:
                # (a session that fell back to per-step launches -- Session.demote_cluster_loops -- takes the three
                # step groups again: the chunks' graphs are captured anew after a demotion)
                ws = self.cluster_ws if (self.cluster_ws is not None and ctx.session.use_cluster_loops) else None
                extra.update(cluster_ws=ws, cluster_ws_bytes=ws.numel() * 4 if ws is not None else 0,
                             sticky_error=ctx.session.error_word() if ws is not None else None)
            call.launch(h_copy=h_out, ld_h_copy=h_out.stride(0), out_state=out_state,
                        ld_out_state=out_state.stride(0), attn_weights=st.weights[st.step], logits=logits,
                        ld_logits=logits.stride(0) if logits is not None else 0, stats=stats,
                        stats_bytes=stats.numel() * 4 if stats is not None else 0, **extra)
            return [AttentionLoopState(st.contexts, st.weights, st.step + 1)]
        self.g1.launch()
        self.g2.patch(0, h_out2=h_out, ldho2=h_out.stride(0) if h_out is not None else 0)
        self.g2.launch()
        self.g3.launch()
        att = self.att
        # energies, softmax, mask renormalisation and context in ONE launch: the chunk workgroup that arrives last
        # for a sentence merges the split-S partials (nm_attn_fwd, csrc/nm_attention.hip)
        ops.attn_fwd(self.y, att.hidden_features(ctx), att.attention_states(ctx), att.attention_mask(ctx),
                     att.var(ctx, "attn_similarity_v"), att.var(ctx, "attn_bias"), att.rows_per_key, self.ctxbuf,
                     st.weights[st.step], self.plan["ws"])
        self.g4.patch(0, C=out_state, ldc=out_state.stride(0))
        self.g4.launch()
        if stats is not None:
            dec.state_to_logits_stats(ctx, out_state, stats, out=logits)
        else:
            dec.state_to_logits(ctx, out_state, logits)
        return [AttentionLoopState(st.contexts, st.weights, st.step + 1)]

    def reorder(self, src_rows: torch.Tensor) -> None:
        ops.gather_rows(self.hbuf[self._cur], src_rows, self.sel)


def make_stepper(dec, ctx, rows: int, tag: str):
    if dec.uses_general_path(False):
        return GeneralStepper(dec, ctx, rows, tag)
    import os
    if not os.environ.get("NM_NO_FUSED_STEP"):
        plan = FusedStepper.supported(dec, ctx, rows)
        if plan is not None:
            # A stepper is a pile of descriptors over persistent buffers plus seven transposed weight matrices:
            # building one costs 0.6 ms of host time in front of every batch's first step.  It is kept per buffer
            # slot and re-used while nothing it points at moved: same variables (signature), same per-batch
            # tensors (keys, values, mask, workspace: persistent buffers keyed by shape -- compared by address).
            att = dec.attentions[0]
            ptr = lambda t: 0 if t is None else t.data_ptr()
            ident = (ctx.session.variables_signature(), plan["ws"].data_ptr(), plan["S"], plan["C"], plan["Bk"],
                     att.hidden_features(ctx).data_ptr(), att.attention_states(ctx).data_ptr(),
                     ptr(att.attention_mask(ctx)), ptr(dec.decoding_bias(ctx)), ptr(dec.embedding_matrix(ctx)),
                     os.environ.get("NM_STEP_GROUPS"), os.environ.get("NM_STEP_TABLES"), os.environ.get("NM_STEP_PAD"))
            cache = ctx.session.__dict__.setdefault("_fused_steppers", weakref.WeakKeyDictionary()).setdefault(dec, {})
            ckey = (tag, rows, ctx.session.slot)
            hit = cache.get(ckey)
            if hit is not None and hit[0] == ident:
                stepper = hit[1]
                stepper.ctx, stepper.plan = ctx, plan
                stepper._pending, stepper._cur = None, 0          # pylint: disable=protected-access
                return stepper
            stepper = FusedStepper(dec, ctx, rows, tag, plan)
            cache[ckey] = (ident, stepper)
            return stepper
    return FastStepper(dec, ctx, rows, tag)
