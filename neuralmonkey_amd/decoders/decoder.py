"""RNN attention decoder (mirror of neuralmonkey/decoders/decoder.py).

One step (Decoder.next_state, decoder.py:279-358):
    cell_output = GRU(rnn_input = embedded_input, prev_rnn_output)
    ctx         = attention(query = cell_output)
    output      = output_projection([cell_output, embedded_input, ctx])
and logits = output . W + b (autoregressive.py:450-459).

Training (teacher forcing): nothing but the GRU state feeds back, so the
embedding gather, the input half of the GRU, the output projection, the
logits GEMM and the cross entropy are hoisted out of the time loop and run
over all T*B rows as large MFMA GEMMs; the loop keeps two skinny recurrent
GEMMs + fused epilogues + the fused attention kernel per step.
Greedy / beam decoding feed the argmax back, so every step runs the full chain.
"""
from typing import Any, List, NamedTuple, Optional, Tuple

import contextlib
import os

import numpy as np
import torch

from .. import ops
from ..attention.base_attention import AttentionLoopState, BaseAttention
from ..model.model_part import InitializerSpecs, ModelPart
from ..model.sequence import EmbeddedSequence
from ..model.stateful import Stateful
from ..nn import gru
from ..nn.dropout import dropout
from ..runtime import tensor
from ..vocabulary import END_TOKEN_INDEX, START_TOKEN_INDEX, Vocabulary, sentence_mask
from .autoregressive import (AutoregressiveDecoder, DecoderConstants, DecoderFeedables,
                             DecoderHistories, LoopState)
from .decoder_general import GeneralDecoderMixin, make_stepper
from .encoder_projection import (EncoderProjection, concat_encoder_projection, empty_initial_state,
                                 linear_encoder_projection)
from .output_projection import OutputProjection, OutputProjectionSpec, nonlinear_output

RNN_CELL_TYPES = ("NematusGRU", "GRU", "LSTM")
# Steps per enqueued chunk (= per HIP graph) of a decoding loop.  The host reads the finished flags one chunk behind
# what it has enqueued (Session.decode_chunks), so a chunk only has to outlast one host round trip (~0.1 ms); a batch
# that finishes early wastes at most one chunk, hence fewer steps per chunk where a step is long.
FORWARD_OVERLAP = int(os.environ.get("NM_FWD_OVERLAP", "1"))     # 0 off, 1 on, 2 on with residency-capped GEMMs
DW_EARLY = os.environ.get("NM_DW_EARLY", "0") != "0"             # A/B: the long leaf GEMM before the attentions' backward
CHECK_EVERY = int(os.environ.get("NM_CHECK_EVERY", "8"))            # greedy RNN step ~0.1 ms
CHECK_EVERY_BEAM = int(os.environ.get("NM_CHECK_EVERY_BEAM", "4"))  # beam-5 step ~0.35 ms (Transformer: 1.4 ms)
CHECK_EVERY_TRANSFORMER = int(os.environ.get("NM_CHECK_EVERY_TRANSFORMER", "4"))   # cached greedy step ~0.65 ms


class RNNFeedables(NamedTuple):
    """decoder.py:34-50."""
    prev_rnn_state: torch.Tensor
    prev_rnn_output: torch.Tensor
    prev_contexts: List[torch.Tensor]


class RNNHistories(NamedTuple):
    """decoder.py:53-66."""
    rnn_outputs: torch.Tensor            # [T,R,H]
    attention_histories: List[Any]


class TrainResult(NamedTuple):
    loss_sum: torch.Tensor               # device scalar: sum of masked token xents
    token_count: float
    steps: int
    saved: dict                          # activations for the backward pass


class RuntimeResult(NamedTuple):
    symbols: torch.Tensor                # [T,B] int32
    mask: torch.Tensor                   # [T,B] int32
    steps: int
    xent_sum: Optional[torch.Tensor]     # device scalar over t < min(T_run, T_target)
    logits: Optional[torch.Tensor]       # [T,B,V] when requested
    output_states: torch.Tensor          # [T,B,E]
    rnn_outputs: torch.Tensor            # [T,B,H]
    attention_weights: List[torch.Tensor]


# pylint: disable=too-many-instance-attributes
class Decoder(GeneralDecoderMixin, AutoregressiveDecoder):
    # pylint: disable=too-many-arguments,too-many-locals
    def __init__(self, encoders: List[Stateful], vocabulary: Vocabulary, data_id: str, name: str,
                 max_output_len: int, dropout_keep_prob: float = 1.0, embedding_size: int = None,
                 embeddings_source: EmbeddedSequence = None, tie_embeddings: bool = False,
                 label_smoothing: float = None, rnn_size: int = None,
                 output_projection: OutputProjectionSpec = None,
                 encoder_projection: EncoderProjection = None,
                 attentions: List[BaseAttention] = None, attention_on_input: bool = False,
                 rnn_cell: str = "GRU", conditional_gru: bool = False, supress_unk: bool = False,
                 reuse: ModelPart = None, save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        AutoregressiveDecoder.__init__(
            self, name=name, vocabulary=vocabulary, data_id=data_id, max_output_len=max_output_len,
            dropout_keep_prob=dropout_keep_prob, embedding_size=embedding_size,
            embeddings_source=embeddings_source, tie_embeddings=tie_embeddings,
            label_smoothing=label_smoothing, supress_unk=supress_unk, reuse=reuse,
            save_checkpoint=save_checkpoint, load_checkpoint=load_checkpoint, initializers=initializers)
        self.encoders = encoders
        self._output_projection_spec = output_projection
        self._conditional_gru = conditional_gru
        self._attention_on_input = attention_on_input
        self._rnn_cell_str = rnn_cell
        self._rnn_size = rnn_size
        self._encoder_projection = encoder_projection
        self.attentions: List[BaseAttention] = attentions if attentions is not None else []
        if not rnn_size and not encoder_projection and not encoders:
            raise ValueError("No RNN size, no encoders and no encoder_projection specified")
        if self._rnn_cell_str not in RNN_CELL_TYPES:
            raise ValueError("RNN cell must be a either 'GRU', 'LSTM', or 'NematusGRU'. Not {}"
                             .format(self._rnn_cell_str))
        for att in self.attentions:
            att.bind_query_size(self.rnn_size)
            if hasattr(att, "bind_decoder"):
                att.bind_decoder(self)
        self._build_cells()
        if self.embedding_size != self.output_dimension:
            raise ValueError("The dimension ({}) of the output projection must be same as the "
                             "dimension of the input embedding ({})"
                             .format(self.output_dimension, self.embedding_size))

    # -- configuration-derived pieces (decoder.py:176-224) --------------------------------
    @property
    def encoder_projection(self) -> EncoderProjection:
        if not hasattr(self, "_enc_proj_cached"):
            if self._encoder_projection is not None:
                proj = self._encoder_projection
            elif not self.encoders:
                proj = empty_initial_state
            elif self._rnn_size is None:
                proj = concat_encoder_projection
            else:
                proj = linear_encoder_projection(self.dropout_keep_prob)
            self._enc_proj_cached = proj
        return self._enc_proj_cached

    @property
    def rnn_size(self) -> int:
        if self._rnn_size is not None:
            return self._rnn_size
        if self._encoder_projection is None:
            assert self.encoders
            return sum(e.output_size for e in self.encoders)
        raise ValueError("Cannot infer RNN size.")

    @property
    def output_projection_spec(self) -> Tuple[OutputProjection, int]:
        if not hasattr(self, "_out_proj_cached"):
            spec = self._output_projection_spec
            if spec is None:
                spec = (nonlinear_output(self.rnn_size)[0], self.rnn_size)
            elif not isinstance(spec, tuple):
                spec = (spec, self.rnn_size)
            self._out_proj_cached = spec
        return self._out_proj_cached

    @property
    def output_projection(self) -> OutputProjection:
        return self.output_projection_spec[0]

    @property
    def output_dimension(self) -> int:
        return self.output_projection_spec[1]

    def declare_variables(self, store) -> None:
        AutoregressiveDecoder.declare_variables(self, store)
        e, h = self.embedding_size, self.rnn_size
        self.encoder_projection.declare_variables(self, store, self.rnn_size, self.encoders)
        self._declare_general_variables(store)      # the cell(s): same names on both paths
        self.output_projection.declare_variables(
            self, store, h, e, [a.context_vector_size for a in self.attentions])

    def _cell(self, ctx):
        pre = "attention_decoder/OrthoGRUCell"
        e = self.embedding_size
        wg, wc = self.var(ctx, pre + "/gates/kernel"), self.var(ctx, pre + "/candidate/kernel")
        return {"wg_x": wg[:e], "wg_h": wg[e:], "bg": self.var(ctx, pre + "/gates/bias"),
                "wc_x": wc[:e], "wc_h": wc[e:], "bc": self.var(ctx, pre + "/candidate/bias")}

    # -- pieces of one step -------------------------------------------------------------
    @tensor
    def initial_state(self, ctx) -> torch.Tensor:
        """decoder.py:226-252 (the double dropout is the identity at keep_prob 1)."""
        bsz = int(ctx.fed(self.batch_size))
        out = ctx.buffer((id(self), "s0"), (bsz, self.rnn_size))
        train = bool(ctx.fed(self.train_mode))
        state = self.encoder_projection.apply(ctx, self, self.rnn_size, self.encoders, out, train)
        return dropout(ctx, state, self.dropout_keep_prob, train)

    def _time_loops_beside(self) -> bool:
        """Is there an encoder whose forward / backward pass is a time loop (work of this decoder that depends on
        nothing of it can then run beside that loop on a side stream)?"""
        encs = list(self.encoders) + [att.encoder for att in self.attentions]
        return any(getattr(enc, "has_time_loop", False) for enc in encs)

    def _input_projection(self, ctx, cell, emb, xp, algo=0):
        """xp[:, :2H] = emb.Wg_x + bg ; xp[:, 2H:] = emb.Wc_x + bc  (input half of the GRU)."""
        h = self.rnn_size
        ops.gemm(emb, cell["wg_x"], out=xp[:, :2 * h], bias=cell["bg"], algo=algo)
        ops.gemm(emb, cell["wc_x"], out=xp[:, 2 * h:], bias=cell["bc"], algo=algo)

    def _recurrent(self, ctx, cell, xp, t_index, x_time_stride, h_prev, h_out, ru, c_save, bufs, rh=None, wt=None):
        """State half of the GRU + fused epilogues: h_out = GRU(x_t, h_prev).  ``wt``: transposed recurrent
        kernels of this time loop (gru.transposed_weights)."""
        h = self.rnn_size
        rows = h_prev.shape[0]
        rh = bufs["rh"] if rh is None else rh
        wg, wc = wt if wt is not None else (cell["wg_h"], cell["wc_h"])
        gru.step_fwd(xp, (0, 3 * h, x_time_stride), h_prev, h_out, wg, wc, ru, rh, c_save,
                     None, (0, 0, 0), None, t_index, 1, rows, h, False, bufs["hg"], bufs["hc"],
                     transposed=wt is not None)

    def _step_bufs(self, ctx, rows):
        h = self.rnn_size
        key = (id(self), "step", rows)
        return {"hg": ctx.buffer(key + ("hg",), (rows, 2 * h)), "hc": ctx.buffer(key + ("hc",), (rows, h)),
                "rh": ctx.buffer(key + ("rh",), (rows, h)), "ru": ctx.buffer(key + ("ru",), (rows, 2 * h)),
                "xp": ctx.buffer(key + ("xp",), (rows, 3 * h))}

    def full_step(self, ctx, cell, emb_in, h_prev, h_out, att_states, out_state, logits, bufs):
        """One inference step on R rows: GRU -> attention(s) -> projection -> logits.
        Returns the new attention loop states."""
        self._input_projection(ctx, cell, emb_in, bufs["xp"])
        self._recurrent(ctx, cell, bufs["xp"], 0, 0, h_prev, h_out, bufs["ru"], None, bufs)
        contexts, new_states = [], []
        for att, st in zip(self.attentions, att_states):
            c, ns = att.attention(ctx, h_out, h_prev, emb_in, st)
            contexts.append(c)
            new_states.append(ns)
        self.output_projection.apply(ctx, self, h_out, emb_in, contexts, out_state)
        self.state_to_logits(ctx, out_state, logits)
        return new_states

    # -- training path ---------------------------------------------------------------------
    def _dec_input_ids(self, ctx) -> torch.Tensor:
        """Step inputs: <s> then the shifted targets (autoregressive.py:377-382,467-480)."""
        def shift(ids):
            tb = np.ascontiguousarray(ids.T)
            out = np.empty_like(tb)
            out[0] = START_TOKEN_INDEX
            out[1:] = tb[:-1]
            return out
        key = (id(self), "dec_in_tb")
        if key not in ctx.memo:        # one H2D copy per run, into a persistent (graph-safe) buffer
            ctx.memo[key] = ctx.session.staged(key, ctx.session.to_device(ctx.fed(self.train_tokens), torch.int32,
                                                                         "dec_in_tb", shift))
        return ctx.memo[key]

    def stage_inputs(self, ctx) -> None:
        AutoregressiveDecoder.stage_inputs(self, ctx)
        if self.has_targets(ctx):
            self._dec_input_ids(ctx)

    def decoding_loop(self, ctx, train_mode: bool, sample: bool = False, temperature: float = 1.0):
        """autoregressive.py:527-562.  ``sample`` / ``temperature`` (get_body, :440-493): the step's logits are
        divided by the temperature and the next symbol is drawn from their softmax instead of taken as the argmax;
        the loop then keeps its logits (the histories the reference's caller reads)."""
        self.check_sampling_args(train_mode, sample, temperature)
        if train_mode:
            return self._train_loop(ctx)
        if sample or temperature != 1.0:
            return self._runtime_loop(ctx, keep_logits=True, sample=sample, temperature=float(temperature))
        return self._runtime_loop(ctx, keep_logits=False)

    def _train_loop(self, ctx, want_grad: bool = False, grad_scale: Optional[torch.Tensor] = None) -> TrainResult:
        if self.uses_general_path(bool(ctx.fed(self.train_mode))):
            return self._general_train_loop(ctx, want_grad, grad_scale)
        key = (id(self), "train")
        tgt = self.train_inputs(ctx)                        # [T,B]
        tmask = self.train_mask(ctx)
        steps, bsz = tgt.shape
        e, h, v = self.embedding_size, self.rnn_size, len(self.vocabulary)
        cell = self._cell(ctx)
        rows = steps * bsz

        # The input half of the GRU (embedding rows, emb.W_x for all steps) needs nothing of the encoders: on a side
        # lane it runs beside the encoders' time loops, which ``initial_state`` evaluates and which leave the chip
        # idle.  Likewise the attention keys (states.W_k), beside this decoder's own time loop.
        overlap = FORWARD_OVERLAP and ctx.session.side_active() and self._time_loops_beside()
        emb_all = ctx.buffer(key + ("emb",), (steps, bsz, e))
        xp = ctx.buffer(key + ("xp",), (rows, 3 * h))
        dec_ids = self._dec_input_ids(ctx)
        def input_half():
            self.embed_input_symbols(ctx, dec_ids.reshape(-1), out=emb_all.view(rows, e))
            self._input_projection(ctx, cell, emb_all.view(rows, e), xp,
                                   algo=ops.GEMM_BACKGROUND if overlap and FORWARD_OVERLAP == 2 else 0)
        if overlap:
            ctx.session.defer_side(input_half)       # started by the first encoder time loop ...
        else:
            input_half()

        s0 = self.initial_state(ctx)
        if overlap:
            ctx.session.start_deferred_side()        # ... or here, when the encoders were evaluated already
            ctx.session.join_side(0)
            for att in self.attentions:
                att.attention_states(ctx)
                att.attention_mask(ctx)
            with ctx.session.side(0):
                for att in self.attentions:
                    att.hidden_features(ctx)
        s_ext = ctx.buffer(key + ("s_ext",), (steps + 1, bsz, h))      # [s0 ; s_1 .. s_T]
        ops.copy(s_ext[0], s0)
        s_all = s_ext[1:]
        ru_all = ctx.buffer(key + ("ru_all",), (steps, bsz, 2 * h))
        c_all = ctx.buffer(key + ("c_all",), (steps, bsz, h))
        rh_all = ctx.buffer(key + ("rh_all",), (steps, bsz, h))
        bufs = self._step_bufs(ctx, bsz)
        att_states = [a.initial_loop_state(ctx, bsz, steps) for a in self.attentions]
        y_all = [ctx.buffer(key + ("y", i), (steps, bsz, a.state_size)) for i, a in enumerate(self.attentions)]
        e_all = [ctx.buffer(key + ("e", i), (steps, bsz, st.weights.shape[2])) for i, st in enumerate(att_states)]
        for att in self.attentions:          # touch lazily built tensors outside the captured region
            att.hidden_features(ctx)
            att.attention_mask(ctx)

        wt = None          # transposed recurrent kernels: measured slower (12.2 vs 10.9 us per step), see encoders/recurrent.py

        def time_loop():
            for t in range(steps):
                self._recurrent(ctx, cell, xp, t, bsz * 3 * h, s_ext[t], s_all[t], ru_all[t], c_all[t], bufs,
                                rh=rh_all[t], wt=wt)
        loop_h = gru.seq_mode(ctx.session, bsz, h, 1, cell["wg_h"], cell["wc_h"])
        if loop_h:
            # the whole recurrence as ONE launch (csrc/nm_gru_cluster.hip; padded when the kernels do not take h)
            gru.seq_fwd(ctx, id(self), loop_h, steps, 1, bsz, h, xp, (0, 3 * h, bsz * 3 * h), s_ext[0], s_ext[1], bsz * h,
                        ru_all[0], bsz * 2 * h, rh_all[0], bsz * h, c_all[0], bsz * h, cell["wg_h"], cell["wc_h"])
        else:
            ctx.session.graphed((id(self), "train_loop", bsz, steps), time_loop)
        if overlap:
            ctx.session.join_side(0)
        # attention of all T steps at once (the contexts do not feed the recurrence)
        for i, att in enumerate(self.attentions):
            att.attention_all_steps(ctx, s_all, y_all[i], att_states[i].contexts, att_states[i].weights, e_all[i])
        att_states = [AttentionLoopState(st.contexts, st.weights, steps) for st in att_states]

        out_all = ctx.buffer(key + ("out",), (rows, self.output_dimension))
        self.output_projection.apply(ctx, self, s_all.view(rows, h), emb_all.view(rows, e),
                                     [st.contexts.view(rows, -1) for st in att_states], out_all)
        logits = ctx.buffer(key + ("logits",), (rows, v))
        self.state_to_logits(ctx, out_all, logits)
        loss_rows = ctx.buffer(key + ("loss_rows",), (rows,))
        db_partial = None
        if want_grad and not self.tie_embeddings and ops.xent_colsum_ok(logits):
            # the bias gradient of the vocabulary projection falls out of the pass that writes dlogits
            db_partial = ctx.buffer(key + ("db_partial",), (min(rows, ops.XENT_COLSUM_ROWS), v))
            ops.xent_colsum(logits, tgt.reshape(-1), self.xent_weights(tmask.reshape(-1)), loss_rows, grad_scale,
                            self.label_smoothing or 0.0, db_partial)
        else:
            ops.xent(logits, tgt.reshape(-1), self.xent_weights(tmask.reshape(-1)), loss_rows, grad_scale, want_grad,
                     self.label_smoothing or 0.0)
        loss_sum = ctx.buffer(key + ("loss_sum",), (1,))
        ops.reduce_sum(loss_rows, loss_sum)
        for att, st in zip(self.attentions, att_states):
            att.finalize_loop("{}_train".format(self.name), st)
        saved = {"emb_all": emb_all, "xp": xp, "s0": s0, "s_all": s_all, "s_ext": s_ext, "ru_all": ru_all,
                 "c_all": c_all, "rh_all": rh_all, "y_all": y_all, "e_all": e_all,
                 "att_states": att_states, "out_all": out_all,
                 "dlogits": logits if want_grad else None, "db_partial": db_partial, "cell": cell, "steps": steps,
                 "bsz": bsz,
                 "loss_rows": loss_rows, "loss_layout": "tb"}
        return TrainResult(loss_sum, self.train_token_count(ctx), steps, saved)

    def backward(self, ctx, res: TrainResult) -> None:
        """Back-propagate the teacher-forced loss (dlogits already in place) into
        ``ctx.store.grad`` and on into the attentions and encoders."""
        store = ctx.store
        sv = res.saved
        if "tape" in sv:
            return self._general_backward(ctx, res)
        steps, bsz = sv["steps"], sv["bsz"]
        rows = steps * bsz
        e, h, v = self.embedding_size, self.rnn_size, len(self.vocabulary)
        key = (id(self), "bwd")
        cell = sv["cell"]
        dlogits, out_all = sv["dlogits"], sv["out_all"]
        emb2 = sv["emb_all"].view(rows, e)
        s2 = sv["s_all"].reshape(rows, h)
        odim = self.output_dimension

        # Leaf work (weight gradients, bias column sums: nothing downstream reads them) is
        # enqueued on the session's side stream so it overlaps the latency-bound BPTT loops.
        # Leaf GEMMs go to side streams and run residency-capped (ops.GEMM_BACKGROUND: the loops' launches find room on
        # every CU) beside the main stream's BPTT loops -- when there are two loops to hide them under, this decoder's
        # and a recurrent encoder's.  With this decoder's alone (captioning: the encoder is a spatial filler, the
        # attention's backward over 64 x 2048 maps is bandwidth work itself) the second stream only adds contention:
        # 10.6 ms per step with it, 10.1 in stream order (tools/captioning_train_probe.py, NM_SIDE_STREAM=0).
        loops = self._time_loops_beside()
        side = ctx.session.side if loops else (lambda lane=0: contextlib.nullcontext())
        bg = ctx.session.leaf_algo() if loops else 0
        acc = self.shares_variables          # another part trains the same variables (reuse=): add, never overwrite
        from .. import distributed
        dp = distributed.current()
        dp_overlap = dp is not None and bool(ctx.memo.get("dp_overlap", False))

        # ---- logits = out . W + b  (autoregressive.py:450-459)
        d_out = ctx.buffer(key + ("d_out",), (rows, odim))
        if self.tie_embeddings:
            emat = self.embedding_matrix(ctx)
            ops.gemm(dlogits, emat, out=d_out)
        else:
            ops.gemm(dlogits, self.var(ctx, "state_to_word_W"), out=d_out, trans_b=True)

        # ---- output projection: out = act([s | emb | ctx...] . Wo + bo)
        proj = self.output_projection
        if proj.activation == "tanh":
            ops.tanh_bwd(d_out, out_all)
        elif proj.activation != "identity":
            raise NotImplementedError("backward of activation {}".format(proj.activation))
        wo = proj.kernel(ctx, self)
        g_wo = store.g(self.var_name("attention_decoder/{}/kernel".format(proj.scope)))
        ctx_all = [st.contexts.view(rows, -1) for st in sv["att_states"]]
        d_s = ctx.buffer(key + ("d_s",), (rows, h))
        d_emb = ctx.buffer(key + ("d_emb",), (rows, e))
        d_ctx = [ctx.buffer(key + ("d_ctx", i), (rows, c.shape[1])) for i, c in enumerate(ctx_all)]
        row = 0
        for x, dx, sz in zip([s2, emb2] + ctx_all, [d_s, d_emb] + d_ctx, proj.sizes):
            ops.gemm(d_out, wo[row:row + sz], out=dx, trans_b=True)
            row += sz
        with side():
            row = 0
            for x, sz in zip([s2, emb2] + ctx_all, proj.sizes):
                ops.gemm(x, d_out, out=g_wo[row:row + sz], trans_a=True, accumulate=acc, algo=bg)
                row += sz
            ops.colsum(d_out, store.g(self.var_name("attention_decoder/{}/bias".format(proj.scope))), accumulate=acc)

        def vocabulary_projection_gradient():
            if self.tie_embeddings:
                ops.gemm(dlogits, out_all, out=store.g(self.embedding_matrix_name), trans_a=True,
                         accumulate=True, algo=bg)
            else:
                ops.gemm(out_all, dlogits, out=store.g(self.var_name("state_to_word_W")), trans_a=True, accumulate=acc,
                         algo=bg)
                ops.colsum(sv["db_partial"] if sv.get("db_partial") is not None else dlogits,
                           store.g(self.var_name("state_to_word_b")), accumulate=acc)
                if dp_overlap:        # the largest gradient slice is final: its all-reduce runs under the BPTT
                    dp.all_reduce_early(store, [self.var_name("state_to_word_W"), self.var_name("state_to_word_b")])
        # The step's one long leaf GEMM ([rows, out]^T . [rows, V], ~2 ms).  Started HERE it runs beside the
        # attentions' backward, which is throughput work itself: the two trade places (a 67 us GEMM of the main
        # stream took 1.1 ms next to it, profiles/r04_train_step_timeline.txt).  Started after them it runs beside
        # the BPTT loops only, whose launches are latency-bound and leave the chip idle.
        early_dw = DW_EARLY
        if early_dw:
            with side(1):
                vocabulary_projection_gradient()

        # ---- attentions (batched over time); adds the query path into d_s
        d_att_states = []
        for i, att in enumerate(self.attentions):
            st = sv["att_states"][i]
            d_att_states.append(att.backward(ctx, d_ctx[i].view(steps, bsz, -1), sv["s_all"], sv["y_all"][i],
                                             st.weights, sv["e_all"][i], d_s))
        if not early_dw:
            with side(1):
                vocabulary_projection_gradient()

        # ---- BPTT through the GRU (the only recurrence)
        dh = ctx.buffer(key + ("dh",), (1, bsz, h), zero=True)
        dxp = ctx.buffer(key + ("dxp",), (rows, 3 * h))
        dgpre = ctx.buffer(key + ("dgpre",), (2, 1, bsz, 2 * h))
        dcpre = ctx.buffer(key + ("dcpre",), (1, bsz, h))
        drh = ctx.buffer(key + ("drh",), (1, bsz, h))
        s_all = sv["s_all"]
        seq_strides = (0, h, bsz * h)
        dxp_strides = (0, 3 * h, bsz * 3 * h)
        def bptt_loop():
            ops.zero(dh)
            gru.bptt(steps, dh, d_s, seq_strides, sv["ru_all"], sv["c_all"], sv["s0"], s_all, seq_strides, dxp,
                     dxp_strides, cell["wg_h"].unsqueeze(0), cell["wc_h"].unsqueeze(0), None, 1, bsz, h, False,
                     dgpre, dcpre, drh)
        loop_h = gru.seq_mode(ctx.session, bsz, h, 1, cell["wg_h"], cell["wc_h"])
        if loop_h:
            ops.zero(dh)
            gru.seq_bwd(ctx, id(self), loop_h, steps, 1, bsz, h, dh, d_s, seq_strides, sv["ru_all"][0], bsz * 2 * h,
                        sv["c_all"][0], bsz * h, sv["s0"], s_all, seq_strides, dxp, dxp_strides, cell["wg_h"],
                        cell["wc_h"])
            if dp is not None:
                dp.after_time_loops()
        else:
            ctx.session.graphed((id(self), "bptt_loop", bsz, steps), bptt_loop)
        ds0 = dh[0]

        # ---- GRU weight gradients, batched over all steps (side stream: overlaps the encoder BPTT)
        pre = "attention_decoder/OrthoGRUCell"
        g_wg, g_wc = store.g(self.var_name(pre + "/gates/kernel")), store.g(self.var_name(pre + "/candidate/kernel"))
        dg_all, dc_all = dxp[:, :2 * h], dxp[:, 2 * h:]
        s_prev = sv["s_ext"][:steps].reshape(rows, h)
        with side():
            ops.gemm(emb2, dg_all, out=g_wg[:e], trans_a=True, accumulate=acc, algo=bg)
            ops.gemm(s_prev, dg_all, out=g_wg[e:], trans_a=True, accumulate=acc, algo=bg)
            ops.gemm(emb2, dc_all, out=g_wc[:e], trans_a=True, accumulate=acc, algo=bg)
            ops.gemm(sv["rh_all"].view(rows, h), dc_all, out=g_wc[e:], trans_a=True, accumulate=acc, algo=bg)
            ops.colsum(dg_all, store.g(self.var_name(pre + "/gates/bias")), accumulate=acc)
            ops.colsum(dc_all, store.g(self.var_name(pre + "/candidate/bias")), accumulate=acc)
            ops.gemm(dg_all, cell["wg_x"], out=d_emb, trans_b=True, accumulate=True, algo=bg)
            ops.gemm(dc_all, cell["wc_x"], out=d_emb, trans_b=True, accumulate=True, algo=bg)
            ops.embedding_scatter_add(store.g(self.embedding_matrix_name), self._dec_input_ids(ctx).reshape(-1),
                                      d_emb)
            if dp_overlap and not self.tie_embeddings and self.embeddings_source is None:
                # the decoder's embedding gradient is final here and touches at most B*T + 1 rows (the step inputs:
                # <s> and the shifted targets): NM_DP_SPARSE_EMB=1 exchanges (ids, rows) instead of the dense [V, E]
                # slice, like the encoder's (model/sequence.py); otherwise the dense collective starts now and runs
                # under the encoder's backward
                sparse = False
                if dp.sparse_embeddings:
                    ids_host = np.concatenate([np.asarray(ctx.fed(self.train_tokens)).reshape(-1),
                                               np.asarray([START_TOKEN_INDEX])])
                    negate = lambda src, dst: ops.ew("scale", src, None, dst, alpha=-1.0)
                    scatter = lambda table, ids, rows: ops.embedding_scatter_add(table, ids, rows)
                    sparse = dp.exchange_sparse_rows(store, self.embedding_matrix_name, ids_host, ops.gather_rows,
                                                     scatter, negate, skip_pad=False)
                if not sparse:
                    dp.all_reduce_early(store, [self.embedding_matrix_name])

        # ---- initial state projection and the encoders
        d_enc_out = self.encoder_projection.backward(ctx, self, self.rnn_size, self.encoders, ds0)
        enc_grads = {}
        for att, dst in zip(self.attentions, d_att_states):
            enc_grads.setdefault(att.encoder, [None, None])[0] = dst
        for enc, dfin in zip(self.encoders, d_enc_out):
            enc_grads.setdefault(enc, [None, None])[1] = dfin
        for enc, (dst, dfin) in enc_grads.items():
            ctx.defer_backward(enc, dst, dfin)

    @tensor
    def train_loss(self, ctx) -> torch.Tensor:
        """sum(xent)/sum(mask) (autoregressive.py:312-316)."""
        res = self.train_loop_result(ctx)
        out = ctx.buffer((id(self), "train_loss"), (1,))
        alpha = 1.0 / res.token_count if res.token_count else float("nan")         # (0 / 0 as the reference's division)
        return ops.ew("scale", res.loss_sum[0:1], None, out, alpha=alpha)[0]

    # -- greedy runtime path -------------------------------------------------------------------
    def _runtime_loop(self, ctx, keep_logits: bool, sample: bool = False, temperature: float = 1.0) -> RuntimeResult:
        key = (id(self), "run", keep_logits)
        plain = not sample and temperature == 1.0          # the greedy loop of the runners
        if not plain:
            key = key + ("sample" if sample else "argmax", temperature)
        bsz = int(ctx.fed(self.batch_size))
        e, h, v = self.embedding_size, self.rnn_size, len(self.vocabulary)
        tmax = self.max_output_len
        has_tgt = self.has_targets(ctx)
        if has_tgt:
            tgt, tmask = self.train_inputs(ctx), self.train_mask(ctx)
            t_target = tgt.shape[0]

        s0 = self.initial_state(ctx)
        s_all = ctx.buffer(key + ("s_all",), (tmax, bsz, h))
        out_all = ctx.buffer(key + ("out_all",), (tmax, bsz, self.output_dimension))
        symbols = ctx.buffer(key + ("sym",), (tmax, bsz), torch.int32, zero=True)
        omask = ctx.buffer(key + ("mask",), (tmax, bsz), torch.int32, zero=True)
        finished = ctx.buffer(key + ("fin",), (bsz,), torch.int32, zero=True)
        allfin = ctx.buffer(key + ("allfin",), (tmax,), torch.int32)
        ops.fill(allfin, 1)
        argmax = ctx.buffer(key + ("argmax",), (bsz,), torch.int32)
        logits_all = ctx.buffer(key + ("logits_all",), (tmax, bsz, v)) if keep_logits else None
        logits_one = ctx.buffer(key + ("logits",), (bsz, v))
        xent_rows = ctx.buffer(key + ("xent_rows",), (tmax, bsz), zero=True) if has_tgt else None
        stepper = make_stepper(self, ctx, bsz, "greedy")
        stepper.start(s0)
        emb = stepper.emb_view                      # the step input is embedded straight into the stepper
        att_states = [a.initial_loop_state(ctx, bsz, tmax) for a in self.attentions]

        go = ctx.buffer(key + ("go",), (bsz,), torch.int32)
        ops.fill(go, START_TOKEN_INDEX)
        self.embed_input_symbols(ctx, go, out=emb)
        att0 = att_states
        graph_ok = getattr(stepper, "graph_safe", False)
        indexed = getattr(stepper, "indexed", False)      # general path: steps addressed by their index
        for att in self.attentions if graph_ok else []:   # lazily built tensors (H2D copies): outside the capture
            att.hidden_features(ctx)
            att.attention_mask(ctx)
        self.decoding_bias(ctx)
        t_xent = min(t_target, tmax) if has_tgt else 0

        # fast path: max / argmax of every vocabulary tile come out of the logits GEMM's epilogue and
        # nm_greedy_finish turns them into the step's symbols, flags and the next input embedding -- the
        # [B,V] logits are neither re-read nor (unless a runner or the runtime loss wants them) written
        # (not when the loop runs in train mode with input dropout: nm_greedy_finish gathers the next input
        # embedding straight from the table, embed_input_symbols applies the dropout of autoregressive.py:199-214)
        drops = bool(ctx.fed(self.train_mode)) and self.dropout_keep_prob < 1.0
        use_stats = plain and graph_ok and not drops and self.logits_stats_ok(ctx, out_all[0])
        salts = self.sampling_salts(ctx, tmax) if sample else None
        stats = ctx.buffer(key + ("stats",), (ops.logits_stats_numel(bsz, v),)) if use_stats else None
        table = self.embedding_matrix(ctx)

        resident = getattr(stepper, "state_resident", False)     # the stepper keeps h_t in its own input row
        # (a resident stepper gets its OWN state row back for t > 0: step() then neither copies nor looks at the
        # initial state start() left pending -- a chunk captured later than chunk 0, whose Python body no longer
        # runs once it is a graph, must not bake a reset to s0 into its graph)
        h_prev_of = lambda t: s0 if t == 0 else (stepper.sel if resident else s_all[t - 1])

        # input tables (FusedStepper.table): the step reads what it needs of its input symbol from a table indexed by
        # the symbol itself -- nobody has to embed the symbols of the previous step
        tabled = getattr(stepper, "table", None) is not None
        ids_kw = (lambda t: {"ids": go if t == 0 else symbols[t - 1]}) if tabled else (lambda t: {})

        def body(t):
            """Step t touches persistent buffers only and depends on nothing but t (graph capturable)."""
            logits = logits_all[t] if keep_logits else logits_one
            st_t = [AttentionLoopState(st.contexts, st.weights, t) for st in att0]
            if use_stats:
                want_logits = keep_logits or t < t_xent
                stepper.step(emb, st_t, out_all[t], logits if want_logits else None, h_out=s_all[t],
                             h_prev=h_prev_of(t), stats=stats, **ids_kw(t))
                if t < t_xent:
                    ops.xent(logits, tgt[t], tmask[t], xent_rows[t])
                ops.greedy_finish(stats, v, finished, symbols[t], omask[t], END_TOKEN_INDEX, allfin[t:t + 1],
                                  table=None if tabled else table, emb_out=None if tabled else emb)
                return
            if graph_ok:
                stepper.step(emb, st_t, out_all[t], logits, h_out=s_all[t], h_prev=h_prev_of(t), **ids_kw(t))
            else:
                if indexed:
                    stepper.set_position(t, 0)
                stepper.step(emb, st_t, out_all[t], logits, h_out=s_all[t])
            if temperature != 1.0:                       # logits /= temperature (autoregressive.py:493)
                ops.ew("scale", logits, None, logits, alpha=1.0 / temperature)
            if sample:                                   # tf.multinomial(logits, 1) (:470-473)
                ops.gumbel_argmax(logits, salts[t], argmax)
            else:
                ops.row_stats(logits, None, None, argmax)
            if t < t_xent:
                ops.xent(logits, tgt[t], tmask[t], xent_rows[t])
            ops.greedy_update(argmax, finished, symbols[t], omask[t], END_TOKEN_INDEX, allfin[t:t + 1])
            if not (tabled and graph_ok):
                self.embed_input_symbols(ctx, symbols[t], out=emb)

        shape_key = tuple(tuple(st.weights.shape) for st in att0)
        graphed = (graph_ok or indexed) and not sample      # (a draw's salt is a launch argument: eager launches)
        chunk_key = key + ("chunk", bsz, t_xent, shape_key, getattr(stepper, "shape_key", ()))

        def launch(t0, n):
            def chunk():
                for t in range(t0, t0 + n):
                    body(t)
            if graphed:                  # one HIP graph per chunk of steps
                ctx.session.graphed(chunk_key + (t0, n), chunk)
            else:
                chunk()
        # the loop criterion (autoregressive.py:425-437) is evaluated on the device; the host reads the flags one
        # chunk behind what it has enqueued
        steps, _ = ctx.session.decode_chunks(tmax, CHECK_EVERY, launch, allfin, run_ahead=graphed)
        xent_sum = None
        if has_tgt:
            xent_sum = ctx.buffer(key + ("xent_sum",), (1,))
            ops.reduce_sum(xent_rows[:min(steps, t_target)].reshape(-1), xent_sum)
        att_states = [AttentionLoopState(st.contexts, st.weights, steps) for st in att_states]
        for att, st in zip(self.attentions, att_states):
            att.finalize_loop("{}_run".format(self.name), st)
        return RuntimeResult(symbols[:steps], omask[:steps], steps, xent_sum,
                             logits_all[:steps] if keep_logits else None, out_all[:steps], s_all[:steps],
                             [st.weights[:steps] for st in att_states])

    @tensor
    def runtime_loss(self, ctx):
        """sum(runtime xent over the cropped time) / sum(runtime_mask) (autoregressive.py:351-371)."""
        res = self.runtime_loop_result(ctx)
        if res.xent_sum is None:
            return 0.0
        return res.xent_sum[0] / res.mask.sum().to(torch.float32)

    @tensor
    def decoded_symbols(self, ctx) -> torch.Tensor:
        """[T,B] argmax symbols of the runtime loop (what GreedyRunner's host
        argmax over ``runtime_logprobs`` yields for a single session)."""
        return self.runtime_loop_result(ctx).symbols

    @tensor
    def runtime_mask(self, ctx) -> torch.Tensor:
        return self.runtime_loop_result(ctx).mask

    @tensor
    def runtime_logits(self, ctx) -> torch.Tensor:
        key = (id(self), "runtime_full")
        if key not in ctx.memo:
            ctx.memo[key] = self._runtime_loop(ctx, keep_logits=True)
        return ctx.memo[key].logits

    @tensor
    def runtime_logprobs(self, ctx) -> torch.Tensor:
        """[T,B,V] log-softmax of the runtime logits (ensembling path of GreedyRunner)."""
        logits = self.runtime_logits(ctx)
        t, b, v = logits.shape
        mx = ctx.buffer((id(self), "lp_max"), (t * b,))
        lse = ctx.buffer((id(self), "lp_lse"), (t * b,))
        ops.row_stats(logits.view(t * b, v), mx, lse, None)
        out = ctx.buffer((id(self), "logprobs"), (t, b, v))
        ops.log_softmax_from_stats(logits.view(t * b, v), mx, lse, out.view(t * b, v))
        return out

    @tensor
    def train_logprobs(self, ctx) -> torch.Tensor:
        """[T,B,V] log-softmax of the teacher-forced logits (autoregressive.py:288-290); the kernels of
        ``runtime_logprobs`` on ``train_logits``."""
        logits = self.train_logits(ctx)
        t, b, v = logits.shape
        logits = logits.contiguous()
        mx = ctx.buffer((id(self), "train_lp_max"), (t * b,))
        lse = ctx.buffer((id(self), "train_lp_lse"), (t * b,))
        ops.row_stats(logits.view(t * b, v), mx, lse, None)
        out = ctx.buffer((id(self), "train_logprobs"), (t, b, v))
        ops.log_softmax_from_stats(logits.view(t * b, v), mx, lse, out.view(t * b, v))
        return out

    @tensor
    def runtime_output_states(self, ctx) -> torch.Tensor:
        return self.runtime_loop_result(ctx).output_states

    @tensor
    def train_logits(self, ctx) -> torch.Tensor:
        res = self.train_loop_result(ctx)
        if res.saved["dlogits"] is not None:
            raise RuntimeError("train_logits were overwritten by their gradient in this run")
        steps, bsz = res.saved["steps"], res.saved["bsz"]
        if "tape" in res.saved:
            return res.saved["logits"].view(steps, bsz, -1)
        return ctx.buffer((id(self), "train", "logits"), (steps * bsz, len(self.vocabulary))) \
            .view(steps, bsz, -1)
