"""Transformer decoder (mirror of neuralmonkey/decoders/transformer.py).

Per layer (transformer.py:360-390): pre-LN masked self-attention + residual (:270-295), pre-LN
encoder-decoder attention per encoder, ``serial`` combination (:297-332,
attention/transformer_cross_layer.py:12-103), pre-LN ReLU feed-forward + residual (:334-358); a final
layer norm; logits = states . W + b with W tied to the embedding matrix by default
(autoregressive.py:226-251).  As in the reference at this commit, the decoder inputs carry no
position signal (``embed_input_symbol`` -- singular -- is dead code, :240-256).

Training is one pass over the shifted targets (:393-453) on the autodiff tape.

Decoding: the reference re-runs all layers over the whole prefix at every step (:487-516).  The
masked self-attention makes the states of earlier positions independent of later ones, so the
same numbers come out of a key/value cache: a step projects only the new position, appends its
keys / values to per-layer [R, Tmax, D] caches and attends with one query per row
(``nm_sdp_attn_fwd`` with Tq = 1).  The key mask column of a new position is ``not finished`` of its
row (:493-497); encoder keys / values are projected once per sentence and shared by the rows of a
beam (row r reads sentence r // k).
"""
import os
from typing import Any, List, NamedTuple, Optional, Union

import numpy as np
import torch

from .. import autodiff as F
from .. import ops
from ..attention.base_attention import Attendable, get_attention_mask, get_attention_states
from ..model.model_part import InitializerSpecs, ModelPart
from ..model.sequence import EmbeddedSequence
from ..nn import transformer_blocks as TB
from ..runtime import tensor
from ..variables import glorot_uniform_initializer, ones_initializer, zeros_initializer
from ..vocabulary import END_TOKEN_INDEX, START_TOKEN_INDEX, Vocabulary
from .autoregressive import AutoregressiveDecoder
from .decoder import CHECK_EVERY_TRANSFORMER, RuntimeResult, TrainResult

STRATEGIES = ["serial", "parallel", "flat", "hierarchical"]


# pylint: disable=too-many-instance-attributes
class TransformerDecoder(AutoregressiveDecoder):
    # pylint: disable=too-many-arguments,too-many-locals
    def __init__(self, name: str, encoders: List[Attendable], vocabulary: Vocabulary, data_id: str,
                 ff_hidden_size: int, n_heads_self: int, n_heads_enc: Union[List[int], int], depth: int,
                 max_output_len: int, attention_combination_strategy: str = "serial", n_heads_hier: int = None,
                 dropout_keep_prob: float = 1.0, embedding_size: int = None,
                 embeddings_source: EmbeddedSequence = None, tie_embeddings: bool = True,
                 label_smoothing: float = None, self_attention_dropout_keep_prob: float = 1.0,
                 attention_dropout_keep_prob: Union[float, List[float]] = 1.0,
                 use_att_transform_bias: bool = False, supress_unk: bool = False, reuse: ModelPart = None,
                 save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        self.encoders = encoders
        if embedding_size is None and embeddings_source is None:
            embedding_size = self._model_dimension(encoders, None)
        AutoregressiveDecoder.__init__(
            self, name=name, vocabulary=vocabulary, data_id=data_id, max_output_len=max_output_len,
            dropout_keep_prob=dropout_keep_prob, embedding_size=embedding_size,
            embeddings_source=embeddings_source, tie_embeddings=tie_embeddings, label_smoothing=label_smoothing,
            supress_unk=supress_unk, reuse=reuse, save_checkpoint=save_checkpoint,
            load_checkpoint=load_checkpoint, initializers=initializers)
        self.ff_hidden_size = ff_hidden_size
        self.n_heads_self = n_heads_self
        if isinstance(n_heads_enc, int):
            self.n_heads_enc = [n_heads_enc] if attention_combination_strategy == "flat" \
                else [n_heads_enc for _ in self.encoders]
        else:
            self.n_heads_enc = list(n_heads_enc)
        self.depth = depth
        if isinstance(attention_dropout_keep_prob, float):
            self.attention_dropout_keep_prob = [attention_dropout_keep_prob for _ in encoders]
        else:
            self.attention_dropout_keep_prob = list(attention_dropout_keep_prob)
        self.self_att_dropout_keep_prob = self_attention_dropout_keep_prob
        self.use_att_transform_bias = use_att_transform_bias
        self.attention_combination_strategy = attention_combination_strategy
        self.n_heads_hier = n_heads_hier
        self.attentions: List[Any] = []          # no Bahdanau-style loop states (runners look at this)
        if self.attention_combination_strategy not in STRATEGIES:
            raise ValueError("Unknown attention combination strategy '{}'. Allowed: {}."
                             .format(self.attention_combination_strategy, ", ".join(STRATEGIES)))
        if self.attention_combination_strategy == "hierarchical" and self.n_heads_hier is None:
            raise ValueError("You must provide n_heads_hier when using the hierarchical attention "
                             "combination strategy.")
        if self.attention_combination_strategy == "flat" and len(self.n_heads_enc) != 1:
            raise ValueError("For the flat attention combination strategy, only a single value is permitted "
                             "in n_heads_enc.")
        if self.attention_combination_strategy == "flat" and len(set(self.attention_dropout_keep_prob)) != 1:
            raise ValueError("For the flat attention combination strategy, the attention dropout must be the "
                             "same for all encoders.")
        if self.depth <= 0:
            raise ValueError("Depth must be a positive integer.")
        _ = self.dimension                        # dimension checks of the reference (:195-222)
        self.set_default_initializer(glorot_uniform_initializer())

    @staticmethod
    def _model_dimension(encoders, embedding_size) -> int:
        dims = [e.dimension for e in encoders]
        if dims:
            for i, dim in enumerate(dims):
                if dim != dims[0]:
                    raise ValueError("Dimension of the {}-th encoder ({}) differs from the dimension of the "
                                     "first one ({}).".format(i, dim, dims[0]))
            if embedding_size is not None and embedding_size != dims[0]:
                raise ValueError("Model dimension and input embedding size do not match")
            return dims[0]
        if embedding_size is None:
            raise ValueError("'embedding_size' must be specified when no encoders are provided")
        return embedding_size

    @property
    def dimension(self) -> int:
        return self._model_dimension(self.encoders, self.embedding_size)

    @property
    def output_dimension(self) -> int:
        return self.dimension

    # -- variables --------------------------------------------------------------------------------
    def declare_variables(self, store) -> None:
        AutoregressiveDecoder.declare_variables(self, store)
        d = self.dimension
        for i in range(self.depth):
            pre = "layer_{}".format(i)
            TB.declare_layer_norm(self, store, pre + "/self_attention", d)
            TB.declare_attention(self, store, pre + "/self_attention", d, self.n_heads_self,
                                 self.use_att_transform_bias)
            strategy = self.attention_combination_strategy
            if strategy == "flat":  # single() over the concatenated encoders, right in the sublayer's scope (:236-268)
                TB.declare_layer_norm(self, store, pre + "/encdec_attention", d)
                TB.declare_attention(self, store, pre + "/encdec_attention", d, self.n_heads_enc[0], False)
                TB.declare_feedforward(self, store, pre + "/feedforward", d, self.ff_hidden_size)
                continue
            if strategy != "serial":   # one normalisation of the queries for all encoders (:139, :182)
                TB.declare_layer_norm(self, store, pre + "/encdec_attention", d)
            for j, heads in enumerate(self.n_heads_enc):
                scope = "{}/encdec_attention/enc_{}".format(pre, j)
                if strategy == "serial":
                    TB.declare_layer_norm(self, store, scope, d)
                TB.declare_attention(self, store, scope, d, heads, False)       # single(): no transform bias
            if strategy == "hierarchical":
                TB.declare_attention(self, store, pre + "/encdec_attention/enc_hier", d, self.n_heads_hier, False)
            TB.declare_feedforward(self, store, pre + "/feedforward", d, self.ff_hidden_size)
        self.declare(store, "LayerNorm/gamma", (d,), ones_initializer())
        self.declare(store, "LayerNorm/beta", (d,), zeros_initializer())

    # -- the layer stack over whole sequences (training) ------------------------------------------------
    def _layers(self, tape: F.Tape, x: F.Var, mask: torch.Tensor, bsz: int, steps: int, enc, train: bool) -> F.Var:
        """layer(depth, inputs, mask) (:360-390).  x [B*T, D]; enc = [(states Var [B*S,D], mask, S)]."""
        ctx = tape.ctx
        keep = self.dropout_keep_prob
        for i in range(self.depth):
            pre = "layer_{}".format(i)
            site = (self.name, pre)
            normed = TB.layer_norm(tape, self, pre + "/self_attention", x)
            att = TB.multihead_attention(tape, self, pre + "/self_attention", normed, normed, mask,
                                         self.n_heads_self, bsz, steps, bsz, steps, True,
                                         self.self_att_dropout_keep_prob, train,
                                         ctx.salt(*site, "self_attention_weights"), self.use_att_transform_bias)
            att = F.dropout(tape, att, keep, train, ctx.salt(*site, "self_attention"))
            x = F.add(tape, att, x)
            x = self._encoder_attention(tape, x, enc, bsz, steps, train, pre, site)
            x = TB.feedforward_sublayer(tape, self, pre + "/feedforward", x, keep, train, site)
        return F.layer_norm(tape, x, tape.param(self, "LayerNorm/gamma"), tape.param(self, "LayerNorm/beta"))

    def _encoder_attention(self, tape: F.Tape, x: F.Var, enc, bsz: int, steps: int, train: bool, pre: str,
                           site) -> F.Var:
        """encoder_attention_sublayer (:297-340) over whole sequences, the four strategies of
        attention/transformer_cross_layer.py: ``serial`` (:68-103) norm + attend + dropout + residual per encoder,
        each on the previous result; ``parallel`` (:106-152) one normalised input queries every encoder, contexts
        and input are summed; ``hierarchical`` (:155-232) a second attention (``enc_hier``) of the same queries over
        the per-encoder contexts; ``flat`` (:236-268) one attention over the encoders concatenated in time."""
        ctx, keep, strategy = tape.ctx, self.dropout_keep_prob, self.attention_combination_strategy
        rows, d = bsz * steps, self.dimension
        top = pre + "/encdec_attention"
        if strategy == "flat":
            evar, emask, elen = concat_in_time(tape, enc, bsz, d, (id(self), "flat_mask", bsz))
            normed = TB.layer_norm(tape, self, top, x)
            att = TB.multihead_attention(tape, self, top, normed, evar, emask, self.n_heads_enc[0], bsz, steps, bsz,
                                         elen, False, self.attention_dropout_keep_prob[0], train,
                                         ctx.salt(*site, "encdec_weights", 0), False)
            return F.add(tape, F.dropout(tape, att, keep, train, ctx.salt(*site, "encdec", 0)), x)
        queries = None if strategy == "serial" else TB.layer_norm(tape, self, top, x)
        contexts = []
        for j, ((evar, emask, elen), heads, att_keep) in enumerate(zip(enc, self.n_heads_enc,
                                                                      self.attention_dropout_keep_prob)):
            scope = "{}/enc_{}".format(top, j)
            normed = queries if queries is not None else TB.layer_norm(tape, self, scope, x)
            att = TB.multihead_attention(tape, self, scope, normed, evar, emask, heads, bsz, steps, bsz, elen,
                                         False, att_keep, train, ctx.salt(*site, "encdec_weights", j), False)
            att = F.dropout(tape, att, keep, train, ctx.salt(*site, "encdec", j))
            if strategy == "hierarchical":
                contexts.append(att)
            else:
                x = F.add(tape, att, x)
        if strategy != "hierarchical":
            return x
        n = len(contexts)
        stacked = tape.view(F.concat(tape, contexts), lambda t: t.view(rows * n, d))    # [B*T, n, D]
        ones = ctx.buffer((id(self), "hier_mask", rows, n), (rows, n))
        ops.fill(ones, 1.0)
        # the weights of the second attention are dropped with dropout_keep_prob (:221-226)
        att = TB.multihead_attention(tape, self, top + "/enc_hier", queries, stacked, ones, self.n_heads_hier, rows, 1,
                                     rows, n, False, keep, train, ctx.salt(*site, "encdec_hier_weights"), False)
        return F.add(tape, F.dropout(tape, att, keep, train, ctx.salt(*site, "encdec_hier")), x)

    def _logit_params(self, tape: F.Tape):
        ctx = tape.ctx
        bias_data = self.decoding_bias(ctx)
        if self.tie_embeddings:
            return tape.named_param(self.embedding_matrix_name), True, tape.leaf(bias_data)
        b = tape.param(self, "state_to_word_b")
        return tape.param(self, "state_to_word_W"), False, F.Var(bias_data, b.grad, b.needs_grad)

    def _train_input_symbols(self, ctx) -> torch.Tensor:
        """[B,T]: <s> then the targets without their last step (:258-268)."""
        def shift(ids):                       # fed ids are [B,T]
            out = np.empty_like(ids)
            out[:, 0] = START_TOKEN_INDEX
            out[:, 1:] = ids[:, :-1]
            return out
        return self._staged(ctx, "tdec_in_bt", torch.int32, shift)

    def graph_safe_training(self, train_mode: bool) -> bool:
        return all(getattr(e, "graph_safe_training", lambda t: False)(train_mode) for e in self.encoders)

    def _staged(self, ctx, tag: str, dtype, derive=None) -> torch.Tensor:
        """Batch-major [B,T] view of the fed target ids in a persistent device buffer (one H2D copy per
        run; the buffer address is stable, so a captured training step can read it)."""
        key = (id(self), tag)
        if key not in ctx.memo:
            ctx.memo[key] = ctx.session.staged(key, ctx.session.to_device(ctx.fed(self.train_tokens), dtype, tag,
                                                                         derive))
        return ctx.memo[key]

    def stage_inputs(self, ctx) -> None:
        AutoregressiveDecoder.stage_inputs(self, ctx)
        if self.has_targets(ctx):
            self._train_input_symbols(ctx)
            self._staged(ctx, "tdec_tgt_bt", torch.int32)
            self._staged(ctx, "tdec_mask_bt", torch.float32, lambda a: (a != 0).astype(np.float32))

    def decoding_loop(self, ctx, train_mode: bool, sample: bool = False, temperature: float = 1.0):
        """autoregressive.py:527-562; ``sample`` / ``temperature`` as in ``Decoder.decoding_loop``."""
        self.check_sampling_args(train_mode, sample, temperature)
        if train_mode:
            return self._train_loop(ctx)
        if sample or temperature != 1.0:
            return self._runtime_loop(ctx, keep_logits=True, sample=sample, temperature=float(temperature))
        return self._runtime_loop(ctx, keep_logits=False)

    def _train_loop(self, ctx, want_grad: bool = False, grad_scale: Optional[torch.Tensor] = None) -> TrainResult:
        train = bool(ctx.fed(self.train_mode))
        tape = F.Tape(ctx, (id(self), "ttrain"), recording=want_grad)
        ids = self._train_input_symbols(ctx)                       # [B,T]
        bsz, steps = ids.shape
        tgt_bt = self._staged(ctx, "tdec_tgt_bt", torch.int32)
        mask_bt = self._staged(ctx, "tdec_mask_bt", torch.float32, lambda a: (a != 0).astype(np.float32))
        table = tape.named_param(self.embedding_matrix_name)
        emb = F.embedding(tape, table, ids.reshape(-1))            # base embed_input_symbols: lookup + dropout
        emb = F.dropout(tape, emb, self.dropout_keep_prob, train, ctx.salt(self.name, "embedded_input"))
        enc = []
        for e in self.encoders:
            st = get_attention_states(e, ctx)
            enc.append((tape.leaf(st.reshape(st.shape[0] * st.shape[1], st.shape[2]), needs_grad=True),
                        get_attention_mask(e, ctx), st.shape[1]))
        states = self._layers(tape, emb, mask_bt, bsz, steps, enc, train)
        w, trans_b, bias = self._logit_params(tape)
        logits = F.linear(tape, states, w, bias, trans_b=trans_b)   # [B*T, V], batch-major rows
        loss_rows = F.xent(tape, logits, tgt_bt.reshape(-1), self.xent_weights(mask_bt.reshape(-1)), grad_scale,
                           self.label_smoothing or 0.0)
        loss_sum = ctx.buffer((id(self), "ttrain", "loss_sum"), (1,))
        ops.reduce_sum(loss_rows, loss_sum)
        saved = {"tape": tape, "enc": enc, "steps": steps, "bsz": bsz, "logits": logits.data,
                 "dlogits": logits.data if want_grad else None, "states": states,
                 "loss_rows": loss_rows, "loss_layout": "bt"}
        return TrainResult(loss_sum, self.train_token_count(ctx), steps, saved)

    def backward(self, ctx, res: TrainResult) -> None:
        sv = res.saved
        sv["tape"].backward()
        bsz = sv["bsz"]
        for e, (var, _, slen) in zip(self.encoders, sv["enc"]):
            if var.grad is not None:
                ctx.defer_backward(e, var.grad.view(bsz, slen, -1), None)

    @tensor
    def train_loss(self, ctx) -> torch.Tensor:
        res = self.train_loop_result(ctx)
        out = ctx.buffer((id(self), "train_loss"), (1,))
        alpha = 1.0 / res.token_count if res.token_count else float("nan")         # (0 / 0 as the reference's division)
        return ops.ew("scale", res.loss_sum[0:1], None, out, alpha=alpha)[0]

    @tensor
    def train_logits(self, ctx) -> torch.Tensor:
        """[T,B,V] (time-major like every decoder history)."""
        res = self.train_loop_result(ctx)
        if res.saved["dlogits"] is not None:
            raise RuntimeError("train_logits were overwritten by their gradient in this run")
        return res.saved["logits"].view(res.saved["bsz"], res.saved["steps"], -1).transpose(0, 1)

    # -- decoding with a key/value cache ------------------------------------------------------------
    def make_stepper(self, ctx, rows: int, tag: str, rows_per_key: int = 1,
                     max_positions: int = 0) -> "TransformerStepper":
        return TransformerStepper(self, ctx, rows, tag, rows_per_key, max_positions)

    def _runtime_loop(self, ctx, keep_logits: bool, sample: bool = False, temperature: float = 1.0) -> RuntimeResult:
        key = (id(self), "trun", keep_logits)
        if sample or temperature != 1.0:
            key = key + ("sample" if sample else "argmax", temperature)
        bsz = int(ctx.fed(self.batch_size))
        d, v = self.dimension, len(self.vocabulary)
        tmax = self.max_output_len
        has_tgt = self.has_targets(ctx)
        if has_tgt:
            tgt, tmask = self.train_inputs(ctx), self.train_mask(ctx)
            t_target = tgt.shape[0]
        out_all = ctx.buffer(key + ("out_all",), (tmax, bsz, d))
        symbols = ctx.buffer(key + ("sym",), (tmax, bsz), torch.int32, zero=True)
        omask = ctx.buffer(key + ("mask",), (tmax, bsz), torch.int32, zero=True)
        finished = ctx.buffer(key + ("fin",), (bsz,), torch.int32, zero=True)
        allfin = ctx.buffer(key + ("allfin",), (tmax,), torch.int32)
        ops.fill(allfin, 1)
        argmax = ctx.buffer(key + ("argmax",), (bsz,), torch.int32)
        logits_all = ctx.buffer(key + ("logits_all",), (tmax, bsz, v)) if keep_logits else None
        logits_one = ctx.buffer(key + ("logits",), (bsz, v))
        xent_rows = ctx.buffer(key + ("xent_rows",), (tmax, bsz), zero=True) if has_tgt else None
        emb = ctx.buffer(key + ("emb",), (2, bsz, d))
        stepper = self.make_stepper(ctx, bsz, "greedy")
        stepper.start()
        go = ctx.buffer(key + ("go",), (bsz,), torch.int32)
        ops.fill(go, START_TOKEN_INDEX)
        self.embed_input_symbols(ctx, go, out=emb[0])
        self.decoding_bias(ctx)              # lazily built tensors: outside the captured chunks
        t_xent = min(t_target, tmax) if has_tgt else 0
        salts = self.sampling_salts(ctx, tmax) if sample else None

        def body(t):
            """Step t touches persistent buffers only and depends on nothing but t (graph capturable)."""
            logits = logits_all[t] if keep_logits else logits_one
            stepper.set_position(t, 0)
            stepper.step(emb[t & 1], [], out_all[t], logits, finished=finished)
            if temperature != 1.0:                       # logits /= temperature (autoregressive.py:493)
                ops.ew("scale", logits, None, logits, alpha=1.0 / temperature)
            if sample:                                   # tf.multinomial(logits, 1) (:470-473)
                ops.gumbel_argmax(logits, salts[t], argmax)
            else:
                ops.row_stats(logits, None, None, argmax)
            if t < t_xent:
                ops.xent(logits, tgt[t], tmask[t], xent_rows[t])
            ops.greedy_update(argmax, finished, symbols[t], omask[t], END_TOKEN_INDEX, allfin[t:t + 1])
            self.embed_input_symbols(ctx, symbols[t], out=emb[(t + 1) & 1])

        def launch(t0, n):
            def chunk():
                for t in range(t0, t0 + n):
                    body(t)
            # one HIP graph per chunk of steps; the host reads the finished flags one chunk behind
            if sample:                                   # (a draw's salt is a launch argument: eager launches)
                chunk()
            else:
                ctx.session.graphed(key + ("chunk", t0, n, bsz, t_xent, stepper.shape_key), chunk)
        steps, _ = ctx.session.decode_chunks(tmax, CHECK_EVERY_TRANSFORMER, launch, allfin, run_ahead=not sample)
        xent_sum = None
        if has_tgt:
            xent_sum = ctx.buffer(key + ("xent_sum",), (1,))
            ops.reduce_sum(xent_rows[:min(steps, t_target)].reshape(-1), xent_sum)
        return RuntimeResult(symbols[:steps], omask[:steps], steps, xent_sum,
                             logits_all[:steps] if keep_logits else None, out_all[:steps], None, [])

    @tensor
    def runtime_loss(self, ctx):
        res = self.runtime_loop_result(ctx)
        if res.xent_sum is None:
            return 0.0
        return res.xent_sum[0] / res.mask.sum().to(torch.float32)

    @tensor
    def decoded_symbols(self, ctx) -> torch.Tensor:
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


def concat_in_time(tape: F.Tape, enc, bsz: int, d: int, mask_key):
    """tf.concat(states, 1), tf.concat(masks, 1) of batch-major encoder states: (Var [B*S,D], mask [B,S], S)."""
    if len(enc) == 1:
        return enc[0]
    total = sum(elen for _, _, elen in enc)
    parts = [tape.view(evar, lambda t, n=elen: t.view(bsz, n * d)) for evar, _, elen in enc]
    states = tape.view(F.concat(tape, parts), lambda t: t.view(bsz * total, d))
    mask = tape.ctx.buffer(mask_key + (total,), (bsz, total))
    col = 0
    for _, emask, elen in enc:
        ops.ew("copy", emask, None, mask[:, col:col + elen])
        col += elen
    return states, mask, total


class TransformerStepper:
    """Cached decoding steps for R rows (greedy: R = B; beam: R = B*k, ``rows_per_key`` = k)."""

    def __init__(self, dec: TransformerDecoder, ctx, rows: int, tag: str, rows_per_key: int = 1,
                 max_positions: int = 0):
        self.dec, self.ctx, self.rows, self.rpk = dec, ctx, rows, rows_per_key
        d = dec.dimension
        self.tmax = max(dec.max_output_len, max_positions) + 1
        key = (id(dec), tag, rows, self.tmax)
        self.tape = F.Tape(ctx, key + ("tape",), recording=False)
        buf = lambda name, shape: ctx.buffer(key + (name,), shape)
        # Per layer: keys / values of the positions decoded so far.  Beam search does not move them: row r's history
        # is addressed through an ancestor table (``anc[r, j]`` = the cache row that holds position j of the
        # hypothesis row r continues; position t is always written to row r itself), and a beam step re-points the
        # rows -- one gather of a [rows, t] int32 table -- where it used to copy the cached prefixes of every
        # layer (12 gathers of up to 65 MB per step at the Transformer-base shape, 13 % of a beam step).
        # NM_KV_ANCESTORS=0: two cache copies and the gathers.
        # (the table is read by the wave-per-(row, head) kernel, which wants head widths of 4 .. 256 floats in powers
        # of two; anything else -- tests/transformer.ini has heads of 2 -- keeps the gathers)
        self.use_anc = (tag.startswith("beam") and os.environ.get("NM_KV_ANCESTORS", "1") != "0" and
                        d % dec.n_heads_self == 0 and d // dec.n_heads_self in (4, 8, 16, 32, 64, 128, 256))
        copies = 1 if (self.use_anc or not tag.startswith("beam")) else 2
        self.kcache = [buf(("k", l), (copies, rows, self.tmax, d)) for l in range(dec.depth)]
        self.vcache = [buf(("v", l), (copies, rows, self.tmax, d)) for l in range(dec.depth)]
        self.anc = ctx.buffer(key + ("anc",), (2, rows, self.tmax), torch.int32) if self.use_anc else None
        self.mask = buf("mask", (2, rows, self.tmax))
        self.hier_ones = buf("hier_ones", (rows, len(dec.encoders)))
        ops.fill(self.hier_ones, 1.0)
        self.cur = 0
        self.t = 0
        self.enc_kv = None

    def start(self) -> None:
        """Project the encoder states to keys / values of every layer once per batch."""
        dec, tape = self.dec, self.tape
        tape._n, tape._slot = 0, 2                # pylint: disable=protected-access
        self.enc_kv = []
        flat = dec.attention_combination_strategy == "flat"
        sources = []
        for e in dec.encoders:
            st = get_attention_states(e, self.ctx)
            bk, slen, d = st.shape
            sources.append((tape.leaf(st.reshape(bk * slen, d)), get_attention_mask(e, self.ctx), slen))
        if flat:
            sources = [concat_in_time(tape, sources, bk, d, (id(dec), "flat_mask_step", bk))]
        for j, ((var, emask, slen), heads) in enumerate(zip(sources, dec.n_heads_enc)):
            per_layer = []
            for l in range(dec.depth):
                scope = "layer_{}/encdec_attention".format(l) if flat else \
                    "layer_{}/encdec_attention/enc_{}".format(l, j)
                k = TB.project(tape, dec, scope, "keys_proj", var, heads, False)
                v = TB.project(tape, dec, scope, "vals_proj", var, heads, False)
                per_layer.append((k.data.view(bk, slen, d), v.data.view(bk, slen, d)))
            self.enc_kv.append((per_layer, emask, bk, slen))
        self.base = tape._n                       # pylint: disable=protected-access
        self.cur, self.t = 0, 0
        self._prepare_qkv()
        if self.anc is not None:                  # every row starts as its own ancestor at every position
            iota = torch.arange(self.rows, dtype=torch.int32, device=self.anc.device).view(1, self.rows, 1)
            ops.copy(self.anc, iota.expand(2, self.rows, self.tmax))

    def _prepare_qkv(self) -> None:
        """The three self-attention projections of a layer read the same normed rows: ONE launch of nm_step_group
        with three problems (queries, the new key row, the new value row) instead of three products -- 12 launches
        of ~6 us fewer per step at depth 6 (decoders/transformer.py:282-300 of the reference: the same three
        tf.layers.dense calls).  nm_step_group multiplies by [N, K] weights: the kernels are transposed once per
        batch (they change between batches only when somebody trains in between).  NM_STEP_QKV=0: three products."""
        dec, ctx = self.dec, self.ctx
        self.qkv = None
        d = dec.dimension
        if (os.environ.get("NM_STEP_QKV", "1") == "0" or dec.n_heads_self <= 1 or d % 16 != 0
                or self.kcache[0].device.type != "cuda"):
            return
        key = (id(dec), "qkvT", self.rows)
        self.qkv = []
        for l in range(dec.depth):
            scope = "layer_{}/self_attention".format(l)
            wts, biases = [], []
            for proj in ("query_proj", "keys_proj", "vals_proj"):
                w = dec.var(ctx, "{}/{}/kernel".format(scope, proj))                 # [d, d]
                wt = ctx.buffer(key + (l, proj), (w.shape[1], w.shape[0]))
                ops.copy(wt, w.t())
                wts.append(wt)
                biases.append(dec.var(ctx, "{}/{}/bias".format(scope, proj)) if dec.use_att_transform_bias else None)
            qbuf = ctx.buffer(key + (l, "q"), (self.rows, d))
            probs = []
            for wt, bias in zip(wts, biases):
                spec = dict(A=qbuf, lda=d, Bt=wt, ldb=wt.stride(0), N=d, K=d, epilogue=0, C=qbuf, ldc=d)
                if bias is not None:
                    spec["bias"] = bias
                probs.append(spec)
            self.qkv.append((ops.StepGroup(self.rows, probs), qbuf))

    def _self_projections(self, l: int, normed, t: int, kc, vc):
        """q (a tape variable) and row t of the layer's key / value caches from the normed input rows."""
        dec, tape = self.dec, self.tape
        scope = "layer_{}/self_attention".format(l)
        if self.qkv is not None:
            group, qbuf = self.qkv[l]
            src = normed.data
            for i in range(3):
                group.patch(i, A=src, lda=src.stride(0))
            group.patch(1, C=kc[:, t], ldc=kc.stride(0))
            group.patch(2, C=vc[:, t], ldc=vc.stride(0))
            group.launch()
            return tape.leaf(qbuf)
        q = TB.project(tape, dec, scope, "query_proj", normed, dec.n_heads_self, dec.use_att_transform_bias)
        if dec.n_heads_self > 1:
            bias = lambda p: tape.param(dec, "{}/{}/bias".format(scope, p)) if dec.use_att_transform_bias else None
            F.linear(tape, normed, tape.param(dec, scope + "/keys_proj/kernel"), bias("keys_proj"),
                     out=tape.leaf(kc[:, t]))
            F.linear(tape, normed, tape.param(dec, scope + "/vals_proj/kernel"), bias("vals_proj"),
                     out=tape.leaf(vc[:, t]))
        else:
            ops.ew("copy", normed.data, None, kc[:, t])
            ops.ew("copy", normed.data, None, vc[:, t])
        return q

    indexed = True       # set_position(t, cur) makes a step a function of its index: HIP-graph capturable

    @property
    def shape_key(self):
        """What a captured step depends on besides its index: the encoder batches it attends to."""
        return tuple((bk, slen) for _, _, bk, slen in self.enc_kv)

    def set_position(self, t: int, cur: int) -> None:
        """Decode position ``t`` next, reading cache copy ``cur`` (graph replays skip the Python-side
        bookkeeping of ``step`` / ``reorder``, so captured callers state the position explicitly)."""
        self.t, self.cur = t, cur

    def step(self, emb, att_states, out_state, logits, h_out=None, finished=None):
        dec, tape, rows, t = self.dec, self.tape, self.rows, self.t
        assert t < self.tmax, "decoding ran past the key/value cache"
        d = dec.dimension
        tape._n, tape._slot = self.base, 0        # pylint: disable=protected-access
        cur = self.cur
        mask = self.mask[cur]
        ops.unfinished_mask(finished, mask[:, t])                               # :493-497
        x = tape.leaf(emb)
        if len(self.enc_kv) == 1 and dec.attention_combination_strategy != "hierarchical" and \
                os.environ.get("NM_STEP_FUSE_LN", "1") != "0":
            self._step_fused(x, t, cur, mask, out_state)
            dec.state_to_logits(self.ctx, out_state, logits)
            self.t += 1
            return att_states
        for l in range(dec.depth):
            pre = "layer_{}".format(l)
            scope = pre + "/self_attention"
            normed = TB.layer_norm(tape, dec, scope, x)
            q = TB.project(tape, dec, scope, "query_proj", normed, dec.n_heads_self, dec.use_att_transform_bias)
            kc, vc = self.kcache[l][cur if len(self.kcache[l]) > 1 else 0], self.vcache[l][cur if len(self.vcache[l]) > 1 else 0]
            if dec.n_heads_self > 1:
                bias = lambda p: tape.param(dec, "{}/{}/bias".format(scope, p)) if dec.use_att_transform_bias else None
                F.linear(tape, normed, tape.param(dec, scope + "/keys_proj/kernel"), bias("keys_proj"),
                         out=tape.leaf(kc[:, t]))
                F.linear(tape, normed, tape.param(dec, scope + "/vals_proj/kernel"), bias("vals_proj"),
                         out=tape.leaf(vc[:, t]))
            else:
                ops.ew("copy", normed.data, None, kc[:, t])
                ops.ew("copy", normed.data, None, vc[:, t])
            att = F.sdp_attention(tape, q, None, None, mask[:, :t + 1], dec.n_heads_self, rows, 1, rows, t + 1,
                                  False, 1.0, 0, k_data=kc[:, :t + 1], v_data=vc[:, :t + 1],
                                  ancestors=self.anc[cur] if self.anc is not None else None)
            att = TB.project(tape, dec, scope, "output_proj", att, dec.n_heads_self, dec.use_att_transform_bias)
            x = F.add(tape, att, x)
            strategy = dec.attention_combination_strategy
            top = pre + "/encdec_attention"
            queries = None if strategy == "serial" else TB.layer_norm(tape, dec, top, x)
            contexts = []
            for j, (heads, (per_layer, emask, bk, slen)) in enumerate(zip(dec.n_heads_enc, self.enc_kv)):
                scope = top if strategy == "flat" else "{}/enc_{}".format(top, j)
                normed = queries if queries is not None else TB.layer_norm(tape, dec, scope, x)
                q = TB.project(tape, dec, scope, "query_proj", normed, heads, False)
                ek, ev = per_layer[l]
                att = F.sdp_attention(tape, q, None, None, emask, heads, rows, 1, bk, slen, False, 1.0, 0,
                                      k_data=ek, v_data=ev)
                att = TB.project(tape, dec, scope, "output_proj", att, heads, False)
                if strategy == "hierarchical":
                    contexts.append(att)
                else:
                    x = F.add(tape, att, x)
            if strategy == "hierarchical":
                n = len(contexts)
                stacked = tape.view(F.concat(tape, contexts), lambda t, n=n: t.view(rows * n, d))
                att = TB.multihead_attention(tape, dec, top + "/enc_hier", queries, stacked, self.hier_ones,
                                             dec.n_heads_hier, rows, 1, rows, n, False, 1.0, False, 0, False)
                x = F.add(tape, att, x)
            x = TB.feedforward_sublayer(tape, dec, pre + "/feedforward", x, 1.0, False, (dec.name, pre))
        ops.layer_norm_fwd(x.data, dec.var(self.ctx, "LayerNorm/gamma"), dec.var(self.ctx, "LayerNorm/beta"),
                           out=out_state)
        dec.state_to_logits(self.ctx, out_state, logits)
        self.t += 1
        return att_states

    def _step_fused(self, x, t: int, cur: int, mask, out_state) -> None:
        """The layers of one step for ONE encoder (decoders/transformer.py:270-358), with every residual connection
        and the layer norm that follows it in one launch (``F.add_layer_norm``) and the feed-forward ReLU in its
        product's epilogue: 24 launches of ~5 us fewer per step than the sub-layer-by-sub-layer formulation in
        ``step``, element for element the same arithmetic."""
        dec, tape, rows = self.dec, self.tape, self.rows
        ln = lambda scope: (tape.param(dec, scope + "/LayerNorm/gamma"), tape.param(dec, scope + "/LayerNorm/beta"))
        strategy = dec.attention_combination_strategy
        per_layer, emask, bk, slen = self.enc_kv[0]
        heads_enc = dec.n_heads_enc[0]
        normed = TB.layer_norm(tape, dec, "layer_0/self_attention", x)
        for l in range(dec.depth):
            pre = "layer_{}".format(l)
            scope = pre + "/self_attention"
            kc, vc = self.kcache[l][cur if len(self.kcache[l]) > 1 else 0], self.vcache[l][cur if len(self.vcache[l]) > 1 else 0]
            q = self._self_projections(l, normed, t, kc, vc)
            att = F.sdp_attention(tape, q, None, None, mask[:, :t + 1], dec.n_heads_self, rows, 1, rows, t + 1,
                                  False, 1.0, 0, k_data=kc[:, :t + 1], v_data=vc[:, :t + 1],
                                  ancestors=self.anc[cur] if self.anc is not None else None)
            att = TB.project(tape, dec, scope, "output_proj", att, dec.n_heads_self, dec.use_att_transform_bias)
            top = pre + "/encdec_attention"
            scope = top if strategy == "flat" else top + "/enc_0"
            x, normed = F.add_layer_norm(tape, att, x, *ln(scope if strategy == "serial" else top))
            q = TB.project(tape, dec, scope, "query_proj", normed, heads_enc, False)
            ek, ev = per_layer[l]
            att = F.sdp_attention(tape, q, None, None, emask, heads_enc, rows, 1, bk, slen, False, 1.0, 0,
                                  k_data=ek, v_data=ev)
            att = TB.project(tape, dec, scope, "output_proj", att, heads_enc, False)
            scope = pre + "/feedforward"
            x, normed = F.add_layer_norm(tape, att, x, *ln(scope))
            hidden = F.linear(tape, normed, tape.param(dec, scope + "/hidden_state/kernel"),
                              tape.param(dec, scope + "/hidden_state/bias"), act="relu")
            out = F.linear(tape, hidden, tape.param(dec, scope + "/output/kernel"), tape.param(dec, scope + "/output/bias"))
            if l + 1 < dec.depth:
                x, normed = F.add_layer_norm(tape, out, x, *ln("layer_{}/self_attention".format(l + 1)))
            else:
                total = tape.new(tuple(x.shape))
                ops.add_layer_norm_fwd(out.data, x.data, dec.var(self.ctx, "LayerNorm/gamma"),
                                       dec.var(self.ctx, "LayerNorm/beta"), total.data, out_state)

    def reorder(self, src_rows: torch.Tensor) -> None:
        """Beam step: row r continues hypothesis ``src_rows[r]`` -- gather the cached prefix."""
        cur, nxt, t = self.cur, self.cur ^ 1, self.t
        d = self.dec.dimension
        width = t * d
        if self.anc is not None:                  # (int32 rows moved as their bit patterns)
            ops.gather_rows(self.anc[cur].view(torch.float32)[:, :t], src_rows, self.anc[nxt].view(torch.float32)[:, :t])
        else:
            for l in range(self.dec.depth):
                for cache in (self.kcache[l], self.vcache[l]):
                    ops.gather_rows(cache[cur].view(self.rows, self.tmax * d)[:, :width], src_rows,
                                    cache[nxt].view(self.rows, self.tmax * d)[:, :width])
        ops.gather_rows(self.mask[cur][:, :t], src_rows, self.mask[nxt][:, :t])
        self.cur = nxt
