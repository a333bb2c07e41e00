"""In-graph-equivalent beam search (mirror of
neuralmonkey/decoders/beam_search_decoder.py).

Per step (body, beam_search_decoder.py:394-556):
  masked log-probs -> + logprob_sum -> / length penalty -> top-k over [B, k*V]
  -> div/mod -> gather search state -> reorder parent state & token history
  -> embed the chosen words -> parent decoder step -> new log-probs.
Everything up to the gathers is the fused HIP kernel pair behind
``nm_beam_topk_step`` reading the parent's raw logits plus their (max, lse)
row statistics, so the [B,k,V] log-softmax tensor is never materialised.

The reference cannot tile Bahdanau keys to the beam and therefore runs RNN
beam search at batch 1 only.  Here hypothesis row r reads the
keys of sentence r // k, which is the same arithmetic for batch 1 and lifts
the restriction for any batch.
"""
from typing import Any, List, NamedTuple, Optional

import numpy as np
import torch

from .. import ops
from ..model.model_part import ModelPart
from ..runtime import Placeholder, tensor
from ..vocabulary import END_TOKEN_INDEX, PAD_TOKEN_INDEX, START_TOKEN_INDEX, Vocabulary
from .autoregressive import AutoregressiveDecoder, DecoderFeedables, LoopState
from .decoder import CHECK_EVERY, CHECK_EVERY_BEAM

INF = 1e9


class SearchState(NamedTuple):
    """beam_search_decoder.py:45-70."""
    logprob_sum: torch.Tensor        # [B,k]
    prev_logprobs: Optional[torch.Tensor]   # [B,k,V]; materialised only for ensembles
    lengths: torch.Tensor            # [B,k] int32
    finished: torch.Tensor           # [B,k] int32 (0/1)


class SearchResults(NamedTuple):
    """beam_search_decoder.py:73-89."""
    scores: torch.Tensor             # [B,k]
    token_ids: torch.Tensor          # [steps+1,B,k]


class BeamSearchLoopState(NamedTuple):
    search_state: SearchState
    search_results: SearchResults
    decoder_loop_state: Any


class BeamSearchOutput(NamedTuple):
    """beam_search_decoder.py:112-126."""
    last_search_step_output: SearchResults
    last_dec_loop_state: Any
    last_search_state: SearchState
    attention_loop_states: List[Any]


class BeamSearchDecoder(ModelPart):
    def __init__(self, name: str, parent_decoder: AutoregressiveDecoder, beam_size: int, max_steps: int,
                 length_normalization: float) -> None:
        ModelPart.__init__(self, name)
        self.parent_decoder = parent_decoder
        self.beam_size = beam_size
        self.length_normalization = length_normalization
        self.max_steps_int = max_steps
        self.max_steps = Placeholder("{}/max_steps".format(name), default=max_steps)
        if beam_size < 1 or beam_size > 16:
            raise ValueError("beam_size must be between 1 and 16 for the HIP top-k kernels, was {}"
                             .format(beam_size))
        if not hasattr(parent_decoder, "make_stepper"):
            raise NotImplementedError("BeamSearchDecoder: parent decoder '{}' has no stepwise inference "
                                      "interface (make_stepper)".format(type(parent_decoder).__name__))

    @property
    def vocabulary(self) -> Vocabulary:
        return self.parent_decoder.vocabulary

    def _length_penalty_table(self, ctx, max_len: int) -> torch.Tensor:
        key = (id(self), "penalty", max_len)
        if key not in ctx.session.__dict__.setdefault("_const", {}):
            ctx.session._const[key] = ops.length_penalty_table(max_len, self.length_normalization,
                                                               ctx.device)
        return ctx.session._const[key]

    def expand_index(self, ctx, bsz: int) -> torch.Tensor:
        """expand_to_beam (beam_search_decoder.py:575-596) as a row index: r -> r // k."""
        key = (id(self), "expand", bsz)
        if key not in ctx.session.__dict__.setdefault("_const", {}):
            idx = (np.arange(bsz * self.beam_size) // self.beam_size).astype(np.int32)
            ctx.session._const[key] = torch.from_numpy(idx).to(ctx.device)
        return ctx.session._const[key]

    @tensor
    def outputs(self, ctx) -> BeamSearchOutput:
        dec = self.parent_decoder
        for att in dec.attentions:
            att.rows_per_key = self.beam_size
        try:
            return self._search(ctx)
        finally:
            for att in dec.attentions:
                att.rows_per_key = 1

    @tensor
    def selection_history(self, ctx):
        """(parent beam [steps,B,k], word [steps,B,k]) of every executed beam body (:481-483): the raw
        selections the token histories are traced back from -- what a parity check needs to tell a flipped
        near-tie from a wrong selection."""
        self.outputs(ctx)
        src, word, steps, bsz = ctx.memo[(id(self), "_raw_selections")]
        k = self.beam_size
        beam = src[:steps].view(steps, bsz, k) - (torch.arange(bsz, device=src.device, dtype=src.dtype) * k).view(1, bsz, 1)
        return beam, word[:steps].view(steps, bsz, k)

    def ensemble_outputs(self, ctxs: List[Any]) -> BeamSearchOutput:
        """Beam search over an ensemble: one run context (= one set of variables) per model, all on
        this device.  Every step each model advances its own decoder state on the SAME hypotheses;
        the step distributions are averaged in log space,
        ``logsumexp_m(log_softmax(logits_m)) - log M`` (runners/beamsearch_runner.py:47-55), and the
        shared beam body (:394-556) selects on that average.  The reference drives this from the host
        with one Session.run per step and model; here the whole search stays on the device."""
        dec = self.parent_decoder
        for att in dec.attentions:
            att.rows_per_key = self.beam_size
        try:
            return self._search_ensemble(ctxs)
        finally:
            for att in dec.attentions:
                att.rows_per_key = 1

    def _search_ensemble(self, ctxs: List[Any]) -> BeamSearchOutput:
        import math
        from ..attention.base_attention import AttentionLoopState
        dec, k, nmod = self.parent_decoder, self.beam_size, len(ctxs)
        ctx0 = ctxs[0]
        bsz = int(ctx0.fed(dec.batch_size))
        rows = bsz * k
        e, v = dec.embedding_size or dec.output_dimension, len(self.vocabulary)
        max_steps = int(ctx0.fed(self.max_steps))
        key = (id(self), "bs_ens", bsz, nmod)
        f32 = lambda name, shape, **kw: ctx0.buffer(key + (name,), shape, torch.float32, **kw)
        i32 = lambda name, shape, **kw: ctx0.buffer(key + (name,), shape, torch.int32, **kw)
        tok = i32("tok", (2, max_steps + 1, rows))
        lps, lens, fin = f32("lps", (2, bsz, k)), i32("lens", (2, bsz, k)), i32("fin", (2, bsz, k))
        scores = f32("scores", (bsz, k), zero=True)
        word, beam, src = i32("word", (bsz, k)), i32("beam", (bsz, k)), i32("src", (bsz, k))
        allfin = i32("allfin", (max(max_steps, 1),))
        ops.fill(allfin, 1)
        ws = ctx0.buffer(key + ("ws",), ((ops._lib.load().nm_beam_workspace_bytes(bsz, k, v) + 3) // 4,))
        penalty = self._length_penalty_table(ctx0, max_steps + 2)
        zero_stat = f32("zero_stat", (rows,), zero=True)      # the averaged log-probs need no max / lse shift
        ens = f32("ens_logprobs", (rows, v))
        lp_m = f32("lp_m", (rows, v))
        rmax, rlse, argmax = f32("rmax", (rows,)), f32("rlse", (rows,)), i32("argmax", (rows,))
        go = i32("go", (rows,))
        ops.fill(go, START_TOKEN_INDEX)
        models = []
        for m, ctx in enumerate(ctxs):
            stepper = dec.make_stepper(ctx, rows, "beam_ens", k, max_positions=max_steps + 1)
            emb = getattr(stepper, "emb_view", None)
            if emb is None:
                emb = ctx.buffer(key + ("emb", m), (rows, e))
            models.append({"ctx": ctx, "stepper": stepper, "emb": emb,
                           "out": ctx.buffer(key + ("out", m), (rows, dec.output_dimension)),
                           "logits": ctx.buffer(key + ("logits", m), (rows, v)),
                           "att": [a.initial_loop_state(ctx, rows, max_steps + 1) for a in dec.attentions]})

        def average(first_symbols: bool) -> None:
            """ens = log mean_m softmax(logits_m)."""
            for m, mod in enumerate(models):
                ops.row_stats(mod["logits"], rmax, rlse, argmax if (first_symbols and m == 0) else None)
                ops.log_softmax_from_stats(mod["logits"], rmax, rlse, ens if m == 0 else lp_m)
                if m > 0:
                    ops.ew("logaddexp", ens, lp_m, ens)
            if nmod > 1:
                ops.ew("add_scalar", ens, None, ens, alpha=-math.log(nmod))

        # ---- initial parent step of every model on the tiled rows (:218-328)
        ops.zero(fin[0])
        for mod in models:
            ctx, stepper = mod["ctx"], mod["stepper"]
            if hasattr(dec, "initial_state"):
                hsel = ctx.buffer(key + ("hsel",), (rows, dec.rnn_size))
                ops.gather_rows(dec.initial_state(ctx), self.expand_index(ctx, bsz), hsel)
                stepper.start(hsel)
            else:
                stepper.start()
            dec.embed_input_symbols(ctx, go, out=mod["emb"])
            mod["att"] = stepper.step(mod["emb"], mod["att"], mod["out"], mod["logits"], finished=fin[0].view(rows))
        average(first_symbols=True)
        ops.copy(tok[0, 0], argmax)
        ops.fill(lps[0], -INF)
        lps[0, :, 0] = 0.0
        ops.zero(lens[0])
        srcf, wordf = src.view(rows), word.view(rows)
        steps = executed = 0
        while steps < max_steps:
            cur, nxt = steps & 1, (steps & 1) ^ 1
            ops.beam_topk_step(ens, bsz, k, zero_stat, zero_stat, lps[cur], lens[cur], fin[cur], penalty,
                               END_TOKEN_INDEX, scores, word, beam, lps[nxt], lens[nxt], fin[nxt], src, ws,
                               allfin[steps:steps + 1])
            ops.beam_reorder_tokens(tok[cur], srcf, wordf, tok[nxt], steps + 1, rows)
            for mod in models:
                mod["stepper"].reorder(srcf)
                dec.embed_input_symbols(mod["ctx"], wordf, out=mod["emb"])
                mod["att"] = mod["stepper"].step(mod["emb"], mod["att"], mod["out"], mod["logits"],
                                                 finished=fin[nxt].view(rows))
            average(first_symbols=False)
            steps += 1
            executed = steps
            if steps % CHECK_EVERY == 0 or steps == max_steps:
                done = np.nonzero(ctx0.session.read_small(allfin[:steps]))[0]
                if done.size:
                    steps = int(done[0]) + 1
                    break
        cur = executed & 1
        token_ids = tok[cur, :steps + 1].view(steps + 1, bsz, k)
        search_state = SearchState(lps[cur], None, lens[cur], fin[cur])
        feedables = DecoderFeedables(step=steps + 1, finished=fin[cur].view(rows), embedded_input=models[0]["emb"],
                                     other=None)
        return BeamSearchOutput(SearchResults(scores, token_ids),
                                LoopState(histories=None, constants=None, feedables=feedables), search_state, [])

    def _search(self, ctx) -> BeamSearchOutput:
        dec = self.parent_decoder
        k = self.beam_size
        bsz = int(ctx.fed(dec.batch_size))
        rows = bsz * k
        e, v = dec.embedding_size or dec.output_dimension, len(self.vocabulary)
        max_steps = int(ctx.fed(self.max_steps))
        key = (id(self), "bs", bsz)
        f32 = lambda name, shape, **kw: ctx.buffer(key + (name,), shape, torch.float32, **kw)
        i32 = lambda name, shape, **kw: ctx.buffer(key + (name,), shape, torch.int32, **kw)

        stepper = dec.make_stepper(ctx, rows, "beam", k, max_positions=max_steps + 1)
        emb = getattr(stepper, "emb_view", None)          # embed straight into the stepper's input slot
        if emb is None:
            emb = f32("emb", (rows, e))
        out_state = f32("out", (rows, dec.output_dimension))
        logits = f32("logits", (rows, v))
        rmax, rlse = f32("rmax", (rows,)), f32("rlse", (rows,))
        argmax = i32("argmax", (rows,))
        tok = i32("tok", (max_steps + 1, rows))           # token history [steps+1, B*k], built once at the end
        src_hist = i32("src_hist", (max(max_steps, 1), rows))      # back-pointers and words of every body
        word_hist = i32("word_hist", (max(max_steps, 1), rows))
        lps = f32("lps", (2, bsz, k))
        lens = i32("lens", (2, bsz, k))
        # the first step's summed log-probabilities: 0 for hypothesis 0, -INF for the copies (built once per shape)
        firsts = ctx.session.__dict__.setdefault("_beam_lps_first", {})
        lps_first = firsts.get((bsz, k))
        if lps_first is None:
            lps_first = torch.full((bsz, k), -INF, dtype=torch.float32, device=lps.device)
            lps_first[:, 0] = 0.0
            firsts[(bsz, k)] = lps_first
        fin = i32("fin", (2, bsz, k))
        scores = f32("scores", (bsz, k), zero=True)
        word, beam, src = i32("word", (bsz, k)), i32("beam", (bsz, k)), i32("src", (bsz, k))
        allfin = i32("allfin", (max(max_steps, 1),))
        ops.fill(allfin, 1)
        ws = ctx.buffer(key + ("ws",), ((ops._lib.load().nm_beam_workspace_bytes(bsz, k, v) + 3) // 4,))
        penalty = self._length_penalty_table(ctx, max_steps + 2)
        att_states = [a.initial_loop_state(ctx, rows, max_steps + 1) for a in dec.attentions]

        # ---- get_initial_loop_state (:218-328): tile, run the parent body once
        fast = getattr(stepper, "graph_safe", False)      # steps keep no Python-side state: HIP-graph capturable
        indexed = getattr(stepper, "indexed", False)      # steps are a function of an explicit position: also capturable
        att0 = att_states
        from ..attention.base_attention import AttentionLoopState
        att_at = lambda i: [AttentionLoopState(a.contexts, a.weights, i) for a in att0]
        # fast path: the vocabulary projection leaves per-tile row statistics behind (nm_logits_stats_gemm) and
        # the beam body reads back only the tiles that can hold a top-k candidate (nm_beam_topk_step_tiles)
        use_stats = fast and dec.logits_stats_ok(ctx, out_state)
        stats = f32("stats", (ops.logits_stats_numel(rows, v),)) if use_stats else None
        tabled = fast and getattr(stepper, "table", None) is not None       # input tables: steps take symbols
        go = i32("go", (rows,))
        first_sym = i32("first_sym", (rows,))
        hsel = s0 = None
        if hasattr(dec, "initial_state"):                 # RNN decoder: tile the initial state (:575-596)
            hsel = f32("hsel", (rows, dec.rnn_size))
            s0 = dec.initial_state(ctx)
        expand = self.expand_index(ctx, bsz)

        def initial_step():
            """Everything up to the first beam body: persistent buffers only, a function of nothing but the encoder
            side of the batch -- on the fast path ONE HIP graph instead of ~20 launches issued from Python in
            front of every search (0.2-0.3 ms of an 18 ms batch during which the GPU mostly waited)."""
            if hsel is not None:
                ops.gather_rows(s0, expand, hsel)
                stepper.start(hsel)
            else:
                stepper.start()
            ops.zero(fin[0])
            ops.fill(go, START_TOKEN_INDEX)
            if not (fast and tabled):
                dec.embed_input_symbols(ctx, go, out=emb)
            if fast:
                stepper.step(emb, att_at(0), out_state, logits, h_prev=hsel, h_out=stepper.hbuf[0], stats=stats,
                             **({"ids": go} if tabled else {}))
            else:
                loop0["att"] = stepper.step(emb, att_states, out_state, logits, finished=fin[0].view(rows))
            ops.row_stats(logits, rmax, rlse, argmax)
            ops.copy(first_sym, argmax)                   # parent's greedy symbol, dropped by the runner
            ops.copy(lps[0], lps_first)                   # 0 for hypothesis 0, -INF for the rest
            ops.zero(lens[0])
        loop0 = {"att": att_states}
        if fast:
            for att in dec.attentions:                    # lazily built tensors (H2D copies): outside the capture
                att.hidden_features(ctx)
                att.attention_mask(ctx)
            dec.decoding_bias(ctx)
            ctx.session.graphed(key + ("init", k, v, tuple(tuple(a.weights.shape) for a in att0),
                                       getattr(stepper, "shape_key", ()), 0 if s0 is None else s0.data_ptr(),
                                       bool(tabled), bool(use_stats)), initial_step)
        else:
            initial_step()
        att_states = loop0["att"]
        loop = {"att": att_states}

        def body(s):
            """Beam body number s (:394-556); every buffer it touches is a function of s alone."""
            cur, nxt = s & 1, (s & 1) ^ 1
            srcf, wordf = src_hist[s], word_hist[s]       # this body's selections ARE the history records
            src, word = srcf.view(bsz, k), wordf.view(bsz, k)
            if use_stats:
                ops.beam_topk_step_tiles(logits, stats, bsz, k, lps[cur], lens[cur], fin[cur], penalty,
                                         END_TOKEN_INDEX, scores, word, beam, lps[nxt], lens[nxt], fin[nxt], src, ws,
                                         rmax, rlse, allfin[s:s + 1])
            else:
                ops.beam_topk_step_fused(logits, bsz, k, lps[cur], lens[cur], fin[cur], penalty, END_TOKEN_INDEX,
                                         scores, word, beam, lps[nxt], lens[nxt], fin[nxt], src, ws, rmax, rlse,
                                         allfin[s:s + 1])     # log-softmax statistics (:537-543) fused in
            if fast:
                ops.gather_rows(stepper.hbuf[cur], srcf, stepper.sel)            # :503-532
            else:
                if indexed:                      # s bodies and the initial step so far: position s + 1, cache copy s & 1
                    stepper.set_position(s + 1, cur)
                stepper.reorder(srcf)
            if not tabled:
                dec.embed_input_symbols(ctx, wordf, out=emb)                     # :507-510 (:546-551: see below)
            if fast:
                stepper.step(emb, att_at(s + 1), out_state, logits, h_prev=stepper.sel,
                             h_out=stepper.hbuf[nxt], stats=stats, **({"ids": wordf} if tabled else {}))   # :534-535
            elif indexed:
                stepper.step(emb, att_at(s + 1), out_state, logits, finished=fin[nxt].view(rows))
            else:
                loop["att"] = stepper.step(emb, loop["att"], out_state, logits, finished=fin[nxt].view(rows))

        shape_key = tuple(tuple(a.weights.shape) for a in att0)
        # ---- loop (:330-355 criterion, :394-556 body): the criterion is evaluated on the device, the host reads the
        # flags one chunk behind what it has enqueued; bodies past the first all-finished one leave the search state
        # unchanged and only append <pad> rows, which are cropped below
        graphed = fast or indexed
        chunk_key = key + ("chunk", k, v, shape_key, getattr(stepper, "shape_key", ()))

        def launch(s0, n):
            def chunk():
                for s in range(s0, s0 + n):
                    body(s)
            if graphed:
                ctx.session.graphed(chunk_key + (s0, n), chunk)
            else:
                chunk()
        steps, executed = ctx.session.decode_chunks(max_steps, CHECK_EVERY_BEAM, launch, allfin, run_ahead=graphed)
        cur = executed & 1                                # buffers the last executed body wrote
        # token histories (:546-551) from the back-pointers, once: out[t+1, r] = word_t[ancestor_t(r)]
        ops.beam_backtrace(src_hist, word_hist, first_sym, tok, executed)
        ctx.memo[(id(self), "_raw_selections")] = (src_hist, word_hist, steps, bsz)
        token_ids = tok[:steps + 1].view(steps + 1, bsz, k)
        prev_logprobs = None
        search_state = SearchState(lps[cur], prev_logprobs, lens[cur], fin[cur])
        results = SearchResults(scores, token_ids)
        feedables = DecoderFeedables(step=steps + 1, finished=fin[cur].view(rows), embedded_input=emb,
                                     other=None)
        dec_ls = LoopState(histories=None, constants=None, feedables=feedables)
        return BeamSearchOutput(results, dec_ls, search_state, [])
