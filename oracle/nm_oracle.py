"""CPU restatement (NumPy) of Neural Monkey's attention-decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may use it, and only as the checker.

PARITY PINNED TO THE REFERENCE'S OWN CODE (round 4).  The reference's arithmetic lives partly in its own Python and
partly in the un-vendored dependency ``tensorflow>=1.12.0,<1.13`` (requirements.txt:13), which cannot be installed
here; it holds no golden vectors for this path.  So ``tests/golden/make_reference_exec_golden.py`` imports
``/root/reference/neuralmonkey`` UNMODIFIED and executes it on a NumPy-eager stand-in for the TensorFlow calls it makes
(``tests/ref_exec/tf_eager.py``, test side only), and the resulting fixtures (``tests/golden/ref_exec/*.npz``: functions,
the beam-search body, whole RNN / Transformer models, finite differences of the reference's loss for the gradients,
six of its acceptance INI files end to end) are what this oracle has to reproduce -- ids exactly, float32 within 2e-6
of a tensor's maximum (``tests/test_reference_exec.py``; ``tests/test_reference_exec_regen.py`` regenerates them
bit for bit where the reference tree exists).  What remains a restatement from knowledge of TF 1.12 is the inside of
the TensorFlow ops themselves (GRUCell / LSTMCell, dynamic_rnn, softmax, top_k's tie order, dense, sequence_loss:
SURVEY section 9) -- stated once, in the stand-in, and shared by every fixture.  Every function below cites the
reference call site (file:line relative to /root/reference) it follows.

All functions take a ``dt`` (np.float32 default, np.float64 for the noise-floor
run) through the dtype of their inputs: no function casts up silently.
"""
from typing import Dict, List, NamedTuple, Optional, Tuple

import numpy as np

PAD, START, END, UNK = 0, 1, 2, 3          # neuralmonkey/vocabulary.py:20-31
INF = 1e9                                  # decoders/beam_search_decoder.py:42


# --------------------------------------------------------------------------- #
# elementary ops (TF-1.12 semantics)
# --------------------------------------------------------------------------- #
def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def softmax(x):
    """tf.nn.softmax, last axis: exp(x-max)/sum(exp(x-max))."""
    z = np.exp(x - x.max(axis=-1, keepdims=True))
    return (z / z.sum(axis=-1, keepdims=True)).astype(x.dtype)


def log_softmax(x):
    """tf.nn.log_softmax, last axis: x - max - log(sum(exp(x-max)))."""
    sh = x - x.max(axis=-1, keepdims=True)
    return (sh - np.log(np.exp(sh).sum(axis=-1, keepdims=True))).astype(x.dtype)


def dense(x, w, b=None):
    """tf.layers.dense / tf.matmul: x.W (+ b)."""
    y = x @ w
    if b is not None:
        y = y + b
    return y.astype(x.dtype)


def layer_norm(x, gamma, beta, eps=1e-6):
    """tf_utils.py:189-219: biased variance, eps inside rsqrt."""
    mean = x.mean(axis=-1, keepdims=True)
    var = np.square(x - mean).mean(axis=-1, keepdims=True)
    eps = np.asarray(eps, dtype=x.dtype)
    return ((x - mean) * (1.0 / np.sqrt(var + eps)) * gamma + beta).astype(x.dtype)


def top_k(x, k):
    """tf.nn.top_k on the last axis: values descending, equal values -> lower
    index first (decoders/beam_search_decoder.py:475)."""
    n = x.shape[-1]
    if n > 4096 and k * 8 < n:
        # a wide row (beam scores over k*V candidates): everything that ties with or beats the k-th largest value
        # is kept by a threshold, then ordered like the plain path -- identical result, no full sort of the row
        kth = np.partition(x, n - k, axis=-1)[..., n - k:n - k + 1]
        keep = x >= kth
        width = int(keep.sum(-1).max())
        cols = np.argsort(~keep, axis=-1, kind="stable")[..., :width]          # kept columns first, ascending index
        vals = np.where(np.take_along_axis(keep, cols, -1), np.take_along_axis(x, cols, -1), -np.inf)
        order = np.argsort(-vals, axis=-1, kind="stable")[..., :k]
        idx = np.take_along_axis(cols, order, -1)
        return np.take_along_axis(x, idx, axis=-1), idx.astype(np.int32)
    # stable argsort of -x keeps the lower index first among equal values
    idx = np.argsort(-x, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(x, idx, axis=-1), idx.astype(np.int32)


# --------------------------------------------------------------------------- #
# a1: padding / masks                              vocabulary.py:331-358
# --------------------------------------------------------------------------- #
def pad_ids(sentences: List[List[int]], max_length: Optional[int] = None,
            add_end_symbol: bool = False) -> np.ndarray:
    """pad_batch on already-indexed sentences (vocabulary.py:331-354)."""
    max_len = max(len(s) for s in sentences)
    if add_end_symbol:
        max_len += 1
    if max_length is not None:
        max_len = min(max_length, max_len)
    out = np.zeros((len(sentences), max_len), dtype=np.int32)
    for i, sent in enumerate(sentences):
        row = list(sent) + ([END] if add_end_symbol else [])
        row = (row + [PAD] * max_len)[:max_len]
        out[i] = row
    return out


def sentence_mask(ids, dt=np.float32):
    """vocabulary.py:357-358."""
    return (ids != PAD).astype(dt)


# --------------------------------------------------------------------------- #
# a2: embedded sequence                            model/sequence.py:170-194
# --------------------------------------------------------------------------- #
def embedded_sequence(emb, ids, scale_by_depth=False):
    """Gather rows, optional *sqrt(E), multiply by the mask."""
    mask = sentence_mask(ids, emb.dtype)
    out = emb[ids]
    if scale_by_depth:
        out = out * np.asarray(emb.shape[-1] ** 0.5, dtype=emb.dtype)
    return (out * mask[..., None]).astype(emb.dtype), mask


# --------------------------------------------------------------------------- #
# a5: RNN cells                                    nn/ortho_gru_cell.py:44-105
# --------------------------------------------------------------------------- #
def gru_cell(x, h, p):
    """TF GRUCell as used by OrthoGRUCell (nn/ortho_gru_cell.py:44-53).

    p: gates_kernel [(D+H),2H], gates_bias [2H] (init 1.0),
       cand_kernel [(D+H),H], cand_bias [H].  Gate order [r,u], kernel rows
       [inputs; state].
    """
    hsz = h.shape[-1]
    g = sigmoid(dense(np.concatenate([x, h], -1), p["gates_kernel"], p["gates_bias"]))
    r, u = g[..., :hsz], g[..., hsz:]
    c = np.tanh(dense(np.concatenate([x, r * h], -1), p["cand_kernel"], p["cand_bias"]))
    return (u * h + (1 - u) * c).astype(x.dtype)


def nematus_gru_cell(x, h, p):
    """nn/ortho_gru_cell.py:73-105 (NematusGRUCell.call)."""
    hsz = h.shape[-1]
    gi = dense(x, p["gates_input_kernel"], p.get("gates_input_bias"))
    gs = dense(h, p["gates_state_kernel"], p.get("gates_state_bias"))
    g = sigmoid(gs + gi)
    r, u = g[..., :hsz], g[..., hsz:]
    ci = dense(x, p["cand_input_kernel"], p.get("cand_input_bias"))
    cs = dense(h, p["cand_state_kernel"], p.get("cand_state_bias"))
    c = np.tanh(cs * r + ci)
    return (u * h + (1 - u) * c).astype(x.dtype)


def lstm_cell(x, c, h, p):
    """tf.contrib.rnn.LSTMCell (decoders/decoder.py:29): gate order i,j,f,o,
    forget_bias 1.0, state (c,h)."""
    z = dense(np.concatenate([x, h], -1), p["kernel"], p["bias"])
    i, j, f, o = np.split(z, 4, axis=-1)
    c2 = sigmoid(f + 1.0) * c + sigmoid(i) * np.tanh(j)
    h2 = sigmoid(o) * np.tanh(c2)
    return c2.astype(x.dtype), h2.astype(x.dtype)


CELLS = {"GRU": gru_cell, "NematusGRU": nematus_gru_cell}


# --------------------------------------------------------------------------- #
# a3: dynamic_rnn / bidirectional_dynamic_rnn      encoders/recurrent.py:71-110
# --------------------------------------------------------------------------- #
def reverse_sequence(x, lengths):
    """tf.reverse_sequence(seq_axis=1): reverse only the first L[b] steps."""
    out = x.copy()
    for b, ln in enumerate(lengths):
        out[b, :ln] = x[b, :ln][::-1]
    return out


def dynamic_rnn(cell, x, lengths, p):
    """tf.nn.dynamic_rnn(sequence_length=L): beyond L[b] the output row is 0
    and the state row is copied through."""
    bsz, steps, _ = x.shape
    hsz = p["cand_bias"].shape[0] if "cand_bias" in p else p["cand_state_kernel"].shape[1]
    h = np.zeros((bsz, hsz), dtype=x.dtype)
    outs = np.zeros((bsz, steps, hsz), dtype=x.dtype)
    for t in range(steps):
        new_h = cell(x[:, t], h, p)
        live = (t < lengths)[:, None]
        h = np.where(live, new_h, h)
        outs[:, t] = np.where(live, new_h, 0)
    return outs, h


def bidirectional_rnn(cell, x, lengths, p_fw, p_bw):
    """tf.nn.bidirectional_dynamic_rnn (encoders/recurrent.py:86-98)."""
    out_fw, fin_fw = dynamic_rnn(cell, x, lengths, p_fw)
    out_bw_r, fin_bw = dynamic_rnn(cell, reverse_sequence(x, lengths), lengths, p_bw)
    out_bw = reverse_sequence(out_bw_r, lengths)
    return (np.concatenate([out_fw, out_bw], 2),
            np.concatenate([fin_fw, fin_bw], 1))


# --------------------------------------------------------------------------- #
# a4: SentenceEncoder                              encoders/recurrent.py:179-314
# --------------------------------------------------------------------------- #
class EncoderOutput(NamedTuple):
    temporal_states: np.ndarray      # [B,S,C]
    temporal_mask: np.ndarray        # [B,S]
    output: np.ndarray               # [B,C]
    rnn_input: np.ndarray            # [B,S,E]


def sentence_encoder(params: Dict[str, np.ndarray], src_ids, name="encoder",
                     rnn_cell="GRU", include_final_layer_norm=True):
    """SentenceEncoder with one bidirectional layer, dropout off
    (encoders/recurrent.py:179-217, 236-314)."""
    emb = params[f"{name}_input/embedding_matrix_0"]
    x, mask = embedded_sequence(emb, src_ids)
    lengths = mask.sum(1).astype(np.int32)          # model/stateful.py:62-69
    cell = CELLS[rnn_cell]
    p_fw = _cell_params(params, f"{name}/rnn_0_bidirectional/bidirectional_rnn/fw", rnn_cell)
    p_bw = _cell_params(params, f"{name}/rnn_0_bidirectional/bidirectional_rnn/bw", rnn_cell)
    states, final = bidirectional_rnn(cell, x, lengths, p_fw, p_bw)
    if include_final_layer_norm:
        g, b = params[f"{name}/LayerNorm/gamma"], params[f"{name}/LayerNorm/beta"]
        states, final = layer_norm(states, g, b), layer_norm(final, g, b)
    return EncoderOutput(states, mask, final, x)


def _cell_params(params, prefix, rnn_cell):
    if rnn_cell == "GRU":
        return {"gates_kernel": params[f"{prefix}/OrthoGRUCell/gates/kernel"],
                "gates_bias": params[f"{prefix}/OrthoGRUCell/gates/bias"],
                "cand_kernel": params[f"{prefix}/OrthoGRUCell/candidate/kernel"],
                "cand_bias": params[f"{prefix}/OrthoGRUCell/candidate/bias"]}
    out = {}
    for k in ("gates_input_kernel", "gates_input_bias", "gates_state_kernel",
              "gates_state_bias", "cand_input_kernel", "cand_input_bias",
              "cand_state_kernel", "cand_state_bias"):
        key = f"{prefix}/{k}"
        if key in params:
            out[k] = params[key]
    return out


# --------------------------------------------------------------------------- #
# a6/a7: Bahdanau attention                        attention/feed_forward.py
# --------------------------------------------------------------------------- #
def attention_keys(states, wk):
    """hidden_features = 1x1 conv == states.Wk, no bias (feed_forward.py:105-118)."""
    return dense(states, wk)


def attention_step(query, hidden_features, states, mask, ap):
    """Attention.attention (feed_forward.py:125-166).

    ap: query_w [Q,A], query_b [A], v [A], bias [] .  ``hidden_features`` and
    ``states`` may have batch 1 and broadcast against ``query`` rows (the
    reference's batch-1 beam search relies on exactly this broadcast).
    Returns (context [R,C], weights [R,S]).
    """
    y = dense(query, ap["query_w"], ap["query_b"])                   # :130-132
    e = (ap["v"] * np.tanh(hidden_features + y[:, None, :])).sum(-1) + ap["bias"]  # :120-123
    e = e.astype(query.dtype)
    if mask is None:
        w = softmax(e)
    else:
        w_all = softmax(e) * mask                                    # :139-141
        norm = w_all.sum(1, keepdims=True) + np.asarray(1e-8, dtype=e.dtype)
        w = w_all / norm
    ctx = (w[:, :, None] * states).sum(1)                            # :151-154
    return ctx.astype(query.dtype), w.astype(query.dtype)


# --------------------------------------------------------------------------- #
# a8-a13: RNN decoder                              decoders/decoder.py, autoregressive.py
# --------------------------------------------------------------------------- #
class DecoderSpec(NamedTuple):
    name: str = "decoder"
    att_name: str = "attention"
    rnn_cell: str = "GRU"
    supress_unk: bool = False
    max_output_len: int = 50


def _dec_params(params, spec):
    n, a = spec.name, spec.att_name
    scope = f"{n}/attention_decoder"
    return {
        "emb": params[f"{n}/word_embeddings"],
        "init_w": params[f"{n}/initial_state/encoders_projection/kernel"],
        "init_b": params[f"{n}/initial_state/encoders_projection/bias"],
        "cell": _cell_params(params, scope, spec.rnn_cell),
        "att": {"query_w": params[f"{a}/Attention/attn_query_projection"],
                "query_b": params[f"{a}/attn_projection_bias"],
                "v": params[f"{a}/attn_similarity_v"],
                "bias": params[f"{a}/attn_bias"]},
        "key_w": params[f"{a}/attn_key_projection"],
        "out_w": params[f"{scope}/dense/kernel"],
        "out_b": params[f"{scope}/dense/bias"],
        "logit_w": params[f"{n}/state_to_word_W"],
        "logit_b": params[f"{n}/state_to_word_b"],
    }


def decoder_initial_state(enc_output, dp):
    """decoder.py:226-252 + encoder_projection.py:47-73 (dropout off)."""
    return dense(enc_output, dp["init_w"], dp["init_b"])


def decoder_step(dp, spec, emb_in, prev_out, hf, states, mask):
    """Decoder.next_state for GRU cells (decoders/decoder.py:279-358).

    Returns (output [R,E], cell_output [R,H], ctx [R,C], weights [R,S]).
    """
    cell = CELLS[spec.rnn_cell]
    cell_output = cell(emb_in, prev_out, dp["cell"])                  # :288-289
    ctx, w = attention_step(cell_output, hf, states, mask, dp["att"])  # :291-297
    out = np.tanh(dense(np.concatenate([cell_output, emb_in, ctx], 1),
                        dp["out_w"], dp["out_b"]))                    # output_projection.py:115-130
    return out.astype(emb_in.dtype), cell_output, ctx, w


def state_to_logits(dp, spec, out):
    """autoregressive.py:450-459."""
    logits = dense(out, dp["logit_w"], dp["logit_b"])
    if spec.supress_unk:
        unk = np.zeros(logits.shape[-1], dtype=logits.dtype)
        unk[UNK] = -1e9
        logits = logits + unk
    return logits.astype(out.dtype)


class DecodeResult(NamedTuple):
    logits: np.ndarray          # [T,R,V]
    output_states: np.ndarray   # [T,R,E]
    symbols: np.ndarray         # [T,R]
    mask: np.ndarray            # [T,R] bool (= not finished, after update)
    rnn_outputs: np.ndarray     # [T,R,H]
    contexts: np.ndarray        # [T,R,C]
    weights: np.ndarray         # [T,R,S]


def gumbel_noise(rows: int, vocab: int, salt: int) -> np.ndarray:
    """The noise of one sampling step of the engine (csrc/nm_logits.hip:gumbel_argmax_kernel), restated: TF's
    tf.multinomial (autoregressive.py:470-473) draws from a Philox stream nobody can replay, the engine draws
    argmax(logits + g) with g = -log(-log(u)) and u a counter-based hash of (salt, row, column).  float32 [rows, vocab]."""
    def mix(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x21F0AAAD)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x735A2D97)
        x ^= x >> np.uint32(15)
        return x
    with np.errstate(over="ignore"):
        key = mix(np.uint32(salt & 0xFFFFFFFF) + np.arange(rows, dtype=np.uint32) * np.uint32(0x85EBCA6B))
        bits = mix(np.arange(vocab, dtype=np.uint32)[None, :] * np.uint32(0x9E3779B1) + key[:, None])
    # 23 bits + 0.5 is exact in float32: u in [2^-24, 1 - 2^-24], the noise is finite for every bit pattern
    u = ((bits >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    return (-np.log(-np.log(u))).astype(np.float32)


def decoding_loop(params, spec: DecoderSpec, enc: EncoderOutput,
                  train_inputs: Optional[np.ndarray], train_mode: bool, temperature: float = 1.0) -> DecodeResult:
    """AutoregressiveDecoder.decoding_loop (autoregressive.py:425-562) around
    Decoder.next_state.  ``train_inputs`` is time-major [T,B] (:216-219).  ``temperature``: logits /= temperature
    (:493).  (A sampled loop is checked by teacher-forcing the engine's draws through this function and restating
    each draw with ``gumbel_noise``: tests/test_sampling_gpu.py.)"""
    dp = _dec_params(params, spec)
    bsz = enc.output.shape[0]
    hf = attention_keys(enc.temporal_states, dp["key_w"])
    state = decoder_initial_state(enc.output, dp)
    emb_in = dp["emb"][np.full(bsz, START)]                           # :201-203, 377-382
    finished = np.zeros(bsz, dtype=bool)
    step = 0
    hist = {k: [] for k in DecodeResult._fields}
    while (not finished.all()) and step < spec.max_output_len:        # :425-437
        out, state, ctx, w = decoder_step(dp, spec, emb_in, state, hf,
                                          enc.temporal_states, enc.temporal_mask)
        logits = state_to_logits(dp, spec, out)
        if temperature != 1.0:
            logits = (logits / np.asarray(temperature, logits.dtype)).astype(logits.dtype)     # :493
        if train_mode:
            nxt = train_inputs[step].astype(np.int64)                 # :467-468
        else:
            nxt = logits.argmax(1)                                    # :470 (pad included)
        nxt = nxt * (~finished)                                       # :472-478
        finished = finished | (nxt == END)                            # :443-445
        emb_in = dp["emb"][nxt]
        for k, v in zip(DecodeResult._fields,
                        (logits, out, nxt, ~finished, state, ctx, w)):
            hist[k].append(v)
        step += 1
    return DecodeResult(*[np.stack(hist[k]) for k in DecodeResult._fields])


def sequence_xent(logits_tbv, targets_tb, mask_tb):
    """contrib.seq2seq.sequence_loss(avg_*=False): -log_softmax[target]*w,
    here kept time-major (autoregressive.py:293-316)."""
    lp = log_softmax(logits_tbv)
    t, b = targets_tb.shape
    picked = lp[np.arange(t)[:, None], np.arange(b)[None, :], targets_tb]
    return (-picked * mask_tb).astype(logits_tbv.dtype)


def train_loss(res: DecodeResult, train_inputs):
    """autoregressive.py:292-316: sum(xent)/sum(mask)."""
    mask = sentence_mask(train_inputs, res.logits.dtype)
    x = sequence_xent(res.logits, train_inputs, mask)
    return (x.sum() / mask.sum()).astype(res.logits.dtype)


def runtime_loss(res: DecodeResult, train_inputs):
    """autoregressive.py:351-371: crop to min time; divide by sum(runtime_mask)."""
    mt = min(train_inputs.shape[0], res.logits.shape[0])
    mask = sentence_mask(train_inputs, res.logits.dtype)
    x = sequence_xent(res.logits[:mt], train_inputs[:mt], mask[:mt])
    return (x.sum() / res.mask.astype(res.logits.dtype).sum()).astype(res.logits.dtype)


def greedy_tokens(res: DecodeResult) -> List[List[int]]:
    """GreedyRunner.collect_results single session (runners/runner.py:35-63):
    argmax of runtime_logprobs == stored argmax of logits; cut at </s>
    (vocabulary.py:257-288).  Returned as id lists (without </s>)."""
    amax = log_softmax(res.logits).argmax(-1)       # [T,B]
    sents = [[] for _ in range(amax.shape[1])]
    for vec in amax:
        for s, wid in zip(sents, vec):
            if not s or s[-1] != END:
                s.append(int(wid))
    return [s[:-1] if s and s[-1] == END else s for s in sents]


# --------------------------------------------------------------------------- #
# a14-a17: beam search                 decoders/beam_search_decoder.py:218-596
# --------------------------------------------------------------------------- #
class BeamResult(NamedTuple):
    scores: np.ndarray       # [B,k]  last top-k scores
    token_ids: np.ndarray    # [steps+1,B,k]
    logprob_sum: np.ndarray  # [B,k]
    lengths: np.ndarray      # [B,k]
    finished: np.ndarray     # [B,k]
    min_gap: float           # smallest (k-th)-( k+1-th) score gap seen (tie report)
    beam_ids: np.ndarray     # [steps,B,k] parent beam of each selection
    word_ids: np.ndarray     # [steps,B,k]
    gaps: Optional[np.ndarray] = None   # [steps,B] smallest NON-ZERO relative gap between adjacent scores of
                                        # the top k+1 (exact ties are ordered by index and are not near-ties)
    tie_sets: Optional[dict] = None     # sentence -> (step, flat candidate ids, scores) of the best k + 16 candidates
                                        # at the sentence's FIRST near-tie step (beam_search(tie_margin=...))
    follow: Optional[dict] = None       # beam_search(follow=...): per step what the oracle would have picked and what
                                        # it thinks of the picks it was made to follow (see beam_search_core)


def length_penalty(lengths, alpha, dt):
    """_length_penalty (:561-573): ((5+len)/6)**alpha in the working dtype."""
    return (((np.asarray(5.0, dt) + lengths.astype(dt)) / np.asarray(6.0, dt))
            ** np.asarray(alpha, dt)).astype(dt)


def beam_search_core(first_logits, step_fn, bsz: int, beam_size: int, max_steps: int, length_normalization: float,
                     tie_margin: Optional[float] = None, follow=None) -> BeamResult:
    """BeamSearchDecoder around ANY parent decoder (beam_search_decoder.py:218-556).

    ``first_logits`` [B*k, V]: the parent step that ``get_initial_loop_state`` runs on the tiled rows (:255-300).
    ``step_fn(src_rows [B*k], words [B*k]) -> logits [B*k, V]``: reorder the parent's per-row state by ``src_rows``
    (gather_flat, :503-532), feed ``words`` and run one parent body (:534-535).

    ``follow = (beam_ids [steps,B,k], word_ids [steps,B,k])`` -- the selections of ANOTHER implementation of the
    search: at every step the oracle scores the candidates from its own state, records its own top k AND its scores
    of the selections it was given, then advances along the GIVEN selections.  A checker can thus account for every
    selection of a whole search even where two fp32 implementations legitimately order near-tied candidates
    differently (the searches would otherwise diverge at the first near-tie and nothing later could be compared).
    ``BeamResult.follow``: ``own_idx`` / ``own_scores`` [steps,B,k+1] (the oracle's best k+1 flat candidates),
    ``given_scores`` [steps,B,k] (its scores of the given picks), ``best_other`` [steps,B] (its best candidate NOT
    among the given picks).
    """
    k = beam_size
    dt = first_logits.dtype
    vsz = first_logits.shape[1]
    logits = first_logits
    first_sym = logits.argmax(1)                     # parent greedy symbol, stored as token_ids[0]
    logprob_sum = np.tile(np.array([0.0] + [-INF] * (k - 1), dtype=dt), (bsz, 1))
    prev_logprobs = log_softmax(logits).reshape(bsz, k, vsz)
    lengths = np.zeros((bsz, k), dtype=np.int32)
    finished = np.zeros((bsz, k), dtype=bool)
    token_ids = first_sym.reshape(1, bsz, k).astype(np.int64)
    scores = np.zeros((bsz, k), dtype=dt)
    dec_step = 1
    min_gap = np.inf
    beam_hist, word_hist, gap_hist = [], [], []
    tie_sets = {}
    rep = {"own_idx": [], "own_scores": [], "given_scores": [], "best_other": []} if follow is not None else None

    finished_row = np.full(vsz, -INF, dtype=dt)
    finished_row[PAD] = 0.0
    bidx = np.arange(bsz)[:, None]

    # --- loop (:330-355 criterion, :394-556 body)
    while (dec_step - 1) < max_steps and not finished.all():
        fmask = finished.astype(dt)[:, :, None]
        logprobs = (1.0 - fmask) * prev_logprobs + fmask * finished_row      # :440-456
        hyp = logprob_sum[:, :, None] + logprobs                              # :460
        hyp_len = lengths + 1 - finished.astype(np.int32)                     # :463-464
        sc = hyp / length_penalty(hyp_len, length_normalization, dt)[:, :, None]  # :467-468
        flat = sc.reshape(bsz, k * vsz).astype(dt)
        top_sc, top_idx = top_k(flat, min(k + 1, flat.shape[1]))
        adj = top_sc[:, :-1] - top_sc[:, 1:]
        adj_rel = np.where(adj > 0, adj / np.maximum(np.abs(top_sc[:, :-1]), 1e-30), np.inf)
        gap_hist.append(np.where(finished.all(axis=1), np.inf, adj_rel.min(axis=1)))
        if tie_margin is not None:               # candidates around a sentence's first near-tie, for the checker
            for b in np.nonzero(gap_hist[-1] <= tie_margin)[0]:
                if int(b) not in tie_sets:
                    ws_sc, ws_idx = top_k(flat[b:b + 1], min(k + 16, flat.shape[1]))
                    tie_sets[int(b)] = (dec_step - 1, ws_idx[0].copy(), ws_sc[0].copy())
        if top_sc.shape[1] > k:
            live = ~finished.all(axis=1)
            if live.any():
                gap = (top_sc[live, k - 1] - top_sc[live, k])
                rel = gap / np.maximum(np.abs(top_sc[live, k - 1]), 1e-30)
                min_gap = min(min_gap, float(rel.min()))
        if follow is not None:
            step_i = dec_step - 1
            if step_i >= len(follow[0]):
                break
            given = (follow[0][step_i].astype(np.int64) * vsz + follow[1][step_i].astype(np.int64))   # [B,k]
            own_sc, own_idx = top_k(flat, min(2 * k + 1, flat.shape[1]))
            rep["own_idx"].append(own_idx[:, :k + 1].copy())
            rep["own_scores"].append(own_sc[:, :k + 1].copy())
            rep["given_scores"].append(flat[bidx, given])
            other = np.where((own_idx[:, :, None] == given[:, None, :]).any(-1), -np.inf, own_sc)
            rep["best_other"].append(other.max(axis=1))
            top_idx = given.astype(np.int32)
            top_sc = flat[bidx, given]
        top_sc, top_idx = top_sc[:, :k], top_idx[:, :k]
        word = (top_idx % vsz).astype(np.int64)                               # :481-483
        beam = top_idx // vsz
        lengths = hyp_len[bidx, beam]                                         # :492
        logprob_sum = hyp.reshape(bsz, k * vsz)[bidx, top_idx]                # :493-496 (unnormalised)
        finished = finished[bidx, beam] | (word == END)                       # :499-501
        flat_src = (bidx * k + beam).reshape(-1)
        logits = step_fn(flat_src, word.reshape(-1))                          # :503-535
        prev_logprobs = log_softmax(logits).reshape(bsz, k, vsz)              # :537-543
        token_ids = np.concatenate(
            [token_ids[:, bidx, beam], word[None]], axis=0)                   # :546-551
        scores = top_sc
        beam_hist.append(beam.copy())
        word_hist.append(word.copy())
        dec_step += 1
    z = np.zeros((0, bsz, k), dtype=np.int64)
    return BeamResult(scores, token_ids, logprob_sum, lengths, finished, min_gap,
                      np.stack(beam_hist) if beam_hist else z,
                      np.stack(word_hist) if word_hist else z,
                      np.stack(gap_hist) if gap_hist else np.zeros((0, bsz)), tie_sets,
                      None if rep is None else {n: np.stack(v) for n, v in rep.items()})


def beam_search(params, spec: DecoderSpec, enc: EncoderOutput, beam_size: int,
                max_steps: int, length_normalization: float, tie_margin: Optional[float] = None,
                follow=None) -> BeamResult:
    """BeamSearchDecoder over the RNN Decoder.

    The reference tiles the parent loop state to B*k rows (expand_to_beam
    :575-596, row order b*k+j) but not the Bahdanau tensors, so it only runs
    at batch 1 through broadcasting.  Tiling the keys is the same
    arithmetic as that broadcast; we tile them here so any batch works and
    batch 1 is bit-identical to the broadcast.
    """
    dp = _dec_params(params, spec)
    bsz, k = enc.output.shape[0], beam_size
    rows = bsz * k
    rep = lambda a: np.repeat(a, k, axis=0)
    states, mask = rep(enc.temporal_states), rep(enc.temporal_mask)
    hf = attention_keys(states, dp["key_w"])

    # --- get_initial_loop_state (:218-328): one parent step on tiled rows
    carried = {"prev": rep(decoder_initial_state(enc.output, dp))}
    emb_in = dp["emb"][np.full(rows, START)]
    out, carried["prev"], _, _ = decoder_step(dp, spec, emb_in, carried["prev"], hf, states, mask)

    def step_fn(flat_src, words):
        prev = carried["prev"][flat_src]                                      # :503-532 gather_flat
        emb = dp["emb"][words]                                                # :507-510
        o, carried["prev"], _, _ = decoder_step(dp, spec, emb, prev, hf, states, mask)   # :534-535
        return state_to_logits(dp, spec, o)

    return beam_search_core(state_to_logits(dp, spec, out), step_fn, bsz, k, max_steps, length_normalization,
                            tie_margin, follow)


def beam_tokens(res: BeamResult, rank: int = 1) -> Tuple[List[List[int]], float]:
    """BeamSearchRunner.prepare_results (runners/beamsearch_runner.py:84-106)."""
    tok = np.transpose(res.token_ids, (1, 2, 0))
    sents = []
    for toks in tok:
        sent = []
        for t in toks[rank - 1][1:]:
            if t == END:
                break
            sent.append(int(t))
        sents.append(sent)
    bs_scores = [s[rank - 1] for s in res.scores]
    return sents, float(np.mean(bs_scores) * len(bs_scores))


# --------------------------------------------------------------------------- #
# a22: trainer arithmetic                  trainers/generic_trainer.py:84-195
# --------------------------------------------------------------------------- #
import re
BIAS_REGEX = re.compile(r"[Bb]ias")               # generic_trainer.py:14


def regularizable(names):
    """generic_trainer.py:87-91: every trainable whose name lacks [Bb]ias."""
    return [n for n in names if not BIAS_REGEX.findall(n)]


def l1_l2(params):
    names = regularizable(sorted(params))
    l1 = sum(np.abs(params[n]).sum() for n in names)
    l2 = sum((params[n] ** 2).sum() for n in names)
    return l1, l2


def clip_by_norm(g, c):
    """tf.clip_by_norm: g*c/max(||g||,c)."""
    n = np.sqrt((g.astype(np.float64) ** 2).sum()).astype(g.dtype)
    return (g * (np.asarray(c, g.dtype) / max(n, np.asarray(c, g.dtype)))).astype(g.dtype)


def adam_step(theta, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (trainers/generic_trainer.py:55-57), step t>=1."""
    dt = theta.dtype
    lr_t = np.asarray(lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t), dt)
    m = (b1 * m + (1 - b1) * g).astype(dt)
    v = (b2 * v + (1 - b2) * g * g).astype(dt)
    theta = (theta - lr_t * m / (np.sqrt(v) + np.asarray(eps, dt))).astype(dt)
    return theta, m, v


# --------------------------------------------------------------------------- #
# synthetic model / data
# --------------------------------------------------------------------------- #
def orthogonal(rng, rows, cols, dt):
    """tf.orthogonal_initializer: QR of a normal matrix, sign-fixed."""
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return q[:rows, :cols].astype(dt)


def init_params(seed=1234, vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512,
                dec_rnn=None, dec_emb=None, att_size=None, std=0.05,
                dt=np.float32, enc_name="encoder", dec_name="decoder",
                att_name="attention") -> Dict[str, np.ndarray]:
    """Random-init weights with the variable names of SURVEY section 9.

    Dense/embedding ~ N(0,std); GRU kernels orthogonal on the recurrent block;
    GRU gate bias 1.0; other biases 0; LayerNorm gamma 1, beta 0.
    """
    rng = np.random.default_rng(seed)
    dec_rnn = dec_rnn or rnn
    dec_emb = dec_emb or emb
    ctx = 2 * rnn
    att = att_size or ctx
    nrm = lambda *s: (rng.standard_normal(s) * std).astype(dt)
    p = {}
    p[f"{enc_name}_input/embedding_matrix_0"] = nrm(vocab_src, emb)
    for d in ("fw", "bw"):
        pre = f"{enc_name}/rnn_0_bidirectional/bidirectional_rnn/{d}/OrthoGRUCell"
        p[f"{pre}/gates/kernel"] = np.concatenate(
            [nrm(emb, 2 * rnn), np.concatenate([orthogonal(rng, rnn, rnn, dt)
                                                for _ in range(2)], 1)], 0)
        p[f"{pre}/gates/bias"] = np.ones(2 * rnn, dt)
        p[f"{pre}/candidate/kernel"] = np.concatenate(
            [nrm(emb, rnn), orthogonal(rng, rnn, rnn, dt)], 0)
        p[f"{pre}/candidate/bias"] = np.zeros(rnn, dt)
    p[f"{enc_name}/LayerNorm/gamma"] = np.ones(ctx, dt)
    p[f"{enc_name}/LayerNorm/beta"] = np.zeros(ctx, dt)
    p[f"{att_name}/attn_key_projection"] = nrm(ctx, att)
    p[f"{att_name}/Attention/attn_query_projection"] = nrm(dec_rnn, att)
    p[f"{att_name}/attn_projection_bias"] = np.zeros(att, dt)
    p[f"{att_name}/attn_similarity_v"] = nrm(att)
    p[f"{att_name}/attn_bias"] = np.zeros((), dt)
    p[f"{dec_name}/word_embeddings"] = nrm(vocab_tgt, dec_emb)
    p[f"{dec_name}/initial_state/encoders_projection/kernel"] = nrm(ctx, dec_rnn)
    p[f"{dec_name}/initial_state/encoders_projection/bias"] = np.zeros(dec_rnn, dt)
    pre = f"{dec_name}/attention_decoder/OrthoGRUCell"
    p[f"{pre}/gates/kernel"] = np.concatenate(
        [nrm(dec_emb, 2 * dec_rnn),
         np.concatenate([orthogonal(rng, dec_rnn, dec_rnn, dt) for _ in range(2)], 1)], 0)
    p[f"{pre}/gates/bias"] = np.ones(2 * dec_rnn, dt)
    p[f"{pre}/candidate/kernel"] = np.concatenate(
        [nrm(dec_emb, dec_rnn), orthogonal(rng, dec_rnn, dec_rnn, dt)], 0)
    p[f"{pre}/candidate/bias"] = np.zeros(dec_rnn, dt)
    p[f"{dec_name}/attention_decoder/dense/kernel"] = nrm(dec_rnn + dec_emb + ctx, dec_emb)
    p[f"{dec_name}/attention_decoder/dense/bias"] = np.zeros(dec_emb, dt)
    p[f"{dec_name}/state_to_word_W"] = nrm(dec_emb, vocab_tgt)
    p[f"{dec_name}/state_to_word_b"] = np.zeros(vocab_tgt, dt)
    return p


def synthetic_batch(seed=1234, batch=128, src_len=50, tgt_len=50, vocab=32000,
                    ragged=False):
    """ids uniform in [4,V); headline = full lengths;
    ragged = lengths uniform [len/2, len]; </s> at the last target position.
    Returns (src_ids [B,S] int32, tgt_ids time-major [T,B] int32)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(4, vocab, size=(batch, src_len)).astype(np.int32)
    tgt = rng.integers(4, vocab, size=(batch, tgt_len)).astype(np.int32)
    if ragged:
        sl = rng.integers(max(1, src_len // 2), src_len + 1, size=batch)
        tl = rng.integers(max(1, tgt_len // 2), tgt_len + 1, size=batch)
    else:
        sl = np.full(batch, src_len)
        tl = np.full(batch, tgt_len)
    for b in range(batch):
        src[b, sl[b]:] = PAD
        tgt[b, tl[b] - 1] = END
        tgt[b, tl[b]:] = PAD
    return src, np.ascontiguousarray(tgt.T)
