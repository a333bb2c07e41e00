"""CPU restatement (torch, autograd) of the configurable parts of the attention-decoder path:
NematusGRU / LSTM cells, stacked / layer-normed / residual encoders, conditional GRU,
attention on input, dropout, the output-projection variants, greedy and beam decoding.

TEST INFRASTRUCTURE ONLY -- imported by ``tests/`` alone; nothing in the product package may
import it.  PARITY PINNED to the reference's own code, see ``oracle/nm_oracle.py``: the fixtures
``rnn_*``, ``captioning*``, ``fd_gradients_rnn_*``, ``fd_gradients_captioning``, ``ini_bahdanau``, ``ini_small`` of
``tests/golden/ref_exec/`` are the reference's own model code executed on the test-side TensorFlow stand-in, and this
module reproduces them; every function cites the reference call site it follows.

Dropout: TF's Philox stream cannot be replayed, so the engine defines its masks by a
counter-based hash (csrc/nm_eltwise.hip ``nm_dropout``); ``dropout_mask`` below restates that
hash in NumPy uint32 arithmetic so that masked forward passes and gradients can be compared
bit-for-bit in the mask and to fp32 tolerance in the values.

Parameters are the engine's own store (``VariableStore.state_dict()``: TF variable names).
"""
import zlib
from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

PAD, START, END, UNK = 0, 1, 2, 3
INF = 1e9


# ------------------------------------------------------------------------------------------------
# dropout mask (nn/utils.py:6-22 semantics: keep iff floor(keep_prob + u) == 1, scale 1/keep_prob)
# ------------------------------------------------------------------------------------------------
def salt_of(global_step: int, *site) -> int:
    base = zlib.crc32("/".join(str(s) for s in site).encode()) & 0xFFFFFFFF
    return (base + global_step * 0x9E3779B9) & 0xFFFFFFFF


def dropout_mask(n: int, keep_prob: float, salt: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint32) * np.uint32(0x9E3779B1) + np.uint32(salt)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x21F0AAAD)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x735A2D97)
        x ^= x >> np.uint32(15)
    uni = (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    keep = (np.float32(keep_prob) + uni) >= np.float32(1.0)
    return keep.astype(np.float32) * (np.float32(1.0) / np.float32(keep_prob))


class Config(NamedTuple):
    enc_name: str = "encoder"
    dec_name: str = "decoder"
    att_name: str = "attention"
    rnn_layers: Tuple = ((4, "bidirectional", "GRU"),)      # (size, direction, cell)
    add_layer_norm: bool = False
    add_residual: bool = False
    include_final_layer_norm: bool = True
    enc_dropout: float = 1.0
    att_dropout: float = 1.0
    dec_cell: str = "GRU"
    conditional_gru: bool = False
    attention_on_input: bool = False
    dec_dropout: float = 1.0
    output_projection: Tuple = ("nonlinear", "tanh", 1.0)   # kind, activation / sizes, keep_prob
    encoder_projection: str = "linear"                      # linear | concat | empty
    tie_embeddings: bool = False
    supress_unk: bool = False
    rnn_size: int = 4
    spatial: Optional[Tuple] = None                         # (ff_hidden_dim, projection_dim): SpatialFiller encoder
    label_smoothing: float = 0.0


def _act(name):
    return {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, "identity": lambda x: x}[name]


class GeneralModel:
    def __init__(self, params: Dict[str, np.ndarray], cfg: Config, dtype=torch.float32, requires_grad=False):
        self.cfg, self.dtype = cfg, dtype
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad)
                  for k, v in params.items()}
        self.global_step = 0

    # -- helpers ---------------------------------------------------------------------------------
    def dropout(self, x, keep, train, *site):
        if keep == 1.0 or not train:
            return x
        mask = dropout_mask(x.numel(), keep, salt_of(self.global_step, *site))
        return x * torch.from_numpy(mask).to(self.dtype).view(x.shape)

    def layer_norm(self, x, prefix):
        """tf_utils.py:189-219: biased variance, eps 1e-6 inside the rsqrt."""
        g, b = self.p[prefix + "/gamma"], self.p[prefix + "/beta"]
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        return (x - mean) * torch.rsqrt(var + 1e-6) * g + b

    def cell(self, kind, scope, x, state, **kw):
        """One step of GRUCell / NematusGRUCell (nn/ortho_gru_cell.py:44-105) / LSTMCell."""
        p = self.p
        if kind == "GRU":
            pre = scope + "/" + kw.get("cell_scope", "OrthoGRUCell")
            (h,) = state
            hs = h.shape[1]
            g = torch.sigmoid(torch.cat([x, h], 1) @ p[pre + "/gates/kernel"] + p[pre + "/gates/bias"])
            r, u = g[:, :hs], g[:, hs:]
            c = torch.tanh(torch.cat([x, r * h], 1) @ p[pre + "/candidate/kernel"] + p[pre + "/candidate/bias"])
            new = u * h + (1 - u) * c
            return new, (new,)
        if kind == "NematusGRU":
            pre = scope + "/" + kw.get("cell_scope", "nematus_gru_cell")
            (h,) = state
            hs = h.shape[1]

            def proj(block, which, inp):
                y = inp @ p["{}/{}/{}_proj/kernel".format(pre, block, which)]
                bias = p.get("{}/{}/{}_proj/bias".format(pre, block, which))
                return y if bias is None else y + bias
            g = torch.sigmoid(proj("gates", "state", h) + proj("gates", "input", x))
            r, u = g[:, :hs], g[:, hs:]
            c = torch.tanh(proj("candidate", "state", h) * r + proj("candidate", "input", x))
            new = u * h + (1 - u) * c
            return new, (new,)
        if kind == "LSTM":
            pre = scope + "/lstm_cell"
            c_prev, h = state
            z = torch.cat([x, h], 1) @ p[pre + "/kernel"] + p[pre + "/bias"]
            i, j, f, o = torch.chunk(z, 4, dim=1)
            c_new = torch.sigmoid(f + 1.0) * c_prev + torch.sigmoid(i) * torch.tanh(j)
            h_new = torch.sigmoid(o) * torch.tanh(c_new)
            return h_new, (c_new, h_new)
        raise ValueError(kind)

    @staticmethod
    def reverse_sequence(x, lengths):
        out = x.clone()
        for b, ln in enumerate(lengths):
            ln = int(ln)
            if ln > 0:
                out[b, :ln] = x[b, :ln].flip(0)
        return out

    def dynamic_rnn(self, kind, scope, x, lengths):
        """tf.nn.dynamic_rnn(sequence_length): zero output, carried state beyond the length."""
        bsz, slen, _ = x.shape
        hs = self._size
        nstate = 2 if kind == "LSTM" else 1
        state = tuple(torch.zeros(bsz, hs, dtype=self.dtype) for _ in range(nstate))
        outs = []
        lens = torch.as_tensor(np.asarray(lengths))
        for t in range(slen):
            live = (t < lens).view(-1, 1)
            out, new_state = self.cell(kind, scope, x[:, t], state)
            state = tuple(torch.where(live, n, s) for n, s in zip(new_state, state))
            outs.append(torch.where(live, out, torch.zeros_like(out)))
        return torch.stack(outs, 1), state[-1]

    # -- encoder (encoders/recurrent.py:71-110, 179-217) ---------------------------------------
    def encode_spatial(self, maps: np.ndarray, name: Optional[str] = None, spatial: Optional[Tuple] = None):
        """SpatialFiller (encoders/numpy_stateful_filler.py:155-245): optional 1x1 convolutions,
        states flattened to [B, H*W, D] with an all-ones mask (attention/base_attention.py:79-122),
        output = mean over positions."""
        cfg, p = self.cfg, self.p
        x = torch.as_tensor(np.asarray(maps), dtype=self.dtype)
        bsz, h, w, _ = x.shape
        x = x.reshape(bsz, h * w, -1)
        ff_dim, proj_dim = spatial if spatial is not None else cfg.spatial
        name = name or cfg.enc_name
        scopes = []
        if ff_dim:
            scopes.append(("conv2d", True))
        if proj_dim:
            scopes.append(("conv2d_1" if scopes else "conv2d", False))
        for scope, use_relu in scopes:
            kernel = p["{}/{}/kernel".format(name, scope)]         # tf.layers.conv2d: [1, 1, in, out]
            x = x @ kernel.reshape(kernel.shape[-2], kernel.shape[-1]) + p["{}/{}/bias".format(name, scope)]
            if use_relu:
                x = torch.relu(x)
        return x, torch.ones(bsz, h * w, dtype=self.dtype), x.mean(1)

    def encode(self, src_ids: np.ndarray, train: bool):
        cfg, p = self.cfg, self.p
        if cfg.spatial is not None:
            return self.encode_spatial(src_ids)
        name = cfg.enc_name
        # model/sequence.py:170-199: one embedding matrix per factor, concatenated along the feature axis; the
        # mask comes from the first factor.  ``src_ids`` is [B,S] or, for a factored input, [F,B,S].
        factors = src_ids if src_ids.ndim == 3 else src_ids[None]
        ids = torch.as_tensor(factors.astype(np.int64))
        mask = (ids[0] != PAD).to(self.dtype)
        lengths = mask.sum(1).to(torch.int64).numpy()
        x = torch.cat([p["{}_input/embedding_matrix_{}".format(name, f)][ids[f]] for f in range(ids.shape[0])], -1)
        x = x * mask.unsqueeze(-1)
        layer_input = self.dropout(x, cfg.enc_dropout, train, name, "rnn_input")
        layer_final = layer_input[:, -1]
        for i, (size, direction, kind) in enumerate(cfg.rnn_layers):
            self._size = size
            scope = "{}/rnn_{}_{}".format(name, i, direction)
            if cfg.add_layer_norm:
                layer_input = self.layer_norm(layer_input, scope + "/LayerNorm")
            if direction == "bidirectional":
                o_fw, f_fw = self.dynamic_rnn(kind, scope + "/bidirectional_rnn/fw", layer_input, lengths)
                o_bw, f_bw = self.dynamic_rnn(kind, scope + "/bidirectional_rnn/bw",
                                              self.reverse_sequence(layer_input, lengths), lengths)
                out = torch.cat([o_fw, self.reverse_sequence(o_bw, lengths)], 2)
                fin = torch.cat([f_fw, f_bw], 1)
            else:
                inp = self.reverse_sequence(layer_input, lengths) if direction == "backward" else layer_input
                out, fin = self.dynamic_rnn(kind, scope + "/rnn", inp, lengths)
                if direction == "backward":
                    out = self.reverse_sequence(out, lengths)
            out = self.dropout(out, cfg.enc_dropout, train, name, "layer_output", i)
            fin = self.dropout(fin, cfg.enc_dropout, train, name, "layer_final", i)
            if cfg.add_residual and layer_input.shape[-1] == out.shape[-1]:
                layer_input, layer_final = layer_input + out, layer_final + fin
            else:
                layer_input, layer_final = out, fin
        if cfg.include_final_layer_norm:
            layer_input = self.layer_norm(layer_input, name + "/LayerNorm")
            layer_final = self.layer_norm(layer_final, name + "/LayerNorm")
        return layer_input, mask, layer_final

    # -- attention (attention/feed_forward.py:47-166) ---------------------------------------------
    def attention_setup(self, states, train):
        cfg, a = self.cfg, self.cfg.att_name
        st = self.dropout(states, cfg.att_dropout, train, a, "attention_states")
        return st, st @ self.p[a + "/attn_key_projection"]

    def attention(self, query, st, hf, mask):
        p, a = self.p, self.cfg.att_name
        y = query @ p[a + "/Attention/attn_query_projection"] + p[a + "/attn_projection_bias"]
        e = (p[a + "/attn_similarity_v"] * torch.tanh(hf + y.unsqueeze(1))).sum(-1) + p[a + "/attn_bias"]
        w_all = torch.softmax(e, -1) * mask
        w = w_all / (w_all.sum(1, keepdim=True) + 1e-8)
        return (w.unsqueeze(-1) * st).sum(1), w

    # -- decoder step (decoders/decoder.py:279-358) -------------------------------------------------
    def output_projection(self, cell_output, emb_in, contexts, train, t):
        cfg, p = self.cfg, self.p
        pre = cfg.dec_name + "/attention_decoder/"
        kind = cfg.output_projection[0]
        site = (cfg.dec_name, "output_projection", t)
        cat = torch.cat([cell_output, emb_in] + contexts, 1)
        if kind == "nonlinear":                                        # output_projection.py:115-130
            _, act, keep = cfg.output_projection
            return self.dropout(_act(act)(cat @ p[pre + "dense/kernel"] + p[pre + "dense/bias"]), keep, train, *site)
        if kind == "nematus":                                          # :76-112
            _, act, keep = cfg.output_projection
            s = (cell_output @ p[pre + "rnn_state/kernel"] + p[pre + "rnn_state/bias"]
                 + emb_in @ p[pre + "prev_out/kernel"] + p[pre + "prev_out/bias"]
                 + torch.cat(contexts, 1) @ p[pre + "context/kernel"] + p[pre + "context/bias"])
            return self.dropout(_act(act)(s), keep, train, *site)
        if kind == "legacy":                                           # :33-72: no previous output, no dropout
            state_with_ctx = torch.cat([cell_output] + contexts, 1)
            return _act(cfg.output_projection[1])(state_with_ctx @ p[pre + "AttnOutputProjection/kernel"]
                                                  + p[pre + "AttnOutputProjection/bias"])
        if kind == "maxout":                                           # :133-160, nn/projection.py:7-35
            _, size, keep = cfg.output_projection
            z = cat @ p[pre + "MaxoutProjection/MaxoutProjection/kernel"] \
                + p[pre + "MaxoutProjection/MaxoutProjection/bias"]
            return self.dropout(torch.maximum(z[:, :size], z[:, size:]), keep, train, *site)
        if kind == "mlp":                                              # :163-188, nn/projection.py:38-58
            _, sizes, act, keep = cfg.output_projection
            x = cat
            base = salt_of(self.global_step, *site)
            for i, _size in enumerate(sizes):
                scope = pre + "deep_output_mlp/mlp_layer_{}".format(i)
                x = _act(act)(x @ p[scope + "/kernel"] + p[scope + "/bias"])
                if keep != 1.0 and train:
                    m = dropout_mask(x.numel(), keep, (base + 0x632BE5AB * (i + 1)) & 0xFFFFFFFF)
                    x = x * torch.from_numpy(m).to(self.dtype).view(x.shape)
            return x
        raise ValueError(kind)

    def decoder_step(self, emb_in, state, st, hf, mask, train, t):
        """state = [prev_rnn_state, prev_rnn_output, *prev_contexts] (RNNFeedables)."""
        cfg, p = self.cfg, self.p
        d = cfg.dec_name
        scope = d + "/attention_decoder"
        prev_state, prev_out, prev_ctxs = state[0], state[1], list(state[2:])
        self._step_t = t
        if cfg.attention_on_input:                                     # :264-277
            x = torch.cat([emb_in] + prev_ctxs, 1) @ p[scope + "/input_projection/kernel"] \
                + p[scope + "/input_projection/bias"]
            rnn_input = self.dropout(x, cfg.dec_dropout, train, d, "input_projection", t)
        else:
            rnn_input = emb_in
        self._size = cfg.rnn_size
        # a.attention(cell_output, prev_rnn_output, rnn_input, loop_state) (decoder.py:291-297): sentinels read these
        self._step_extra = (prev_out, rnn_input)
        if cfg.dec_cell == "LSTM":                                     # :309-325
            cell_output, (next_state, _) = self.cell("LSTM", scope, rnn_input, (prev_state, prev_out))
            ctx, w = self.attention(cell_output, st, hf, mask)
            contexts = [ctx]
        else:
            cell_output, (next_state,) = self.cell(cfg.dec_cell, scope, rnn_input, (prev_out,))
            ctx, w = self.attention(cell_output, st, hf, mask)
            contexts = [ctx]
            if cfg.conditional_gru:                                    # :303-307
                cell_output, (next_state,) = self.cell(cfg.dec_cell, scope, torch.cat(contexts, 1), (next_state,),
                                                       cell_scope="cond_gru_2_cell")
        contexts = [self.dropout(c, cfg.dec_dropout, train, d, "context", i, t) for i, c in enumerate(contexts)]
        cell_output = self.dropout(cell_output, cfg.dec_dropout, train, d, "cell_output", t)
        output = self.output_projection(cell_output, emb_in, contexts, train, t)
        return output, [next_state, cell_output] + contexts, w

    def logits(self, output):
        cfg, p = self.cfg, self.p
        if cfg.tie_embeddings:
            lg = output @ p[cfg.dec_name + "/word_embeddings"].t()
        else:
            lg = output @ p[cfg.dec_name + "/state_to_word_W"] + p[cfg.dec_name + "/state_to_word_b"]
        if cfg.supress_unk:
            unk = torch.zeros(lg.shape[-1], dtype=self.dtype)
            unk[UNK] = -1e9
            lg = lg + unk
        return lg

    def initial_state(self, final, train, states=None, mask=None):
        cfg, p = self.cfg, self.p
        d = cfg.dec_name
        bsz = final.shape[0]
        if cfg.encoder_projection == "nematus":                        # encoder_projection.py:99-145
            means = (states * mask.unsqueeze(-1)).sum(1) / mask.sum(1, keepdim=True)
            means = self.dropout(means, cfg.dec_dropout, train, d, "encoders_projection")
            s0 = torch.tanh(means @ p[d + "/initial_state/encoders_projection/kernel"]
                            + p[d + "/initial_state/encoders_projection/bias"])
        elif cfg.encoder_projection == "linear":                         # encoder_projection.py:47-73
            s0 = final @ p[d + "/initial_state/encoders_projection/kernel"] \
                + p[d + "/initial_state/encoders_projection/bias"]
            s0 = self.dropout(s0, cfg.dec_dropout, train, d, "encoders_projection")
        elif cfg.encoder_projection == "concat":
            s0 = final
        else:
            s0 = torch.zeros(bsz, cfg.rnn_size, dtype=self.dtype)
        return self.dropout(s0, cfg.dec_dropout, train, d, "initial_state")   # decoder.py:235-240

    # -- training loss (autoregressive.py:289-316) -----------------------------------------------
    def train_loss(self, src_ids, tgt_tb, train=True):
        cfg, p = self.cfg, self.p
        states, mask, final = self.encode(src_ids, train)
        st, hf = self.attention_setup(states, train)
        steps, bsz = tgt_tb.shape
        dec_in = np.concatenate([np.full((1, bsz), START, tgt_tb.dtype), tgt_tb[:-1]], 0)
        emb_all = p[cfg.dec_name + "/word_embeddings"][torch.as_tensor(dec_in.reshape(-1).astype(np.int64))]
        emb_all = self.dropout(emb_all, cfg.dec_dropout, train, cfg.dec_name, "embedded_input")
        emb_all = emb_all.view(steps, bsz, -1)
        s0 = self.initial_state(final, train, states, mask)
        state = [s0, s0, torch.zeros(bsz, self.context_size(st), dtype=self.dtype)]
        outs, weights = [], []
        for t in range(steps):
            out, state, w = self.decoder_step(emb_all[t], state, st, hf, mask, train, t)
            outs.append(out)
            weights.append(w)
        logits = self.logits(torch.cat(outs, 0))                       # [T*B, V]
        tgt = torch.as_tensor(tgt_tb.reshape(-1).astype(np.int64))
        tmask = (tgt != PAD).to(self.dtype)
        lp = torch.log_softmax(logits, -1)
        if cfg.label_smoothing:
            # autoregressive.py:292-310: tf.losses.softmax_cross_entropy(onehot, logits, label_smoothing)
            # reduces to ONE scalar (mean over all B*T positions, weights 1.0); sequence_loss broadcasts it
            # against the mask, so sum(xents)/sum(mask) is that mean itself
            eps, vsz = cfg.label_smoothing, logits.shape[-1]
            q = torch.full_like(lp, eps / vsz)
            q[torch.arange(tgt.numel()), tgt] += 1.0 - eps
            mean_all = -(q * lp).sum(-1).mean()
            xents = mean_all * tmask
            return xents.sum() / tmask.sum(), logits.view(steps, bsz, -1), torch.stack(weights)
        xent = -lp[torch.arange(tgt.numel()), tgt] * tmask
        return xent.sum() / tmask.sum(), logits.view(steps, bsz, -1), torch.stack(weights)

    def train_grads(self, src_ids, tgt_tb, train=True):
        loss, _, _ = self.train_loss(src_ids, tgt_tb, train)
        names = list(self.p)
        grads = torch.autograd.grad(loss, [self.p[n] for n in names], allow_unused=True)
        return float(loss.detach()), {n: (None if g is None else g.detach().numpy()) for n, g in zip(names, grads)}

    # -- greedy decoding (autoregressive.py:425-562) -----------------------------------------------
    def _decode_setup(self, src_ids, rep: int = 1):
        states, mask, final = self.encode(src_ids, False)
        st, hf = self.attention_setup(states, False)
        s0 = self.initial_state(final, False, states, mask)
        if rep > 1:
            st, hf, mask = self.repeat_sources(st, hf, mask, rep)
            s0 = s0.repeat_interleave(rep, 0)
        rows = s0.shape[0]
        state = [s0, s0, torch.zeros(rows, self.context_size(st), dtype=self.dtype)]
        return st, hf, mask, state

    # hooks for models whose attention reads several sources (oracle/multisource_ref.py)
    def context_size(self, st) -> int:
        return st.shape[-1]

    def reorder_attention(self, src) -> None:
        """Hook for attentions that carry per-hypothesis state through a beam search (oracle/coverage_ref.py)."""

    def repeat_sources(self, st, hf, mask, rep: int):
        """expand_to_beam (beam_search_decoder.py:575-596): row order b*k + j."""
        return tuple(x.repeat_interleave(rep, 0) for x in (st, hf, mask))

    def greedy(self, src_ids, max_len: int):
        with torch.no_grad():
            p, d = self.p, self.cfg.dec_name
            st, hf, mask, state = self._decode_setup(src_ids)
            rows = state[0].shape[0]
            emb = p[d + "/word_embeddings"][torch.full((rows,), START)]
            finished = torch.zeros(rows, dtype=torch.bool)
            syms, masks, logit_hist = [], [], []
            t = 0
            while (not bool(finished.all())) and t < max_len:
                out, state, _ = self.decoder_step(emb, state, st, hf, mask, False, t)
                lg = self.logits(out)
                nxt = lg.argmax(1) * (~finished)
                finished = finished | (nxt == END)
                emb = p[d + "/word_embeddings"][nxt]
                syms.append(nxt.numpy())
                masks.append((~finished).numpy())
                logit_hist.append(lg.numpy())
                t += 1
            return np.stack(syms), np.stack(masks), np.stack(logit_hist)

    def beam_follow(self, src_ids, k: int, max_steps: int, alpha: float, follow=None, tie_margin=None):
        """The beam search through ``nm_oracle.beam_search_core`` (one restatement of the beam body for every parent
        decoder), optionally FOLLOWING another implementation's selections -- see ``beam_search_core``."""
        from . import nm_oracle as O
        with torch.no_grad():
            p, d = self.p, self.cfg.dec_name
            st, hf, mask, state = self._decode_setup(src_ids, k)
            rows = state[0].shape[0]
            table = p[d + "/word_embeddings"]
            out, state, _ = self.decoder_step(table[torch.full((rows,), START)], state, st, hf, mask, False, 0)
            box = {"state": state, "t": 1}

            def step_fn(src_rows, words):
                src = torch.as_tensor(np.asarray(src_rows, dtype=np.int64))
                prev = [s[src] for s in box["state"]]
                self.reorder_attention(src)
                o, box["state"], _ = self.decoder_step(table[torch.as_tensor(np.asarray(words, dtype=np.int64))], prev,
                                                       st, hf, mask, False, box["t"])
                box["t"] += 1
                return self.logits(o).numpy()
            return O.beam_search_core(self.logits(out).numpy(), step_fn, rows // k, k, max_steps, alpha,
                                      tie_margin=tie_margin, follow=follow)

    # -- beam search (decoders/beam_search_decoder.py:218-556) --------------------------------------
    def beam(self, src_ids, k: int, max_steps: int, alpha: float):
        with torch.no_grad():
            p, d, dt = self.p, self.cfg.dec_name, self.dtype
            st, hf, mask, state = self._decode_setup(src_ids, k)
            bsz = state[0].shape[0] // k
            rows = bsz * k
            table = p[d + "/word_embeddings"]
            out, state, _ = self.decoder_step(table[torch.full((rows,), START)], state, st, hf, mask, False, 0)
            lg = self.logits(out)
            vsz = lg.shape[1]
            token_ids = lg.argmax(1).view(1, bsz, k)
            logprob_sum = torch.tensor([0.0] + [-INF] * (k - 1), dtype=dt).repeat(bsz, 1)
            prev_lp = torch.log_softmax(lg, -1).view(bsz, k, vsz)
            lengths = torch.zeros(bsz, k, dtype=torch.int64)
            finished = torch.zeros(bsz, k, dtype=torch.bool)
            scores = torch.zeros(bsz, k, dtype=dt)
            fin_row = torch.full((vsz,), -INF, dtype=dt)
            fin_row[PAD] = 0.0
            bidx = torch.arange(bsz).view(-1, 1)
            step, min_gap = 1, float("inf")
            self.beam_gaps = []        # per step [B]: smallest non-zero relative gap among the top k+1 scores
            while (step - 1) < max_steps and not bool(finished.all()):
                fm = finished.to(dt).unsqueeze(-1)
                lp = (1.0 - fm) * prev_lp + fm * fin_row
                hyp = logprob_sum.unsqueeze(-1) + lp
                hyp_len = lengths + 1 - finished.to(torch.int64)
                pen = ((5.0 + hyp_len.to(dt)) / 6.0) ** alpha
                flat = (hyp / pen.unsqueeze(-1)).reshape(bsz, k * vsz)
                order = torch.argsort(-flat, dim=1, stable=True)[:, :k + 1]     # ties: lower index first
                top = torch.gather(flat, 1, order)
                live = ~finished.all(1)
                adj = top[:, :-1] - top[:, 1:]
                adj = torch.where(adj > 0, adj / top[:, :-1].abs().clamp_min(1e-30), torch.full_like(adj, float("inf")))
                self.beam_gaps.append(torch.where(live, adj.min(1).values, torch.full_like(adj[:, 0], float("inf"))).numpy())
                if order.shape[1] > k and bool(live.any()):
                    gap = (top[live, k - 1] - top[live, k]) / top[live, k - 1].abs().clamp_min(1e-30)
                    min_gap = min(min_gap, float(gap.min()))
                idx, top = order[:, :k], top[:, :k]
                word, beam = idx % vsz, idx // vsz
                lengths = hyp_len[bidx, beam]
                logprob_sum = hyp.reshape(bsz, k * vsz)[bidx, idx]
                finished = finished[bidx, beam] | (word == END)
                src = (bidx * k + beam).reshape(-1)
                state = [s[src] for s in state]
                self.reorder_attention(src)
                out, state, _ = self.decoder_step(table[word.reshape(-1)], state, st, hf, mask, False, step)
                prev_lp = torch.log_softmax(self.logits(out), -1).view(bsz, k, vsz)
                token_ids = torch.cat([token_ids[:, bidx, beam], word.unsqueeze(0)], 0)
                scores = top
                step += 1
            return token_ids.numpy(), scores.numpy(), min_gap
