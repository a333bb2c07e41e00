"""CPU restatement (torch, autograd) of the Transformer path: scaled dot-product attention
(attention/scaled_dot_product.py:24-226), TransformerEncoder (encoders/transformer.py:23-322),
TransformerDecoder (decoders/transformer.py:258-516, serial encoder-decoder attention of
attention/transformer_cross_layer.py:12-103), greedy and beam decoding around it.

TEST INFRASTRUCTURE ONLY.  PARITY PINNED to the reference's own code (see oracle/nm_oracle.py): fixtures
``transformer*``, ``fd_gradients_transformer*``, ``ini_beamsearch``, ``ensemble``.  Decoding here follows the
reference literally -- every step re-runs all layers over the whole prefix (:487-516) -- which is
what pins the engine's key/value-cache implementation to the reference arithmetic.
"""
import math
from typing import Dict, List, NamedTuple, Optional, Tuple

import numpy as np
import torch

from .general_ref import END, INF, PAD, START, UNK, dropout_mask, salt_of


class TConfig(NamedTuple):
    enc_name: str = "encoder"
    dec_name: str = "decoder"
    depth: int = 2
    n_heads: int = 2
    n_heads_self: int = 2
    n_heads_enc: int = 2
    enc_dropout: float = 1.0
    enc_att_dropout: float = 1.0
    dec_dropout: float = 1.0
    self_att_dropout: float = 1.0
    encdec_att_dropout: float = 1.0
    use_att_transform_bias: bool = False
    use_positional_encoding: bool = True
    tie_embeddings: bool = True
    supress_unk: bool = False
    target_space_id: Optional[int] = None
    extra_encoders: Tuple[str, ...] = ()      # names of further encoders (same hyper-parameters) the decoder attends to
    strategy: str = "serial"                  # attention_combination_strategy: serial | parallel | flat | hierarchical
    n_heads_hier: int = 1
    shared_embeddings: bool = False           # decoder(embeddings_source=<the encoder's input sequence>): one matrix
    scale_embeddings: bool = False            # EmbeddedSequence(scale_embeddings_by_depth=True), model/sequence.py:185-187


def position_signal(dimension: int, length: int) -> torch.Tensor:
    """encoders/transformer.py:23-45."""
    positions = torch.arange(length, dtype=torch.float32)
    num_timescales = dimension // 2
    inc = math.log(1.0e4) / (num_timescales - 1)
    inv = torch.exp(torch.arange(num_timescales, dtype=torch.float32) * -inc)
    scaled = positions[:, None] * inv[None, :]
    signal = torch.cat([torch.sin(scaled), torch.cos(scaled)], 1)
    if dimension % 2:
        signal = torch.nn.functional.pad(signal, (0, 1))
    return signal


def mask_future(e: torch.Tensor, mask_value: float = -1e9) -> torch.Tensor:
    """scaled_dot_product.py:72-93: keep the lower triangle (key <= query), the rest becomes ``mask_value``."""
    tril = torch.tril(torch.ones_like(e))
    return torch.where(tril == 1, e, torch.full_like(e, mask_value))


def mask_energies(e: torch.Tensor, keys_mask: torch.Tensor, mask_value: float = -1e9) -> torch.Tensor:
    """scaled_dot_product.py:45-69: ``e * m + (1 - m) * mask_value`` with the key mask broadcast over heads and queries."""
    m4 = keys_mask[:, None, None, :]
    return e * m4 + (1.0 - m4) * mask_value


class TransformerModel:
    def __init__(self, params: Dict[str, np.ndarray], cfg: TConfig, dtype=torch.float32, requires_grad=False):
        self.cfg, self.dtype = cfg, dtype
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in params.items()}
        self.global_step = 0

    def dropout(self, x, keep, train, *site):
        if keep == 1.0 or not train:
            return x
        mask = dropout_mask(x.numel(), keep, salt_of(self.global_step, *site))
        return x * torch.from_numpy(mask).to(self.dtype).view(x.shape)

    def layer_norm(self, x, prefix):
        g, b = self.p[prefix + "LayerNorm/gamma"], self.p[prefix + "LayerNorm/beta"]
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        return (x - mean) * torch.rsqrt(var + 1e-6) * g + b

    def attention(self, scope, queries, keys, keys_mask, heads, masked, att_keep, train, use_bias, site,
                  return_weights=False):
        """scaled_dot_product.py:98-226 with values == keys."""
        p = self.p
        dim = queries.shape[-1]
        dh = dim // heads

        def dense(x, name):
            y = x @ p["{}/{}/kernel".format(scope, name)]
            return y + p["{}/{}/bias".format(scope, name)] if use_bias else y
        q, k, v = queries, keys, keys
        if heads > 1:
            q, k, v = dense(queries, "query_proj"), dense(keys, "keys_proj"), dense(keys, "vals_proj")
        q = q / math.sqrt(dh)

        def split(x):
            b, t, _ = x.shape
            return x.view(b, t, heads, dh).permute(0, 2, 1, 3)
        q, k, v = split(q), split(k), split(v)
        e = q @ k.transpose(-1, -2)                                             # [B,H,Tq,Tk]
        if masked:                                                              # mask_future BEFORE the key mask
            e = mask_future(e)
        if keys_mask is not None:
            e = mask_energies(e, keys_mask)
        w = torch.softmax(e, -1)
        w = self.dropout(w, att_keep, train, *site)
        ctx = (w @ v).permute(0, 2, 1, 3).reshape(queries.shape[0], queries.shape[1], dim)
        if heads > 1:
            ctx = dense(ctx, "output_proj")
        return (ctx, w) if return_weights else ctx

    def target_embeddings(self):
        """autoregressive.py:253-267: the decoder's own ``word_embeddings`` or, with ``embeddings_source``, the
        source sequence's matrix (decoder inputs are NOT scaled by sqrt(E): the loop embeds through the base
        ``embed_input_symbols``, :269-272)."""
        if self.cfg.shared_embeddings:
            return self.p[self.cfg.enc_name + "_input/embedding_matrix_0"]
        return self.p[self.cfg.dec_name + "/word_embeddings"]

    def feedforward(self, scope, x, keep, train, site):
        p = self.p
        normed = self.layer_norm(x, scope + "/")
        hidden = torch.relu(normed @ p[scope + "/hidden_state/kernel"] + p[scope + "/hidden_state/bias"])
        hidden = self.dropout(hidden, keep, train, *site, "ff_hidden")
        out = hidden @ p[scope + "/output/kernel"] + p[scope + "/output/bias"]
        out = self.dropout(out, keep, train, *site, "ff_output")
        return out + x

    # -- encoder -------------------------------------------------------------------------------------
    def encode_all(self, src_ids, train: bool):
        """States and masks of every encoder the decoder attends to: ``src_ids`` is one id matrix, or a list
        with one matrix per encoder (cfg.enc_name, then cfg.extra_encoders)."""
        if isinstance(src_ids, np.ndarray):
            src_ids = [src_ids]
        names = (self.cfg.enc_name,) + tuple(self.cfg.extra_encoders)
        assert len(src_ids) == len(names)
        encoded = [self.encode(ids, train, name) for ids, name in zip(src_ids, names)]
        return [e[0] for e in encoded], [e[1] for e in encoded]

    def encode(self, src_ids: np.ndarray, train: bool, name: Optional[str] = None):
        cfg, p, name = self.cfg, self.p, name or self.cfg.enc_name
        ids = torch.as_tensor(src_ids.astype(np.int64))
        mask = (ids != PAD).to(self.dtype)
        x = p[name + "_input/embedding_matrix_0"][ids]
        if cfg.scale_embeddings:                   # model/sequence.py:185-187: scaled before the mask is applied
            x = x * (x.shape[-1] ** 0.5)
        x = x * mask.unsqueeze(-1)
        if cfg.target_space_id is not None:        # encoders/transformer.py:175-203: one row of a [32, D] table
            x = x + p[name + "/target_modality_embedding_matrix"][cfg.target_space_id].reshape(1, 1, -1)
        if cfg.use_positional_encoding:
            x = x + position_signal(x.shape[-1], x.shape[1]).to(self.dtype)
        x = self.dropout(x, cfg.enc_dropout, train, name, "encoder_inputs")
        for i in range(cfg.depth):
            pre = "{}/layer_{}".format(name, i)
            site = (name, "layer_{}".format(i))
            normed = self.layer_norm(x, pre + "/self_attention/")
            att = self.attention(pre + "/self_attention", normed, normed, mask, cfg.n_heads, False,
                                 cfg.enc_att_dropout, train, cfg.use_att_transform_bias,
                                 site + ("self_attention_weights",))
            x = self.dropout(att, cfg.enc_dropout, train, *site, "self_attention") + x
            x = self.feedforward(pre + "/feedforward", x, cfg.enc_dropout, train, site)
        x = self.layer_norm(x, name + "/")
        return x, mask, x.sum(1)

    # -- decoder layer stack over a whole (prefix) sequence ------------------------------------------
    def decoder_layers(self, x, mask, enc_states, enc_mask, train):
        cfg, name = self.cfg, self.cfg.dec_name
        for i in range(cfg.depth):
            pre = "{}/layer_{}".format(name, i)
            site = (name, "layer_{}".format(i))
            normed = self.layer_norm(x, pre + "/self_attention/")
            att = self.attention(pre + "/self_attention", normed, normed, mask, cfg.n_heads_self, True,
                                 cfg.self_att_dropout, train, cfg.use_att_transform_bias,
                                 site + ("self_attention_weights",))
            x = self.dropout(att, cfg.dec_dropout, train, *site, "self_attention") + x
            # attention/transformer_cross_layer.py: serial (:68-103) re-normalises the running result for every
            # encoder; parallel (:106-152) queries all encoders with one normalised input and sums
            if not isinstance(enc_states, (list, tuple)):
                enc_states, enc_mask = [enc_states], [enc_mask]
            top = pre + "/encdec_attention"
            if cfg.strategy == "flat":              # :236-268: one attention over the encoders concatenated in time
                enc_states, enc_mask = [torch.cat(list(enc_states), 1)], [torch.cat(list(enc_mask), 1)]
            queries = self.layer_norm(x, top + "/") if cfg.strategy != "serial" else None
            contexts = []
            for j, (states, smask) in enumerate(zip(enc_states, enc_mask)):
                scope = top if cfg.strategy == "flat" else top + "/enc_{}".format(j)
                normed = queries if queries is not None else self.layer_norm(x, scope + "/")
                att = self.attention(scope, normed, states, smask, cfg.n_heads_enc, False, cfg.encdec_att_dropout,
                                     train, False, site + ("encdec_weights", j))
                att = self.dropout(att, cfg.dec_dropout, train, *site, "encdec", j)
                if cfg.strategy == "hierarchical":
                    contexts.append(att)
                else:
                    x = att + x
            if cfg.strategy == "hierarchical":      # :155-232: the same queries attend to the stacked contexts
                bsz, steps, dim = x.shape
                stacked = torch.stack(contexts, 2).reshape(bsz * steps, len(contexts), dim)
                ones = torch.ones(bsz * steps, len(contexts), dtype=self.dtype)
                att = self.attention(top + "/enc_hier", queries.reshape(bsz * steps, 1, dim), stacked, ones,
                                     cfg.n_heads_hier, False, cfg.dec_dropout, train, False,
                                     site + ("encdec_hier_weights",))
                x = self.dropout(att.reshape(bsz, steps, dim), cfg.dec_dropout, train, *site, "encdec_hier") + x
            x = self.feedforward(pre + "/feedforward", x, cfg.dec_dropout, train, site)
        return self.layer_norm(x, name + "/")

    def logits(self, states):
        cfg, p = self.cfg, self.p
        if cfg.tie_embeddings:
            lg = states @ self.target_embeddings().t()
        else:
            lg = states @ p[cfg.dec_name + "/state_to_word_W"] + p[cfg.dec_name + "/state_to_word_b"]
        if cfg.supress_unk:
            unk = torch.zeros(lg.shape[-1], dtype=self.dtype)
            unk[UNK] = -1e9
            lg = lg + unk
        return lg

    def train_loss(self, src_ids, tgt_bt, train=True):
        """train_loop_result (:393-453): one pass over <s> + targets[:-1]; loss autoregressive.py:289-316."""
        cfg, p = self.cfg, self.p
        enc_states, enc_mask = self.encode_all(src_ids, train)
        bsz, steps = tgt_bt.shape
        dec_in = np.concatenate([np.full((bsz, 1), START, tgt_bt.dtype), tgt_bt[:, :-1]], 1)
        emb = self.target_embeddings()[torch.as_tensor(dec_in.astype(np.int64))]
        emb = self.dropout(emb, cfg.dec_dropout, train, cfg.dec_name, "embedded_input")
        tgt = torch.as_tensor(tgt_bt.astype(np.int64))
        tmask = (tgt != PAD).to(self.dtype)
        states = self.decoder_layers(emb, tmask, enc_states, enc_mask, train)
        logits = self.logits(states)                                  # [B,T,V]
        lp = torch.log_softmax(logits, -1)
        xent = -torch.gather(lp, 2, tgt.unsqueeze(-1)).squeeze(-1) * tmask
        return xent.sum() / tmask.sum(), logits

    def train_grads(self, src_ids, tgt_bt, train=True):
        loss, _ = self.train_loss(src_ids, tgt_bt, train)
        names = list(self.p)
        grads = torch.autograd.grad(loss, [self.p[n] for n in names], allow_unused=True)
        return float(loss.detach()), {n: (None if g is None else g.detach().numpy()) for n, g in zip(names, grads)}

    # -- decoding: next_state (:487-516) re-runs the stack over the whole prefix --------------------
    def _next_output(self, seq, seq_mask, enc_states, enc_mask):
        return self.decoder_layers(seq, seq_mask, enc_states, enc_mask, False)[:, -1]

    def greedy(self, src_ids, max_len: int, pick=None, temperature: float = 1.0):
        """``pick(t, logits) -> symbols`` replaces the argmax (autoregressive.py:466-473: tf.multinomial when the body
        samples -- the checker hands the engine's draws back in and restates each of them, tests/test_sampling_gpu.py);
        ``temperature``: logits /= temperature (:493)."""
        with torch.no_grad():
            table = self.target_embeddings()
            enc_states, enc_mask = self.encode_all(src_ids, False)
            rows = enc_states[0].shape[0]
            emb = table[torch.full((rows,), START)]
            finished = torch.zeros(rows, dtype=torch.bool)
            seq = torch.zeros(rows, 0, table.shape[1], dtype=self.dtype)
            seq_mask = torch.zeros(rows, 0, dtype=self.dtype)
            syms, masks, logit_hist = [], [], []
            t = 0
            while (not bool(finished.all())) and t < max_len:
                seq = torch.cat([seq, emb.unsqueeze(1)], 1)
                seq_mask = torch.cat([seq_mask, (~finished).to(self.dtype).unsqueeze(1)], 1)
                lg = self.logits(self._next_output(seq, seq_mask, enc_states, enc_mask))
                if temperature != 1.0:
                    lg = lg / temperature
                nxt = (torch.as_tensor(np.asarray(pick(t, lg.numpy())).astype(np.int64)) if pick else lg.argmax(1))
                nxt = nxt * (~finished)
                finished = finished | (nxt == END)
                emb = table[nxt]
                syms.append(nxt.numpy())
                masks.append((~finished).numpy())
                logit_hist.append(lg.numpy())
                t += 1
            return np.stack(syms), np.stack(masks), np.stack(logit_hist)

    def beam(self, src_ids, k: int, max_steps: int, alpha: float):
        """decoders/beam_search_decoder.py:218-556 around the Transformer parent."""
        with torch.no_grad():
            dt = self.dtype
            table = self.target_embeddings()
            enc_states, enc_mask = self.encode_all(src_ids, False)
            bsz = enc_states[0].shape[0]
            enc_states = [e.repeat_interleave(k, 0) for e in enc_states]
            enc_mask = [m.repeat_interleave(k, 0) for m in enc_mask]
            rows = bsz * k
            seq = table[torch.full((rows,), START)].unsqueeze(1)
            seq_mask = torch.ones(rows, 1, dtype=dt)
            lg = self.logits(self._next_output(seq, seq_mask, enc_states, enc_mask))
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
                order = torch.argsort(-flat, dim=1, stable=True)[:, :k + 1]
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
                seq = torch.cat([seq[src], table[word.reshape(-1)].unsqueeze(1)], 1)
                seq_mask = torch.cat([seq_mask[src], (~finished).reshape(-1, 1).to(dt)], 1)
                prev_lp = torch.log_softmax(self.logits(self._next_output(seq, seq_mask, enc_states, enc_mask)),
                                            -1).view(bsz, k, vsz)
                token_ids = torch.cat([token_ids[:, bidx, beam], word.unsqueeze(0)], 0)
                scores = top
                step += 1
            return token_ids.numpy(), scores.numpy(), min_gap
