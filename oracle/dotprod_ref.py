"""CPU restatement (torch, autograd) of scaled dot-product attention on the RNN decoder path:
``MultiHeadAttention`` / ``ScaledDotProdAttention`` (neuralmonkey/attention/scaled_dot_product.py:
98-226 the function, :247-400 the classes) plugged into ``oracle/general_ref.py``'s decoder -- the
model family of the reference's tests/factored.ini and tests/post-edit.ini.

TEST INFRASTRUCTURE ONLY -- imported by ``tests/`` alone.  PARITY PINNED to the reference's own code (see
``oracle/nm_oracle.py``): fixtures ``dotprod_heads*``, ``factored_smoothing``, ``fd_gradients_dotprod``, ``ini_factored``,
``ini_postedit`` of ``tests/golden/ref_exec/``.
"""
import math

import torch

from .general_ref import Config, GeneralModel


class DotProdModel(GeneralModel):
    """``n_heads`` heads over the encoder states (keys == values == the sentence encoder)."""

    def __init__(self, params, cfg: Config, n_heads: int, att_dropout: float = 1.0, dtype=torch.float32,
                 requires_grad=False):
        GeneralModel.__init__(self, params, cfg, dtype, requires_grad)
        self.n_heads = n_heads
        self.att_keep = att_dropout

    def _w(self, proj: str):
        return self.p["{}/attention_decoder/{}/kernel".format(self.cfg.dec_name, proj)]

    def attention_setup(self, states, train):
        self._train = train
        if self.n_heads > 1:                                              # :170-176 (no bias)
            return states @ self._w("vals_proj"), states @ self._w("keys_proj")
        return states, states

    def context_size(self, st) -> int:
        return self.cfg.rnn_size

    def attention(self, query, st, hf, mask):
        heads = self.n_heads
        values, keys = st, hf
        q = query @ self._w("query_proj") if heads > 1 else query
        dim = q.shape[-1]
        dh = dim // heads
        q = (q / math.sqrt(dh)).unsqueeze(1)                              # [R,1,D]

        def split(x):
            b, t, _ = x.shape
            return x.view(b, t, heads, dh).permute(0, 2, 1, 3)
        e = split(q) @ split(keys).transpose(-1, -2)                      # [R,H,1,S]
        m4 = mask[:, None, None, :]
        e = e * m4 + (1.0 - m4) * -1e9                                    # mask_energies :45-69
        w = torch.softmax(e, -1)
        w = self.dropout(w, self.att_keep, self._train, self.cfg.att_name, "weights", self._step_t)
        ctx = (w @ split(values)).permute(0, 2, 1, 3).reshape(query.shape[0], dim)
        if heads > 1:
            ctx = ctx @ self._w("output_proj")
        return ctx, w.reshape(query.shape[0], -1)


class PostEditModel(GeneralModel):
    """The model of the reference's tests/post-edit.ini: ``decoder.encoders = [translation encoder, source encoder]``,
    ``decoder.attentions = [MultiHeadAttention(keys: source encoder, values: translation encoder), ScaledDotProd-
    Attention(source encoder)]``, decoder embeddings borrowed from the translation's input sequence
    (``embeddings_source``, decoders/autoregressive.py:226-251).  ``src`` everywhere is the pair
    (source ids [B,S], translation ids [B,S']); the two contexts travel as one concatenated vector, in the order of
    ``attentions`` -- which is the order the output projection concatenates them in (decoders/decoder.py:291-346).

    ``cfg`` describes the decoder and the source encoder (``enc_name``), ``trans`` the translation encoder; the
    parameter dictionary follows the oracle's naming for input sequences (``<encoder>_input/...``) and for the
    decoder's own embedding matrix -- alias the reference's names before building (see the test)."""

    def __init__(self, params, cfg: Config, trans: Config, n_heads: int, dtype=torch.float32, requires_grad=False):
        GeneralModel.__init__(self, params, cfg, dtype, requires_grad)
        self.n_heads = n_heads
        self.translation = GeneralModel({}, trans, dtype)
        self.translation.p = self.p                                    # one set of variables

    def _w(self, proj: str):
        return self.p["{}/attention_decoder/{}/kernel".format(self.cfg.dec_name, proj)]

    def encode(self, src, train: bool):
        ids, mt_ids = src
        s_src, m_src, f_src = GeneralModel.encode(self, ids, train)
        self.translation.global_step = self.global_step
        s_mt, m_mt, f_mt = self.translation.encode(mt_ids, train)
        return (s_mt, s_src), (m_mt, m_src), torch.cat([f_mt, f_src], 1)

    def attention_setup(self, states, train):
        s_mt, s_src = states
        # scaled_dot_product.py:170-176: keys and values of the multi-head attention are projected (no bias); with one
        # head nothing is (:163-168)
        st = {"values": s_mt @ self._w("vals_proj"), "keys": s_src @ self._w("keys_proj"), "source": s_src}
        return st, None

    def context_size(self, st) -> int:
        return st["values"].shape[-1] + st["source"].shape[-1]

    def repeat_sources(self, st, hf, mask, rep: int):
        r = lambda x: x.repeat_interleave(rep, 0)
        return {k: r(v) for k, v in st.items()}, None, tuple(r(m) for m in mask)

    @staticmethod
    def _dot_attention(q, keys, values, key_mask, heads: int):
        """attention() of scaled_dot_product.py:98-226 for one query per row: q [R,D], keys / values [R,S,D]."""
        dim = q.shape[-1]
        dh = dim // heads
        q = (q / math.sqrt(dh)).unsqueeze(1)

        def split(x):
            b, t, _ = x.shape
            return x.view(b, t, heads, dh).permute(0, 2, 1, 3)
        e = split(q) @ split(keys).transpose(-1, -2)                      # [R,H,1,S]
        m4 = key_mask[:, None, None, :]
        e = e * m4 + (1.0 - m4) * -1e9                                    # mask_energies :45-69
        w = torch.softmax(e, -1)
        return (w @ split(values)).permute(0, 2, 1, 3).reshape(q.shape[0], dim), w.reshape(q.shape[0], -1)

    def attention(self, query, st, hf, mask):
        m_mt, m_src = mask
        # both attentions mask by their KEYS encoder (attention_mask :289-291): the source sentence
        ctx_mt, w = self._dot_attention(query @ self._w("query_proj"), st["keys"], st["values"], m_src, self.n_heads)
        ctx_mt = ctx_mt @ self._w("output_proj")
        ctx_src, _ = self._dot_attention(query, st["source"], st["source"], m_src, 1)
        return torch.cat([ctx_mt, ctx_src], 1), w
