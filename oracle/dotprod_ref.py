"""CPU restatement (torch, autograd) of scaled dot-product attention on the RNN decoder path:
``MultiHeadAttention`` / ``ScaledDotProdAttention`` (neuralmonkey/attention/scaled_dot_product.py:
98-226 the function, :247-400 the classes) plugged into ``oracle/general_ref.py``'s decoder -- the
model family of the reference's tests/factored.ini and tests/post-edit.ini.

TEST INFRASTRUCTURE ONLY -- imported by ``tests/`` alone.  PARITY UNPINNED (see
``oracle/nm_oracle.py``).
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
