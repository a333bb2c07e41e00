"""CPU restatement (torch, autograd) of the multi-source attention combinations
(neuralmonkey/attention/combination.py) on top of ``oracle/general_ref.py``'s decoder: a sentence
encoder and a spatial (image feature map) encoder, attended either by one FlatMultiAttention or by
a HierarchicalMultiAttention over one Bahdanau attention per encoder -- the model family of the
reference's tests/flat-multiattention.ini and tests/hier-multiattention.ini.

TEST INFRASTRUCTURE ONLY -- imported by ``tests/`` alone.  PARITY PINNED to the reference's own code (see
``oracle/nm_oracle.py``): fixtures ``ms_flat*``, ``ms_hier*``, ``fd_gradients_ms_*``, ``ini_flat`` of
``tests/golden/ref_exec/``; each function cites the reference lines it restates.
"""
from typing import NamedTuple, Tuple

import numpy as np
import torch

from .general_ref import Config, GeneralModel


class MultiConfig(NamedTuple):
    kind: str = "flat"                     # "flat" | "hier"
    att_name: str = "wrapper"
    state_size: int = 5                    # attention_state_size
    share: bool = False                    # share_attn_projections
    sentinel: bool = False                 # use_sentinels
    image_name: str = "imagenet"
    image_spatial: Tuple = (None, None)    # SpatialFiller (ff_hidden_dim, projection_dim)
    child_names: Tuple = ("att_text", "att_image")      # hierarchical: one Attention per encoder


class MultiSourceModel(GeneralModel):
    """``src`` everywhere is the pair (source ids [B,S], feature maps [B,H,W,D])."""

    def __init__(self, params, cfg: Config, mcfg: MultiConfig, dtype=torch.float32, requires_grad=False):
        GeneralModel.__init__(self, params, cfg, dtype, requires_grad)
        self.m = mcfg

    # -- encoders: decoder.encoders = [sentence encoder, spatial filler] ---------------------------
    def encode(self, src, train: bool):
        ids, maps = src
        s_txt, m_txt, f_txt = GeneralModel.encode(self, ids, train)
        s_img, m_img, f_img = self.encode_spatial(maps, self.m.image_name, self.m.image_spatial)
        return (s_txt, s_img), (m_txt, m_img), torch.cat([f_txt, f_img], 1)

    def _step_name(self, local: str) -> str:
        """Variables created inside attention(): decoder step scope / attention_<name> (combination.py:254,396)."""
        return "{}/attention_decoder/attention_{}/{}".format(self.cfg.dec_name, self.m.att_name, local)

    def attention_setup(self, states, train):
        m, p = self.m, self.p
        a = m.att_name
        if m.kind == "flat":                                              # get_encoder_projections (:199-232)
            keys, vals = [], []
            for i, st in enumerate(states):
                k = st @ p["{}/logits_projections/proj_matrix_{}".format(a, i)] \
                    + p["{}/logits_projections/proj_bias_{}".format(a, i)]
                keys.append(k)
                vals.append(k if m.share else st @ p["{}/context_projections/proj_matrix_{}".format(a, i)]
                            + p["{}/context_projections/proj_bias_{}".format(a, i)])
            return {"keys": keys, "vals": vals}, None
        children = []
        for name, st in zip(m.child_names, states):                       # feed_forward.py:47-51,105-118
            children.append({"name": name, "states": st, "hf": st @ p[name + "/attn_key_projection"]})
        return {"children": children}, None

    def context_size(self, st) -> int:
        return self.m.state_size

    def repeat_sources(self, st, hf, mask, rep: int):
        r = lambda x: x.repeat_interleave(rep, 0)
        if self.m.kind == "flat":
            st = {"keys": [r(x) for x in st["keys"]], "vals": [r(x) for x in st["vals"]]}
        else:
            st = {"children": [{"name": c["name"], "states": r(c["states"]), "hf": r(c["hf"])}
                               for c in st["children"]]}
        return st, None, tuple(r(x) for x in mask)

    # -- pieces of combination.py ----------------------------------------------------------------
    def _vector_logit(self, projected_state, vector, scope: str):
        """:74-103 -> (projection for the context [R,A], logit [R,1])."""
        p, pre = self.p, self._step_name("{}_logit".format(scope))
        proj_logit = vector @ p[pre + "/vector_projection/kernel"] + p[pre + "/vector_projection/bias"]
        if self.m.share:
            proj_ctx = proj_logit
        else:
            proj_ctx = vector @ p[pre + "/vector_ctx_proj/kernel"] + p[pre + "/vector_ctx_proj/bias"]
        v = p[self.m.att_name + "/attn_v"].reshape(-1)      # the reference declares it [1, 1, A] (combination.py:65-69)
        logit = (v * torch.tanh(projected_state + proj_logit)).sum(-1, keepdim=True) + p[pre + "/vector_bias"]
        return proj_ctx, logit

    def _sentinel(self, state, prev_state, input_):
        """:326-342."""
        p = self.p
        gate = torch.sigmoid(torch.cat([prev_state, input_], 1) @ p[self._step_name("sentinel/dense/kernel")]
                             + p[self._step_name("sentinel/dense/bias")])
        return gate * state

    def attention(self, query, st, hf, mask):
        m, p = self.m, self.p
        prev_state, rnn_input = self._step_extra
        projected = query @ p[self._step_name("dense/kernel")] + p[self._step_name("dense/bias")]     # [R,A]
        v = p[m.att_name + "/attn_v"].reshape(-1)           # [1, 1, A] in the reference (combination.py:65-69)
        if m.kind == "flat":                                              # :245-296
            logits = []
            for i, k in enumerate(st["keys"]):
                logits.append((v * torch.tanh(projected.unsqueeze(1) + k)).sum(-1)
                              + p["{}/attn_bias_{}".format(m.att_name, i)])
            vals, masks = list(st["vals"]), list(mask)
            if m.sentinel:
                value = self._sentinel(query, prev_state, rnn_input)
                proj_sent, sent_logit = self._vector_logit(projected, value, "sentinel")
                logits.append(sent_logit)
                vals.append(proj_sent.unsqueeze(1))
                masks.append(torch.ones(query.shape[0], 1, dtype=self.dtype))
            soft = torch.softmax(torch.cat(logits, 1), -1) * torch.cat(masks, 1)          # _renorm_softmax :301-307
            w = soft / (soft.sum(1, keepdim=True) + 1e-8)
            return (w.unsqueeze(-1) * torch.cat(vals, 1)).sum(1), w
        # hierarchical (:389-457)
        vectors, names = [], []
        for child, cmask in zip(st["children"], mask):
            n = child["name"]
            y = query @ p[n + "/Attention/attn_query_projection"] + p[n + "/attn_projection_bias"]
            e = (p[n + "/attn_similarity_v"] * torch.tanh(child["hf"] + y.unsqueeze(1))).sum(-1) + p[n + "/attn_bias"]
            w_all = torch.softmax(e, -1) * cmask
            cw = w_all / (w_all.sum(1, keepdim=True) + 1e-8)
            vectors.append((cw.unsqueeze(-1) * child["states"]).sum(1))
            names.append(n)
        if m.sentinel:
            vectors.append(self._sentinel(query, prev_state, rnn_input))
            names.append("sentinel")
        proj_ctxs, logits = zip(*[self._vector_logit(projected, vec, name) for vec, name in zip(vectors, names)])
        distr = torch.softmax(torch.cat(logits, 1), -1)
        if m.share:
            outputs = proj_ctxs
        else:
            outputs = []
            for vec, name in zip(vectors, names):
                scope = self._step_name("proj_sentinel" if name == "sentinel" else "proj_attn_{}".format(name))
                outputs.append(vec @ p[scope + "/kernel"] + p[scope + "/bias"])
        ctx = sum(distr[:, i:i + 1] * o for i, o in enumerate(outputs))
        return ctx, distr


def pack_sources(src_ids: np.ndarray, maps: np.ndarray):
    return (src_ids, maps)
