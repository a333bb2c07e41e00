"""CPU restatement (torch, autograd) of CoverageAttention (attention/coverage.py:19-66, Tu et al. 2016) on top of
``oracle/general_ref.py``.

TEST INFRASTRUCTURE ONLY -- imported by ``tests/`` alone.  PARITY UNPINNED FOR THIS ONE MODULE (the rest of the oracle is pinned to the reference's own code, see
``oracle/nm_oracle.py``): the fixture ``defects`` records that upstream
line 52 (``weights_in_time.size()`` on a tf.Tensor) cannot build, so there is no reference behaviour to pin;
what is restated is the arithmetic the lines spell out, with ``weights_in_time`` read as the loop state's
``weights`` history [t,B,S] (feed_forward.py:158-159), whose sum over t is zero at the first step.
"""
import torch

from . import general_ref as G


class CoverageModel(G.GeneralModel):
    def __init__(self, params, cfg: G.Config, max_fertility: int = 5, dtype=torch.float32, requires_grad=False):
        G.GeneralModel.__init__(self, params, cfg, dtype, requires_grad)
        self.max_fertility = max_fertility
        self._wsum = None

    def attention(self, query, st, hf, mask):
        p, a = self.p, self.cfg.att_name
        if self._step_t == 0:                                                   # empty history (:53-56)
            self._wsum = torch.zeros(query.shape[0], st.shape[1], dtype=self.dtype)
        fertility = 1e-8 + self.max_fertility * torch.sigmoid(
            (p[a + "/fertility_matrix"].view(1, 1, -1) * st).sum(2))            # :47-50
        coverage = self._wsum / fertility * mask                                # :57
        y = query @ p[a + "/Attention/attn_query_projection"] + p[a + "/attn_projection_bias"]
        e = (p[a + "/attn_similarity_v"] * torch.tanh(
            hf + y.unsqueeze(1) + p[a + "/coverage_matrix"].view(1, 1, -1) * coverage.unsqueeze(-1))).sum(-1)   # :58-64
        w_all = torch.softmax(e, -1) * mask                                     # feed_forward.py:139-144
        w = w_all / (w_all.sum(1, keepdim=True) + 1e-8)
        self._wsum = self._wsum + w
        return (w.unsqueeze(-1) * st).sum(1), w

    def reorder_attention(self, src):
        """Beam search keeps the coverage of the hypothesis each survivor continues."""
        self._wsum = self._wsum[src]
