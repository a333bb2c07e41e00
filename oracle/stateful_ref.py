"""TEST INFRASTRUCTURE -- CPU restatement of ``StatefulContext`` (attention/stateful_context.py:14-90): the encoder's
OUTPUT vector as the decoder's context at every step (attention weights: a column of ones), on top of
oracle/general_ref.py.  Pinned to the reference's own execution by ``tests/golden/ref_exec/stateful_context.npz``."""
import torch

from oracle import general_ref as G


class StaticContextModel(G.GeneralModel):
    """The oracle's decoder with ``attention()`` == the encoder output (stateful_context.py:60-78)."""

    def encode(self, src_ids, train):
        states, mask, final = super().encode(src_ids, train)
        self._final = final
        return states, mask, final

    def attention_setup(self, states, train):
        return self._final, self._final

    def context_size(self, st) -> int:
        return st.shape[-1]

    def repeat_sources(self, st, hf, mask, rep: int):
        return st.repeat_interleave(rep, 0), hf.repeat_interleave(rep, 0), mask.repeat_interleave(rep, 0)

    def attention(self, query, st, hf, mask):
        return st, torch.ones(st.shape[0], 1, dtype=self.dtype)            # :70: weights of width 1
