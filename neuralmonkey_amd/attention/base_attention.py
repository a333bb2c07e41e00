"""Attention base class (mirror of neuralmonkey/attention/base_attention.py).

The reference threads an ``AttentionLoopState`` (growing contexts / weights
tensors, namedtuples.py) through tf.while_loop.  Here the loop state is a
pre-allocated [T,R,*] history buffer plus the step index; ``attention()`` has
the same argument list and returns ``(context, next_loop_state)``."""
from typing import Any, Dict, NamedTuple, Optional, Union

import torch

from ..model.model_part import InitializerSpecs, ModelPart
from ..model.stateful import SpatialStateful, TemporalStateful

Attendable = Union[TemporalStateful, SpatialStateful]


class AttentionLoopState(NamedTuple):
    contexts: torch.Tensor     # [T_max, R, C]  rows < step are valid
    weights: torch.Tensor      # [T_max, R, S]
    step: int


def get_attention_states(encoder: Attendable, ctx) -> torch.Tensor:
    """base_attention.py:79-97: [B, S, C] (spatial maps flattened)."""
    if isinstance(encoder, TemporalStateful):
        return encoder.temporal_states(ctx)
    if isinstance(encoder, SpatialStateful):
        st = encoder.spatial_states(ctx)
        return st.reshape(st.shape[0], st.shape[1] * st.shape[2], st.shape[3])
    raise TypeError("Unknown encoder type")


def get_attention_mask(encoder: Attendable, ctx) -> Optional[torch.Tensor]:
    """base_attention.py:100-122."""
    if isinstance(encoder, TemporalStateful):
        mask = encoder.temporal_mask(ctx)
        if mask is None:
            raise ValueError("The encoder temporal mask should not be none")
        return mask
    if isinstance(encoder, SpatialStateful):
        mask = encoder.spatial_mask(ctx)
        if mask is None:
            return None
        return mask.reshape(mask.shape[0], mask.shape[1] * mask.shape[2])
    raise TypeError("Unknown encoder type")


class BaseAttention(ModelPart):
    def __init__(self, name: str, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        ModelPart.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.query_state_size: Optional[int] = None
        self._histories: Dict[str, Any] = {}

    @property
    def histories(self) -> Dict[str, Any]:
        return self._histories

    def attention(self, ctx, query, decoder_prev_state, decoder_input, loop_state):
        raise NotImplementedError("Abstract method")

    def initial_loop_state(self, ctx, rows: int, max_steps: int):
        raise NotImplementedError("Abstract method")

    def finalize_loop(self, key: str, last_loop_state: Any) -> None:
        raise NotImplementedError("Abstract method")

    @property
    def context_vector_size(self) -> int:
        raise NotImplementedError("Abstract property")

    def visualize_attention(self, key: str, max_outputs: int = 16) -> None:
        """TensorBoard image summaries are out of scope."""
