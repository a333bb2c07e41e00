"""``CrossEntropyTrainer``: the trainer INI files name for maximum-likelihood training
(interface of neuralmonkey/trainers/cross_entropy_trainer.py:21-53).

It is nothing but a ``GenericTrainer`` whose objectives are the ``cost`` of each listed decoder,
optionally weighted: everything that happens in a step -- one fused forward + backward over all
objectives, the data-parallel all-reduce, L1/L2 terms, per-tensor clipping, the Adam kernel -- lives in
``generic_trainer.py``.  ``decoder_weights`` entries may be numbers or ``None`` (weight 1).
"""
from typing import Any, List, Optional

from ..optimizers import Optimizer
from .generic_trainer import GenericTrainer
from .objective import CostObjective, Objective, ObjectiveWeight


def _weighted_costs(decoders: List[Any], weights: Optional[List[ObjectiveWeight]]) -> List[Objective]:
    if weights is None:
        return [CostObjective(decoder, None) for decoder in decoders]
    if len(weights) != len(decoders):
        raise ValueError("decoder_weights (length {}) do not match decoders (length {})"
                         .format(len(weights), len(decoders)))
    return [CostObjective(decoder, weight) for decoder, weight in zip(decoders, weights)]


class CrossEntropyTrainer(GenericTrainer):
    # pylint: disable=too-many-arguments
    def __init__(self, decoders: List[Any], decoder_weights: List[ObjectiveWeight] = None,
                 l1_weight: float = 0.0, l2_weight: float = 0.0, clip_norm: float = None,
                 optimizer: Optimizer = None, var_scopes: List[str] = None,
                 var_collection: str = None) -> None:
        super().__init__(_weighted_costs(decoders, decoder_weights), l1_weight, l2_weight, clip_norm, optimizer,
                         var_scopes, var_collection)


def xent_objective(decoder: Any, weight: ObjectiveWeight = None) -> Objective:
    """The name old configurations use for ``CostObjective`` (:13-17)."""
    return CostObjective(decoder, weight)
