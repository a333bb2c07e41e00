"""CrossEntropyTrainer (mirror of neuralmonkey/trainers/cross_entropy_trainer.py)."""
from typing import Any, List

from ..optimizers import Optimizer
from .generic_trainer import GenericTrainer
from .objective import CostObjective, Objective, ObjectiveWeight


def xent_objective(decoder, weight=None) -> Objective:
    """Deprecated alias kept for old configs (cross_entropy_trainer.py:13-17)."""
    return CostObjective(decoder, weight)


# pylint: disable=too-many-arguments
class CrossEntropyTrainer(GenericTrainer):
    def __init__(self, decoders: List[Any], decoder_weights: List[ObjectiveWeight] = None,
                 l1_weight: float = 0., l2_weight: float = 0., clip_norm: float = None,
                 optimizer: Optimizer = None, var_scopes: List[str] = None,
                 var_collection: str = None) -> None:
        if decoder_weights is None:
            decoder_weights = [None for _ in decoders]
        if len(decoder_weights) != len(decoders):
            raise ValueError("decoder_weights (length {}) do not match decoders (length {})"
                             .format(len(decoder_weights), len(decoders)))
        objectives = [CostObjective(dec, w) for dec, w in zip(decoders, decoder_weights)]
        GenericTrainer.__init__(self, objectives=objectives, l1_weight=l1_weight, l2_weight=l2_weight,
                                clip_norm=clip_norm, optimizer=optimizer, var_scopes=var_scopes,
                                var_collection=var_collection)
