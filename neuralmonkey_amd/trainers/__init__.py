"""Trainers of the attention-decoder path: one fused forward + backward per step on the HIP engine
(``generic_trainer``), thin front ends under the class names INI files use."""
from .objective import CostObjective                        # noqa: F401
from .generic_trainer import GenericTrainer                 # noqa: F401
from .cross_entropy_trainer import CrossEntropyTrainer      # noqa: F401
from .delayed_update_trainer import DelayedUpdateTrainer    # noqa: F401
from .multitask_trainer import MultitaskTrainer             # noqa: F401

__all__ = ["CostObjective", "CrossEntropyTrainer", "DelayedUpdateTrainer", "GenericTrainer", "MultitaskTrainer"]
