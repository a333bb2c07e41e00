from .cross_entropy_trainer import CrossEntropyTrainer      # noqa: F401
from .generic_trainer import GenericTrainer                 # noqa: F401
from .objective import CostObjective                        # noqa: F401
from .delayed_update_trainer import DelayedUpdateTrainer  # noqa: F401
from .multitask_trainer import MultitaskTrainer            # noqa: F401
