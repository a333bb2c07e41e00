"""``MultitaskTrainer``: task switching by round robin (interface of
neuralmonkey/trainers/multitask_trainer.py:12-48; used by tests/bahdanau.ini).

The object is a trainer only by name: it owns no step of its own.  Each call of ``get_executable``
returns the executable of the next wrapped trainer, so batch i is trained by trainer ``i mod n``.
Every wrapped trainer keeps its own Adam slots and update count (see ``GenericTrainer._adam_state``),
as TensorFlow's per-optimizer slot variables do in the reference.
"""
from itertools import chain
from typing import Any, Dict, List

from ..runners.base_runner import GraphExecutor
from .generic_trainer import GenericTrainer


class MultitaskTrainer(GraphExecutor):
    def __init__(self, trainers: List[GenericTrainer]) -> None:
        if len(trainers) == 0:
            raise ValueError("MultitaskTrainer needs at least one trainer")
        super().__init__(set(trainers))
        self.trainers = trainers
        self.trainer_idx = 0                       # whose turn it is (the reference's attribute name)

    def get_executable(self, compute_losses: bool = True, summaries: bool = True, num_sessions: int = 1):
        turn, self.trainer_idx = self.trainer_idx, (self.trainer_idx + 1) % len(self.trainers)
        return self.trainers[turn].get_executable(compute_losses, summaries, num_sessions)

    @property
    def fetches(self) -> Dict[str, Any]:
        """Union of the wrapped trainers' fetches (later trainers win on equal keys)."""
        return dict(chain.from_iterable(trainer.fetches.items() for trainer in self.trainers))

    def var_list(self, store) -> List[str]:
        """Every variable some wrapped trainer updates, once, in first-seen order."""
        seen: Dict[str, None] = {}
        for trainer in self.trainers:
            seen.update(dict.fromkeys(trainer.var_list(store)))
        return list(seen)
