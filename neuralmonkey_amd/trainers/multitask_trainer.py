"""MultitaskTrainer (mirror of neuralmonkey/trainers/multitask_trainer.py:12-48): a task-switching
schedule -- every ``get_executable`` hands out the next trainer of the list, round robin
(tests/bahdanau.ini)."""
from typing import Any, Dict, List

from ..runners.base_runner import GraphExecutor
from .generic_trainer import GenericTrainer


class MultitaskTrainer(GraphExecutor):
    def __init__(self, trainers: List[GenericTrainer]) -> None:
        if not trainers:
            raise ValueError("MultitaskTrainer needs at least one trainer")
        GraphExecutor.__init__(self, set(trainers))
        self.trainers = trainers
        self.trainer_idx = 0

    def var_list(self, store) -> List[str]:
        names: List[str] = []
        for trainer in self.trainers:
            names.extend(n for n in trainer.var_list(store) if n not in names)
        return names

    def get_executable(self, compute_losses: bool = True, summaries: bool = True, num_sessions: int = 1):
        focused = self.trainers[self.trainer_idx]
        self.trainer_idx = (self.trainer_idx + 1) % len(self.trainers)
        return focused.get_executable(compute_losses, summaries, num_sessions)

    @property
    def fetches(self) -> Dict[str, Any]:
        fetches: Dict[str, Any] = {}
        for trainer in self.trainers:
            fetches.update(trainer.fetches)
        return fetches
