"""Training objectives (mirror of neuralmonkey/trainers/objective.py)."""
from typing import Optional

from ..model.model_part import GenericModelPart

ObjectiveWeight = Optional[float]


class Objective:
    def __init__(self, name: str, decoder) -> None:
        self._name = name
        self._decoder = decoder

    @property
    def decoder(self):
        return self._decoder

    @property
    def name(self) -> str:
        return self._name

    @property
    def loss(self):
        raise NotImplementedError()

    @property
    def gradients(self):
        return None

    @property
    def weight(self) -> ObjectiveWeight:
        return None


class CostObjective(Objective):
    """objective.py:62-102: the decoder's ``cost`` with an optional weight."""

    def __init__(self, decoder: GenericModelPart, weight: ObjectiveWeight = None) -> None:
        if "cost" not in dir(decoder):
            raise TypeError("The decoder does not have the 'cost' attribute")
        super().__init__("{} - cost".format(str(decoder)), decoder)
        self._weight = weight

    @property
    def loss(self):
        return getattr(self.decoder, "cost")

    @property
    def weight(self) -> ObjectiveWeight:
        return self._weight
