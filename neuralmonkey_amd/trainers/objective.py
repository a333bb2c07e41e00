"""What a trainer optimises: named losses of model parts, each with an optional weight (the interface of
neuralmonkey/trainers/objective.py; ``CostObjective`` is its :62-102 -- the ``cost`` of a decoder)."""
from typing import Optional

from ..model.model_part import GenericModelPart

ObjectiveWeight = Optional[float]


class Objective:
    """Base: a name for logs, the part whose loss it is.  ``gradients`` (hand-made gradients, used by the reference's
    reinforcement-learning trainers only) and ``weight`` default to None = "differentiate the loss" / "weight 1"."""
    gradients = None
    weight: ObjectiveWeight = None

    def __init__(self, name: str, decoder) -> None:
        self._name, self._decoder = name, decoder

    name = property(lambda self: self._name)
    decoder = property(lambda self: self._decoder)

    @property
    def loss(self):
        raise NotImplementedError()


class CostObjective(Objective):
    def __init__(self, decoder: GenericModelPart, weight: ObjectiveWeight = None) -> None:
        if "cost" not in dir(decoder):
            raise TypeError("The decoder does not have the 'cost' attribute")
        Objective.__init__(self, "{} - cost".format(decoder), decoder)
        self.weight = weight

    @property
    def loss(self):
        return self.decoder.cost
