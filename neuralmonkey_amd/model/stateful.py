"""Stateful interfaces (mirror of neuralmonkey/model/stateful.py:22-103).

Every ``@tensor`` of the reference is a Fetch here: call it with the run
context to obtain the device tensor.  The static sizes (``output_size``,
``dimension``) are plain ints so that dependent parts can size their variables
before any data flows.
"""
from .model_part import GenericModelPart


class Stateful(GenericModelPart):
    @property
    def output(self):
        """Fetch of a [batch, state_size] tensor."""
        raise NotImplementedError("Abstract property")

    @property
    def output_size(self) -> int:
        raise NotImplementedError("Abstract property")


class TemporalStateful(GenericModelPart):
    @property
    def temporal_states(self):
        """Fetch of a [batch, time, state_size] tensor."""
        raise NotImplementedError("Abstract property")

    @property
    def temporal_mask(self):
        """Fetch of a [batch, time] float 0/1 tensor."""
        raise NotImplementedError("Abstract property")

    @property
    def dimension(self) -> int:
        raise NotImplementedError("Abstract property")


class SpatialStateful(GenericModelPart):
    @property
    def spatial_states(self):
        """Fetch of a [batch, width, height, state_size] tensor."""
        raise NotImplementedError("Abstract property")

    @property
    def spatial_mask(self):
        raise NotImplementedError("Abstract property")

    @property
    def dimension(self) -> int:
        raise NotImplementedError("Abstract property")


class TemporalStatefulWithOutput(Stateful, TemporalStateful):
    pass


class SpatialStatefulWithOutput(Stateful, SpatialStateful):
    pass
