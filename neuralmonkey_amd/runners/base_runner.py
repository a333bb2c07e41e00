"""Graph-executor protocol (mirror of neuralmonkey/runners/base_runner.py).

``GraphExecutor.get_executable`` -> ``Executable.next_to_execute()`` returns
``(fetches, [feed_dict per session])``; the manager evaluates the fetches and
hands one result dict per session to ``collect_results`` until ``result`` is
set.  Fetches are ``runtime.Fetch`` handles instead of tf.Tensors."""
from typing import Any, Dict, Generic, List, NamedTuple, Optional, Set, Tuple, TypeVar, Union

import numpy as np

from ..checking import check_constructor_chain
from ..model.model_part import Feedable, GenericModelPart, Parameterized


def encoder_side_fetches(decoder) -> List[Any]:
    """What a decoding run needs that does not depend on the decoding loop itself: the states / masks / outputs
    of the decoder's encoders, the keys of its attentions, its initial state.  ``TensorFlowManager.execute(...,
    lookahead=next_batch)`` evaluates these for the NEXT batch on a second stream while the current batch decodes."""
    from ..runtime import Fetch
    found: List[Any] = []

    def take(owner, names):
        for name in names:
            try:
                handle = getattr(owner, name, None)
            except Exception:                # pylint: disable=broad-except   (a property that needs a run context)
                handle = None
            if isinstance(handle, Fetch):
                found.append(handle)
    for enc in getattr(decoder, "encoders", None) or []:
        take(enc, ("temporal_states", "temporal_mask", "spatial_states", "spatial_mask", "output"))
    for att in getattr(decoder, "attentions", None) or []:
        take(att, ("attention_states", "attention_mask", "hidden_features"))
    take(decoder, ("initial_state",))
    return found

FeedDict = Dict[Any, Any]
NextExecute = Tuple[Union[Dict, List], List[FeedDict]]
MP = TypeVar("MP", bound=GenericModelPart)
OutputSeries = Union[List, np.ndarray]


def _after_fill(name):
    """A dict method that first makes sure the values have arrived."""
    plain = getattr(dict, name)

    def method(self, *args, **kwargs):
        self._fill()          # pylint: disable=protected-access
        return plain(self, *args, **kwargs)
    method.__name__ = name
    return method


class LazyLosses(dict):
    """``ExecutionResult.losses`` of a training step whose scalars are still being copied to the host
    (runtime.HostPending): a dict that fills itself in on first access."""

    def __init__(self, names, pending) -> None:
        # The keys are there from the start (values NaN until read): C-level consumers of dict subclasses -- json's
        # encoder -- look at the raw size first and take the mapping protocol, i.e. the methods below, only for a
        # non-empty dict; anything that reads the raw storage directly sees NaN rather than nothing.
        self._names, self._pending = list(names), pending
        super().__init__((name, float("nan")) for name in self._names)

    def _fill(self) -> None:
        pending, self._pending = self._pending, None
        if pending is not None:
            values = [float(x) for x in pending.get()]
            # one value more than there are names: the session's device error word of that step (Session.error_word).
            # Set = a GRU time loop of the step gave up and its update was skipped on the device: the session runs the
            # step (and those enqueued since) again on the per-step path and these losses become that run's
            if len(values) > len(self._names) and values[len(self._names)] != 0.0:
                sess = getattr(pending, "session", None)
                if sess is None:
                    raise RuntimeError("the training step that produced these losses ran a GRU time loop that gave up "
                                       "waiting for a hand-off between workgroups: its results are garbage")
                sess.recover_training(pending=pending)
                values = [float(x) for x in pending.get()]
                if values[len(self._names)] != 0.0:
                    sess.raise_device_error()
            dict.update(self, zip(self._names, values))

    __getitem__ = _after_fill("__getitem__")
    __iter__ = _after_fill("__iter__")
    __len__ = _after_fill("__len__")
    __contains__ = _after_fill("__contains__")
    __repr__ = _after_fill("__repr__")
    __eq__ = _after_fill("__eq__")
    __ne__ = _after_fill("__ne__")
    __reversed__ = _after_fill("__reversed__")
    __or__ = _after_fill("__or__")
    __ror__ = _after_fill("__ror__")
    keys = _after_fill("keys")
    values = _after_fill("values")
    items = _after_fill("items")
    get = _after_fill("get")
    copy = _after_fill("copy")
    __hash__ = None

    def __reduce__(self):
        self._fill()
        return (dict, (dict(self),))


class ExecutionResult(NamedTuple):
    """base_runner.py:21-39."""
    outputs: Dict[str, OutputSeries]
    losses: Dict[str, float]
    size: int
    summaries: List[Any]


class GraphExecutor(GenericModelPart):
    class Executable:
        def __init__(self, executor: "GraphExecutor", compute_losses: bool, summaries: bool,
                     num_sessions: int) -> None:
            self._executor = executor
            self.compute_losses = compute_losses
            self.summaries = summaries
            self.num_sessions = num_sessions
            self._result: Optional[ExecutionResult] = None

        def set_result(self, outputs: Dict[str, OutputSeries], losses: Dict[str, float], size: int,
                       summaries: List[Any]) -> None:
            self._result = ExecutionResult(outputs, losses, size, summaries)

        @property
        def result(self) -> Optional[ExecutionResult]:
            return self._result

        @property
        def executor(self):
            return self._executor

        def next_to_execute(self) -> NextExecute:
            return self.executor.fetches, []

        def collect_results(self, results: List[Dict]) -> None:
            raise NotImplementedError

    def __init__(self, dependencies: Set[GenericModelPart]) -> None:
        check_constructor_chain(self)                       # runners / trainers: check_argument_types()
        self._dependencies = dependencies
        self._feedables, self._parameterizeds = self.get_dependencies()

    def get_executable(self, compute_losses: bool, summaries: bool, num_sessions: int):
        return self.Executable(self, compute_losses, summaries, num_sessions)

    @property
    def fetches(self) -> Dict[str, Any]:
        raise NotImplementedError()

    @property
    def dependencies(self) -> List[str]:
        return ["_dependencies"]

    @property
    def feedables(self) -> Set[Feedable]:
        return self._feedables

    @property
    def parameterizeds(self) -> Set[Parameterized]:
        return self._parameterizeds


class BaseRunner(GraphExecutor, Generic[MP]):
    class Executable(GraphExecutor.Executable):
        def next_to_execute(self) -> NextExecute:
            fetches = dict(self.executor.fetches)
            if not self.compute_losses:
                for loss in self.executor.loss_names:
                    fetches[loss] = 0.0
            return fetches, []

        def set_runner_result(self, outputs: OutputSeries, losses: List[float], size: int = None,
                              summaries: List[Any] = None) -> None:
            if summaries is None:
                summaries = []
            if size is None:
                size = len(outputs)
            loss_names = ["{}/{}".format(self.executor.output_series, loss)
                          for loss in self.executor.loss_names]
            self.set_result({self.executor.output_series: outputs}, dict(zip(loss_names, losses)), size,
                            summaries)

    def __init__(self, output_series: str, decoder: MP) -> None:
        GraphExecutor.__init__(self, {decoder})
        self.output_series = output_series
        self.decoder = decoder

    @property
    def decoder_data_id(self) -> Optional[str]:
        return getattr(self.decoder, "data_id", None)

    @property
    def loss_names(self) -> List[str]:
        raise NotImplementedError()
