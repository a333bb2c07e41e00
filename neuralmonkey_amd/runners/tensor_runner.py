"""TensorRunner / RepresentationRunner (mirror of neuralmonkey/runners/tensor_runner.py:13-204):
fetch named ``@tensor`` attributes of model parts (encoder states, attention histories ...) for a
dataset and hand them out example by example.

``tensors_by_name`` addresses nodes of the TensorFlow graph by their graph name; there is no graph
here, so such names are reported as missing exactly as the reference reports names that are not in
its graph (:120-128) and the remaining tensors are still delivered."""
import warnings
from typing import Any, Dict, List

import numpy as np

from .base_runner import BaseRunner


class TensorRunner(BaseRunner):
    class Executable(BaseRunner.Executable):
        def collect_results(self, results: List[Dict]) -> None:
            if len(results) > 1 and self.executor.select_session is None:
                per_session = [self._examples(res) for res in results]
                batched = list(zip(*per_session))
            else:
                # tensor_runner.py:33-34: with ``select_session`` set the reference hands out the FIRST session's
                # tensors, whatever the number says (pinned by the reference-executed fixture ``tensor_runner``)
                batched = self._examples(results[0])
            self.set_runner_result(outputs=batched, losses=[])

        def _examples(self, sess_results: Dict) -> List:
            """Move each tensor's batch axis to the front and split it into per-example entries."""
            by_example: Dict[str, np.ndarray] = {}
            for name, value in sess_results.items():
                value = np.asarray(value)
                by_example[name] = np.moveaxis(value, self.executor.batch_ids[name], 0)
            rows = [dict(zip(by_example, column)) for column in zip(*by_example.values())]
            if self.executor.single_tensor:
                rows = [next(iter(row.values())) for row in rows]
            return rows

    # pylint: disable=too-many-arguments
    def __init__(self, output_series: str, modelparts: List[Any], tensors: List[str], batch_dims: List[int],
                 tensors_by_name: List[str], batch_dims_by_name: List[int], select_session: int = None,
                 single_tensor: bool = False) -> None:
        if not modelparts:
            raise ValueError("At least one model part is expected")
        super().__init__(output_series, modelparts[0])
        if len(modelparts) != len(tensors):
            raise ValueError("TensorRunner: 'modelparts' and 'tensors' lists must have the same length")
        total = len(tensors_by_name) + len(tensors)
        if single_tensor and total > 1:
            raise ValueError("single_tensor is True, but {} tensors were given".format(total))
        for part in modelparts[1:]:                       # every part must be fed and have its variables
            feeds, params = part.get_dependencies()
            self._feedables |= feeds
            self._parameterizeds |= params
        self._names = tensors_by_name
        self._modelparts = modelparts
        self._tensors = tensors
        self._batch_dims_name = batch_dims_by_name
        self.batch_dims = batch_dims
        self.select_session = select_session
        self.single_tensor = single_tensor
        self.batch_ids: Dict[str, int] = {}

    @property
    def fetches(self) -> Dict[str, Any]:
        fetches: Dict[str, Any] = {}
        for name in self._names:
            warnings.warn("The tensor of name '{}' is not present in the graph.".format(name))
        for part, tname, bid in zip(self._modelparts, self._tensors, self.batch_dims):
            if not hasattr(part, tname):
                raise ValueError("Model part {} does not have a tensor called {}.".format(part, tname))
            key = "{}/{}".format(getattr(part, "name", part), tname)
            fetches[key] = getattr(part, tname)
            self.batch_ids[key] = bid
        return fetches

    @property
    def loss_names(self) -> List[str]:
        return []


class RepresentationRunner(TensorRunner):
    """One attribute of one encoder (default ``output``), one vector per example (:166-204)."""

    def __init__(self, output_series: str, encoder: Any, attribute: str = "output",
                 select_session: int = None) -> None:
        if attribute not in dir(encoder):
            warnings.warn("The encoder '{}' seems not to have the specified attribute '{}'".format(encoder, attribute))
        TensorRunner.__init__(self, output_series, modelparts=[encoder], tensors=[attribute], batch_dims=[0],
                              tensors_by_name=[], batch_dims_by_name=[], select_session=select_session,
                              single_tensor=True)
