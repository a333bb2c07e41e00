"""Readers of pre-extracted feature maps (the series readers of
neuralmonkey/readers/numpy_reader.py): a dataset series given as ``(files, reader)`` yields one
NumPy array per example.  Host-side only; the arrays reach the device through
``SpatialFiller.feed_dict`` / ``Session.to_device``."""
import os
from typing import Callable, Iterator, List, Sequence

import numpy as np


def single_tensor(files: List[str]) -> np.ndarray:
    """All examples stacked in one ``.npy`` tensor -- or in several, joined along the example axis
    (numpy_reader.py:9-15)."""
    parts = [np.load(path) for path in files]
    return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0)


def from_file_list(prefix: str, shape: Sequence[int], suffix: str = "",
                   default_tensor_name: str = "arr_0") -> Callable[[List[str]], Iterator[np.ndarray]]:
    """A reader over list files: every line names one ``.npz`` archive relative to ``prefix``
    (+ ``suffix``) holding the example's map under ``default_tensor_name`` (numpy_reader.py:18-52).
    A map whose shape is not ``shape`` is an error."""
    want = [int(d) for d in shape]

    def read(list_files: List[str]) -> Iterator[np.ndarray]:
        for list_file in list_files:
            with open(list_file, encoding="utf-8") as names:
                for name in names:
                    archive = os.path.join(prefix, name.rstrip()) + suffix
                    with np.load(archive) as contents:
                        feature_map = contents[default_tensor_name]
                    if list(feature_map.shape) != want:
                        raise ValueError("Shapes do not match: expected {}, found {}"
                                         .format(want, list(feature_map.shape)))
                    yield feature_map
    return read
