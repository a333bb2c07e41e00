"""Series readers for pre-extracted feature maps (the two readers of neuralmonkey/readers/numpy_reader.py): a
dataset series given as ``(files, reader)`` yields one NumPy array per example.  Host side only -- the arrays reach
the device through ``SpatialFiller.feed_dict`` / ``Session.to_device``."""
import os
from typing import Callable, Iterator, List, Sequence

import numpy as np


def single_tensor(files: List[str]) -> np.ndarray:
    """Every example in ONE ``.npy`` tensor (first axis = examples); several files are several runs of examples
    (numpy_reader.py:9-15)."""
    tensors = list(map(np.load, files))
    if len(tensors) > 1:
        return np.concatenate(tensors, axis=0)
    return tensors[0]


def _archive_paths(list_files: List[str], prefix: str, suffix: str) -> Iterator[str]:
    for list_file in list_files:
        with open(list_file, encoding="utf-8") as names:
            yield from (os.path.join(prefix, name.rstrip()) + suffix for name in names)


def from_file_list(prefix: str, shape: Sequence[int], suffix: str = "",
                   default_tensor_name: str = "arr_0") -> Callable[[List[str]], Iterator[np.ndarray]]:
    """The reader of "list files" (numpy_reader.py:18-52): every line of such a file names an ``.npz`` archive
    (``prefix``/line + ``suffix``) whose entry ``default_tensor_name`` is one example's map; a map of another
    shape than ``shape`` stops the reader with the reference's ValueError."""
    expected = [int(extent) for extent in shape]

    def reader(list_files: List[str]) -> Iterator[np.ndarray]:
        for path in _archive_paths(list_files, prefix, suffix):
            with np.load(path) as archive:
                feature_map = archive[default_tensor_name]
            found = list(feature_map.shape)
            if found != expected:
                raise ValueError("Shapes do not match: expected {}, found {}".format(expected, found))
            yield feature_map
    return reader
