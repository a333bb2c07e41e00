"""Readers of pre-extracted feature maps (mirror of neuralmonkey/readers/numpy_reader.py).
Host-side only."""
import os
from typing import Callable, Iterable, List

import numpy as np


def single_tensor(files: List[str]) -> np.ndarray:
    """numpy_reader.py:9-15: one tensor, or several concatenated along axis 0."""
    if len(files) == 1:
        return np.load(files[0])
    return np.concatenate([np.load(f) for f in files], axis=0)


def from_file_list(prefix: str, shape: List[int], suffix: str = "",
                   default_tensor_name: str = "arr_0") -> Callable:
    """numpy_reader.py:18-52: every line of the list files names an .npz under ``prefix``."""
    def load(files: List[str]) -> Iterable[np.ndarray]:
        for list_file in files:
            with open(list_file, encoding="utf-8") as f_list:
                for line in f_list:
                    path = os.path.join(prefix, line.rstrip()) + suffix
                    with np.load(path) as npz:
                        arr = npz[default_tensor_name]
                        if list(arr.shape) != list(shape):
                            raise ValueError("Shapes do not match: expected {}, found {}"
                                             .format(shape, list(arr.shape)))
                        yield arr
    return load
