"""Vectors written as whitespace-separated numbers, one per line (mirror of
neuralmonkey/readers/string_vector_reader.py: ``get_string_vector_reader``, ``FloatVectorReader``,
``IntVectorReader``).  Blank lines are skipped but counted, so the line number of the error message
is the one an editor shows.  gzip files are decoded as text (the reference applies ``str()`` to the
raw bytes there, which cannot be parsed)."""
import gzip
from typing import Iterator, List, Type

import numpy as np


def get_string_vector_reader(dtype: Type = np.float32, columns: int = None):
    def reader(files: List[str]) -> Iterator[np.ndarray]:
        for path in files:
            opener = (lambda p: gzip.open(p, "rt")) if path.endswith(".gz") else open
            with opener(path) as handle:
                for lineno, line in enumerate(handle, start=1):
                    numbers = line.split()
                    if not numbers:
                        continue
                    if columns is not None and len(numbers) != columns:
                        raise ValueError("Wrong number of columns ({}) on line {}, file {}"
                                         .format(len(numbers), lineno, path))
                    yield np.array(numbers, dtype=dtype)
    return reader


# pylint: disable=invalid-name
FloatVectorReader = get_string_vector_reader(np.float32)
IntVectorReader = get_string_vector_reader(np.int32)
# pylint: enable=invalid-name
