"""Minimal dataset layer (subset of neuralmonkey/dataset.py needed by the
attention-decoder path): named series of token lists, plain-text loading,
fixed-size and length-bucketed batching.  Host-side only."""
import glob
import os
import random
import warnings
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple, Union


class BatchingScheme:
    """dataset.py:55-97."""

    def __init__(self, batch_size: int = None, drop_remainder: bool = False,
                 bucket_boundaries: List[int] = None, bucket_batch_sizes: List[int] = None,
                 ignore_series: List[str] = None) -> None:
        self.batch_size = batch_size
        self.drop_remainder = drop_remainder
        self.bucket_boundaries = bucket_boundaries
        self.bucket_batch_sizes = bucket_batch_sizes
        self.ignore_series = ignore_series or []
        if (self.batch_size is None) == (self.bucket_boundaries is None):
            raise ValueError("You must specify either batch_size or bucket_boundaries, not both")
        if self.bucket_boundaries is not None:
            if self.bucket_batch_sizes is None:
                raise ValueError("You must specify bucket_batch_sizes")
            if len(self.bucket_batch_sizes) != len(self.bucket_boundaries) + 1:
                raise ValueError("There should be N+1 batch sizes for N bucket boundaries")


def plain_text_reader(files: List[str], encoding: str = "utf-8") -> Iterator[List[str]]:
    for path in files:
        with open(path, encoding=encoding) as handle:
            for line in handle:
                yield line.strip().split()


class Dataset:
    """Named series of equal length; ``batches()`` yields sub-``Dataset``s (dataset.py:335-640).

    The series come as lists (this package's own callers) or, as in the reference, as FACTORIES ``() -> iterator``.
    Without ``buffer_size`` everything is read once and kept; with ``buffer_size=(low, high)`` the dataset is LAZY:
    nothing is read until somebody iterates, ``len()`` is refused, and ``batches`` keeps at most ``high`` rows in a
    buffer that it tops up whenever fewer than ``low`` are left.  ``shuffled`` shuffles that buffer -- all rows of an
    eager dataset once per pass, the buffered rows of a lazy one at every top-up -- with the ``random`` module, list
    lengths and call order as in the reference, so that a seeded run sees the reference's batches."""

    def __init__(self, name: str, iterators: Dict[str, Any] = None, batching: BatchingScheme = None,
                 outputs: Dict[str, Tuple[str, Any]] = None, buffer_size: Tuple[int, int] = None,
                 shuffled: bool = False, series: Dict[str, List[Any]] = None) -> None:
        self.name = name
        self.batching = batching
        self.outputs = outputs or {}
        self.shuffled = shuffled
        given = iterators if iterators is not None else (series or {})
        self.lazy = buffer_size is not None
        self._lists: Optional[Dict[str, List[Any]]] = None
        if self.lazy:
            self.buffer_min_size, self.buffer_size = buffer_size
            self.iterators = {k: (v if callable(v) else (lambda items=v: iter(items))) for k, v in given.items()}
            self._length = None
            return
        self._lists = {k: list(v() if callable(v) else v) for k, v in given.items()}
        length_dict = {k: len(v) for k, v in self._lists.items()}
        lengths = set(length_dict.values())
        if len(lengths) > 1:                  # dataset.py:398-405, the reference's text
            raise ValueError("Lengths of data series do not match: {}".format(str(length_dict)))
        self._length = lengths.pop() if lengths else 0
        self.iterators = {k: (lambda key=k: iter(self._lists[key])) for k in self._lists}

    @property
    def _series(self) -> Dict[str, List[Any]]:
        """name -> list of an eager dataset (a lazy one is read through ``get_series`` / ``batches`` only)."""
        if self._lists is None:
            raise NotImplementedError("a lazy dataset holds no lists")
        return self._lists

    def __len__(self) -> int:
        if self.lazy:
            raise NotImplementedError("Querying the len of a lazy dataset.")
        return self._length

    def __contains__(self, name: str) -> bool:
        return name in self.iterators

    @property
    def series(self) -> List[str]:
        return sorted(self.iterators)

    def get_series(self, name: str) -> Iterator:
        return self.iterators[name]()             # (KeyError for a series the dataset does not have)

    def maybe_get_series(self, name: str) -> Optional[Iterator]:
        return self.iterators[name]() if name in self.iterators else None

    def subset(self, start: int, length: int) -> "Dataset":
        """Rows ``start .. start + length``; laziness, buffer sizes and shuffling are inherited, output files get the
        start offset appended (dataset.py:586-619)."""
        import itertools
        outputs = {key: ("{}.{:010}".format(path, start), writer) for key, (path, writer) in self.outputs.items()}
        name = "{}.{}.{}".format(self.name, start, length)
        if not self.lazy:
            return Dataset(name, {k: v[start:start + length] for k, v in self._lists.items()}, self.batching,
                           outputs, None, self.shuffled)
        slices = {k: (lambda key=k: itertools.islice(self.get_series(key), start, start + length))
                  for k in self.iterators}
        return Dataset(name, slices, self.batching, outputs, (self.buffer_min_size, self.buffer_size), self.shuffled)

    def _rows(self, idx: List[int], number: int = None) -> "Dataset":
        name = self.name if number is None else "{}.batch.{}".format(self.name, number)
        return Dataset(name, {k: [v[i] for i in idx] for k, v in self._lists.items()}, self.batching, self.outputs)

    @staticmethod
    def _longest(items) -> int:
        # dataset.py:525: max(len(row[key])) over the series -- token lists, pre-indexed id arrays
        # (input_pipeline.preindex) and feature arrays alike
        return max((len(item) for item in items if hasattr(item, "__len__") and not isinstance(item, (str, bytes))),
                   default=0)

    @staticmethod
    def _bucket_of(longest: int, bounds: List[int]) -> int:
        # dataset.py:524-535: the TIGHTEST boundary that fits (the boundaries need not be sorted); none fits: the
        # last bucket (the reference's ``buckets[-1]``)
        best = -1
        for cand, limit in enumerate(bounds):
            if longest <= limit and (best == -1 or limit < bounds[best]):
                best = cand
        return best if best != -1 else len(bounds)

    def batches(self, batching: BatchingScheme = None) -> Iterator["Dataset"]:
        """dataset.py:467-579.  (``scheme.ignore_series`` is accepted and, as in the reference at this commit, not
        consulted: dataset.py:521 "TODO: use only specific series to determine the bucket number" -- the longest of
        ALL series decides; the reference-executed fixture "dataset_batching" under tests/golden, scheme
        "buckets_ignore".)"""
        scheme = batching or self.batching
        if scheme is None:
            raise ValueError("No batching scheme for dataset '{}'".format(self.name))
        if self.lazy:
            yield from self._lazy_batches(scheme)
            return
        order = list(range(self._length))
        if self.shuffled:
            random.shuffle(order)                 # the same permutation ``random.shuffle`` gives the reference's rows
        if scheme.bucket_boundaries is None:
            sizes, keys = [scheme.batch_size], None
        else:
            sizes, keys = scheme.bucket_batch_sizes, list(self._lists)
        buckets: List[List[int]] = [[] for _ in sizes]
        number = 0
        for i in order:
            b = 0 if keys is None else self._bucket_of(self._longest(self._lists[k][i] for k in keys),
                                                       scheme.bucket_boundaries)
            buckets[b].append(i)
            if len(buckets[b]) >= sizes[b]:
                yield self._rows(buckets[b], number)
                number += 1
                buckets[b] = []
        if not scheme.drop_remainder:
            for bucket in buckets:
                if bucket:
                    yield self._rows(bucket, number)
                    number += 1

    def _lazy_batches(self, scheme: BatchingScheme) -> Iterator["Dataset"]:
        """The buffered pass of a lazy dataset: rows are drawn from the series in step; the buffer starts with
        ``buffer_size`` rows and is topped up to that size whenever it holds fewer than ``buffer_min_size`` after a
        row has been taken (shuffled again then, if the dataset shuffles)."""
        import collections
        import itertools
        largest = scheme.batch_size if scheme.batch_size is not None else max(scheme.bucket_batch_sizes)
        if self.buffer_min_size < largest:
            warnings.warn("Minimum buffer size ({}) lower than batch size ({}). It is recommended to use large buffer "
                          "size.".format(self.buffer_min_size, largest))
        keys = list(self.iterators)
        rows = zip(*[self.iterators[k]() for k in keys])
        pending = list(itertools.islice(rows, self.buffer_size))
        if self.shuffled:
            random.shuffle(pending)
        buffer = collections.deque(pending)
        sizes = [scheme.batch_size] if scheme.bucket_boundaries is None else scheme.bucket_batch_sizes
        buckets: List[List[tuple]] = [[] for _ in sizes]
        number = 0

        def batch_of(taken: List[tuple]) -> "Dataset":
            columns = {k: [row[j] for row in taken] for j, k in enumerate(keys)}
            return Dataset("{}.batch.{}".format(self.name, number), columns, self.batching)
        while buffer:
            row = buffer.popleft()
            b = 0 if scheme.bucket_boundaries is None else self._bucket_of(self._longest(row),
                                                                           scheme.bucket_boundaries)
            buckets[b].append(row)
            if len(buckets[b]) >= sizes[b]:
                yield batch_of(buckets[b])
                number += 1
                buckets[b] = []
            if len(buffer) < self.buffer_min_size:
                buffer.extend(itertools.islice(rows, self.buffer_size - len(buffer)))
                if self.shuffled:
                    again = list(buffer)
                    random.shuffle(again)
                    buffer = collections.deque(again)
        if not scheme.drop_remainder:
            for bucket in buckets:
                if bucket:
                    yield batch_of(bucket)
                    number += 1


def _expand(patterns: Union[str, List[str]]) -> List[str]:
    if isinstance(patterns, str):
        patterns = [patterns]
    paths: List[str] = []
    for pat in patterns:
        matched = sorted(glob.glob(pat))
        if not matched:
            raise FileNotFoundError("Pattern did not match any files: {}".format(pat))
        # absolute: a lazy dataset opens its files when it is iterated, wherever the process stands by then
        paths.extend(os.path.abspath(path) for path in matched)
    return paths


# [main] batch_size of the configuration that is being built right now (innermost last): what the reference reads
# through ``Experiment.get_current().config.args.batch_size`` (dataset.py:237-246).  Pushed by config.Configuration.
_MAIN_BATCH_SIZE: List[Any] = []


def load(name: str, series: List[str], data: List[Any], batching: BatchingScheme = None,
         outputs: List[Tuple] = None, buffer_size: int = None, shuffled: bool = False) -> Dataset:
    """dataset.py:207-333.  A series is given by one of

    * a path / glob / list of them, or ``(files, reader)`` -- read from files (whitespace tokens by default);
    * ``(preprocessor, source_series)`` -- the preprocessor applied to every item of a series read from
      files (series-level preprocessors do not stack, as in the reference);
    * a callable -- a dataset-level preprocessor, called once with ``{series: () -> iterator}`` of all
      the series above and returning the items of the new series.

    Without ``buffer_size`` the series are read here, once; with it the dataset stays lazy (``Dataset``)."""
    def from_file(spec) -> bool:         # a ReaderDef of the reference: files, or (files, reader)
        if isinstance(spec, (str, list)):
            return True
        return isinstance(spec, tuple) and len(spec) == 2 and isinstance(spec[0], (str, list)) and callable(spec[1])
    # a dataset section without ``batching`` batches by [main] batch_size (dataset.py:237-246); outside a
    # configuration the scheme stays open and ``batches`` asks for one
    if batching is None and _MAIN_BATCH_SIZE:
        if _MAIN_BATCH_SIZE[-1] is None:
            raise ValueError("Argument main.batch_size is not specified, cannot use default batching scheme.")
        batching = BatchingScheme(batch_size=_MAIN_BATCH_SIZE[-1])
    # the checks of dataset.py:246-266, in the reference's order and with its texts
    if not series:
        raise ValueError("No dataset series specified.")
    if not any(from_file(spec) for spec in data):
        raise ValueError("At least one data series should be from a file")
    if len(series) != len(data):
        raise ValueError("The 'series' and 'data' lists should have the same number of elements: "
                         "{} vs {}.".format(len(series), len(data)))
    if len(series) != len(set(series)):
        raise ValueError("There are duplicate series.")
    if outputs is not None and len({spec[0] for spec in outputs}) != len(outputs):
        raise ValueError("Multiple outputs for a single series")
    factories: Dict[str, Callable[[], Iterator]] = {}
    series_level: Dict[str, Tuple[Callable, str]] = {}
    dataset_level: Dict[str, Callable] = {}

    def reading(reader: Callable, files: List[str]) -> Callable[[], Iterator]:
        return lambda: reader(files)

    def mapped(preprocessor: Callable, source: str) -> Callable[[], Iterator]:
        return lambda: (preprocessor(item) for item in factories[source]())
    for sid, spec in zip(series, data):
        if isinstance(spec, (str, list)):
            factories[sid] = reading(plain_text_reader, _expand(spec))
        elif isinstance(spec, tuple) and len(spec) == 2 and isinstance(spec[0], (str, list)) and callable(spec[1]):
            factories[sid] = reading(spec[1], _expand(spec[0]))
        elif isinstance(spec, tuple) and len(spec) == 2 and callable(spec[0]) and isinstance(spec[1], str):
            series_level[sid] = spec
        elif callable(spec):
            dataset_level[sid] = spec
        else:
            raise TypeError("series '{}': {!r} is neither files, (files, reader), (preprocessor, series) "
                            "nor a dataset-level preprocessor".format(sid, spec))
    from_files = set(factories)
    for sid, (preprocessor, source) in series_level.items():
        if source not in from_files:
            # (dataset.py:309-312 never fills the two placeholders in: the text below IS the reference's)
            raise ValueError("Source series for series-level preprocessor nonexistent: "
                             "Preprocessed series '{}', source series '{}'")
    for sid, (preprocessor, source) in series_level.items():
        factories[sid] = mapped(preprocessor, source)
    for sid, func in dataset_level.items():
        # called with the factories of ALL series, its own and the other dataset-level ones included (they are
        # opened only when iterated: dataset.py:291-294,318-319)
        factories[sid] = (lambda call=func: call(factories))
    out_specs = {}
    for spec in outputs or []:
        out_specs[spec[0]] = (spec[1], spec[2] if len(spec) > 2 else None)
    # ``buffer_size`` makes the dataset lazy with a buffer topped up at half that size (dataset.py:327-331)
    buffering = None if buffer_size is None else (buffer_size // 2, buffer_size)
    return Dataset(name, factories, batching, out_specs, buffering, shuffled)


def from_ids(name: str, series: Dict[str, List[List[str]]], batching: BatchingScheme = None) -> Dataset:
    return Dataset(name, series, batching)
