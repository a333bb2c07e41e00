"""Minimal dataset layer (subset of neuralmonkey/dataset.py needed by the
attention-decoder path): named series of token lists, plain-text loading,
fixed-size and length-bucketed batching.  Host-side only."""
import glob
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
    """Named series of equal length; ``batches()`` yields sub-``Dataset``s."""

    def __init__(self, name: str, series: Dict[str, List[Any]], batching: BatchingScheme = None,
                 outputs: Dict[str, Tuple[str, Any]] = None, shuffled: bool = False) -> None:
        self.name = name
        self._series = {k: list(v) for k, v in series.items()}
        length_dict = {k: len(v) for k, v in self._series.items()}
        lengths = set(length_dict.values())
        if len(lengths) > 1:                  # dataset.py:398-405, the reference's text
            raise ValueError("Lengths of data series do not match: {}".format(str(length_dict)))
        self._length = lengths.pop() if lengths else 0
        self.batching = batching
        self.outputs = outputs or {}
        self.shuffled = shuffled

    def __len__(self) -> int:
        return self._length

    def __contains__(self, name: str) -> bool:
        return name in self._series

    @property
    def series(self) -> List[str]:
        return list(self._series)

    def get_series(self, name: str) -> Iterator:
        if name not in self._series:
            raise KeyError("Series '{}' is not in the dataset".format(name))
        return iter(self._series[name])

    def maybe_get_series(self, name: str) -> Optional[Iterator]:
        return iter(self._series[name]) if name in self._series else None

    def subset(self, start: int, length: int) -> "Dataset":
        return Dataset("{}.{}.{}".format(self.name, start, length),
                       {k: v[start:start + length] for k, v in self._series.items()},
                       self.batching, self.outputs)

    def _rows(self, idx: List[int]) -> "Dataset":
        return Dataset(self.name, {k: [v[i] for i in idx] for k, v in self._series.items()},
                       self.batching, self.outputs)

    def batches(self, batching: BatchingScheme = None) -> Iterator["Dataset"]:
        scheme = batching or self.batching
        if scheme is None:
            raise ValueError("No batching scheme for dataset '{}'".format(self.name))
        if scheme.batch_size is not None:
            for start in range(0, self._length, scheme.batch_size):
                idx = list(range(start, min(self._length, start + scheme.batch_size)))
                if len(idx) < scheme.batch_size and scheme.drop_remainder:
                    break
                yield self._rows(idx)
            return
        bounds = scheme.bucket_boundaries
        sizes = scheme.bucket_batch_sizes
        buckets: List[List[int]] = [[] for _ in sizes]
        # (``scheme.ignore_series`` is accepted and, as in the reference at this commit, not consulted: dataset.py:521
        # "TODO: use only specific series to determine the bucket number" -- the longest of ALL series decides;
        # the reference-executed fixture "dataset_batching" under tests/golden, scheme "buckets_ignore")
        keys = list(self._series)
        for i in range(self._length):
            longest = 0
            for k in keys:
                item = self._series[k][i]
                # dataset.py:525: max(len(row[key])) over the series -- token lists, pre-indexed id arrays
                # (input_pipeline.preindex) and feature arrays alike
                if hasattr(item, "__len__") and not isinstance(item, (str, bytes)):
                    longest = max(longest, len(item))
            # dataset.py:524-535: the TIGHTEST boundary that fits (the boundaries need not be sorted); none fits: the
            # last bucket (the reference's ``buckets[-1]``)
            b = -1
            for cand, limit in enumerate(bounds):
                if longest <= limit and (b == -1 or limit < bounds[b]):
                    b = cand
            if b == -1:
                b = len(buckets) - 1
            buckets[b].append(i)
            if len(buckets[b]) == sizes[b]:
                yield self._rows(buckets[b])
                buckets[b] = []
        if not scheme.drop_remainder:
            for bucket in buckets:
                if bucket:
                    yield self._rows(bucket)


def _expand(patterns: Union[str, List[str]]) -> List[str]:
    if isinstance(patterns, str):
        patterns = [patterns]
    paths: List[str] = []
    for pat in patterns:
        matched = sorted(glob.glob(pat))
        if not matched:
            raise FileNotFoundError("Pattern did not match any files: {}".format(pat))
        paths.extend(matched)
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

    The series are materialised here (``buffer_size`` is accepted and ignored: the lazy refill of the
    reference bounds host memory, the batches are the same)."""
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
    loaded: Dict[str, List[Any]] = {}
    series_level: Dict[str, Tuple[Callable, str]] = {}
    dataset_level: Dict[str, Callable] = {}
    for sid, spec in zip(series, data):
        if isinstance(spec, (str, list)):
            loaded[sid] = list(plain_text_reader(_expand(spec)))
        elif isinstance(spec, tuple) and len(spec) == 2 and isinstance(spec[0], (str, list)) and callable(spec[1]):
            loaded[sid] = list(spec[1](_expand(spec[0])))
        elif isinstance(spec, tuple) and len(spec) == 2 and callable(spec[0]) and isinstance(spec[1], str):
            series_level[sid] = spec
        elif callable(spec):
            dataset_level[sid] = spec
        else:
            raise TypeError("series '{}': {!r} is neither files, (files, reader), (preprocessor, series) "
                            "nor a dataset-level preprocessor".format(sid, spec))
    from_files = set(loaded)
    for sid, (preprocessor, source) in series_level.items():
        if source not in from_files:
            # (dataset.py:309-312 never fills the two placeholders in: the text below IS the reference's)
            raise ValueError("Source series for series-level preprocessor nonexistent: "
                             "Preprocessed series '{}', source series '{}'")
        loaded[sid] = [preprocessor(item) for item in loaded[source]]
    if dataset_level:
        def _factory(items):
            return lambda: iter(items)
        iterators = {sid: _factory(items) for sid, items in loaded.items()}
        for sid, func in dataset_level.items():
            loaded[sid] = list(func(iterators))
    out_specs = {}
    for spec in outputs or []:
        out_specs[spec[0]] = (spec[1], spec[2] if len(spec) > 2 else None)
    return Dataset(name, loaded, batching, out_specs, shuffled)


def from_ids(name: str, series: Dict[str, List[List[str]]], batching: BatchingScheme = None) -> Dataset:
    return Dataset(name, series, batching)
