"""Text readers of the host input pipeline (mirror of neuralmonkey/readers/plain_text_reader.py:
``UtfPlainTextReader``, ``T2TReader``, ``tokenized_text_reader``, ``t2t_tokenized_text_reader``,
``column_separated_reader``, ``csv_reader``, ``tsv_reader``, ``string_reader``).

A reader is a callable ``files -> iterable of token lists``; ``dataset.load`` accepts one per series
(``s_<name> = (path, reader)`` in the INI).  Paths ending in ``.gz`` are read through gzip.
"""
import csv
import gzip
import itertools
import sys
import unicodedata
import warnings
from functools import lru_cache
from typing import Callable, Iterable, Iterator, List


# pylint: disable=invalid-name
PlainTextFileReader = Callable[[List[str]], Iterable[List[str]]]
# pylint: enable=invalid-name

csv.field_size_limit(sys.maxsize)


def _lines(path: str, encoding: str) -> Iterator[str]:
    if path.endswith(".gz"):
        with gzip.open(path, "rt", encoding="utf-8") as handle:        # :28-31: gzip input is always UTF-8
            yield from handle
    else:
        with open(path, encoding=encoding) as handle:
            yield from handle


def string_reader(encoding: str = "utf-8") -> Callable[[List[str]], Iterable[str]]:
    """Lines of the files, one after another, line ends included."""
    def reader(files: List[str]) -> Iterator[str]:
        return itertools.chain.from_iterable(_lines(path, encoding) for path in files)
    return reader


def tokenized_text_reader(encoding: str = "utf-8") -> PlainTextFileReader:
    """Whitespace-separated tokens."""
    def reader(files: List[str]) -> Iterator[List[str]]:
        return (line.split() for line in string_reader(encoding)(files))
    return reader


@lru_cache(maxsize=None)
def _is_alnum(char: str) -> bool:
    """Unicode letters and numbers (categories L* and N*), the tensor2tensor tokenizer's alphabet."""
    return unicodedata.category(char)[0] in "LN"


def t2t_tokenize(line: str) -> List[str]:
    """Runs of alphanumeric and of other characters alternate; a run that is exactly one space is a
    separator and is dropped (plain_text_reader.py:49-86).  Odd whitespace survives as tokens, so the
    text can be put back together."""
    line = line.strip()
    runs = ["".join(chars) for _, chars in itertools.groupby(line, key=_is_alnum)]
    if not runs:
        return [""]                                                     # the reference's final token of an empty line
    last = len(runs) - 1
    return [run for i, run in enumerate(runs) if run != " " or i in (0, last)]


def t2t_tokenized_text_reader(encoding: str = "utf-8") -> PlainTextFileReader:
    def reader(files: List[str]) -> Iterator[List[str]]:
        return (t2t_tokenize(line) for line in string_reader(encoding)(files))
    return reader


def column_separated_reader(column: int, delimiter: str = "\t", quotechar: str = None,
                            encoding: str = "utf-8") -> PlainTextFileReader:
    """Tokens of the ``column``-th field (counted from 1) of delimiter-separated lines; a line with too
    few fields gives an empty sentence and a warning (:89-124)."""
    dialect = dict(delimiter=delimiter, skipinitialspace=True)
    if quotechar is None:
        dialect["quoting"] = csv.QUOTE_NONE
    else:
        dialect["quotechar"] = quotechar

    def reader(files: List[str]) -> Iterator[List[str]]:
        expected = None
        for line in string_reader(encoding)(files):
            fields = next(csv.reader([line.strip()], **dialect), [])
            if expected is None:
                expected = len(fields)
            elif len(fields) != expected:
                warnings.warn("A mismatch in number of columns. Expected {} got {}".format(expected, len(fields)))
            if len(fields) < column:
                warnings.warn("There is a missing column number {} in the dataset.".format(column))
                yield []
            else:
                yield fields[column - 1].split()
    return reader


def csv_reader(column: int) -> PlainTextFileReader:
    return column_separated_reader(column, delimiter=",", quotechar='"')


def tsv_reader(column: int) -> PlainTextFileReader:
    return column_separated_reader(column, delimiter="\t", quotechar=None)


# pylint: disable=invalid-name
UtfPlainTextReader = tokenized_text_reader()
T2TReader = t2t_tokenized_text_reader()
# pylint: enable=invalid-name
