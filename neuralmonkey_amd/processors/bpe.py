"""Byte-pair-encoding pre-/post-processors of the translation pipeline (mirror of
neuralmonkey/processors/bpe.py:10-60; the segmentation itself is Sennrich et al. 2016, which the
reference takes from its vendored ``lib/subword_nmt/apply_bpe.py``).

Host-side string work, run once per dataset by the input pipeline (``input_pipeline.preindex``
consumes the segmented tokens) and once per decoded batch; nothing here is on the device path.

Segmentation of one word: start from its characters plus the end-of-word symbol ``</w>``; while
any adjacent pair of symbols is in the merge table, merge every occurrence (left to right) of the
pair with the LOWEST merge rank; finally drop ``</w>``.  Duplicate merge lines keep their first
rank.  Golden vectors produced by the reference's own code: tests/golden/bpe_golden.json.
"""
import re
from typing import Dict, List, Tuple

END_OF_WORD = "</w>"


class BPEPreprocessor:
    def __init__(self, merge_file: str, separator: str = "@@", encoding: str = "utf-8") -> None:
        with open(merge_file, "r", encoding=encoding) as handle:
            self.ranks = read_merges(handle)
        self.separator = separator
        self._cache: Dict[str, Tuple[str, ...]] = {}

    def __call__(self, sentence: List[str]) -> List[str]:
        output: List[str] = []
        for word in sentence:
            if not word:                              # the reference passes empty tokens through (bpe.py:33-36)
                output.append(word)
                continue
            pieces = self._cache.get(word)
            if pieces is None:
                pieces = self._cache[word] = segment_word(word, self.ranks)
            output.extend(piece + self.separator for piece in pieces[:-1])
            output.append(pieces[-1])
        return output


def read_merges(lines) -> Dict[Tuple[str, ...], int]:
    ranks: Dict[Tuple[str, ...], int] = {}
    for rank, line in enumerate(lines):
        pair = tuple(line.split())
        ranks.setdefault(pair, rank)                  # first instance of a duplicate wins
    return ranks


def segment_word(word: str, ranks: Dict[Tuple[str, ...], int]) -> Tuple[str, ...]:
    symbols = list(word) + [END_OF_WORD]
    while len(symbols) > 1:
        best_rank, best = None, None
        for left, right in zip(symbols, symbols[1:]):
            rank = ranks.get((left, right))
            if rank is not None and (best_rank is None or rank < best_rank):
                best_rank, best = rank, (left, right)
        if best is None:
            break
        merged, i = [], 0
        while i < len(symbols):
            if i + 1 < len(symbols) and symbols[i] == best[0] and symbols[i + 1] == best[1]:
                merged.append(best[0] + best[1])
                i += 2
            else:
                merged.append(symbols[i])
                i += 1
        symbols = merged
    if symbols[-1] == END_OF_WORD:
        symbols = symbols[:-1]
    elif symbols[-1].endswith(END_OF_WORD):
        symbols[-1] = symbols[-1].replace(END_OF_WORD, "")
    return tuple(symbols)


class BPEPostprocessor:
    """bpe.py:46-60: glue ``piece@@ piece`` back together."""

    def __init__(self, separator: str = "@@") -> None:
        self.pattern = re.compile(re.escape(separator) + r" ")

    def __call__(self, decoded_sentences: List[List[str]]) -> List[List[str]]:
        return [self.decode(s) for s in decoded_sentences]

    def decode(self, sentence: List[str]) -> List[str]:
        return self.pattern.sub("", " ".join(sentence)).split(" ")


bpe_postprocess = BPEPostprocessor()       # the instance the reference's configs name (examples/translation.ini)
