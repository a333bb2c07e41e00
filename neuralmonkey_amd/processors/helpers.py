"""Sentence-level odds and ends of the host pipeline, under the names INI files use
(neuralmonkey/processors/helpers.py): ``pipeline``, ``untruecase``, the character-level pair
``preprocess_char_based`` / ``postprocess_char_based`` and ``preprocess_add_noise``.

Preprocessors map ONE sentence (a token list) to a token list and are used as series-level
preprocessors in ``dataset.load``; postprocessors map a BATCH of decoded sentences and are handed to
runners (``GreedyRunner(postprocess=...)``).
"""
import random
from functools import reduce
from typing import Any, Callable, Iterator, List

Sentence = List[str]


def pipeline(processors: List[Callable]) -> Callable:
    """``pipeline([f, g, h])(x) == h(g(f(x)))``."""
    return lambda data: reduce(lambda value, step: step(value), processors, data)


def untruecase(sentences: List[Sentence]) -> Iterator[Sentence]:
    """Capitalise the first token of every non-empty sentence (lazy, like the reference's generator).
    ``str.capitalize`` also lower-cases the rest of that token -- kept, it is what the reference does."""
    return ([head.capitalize()] + tail if head is not None else []
            for head, tail in ((s[0] if s else None, list(s[1:])) for s in sentences))


def preprocess_char_based(sentence: Sentence) -> Sentence:
    """Tokens -> characters, the token boundaries as single-space tokens."""
    return [char for char in " ".join(sentence)]


def postprocess_char_based(sentences: List[Sentence]) -> List[Sentence]:
    """Characters -> tokens: glue everything, cut at the spaces (two spaces in a row give an empty token,
    as ``str.split(" ")`` does in the reference)."""
    return ["".join(chars).split(" ") for chars in sentences]


def preprocess_add_noise(sentence: Sentence) -> Sentence:
    """len // 2 times: overwrite a random position with its right neighbour.  (The reference means to
    swap the two but assigns the left one first, so the pair ends up as two copies of the right token;
    reproduced as is.)  Uses the ``random`` module's global state, like the reference."""
    noisy = list(sentence)
    last = len(noisy) - 2
    for _ in range(len(noisy) // 2 if last >= 0 else 0):
        pos = random.randint(0, last)
        noisy[pos] = noisy[pos + 1]
    return noisy
