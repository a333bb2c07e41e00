"""Small sentence processors (mirror of neuralmonkey/processors/helpers.py): character-level
pre/post-processing, ``untruecase``, ``pipeline``.  ``preprocess_add_noise`` keeps the reference's
behaviour of copying the right neighbour over a random position (it does not swap)."""
from random import randint
from typing import Any, Callable, Iterator, List


def preprocess_char_based(sentence: List[str]) -> List[str]:
    return list(" ".join(sentence))


def postprocess_char_based(sentences: List[List[str]]) -> List[List[str]]:
    return ["".join(sentence).split(" ") for sentence in sentences]


def preprocess_add_noise(sentence: List[str]) -> List[str]:
    noisy = list(sentence)
    if len(noisy) > 1:
        for _ in range(len(noisy) // 2):
            pos = randint(0, len(noisy) - 2)
            noisy[pos] = noisy[pos + 1]
    return noisy


def untruecase(sentences: List[List[str]]) -> Iterator[List[str]]:
    for sentence in sentences:
        yield [sentence[0].capitalize()] + sentence[1:] if sentence else []


def pipeline(processors: List[Callable]) -> Callable:
    """The processors one after another."""
    def process(data: Any) -> Any:
        for processor in processors:
            data = processor(data)
        return data
    return process
