"""Edit-operation series for automatic post-editing (neuralmonkey/processors/editops.py, used by the reference's
acceptance configuration tests/post-edit.ini): the decoder is trained to emit, for a machine-translated sentence,
the script that turns it into its post-edited version -- ``<keep>`` (copy the next token), ``<delete>`` (skip it), or
a word (insert it) -- and the script is applied to the translation again after decoding.

``Preprocess`` is a dataset-level preprocessor (``dataset.load``: a callable in ``data`` receives the iterators of
the other series), ``Postprocess`` a [main] ``postprocess`` entry (dataset series + generated series -> sentences).
Host side only.

The script is the cheapest insert / delete alignment of the two sentences (no substitutions), with the reference's
preference among equally cheap continuations: keep before delete before insert, decided from the END of both
sentences backwards (editops.py:62-95 grows a script per table cell by exactly these choices; here the cost table
is filled first and one path is read back, which visits the same cells and takes the same choices).
"""
from typing import Any, Callable, Dict, Iterable, Iterator, List

import numpy as np

KEEP = "<keep>"
DELETE = "<delete>"


def _cost_table(source: List[str], target: List[str]) -> np.ndarray:
    """cost[i, j] = fewest insertions + deletions that turn source[:i] into target[:j]."""
    n, m = len(source), len(target)
    cost = np.zeros((n + 1, m + 1))
    cost[:, 0] = np.arange(n + 1)
    cost[0, :] = np.arange(m + 1)
    for j in range(1, m + 1):
        word = target[j - 1]
        for i in range(1, n + 1):
            best = min(cost[i - 1, j], cost[i, j - 1]) + 1
            if source[i - 1] == word and cost[i - 1, j - 1] < best:
                best = cost[i - 1, j - 1]
            cost[i, j] = best
    return cost


def convert_to_edits(source: List[str], target: List[str]) -> List[str]:
    cost = _cost_table(source, target)
    i, j = len(source), len(target)
    script: List[str] = []
    while i > 0 and j > 0:
        here = cost[i, j]
        if source[i - 1] == target[j - 1] and here == cost[i - 1, j - 1]:
            script.append(KEEP)
            i, j = i - 1, j - 1
        elif here == cost[i - 1, j] + 1:
            script.append(DELETE)
            i -= 1
        else:
            script.append(target[j - 1])
            j -= 1
    # one of the sentences is used up: the rest of the other is deleted / inserted
    script.extend([DELETE] * i)
    script.extend(reversed(target[:j]))
    script.reverse()
    return script


def reconstruct(source: List[str], edits: List[str]) -> List[str]:
    """Apply a script.  A ``<keep>`` beyond the end of the source copies nothing; source tokens the script never
    reached (a script cut short by the decoder's length limit) are copied at the end."""
    position, rebuilt = 0, []
    for op in edits:
        if op == DELETE:
            position += 1
        elif op == KEEP:
            rebuilt.extend(source[position:position + 1])
            position += 1
        else:
            rebuilt.append(op)
    rebuilt.extend(source[position:])
    return rebuilt


class Preprocess:
    """The series of scripts that turn ``source_id``'s sentences into ``target_id``'s."""

    def __init__(self, source_id: str, target_id: str) -> None:
        self._source_id, self._target_id = source_id, target_id

    def __call__(self, iterators: Dict[str, Callable[[], Iterator[List[str]]]]) -> Iterator[List[str]]:
        # a generator function: the two series are opened when the first script is asked for, not before
        yield from map(convert_to_edits, iterators[self._source_id](), iterators[self._target_id]())


class Postprocess:
    """The generated scripts (``edits_id``) applied to the dataset's ``source_id`` sentences."""

    def __init__(self, source_id: str, edits_id: str) -> None:
        self._source_id, self._edits_id = source_id, edits_id

    def __call__(self, dataset: Dict[str, Iterable[Any]], generated: Dict[str, Iterable[Any]]) -> List[List[str]]:
        if self._source_id not in dataset:
            raise ValueError("Source series not present in the input dataset")
        if self._edits_id not in generated:
            raise ValueError("Edits series not present in the output dataset")
        return [reconstruct(sentence, script)
                for sentence, script in zip(dataset[self._source_id], generated[self._edits_id])]
