"""Vocabulary, padding and masks (mirror of neuralmonkey/vocabulary.py).

The reference pads *string* tokens on the host and looks them up inside the TF
graph with a hash table (vocabulary.py:187-195, 224-244).  Here the lookup
happens on the host at feed time and the device only ever sees int32 ids
(SURVEY section 8 row a1).
"""
from typing import Dict, Iterable, List, Optional, Set, Union

import numpy as np

PAD_TOKEN = "<pad>"
START_TOKEN = "<s>"
END_TOKEN = "</s>"
UNK_TOKEN = "<unk>"
SPECIAL_TOKENS = [PAD_TOKEN, START_TOKEN, END_TOKEN, UNK_TOKEN]
PAD_TOKEN_INDEX = 0
START_TOKEN_INDEX = 1
END_TOKEN_INDEX = 2
UNK_TOKEN_INDEX = 3


class Vocabulary:
    def __init__(self, words: List[str], num_oov_buckets: int = 0) -> None:
        self._vocabulary = SPECIAL_TOKENS + list(words)
        self._alphabet = {c for word in words for c in word}
        self._word_to_index: Dict[str, int] = {}
        for i, w in enumerate(self._vocabulary):
            self._word_to_index.setdefault(w, i)
        self.num_oov_buckets = num_oov_buckets

    def __len__(self) -> int:
        return len(self._vocabulary)

    def __contains__(self, word: str) -> bool:
        return word in self._word_to_index

    @property
    def alphabet(self) -> Set[str]:
        return self._alphabet

    @property
    def index_to_word(self) -> List[str]:
        return self._vocabulary

    def strings_to_indices(self, sentences: Iterable[Iterable[str]]) -> np.ndarray:
        """[batch, time] padded token strings -> int32 ids, OOV -> <unk>."""
        rows = [[self._word_to_index.get(tok, UNK_TOKEN_INDEX) for tok in sent] for sent in sentences]
        return np.asarray(rows, dtype=np.int32).reshape(len(rows), -1)

    def indices_to_strings(self, vectors) -> List[List[str]]:
        n = len(self._vocabulary)
        return [[self._vocabulary[i] if 0 <= i < n else UNK_TOKEN for i in row] for row in vectors]

    def vectors_to_sentences(self, vectors: Union[List[np.ndarray], np.ndarray]) -> List[List[str]]:
        """TIME-MAJOR id vectors -> token lists cut at </s> (vocabulary.py:257-288)."""
        if isinstance(vectors, list):
            if not vectors:
                raise ValueError("Cannot infer batch size because decoder returned an empty output.")
            batch_size = vectors[0].shape[0]
        elif isinstance(vectors, np.ndarray):
            batch_size = vectors.shape[1]
        else:
            raise TypeError("Unexpected type of decoder output: {}".format(type(vectors)))
        # vectorised: one fancy-index lookup for the whole [T,B] block, every sentence cut before its first </s>
        # (the reference appends token by token in Python: 1.5 ms per 128 x 50 batch, a fifth of a decoded batch here)
        arr = np.stack([np.asarray(v) for v in vectors]) if isinstance(vectors, list) else np.asarray(vectors)
        if arr.shape[0] == 0:                       # a loop that ran zero steps: empty sentences, as the token loop gave
            return [[] for _ in range(batch_size)]
        arr = arr.reshape(arr.shape[0], batch_size).astype(np.int64, copy=False)
        table = self.__dict__.get("_i2w_array")
        if table is None or len(table) != len(self.index_to_word):
            table = np.empty(len(self.index_to_word), dtype=object)
            table[:] = self.index_to_word
            self.__dict__["_i2w_array"] = table
        # (</s> is found by its index -- the word list holds it exactly once, at END_TOKEN_INDEX -- on the integer
        # block, and the words are looked up batch-major so that a sentence is a contiguous row: 0.29 -> 0.15 ms per
        # 128 x 50 batch, host time that sits between two decoded batches)
        is_end = arr == END_TOKEN_INDEX
        first_end = np.where(is_end.any(axis=0), is_end.argmax(axis=0), arr.shape[0])
        words = table[arr.T]                                            # object array [B,T]
        return [row[:n].tolist() for row, n in zip(words, first_end)]

    def save_wordlist(self, path: str, overwrite: bool = False, encoding: str = "utf-8") -> None:
        import os
        if os.path.exists(path) and not overwrite:
            raise FileExistsError("Cannot save vocabulary: File exists and overwrite is disabled. {}"
                                  .format(path))
        with open(path, "w", encoding=encoding) as out:
            for word in self._vocabulary:
                out.write("{}\n".format(word))


def from_wordlist(path: str, encoding: str = "utf-8", contains_header: bool = True,
                  contains_frequencies: bool = True) -> Vocabulary:
    """vocabulary.py:34-99: one word per line (optionally ``word<TAB>count``
    with a header line); the four leading special tokens are skipped."""
    words: List[str] = []
    with open(path, encoding=encoding) as handle:
        lines = handle.read().split("\n")
    first_data = 1 if contains_header else 0
    for lineno, raw in enumerate(lines):
        if lineno < first_data:
            continue
        line = raw.strip()
        if not line:
            continue
        if contains_frequencies:
            cols = line.split("\t")
            if len(cols) != 2:
                raise ValueError("Vocabulary file {}:{}: line does not have two columns"
                                 .format(path, lineno + 1))
            word = cols[0]
        else:
            word = line
        slot = lineno - first_data
        if slot < len(SPECIAL_TOKENS):
            if word != SPECIAL_TOKENS[slot]:
                words.append(word)
            continue
        words.append(word)
    return Vocabulary(words)


def from_dataset(datasets, series_ids: List[str], max_size: int, save_to_file: str = None,
                 overwrite: bool = False, min_freq: Optional[int] = None,
                 unk_sample_prob: float = 0.5) -> Vocabulary:
    """Most frequent tokens of the given series (vocabulary.py from_dataset)."""
    from collections import Counter
    counts: Counter = Counter()
    for ds in datasets:
        for sid in series_ids:
            for sent in ds.get_series(sid):
                counts.update(sent)
    items = sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))
    if min_freq is not None:
        items = [kv for kv in items if kv[1] >= min_freq]
    words = [w for w, _ in items][:max(0, max_size - len(SPECIAL_TOKENS))] if max_size > 0 else \
        [w for w, _ in items]
    vocab = Vocabulary(words)
    if save_to_file is not None:
        vocab.save_wordlist(save_to_file, overwrite)
    return vocab


def from_t2t_vocabulary(path: str, encoding: str = "utf-8") -> Vocabulary:
    """vocabulary.py:102-134: a tensor2tensor subword vocabulary, one (optionally quoted) token per
    line; its own ``<pad>`` / ``<EOS>`` entries are dropped (ours come first anyway)."""
    words: List[str] = []
    with open(path, encoding=encoding) as handle:
        for raw in handle:
            token = raw.strip()
            if len(token) >= 2 and token[0] == token[-1] and token[0] in "'\"":
                token = token[1:-1]
            if token not in ("<pad>", "<EOS>"):
                words.append(token)
    return Vocabulary(words)


def from_nematus_json(path: str, max_size: int = None, pad_to_max_size: bool = False) -> Vocabulary:
    """vocabulary.py:137-173: a Nematus word -> index dictionary.  Words come in index order, the two
    lowest indices (Nematus' own <eos> / unk slots) are skipped, ``max_size`` counts real words, and
    ``pad_to_max_size`` fills up with ``<pad_i>`` dummies so that an imported embedding matrix keeps
    its row count (the import path of ``scripts/import_nematus.py``)."""
    import json
    with open(path, "r", encoding="utf-8") as handle:
        index_of = json.load(handle)
    words: List[str] = []
    for word in sorted(index_of, key=lambda w: index_of[w]):
        if index_of[word] < 2:
            continue
        words.append(word)
        if max_size is not None and len(words) == max_size:
            break
    if max_size is None:
        max_size = len(words) - 2          # the reference's own off-by-two, kept: no padding is added below
    if pad_to_max_size:
        words.extend("<pad_{}>".format(i) for i in range(max_size - len(words) + 2))
    return Vocabulary(words)


def pad_batch(sentences: List[List[str]], max_length: int = None, add_start_symbol: bool = False,
              add_end_symbol: bool = False) -> List[List[str]]:
    """vocabulary.py:331-354 (</s> may be truncated away by max_length)."""
    longest = max(len(s) for s in sentences)
    if add_end_symbol:
        longest += 1
    if max_length is not None:
        longest = min(max_length, longest)
    padded = []
    for sent in sentences:
        row = list(sent)
        if add_end_symbol:
            row.append(END_TOKEN)
        row = (row + [PAD_TOKEN] * longest)[:longest]
        if add_start_symbol:
            row.insert(0, START_TOKEN)
        padded.append(row)
    return padded


def sentence_mask(ids: np.ndarray) -> np.ndarray:
    """vocabulary.py:357-358."""
    return (np.asarray(ids) != PAD_TOKEN_INDEX).astype(np.float32)
