"""Known answers taken from the reference's own unit tests (test data, not code): the INI value grammar
(neuralmonkey/tests/test_config.py:8-43), piecewise_function (test_functions.py:12-22) and the vocabulary
(test_vocabulary.py:14-66) -- the few numbers on this path that the reference itself pins."""
import pytest

from neuralmonkey_amd.config import parsing

SPLITTER = [
    ("", []),
    (",,,,,,", []),
    (",    ,   ,,   , , ", []),
    ("without", ["without"]),
    ("a,b,c", ["a", "b", "c"]),
    ("(brackets),(brac,kets)", ["(brackets)", "(brac,kets)"]),
]


@pytest.mark.parametrize("text,expected", SPLITTER)
def test_top_level_comma_splitter(text, expected):
    assert parsing._split_top_level(text) == expected          # pylint: disable=protected-access


def test_splitter_rejects_mismatched_brackets():
    with pytest.raises(Exception):
        parsing._split_top_level("(omg,brac],kets")            # pylint: disable=protected-access


def test_numbers():
    assert parsing.parse_value("42", {}) == 42
    assert parsing.parse_value("-42", {}) == -42
    for text in ("0.5e-1", ".5e-1", "-.5e-1", "5.e-1"):
        assert parsing.parse_value(text, {}) == pytest.approx(float(text))
        assert isinstance(parsing.parse_value(text, {}), float)


def test_strings_substitute_variables():
    variables = {"pi": 3.14159, "greeting": "hello"}
    assert parsing.parse_value('"{greeting}"world"', variables) == 'hello"world'
    assert parsing.parse_value('"pi = {pi:.0f}"', variables) == "pi = 3"


# ---- neuralmonkey/tests/test_functions.py:12-22 ---------------------------------------------------------
def test_piecewise_function_known_answers():
    from neuralmonkey_amd.functions import piecewise_function
    y = piecewise_function(lambda x: x, [-0.5, 1.2, 3, 2], [-1, 2, 1000])
    assert [y(x) for x in (-2, -1, 999, 1000, 1001)] == [-0.5, 1.2, 3, 2, 2]
    with pytest.raises(ValueError):
        piecewise_function(lambda x: x, [1, 2], [0, 1])


# ---- neuralmonkey/tests/test_vocabulary.py:14-66 ---------------------------------------------------------
CORPUS = ["the colorless ideas slept furiously", "pooh slept all night", "working class hero is something to be",
          "I am the working class walrus", "walrus for president"]


def test_vocabulary_known_answers():
    from neuralmonkey_amd.vocabulary import Vocabulary, pad_batch
    tokenized = [s.split(" ") for s in CORPUS]
    vocabulary = Vocabulary(sorted({w for sent in tokenized for w in sent}))
    assert all(word in vocabulary for sent in tokenized for word in sent)
    assert "jindrisek" not in vocabulary
    assert all(len(p) == 7 for p in pad_batch(tokenized))
    padded = pad_batch(tokenized, max_length=20, add_start_symbol=False, add_end_symbol=True)
    vectors = vocabulary.strings_to_indices(padded).T                       # time-major, as the runners hand it over
    assert vocabulary.vectors_to_sentences(vectors) == tokenized            # there and back


# ---- neuralmonkey/tests/test_dataset.py:169-258: bucketed batching --------------------------------------
def _word_lists(lengths):
    return [["word"] * n for n in lengths]


def test_bucketing_known_answers():
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    scheme = BatchingScheme(bucket_boundaries=[9, 19, 29, 39, 49], bucket_batch_sizes=[7, 7, 7, 7, 7, 7])
    dataset = Dataset("dataset", {"sentences": _word_lists(range(1, 50))}, scheme)
    batches = [list(batch.get_series("sentences")) for batch in dataset.batches()]
    expected = [_word_lists(r) for r in (range(1, 8), range(10, 17), range(20, 27), range(30, 37), range(40, 47),
                                         range(8, 10), range(17, 20), range(27, 30), range(37, 40), range(47, 50))]
    assert batches == expected


def test_bucketing_drop_remainder_known_answers():
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    scheme = BatchingScheme(bucket_boundaries=[9, 19, 29, 39, 49], bucket_batch_sizes=[7, 7, 7, 7, 7, 7],
                            drop_remainder=True)
    dataset = Dataset("dataset", {"sentences": _word_lists(range(1, 50))}, scheme)
    batches = [list(batch.get_series("sentences")) for batch in dataset.batches()]
    assert batches == [_word_lists(r) for r in (range(1, 8), range(10, 17), range(20, 27), range(30, 37),
                                                range(40, 47))]


def test_buckets_hold_similar_lengths():
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    scheme = BatchingScheme(bucket_boundaries=[1, 3, 5], bucket_batch_sizes=[6, 6, 6, 6])
    dataset = Dataset("dataset", {"sentences": _word_lists(list(range(6)) * 3)}, scheme)
    batches = [list(batch.get_series("sentences")) for batch in dataset.batches()]
    assert len(batches) == 3
    for batch in batches:
        assert len(batch) == 6
        lengths = {len(sentence) for sentence in batch}
        assert len(lengths) == 2 and max(lengths) - min(lengths) == 1
