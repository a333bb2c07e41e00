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


# ---- neuralmonkey/tests/test_encoders_init.py, test_decoder.py: constructors refuse bad arguments -------
def _refuses_each(cls, good, bad):
    serial = 0
    for key, values in bad.items():
        for value in values:
            options = dict(good)
            options[key] = value
            if key != "name":
                options["name"] = "{}_{}".format(good["name"], serial)
            serial += 1
            with pytest.raises(Exception):
                cls(**options)
                pytest.fail("{} accepted {}={!r}".format(cls.__name__, key, value))
    return serial


def test_sentence_encoder_constructor_table():
    from neuralmonkey_amd.encoders.recurrent import SentenceEncoder
    from neuralmonkey_amd.vocabulary import Vocabulary
    vocabulary = Vocabulary(["ich", "bin", "der", "walrus"])
    good = dict(name="encoder", vocabulary=vocabulary, data_id="marmelade", embedding_size=20, rnn_size=30,
                max_input_len=None, dropout_keep_prob=0.5)
    junk = ["ahoj", 3.14, vocabulary, SentenceEncoder]
    bad = {"nonexistent": ["ahoj"], "name": [None, 1], "vocabulary": [0, None, "ahoj", dict()],
           "data_id": [0, None, vocabulary], "embedding_size": [-1, 0, None] + junk, "rnn_size": [-1, 0, None] + junk,
           "max_input_len": [-1, 0] + junk, "dropout_keep_prob": [0.0, 0, -1.0, 2.0, "ahoj", vocabulary, None]}
    with pytest.raises(TypeError):
        SentenceEncoder()                                       # pylint: disable=no-value-for-parameter
    serial = _refuses_each(SentenceEncoder, good, bad)
    for max_len in (None, 15):
        for keep in (0.5, 1.0):
            serial += 1
            SentenceEncoder(**dict(good, name="encoder_{}".format(serial), max_input_len=max_len,
                                   dropout_keep_prob=keep))


def test_transformer_encoder_constructor_table():
    from neuralmonkey_amd.encoders.transformer import TransformerEncoder
    from neuralmonkey_amd.model.sequence import EmbeddedSequence
    from neuralmonkey_amd.vocabulary import Vocabulary
    vocabulary = Vocabulary(["ich", "bin", "der", "walrus"])
    sequence = EmbeddedSequence("seq", vocabulary, "marmelade", 300)
    good = dict(name="transformer_encoder", input_sequence=sequence, ff_hidden_size=10, depth=6, n_heads=3,
                dropout_keep_prob=0.5)
    junk = ["ahoj", 3.14, TransformerEncoder, None]
    bad = {"nonexistent": ["ahoj"], "name": [None, 1], "input_sequence": [0, None, vocabulary],
           "ff_hidden_size": [-1, 0, vocabulary] + junk, "depth": [-1] + junk, "n_heads": [-1] + junk,
           "dropout_keep_prob": [0.0, 0, -1.0, 2.0, "ahoj", vocabulary, None]}
    serial = _refuses_each(TransformerEncoder, good, bad)
    TransformerEncoder(**dict(good, name="transformer_encoder_{}".format(serial)))


def test_decoder_constructor_known_answers():
    from neuralmonkey_amd.decoders.decoder import Decoder
    from neuralmonkey_amd.vocabulary import Vocabulary
    good = dict(encoders=[], vocabulary=Vocabulary(["a", "b", "c"]), data_id="foo", name="test-decoder",
                max_output_len=5, dropout_keep_prob=1.0, embedding_size=10, rnn_size=10)
    Decoder(**good)
    bad = {"max_output_len": [-10], "dropout_keep_prob": [-0.5, 1.5], "embedding_size": [None, -10],
           "rnn_cell": ["bogus_cell"]}
    for key, values in bad.items():
        for value in values:
            with pytest.raises(ValueError):
                Decoder(**dict(good, **{key: value}))
    for cell in ("GRU", "LSTM", "NematusGRU"):
        Decoder(**dict(good, rnn_cell=cell, name="test-decoder-{}".format(cell)))


def test_argument_type_matching():
    from typing import Callable, Dict, List, Optional, Tuple, Union
    from neuralmonkey_amd.checking import check_argument_types, matches
    assert matches(3, float) and matches(3.0, float) and not matches(True, float) and not matches("3", float)
    assert matches(3, int) and not matches(3.0, int) and not matches(True, int)
    assert matches(None, Optional[int]) and matches(2, Optional[int]) and not matches("x", Optional[int])
    assert matches([1, 2], List[int]) and not matches([1, "2"], List[int]) and not matches((1, 2), List[int])
    assert matches([(2, 10)], List[Tuple[int, int]]) and not matches([(1, 2, 3)], List[Tuple[int, int]])
    assert matches({"a": 1}, Dict[str, int]) and not matches({"a": "b"}, Dict[str, int])
    assert matches(len, Callable[[List[str]], int]) and not matches(3, Callable)
    assert matches("x", Union[str, List[str]]) and matches(["x"], Union[str, List[str]])
    assert matches(3, "int") and not matches(3, "Vocabulary")          # forward references go by class name

    def build(size: int, names: List[str] = None, rate: float = 1.0):
        check_argument_types()
        return size, names, rate

    build(3), build(3, ["a"], 1)
    for args in ((3.5,), (3, "a"), (3, None, "fast")):
        with pytest.raises(TypeError):
            build(*args)
