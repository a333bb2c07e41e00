"""Known answers taken from the reference's own unit tests (test data, not code): the INI value grammar
(neuralmonkey/tests/test_config.py:8-43), piecewise_function (test_functions.py:12-22) and the vocabulary
(test_vocabulary.py:14-66), bucketed batching (test_dataset.py:169-258), the constructor tables
(test_encoders_init.py, test_decoder.py), variable sharing and per-part checkpoints (test_model_part.py)
-- the few answers on this path that the reference itself pins."""
import numpy as np
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


# ---- neuralmonkey/tests/test_model_part.py: variable sharing by ``reuse``, per-part checkpoints ---------
class _HostSession:
    """Parameterized.save / load only touch ``session.store``; a host-memory store has no kernels to run."""

    def __init__(self, parts, seed):
        from neuralmonkey_amd.variables import VariableStore
        self.store = VariableStore("cpu", seed=seed)
        for part in parts:
            part.declare_variables(self.store)
        self.store.finalize()


def test_reuse_shares_the_embedding_matrix():
    from neuralmonkey_amd.model.sequence import EmbeddedSequence
    from neuralmonkey_amd.vocabulary import Vocabulary
    vocabulary = Vocabulary(["a", "b"])
    seq1 = EmbeddedSequence(name="seq1", vocabulary=vocabulary, data_id="id", embedding_size=10)
    seq2 = EmbeddedSequence(name="seq2", vocabulary=vocabulary, embedding_size=10, data_id="id")
    seq3 = EmbeddedSequence(name="seq3", vocabulary=vocabulary, data_id="id", embedding_size=10, reuse=seq1)
    sess = _HostSession([seq1, seq2, seq3], seed=3)
    mats = [part.var(sess, part.variable_names(sess.store)[0].split("/", 1)[1]) for part in (seq1, seq2, seq3)]
    assert not np.array_equal(mats[0].numpy(), mats[1].numpy())
    assert mats[0].data_ptr() == mats[2].data_ptr()                  # one variable, not two equal ones
    assert seq3.variable_names(sess.store) == seq1.variable_names(sess.store)
    with pytest.raises(ValueError):                                  # parameterized.py:48-52
        EmbeddedSequence(name="seq4", vocabulary=vocabulary, data_id="id", embedding_size=10, reuse=seq1,
                         initializers=[("word_embeddings", lambda shape: np.zeros(shape))])


def test_part_checkpoint_save_and_load(tmp_path):
    from neuralmonkey_amd import tf_bundle
    from neuralmonkey_amd.encoders.recurrent import SentenceEncoder
    from neuralmonkey_amd.vocabulary import Vocabulary
    prefix = str(tmp_path / "enc.ckpt")
    encoder = SentenceEncoder(name="enc", vocabulary=Vocabulary(["a", "b"]), data_id="data_id", embedding_size=10,
                              rnn_size=20, max_input_len=30, save_checkpoint=prefix, load_checkpoint=prefix)
    parts = [encoder.input_sequence, encoder]
    sess_1, sess_2 = _HostSession(parts, seed=1), _HostSession(parts, seed=2)
    names = encoder.variable_names(sess_1.store)
    assert names and any(not np.array_equal(sess_1.store[n].numpy(), sess_2.store[n].numpy()) for n in names)
    encoder.save(sess_1)
    bundle = tf_bundle.read_bundle(prefix)                           # a TensorFlow tensor bundle of the scope "enc"
    assert sorted(bundle) == sorted(names) and all(n.startswith("enc/") for n in names)
    before = {n: sess_2.store[n].numpy().copy() for n in sess_2.store.names() if n not in names}
    encoder.load(sess_2)
    for n in names:
        assert np.array_equal(sess_1.store[n].numpy(), sess_2.store[n].numpy()), n
    for n, value in before.items():                                  # the embeddings live in their own scope
        assert np.array_equal(value, sess_2.store[n].numpy()), n
    # a checkpoint that lacks one of the scope's variables is refused, as Saver.restore refuses it
    partial = dict(bundle)
    partial.pop(names[0])
    tf_bundle.write_bundle(prefix, partial)
    with pytest.raises(KeyError):
        encoder.load(sess_2)


# ---- neuralmonkey/tests/test_readers.py: vectors as text, the T2T tokenizer -----------------------------
INT_ROWS = "\n1   2 3\n4 5   6\n7 8 9 10\n\n"
FLOAT_ROWS = "\n1 2       3.5\n      4 -5.0e10     6\n7 8 9.2e-12 10.1123213213214123141234123112312312\n"
SQUARE_INT_ROWS = "\n1 2 3\n4 5 6\n7 8 9\n"


def _rows(text, dtype):
    return [np.array(row.split(), dtype=dtype) for row in text.strip().split("\n")]


def _same(got, want):
    return len(got) == len(want) and all(a.dtype == b.dtype and np.array_equal(a, b) for a, b in zip(got, want))


def test_string_vector_reader_known_answers(tmp_path):
    from neuralmonkey_amd.readers.string_vector_reader import get_string_vector_reader
    paths = {}
    for name, text in (("ints", INT_ROWS), ("floats", FLOAT_ROWS), ("square", SQUARE_INT_ROWS)):
        paths[name] = str(tmp_path / name)
        with open(paths[name], "w") as handle:
            handle.write(text)
    assert _same(list(get_string_vector_reader(np.float32)([paths["floats"]])), _rows(FLOAT_ROWS, np.float32))
    assert _same(list(get_string_vector_reader(np.int32)([paths["ints"], paths["square"]])),
                 _rows(INT_ROWS, np.int32) + _rows(SQUARE_INT_ROWS, np.int32))
    for cols in (2, 3):
        for name, dtype in (("ints", np.int32), ("floats", np.float32)):
            with pytest.raises(ValueError, match="Wrong number of columns"):
                list(get_string_vector_reader(dtype, columns=cols)([paths[name]]))
    with pytest.raises(ValueError, match=r"Wrong number of columns \(4\) on line 4"):   # blank lines are counted
        list(get_string_vector_reader(np.int32, columns=3)([paths["ints"]]))
    with pytest.raises(ValueError, match="Wrong number of columns"):
        list(get_string_vector_reader(np.int32, columns=2)([paths["square"]]))
    assert _same(list(get_string_vector_reader(np.int32, columns=3)([paths["square"]])), _rows(SQUARE_INT_ROWS, np.int32))


@pytest.mark.filterwarnings("ignore:A mismatch in number of columns")
def test_text_readers_known_answers(tmp_path):
    import gzip
    from neuralmonkey_amd.readers.plain_text_reader import (T2TReader, UtfPlainTextReader, csv_reader, t2t_tokenize,
                                                            tsv_reader)
    path = str(tmp_path / "text")
    with open(path, "w", encoding="utf-8") as handle:
        handle.write("Ich bin  der čermák -=- - !!! alfonso ")
    assert list(T2TReader([path])) == [["Ich", "bin", "  ", "der", "čermák", " -=- - !!! ", "alfonso"]]
    assert list(UtfPlainTextReader([path])) == [["Ich", "bin", "der", "čermák", "-=-", "-", "!!!", "alfonso"]]
    assert t2t_tokenize("") == [""] and t2t_tokenize("a b") == ["a", "b"] and t2t_tokenize(", a") == [", ", "a"]
    zipped = str(tmp_path / "text.gz")
    with gzip.open(zipped, "wt", encoding="utf-8") as handle:
        handle.write("a b\n\nčermák\n")
    assert list(UtfPlainTextReader([zipped, path]))[:3] == [["a", "b"], [], ["čermák"]]
    table = str(tmp_path / "table")
    with open(table, "w", encoding="utf-8") as handle:
        handle.write('one two\t"x, y" z\nthree\n')
    with pytest.warns(UserWarning, match="missing column number 2"):
        assert list(tsv_reader(2)([table])) == [['"x,', 'y"', "z"], []]
    assert list(tsv_reader(1)([table])) == [["one", "two"], ["three"]]
    with open(table, "w", encoding="utf-8") as handle:
        handle.write('one two, "x, y" z\n')
    assert list(csv_reader(2)([table])) == [["x,", "y", "z"]]


# ---- neuralmonkey/tests/test_wordpiece.py -----------------------------------------------------------------
def _wordpiece_vocabulary():
    from neuralmonkey_amd.vocabulary import Vocabulary
    corpus = ["the colorless ideas slept furiously", "pooh slept all night", "working class hero is something to be",
              "I am the working class walrus", "walrus for president"]
    words = {word + "_" for sentence in corpus for word in sentence.split()}
    chars = {piece for char in set("".join(corpus)) for piece in (char, char + "_")}
    return Vocabulary(list(chars | set("\\_u0987654321;") | words) + ["\\269;", "225"])


WORDPIECES = [
    ("I am the walrus", "I_ am_ the_ walrus_"),
    ("Ich bin der walrus", "I c h_ b i n_ d e r_ walrus_"),
    ("Ich bin der čermák", "I c h_ b i n_ d e r_ \\269; e r m \\ 225 ; k_"),
]


@pytest.mark.parametrize("raw,pieces", WORDPIECES)
def test_wordpiece_known_answers(raw, pieces):
    from neuralmonkey_amd.processors.wordpiece import WordpiecePostprocessor, WordpiecePreprocessor
    preprocessor = WordpiecePreprocessor(_wordpiece_vocabulary())
    assert preprocessor(raw.split()) == pieces.split()
    assert WordpiecePostprocessor([pieces.split()]) == [raw.split()]


def test_wordpiece_refusals():
    from neuralmonkey_amd.processors.wordpiece import WordpiecePreprocessor, unescape_token
    from neuralmonkey_amd.vocabulary import Vocabulary
    with pytest.raises(TypeError):
        WordpiecePreprocessor(["not", "a", "vocabulary"])
    with pytest.raises(AssertionError, match="No token substring"):
        WordpiecePreprocessor(Vocabulary(["a", "a_"]))(["ab"])        # "b" is written as its code point, which has no pieces
    assert unescape_token("a\\ub\\\\c\\99999999999;_") == "a_b\\c〓"


# ---- neuralmonkey/dataset.py:207-333: the three kinds of series -----------------------------------------
def test_dataset_load_series_kinds(tmp_path):
    from neuralmonkey_amd import dataset
    from neuralmonkey_amd.processors.helpers import pipeline, preprocess_char_based
    from neuralmonkey_amd.readers.plain_text_reader import T2TReader
    path = str(tmp_path / "src")
    with open(path, "w") as handle:
        handle.write("ab c\nd, e\n")

    def lengths(iterators):
        return ([str(len(a)), str(len(b))] for a, b in zip(iterators["source"](), iterators["chars"]()))

    data = dataset.load("d", ["source", "t2t", "chars", "lens"],
                        [path, (path, T2TReader), (pipeline([preprocess_char_based]), "source"), lengths],
                        dataset.BatchingScheme(batch_size=2))
    assert list(data.get_series("source")) == [["ab", "c"], ["d,", "e"]]
    assert list(data.get_series("t2t")) == [["ab", "c"], ["d", ", ", "e"]]
    assert list(data.get_series("chars")) == [list("ab c"), list("d, e")]
    assert list(data.get_series("lens")) == [["2", "4"], ["2", "4"]]
    with pytest.raises(ValueError, match="series-level preprocessor"):   # preprocessors do not stack
        dataset.load("d", ["source", "chars", "chars2"], [path, (preprocess_char_based, "source"),
                                                          (preprocess_char_based, "chars")],
                     dataset.BatchingScheme(batch_size=2))
    with pytest.raises(ValueError, match="Multiple outputs"):
        dataset.load("d", ["source"], [path], dataset.BatchingScheme(batch_size=2),
                     outputs=[("target", "a.txt"), ("target", "b.txt")])
    with pytest.raises(ValueError, match="duplicate"):
        dataset.load("d", ["source", "source"], [path, path], dataset.BatchingScheme(batch_size=2))
    with pytest.raises(ValueError, match="from a file"):        # dataset.py:249-250: nothing here is a file series
        dataset.load("d", ["source"], [3], dataset.BatchingScheme(batch_size=2))
    with pytest.raises(TypeError):                              # (the reference: a bare assert, dataset.py:300)
        dataset.load("d", ["source", "x"], [path, 3], dataset.BatchingScheme(batch_size=2))
