"""Host-side plugin surface on CPU: INI grammar and builder, vocabulary /
padding, batching, model-part protocol, constructor validation, variable
naming.  Mirrors what the reference's unit tests check
(neuralmonkey/tests/test_{config,vocabulary,dataset,decoder,encoders_init}.py)."""
import os

import numpy as np
import pytest

from neuralmonkey_amd import dataset as D
from neuralmonkey_amd import vocabulary as V
from neuralmonkey_amd.config import parsing
from neuralmonkey_amd.config.exceptions import ParseError


# ------------------------------------------------------------------ INI grammar
def pv(text, **variables):
    vd = parsing.VarsDict()
    vd.update(variables)
    return parsing.parse_value(text, vd)


def test_value_grammar():
    assert pv("42") == 42 and pv("-3") == -3
    assert pv("1.0e-8") == 1e-8 and pv(".5") == 0.5 and pv("3e4") == 3e4 and pv("-2.") == -2.0
    assert pv("True") is True and pv("False") is False and pv("None") is None
    assert pv('"a {x} b"', x=7) == "a 7 b"
    assert pv("$x", x=[1, 2]) == [1, 2]
    assert pv("[1, 2.5, \"s\"]") == [1, 2.5, "s"] and pv("[]") == []
    assert pv("(1, [2, 3], (4, 5))") == (1, [2, 3], (4, 5))
    ref = pv("<encoder.input_sequence>")
    assert isinstance(ref, parsing.ObjectRef) and ref.name == "encoder" and ref.attr_chain == ["input_sequence"]
    cls = pv("decoders.output_projection.nonlinear_output")
    assert isinstance(cls, parsing.ClassSymbol) and cls.clazz.endswith("nonlinear_output")
    tup = pv('[("target", evaluators.BLEU), ("a,b", 1)]')
    assert tup[0][0] == "target" and tup[1] == ("a,b", 1)
    with pytest.raises(ParseError):
        pv("[1, 2")
    with pytest.raises(ParseError):
        pv("$undefined_variable_xyz")


def test_env_fallback_and_overrides(monkeypatch):
    monkeypatch.setenv("NM_EXPERIMENT_NAME", "exp1")
    ini = ["[vars]", "dim=8", "[main]", 'name="{NM_EXPERIMENT_NAME}-{dim}"', "size=$dim", "out=<x>",
           "[x]", "class=dataset.BatchingScheme", "batch_size=$dim"]
    raw, parsed = parsing.parse_file(ini, changes=["main.extra=5", "x.batch_size=3"])
    assert parsed["main"]["name"] == "exp1-8" and parsed["main"]["size"] == 8
    assert parsed["main"]["extra"] == 5 and parsed["x"]["batch_size"] == 3
    assert raw["main"]["size"] == "$dim"
    with pytest.raises(ParseError) as err:
        parsing.parse_file(["[main]", "ok=1", "bad=[1,"])
    assert "line 3" in str(err.value)


# ------------------------------------------------------------------ vocabulary
def test_pad_batch_and_roundtrip(tmp_path):
    sents = [["a", "b", "c"], ["a"]]
    assert V.pad_batch(sents) == [["a", "b", "c"], ["a", "<pad>", "<pad>"]]
    assert V.pad_batch(sents, 2, add_end_symbol=True) == [["a", "b"], ["a", "</s>"]]      # </s> truncated away
    assert V.pad_batch(sents, None, True, True)[1] == ["<s>", "a", "</s>", "<pad>", "<pad>"]
    vocab = V.Vocabulary(["a", "b"])
    assert len(vocab) == 6 and vocab.index_to_word[4] == "a"
    ids = vocab.strings_to_indices(V.pad_batch([["a", "zzz"], ["b"]], add_end_symbol=True))
    assert ids.tolist() == [[4, 3, 2], [5, 2, 0]] and ids.dtype == np.int32
    assert V.sentence_mask(ids).tolist() == [[1, 1, 1], [1, 1, 0]]
    assert vocab.vectors_to_sentences(list(ids.T)) == [["a", "<unk>"], ["b"]]
    path = tmp_path / "vocab.tsv"
    path.write_text("Word\tWord counts\n<pad>\t1\n<s>\t1\n</s>\t1\n<unk>\t1\ntwo\t129\nyoung\t87\n")
    loaded = V.from_wordlist(str(path))
    assert loaded.index_to_word == V.SPECIAL_TOKENS + ["two", "young"]


# ------------------------------------------------------------------ dataset
def test_batching_fixed_and_bucketed():
    ds = D.Dataset("d", {"source": [["a"] * n for n in (1, 7, 3, 12, 2, 6, 4)]},
                   D.BatchingScheme(batch_size=3))
    assert [len(b) for b in ds.batches()] == [3, 3, 1]
    assert [len(b) for b in ds.batches(D.BatchingScheme(batch_size=3, drop_remainder=True))] == [3, 3]
    bucketed = D.BatchingScheme(bucket_boundaries=[2, 5], bucket_batch_sizes=[2, 2, 1])
    lens = [[len(s) for s in b.get_series("source")] for b in ds.batches(bucketed)]
    assert lens == [[7], [12], [1, 2], [6], [3, 4]]
    with pytest.raises(ValueError):
        D.BatchingScheme(batch_size=2, bucket_boundaries=[1], bucket_batch_sizes=[1, 1])
    with pytest.raises(ValueError):
        D.BatchingScheme(bucket_boundaries=[1, 2], bucket_batch_sizes=[1, 1])
    assert ds.maybe_get_series("target") is None and "source" in ds
    assert len(ds.subset(2, 3)) == 3


# ------------------------------------------------------------------ model parts
def small_model(**kw):
    from neuralmonkey_amd import synthetic
    args = dict(vocab_src=30, vocab_tgt=40, emb=8, rnn=8, max_len=6, beam_size=2, max_steps=4, device="cpu")
    args.update(kw)
    return synthetic.build_translation_model(**args)


def test_dependency_collection_and_variable_names():
    m = small_model()
    feeds, params = m.greedy_runner.get_dependencies()
    names = sorted(p.name for p in params)
    assert names == ["attention", "decoder", "encoder", "encoder_input"]
    assert {type(f).__name__ for f in feeds} == {"Attention", "Decoder", "SentenceEncoder", "EmbeddedSequence"}
    bfeeds, bparams = m.beam_runner.get_dependencies()
    assert "beam_decoder" in {p.name for p in bparams} and m.decoder in bfeeds
    store = m.tf_manager.sessions[0].store
    assert "decoder/attention_decoder/OrthoGRUCell/gates/kernel" in store
    assert store["encoder/rnn_0_bidirectional/bidirectional_rnn/bw/OrthoGRUCell/gates/bias"].tolist() == [1.0] * 16
    assert store["decoder/state_to_word_W"].shape == (8, 40)
    assert float(store["decoder/state_to_word_W"].abs().max()) <= 0.5       # U(-0.5, 0.5)
    assert float(store["attention/attn_key_projection"].std()) < 0.01      # N(0, 0.001) scope default
    gk = store["decoder/attention_decoder/OrthoGRUCell/candidate/kernel"].numpy()
    assert np.allclose(gk.T @ gk, np.eye(8), atol=1e-5)                    # orthogonal columns
    assert all(off % 4 == 0 for off in (store.offset(n) for n in store.names()))
    assert m.trainer.regularizable(store) == [n for n in store.names() if "ias" not in n]


def test_feed_dicts_carry_int_ids():
    m = small_model()
    ds = D.Dataset("d", {"source": [["w1", "w2"], ["w3"]], "target": [["w5"], ["w6", "w7", "w8"]]})
    fd = {}
    for f in m.greedy_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    by_name = {k.name: v for k, v in fd.items()}
    assert by_name["batch_size"] == 2 and by_name["train_mode"] is False
    assert by_name["encoder_input/source"].tolist() == [[5, 6], [7, 0]]
    assert by_name["decoder/target"].tolist() == [[9, 2, 0, 0], [10, 11, 12, 2]]
    with pytest.raises(ValueError):
        m.decoder.feed_dict(D.Dataset("d", {"source": [["w1"]]}), train=True)


def test_constructor_validation_tables():
    """neuralmonkey/tests/test_decoder.py:37-75 / test_encoders_init.py:18-37."""
    from neuralmonkey_amd.decoders import Decoder
    from neuralmonkey_amd.encoders import SentenceEncoder
    from neuralmonkey_amd.runtime import reset_registry
    reset_registry()
    vocab = V.Vocabulary(["a"])
    good = dict(encoders=[], vocabulary=vocab, data_id="t", name="d", max_output_len=5, embedding_size=4,
                rnn_size=4)
    Decoder(**good)
    for bad in (dict(max_output_len=0), dict(max_output_len=-1), dict(dropout_keep_prob=1.5),
                dict(dropout_keep_prob=-0.1), dict(embedding_size=-4), dict(rnn_cell="bogus"),
                dict(embedding_size=None)):
        with pytest.raises(ValueError):
            Decoder(**{**good, **bad})
    enc_good = dict(name="e", vocabulary=vocab, data_id="s", embedding_size=4, rnn_size=4)
    SentenceEncoder(**enc_good)
    for bad in (dict(embedding_size=-1), dict(rnn_size=0), dict(dropout_keep_prob=0.0),
                dict(dropout_keep_prob=2.0), dict(max_input_len=-2), dict(rnn_direction="sideways"),
                dict(rnn_cell="bogus")):
        with pytest.raises(ValueError):
            SentenceEncoder(**{**enc_good, **bad})
    from neuralmonkey_amd.nn.dropout import dropout
    for keep in (-1.0, 2.0, 0.0):                      # test_nn_utils.py:12-20
        with pytest.raises(ValueError):
            dropout(None, None, keep, True)


def test_unsupported_features_refuse_loudly():
    from neuralmonkey_amd.decoders import Decoder
    from neuralmonkey_amd.runtime import reset_registry
    reset_registry()
    vocab = V.Vocabulary(["a"])
    from neuralmonkey_amd.decoders.encoder_projection import nematus_projection
    with pytest.raises(ValueError):
        Decoder(encoders=[], vocabulary=vocab, data_id="t", name="d", max_output_len=5, embedding_size=4,
                rnn_size=4, label_smoothing=1.5)
    with pytest.raises(ValueError):          # nematus_projection: exactly one encoder
        nematus_projection().declare_variables(None, None, 4, [])
    with pytest.raises(NotImplementedError):     # sampling inside a teacher-forced loop (no caller in the reference)
        Decoder(encoders=[], vocabulary=vocab, data_id="t", name="ds", max_output_len=5, embedding_size=4,
                rnn_size=4).decoding_loop(None, True, sample=True)
    with pytest.raises(ValueError):
        Decoder(encoders=[], vocabulary=vocab, data_id="t", name="dt", max_output_len=5, embedding_size=4,
                rnn_size=4).decoding_loop(None, False, temperature=-1.0)
    # cells / conditional GRU / attention on input are served by the general (taped) path
    for i, kw in enumerate((dict(rnn_cell="LSTM"), dict(rnn_cell="NematusGRU"), dict(attention_on_input=True))):
        dec = Decoder(encoders=[], vocabulary=vocab, data_id="t", name="dg{}".format(i), max_output_len=5,
                      embedding_size=4, rnn_size=4, **kw)
        assert dec.uses_general_path(False)
    with pytest.raises(ValueError):          # a conditional GRU conditions on attention contexts
        Decoder(encoders=[], vocabulary=vocab, data_id="t", name="d2", max_output_len=5, embedding_size=4,
                rnn_size=4, conditional_gru=True)


# ------------------------------------------------------------------ INI -> objects
INI = """
[vars]
dim=8
[main]
name="ini test {dim}"
tf_manager=<tf_manager>
batch_size=4
epochs=1
train_dataset=<train_data>
trainer=<trainer>
runners=[<runner>, <beam_runners>]
[tf_manager]
class=tf_manager.TensorFlowManager
num_threads=4
num_sessions=1
device="cpu"
[batching]
class=dataset.BatchingScheme
batch_size=4
[train_data]
class=dataset.load
series=["source", "target"]
data=["{src}", "{tgt}"]
batching=<batching>
[encoder_vocabulary]
class=vocabulary.from_wordlist
path="{vocab}"
[emb_init]
class=tf.random_uniform_initializer
minval=-0.5
maxval=0.5
[encoder]
class=encoders.recurrent.SentenceEncoder
rnn_size=$dim
max_input_len=10
embedding_size=$dim
data_id="source"
vocabulary=<encoder_vocabulary>
embedding_initializer=<emb_init>
[attention]
class=attention.Attention
encoder=<encoder>
initializers=[("Attention/attn_query_projection", <qinit>)]
[qinit]
class=tf.random_normal_initializer
stddev=0.5
[decoder]
class=decoders.decoder.Decoder
encoders=[<encoder>]
rnn_size=$dim
embedding_size=$dim
attentions=[<attention>]
output_projection=<out_proj>
data_id="target"
max_output_len=10
vocabulary=<encoder_vocabulary>
supress_unk=True
[out_proj]
class=decoders.output_projection.nonlinear_output
output_size=$dim
activation_fn=tf.tanh
[optimizer]
class=tf.contrib.opt.LazyAdamOptimizer
learning_rate=0.001
[trainer]
class=trainers.cross_entropy_trainer.CrossEntropyTrainer
decoders=[<decoder>]
l2_weight=1.0e-8
clip_norm=1.0
optimizer=<optimizer>
[runner]
class=runners.GreedyRunner
output_series="target"
decoder=<decoder>
[beam_decoder]
class=decoders.beam_search_decoder.BeamSearchDecoder
parent_decoder=<decoder>
beam_size=3
max_steps=6
length_normalization=0.6
[beam_runners]
class=runners.beam_search_runner_range
output_series="target_beam"
decoder=<beam_decoder>
max_rank=2
"""


def write_ini(tmp_path):
    (tmp_path / "src.txt").write_text("a b c\nb c\nc a a b\na\n")
    (tmp_path / "tgt.txt").write_text("x y\ny\nx x y\ny y\n")
    (tmp_path / "vocab.tsv").write_text("Word\tCount\n<pad>\t1\n<s>\t1\n</s>\t1\n<unk>\t1\n"
                                        "a\t9\nb\t8\nc\t7\nx\t6\ny\t5\n")
    ini = INI.format(src=tmp_path / "src.txt", tgt=tmp_path / "tgt.txt", vocab=tmp_path / "vocab.tsv",
                     dim="{dim}")
    path = tmp_path / "exp.ini"
    path.write_text(ini)
    return str(path)


def test_ini_with_the_adadelta_section_of_bpe_ini(tmp_path):
    """``[adadelta] class=tf.train.AdadeltaOptimizer`` with the arguments of tests/bpe.ini:102-108 / tests/str.ini
    builds through the tf.* shim; slots are named after the optimizer (TF: ``<var>/<name>``, ``<var>/<name>_1``)."""
    from neuralmonkey_amd.config.configuration import load_experiment
    from neuralmonkey_amd.optimizers import AdadeltaOptimizer
    path = write_ini(tmp_path)
    with open(path) as fh:
        text = fh.read()
    text = text.replace("class=tf.contrib.opt.LazyAdamOptimizer\nlearning_rate=0.001",
                        'class=tf.train.AdadeltaOptimizer\nname="adadelta"\nlearning_rate=1e-4\nepsilon=1.0e-6\nrho=0.95')
    assert "AdadeltaOptimizer" in text
    with open(path, "w") as fh:
        fh.write(text)
    model = load_experiment(path)
    opt = model.trainer.optimizer
    assert isinstance(opt, AdadeltaOptimizer)
    assert (opt.learning_rate(1), opt.epsilon, opt.rho) == (1e-4, 1e-6, 0.95)
    assert opt.slot_suffixes == ("/adadelta", "/adadelta_1") and AdadeltaOptimizer().slot_suffixes[0] == "/Adadelta"


def test_ini_builds_the_plugin_surface(tmp_path):
    from neuralmonkey_amd.config.configuration import load_experiment
    model = load_experiment(write_ini(tmp_path))
    assert model.name == "ini test 8" and model.batch_size == 4
    assert type(model.trainer).__name__ == "CrossEntropyTrainer"
    assert [r.output_series for r in model.runners] == ["target", "target_beam.rank001", "target_beam.rank002"]
    assert model.trainer.optimizer.learning_rate(1) == 0.001
    dec = model.runners[0].decoder
    assert dec.name == "decoder" and dec.supress_unk and dec.output_dimension == 8
    store = model.tf_manager.sessions[0].store
    assert float(store["encoder_input/embedding_matrix_0"].abs().max()) <= 0.5
    assert float(store["attention/Attention/attn_query_projection"].std()) > 0.2      # initializer override
    batch = next(model.train_dataset.batches())
    assert len(batch) == 4 and list(batch.get_series("target"))[0] == ["x", "y"]
    # section override from the command line (-s section.key=value)
    model2 = load_experiment(write_ini(tmp_path), changes=["beam_decoder.beam_size=2", "beam_runners.max_rank=1"])
    assert len(model2.runners) == 2


def test_lazy_losses_fill_in_on_first_access():
    """ExecutionResult.losses of a training step: a dict whose values are still on their way to the host."""
    import json
    import pickle

    from neuralmonkey_amd.runners.base_runner import LazyLosses
    from neuralmonkey_amd.runtime import HostPending

    class Counting(HostPending):
        reads = 0

        def get(self):
            Counting.reads += 1
            return super().get()

    make = lambda: LazyLosses(["decoder - cost", "L1", "L2"], Counting(value=np.array([1.5, 0.0, 2.5], np.float32)))
    losses = make()
    assert Counting.reads == 0                                  # nothing is read until somebody looks
    assert losses["L2"] == 2.5 and Counting.reads == 1
    assert list(losses) == ["decoder - cost", "L1", "L2"] and len(losses) == 3 and Counting.reads == 1
    for view in (lambda d: sum(d.values()), lambda d: dict(d)["L1"], lambda d: json.loads(json.dumps(d))["L2"],
                 lambda d: pickle.loads(pickle.dumps(d))["decoder - cost"], lambda d: d.get("L1", 7.0),
                 lambda d: "{}".format(d), lambda d: d == {"decoder - cost": 1.5, "L1": 0.0, "L2": 2.5},
                 lambda d: "L1" in d, lambda d: bool(d)):
        assert view(make()) is not None
    assert isinstance(make(), dict) and all(isinstance(v, float) for v in make().values())


def test_variables_signature_follows_writes():
    """What input tables and transposed step weights are cached under: torch-side writes to any variable view and
    the explicit hook of raw-pointer writers both change it; reads do not."""
    from neuralmonkey_amd.runtime import Session
    from neuralmonkey_amd.variables import zeros_initializer
    sess = Session("cpu", seed=1)
    sess.store.declare("a/w", (4, 3), zeros_initializer())
    sess.store.declare("a/b", (3,), zeros_initializer())
    sess.store.finalize()
    sig0 = sess.variables_signature()
    float(sess.store["a/w"].sum())
    assert sess.variables_signature() == sig0
    sess.store["a/b"][1] = 2.0                                  # a test poking a row, a restored checkpoint
    sig1 = sess.variables_signature()
    assert sig1 != sig0
    sess.variables_changed()                                    # optimizer kernels, collectives
    assert sess.variables_signature() not in (sig0, sig1)


def test_side_lanes_degrade_to_stream_order_without_a_gpu():
    """`Session.side` / `defer_side` / `join_side` are a schedule, not a semantics: where there is no second stream
    (CPU sessions, inside a graph capture) the enclosed work simply runs in place, deferred work runs when its time
    loop -- or whoever deferred it -- starts it, and nothing is left pending."""
    from neuralmonkey_amd.runtime import Session
    sess = Session("cpu", seed=1)
    assert not sess.side_active() and sess.leaf_algo() == 0
    order = []
    with sess.side(1):
        order.append("leaf")
    sess.defer_side(lambda: order.append("deferred a"))
    sess.defer_side(lambda: order.append("deferred b"))
    order.append("main")
    sess.start_deferred_side()
    sess.start_deferred_side()            # nothing left: a no-op
    sess.join_side(0)
    sess.join_side()
    assert order == ["leaf", "main", "deferred a", "deferred b"]
    assert not sess._deferred_side and not sess._side_dirty


@pytest.mark.parametrize("finish_at", [1, 3, 8, 11, 16, None])
@pytest.mark.parametrize("run_ahead", [False, True])
def test_decode_chunks_stops_where_the_reference_loop_stops(finish_at, run_ahead):
    """``Session.decode_chunks`` is the host side of tf.while_loop around a decoding body
    (decoders/autoregressive.py:425-437): the criterion lives in the flags the steps leave behind; the host only
    decides when to stop enqueueing.  Whatever the chunking and whether or not it reads the flags one chunk behind,
    the number of steps of the reference's loop it reports is the index of the first all-finished step + 1 (or the
    maximum), and it never enqueues more than one chunk past that."""
    import torch
    from neuralmonkey_amd.runtime import Session
    sess = Session("cpu", seed=1)
    total, every = 16, 4
    allfin = torch.zeros(total, dtype=torch.int32)
    launched = []

    def launch(t0, n):
        launched.append((t0, n))
        for t in range(t0, t0 + n):                       # step t leaves allfin[t] != 0 once every row has finished
            allfin[t] = 1 if finish_at is not None and t + 1 >= finish_at else 0
    steps, enqueued = sess.decode_chunks(total, every, launch, allfin, run_ahead=run_ahead)
    want = total if finish_at is None else finish_at
    assert steps == want
    assert enqueued == sum(n for _, n in launched) and enqueued >= steps
    chunks_needed = -(-want // every)
    assert len(launched) <= min(total // every, chunks_needed + (1 if run_ahead else 0))
    assert [t0 for t0, _ in launched] == list(range(0, enqueued, every))



def test_stateful_filler_surface():
    """encoders/numpy_stateful_filler.py:16-72 of the reference: constructor checks with its texts, the variables of
    the optional projection under ``tf.layers.dense``'s names, the fed array."""
    from neuralmonkey_amd.encoders.numpy_stateful_filler import StatefulFiller
    from neuralmonkey_amd.runtime import reset_registry
    from neuralmonkey_amd.variables import VariableStore
    reset_registry()
    with pytest.raises(ValueError, match="Input vector dimension must be positive."):
        StatefulFiller("bad", 0, "vectors")
    with pytest.raises(ValueError, match="Output vector dimension must be positive."):
        StatefulFiller("bad2", 3, "vectors", output_shape=-1)
    plain, same_size, projected = (StatefulFiller("plain", 7, "vectors"), StatefulFiller("same", 7, "vectors", 7),
                                   StatefulFiller("proj", 7, "vectors", output_shape=4))
    store = VariableStore("cpu")
    for part in (plain, same_size, projected):
        part.declare_variables(store)
    assert {n: tuple(store.specs[n].shape) for n in store.names()} == {"proj/dense/kernel": (7, 4), "proj/dense/bias": (4,)}
    assert (plain.output_size, same_size.output_size, projected.output_size) == (7, 7, 4)
    assert projected.input_shapes == {"vectors": [None, 7]}
    ds = D.Dataset("d", {"vectors": [np.arange(7.0), np.ones(7)]})
    fd = projected.feed_dict(ds, train=False)
    assert fd[projected.vector_input].shape == (2, 7) and fd[projected.vector_input].dtype == np.float32
