"""Host input pipeline: pre-indexed datasets, the prefetching worker, feed-dict
caching, BPE processors against vectors produced by the reference's own BPE code
(tests/golden/make_bpe_golden.py -> bpe_golden.json).  CPU-only except the last test."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "bpe_golden.json")


def test_bpe_preprocessor_matches_the_reference_vectors(tmp_path):
    from neuralmonkey_amd.processors.bpe import BPEPostprocessor, BPEPreprocessor
    with open(GOLDEN, encoding="utf-8") as handle:
        golden = json.load(handle)
    merges = tmp_path / "merges.bpe"
    merges.write_text("\n".join(golden["merges"]) + "\n", encoding="utf-8")
    pre, post = BPEPreprocessor(str(merges), golden["separator"]), BPEPostprocessor(golden["separator"])
    segmented = 0
    for case in golden["cases"]:
        out = pre(case["input"])
        assert out == case["output"], case["input"]
        segmented += out != case["input"]
        if all(case["input"]):                       # round trip (empty tokens do not survive a join/split)
            assert post([out])[0] == case["input"]
    assert segmented > 30
    assert post([["ab@@", "c", "d@@", "e@@", "f"]]) == [["abc", "def"]]


def _model(device="cpu"):
    from neuralmonkey_amd.attention import Attention
    from neuralmonkey_amd.decoders import Decoder
    from neuralmonkey_amd.encoders import SentenceEncoder
    from neuralmonkey_amd.runtime import reset_registry
    from neuralmonkey_amd.synthetic import synthetic_vocabulary
    from neuralmonkey_amd.tf_manager import TensorFlowManager
    from neuralmonkey_amd.trainers import CrossEntropyTrainer
    reset_registry()
    vocab = synthetic_vocabulary(60)
    enc = SentenceEncoder(name="enc", vocabulary=vocab, data_id="source", embedding_size=8, rnn_size=4,
                          max_input_len=12)
    att = Attention(name="att", encoder=enc)
    dec = Decoder(encoders=[enc], vocabulary=vocab, data_id="target", name="dec", max_output_len=12,
                  embedding_size=8, rnn_size=8, attentions=[att])
    trainer = CrossEntropyTrainer(decoders=[dec], l2_weight=0.0, clip_norm=1.0)
    tfm = TensorFlowManager(num_sessions=1, num_threads=1, device=device, seed=5)
    tfm.initialize_sessions()
    return dict(vocab=vocab, enc=enc, dec=dec, trainer=trainer, tfm=tfm)


def _dataset(n=40, seed=0, batching=None):
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    rng = np.random.default_rng(seed)
    words = lambda k: ["w{}".format(int(i)) for i in rng.integers(0, 56, size=int(k))]
    src = [words(k) for k in rng.integers(1, 12, size=n)]
    tgt = [words(k) + (["never-seen"] if i % 7 == 0 else []) for i, k in enumerate(rng.integers(1, 11, size=n))]
    return Dataset("d", {"source": src, "target": tgt},
                   batching or BatchingScheme(bucket_boundaries=[4, 8], bucket_batch_sizes=[6, 5, 4]))


def test_preindexed_dataset_feeds_the_same_ids():
    from neuralmonkey_amd.input_pipeline import preindex, series_vocabularies
    from neuralmonkey_amd.tf_manager import _feed_dicts
    m = _model()
    feedables = m["trainer"].feedables
    assert set(series_vocabularies(feedables)) == {"source", "target"}
    ds = _dataset()
    pre = preindex(ds, feedables)
    assert all(isinstance(s, np.ndarray) and s.dtype == np.int32 for s in pre.get_series("source"))
    n = 0
    for b_str, b_int in zip(ds.batches(), pre.batches()):        # same bucketing on both
        fd_s, fd_i = _feed_dicts(b_str, feedables, train=True), _feed_dicts(b_int, feedables, train=True)
        assert set(fd_s) == set(fd_i)
        for key, val in fd_s.items():
            if isinstance(val, np.ndarray):
                assert val.dtype == fd_i[key].dtype and np.array_equal(val, fd_i[key]), key.name
            else:
                assert val == fd_i[key]
        n += 1
    assert n >= 7
    assert list(pre.get_series("target"))[0][-1] == 3          # OOV -> <unk> (vocabulary.py:224-244)


def test_feed_dict_is_cached_on_the_batch_object():
    from neuralmonkey_amd.tf_manager import _feed_dicts
    m = _model()
    feedables = m["trainer"].feedables
    batch = next(_dataset().batches())
    a, b = _feed_dicts(batch, feedables, train=True), _feed_dicts(batch, feedables, train=True)
    assert a is not b and all(a[k] is b[k] for k in a)
    c = _feed_dicts(batch, feedables, train=False)
    assert any(c[k] is not a[k] for k in a if isinstance(a[k], np.ndarray)) or c != a
    m2 = _model()                          # other model parts, same batch object: no stale placeholders
    d = _feed_dicts(batch, m2["trainer"].feedables, train=True)
    assert not set(d) & set(a)


def test_prefetcher_keeps_order_and_uploads_ahead():
    from neuralmonkey_amd.input_pipeline import Prefetcher, preindex
    from neuralmonkey_amd.tf_manager import _feed_dicts
    m = _model()
    feedables = m["trainer"].feedables
    ds = preindex(_dataset(60, seed=3), feedables)
    want = [list(map(len, b.get_series("source"))) for b in ds.batches()]
    sess = m["tfm"].sessions[0]
    got = []
    for batch in Prefetcher(m["tfm"], feedables, train=True, depth=2).iterate(ds.batches()):
        got.append(list(map(len, batch.get_series("source"))))
        fd = _feed_dicts(batch, feedables, train=True)
        # every array a model part will ask for is already in the session's device cache
        ids = fd[m["enc"].input_sequence.input_factors[0]]
        key = (id(ids), None, __import__("torch").int32)
        assert key in sess._h2d and sess._h2d[key][0] is ids
    assert got == want

    def boom():
        yield next(ds.batches())
        raise RuntimeError("reader failed")
    with pytest.raises(RuntimeError, match="reader failed"):
        list(Prefetcher(m["tfm"], feedables, train=True).iterate(boom()))
    with pytest.raises(ValueError):
        Prefetcher(m["tfm"], feedables, depth=0)


@pytest.mark.gpu
def test_training_through_the_prefetcher_equals_direct_feeding(dev):
    """Same batches, same seed: losses of a run fed through pre-indexing + the prefetching worker are
    the losses of a run fed batch by batch from strings (the copy-stream events order every upload
    before its use)."""
    from neuralmonkey_amd.input_pipeline import Prefetcher, preindex

    def run(pipeline):
        m = _model(str(dev))
        feedables = m["trainer"].feedables
        ds = _dataset(80, seed=1)
        if pipeline:
            batches = Prefetcher(m["tfm"], feedables, train=True, depth=3).iterate(preindex(ds, feedables).batches())
        else:
            batches = ds.batches()
        losses = []
        for _ in range(2):                   # second epoch re-uses cached feed dicts / resident arrays
            for batch in batches:
                losses.append(m["tfm"].execute(batch, feedables, [m["trainer"]], train=True)[0].losses["dec - cost"])
            batches = (Prefetcher(m["tfm"], feedables, train=True).iterate(preindex(ds, feedables).batches())
                       if pipeline else ds.batches())
        return np.asarray(losses)
    direct, piped = run(False), run(True)
    assert direct.shape == piped.shape and direct.size > 20
    assert np.abs(direct - piped).max() <= 1e-6 * np.abs(direct).max()
