"""GPU half of test_reference_inis.py: each of the reference's acceptance configs, loaded byte for byte
from tests/golden/reference_tests.tar.gz with the working directory at the bundle root (its own
tests/data), initialises its variables on the MI355X, runs optimizer steps through
``TensorFlowManager.execute`` exactly as the reference's training loop does (learning_utils.py:193-235)
and decodes a validation batch with every runner the file lists.

The last test is the reference's ensemble invariant on its own files (tests/tests_run.sh:41-50): the
model of tests/beamsearch.ini, trained for a few steps and saved, scores the validation data exactly as
tests/beamsearch_ensembles.ini does with that checkpoint loaded into all four of its sessions.
"""
import os

import numpy as np
import pytest

from .test_reference_inis import INIS, load_verbatim, ref_root  # noqa: F401  pylint: disable=unused-import

pytestmark = pytest.mark.gpu


def _feedables(model):
    return set.union(*[r.feedables for r in model.runners + model.trainers])


@pytest.mark.parametrize("name", INIS)
def test_reference_ini_trains_and_decodes(dev, ref_root, name):      # noqa: F811
    model = load_verbatim(ref_root, name, device=str(dev), seed=1234)
    tfm = model.tf_manager
    feedables = _feedables(model)
    scheme = getattr(model.train_dataset, "batching", None)
    batches = []
    for batch in model.train_dataset.batches() if scheme is not None and scheme.batch_size else \
            model.train_dataset.batches(_scheme(model.batch_size)):
        batches.append(batch)
        if len(batches) == 3:
            break
    assert len(batches) == 3
    step0 = tfm.sessions[0].global_step
    losses = []
    if len(tfm.sessions) != 1:
        # beamsearch_ensembles.ini (num_sessions=4) is a neuralmonkey-run config (tests/tests_run.sh:41-50): the
        # reference's trainer refuses several sessions with this very error (trainers/generic_trainer.py:25-27)
        with pytest.raises(ValueError, match="single session"):
            tfm.execute(batches[0], feedables, model.trainers, train=True)
        batches = []
    for batch in batches:
        res = tfm.execute(batch, feedables, model.trainers, train=True)
        assert len(res) == len(model.trainers)
        for r in res:
            assert r.losses and all(np.isfinite(v) for v in r.losses.values()), r.losses
        losses.append(sum(res[0].losses.values()))
    # DelayedUpdateTrainer (transformer.ini): one update every 2nd batch
    assert tfm.sessions[0].global_step > step0 or name in ("transformer", "beamsearch_ensembles")
    val_ds = model.val_dataset[0] if isinstance(model.val_dataset, list) else model.val_dataset     # bahdanau.ini lists two
    val = next(val_ds.batches() if getattr(val_ds, "batching", None) is not None
               and val_ds.batching.batch_size else val_ds.batches(_scheme(model.batch_size)))
    out = tfm.execute(val, feedables, model.runners, compute_losses=True)
    assert len(out) == len(model.runners)
    for runner, result in zip(model.runners, out):
        series = runner.output_series
        rows = result.outputs[series] if isinstance(result.outputs, dict) else result.outputs
        assert len(rows) == len(val), (series, len(rows), len(val))
    # every evaluated series of the file is produced by one of its runners (the evaluators themselves are
    # host control plane: placeholders)
    produced = {r.output_series for r in model.runners}
    assert {item[0] for item in model.evaluation} <= produced


def test_coverage_attention_variant_of_bahdanau_ini_trains_and_decodes(dev, ref_root):     # noqa: F811
    """tests/bahdanau.ini with its attention section switched to attention.CoverageAttention (coverage.py:19-36 takes
    the same arguments + max_fertility): the same three optimizer steps and decoding runs."""
    with open(os.path.join(ref_root, "tests", "bahdanau.ini")) as fh:
        text = fh.read()
    with open(os.path.join(ref_root, "tests", "bahdanau_coverage.ini"), "w") as fh:
        fh.write(text.replace("class=attention.Attention", "class=attention.CoverageAttention\nmax_fertility=4"))
    test_reference_ini_trains_and_decodes(dev, ref_root, "bahdanau_coverage")


def _scheme(batch_size):
    from neuralmonkey_amd.dataset import BatchingScheme
    return BatchingScheme(batch_size=batch_size)


def test_reference_ensemble_invariant_on_its_own_configs(dev, ref_root, tmp_path):     # noqa: F811
    single = load_verbatim(ref_root, "beamsearch", device=str(dev), seed=1234)
    tfm = single.tf_manager
    feedables = _feedables(single)
    n = 0
    for batch in single.train_dataset.batches(_scheme(single.batch_size)):
        tfm.execute(batch, feedables, single.trainers, train=True)
        n += 1
        if n == 12:
            break
    ckpt = str(tmp_path / "variables.data.0")
    tfm.save(ckpt)
    val = single.val_dataset.subset(0, 10)                  # tests/test_data_ensembles_*.ini: batch_size=10
    run_feed = set.union(*[r.feedables for r in single.runners])
    one = tfm.execute(val, run_feed, single.runners, compute_losses=True)
    ens = load_verbatim(ref_root, "beamsearch_ensembles", device=str(dev), seed=99)
    assert len(ens.tf_manager.sessions) == 4
    ens.tf_manager.restore([ckpt] * 4)                      # test_data_ensembles_duplicate.ini: variables=[x]*4
    ens_feed = set.union(*[r.feedables for r in ens.runners])
    four = ens.tf_manager.execute(ens.val_dataset.subset(0, 10), ens_feed, ens.runners, compute_losses=True)
    for a, b in zip(one, four):
        assert a.outputs == b.outputs
        ka = [k for k in a.losses if k.endswith("beam_search_score")]
        for k in ka:          # tests_run.sh compares the first 8 characters of the two printed scores
            assert abs(a.losses[k] - b.losses[k]) <= 1e-5 * max(1.0, abs(a.losses[k])), (k, a.losses[k], b.losses[k])


def test_bahdanau_ini_on_the_engine_equals_the_reference_built_model(dev, ref_root):      # noqa: F811
    """tests/bahdanau.ini end to end on both sides: the REFERENCE'S parser, builder, data pipeline and model parts
    produced ``tests/golden/ref_exec/ini_bahdanau.npz`` (first training batch, train_mode False); the product loads
    the same file from the bundle, takes the reference's variables under their own names, and must give the encoder
    states, teacher-forced logits, greedy symbols and the GreedyRunner's sentences of the reference."""
    fixture = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec",
                                   "ini_bahdanau.npz"))
    params = {k[2:]: fixture[k] for k in fixture.files if k.startswith("p/")}
    model = load_verbatim(ref_root, "bahdanau", device=str(dev), seed=1234)
    tfm = model.tf_manager
    store = tfm.sessions[0].store
    assert sorted(store.names()) == sorted(params), set(store.names()) ^ set(params)
    store.load_state_dict(params)
    batch = next(iter(model.train_dataset.batches()))
    runner = model.runners[0]
    dec = runner.decoder
    enc = dec.encoders[0]
    fd = {}
    for part in runner.feedables:
        fd.update(part.feed_dict(batch, train=False))
    out = tfm.sessions[0].run({"enc": enc.temporal_states, "final": enc.output, "train_logits": dec.train_logits,
                               "sym": dec.decoded_symbols, "mask": dec.runtime_mask,
                               "logits": dec.runtime_logits}, fd)

    def close(got, want, what, tol=1e-4):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape, (what, got.shape, want.shape)
        keep = np.abs(want) < 1e8                      # the -1e9 of supress_unk aside
        err = np.abs(got - want)[keep].max()
        assert err <= tol * max(np.abs(want[keep]).max(), 1e-6), "{}: {:.3e}".format(what, err)
    close(out["enc"], fixture["out/enc_states"], "encoder states")
    close(out["final"], fixture["out/enc_output"], "encoder output")
    close(out["train_logits"], fixture["out/train_logits"], "train logits")
    assert np.array_equal(np.asarray(out["sym"]), fixture["out/runtime_symbols"])
    assert np.array_equal(np.asarray(out["mask"]).astype(bool), fixture["out/runtime_mask"])
    close(out["logits"], fixture["out/runtime_logits"], "runtime logits")
    res = tfm.execute(batch, runner.feedables, [runner], compute_losses=True)[0]
    assert [" ".join(s) for s in res.outputs[runner.output_series]] == [str(s) for s in fixture["out/runner_sentences"]]
    want_losses = fixture["out/runner_losses"]
    got_losses = [res.losses["{}/{}".format(runner.output_series, n)] for n in runner.loss_names]
    assert np.allclose(got_losses, want_losses, rtol=1e-5)


def test_beamsearch_ini_on_the_engine_equals_the_reference_built_model(dev, ref_root):      # noqa: F811
    """tests/beamsearch.ini (Transformer + BeamSearchDecoder + beam_search_runner_range) end to end on both sides: the
    fixture ``ini_beamsearch`` is what the REFERENCE'S parser, builder and model parts gave for the first six sentence
    pairs of the file's training data; the product loads the same file from the bundle, takes the reference's
    variables under their own names and must give its encoder states, logits, greedy symbols, beam search and the
    sentences / scores of its rank-1 and rank-2 runners."""
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    fixture = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec",
                                   "ini_beamsearch.npz"))
    params = {k[2:]: fixture[k] for k in fixture.files if k.startswith("p/")}
    model = load_verbatim(ref_root, "beamsearch", device=str(dev), seed=1234)
    tfm = model.tf_manager
    store = tfm.sessions[0].store
    assert sorted(store.names()) == sorted(params), set(store.names()) ^ set(params)
    store.load_state_dict(params)
    unpad = lambda rows, drop: [[str(t) for t in row if str(t) not in drop] for row in rows]
    batch = Dataset("fixture", {"source": unpad(fixture["in/src_tokens"], ("<pad>",)),
                                "target": unpad(fixture["in/tgt_tokens"], ("<pad>", "</s>"))},
                    BatchingScheme(batch_size=int(fixture["in/src_ids"].shape[0])))
    runners = model.runners
    assert [r.rank for r in runners] == [1, 2]
    bdec = runners[0].decoder
    dec = bdec.parent_decoder
    enc = dec.encoders[0]
    feedables = set.union(*[r.feedables for r in runners])
    fd = {}
    for part in feedables:
        fd.update(part.feed_dict(batch, train=False))
    out = tfm.sessions[0].run({"enc": enc.temporal_states, "train_logits": dec.train_logits, "sym": dec.decoded_symbols,
                               "logits": dec.runtime_logits, "bs": bdec.outputs}, fd)

    def close(got, want, what, tol=1e-4):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape, (what, got.shape, want.shape)
        err = np.abs(got - want).max()
        assert err <= tol * max(np.abs(want).max(), 1e-6), "{}: {:.3e}".format(what, err)
    close(out["enc"], fixture["out/enc_states"], "encoder states")
    close(out["train_logits"], fixture["out/train_logits"], "train logits")
    assert np.array_equal(np.asarray(out["sym"]), fixture["out/runtime_symbols"])
    close(out["logits"], fixture["out/runtime_logits"], "runtime logits")
    tok = np.asarray(out["bs"].last_search_step_output.token_ids)
    assert np.array_equal(tok[1:], fixture["out/beam_token_ids"][1:])
    close(np.asarray(out["bs"].last_search_step_output.scores), fixture["out/beam_scores"], "beam scores")
    res = tfm.execute(batch, feedables, runners, compute_losses=False)
    for rank, r in zip((1, 2), res):
        want = [str(s) for s in fixture["out/rank{}_sentences".format(rank)]]
        got = [" ".join(s) for s in r.outputs[runners[rank - 1].output_series]]
        for g, w, first in zip(got, want, fixture["out/beam_token_ids"][1, :, rank - 1]):
            if first != 2:                        # (a hypothesis that starts with </s>: raw ids in the reference)
                assert g == w
        loss_name = "{}/beam_search_score".format(runners[rank - 1].output_series)
        assert abs(r.losses[loss_name] - float(fixture["out/rank{}_loss".format(rank)])) <= \
            1e-4 * abs(float(fixture["out/rank{}_loss".format(rank)]))


def test_factored_ini_on_the_engine_equals_the_reference_built_model(dev, ref_root):      # noqa: F811
    """tests/factored.ini (FactoredEncoder + ScaledDotProdAttention + Decoder) end to end on both sides, as for
    tests/bahdanau.ini: the fixture ``ini_factored`` is the reference-built model on the first six lines of the
    file's training data."""
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    fixture = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec",
                                   "ini_factored.npz"))
    params = {k[2:]: fixture[k] for k in fixture.files if k.startswith("p/")}
    model = load_verbatim(ref_root, "factored", device=str(dev), seed=1234)
    tfm = model.tf_manager
    store = tfm.sessions[0].store
    assert sorted(store.names()) == sorted(params), set(store.names()) ^ set(params)
    store.load_state_dict(params)
    unpad = lambda rows, drop: [[str(t) for t in row if str(t) not in drop] for row in rows]
    batch = Dataset("fixture", {"source": unpad(fixture["in/src_tokens"], ("<pad>",)),
                                "tags": unpad(fixture["in/tag_tokens"], ("<pad>",)),
                                "target": unpad(fixture["in/tgt_tokens"], ("<pad>", "</s>"))},
                    BatchingScheme(batch_size=int(fixture["in/src_ids"].shape[0])))
    runner = model.runners[0]
    dec = runner.decoder
    enc = dec.encoders[0]
    fd = {}
    for part in runner.feedables:
        fd.update(part.feed_dict(batch, train=False))
    out = tfm.sessions[0].run({"enc": enc.temporal_states, "train_logits": dec.train_logits,
                               "sym": dec.decoded_symbols, "logits": dec.runtime_logits}, fd)

    def close(got, want, what, tol=1e-4):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape, (what, got.shape, want.shape)
        err = np.abs(got - want).max()
        assert err <= tol * max(np.abs(want).max(), 1e-6), "{}: {:.3e}".format(what, err)
    close(out["enc"], fixture["out/enc_states"], "encoder states")
    close(out["train_logits"], fixture["out/train_logits"], "train logits")
    assert np.array_equal(np.asarray(out["sym"]), fixture["out/runtime_symbols"])
    close(out["logits"], fixture["out/runtime_logits"], "runtime logits")
    res = tfm.execute(batch, runner.feedables, [runner], compute_losses=True)[0]
    assert [" ".join(s) for s in res.outputs["target"]] == [str(s) for s in fixture["out/runner_sentences"]]
    assert np.allclose([res.losses["target/train_xent"], res.losses["target/runtime_xent"]],
                       fixture["out/runner_losses"], rtol=1e-4)
