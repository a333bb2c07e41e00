"""GPU tests written AFTER round 4's GPU budget was spent: they have never run on a device.  The driver's round-end
suite is `pytest -x`, so they are skipped unless NM_RUN_PENDING=1 -- the first GPU call of the next round runs them
(tools/r05_first_call.sh), fixes what they find, and moves them into the files they belong to
(test_reference_inis_gpu.py, test_background_gpu.py) without the gate.

What they cover:
  * tests/small.ini, tests/post-edit.ini and tests/flat-multiattention.ini on the engine against the models the
    REFERENCE built from those files (fixtures ``ini_small``, ``ini_postedit``, ``ini_flat``; neuralmonkey/config/builder.py:159-176 names, NematusGRU cells,
    conditional GRU decoder; two encoders under dot-product attentions, attention/scaled_dot_product.py:247-400);
  * the early half of the optimizer step (NM_OPT_EARLY, trainers/generic_trainer.py: the decoders' variables are
    updated on a side lane beside the encoders' backward; trainers/generic_trainer.py:136-195 of the reference is the
    arithmetic, which must not change): three steps with it equal three steps without it;
  * the launches in front of the encoder's BPTT loop inside that loop's graph (NM_ENC_BWD_GRAPH): nothing changes;
  * a checkpoint written by the engine for a NematusGRU model holds exactly the reference's variables;
  * the trainers' reported losses against the reference trainers' ``objective_values``;
  * ``StatefulFiller`` (new) as a decoder's encoder.
"""
import os

import numpy as np
import pytest

from .test_reference_inis import load_verbatim, ref_root, reference_variables  # noqa: F401  pylint: disable=unused-import

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("NM_RUN_PENDING") != "1",
                                 reason="never run on a GPU yet: NM_RUN_PENDING=1 (tools/r05_first_call.sh)")]

HERE = os.path.dirname(os.path.abspath(__file__))


def test_small_ini_on_the_engine_equals_the_reference_built_model(dev, ref_root):      # noqa: F811
    fixture = np.load(os.path.join(HERE, "golden", "ref_exec", "ini_small.npz"))
    params = {k[2:]: fixture[k] for k in fixture.files if k.startswith("p/")}
    model = load_verbatim(ref_root, "small", device=str(dev), seed=1234)
    tfm = model.tf_manager
    store = tfm.sessions[0].store
    # the engine's variables + GRUCell.build's unread ones == the reference's graph
    assert sorted(list(store.names()) + list(store.checkpoint_only)) == sorted(params)
    store.load_state_dict(params)
    # the fixture's batch: the first batch of the file's validation data under its bucketed scheme
    val = model.val_dataset[0] if isinstance(model.val_dataset, list) else model.val_dataset
    batch = next(iter(val.batches()))
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
        err = np.abs(got - want).max()
        assert err <= tol * max(np.abs(want).max(), 1e-6), "{}: {:.3e}".format(what, err)
    close(out["enc"], fixture["out/enc_states"], "encoder states")
    close(out["final"], fixture["out/enc_output"], "encoder output")
    close(out["train_logits"], fixture["out/train_logits"], "train logits")
    assert np.array_equal(np.asarray(out["sym"]), fixture["out/runtime_symbols"])
    assert np.array_equal(np.asarray(out["mask"]).astype(bool), fixture["out/runtime_mask"])
    close(out["logits"], fixture["out/runtime_logits"], "runtime logits")
    res = tfm.execute(batch, runner.feedables, [runner], compute_losses=True)[0]
    assert [" ".join(s) for s in res.outputs[runner.output_series]] == [str(s) for s in fixture["out/runner_sentences"]]
    got_losses = [res.losses["{}/{}".format(runner.output_series, n)] for n in runner.loss_names]
    assert np.allclose(got_losses, fixture["out/runner_losses"], rtol=1e-5)


def test_post_edit_ini_on_the_engine_equals_the_reference_built_model(dev, ref_root):      # noqa: F811
    """tests/post-edit.ini end to end on both sides (fixture ``ini_postedit``): two encoders, a multi-head attention
    with keys and values from different encoders plus a one-head attention, borrowed embeddings, the edit-script
    series made while the data load."""
    fixture = np.load(os.path.join(HERE, "golden", "ref_exec", "ini_postedit.npz"))
    params = {k[2:]: fixture[k] for k in fixture.files if k.startswith("p/")}
    model = load_verbatim(ref_root, "post-edit", device=str(dev), seed=1234)
    tfm = model.tf_manager
    store = tfm.sessions[0].store
    assert sorted(store.names()) == sorted(params), set(store.names()) ^ set(params)
    store.load_state_dict(params)
    batch = next(iter(model.train_dataset.batches()))
    runner = model.runners[0]
    dec = runner.decoder
    trans, src = dec.encoders
    fd = {}
    for part in runner.feedables:
        fd.update(part.feed_dict(batch, train=False))
    out = tfm.sessions[0].run({"src": src.temporal_states, "mt": trans.temporal_states, "src_out": src.output,
                               "mt_out": trans.output, "train_logits": dec.train_logits, "sym": dec.decoded_symbols,
                               "mask": dec.runtime_mask, "logits": dec.runtime_logits}, fd)

    def close(got, want, what, tol=1e-4):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape, (what, got.shape, want.shape)
        err = np.abs(got - want).max()
        assert err <= tol * max(np.abs(want).max(), 1e-6), "{}: {:.3e}".format(what, err)
    close(out["src"], fixture["out/src_states"], "source encoder states")
    close(out["mt"], fixture["out/mt_states"], "translation encoder states")
    close(out["src_out"], fixture["out/src_output"], "source encoder output")
    close(out["mt_out"], fixture["out/mt_output"], "translation encoder output")
    close(out["train_logits"], fixture["out/train_logits"], "train logits")
    assert np.array_equal(np.asarray(out["sym"]), fixture["out/runtime_symbols"])
    assert np.array_equal(np.asarray(out["mask"]).astype(bool), fixture["out/runtime_mask"])
    close(out["logits"], fixture["out/runtime_logits"], "runtime logits")
    res = tfm.execute(batch, runner.feedables, [runner], compute_losses=True)[0]
    scripts = res.outputs[runner.output_series]
    assert [" ".join(s) for s in scripts] == [str(s) for s in fixture["out/runner_sentences"]]
    got_losses = [res.losses["{}/{}".format(runner.output_series, n)] for n in runner.loss_names]
    assert np.allclose(got_losses, fixture["out/runner_losses"], rtol=1e-5)
    (_, post), = model.postprocess
    rebuilt = post({"translated": list(batch.get_series("translated"))}, {"edits": scripts})
    assert [" ".join(r) for r in rebuilt] == fixture["out/postprocessed"].tolist()


def test_flat_multiattention_ini_on_the_engine_equals_the_reference_built_model(dev, ref_root):      # noqa: F811
    """tests/flat-multiattention.ini end to end on both sides (fixture ``ini_flat``): the four decoders' teacher-forced
    logits, greedy loops and runner outputs, and the RNN beam search with its rank-2 runner."""
    fixture = np.load(os.path.join(HERE, "golden", "ref_exec", "ini_flat.npz"))
    params = {k[2:]: fixture[k] for k in fixture.files if k.startswith("p/")}
    model = load_verbatim(ref_root, "flat-multiattention", device=str(dev), seed=1234)
    tfm = model.tf_manager
    store = tfm.sessions[0].store
    assert sorted(store.names()) == sorted(params), set(store.names()) ^ set(params)
    store.load_state_dict(params)
    batch = next(iter(model.train_dataset.batches()))
    assert len(batch) == 1
    feedables = set.union(*[r.feedables for r in model.runners])
    fd = {}
    for part in feedables:
        fd.update(part.feed_dict(batch, train=False))

    def close(got, want, what, tol=1e-4):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape, (what, got.shape, want.shape)
        err = np.abs(got - want).max()
        assert err <= tol * max(np.abs(want).max(), 1e-6), "{}: {:.3e}".format(what, err)
    for runner in model.runners[:4]:
        dec = runner.decoder
        tag = dec.name[len("decoder_"):]
        out = tfm.sessions[0].run({"train_logits": dec.train_logits, "sym": dec.decoded_symbols,
                                   "mask": dec.runtime_mask, "logits": dec.runtime_logits}, fd)
        pre = "out/{}/".format(tag)
        close(out["train_logits"], fixture[pre + "train_logits"], tag + " train logits")
        assert np.array_equal(np.asarray(out["sym"]), fixture[pre + "runtime_symbols"]), tag
        assert np.array_equal(np.asarray(out["mask"]).astype(bool), fixture[pre + "runtime_mask"]), tag
        close(out["logits"], fixture[pre + "runtime_logits"], tag + " runtime logits")
    results = tfm.execute(batch, feedables, model.runners, compute_losses=True)
    for runner, res in zip(model.runners[:4], results):
        tag = runner.decoder.name[len("decoder_"):]
        assert [" ".join(s) for s in res.outputs[runner.output_series]] == \
            [str(s) for s in fixture["out/{}/runner_sentences".format(tag)]]
        got = [res.losses["{}/{}".format(runner.output_series, n)] for n in runner.loss_names]
        assert np.allclose(got, fixture["out/{}/runner_losses".format(tag)], rtol=1e-5)
    beam_runner, beam = model.runners[4], results[4]
    assert [" ".join(s) for s in beam.outputs[beam_runner.output_series]] == \
        [str(s) for s in fixture["out/beam_runner_sentences"]]
    assert abs(beam.losses[beam_runner.output_series + "/beam_search_score"] - float(fixture["out/beam_runner_loss"])) \
        <= 1e-5 * abs(float(fixture["out/beam_runner_loss"]))


@pytest.mark.parametrize("name,section", [("bahdanau", "greedy_trainer"), ("bahdanau", "trainer1"),
                                          ("post-edit", "trainer")])
def test_trainer_losses_on_the_engine_equal_the_reference_trainers(dev, ref_root, name, section):      # noqa: F811
    """``objective_values`` of the reference's trainers on their first training batch (fixture
    ``ini_trainer_objectives``: decoder cost, L1, L2 over the reference's choice of variables, train_mode False)
    against the losses the engine's trainer reports for the same variables and batch."""
    import json
    from .test_reference_inis import _product_trainers
    z = np.load(os.path.join(HERE, "golden", "ref_exec", "ini_trainer_objectives.npz"))
    want = json.loads(str(z["out/" + name]))[section]
    model = load_verbatim(ref_root, name, device=str(dev), seed=3)
    tfm = model.tf_manager
    store = tfm.sessions[0].store
    store.load_state_dict({n: z["vars/{}/{}".format(name, n)] for n in store.names()})
    trainer = _product_trainers(model, name)[section]
    batch = next(iter(model.train_dataset.batches()))
    res = tfm.execute(batch, trainer.feedables, [trainer], train=False)[0]
    loss, l1, l2 = want["objective_values"]
    keys = list(res.losses)
    assert keys[-2:] == ["L1", "L2"] and keys[0] == want["objective_names"][0]
    assert abs(res.losses[keys[0]] - loss) <= 1e-4 * abs(loss)
    assert abs(res.losses["L1"] - l1) <= 1e-5 * l1 and abs(res.losses["L2"] - l2) <= 1e-5 * l2


def test_small_ini_checkpoint_holds_the_references_variables(dev, ref_root, tmp_path):      # noqa: F811
    """``variables.data`` as the engine writes it for tests/small.ini: the names and shapes the reference's Saver
    would look for (fixture ``ini_variables``), GRUCell.build's unread variables included; read back, it restores
    every variable bit for bit."""
    import torch
    from neuralmonkey_amd import tf_bundle
    model = load_verbatim(ref_root, "small", device=str(dev), seed=7)
    store = model.tf_manager.sessions[0].store
    prefix = str(tmp_path / "variables.data")
    model.tf_manager.checkpoint_format = "tf"
    model.tf_manager.save(prefix)
    written = tf_bundle.read_bundle(prefix)
    theirs = {n: tuple(s) for n, s in reference_variables("small")["variables"]}
    mine = {n: tuple(a.shape) for n, a in written.items() if n != "global_step" and not n.endswith(("/Adam", "/Adam_1"))
            and n not in ("beta1_power", "beta2_power")}
    assert mine == theirs
    before = {n: store[n].clone() for n in store.names()}
    store.theta.zero_()
    model.tf_manager.restore(prefix)
    assert all(torch.equal(store[n], before[n]) for n in store.names())


def _three_steps(dev, monkeypatch, early, prologue_in_graph=False, steps=3):
    import torch
    from neuralmonkey_amd import synthetic
    from neuralmonkey_amd.encoders import recurrent
    from neuralmonkey_amd.trainers import generic_trainer
    from oracle import nm_oracle as O
    monkeypatch.setattr(generic_trainer, "OPT_EARLY", early)
    monkeypatch.setattr(recurrent, "BWD_PROLOGUE_IN_GRAPH", prologue_in_graph)
    model = synthetic.build_translation_model(vocab_src=2000, vocab_tgt=2000, emb=64, rnn=64, max_len=24,
                                              beam_size=0, device=str(dev), l2_weight=1e-6, clip_norm=1.0)
    sess = model.tf_manager.sessions[0]
    sess.store.load_state_dict(O.init_params(seed=3, vocab_src=2000, vocab_tgt=2000, emb=64, rnn=64, std=0.1))
    ds = synthetic.synthetic_dataset(seed=4, batch=32, src_len=24, tgt_len=20, vocab=2000, ragged=True)
    losses = []
    for _ in range(steps):
        res = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
        losses.append([float(res.losses[k]) for k in ("decoder - cost", "L1", "L2")])
    torch.cuda.synchronize()
    halves = [key for key in model.trainer._tables if isinstance(key, tuple)]
    return losses, {n: sess.store[n].cpu().numpy().copy() for n in sess.store.names()}, halves, sess.global_step


def test_early_optimizer_half_changes_no_parameter(dev, monkeypatch):
    """Clipping, the regulariser's gradient and Adam are per tensor: splitting the variables into two launches
    changes no update.  Only the reported L1 / L2 sums are added in another order."""
    l_early, p_early, halves, step_early = _three_steps(dev, monkeypatch, True)
    l_plain, p_plain, none, step_plain = _three_steps(dev, monkeypatch, False)
    assert len(halves) == 2 and not none and step_early == step_plain == 3
    assert np.allclose(l_early, l_plain, rtol=1e-6, atol=0.0), (l_early, l_plain)
    for name, want in p_plain.items():
        assert np.array_equal(p_early[name], want), name


def test_encoder_backward_prologue_inside_the_loop_graph_changes_nothing(dev, monkeypatch):
    """NM_ENC_BWD_GRAPH (encoders/recurrent.py): the same launches in the same order, ten of them moved into the BPTT
    loop's HIP graph -- eager pass, capture and three replays against the plain path, bit for bit."""
    l_graph, p_graph, _, _ = _three_steps(dev, monkeypatch, False, prologue_in_graph=True, steps=5)
    l_plain, p_plain, _, _ = _three_steps(dev, monkeypatch, False, prologue_in_graph=False, steps=5)
    assert l_graph == l_plain
    for name, want in p_plain.items():
        assert np.array_equal(p_graph[name], want), name


def test_stateful_filler_under_a_decoder(dev):
    """``StatefulFiller`` (encoders/numpy_stateful_filler.py:16-72) with its dense projection as the only encoder of an
    RNN decoder without attention: the decoder's initial state comes from the projected vectors, and a training step
    gives the projection's gradients that float64 autograd gives (``initial_state = dense(output)``,
    decoders/encoder_projection.py:47-73)."""
    import torch
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    from neuralmonkey_amd.decoders import Decoder
    from neuralmonkey_amd.encoders.numpy_stateful_filler import StatefulFiller
    from neuralmonkey_amd.runtime import reset_registry
    from neuralmonkey_amd.tf_manager import TensorFlowManager
    from neuralmonkey_amd.trainers.cross_entropy_trainer import CrossEntropyTrainer
    from neuralmonkey_amd.vocabulary import Vocabulary
    reset_registry()
    rng = np.random.default_rng(5)
    vocab = Vocabulary(["w{}".format(i) for i in range(11)])
    filler = StatefulFiller("vec", 7, "vectors", output_shape=5)
    dec = Decoder(encoders=[filler], vocabulary=vocab, data_id="target", name="decoder", max_output_len=4,
                  embedding_size=6, rnn_size=6, dropout_keep_prob=1.0)
    trainer = CrossEntropyTrainer(decoders=[dec])
    tfm = TensorFlowManager(num_sessions=1, num_threads=1, device=str(dev), seed=3)
    tfm.initialize_sessions()
    store = tfm.sessions[0].store
    vectors = [rng.normal(size=7).astype(np.float32) for _ in range(3)]
    ds = Dataset("d", {"vectors": vectors, "target": [["w1", "w2"], ["w3"], ["w4", "w5", "w6"]]},
                 BatchingScheme(batch_size=3))
    fd = {}
    for part in trainer.feedables:
        fd.update(part.feed_dict(ds, train=False))
    out = tfm.sessions[0].run({"enc": filler.output}, fd)["enc"]
    w, b = store["vec/dense/kernel"].cpu().numpy(), store["vec/dense/bias"].cpu().numpy()
    assert np.allclose(np.asarray(out), np.stack(vectors) @ w + b, atol=1e-5)
    before = w.copy()
    tfm.execute(ds, trainer.feedables, [trainer], train=True)
    grad = store.g("vec/dense/kernel").cpu().numpy()
    assert np.abs(grad).max() > 0 and not np.array_equal(store["vec/dense/kernel"].cpu().numpy(), before)
    # d loss / d kernel = vectors^T . d loss / d output: rank <= batch size, rows in the span of the fed vectors
    assert np.linalg.matrix_rank(grad.astype(np.float64), tol=1e-6 * np.abs(grad).max()) <= 3


def test_train_logprobs_is_the_log_softmax_of_the_train_logits(dev):
    """``AutoregressiveDecoder.train_logprobs`` (autoregressive.py:288-290; new here, for runners that fetch it by
    name): tf.nn.log_softmax of the teacher-forced logits."""
    import torch
    from neuralmonkey_amd import synthetic
    model = synthetic.build_translation_model(vocab_src=300, vocab_tgt=300, emb=32, rnn=32, max_len=12, beam_size=0,
                                              device=str(dev))
    ds = synthetic.synthetic_dataset(seed=4, batch=6, src_len=9, tgt_len=8, vocab=300, ragged=True)
    dec = model.decoder
    fd = {}
    for part in model.trainer.feedables:
        fd.update(part.feed_dict(ds, train=False))
    out = model.tf_manager.sessions[0].run({"logits": dec.train_logits, "logprobs": dec.train_logprobs}, fd)
    want = torch.log_softmax(torch.as_tensor(np.asarray(out["logits"])).double(), -1).numpy()
    assert np.abs(np.asarray(out["logprobs"]) - want).max() < 1e-5
