"""PlainRunner / XentRunner / MultitaskTrainer (runners/plain_runner.py, runners/xent_runner.py,
trainers/multitask_trainer.py of the reference) over the attention decoder: their device-side
fetches (``decoded``: argmax with <pad> excluded; ``train_xents``: [B,T] masked cross entropies)
against the same quantities derived on the host from the logits the decoder already exposes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def model(dev):
    from neuralmonkey_amd import synthetic
    return synthetic.build_translation_model(vocab_src=60, vocab_tgt=60, emb=8, rnn=8, max_len=9, beam_size=0,
                                             device=str(dev), seed=11)


def _log_softmax(x):
    x = x - x.max(-1, keepdims=True)
    return x - np.log(np.exp(x).sum(-1, keepdims=True))


def test_plain_and_xent_runner(model):
    from neuralmonkey_amd import synthetic
    from neuralmonkey_amd.runners import PlainRunner, XentRunner
    ds = synthetic.synthetic_dataset(seed=2, batch=6, src_len=7, tgt_len=6, vocab=60, ragged=True)
    dec, tfm = model.decoder, model.tf_manager
    plain, xent = PlainRunner("target_plain", dec), XentRunner("target_xent", dec)
    feedables = plain.feedables | xent.feedables
    res_plain, res_xent = tfm.execute(ds, feedables, [plain, xent])
    sess = tfm.sessions[0]
    fd = {}
    for part in feedables:
        fd.update(part.feed_dict(ds, train=False))
    out = sess.run({"logits": dec.runtime_logits, "train_logits": dec.train_logits, "loss": dec.train_loss,
                    "tgt": dec.train_inputs, "mask": dec.train_mask}, fd)
    # decoded = argmax(logits[:, :, 1:]) + 1 (autoregressive.py:341-349), cut at </s> by the vocabulary
    want_ids = out["logits"][:, :, 1:].argmax(-1) + 1
    want = dec.vocabulary.vectors_to_sentences(list(want_ids))
    assert res_plain.outputs["target_plain"] == want
    assert set(res_plain.losses) == {"target_plain/train_loss", "target_plain/runtime_loss"}
    # train_xents [B,T] = -log p(target) * mask (autoregressive.py:289-310)
    lp = _log_softmax(out["train_logits"].astype(np.float64))                     # [T,B,V]
    t, b = out["tgt"].shape
    want_x = -(lp[np.arange(t)[:, None], np.arange(b)[None, :], out["tgt"]]) * out["mask"]
    got_x = np.asarray(res_xent.outputs["target_xent"])
    assert got_x.shape == (b, t)
    assert np.abs(got_x - want_x.T).max() < 1e-4
    assert abs(got_x.sum() / out["mask"].sum() - float(out["loss"])) < 1e-4
    assert abs(res_xent.losses["target_xent/xent"] - got_x.mean()) < 1e-5


def test_tensor_and_representation_runner(model):
    from neuralmonkey_amd import synthetic
    from neuralmonkey_amd.runners import RepresentationRunner, TensorRunner
    ds = synthetic.synthetic_dataset(seed=4, batch=5, src_len=7, tgt_len=6, vocab=60, ragged=True)
    enc, dec, tfm = model.encoder, model.decoder, model.tf_manager
    rep = RepresentationRunner("encoded", enc)
    both = TensorRunner("tensors", modelparts=[enc, dec], tensors=["temporal_states", "decoded_symbols"],
                        batch_dims=[0, 1], tensors_by_name=[], batch_dims_by_name=[])
    feedables = rep.feedables | both.feedables
    assert dec in both.feedables and enc.input_sequence in rep.feedables
    res_rep, res_both = tfm.execute(ds, feedables, [rep, both], compute_losses=False)
    sess = tfm.sessions[0]
    fd = {}
    for part in feedables:
        fd.update(part.feed_dict(ds, train=False))
    want = sess.run({"out": enc.output, "states": enc.temporal_states, "sym": dec.decoded_symbols}, fd)
    vectors = res_rep.outputs["encoded"]
    assert len(vectors) == 5 and res_rep.losses == {}
    assert np.array_equal(np.stack(vectors), want["out"])
    rows = res_both.outputs["tensors"]
    assert len(rows) == 5 and set(rows[0]) == {"encoder/temporal_states", "decoder/decoded_symbols"}
    for b, row in enumerate(rows):                     # batch axis 0 and batch axis 1 both end up per example
        assert np.array_equal(row["encoder/temporal_states"], want["states"][b])
        assert np.array_equal(row["decoder/decoded_symbols"], want["sym"][:, b])
    with pytest.raises(ValueError):
        TensorRunner("x", modelparts=[enc], tensors=["output", "temporal_states"], batch_dims=[0, 0],
                     tensors_by_name=[], batch_dims_by_name=[])
    with pytest.raises(ValueError):
        TensorRunner("x", modelparts=[enc, dec], tensors=["output", "decoded_symbols"], batch_dims=[0, 1],
                     tensors_by_name=[], batch_dims_by_name=[], single_tensor=True)


def test_multitask_trainer_switches_tasks(model):
    from neuralmonkey_amd import synthetic
    from neuralmonkey_amd.trainers import CrossEntropyTrainer, MultitaskTrainer
    ds = synthetic.synthetic_dataset(seed=3, batch=6, src_len=7, tgt_len=6, vocab=60, ragged=True)
    dec, tfm = model.decoder, model.tf_manager
    a = CrossEntropyTrainer(decoders=[dec], l2_weight=0.0, clip_norm=1.0)
    b = CrossEntropyTrainer(decoders=[dec], l2_weight=1.0, clip_norm=1.0)
    multi = MultitaskTrainer([a, b])
    assert multi.feedables >= a.feedables | b.feedables
    order = [multi.get_executable(True, True, 1).executor for _ in range(4)]
    assert order == [a, b, a, b] and multi.trainer_idx == 0
    losses = [tfm.execute(ds, multi.feedables, [multi], train=True)[0].losses for _ in range(4)]
    assert all(set(l) == {"decoder - cost", "L1", "L2"} for l in losses)
    assert losses[3]["decoder - cost"] < losses[0]["decoder - cost"]
    assert multi.trainer_idx == 0 and tfm.sessions[0].global_step == 4
    with pytest.raises(ValueError):
        MultitaskTrainer([])
