"""Training-step parity: hand-written HIP backward + fused clip/Adam against
torch-CPU autograd of the oracle's forward (oracle/torch_ref.py) on the same
weights and batch.  Tolerances: gradients 1e-3 of the tensor's max magnitude
(fp32 accumulation over T steps), loss 1e-4 relative."""
import numpy as np
import pytest
import torch

from oracle import nm_oracle as O
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu


def _build(dev, vocab, emb, rnn, batch, slen, tlen, ragged, seed=11, l1=0.0, l2=1e-8, clip=1.0):
    from neuralmonkey_amd import synthetic
    model = synthetic.build_translation_model(vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn,
                                              max_len=max(slen, tlen), beam_size=0, device=str(dev),
                                              l2_weight=l2, clip_norm=clip)
    model.trainer.l1_weight = l1
    params = O.init_params(seed=seed, vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, std=0.1)
    model.tf_manager.sessions[0].store.load_state_dict(params)
    ds = synthetic.synthetic_dataset(seed=seed + 1, batch=batch, src_len=slen, tgt_len=tlen, vocab=vocab,
                                     ragged=ragged)
    src = O.pad_ids([list(s) for s in ds.get_series("source")], max(slen, tlen))
    tgt = O.pad_ids([list(s) for s in ds.get_series("target")], max(slen, tlen), add_end_symbol=True)
    return model, params, ds, src, np.ascontiguousarray(tgt.T)


@pytest.mark.parametrize("vocab,emb,rnn,batch,slen,tlen,ragged,l1,l2,clip", [
    (64, 12, 12, 5, 7, 6, True, 0.0, 1e-8, 1.0),
    (300, 32, 32, 12, 15, 11, True, 1e-4, 1e-3, 0.05),
    (1000, 64, 64, 16, 20, 16, False, 0.0, 0.0, None),
])
def test_gradients_and_adam_step_match_autograd(dev, vocab, emb, rnn, batch, slen, tlen, ragged, l1, l2, clip):
    model, params, ds, src, tgt = _build(dev, vocab, emb, rnn, batch, slen, tlen, ragged, l1=l1, l2=l2, clip=clip)
    sess = model.tf_manager.sessions[0]
    store = sess.store

    tp = TR.to_torch(params)
    ref_loss, ref_l1, ref_l2, ref_g = TR.train_step_grads(tp, src, tgt, l1_weight=l1, l2_weight=l2)

    res = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
    assert set(res.losses) == {"decoder - cost", "L1", "L2"}
    assert res.size == batch
    assert abs(res.losses["decoder - cost"] - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    assert abs(res.losses["L1"] - float(ref_l1)) < 1e-4 * float(ref_l1)
    assert abs(res.losses["L2"] - float(ref_l2)) < 1e-4 * float(ref_l2)

    worst = {}
    for name in store.names():
        got = store.g(name).cpu().numpy().reshape(-1)
        want = ref_g[name].numpy().reshape(-1)
        if name.endswith("attn_bias"):
            # d/d(attn_bias) is identically 0 (softmax shift invariance): both sides hold rounding noise
            assert abs(got[0]) < 1e-5 and abs(want[0]) < 1e-5
            continue
        scale = max(np.abs(want).max(), 1e-6)   # floor: d/d(attn_bias) is identically 0 (softmax shift invariance)
        worst[name] = float(np.abs(got - want).max() / scale)
    bad = {k: v for k, v in worst.items() if v > 1e-3}
    assert not bad, "gradient mismatch: {}".format(bad)

    # Adam moments after step 1 pin the clipped gradient: m = (1-b1)*g_clip, v = (1-b2)*g_clip^2
    m, v = store.ensure_adam()
    for name in store.names():
        if name.endswith("attn_bias"):
            continue
        g = ref_g[name]
        if clip:
            g = g * (clip / max(float(g.norm()), clip))
        spec = store.specs[name]
        got_m = m[spec.offset:spec.offset + spec.size].cpu().numpy()
        want_m = (0.1 * g).numpy().reshape(-1)
        assert np.abs(got_m - want_m).max() <= 1e-3 * max(np.abs(want_m).max(), 1e-7), name
    # parameters moved by at most lr (|update| <= lr_t * ... ~ lr at step 1) and in the right direction
    for name in ("decoder/state_to_word_W", "attention/attn_similarity_v"):
        before = params[name].reshape(-1)
        after = store[name].cpu().numpy().reshape(-1)
        g = ref_g[name].numpy().reshape(-1)
        big = np.abs(g) > 1e-3 * np.abs(g).max()
        assert np.all(np.sign(before - after)[big] == np.sign(g)[big])
        assert np.abs(before - after).max() <= 1.01e-4


def test_three_steps_track_the_reference_optimizer(dev):
    """Loss trajectory of 3 optimizer steps == torch autograd + clip + Adam."""
    model, params, ds, src, tgt = _build(dev, 200, 32, 32, 8, 10, 9, True, l2=1e-8, clip=1.0)
    tp = TR.to_torch(params)
    m = {k: torch.zeros_like(v) for k, v in tp.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in tp.items()}
    ref_losses, got_losses = [], []
    for step in range(1, 4):
        loss, _, _, grads = TR.train_step_grads(tp, src, tgt, l1_weight=0.0, l2_weight=1e-8)
        ref_losses.append(float(loss))
        TR.clip_and_adam(tp, grads, m, v, step, 1.0)
        res = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
        got_losses.append(res.losses["decoder - cost"])
    assert np.allclose(got_losses, ref_losses, rtol=2e-4), (got_losses, ref_losses)
    assert got_losses[2] < got_losses[0]


def test_training_then_greedy_share_one_run(dev):
    """Trainer and runner executed in one ``execute`` call (logging_period path,
    learning_utils.py:110-125): one forward result is shared, both results come back."""
    model, params, ds, src, tgt = _build(dev, 64, 12, 12, 5, 7, 6, True)
    out = model.tf_manager.execute(ds, model.trainer.feedables | model.greedy_runner.feedables,
                                   [model.trainer, model.greedy_runner], train=True)
    assert out[0].losses["decoder - cost"] > 0
    assert len(out[1].outputs["target"]) == 5


def test_delayed_update_trainer_accumulates_and_averages(dev):
    """DelayedUpdateTrainer (trainers/delayed_update_trainer.py:142-204): no update before the N-th batch; the N-th
    applies the MEAN of the N accumulated gradients (each ``raw_gradients`` of generic_trainer.py:136-142, i.e. with
    the regulariser's share) through clip_by_norm and Adam.  Six DISTINCT batches = two updates, against the oracle's
    autograd -> mean -> clip -> Adam on the same batches.  Adam's first moment after update 1 is 0.1 x the clipped
    mean gradient, so a sum instead of a mean shows there (x3 on every tensor the clip leaves alone) -- the
    parameters alone could not tell (Adam's first step is scale invariant)."""
    from neuralmonkey_amd import synthetic
    from neuralmonkey_amd.trainers import DelayedUpdateTrainer
    from neuralmonkey_amd.trainers.objective import CostObjective
    vocab, n, l2, clip = 64, 3, 1e-3, 0.02
    model, params, _, _, _ = _build(dev, vocab, 12, 12, 5, 7, 6, True, l2=l2, clip=clip)
    tfm = model.tf_manager
    sess = tfm.sessions[0]
    store = sess.store
    delayed = DelayedUpdateTrainer(batches_per_update=n, objectives=[CostObjective(model.decoder)],
                                   l2_weight=l2, clip_norm=clip)
    batches = []
    for i in range(2 * n):
        ds = synthetic.synthetic_dataset(seed=40 + i, batch=4 + i % 3, src_len=7, tgt_len=6, vocab=vocab, ragged=True)
        src = O.pad_ids([list(s) for s in ds.get_series("source")], 7)
        tgt = O.pad_ids([list(s) for s in ds.get_series("target")], 7, add_end_symbol=True)
        batches.append((ds, src, np.ascontiguousarray(tgt.T)))

    tp = TR.to_torch(params)
    ref_m = {k: torch.zeros_like(v) for k, v in tp.items()}
    ref_v = {k: torch.zeros_like(v) for k, v in tp.items()}
    clipped, solid = 0, {}
    for update in range(2):
        before = {name: store[name].cpu().numpy().copy() for name in store.names()}
        mean = {k: torch.zeros_like(v) for k, v in tp.items()}
        for i in range(n):
            ds, src, tgt = batches[update * n + i]
            loss, _, _, grads = TR.train_step_grads(tp, src, tgt, l1_weight=0.0, l2_weight=l2)
            for k in mean:
                mean[k] += grads[k] / n
            res = tfm.execute(ds, delayed.feedables, [delayed], train=True)[0]
            assert set(res.losses) == {"decoder - cost", "L1", "L2"}
            assert abs(res.losses["decoder - cost"] - float(loss)) < 1e-4 * float(loss)
            if i < n - 1:       # accumulating: parameters and global step untouched
                assert sess.global_step == update
                assert all(np.array_equal(store[name].cpu().numpy(), before[name]) for name in store.names())
        assert sess.global_step == update + 1
        clipped += sum(1 for g in mean.values() if float(g.norm()) > clip)
        TR.clip_and_adam(tp, mean, ref_m, ref_v, update + 1, clip)
        m, _ = store.ensure_adam()
        for name in store.names():
            if name.endswith("attn_bias"):
                continue
            spec = store.specs[name]
            got_m = m[spec.offset:spec.offset + spec.size].cpu().numpy()
            want_m = ref_m[name].numpy().reshape(-1)
            assert np.abs(got_m - want_m).max() <= 2e-3 * max(np.abs(want_m).max(), 1e-7), (update, name)
            # parameters: Adam's early updates are ~ lr * g / |g|, so an entry whose gradient is noise-sized may go
            # either way (bounded by 2 lr per update); entries with a real gradient must land on the oracle's value
            got, want = store[name].cpu().numpy().reshape(-1), tp[name].detach().numpy().reshape(-1)
            g = mean[name].numpy().reshape(-1)
            big = solid[name] = solid.get(name, True) & (np.abs(g) > 2e-2 * np.abs(g).max())
            assert np.abs(got - want).max() <= 2.1e-4 * (update + 1), (update, name)
            assert np.abs(got - want)[big].max() <= 2e-6 + 1e-5 * np.abs(want).max(), (update, name)
    assert 0 < clipped < 2 * len(tp), "the clip must bite on some tensors and spare others for the test to mean anything"


def test_adadelta_steps_track_the_oracle(dev, tmp_path):
    """tf.train.AdadeltaOptimizer with the arguments of tests/bpe.ini:102-108 behind the trainer's clip: four updates
    against the oracle's restatement of TF 1.12's ApplyAdadelta (oracle/torch_ref.py:clip_and_adadelta).  Both slots
    and the parameters are compared entry by entry (Adadelta has no sign-like first step: update ~ sqrt(eps / ((1-rho)
    g^2 + eps)) * g, smooth in g), and a checkpoint carries the slots under the optimizer's name."""
    from neuralmonkey_amd.optimizers import AdadeltaOptimizer
    from neuralmonkey_amd.trainers import CrossEntropyTrainer
    lr, eps, rho, clip, l2 = 0.5, 1e-6, 0.95, 0.05, 1e-3          # (lr 0.5: four steps must move the loss visibly)
    model, params, ds, src, tgt = _build(dev, 200, 32, 32, 8, 10, 9, True, l2=l2, clip=clip)
    trainer = CrossEntropyTrainer(decoders=[model.decoder], l2_weight=l2, clip_norm=clip,
                                  optimizer=AdadeltaOptimizer(learning_rate=lr, epsilon=eps, rho=rho, name="adadelta"))
    tfm = model.tf_manager
    store = tfm.sessions[0].store
    tp = TR.to_torch(params)
    acc = {k: torch.zeros_like(v) for k, v in tp.items()}
    acc_u = {k: torch.zeros_like(v) for k, v in tp.items()}
    ref_losses, got_losses = [], []
    for _ in range(4):
        loss, _, _, grads = TR.train_step_grads(tp, src, tgt, l1_weight=0.0, l2_weight=l2)
        ref_losses.append(float(loss))
        TR.clip_and_adadelta(tp, grads, acc, acc_u, clip, lr=lr, rho=rho, eps=eps)
        res = tfm.execute(ds, trainer.feedables, [trainer], train=True)[0]
        got_losses.append(res.losses["decoder - cost"])
    assert np.allclose(got_losses, ref_losses, rtol=2e-4), (got_losses, ref_losses)
    assert got_losses[3] < got_losses[0] - 1e-3
    a, au = store.ensure_adam()
    for name in store.names():
        if name.endswith("attn_bias"):
            continue
        spec = store.specs[name]
        sl = slice(spec.offset, spec.offset + spec.size)
        for got, want, what in ((a[sl], acc[name], "accum"), (au[sl], acc_u[name], "accum_update"),
                                (store[name], tp[name].detach(), "variable")):
            got, want = got.cpu().numpy().reshape(-1), want.numpy().reshape(-1)
            assert np.abs(got - want).max() <= 5e-3 * max(np.abs(want).max(), 1e-12), (name, what)
    tfm.checkpoint_format = "tf"
    prefix = str(tmp_path / "variables.data")
    tfm.save(prefix)
    from neuralmonkey_amd import tf_bundle
    keys = set(tf_bundle.read_bundle(prefix))
    assert all(n + "/adadelta" in keys and n + "/adadelta_1" in keys for n in store.names())
    assert "beta1_power" not in keys and not any(k.endswith("/Adam") for k in keys)
    kept = a.clone()
    a.zero_()
    store.slot_suffixes = ("/Adam", "/Adam_1")                # a fresh process has not named the slots yet
    tfm.restore(prefix)
    assert torch.equal(store.ensure_adam()[0], kept) and store.slot_suffixes == ("/adadelta", "/adadelta_1")


@pytest.mark.parametrize("rnn,batch,slen,tlen", [(384, 20, 11, 9), (256, 37, 14, 12)])
def test_cluster_time_loops_train_like_the_stepwise_launches(dev, rnn, batch, slen, tlen):
    """All four time loops of a training step as one launch each (nm_gru_seq_fwd / nm_gru_seq_bwd, the default where
    the shape allows) against two graph-replayed launches per step (NM_CLUSTER_LOOPS=0): the same products and
    epilogues in the same order -- losses, every gradient and the parameters after three updates agree to rounding."""
    from neuralmonkey_amd import ops
    assert ops.gru_seq_supported(batch, rnn, 2) and ops.gru_seq_supported(batch, rnn, 1), "must take the cluster kernels"
    results = []
    for cluster in (False, True):
        model, params, ds, src, tgt = _build(dev, 200, rnn, rnn, batch, slen, tlen, True, l2=1e-6, clip=1.0)
        sess = model.tf_manager.sessions[0]
        sess.use_cluster_loops = cluster
        store = sess.store
        losses = []
        for _ in range(3):
            res = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
            losses.append(res.losses["decoder - cost"])
        grads = {n: store.g(n).cpu().numpy().copy() for n in store.names()}
        results.append((losses, grads, {n: store[n].cpu().numpy().copy() for n in store.names()}))
    (l0, g0, p0), (l1, g1, p1) = results
    assert np.allclose(l0, l1, rtol=2e-6), (l0, l1)
    for n in g0:
        if n.endswith("attn_bias"):          # identically zero: rounding noise on both sides
            continue
        assert np.abs(g0[n] - g1[n]).max() <= 2e-5 * max(np.abs(g0[n]).max(), 1e-8), n
        assert np.abs(p0[n] - p1[n]).max() <= 2.1e-4 * 3, n            # (Adam's early steps are ~ lr * sign(g))


@pytest.mark.parametrize("fmt", ["npz", "tf"])
def test_checkpoint_carries_optimizer_state_and_global_step(dev, tmp_path, fmt, monkeypatch):
    """save -> restore -> continue == uninterrupted training: a checkpoint holds the variables, the Adam slots
    and global_step, as tf.train.Saver over all global variables does (tf_manager.py:257-277); without them
    Adam's bias correction and moments restart and the two runs part ways at the first resumed step."""
    from neuralmonkey_amd import synthetic
    monkeypatch.setenv("NM_CHECKPOINT_FORMAT", fmt)

    def fresh():
        model = synthetic.build_translation_model(vocab_src=300, vocab_tgt=300, emb=32, rnn=32, max_len=12,
                                                  beam_size=0, device=str(dev), seed=5)
        synthetic.load_baseline_weights(model.tf_manager.sessions[0].store, seed=9, std=0.1)
        return model
    batches = [synthetic.synthetic_dataset(seed=40 + i, batch=16, src_len=12, tgt_len=12, vocab=300, ragged=True)
               for i in range(5)]
    step = lambda model, ds: model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
    a = fresh()
    for ds in batches[:3]:
        step(a, ds)
    ckpt = str(tmp_path / "variables.data")
    a.tf_manager.save(ckpt)
    losses_a = [step(a, ds).losses["decoder - cost"] for ds in batches[3:]]
    final_a = a.tf_manager.sessions[0].store.state_dict()
    b = fresh()
    b.tf_manager.restore(ckpt)
    sess_b = b.tf_manager.sessions[0]
    assert sess_b.global_step == 3
    assert float(sess_b.store.adam_v.abs().max()) > 0.0
    b_start = sess_b.store.state_dict()
    losses_b = [step(b, batches[3]).losses["decoder - cost"]]
    after_one = sess_b.store.state_dict()
    losses_b.append(step(b, batches[4]).losses["decoder - cost"])
    assert sess_b.global_step == 5
    assert np.allclose(losses_a, losses_b, rtol=1e-6, atol=0)
    final_b = sess_b.store.state_dict()
    for name, want in final_a.items():      # (the embedding-gradient scatter adds with atomics: last-bit differences)
        assert np.abs(final_b[name] - want).max() <= 1e-6, name
    # and a restore that dropped the optimizer state would NOT reproduce the run: Adam's first-step update is
    # lr * sign(g), three orders of magnitude above this tolerance
    c = fresh()
    c.tf_manager.sessions[0].store.load_state_dict(b_start)
    step(c, batches[3])
    worst = max(float(np.abs(c.tf_manager.sessions[0].store.state_dict()[n] - after_one[n]).max()) for n in after_one)
    assert worst > 1e-5


def test_train_op_differentiates_with_train_mode_fed_false(dev):
    """The reference's train_op differentiates whatever ``train_mode`` is fed (the placeholder only switches
    dropout, model/model_part.py): a trainer run with train=False yields the gradients of a train=True run of
    the same dropout-free model, not a crash on forward state that was never kept."""
    model, _, ds, _, _ = _build(dev, 64, 12, 12, 5, 7, 6, True)
    store = model.tf_manager.sessions[0].store
    theta0 = store.theta.clone()
    res_t = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
    g_t = store.ensure_grad().clone()
    store.theta.copy_(theta0)
    res_f = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=False)[0]
    g_f = store.ensure_grad().clone()
    assert res_f.losses["decoder - cost"] == pytest.approx(res_t.losses["decoder - cost"], rel=1e-6)
    assert float((g_t - g_f).abs().max()) <= 1e-6 * float(g_t.abs().max())
    assert float(g_t.abs().max()) > 0
