"""A GRU time loop launched as ONE cluster kernel (csrc/nm_gru_cluster.hip) needs all its workgroups resident at
once; when something else holds compute units its hand-offs time out, it raises the session's error word and its
results are garbage.  That must cost a slow step, not the run:

  * the optimizer kernels skip their update while the word is set (nm_optim_apply, ``skip_word``), so nothing of
    the garbage reaches the variables or the optimizer slots;
  * the session switches to the per-step path (ONE warning), runs the affected steps / batches again from their
    saved feeds and hands their losses to whoever holds the lazily-read losses of the failed runs;
  * the result equals a run in which the same steps were taken on the per-step path from the start: the loss of
    the step that is run again is bit-identical (same variables, same kernels); everything after it to rounding
    only, because the embedding gradients are scattered with float atomics (csrc/nm_backward.hip:
    embedding_scatter_kernel) and no two runs add them in the same order.  A skipped, doubled or garbage update
    would move every variable by ~learning rate = 1e-4; the tests allow a mean difference of 2e-8.

The give-up is forced by the library's test hook (nm_gru_seq_force_give_up: the launch raises its error word at
once instead of after 0.2 s) and, once, provoked for real: 32 workgroups of another stream sit on 32 CUs with all
their LDS while a training step is launched.  Reference semantics: trainers/generic_trainer.py:179-195 (one update
per batch), decoders/autoregressive.py:313-316.
"""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VOCAB, DIM, BATCH, SLEN, TLEN = 300, 256, 24, 9, 8


def _model(dev, trainer_cls=None, **kw):
    from neuralmonkey_amd import synthetic
    from oracle import nm_oracle as O
    model = synthetic.build_translation_model(vocab_src=VOCAB, vocab_tgt=VOCAB, emb=DIM, rnn=DIM, max_len=12,
                                              beam_size=0, device=str(dev))
    params = O.init_params(seed=3, vocab_src=VOCAB, vocab_tgt=VOCAB, emb=DIM, rnn=DIM, std=0.1)
    model.tf_manager.sessions[0].store.load_state_dict(params)
    if trainer_cls is not None:
        trainer = trainer_cls(objectives=model.trainer.objectives, l2_weight=1e-8, clip_norm=1.0, **kw)
        model = model._replace(trainer=trainer)
    return model


def _batches(n):
    from neuralmonkey_amd import synthetic
    return [synthetic.synthetic_dataset(seed=10 + i, batch=BATCH, src_len=SLEN, tgt_len=TLEN, vocab=VOCAB, ragged=True)
            for i in range(n)]


def _close(losses, ref_losses, exact_at=None):
    assert len(losses) == len(ref_losses)
    for i, (got, want) in enumerate(zip(losses, ref_losses)):
        assert got.keys() == want.keys()
        for name in got:
            if i == exact_at:
                assert got[name] == want[name], "step {}: {} {} != {}".format(i, name, got[name], want[name])
            else:
                assert abs(got[name] - want[name]) <= 1e-5 * abs(want[name]), (i, name, got[name], want[name])


def _same_variables(theta, ref_theta):
    """A lost, doubled or garbage update moves EVERY element with a gradient by about the learning rate (Adam: 1e-4 per
    step); the order of the float atomics moves a few elements whose gradient nearly cancels by a few 1e-6 and the
    rest by 1e-9 or less."""
    assert torch.isfinite(theta).all()
    diff = (theta - ref_theta).abs()
    assert float(diff.mean()) < 2e-8 and float(diff.max()) < 3e-5, \
        "variables differ (mean {:.2e}, max {:.2e}): an update was lost, doubled or garbage".format(float(diff.mean()),
                                                                                                  float(diff.max()))


def _uses_cluster_loops(model):
    from neuralmonkey_amd import ops
    return model.tf_manager.sessions[0].use_cluster_loops and ops.gru_seq_supported(BATCH, DIM, 1)


def _train(model, batches, fail_before=None, stepwise_from=None, read_losses="end"):
    """Training steps over ``batches``; ``fail_before`` = index of the step whose first cluster loop is forced to
    give up; ``stepwise_from`` = index from which the session is put on the per-step path by hand (the reference
    run); ``read_losses``: "each" reads every step's losses right away, "end" only after the last step."""
    from neuralmonkey_amd import ops
    tfm = model.tf_manager
    sess = tfm.sessions[0]
    results = []
    for i, ds in enumerate(batches):
        if stepwise_from is not None and i == stepwise_from:
            torch.cuda.synchronize()
            sess.use_cluster_loops = False
            sess._graphs.clear()                                   # pylint: disable=protected-access
        if fail_before is not None and i == fail_before:
            ops.gru_seq_force_give_up(1)
        res = tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
        if read_losses == "each":
            dict(res.losses)
        results.append(res)
    losses = [dict(r.losses) for r in results]
    torch.cuda.synchronize()
    assert ops.gru_seq_force_give_up(0) == 0, "the forced give-up was never consumed: no cluster loop was launched"
    return losses, sess.store.theta.clone(), sess


@pytest.mark.parametrize("read_losses", ["end", "each"])
@pytest.mark.parametrize("fail_at", [0, 2])
def test_a_given_up_time_loop_costs_a_slow_step_not_the_run(dev, fail_at, read_losses):
    batches = _batches(5)
    ref_model = _model(dev)
    if not _uses_cluster_loops(ref_model):
        pytest.skip("cluster loops are off or unsupported on this device")
    ref_losses, ref_theta, ref_sess = _train(ref_model, batches, stepwise_from=fail_at, read_losses=read_losses)
    model = _model(dev)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        losses, theta, sess = _train(model, batches, fail_before=fail_at, read_losses=read_losses)
    told = [w for w in caught if "gave up waiting" in str(w.message)]
    assert len(told) == 1, "exactly one warning"
    assert not sess.use_cluster_loops and sess.cluster_demotions == 1
    assert sess.global_step == ref_sess.global_step == len(batches)
    _close(losses, ref_losses, exact_at=fail_at if fail_at == 0 else None)
    _same_variables(theta, ref_theta)
    assert not sess.cluster_failure()
    m, v = sess.store.ensure_adam()
    rm, rv = ref_sess.store.ensure_adam()
    assert torch.isfinite(m).all() and torch.isfinite(v).all()
    assert float((m - rm).abs().max()) <= 1e-4 * float(rm.abs().max()), "optimizer slots differ: a garbage update got through"
    assert float((v - rv).abs().max()) <= 1e-4 * float(rv.abs().max())


def test_the_garbage_update_is_skipped_on_the_device(dev):
    """The gate itself: with the error word set by hand the optimizer launch leaves variables and slots alone."""
    model = _model(dev)
    tfm, sess = model.tf_manager, model.tf_manager.sessions[0]
    ds = _batches(1)[0]
    dict(tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0].losses)
    torch.cuda.synchronize()
    before = sess.store.theta.clone()
    m0 = sess.store.ensure_adam()[0].clone()
    from neuralmonkey_amd import ops
    tables = model.trainer._optim_tables(sess.store)               # pylint: disable=protected-access
    grad = sess.store.ensure_grad()
    grad.fill_(0.5)
    m, v = sess.store.ensure_adam()
    word = torch.ones(1, dtype=torch.int32, device=dev)
    tables.regularize_and_norms(sess.store.theta, grad, 0.0, 0.0)
    tables.clip_adam(sess.store.theta, grad, m, v, 1.0, 1e-3, 0.9, 0.999, 1e-8, skip=word)
    torch.cuda.synchronize()
    assert torch.equal(sess.store.theta, before) and torch.equal(m, m0)
    word.zero_()
    tables.clip_adam(sess.store.theta, grad, m, v, 1.0, 1e-3, 0.9, 0.999, 1e-8, skip=word)
    torch.cuda.synchronize()
    assert not torch.equal(sess.store.theta, before)
    ops.zero_if(word, grad)
    assert float(grad.abs().max()) > 0
    word.fill_(1)
    ops.zero_if(word, grad)
    assert float(grad.abs().max()) == 0.0


def test_delayed_updates_survive_a_given_up_loop_in_the_middle_of_a_window(dev):
    """trainers/delayed_update_trainer.py:142-204: the accumulation buffer lives across steps; the failed step's
    gradient is zeroed on the device before it is added, and the window goes on from the step that is run again."""
    from neuralmonkey_amd.trainers import DelayedUpdateTrainer
    batches = _batches(6)
    for fail_at in (1, 3):                                        # second batch of a window / first of the next
        ref_model = _model(dev, DelayedUpdateTrainer, batches_per_update=3)
        if not _uses_cluster_loops(ref_model):
            pytest.skip("cluster loops are off or unsupported on this device")
        ref_losses, ref_theta, _ = _train(ref_model, batches, stepwise_from=fail_at)
        model = _model(dev, DelayedUpdateTrainer, batches_per_update=3)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            losses, theta, sess = _train(model, batches, fail_before=fail_at)
        _close(losses, ref_losses)
        _same_variables(theta, ref_theta)
        assert sess.global_step == 2 and not sess.use_cluster_loops


def test_inference_runs_the_batch_again_on_the_per_step_path(dev):
    from neuralmonkey_amd import ops
    model = _model(dev)
    if not _uses_cluster_loops(model):
        pytest.skip("cluster loops are off or unsupported on this device")
    tfm, sess = model.tf_manager, model.tf_manager.sessions[0]
    a, b = _batches(2)
    want_a = tfm.execute(a, model.greedy_runner.feedables, [model.greedy_runner])[0].outputs["target"]
    want_b = tfm.execute(b, model.greedy_runner.feedables, [model.greedy_runner])[0].outputs["target"]
    ops.gru_seq_force_give_up(1)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got_a = tfm.execute(a, model.greedy_runner.feedables, [model.greedy_runner], lookahead=b)[0].outputs["target"]
        got_b = tfm.execute(b, model.greedy_runner.feedables, [model.greedy_runner])[0].outputs["target"]
    assert ops.gru_seq_force_give_up(0) == 0
    assert len([w for w in caught if "gave up waiting" in str(w.message)]) == 1
    assert got_a == want_a and got_b == want_b
    assert not sess.use_cluster_loops
    # a raised word with the loops already off is the fallback's own failure: that one raises
    sess.error_word().fill_(1)
    with pytest.raises(RuntimeError, match="gave up waiting"):
        tfm.execute(a, model.greedy_runner.feedables, [model.greedy_runner])
    sess.error_word().zero_()


def test_two_cluster_loops_on_two_streams_take_turns(dev):
    """One workgroup of a cluster loop fits a CU (8 waves x 160 registers): two loops launched at once on two streams
    would interleave over the CUs and both time out.  ops.gru_seq_fwd orders every launch behind the previous one's
    event, whatever its stream."""
    from neuralmonkey_amd import ops
    rows, steps, h = 48, 12, 256
    if not ops.gru_seq_supported(rows, h, 1):
        pytest.skip("cluster loops unsupported on this device")
    g = torch.Generator(device=dev).manual_seed(5)
    xp = torch.randn(rows * steps, 3 * h, device=dev, generator=g) * 0.5
    wgh = torch.randn(1, h, 2 * h, device=dev, generator=g) * 0.09
    wch = torch.randn(1, h, h, device=dev, generator=g) * 0.09
    strides = (3 * h, steps * 3 * h, 3 * h)

    def loop(ws):
        hb = torch.zeros(1, rows, h, device=dev)
        ru = torch.empty(steps, 1, rows, 2 * h, device=dev)
        cs = torch.empty(steps, 1, rows, h, device=dev)
        out = torch.zeros(rows, steps, h, device=dev)
        ops.gru_seq_fwd(steps, 1, rows, h, xp, strides, hb, hb, 0, ru[0], rows * 2 * h, None, 0, cs[0], rows * h,
                        wgh, wch, ws, out=out, out_strides=(h, steps * h, h))
        return out, hb

    ws = [ops.gru_seq_workspace(rows, h, 1, dev) for _ in range(3)]
    want, want_h = loop(ws[0])
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outs = []
    for _ in range(4):
        with torch.cuda.stream(s1):
            outs.append((loop(ws[1]), ws[1]))
        with torch.cuda.stream(s2):
            outs.append((loop(ws[2]), ws[2]))
        torch.cuda.synchronize()
        for (out, hb), w in outs[-2:]:
            assert not ops.gru_seq_failed(w), "a cluster loop gave up: two loops were resident at once"
            assert torch.equal(out, want) and torch.equal(hb, want_h)


@pytest.mark.slow
def test_a_real_compute_unit_hog_makes_the_loop_give_up_and_the_step_is_recovered(dev):
    """Not the hook: 32 workgroups of another stream hold 32 CUs' LDS for 0.6 s while a training step is launched.
    The step's first cluster loop cannot place all its workgroups, gives up after 0.2 s, and the step is run again
    on the per-step path -- same losses and variables as a per-step run."""
    from neuralmonkey_amd import ops
    batches = _batches(3)
    ref_model = _model(dev)
    if not _uses_cluster_loops(ref_model):
        pytest.skip("cluster loops are off or unsupported on this device")
    ref_losses, ref_theta, _ = _train(ref_model, batches, stepwise_from=1)
    model = _model(dev)
    tfm, sess = model.tf_manager, model.tf_manager.sessions[0]
    side = torch.cuda.Stream(device=dev)
    losses = []
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for i, ds in enumerate(batches):
            if i == 1:
                torch.cuda.synchronize()
                ops.gru_seq_test_hog(32, 160 * 1024, 600000, stream=side)
            losses.append(dict(tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0].losses))
    torch.cuda.synchronize()
    if sess.use_cluster_loops:
        pytest.skip("the hog did not keep the loop from becoming resident on this device")
    assert len([w for w in caught if "gave up waiting" in str(w.message)]) == 1
    _close(losses, ref_losses)
    _same_variables(sess.store.theta, ref_theta)
