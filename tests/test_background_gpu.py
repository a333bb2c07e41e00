"""Background launches (round 4): `nm_gemm_f32` algo 4 and a context in background mode run the SAME kernels with a
residency cap (unused dynamic LDS) -- their results must equal the foreground launches bit for bit; and the training
step that spreads its leaf work over side streams must produce the gradients of the step in stream order.
Reference for what is computed: tf.matmul / GRUCell (nn/ortho_gru_cell.py:44-53); the schedule has no counterpart in
the reference (TensorFlow's executor orders the ops of tf.gradients, trainers/generic_trainer.py:136-195)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n,k,ta,tb", [
    (512, 32000, 640, True, False),      # the vocabulary projection's weight gradient, shortened in K: 128x128 tiles
    (512, 1024, 6400, True, False),      # a recurrent cell's weight gradient: split-K + reduction
    (2048, 512, 96, False, True),        # 64x64 tiles (a few hundred rows would take the medium-M route in the
                                         # foreground, which a background launch skips: another kernel)
    (70, 50, 33, False, False),          # ragged, unaligned: the scalar-load instances
])
def test_background_gemm_equals_the_foreground_gemm_bit_for_bit(dev, m, n, k, ta, tb):
    from neuralmonkey_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    a = torch.randn((k, m) if ta else (m, k), device=dev, generator=g)
    b = torch.randn((n, k) if tb else (k, n), device=dev, generator=g)
    base = torch.randn(m, n, device=dev, generator=g)
    for acc in (False, True):
        want, got = base.clone(), base.clone()
        ops.gemm(a, b, out=want, trans_a=ta, trans_b=tb, accumulate=acc)
        ops.gemm(a, b, out=got, trans_a=ta, trans_b=tb, accumulate=acc, algo=ops.GEMM_BACKGROUND)
        torch.cuda.synchronize()
        assert torch.equal(want, got)
    ref = (a.double().T if ta else a.double()) @ (b.double().T if tb else b.double())
    assert float((want.double() - base.double() - ref).abs().max()) < 1e-3 * float(ref.abs().max())


def test_context_background_mode_changes_no_result_and_is_switched_off_again(dev):
    from neuralmonkey_amd import _lib, ops
    from neuralmonkey_amd.nn import gru
    lib = _lib.load()
    rows, h, ndir = 48, 64, 2
    g = torch.Generator(device=dev).manual_seed(9)
    r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.3
    xp, wg, wc = r(rows, 7, ndir * 3 * h), r(ndir, h, 2 * h), r(ndir, h, h)
    a, b = r(300, 200), r(200, 500)

    def run():
        hcur = torch.zeros(ndir, rows, h, device=dev)
        ru, rh = torch.empty(ndir, rows, 2 * h, device=dev), torch.empty(ndir, rows, h, device=dev)
        hg, hc = torch.empty(ndir, rows, 2 * h, device=dev), torch.empty(ndir, rows, h, device=dev)
        out = torch.zeros(rows, 7, ndir * h, device=dev)
        lengths = torch.full((rows,), 7, dtype=torch.int32, device=dev)
        for t in range(7):
            gru.step_fwd(xp, (3 * h, 7 * ndir * 3 * h, ndir * 3 * h), hcur, hcur, wg, wc, ru, rh, None, out,
                         (h, 7 * ndir * h, ndir * h), lengths, t, ndir, rows, h, False, hg, hc)
        c = ops.gemm(a, b)
        torch.cuda.synchronize()
        return out.clone(), hcur.clone(), c
    want = run()
    _lib.check(lib.nm_ctx_set_background(None, 1), "nm_ctx_set_background")
    try:
        got = run()
    finally:
        _lib.check(lib.nm_ctx_set_background(None, 0), "nm_ctx_set_background")
    for w, x in zip(want, got):
        assert torch.equal(w, x)
    assert float(want[0].abs().max()) > 0.0


def _gradients(dev, side_stream, monkeypatch):
    from neuralmonkey_amd import synthetic
    from oracle import nm_oracle as O
    monkeypatch.setenv("NM_SIDE_STREAM", "1" if side_stream else "0")
    model = synthetic.build_translation_model(vocab_src=2000, vocab_tgt=2000, emb=64, rnn=64, max_len=24,
                                              beam_size=0, device=str(dev), l2_weight=1e-6, clip_norm=1.0)
    sess = model.tf_manager.sessions[0]
    assert sess.use_side_stream == side_stream
    sess.store.load_state_dict(O.init_params(seed=3, vocab_src=2000, vocab_tgt=2000, emb=64, rnn=64, std=0.1))
    ds = synthetic.synthetic_dataset(seed=4, batch=32, src_len=24, tgt_len=20, vocab=2000, ragged=True)
    losses = []
    for _ in range(3):       # eager pass, capture pass, replay: the side lanes in all three
        res = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
        losses.append(float(res.losses["decoder - cost"]))
    torch.cuda.synchronize()
    return losses, {n: sess.store[n].cpu().numpy().copy() for n in sess.store.names()}


def test_training_over_side_lanes_equals_training_in_stream_order(dev, monkeypatch):
    """Three optimizer steps with the leaf work on side streams (capped GEMMs, deferred input half, attention keys
    beside the decoder loop) against the same three steps enqueued on one stream: split-K reductions are in a fixed
    order and every buffer has one writer, so losses and parameters agree to rounding of the different GEMM
    instances only (the capped launches are the same kernels: bit-equal)."""
    l_side, p_side = _gradients(dev, True, monkeypatch)
    l_one, p_one = _gradients(dev, False, monkeypatch)
    assert np.allclose(l_side, l_one, rtol=1e-6, atol=0.0), (l_side, l_one)
    for name, want in p_one.items():
        if name.endswith("attn_bias"):       # its gradient is identically 0 (softmax shift invariance): Adam
            continue                         # normalises pure rounding noise to +-lr per step
        got = p_side[name]
        # three Adam steps of lr 1e-4 move a parameter by ~3e-4: agreement to 0.1 % of that movement
        assert np.abs(got - want).max() <= 3e-7 + 1e-6 * np.abs(want).max(), name
