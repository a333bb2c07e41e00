"""nm_attn_energy_bwd (the fused tanh-energies backward of the Bahdanau attention,
feed_forward.py:120-123) against a plain PyTorch fp32 reference of the same op, over position counts
that exercise every register-chunk size, ragged last chunks, feature counts off the block size, and
the accumulate flag.  Tolerance 2e-5 of each output's max magnitude (fp32 sums in another order,
fast tanh)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(de, hf, y, v):
    z = torch.tanh(hf.unsqueeze(0) + y.unsqueeze(2))            # [T,B,S,A]
    g = de.unsqueeze(-1) * (1.0 - z * z)
    dhf = v * g.sum(0)                                          # [B,S,A]
    dvp = (de.unsqueeze(-1) * z).sum(0)                         # [B,S,A]
    dy = v * g.sum(2)                                           # [T,B,A]
    return dhf, dvp, dy


@pytest.mark.parametrize("t,b,s,a", [(1, 3, 5, 7), (4, 2, 8, 36), (3, 2, 9, 130), (2, 3, 17, 128), (5, 2, 50, 260),
                                     (2, 2, 33, 64), (3, 1, 64, 12), (2, 2, 100, 40), (6, 4, 13, 256)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_attn_energy_bwd_matches_torch(dev, t, b, s, a, accumulate):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(t * 1000 + s * 10 + a)
    mk = lambda *shape: torch.tensor(rng.standard_normal(shape).astype(np.float32), device=dev)
    de, hf, y, v = mk(t, b, s), mk(b, s, a), mk(t, b, a), mk(a)
    dhf0, dvp0 = mk(b, s, a), mk(b * s, a)
    dhf, dvp, dy = dhf0.clone(), dvp0.clone(), torch.full((t, b, a), 7.0, device=dev)
    ops.attn_energy_bwd(de, hf, y, v, dhf, dvp, dy, accumulate=accumulate)
    want_dhf, want_dvp, want_dy = _reference(de, hf, y, v)
    if accumulate:
        want_dhf, want_dvp = want_dhf + dhf0, want_dvp + dvp0.view(b, s, a)
    for got, want in ((dhf, want_dhf), (dvp.view(b, s, a), want_dvp), (dy, want_dy)):
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * max(scale, 1.0)


@pytest.mark.parametrize("t,b,s,a", [(4, 2, 50, 260), (3, 2, 9, 130)])
def test_attn_energy_bwd_beyond_the_product_form_range(dev, t, b, s, a):
    """Keys / queries of magnitude up to ~60 that nearly cancel: exp(2 x) overflows fp32 beyond |x| = 44, so the
    product form tanh(h + y) = 1 - 2 / (1 + exp(2h) exp(2y)) is only used up to NM_EXP2X_MAX = 43 and the kernel
    must take tanh(h + y) itself for the affected elements -- same tolerance as everywhere else."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(s * 7 + a)
    mk = lambda *shape: torch.tensor(rng.standard_normal(shape).astype(np.float32), device=dev)
    de, v = mk(t, b, s), mk(a)
    hf = mk(b, s, a)
    hf[:, ::3, ::5] += 55.0                                     # some key elements far outside
    hf[:, 1::4, 1::7] -= 60.0
    y = mk(t, b, a)
    y[:, :, ::5] -= 54.0                                        # ... cancelled by the matching query columns
    y[1:, :, 1::7] += 59.0
    dhf, dvp, dy = torch.zeros(b, s, a, device=dev), torch.zeros(b * s, a, device=dev), torch.zeros(t, b, a, device=dev)
    ops.attn_energy_bwd(de, hf, y, v, dhf, dvp, dy, accumulate=False)
    want_dhf, want_dvp, want_dy = _reference(de, hf, y, v)
    for got, want in ((dhf, want_dhf), (dvp.view(b, s, a), want_dvp), (dy, want_dy)):
        assert torch.isfinite(got).all()
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * max(scale, 1.0)


@pytest.mark.parametrize("b,s,c,a", [(3, 5, 8, 7), (2, 50, 1024, 512), (4, 33, 260, 130), (64, 50, 1024, 512), (2, 130, 64, 40),
                                     (1, 1, 4, 1)])
@pytest.mark.parametrize("masked", [False, True])
def test_attn_step_bwd_matches_float64_autograd(dev, b, s, c, a, masked):
    """nm_attn_step_bwd -- a taped decoder step's attention backward up to the query in one launch -- against float64
    autograd of the reference's arithmetic (feed_forward.py:120-149: energies, softmax, mask, renormalisation with
    1e-8, context sum), with the context gradient read from rows of a wider buffer as the decoder hands it over.
    Tolerance 2e-5 of each output's largest magnitude."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(b * 1000 + s * 10 + a)
    mk = lambda *shape: torch.tensor(rng.standard_normal(shape).astype(np.float32), device=dev)
    states, hf, y, v = mk(b, s, c), mk(b, s, a), mk(b, a), mk(a)
    wide = mk(b, c + 8)
    dctx = wide[:, :c]                                           # strided rows, 16-byte aligned
    mask = None
    if masked:
        lens = rng.integers(1, s + 1, size=b)
        mask = torch.tensor((np.arange(s)[None, :] < lens[:, None]).astype(np.float32), device=dev)
    y64 = y.double().requires_grad_(True)
    e64 = (torch.tanh(hf.double() + y64.unsqueeze(1)) * v.double()).sum(-1)
    e64.retain_grad()
    p = torch.softmax(e64, dim=-1)
    if mask is not None:
        p = p * mask.double()
    w = p / (p.sum(-1, keepdim=True) + 1e-8)
    ctxv = (w.unsqueeze(-1) * states.double()).sum(1)
    ctxv.backward(dctx.double())
    e = e64.detach().float().contiguous()
    de = torch.full((b, s), 3.0, device=dev)
    dy = torch.full((b, a + 3), 5.0, device=dev)[:, :a]
    assert ops.attn_step_bwd_ok(dctx, c)
    ops.attn_step_bwd(dctx, states, e, mask, hf, y, v, de, dy)
    for got, want in ((de, e64.grad), (dy, y64.grad)):
        scale = max(float(want.abs().max()), 1e-3)
        assert float((got.double() - want).abs().max()) <= 2e-5 * scale
