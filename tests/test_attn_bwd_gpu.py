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
