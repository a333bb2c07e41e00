"""The vocabulary projection with its row statistics in the GEMM epilogue (nm_logits_stats_gemm) and the
two consumers of those statistics: nm_greedy_finish (argmax + symbol / finished update + next input
embedding, decoders/autoregressive.py:461-480) and nm_beam_topk_step_tiles (beam body that reads back
only the vocabulary tiles that can hold a top-k candidate, beam_search_decoder.py:440-501).

  * the logits are bit-identical to nm_gemm_f32's (gemm_tiled's statistics epilogue) or -- the activation-stationary
    kernel of csrc/nm_proj.hip -- as close to the float64 product as nm_gemm_f32's; the merged tile maxima / first
    argmax are exact for the kernel's own logits, the merged lse matches float64 NumPy to 1e-6;
  * greedy tail against a NumPy restatement of autoregressive.py:461-480;
  * beam step against nm_beam_topk_step_fused (itself checked against the two-pass kernel and the oracle)
    on the filter-stress inputs of test_beam_fused_gpu.py: every index output identical, scores within
    2e-6 (the lse of merged tiles and of one pass differ in the last bits).
"""
import numpy as np
import pytest
import torch

from .test_beam_fused_gpu import _state

pytestmark = pytest.mark.gpu
END = 2


def tile_of(rows):
    """Columns per statistics tile: the N extent of the GEMM's block tile (64 for <= 256 rows, else 128)."""
    from neuralmonkey_amd import _lib
    return int(_lib.load().nm_logits_stats_tile(rows))


def T(a, dev, dt=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)


def merged(stats, rows, v):
    """NumPy merge of the kernel's tile records -> (max, argmax, lse)."""
    tile = tile_of(rows)
    nt = (v + tile - 1) // tile
    st = stats.cpu().numpy().reshape(rows, nt, 4)
    mx = st[:, :, 0]
    sm = st[:, :, 1].astype(np.float64)
    arg = st[:, :, 2].copy().view(np.int32)
    big = mx.max(1)
    first = np.array([arg[r, np.nonzero(mx[r] == big[r])[0]].min() for r in range(rows)])
    lse = np.log((sm * np.exp(mx.astype(np.float64) - big[:, None])).sum(1))
    return big, first, lse


@pytest.mark.parametrize("m,n,k,tb", [(128, 32000, 512, False), (640, 32000, 512, False), (37, 4104, 64, False),
                                      (128, 1000, 512, True), (5, 132, 8, False), (130, 128, 16, True),
                                      (37, 4104, 128, False), (130, 1000, 256, False), (640, 32004, 384, False),
                                      (16, 64, 128, False), (257, 68, 512, False)])
def test_stats_gemm_logits_and_merged_statistics(dev, m, n, k, tb):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k) if tb else (k, n)) * 0.3).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    bias[3 % n] = -1e9                                            # supress_unk bias
    ad, wd, bd = T(a, dev), T(w, dev), T(bias, dev)
    want = ops.gemm(ad, wd, bias=bd, trans_b=tb, algo=1)
    stats = ops.logits_stats_buffer(m, n, dev)
    got = torch.full((m, n), float("nan"), device=dev)
    ops.logits_stats_gemm(ad, wd, bd, stats, out=got, trans_b=tb)
    if not torch.equal(got, want):
        # the activation-stationary kernel of the decoding steps (csrc/nm_proj.hip: W stored [K, N], K a multiple of
        # 128 up to 512) adds the same exact-fp32 products in another order than gemm_tiled: its logits agree with
        # nm_gemm_f32 to rounding, and are as close to the float64 product as nm_gemm_f32's are
        assert not tb and k % 128 == 0 and k <= 512, "only the activation-stationary kernel may differ from nm_gemm_f32"
        ref = a.astype(np.float64) @ w.astype(np.float64) + bias
        keep = ref > -1e8
        scale = np.abs(ref[keep]).max()
        err_got = np.abs(got.cpu().numpy() - ref)[keep].max() / scale
        err_want = np.abs(want.cpu().numpy() - ref)[keep].max() / scale
        assert err_got <= max(1.5 * err_want, 2e-7 * np.sqrt(k)), (err_got, err_want)
        assert not torch.isnan(got).any()
    stats2 = ops.logits_stats_buffer(m, n, dev)
    ops.logits_stats_gemm(ad, wd, bd, stats2, out=None, trans_b=tb)          # statistics only
    assert torch.equal(stats, stats2)
    x = got.cpu().numpy()                      # the statistics are exact for the kernel's OWN logits
    mx, arg, lse = merged(stats, m, n)
    assert np.array_equal(mx, x.max(1))
    assert np.array_equal(arg, x.argmax(1))
    x64 = x.astype(np.float64)
    ref_lse = np.log(np.exp(x64 - x64.max(1, keepdims=True)).sum(1))
    assert np.abs(lse - ref_lse).max() < 1e-6 * max(1.0, np.abs(ref_lse).max())


def _identity_logits(dev, logits):
    """Plant an exact logits matrix through the kernel under test: logits = I . L."""
    from neuralmonkey_amd import ops
    rows, v = logits.shape
    kpad = (rows + 3) // 4 * 4
    eye = np.zeros((rows, kpad), np.float32)
    eye[np.arange(rows), np.arange(rows)] = 1.0
    lpad = np.zeros((kpad, v), np.float32)
    lpad[:rows] = logits
    stats = ops.logits_stats_buffer(rows, v, dev)
    out = torch.empty((rows, v), device=dev)
    ops.logits_stats_gemm(T(eye, dev), T(lpad, dev), None, stats, out=out)
    assert np.array_equal(out.cpu().numpy(), logits)
    return out, stats


def test_greedy_finish_matches_the_reference_update(dev):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(3)
    rows, v, e = 70, 4000, 20
    logits = rng.standard_normal((rows, v)).astype(np.float32)
    logits[0, [5, 900, 3999]] = 7.0                                # tie for the maximum: first index wins
    logits[1, END] = 9.0                                           # emits </s>
    logits[2, 0] = 9.0                                             # emits <pad> while alive
    ld, stats = _identity_logits(dev, logits)
    fin0 = rng.random(rows) < 0.3
    fin0[:3] = False
    finished = T(fin0.astype(np.int32), dev, torch.int32)
    sym, mask = torch.empty(rows, dtype=torch.int32, device=dev), torch.empty(rows, dtype=torch.int32, device=dev)
    allfin = torch.ones(1, dtype=torch.int32, device=dev)
    table = T(rng.standard_normal((v, e)).astype(np.float32), dev)
    cat = torch.zeros((rows, 3 * e + 4), device=dev)
    emb = cat[:, e:2 * e]                                           # a column slice of a wider buffer
    amax, mx, lse = torch.empty(rows, dtype=torch.int32, device=dev), torch.empty(rows, device=dev), \
        torch.empty(rows, device=dev)
    ops.greedy_finish(stats, v, finished, sym, mask, END, allfin, table=table, emb_out=emb, argmax_out=amax,
                      max_out=mx, lse_out=lse)
    arg = logits.argmax(1)
    want_sym = np.where(fin0, 0, arg)
    want_fin = fin0 | (want_sym == END)
    assert np.array_equal(amax.cpu().numpy(), arg) and arg[0] == 5
    assert np.array_equal(sym.cpu().numpy(), want_sym)
    assert np.array_equal(finished.cpu().numpy().astype(bool), want_fin)
    assert np.array_equal(mask.cpu().numpy().astype(bool), ~want_fin)
    assert int(allfin.item()) == int(want_fin.all())
    assert np.array_equal(emb.cpu().numpy(), table.cpu().numpy()[want_sym])
    assert float(cat[:, :e].abs().max()) == 0.0 and float(cat[:, 2 * e:].abs().max()) == 0.0
    assert np.array_equal(mx.cpu().numpy(), logits.max(1))
    x64 = logits.astype(np.float64)
    assert np.abs(lse.cpu().numpy() - np.log(np.exp(x64 - x64.max(1, keepdims=True)).sum(1))).max() < 1e-5
    # everybody finished -> the flag stays set
    finished.fill_(1)
    allfin.fill_(1)
    ops.greedy_finish(stats, v, finished, sym, mask, END, allfin)
    assert int(allfin.item()) == 1 and int(sym.abs().sum().item()) == 0


def _run_tiles_vs_fused(dev, logits, k, lps, lens, fin, alpha=0.6):
    from neuralmonkey_amd import ops
    rows, v = logits.shape
    b = rows // k
    ld, stats = _identity_logits(dev, logits)
    lpsd, lensd, find = T(lps, dev), T(lens, dev, torch.int32), T(fin.astype(np.int32), dev, torch.int32)
    pen = ops.length_penalty_table(64, alpha, dev)
    i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)

    def outputs():
        return [torch.empty((b, k), device=dev), i32(b, k), i32(b, k), torch.empty((b, k), device=dev), i32(b, k),
                i32(b, k), i32(b, k)]
    ws = ops.beam_workspace(b, k, v, dev)
    ref, got = outputs(), outputs()
    mx, lse = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.beam_topk_step_fused(ld, b, k, lpsd, lensd, find, pen, END, *ref, ws, mx, lse)
    mx2, lse2 = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.beam_topk_step_tiles(ld, stats, b, k, lpsd, lensd, find, pen, END, *got, ws, mx2, lse2)
    live = ~fin.reshape(-1)
    assert torch.equal(mx2.cpu()[live], mx.cpu()[live])
    assert float((lse2.cpu()[live] - lse.cpu()[live]).abs().max()) < 2e-6 * max(1.0, float(lse.cpu()[live].abs().max()))
    names = ["score", "word", "beam", "logprob_sum", "lengths", "finished", "src_row"]
    for name, g, r in zip(names, got, ref):
        if g.dtype == torch.int32:
            assert torch.equal(g.cpu(), r.cpu()), name
        else:
            scale = max(1.0, float(r.abs().max()))
            assert float((g.cpu() - r.cpu()).abs().max()) <= 2e-6 * scale, name
    return [g.cpu().numpy() for g in got]


@pytest.mark.parametrize("b,k,v", [(16, 5, 32000), (4, 8, 32000), (3, 3, 4096), (2, 5, 32768), (5, 2, 2048),
                                   (2, 5, 131072), (3, 4, 1000), (2, 3, 132)])
def test_tiles_equal_fused_on_random_rows(dev, b, k, v):
    rng = np.random.default_rng(b * 7 + k + v)
    logits = (rng.standard_normal((b * k, v)) * 4).astype(np.float32)
    _run_tiles_vs_fused(dev, logits, k, *_state(rng, b, k, finished_frac=0.25))
    _run_tiles_vs_fused(dev, logits, k, *_state(rng, b, k, first_step=True))


def test_tiles_exact_ties_and_one_ulp_neighbours(dev):
    rng = np.random.default_rng(5)
    b, k, v = 6, 5, 32000
    logits = (rng.standard_normal((b * k, v)) * 2).astype(np.float32)
    for r in range(b * k):
        top = np.float32(9.0 + r * 0.01)
        idx = rng.choice(v, size=12, replace=False)
        logits[r, idx[:4]] = top
        logits[r, idx[4:8]] = np.nextafter(top, np.float32(0))
        logits[r, idx[8:]] = np.nextafter(np.nextafter(top, np.float32(0)), np.float32(0))
    lps, lens, fin = _state(rng, b, k)
    lps[:] = lps[:, :1]
    lens[:] = lens[:, :1]
    _run_tiles_vs_fused(dev, logits, k, lps, lens, fin)


def test_tiles_overflowing_lists_take_the_full_path(dev):
    rng = np.random.default_rng(6)
    b, k, v = 3, 5, 32000
    logits = (rng.standard_normal((b * k, v))).astype(np.float32)
    logits[0, :] = 0.0                                             # every tile qualifies, every logit equal
    logits[1, rng.choice(v, size=700, replace=False)] = 30.0       # 700 equal maxima, spread over > 64 tiles
    logits[2, rng.choice(v, size=256, replace=False)] = 30.0
    logits[3, rng.choice(v, size=257, replace=False)] = 30.0
    logits[4] = np.float32(1e-3) * np.arange(v, dtype=np.float32)  # a ramp: neighbours far closer than the margin
    logits[5, 128 * 7:128 * 9] = 25.0                              # 256 equal maxima inside two tiles: list exactly full
    logits[6, 128 * 7:128 * 9 + 1] = 25.0                          # one more: overflow
    lps, lens, fin = _state(rng, b, k)
    out = _run_tiles_vs_fused(dev, logits, k, lps, lens, fin)
    assert (np.diff(out[0], axis=1) <= 0).all()


def test_tiles_huge_and_tiny_magnitudes(dev):
    rng = np.random.default_rng(8)
    b, k, v = 4, 5, 32000
    logits = (rng.standard_normal((b * k, v))).astype(np.float32)
    logits[:5] *= 1e4
    logits[5:10] *= 1e-6
    logits[10:15] += 1e5
    lps, lens, fin = _state(rng, b, k)
    lps[1] = np.float32(-3e4)
    _run_tiles_vs_fused(dev, logits, k, lps, lens, fin)
