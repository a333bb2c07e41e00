"""GPU parity of every libnmhip kernel against the CPU oracle (oracle/) on the
same seeded inputs.  fp32 tolerance: 1e-4 relative (BASELINE.json north_star);
integer / index outputs bit-exact."""
import numpy as np
import pytest
import torch

from oracle import nm_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6))


def T(x, dev, dtype=torch.float32):
    return torch.tensor(np.asarray(x), dtype=dtype, device=dev)


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("m,n,k,ta,tb,algo", [
    (128, 1024, 512, False, False, 0),      # decoder-step shape -> skinny
    (128, 512, 1024, False, True, 0),       # skinny NT
    (128, 1024, 512, False, False, 2),      # same through tiled 64
    (128, 1024, 512, False, False, 1),      # same through tiled 128
    (640, 512, 2048, False, False, 0),      # output projection shape
    (300, 260, 130, False, False, 0),       # ragged, non-vector path
    (300, 260, 130, True, True, 0),
    (259, 131, 77, True, False, 0),
    (96, 200, 64, False, True, 3),          # skinny forced, ragged M/N
    (1024, 2048, 256, True, False, 1),      # TN weight-gradient form
    (512, 384, 640, False, True, 1),
    (37, 1000, 8, False, False, 0),
    (80, 260, 520, False, False, 0),        # skinny, K = 16 * 16 * 2 + 8: the last eight k-values belong to the last wave
    (80, 264, 264, False, True, 0),         # K = 16 * 16 + 8
    (16, 64, 776, False, False, 3),         # K = 16 * 16 * 3 + 8, skinny forced
    (90, 520, 264, False, False, 0),
    (512, 1024, 6400, True, False, 0),      # weight-gradient form, deep K -> split-K slabs
    (640, 512, 8192, False, True, 0),       # dlogits . W^T form -> split-K
    (96, 136, 4096, True, True, 2),         # split-K on the 64 tile, ragged
    (128, 512, 1024, False, True, 3),       # skinny NT with a 64-deep slice per wave
    (128, 1536, 256, False, False, 3),      # skinny KS=8
    # a few hundred rows, too few 64x64 tiles for the chip: 32x32 K-split tiles (nm_medium_gemm, nm_step.hip)
    (640, 512, 512, False, False, 0),       # a Transformer beam-step projection, weights [K,N]
    (640, 2048, 512, False, False, 0),      # its feed-forward expansion (relu checked below)
    (640, 512, 2048, False, False, 0),
    (300, 96, 48, False, True, 0),          # ragged rows, a K slice that ends in the middle of a trip, weights [N,K]
    (1000, 64, 80, False, False, 0),
    (512, 32, 16, False, True, 0),          # K shorter than one trip (a 16-channel projection's input gradient)
    (512, 32, 272, False, False, 0),        # 8 full trips + half a trip: the half belongs to the LAST K slice
])
def test_gemm(dev, m, n, k, ta, tb, algo):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
    b = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    out = ops.gemm(T(a, dev), T(b, dev), trans_a=ta, trans_b=tb, algo=algo)
    assert rel_err(out.cpu().numpy(), ref) < 2e-6 * np.sqrt(k) + 1e-6
    a_s = (a / np.sqrt(k)).astype(np.float32)          # O(1) pre-activations for the tanh epilogue
    ref_s = (a_s.T if ta else a_s).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    out2 = ops.gemm(T(a_s, dev), T(b, dev), bias=T(bias, dev), act="tanh", trans_a=ta, trans_b=tb, algo=algo)
    assert rel_err(out2.cpu().numpy(), np.tanh(ref_s + bias)) < 1e-5
    c0 = rng.standard_normal((m, n)).astype(np.float32)
    out3 = T(c0, dev)
    ops.gemm(T(a, dev), T(b, dev), out=out3, trans_a=ta, trans_b=tb, accumulate=True, algo=algo)
    assert rel_err(out3.cpu().numpy(), ref + c0) < 2e-6 * np.sqrt(k) + 1e-6
    out4 = ops.gemm(T(a_s, dev), T(b, dev), bias=T(bias, dev), act="relu", trans_a=ta, trans_b=tb, algo=algo)
    assert rel_err(out4.cpu().numpy(), np.maximum(ref_s + bias, 0.0)) < 1e-5


def test_gemm_random_shapes(dev):
    """Seeded sweep over shapes nobody picked by hand: every layout, odd and aligned sizes, row strides larger than
    the row, bias / activation / accumulate, every kernel family (auto dispatch and forced), with and without the
    split-K workspace -- against float64."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(20260925)
    sizes = [1, 2, 3, 4, 5, 7, 8, 12, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128, 129, 200, 256, 260, 384, 511,
             512, 640, 1000, 1024, 1300, 2048]
    for case in range(60):
        m, n = (int(rng.choice(sizes)) for _ in range(2))
        k = int(rng.choice(sizes + [4096, 6400]))
        ta, tb = bool(rng.integers(2)), bool(rng.integers(2))
        algo = int(rng.choice([0, 0, 0, 1, 2, 3]))
        pad_a, pad_b, pad_c = (int(rng.choice([0, 0, 4, 3])) for _ in range(3))
        if algo == 3 and (ta or k % 8 or (k + pad_a) % 4 or (tb and (k + pad_b) % 4) or (not tb and (n + pad_b) % 4)):
            algo = 0            # the skinny kernels are only ever forced where they apply (they refuse loudly otherwise)
        scale = 1.0 / np.sqrt(k)
        a_full = (rng.standard_normal(((k, m + pad_a) if ta else (m, k + pad_a))) * scale).astype(np.float32)
        b_full = rng.standard_normal(((n, k + pad_b) if tb else (k, n + pad_b))).astype(np.float32)
        a_np = a_full[:, :m] if ta else a_full[:, :k]
        b_np = b_full[:, :k] if tb else b_full[:, :n]
        a_d = T(a_full, dev)[:, :a_np.shape[1]]
        b_d = T(b_full, dev)[:, :b_np.shape[1]]
        bias = rng.standard_normal(n).astype(np.float32) if rng.integers(2) else None
        act = [None, "tanh", "relu"][int(rng.integers(3))]
        acc = bool(rng.integers(2)) and act is None
        ref = (a_np.T if ta else a_np).astype(np.float64) @ (b_np.T if tb else b_np).astype(np.float64)
        if bias is not None:
            ref = ref + bias
        c0 = rng.standard_normal((m, n + pad_c)).astype(np.float32)
        out_full = T(c0 if acc else np.full_like(c0, np.nan), dev)
        out = out_full[:, :n]
        if acc:
            ref = ref + c0[:, :n]
        ref = {"tanh": np.tanh, "relu": lambda x: np.maximum(x, 0), None: lambda x: x}[act](ref)
        ops.gemm(a_d, b_d, out=out, bias=None if bias is None else T(bias, dev), act=act, trans_a=ta, trans_b=tb,
                 accumulate=acc, algo=algo)
        got = out_full.cpu().numpy()
        what = "case {}: m={} n={} k={} ta={} tb={} algo={} pads=({},{},{}) bias={} act={} acc={}".format(
            case, m, n, k, ta, tb, algo, pad_a, pad_b, pad_c, bias is not None, act, acc)
        assert rel_err(got[:, :n], ref) < 3e-5, what
        if pad_c and not acc:
            assert np.isnan(got[:, n:]).all(), what               # nothing beyond the N columns is written


def test_gemm_transpose_detecting(dev):
    """A = I with an asymmetric B catches a swapped C layout."""
    from neuralmonkey_amd import ops
    for n in (96, 256):                  # 256: the skinny path switches to its 16x16-tile kernel
        a = np.eye(n, dtype=np.float32)
        b = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 97) * 0.25
        for algo in (1, 2, 3):
            out = ops.gemm(T(a, dev), T(b, dev), algo=algo)
            assert np.array_equal(out.cpu().numpy(), b)
            out = ops.gemm(T(a, dev), T(np.ascontiguousarray(b.T), dev), trans_b=True, algo=algo)
            assert np.array_equal(out.cpu().numpy(), b)


def test_gemm_batched_strided(dev):
    """Per-sentence attention-backward forms: batch over b of [T,C]x[C,S]."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(5)
    t, b, c, s = 50, 6, 1024, 50
    dctx = rng.standard_normal((t, b, c)).astype(np.float32)
    states = rng.standard_normal((b, s, c)).astype(np.float32)
    w = rng.standard_normal((t, b, s)).astype(np.float32)
    d_dctx, d_states, d_w = T(dctx, dev), T(states, dev), T(w, dev)
    dw = torch.empty((t, b, s), dtype=torch.float32, device=dev)
    ops.gemm(d_dctx.permute(1, 0, 2), d_states, out=dw.permute(1, 0, 2), trans_b=True)
    ref = np.einsum("tbc,bsc->tbs", dctx.astype(np.float64), states.astype(np.float64))
    assert rel_err(dw.cpu().numpy(), ref) < 1e-5
    dst = torch.empty((b, s, c), dtype=torch.float32, device=dev)
    ops.gemm(d_w.permute(1, 0, 2), d_dctx.permute(1, 0, 2), out=dst, trans_a=True)
    ref2 = np.einsum("tbs,tbc->bsc", w.astype(np.float64), dctx.astype(np.float64))
    assert rel_err(dst.cpu().numpy(), ref2) < 1e-5


# --------------------------------------------------------------------------- #
def test_embedding_gather(dev):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((100, 512)).astype(np.float32)
    ids = rng.integers(0, 100, size=(7, 9)).astype(np.int32)
    ids[2, 4:] = 0
    ref, _ = O.embedded_sequence(emb, ids)
    got = ops.embedding_gather(T(emb, dev), T(ids, dev, torch.int32), mask_pad=True)
    assert np.array_equal(got.cpu().numpy(), ref)
    got2 = ops.embedding_gather(T(emb, dev), T(ids, dev, torch.int32))
    assert np.array_equal(got2.cpu().numpy(), emb[ids])
    emb2 = rng.standard_normal((50, 10)).astype(np.float32)     # E % 4 != 0
    got3 = ops.embedding_gather(T(emb2, dev), T(ids % 50, dev, torch.int32))
    assert np.array_equal(got3.cpu().numpy(), emb2[ids % 50])


def test_layer_norm(dev):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(1)
    for d in (1024, 14, 513):
        x = rng.standard_normal((33, d)).astype(np.float32) * 3 + 1
        x[5] = 0.0                                   # padded encoder row -> beta
        g = rng.standard_normal(d).astype(np.float32)
        b = rng.standard_normal(d).astype(np.float32)
        got = ops.layer_norm_fwd(T(x, dev), T(g, dev), T(b, dev))
        ref = O.layer_norm(x.astype(np.float64), g.astype(np.float64), b.astype(np.float64))
        assert rel_err(got.cpu().numpy(), ref) < 1e-5


def test_add_layer_norm_is_add_then_layer_norm(dev):
    """nm_add_layer_norm_fwd: the residual sum and its layer norm, bit for bit what the two separate launches give."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(11)
    for rows, d in ((640, 512), (7, 14), (33, 1030)):
        a = T(rng.standard_normal((rows, d)).astype(np.float32) * 2, dev)
        x = T(rng.standard_normal((rows, d)).astype(np.float32) * 3 + 1, dev)
        g, b = T(rng.standard_normal(d).astype(np.float32), dev), T(rng.standard_normal(d).astype(np.float32), dev)
        want_sum = ops.ew("add", a, x, torch.empty_like(x))
        want = ops.layer_norm_fwd(want_sum, g, b)
        got_sum, got = ops.add_layer_norm_fwd(a, x, g, b, torch.empty_like(x), torch.empty_like(x))
        assert torch.equal(got_sum, want_sum) and torch.equal(got, want)


# --------------------------------------------------------------------------- #
def _gru_params(rng, d_in, h, std=0.3):
    return {"gates_kernel": (rng.standard_normal((d_in + h, 2 * h)) * std).astype(np.float32),
            "gates_bias": np.ones(2 * h, np.float32),
            "cand_kernel": (rng.standard_normal((d_in + h, h)) * std).astype(np.float32),
            "cand_bias": (rng.standard_normal(h) * 0.1).astype(np.float32)}


def test_gru_decoder_step(dev):
    """Split GRU (x-part hoisted) == TF GRUCell on concatenated [x,h]."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(2)
    b, e, h = 128, 512, 512
    p = _gru_params(rng, e, h, 0.05)
    x = rng.standard_normal((b, e)).astype(np.float32)
    h0 = rng.standard_normal((b, h)).astype(np.float32)
    ref = O.gru_cell(x.astype(np.float64), h0.astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()})
    wx = np.concatenate([p["gates_kernel"][:e], p["cand_kernel"][:e]], 1)
    bx = np.concatenate([p["gates_bias"], p["cand_bias"]])
    xp = ops.gemm(T(x, dev), T(wx, dev), bias=T(bx, dev))
    hd = T(h0, dev)
    hg = ops.gemm(hd, T(p["gates_kernel"][e:], dev))
    ru = torch.empty((b, 2 * h), device=dev)
    rh = torch.empty((b, h), device=dev)
    ops.gru_gates_fwd(xp, 0, 3 * h, 0, hg, hd, ru, rh, None, 0, 1, b, h)
    hc = ops.gemm(rh, T(p["cand_kernel"][e:], dev))
    hn = torch.empty((b, h), device=dev)
    ops.gru_blend_fwd(xp, 0, 3 * h, 0, hc, ru, hd, hn, None, None, 0, 0, 0, None, 0, 1, b, h)
    assert rel_err(hn.cpu().numpy(), ref) < RTOL


def test_gru_bidirectional_encoder(dev):
    """Length-masked biGRU sequence == bidirectional_dynamic_rnn restatement."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(3)
    b, s, e, h = 9, 11, 16, 12
    pf, pb = _gru_params(rng, e, h), _gru_params(rng, e, h)
    x = rng.standard_normal((b, s, e)).astype(np.float32)
    lengths = np.array([11, 1, 5, 11, 7, 2, 10, 3, 6], dtype=np.int32)
    for i, ln in enumerate(lengths):
        x[i, ln:] = 0
    ref_states, ref_final = O.bidirectional_rnn(O.gru_cell, x, lengths, pf, pb)
    wx = np.concatenate([pf["gates_kernel"][:e], pf["cand_kernel"][:e],
                         pb["gates_kernel"][:e], pb["cand_kernel"][:e]], 1)       # [E, 6H]
    bx = np.concatenate([pf["gates_bias"], pf["cand_bias"], pb["gates_bias"], pb["cand_bias"]])
    xp = ops.gemm(T(x.reshape(b * s, e), dev), T(wx, dev), bias=T(bx, dev))      # [B*S, 6H]
    wgh = T(np.stack([pf["gates_kernel"][e:], pb["gates_kernel"][e:]]), dev)      # [2,H,2H]
    wch = T(np.stack([pf["cand_kernel"][e:], pb["cand_kernel"][e:]]), dev)        # [2,H,H]
    hcur = torch.zeros((2, b, h), device=dev)
    out = torch.zeros((b, s, 2 * h), device=dev)
    hg = torch.empty((2, b, 2 * h), device=dev)
    ru = torch.empty((2, b, 2 * h), device=dev)
    rh = torch.empty((2, b, h), device=dev)
    hc = torch.empty((2, b, h), device=dev)
    ld = T(lengths, dev, torch.int32)
    for t in range(s):
        ops.gemm(hcur, wgh, out=hg)
        ops.gru_gates_fwd(xp, 3 * h, s * 6 * h, 6 * h, hg, hcur, ru, rh, ld, t, 2, b, h)
        ops.gemm(rh, wch, out=hc)
        ops.gru_blend_fwd(xp, 3 * h, s * 6 * h, 6 * h, hc, ru, hcur, hcur, None, out,
                          h, s * 2 * h, 2 * h, ld, t, 2, b, h)
    assert rel_err(out.cpu().numpy(), ref_states) < RTOL
    fin = torch.cat([hcur[0], hcur[1]], 1).cpu().numpy()
    assert rel_err(fin, ref_final) < RTOL
    assert np.all(out.cpu().numpy()[1, 1:] == 0)          # beyond length stays zero


def _split_gru_ref(xp, wgh, wch, lengths, h0=None):
    """torch restatement of the biGRU on pre-projected inputs (the arithmetic of O.gru_cell /
    O.bidirectional_rnn with the input half hoisted): xp [B,S,ndir,3H] = x.[Wg_x|Wc_x]+[bg|bc];
    direction 1 walks each sentence backwards over its own length (reverse_sequence).  Returns
    (states [B,S,ndir*H], final [ndir,B,H])."""
    b, s, ndir, h3 = xp.shape
    h = h3 // 3
    outs = [[None] * s for _ in range(ndir)]
    finals = []
    ar = torch.arange(b)
    for d in range(ndir):
        hcur = xp.new_zeros(b, h) if h0 is None else h0[d]
        per_pos = [[] for _ in range(s)]
        for t in range(s):
            live = t < lengths
            pos = torch.where(live, (lengths - 1 - t) if d == 1 else torch.full_like(lengths, t),
                              torch.zeros_like(lengths))
            x_t = xp[ar, pos, d]
            g = torch.sigmoid(x_t[:, :2 * h] + hcur @ wgh[d])
            r, u = g[:, :h], g[:, h:]
            c = torch.tanh(x_t[:, 2 * h:] + (r * hcur) @ wch[d])
            hn = u * hcur + (1 - u) * c
            hcur = torch.where(live[:, None], hn, hcur)
            per_pos[t] = (pos, live, hn)
        rows = []
        for i in range(b):
            row = [xp.new_zeros(h) for _ in range(s)]
            for t in range(s):
                pos, live, hn = per_pos[t]
                if bool(live[i]):
                    row[int(pos[i])] = hn[i]
            rows.append(torch.stack(row))
        outs[d] = torch.stack(rows)                   # [B,S,H]
        finals.append(hcur)
    return torch.cat(outs, 2), torch.stack(finals)


@pytest.mark.parametrize("bk,qpk", [(128, 1), (32, 5), (7, 3)])
def test_attention_in_kernel_merge_survives_buffer_reuse(dev, bk, qpk):
    """The split-S partials are handed to the last-arriving chunk workgroup INSIDE the launch (write-through
    stores, one arrival counter per sentence, one agent-scope acquire): the same device buffers -- keys, queries,
    workspace, outputs -- are refilled with new values and the step is re-run back to back, with other kernels
    dirtying the caches in between; a consumer that read a stale partial (its own L1, another XCD's L2) would
    reproduce the previous round's context.  Every round is checked against the float64 oracle
    (attention/feed_forward.py:120-166) and the arrival counters must be back at zero."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(bk + qpk)
    s, a, c = 50, 1024, 1024
    r = bk * qpk
    yd, hfd, std = (torch.empty(shape, device=dev) for shape in ((r, a), (bk, s, a), (bk, s, c)))
    maskd = torch.empty((bk, s), device=dev)
    vd, biasd = torch.empty(a, device=dev), torch.empty(1, device=dev)
    ctx, w = torch.empty((r, c), device=dev), torch.empty((r, s), device=dev)
    ws = ops.attn_workspace(r, s, c, dev)
    nbytes = ops._lib.load().nm_attn_workspace_bytes(r, s, c)
    junk = torch.empty(64 << 20, device=dev)
    f64 = lambda x: np.asarray(x, dtype=np.float64)
    for rnd in range(5):
        y = (rng.standard_normal((r, a)) * 0.5).astype(np.float32)
        hf = rng.standard_normal((bk, s, a)).astype(np.float32)
        states = rng.standard_normal((bk, s, c)).astype(np.float32) * (rnd + 1)
        mask = np.ones((bk, s), np.float32)
        for i in range(bk):
            mask[i, rng.integers(1, s + 1):] = 0
        v = (rng.standard_normal(a) * 0.3).astype(np.float32)
        for dst, src in ((yd, y), (hfd, hf), (std, states), (maskd, mask), (vd, v), (biasd, np.float32([0.1 * rnd]))):
            dst.copy_(torch.from_numpy(np.ascontiguousarray(src)))
        for rep in range(3):                                  # back to back on the same buffers
            junk.fill_(float(rep))                            # 256 MB of dirty lines between the launches
            ops.attn_fwd(yd, hfd, std, maskd, vd, biasd, qpk, ctx, w, ws)
        e = (f64(v) * np.tanh(np.repeat(f64(hf), qpk, 0) + f64(y)[:, None, :])).sum(-1) + 0.1 * rnd
        sm = np.exp(e - e.max(1, keepdims=True))
        sm /= sm.sum(1, keepdims=True)
        wm = sm * np.repeat(f64(mask), qpk, 0)
        ref_w = wm / (wm.sum(1, keepdims=True) + 1e-8)
        ref_ctx = np.einsum("rs,rsc->rc", ref_w, np.repeat(f64(states), qpk, 0))
        assert rel_err(w.cpu().numpy(), ref_w) < RTOL, rnd
        assert rel_err(ctx.cpu().numpy(), ref_ctx) < RTOL, rnd
        tail = ws.view(torch.int32)[nbytes // 4 - ((r + 3) // 4) * 4:]
        assert int(tail.abs().sum().item()) == 0, "arrival counters not restored"


@pytest.mark.parametrize("b,s,e,h", [(128, 6, 512, 512), (16, 5, 64, 64), (128, 4, 256, 1024)])
def test_gru_gemm_fused_epilogues_fwd_bwd(dev, b, s, e, h):
    """nm_gru_gemm modes 1-4 (recurrent GEMM + fused gate / blend epilogues, forward and BPTT) at the
    headline shape R=128, H=512, both directions, ragged lengths: forward against O.bidirectional_rnn
    (nn/ortho_gru_cell.py:44-53, encoders/recurrent.py:86-102), backward against autograd of the
    restated forward."""
    from neuralmonkey_amd import ops
    from neuralmonkey_amd.nn import gru
    rng = np.random.default_rng(b + s + h)
    ndir = 2
    pf, pb = _gru_params(rng, e, h, 0.05), _gru_params(rng, e, h, 0.05)
    x = rng.standard_normal((b, s, e)).astype(np.float32)
    lengths = rng.integers(1, s + 1, size=b).astype(np.int32)
    lengths[0] = s
    for i, ln in enumerate(lengths):
        x[i, ln:] = 0
    ref_states, ref_final = O.bidirectional_rnn(O.gru_cell, x, lengths, pf, pb)
    wx = np.concatenate([pf["gates_kernel"][:e], pf["cand_kernel"][:e],
                         pb["gates_kernel"][:e], pb["cand_kernel"][:e]], 1)       # [E, 6H]
    bx = np.concatenate([pf["gates_bias"], pf["cand_bias"], pb["gates_bias"], pb["cand_bias"]])
    xp = ops.gemm(T(x.reshape(b * s, e), dev), T(wx, dev), bias=T(bx, dev))      # [B*S, ndir*3H]
    wgh_np = np.stack([pf["gates_kernel"][e:], pb["gates_kernel"][e:]])
    wch_np = np.stack([pf["cand_kernel"][e:], pb["cand_kernel"][e:]])
    wgh, wch = T(wgh_np, dev), T(wch_np, dev)
    assert gru.fused_ok(b, h)
    c_out = ndir * h
    hcur = torch.zeros((ndir, b, h), device=dev)
    states = torch.zeros((b, s, c_out), device=dev)
    ru_all = torch.empty((s, ndir, b, 2 * h), device=dev)
    c_all = torch.empty((s, ndir, b, h), device=dev)
    rh = torch.empty((ndir, b, h), device=dev)
    ld = T(lengths, dev, torch.int32)
    xrs, xts, ors, ots = s * ndir * 3 * h, ndir * 3 * h, s * c_out, c_out
    for t in range(s):
        gru.step_fwd(xp, (3 * h, xrs, xts), hcur, hcur, wgh, wch, ru_all[t], rh, c_all[t], states,
                     (h, ors, ots), ld, t, ndir, b, h, False, None, None)
    assert rel_err(states.cpu().numpy(), ref_states) < RTOL
    assert rel_err(torch.cat([hcur[0], hcur[1]], 1).cpu().numpy(), ref_final) < RTOL

    # ---- backward: random upstream gradients on the states and on the final states
    d_states = rng.standard_normal((b, s, c_out)).astype(np.float32)
    for i, ln in enumerate(lengths):
        d_states[i, ln:] = 0                              # dead positions emit zeros: no gradient path
    d_final = rng.standard_normal((ndir, b, h)).astype(np.float32)
    xp_ref = torch.tensor(xp.cpu().numpy().astype(np.float64).reshape(b, s, ndir, 3 * h), requires_grad=True)
    wg_ref = torch.tensor(wgh_np.astype(np.float64), requires_grad=True)
    wc_ref = torch.tensor(wch_np.astype(np.float64), requires_grad=True)
    st, fin = _split_gru_ref(xp_ref, wg_ref, wc_ref, torch.tensor(lengths.astype(np.int64)))
    assert rel_err(st.detach().numpy(), ref_states) < 1e-5        # the restatement IS the oracle's forward
    ((st * torch.tensor(d_states.astype(np.float64))).sum()
     + (fin * torch.tensor(d_final.astype(np.float64))).sum()).backward()
    dh = T(d_final, dev)
    dxp = torch.zeros((b * s, ndir * 3 * h), device=dev)
    dgpre = torch.empty((2, ndir, b, 2 * h), device=dev)
    dcpre = torch.empty((ndir, b, h), device=dev)
    drh = torch.empty((ndir, b, h), device=dev)
    seq_strides = (h, s * c_out, c_out)
    gru.bptt(s, dh, T(d_states, dev), seq_strides, ru_all, c_all, None, states, seq_strides, dxp,
             (3 * h, xrs, xts), wgh, wch, ld, ndir, b, h, False, dgpre, dcpre, drh)
    want = xp_ref.grad.numpy().reshape(b * s, ndir * 3 * h)
    assert rel_err(dxp.cpu().numpy(), want) < 1e-4
    # weight gradients assembled from the kernel's pre-activation gradients == autograd's
    hprev = torch.empty((b, s, ndir, h), device=dev)
    rh_seq = torch.empty((b, s, ndir, h), device=dev)
    ops.gru_seq_shift(states, hprev, ld, ndir, h)
    ops.gru_rh_seq(ru_all, hprev, rh_seq, ld, ndir, h)
    for d in range(ndir):
        dg = dxp[:, d * 3 * h:d * 3 * h + 2 * h]
        dc = dxp[:, d * 3 * h + 2 * h:(d + 1) * 3 * h]
        g_wg = ops.gemm(hprev.view(b * s, c_out)[:, d * h:(d + 1) * h], dg, trans_a=True)
        g_wc = ops.gemm(rh_seq.view(b * s, c_out)[:, d * h:(d + 1) * h], dc, trans_a=True)
        assert rel_err(g_wg.cpu().numpy(), wg_ref.grad[d].numpy()) < 2e-4
        assert rel_err(g_wc.cpu().numpy(), wc_ref.grad[d].numpy()) < 2e-4


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("bk,qpk,s,a,c,ragged", [
    (16, 1, 50, 1024, 1024, False),
    (16, 1, 50, 1024, 1024, True),
    (128, 1, 50, 1024, 1024, True),      # headline step: whole-sentence workgroups (>= 96 sentences, 40..52 positions)
    (100, 1, 41, 200, 72, True),         # the same kernel with partial column waves and a short last row group
    (96, 1, 52, 1024, 512, False),
    (4, 5, 50, 1024, 1024, True),
    (128, 5, 50, 1024, 1024, True),      # the beam step as benchmarked
    (100, 3, 47, 600, 520, True),        # 3 queries on the 4-query instance, partial column waves
    (96, 2, 52, 1024, 1024, False),
    (97, 4, 41, 256, 1024, True),
    (1, 5, 50, 1024, 1024, True),        # reference-compatible batch-1 beam
    (3, 3, 7, 64, 32, True),             # single chunk, tiny dims
    (5, 1, 64, 128, 2048, False),        # captioning shape: S=64, C=2048, state_size 128
    (128, 1, 64, 512, 2048, True),       # config 4 as benchmarked: two whole-sentence workgroups per sentence, split by value columns
    (70, 1, 61, 300, 1540, True),        # the same kernel: partial key / value column waves, a short last row group
    (4, 1, 120, 256, 256, True),         # long sources: 10 chunks -> too many for the in-kernel merge, combine launch
    (2, 5, 200, 128, 128, True),         # 17 chunks, five queries per sentence
    (6, 1, 96, 512, 1024, True),         # 8 chunks: the most the in-kernel merge takes
    (2, 8, 13, 512, 1024, True),
    (2, 2, 100, 256, 512, True),
    (2, 3, 50, 256, 256, True),          # beam widths that run a wider register instance (3 -> 4, 6 / 7 -> 8)
    (3, 4, 24, 512, 256, True),
    (2, 6, 30, 128, 512, False),
    (2, 7, 50, 1024, 1024, True),
    (2, 8, 50, 640, 1024, True),
    (3, 12, 50, 1024, 1024, True),       # beams wider than one query group: two groups of 8 + 4
    (2, 16, 23, 256, 512, True),
    (3, 2, 9, 10, 14, True),             # tests/small.ini-like: C = 2*7, no dim a multiple of 4
    (2, 1, 300, 6, 7, True),             # any-shape kernel, long S
])
def test_attention_fwd(dev, bk, qpk, s, a, c, ragged):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(bk * 100 + qpk * 10 + s)
    r = bk * qpk
    q = rng.standard_normal((r, 48)).astype(np.float32)
    states = rng.standard_normal((bk, s, c)).astype(np.float32)
    wk = (rng.standard_normal((c, a)) * 0.05).astype(np.float32)
    ap = {"query_w": (rng.standard_normal((48, a)) * 0.2).astype(np.float32),
          "query_b": (rng.standard_normal(a) * 0.1).astype(np.float32),
          "v": rng.standard_normal(a).astype(np.float32) * 0.3,
          "bias": np.float32(0.37)}
    mask = np.ones((bk, s), np.float32)
    if ragged:
        for i in range(bk):
            mask[i, rng.integers(1, s + 1):] = 0
    hf = O.attention_keys(states, wk)
    f64 = lambda x: np.asarray(x, dtype=np.float64)
    ref_ctx, ref_w = O.attention_step(f64(q), np.repeat(f64(hf), qpk, 0), np.repeat(f64(states), qpk, 0),
                                      np.repeat(f64(mask), qpk, 0), {k: f64(v) for k, v in ap.items()})
    y = ops.gemm(T(q, dev), T(ap["query_w"], dev), bias=T(ap["query_b"], dev))
    ctx = torch.empty((r, c), device=dev)
    w = torch.empty((r, s), device=dev)
    ws = ops.attn_workspace(r, s, c, dev)
    ops.attn_fwd(y, T(hf, dev), T(states, dev), T(mask, dev), T(ap["v"], dev),
                 T(np.array([ap["bias"]]), dev), qpk, ctx, w, ws)
    assert rel_err(w.cpu().numpy(), ref_w) < RTOL
    assert rel_err(ctx.cpu().numpy(), ref_ctx) < RTOL
    assert np.all(w.cpu().numpy()[np.repeat(mask, qpk, 0) == 0] == 0)


def test_attention_fwd_random_shapes(dev):
    """Seeded sweep of the attention step over sizes nobody picked by hand (every kernel: whole-sentence, split-S with
    in-kernel merge, split-S + combine, several queries per sentence, any-shape) against float64."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(424242)
    for case in range(40):
        qpk = int(rng.choice([1, 1, 1, 2, 3, 5, 8, 11]))
        bk = int(rng.choice([1, 2, 3, 7, 33, 97, 130])) if qpk == 1 else int(rng.choice([1, 2, 5, 9]))
        s = int(rng.choice([1, 2, 5, 9, 13, 26, 39, 40, 47, 52, 53, 64, 97, 130]))
        a = int(rng.choice([4, 6, 32, 100, 256, 512, 1024]))
        c = int(rng.choice([4, 7, 48, 256, 516, 1024, 2048]))
        r = bk * qpk
        y = rng.standard_normal((r, a)).astype(np.float32)
        hf = rng.standard_normal((bk, s, a)).astype(np.float32)
        states = rng.standard_normal((bk, s, c)).astype(np.float32)
        v = (rng.standard_normal(a) * 0.3).astype(np.float32)
        mask = (np.arange(s)[None, :] < rng.integers(1, s + 1, bk)[:, None]).astype(np.float32)
        e = (v.astype(np.float64) * np.tanh(np.repeat(hf, qpk, 0).astype(np.float64) + y[:, None, :])).sum(-1) + 0.25
        p = np.exp(e - e.max(1, keepdims=True))
        w_all = p / p.sum(1, keepdims=True) * np.repeat(mask, qpk, 0)
        ref_w = w_all / (w_all.sum(1, keepdims=True) + 1e-8)
        ref_ctx = (ref_w[:, :, None] * np.repeat(states, qpk, 0).astype(np.float64)).sum(1)
        ctx = torch.full((r, c + 4), float("nan"), device=dev)
        w = torch.empty((r, s), device=dev)
        en = torch.empty((r, s), device=dev)
        ws = ops.attn_workspace(r, s, c, dev)
        for _ in range(2):                      # twice: arrival counters left at zero
            ops.attn_fwd(T(y, dev), T(hf, dev), T(states, dev), T(mask, dev), T(v, dev), T(np.array([0.25]), dev), qpk,
                         ctx[:, :c], w, ws, en)
        what = "case {}: bk={} qpk={} s={} a={} c={}".format(case, bk, qpk, s, a, c)
        assert rel_err(en.cpu().numpy(), e) < RTOL, what
        assert np.abs(w.cpu().numpy() - ref_w).max() < 2e-6, what
        assert rel_err(ctx[:, :c].cpu().numpy(), ref_ctx) < RTOL, what
        assert torch.isnan(ctx[:, c:]).all(), what


@pytest.mark.parametrize("whole", ["0", "1"])
def test_attention_fwd_under_either_dispatch(whole):
    """The one-query step has two kernels (split-S + in-kernel merge / one workgroup per sentence) chosen by shape;
    NM_ATTN_WHOLE forces either wherever it applies: every shape above passes under both (a fresh process each,
    the switch is read once)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NM_ATTN_WHOLE=whole)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "pytest", "tests/test_kernels_gpu.py", "-q", "-x", "-k",
                          "test_attention_fwd and not dispatch or all_masked_row or merge_survives"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]


@pytest.mark.parametrize("t,b,s,a,c", [
    (50, 16, 50, 1024, 1024),            # teacher-forced training shape (fewer sentences)
    (7, 3, 13, 64, 32),                  # one query group, single chunk
    (11, 5, 100, 256, 512),              # query groups of 8 + a ragged tail of 3
    (1, 4, 20, 128, 128),
])
def test_attention_time_major_all_steps(dev, t, b, s, a, c):
    """nm_attn_fwd_multi, time-major layout: all T teacher-forced queries of a sentence in
    one launch must equal T independent attention steps (attention/feed_forward.py:66-104)."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(t * 1000 + b * 10 + s)
    q = rng.standard_normal((t, b, 48)).astype(np.float32)
    states = rng.standard_normal((b, s, c)).astype(np.float32)
    wk = (rng.standard_normal((c, a)) * 0.05).astype(np.float32)
    ap = {"query_w": (rng.standard_normal((48, a)) * 0.2).astype(np.float32),
          "query_b": (rng.standard_normal(a) * 0.1).astype(np.float32),
          "v": rng.standard_normal(a).astype(np.float32) * 0.3,
          "bias": np.float32(-0.21)}
    mask = np.ones((b, s), np.float32)
    for i in range(b):
        mask[i, rng.integers(1, s + 1):] = 0
    hf = O.attention_keys(states, wk)
    f64 = lambda x: np.asarray(x, dtype=np.float64)
    ap64 = {k: f64(v) for k, v in ap.items()}
    ref = [O.attention_step(f64(q[i]), f64(hf), f64(states), f64(mask), ap64) for i in range(t)]
    ref_ctx = np.stack([r[0] for r in ref])
    ref_w = np.stack([r[1] for r in ref])
    y_all = ops.gemm(T(q.reshape(t * b, 48), dev), T(ap["query_w"], dev),
                     bias=T(ap["query_b"], dev)).view(t, b, a)
    ctx = torch.empty((t, b, c), device=dev)
    w = torch.empty((t, b, s), device=dev)
    e = torch.empty((t, b, s), device=dev)
    ops.attn_fwd_time_major(y_all, T(hf, dev), T(states, dev), T(mask, dev), T(ap["v"], dev),
                            T(np.array([ap["bias"]]), dev), ctx, w, ops.attn_workspace(t * b, s, c, dev), e)
    assert rel_err(w.cpu().numpy(), ref_w) < RTOL
    assert rel_err(ctx.cpu().numpy(), ref_ctx) < RTOL
    assert np.all(w.cpu().numpy()[:, mask == 0] == 0)
    # the same numbers as T single-step launches of the decoding kernel
    ctx1 = torch.empty((b, c), device=dev)
    w1 = torch.empty((b, s), device=dev)
    ws1 = ops.attn_workspace(b, s, c, dev)
    for i in (0, t - 1):
        ops.attn_fwd(y_all[i], T(hf, dev), T(states, dev), T(mask, dev), T(ap["v"], dev),
                     T(np.array([ap["bias"]]), dev), 1, ctx1, w1, ws1)
        assert rel_err(w1.cpu().numpy(), w[i].cpu().numpy()) < 1e-5
        assert rel_err(ctx1.cpu().numpy(), ctx[i].cpu().numpy()) < 1e-5


@pytest.mark.parametrize("bk,qpk,t", [(4, 5, 0), (3, 1, 6), (100, 5, 0)])
def test_attention_beyond_the_product_form_range(dev, bk, qpk, t):
    """The multi-query kernels evaluate tanh(hf + y) as 1 - 2 / (1 + exp(2 hf) exp(2 y)) -- exact only while
    exp(2 x) stays inside fp32 (|x| <= 43).  Keys and queries of magnitude ~55 that nearly cancel must come out
    like everything else (beam queries per sentence: qpk = 5; all steps of a sentence: t = 6)."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(bk * 17 + qpk)
    s, a, c = 50, 1024, 1024
    states = rng.standard_normal((bk, s, c)).astype(np.float32)
    hf = rng.standard_normal((bk, s, a)).astype(np.float32)
    hf[:, ::3, ::5] += 55.0
    hf[:, 1::4, 1::7] -= 60.0
    mask = np.ones((bk, s), np.float32)
    mask[0, 40:] = 0
    v = (rng.standard_normal(a) * 0.3).astype(np.float32)
    nq = t or qpk
    y = rng.standard_normal((bk * nq if not t else t * bk, a)).astype(np.float32)
    y[:, ::5] -= 54.0
    y[1::2, 1::7] += 59.0
    f64 = lambda x: np.asarray(x, dtype=np.float64)

    def ref(yrows, b):
        e = np.tanh(f64(hf[b])[None] + f64(yrows)[:, None, :]) @ f64(v) + 0.37        # [nq, S]
        ex = np.exp(e - e.max(1, keepdims=True))
        sm = ex / ex.sum(1, keepdims=True)
        w = sm * f64(mask[b])
        w = w / (w.sum(1, keepdims=True) + 1e-8)
        return w @ f64(states[b]), w
    bias = T(np.array([0.37], np.float32), dev)
    if t:
        ctx = torch.empty((t, bk, c), device=dev)
        w = torch.empty((t, bk, s), device=dev)
        e = torch.empty((t, bk, s), device=dev)
        ops.attn_fwd_time_major(T(y, dev).view(t, bk, a), T(hf, dev), T(states, dev), T(mask, dev), T(v, dev), bias,
                                ctx, w, ops.attn_workspace(t * bk, s, c, dev), e)
        got_ctx, got_w = ctx.cpu().numpy(), w.cpu().numpy()
        for b in range(bk):
            rc, rw = ref(y.reshape(t, bk, a)[:, b], b)
            assert rel_err(got_w[:, b], rw) < RTOL and rel_err(got_ctx[:, b], rc) < RTOL
    else:
        ctx = torch.empty((bk * qpk, c), device=dev)
        w = torch.empty((bk * qpk, s), device=dev)
        ops.attn_fwd(T(y, dev), T(hf, dev), T(states, dev), T(mask, dev), T(v, dev), bias, qpk, ctx, w,
                     ops.attn_workspace(bk * qpk, s, c, dev))
        got_ctx, got_w = ctx.cpu().numpy(), w.cpu().numpy()
        assert np.isfinite(got_ctx).all()
        for b in range(bk):
            rc, rw = ref(y[b * qpk:(b + 1) * qpk], b)
            assert rel_err(got_w[b * qpk:(b + 1) * qpk], rw) < RTOL and rel_err(got_ctx[b * qpk:(b + 1) * qpk], rc) < RTOL


def test_attention_all_masked_row(dev):
    """A fully masked sentence gives zero weights (0/(0+1e-8)), not NaN."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(9)
    bk, s, a, c = 2, 20, 64, 64
    y = T(rng.standard_normal((bk, a)), dev)
    hf = T(rng.standard_normal((bk, s, a)), dev)
    st = T(rng.standard_normal((bk, s, c)), dev)
    mask = np.ones((bk, s), np.float32)
    mask[1] = 0
    ctx = torch.empty((bk, c), device=dev)
    w = torch.empty((bk, s), device=dev)
    ops.attn_fwd(y, hf, st, T(mask, dev), T(rng.standard_normal(a), dev), None, 1, ctx, w,
                 ops.attn_workspace(bk, s, c, dev))
    assert np.all(w.cpu().numpy()[1] == 0) and np.all(ctx.cpu().numpy()[1] == 0)
    assert abs(w.cpu().numpy()[0].sum() - 1) < 1e-5


# --------------------------------------------------------------------------- #
def test_row_stats_and_argmax_ties(dev):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(4)
    r, v = 37, 32000
    x = (rng.standard_normal((r, v)) * 3).astype(np.float32)
    x[3, 100] = x[3, 31999] = 50.0       # tie: first index wins (tf.argmax)
    x[4, :] = 1.0                        # all equal -> index 0
    xd = T(x, dev)
    mx = torch.empty(r, device=dev)
    lse = torch.empty(r, device=dev)
    am = torch.empty(r, dtype=torch.int32, device=dev)
    ops.row_stats(xd, mx, lse, am)
    assert np.array_equal(am.cpu().numpy(), x.argmax(1).astype(np.int32))
    assert np.array_equal(mx.cpu().numpy(), x.max(1))
    ref_lse = np.log(np.exp(x.astype(np.float64) - x.max(1, keepdims=True)).sum(1))
    assert rel_err(lse.cpu().numpy(), ref_lse) < 1e-6


def test_xent_fwd_and_grad(dev):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(6)
    r, v = 64, 5000
    x = (rng.standard_normal((r, v)) * 2).astype(np.float32)
    tgt = rng.integers(0, v, size=r).astype(np.int32)
    w = (rng.random(r) > 0.3).astype(np.float32)
    ref = O.sequence_xent(x.astype(np.float64)[None], tgt[None].astype(np.int64), w[None].astype(np.float64))[0]
    xd = T(x, dev)
    loss = torch.empty(r, device=dev)
    ops.xent(xd, T(tgt, dev, torch.int32), T(w, dev), loss)
    assert rel_err(loss.cpu().numpy(), ref) < 1e-5
    assert np.array_equal(xd.cpu().numpy(), x)                     # untouched without grad
    scale = np.float32(1.0 / w.sum())
    ops.xent(xd, T(tgt, dev, torch.int32), T(w, dev), loss, T(np.array([scale]), dev), True)
    sm = O.softmax(x.astype(np.float64))
    sm[np.arange(r), tgt] -= 1
    ref_g = sm * w[:, None] * scale
    assert rel_err(xd.cpu().numpy(), ref_g) < 1e-5


@pytest.mark.parametrize("r,v,g,smoothing", [(64, 5000, 7, 0.0), (300, 32000, 256, 0.0), (33, 1024, 33, 0.1),
                                             (5, 36, 1, 0.0)])
def test_xent_with_column_sums(dev, r, v, g, smoothing):
    """nm_xent_colsum: loss and in-place gradient bit-identical to nm_xent, partial column sums add up to the
    column sums of that gradient (float64 yardstick)."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(r + v)
    x = (rng.standard_normal((r, v)) * 2).astype(np.float32)
    tgt = rng.integers(0, v, size=r).astype(np.int32)
    w = (rng.random(r) > 0.3).astype(np.float32)
    scale = T(np.array([1.0 / max(w.sum(), 1.0)], np.float32), dev)
    want, got = T(x, dev), T(x, dev)
    loss_want, loss_got = torch.empty(r, device=dev), torch.empty(r, device=dev)
    ops.xent(want, T(tgt, dev, torch.int32), T(w, dev), loss_want, scale, True, smoothing)
    assert ops.xent_colsum_ok(got)
    partial = torch.full((g, v), float("nan"), device=dev)
    ops.xent_colsum(got, T(tgt, dev, torch.int32), T(w, dev), loss_got, scale, smoothing, partial)
    assert torch.equal(got, want) and torch.equal(loss_got, loss_want)
    ref = want.cpu().numpy().astype(np.float64).sum(0)
    bias = torch.zeros(v, device=dev)
    ops.colsum(partial, bias)
    assert np.abs(bias.cpu().numpy() - ref).max() < 1e-6 * max(np.abs(ref).max(), 1e-3) + 1e-9
    assert np.abs(partial.cpu().numpy().astype(np.float64).sum(0) - ref).max() < 1e-6 * max(np.abs(ref).max(), 1e-3) + 1e-9


# --------------------------------------------------------------------------- #
def _beam_step_ref(logits, k, logprob_sum, lengths, finished, alpha):
    """One beam_search_decoder body top-k (oracle arithmetic, fp32)."""
    r, v = logits.shape
    b = r // k
    lp = O.log_softmax(logits).reshape(b, k, v)
    fin_row = np.full(v, -O.INF, np.float32)
    fin_row[0] = 0
    fm = finished.astype(np.float32)[:, :, None]
    lp = (1 - fm) * lp + fm * fin_row
    hyp = logprob_sum[:, :, None] + lp
    hl = lengths + 1 - finished.astype(np.int32)
    sc = (hyp / O.length_penalty(hl, alpha, np.float32)[:, :, None]).reshape(b, k * v).astype(np.float32)
    ts, ti = O.top_k(sc, k + 1)
    return sc, hyp.reshape(b, k * v), hl, ts, ti


@pytest.mark.parametrize("b,k,v", [(128, 5, 32000), (3, 3, 70), (1, 2, 17), (7, 8, 1000), (16, 12, 32000), (5, 16, 516),
                                   (2, 10, 37)])
def test_beam_topk_step(dev, b, k, v):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(b + k + v)
    logits = (rng.standard_normal((b * k, v)) * 4).astype(np.float32)
    lps = (-rng.random((b, k)) * 20).astype(np.float32)
    lens = rng.integers(0, 30, size=(b, k)).astype(np.int32)
    fin = rng.random((b, k)) < 0.3
    fin[0] = True                                       # a fully finished sentence
    if b > 1:
        lps[1, 1:] = -O.INF                             # first-step state
    sc, hyp, hl, ts, ti = _beam_step_ref(logits, k, lps, lens, fin, 0.6)
    ld = T(logits, dev)
    mx, lse = torch.empty(b * k, device=dev), torch.empty(b * k, device=dev)
    ops.row_stats(ld, mx, lse, None)
    pen = ops.length_penalty_table(64, 0.6, dev)
    i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
    o_sc, o_lps = torch.empty((b, k), device=dev), torch.empty((b, k), device=dev)
    o_w, o_b, o_len, o_fin, o_src = i32(b, k), i32(b, k), i32(b, k), i32(b, k), i32(b, k)
    ops.beam_topk_step(ld, b, k, mx, lse, T(lps, dev), T(lens, dev, torch.int32),
                       T(fin.astype(np.int32), dev, torch.int32), pen, O.END, o_sc, o_w, o_b, o_lps,
                       o_len, o_fin, o_src, ops.beam_workspace(b, k, v, dev))
    got_idx = (o_b.cpu().numpy().astype(np.int64) * v + o_w.cpu().numpy())
    ref_idx = ti[:, :k]
    # exact unless the oracle itself reports a near-tie between neighbours
    gaps = np.abs(np.diff(ts, axis=1)) / np.maximum(np.abs(ts[:, :-1]), 1e-30)
    for bi in range(b):
        if np.array_equal(got_idx[bi], ref_idx[bi]):
            continue
        assert gaps[bi].min() < 1e-5, f"beam index mismatch without a near-tie, sentence {bi}"
    same = np.all(got_idx == ref_idx, axis=1)
    assert same.mean() > 0.98
    bi = np.arange(b)[:, None]
    assert rel_err(o_sc.cpu().numpy()[same], ts[:, :k][same]) < 1e-5
    assert rel_err(o_lps.cpu().numpy()[same], hyp[bi, ref_idx][same]) < 1e-5
    beam_ref = ref_idx // v
    assert np.array_equal(o_len.cpu().numpy()[same], hl[bi, beam_ref][same])
    fin_ref = fin[bi, beam_ref] | ((ref_idx % v) == O.END)
    assert np.array_equal(o_fin.cpu().numpy()[same].astype(bool), fin_ref[same])
    assert np.array_equal(o_src.cpu().numpy()[same], (bi * k + beam_ref)[same])


def test_beam_exact_ties_take_lower_index(dev):
    """Equal scores: tf.nn.top_k returns the lower flat index first."""
    from neuralmonkey_amd import ops
    b, k, v = 2, 3, 50
    logits = np.zeros((b * k, v), np.float32)          # uniform -> all candidates tie per beam
    lps = np.zeros((b, k), np.float32)
    lens = np.zeros((b, k), np.int32)
    fin = np.zeros((b, k), np.int32)
    ld = T(logits, dev)
    mx, lse = torch.empty(b * k, device=dev), torch.empty(b * k, device=dev)
    ops.row_stats(ld, mx, lse, None)
    i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
    o_sc, o_lps = torch.empty((b, k), device=dev), torch.empty((b, k), device=dev)
    o_w, o_b, o_len, o_fin, o_src = i32(b, k), i32(b, k), i32(b, k), i32(b, k), i32(b, k)
    ops.beam_topk_step(ld, b, k, mx, lse, T(lps, dev), T(lens, dev, torch.int32), T(fin, dev, torch.int32),
                       ops.length_penalty_table(8, 1.0, dev), O.END, o_sc, o_w, o_b, o_lps, o_len,
                       o_fin, o_src, ops.beam_workspace(b, k, v, dev))
    assert np.array_equal(o_b.cpu().numpy(), np.zeros((b, k), np.int32))
    assert np.array_equal(o_w.cpu().numpy(), np.tile(np.arange(k, dtype=np.int32), (b, 1)))


def test_gather_and_token_reorder(dev):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(8)
    src = rng.standard_normal((15, 512)).astype(np.float32)
    idx = rng.integers(0, 15, size=15).astype(np.int32)
    dst = torch.empty((15, 512), device=dev)
    ops.gather_rows(T(src, dev), T(idx, dev, torch.int32), dst)
    assert np.array_equal(dst.cpu().numpy(), src[idx])
    steps = 4
    tok = rng.integers(0, 99, size=(6, 15)).astype(np.int32)
    word = rng.integers(0, 99, size=15).astype(np.int32)
    out = torch.zeros((6, 15), dtype=torch.int32, device=dev)
    ops.beam_reorder_tokens(T(tok, dev, torch.int32), T(idx, dev, torch.int32), T(word, dev, torch.int32),
                            out, steps, 15)
    ref = np.concatenate([tok[:steps][:, idx], word[None]], 0)
    assert np.array_equal(out.cpu().numpy()[:steps + 1], ref)


@pytest.mark.parametrize("rows,cols,ld", [
    (6400, 512, 512),        # a Transformer-base bias gradient: the float4 kernel, 48 row slices
    (6400, 2048, 2048),
    (6400, 512, 1536),       # a column block of a wider gradient buffer
    (640, 32000, 32000),     # the output bias
    (64, 8, 8), (65, 4, 12), (200, 516, 516),
    (63, 512, 512),          # fewer than 64 rows: the one-float-per-thread kernel
    (777, 130, 131),         # neither aligned nor a multiple of 4
])
def test_colsum_both_kernels(dev, rows, cols, ld):
    """nm_colsum (bias gradients) against a float64 sum: the float4 kernel (16-byte aligned, ld and cols multiples of
    4, >= 64 rows) and the scalar one; deterministic from launch to launch; ``accumulate`` adds to what is there."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(rows + cols)
    full = rng.standard_normal((rows, ld)).astype(np.float32)
    xd = T(full, dev)[:, :cols]
    want = full[:, :cols].astype(np.float64).sum(0)
    out = torch.full((cols,), float("nan"), device=dev)
    ops.colsum(xd, out)
    tol = 2e-6 * np.sqrt(rows) * max(1.0, np.abs(full).max())
    assert np.abs(out.cpu().numpy() - want).max() <= tol
    again = torch.empty_like(out)
    ops.colsum(xd, again)
    assert torch.equal(out, again)
    base = T(rng.standard_normal(cols).astype(np.float32), dev)
    acc = base.clone()
    ops.colsum(xd, acc, accumulate=True)
    assert torch.equal(acc, base + out)


@pytest.mark.parametrize("rows,count,m,n,ld_extra", [
    (128, 50, 512, 1536, 0),      # a NematusGRU gates kernel of the general-path model: 50 steps of 128 sentences
    (128, 50, 512, 512, 0),
    (16, 3, 8, 12, 4),            # small, padded rows
    (48, 7, 132, 68, 0),          # ragged tiles
    (640, 12, 512, 512, 0),       # beam-sized members
    (32, 1, 64, 64, 0),           # a chain of one
])
def test_chained_weight_and_bias_gradients(dev, rows, count, m, n, ld_extra):
    """nm_gemm_f32_chain / nm_colsum_chain: sum_i a_i^T b_i and the column sums of the b_i over members that live in
    separate buffers, against float64; ``accumulate`` adds to what is there; deterministic from launch to launch."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(rows + count + m)
    a_np = [rng.standard_normal((rows, m + ld_extra)).astype(np.float32) for _ in range(count)]
    b_np = [rng.standard_normal((rows, n + ld_extra)).astype(np.float32) for _ in range(count)]
    a_d = [T(a, dev)[:, :m] for a in a_np]
    b_d = [T(b, dev)[:, :n] for b in b_np]
    want = sum(a[:, :m].astype(np.float64).T @ b[:, :n].astype(np.float64) for a, b in zip(a_np, b_np))
    base = rng.standard_normal((m, n)).astype(np.float32)
    out = T(base, dev)
    ops.gemm_chain(list(zip(a_d, b_d)), out, accumulate=True)
    tol = 3e-6 * np.sqrt(rows * count) * 4.0
    assert np.abs(out.cpu().numpy() - (want + base)).max() <= tol * max(1.0, np.abs(want).max() / np.sqrt(rows * count))
    again = T(base, dev)
    ops.gemm_chain(list(zip(a_d, b_d)), again, accumulate=True)
    assert torch.equal(out, again)
    fresh = torch.full((m, n), float("nan"), device=dev)
    ops.gemm_chain(list(zip(a_d, b_d)), fresh, accumulate=False)
    assert np.abs(fresh.cpu().numpy() - want).max() <= tol * max(1.0, np.abs(want).max() / np.sqrt(rows * count))
    # the bias gradient of the same chain
    want_b = sum(b[:, :n].astype(np.float64).sum(0) for b in b_np)
    bias = T(base[0].copy(), dev)
    ops.colsum_chain(b_d, bias, accumulate=True)
    assert np.abs(bias.cpu().numpy() - (want_b + base[0])).max() <= 2e-6 * np.sqrt(rows * count) * 4.0
    bias2 = torch.full((n,), float("nan"), device=dev)
    ops.colsum_chain(b_d, bias2, accumulate=False)
    assert np.abs(bias2.cpu().numpy() - want_b).max() <= 2e-6 * np.sqrt(rows * count) * 4.0


@pytest.mark.parametrize("rows,d", [(6400, 512), (37, 512), (640, 2048), (5, 8), (300, 132), (1, 1024)])
def test_layer_norm_bwd_with_parameter_gradients(dev, rows, d):
    """nm_layer_norm_bwd_params (dx, dgamma, dbeta in one call; tf_utils.py:189-219 differentiated) against float64
    autograd: written, and added to what dx / the parameter gradients hold (bits 1 / 0 of ``accumulate``)."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(rows + d)
    x = rng.standard_normal((rows, d)).astype(np.float32) * 2.0 + 0.5
    dy = rng.standard_normal((rows, d)).astype(np.float32)
    gamma = (1.0 + 0.3 * rng.standard_normal(d)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(d)).astype(np.float32)
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    g64 = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
    b64 = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    mu = x64.mean(-1, keepdim=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdim=True)
    y = (x64 - mu) * torch.rsqrt(var + 1e-6) * g64 + b64
    (y * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    xd, dyd, gd, bd = T(x, dev), T(dy, dev), T(gamma, dev), T(beta, dev)
    out, mean, rstd = torch.empty_like(xd), torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.layer_norm_fwd(xd, gd, bd, out=out, mean=mean, rstd=rstd, eps=1e-6)
    dx = torch.full_like(xd, float("nan"))
    dgam, dbet = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    ops.layer_norm_bwd_params(dyd, xd, mean, rstd, gd, dx, dgam, dbet, accumulate=False)
    tol = lambda ref: 2e-5 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(dx.cpu().numpy() - x64.grad.numpy()).max() <= tol(x64.grad.numpy())
    assert np.abs(dgam.cpu().numpy() - g64.grad.numpy()).max() <= tol(g64.grad.numpy()) * np.sqrt(rows)
    assert np.abs(dbet.cpu().numpy() - b64.grad.numpy()).max() <= tol(b64.grad.numpy()) * np.sqrt(rows)
    # accumulate: dx += (the gradient a residual connection left there), the parameter gradients likewise
    base = rng.standard_normal((rows, d)).astype(np.float32)
    dx2 = T(base, dev)
    dgam2, dbet2 = dgam.clone(), dbet.clone()
    ops.layer_norm_bwd_params(dyd, xd, mean, rstd, gd, dx2, dgam2, dbet2, accumulate=True, accumulate_dx=True)
    assert torch.allclose(dx2, T(base, dev) + dx, rtol=1e-6, atol=1e-6)      # (the add may be contracted into an fma)
    assert torch.allclose(dgam2, 2 * dgam, rtol=1e-6, atol=1e-6) and torch.allclose(dbet2, 2 * dbet, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("rows,d", [(6400, 512), (37, 512), (640, 2048), (5, 8), (300, 132), (128, 1024), (3, 2052)])
def test_layer_norm_forward_kernels(dev, rows, d):
    """nm_layer_norm_fwd (one wave per row for contiguous rows of D <= 2048, D % 4 == 0; a workgroup per row otherwise),
    nm_add_layer_norm_fwd and nm_add_layer_norm_stats_fwd against float64 (tf_utils.py:189-219: biased variance, eps
    inside the root)."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(rows * 7 + d)
    x = (rng.standard_normal((rows, d)) * 3.0 + 1.0).astype(np.float32)
    a = rng.standard_normal((rows, d)).astype(np.float32)
    gamma = (1.0 + 0.3 * rng.standard_normal(d)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(d)).astype(np.float32)

    def ref(v):
        v = v.astype(np.float64)
        mu = v.mean(-1, keepdims=True)
        var = ((v - mu) ** 2).mean(-1, keepdims=True)
        rs = 1.0 / np.sqrt(var + 1e-6)
        return (v - mu) * rs * gamma + beta, mu[:, 0], rs[:, 0]
    xd, ad, gd, bd = T(x, dev), T(a, dev), T(gamma, dev), T(beta, dev)
    out, mean, rstd = torch.empty_like(xd), torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.layer_norm_fwd(xd, gd, bd, out=out, mean=mean, rstd=rstd, eps=1e-6)
    want, mu, rs = ref(x)
    assert np.abs(out.cpu().numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    assert np.abs(mean.cpu().numpy() - mu).max() <= 1e-5 and rel_err(rstd.cpu().numpy(), rs) <= 1e-5
    want2, mu2, rs2 = ref(a + x)                       # (the fp32 sum, as the kernels form it)
    s1, o1 = torch.empty_like(xd), torch.empty_like(xd)
    ops.add_layer_norm_fwd(ad, xd, gd, bd, s1, o1)
    assert torch.equal(s1, ad + xd)
    assert np.abs(o1.cpu().numpy() - want2).max() <= 2e-5 * max(1.0, np.abs(want2).max())
    if ops.add_layer_norm_stats_ok(ad, xd, gd, bd):
        s2, o2 = torch.empty_like(xd), torch.empty_like(xd)
        ops.add_layer_norm_stats_fwd(ad, xd, gd, bd, s2, o2, mean, rstd)
        # (from 1024 rows on nm_add_layer_norm_fwd runs this very kernel; below, a workgroup per row sums in another order)
        assert torch.equal(s2, s1) and torch.allclose(o2, o1, rtol=2e-5, atol=2e-5)
        if rows >= 1024:
            assert torch.equal(o2, o1)
        assert np.abs(mean.cpu().numpy() - mu2).max() <= 1e-5 and rel_err(rstd.cpu().numpy(), rs2) <= 1e-5
    else:
        assert d % 4 != 0 or d > 2048


@pytest.mark.parametrize("blocks,threads", [(256, 1024), (272, 1024), (800, 256), (2048, 512)])
def test_workgroups_are_dealt_round_robin_to_the_xcds(dev, blocks, threads):
    """What attn_whole_wide (the two halves of a sentence are workgroups i and i + 8: one XCD, one L2 -- DESIGN 4.2) and
    gemm_tiled's XCD-aware tile order rest their SPEED on, checked where the cluster kernels check it at run time: on an
    idle device workgroup i of a launch lands on XCD i mod 8 (HW_REG_XCC_ID)."""
    from neuralmonkey_amd import _lib
    lib = _lib.load()
    out = torch.full((blocks,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.nm_test_xcc_ids(torch.cuda.current_stream().cuda_stream, out.data_ptr(), blocks, threads),
               "nm_test_xcc_ids")
    ids = out.cpu().numpy()
    assert ids.min() >= 0 and ids.max() <= 7 and len(set(ids.tolist())) == 8
    first = ids[:8]
    assert sorted(first.tolist()) == list(range(8)), first            # eight consecutive workgroups: eight XCDs
    assert (ids == np.tile(first, blocks // 8 + 1)[:blocks]).all()    # ... and the deal repeats: i and i + 8 share one


@pytest.mark.parametrize("count,b,s,c,pad", [(50, 128, 50, 1024, 0), (1, 3, 7, 20, 0), (64, 5, 9, 260, 3), (13, 40, 64, 2048, 0)])
def test_outer_products_of_a_loop_summed_in_one_launch(dev, count, b, s, c, pad):
    """nm_outer_chain: out[b, s, c] (+)= sum_t w_t[b, s] * d_t[b, c] (the attended states' gradient through the context
    sums of a taped time loop) against float64; the weights may be the first S columns of wider rows."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(count + b)
    ws = [rng.standard_normal((b, s + pad)).astype(np.float32) for _ in range(count)]
    ds = [rng.standard_normal((b, c)).astype(np.float32) for _ in range(count)]
    want = sum(w[:, :s, None].astype(np.float64) * d[:, None, :].astype(np.float64) for w, d in zip(ws, ds))
    base = rng.standard_normal((b, s, c)).astype(np.float32)
    out = T(base, dev)
    members = [(T(w, dev)[:, :s], T(d, dev)) for w, d in zip(ws, ds)]
    ops.outer_chain(members, out, accumulate=True)
    assert np.abs(out.cpu().numpy() - (want + base)).max() <= 1e-5 * np.sqrt(count) * 4
    fresh = torch.full((b, s, c), float("nan"), device=dev)
    ops.outer_chain(members, fresh, accumulate=False)
    assert np.abs(fresh.cpu().numpy() - want).max() <= 1e-5 * np.sqrt(count) * 4
