"""The GRU time loops as ONE launch each (csrc/nm_gru_cluster.hip: weight-stationary workgroup clusters, tagged-granule
hand-offs) against the per-step launches they replace (nm_gru_gemm modes 1-4, themselves checked against the oracle's
GRUCell / dynamic_rnn restatement in test_kernels_gpu.py -- nn/ortho_gru_cell.py:44-53, encoders/recurrent.py:86-102).

Same products, same accumulation order, same epilogues -- but the two paths are compiled apart, hipcc contracts their
multiply-adds differently, and the per-step path takes 32x32 tiles for some shapes: the comparison is to 2e-6 of the
largest value (forward) / 1e-5 (backward, whose sums run over 50 steps of products).  Every case is run
several times, with the hand-off buffers dirtied in between, under a GEMM that keeps the chip busy on a second stream
for some of the runs (uneven load is what exposes a broken hand-off), and the error word must stay clear."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(dev, rows, steps, h, ndir, ragged, reverse_only, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *shape: torch.randn(*shape, device=dev, generator=g)
    xp = rn(rows * steps, ndir * 3 * h) * 0.5
    wgh, wch = rn(ndir, h, 2 * h) * (1.5 / h ** 0.5), rn(ndir, h, h) * (1.5 / h ** 0.5)
    h0 = rn(ndir, rows, h) * 0.3
    lengths = None
    if ragged:
        lens = np.random.default_rng(seed).integers(1, steps + 1, size=rows).astype(np.int32)
        lens[0] = steps
        lengths = torch.tensor(lens, device=dev)
    return xp, wgh, wch, h0, lengths


def _stepwise(dev, rows, steps, h, ndir, xp, wgh, wch, h0, lengths, rev0):
    from neuralmonkey_amd.nn import gru
    hcur = h0.clone()
    out = torch.zeros(rows, steps, ndir * h, device=dev)
    ru_all = torch.empty(steps, ndir, rows, 2 * h, device=dev)
    c_all = torch.empty(steps, ndir, rows, h, device=dev)
    rh_all = torch.empty(steps, ndir, rows, h, device=dev)
    xrs, xts, ors, ots = steps * ndir * 3 * h, ndir * 3 * h, steps * ndir * h, ndir * h
    for t in range(steps):
        gru.step_fwd(xp, (3 * h, xrs, xts), hcur, hcur, wgh, wch, ru_all[t], rh_all[t], c_all[t], out, (h, ors, ots),
                     lengths, t, ndir, rows, h, rev0, None, None)
    return hcur, out, ru_all, c_all, rh_all


@pytest.mark.parametrize("rows,steps,h,ndir,ragged,rev0", [
    (128, 50, 512, 1, False, False),        # the decoder's loop of the headline model
    (128, 50, 512, 2, True, False),         # the encoder's: both directions, ragged
    (128, 9, 512, 1, True, True),           # a single reversed direction
    (37, 7, 256, 2, True, False),           # rows that do not fill the row tiles, four waves
    (16, 6, 512, 1, False, False),          # one cluster (strong scaling's 16 sentences per GPU)
    (40, 5, 384, 2, True, False),           # six waves
    (128, 6, 256, 2, True, False),          # 16 clusters: two per XCD
    (100, 5, 256, 2, True, True),           # 14 clusters: the last XCD hosts none; first direction reversed
])
def test_forward_loop_in_one_launch_equals_the_stepwise_launches(dev, rows, steps, h, ndir, ragged, rev0):
    from neuralmonkey_amd import ops
    assert ops.gru_seq_supported(rows, h, ndir)
    xp, wgh, wch, h0, lengths = _case(dev, rows, steps, h, ndir, ragged, rev0, seed=rows + steps + h)
    want = _stepwise(dev, rows, steps, h, ndir, xp, wgh, wch, h0, lengths, rev0)
    ws = ops.gru_seq_workspace(rows, h, ndir, dev)
    xrs, xts, ors, ots = steps * ndir * 3 * h, ndir * 3 * h, steps * ndir * h, ndir * h
    side = torch.cuda.Stream(device=dev)
    big = torch.randn(4096, 4096, device=dev)
    for attempt in range(4):
        hcur = h0.clone()
        out = torch.zeros(rows, steps, ndir * h, device=dev)
        ru_all = torch.full((steps, ndir, rows, 2 * h), 7.0, device=dev)
        c_all = torch.full((steps, ndir, rows, h), 7.0, device=dev)
        rh_all = torch.full((steps, ndir, rows, h), 7.0, device=dev)
        ws.uniform_(-1e30, 1e30)                       # stale tags / values of any earlier launch must not matter
        if attempt % 2 == 1:                           # a GEMM beside the loop: workgroups start late and unevenly
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    ops.gemm(big, big)
        ops.gru_seq_fwd(steps, ndir, rows, h, xp, (3 * h, xrs, xts), hcur, hcur, 0, ru_all[0], ndir * rows * 2 * h,
                        rh_all[0], ndir * rows * h, c_all[0], ndir * rows * h, wgh, wch, ws, lengths=lengths,
                        reverse_dir0=rev0, out=out, out_strides=(h, ors, ots))
        torch.cuda.synchronize()
        assert not ops.gru_seq_failed(ws)
        got = (hcur, out, ru_all, c_all, rh_all)
        for name, a, b in zip(("final", "states", "gates", "candidates", "r*h"), got, want):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), (attempt, name)


@pytest.mark.parametrize("rows,steps,h,ndir,ragged,rev0,with_dout,with_h0", [
    (128, 50, 512, 1, False, False, True, True),     # the decoder's BPTT of the headline model
    (128, 50, 512, 2, True, False, True, False),     # the encoder's: both directions, ragged
    (128, 9, 512, 1, True, True, False, False),      # one reversed direction, gradient only through the final state
    (37, 7, 256, 2, True, False, True, True),
    (16, 6, 512, 1, False, False, True, False),
    (40, 5, 384, 2, True, False, True, False),
    (128, 6, 256, 2, True, False, True, True),
    (100, 5, 256, 2, True, True, True, False),
])
def test_bptt_loop_in_one_launch_equals_the_stepwise_launches(dev, rows, steps, h, ndir, ragged, rev0, with_dout, with_h0):
    from neuralmonkey_amd import ops
    from neuralmonkey_amd.nn import gru
    xp, wgh, wch, h0, lengths = _case(dev, rows, steps, h, ndir, ragged, rev0, seed=7 * rows + steps + h)
    if not with_h0:
        h0 = torch.zeros_like(h0)
    _, out, ru_all, c_all, _ = _stepwise(dev, rows, steps, h, ndir, xp, wgh, wch, h0, lengths, rev0)
    g = torch.Generator(device=dev).manual_seed(5)
    d_final = torch.randn(ndir, rows, h, device=dev, generator=g)
    d_out = torch.randn(rows, steps, ndir * h, device=dev, generator=g) if with_dout else None
    if d_out is not None and lengths is not None:
        d_out *= (torch.arange(steps, device=dev)[None, :] < lengths[:, None]).float()[:, :, None]
    seq = (h, steps * ndir * h, ndir * h)
    xrs, xts = steps * ndir * 3 * h, ndir * 3 * h
    h0_arg = h0 if with_h0 else None

    dh_want = d_final.clone()
    dxp_want = torch.zeros(rows * steps, ndir * 3 * h, device=dev)
    gru.bptt(steps, dh_want, d_out, seq if d_out is not None else None, ru_all, c_all, h0_arg, out, seq, dxp_want,
             (3 * h, xrs, xts), wgh, wch, lengths, ndir, rows, h, rev0,
             torch.empty(2, ndir, rows, 2 * h, device=dev), torch.empty(ndir, rows, h, device=dev),
             torch.empty(ndir, rows, h, device=dev))
    ws = ops.gru_seq_workspace(rows, h, ndir, dev)
    side = torch.cuda.Stream(device=dev)
    big = torch.randn(4096, 4096, device=dev)
    for attempt in range(4):
        dh = d_final.clone()
        dxp = torch.zeros(rows * steps, ndir * 3 * h, device=dev)
        ws.uniform_(-1e30, 1e30)
        if attempt % 2 == 1:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    ops.gemm(big, big)
        ops.gru_seq_bwd(steps, ndir, rows, h, dh, d_out, seq if d_out is not None else None, ru_all[0],
                        ndir * rows * 2 * h, c_all[0], ndir * rows * h, h0_arg, out, seq, dxp, (3 * h, xrs, xts),
                        wgh, wch, ws, lengths=lengths, reverse_dir0=rev0)
        torch.cuda.synchronize()
        assert not ops.gru_seq_failed(ws)
        for name, a, b in (("dh0", dh, dh_want), ("dxp", dxp, dxp_want)):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), (attempt, name)


def test_the_placement_independent_path_gives_the_same_loops():
    """When an XCD does not get the workgroups its clusters need, roles follow blockIdx and the granules are stored
    write-through (correct under any placement, ~2x slower per hop).  On this hardware the tickets always work out, so
    the path is forced (NM_CLUSTER_PLACEMENT=blockidx, read per launch) and every case above is run through it in a
    fresh process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NM_CLUSTER_PLACEMENT="blockidx")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gru_cluster_gpu.py", "-q", "-x", "-k",
                          "one_launch_equals"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]


def test_unsupported_shapes_are_refused_not_run(dev):
    from neuralmonkey_amd import ops
    assert not ops.gru_seq_supported(128, 128, 1)         # fewer than four waves: no thread per element
    assert not ops.gru_seq_supported(128, 320, 1)         # the halves of the backward operand must fall on whole waves
    assert not ops.gru_seq_supported(128, 1024, 1)        # the kernels' slices would not fit the registers
    assert not ops.gru_seq_supported(640, 512, 1)         # more clusters than XCDs can hold: not all resident at once
    assert ops.gru_seq_supported(256, 512, 1) and not ops.gru_seq_supported(256, 512, 2)
    assert ops.gru_seq_supported(128, 256, 2) and not ops.gru_seq_supported(256, 256, 2)      # 16-row clusters only at H = 256


# ---- straight against the oracle (not through the per-step kernels) ----------------------------------------------------
def _oracle_case(rows, steps, e, h, ndir, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, steps, e)) * 0.7).astype(np.float32)
    lens = rng.integers(1, steps + 1, size=rows).astype(np.int32)
    lens[0] = steps
    cells = []
    for _ in range(ndir):
        cells.append({"gates_kernel": (rng.standard_normal((e + h, 2 * h)) * (1.2 / (e + h) ** 0.5)).astype(np.float32),
                      "gates_bias": np.ones(2 * h, np.float32),
                      "cand_kernel": (rng.standard_normal((e + h, h)) * (1.2 / (e + h) ** 0.5)).astype(np.float32),
                      "cand_bias": (rng.standard_normal(h) * 0.1).astype(np.float32)})
    return x, lens, cells


@pytest.mark.parametrize("rows,steps,h,ndir", [
    (37, 7, 256, 2),         # rows that do not fill the row tiles
    (100, 5, 256, 2),        # 14 clusters: the last XCD hosts none
    (128, 6, 256, 2),        # 16 clusters: two per XCD
    (16, 6, 512, 1),         # one cluster
    (128, 12, 512, 2),       # the headline encoder's shape, ragged
])
def test_cluster_loops_against_the_oracle(dev, rows, steps, h, ndir):
    """The one-launch loops against the oracle's tf.nn.(bidirectional_)dynamic_rnn over TF GRUCells
    (oracle.nm_oracle.bidirectional_rnn / dynamic_rnn, encoders/recurrent.py:86-102, nn/ortho_gru_cell.py:44-53) on
    ragged batches, and their BPTT against float64 autograd of the same recurrence -- no per-step HIP kernel in
    between.  Forward 2e-5 of the largest state, gradients 1e-4 of the largest gradient."""
    from neuralmonkey_amd import ops
    from oracle import nm_oracle as O
    e = 48
    assert ops.gru_seq_supported(rows, h, ndir)
    x, lens, cells = _oracle_case(rows, steps, e, h, ndir, seed=rows + 3 * steps + h)
    if ndir == 2:
        want_out, want_fin = O.bidirectional_rnn(O.gru_cell, x, lens, cells[0], cells[1])
        want_fin = np.stack([want_fin[:, :h], want_fin[:, h:]])
    else:
        want_out, fin = O.dynamic_rnn(O.gru_cell, x, lens, cells[0])
        want_fin = fin[None]
    # input halves of the two kernels as one product per direction (what the encoder hoists out of the loop)
    xp = np.concatenate([np.concatenate([x.reshape(-1, e) @ c["gates_kernel"][:e] + c["gates_bias"],
                                         x.reshape(-1, e) @ c["cand_kernel"][:e] + c["cand_bias"]], 1) for c in cells], 1)
    T = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    xpd = T(xp.astype(np.float32))
    wgh = T(np.stack([c["gates_kernel"][e:] for c in cells]))
    wch = T(np.stack([c["cand_kernel"][e:] for c in cells]))
    lengths = T(lens, torch.int32)
    hcur = torch.zeros(ndir, rows, h, device=dev)
    out = torch.zeros(rows, steps, ndir * h, device=dev)
    ru_all = torch.empty(steps, ndir, rows, 2 * h, device=dev)
    c_all = torch.empty(steps, ndir, rows, h, device=dev)
    ws = ops.gru_seq_workspace(rows, h, ndir, dev)
    xrs, xts, ors, ots = steps * ndir * 3 * h, ndir * 3 * h, steps * ndir * h, ndir * h
    ops.gru_seq_fwd(steps, ndir, rows, h, xpd, (3 * h, xrs, xts), hcur, hcur, 0, ru_all[0], ndir * rows * 2 * h,
                    None, 0, c_all[0], ndir * rows * h, wgh, wch, ws, lengths=lengths, out=out, out_strides=(h, ors, ots))
    torch.cuda.synchronize()
    assert not ops.gru_seq_failed(ws)
    scale = max(1.0, float(np.abs(want_out).max()))
    assert np.abs(out.cpu().numpy() - want_out).max() <= 2e-5 * scale
    assert np.abs(hcur.cpu().numpy() - want_fin).max() <= 2e-5 * scale

    # BPTT: float64 autograd of the same recurrence over the pre-activations xp and the (zero) initial state
    rng = np.random.default_rng(5)
    d_out = rng.standard_normal((rows, steps, ndir * h)) * (np.arange(steps)[None, :] < lens[:, None])[:, :, None]
    d_fin = rng.standard_normal((ndir, rows, h))
    xp64 = torch.tensor(xp, dtype=torch.float64, requires_grad=True)
    h0_64 = torch.zeros(ndir, rows, h, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(lens)
    loss = 0.0
    for d in range(ndir):
        wg64, wc64 = torch.tensor(cells[d]["gates_kernel"][e:], dtype=torch.float64), \
            torch.tensor(cells[d]["cand_kernel"][e:], dtype=torch.float64)
        xd = xp64.view(rows, steps, ndir, 3 * h)[:, :, d]
        hh = h0_64[d]
        outs = [None] * steps
        for t in range(steps):
            pos = (lt - 1 - t).clamp(min=0) if d == 1 else torch.full((rows,), t)
            live = (t < lt)[:, None]
            xt = xd[torch.arange(rows), pos]
            g = torch.sigmoid(xt[:, :2 * h] + hh @ wg64)
            r, u = g[:, :h], g[:, h:]
            c = torch.tanh(xt[:, 2 * h:] + (r * hh) @ wc64)
            new = torch.where(live, u * hh + (1 - u) * c, hh)
            contrib = torch.tensor(d_out[torch.arange(rows), pos.numpy(), d * h:(d + 1) * h]) * live
            loss = loss + (torch.where(live, new, torch.zeros_like(new)) * contrib).sum()
            hh = new
        loss = loss + (hh * torch.tensor(d_fin[d])).sum()
    loss.backward()
    dh = T(d_fin)
    dxp = torch.zeros(rows * steps, ndir * 3 * h, device=dev)
    seq = (h, steps * ndir * h, ndir * h)
    ops.gru_seq_bwd(steps, ndir, rows, h, dh, T(d_out), seq, ru_all[0], ndir * rows * 2 * h, c_all[0], ndir * rows * h,
                    None, out, seq, dxp, (3 * h, xrs, xts), wgh, wch, ws, lengths=lengths)
    torch.cuda.synchronize()
    assert not ops.gru_seq_failed(ws)
    want_dxp = xp64.grad.numpy().reshape(rows * steps, ndir * 3 * h)
    gscale = np.abs(want_dxp).max()
    assert np.abs(dxp.cpu().numpy() - want_dxp).max() <= 1e-4 * gscale
    assert np.abs(dh.cpu().numpy() - h0_64.grad.numpy()).max() <= 1e-4 * max(gscale, np.abs(h0_64.grad.numpy()).max())
