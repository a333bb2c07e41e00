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
