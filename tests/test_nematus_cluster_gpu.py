"""The one-launch time loops of NematusGRUCell layers (csrc/nm_gru_cluster.hip: nematus_cluster_fwd_kernel /
nematus_cluster_bwd_kernel behind nm_nematus_seq_fwd / nm_nematus_seq_bwd).

Checkers: oracle.nm_oracle (bidirectional_rnn / dynamic_rnn over nematus_gru_cell: nn/ortho_gru_cell.py:73-105,
encoders/recurrent.py:71-110) for the forward loop, float64 autograd of the same recurrence for BPTT, and
oracle.general_ref (the whole model) for a training step whose encoder layer takes the loops.
Tolerances: forward 2e-5 of the largest state; gradients 1e-4 of the largest gradient (ops level), 1e-3 of each
tensor's largest gradient (model level, as tests/test_general_gpu.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(rows, steps, e, h, ndir, seed, state_bias):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, steps, e)) * 0.7).astype(np.float32)
    lens = rng.integers(1, steps + 1, size=rows).astype(np.int32)
    lens[0] = steps
    cells = []
    for _ in range(ndir):
        p = {"gates_input_kernel": (rng.standard_normal((e, 2 * h)) * (1.2 / e ** 0.5)).astype(np.float32),
             "gates_input_bias": (rng.standard_normal(2 * h) * 0.1).astype(np.float32),
             "gates_state_kernel": (rng.standard_normal((h, 2 * h)) * (1.2 / h ** 0.5)).astype(np.float32),
             "cand_input_kernel": (rng.standard_normal((e, h)) * (1.2 / e ** 0.5)).astype(np.float32),
             "cand_input_bias": (rng.standard_normal(h) * 0.1).astype(np.float32),
             "cand_state_kernel": (rng.standard_normal((h, h)) * (1.2 / h ** 0.5)).astype(np.float32)}
        if state_bias:
            p["gates_state_bias"] = (rng.standard_normal(2 * h) * 0.2).astype(np.float32)
            p["cand_state_bias"] = (rng.standard_normal(h) * 0.2).astype(np.float32)
        cells.append(p)
    return x, lens, cells


@pytest.mark.parametrize("rows,steps,h,ndir,state_bias", [
    (37, 7, 256, 2, False),      # rows that do not fill the row tiles
    (100, 5, 256, 2, True),      # 14 clusters: the last XCD hosts none
    (16, 6, 512, 1, True),       # one cluster
    (20, 9, 384, 2, False),
    (128, 12, 512, 2, False),    # the headline encoder's shape, ragged
])
def test_nematus_loops_against_the_oracle(dev, rows, steps, h, ndir, state_bias):
    from neuralmonkey_amd import ops
    from oracle import nm_oracle as O
    e = 48
    assert ops.gru_seq_supported(rows, h, ndir)
    x, lens, cells = _case(rows, steps, e, h, ndir, rows + 3 * steps + h, state_bias)
    if ndir == 2:
        want_out, want_fin = O.bidirectional_rnn(O.nematus_gru_cell, x, lens, cells[0], cells[1])
        want_fin = np.stack([want_fin[:, :h], want_fin[:, h:]])
    else:
        want_out, fin = O.dynamic_rnn(O.nematus_gru_cell, x, lens, cells[0])
        want_fin = fin[None]
    x2 = x.reshape(-1, e)
    xp = np.concatenate([np.concatenate([x2 @ c["gates_input_kernel"] + c["gates_input_bias"],
                                         x2 @ c["cand_input_kernel"] + c["cand_input_bias"]], 1) for c in cells], 1)
    T = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    xpd = T(xp.astype(np.float32))
    ug = T(np.stack([c["gates_state_kernel"] for c in cells]))
    uc = T(np.stack([c["cand_state_kernel"] for c in cells]))
    bgs = T(np.stack([c["gates_state_bias"] for c in cells])) if state_bias else None
    bcs = T(np.stack([c["cand_state_bias"] for c in cells])) if state_bias else None
    lengths = T(lens, torch.int32)
    hcur = torch.zeros(ndir, rows, h, device=dev)
    out = torch.zeros(rows, steps, ndir * h, device=dev)
    ru_all = torch.empty(steps, ndir, rows, 2 * h, device=dev)
    c_all = torch.empty(steps, ndir, rows, h, device=dev)
    sc_all = torch.empty(steps, ndir, rows, h, device=dev)
    ws = torch.empty(ops.nematus_seq_workspace_floats(rows, h, ndir), device=dev)
    xrs, xts = steps * ndir * 3 * h, ndir * 3 * h
    seq = (h, steps * ndir * h, ndir * h)
    ops.nematus_seq_fwd(steps, ndir, rows, h, xpd, (3 * h, xrs, xts), torch.zeros_like(hcur), hcur, 0, ru_all[0],
                        ndir * rows * 2 * h,
                        sc_all[0], ndir * rows * h, c_all[0], ndir * rows * h, ug, uc, ws, bgs=bgs, bcs=bcs,
                        lengths=lengths, out=out, out_strides=seq)
    torch.cuda.synchronize()
    assert not ops.gru_seq_failed(ws)
    scale = max(1.0, float(np.abs(want_out).max()))
    assert np.abs(out.cpu().numpy() - want_out).max() <= 2e-5 * scale
    assert np.abs(hcur.cpu().numpy() - want_fin).max() <= 2e-5 * scale

    # BPTT: float64 autograd of the same recurrence over xp, the state projections' outputs (through a zero offset
    # added to them: its gradient is what the state kernels and biases see) and the zero initial state
    rng = np.random.default_rng(5)
    d_out = rng.standard_normal((rows, steps, ndir * h)) * (np.arange(steps)[None, :] < lens[:, None])[:, :, None]
    d_fin = rng.standard_normal((ndir, rows, h))
    xp64 = torch.tensor(xp, dtype=torch.float64, requires_grad=True)
    h0_64 = torch.zeros(ndir, rows, h, dtype=torch.float64, requires_grad=True)
    off_sc = torch.zeros(rows, steps, ndir, h, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(lens)
    loss = 0.0
    ar = torch.arange(rows)
    for d in range(ndir):
        ug64 = torch.tensor(cells[d]["gates_state_kernel"], dtype=torch.float64)
        uc64 = torch.tensor(cells[d]["cand_state_kernel"], dtype=torch.float64)
        bg64 = torch.tensor(cells[d]["gates_state_bias"], dtype=torch.float64) if state_bias else 0.0
        bc64 = torch.tensor(cells[d]["cand_state_bias"], dtype=torch.float64) if state_bias else 0.0
        xd = xp64.view(rows, steps, ndir, 3 * h)[:, :, d]
        hh = h0_64[d]
        for t in range(steps):
            pos = (lt - 1 - t).clamp(min=0) if d == 1 else torch.full((rows,), t)
            live = (t < lt)[:, None]
            xt = xd[ar, pos]
            g = torch.sigmoid(xt[:, :2 * h] + hh @ ug64 + bg64)
            r, u = g[:, :h], g[:, h:]
            sc = hh @ uc64 + bc64 + off_sc[ar, pos, d]
            c = torch.tanh(xt[:, 2 * h:] + sc * r)
            new = torch.where(live, u * hh + (1 - u) * c, hh)
            contrib = torch.tensor(d_out[ar, pos.numpy(), d * h:(d + 1) * h]) * live
            loss = loss + (torch.where(live, new, torch.zeros_like(new)) * contrib).sum()
            hh = new
        loss = loss + (hh * torch.tensor(d_fin[d])).sum()
    loss.backward()
    dh = T(d_fin)
    dxp = torch.zeros(rows * steps, ndir * 4 * h, device=dev)
    ops.nematus_seq_bwd(steps, ndir, rows, h, dh, T(d_out), seq, ru_all[0], ndir * rows * 2 * h, sc_all[0],
                        ndir * rows * h, c_all[0], ndir * rows * h, None, out, seq, dxp,
                        (4 * h, steps * ndir * 4 * h, ndir * 4 * h), ug, uc, ws, lengths=lengths)
    torch.cuda.synchronize()
    assert not ops.gru_seq_failed(ws)
    got = dxp.cpu().numpy().reshape(rows, steps, ndir, 4 * h)
    want_dxp = xp64.grad.numpy().reshape(rows, steps, ndir, 3 * h)
    live_pos = (np.arange(steps)[None, :] < lens[:, None])
    want_dsc = off_sc.grad.numpy() * live_pos[:, :, None, None]
    gscale = np.abs(want_dxp).max()
    assert np.abs(got[..., :3 * h] - want_dxp).max() <= 1e-4 * gscale
    assert np.abs(got[..., 3 * h:] - want_dsc).max() <= 1e-4 * max(gscale, np.abs(want_dsc).max())
    assert np.abs(dh.cpu().numpy() - h0_64.grad.numpy()).max() <= 1e-4 * max(gscale, np.abs(h0_64.grad.numpy()).max())


@pytest.mark.parametrize("direction,h", [("bidirectional", 256), ("backward", 256), ("forward", 384),
                                         ("bidirectional", 300), ("forward", 260)])      # 300 / 260: padded to 384
def test_encoder_layer_takes_the_loops_and_matches_the_model_oracle(dev, direction, h, monkeypatch):
    """A model whose NematusGRU encoder layer is wide enough for the cluster kernels: the training step (loss and
    every gradient) against oracle.general_ref, with the loops (and with NM_NEMATUS_CLUSTER=0: the step-by-step tape)."""
    from oracle import general_ref as G
    from neuralmonkey_amd import ops
    from tests.test_general_gpu import _build, _data
    cfg = G.Config(rnn_layers=((h, direction, "NematusGRU"),), dec_cell="NematusGRU", conditional_gru=True, rnn_size=8,
                   output_projection=("nematus", "tanh", 1.0))
    calls = {"fwd": 0, "bwd": 0}
    real_f, real_b = ops.nematus_seq_fwd, ops.nematus_seq_bwd

    def spy_f(*a, **k):
        calls["fwd"] += 1
        return real_f(*a, **k)

    def spy_b(*a, **k):
        calls["bwd"] += 1
        return real_b(*a, **k)
    monkeypatch.setattr(ops, "nematus_seq_fwd", spy_f)
    monkeypatch.setattr(ops, "nematus_seq_bwd", spy_b)
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NM_NEMATUS_CLUSTER", mode)
        m = _build(dev, cfg, 12, 8, init_std=0.08)
        ds, src, tgt = _data(5, 7, 6, 8)
        before = dict(calls)
        res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]
        if mode == "1":
            assert calls["fwd"] == before["fwd"] + 1 and calls["bwd"] == before["bwd"] + 1
        else:
            assert calls == before
        ref = G.GeneralModel(m["params"], cfg, requires_grad=True)
        ref_loss, ref_g = ref.train_grads(src, tgt, train=True)
        assert abs(res.losses[cfg.dec_name + " - cost"] - ref_loss) < 1e-4 * abs(ref_loss)
        store = m["store"]
        bad = {}
        for name in store.names():
            got = store.g(name).cpu().numpy().reshape(-1)
            want = ref_g[name]
            want = np.zeros_like(got) if want is None else want.reshape(-1)
            if name.endswith("attn_bias"):
                continue
            err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6))
            if err > 1e-3:
                bad[name] = err
        assert not bad, "mode {}: gradient mismatch: {}".format(mode, bad)
        grads[mode] = {n: store.g(n).cpu().numpy().copy() for n in store.names()}
    # inference through the same layer (a tape that does not record keeps one step's worth of saved gates)
    monkeypatch.setenv("NM_NEMATUS_CLUSTER", "1")
    m = _build(dev, cfg, 12, 8, init_std=0.08)
    ds, src, _ = _data(4, 7, 6, 8, with_target=False)
    ref = G.GeneralModel(m["params"], cfg)
    ref_sym, ref_mask, ref_logits = ref.greedy(src, 8)
    sess = m["tfm"].sessions[0]
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], m["att"], m["dec"]):
        fd.update(part.feed_dict(ds, train=False))
    before = calls["fwd"]
    out = sess.run({"sym": m["dec"].decoded_symbols, "logits": m["dec"].runtime_logits}, fd)
    assert calls["fwd"] == before + 1
    assert np.array_equal(out["sym"], ref_sym)
    assert np.abs(out["logits"] - ref_logits).max() <= 1e-4 * np.abs(ref_logits).max()


@pytest.mark.parametrize("direction,h,batch", [("bidirectional", 256, 5), ("backward", 384, 9), ("forward", 512, 40),
                                               ("bidirectional", 512, 33)])
def test_lstm_layer_takes_the_loops_and_matches_the_model_oracle(dev, direction, h, batch, monkeypatch):
    """LSTMCell encoder layers (tf.nn.rnn_cell.LSTMCell: encoders/recurrent.py:21) as one cluster launch each way
    (nm_lstm_seq_fwd / nm_lstm_seq_bwd) against oracle.general_ref: the training step (loss, every gradient) with the
    loops and with NM_LSTM_CLUSTER=0 (the step-by-step tape), ragged batches, then greedy decoding through the layer."""
    from oracle import general_ref as G
    from neuralmonkey_amd import ops
    from tests.test_general_gpu import _build, _data
    cfg = G.Config(rnn_layers=((h, direction, "LSTM"),), dec_cell="LSTM", rnn_size=8)
    calls = {"fwd": 0, "bwd": 0}
    real_f, real_b = ops.lstm_seq_fwd, ops.lstm_seq_bwd

    def spy_f(*a, **k):
        calls["fwd"] += 1
        return real_f(*a, **k)

    def spy_b(*a, **k):
        calls["bwd"] += 1
        return real_b(*a, **k)
    monkeypatch.setattr(ops, "lstm_seq_fwd", spy_f)
    monkeypatch.setattr(ops, "lstm_seq_bwd", spy_b)
    for mode in ("1", "0"):
        monkeypatch.setenv("NM_LSTM_CLUSTER", mode)
        m = _build(dev, cfg, 12, 8, init_std=0.08)
        ds, src, tgt = _data(batch, 7, 6, 8)
        before = dict(calls)
        res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]
        if mode == "1":
            assert calls["fwd"] == before["fwd"] + 1 and calls["bwd"] == before["bwd"] + 1
        else:
            assert calls == before
        ref = G.GeneralModel(m["params"], cfg, requires_grad=True)
        ref_loss, ref_g = ref.train_grads(src, tgt, train=True)
        assert abs(res.losses[cfg.dec_name + " - cost"] - ref_loss) < 1e-4 * abs(ref_loss), mode
        store = m["store"]
        bad = {}
        for name in store.names():
            got = store.g(name).cpu().numpy().reshape(-1)
            want = ref_g[name]
            want = np.zeros_like(got) if want is None else want.reshape(-1)
            if name.endswith("attn_bias"):
                continue
            err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6))
            if err > 1e-3:
                bad[name] = err
        assert not bad, "mode {}: gradient mismatch: {}".format(mode, bad)
    monkeypatch.setenv("NM_LSTM_CLUSTER", "1")
    m = _build(dev, cfg, 12, 8, init_std=0.08)
    ds, src, _ = _data(4, 7, 6, 8, with_target=False)
    ref = G.GeneralModel(m["params"], cfg)
    ref_sym, _, ref_logits = ref.greedy(src, 8)
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], m["att"], m["dec"]):
        fd.update(part.feed_dict(ds, train=False))
    before = calls["fwd"]
    out = m["tfm"].sessions[0].run({"sym": m["dec"].decoded_symbols, "logits": m["dec"].runtime_logits}, fd)
    assert calls["fwd"] == before + 1
    assert np.array_equal(out["sym"], ref_sym)
    assert np.abs(out["logits"] - ref_logits).max() <= 1e-4 * np.abs(ref_logits).max()


def test_stacked_gru_layers_of_the_general_path_take_the_loops(dev, monkeypatch):
    """TF-GRUCell layers of a stacked, layer-normed encoder with dropout (the general path): every layer's
    time loop is one cluster launch each way (encoders/recurrent.py::_gru_cluster_layer -> nn/gru.py seq_fwd / seq_bwd;
    the 12-unit layer padded to 256) -- against oracle.general_ref, and NM_GRU_LAYER_CLUSTER=0 (the step-by-step tape)."""
    from oracle import general_ref as G
    from neuralmonkey_amd import ops
    from tests.test_general_gpu import _build, _data
    cfg = G.Config(rnn_layers=((256, "bidirectional", "GRU"), (512, "backward", "GRU"), (12, "forward", "GRU")),
                   add_layer_norm=True, enc_dropout=0.8, dec_cell="GRU", rnn_size=8)
    calls = {"fwd": [], "bwd": []}
    real_f, real_b = ops.gru_seq_fwd, ops.gru_seq_bwd

    def spy_f(steps, ndir, rows, hsz, *a, **k):
        calls["fwd"].append((ndir, hsz))
        return real_f(steps, ndir, rows, hsz, *a, **k)

    def spy_b(steps, ndir, rows, hsz, *a, **k):
        calls["bwd"].append((ndir, hsz))
        return real_b(steps, ndir, rows, hsz, *a, **k)
    monkeypatch.setattr(ops, "gru_seq_fwd", spy_f)
    monkeypatch.setattr(ops, "gru_seq_bwd", spy_b)
    for mode in ("1", "0"):
        monkeypatch.setenv("NM_GRU_LAYER_CLUSTER", mode)
        m = _build(dev, cfg, 12, 8, init_std=0.08)
        ds, src, tgt = _data(6, 7, 6, 8)
        before = {k: len(v) for k, v in calls.items()}
        res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]
        if mode == "1":
            # the three encoder layers, then the decoder's own loop (8 units, hand-scheduled path, padded to 256)
            assert calls["fwd"][before["fwd"]:] == [(2, 256), (1, 512), (1, 256), (1, 256)]
            assert sorted(calls["bwd"][before["bwd"]:]) == [(1, 256), (1, 256), (1, 512), (2, 256)]
        else:                                            # only the decoder's
            assert [len(v) - before[k] for k, v in calls.items()] == [1, 1]
        ref = G.GeneralModel(m["params"], cfg, requires_grad=True)
        ref_loss, ref_g = ref.train_grads(src, tgt, train=True)
        assert abs(res.losses[cfg.dec_name + " - cost"] - ref_loss) < 1e-4 * abs(ref_loss), mode
        store = m["store"]
        bad = {}
        for name in store.names():
            got = store.g(name).cpu().numpy().reshape(-1)
            want = ref_g[name]
            want = np.zeros_like(got) if want is None else want.reshape(-1)
            if name.endswith("attn_bias"):
                continue
            err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6))
            if err > 1e-3:
                bad[name] = err
        assert not bad, "mode {}: {}".format(mode, bad)
