"""nm_step_group (csrc/nm_step.hip): the GEMM groups of one inference step of the RNN attention decoder
(Decoder.next_state, decoders/decoder.py:279-358) against the CPU oracle.

  * plain problems: C = act(A . Bt^T + bias + add), several independent problems in one launch;
  * GRU gates / candidate epilogues == O.gru_cell (nn/ortho_gru_cell.py:44-53) at R = 128, H = 512 and at
    ragged sizes;
  * the attention partials merged in the operand loader: ctx . W computed from nm_attn_fwd_partials'
    workspace == O.attention_step's context . W, and the weights written by the same launch == the oracle's
    distribution (feed_forward.py:139-154), one and five queries per key batch;
  * FusedStepper == FastStepper: the same greedy / beam decode through both step drivers.
"""
import os

import numpy as np
import pytest
import torch

from oracle import nm_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def T(a, dev, dt=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)


def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6))


@pytest.fixture(params=["register-staged", "lds-dma"])
def medium_path(request, monkeypatch):
    """Both medium-M kernels: the register-staged tiles (default) and the opt-in 64x64 LDS-DMA tiles (NM_STEP_DMA=1,
    step_group_dma_kernel: kept as a measured negative result, DESIGN 4.5 -- but kept correct)."""
    monkeypatch.setenv("NM_STEP_DMA", "1" if request.param == "lds-dma" else "0")
    return request.param


@pytest.mark.parametrize("m", [128, 37, 640, 300])     # > 256 rows: 32x32 tiles (step_group_medium_kernel)
def test_plain_problems_share_a_launch(dev, m, medium_path):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(m)
    shapes = [(1024, 512, True, 1), (512, 1024, False, 0), (40, 64, True, 0)]          # (N, K, add?, act)
    probs, want, outs = [], [], []
    for n, k, has_add, act in shapes:
        a = rng.standard_normal((m, k + 8)).astype(np.float32)                          # lda > K
        w = (rng.standard_normal((k, n)) * 0.05).astype(np.float32)
        bias = rng.standard_normal(n).astype(np.float32)
        add = rng.standard_normal((m, n)).astype(np.float32) if has_add else None
        ref = a[:, :k].astype(np.float64) @ w.astype(np.float64) + bias + (add if has_add else 0.0)
        want.append(np.tanh(ref) if act else ref)
        ad, wt, bd = T(a, dev), T(w.T, dev), T(bias, dev)
        out = torch.full((m, n + 4), float("nan"), device=dev)
        spec = dict(A=ad, lda=k + 8, Bt=wt, ldb=k, N=n, K=k, epilogue=0, act=act, bias=bd, C=out, ldc=n + 4)
        if has_add:
            spec.update(add=T(add, dev), ldadd=n)
        probs.append(spec)
        outs.append(out)
    ops.StepGroup(m, probs).launch()
    for out, ref, (n, _, _, _) in zip(outs, want, shapes):
        got = out.cpu().numpy()
        assert rel(got[:, :n], ref) < RTOL
        assert np.isnan(got[:, n:]).all()                                                # nothing beyond N is touched


@pytest.mark.parametrize("rows,e,h", [(128, 512, 512), (23, 48, 80), (640, 512, 512)])
def test_gru_groups_match_the_cell(dev, rows, e, h, medium_path):
    """group 1 ([emb | h] . Wg -> r, u, r*h ; emb . Wc_x -> xc) + group 2 (candidate, blend in place)."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(rows + h)
    p = {"gates_kernel": (rng.standard_normal((e + h, 2 * h)) * 0.05).astype(np.float32),
         "gates_bias": np.ones(2 * h, np.float32),
         "cand_kernel": (rng.standard_normal((e + h, h)) * 0.05).astype(np.float32),
         "cand_bias": (rng.standard_normal(h) * 0.1).astype(np.float32)}
    x = rng.standard_normal((rows, e)).astype(np.float32)
    h0 = rng.standard_normal((rows, h)).astype(np.float32)
    ref = O.gru_cell(x.astype(np.float64), h0.astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()})
    cat = T(np.concatenate([x, h0], 1), dev)
    sel = cat[:, e:]
    ru, rh, xc = (torch.empty((rows, w), device=dev) for w in (2 * h, h, h))
    hist = torch.empty((rows, h), device=dev)
    wg_t, wcx_t, wch_t = T(p["gates_kernel"].T, dev), T(p["cand_kernel"][:e].T, dev), T(p["cand_kernel"][e:].T, dev)
    ld = e + h
    ops.StepGroup(rows, [
        dict(A=cat, lda=ld, Bt=wg_t, ldb=ld, N=2 * h, K=ld, epilogue=1, bias=T(p["gates_bias"], dev), h=sel, ldh=ld,
             ru=ru, rh=rh),
        dict(A=cat, lda=ld, Bt=wcx_t, ldb=e, N=h, K=e, epilogue=0, bias=T(p["cand_bias"], dev), C=xc, ldc=h)]).launch()
    ops.StepGroup(rows, [
        dict(A=rh, lda=h, Bt=wch_t, ldb=h, N=h, K=h, epilogue=2, xc=xc, ldxc=h, ru=ru, h=sel, ldh=ld, h_out=sel,
             ldho=ld, h_out2=hist, ldho2=h)]).launch()
    assert rel(hist.cpu().numpy(), ref) < RTOL
    assert torch.equal(cat[:, e:], hist)                              # the new state replaced the old one in place
    assert np.array_equal(cat[:, :e].cpu().numpy(), x)                # the input half is untouched


@pytest.mark.parametrize("bk,qpk,s,a,c,o", [(128, 1, 50, 1024, 1024, 512), (16, 5, 50, 1024, 1024, 512),
                                            (9, 1, 23, 64, 48, 20), (3, 3, 7, 32, 2048, 36)])
def test_partials_merged_in_the_operand_loader(dev, bk, qpk, s, a, c, o):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(bk * 7 + qpk)
    r = bk * qpk
    q = rng.standard_normal((r, 40)).astype(np.float32)
    ap = {"query_w": (rng.standard_normal((40, a)) * 0.2).astype(np.float32),
          "query_b": (rng.standard_normal(a) * 0.1).astype(np.float32),
          "v": (rng.standard_normal(a) * 0.3).astype(np.float32), "bias": np.float32(0.1)}
    hf = rng.standard_normal((bk, s, a)).astype(np.float32)
    states = rng.standard_normal((bk, s, c)).astype(np.float32)
    mask = np.ones((bk, s), np.float32)
    for i in range(bk):
        mask[i, rng.integers(1, s + 1):] = 0
    rep = lambda x: np.repeat(x, qpk, axis=0)
    ctx_ref, w_ref = O.attention_step(q.astype(np.float64), rep(hf).astype(np.float64), rep(states).astype(np.float64),
                                      rep(mask).astype(np.float64), {k: np.asarray(v, np.float64) for k, v in ap.items()})
    wo = (rng.standard_normal((c, o)) * 0.05).astype(np.float32)
    pre = rng.standard_normal((r, o)).astype(np.float32)
    want = np.tanh(ctx_ref @ wo.astype(np.float64) + pre)
    lay = ops.attn_partials_layout(r, s, a, c)
    assert lay is not None
    nchunk, pctx_off, pstat_off = lay
    ws = ops.attn_workspace(r, s, c, dev)
    y = T(q @ ap["query_w"] + ap["query_b"], dev)
    ops.attn_fwd_partials(y, T(hf, dev), T(states, dev), T(mask, dev), T(ap["v"], dev), T([ap["bias"]], dev), qpk, ws)
    out = torch.empty((r, o), device=dev)
    weights = torch.empty((r, s), device=dev)
    pre_d = T(pre, dev)
    ops.StepGroup(r, [dict(a_kind=1, Bt=T(wo.T, dev), ldb=c, N=o, K=c, epilogue=0, act=1, add=pre_d, ldadd=o, C=out,
                           ldc=o, pctx=ws[pctx_off:], pstat=ws[pstat_off:], nchunk=nchunk, energies=ws,
                           mask=T(mask, dev), weights=weights, S=s, mask_div=qpk, mask_mod=bk)]).launch()
    assert rel(out.cpu().numpy(), want) < RTOL
    assert np.abs(weights.cpu().numpy() - w_ref).max() < 2e-6
    # and identical (to rounding) to the two-launch path: attention + combine, then a plain GEMM
    ctx2 = torch.empty((r, c), device=dev)
    w2 = torch.empty((r, s), device=dev)
    ops.attn_fwd(y, T(hf, dev), T(states, dev), T(mask, dev), T(ap["v"], dev), T([ap["bias"]], dev), qpk, ctx2, w2, ws)
    assert rel(ctx2.cpu().numpy(), ctx_ref) < RTOL
    assert torch.allclose(w2, weights, atol=1e-6, rtol=0)     # (whole-sentence kernel: sums in another order)


def _decode_both_ways(dev, beam, batch=24):
    """The same model and batch through FusedStepper and (NM_NO_FUSED_STEP=1) FastStepper."""
    from neuralmonkey_amd import synthetic
    from neuralmonkey_amd.decoders import decoder as decoder_mod
    outs = []
    for no_fused in ("", "1"):
        if no_fused:
            os.environ["NM_NO_FUSED_STEP"] = "1"
        else:
            os.environ.pop("NM_NO_FUSED_STEP", None)
        try:
            model = synthetic.build_translation_model(vocab_src=2000, vocab_tgt=2000, emb=64, rnn=64, max_len=20,
                                                      beam_size=beam, max_steps=12, with_trainer=False,
                                                      device=str(dev))
            store = model.tf_manager.sessions[0].store
            store.load_state_dict(O.init_params(seed=5, vocab_src=2000, vocab_tgt=2000, emb=64, rnn=64, std=0.1))
            ds = synthetic.synthetic_dataset(seed=6, batch=batch, src_len=20, tgt_len=15, vocab=2000, ragged=True)
            sess = model.tf_manager.sessions[0]
            fd = {}
            for f in model.greedy_runner.feedables | model.beam_runner.feedables:
                fd.update(f.feed_dict(ds, train=False))
            got = sess.run({"sym": model.decoder.decoded_symbols, "logits": model.decoder.runtime_logits,
                            "bs": model.beam_decoder.outputs, "w": model.decoder.runtime_loop_result}, fd)
            kinds = set()
            real = decoder_mod.make_stepper

            def spy(*args, **kw):
                st = real(*args, **kw)
                kinds.add(type(st).__name__)
                return st
            decoder_mod.make_stepper = spy
            try:
                sess.run({"sym": model.decoder.decoded_symbols}, fd)
            finally:
                decoder_mod.make_stepper = real
            outs.append((got, kinds))
        finally:
            os.environ.pop("NM_NO_FUSED_STEP", None)
    return outs


def test_fused_stepper_equals_the_six_gemm_stepper(dev):
    (fused, kf), (plain, kp) = _decode_both_ways(dev, beam=4)
    assert kf == {"FusedStepper"} and kp == {"FastStepper"}
    assert np.array_equal(fused["sym"], plain["sym"])
    assert rel(fused["logits"], plain["logits"]) < 2e-5
    wf, wp = np.asarray(fused["w"].attention_weights[0]), np.asarray(plain["w"].attention_weights[0])
    assert wf.shape == wp.shape and np.abs(wf - wp).max() < 1e-6
    tf_, tp = (np.asarray(x["bs"].last_search_step_output.token_ids) for x in (fused, plain))
    assert tf_.shape == tp.shape and (tf_ == tp).mean() > 0.98          # near-ties may flip between roundings
    sf, sp = (np.asarray(x["bs"].last_search_step_output.scores) for x in (fused, plain))
    assert np.abs(sf - sp).max() < 1e-4 * np.abs(sp).max()


def test_fused_stepper_serves_a_beam_of_hundreds_of_rows(dev):
    """72 sentences x beam 4 = 288 hypotheses: the beam body's step runs the fused groups on 32x32 tiles
    (step_group_medium_kernel) and must select what the six-GEMM stepper selects."""
    (fused, _), (plain, _) = _decode_both_ways(dev, beam=4, batch=72)
    tf_, tp = (np.asarray(x["bs"].last_search_step_output.token_ids) for x in (fused, plain))
    assert tf_.shape == tp.shape and (tf_ == tp).mean() > 0.98          # near-ties may flip between roundings
    sf, sp = (np.asarray(x["bs"].last_search_step_output.scores) for x in (fused, plain))
    assert np.abs(sf - sp).max() < 1e-4 * np.abs(sp).max()
    assert np.array_equal(fused["sym"], plain["sym"])


@pytest.mark.parametrize("bk,qpk,s,e,h,a,c,o,v,with_stats", [
    (128, 1, 50, 512, 512, 1024, 1024, 512, 32000, True),       # the headline greedy step
    (128, 5, 50, 512, 512, 1024, 1024, 512, 32000, True),       # the headline beam step: 640 rows, medium-M groups
    (16, 5, 50, 512, 512, 1024, 1024, 512, 32000, True),        # beam rows share the keys of their sentence
    (7, 1, 13, 32, 48, 64, 96, 40, 516, False),                 # ragged, plain logits GEMM
    (3, 2, 9, 16, 16, 32, 32, 16, 260, True)])
def test_decoder_step_fused_is_the_oracle_step(dev, bk, qpk, s, e, h, a, c, o, v, with_stats):
    """nm_decoder_step_fused through the C-ABI == O.decoder_step + O.state_to_logits (decoders/decoder.py:279-358,
    autoregressive.py:450-459) in float64: new state (in place and copy), attention distribution, projected
    output, logits, and -- with statistics -- the arg max of every row."""
    from neuralmonkey_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(bk * 131 + s)
    rows = bk * qpk
    f = lambda *shape, sc=0.05: (rng.standard_normal(shape) * sc).astype(np.float32)
    cellp = {"gates_kernel": f(e + h, 2 * h), "gates_bias": np.ones(2 * h, np.float32),
             "cand_kernel": f(e + h, h), "cand_bias": f(h, sc=0.1)}
    attp = {"query_w": f(h, a), "query_b": f(a, sc=0.1), "v": f(a), "bias": f(1)[0]}
    out_w, out_b = f(h + e + c, o), f(o, sc=0.1)
    logit_w, logit_b = f(o, v), f(v, sc=0.5)
    emb, h0 = f(rows, e, sc=1.0), f(rows, h, sc=1.0)
    keys, vals = f(bk, s, a, sc=1.0), f(bk, s, c, sc=1.0)
    mask = (np.arange(s)[None, :] < rng.integers(1, s + 1, bk)[:, None]).astype(np.float32)
    d64 = lambda x: {k: np.asarray(w, np.float64) for k, w in x.items()}
    dp = {"cell": d64(cellp), "att": d64(attp), "out_w": out_w.astype(np.float64), "out_b": out_b.astype(np.float64),
          "logit_w": logit_w.astype(np.float64), "logit_b": logit_b.astype(np.float64)}
    rep = lambda x: np.repeat(x, qpk, axis=0).astype(np.float64)
    r_out, r_h, _, r_w = O.decoder_step(dp, O.DecoderSpec(), emb.astype(np.float64), h0.astype(np.float64),
                                        rep(keys), rep(vals), rep(mask))
    r_logits = O.state_to_logits(dp, O.DecoderSpec(), r_out)

    cat = T(np.concatenate([emb, h0], 1), dev)
    z = lambda *shape: torch.full(shape, float("nan"), device=dev)
    h_copy, out_state, w_out, logits = z(rows, h), z(rows, o), z(rows, s), z(rows, v)
    stats = torch.empty(lib.nm_logits_stats_bytes(rows, v) // 4, device=dev) if with_stats else None
    ws = ops.attn_workspace(rows, s, c, dev)
    call = ops.DecoderStepCall(dict(
        rows=rows, emb=e, rnn=h, attn_state=a, ctx_width=c, out=o, vocab=v, src_len=s, rows_per_key=qpk, cat=cat,
        ru=z(rows, 2 * h), rh=z(rows, h), xc=z(rows, h), y=z(rows, a), pre_e=z(rows, o), pre=z(rows, o), ctx=z(rows, c),
        attn_workspace=ws, attn_workspace_bytes=ws.numel() * 4,
        wg_t=T(cellp["gates_kernel"].T, dev), bg=T(cellp["gates_bias"], dev), wcx_t=T(cellp["cand_kernel"][:e].T, dev),
        wch_t=T(cellp["cand_kernel"][e:].T, dev), bc=T(cellp["cand_bias"], dev), wq_t=T(attp["query_w"].T, dev),
        bq=T(attp["query_b"], dev), keys=T(keys, dev), values=T(vals, dev), mask=T(mask, dev), v=T(attp["v"], dev),
        attn_bias=T(np.asarray([attp["bias"]]), dev), wo_h_t=T(out_w[:h].T, dev), wo_e_t=T(out_w[h:h + e].T, dev),
        wo_c_t=T(out_w[h + e:].T, dev), bo=T(out_b, dev), out_act=1, w_vocab=T(logit_w, dev), ld_w_vocab=v,
        b_vocab=T(logit_b, dev), vocab_trans_b=0))
    for _ in range(2):                                   # twice: the arrival counters of the workspace are left at zero
        cat[:, e:] = T(h0, dev)
        call.launch(h_copy=h_copy, ld_h_copy=h, out_state=out_state, ld_out_state=o, attn_weights=w_out, logits=logits,
                    ld_logits=v, stats=stats, stats_bytes=stats.numel() * 4 if with_stats else 0)
        torch.cuda.synchronize()
        assert rel(h_copy.cpu().numpy(), r_h) < RTOL
        assert torch.equal(cat[:, e:], h_copy) and np.array_equal(cat[:, :e].cpu().numpy(), emb)
        assert np.abs(w_out.cpu().numpy() - r_w).max() < 1e-5
        assert rel(out_state.cpu().numpy(), r_out) < RTOL
        assert rel(logits.cpu().numpy(), r_logits) < RTOL
    # ---- the same step with INPUT TABLES: the rows' input symbols index a table [V', 2H + H + O] = [E.Wg_x | E.Wc_x
    # + bc | E.Wo_e]; the embedded half of the input row is not read (poisoned here)
    vt = 37
    ids = rng.integers(0, vt, rows).astype(np.int32)
    etab = f(vt, e, sc=1.0)
    etab[ids] = emb                                       # rows with the same id share an embedding: last one wins ...
    emb_t = etab[ids]                                     # ... so re-read what each row really gets
    table = np.concatenate([etab.astype(np.float64) @ cellp["gates_kernel"][:e],
                            etab.astype(np.float64) @ cellp["cand_kernel"][:e] + cellp["cand_bias"],
                            etab.astype(np.float64) @ out_w[h:h + e]], 1).astype(np.float32)
    t_out, t_h, _, t_w = O.decoder_step(dp, O.DecoderSpec(), emb_t.astype(np.float64), h0.astype(np.float64),
                                        rep(keys), rep(vals), rep(mask))
    t_logits = O.state_to_logits(dp, O.DecoderSpec(), t_out)
    cat[:, :e] = float("nan")
    cat[:, e:] = T(h0, dev)
    table_d, ids_d = T(table, dev), T(ids, dev, torch.int32)
    call.launch(h_copy=h_copy, ld_h_copy=h, out_state=out_state, ld_out_state=o, attn_weights=w_out, logits=logits,
                ld_logits=v, stats=stats, stats_bytes=stats.numel() * 4 if with_stats else 0,
                in_table=table_d, ld_table=table.shape[1], in_ids=ids_d)
    torch.cuda.synchronize()
    assert rel(h_copy.cpu().numpy(), t_h) < RTOL and torch.equal(cat[:, e:], h_copy)
    assert np.abs(w_out.cpu().numpy() - t_w).max() < 1e-5
    assert rel(out_state.cpu().numpy(), t_out) < RTOL
    assert rel(logits.cpu().numpy(), t_logits) < RTOL
    call._set(dict(in_table=None, ld_table=0, in_ids=None))          # pylint: disable=protected-access
    if with_stats:
        sym, fin = torch.zeros(rows, dtype=torch.int32, device=dev), torch.zeros(rows, dtype=torch.int32, device=dev)
        arg = torch.zeros(rows, dtype=torch.int32, device=dev)
        ops.greedy_finish(stats, v, fin, sym, None, 2, argmax_out=arg)
        got = logits.cpu().numpy()
        assert np.array_equal(arg.cpu().numpy(), got.argmax(1))          # first maximum of the fp32 logits it wrote
    # a descriptor without any output for the projection is refused, not ignored
    with pytest.raises(_lib.NMHipError):
        call.launch(logits=None, stats=None, stats_bytes=0)
