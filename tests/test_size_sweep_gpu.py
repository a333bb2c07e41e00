"""Parity of the headline model family over sizes nobody tuned for.  Every size-specialised kernel on the path (skinny
/ tiled / medium products, the attention variants by S / A / C, cluster loops direct and padded, projection tiles,
beam scans) picks itself from the shape: a seeded sweep over embedding / hidden / attention / vocabulary sizes, batch
sizes and lengths runs one training step, greedy decoding and beam search against the oracle (oracle.torch_ref
float32 autograd, oracle.nm_oracle decoding loop / beam search).  Round 6 found `gemm_skinny16` dropping the K tail
at hidden sizes 260 / 264 this way (tests/test_cluster_pad_gpu.py).
Tolerances: loss 1e-4 relative; gradients 1e-3 of each tensor's largest entry; greedy tokens exact; beam tokens exact
unless the oracle reports a near-tie."""
import numpy as np
import pytest

from oracle import nm_oracle as O
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(2026)
    out = []
    hidden = [4, 12, 36, 64, 100, 132, 200, 248, 256, 260, 264, 268, 280, 300, 328, 384, 392]
    for i in range(36):
        if i == 26:                                      # a second stream: larger batches (the beam steps' medium tiles),
            hidden = [8, 128, 256, 264, 300, 512, 516, 520]         # longer sentences, the widest hidden sizes
        h = int(rng.choice(hidden))
        e = int(rng.choice([8, 20, 64, 100, 260, 300])) if rng.random() < 0.7 else h
        a = int(rng.choice([0, 12, 64, 132, 264, 520]))
        v = int(rng.choice([17, 64, 130, 257, 1000, 1031]))
        b = int(rng.choice([1, 2, 3, 7, 16, 33, 100, 129] if i < 26 else [40, 96, 128, 130, 200]))
        s = int(rng.choice([1, 2, 5, 13, 33, 41] if i < 26 else [9, 40, 50, 52, 64]))
        t = int(rng.choice([1, 2, 6, 11, 25] if i < 26 else [7, 12, 30]))
        if i >= 26:
            v = int(rng.choice([130, 1000, 4100]))
        cap = 600000 if i < 26 else 2400000
        if b * max(s, t) * max(h, e) > cap:             # keep the CPU oracle in seconds
            b = max(1, cap // (max(s, t) * max(h, e)))
        out.append((h, e, a, v, b, s, t, 100 + i))
    # long sentences: more positions than the whole-sentence attention kernels take, more steps than one chunk of
    # anything (decode graphs, the stacks of the taped attention)
    out += [(64, 20, 0, 130, 3, 150, 80, 190), (256, 36, 132, 257, 17, 90, 70, 191), (12, 8, 12, 64, 2, 301, 5, 192),
            (300, 64, 64, 130, 9, 70, 66, 193)]
    return out


@pytest.mark.parametrize("h,e,a,v,b,s,t,seed", _cases())
def test_training_step_and_decoding_at_untuned_sizes(dev, h, e, a, v, b, s, t, seed):
    from neuralmonkey_amd import synthetic
    max_len = max(s, t + 1)
    # (the decoder's embedding size is its output projection's -- decoders/decoder.py checks it; the encoder's is free)
    params = O.init_params(seed=seed, vocab_src=v, vocab_tgt=v, emb=e, rnn=h, dec_emb=h, att_size=a or None, std=0.08)
    ds = synthetic.synthetic_dataset(seed=seed + 1, batch=b, src_len=s, tgt_len=t, vocab=v, ragged=True)
    src = O.pad_ids([list(x) for x in ds.get_series("source")], max_len)
    tgt = np.ascontiguousarray(O.pad_ids([list(x) for x in ds.get_series("target")], max_len, add_end_symbol=True).T)
    model = synthetic.build_translation_model(vocab_src=v, vocab_tgt=v, emb=e, rnn=h, dec_emb=h, att_size=a or None, max_len=max_len,
                                              beam_size=3, max_steps=max_len, device=str(dev))
    store = model.tf_manager.sessions[0].store
    store.load_state_dict(params)
    spec = O.DecoderSpec(max_output_len=max_len)
    enc = O.sentence_encoder(params, src)
    want = O.greedy_tokens(O.decoding_loop(params, spec, enc, None, False))
    res = model.tf_manager.execute(ds, model.greedy_runner.feedables | model.beam_runner.feedables,
                                   [model.greedy_runner, model.beam_runner], compute_losses=False)
    w2i = model.tgt_vocab._word_to_index             # pylint: disable=protected-access
    assert [[w2i[w] for w in sent] for sent in res[0].outputs["target"]] == want
    bres = O.beam_search(params, spec, enc, 3, max_len, 0.6)
    if bres.min_gap > 1e-5:
        assert [[w2i[w] for w in sent] for sent in res[1].outputs["target_beam"]] == O.beam_tokens(bres, 1)[0]
    ref_loss, _, _, ref_g = TR.train_step_grads(TR.to_torch(params), src, tgt, l1_weight=0.0, l2_weight=1e-8)
    out = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
    assert abs(out.losses["decoder - cost"] - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    bad = {}
    for name in store.names():
        got, ref = store.g(name).cpu().numpy(), ref_g[name].numpy()
        if name.endswith("attn_bias"):
            continue
        err = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-8))
        if err > 1e-3:
            bad[name] = err
    assert not bad, bad


def _transformer_cases():
    rng = np.random.default_rng(77)
    out = []
    for i in range(14):
        heads = int(rng.choice([1, 2, 4, 8]))
        dh = int(rng.choice([2, 4, 8, 16, 33, 64]))
        d = heads * dh
        ff = int(rng.choice([8, 20, 64, 132, 512]))
        depth = int(rng.choice([1, 2, 3]))
        if i < 8:
            b, s, t = int(rng.choice([1, 3, 9, 20])), int(rng.choice([1, 4, 11, 23])), int(rng.choice([1, 5, 12]))
        else:                # >= 1024 rows of B x T: the wave-per-row layer norm, the residual sum folded into it, its
            b, s, t = int(rng.choice([48, 64, 90])), int(rng.choice([24, 30])), int(rng.choice([17, 25]))    # backward
        bias = bool(rng.integers(0, 2))
        out.append((heads, dh, ff, depth, b, s, t, bias, 300 + i))
    return out


@pytest.mark.parametrize("heads,dh,ff,depth,b,s,t,bias,seed", _transformer_cases())
def test_transformer_training_step_and_greedy_at_untuned_sizes(dev, heads, dh, ff, depth, b, s, t, bias, seed):
    """The same for the Transformer (oracle.transformer_ref): model width = heads x head width, including widths that
    are no multiple of 4 (every vector path falls back) and batches of >= 1024 rows (the round-6 training kernels)."""
    from oracle import transformer_ref as TRF
    from tests.test_transformer_gpu import _build, _data
    d = heads * dh
    max_len = max(s, t + 1)
    cfg = TRF.TConfig(depth=depth, n_heads=heads, n_heads_self=heads, n_heads_enc=heads, use_att_transform_bias=bias)
    m = _build(dev, cfg, d, ff, max_len=max_len, seed=seed, init_std=0.3)
    ds, src, tgt = _data(b, s, t, max_len, seed=seed + 1)
    ref = TRF.TransformerModel(m["params"], cfg, requires_grad=True)
    ref_loss, ref_g = ref.train_grads(src, tgt, train=True)
    res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]
    assert abs(res.losses[cfg.dec_name + " - cost"] - ref_loss) < 1e-4 * abs(ref_loss)
    store = m["store"]
    gmax = max(float(np.abs(g).max()) for g in ref_g.values() if g is not None)
    bad = {}
    for name in store.names():
        got = store.g(name).cpu().numpy().reshape(-1)
        want = ref_g[name]
        want = np.zeros_like(got) if want is None else want.reshape(-1)
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-3 * gmax))
        if err > 1e-3:
            bad[name] = err
    assert not bad, bad
    if b <= 20:
        store.load_state_dict(m["params"])
        dsd, srcd, _ = _data(b, s, t, max_len, seed=seed + 2, with_target=False)
        ref2 = TRF.TransformerModel(m["params"], cfg)
        sym, mask, _ = ref2.greedy(srcd, max_len)
        dec = m["dec"]
        fd = {}
        for part in (m["enc"].input_sequence, m["enc"], dec):
            fd.update(part.feed_dict(dsd, train=False))
        out = m["tfm"].sessions[0].run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask}, fd)
        assert np.array_equal(out["sym"], sym) and np.array_equal(out["mask"].astype(bool), mask)


def _general_cases():
    rng = np.random.default_rng(991)
    out = []
    for i in range(16):
        h = int(rng.choice([4, 6, 8, 12, 20, 36, 64] if i < 13 else [256]))
        direction = str(rng.choice(["bidirectional", "forward", "backward"]))
        cell = str(rng.choice(["NematusGRU", "GRU", "LSTM"])) if i < 13 else "NematusGRU"
        dec_cell = str(rng.choice(["NematusGRU", "GRU", "LSTM"]))
        cond = bool(rng.integers(0, 2)) and dec_cell != "LSTM"
        r = int(rng.choice([4, 8, 12, 16, 36]))
        es, et = int(rng.choice([4, 8, 10, 12, 20])), int(rng.choice([4, 8, 12, 20]))
        # batches of 16 / 32 / 48 rows: the steps' weight and bias gradients go through the chained products
        # (Tape.defer_wgrad: rows % 16 == 0); the others launch one product per step
        b = int(rng.choice([16, 32, 48] if i % 2 == 0 else [1, 5, 9, 21]))
        s, t = int(rng.choice([2, 5, 9])), int(rng.choice([1, 4, 7]))
        out.append((h, direction, cell, dec_cell, cond, r, es, et, b, s, t, 500 + i))
    # more steps than the taped attention stacks (64): the steps beyond take the per-step backward
    out += [(8, "bidirectional", "NematusGRU", "NematusGRU", True, 8, 8, 8, 16, 12, 70, 590),
            (12, "forward", "LSTM", "LSTM", False, 12, 8, 12, 3, 80, 66, 591)]
    return out


@pytest.mark.parametrize("h,direction,cell,dec_cell,cond,r,es,et,b,s,t,seed", _general_cases())
def test_general_path_training_step_at_untuned_sizes(dev, h, direction, cell, dec_cell, cond, r, es, et, b, s, t, seed):
    """The taped general path (oracle.general_ref): cell types, directions, conditional GRU and sizes drawn at random;
    half of the batches are multiples of 16 rows, where the per-step weight / bias gradients are chained products
    (nm_gemm_f32_chain / nm_colsum_chain), the fused and merged NematusGRU cell steps wherever the sizes allow."""
    from oracle import general_ref as G
    from tests.test_general_gpu import _build, _data
    cfg = G.Config(rnn_layers=((h, direction, cell),), dec_cell=dec_cell, conditional_gru=cond, rnn_size=r)
    max_len = max(s, t + 1)
    m = _build(dev, cfg, es, et, max_len=max_len, seed=seed, init_std=0.3 if h < 100 else 0.08)
    ds, src, tgt = _data(b, s, t, max_len, seed=seed + 1)
    ref = G.GeneralModel(m["params"], cfg, requires_grad=True)
    ref_loss, ref_g = ref.train_grads(src, tgt, train=True)
    res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]
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
    assert not bad, bad
