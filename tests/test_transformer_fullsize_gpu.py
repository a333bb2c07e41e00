"""Oracle parity AT BASELINE.json's configs[4] shape: Transformer-base encoder-decoder, 6 + 6 layers, d = 512,
8 heads (head width 64), feed-forward 2048, tied embeddings, V = 32000, B = 128 sentences of up to 50 tokens
(ragged), beam 5 (encoders/transformer.py:290-322, decoders/transformer.py:393-516,
attention/scaled_dot_product.py:98-226).

At this size every attention core runs on the matrix-core kernels of csrc/nm_sdp_mfma.hip (4 key tiles, head
width 64), the 6400-row GEMMs take their split-K instances and the whole taped step is one HIP graph; none of
that is reached by the small cases of test_transformer_gpu.py.

  * one training step: loss 1e-4 relative against the float64 oracle; EVERY gradient against the float64
    oracle, next to the float32 oracle's own distance from it (the noise two fp32 implementations of twelve
    LayerNorm-ed residual blocks show against each other) -- printed per tensor (``-s``), and bounded by
    max(floor, factor x noise) per tensor;
  * greedy decoding through the key/value cache against the oracle's literal prefix recompute: logits of the
    first 10 steps within 1e-4 relative, symbols exact up to a sentence's first near-tie;
  * beam-5, 10 steps: token histories exact and scores 1e-4 for every sentence the oracle decides by more than
    the near-tie margin (1e-5 relative between adjacent candidates of the top k+1) at every step.

The oracle needs a few minutes of host CPU (float64 autograd over 2.4 TFLOP, then 10 + 10 decoding steps that
re-run the stack over the prefix)."""
import numpy as np
import pytest
import torch

from oracle import transformer_ref as TRF
from tests.test_transformer_gpu import _build, _data

pytestmark = pytest.mark.gpu

B, LEN, VOCAB, D, FF, DEPTH = 128, 50, 32000, 512, 2048, 6
DECODE_STEPS = 10
NEAR_TIE = 1e-5


@pytest.fixture(scope="module")
def world(dev):
    cfg = TRF.TConfig(depth=DEPTH, n_heads=8, n_heads_self=8, n_heads_enc=8)
    # init_std as in the base-width test: sharp enough distributions for decided beam steps, fp32 logits still
    # within 2e-5 of float64
    m = _build(dev, cfg, D, FF, max_len=LEN, beam=5, seed=13, init_std=1.2, vocab_size=VOCAB,
               beam_steps=DECODE_STEPS)
    ds, src, tgt = _data(B, LEN, LEN - 1, LEN, seed=17, vocab_size=VOCAB)
    m.update(cfg=cfg, ds=ds, src=src, tgt=tgt)
    return m


def _feed(m):
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], m["dec"]):
        fd.update(part.feed_dict(m["ds"], train=False))
    return fd


def test_greedy_and_beam_through_the_cache_match_the_prefix_recompute(world):
    """Yardstick: the oracle in float64.  Twelve LayerNorm-ed residual blocks amplify fp32 rounding, so next to the
    engine's distance from the float64 logits the test measures the float32 ORACLE's own distance from them (two
    fp32 implementations of the same arithmetic cannot be closer to each other than each is to the exact result) and
    prints both; the engine has to stay within 1e-4 relative or 3x that noise, whichever is larger."""
    m, cfg = world, world["cfg"]
    m["store"].load_state_dict(m["params"])
    sess = m["tfm"].sessions[0]
    plain = TRF.TransformerModel(m["params"], cfg)
    exact = TRF.TransformerModel(m["params"], cfg, dtype=torch.float64)
    enc_states, _, _ = exact.encode(m["src"], False)
    enc32, _, _ = plain.encode(m["src"], False)
    fd = _feed(m)
    out = sess.run({"sym": m["dec"].decoded_symbols, "logits": m["dec"].runtime_logits,
                    "enc": m["enc"].temporal_states}, fd)
    es = np.abs(enc_states.numpy()).max()
    enc_err, enc_noise = np.abs(out["enc"] - enc_states.numpy()).max() / es, np.abs(enc32.numpy() - enc_states.numpy()).max() / es
    print("encoder states vs float64: engine {:.3g}, fp32 oracle {:.3g} (relative to the max)".format(enc_err, enc_noise))
    assert enc_err <= max(1e-4, 3 * enc_noise)
    ref_sym, _, ref_logits = exact.greedy(m["src"], DECODE_STEPS)
    _, _, logits32 = plain.greedy(m["src"], DECODE_STEPS)
    steps = min(len(ref_sym), len(out["sym"]))
    assert steps == DECODE_STEPS
    top2 = np.partition(ref_logits[:steps], VOCAB - 2, axis=-1)[..., -2:]
    safe = np.minimum.accumulate((top2[..., 1] - top2[..., 0]) > NEAR_TIE * np.abs(top2[..., 1]), axis=0)
    assert safe.mean() > 0.9, "too many near-ties in the oracle: {}".format(safe.mean())
    assert np.array_equal(out["sym"][:steps][safe], ref_sym[:steps][safe]), "greedy symbols differ"
    scale = np.abs(ref_logits[:steps]).max()
    err = np.abs(out["logits"][:steps] - ref_logits[:steps]).max(-1)[safe] / scale
    noise = np.abs(logits32[:steps] - ref_logits[:steps]).max(-1)[safe] / scale
    print("greedy logits vs float64 over {} decided (sentence, step) pairs, relative to max |logit| = {:.3g}:\n"
          "  engine      max {:.3g}  median {:.3g}\n  fp32 oracle max {:.3g}  median {:.3g}".format(
              int(safe.sum()), scale, err.max(), np.median(err), noise.max(), np.median(noise)))
    assert err.max() <= max(1e-4, 3 * noise.max())

    tok, scores, _ = exact.beam(m["src"], 5, DECODE_STEPS, 0.6)
    gaps = np.stack(exact.beam_gaps)
    _, scores32, _ = plain.beam(m["src"], 5, DECODE_STEPS, 0.6)
    got = sess.run(m["bdec"].outputs, fd)
    got_tok = np.asarray(got.last_search_step_output.token_ids)
    assert got_tok.shape == tok.shape == (DECODE_STEPS + 1, B, 5)
    clean = (gaps > NEAR_TIE).all(axis=0)
    print("beam-5: {:.0%} of the sentences are decided by more than {} at every step".format(clean.mean(), NEAR_TIE))
    assert clean.mean() >= 0.7, "too many near-ties in the oracle ({} clean)".format(clean.mean())
    assert np.array_equal(got_tok[1:][:, clean], tok[1:][:, clean]), "beam token ids differ"
    got_scores = np.asarray(got.last_search_step_output.scores)
    sscale = np.abs(scores[clean]).max()
    serr, snoise = np.abs(got_scores[clean] - scores[clean]).max() / sscale, np.abs(scores32[clean] - scores[clean]).max() / sscale
    print("beam scores vs float64: engine {:.3g}, fp32 oracle {:.3g}".format(serr, snoise))
    assert serr <= max(1e-4, 3 * snoise)


def test_training_step_loss_and_every_gradient(world):
    m, cfg = world, world["cfg"]
    m["store"].load_state_dict(m["params"])
    ref_loss, ref_g = TRF.TransformerModel(m["params"], cfg, dtype=torch.float64, requires_grad=True).train_grads(
        m["src"], m["tgt"], train=True)
    _, g32 = TRF.TransformerModel(m["params"], cfg, requires_grad=True).train_grads(m["src"], m["tgt"], train=True)
    res = m["tfm"].execute(m["ds"], m["trainer"].feedables, [m["trainer"]], train=True)[0]     # mutates the variables
    loss = res.losses[cfg.dec_name + " - cost"]
    print("loss: engine {:.7f}  float64 oracle {:.7f}  relative {:.2e}".format(loss, ref_loss,
                                                                             abs(loss - ref_loss) / abs(ref_loss)))
    assert abs(loss - ref_loss) < 1e-4 * abs(ref_loss)
    store = m["store"]
    gmax = max(float(np.abs(g).max()) for g in ref_g.values() if g is not None)
    bad, worst = {}, (0.0, 0.0, "")
    print("{:<64} {:>10} {:>10} {:>10} {:>10}".format("gradient (vs the float64 oracle)", "l2 err", "l2 noise", "max err",
                                                       "max noise"))
    for name in store.names():
        got = store.g(name).cpu().numpy().reshape(-1).astype(np.float64)
        want = ref_g[name]
        want = np.zeros_like(got) if want is None else want.reshape(-1)
        w32 = np.zeros_like(got) if g32[name] is None else g32[name].reshape(-1).astype(np.float64)
        scale2, scalem = max(np.linalg.norm(want), 1e-3 * gmax), max(np.abs(want).max(), 1e-3 * gmax)
        err2, noise2 = np.linalg.norm(got - want) / scale2, np.linalg.norm(w32 - want) / scale2
        errm, noisem = np.abs(got - want).max() / scalem, np.abs(w32 - want).max() / scalem
        print("{:<64} {:>10.2e} {:>10.2e} {:>10.2e} {:>10.2e}".format(name[-64:], err2, noise2, errm, noisem))
        if err2 / max(noise2, 1e-12) > worst[0]:
            worst = (err2 / max(noise2, 1e-12), err2, name)
        # the engine may be as far from float64 as a second fp32 implementation is: 3x the fp32 oracle's own
        # distance in the L2 norm, 6x in the max norm (a max over 1e5..1e7 entries of two noise samples), with
        # floors where the fp32 oracle happens to sit closer than fp32 resolution of the tensor allows
        if err2 > max(1e-3, 3 * noise2) or errm > max(1e-3, 6 * noisem):
            bad[name] = (float(err2), float(noise2), float(errm), float(noisem))
    print("largest l2 error in units of the fp32 oracle's own noise: {:.2f}x ({:.2e}) on {}".format(*worst))
    assert not bad, "gradient mismatch (l2 err, l2 fp32-oracle noise, max err, max noise): {}".format(bad)


def test_logits_on_the_benchmarked_weights_meet_1e_4_outright(dev):
    """The configuration ``bench.py`` measures (``configs.transformer``): ``synthetic.build_transformer_model`` with
    BASELINE.md section 3 weights (N(0, 0.05), ``load_baseline_weights``), source embeddings scaled by sqrt(d).  On
    THESE weights the north-star tolerance holds without any allowance for fp32 noise: encoder states and the greedy
    logits of the first steps within 1e-4 of the float64 oracle, relative to the tensor's largest magnitude (the
    test above uses 24x larger weights to get decided beam steps, where no fp32 implementation stays within 1e-4)."""
    from neuralmonkey_amd import synthetic
    steps = 3
    m = synthetic.build_transformer_model(vocab=VOCAB, max_len=LEN, max_steps=LEN, with_trainer=False,
                                          device=str(dev), seed=1234)
    store = m.tf_manager.sessions[0].store
    synthetic.load_baseline_weights(store, seed=1234, std=0.05)
    params = store.state_dict()
    ds = synthetic.synthetic_dataset(seed=4000, batch=B, src_len=LEN, tgt_len=LEN, vocab=VOCAB, ragged=True,
                                     with_target=False)
    from oracle import nm_oracle as O
    src = O.pad_ids([list(s) for s in ds.get_series("source")], LEN)
    cfg = TRF.TConfig(depth=DEPTH, n_heads=8, n_heads_self=8, n_heads_enc=8, scale_embeddings=True)
    exact = TRF.TransformerModel(params, cfg, dtype=torch.float64)
    plain = TRF.TransformerModel(params, cfg)
    enc64 = exact.encode(src, False)[0].numpy()
    enc32 = plain.encode(src, False)[0].numpy()
    _, _, lg64 = exact.greedy(src, steps)
    _, _, lg32 = plain.greedy(src, steps)
    fd = {}
    for part in (m.encoder.input_sequence, m.encoder, m.decoder):
        fd.update(part.feed_dict(ds, train=False))
    out = m.tf_manager.sessions[0].run({"logits": m.decoder.runtime_logits, "enc": m.encoder.temporal_states}, fd)
    es, ls = np.abs(enc64).max(), np.abs(lg64).max()
    enc_err, enc_noise = np.abs(np.asarray(out["enc"]) - enc64).max() / es, np.abs(enc32 - enc64).max() / es
    lg_err = np.abs(np.asarray(out["logits"])[:steps] - lg64).max() / ls
    lg_noise = np.abs(lg32 - lg64).max() / ls
    print("benchmarked weights, vs float64: encoder states engine {:.3g} (fp32 oracle {:.3g}); logits of {} steps "
          "engine {:.3g} (fp32 oracle {:.3g})".format(enc_err, enc_noise, steps, lg_err, lg_noise))
    assert enc_err <= 1e-4
    assert lg_err <= 1e-4
