"""Oracle parity AT BASELINE.json's configs[3] shape, the one ``bench.py`` measures (``configs.captioning``):
pre-extracted 8x8x2048 convolutional maps (S = 64 positions, C = 2048) -> SpatialFiller
(encoders/numpy_stateful_filler.py:209-245) -> Bahdanau attention with state 512 (attention/feed_forward.py:105-166,
attention/base_attention.py:79-122) -> GRU-512 decoder, B = 128, V = 32000, beam 5, BASELINE.md section 3 weights.

At this shape the attention step streams 128 x 64 x (2048 + 512) floats per decoding step through the kernels the
translation model uses at S = 50 / C = A = 1024 (other tile counts, the spatial mask of ones); the small captioning
cases (V = 8000, 13 maps) never reach these instances.

  * attention states (the flattened maps), keys, initial state: 1e-4 relative;
  * greedy decoding: logits of the first 10 steps 1e-4 relative, symbols exact wherever the oracle's own argmax is
    decided by more than 1e-5;
  * beam-5, all 50 steps: every one of the 50 x 128 x 5 selections accounted for by the oracle following the
    engine's picks (tests/beam_accounting.py)."""
import numpy as np
import pytest
import torch

from oracle import general_ref as G
from tests.beam_accounting import account_for_every_selection

pytestmark = pytest.mark.gpu

B, LEN, VOCAB, SHAPE, ATT = 128, 50, 32000, (8, 8, 2048), 512
DECODE_STEPS = 10
NEAR_TIE = 1e-5


@pytest.fixture(scope="module")
def world(dev):
    from neuralmonkey_amd import synthetic
    m = synthetic.build_captioning_model(vocab=VOCAB, shape=SHAPE, att_size=ATT, max_len=LEN, max_steps=LEN,
                                         with_trainer=False, device=str(dev), seed=1234)
    store = m.tf_manager.sessions[0].store
    synthetic.load_baseline_weights(store, seed=1234, std=0.05)
    ds = synthetic.synthetic_captioning_dataset(seed=6001, batch=B, shape=SHAPE, tgt_len=LEN, vocab=VOCAB,
                                                with_target=False)
    maps = np.stack(list(ds.get_series("images")))
    cfg = G.Config(enc_name="image_encoder", spatial=(None, None), rnn_size=512)
    return dict(m=m, store=store, ds=ds, maps=maps, cfg=cfg, params=store.state_dict())


def _feed(m, ds, extra=()):
    fd = {}
    for part in (m.encoder, m.attention, m.decoder) + tuple(extra):
        fd.update(part.feed_dict(ds, train=False))
    return fd


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-12))


def test_maps_keys_and_greedy_logits_at_the_benched_shape(world):
    m, cfg = world["m"], world["cfg"]
    ref = G.GeneralModel(world["params"], cfg)
    with torch.no_grad():
        states, mask, final = ref.encode(world["maps"], False)
        st, hf = ref.attention_setup(states, False)
        s0 = ref.initial_state(final, False, states, mask)
    sess = m.tf_manager.sessions[0]
    got = sess.run({"hf": m.attention.hidden_features, "final": m.encoder.output, "s0": m.decoder.initial_state,
                    "sym": m.decoder.decoded_symbols, "logits": m.decoder.runtime_logits}, _feed(m, world["ds"]))
    assert rel(np.asarray(got["hf"]).reshape(hf.shape), hf.numpy()) < 1e-4
    assert rel(got["final"], final.numpy()) < 1e-4
    assert rel(got["s0"], s0.numpy()) < 1e-4
    ref_sym, _, ref_logits = ref.greedy(world["maps"], DECODE_STEPS)
    steps = ref_logits.shape[0]
    assert steps == DECODE_STEPS and got["logits"].shape[0] >= steps
    top2 = np.partition(ref_logits, VOCAB - 2, axis=-1)[..., -2:]
    safe = np.minimum.accumulate((top2[..., 1] - top2[..., 0]) > NEAR_TIE * np.abs(top2[..., 1]), axis=0)
    assert safe.mean() >= 0.9, "too many near-ties in the oracle: {}".format(safe.mean())
    assert np.array_equal(np.asarray(got["sym"])[:steps][safe], ref_sym[safe].astype(np.int32)), "greedy symbols differ"
    scale = np.abs(ref_logits).max()
    err = np.abs(np.asarray(got["logits"])[:steps] - ref_logits).max(axis=-1)
    print("greedy logits, {} decided (image, step) pairs: max error {:.3g} of max |logit| {:.3g}".format(
        int(safe.sum()), float(err[safe].max() / scale), scale))
    assert float(err[safe].max()) <= 1e-4 * scale


def test_every_beam_selection_of_all_50_steps_is_accounted_for(world):
    m, cfg = world["m"], world["cfg"]
    sess = m.tf_manager.sessions[0]
    fd = _feed(m, world["ds"], extra=(m.beam_decoder,))
    got = sess.run({"bs": m.beam_decoder.outputs, "sel": m.beam_decoder.selection_history}, fd)
    out = got["bs"]
    sel_beam, sel_word = (np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in got["sel"])
    tok = np.asarray(out.last_search_step_output.token_ids)
    assert tok.shape == (LEN + 1, B, 5) and sel_beam.shape == (LEN, B, 5)
    ref = G.GeneralModel(world["params"], cfg).beam_follow(world["maps"], 5, LEN, 0.6, follow=(sel_beam, sel_word))
    exact, reordered, never = account_for_every_selection(ref, sel_beam, sel_word, tok, out, 5, VOCAB, NEAR_TIE)
    print("captioning beam-5, {} steps: all {} selections accounted for -- {:.2%} of the (step, image) top-5 lists "
          "exactly the oracle's, {} lists a legitimate ordering of a near-tie; {:.0%} of the images never met a "
          "near-tie".format(LEN, LEN * B * 5, exact, reordered, never))
