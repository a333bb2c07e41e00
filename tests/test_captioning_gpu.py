"""Captioning shape (BASELINE configs[3], tests/captioning.ini / flat-multiattention.ini):
pre-extracted 8x8x2048 convolutional maps -> SpatialFiller -> the same Bahdanau attention kernels
-> attention GRU decoder.  Checker: oracle/general_ref.py (encode_spatial + the decoder
restatement, autograd for the gradients).  Tolerances as in test_general_gpu.py."""
import numpy as np
import pytest

from oracle import general_ref as G
from oracle import nm_oracle as O

pytestmark = pytest.mark.gpu

VOCAB = 40


def _build(dev, cfg: G.Config, shape, state_size, emb, seed=9, vocab_size=VOCAB, max_len=8, maps=None, beam=3,
           weight_std=None, logit_std=None):
    from neuralmonkey_amd.attention import Attention
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    from neuralmonkey_amd.decoders import BeamSearchDecoder, Decoder
    from neuralmonkey_amd.encoders import SpatialFiller
    from neuralmonkey_amd.runtime import reset_registry
    from neuralmonkey_amd.synthetic import synthetic_vocabulary
    from neuralmonkey_amd.tf_manager import TensorFlowManager
    from neuralmonkey_amd.trainers import CrossEntropyTrainer
    reset_registry()
    vocab = synthetic_vocabulary(vocab_size)
    ff, proj = cfg.spatial
    enc = SpatialFiller(name=cfg.enc_name, input_shape=list(shape), data_id="images", projection_dim=proj,
                        ff_hidden_dim=ff)
    att = Attention(name=cfg.att_name, encoder=enc, state_size=state_size, dropout_keep_prob=cfg.att_dropout)
    dec = Decoder(encoders=[enc], vocabulary=vocab, data_id="target", name=cfg.dec_name, max_output_len=max_len,
                  dropout_keep_prob=cfg.dec_dropout, embedding_size=emb, rnn_size=cfg.rnn_size, attentions=[att],
                  rnn_cell=cfg.dec_cell, conditional_gru=cfg.conditional_gru)
    bdec = BeamSearchDecoder(name="beam_decoder", parent_decoder=dec, beam_size=beam, max_steps=max_len,
                             length_normalization=0.6)
    trainer = CrossEntropyTrainer(decoders=[dec], l2_weight=0.0, clip_norm=None)
    tfm = TensorFlowManager(num_sessions=1, num_threads=1, device=str(dev), seed=seed)
    tfm.initialize_sessions()
    store = tfm.sessions[0].store
    rng = np.random.default_rng(seed)
    vals = store.state_dict()
    for name, v in vals.items():
        if v.ndim >= 2 or name.endswith("attn_similarity_v"):
            std = 0.35 if v.shape[0] < 256 else 0.35 * (64.0 / v.shape[0]) ** 0.5      # keep pre-activations O(1)
            if weight_std is not None and "OrthoGRUCell" not in name:
                std = weight_std
            if logit_std is not None and name.endswith("state_to_word_W"):
                std = logit_std
            vals[name] = (rng.standard_normal(v.shape) * std).astype(np.float32)
        elif "bias" in name or name.endswith("_b"):
            vals[name] = (v + rng.standard_normal(v.shape) * 0.1).astype(np.float32)
    store.load_state_dict(vals)
    if maps is None:
        bsz = 5
        maps = np.maximum(rng.standard_normal((bsz,) + tuple(shape)), 0).astype(np.float32)
    bsz = len(maps)
    tgt_sents = [["w{}".format(int(i)) for i in rng.integers(0, vocab_size - 4, size=int(n))]
                 for n in rng.integers(2, max_len - 1, size=bsz)]
    ds = Dataset("captions", {"images": list(maps), "target": tgt_sents}, BatchingScheme(batch_size=bsz))
    ids = [[dec.vocabulary._word_to_index[w] for w in s] for s in tgt_sents]
    tgt = np.ascontiguousarray(O.pad_ids(ids, max_len, add_end_symbol=True).T)
    return dict(enc=enc, att=att, dec=dec, bdec=bdec, trainer=trainer, tfm=tfm, store=store,
                params=store.state_dict(), ds=ds, maps=maps, tgt=tgt)


CASES = {
    # raw ResNet-shaped maps, vectorised attention kernels (C = 2048 -> two context groups), fused GRU path
    "resnet_maps": (G.Config(spatial=(None, None), rnn_size=8), (8, 8, 2048), 128, 8),
    # projected maps, tests/captioning.ini attention state_size=10 (any-shape kernel), dropout 0.5
    "projected_dropout": (G.Config(spatial=(32, 16), rnn_size=8, dec_dropout=0.5, att_dropout=0.8), (8, 8, 64), 10, 8),
    "projection_only_condgru": (G.Config(spatial=(None, 24), rnn_size=8, dec_cell="NematusGRU", conditional_gru=True),
                                (4, 6, 40), 12, 8),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_captioning_train_step(dev, case):
    cfg, shape, asz, emb = CASES[case]
    m = _build(dev, cfg, shape, asz, emb)
    ref = G.GeneralModel(m["params"], cfg, requires_grad=True)
    ref_loss, ref_g = ref.train_grads(m["maps"], m["tgt"], train=True)
    res = m["tfm"].execute(m["ds"], m["trainer"].feedables, [m["trainer"]], train=True)[0]
    assert abs(res.losses["decoder - cost"] - ref_loss) < 1e-4 * abs(ref_loss)
    store = m["store"]
    bad = {}
    for name in store.names():
        got = store.g(name).cpu().numpy().reshape(-1)
        want = ref_g[name]
        want = np.zeros_like(got) if want is None else want.reshape(-1)
        if name.endswith("attn_bias"):
            assert abs(got[0]) < 1e-5 and abs(want[0]) < 1e-5
            continue
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6))
        if err > 1e-3:
            bad[name] = err
    assert not bad, "gradient mismatch: {}".format(bad)


@pytest.mark.parametrize("case", sorted(CASES))
def test_captioning_decoding(dev, case):
    cfg, shape, asz, emb = CASES[case]
    m = _build(dev, cfg, shape, asz, emb)
    ref = G.GeneralModel(m["params"], cfg)
    ref_sym, ref_mask, ref_logits = ref.greedy(m["maps"], 8)
    dec, sess = m["dec"], m["tfm"].sessions[0]
    fd = {}
    for part in (m["enc"], m["att"], dec):
        fd.update(part.feed_dict(m["ds"], train=False))
    out = sess.run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask, "logits": dec.runtime_logits,
                    "enc_out": m["enc"].output, "smask": m["enc"].spatial_mask}, fd)
    states, _, final = ref.encode_spatial(m["maps"])
    assert np.abs(out["enc_out"] - final.numpy()).max() <= 1e-4 * np.abs(final.numpy()).max()
    assert out["smask"].shape == (5,) + tuple(shape[:2]) and np.all(out["smask"] == 1.0)
    assert np.array_equal(out["sym"], ref_sym)
    assert np.array_equal(out["mask"].astype(bool), ref_mask)
    assert np.abs(out["logits"] - ref_logits).max() <= 1e-4 * np.abs(ref_logits).max()
    tok, scores, gap = ref.beam(m["maps"], 3, 8, 0.6)
    got = sess.run(m["bdec"].outputs, fd)
    if gap > 1e-5:          # exact indices unless the oracle itself reports a near-tie
        assert np.array_equal(np.asarray(got.last_search_step_output.token_ids)[1:], tok[1:])
    assert np.abs(np.asarray(got.last_search_step_output.scores) - scores).max() <= 1e-4 * np.abs(scores).max()


# ------------------------------------------------------------------------------------------------
# BASELINE configs[3] at model size, on the reference's own pre-extracted maps
# ------------------------------------------------------------------------------------------------
def _reference_maps():
    """The 13 8x8x2048 ResNet maps of the reference's tests/data/flickr30k (the inputs of
    tests/flat-multiattention.ini through readers.numpy_reader), from the committed bundle."""
    import io
    import os
    import tarfile
    bundle = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_tests.tar.gz")
    maps = []
    with tarfile.open(bundle) as tar:
        for m in sorted(tar.getmembers(), key=lambda m: m.name):
            if m.name.endswith(".jpg.npz"):
                maps.append(np.load(io.BytesIO(tar.extractfile(m).read()))["arr_0"].astype(np.float32))
    assert len(maps) == 13 and maps[0].shape == (8, 8, 2048)
    return np.stack(maps)


def test_captioning_model_size_on_reference_maps(dev):
    """ResNet maps of the reference's fixtures -> SpatialFiller -> Bahdanau attention (state 512 over
    C=2048 keys) -> GRU-512 decoder, V=8000, N(0, 0.05) weights: loss, every
    gradient, 12 greedy steps (logits 1e-4, symbols exact) and beam-5 against oracle/general_ref.py."""
    maps = _reference_maps()
    cfg = G.Config(spatial=(None, None), rnn_size=512)
    max_len = 12
    # vocabulary projection N(0, 0.3): with 0.05 the softmax over 8000 words is nearly flat and the beam's
    # candidate scores are packed within the 1e-5 near-tie margin for most sentences
    m = _build(dev, cfg, (8, 8, 2048), 512, 512, seed=11, vocab_size=8000, max_len=max_len, maps=maps, beam=5,
               weight_std=0.05, logit_std=0.3)
    ref = G.GeneralModel(m["params"], cfg, requires_grad=True)
    ref_loss, ref_g = ref.train_grads(m["maps"], m["tgt"], train=True)
    dec, sess = m["dec"], m["tfm"].sessions[0]
    fd = {}
    for part in (m["enc"], m["att"], dec):
        fd.update(part.feed_dict(m["ds"], train=False))
    # decode BEFORE the optimizer step changes the variables
    out = sess.run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask, "logits": dec.runtime_logits}, fd)
    got_beam = sess.run(m["bdec"].outputs, fd)
    res = m["tfm"].execute(m["ds"], m["trainer"].feedables, [m["trainer"]], train=True)[0]
    assert abs(res.losses["decoder - cost"] - ref_loss) < 1e-4 * abs(ref_loss)
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
    assert not bad, "gradient mismatch: {}".format(bad)

    plain = G.GeneralModel(m["params"], cfg)
    ref_sym, ref_mask, ref_logits = plain.greedy(m["maps"], max_len)
    steps = min(len(ref_sym), len(out["sym"]))
    assert np.abs(out["logits"][0] - ref_logits[0]).max() <= 1e-4 * np.abs(ref_logits[0]).max()
    top2 = np.sort(ref_logits, axis=-1)[..., -2:]
    safe = np.minimum.accumulate((top2[..., 1] - top2[..., 0]) > 1e-5 * np.abs(top2[..., 1]), axis=0)   # [T,B]
    assert safe.mean() > 0.9
    assert np.array_equal(out["sym"][:steps][safe[:steps]], ref_sym[:steps][safe[:steps]])
    ok = safe[:steps]
    assert np.abs(out["logits"][:steps][ok] - ref_logits[:steps][ok]).max() <= 1e-4 * np.abs(ref_logits).max()
    tok, scores, gap = plain.beam(m["maps"], 5, max_len, 0.6)
    got_tok = np.asarray(got_beam.last_search_step_output.token_ids)
    # near-tie rule per sentence: compared exactly unless the oracle itself saw adjacent candidates within 1e-5
    clean = (np.stack(plain.beam_gaps) > 1e-5).all(axis=0)
    assert clean.mean() >= 0.8, "too many near-ties in the oracle ({} clean)".format(clean.mean())
    assert np.array_equal(got_tok[1:][:, clean], tok[1:][:, clean])
    got_scores = np.asarray(got_beam.last_search_step_output.scores)
    assert np.abs(got_scores[clean] - scores[clean]).max() <= 1e-4 * np.abs(scores[clean]).max()
