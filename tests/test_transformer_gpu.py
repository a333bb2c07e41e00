"""Transformer path parity (BASELINE configs[4], tests/transformer.ini shape and larger):
scaled dot-product attention kernels, TransformerEncoder / TransformerDecoder training step
(gradients of every variable against autograd), greedy and beam decoding through the key/value
cache against the oracle's literal re-run-the-prefix decoding.

Checker: oracle/transformer_ref.py.  Tolerances: loss 1e-4 relative; gradients 1e-3 of each
tensor's max magnitude; greedy / beam indices exact (beam unless the oracle reports a near-tie);
logits 1e-4 relative."""
import math

import numpy as np
import pytest
import torch

from oracle import general_ref as G
from oracle import nm_oracle as O
from oracle import transformer_ref as TRF

pytestmark = pytest.mark.gpu

VOCAB = 40


# ------------------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("b,tq,tk,heads,dh,causal,masked,keep,rpk", [
    (3, 7, 7, 3, 2, True, True, 1.0, 1),          # transformer.ini: d=6, 3 heads
    (4, 9, 13, 2, 8, False, True, 1.0, 1),        # cross attention, Tq != Tk
    (5, 12, 12, 8, 64, True, True, 0.8, 1),       # base-model head shape with attention dropout
    (2, 50, 50, 8, 64, False, True, 1.0, 1),
    (6, 1, 11, 4, 16, False, True, 1.0, 3),       # decoding step of a beam (3 rows per sentence)
    (2, 70, 130, 1, 32, False, False, 1.0, 1),    # more keys than one wave pass, no mask
    # matrix-core kernels (nm_sdp_mfma.hip): head width 16..128, up to 128 positions
    (128, 50, 50, 8, 64, False, True, 1.0, 1),    # BASELINE configs[4]: encoder self-attention at full size
    (3, 50, 50, 8, 64, True, True, 0.9, 1),       # decoder self-attention with attention dropout
    (2, 70, 100, 4, 32, False, True, 1.0, 1),     # two query blocks of 64, 8 key tiles
    (2, 100, 128, 2, 16, True, False, 0.8, 1),    # narrow heads, the largest tile count
    (4, 20, 30, 2, 128, False, True, 1.0, 1),     # wide heads
    (6, 9, 40, 4, 64, False, True, 1.0, 3),       # several query rows per key batch (forward only)
    (2, 33, 17, 8, 64, True, True, 1.0, 1),       # odd key count (unpaired weight stores), more queries than keys
    (3, 64, 64, 2, 64, True, True, 0.7, 1),       # exactly full tiles
    # one query per row (sdp_decode_kernel: a wave per (row, head), online softmax over lane groups)
    (10, 1, 50, 8, 64, False, True, 1.0, 5),      # BASELINE configs[4] beam step: cross attention, 5 rows per sentence
    (7, 1, 37, 8, 64, True, True, 1.0, 1),        # self attention against a cache (the future mask is a no-op)
    (4, 1, 1, 2, 32, False, True, 1.0, 1),        # a single key: three of the four lane groups see nothing
    (3, 1, 130, 1, 128, False, False, 1.0, 1),    # more keys than an unrolled pass, two keys per load
    (2, 1, 9, 20, 256, False, True, 1.0, 1),      # a key row per wave pass; more heads than waves
    (5, 1, 23, 3, 4, False, True, 0.8, 1),        # one lane per key, dropout on the weights
])
def test_sdp_attention_fwd_bwd(dev, b, tq, tk, heads, dh, causal, masked, keep, rpk):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(b * 100 + tq)
    d = heads * dh
    bk = b // rpk
    q = torch.tensor(rng.standard_normal((b, tq, d)).astype(np.float32), requires_grad=True)
    k = torch.tensor(rng.standard_normal((bk, tk, d)).astype(np.float32), requires_grad=True)
    v = torch.tensor(rng.standard_normal((bk, tk, d)).astype(np.float32), requires_grad=True)
    mask = np.ones((bk, tk), np.float32)
    if masked:
        for i in range(bk):
            mask[i, rng.integers(1, tk + 1):] = 0
    mt = torch.tensor(mask)
    salt = 4242

    def ref():
        split = lambda x, n: x.view(x.shape[0], x.shape[1], heads, dh).permute(0, 2, 1, 3)
        qq = split(q / math.sqrt(dh), b)
        kk = split(k, bk).repeat_interleave(rpk, 0)
        vv = split(v, bk).repeat_interleave(rpk, 0)
        e = qq @ kk.transpose(-1, -2)
        if causal:
            i = torch.arange(tq)[:, None]
            j = torch.arange(tk)[None, :]
            e = torch.where(j <= i + tk - tq, e, torch.full_like(e, -1e9))
        m4 = mt.repeat_interleave(rpk, 0)[:, None, None, :]
        e = e * m4 + (1 - m4) * -1e9
        w = torch.softmax(e, -1)
        wd = w
        if keep < 1.0:
            wd = w * torch.from_numpy(G.dropout_mask(w.numel(), keep, salt)).view(w.shape)
        return (wd @ vv).permute(0, 2, 1, 3).reshape(b, tq, d), w
    want_ctx, want_w = ref()
    qd, kd, vd = (x.detach().to(dev) for x in (q, k, v))
    ctx = torch.empty((b, tq, d), device=dev)
    w = torch.empty((b, heads, tq, tk), device=dev)
    ops.sdp_attn_fwd(qd, kd, vd, mt.to(dev), heads, ctx, w, causal, rpk, keep, salt)
    assert np.abs(w.cpu().numpy() - want_w.detach().numpy()).max() < 2e-6
    scale = float(want_ctx.abs().max())
    assert float((ctx.cpu() - want_ctx.detach()).abs().max()) < 1e-5 * max(scale, 1.0)
    if rpk != 1:
        return
    g = torch.tensor(rng.standard_normal((b, tq, d)).astype(np.float32))
    want_ctx.backward(g)
    dq, dk, dv = (torch.zeros_like(x) for x in (qd, kd, vd))
    de = torch.empty((b, heads, tq, tk), device=dev)
    ops.sdp_attn_bwd(qd, kd, vd, mt.to(dev), w, g.to(dev), heads, dq, dk, dv, de, causal, keep, salt, accumulate=True)
    for got, want in ((dq, q.grad), (dk, k.grad), (dv, v.grad)):
        assert float((got.cpu() - want).abs().max()) < 2e-5 * max(float(want.abs().max()), 1.0)


@pytest.mark.parametrize("rows,tk,tmax,heads,dh", [(10, 7, 12, 8, 64), (640, 50, 51, 8, 64), (6, 1, 4, 2, 16)])
def test_sdp_step_through_an_ancestor_table(dev, rows, tk, tmax, heads, dh):
    """nm_sdp_attn_step reads position j of row r from cache row ancestors[r, j]: the same numbers as gathering the
    caches first (what TransformerStepper.reorder did at every beam step) and attending to the copy."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(rows + tk)
    d = heads * dh
    q = torch.tensor(rng.standard_normal((rows, 1, d)).astype(np.float32), device=dev)
    kc = torch.tensor(rng.standard_normal((rows, tmax, d)).astype(np.float32), device=dev)
    vc = torch.tensor(rng.standard_normal((rows, tmax, d)).astype(np.float32), device=dev)
    anc = torch.tensor(rng.integers(0, rows, (rows, tmax)).astype(np.int32), device=dev)
    mask = torch.tensor((rng.random((rows, tmax)) < 0.8).astype(np.float32), device=dev)
    mask[:, 0] = 1.0
    pos = torch.arange(tmax, device=dev)[None, :].expand(rows, tmax)
    kg, vg = kc[anc.long(), pos], vc[anc.long(), pos]            # [rows, tmax, d] gathered copies (indexing: plumbing)
    want, want_w = torch.empty((rows, 1, d), device=dev), torch.empty((rows, heads, 1, tk), device=dev)
    ops.sdp_attn_fwd(q, kg[:, :tk], vg[:, :tk], mask, heads, want, want_w)
    got, got_w = torch.empty_like(want), torch.empty_like(want_w)
    ops.sdp_attn_step(q, kc[:, :tk], vc[:, :tk], mask, heads, anc, got, got_w)
    assert torch.equal(got, want) and torch.equal(got_w, want_w)


def test_position_signal_matches_the_reference_formula(dev):
    from neuralmonkey_amd.nn.transformer_blocks import position_signal
    for dim, length in ((6, 7), (16, 50), (7, 5), (512, 64)):
        got = position_signal(dim, length)
        want = TRF.position_signal(dim, length).numpy()
        assert got.shape == (length, dim)
        assert np.abs(got - want).max() < 1e-5


# ------------------------------------------------------------------------------------------- models
def _build(dev, cfg: TRF.TConfig, d, ff, max_len=8, beam=3, seed=7, init_std=0.4, vocab_size=VOCAB,
           beam_steps=None):
    from neuralmonkey_amd.decoders import BeamSearchDecoder, TransformerDecoder
    from neuralmonkey_amd.encoders import TransformerEncoder
    from neuralmonkey_amd.model.sequence import EmbeddedSequence
    from neuralmonkey_amd.runtime import reset_registry
    from neuralmonkey_amd.synthetic import synthetic_vocabulary
    from neuralmonkey_amd.tf_manager import TensorFlowManager
    from neuralmonkey_amd.trainers import CrossEntropyTrainer
    reset_registry()
    vocab = synthetic_vocabulary(vocab_size)
    seq = EmbeddedSequence(name=cfg.enc_name + "_input", vocabulary=vocab, data_id="source", embedding_size=d,
                           max_length=max_len)
    enc = TransformerEncoder(name=cfg.enc_name, input_sequence=seq, ff_hidden_size=ff, depth=cfg.depth,
                             n_heads=cfg.n_heads, dropout_keep_prob=cfg.enc_dropout,
                             attention_dropout_keep_prob=cfg.enc_att_dropout,
                             use_att_transform_bias=cfg.use_att_transform_bias,
                             use_positional_encoding=cfg.use_positional_encoding,
                             target_space_id=cfg.target_space_id)
    dec = TransformerDecoder(name=cfg.dec_name, encoders=[enc], vocabulary=vocab, data_id="target",
                             ff_hidden_size=ff, n_heads_self=cfg.n_heads_self, n_heads_enc=cfg.n_heads_enc,
                             depth=cfg.depth, max_output_len=max_len, dropout_keep_prob=cfg.dec_dropout,
                             embedding_size=d, tie_embeddings=cfg.tie_embeddings,
                             self_attention_dropout_keep_prob=cfg.self_att_dropout,
                             attention_dropout_keep_prob=cfg.encdec_att_dropout,
                             use_att_transform_bias=cfg.use_att_transform_bias, supress_unk=cfg.supress_unk)
    bdec = BeamSearchDecoder(name="beam_decoder", parent_decoder=dec, beam_size=beam, max_steps=beam_steps or max_len,
                             length_normalization=0.6)
    trainer = CrossEntropyTrainer(decoders=[dec], l2_weight=0.0, clip_norm=None)
    tfm = TensorFlowManager(num_sessions=1, num_threads=1, device=str(dev), seed=seed)
    tfm.initialize_sessions()
    store = tfm.sessions[0].store
    rng = np.random.default_rng(seed)
    vals = store.state_dict()
    for name, v in vals.items():
        if v.ndim >= 2:
            vals[name] = (rng.standard_normal(v.shape) * init_std / max(1.0, (v.shape[0] / 16.0) ** 0.5)
                          ).astype(np.float32)
        else:
            vals[name] = (v + rng.standard_normal(v.shape) * 0.1).astype(np.float32)
    store.load_state_dict(vals)
    return dict(enc=enc, dec=dec, bdec=bdec, trainer=trainer, tfm=tfm, store=store, params=store.state_dict())


def _data(batch, slen, tlen, max_len, seed=3, with_target=True, vocab_size=VOCAB):
    from neuralmonkey_amd import synthetic
    ds = synthetic.synthetic_dataset(seed=seed, batch=batch, src_len=slen, tgt_len=tlen, vocab=vocab_size, ragged=True,
                                     with_target=with_target)
    src = O.pad_ids([list(s) for s in ds.get_series("source")], max_len)
    tgt = O.pad_ids([list(s) for s in ds.get_series("target")], max_len, add_end_symbol=True) if with_target else None
    return ds, src, tgt


CASES = {
    # tests/transformer.ini: d=6, 3 self-attention heads, 2 enc-dec heads, ff 10, depth 2, dropout 0.9 / 0.5
    "transformer_ini": (TRF.TConfig(depth=2, n_heads=3, n_heads_self=3, n_heads_enc=2, enc_dropout=0.9,
                                    dec_dropout=0.5), 6, 10),
    "single_head_bias_untied": (TRF.TConfig(depth=1, n_heads=1, n_heads_self=1, n_heads_enc=1,
                                            use_att_transform_bias=True, tie_embeddings=False, supress_unk=True),
                                16, 24),
    "attention_dropouts": (TRF.TConfig(depth=2, n_heads=4, n_heads_self=2, n_heads_enc=4, enc_att_dropout=0.8,
                                       self_att_dropout=0.7, encdec_att_dropout=0.9,
                                       use_att_transform_bias=True, use_positional_encoding=False), 16, 32),
    "wide": (TRF.TConfig(depth=3, n_heads=8, n_heads_self=8, n_heads_enc=8), 64, 128),
    # a target-space modality embedding (row 5 of the 32) added to the encoder input, as in the T2T imports
    "target_space": (TRF.TConfig(depth=1, n_heads=2, n_heads_self=2, n_heads_enc=2, target_space_id=5), 8, 16),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_transformer_train_step_gradients(dev, case):
    cfg, d, ff = CASES[case]
    m = _build(dev, cfg, d, ff)
    ds, src, tgt = _data(5, 7, 6, 8)
    ref = TRF.TransformerModel(m["params"], cfg, requires_grad=True)
    ref_loss, ref_g = ref.train_grads(src, tgt, train=True)
    res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]
    assert abs(res.losses[cfg.dec_name + " - cost"] - ref_loss) < 1e-4 * abs(ref_loss)
    store = m["store"]
    bad = {}
    # floor of the per-tensor scale: d/d(keys_proj/bias) is identically zero (a constant added to every
    # key shifts all energies of a query alike and softmax is shift invariant), both sides hold noise
    gmax = max(float(np.abs(g).max()) for g in ref_g.values() if g is not None)
    for name in store.names():
        got = store.g(name).cpu().numpy().reshape(-1)
        want = ref_g[name]
        want = np.zeros_like(got) if want is None else want.reshape(-1)
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-3 * gmax))
        if err > 1e-3:
            bad[name] = err
    assert not bad, "gradient mismatch: {}".format(bad)


@pytest.mark.parametrize("case", sorted(CASES))
def test_transformer_greedy_and_beam(dev, case):
    cfg, d, ff = CASES[case]
    m = _build(dev, cfg, d, ff)
    ds, src, _ = _data(4, 7, 6, 8, with_target=False)
    ref = TRF.TransformerModel(m["params"], cfg)
    enc_states, _, enc_out = ref.encode(src, False)
    ref_sym, ref_mask, ref_logits = ref.greedy(src, 8)
    dec, sess = m["dec"], m["tfm"].sessions[0]
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], dec):
        fd.update(part.feed_dict(ds, train=False))
    out = sess.run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask, "logits": dec.runtime_logits,
                    "enc": m["enc"].temporal_states, "enc_out": m["enc"].output}, fd)
    assert np.abs(out["enc"] - enc_states.numpy()).max() <= 1e-4 * np.abs(enc_states.numpy()).max()
    assert np.abs(out["enc_out"] - enc_out.numpy()).max() <= 1e-4 * np.abs(enc_out.numpy()).max()
    assert out["sym"].shape == ref_sym.shape
    assert np.array_equal(out["sym"], ref_sym)
    assert np.array_equal(out["mask"].astype(bool), ref_mask)
    keep = np.abs(ref_logits) < 1e8
    assert np.abs(out["logits"] - ref_logits)[keep].max() <= 1e-4 * np.abs(ref_logits[keep]).max()

    tok, scores, gap = ref.beam(src, 3, 8, 0.6)
    got = sess.run(m["bdec"].outputs, fd)
    got_tok = np.asarray(got.last_search_step_output.token_ids)
    assert got_tok.shape == tok.shape
    if gap > 1e-5:
        assert np.array_equal(got_tok[1:], tok[1:])
    assert np.abs(np.asarray(got.last_search_step_output.scores) - scores).max() <= 1e-4 * np.abs(scores).max()


def test_transformer_base_width_matches_the_oracle(dev):
    """BASELINE configs[4] at the model width: d = 512, 8 heads (dh = 64), ff 2048, 2 + 2 layers, V = 4000,
    32 sentences of up to 24 tokens: loss, every gradient, greedy (logits 1e-4, symbols exact up to the
    oracle's first near-tie per sentence) and beam-5 through the key/value cache against the literal
    prefix-recompute decoding of oracle/transformer_ref.py (decoders/transformer.py:487-516).

    Gradients at this width carry visible fp32 noise on BOTH sides (ReLU units whose pre-activation is within
    rounding of zero flip their derivative; four LayerNorm-ed residual blocks amplify it): the fp32 oracle
    itself is 0.1-3 % away from the same oracle run in float64.  The yardstick is therefore the float64
    oracle, and the engine must be within 4x (L2 norm) / 8x (max norm) the fp32 oracle's own distance from it (per
    tensor, floors 2e-3 / 1e-3)."""
    cfg = TRF.TConfig(depth=2, n_heads=8, n_heads_self=8, n_heads_enc=8)
    vsz, max_len, bsz = 4000, 24, 32
    # init_std 1.2: distributions sharp enough that most beam decisions clear the near-tie margin (with the
    # small cases' 0.4 the random model repeats one token and hypotheses are permutations of each other),
    # still conditioned well enough that fp32 logits stay within 2e-5 of float64
    m = _build(dev, cfg, 512, 2048, max_len=max_len, beam=5, seed=13, init_std=1.2, vocab_size=vsz)
    ds, src, tgt = _data(bsz, max_len - 1, max_len - 2, max_len, seed=17, vocab_size=vsz)
    ref_loss, ref_g = TRF.TransformerModel(m["params"], cfg, dtype=torch.float64, requires_grad=True).train_grads(
        src, tgt, train=True)
    _, g32 = TRF.TransformerModel(m["params"], cfg, requires_grad=True).train_grads(src, tgt, train=True)
    dec, sess = m["dec"], m["tfm"].sessions[0]
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], dec):
        fd.update(part.feed_dict(ds, train=False))
    out = sess.run({"sym": dec.decoded_symbols, "logits": dec.runtime_logits, "enc": m["enc"].temporal_states}, fd)
    got_beam = sess.run(m["bdec"].outputs, fd)
    res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]      # mutates the variables
    assert abs(res.losses[cfg.dec_name + " - cost"] - ref_loss) < 1e-4 * abs(ref_loss)
    store = m["store"]
    gmax = max(float(np.abs(g).max()) for g in ref_g.values() if g is not None)
    bad = {}
    for name in store.names():
        got = store.g(name).cpu().numpy().reshape(-1).astype(np.float64)
        want = ref_g[name]
        want = np.zeros_like(got) if want is None else want.reshape(-1)
        w32 = np.zeros_like(got) if g32[name] is None else g32[name].reshape(-1).astype(np.float64)
        scale2, scalem = max(np.linalg.norm(want), 1e-3 * gmax), max(np.abs(want).max(), 1e-3 * gmax)
        err2, noise2 = np.linalg.norm(got - want) / scale2, np.linalg.norm(w32 - want) / scale2
        errm, noisem = np.abs(got - want).max() / scalem, np.abs(w32 - want).max() / scalem
        if err2 > max(2e-3, 4 * noise2) or errm > max(1e-3, 8 * noisem):
            bad[name] = (float(err2), float(noise2), float(errm), float(noisem))
    assert not bad, "gradient mismatch (l2 err, l2 fp32-oracle noise, max err, max noise): {}".format(bad)

    plain = TRF.TransformerModel(m["params"], cfg)
    enc_states, _, _ = plain.encode(src, False)
    assert np.abs(out["enc"] - enc_states.numpy()).max() <= 1e-4 * np.abs(enc_states.numpy()).max()
    ref_sym, _, ref_logits = plain.greedy(src, max_len)
    steps = min(len(ref_sym), len(out["sym"]))
    keep = np.abs(ref_logits[:steps]) < 1e8
    masked = np.where(keep, ref_logits[:steps], -np.inf)
    top2 = np.sort(masked, axis=-1)[..., -2:]
    safe = np.minimum.accumulate((top2[..., 1] - top2[..., 0]) > 1e-5 * np.abs(top2[..., 1]), axis=0)
    assert safe.mean() > 0.9
    assert np.array_equal(out["sym"][:steps][safe], ref_sym[:steps][safe])
    diff = np.where(keep, np.abs(out["logits"][:steps] - ref_logits[:steps]), 0.0).max(-1)
    assert diff[safe].max() <= 1e-4 * np.abs(ref_logits[:steps][keep]).max()
    tok, scores, gap = plain.beam(src, 5, max_len, 0.6)
    got_tok = np.asarray(got_beam.last_search_step_output.token_ids)
    assert got_tok.shape == tok.shape
    # near-tie rule per sentence: compared exactly unless the oracle itself saw adjacent candidates within 1e-5
    clean = (np.stack(plain.beam_gaps) > 1e-5).all(axis=0)
    assert clean.mean() >= 0.8, "too many near-ties in the oracle ({} clean)".format(clean.mean())
    assert np.array_equal(got_tok[1:][:, clean], tok[1:][:, clean])
    got_scores = np.asarray(got_beam.last_search_step_output.scores)
    assert np.abs(got_scores[clean] - scores[clean]).max() <= 1e-4 * np.abs(scores[clean]).max()


TRANSFORMER_INI = """
; model / trainer / runner sections of the reference's tests/transformer.ini and tests/beamsearch.ini
[main]
name="transformer.ini shape"
tf_manager=<tf_manager>
batch_size=4
epochs=1
train_dataset=<train_data>
trainer=<trainer>
runners=[<runner>, <beam_runners>]
[tf_manager]
class=tf_manager.TensorFlowManager
num_threads=4
num_sessions={sessions}
[batching]
class=dataset.BatchingScheme
batch_size=4
[train_data]
class=dataset.load
series=["source", "target"]
data=["{src}", "{tgt}"]
batching=<batching>
[encoder_vocabulary]
class=vocabulary.from_wordlist
path="{vocab}"
[inpseq]
class=model.sequence.EmbeddedSequence
name="input"
embedding_size=6
max_length=7
data_id="source"
vocabulary=<encoder_vocabulary>
[encoder]
class=encoders.transformer.TransformerEncoder
name="transformer_encoder"
input_sequence=<inpseq>
ff_hidden_size=10
depth=2
n_heads=3
dropout_keep_prob=0.9
[decoder]
class=decoders.transformer.TransformerDecoder
name="decoder"
encoders=[<encoder>]
dropout_keep_prob=0.5
data_id="target"
max_output_len=5
vocabulary=<encoder_vocabulary>
embedding_size=6
ff_hidden_size=10
depth=2
n_heads_self=3
n_heads_enc=2
[trainer]
class=trainers.delayed_update_trainer.DelayedUpdateTrainer
batches_per_update=2
l2_weight=1.0e-8
clip_norm=1.0
objectives=[<obj>]
optimizer=<lazyadam_g>
[obj]
class=trainers.cross_entropy_trainer.CostObjective
decoder=<decoder>
[decayed_lr]
class=functions.noam_decay
learning_rate=0.2
model_dimension=6
warmup_steps=20
[lazyadam_g]
class=tf.contrib.opt.LazyAdamOptimizer
beta1=0.9
beta2=0.98
epsilon=1.0e-9
learning_rate=<decayed_lr>
[runner]
class=runners.GreedyRunner
decoder=<decoder>
output_series="target"
[beam_decoder]
class=decoders.beam_search_decoder.BeamSearchDecoder
parent_decoder=<decoder>
beam_size=3
max_steps=5
length_normalization=0.6
[beam_runners]
class=runners.beam_search_runner_range
output_series="target_beam"
decoder=<beam_decoder>
max_rank=2
"""


def _write_transformer_ini(tmp_path, sessions):
    (tmp_path / "src.txt").write_text("a b c\nb c\nc a a b\na\n")
    (tmp_path / "tgt.txt").write_text("b a\nc\nb b a\nc c\n")
    (tmp_path / "vocab.tsv").write_text("Word\tCount\n<pad>\t1\n<s>\t1\n</s>\t1\n<unk>\t1\na\t9\nb\t8\nc\t7\n")
    path = tmp_path / "transformer_{}.ini".format(sessions)
    path.write_text(TRANSFORMER_INI.format(src=tmp_path / "src.txt", tgt=tmp_path / "tgt.txt",
                                           vocab=tmp_path / "vocab.tsv", sessions=sessions))
    return str(path)


def test_transformer_ini_experiment_and_ensemble_invariant(dev, tmp_path):
    """The model of tests/transformer.ini (+ the beam runners of tests/beamsearch.ini) builds from INI
    text, trains with DelayedUpdateTrainer / noam_decay / LazyAdam (falling loss), decodes greedily and
    with beam search; a 2-session ensemble of the trained model with itself decodes the same sentences
    (tests/beamsearch_ensembles.ini, tests/tests_run.sh:41-50)."""
    from neuralmonkey_amd.config.configuration import load_experiment
    model = load_experiment(_write_transformer_ini(tmp_path, 1), device=str(dev), seed=1234)
    tfm = model.tf_manager
    batch = next(model.train_dataset.batches())
    feedables = set.union(*[r.feedables for r in model.runners + model.trainers])
    losses = [tfm.execute(batch, feedables, model.trainers, train=True)[0].losses["decoder - cost"]
              for _ in range(80)]
    assert tfm.sessions[0].global_step == 40                      # one update per two batches
    assert np.mean(losses[-10:]) < np.mean(losses[:10]) - 0.05, (losses[:3], losses[-3:])
    single = tfm.execute(batch, feedables, model.runners, compute_losses=False)
    assert [len(r.outputs[s]) for r, s in zip(single, ("target", "target_beam.rank001", "target_beam.rank002"))] \
        == [4, 4, 4]
    trained = tfm.sessions[0].store.state_dict()
    ens = load_experiment(_write_transformer_ini(tmp_path, 2), device=str(dev), seed=99)
    for sess in ens.tf_manager.sessions:
        sess.store.load_state_dict(trained)
    feedables2 = set.union(*[r.feedables for r in ens.runners])
    both = ens.tf_manager.execute(next(ens.train_dataset.batches()), feedables2, ens.runners, compute_losses=False)
    for one, two, series in zip(single, both, ("target", "target_beam.rank001", "target_beam.rank002")):
        assert one.outputs[series] == two.outputs[series], series
