"""General (taped) path parity: NematusGRU / LSTM cells, conditional GRU, attention on input,
dropout, stacked encoders with layer norm / residual, output-projection variants -- the
configurations of the reference's tests/small.ini, tests/nematus.ini and friends.

Checker: oracle/general_ref.py (torch-CPU restatement + autograd) on the engine's own weights.
Tolerances: loss 1e-4 relative; gradients 1e-3 of each tensor's max magnitude (fp32 sums over
T steps in a different order); greedy / beam indices exact; logits 1e-4 relative."""
import numpy as np
import pytest

from oracle import general_ref as G
from oracle import nm_oracle as O

pytestmark = pytest.mark.gpu

VOCAB = 40


def _build(dev, cfg: G.Config, emb_src, emb_tgt, max_len=8, beam=3, seed=5, init_std=0.35, make_attention=None):
    from neuralmonkey_amd.attention import Attention
    from neuralmonkey_amd.decoders import BeamSearchDecoder, Decoder
    from neuralmonkey_amd.decoders import encoder_projection as EP
    from neuralmonkey_amd.decoders import output_projection as OP
    from neuralmonkey_amd.encoders import RecurrentEncoder
    from neuralmonkey_amd.model.sequence import EmbeddedSequence
    from neuralmonkey_amd.runners import BeamSearchRunner, GreedyRunner
    from neuralmonkey_amd.runtime import reset_registry
    from neuralmonkey_amd.synthetic import synthetic_vocabulary
    from neuralmonkey_amd.tf_manager import TensorFlowManager
    from neuralmonkey_amd.trainers import CrossEntropyTrainer
    reset_registry()
    vocab = synthetic_vocabulary(VOCAB)
    seq = EmbeddedSequence(name=cfg.enc_name + "_input", vocabulary=vocab, data_id="source",
                           embedding_size=emb_src, max_length=max_len)
    enc = RecurrentEncoder(name=cfg.enc_name, input_sequence=seq, rnn_layers=[tuple(l) for l in cfg.rnn_layers],
                           add_residual=cfg.add_residual, add_layer_norm=cfg.add_layer_norm,
                           include_final_layer_norm=cfg.include_final_layer_norm,
                           dropout_keep_prob=cfg.enc_dropout)
    att = make_attention(enc) if make_attention else Attention(name=cfg.att_name, encoder=enc,
                                                                 dropout_keep_prob=cfg.att_dropout)
    kind = cfg.output_projection[0]
    act = lambda name: type("Act", (), {"nm_name": name})()
    if kind == "nonlinear":
        proj = OP.nonlinear_output(emb_tgt, act(cfg.output_projection[1]), cfg.output_projection[2])
    elif kind == "nematus":
        proj = OP.nematus_output(emb_tgt, act(cfg.output_projection[1]), cfg.output_projection[2])
    elif kind == "maxout":
        proj = OP.maxout_output(cfg.output_projection[1], cfg.output_projection[2])
    elif kind == "legacy":
        legacy = {"relu": OP._legacy_relu, "identity": OP._legacy_linear}     # pylint: disable=protected-access
        proj = legacy[cfg.output_projection[1]](emb_tgt)
    else:
        proj = OP.mlp_output(list(cfg.output_projection[1]), act(cfg.output_projection[2]), cfg.output_projection[3])
    # concat is what the reference infers from rnn_size=None without a projection (decoder.py:176-191)
    enc_proj = {"linear": None, "concat": None, "empty": EP.empty_initial_state,
                "nematus": EP.nematus_projection(cfg.dec_dropout)}[cfg.encoder_projection]
    dec = Decoder(encoders=[enc], vocabulary=vocab, data_id="target", name=cfg.dec_name, max_output_len=max_len,
                  dropout_keep_prob=cfg.dec_dropout, embedding_size=emb_tgt,
                  rnn_size=None if cfg.encoder_projection == "concat" else cfg.rnn_size,
                  output_projection=proj, encoder_projection=enc_proj, attentions=[att],
                  attention_on_input=cfg.attention_on_input, rnn_cell=cfg.dec_cell,
                  conditional_gru=cfg.conditional_gru, supress_unk=cfg.supress_unk,
                  tie_embeddings=cfg.tie_embeddings, label_smoothing=cfg.label_smoothing or None)
    bdec = BeamSearchDecoder(name="beam_decoder", parent_decoder=dec, beam_size=beam, max_steps=max_len,
                             length_normalization=0.6)
    greedy = GreedyRunner(output_series="target", decoder=dec)
    brun = BeamSearchRunner(output_series="target_beam", decoder=bdec, rank=1)
    trainer = CrossEntropyTrainer(decoders=[dec], l2_weight=0.0, clip_norm=None)
    tfm = TensorFlowManager(num_sessions=1, num_threads=1, device=str(dev), seed=seed)
    tfm.initialize_sessions()
    store = tfm.sessions[0].store
    # non-degenerate weights: keep the structured initial values of biases / layer-norm scales
    rng = np.random.default_rng(seed)
    vals = store.state_dict()
    for name, v in vals.items():
        if v.ndim >= 2 or name.endswith("attn_similarity_v"):
            vals[name] = (rng.standard_normal(v.shape) * init_std).astype(np.float32)
        elif "bias" in name or name.endswith("_b") or name.endswith("beta"):
            vals[name] = (v + rng.standard_normal(v.shape) * 0.1).astype(np.float32)
    store.load_state_dict(vals)
    return dict(enc=enc, att=att, dec=dec, bdec=bdec, greedy=greedy, brun=brun, trainer=trainer, tfm=tfm,
                store=store, params=store.state_dict())


def _data(batch, slen, tlen, max_len, seed=3, with_target=True):
    from neuralmonkey_amd import synthetic
    ds = synthetic.synthetic_dataset(seed=seed, batch=batch, src_len=slen, tgt_len=tlen, vocab=VOCAB, ragged=True,
                                     with_target=with_target)
    src = O.pad_ids([list(s) for s in ds.get_series("source")], max_len)
    tgt = None
    if with_target:
        tgt = np.ascontiguousarray(O.pad_ids([list(s) for s in ds.get_series("target")], max_len,
                                             add_end_symbol=True).T)
    return ds, src, tgt


CASES = {
    # tests/small.ini shape: NematusGRU encoder and decoder, conditional GRU, dropout 0.5
    "small_ini": (G.Config(rnn_layers=((7, "bidirectional", "NematusGRU"),), enc_dropout=0.5, dec_cell="NematusGRU",
                           conditional_gru=True, dec_dropout=0.5, rnn_size=9,
                           output_projection=("nonlinear", "tanh", 1.0)), 11, 9),
    "nematus_nodrop": (G.Config(rnn_layers=((6, "bidirectional", "NematusGRU"),), dec_cell="NematusGRU",
                                conditional_gru=True, rnn_size=8, output_projection=("nematus", "tanh", 1.0)),
                       10, 8),
    "lstm_att_on_input": (G.Config(rnn_layers=((6, "bidirectional", "LSTM"),), dec_cell="LSTM",
                                   attention_on_input=True, rnn_size=8), 8, 8),
    "stacked_ln_residual": (G.Config(rnn_layers=((8, "forward", "GRU"), (8, "backward", "GRU"),
                                                 (4, "bidirectional", "LSTM")),
                                     add_layer_norm=True, add_residual=True, dec_cell="GRU", conditional_gru=True,
                                     rnn_size=8, output_projection=("maxout", 8, 1.0)), 8, 8),
    "gru_dropout_mlp": (G.Config(rnn_layers=((4, "bidirectional", "GRU"),), enc_dropout=0.8, att_dropout=0.9,
                                 dec_dropout=0.7, rnn_size=8, supress_unk=True,
                                 output_projection=("mlp", (12, 8), "relu", 0.8)), 8, 8),
    # the output projections of old experiments: dense(concat[state, context]) without the previous output
    "legacy_relu": (G.Config(rnn_layers=((4, "bidirectional", "GRU"),), rnn_size=8, dec_dropout=0.8,
                             output_projection=("legacy", "relu")), 8, 8),
    "legacy_linear_lstm": (G.Config(rnn_layers=((4, "bidirectional", "LSTM"),), dec_cell="LSTM", rnn_size=8,
                                    output_projection=("legacy", "identity")), 8, 8),
    # plain GRU model: dropout sends training to the tape, inference stays on the fused fast path
    "gru_dropout_only": (G.Config(rnn_layers=((4, "bidirectional", "GRU"),), enc_dropout=0.6, dec_dropout=0.6,
                                  rnn_size=8), 8, 8),
    # label smoothing (autoregressive.py:292-299) on the hand-scheduled fast path and on the tape
    "smoothing_fast": (G.Config(rnn_layers=((4, "bidirectional", "GRU"),), rnn_size=8, label_smoothing=0.1), 8, 8),
    "smoothing_taped": (G.Config(rnn_layers=((4, "bidirectional", "NematusGRU"),), dec_cell="NematusGRU",
                                 rnn_size=8, label_smoothing=0.2), 8, 8),
    # tests/nematus.ini shape: NematusGRU everywhere, conditional GRU, nematus initial state and output
    "nematus_ini": (G.Config(rnn_layers=((6, "bidirectional", "NematusGRU"),), enc_dropout=0.9, dec_cell="NematusGRU",
                             conditional_gru=True, dec_dropout=0.8, rnn_size=8, encoder_projection="nematus",
                             output_projection=("nematus", "tanh", 0.8)), 10, 8),
    "concat_tied": (G.Config(rnn_layers=((4, "bidirectional", "GRU"),), include_final_layer_norm=False,
                             encoder_projection="concat", rnn_size=8, tie_embeddings=True, attention_on_input=True,
                             output_projection=("nonlinear", "relu", 1.0)), 8, 8),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_train_step_gradients(dev, case):
    cfg, es, et = CASES[case]
    m = _build(dev, cfg, es, et)
    ds, src, tgt = _data(5, 7, 6, 8)
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
        if name.endswith("attn_bias"):       # identically zero (softmax shift invariance)
            assert abs(got[0]) < 1e-5 and abs(want[0]) < 1e-5
            continue
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6))
        if err > 1e-3:
            bad[name] = err
    assert not bad, "gradient mismatch: {}".format(bad)


@pytest.mark.parametrize("case", sorted(CASES))
def test_greedy_and_beam_decoding(dev, case):
    cfg, es, et = CASES[case]
    m = _build(dev, cfg, es, et)
    ds, src, _ = _data(4, 7, 6, 8, with_target=False)
    ref = G.GeneralModel(m["params"], cfg)
    ref_sym, ref_mask, ref_logits = ref.greedy(src, 8)
    dec = m["dec"]
    sess = m["tfm"].sessions[0]
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], m["att"], dec):
        fd.update(part.feed_dict(ds, train=False))
    out = sess.run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask, "logits": dec.runtime_logits}, fd)
    assert out["sym"].shape == ref_sym.shape
    assert np.array_equal(out["sym"], ref_sym)
    assert np.array_equal(out["mask"].astype(bool), ref_mask)
    lg = out["logits"]
    keep = np.abs(ref_logits) < 1e8                      # the -1e9 of supress_unk aside
    assert np.abs(lg - ref_logits)[keep].max() <= 1e-4 * np.abs(ref_logits[keep]).max()

    tok, scores, gap = ref.beam(src, 3, 8, 0.6)
    got = sess.run(m["bdec"].outputs, fd)
    got_tok = np.asarray(got.last_search_step_output.token_ids)
    got_sc = np.asarray(got.last_search_step_output.scores)
    assert gap > 1e-5, "oracle reports a near-tie ({}): pick another seed".format(gap)
    assert got_tok.shape == tok.shape
    assert np.array_equal(got_tok[1:], tok[1:])
    assert np.abs(got_sc - scores).max() <= 1e-4 * np.abs(scores).max()


def test_dropout_mask_restatement(dev):
    """nm_dropout's counter-based mask == oracle.general_ref.dropout_mask, bit for bit."""
    import torch
    from neuralmonkey_amd import ops
    x = torch.ones((37, 53), device=dev)
    for keep, salt in ((0.5, 123), (0.8, 0xDEADBEEF), (0.31, 7)):
        out = torch.empty_like(x)
        ops.dropout(x, out, keep, salt)
        want = G.dropout_mask(x.numel(), keep, salt).reshape(37, 53)
        assert np.array_equal(out.cpu().numpy(), want)
        frac = float((out > 0).float().mean())
        assert abs(frac - keep) < 0.05
    # strided views draw the same mask as the contiguous tensor of the same logical shape
    big = torch.ones((37, 80), device=dev)
    out = torch.zeros((37, 64), device=dev)
    ops.dropout(big[:, 3:56], out[:, :53], 0.5, 123)
    assert np.array_equal(out[:, :53].cpu().numpy(), G.dropout_mask(37 * 53, 0.5, 123).reshape(37, 53))


def test_elementwise_primitives(dev):
    import torch
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(0)
    a = torch.tensor(rng.standard_normal((19, 24)).astype(np.float32), device=dev)
    b = torch.tensor(rng.standard_normal((19, 24)).astype(np.float32), device=dev)
    out = torch.empty_like(a)
    for op, fn in (("add", lambda x, y: x + y), ("sub", lambda x, y: x - y), ("mul", lambda x, y: x * y)):
        ops.ew(op, a, b, out)
        assert torch.allclose(out, fn(a, b), atol=1e-6)
    for op, fn in (("sigmoid", torch.sigmoid), ("tanh", torch.tanh), ("relu", torch.relu)):
        ops.ew(op, a, None, out)
        assert torch.allclose(out.cpu(), fn(a.cpu()), atol=1e-6)
    ops.ew("sigmoid", a, None, out, alpha=1.0)
    assert torch.allclose(out.cpu(), torch.sigmoid(a.cpu() + 1.0), atol=1e-6)
    # accumulate + column slices
    acc = torch.ones((19, 40), device=dev)
    ops.ew("mul", a[:, 2:10], b[:, 5:13], acc[:, 7:15], accumulate=True)
    want = torch.ones((19, 40))
    want[:, 7:15] += (a[:, 2:10] * b[:, 5:13]).cpu()
    assert torch.allclose(acc.cpu(), want, atol=1e-6)
    y = torch.tanh(a)
    ops.ew("tanh_bwd", y, b, out)
    assert torch.allclose(out.cpu(), (b * (1 - y * y)).cpu(), atol=1e-6)
    # maxout pairs columns g and g + G (nn/projection.py:7-35)
    x = torch.tensor(rng.standard_normal((6, 10)).astype(np.float32), device=dev)
    mo, arg = torch.empty((6, 5), device=dev), torch.empty((6, 5), dtype=torch.int32, device=dev)
    ops.maxout_fwd(x, mo, arg, 2)
    assert torch.equal(mo.cpu(), torch.maximum(x[:, :5], x[:, 5:]).cpu())
    # reverse_sequence
    seq = torch.tensor(rng.standard_normal((3, 5, 4)).astype(np.float32), device=dev)
    lens = torch.tensor([5, 2, 0], dtype=torch.int32, device=dev)
    rev = torch.empty_like(seq)
    ops.reverse_sequence(seq, rev, lens)
    want = seq.cpu().clone()
    want[0] = seq[0].cpu().flip(0)
    want[1, :2] = seq[1, :2].cpu().flip(0)
    assert torch.equal(rev.cpu(), want)


SMALL_INI = """
; model sections of the reference's tests/small.ini (NematusGRU encoder/decoder, conditional GRU,
; dropout 0.5); data, evaluation and logging sections belong to the control plane
[vars]
drop_keep_p=0.5
[main]
name="small.ini shape"
batch_size=4
epochs=1
train_dataset=<train_data>
trainer=<trainer>
runners=[<runner>]
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
[decoder_vocabulary]
class=vocabulary.from_wordlist
path="{vocab}"
[my_encoder]
class=encoders.SentenceEncoder
rnn_size=7
max_input_len=5
embedding_size=11
dropout_keep_prob=$drop_keep_p
data_id="source"
vocabulary=<encoder_vocabulary>
rnn_cell="NematusGRU"
embedding_initializer=<embedding_initializer>
[embedding_initializer]
class=tf.random_uniform_initializer
minval=-0.5
maxval=0.5
[my_attention]
class=attention.Attention
encoder=<my_encoder>
initializers=[("Attention/attn_query_projection", <query_projection_initializer>)]
[query_projection_initializer]
class=tf.random_normal_initializer
stddev=0.001
[my_decoder]
class=decoders.Decoder
conditional_gru=True
encoders=[<my_encoder>]
attentions=[<my_attention>]
rnn_size=9
embedding_size=9
dropout_keep_prob=$drop_keep_p
data_id="target"
max_output_len=5
vocabulary=<decoder_vocabulary>
attention_on_input=False
rnn_cell="NematusGRU"
[optimizer]
class=tf.train.AdamOptimizer
learning_rate=0.01
[trainer]
class=trainers.CrossEntropyTrainer
decoders=[<my_decoder>]
l2_weight=1.0e-8
clip_norm=1.0
optimizer=<optimizer>
[runner]
class=runners.GreedyRunner
decoder=<my_decoder>
output_series="target"
"""


def test_small_ini_shape_experiment(dev, tmp_path):
    """BASELINE configs[0]: the model of tests/small.ini builds from INI text through the plugin
    surface, trains with dropout (falling loss) and decodes."""
    from neuralmonkey_amd.config.configuration import load_experiment
    (tmp_path / "src.txt").write_text("a b c\nb c\nc a a b\na\n")
    (tmp_path / "tgt.txt").write_text("x y\ny\nx x y\ny y\n")
    (tmp_path / "vocab.tsv").write_text("Word\tCount\n<pad>\t1\n<s>\t1\n</s>\t1\n<unk>\t1\n"
                                        "a\t9\nb\t8\nc\t7\nx\t6\ny\t5\n")
    path = tmp_path / "small.ini"
    path.write_text(SMALL_INI.format(src=tmp_path / "src.txt", tgt=tmp_path / "tgt.txt",
                                     vocab=tmp_path / "vocab.tsv"))
    model = load_experiment(str(path), device=str(dev), seed=4321)
    dec = model.trainers[0].objectives[0].decoder
    assert dec.uses_general_path(True) and dec.uses_general_path(False)
    store = model.tf_manager.sessions[0].store
    for name in ("my_encoder/rnn_0_bidirectional/bidirectional_rnn/fw/nematus_gru_cell/gates/state_proj/kernel",
                 "my_decoder/attention_decoder/nematus_gru_cell/candidate/input_proj/bias",
                 "my_decoder/attention_decoder/cond_gru_2_cell/gates/state_proj/bias",
                 "my_decoder/attention_decoder/cond_gru_2_cell/candidate/input_proj/kernel"):
        assert name in store, name
    assert "my_decoder/attention_decoder/cond_gru_2_cell/gates/input_proj/bias" not in store
    batch = next(model.train_dataset.batches())
    feedables = set.union(*[r.feedables for r in model.runners + model.trainers])
    losses = [model.tf_manager.execute(batch, feedables, model.trainers, train=True)[0].losses["my_decoder - cost"]
              for _ in range(60)]
    assert np.mean(losses[-10:]) < np.mean(losses[:10]) - 0.05, (losses[:3], losses[-3:])
    out = model.tf_manager.execute(batch, feedables, model.runners)[0]
    assert len(out.outputs["target"]) == 4
