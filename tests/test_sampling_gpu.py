"""The sampling decoder body (decoders/autoregressive.py:440-493: ``logits /= temperature``; ``tf.multinomial(logits,
1)`` instead of the argmax when ``sample``; the reference's only caller is trainers/rl_trainer.py:122-125 with
train_mode=False).  TF draws from a Philox stream that no other implementation can replay, so what is checked is
  * the draw itself: ``nm_gumbel_argmax`` == argmax(x + noise) with the oracle's restatement of the noise
    (oracle/nm_oracle.py:gumbel_noise), and its distribution == softmax(x);
  * the loop around it: the engine's draws are handed back to the oracle (teacher forcing), whose logits / temperature
    must be the engine's logits (1e-4), and every draw must be the argmax of the ORACLE's logits + the restated noise
    (or within 1e-4 of it: float32 logs differ in the last bit between the two);
  * the arguments the engine refuses."""
import numpy as np
import pytest
import torch

from oracle import nm_oracle as O
from oracle import transformer_ref as TRF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,vocab", [(37, 1000), (128, 32000), (3, 7)])
def test_draw_equals_the_restated_gumbel_argmax(dev, rows, vocab):
    from neuralmonkey_amd import ops
    g = torch.Generator(device=dev).manual_seed(rows)
    x = torch.randn(rows, vocab, device=dev, generator=g) * 2.0
    out = torch.empty(rows, dtype=torch.int32, device=dev)
    for salt in (0, 12345, 0xFFFFFFFF):
        ops.gumbel_argmax(x, salt, out)
        noisy = x.cpu().numpy() + O.gumbel_noise(rows, vocab, salt)
        got = out.cpu().numpy()
        want = noisy.argmax(1)
        gap = noisy[np.arange(rows), want] - noisy[np.arange(rows), got]
        assert (gap <= 1e-4).all(), gap.max()
        assert (got == want).mean() > 0.95
    ops.gumbel_argmax(x, 1, out)
    first = out.clone()
    ops.gumbel_argmax(x, 2, out)
    assert rows < 8 or not torch.equal(first, out)          # another salt, another draw


def test_draws_follow_the_softmax_distribution(dev):
    from neuralmonkey_amd import ops
    logits = np.log(np.array([0.5, 0.25, 0.125, 0.0625, 0.0625], np.float32))
    rows = 1 << 16
    x = torch.as_tensor(np.tile(logits, (rows, 1))).to(dev)
    out = torch.empty(rows, dtype=torch.int32, device=dev)
    ops.gumbel_argmax(x, 99, out)
    counts = np.bincount(out.cpu().numpy(), minlength=5)
    p = np.exp(logits)
    sigma = np.sqrt(rows * p * (1 - p))
    assert (np.abs(counts - rows * p) < 5 * sigma).all(), counts


def _check_draws(symbols, logits_oracle, salts, end=O.END):
    """Every draw of an unfinished row is the argmax of the oracle's logits + the restated noise."""
    steps, rows = symbols.shape
    finished = np.zeros(rows, bool)
    exact = total = 0
    for t in range(steps):
        noisy = logits_oracle[t] + O.gumbel_noise(rows, logits_oracle.shape[2], salts[t])
        want = noisy.argmax(1)
        live = ~finished
        gap = noisy[np.arange(rows), want] - noisy[np.arange(rows), symbols[t]]
        assert (gap[live] <= 1e-4).all(), (t, gap[live].max())
        assert (symbols[t][finished] == 0).all()                  # <pad> once a row has emitted </s>
        exact += int((symbols[t] == want)[live].sum())
        total += int(live.sum())
        finished |= symbols[t] == end
    return exact / max(total, 1)


@pytest.mark.parametrize("temperature", [1.0, 0.7, 2.5])
def test_rnn_decoder_sampling_loop_against_the_oracle(dev, temperature):
    from neuralmonkey_amd.runtime import RunContext
    from tests.test_engine_gpu import _setup
    vocab, rnn, batch, slen, tlen = 300, 32, 12, 14, 10
    model, params, ds, src, _ = _setup(dev, vocab, rnn, rnn, batch, slen, tlen, True, beam=0, std=0.3)
    sess = model.tf_manager.sessions[0]
    fd = {}
    for f in model.greedy_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    ctx = RunContext(sess, fd)
    with torch.no_grad():
        res = model.decoder.decoding_loop(ctx, train_mode=False, sample=True, temperature=temperature)
        sym, logits = res.symbols.cpu().numpy(), res.logits.cpu().numpy()
        salts = ctx.memo[(id(model.decoder), "sampling_salts")]
        greedy = model.decoder.decoding_loop(RunContext(sess, fd), train_mode=False).symbols.cpu().numpy()
    enc = O.sentence_encoder(params, src)
    spec = O.DecoderSpec(max_output_len=max(slen, tlen))
    forced = np.zeros((spec.max_output_len, batch), np.int64)
    forced[:len(sym)] = sym
    ref = O.decoding_loop(params, spec, enc, forced, True, temperature=temperature)
    steps = len(sym)
    assert np.array_equal(ref.symbols[:steps], sym)               # the loop's own masking of finished rows
    assert float(np.abs(logits - ref.logits[:steps]).max() / np.abs(ref.logits[:steps]).max()) < 1e-4
    assert _check_draws(sym, ref.logits[:steps], salts) > 0.98
    if temperature >= 1.0:
        assert not np.array_equal(sym[:len(greedy)], greedy[:steps])      # a sample, not the argmax path
    # the next loop of the same session draws with other salts
    ctx2 = RunContext(sess, fd)
    with torch.no_grad():
        again = model.decoder.decoding_loop(ctx2, train_mode=False, sample=True, temperature=temperature)
    assert ctx2.memo[(id(model.decoder), "sampling_salts")] != salts
    assert again.symbols.shape[1] == batch


def test_temperature_alone_scales_the_logits_and_keeps_the_greedy_symbols(dev):
    from neuralmonkey_amd.runtime import RunContext
    from tests.test_engine_gpu import _setup
    model, params, ds, src, _ = _setup(dev, 300, 32, 32, 8, 12, 9, True, beam=0, std=0.3)
    sess = model.tf_manager.sessions[0]
    fd = {}
    for f in model.greedy_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    with torch.no_grad():
        cold = model.decoder.decoding_loop(RunContext(sess, fd), train_mode=False, temperature=0.5)
        sym_t, logits_t = cold.symbols.cpu().numpy(), cold.logits.cpu().numpy()
    enc = O.sentence_encoder(params, src)
    ref = O.decoding_loop(params, O.DecoderSpec(max_output_len=12), enc, None, False, temperature=0.5)
    assert np.array_equal(sym_t, ref.symbols.astype(np.int32))
    assert float(np.abs(logits_t - ref.logits).max() / np.abs(ref.logits).max()) < 1e-4


def test_transformer_sampling_loop_against_the_oracle(dev):
    from neuralmonkey_amd.runtime import RunContext
    from tests.test_transformer_gpu import _build, _data
    cfg = TRF.TConfig(depth=2, n_heads=2, n_heads_self=2, n_heads_enc=2)
    temperature, max_len, batch = 1.3, 9, 6
    m = _build(dev, cfg, 16, 24, max_len=max_len, beam=2, seed=5, init_std=0.6)
    ds, src, _ = _data(batch, 7, 6, max_len, seed=8, with_target=False)
    sess = m["tfm"].sessions[0]
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], m["dec"]):
        fd.update(part.feed_dict(ds, train=False))
    ctx = RunContext(sess, fd)
    with torch.no_grad():
        res = m["dec"].decoding_loop(ctx, train_mode=False, sample=True, temperature=temperature)
    sym, logits = res.symbols.cpu().numpy(), res.logits.cpu().numpy()
    salts = ctx.memo[(id(m["dec"]), "sampling_salts")]
    model = TRF.TransformerModel(m["params"], cfg)
    ref_sym, _, ref_logits = model.greedy(src, max_len, pick=lambda t, lg: sym[t] if t < len(sym) else lg.argmax(1),
                                          temperature=temperature)
    steps = len(sym)
    assert np.array_equal(ref_sym[:steps], sym)
    assert float(np.abs(logits - ref_logits[:steps]).max() / np.abs(ref_logits[:steps]).max()) < 1e-4
    assert _check_draws(sym, ref_logits[:steps], salts) > 0.95


def test_arguments_the_sampling_body_refuses(dev):
    from neuralmonkey_amd.runtime import RunContext
    from tests.test_engine_gpu import _setup
    model, _, ds, _, _ = _setup(dev, 64, 12, 12, 3, 6, 5, True, beam=0)
    sess = model.tf_manager.sessions[0]
    fd = {}
    for f in model.greedy_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    with pytest.raises(ValueError, match="temperature must be positive"):
        model.decoder.decoding_loop(RunContext(sess, fd), train_mode=False, sample=True, temperature=0.0)
    with pytest.raises(NotImplementedError, match="teacher-forced"):
        model.decoder.decoding_loop(RunContext(sess, fd), train_mode=True, sample=True)
