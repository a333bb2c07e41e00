"""BASELINE.json's full configuration (configs[1]: biGRU-512 encoder + Bahdanau attention + GRU-512
decoder, B=128, src_len=tgt_len=50, V=32000) checked through size-independent properties -- the CPU
oracle would need minutes per step at this size:

  * attention weights are a distribution over the unmasked source positions, zero on padding;
  * greedy decoding is bit-reproducible and HIP-graph replay equals eager launches bit for bit;
  * beam search: scores sorted per sentence, lengths / finished flags consistent with the emitted
    tokens, and a beam of 1 without length normalisation emits the greedy sentence;
  * training: the gradient is linear in the loss weight, and the gradient of a batch equals the sum
    of the gradients of its two halves when each half is scaled by the GLOBAL token count -- the
    identity the data-parallel all-reduce relies on, here at the real shard size.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, LEN, VOCAB, HID = 128, 50, 32000, 512


@pytest.fixture(scope="module")
def model(dev):
    from neuralmonkey_amd import synthetic
    m = synthetic.build_translation_model(vocab_src=VOCAB, vocab_tgt=VOCAB, emb=HID, rnn=HID, max_len=LEN, beam_size=5,
                                          max_steps=LEN, device=str(dev), seed=1234)
    store = m.tf_manager.sessions[0].store
    rng = np.random.default_rng(1234)
    vals = store.state_dict()
    for name, v in vals.items():          # N(0, 0.05) so that attention / beams are not degenerate
        if v.ndim >= 2 and "OrthoGRUCell" not in name:
            vals[name] = (rng.standard_normal(v.shape) * 0.05).astype(np.float32)
    store.load_state_dict(vals)
    return m


def _batch(seed, ragged, with_target=True, batch=B):
    from neuralmonkey_amd import synthetic
    return synthetic.synthetic_dataset(seed=seed, batch=batch, src_len=LEN, tgt_len=LEN, vocab=VOCAB, ragged=ragged,
                                       with_target=with_target)


def _feed(model, ds, train=False):
    fd = {}
    for part in (model.encoder.input_sequence, model.encoder, model.attention, model.decoder):
        fd.update(part.feed_dict(ds, train=train))
    return fd


def test_attention_weights_are_masked_distributions(model):
    ds = _batch(5, ragged=True, with_target=False)
    sess = model.tf_manager.sessions[0]
    out = sess.run({"res": model.decoder.runtime_loop_result, "mask": model.encoder.temporal_mask}, _feed(model, ds))
    w = np.asarray(out["res"].attention_weights[0])            # [T,B,S]
    mask = np.asarray(out["mask"])                              # [B,S]
    assert w.shape[1:] == (B, LEN) and w.min() >= 0.0
    assert np.all(w[:, mask == 0] == 0.0)
    assert np.abs(w.sum(-1) - 1.0).max() < 1e-5


def test_greedy_is_reproducible_and_graph_replay_is_exact(model):
    ds = _batch(6, ragged=True, with_target=False)
    sess = model.tf_manager.sessions[0]
    fd = _feed(model, ds)
    runs = [sess.run({"sym": model.decoder.decoded_symbols, "logits": model.decoder.runtime_logits}, fd)
            for _ in range(3)]                                  # eager, capture, replay
    sess.use_graphs = False
    try:
        eager = sess.run({"sym": model.decoder.decoded_symbols, "logits": model.decoder.runtime_logits}, fd)
    finally:
        sess.use_graphs = True
    for r in runs:
        assert np.array_equal(r["sym"], eager["sym"])
        assert np.array_equal(r["logits"], eager["logits"])


def test_beam_search_invariants(model, dev):
    from neuralmonkey_amd.decoders import BeamSearchDecoder
    ds = _batch(7, ragged=True, with_target=False)
    sess = model.tf_manager.sessions[0]
    fd = _feed(model, ds)
    out = sess.run(model.beam_decoder.outputs, fd)
    tok = np.asarray(out.last_search_step_output.token_ids)     # [steps+1,B,k]
    scores = np.asarray(out.last_search_step_output.scores)
    lens = np.asarray(out.last_search_state.lengths)
    fin = np.asarray(out.last_search_state.finished)
    assert tok.shape[1:] == (B, 5) and np.all(np.diff(scores, axis=1) <= 0)          # best first
    body = tok[1:]                                              # the initial parent symbol is dropped
    has_end = (body == 2).any(0)
    assert np.array_equal(has_end, fin.astype(bool))
    first_end = np.where(has_end, (body == 2).argmax(0) + 1, body.shape[0])
    assert np.array_equal(first_end, lens)
    assert np.all(body[np.arange(body.shape[0])[:, None, None] >= first_end[None]] == 0)   # <pad> after </s>
    # beam of 1, no length normalisation == greedy (scores are sums of greedy log-probs)
    greedy = sess.run({"sym": model.decoder.decoded_symbols, "mask": model.decoder.runtime_mask}, fd)
    bd1 = BeamSearchDecoder(name="beam1", parent_decoder=model.decoder, beam_size=1, max_steps=LEN,
                            length_normalization=0.0)
    one = sess.run(bd1.outputs, fd)
    t1 = np.asarray(one.last_search_step_output.token_ids)[1:, :, 0]
    g = np.asarray(greedy["sym"])
    steps = min(t1.shape[0], g.shape[0])
    assert np.array_equal(t1[:steps], g[:steps])


def _grads(model, ds, weight=None, count_override=None):
    """Objective gradients (before regularisation / clipping) of one batch."""
    from neuralmonkey_amd.runtime import RunContext
    from neuralmonkey_amd.trainers.objective import CostObjective
    sess = model.tf_manager.sessions[0]
    trainer = model.trainer
    old = trainer.objectives
    trainer.objectives = [CostObjective(model.decoder, weight)]
    fd = _feed(model, ds, train=True)
    fd.update(trainer.feed_dict(ds, train=True))
    ctx = RunContext(sess, fd)
    dec = model.decoder
    real_count = dec.train_token_count
    if count_override is not None:
        dec.train_token_count = lambda c: count_override
    try:
        with torch.no_grad():
            trainer._objective_gradients(ctx)        # pylint: disable=protected-access
    finally:
        trainer.objectives = old
        if count_override is not None:
            del dec.train_token_count
    torch.cuda.synchronize()
    return sess.store.ensure_grad().clone(), real_count(ctx)


def test_gradient_is_linear_and_shards_add_up(model):
    full = _batch(8, ragged=True)
    g1, count = _grads(model, full)
    g2, _ = _grads(model, full, weight=2.0)
    scale = float(g1.abs().max())
    assert float((g2 - 2.0 * g1).abs().max()) <= 2e-6 * scale
    half_a, half_b = full.subset(0, B // 2), full.subset(B // 2, B // 2)
    ga, ca = _grads(model, half_a, count_override=count)
    gb, cb = _grads(model, half_b, count_override=count)
    assert ca + cb == count
    err = float((ga + gb - g1).abs().max())
    assert err <= 2e-4 * scale, (err, scale)
