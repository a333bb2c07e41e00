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
