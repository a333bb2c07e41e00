"""The CPU oracle against its committed golden vectors and against itself
(NumPy restatement vs torch restatement vs fp64), plus the size-independent
properties the domain offers.  No GPU."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import nm_oracle as O
from oracle import torch_ref as TR

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)


@pytest.mark.parametrize("case", ["tiny", "mid"])
def test_oracle_reproduces_golden_vectors(case):
    gold = np.load(os.path.join(HERE, "golden", case + ".npz"))
    _, out = MG.build(case)
    for key in ("src", "tgt", "greedy_symbols", "beam_token_ids", "meta"):
        assert np.array_equal(out[key], gold[key]), key
    for key in ("enc_states", "enc_final", "greedy_logits", "beam_scores", "train_loss", "runtime_loss",
                "torch_loss", "l2", "grad_logit_w", "grad_attn_v", "grad_enc_emb"):
        assert np.allclose(out[key], gold[key], rtol=2e-5, atol=1e-6), key


def test_numpy_and_torch_restatements_agree():
    p = O.init_params(seed=1, vocab_src=50, vocab_tgt=60, emb=12, rnn=8, dec_rnn=12, std=0.3)
    src, tgt = O.synthetic_batch(seed=2, batch=5, src_len=7, tgt_len=6, vocab=50, ragged=True)
    enc = O.sentence_encoder(p, src)
    res = O.decoding_loop(p, O.DecoderSpec(max_output_len=6), enc, tgt, True)
    tp = TR.to_torch(p, requires_grad=False)
    st, mask, fin = TR.encoder(tp, src)
    assert np.allclose(st.numpy(), enc.temporal_states, atol=2e-6)
    assert np.allclose(fin.numpy(), enc.output, atol=2e-6)
    assert abs(float(TR.train_forward(tp, src, tgt)) - float(O.train_loss(res, tgt))) < 1e-5
    assert abs(float(TR.train_forward(tp, src, tgt, hoist_logits=True)) - float(O.train_loss(res, tgt))) < 1e-5


def test_fp32_noise_floor_against_fp64():
    """Protocol step (1) of SURVEY 8c: oracle fp32 vs fp64 bounds intrinsic noise."""
    p32 = O.init_params(seed=5, vocab_src=80, vocab_tgt=80, emb=16, rnn=16, std=0.2)
    p64 = {k: v.astype(np.float64) for k, v in p32.items()}
    src, tgt = O.synthetic_batch(seed=6, batch=6, src_len=9, tgt_len=8, vocab=80, ragged=True)
    spec_ = O.DecoderSpec(max_output_len=9)
    l32 = O.decoding_loop(p32, spec_, O.sentence_encoder(p32, src), tgt, True).logits
    l64 = O.decoding_loop(p64, spec_, O.sentence_encoder(p64, src), tgt, True).logits
    assert l64.dtype == np.float64
    assert np.abs(l32 - l64).max() / np.abs(l64).max() < 1e-5


def test_reverse_sequence_and_masking_semantics():
    x = np.arange(2 * 4 * 1, dtype=np.float32).reshape(2, 4, 1)
    r = O.reverse_sequence(x, np.array([4, 2]))
    assert r[0, :, 0].tolist() == [3, 2, 1, 0] and r[1, :, 0].tolist() == [5, 4, 6, 7]
    p = {"gates_kernel": np.zeros((3, 4), np.float32), "gates_bias": np.ones(4, np.float32),
         "cand_kernel": np.ones((3, 2), np.float32), "cand_bias": np.zeros(2, np.float32)}
    xs = np.ones((2, 3, 1), np.float32)
    outs, fin = O.dynamic_rnn(O.gru_cell, xs, np.array([3, 1]), p)
    assert np.all(outs[1, 1:] == 0)                    # output zeroed past the length
    assert np.allclose(fin[1], outs[1, 0])             # state copied through


def test_attention_weights_properties():
    rng = np.random.default_rng(0)
    q, hf, st = rng.standard_normal((4, 6)), rng.standard_normal((4, 5, 8)), rng.standard_normal((4, 5, 3))
    mask = np.ones((4, 5))
    mask[1, 3:] = 0
    mask[2, :] = 0
    ap = {"query_w": rng.standard_normal((6, 8)), "query_b": np.zeros(8), "v": rng.standard_normal(8),
          "bias": np.float64(0.3)}
    ctx, w = O.attention_step(q, hf, st, mask, ap)
    assert np.allclose(w[0].sum(), 1.0) and np.allclose(w[1].sum(), 1.0, atol=1e-6)
    assert np.all(w[1, 3:] == 0)
    assert np.all(w[2] == 0) and np.all(ctx[2] == 0)           # 0/(0+1e-8)
    ap2 = dict(ap, bias=np.float64(7.0))                        # scalar bias: softmax shift invariance
    assert np.allclose(O.attention_step(q, hf, st, mask, ap2)[1], w)


def test_top_k_tie_order_and_beam_invariants():
    x = np.array([[1.0, 3.0, 3.0, 2.0, 3.0]])
    vals, idx = O.top_k(x, 3)
    assert idx.tolist() == [[1, 2, 4]] and vals.tolist() == [[3.0, 3.0, 3.0]]
    p = O.init_params(seed=9, vocab_src=40, vocab_tgt=40, emb=8, rnn=8, std=0.3)
    src, _ = O.synthetic_batch(seed=10, batch=3, src_len=6, tgt_len=5, vocab=40, ragged=True)
    spec_ = O.DecoderSpec(max_output_len=6)
    enc = O.sentence_encoder(p, src)
    res = O.beam_search(p, spec_, enc, 3, 5, 0.6)
    assert res.token_ids.shape[0] <= 6
    assert np.all(np.diff(res.scores, axis=1) <= 1e-6)          # top_k output is sorted
    # batch-1 (the reference's broadcast regime) equals row 0 of the batched search
    enc1 = O.sentence_encoder(p, src[:1])
    one = O.beam_search(p, spec_, enc1, 3, 5, 0.6)
    n = min(len(one.token_ids), len(res.token_ids))
    assert np.array_equal(one.token_ids[:n, 0], res.token_ids[:n, 0])
    # beam size 1 is greedy search
    g = O.decoding_loop(p, spec_, enc, None, False)
    b1 = O.beam_search(p, spec_, enc, 1, 6, 0.0)
    m = min(len(g.symbols), len(b1.token_ids) - 1)
    for b in range(3):
        gs = g.symbols[:m, b]
        cut = (list(gs).index(O.END) + 1) if O.END in gs else m
        assert np.array_equal(b1.token_ids[1:cut + 1, b, 0], gs[:cut])


def test_adam_and_clip_match_torch():
    rng = np.random.default_rng(3)
    theta = rng.standard_normal(50).astype(np.float32)
    tt = torch.tensor(theta, requires_grad=True)
    opt = torch.optim.Adam([tt], lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
    m = np.zeros_like(theta)
    v = np.zeros_like(theta)
    for t in range(1, 4):
        g = rng.standard_normal(50).astype(np.float32)
        tt.grad = torch.tensor(g)
        opt.step()
        theta, m, v = O.adam_step(theta, g, m, v, t)
    assert np.allclose(theta, tt.detach().numpy(), atol=2e-7)
    g = rng.standard_normal(10).astype(np.float32) * 5
    assert abs(np.linalg.norm(O.clip_by_norm(g, 1.0)) - 1.0) < 1e-6
    assert np.array_equal(O.clip_by_norm(g * 1e-3, 1.0), (g * 1e-3).astype(np.float32))


def test_regularizable_follows_bias_regex():
    names = ["a/kernel", "a/bias", "att/attn_bias", "att/attn_projection_bias", "dec/state_to_word_b",
             "enc/LayerNorm/beta", "enc/gates/Bias"]
    assert O.regularizable(names) == ["a/kernel", "dec/state_to_word_b", "enc/LayerNorm/beta"]


def test_gumbel_noise_is_finite_for_every_bit_pattern():
    """u = ((bits >> 9) + 0.5) / 2^23 stays strictly inside (0, 1) in float32 (csrc/nm_logits.hip:gumbel_argmax_kernel
    and its restatement): the round-4 form ((bits >> 8) + 0.5) / 2^24 rounded to 1.0 for bits = 0xFFFFFFFF, an
    infinite noise term that made that column win whatever its logit."""
    for bits in (0x00000000, 0x000001FF, 0xFFFFFE00, 0xFFFFFFFF):
        u = (np.float32(bits >> 9) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
        assert np.float32(0.0) < u < np.float32(1.0)
        assert np.isfinite(-np.log(-np.log(u)))
    old = (np.float32(0xFFFFFFFF >> 8) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    assert old == np.float32(1.0)
    noise = O.gumbel_noise(64, 4096, 12345)
    assert noise.dtype == np.float32 and np.isfinite(noise).all()


def test_adadelta_restatement_follows_tf_apply_adadelta():
    """``TR.clip_and_adadelta`` (the checker of the HIP Adadelta kernel) against the four lines of TensorFlow 1.12's
    ApplyAdadelta written out in NumPy float64, two updates, with the clip biting on one tensor and sparing the other
    (tests/bpe.ini:102-108: rho 0.95, epsilon 1e-6)."""
    rng = np.random.default_rng(4)
    p = {"a": torch.tensor(rng.standard_normal((3, 4)), dtype=torch.float32),
         "b": torch.tensor(rng.standard_normal(5), dtype=torch.float32)}
    acc = {k: torch.zeros_like(v) for k, v in p.items()}
    acc_u = {k: torch.zeros_like(v) for k, v in p.items()}
    ref = {k: v.double().numpy().copy() for k, v in p.items()}
    r_acc = {k: np.zeros_like(v) for k, v in ref.items()}
    r_acc_u = {k: np.zeros_like(v) for k, v in ref.items()}
    lr, rho, eps, clip = 0.5, 0.95, 1e-6, 1.0
    for step in range(2):
        grads = {"a": torch.tensor(rng.standard_normal((3, 4)) * 3.0, dtype=torch.float32),        # norm > clip
                 "b": torch.tensor(rng.standard_normal(5) * 0.1, dtype=torch.float32)}             # norm < clip
        assert float(grads["a"].norm()) > clip > float(grads["b"].norm())
        TR.clip_and_adadelta(p, grads, acc, acc_u, clip, lr=lr, rho=rho, eps=eps)
        for k in ref:
            g = grads[k].double().numpy()
            g = g * (clip / max(np.sqrt((g * g).sum()), clip))
            r_acc[k] = rho * r_acc[k] + (1 - rho) * g * g
            update = np.sqrt(r_acc_u[k] + eps) / np.sqrt(r_acc[k] + eps) * g
            ref[k] = ref[k] - lr * update
            r_acc_u[k] = rho * r_acc_u[k] + (1 - rho) * update * update
    for k in ref:
        assert np.allclose(p[k].numpy(), ref[k], rtol=1e-5, atol=1e-7)
        assert np.allclose(acc[k].numpy(), r_acc[k], rtol=1e-5) and np.allclose(acc_u[k].numpy(), r_acc_u[k], rtol=1e-5)

