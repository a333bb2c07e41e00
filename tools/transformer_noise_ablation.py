"""Which op class makes the engine's Transformer-base logits noisier than a plain torch fp32 implementation?

At BASELINE.json's configs[4] shape (6+6 layers, d=512, 8 heads, ff 2048, B=128, len 50, V=32000; the model and data
of tests/test_transformer_fullsize_gpu.py) the float64 oracle is the yardstick.  The engine's encoder states and the
greedy logits of the first steps are measured against it
  * as shipped,
  * with ONE op class at a time replaced by its float64-exact result (inputs up-cast, computed in float64 by torch on
    the GPU, rounded once to fp32): dense products (``ops.gemm`` / ``ops.logits_stats_gemm``), layer norm
    (``ops.layer_norm_fwd`` / ``ops.add_layer_norm_fwd``), the attention core (``ops.sdp_attn_fwd`` /
    ``ops.sdp_attn_step``: QK^T, masks, softmax, PV),
  * with all three replaced (what is left is the rounding of the tensors handed from op to op),
next to the fp32 ORACLE's own distance from float64.  Forward only: the backward kernels have no Python-level seam.

    python tools/transformer_noise_ablation.py [steps=3] > profiles/r04_transformer_noise_ablation.txt
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import ops  # noqa: E402
from oracle import transformer_ref as TRF  # noqa: E402
from tests.test_transformer_gpu import _build, _data  # noqa: E402

B, LEN, VOCAB, D, FF, DEPTH = 128, 50, 32000, 512, 2048, 6
ORIG = {n: getattr(ops, n) for n in ("gemm", "logits_stats_gemm", "layer_norm_fwd", "add_layer_norm_fwd",
                                     "sdp_attn_fwd", "sdp_attn_step")}
ACT = {None: lambda x: x, "tanh": torch.tanh, "relu": torch.relu}


def gemm64(a, b, out=None, bias=None, act=None, trans_a=False, trans_b=False, accumulate=False, algo=0):
    a64, b64 = a.double(), b.double()
    if trans_a:
        a64 = a64.transpose(-1, -2)
    if trans_b:
        b64 = b64.transpose(-1, -2)
    y = a64 @ b64
    if bias is not None:
        y = y + bias.double()
    if accumulate:
        y = y + out.double()
    y = ACT[act](y).float()
    if out is None:
        return y
    out.copy_(y)
    return out


def logits_stats_gemm64(state, w, bias, stats, out=None, trans_b=False):
    # the statistics come from the fp32 kernel; only the stored logits (what this tool compares) are exact
    ORIG["logits_stats_gemm"](state, w, bias, stats, out, trans_b)
    if out is not None:
        gemm64(state, w, out=out, bias=bias, trans_b=trans_b)


def _ln64(x, gamma, beta, eps):
    x64 = x.double()
    mean = x64.mean(-1, keepdim=True)
    var = ((x64 - mean) ** 2).mean(-1, keepdim=True)
    return ((x64 - mean) * torch.rsqrt(var + eps) * gamma.double() + beta.double()).float()


def layer_norm_fwd64(x, gamma, beta, out=None, mean=None, rstd=None, eps=1e-6):
    y = _ln64(x, gamma, beta, eps)
    if out is None:
        return y
    out.copy_(y.view(out.shape))
    return out


def add_layer_norm_fwd64(a, x, gamma, beta, sum_out, out, eps=1e-6):
    sum_out.copy_((a + x).view(sum_out.shape))          # the fp32 sum IS what the next residual reads
    out.copy_(_ln64(sum_out, gamma, beta, eps).view(out.shape))
    return sum_out, out


def _sdp64(q, k, v, key_mask, heads, causal, rows_per_key):
    bq, tq, d = q.shape
    tk = k.shape[1]
    dh = d // heads
    if rows_per_key > 1:
        k, v = k.repeat_interleave(rows_per_key, 0), v.repeat_interleave(rows_per_key, 0)
        if key_mask is not None:
            key_mask = key_mask.repeat_interleave(rows_per_key, 0)
    split = lambda x, t: x.double().reshape(bq, t, heads, dh).permute(0, 2, 1, 3)
    e = (split(q, tq) / math.sqrt(dh)) @ split(k, tk).transpose(-1, -2)
    if causal:
        i = torch.arange(tq, device=q.device)[:, None] + (tk - tq)
        j = torch.arange(tk, device=q.device)[None, :]
        e = torch.where(j <= i, e, torch.full_like(e, -1e9))
    if key_mask is not None:
        m = key_mask[:, :tk].double()[:, None, None, :]
        e = e * m + (1.0 - m) * -1e9
    w = torch.softmax(e, -1)
    return (w @ split(v, tk)).permute(0, 2, 1, 3).reshape(bq, tq, d).float(), w.float()


def sdp_attn_fwd64(q, k, v, key_mask, heads, ctx, weights=None, causal=False, rows_per_key=1, keep_prob=1.0,
                   salt=0, step=None):
    assert keep_prob >= 1.0
    c, w = _sdp64(q, k, v, key_mask, heads, causal, rows_per_key)
    ctx.copy_(c)
    if weights is not None:
        weights.copy_(w.view(weights.shape))
    return ctx


def sdp_attn_step64(q, k, v, key_mask, heads, ancestors, ctx, weights=None):
    rows, tk = q.shape[0], k.shape[1]
    idx = ancestors[:, :tk].long()
    pos = torch.arange(tk, device=q.device)[None, :].expand(rows, tk)
    c, w = _sdp64(q, k[idx, pos], v[idx, pos], key_mask, heads, False, 1)
    ctx.copy_(c)
    if weights is not None:
        weights.copy_(w.view(weights.shape))
    return ctx


PATCHES = {
    "as shipped": {},
    "dense products in float64": {"gemm": gemm64, "logits_stats_gemm": logits_stats_gemm64},
    "layer norm in float64": {"layer_norm_fwd": layer_norm_fwd64, "add_layer_norm_fwd": add_layer_norm_fwd64},
    "attention core in float64": {"sdp_attn_fwd": sdp_attn_fwd64, "sdp_attn_step": sdp_attn_step64},
}
PATCHES["all three in float64"] = {k: v for p in list(PATCHES.values()) for k, v in p.items()}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    only_shipped = len(sys.argv) > 2 and sys.argv[2] == "shipped"      # e.g. under another NM_GEMM_CHAINS
    dev = torch.device("cuda:0")
    cfg = TRF.TConfig(depth=DEPTH, n_heads=8, n_heads_self=8, n_heads_enc=8)
    m = _build(dev, cfg, D, FF, max_len=LEN, beam=5, seed=13, init_std=1.2, vocab_size=VOCAB, beam_steps=10)
    ds, src, _ = _data(B, LEN, LEN - 1, LEN, seed=17, vocab_size=VOCAB)
    m["store"].load_state_dict(m["params"])
    sess = m["tfm"].sessions[0]
    sess.use_graphs = False                      # the replacements are torch code
    exact = TRF.TransformerModel(m["params"], cfg, dtype=torch.float64)
    plain = TRF.TransformerModel(m["params"], cfg)
    enc64 = exact.encode(src, False)[0].numpy()
    enc32 = plain.encode(src, False)[0].numpy()
    sym64, _, lg64 = exact.greedy(src, steps)
    _, _, lg32 = plain.greedy(src, steps)
    top2 = np.partition(lg64, VOCAB - 2, axis=-1)[..., -2:]
    safe = np.minimum.accumulate((top2[..., 1] - top2[..., 0]) > 1e-5 * np.abs(top2[..., 1]), axis=0)
    es, ls = np.abs(enc64).max(), np.abs(lg64).max()
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], m["dec"]):
        fd.update(part.feed_dict(ds, train=False))
    print("Transformer-base, B={} len<={} V={}, weights N(0, 1.2 x fan-avg scale); errors relative to the largest "
          "magnitude of the float64 tensor; greedy logits over the first {} steps, {} decided (sentence, step) pairs"
          .format(B, LEN, VOCAB, steps, int(safe.sum())))
    row = "{:<34} {:>12} {:>12} {:>12}"
    print(row.format("", "enc states", "logits max", "logits med"))
    lerr = np.abs(lg32 - lg64).max(-1)[safe] / ls
    print(row.format("fp32 oracle (torch-CPU)", "%.3g" % (np.abs(enc32 - enc64).max() / es), "%.3g" % lerr.max(),
                     "%.3g" % np.median(lerr)))
    for name, patch in PATCHES.items():
        if only_shipped and patch:
            continue
        if only_shipped:
            name += " (NM_GEMM_CHAINS={})".format(os.environ.get("NM_GEMM_CHAINS", "default"))
        for k, fn in ORIG.items():
            setattr(ops, k, patch.get(k, fn))
        out = sess.run({"logits": m["dec"].runtime_logits, "enc": m["enc"].temporal_states}, fd)
        lerr = np.abs(np.asarray(out["logits"])[:steps] - lg64).max(-1)[safe] / ls
        print(row.format("engine, " + name, "%.3g" % (np.abs(np.asarray(out["enc"]) - enc64).max() / es),
                         "%.3g" % lerr.max(), "%.3g" % np.median(lerr)))
    for k, fn in ORIG.items():
        setattr(ops, k, fn)


if __name__ == "__main__":
    main()
