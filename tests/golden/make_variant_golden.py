"""Golden vectors of the configurable-model oracles (oracle/general_ref.py, multisource_ref.py,
dotprod_ref.py, transformer_ref.py): loss, gradient norms, greedy symbols and beam hypotheses of one
seeded model per family.  Like tiny.npz / mid.npz these pin the ORACLE against accidental change
(frozen oracle outputs; parity with the reference itself is pinned separately, by the reference-executed fixtures of
tests/golden/make_reference_exec_golden.py).  The
parameters come from the engine's own variable store built on the CPU device (host-side plumbing
only; no kernel runs), so the generator also exercises the plugin surface.

    python tests/golden/make_variant_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CPU = torch.device("cpu")


def _summary(loss, grads, greedy, beam):
    names = sorted(n for n, g in grads.items() if g is not None)
    sym, mask, logits = greedy
    tok, scores, gap = beam
    return {"loss": np.float64(loss), "grad_names": np.array(names),
            "grad_norms": np.array([float(np.sqrt((grads[n].astype(np.float64) ** 2).sum())) for n in names]),
            "greedy_symbols": sym.astype(np.int64), "greedy_mask": mask.astype(np.int8),
            "greedy_logit_sum": np.float64(logits.astype(np.float64).sum()),
            "beam_tokens": tok.astype(np.int64), "beam_scores": scores.astype(np.float64), "beam_gap": np.float64(gap)}


def general():
    from oracle import general_ref as G
    from tests import test_general_gpu as T
    cfg, es, et = T.CASES["small_ini"]
    m = T._build(CPU, cfg, es, et)
    _, src, tgt = T._data(5, 7, 6, 8)
    ref = G.GeneralModel(m["params"], cfg, requires_grad=True)
    loss, grads = ref.train_grads(src, tgt, train=True)
    plain = G.GeneralModel(m["params"], cfg)
    return _summary(loss, grads, plain.greedy(src, 8), plain.beam(src, 3, 8, 0.6))


def multisource():
    from oracle import multisource_ref as M
    from tests import test_multisource_gpu as T
    cfg, mcfg = T.CASES["flat_share_sentinel"]
    m = T._build(CPU, cfg, mcfg)
    _, src, tgt = T._data(m, 5)
    ref = M.MultiSourceModel(m["params"], cfg, mcfg, requires_grad=True)
    loss, grads = ref.train_grads(src, tgt, train=True)
    plain = M.MultiSourceModel(m["params"], cfg, mcfg)
    return _summary(loss, grads, plain.greedy(src, 8), plain.beam(src, 3, 8, 0.6))


def hierarchical():
    from oracle import multisource_ref as M
    from tests import test_multisource_gpu as T
    cfg, mcfg = T.CASES["hier_noshare_sentinel_lstm"]
    m = T._build(CPU, cfg, mcfg)
    _, src, tgt = T._data(m, 5)
    ref = M.MultiSourceModel(m["params"], cfg, mcfg, requires_grad=True)
    loss, grads = ref.train_grads(src, tgt, train=True)
    plain = M.MultiSourceModel(m["params"], cfg, mcfg)
    return _summary(loss, grads, plain.greedy(src, 8), plain.beam(src, 3, 8, 0.6))


def dotprod():
    from oracle import dotprod_ref as D
    from tests import test_dotprod_gpu as T
    cfg, heads, keep = T.CASES["four_heads_dropout_lstm"]
    m = T._build(CPU, cfg, heads, keep)
    _, src, tgt = T._data(5)
    ref = D.DotProdModel(m["params"], cfg, heads, keep, requires_grad=True)
    loss, grads = ref.train_grads(src, tgt, train=True)
    plain = D.DotProdModel(m["params"], cfg, heads, keep)
    return _summary(loss, grads, plain.greedy(src, 8), plain.beam(src, 3, 8, 0.6))


def transformer():
    from oracle import transformer_ref as TRF
    from tests import test_transformer_gpu as T
    cfg, d, ff = T.CASES["transformer_ini"]
    m = T._build(CPU, cfg, d, ff)
    _, src, tgt = T._data(5, 7, 6, 8)
    ref = TRF.TransformerModel(m["params"], cfg, requires_grad=True)
    loss, grads = ref.train_grads(src, tgt, train=True)
    plain = TRF.TransformerModel(m["params"], cfg)
    return _summary(loss, grads, plain.greedy(src, 8), plain.beam(src, 3, 8, 0.6))


FAMILIES = {"general_small_ini": general, "flat_share_sentinel": multisource,
            "hier_noshare_sentinel_lstm": hierarchical, "dotprod_four_heads": dotprod,
            "transformer_ini": transformer}

if __name__ == "__main__":
    out = {}
    for name, fn in FAMILIES.items():
        for key, val in fn().items():
            out["{}/{}".format(name, key)] = val
        print(name, "loss", float(out[name + "/loss"]), "grads", len(out[name + "/grad_names"]),
              "beam gap", float(out[name + "/beam_gap"]))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants.npz"), **out)
