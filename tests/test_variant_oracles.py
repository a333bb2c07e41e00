"""The configurable-model oracles against their committed vectors (tests/golden/variants.npz, made by
tests/golden/make_variant_golden.py): one seeded model per family -- the tests/small.ini shape, a
flat and a hierarchical multi-source model, multi-head dot-product attention on an LSTM decoder, the
tests/transformer.ini shape.  CPU only: the models are built through the plugin surface on the CPU
device (no kernel runs), the oracle supplies every number."""
import os

import numpy as np
import pytest

from tests.golden import make_variant_golden as V

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "variants.npz")


@pytest.mark.parametrize("family", sorted(V.FAMILIES))
def test_variant_oracle_reproduces_golden(family):
    got = V.FAMILIES[family]()
    with np.load(GOLDEN) as want:
        w = {k.split("/", 1)[1]: want[k] for k in want.files if k.startswith(family + "/")}
    assert set(w) == set(got)
    assert abs(float(got["loss"]) - float(w["loss"])) <= 1e-5 * abs(float(w["loss"]))
    assert list(got["grad_names"]) == list(w["grad_names"])
    scale = float(np.max(w["grad_norms"]))
    assert np.abs(got["grad_norms"] - w["grad_norms"]).max() <= 1e-4 * scale
    assert np.array_equal(got["greedy_symbols"], w["greedy_symbols"])
    assert np.array_equal(got["greedy_mask"], w["greedy_mask"])
    assert abs(float(got["greedy_logit_sum"]) - float(w["greedy_logit_sum"])) <= 1e-3
    assert float(w["beam_gap"]) > 1e-5                 # no near-tie: the hypotheses are well defined
    assert np.array_equal(got["beam_tokens"], w["beam_tokens"])
    assert np.abs(got["beam_scores"] - w["beam_scores"]).max() <= 1e-5 * np.abs(w["beam_scores"]).max()


def test_variant_oracles_in_double_precision_bound_the_fp32_noise():
    """SURVEY 8c protocol (1): the fp32 oracle against itself in fp64 on the same weights."""
    import torch
    from oracle import general_ref as G
    from tests import test_general_gpu as T
    cfg, es, et = T.CASES["nematus_nodrop"]
    m = T._build(torch.device("cpu"), cfg, es, et)
    _, src, tgt = T._data(5, 7, 6, 8)
    l32, _, _ = G.GeneralModel(m["params"], cfg).train_loss(src, tgt, train=False)
    l64, _, _ = G.GeneralModel(m["params"], cfg, dtype=torch.float64).train_loss(src, tgt, train=False)
    assert abs(float(l32) - float(l64)) <= 2e-6 * abs(float(l64))
