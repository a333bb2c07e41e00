"""Whole-training-step HIP graphs of the taped path: a step that is replayed from a captured graph
must produce exactly what the same step launched kernel by kernel produces -- including dropout,
whose masks advance with the device-side copy of the global step (nm_dropout / nm_sdp_attn_* read
it through a pointer, so nothing about a step is baked into the graph).

Two identical models train over the same batches, one with ``use_step_graphs`` off.  Each batch
shape is seen three times (eager, capture, replay); losses and final parameters are compared."""
import numpy as np
import pytest

from oracle import general_ref as G
from oracle import transformer_ref as TRF

pytestmark = pytest.mark.gpu


def _general(dev):
    from tests import test_general_gpu as TG
    cfg, es, et = TG.CASES["small_ini"]
    return TG._build(dev, cfg, es, et), TG._data, cfg.dec_name


def _general_mlp(dev):
    from tests import test_general_gpu as TG
    cfg, es, et = TG.CASES["gru_dropout_mlp"]
    return TG._build(dev, cfg, es, et), TG._data, cfg.dec_name


def _transformer(dev):
    from tests import test_transformer_gpu as TT
    cfg, d, ff = TT.CASES["transformer_ini"]
    return TT._build(dev, cfg, d, ff), TT._data, cfg.dec_name


def _transformer_attdrop(dev):
    from tests import test_transformer_gpu as TT
    cfg, d, ff = TT.CASES["attention_dropouts"]
    return TT._build(dev, cfg, d, ff), TT._data, cfg.dec_name


BUILDERS = {"small_ini": _general, "gru_dropout_mlp": _general_mlp, "transformer_ini": _transformer,
            "transformer_attention_dropouts": _transformer_attdrop}


def _train(dev, builder, graphs):
    m, data, dec_name = builder(dev)
    sess = m["tfm"].sessions[0]
    sess.use_step_graphs = graphs
    # two batch shapes, interleaved, three visits each
    batches = [data(5, 7, 6, 8, seed=11)[0], data(3, 5, 4, 8, seed=12)[0], data(5, 7, 6, 8, seed=13)[0],
               data(3, 5, 4, 8, seed=14)[0], data(5, 7, 6, 8, seed=15)[0], data(3, 5, 4, 8, seed=16)[0],
               data(5, 7, 6, 8, seed=17)[0]]
    losses = []
    for ds in batches:
        res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]
        losses.append(res.losses[dec_name + " - cost"])
    replayed = sum(1 for st in sess.__dict__.get("_step_graphs", {}).values() if st[0] == 2)
    return np.asarray(losses), m["store"].state_dict(), replayed


@pytest.mark.parametrize("case", sorted(BUILDERS))
def test_replayed_training_step_equals_eager(dev, case):
    l_eager, p_eager, n_eager = _train(dev, BUILDERS[case], False)
    l_graph, p_graph, n_graph = _train(dev, BUILDERS[case], True)
    assert n_eager == 0
    assert n_graph >= 1, "no training step was captured: graph_safe_training refused this model"
    assert np.all(np.isfinite(l_graph))
    assert np.abs(l_graph - l_eager).max() <= 1e-5 * np.abs(l_eager).max(), (l_eager, l_graph)
    for name, want in p_eager.items():
        if name.endswith("attn_bias") or name.endswith("keys_proj/bias"):
            continue        # their gradient is identically zero: Adam turns rounding noise into +-lr steps
        got = p_graph[name]
        assert np.abs(got - want).max() <= 1e-5 * max(float(np.abs(want).max()), 1e-3), name
    # dropout really advances: the same batch shape at steps 0 and 2 gives different losses anyway (different
    # data); check instead that the losses are not constant, i.e. replays read fresh inputs
    assert np.unique(np.round(l_graph, 6)).size > 3
