"""Greedy and beam decoding of general-path (taped) decoders through captured HIP graphs: the steps
are index-addressed (``GeneralStepper.set_position``), chunks of 8 are captured on the second batch
of a shape and replayed afterwards.  Replays must reproduce exactly what the same steps launched
one by one produce -- NematusGRU / LSTM / conditional GRU decoders, flat and hierarchical
multi-source attention (whose captured steps also depend on the lengths of every encoder), multi-head
dot-product attention -- over batches of different sentences and lengths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _general(dev, case):
    from tests import test_general_gpu as T
    cfg, es, et = T.CASES[case]
    m = T._build(dev, cfg, es, et, max_len=12)
    parts = (m["enc"].input_sequence, m["enc"], m["att"], m["dec"])
    data = lambda batch, slen, seed: T._data(batch, slen, 6, 12, seed=seed, with_target=False)[0]
    return m, parts, data


def _multisource(dev, case):
    from tests import test_multisource_gpu as T
    cfg, mcfg = T.CASES[case]
    m = T._build(dev, cfg, mcfg)
    parts = (m["enc"].input_sequence, m["enc"], m["img"], m["att"], m["dec"])
    data = lambda batch, slen, seed: T._data(m, batch, seed=seed, with_target=False)[0]
    return m, parts, data


def _dotprod(dev, case):
    from tests import test_dotprod_gpu as T
    cfg, heads, keep = T.CASES[case]
    m = T._build(dev, cfg, heads, keep)
    parts = (m["enc"].input_sequence, m["enc"], m["att"], m["dec"])
    data = lambda batch, slen, seed: T._data(batch, seed=seed, with_target=False)[0]
    return m, parts, data


FAMILIES = {"small_ini": (_general, "small_ini"), "lstm_att_on_input": (_general, "lstm_att_on_input"),
            "flat_share_sentinel": (_multisource, "flat_share_sentinel"),
            "hier_noshare_sentinel_lstm": (_multisource, "hier_noshare_sentinel_lstm"),
            "two_heads": (_dotprod, "two_heads")}


def _decode(dev, family, graphs):
    build, case = FAMILIES[family]
    m, parts, data = build(dev, case)
    sess = m["tfm"].sessions[0]
    sess.use_graphs = graphs
    dec = m["dec"]
    outs = []
    for batch, slen, seed in [(4, 7, 1), (3, 5, 2), (4, 7, 3), (3, 5, 4), (4, 7, 5), (3, 5, 6), (4, 7, 7)]:
        ds = data(batch, slen, seed)
        fd = {}
        for part in parts:
            fd.update(part.feed_dict(ds, train=False))
        got = sess.run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask, "beam": m["bdec"].outputs}, fd)
        beam = got["beam"].last_search_step_output
        outs.append((got["sym"].copy(), got["mask"].copy(), np.asarray(beam.token_ids).copy(),
                     np.asarray(beam.scores).copy()))
    captured = sum(1 for g in sess._graphs.values() if g != 1) if graphs else 0
    return outs, captured


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_replayed_general_decoding_equals_eager(dev, family):
    eager, n0 = _decode(dev, family, False)
    graphed, n1 = _decode(dev, family, True)
    assert n0 == 0 and n1 >= 2, "no decode chunk was captured"
    for i, ((s0, m0, t0, c0), (s1, m1, t1, c1)) in enumerate(zip(eager, graphed)):
        assert s0.shape == s1.shape and np.array_equal(s0, s1), i
        assert np.array_equal(m0, m1), i
        assert t0.shape == t1.shape and np.array_equal(t0, t1), i
        assert np.array_equal(c0, c1), i
