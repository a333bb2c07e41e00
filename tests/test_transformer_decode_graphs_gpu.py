"""Transformer decoding through captured HIP graphs: greedy and beam-search step chunks are captured
on the second batch of a shape and replayed afterwards (the steps are index-addressed:
``TransformerStepper.set_position``); what a replay produces must be exactly what the same steps
launched one by one produce, batch after batch with different sentences and lengths."""
import numpy as np
import pytest

from oracle import transformer_ref as TRF

pytestmark = pytest.mark.gpu


def _decode(dev, graphs: bool, case: str):
    from tests import test_transformer_gpu as T
    cfg, d, ff = T.CASES[case]
    m = T._build(dev, cfg, d, ff, max_len=12, beam=3)
    sess = m["tfm"].sessions[0]
    sess.use_graphs = graphs
    dec, enc = m["dec"], m["enc"]
    outs = []
    # two source shapes, interleaved: every shape is decoded eagerly, captured, then replayed twice
    plan = [(4, 7, 1), (3, 5, 2), (4, 7, 3), (3, 5, 4), (4, 7, 5), (3, 5, 6), (4, 7, 7), (3, 5, 8)]
    for batch, slen, seed in plan:
        ds, _, _ = T._data(batch, slen, 6, 12, seed=seed, with_target=False)
        fd = {}
        for part in (enc.input_sequence, enc, dec):
            fd.update(part.feed_dict(ds, train=False))
        got = sess.run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask, "beam": m["bdec"].outputs}, fd)
        beam = got["beam"].last_search_step_output
        outs.append((got["sym"].copy(), got["mask"].copy(), np.asarray(beam.token_ids).copy(),
                     np.asarray(beam.scores).copy()))
    replayed = sum(1 for g in sess._graphs.values() if g != 1) if graphs else 0
    return outs, replayed


@pytest.mark.parametrize("case", ["transformer_ini", "wide"])
def test_replayed_transformer_decoding_equals_eager(dev, case):
    eager, n0 = _decode(dev, False, case)
    graphed, n1 = _decode(dev, True, case)
    assert n0 == 0 and n1 >= 2, "no decode chunk was captured"
    for (s0, m0, t0, c0), (s1, m1, t1, c1) in zip(eager, graphed):
        assert s0.shape == s1.shape and np.array_equal(s0, s1)
        assert np.array_equal(m0, m1)
        assert t0.shape == t1.shape and np.array_equal(t0, t1)
        assert np.array_equal(c0, c1)


def test_transformer_decode_oracle_parity_survives_replay(dev):
    """The third decode of a shape (a graph replay) against the CPU oracle."""
    from tests import test_transformer_gpu as T
    cfg, d, ff = T.CASES["transformer_ini"]
    m = T._build(dev, cfg, d, ff)
    sess, dec, enc = m["tfm"].sessions[0], m["dec"], m["enc"]
    ref = TRF.TransformerModel(m["params"], cfg)
    for seed in (11, 12, 13):
        ds, src, _ = T._data(4, 7, 6, 8, seed=seed, with_target=False)
        fd = {}
        for part in (enc.input_sequence, enc, dec):
            fd.update(part.feed_dict(ds, train=False))
        got = sess.run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask}, fd)
        want_sym, want_mask, _ = ref.greedy(src, 8)
        assert np.array_equal(got["sym"], want_sym), seed
        assert np.array_equal(got["mask"].astype(bool), want_mask), seed
