"""CoverageAttention (attention/coverage.py:19-66, Tu et al. 2016) on the general (taped) decoder path.

Checker: oracle/coverage_ref.py (torch-CPU restatement + autograd) on the engine's own weights.  Upstream line 52
cannot build, so both sides follow the arithmetic the lines spell out (see the oracle's header).
Tolerances as tests/test_general_gpu.py: loss 1e-4 relative, gradients 1e-3 of each tensor's max magnitude,
greedy / beam indices exact unless the oracle reports a near-tie, logits 1e-4 relative.  The weights history the
decoder keeps (what coverage sums) is compared step by step."""
import numpy as np
import pytest

from oracle import coverage_ref as C
from oracle import general_ref as G

from .test_general_gpu import _build, _data

pytestmark = pytest.mark.gpu

CASES = {
    # plain GRU decoder: the coverage term is the only reason the step runs on the tape
    "gru": (G.Config(rnn_layers=((4, "bidirectional", "GRU"),), rnn_size=8), 8, 8, 5),
    # attention dropout + conditional GRU + a fertility bound other than the default
    "cond_nematus_dropout": (G.Config(rnn_layers=((6, "bidirectional", "NematusGRU"),), dec_cell="NematusGRU",
                                      conditional_gru=True, att_dropout=0.8, dec_dropout=0.7, rnn_size=8), 10, 8, 3),
    "lstm_att_on_input": (G.Config(rnn_layers=((6, "bidirectional", "LSTM"),), dec_cell="LSTM",
                                   attention_on_input=True, rnn_size=8), 8, 8, 7),
}


def _make(cfg, fert):
    from neuralmonkey_amd.attention import CoverageAttention
    return lambda enc: CoverageAttention(name=cfg.att_name, encoder=enc, dropout_keep_prob=cfg.att_dropout,
                                         state_size=6, max_fertility=fert)


@pytest.mark.parametrize("case", sorted(CASES))
def test_coverage_train_step_gradients(dev, case):
    cfg, es, et, fert = CASES[case]
    m = _build(dev, cfg, es, et, make_attention=_make(cfg, fert))
    assert {"attention/coverage_matrix", "attention/fertility_matrix"} <= set(m["store"].names())
    assert "attention/attn_bias" not in m["store"].names()        # the override never creates bias_term
    assert m["params"]["attention/coverage_matrix"].shape == (1, 1, 1, 6)          # TF shapes (coverage.py:40-46)
    ds, src, tgt = _data(5, 7, 6, 8)
    ref = C.CoverageModel(m["params"], cfg, fert, requires_grad=True)
    ref_loss, ref_g = ref.train_grads(src, tgt, train=True)
    res = m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)[0]
    assert abs(res.losses[cfg.dec_name + " - cost"] - ref_loss) < 1e-4 * abs(ref_loss)
    store = m["store"]
    bad = {}
    for name in store.names():
        got = store.g(name).cpu().numpy().reshape(-1)
        want = ref_g[name]
        want = np.zeros_like(got) if want is None else want.reshape(-1)
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6))
        if err > 1e-3:
            bad[name] = err
    assert not bad, "gradient mismatch: {}".format(bad)
    for name in ("attention/coverage_matrix", "attention/fertility_matrix"):       # and they do get a gradient
        assert np.abs(ref_g[name]).max() > 1e-6
    hist = m["att"].histories[cfg.dec_name + "_train"]
    _, _, ref_w = ref.train_loss(src, tgt, train=True)
    assert np.abs(hist.cpu().numpy() - ref_w.detach().numpy()).max() < 1e-5


@pytest.mark.parametrize("case", sorted(CASES))
def test_coverage_greedy_and_beam_decoding(dev, case):
    cfg, es, et, fert = CASES[case]
    m = _build(dev, cfg, es, et, make_attention=_make(cfg, fert))
    ds, src, _ = _data(4, 7, 6, 8, with_target=False)
    ref = C.CoverageModel(m["params"], cfg, fert)
    ref_sym, ref_mask, ref_logits = ref.greedy(src, 8)
    dec = m["dec"]
    sess = m["tfm"].sessions[0]
    fd = {}
    for part in (m["enc"].input_sequence, m["enc"], m["att"], dec):
        fd.update(part.feed_dict(ds, train=False))
    for _ in range(2):                 # twice: the second run replays captured step graphs where the loop allows
        out = sess.run({"sym": dec.decoded_symbols, "mask": dec.runtime_mask, "logits": dec.runtime_logits}, fd)
        assert np.array_equal(out["sym"], ref_sym)
        assert np.array_equal(out["mask"].astype(bool), ref_mask)
        assert np.abs(out["logits"] - ref_logits).max() <= 1e-4 * np.abs(ref_logits).max()
    # beam search: every hypothesis carries its own coverage through the reordering
    tok, scores, gap = ref.beam(src, 3, 8, 0.6)
    assert gap > 1e-5, "oracle reports a near-tie ({}): pick another seed".format(gap)
    for _ in range(2):
        got = sess.run(m["bdec"].outputs, fd)
        got_tok = np.asarray(got.last_search_step_output.token_ids)
        got_sc = np.asarray(got.last_search_step_output.scores)
        assert got_tok.shape == tok.shape
        assert np.array_equal(got_tok[1:], tok[1:])
        assert np.abs(got_sc - scores).max() <= 1e-4 * np.abs(scores).max()


def test_coverage_changes_the_distribution(dev):
    """Not a no-op: with the coverage matrix zeroed the model is the plain Bahdanau attention (minus its bias)."""
    cfg, es, et, fert = CASES["gru"]
    m = _build(dev, cfg, es, et, make_attention=_make(cfg, fert))
    ds, src, tgt = _data(5, 7, 6, 8)
    with_cov = C.CoverageModel(m["params"], cfg, fert)
    _, _, w1 = with_cov.train_loss(src, tgt, train=False)
    params = dict(m["params"])
    params["attention/coverage_matrix"] = np.zeros_like(params["attention/coverage_matrix"])
    params["attention/attn_bias"] = np.zeros(1, np.float32)
    _, _, w0 = G.GeneralModel(params, cfg).train_loss(src, tgt, train=False)
    assert np.abs(w1[0].numpy() - w0[0].numpy()).max() < 1e-6       # first step: empty history, zero coverage
    assert np.abs(w1[1:].numpy() - w0[1:].numpy()).max() > 1e-3     # later steps differ
    m["store"].load_state_dict({**m["params"], "attention/coverage_matrix": params["attention/coverage_matrix"]})
    m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=False)
    hist = m["att"].histories[cfg.dec_name + "_train"].cpu().numpy()
    assert np.abs(hist - w0.numpy()).max() < 1e-5
