"""Hidden sizes the cluster kernels do not take -- H = 300, what the reference's examples/translation.ini sets
(rnn_size=300, embedding_size=300) -- run their GRU time loops as one launch each at the next size the kernels do
take, on zero-padded operands (nn/gru.py: seq_mode / seq_fwd / seq_bwd).  Checker: oracle.torch_ref (float32 autograd
of the reference step) and the per-step path (NM_CLUSTER_PAD=0) on the same weights.  Tolerances as smoke(): loss 1e-4
relative, gradients 1e-3 of each tensor's largest entry; greedy tokens exact."""
import numpy as np
import pytest

from oracle import nm_oracle as O
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hidden", [300, 260, 264])
def test_padded_time_loops_train_and_decode_like_the_oracle(dev, hidden, monkeypatch):
    from neuralmonkey_amd import ops, synthetic
    from neuralmonkey_amd.nn import gru
    vocab, batch, slen, tlen = 300, 10, 9, 8
    calls = {"fwd": [], "bwd": []}
    real_f, real_b = ops.gru_seq_fwd, ops.gru_seq_bwd

    def spy_f(steps, ndir, rows, hsz, *a, **k):
        calls["fwd"].append((ndir, hsz))
        return real_f(steps, ndir, rows, hsz, *a, **k)

    def spy_b(steps, ndir, rows, hsz, *a, **k):
        calls["bwd"].append((ndir, hsz))
        return real_b(steps, ndir, rows, hsz, *a, **k)
    monkeypatch.setattr(ops, "gru_seq_fwd", spy_f)
    monkeypatch.setattr(ops, "gru_seq_bwd", spy_b)
    params = O.init_params(seed=3, vocab_src=vocab, vocab_tgt=vocab, emb=hidden, rnn=hidden, std=0.06)
    ds = synthetic.synthetic_dataset(seed=4, batch=batch, src_len=slen, tgt_len=tlen, vocab=vocab, ragged=True)
    src = O.pad_ids([list(s) for s in ds.get_series("source")], slen)
    tgt = np.ascontiguousarray(O.pad_ids([list(s) for s in ds.get_series("target")], slen, add_end_symbol=True).T)
    ref_loss, _, _, ref_g = TR.train_step_grads(TR.to_torch(params), src, tgt, l1_weight=0.0, l2_weight=1e-8)
    ref_tokens = O.greedy_tokens(O.decoding_loop(params, O.DecoderSpec(max_output_len=slen),
                                                 O.sentence_encoder(params, src), None, False))
    grads = {}
    for pad in (True, False):
        monkeypatch.setattr(gru, "PAD_LOOPS", pad)
        model = synthetic.build_translation_model(vocab_src=vocab, vocab_tgt=vocab, emb=hidden, rnn=hidden, max_len=slen,
                                                  beam_size=0, device=str(dev))
        store = model.tf_manager.sessions[0].store
        store.load_state_dict(params)
        before = {k: len(v) for k, v in calls.items()}
        out = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
        if pad:       # encoder (two directions) and decoder loops, forward and backward, at the padded size
            assert calls["fwd"][before["fwd"]:] == [(2, 384), (1, 384)] and sorted(calls["bwd"][before["bwd"]:]) == [(1, 384), (2, 384)]
        else:
            assert {k: len(v) for k, v in calls.items()} == before
        assert abs(out.losses["decoder - cost"] - float(ref_loss)) < 1e-4 * float(ref_loss), "pad={}".format(pad)
        bad = {}
        for name in store.names():
            got, want = store.g(name).cpu().numpy(), ref_g[name].numpy()
            if name.endswith("attn_bias"):               # identically zero (softmax shift invariance): rounding only
                assert abs(float(got.reshape(-1)[0])) < 1e-5
                continue
            err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-8))
            if err > 1e-3:
                bad[name] = err
        assert not bad, "pad={}: {}".format(pad, bad)
        grads[pad] = {n: store.g(n).cpu().numpy().copy() for n in store.names()}
        store.load_state_dict(params)                    # (the step above moved the variables)
        greedy = model.tf_manager.execute(ds, model.greedy_runner.feedables, [model.greedy_runner])[0]
        assert greedy.outputs["target"] == [[model.tgt_vocab.index_to_word[i] for i in s] for s in ref_tokens]
    for name in grads[True]:                             # both schedules: the same numbers up to summation order
        if name.endswith("attn_bias"):
            continue
        scale = max(float(np.abs(grads[False][name]).max()), 1e-8)
        assert np.abs(grads[True][name] - grads[False][name]).max() <= 2e-5 * scale, name
