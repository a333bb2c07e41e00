"""Variable sharing through ``reuse=`` on the hand-scheduled fast path: two SentenceEncoders in ONE variable
scope feed a decoder; tf.gradients adds the contributions of both uses of every shared variable
(model/parameterized.py:68-72 reuse, trainers/generic_trainer.py:136-142).  Checked by linearity against the
same model with two independent scopes holding identical weights:
    grad_shared[v] == grad_indep[v of encoder] + grad_indep[v of the copy]."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(dev, shared: bool):
    from neuralmonkey_amd.attention import Attention
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    from neuralmonkey_amd.decoders import Decoder
    from neuralmonkey_amd.encoders import SentenceEncoder
    from neuralmonkey_amd.runtime import reset_registry
    from neuralmonkey_amd.synthetic import synthetic_vocabulary
    from neuralmonkey_amd.tf_manager import TensorFlowManager
    from neuralmonkey_amd.trainers import CrossEntropyTrainer
    reset_registry()
    vocab = synthetic_vocabulary(60)
    enc = SentenceEncoder(name="encoder", vocabulary=vocab, data_id="source", embedding_size=16, rnn_size=16,
                          max_input_len=9)
    enc2 = SentenceEncoder(name="encoder_b", vocabulary=vocab, data_id="source2", embedding_size=16, rnn_size=16,
                           max_input_len=9, reuse=enc if shared else None)
    att = Attention(name="attention", encoder=enc)
    dec = Decoder(encoders=[enc, enc2], vocabulary=vocab, data_id="target", name="decoder", max_output_len=9,
                  embedding_size=16, rnn_size=16, attentions=[att])
    assert not dec.uses_general_path(True) and not enc.uses_general_path(True)        # the hand-scheduled path
    trainer = CrossEntropyTrainer(decoders=[dec], l2_weight=0.0, clip_norm=None)
    tfm = TensorFlowManager(num_sessions=1, num_threads=1, device=str(dev), seed=3)
    tfm.initialize_sessions()
    rng = np.random.default_rng(0)
    sents = lambda n, lo, hi: [["w{}".format(int(i)) for i in rng.integers(0, 56, size=int(m))]
                               for m in rng.integers(lo, hi, size=n)]
    ds = Dataset("d", {"source": sents(6, 2, 9), "source2": sents(6, 2, 9), "target": sents(6, 2, 8)},
                 BatchingScheme(batch_size=6))
    return enc, enc2, trainer, tfm, ds


def test_shared_scope_gradients_add_up(dev):
    enc, enc2, trainer, tfm, ds = _build(dev, shared=False)
    store = tfm.sessions[0].store
    rng = np.random.default_rng(1)
    vals = {n: (rng.standard_normal(v.shape) * 0.3).astype(np.float32) for n, v in store.state_dict().items()}
    for name in list(vals):                                   # the copy starts from the same weights
        if name.startswith("encoder_b/"):
            vals[name] = vals["encoder/" + name[len("encoder_b/"):]]
    # the input sequences are separate parts in both models: give them the same table too
    vals["encoder_b_input/embedding_matrix_0"] = vals["encoder_input/embedding_matrix_0"]
    store.load_state_dict(vals)
    tfm.execute(ds, trainer.feedables, [trainer], train=True)
    g_indep = {n: store.g(n).cpu().numpy().copy() for n in store.names()}

    enc, enc2, trainer, tfm, ds2 = _build(dev, shared=True)
    assert enc.shares_variables and enc2.shares_variables and enc2.scope == "encoder"
    store = tfm.sessions[0].store
    assert not any(n.startswith("encoder_b/") for n in store.names())
    store.load_state_dict({n: v for n, v in vals.items() if not n.startswith("encoder_b/")})
    tfm.execute(ds, trainer.feedables, [trainer], train=True)
    checked = 0
    for name in store.names():
        got = store.g(name).cpu().numpy()
        want = g_indep[name]
        if name.startswith("encoder/"):
            want = want + g_indep["encoder_b/" + name[len("encoder/"):]]
            checked += 1
        scale = max(float(np.abs(want).max()), 1e-6)
        assert np.abs(got - want).max() <= 2e-4 * scale, name
    assert checked >= 10
