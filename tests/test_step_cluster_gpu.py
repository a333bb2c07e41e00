"""The recurrent part of a greedy decoding step as ONE launch of workgroup clusters (dec_step_cluster_kernel,
csrc/nm_gru_cluster.hip: gates -> candidate + blend -> attention query + state part of the output projection, with
tagged hand-offs between the stages) against the three dependent step groups it replaces (nm_step_group, pinned by
tests/test_step_group_gpu.py against the oracle's Decoder.next_state, decoders/decoder.py:279-325):

  * greedy decodes through both paths give the same symbols and output states within 5e-5 after 12 recurrent
    steps (the cluster kernel splits K over 8 waves, the step groups over 16: another order of the same fp32 sums);
  * the launches really are the cluster kernel's (its workspace counts them) and leave no error behind;
  * the placement-independent variant (roles by blockIdx, write-through stores) gives the same;
  * a step kernel that gives up is recovered like a time loop: the batch is run again on the three launches.
"""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decode(dev, rnn, batch, cluster, placement=None, vocab=900, steps=12, fail=0):
    from neuralmonkey_amd import ops, synthetic
    from oracle import nm_oracle as O
    os.environ["NM_STEP_CLUSTER"] = "1" if cluster else "0"
    if placement:
        os.environ["NM_CLUSTER_PLACEMENT"] = placement
    try:
        model = synthetic.build_translation_model(vocab_src=vocab, vocab_tgt=vocab, emb=rnn, rnn=rnn, max_len=steps,
                                                  beam_size=0, device=str(dev))
        params = O.init_params(seed=3, vocab_src=vocab, vocab_tgt=vocab, emb=rnn, rnn=rnn, std=0.1)
        sess = model.tf_manager.sessions[0]
        sess.store.load_state_dict(params)
        ds = synthetic.synthetic_dataset(seed=8, batch=batch, src_len=9, tgt_len=8, vocab=vocab, ragged=True,
                                         with_target=False)
        dec = model.decoder
        fd = {}
        for part in model.greedy_runner.feedables:
            fd.update(part.feed_dict(ds, train=False))
        if fail:
            ops.gru_seq_force_give_up(fail)
        out = None
        for _ in range(3):          # eager, capture, replay
            out = model.tf_manager.execute(ds, model.greedy_runner.feedables, [model.greedy_runner],
                                           compute_losses=False)[0]
        res = sess.run({"states": dec.runtime_output_states, "sym": dec.decoded_symbols}, fd)
        steppers = [st for per in sess.__dict__.get("_fused_steppers", {}).values() for _, st in per.values()]
        epochs = [int(st.cluster_ws.view(torch.int32)[3].item()) for st in steppers
                  if getattr(st, "cluster_ws", None) is not None]
        return out.outputs["target"], np.asarray(res["states"]), np.asarray(res["sym"]), epochs, sess
    finally:
        os.environ.pop("NM_STEP_CLUSTER", None)
        os.environ.pop("NM_CLUSTER_PLACEMENT", None)


@pytest.mark.parametrize("rnn,batch", [(512, 128), (256, 40), (384, 16), (512, 7)])
def test_cluster_step_equals_the_three_step_groups(dev, rnn, batch):
    from neuralmonkey_amd import _lib
    if not _lib.load().nm_dec_step_cluster_supported(batch, rnn, 2 * rnn, rnn):
        pytest.skip("shape not taken on this device")
    want_tok, want_states, want_sym, none, _ = _decode(dev, rnn, batch, cluster=False)
    assert none == []
    got_tok, got_states, got_sym, epochs, sess = _decode(dev, rnn, batch, cluster=True)
    assert epochs and sum(epochs) >= 3, "the cluster kernel was never launched"
    assert not sess.cluster_failure() and sess.use_cluster_loops
    assert np.array_equal(got_sym, want_sym) and got_tok == want_tok
    scale = np.abs(want_states).max()
    assert np.abs(got_states - want_states).max() <= 5e-5 * max(scale, 1.0)


def test_cluster_step_without_the_placement_assumption(dev):
    from neuralmonkey_amd import _lib
    if not _lib.load().nm_dec_step_cluster_supported(40, 256, 512, 256):
        pytest.skip("shape not taken on this device")
    want_tok, want_states, _, _, _ = _decode(dev, 256, 40, cluster=False)
    got_tok, got_states, _, epochs, sess = _decode(dev, 256, 40, cluster=True, placement="blockidx")
    assert epochs and not sess.cluster_failure()
    assert got_tok == want_tok and np.abs(got_states - want_states).max() <= 5e-5 * max(np.abs(want_states).max(), 1.0)


def test_a_step_kernel_that_gives_up_is_recovered(dev):
    from neuralmonkey_amd import _lib, ops
    if not _lib.load().nm_dec_step_cluster_supported(40, 256, 512, 256):
        pytest.skip("shape not taken on this device")
    want_tok, _, _, _, _ = _decode(dev, 256, 40, cluster=False)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got_tok, _, _, _, sess = _decode(dev, 256, 40, cluster=True, fail=4)     # the encoder's loop and three steps
    ops.gru_seq_force_give_up(0)
    assert len([w for w in caught if "gave up waiting" in str(w.message)]) == 1
    assert not sess.use_cluster_loops and not sess.cluster_failure()
    assert got_tok == want_tok
