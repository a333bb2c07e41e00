"""Transformer-base shape (BASELINE configs[4]: 6+6 layers, d=512, 8 heads, ff 2048, tied embeddings,
B=128, len 50, V=32000): training step time, greedy and beam-5 decode time on one GPU.
Not a bench.py line (configs[4] is a parity case); numbers go into DESIGN.md."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralmonkey_amd import synthetic  # noqa: E402
from neuralmonkey_amd.decoders import BeamSearchDecoder, TransformerDecoder  # noqa: E402
from neuralmonkey_amd.encoders import TransformerEncoder  # noqa: E402
from neuralmonkey_amd.model.sequence import EmbeddedSequence  # noqa: E402
from neuralmonkey_amd.runners import BeamSearchRunner, GreedyRunner  # noqa: E402
from neuralmonkey_amd.runtime import reset_registry  # noqa: E402
from neuralmonkey_amd.tf_manager import TensorFlowManager  # noqa: E402
from neuralmonkey_amd.trainers import CrossEntropyTrainer  # noqa: E402


def main():
    batch, length, vocab_size, d, depth = 128, 50, 32000, 512, 6
    reset_registry()
    vocab = synthetic.synthetic_vocabulary(vocab_size)
    seq = EmbeddedSequence(name="input", vocabulary=vocab, data_id="source", embedding_size=d, max_length=length,
                           scale_embeddings_by_depth=True)
    enc = TransformerEncoder(name="encoder", input_sequence=seq, ff_hidden_size=2048, depth=depth, n_heads=8)
    dec = TransformerDecoder(name="decoder", encoders=[enc], vocabulary=vocab, data_id="target", ff_hidden_size=2048,
                             n_heads_self=8, n_heads_enc=8, depth=depth, max_output_len=length, embedding_size=d)
    bdec = BeamSearchDecoder(name="beam", parent_decoder=dec, beam_size=5, max_steps=length, length_normalization=0.6)
    greedy, beam = GreedyRunner("target", dec), BeamSearchRunner("target_beam", bdec)
    trainer = CrossEntropyTrainer(decoders=[dec], l2_weight=1e-8, clip_norm=1.0)
    tfm = TensorFlowManager(num_sessions=1, num_threads=4, device="cuda:0", seed=1234)
    tfm.initialize_sessions()
    store = tfm.sessions[0].store
    print("parameters: {:.1f} M".format(store.total / 1e6))
    ds = synthetic.synthetic_dataset(seed=1, batch=batch, src_len=length, tgt_len=length, vocab=vocab_size)
    tokens = batch * length

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    only = [a[2:-5] for a in sys.argv if a.startswith("--") and a.endswith("-only")]       # --train-only, --beam-5-only ...
    if not only or "train" in only:
        t_train = timed(lambda: tfm.execute(ds, trainer.feedables, [trainer], train=True), 2, 5)
        print("train: {:.2f} ms/step  {:.0f} tok/s".format(t_train * 1e3, tokens / t_train))
    if only == ["train"]:
        return
    dsd = synthetic.synthetic_dataset(seed=2, batch=batch, src_len=length, tgt_len=length, vocab=vocab_size,
                                      with_target=False)
    # decode the full length: with tied embeddings (W = E^T, b = 0) there is no bias to push </s> down, so
    # give the final layer norm a large offset along one direction u and point the </s> embedding the other way
    u = torch.zeros(d, device="cuda:0")
    u[0] = 1.0
    store["decoder/LayerNorm/beta"].copy_(10.0 * u)
    store["decoder/word_embeddings"][2].copy_(-100.0 * u)
    for name, runner, series in (("greedy", greedy, "target"), ("beam-5", beam, "target_beam")):
        if only and name not in only:
            continue
        t = timed(lambda: tfm.execute(dsd, runner.feedables, [runner], compute_losses=False), 2, 3)
        out = tfm.execute(dsd, runner.feedables, [runner], compute_losses=False)[0]
        steps = max(len(sent) for sent in out.outputs[series])
        print("{}: {:.2f} ms/batch ({} steps, {:.0f} tok/s)".format(name, t * 1e3, steps, batch * steps / t))


if __name__ == "__main__":
    main()
