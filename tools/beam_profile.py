"""Beam-5 decode batches only, headline shape (B=128, 50 steps, H=512, V=32000; </s> unreachable so
every hypothesis runs all steps) -- the workload of bench.py's beam leg, for per-kernel profiles."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=4)
    args = ap.parse_args()
    from neuralmonkey_amd import synthetic
    model = synthetic.build_translation_model(vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512, max_len=50,
                                              beam_size=5, max_steps=50, length_normalization=0.6, device="cuda:0")
    store = model.tf_manager.sessions[0].store
    store["decoder/state_to_word_b"][2] = -1e9
    ds = synthetic.synthetic_dataset(seed=99, batch=128, src_len=50, tgt_len=50, vocab=32000, with_target=False)
    tfm, runner = model.tf_manager, model.beam_runner
    for _ in range(2):
        tfm.execute(ds, runner.feedables, [runner], compute_losses=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.batches):
        tfm.execute(ds, runner.feedables, [runner], compute_losses=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.batches
    print("beam-5: {:.2f} ms/batch  {:.0f} tok/s".format(dt * 1e3, 128 * 50 / dt))


if __name__ == "__main__":
    main()
