"""Greedy or beam-5 decode batches only, headline shape (B=128, 50 steps, H=512, V=32000, BASELINE.md weights;
</s> unreachable so every sentence runs all steps) -- the workload of bench.py's decode legs, for per-kernel
profiles:  rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/decode_profile.py --mode greedy
           python tools/trace_window.py DIR/*/*_kernel_trace.csv 0.4"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["greedy", "beam"], default="greedy")
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--batch", type=int, default=128)
    args = ap.parse_args()
    from neuralmonkey_amd import synthetic
    model = synthetic.build_translation_model(vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512, max_len=50,
                                              beam_size=5, max_steps=50, length_normalization=0.6, device="cuda:0")
    store = model.tf_manager.sessions[0].store
    synthetic.load_baseline_weights(store)
    store["decoder/state_to_word_b"][2] = -1e9
    sets = [synthetic.synthetic_dataset(seed=99 + i, batch=args.batch, src_len=50, tgt_len=50, vocab=32000,
                                        with_target=False) for i in range(2)]
    tfm = model.tf_manager
    runner = model.beam_runner if args.mode == "beam" else model.greedy_runner
    for i in range(5):           # eager + capture passes in both buffer slots (look-ahead alternates them)
        tfm.execute(sets[i % 2], runner.feedables, [runner], compute_losses=False,
                    lookahead=None if os.environ.get("NM_NO_LOOKAHEAD") else sets[(i + 1) % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ahead = not os.environ.get("NM_NO_LOOKAHEAD")
    for i in range(args.batches):
        nxt = sets[(i + 1) % 2] if (ahead and i + 1 < args.batches) else None
        tfm.execute(sets[i % 2], runner.feedables, [runner], compute_losses=False, lookahead=nxt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.batches
    print("{}: {:.2f} ms/batch  {:.1f} us/step  {:.0f} tok/s".format(args.mode, dt * 1e3, dt * 1e6 / 50,
                                                                      args.batch * 50 / dt))


if __name__ == "__main__":
    main()
