"""cProfile of the host side of decode batches (greedy / beam) or training steps at the headline shape: where the
Python thread spends its time between kernel launches.   python tools/host_profile.py greedy|beam|train [n]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import synthetic  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "greedy"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    model = synthetic.build_translation_model(vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512, max_len=50,
                                              beam_size=5, max_steps=50, length_normalization=0.6, device="cuda:0")
    store = model.tf_manager.sessions[0].store
    synthetic.load_baseline_weights(store)
    tfm = model.tf_manager
    if mode == "train":
        ds = synthetic.synthetic_dataset(seed=1234, batch=128, src_len=50, tgt_len=50, vocab=32000)
        run = lambda i: tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)
    else:
        store["decoder/state_to_word_b"][2] = -1e9
        sets = [synthetic.synthetic_dataset(seed=99 + i, batch=128, src_len=50, tgt_len=50, vocab=32000,
                                            with_target=False) for i in range(2)]
        runner = model.beam_runner if mode == "beam" else model.greedy_runner
        run = lambda i: tfm.execute(sets[i % 2], runner.feedables, [runner], compute_losses=False,
                                    lookahead=sets[(i + 1) % 2])
    for i in range(5):
        run(i)
    torch.cuda.synchronize()
    prof = cProfile.Profile()
    prof.enable()
    for i in range(n):
        run(i)
    torch.cuda.synchronize()
    prof.disable()
    st = pstats.Stats(prof)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
