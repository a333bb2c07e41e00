"""Where the host time between two greedy batches goes: timestamps around the chunk launches and flag reads
(monkeypatched ``Session.graphed`` / ``Session.read_small``), relative to the start of ``execute``."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import runtime, synthetic  # noqa: E402

LOG = []


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "greedy"
    model = synthetic.build_translation_model(vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512, max_len=50,
                                              beam_size=5, max_steps=50, length_normalization=0.6, device="cuda:0")
    store = model.tf_manager.sessions[0].store
    synthetic.load_baseline_weights(store)
    store["decoder/state_to_word_b"][2] = -1e9
    sets = [synthetic.synthetic_dataset(seed=99 + i, batch=128, src_len=50, tgt_len=50, vocab=32000,
                                        with_target=False) for i in range(2)]
    tfm = model.tf_manager
    runner = model.beam_runner if mode == "beam" else model.greedy_runner
    run = lambda i: tfm.execute(sets[i % 2], runner.feedables, [runner], compute_losses=False,
                                lookahead=sets[(i + 1) % 2])
    for i in range(5):
        run(i)
    torch.cuda.synchronize()
    g0, r0 = runtime.Session.graphed, runtime.Session.read_small

    def graphed(self, key, fn):
        LOG.append(("launch", time.perf_counter()))
        return g0(self, key, fn)

    def read_small(self, t):
        LOG.append(("read>", time.perf_counter()))
        out = r0(self, t)
        LOG.append(("read<", time.perf_counter()))
        return out
    runtime.Session.graphed, runtime.Session.read_small = graphed, read_small
    marks = []
    for i in range(6):
        LOG.append(("exec>", time.perf_counter()))
        run(i)
        LOG.append(("exec<", time.perf_counter()))
    torch.cuda.synchronize()
    # per batch: exec> -> first launch, last read< -> exec<, read waits
    batches, cur = [], None
    for name, t in LOG:
        if name == "exec>":
            cur = {"t0": t, "ev": []}
        elif name == "exec<":
            cur["t1"] = t
            batches.append(cur)
        else:
            cur["ev"].append((name, t))
    for b in batches[1:]:
        ev = b["ev"]
        first_launch = next(t for n, t in ev if n == "launch")
        last_read = [t for n, t in ev if n == "read<"][-1]
        waits = sum(t2 - t1 for (n1, t1), (n2, t2) in zip(ev, ev[1:]) if n1 == "read>" and n2 == "read<")
        gaps = [t2 - t1 for (n1, t1), (n2, t2) in zip(ev, ev[1:]) if n1 == "read<" and n2 == "launch"]
        print("{}: batch {:.2f} ms | entry -> first launch {:.0f} us | {} flag reads waiting {:.2f} ms | read -> next launch "
              "{:.0f} us each | last read -> return {:.0f} us".format(
                  mode, (b["t1"] - b["t0"]) * 1e3, (first_launch - b["t0"]) * 1e6, len(gaps) + 1, waits * 1e3,
                  1e6 * sum(gaps) / max(len(gaps), 1), (b["t1"] - last_read) * 1e6))


if __name__ == "__main__":
    main()
