"""Where the time of a decoding batch goes, host side and GPU side, without a profiler attached (rocprofv3 slows
HIP-graph launches by milliseconds): ``Session.decode_chunks`` (the loop of step chunks) is wrapped with a HIP event
on either side and host time stamps; flag read-backs (``HostPending.get``) are timed on the host.

    python tools/batch_boundary_probe.py greedy|beam [batches=8]

Per batch: wall time of ``execute``; host time from entry to the loop, inside the loop, from the loop to the return;
GPU time of the loop itself (event to event) and GPU time between the end of the previous batch's loop and the start
of this one (what of the encoder / set-up / result hand-over is NOT hidden under decoding); host time spent waiting
for flag read-backs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import runtime, synthetic  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "greedy"
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    if os.environ.get("NM_MAIN_PRIO"):        # experiment: the decoding stream above the look-ahead stream
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    model = synthetic.build_translation_model(vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512, max_len=50,
                                              beam_size=5, max_steps=50, length_normalization=0.6, device="cuda:0")
    store = model.tf_manager.sessions[0].store
    synthetic.load_baseline_weights(store)
    store["decoder/state_to_word_b"][2] = -1e9
    sets = [synthetic.synthetic_dataset(seed=99 + i, batch=128, src_len=50, tgt_len=50, vocab=32000,
                                        with_target=False) for i in range(2)]
    tfm = model.tf_manager
    runner = model.beam_runner if mode == "beam" else model.greedy_runner
    ahead = not os.environ.get("NM_NO_LOOKAHEAD")
    run = lambda i: tfm.execute(sets[i % 2], runner.feedables, [runner], compute_losses=False,
                                lookahead=sets[(i + 1) % 2] if ahead else None)
    for i in range(5):
        run(i)
    torch.cuda.synchronize()

    log = []
    cur = {}
    d0, g0 = runtime.Session.decode_chunks, runtime.HostPending.get

    def decode_chunks(self, *a, **k):
        cur["loop_in"] = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = d0(self, *a, **k)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        cur["loop_out"] = time.perf_counter()
        cur["ev"] = (e0, e1)
        return out

    def get(self):
        t = time.perf_counter()
        out = g0(self)
        cur["wait"] = cur.get("wait", 0.0) + time.perf_counter() - t
        cur["reads"] = cur.get("reads", 0) + 1
        return out
    runtime.Session.decode_chunks, runtime.HostPending.get = decode_chunks, get
    for i in range(batches):
        cur = {"t0": time.perf_counter()}
        run(i)
        cur["t1"] = time.perf_counter()
        log.append(cur)
    torch.cuda.synchronize()
    prev_end = None
    for b in log:
        e0, e1 = b["ev"]
        loop_gpu = e0.elapsed_time(e1)
        between = prev_end.elapsed_time(e0) if prev_end is not None else float("nan")
        prev_end = e1
        print("{}: batch {:.2f} ms | host: entry->loop {:.0f} us, in loop {:.2f} ms ({} flag reads waiting {:.2f} ms), "
              "loop->return {:.0f} us | GPU: loop {:.2f} ms, previous loop end -> this loop start {:.2f} ms".format(
                  mode, (b["t1"] - b["t0"]) * 1e3, (b["loop_in"] - b["t0"]) * 1e6,
                  (b["loop_out"] - b["loop_in"]) * 1e3, b.get("reads", 0), b.get("wait", 0.0) * 1e3,
                  (b["t1"] - b["loop_out"]) * 1e6, loop_gpu, between))


if __name__ == "__main__":
    main()
