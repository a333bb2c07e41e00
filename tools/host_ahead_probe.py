"""Does the host thread run ahead of the GPU in the training loop?  Prints the host time of each
``execute`` call (no synchronisation in between) next to the wall time per step, with torch's
synchronisation warnings switched on."""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import synthetic  # noqa: E402


def main():
    model = synthetic.build_translation_model(vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512, max_len=50,
                                              beam_size=0, device="cuda:0")
    synthetic.load_baseline_weights(model.tf_manager.sessions[0].store)
    ds = synthetic.synthetic_dataset(seed=1234, batch=128, src_len=50, tgt_len=50, vocab=32000)
    tfm, trainer = model.tf_manager, model.trainer
    for _ in range(3):
        tfm.execute(ds, trainer.feedables, [trainer], train=True)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode(1)
    warnings.simplefilter("always")
    host = []
    t0 = time.perf_counter()
    for _ in range(10):
        t = time.perf_counter()
        tfm.execute(ds, trainer.feedables, [trainer], train=True)
        host.append((time.perf_counter() - t) * 1e3)
    t_host = time.perf_counter() - t0
    torch.cuda.set_sync_debug_mode(0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("host ms per execute:", " ".join("%.2f" % h for h in host))
    print("host total %.2f ms, wall %.2f ms for 10 steps" % (t_host * 1e3, wall * 1e3))


if __name__ == "__main__":
    main()
