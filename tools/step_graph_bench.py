"""Effect of HIP-graph replay on the taped (general) path: training steps and greedy decoding of the
tests/small.ini-shaped model (NematusGRU + conditional GRU) and of a small Transformer, graphs on / off."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import test_general_gpu as TG  # noqa: E402
from tests import test_transformer_gpu as TT  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run(build, data, parts_of, graphs):
    m = build()
    sess = m["tfm"].sessions[0]
    sess.use_step_graphs = graphs
    sess.use_graphs = graphs
    ds = data(64, 20, 20, 24, seed=1)[0]
    train = timed(lambda: m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True), 3, 20)
    dec = m["dec"]
    # untrained models stop at once: push </s> down so that all 24 steps run
    bias = "decoder/state_to_word_b"
    if bias in sess.store:
        sess.store[bias][2] = -1e9
    dsd = data(64, 20, 20, 24, seed=2, with_target=False)[0]
    fd = {}
    for part in parts_of(m):
        fd.update(part.feed_dict(dsd, train=False))
    steps = sess.run({"sym": dec.decoded_symbols}, fd)["sym"].shape[0]
    greedy = timed(lambda: sess.run({"sym": dec.decoded_symbols}, fd), 2, 10)
    return train, greedy, steps


cfg, es, et = TG.CASES["small_ini"]
for g in (False, True):
    t, d, n = run(lambda: TG._build(dev, cfg, es, et, max_len=24), TG._data,
                  lambda m: (m["enc"].input_sequence, m["enc"], m["att"], m["dec"]), g)
    print("small_ini    graphs {!s:5}  train {:6.2f} ms/step   greedy {:6.2f} ms / {} steps".format(g, t, d, n))
c2, d_model, ff = TT.CASES["wide"]
for g in (False, True):
    t, d, n = run(lambda: TT._build(dev, c2, d_model, ff, max_len=24), TT._data,
                  lambda m: (m["enc"].input_sequence, m["enc"], m["dec"]), g)
    print("transformer  graphs {!s:5}  train {:6.2f} ms/step   greedy {:6.2f} ms / {} steps".format(g, t, d, n))
