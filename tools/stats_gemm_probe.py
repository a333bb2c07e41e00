"""Where does the time of the vocabulary projection of ONE decoding step go (nm_logits_stats_gemm, M = 128)?
Times the launch for several K (slope = cost per k-tile, intercept = fixed prologue / epilogue / launch cost) and
several N (half the tiles in the same time = latency-bound, half the time = throughput-bound).

    python tools/stats_gemm_probe.py            table of microseconds
    python tools/stats_gemm_probe.py pmc        12 launches of the headline shape only (target of rocprofv3 --pmc)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import ops  # noqa: E402


def timed(m, n, k, dev, g, reps=30):
    a = torch.randn(m, k, device=dev, generator=g)
    w = torch.randn(k, n, device=dev, generator=g)
    bias = torch.randn(n, device=dev, generator=g)
    stats = ops.logits_stats_buffer(m, n, dev)
    for _ in range(3):
        ops.logits_stats_gemm(a, w, bias, stats)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.logits_stats_gemm(a, w, bias, stats)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":
        m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
        timed(m, 32000, 512, dev, g, reps=12)
        return
    for m in (128, 640):
        print("M =", m)
        print("  K sweep (N=32000):", "  ".join("K={} {:.1f}us".format(k, timed(m, 32000, k, dev, g))
                                               for k in (64, 128, 256, 512, 1024, 2048)))
        print("  N sweep (K=512):  ", "  ".join("N={} {:.1f}us".format(n, timed(m, n, 512, dev, g))
                                               for n in (4000, 8000, 16000, 32000, 64000)))


if __name__ == "__main__":
    main()
