"""The mid-sized products of a Transformer-base training step (6400 rows, 512 / 2048 wide) under the tile of
NM_GEMM_CFG64 (read when the library's context is created: one process per setting):
    for c in 0 1 2 3; do NM_GEMM_CFG64=$c python tools/gemm_mid_sweep.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import ops  # noqa: E402

SHAPES = [("x.W 512x512", 6400, 512, 512, False, False), ("dy.W^T 512x512", 6400, 512, 512, False, True),
          ("ff2 x.W 2048->512", 6400, 512, 2048, False, False), ("ff1 dy.W^T 2048->512", 6400, 512, 2048, False, True),
          ("x.W 512->2048", 6400, 2048, 512, False, False), ("logits-sized row block 640x32000", 640, 32000, 512, False, False)]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    print("NM_GEMM_CFG64 =", os.environ.get("NM_GEMM_CFG64", "0"))
    for name, m, n, k, ta, tb in SHAPES:
        a = rn(k, m) if ta else rn(m, k)
        b = rn(n, k) if tb else rn(k, n)
        c = torch.empty(m, n, device=dev)
        for _ in range(5):
            ops.gemm(a, b, out=c, trans_a=ta, trans_b=tb)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(50):
            ops.gemm(a, b, out=c, trans_a=ta, trans_b=tb)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1000 / 50
        print("  {:36s} {:8.1f} us  {:6.1f} TFLOP/s".format(name, us, 2.0 * m * n * k / us / 1e6))


if __name__ == "__main__":
    main()
