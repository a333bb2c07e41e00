"""One large GEMM shape, repeated -- the target of PMC passes (MFMA busy cycles, wait buckets):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES ... -- python tools/gemm_one.py NN"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import ops  # noqa: E402

SHAPES = {"NN": (6400, 32000, 512, False, False), "NT": (6400, 512, 32000, False, True),
          "TN": (512, 32000, 6400, True, False)}


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "NN"
    m, n, k, ta, tb = SHAPES[kind]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    a = rn(k, m) if ta else rn(m, k)
    b = rn(n, k) if tb else rn(k, n)
    c = torch.empty(m, n, device=dev)
    for _ in range(12):
        ops.gemm(a, b, out=c, trans_a=ta, trans_b=tb)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
