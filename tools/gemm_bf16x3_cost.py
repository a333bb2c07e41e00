"""What a split-bf16 emulation of the logits-sized fp32 products would cost and buy (cost, do not
ship).  For the two "NT" shapes of the vocabulary projection -- states . E^T (tied embeddings: M=6400, N=32000,
K=512) and dlogits . W^T (M=6400, N=512, K=32000) -- and the beam / greedy step rows (M=640 / 128): time of the exact
fp32 product (nm_gemm_f32, v_mfma_f32_32x32x2_f32), of the three-term bf16 emulation (nm_gemm_bf16x3_nt) and of a plain
bf16 product (one term), with each result's error against float64, relative to the largest |entry| of the exact
result -- the measure the parity tests use (1e-4).

    python tools/gemm_bf16x3_cost.py > profiles/r04_gemm_bf16x3_cost.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import _lib, ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3         # us


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1)
    shapes = [("states . E^T (tied projection), training rows", 6400, 32000, 512),
              ("states . E^T, beam step rows", 640, 32000, 512),
              ("states . E^T, greedy step rows", 128, 32000, 512),
              ("dlogits . W^T", 6400, 512, 32000)]
    print("{:<48} {:>9} {:>9} {:>9} | {:>9} {:>9} {:>9}".format("C[M,N] = A[M,K] . B[N,K]^T", "fp32 us", "bf16x3 us",
                                                                   "bf16 us", "fp32 err", "bf16x3 err", "bf16 err"))
    for name, m, n, k in shapes:
        # activations ~ N(0,1) against weights ~ N(0, 0.05); dlogits are softmax gradients
        a = torch.randn(m, k, device=dev, generator=gen)
        b = torch.randn(n, k, device=dev, generator=gen) * 0.05
        if k == 32000:
            a = torch.softmax(a, -1) - torch.nn.functional.one_hot(torch.randint(0, k, (m,), device=dev), k)
        c32, c3, c1 = (torch.empty(m, n, device=dev) for _ in range(3))
        f32 = lambda: ops.gemm(a, b, out=c32, trans_b=True)
        call = lambda out, terms, var=0: _lib.check(
            lib.nm_gemm_bf16x3_nt(ops._stream(), m, n, k, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                  out.data_ptr(), out.stride(0), terms, var), "nm_gemm_bf16x3_nt")
        variants = {v: timed(lambda v=v: call(c3, 3, v)) for v in range(4)}
        best = min(variants, key=variants.get)
        t32, t3, t1 = timed(f32), variants[best], timed(lambda: call(c1, 1, best))
        call(c3, 3, best)
        rows = slice(0, min(m, 256))                 # float64 check on a slab of rows
        exact = a[rows].double() @ b.double().t()
        scale = float(exact.abs().max())
        err = lambda c: float((c[rows].double() - exact).abs().max()) / scale
        flops = 2.0 * m * n * k
        print("{:<48} {:>9.1f} {:>9.1f} {:>9.1f} | {:>9.2e} {:>9.2e} {:>9.2e}   M={} N={} K={}: {:.0f} / {:.0f} / {:.0f} "
              "TFLOP/s-equivalent; bf16x3 by variant (128x128 BK16 / BK32, 256x128 BK16 / BK32): {}".format(
                  name, t32, t3, t1, err(c32), err(c3), err(c1), m, n, k, flops / t32 / 1e6, flops / t3 / 1e6,
                  flops / t1 / 1e6, " / ".join("%.0f" % variants[v] for v in range(4))))


if __name__ == "__main__":
    main()
