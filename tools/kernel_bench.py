"""Micro-benchmarks of the libnmhip kernels at the BASELINE.md shapes.
Prints one JSON object per kernel (time, algorithmic GB/s or TFLOP/s)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralmonkey_amd import ops  # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    dev = torch.device("cuda:0")
    out = []
    B, S, A, C, H, E, V = 128, 50, 1024, 1024, 512, 512, 32000
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)

    # --- attention step
    for qpk in (1, 5):
        r = B * qpk
        y, hf, st = rn(r, A), rn(B, S, A), rn(B, S, C)
        mask = torch.ones(B, S, device=dev)
        v, bias = rn(A), rn(1)
        ctx, w = torch.empty(r, C, device=dev), torch.empty(r, S, device=dev)
        ws = ops.attn_workspace(r, S, C, dev)
        t = timeit(lambda: ops.attn_fwd(y, hf, st, mask, v, bias, qpk, ctx, w, ws))
        nbytes = 4 * (B * S * A + B * S * C + B * S + r * A + r * S + r * C)
        out.append({"kernel": f"attn_fwd qpk={qpk}", "us": t * 1e6, "alg_GBps": nbytes / t / 1e9,
                    "frac_8TBps": nbytes / t / 8e12})

    # --- GEMMs
    shapes = [("step h.Wg_h", 128, 1024, 512, False, False, 0),
              ("step rh.Wc_h", 128, 512, 512, False, False, 0),
              ("step s.Wq", 128, 1024, 512, False, False, 0),
              ("step bwd dG.WgT", 128, 512, 1024, False, True, 0),
              ("step tiled64", 128, 1024, 512, False, False, 2),
              ("beam step h.Wg_h", 640, 1024, 512, False, False, 0),
              ("greedy logits", 128, 32000, 512, False, False, 0),
              ("beam logits", 640, 32000, 512, False, False, 0),
              ("train logits", 6400, 32000, 512, False, False, 0),
              ("train dlogits.WT", 6400, 512, 32000, False, True, 0),
              ("train OT.dlogits", 512, 32000, 6400, True, False, 0),
              ("keys", 6400, 1024, 1024, False, False, 0),
              ("keys 128tile", 6400, 1024, 1024, False, False, 1),
              ("outproj", 6400, 512, 2048, False, False, 0),
              ("enc xproj", 6400, 3072, 512, False, False, 0),
              ("wgrad embT.dxp", 512, 1024, 6400, True, False, 0),
              ("wgrad sT.dy", 512, 1024, 6400, True, False, 0),
              ("wgrad statesT.dhf", 1024, 1024, 6400, True, False, 0),
              ("dx = dxp.WT", 6400, 512, 1024, False, True, 0)]
    for name, m, n, k, ta, tb, algo in shapes:
        a = rn(k, m) if ta else rn(m, k)
        b = rn(n, k) if tb else rn(k, n)
        c = torch.empty(m, n, device=dev)
        t = timeit(lambda: ops.gemm(a, b, out=c, trans_a=ta, trans_b=tb, algo=algo), iters=20)
        out.append({"kernel": f"gemm {name} {m}x{n}x{k}", "us": t * 1e6, "TFLOPs": 2.0 * m * n * k / t / 1e12})

    # --- vocabulary row kernels
    for rows in (128, 640):
        x = rn(rows, V)
        mx, lse = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        am = torch.empty(rows, dtype=torch.int32, device=dev)
        t = timeit(lambda: ops.row_stats(x, mx, lse, am))
        out.append({"kernel": f"row_stats {rows}x{V}", "us": t * 1e6, "alg_GBps": 4.0 * rows * V / t / 1e9})
    k = 5
    x = rn(B * k, V)
    mx, lse = torch.empty(B * k, device=dev), torch.empty(B * k, device=dev)
    ops.row_stats(x, mx, lse, None)
    i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
    lps, lens, fin = torch.zeros(B, k, device=dev), i32(B, k), i32(B, k)
    pen = ops.length_penalty_table(64, 0.6, dev)
    o = [torch.empty(B, k, device=dev), i32(B, k), i32(B, k), torch.empty(B, k, device=dev), i32(B, k),
         i32(B, k), i32(B, k)]
    ws = ops.beam_workspace(B, k, V, dev)
    t = timeit(lambda: ops.beam_topk_step(x, B, k, mx, lse, lps, lens, fin, pen, 2, *o, ws))
    out.append({"kernel": "beam_topk_step 128x5x32000", "us": t * 1e6, "alg_GBps": 4.0 * B * k * V / t / 1e9})
    xl = rn(6400, V)
    tg, wt, ls = i32(6400), torch.ones(6400, device=dev), torch.empty(6400, device=dev)
    t = timeit(lambda: ops.xent(xl, tg, wt, ls), iters=10)
    out.append({"kernel": "xent fwd 6400x32000", "us": t * 1e6, "alg_GBps": 4.0 * 6400 * V / t / 1e9})
    for o_ in out:
        print(json.dumps(o_))


if __name__ == "__main__":
    main()
