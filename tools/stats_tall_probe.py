"""The statistics GEMM of one beam step (640 x 32000 x 512, logits stored): time and distance from torch.addmm.
(Round 4 used it to compare the 640x128-tile instance, since removed: 314 us against 228 us.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for m in (640, 600):
    a = torch.randn(m, 512, device=dev, generator=g)
    w = torch.randn(512, 32000, device=dev, generator=g)
    bias = torch.randn(32000, device=dev, generator=g)
    stats = ops.logits_stats_buffer(m, 32000, dev)
    out = torch.empty(m, 32000, device=dev)
    for _ in range(3):
        ops.logits_stats_gemm(a, w, bias, stats, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.logits_stats_gemm(a, w, bias, stats, out)
    e1.record()
    torch.cuda.synchronize()
    ref = torch.addmm(bias, a, w)
    print("M={}: {:.1f} us, max |logit - torch| {:.2e}".format(
        m, e0.elapsed_time(e1) * 1e3 / 30, float((out - ref).abs().max())))
