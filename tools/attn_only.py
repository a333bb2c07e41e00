"""Runs only the fused attention step at the benchmark shape (for PMC / kernel-trace passes:
rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/attn_only.py [queries_per_key] [iters] [cold|warm]).
"cold" (default): a 1 GiB read sweep between launches evicts hf / states from L2 and the 256 MB Infinity
Cache; "dirty": a 1 GiB write sweep instead (the launch also pays for the write-back of dirty lines)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralmonkey_amd import ops  # noqa: E402

B, S, A, C = 128, 50, 1024, 1024
qpk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "cold"      # cold (read sweep) | dirty (write sweep) | warm
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
r = B * qpk
y = torch.randn(r, A, device=dev, generator=g)
hf = torch.randn(B, S, A, device=dev, generator=g)
st = torch.randn(B, S, C, device=dev, generator=g)
mask = torch.ones(B, S, device=dev)
v = torch.randn(A, device=dev, generator=g)
bias = torch.zeros(1, device=dev)
ctx = torch.empty(r, C, device=dev)
w = torch.empty(r, S, device=dev)
ws = ops.attn_workspace(r, S, C, dev)
# a 1 GiB sweep between launches evicts hf/states from the 256 MB Infinity Cache so the
# counters see HBM traffic
flush = torch.zeros(256 << 20, device=dev)
sink = torch.zeros(1, device=dev)
for i in range(iters):
    if mode == "dirty":
        flush.fill_(float(i))
    elif mode == "cold":
        sink.add_(flush.sum())
    ops.attn_fwd(y, hf, st, mask, v, bias, qpk, ctx, w, ws)
torch.cuda.synchronize()
print("done", float(ctx.sum()))
