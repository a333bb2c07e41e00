"""Runs only the fused attention step at the benchmark shape (for PMC / kernel-trace passes:
rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/attn_only.py [queries_per_key] [iters] [cold|warm]).
"cold" (default): a 1 GiB read sweep between launches evicts hf / states from L2 and the 256 MB Infinity
Cache; "dirty": a 1 GiB write sweep instead (the launch also pays for the write-back of dirty lines)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralmonkey_amd import ops  # noqa: E402

B, S = int(os.environ.get("NM_B", "128")), int(os.environ.get("NM_S", "50"))
A, C = int(os.environ.get("NM_A", "1024")), int(os.environ.get("NM_C", "1024"))     # captioning: NM_S=64 NM_A=512 NM_C=2048
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
import ctypes  # noqa: E402

from neuralmonkey_amd import _lib  # noqa: E402

lib = _lib.load()


def run(n):
    for i in range(n):
        if mode == "dirty":
            flush.fill_(float(i))
        elif mode == "cold":
            sink.add_(flush.sum())
        ops.attn_fwd(y, hf, st, mask, v, bias, qpk, ctx, w, ws)
    torch.cuda.synchronize()


run(3)
lib.nm_prof_enable(None, 1)
run(iters)
lib.nm_prof_enable(None, 0)
tot, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
lib.nm_prof_attn_step(None, ctypes.byref(tot), ctypes.byref(cnt))
# reference of the same step in float64 (feed_forward.py:120-166)
yd, hfd, std = y.double(), hf.double(), st.double()
e = (v.double() * torch.tanh(hfd.repeat_interleave(qpk, 0) + yd[:, None, :])).sum(-1)
wr = torch.softmax(e, -1) * mask.double().repeat_interleave(qpk, 0)
wr = wr / (wr.sum(1, keepdim=True) + 1e-8)
cr = (wr[:, :, None] * std.repeat_interleave(qpk, 0)).sum(1)
err = float((ctx.double() - cr).abs().max() / cr.abs().max())
print("NM_ATTN_MAXROWS={} whole={} B={} S={} mode={} qpk={}: {:.2f} us/launch (HIP events, {} launches)  rel err {:.1e}".format(
    os.environ.get("NM_ATTN_MAXROWS", "-"), os.environ.get("NM_ATTN_WHOLE", "-"), B, S, mode, qpk, tot.value * 1e3 / max(cnt.value, 1), cnt.value, err))
