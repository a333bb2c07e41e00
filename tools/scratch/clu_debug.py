import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from neuralmonkey_amd import ops
dev = torch.device("cuda:0")
rows, steps, h, ndir = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 2, 512, 1
g = torch.Generator(device=dev).manual_seed(0)
xp = torch.randn(rows * steps, 3 * h, device=dev, generator=g)
wgh = torch.randn(1, h, 2 * h, device=dev, generator=g) * 0.05
wch = torch.randn(1, h, h, device=dev, generator=g) * 0.05
hcur = torch.randn(1, rows, h, device=dev, generator=g)
out = torch.zeros(rows, steps, h, device=dev)
ru = torch.zeros(steps, 1, rows, 2 * h, device=dev); c = torch.zeros(steps, 1, rows, h, device=dev)
ws = ops.gru_seq_workspace(rows, h, ndir, dev)
ops.gru_seq_fwd(steps, ndir, rows, h, xp, (3 * h, steps * 3 * h, 3 * h), hcur, hcur, 0, ru[0], rows * 2 * h, None, 0, c[0], rows * h, wgh, wch, ws, out=out, out_strides=(h, steps * h, h))
torch.cuda.synchronize()
print("failed", ops.gru_seq_failed(ws))
w = ws.view(torch.int32).cpu().numpy()
print("err word", w[0])
gran = w[64:].reshape(-1, 2)
n = 32 * h   # granules per stage (padded rows 32)
xa = gran[:n]; xb = gran[n:2 * n]
print("XA tags:", np.unique(xa[:, 1], return_counts=True))
print("XB tags:", np.unique(xb[:, 1], return_counts=True))
print("XA first rows of tag!=1:", np.nonzero(xa[:16 * h, 1] != 1)[0][:20])
