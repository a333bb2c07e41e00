"""Where a recurrent step of the cluster forward loop spends its time (in-kernel 100 MHz clock stamps of one workgroup,
steps 20..23; needs a build with NM_HIPCC_FLAGS=-DNM_CLU_DEBUG):

    NM_HIPCC_FLAGS=-DNM_CLU_DEBUG python -m neuralmonkey_amd.build --force && python tools/clu_stamps.py 128 1
"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
dev = torch.device("cuda:0")
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
os.environ["NM_CLU_DEBUG_PTR"] = str(dbg.data_ptr())
from neuralmonkey_amd import ops
rows, steps, h = int(sys.argv[1]), 50, 512
ndir = int(sys.argv[2])
g = torch.Generator(device=dev).manual_seed(0)
xp = torch.randn(rows * steps, ndir * 3 * h, device=dev, generator=g)
wgh = torch.randn(ndir, h, 2 * h, device=dev, generator=g) * 0.05
wch = torch.randn(ndir, h, h, device=dev, generator=g) * 0.05
hcur = torch.randn(ndir, rows, h, device=dev, generator=g)
out = torch.zeros(rows, steps, ndir * h, device=dev)
ru = torch.zeros(steps, ndir, rows, 2 * h, device=dev); c = torch.zeros(steps, ndir, rows, h, device=dev)
ws = ops.gru_seq_workspace(rows, h, ndir, dev)
for rep in range(3):
    ops.gru_seq_fwd(steps, ndir, rows, h, xp, (3 * h, steps * ndir * 3 * h, ndir * 3 * h), hcur, hcur, 0, ru[0], ndir * rows * 2 * h, None, 0, c[0], ndir * rows * h, wgh, wch, ws, out=out, out_strides=(h, steps * ndir * h, ndir * h))
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(4, 2, 8)
print("rows", rows, "ndir", ndir, "failed", ops.gru_seq_failed(ws))
names = ["top", "sweepA", "mmaA", "barA", "epiA", "sweepB", "barB(mma+bar)"]
for t in range(4):
    for w in range(2):
        x = d[t, w]
        print("step", 20 + t, "wave", "last" if w else 0, " ".join("{}={:.2f}".format(n, (x[i + 1] - x[i]) / 100.0) for i, n in enumerate(names[1:])),
              "| step total {:.2f} us".format((d[t + 1, w, 0] - x[0]) / 100.0 if t < 3 else float("nan")))
