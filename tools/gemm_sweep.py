"""Sweep the tiled-GEMM configurations (NM_GEMM_CFG) over the large GEMM shapes of one training step.
Each configuration runs in its own process (the knob is read once per process)."""
import json
import os
import subprocess
import sys

SHAPES = [("beam logits NN", 640, 32000, 512, False, False),
          ("greedy logit NN", 128, 32000, 512, False, False),
          ("beam outproj NN", 640, 512, 2048, False, False),
          ("logits fwd NN", 6400, 32000, 512, False, False),
          ("dlogits.WT NT", 6400, 512, 32000, False, True),
          ("OT.dlogits TN", 512, 32000, 6400, True, False),
          ("keys NN", 6400, 1024, 1024, False, False),
          ("outproj NN", 6400, 512, 2048, False, False),
          ("enc xproj NN", 6400, 3072, 512, False, False),
          ("wgrad TN 1024", 1024, 1024, 6400, True, False),
          ("dx NT", 6400, 512, 1024, False, True),
          ("dstates NT", 6400, 1024, 1024, False, True)]


def child():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from neuralmonkey_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    res = {}
    for name, m, n, k, ta, tb in SHAPES:
        a = rn(k, m) if ta else rn(m, k)
        b = rn(n, k) if tb else rn(k, n)
        c = torch.empty(m, n, device=dev)
        for _ in range(3):
            ops.gemm(a, b, out=c, trans_a=ta, trans_b=tb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, b, out=c, trans_a=ta, trans_b=tb)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / 10
        res[name] = round(2.0 * m * n * k / t / 1e12, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        cfgs = sys.argv[1:] or ["1", "2", "4", "5", "6"]
        rows = {}
        for cfg in cfgs:
            env = dict(os.environ, NM_GEMM_CFG=cfg)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True,
                                 text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            rows[cfg] = json.loads(line[-1]) if line else {"error": out.stderr[-300:]}
        names = [s[0] for s in SHAPES]
        print("{:16s}".format("TFLOP/s  cfg:") + "".join("{:>8s}".format(c) for c in cfgs))
        for n in names:
            print("{:16s}".format(n) + "".join("{:>8}".format(rows[c].get(n, "-")) for c in cfgs))
