"""Sweep the tiled-GEMM configurations (NM_GEMM_CFG) over the large GEMM shapes of one training step.
Each configuration runs in its own process (the knob is read once per process).

    python tools/gemm_sweep.py 1 6 2 7 1+NOSTORE      a column per NM_GEMM_CFG value; +NOSTORE = the timing
                                                       ablation that skips the C stores
A second table times the vocabulary projection with the statistics epilogue (nm_logits_stats_gemm) at one
greedy step (M=128) and one beam step (M=640) for NM_STATS_CFG = 0..3."""
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
    for m in (128, 640):                      # vocabulary projection with the statistics epilogue
        n, k = 32000, 512
        a, bias = rn(m, k), rn(n)
        stats = ops.logits_stats_buffer(m, n, dev)
        out = torch.empty(m, n, device=dev)
        for tb in (False, True):              # W as stored [K,N] / transposed [N,K] (a tile is then contiguous)
            w = rn(n, k) if tb else rn(k, n)
            for with_c in (False, True):
                for _ in range(3):
                    ops.logits_stats_gemm(a, w, bias, stats, out=out if with_c else None, trans_b=tb)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.logits_stats_gemm(a, w, bias, stats, out=out if with_c else None, trans_b=tb)
                e1.record()
                torch.cuda.synchronize()
                res["stats M={} {}{}".format(m, "WT " if tb else "W  ", "C" if with_c else "noC")] = \
                    round(e0.elapsed_time(e1) * 1e3 / 20, 1)   # us
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        cfgs = sys.argv[1:] or ["1", "2", "4", "5", "6"]
        rows = {}
        for cfg in cfgs:
            num = cfg.split("+")[0]
            env = dict(os.environ, NM_GEMM_CFG=num, NM_STATS_CFG=num if int(num) <= 3 else "0")
            if "+NOSTORE" in cfg:
                env["NM_GEMM_NOSTORE"] = "1"
            for part in cfg.split("+")[1:]:
                if part.startswith("SK"):
                    env["NM_GEMM_SK"] = part[2:]
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True,
                                 text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            rows[cfg] = json.loads(line[-1]) if line else {"error": out.stderr[-300:]}
        names = [s[0] for s in SHAPES]
        print("{:16s}".format("TFLOP/s  cfg:") + "".join("{:>11s}".format(c) for c in cfgs))
        for n in names:
            print("{:16s}".format(n) + "".join("{:>11}".format(rows[c].get(n, "-")) for c in cfgs))
        print("us per launch, nm_logits_stats_gemm (NM_STATS_CFG = cfg when cfg <= 3: 0 default, 1 128-wide tiles, "
              "+2 loads two k-tiles ahead)")
        for n in sorted(k for k in rows[cfgs[0]] if k.startswith("stats")):
            print("{:16s}".format(n) + "".join("{:>11}".format(rows[c].get(n, "-")) for c in cfgs))
        for c in cfgs:
            if "error" in rows[c]:
                print(c, rows[c]["error"])
