"""The decoding steps' vocabulary projection: the activation-stationary kernel (csrc/nm_proj.hip) against gemm_tiled's
statistics epilogue -- error against float64, the statistics against the kernel's own logits, microseconds per launch
(logits stored / statistics only).

    python tools/proj_astat_probe.py            both kernels (the switch is read once: one subprocess each)
    python tools/proj_astat_probe.py one        this process only (NM_PROJ_ASTAT / NM_PROJ_ASTAT_CK from the environment)"""
import json
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import _lib, ops  # noqa: E402


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def check(m, n, k, dev):
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) * 0.3).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    bias[3 % n] = -1e9
    ad, wd, bd = (torch.tensor(x, device=dev) for x in (a, w, bias))
    stats = ops.logits_stats_buffer(m, n, dev)
    got = torch.full((m, n), float("nan"), device=dev)
    ops.logits_stats_gemm(ad, wd, bd, stats, out=got)
    stats2 = ops.logits_stats_buffer(m, n, dev)
    ops.logits_stats_gemm(ad, wd, bd, stats2, out=None)
    x = got.cpu().numpy()
    ref = a.astype(np.float64) @ w.astype(np.float64) + bias
    finite = ref > -1e8
    err = float(np.abs(x - ref)[finite].max() / np.abs(ref[finite]).max())
    tile = int(_lib.load().nm_logits_stats_tile(m))
    nt = (n + tile - 1) // tile
    st = stats.cpu().numpy().reshape(m, nt, 4)
    pad = np.full((m, nt * tile), -np.inf, np.float32)
    pad[:, :n] = x
    tiles = pad.reshape(m, nt, tile)
    ok_max = bool(np.array_equal(st[:, :, 0], tiles.max(2)))
    arg = st[:, :, 2].copy().view(np.int32)
    ok_arg = bool(np.array_equal(arg, tiles.argmax(2) + np.arange(nt)[None, :] * tile))
    t64 = tiles.astype(np.float64)
    sums = np.exp(t64 - t64.max(2, keepdims=True)).sum(2)
    err_sum = float(np.abs(st[:, :, 1] - sums).max() / sums.max())
    same = bool(torch.equal(stats, stats2))
    t_c = timed(lambda: ops.logits_stats_gemm(ad, wd, bd, stats, out=got))
    t_s = timed(lambda: ops.logits_stats_gemm(ad, wd, bd, stats, out=None))
    return {"m": m, "n": n, "k": k, "err_vs_f64": err, "nan": bool(np.isnan(x).any()), "tile_max_exact": ok_max,
            "tile_argmax_exact": ok_arg, "sumexp_rel_err": err_sum, "stats_same_without_logits": same,
            "us_logits_stored": t_c, "us_stats_only": t_s}


def stamps():
    """Per-chunk clock stamps (100 MHz wall clock) of workgroups 0 and 101 of one launch."""
    dev = torch.device("cuda:0")
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    for m in (128, 640):
        n, k = 32000, 512
        g = torch.Generator(device=dev).manual_seed(0)
        a = torch.randn(m, k, device=dev, generator=g)
        w = torch.randn(k, n, device=dev, generator=g)
        bias = torch.randn(n, device=dev, generator=g)
        stats = ops.logits_stats_buffer(m, n, dev)
        for _ in range(3):
            ops.logits_stats_gemm(a, w, bias, stats)
        torch.cuda.synchronize()
        os.environ["NM_PROJ_ASTAT_DBG_PTR"] = str(buf.data_ptr())
        buf.zero_()
        ops.logits_stats_gemm(a, w, bias, stats)
        torch.cuda.synchronize()
        del os.environ["NM_PROJ_ASTAT_DBG_PTR"]
        t = buf.cpu().numpy()
        for wg, off in ((0, 0), (101, 64)):
            nchunk = int(t[off + 62])
            t0 = t[off + 61]
            marks = [(t[off + i] - t0) / 100.0 for i in range(min(nchunk, 60))]
            wall_us = (t[off + 60] - t0) / 100.0
            print("M={} wg {:3d}: chunks {} end {:.2f} us, shader clock {:.0f} MHz; chunk starts (us): {}".format(
                m, wg, nchunk, wall_us, t[off + 63] / max(wall_us, 1e-9),
                " ".join("{:.2f}".format(x) for x in marks)), flush=True)


def one():
    dev = torch.device("cuda:0")
    shapes = [(128, 32000, 512), (640, 32000, 512), (5, 32000, 512), (37, 4104, 128), (130, 1000, 256), (128, 32000, 384),
              (640, 32004, 512), (16, 64, 128)]
    for m, n, k in shapes[:int(os.environ.get("NM_PROBE_SHAPES", "99"))]:
        print(json.dumps(check(m, n, k, dev)), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "stamps":
        stamps()
        return
    variants = [("gemm_tiled", {"NM_PROJ_ASTAT": "0"}), ("astat", {"NM_PROJ_ASTAT": "1"})]
    if "ablate" in sys.argv:          # timing only (results are wrong by construction): where does the time go?
        def ab(bits, **more):
            env = {"NM_PROJ_ASTAT_ABLATE": str(bits), "NM_PROBE_SHAPES": "2"}
            env.update(more)
            return env
        variants += [("astat_no_weight_stream", ab(1)), ("astat_no_statistics", ab(16)), ("astat_no_sum_exp", ab(32))]
    for tag, env in variants:
        print("==", tag, flush=True)
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=e, timeout=600)


if __name__ == "__main__":
    main()
