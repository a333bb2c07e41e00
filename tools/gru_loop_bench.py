"""Microseconds per step of the recurrent time loops (forward: gates + candidate launches; BPTT: two launches),
HIP-graph replayed as in training: R rows, H = 512, one or two directions, with / without length masking.

    python tools/gru_loop_bench.py [rows=128]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import ops  # noqa: E402
from neuralmonkey_amd.nn import gru  # noqa: E402


def bench(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    h, s = 512, 50
    for ndir, masked in ((1, False), (2, True)):
        rn = lambda *shape: torch.randn(*shape, device=dev, generator=g) * 0.1
        xp = rn(rows * s, ndir * 3 * h)
        wgh, wch = rn(ndir, h, 2 * h), rn(ndir, h, h)
        wg_t, wc_t = wgh.transpose(1, 2).contiguous(), wch.transpose(1, 2).contiguous()
        lengths = torch.tensor(np.random.default_rng(0).integers(25, s + 1, size=rows), dtype=torch.int32,
                               device=dev) if masked else None
        hcur = torch.zeros(ndir, rows, h, device=dev)
        out = torch.zeros(rows, s, ndir * h, device=dev)
        ru_all, c_all = torch.empty(s, ndir, rows, 2 * h, device=dev), torch.empty(s, ndir, rows, h, device=dev)
        rh = torch.empty(ndir, rows, h, device=dev)
        xrs, xts, ors, ots = s * ndir * 3 * h, ndir * 3 * h, s * ndir * h, ndir * h

        def fwd(transposed):
            hcur.zero_()
            for t in range(s):
                gru.step_fwd(xp, (3 * h, xrs, xts), hcur, hcur, wg_t if transposed else wgh,
                             wc_t if transposed else wch, ru_all[t], rh, c_all[t], out, (h, ors, ots), lengths, t,
                             ndir, rows, h, False, None, None, transposed=transposed)
        dh = torch.zeros(ndir, rows, h, device=dev)
        d_out = rn(rows, s, ndir * h)
        dxp = torch.zeros(rows * s, ndir * 3 * h, device=dev)
        dgpre, dcpre, drh = (torch.empty(2, ndir, rows, 2 * h, device=dev), torch.empty(ndir, rows, h, device=dev),
                             torch.empty(ndir, rows, h, device=dev))
        seq = (h, s * ndir * h, ndir * h)

        def bwd():
            dh.zero_()
            gru.bptt(s, dh, d_out, seq, ru_all, c_all, None, out, seq, dxp, (3 * h, xrs, xts), wgh, wch, lengths, ndir,
                     rows, h, False, dgpre, dcpre, drh)
        ws = ops.gru_seq_workspace(rows, h, ndir, dev) if ops.gru_seq_supported(rows, h, ndir) else None

        def fwd_cluster():
            hcur.zero_()
            ops.gru_seq_fwd(s, ndir, rows, h, xp, (3 * h, xrs, xts), hcur, hcur, 0, ru_all[0], ndir * rows * 2 * h,
                            None, 0, c_all[0], ndir * rows * h, wgh, wch, ws, lengths=lengths, out=out,
                            out_strides=(h, ors, ots))
        if ws is not None:
            t_c = bench(fwd_cluster)
            torch.cuda.synchronize()
            print("rows {} ndir {} masked {}: forward as ONE cluster launch {:.2f} us/step ({:.1f} us per loop){}".format(
                rows, ndir, masked, t_c / s, t_c, "  GAVE UP WAITING" if ops.gru_seq_failed(ws) else ""))
        def bwd_cluster():
            dh.zero_()
            ops.gru_seq_bwd(s, ndir, rows, h, dh, d_out, seq, ru_all[0], ndir * rows * 2 * h, c_all[0], ndir * rows * h,
                            None, out, seq, dxp, (3 * h, xrs, xts), wgh, wch, ws, lengths=lengths)
        if ws is not None:
            t_cb = bench(bwd_cluster)
            torch.cuda.synchronize()
            print("rows {} ndir {} masked {}: BPTT as ONE cluster launch {:.2f} us/step ({:.1f} us per loop){}".format(
                rows, ndir, masked, t_cb / s, t_cb, "  GAVE UP WAITING" if ops.gru_seq_failed(ws) else ""))
        t_f = bench(lambda: fwd(False))
        t_ft = bench(lambda: fwd(True))
        t_b = bench(bwd)
        print("rows {} ndir {} masked {}: forward {:.1f} us/step ([K,N] weights) {:.1f} us/step (transposed), "
              "BPTT {:.1f} us/step".format(rows, ndir, masked, t_f / s, t_ft / s, t_b / s))


if __name__ == "__main__":
    main()
