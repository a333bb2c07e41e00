"""Soak test of the cluster time loops: the forward and the BPTT loop of the headline shapes, several hundred times, with
GEMMs of varying size launched beside them on a second stream (uneven load is what exposes a broken hand-off), every
result compared with the first run's, the error word checked every time.

    python tools/cluster_soak.py [iterations=300]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuralmonkey_amd import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    side = torch.cuda.Stream(device=dev)
    mats = [torch.randn(n, n, device=dev, generator=g) for n in (512, 2048, 4096)]
    bad = 0
    for rows, ndir, h, s in ((128, 1, 512, 50), (128, 2, 512, 50), (37, 2, 256, 23)):
        rn = lambda *shape: torch.randn(*shape, device=dev, generator=g) * 0.2
        xp, wgh, wch = rn(rows * s, ndir * 3 * h), rn(ndir, h, 2 * h) * 0.3, rn(ndir, h, h) * 0.3
        lengths = torch.randint(1, s + 1, (rows,), device=dev, dtype=torch.int32, generator=g) if ndir == 2 else None
        d_out = rn(rows, s, ndir * h)
        ws = ops.gru_seq_workspace(rows, h, ndir, dev)
        xst, seq = (3 * h, s * ndir * 3 * h, ndir * 3 * h), (h, s * ndir * h, ndir * h)
        first = None
        for it in range(iters):
            hcur = torch.zeros(ndir, rows, h, device=dev)
            states = torch.zeros(rows, s, ndir * h, device=dev)
            ru, cs = torch.empty(s, ndir, rows, 2 * h, device=dev), torch.empty(s, ndir, rows, h, device=dev)
            dh, dxp = torch.zeros(ndir, rows, h, device=dev), torch.zeros(rows * s, ndir * 3 * h, device=dev)
            if it % 3:
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(1 + it % 4):
                        ops.gemm(mats[it % 3], mats[it % 3])
            ops.gru_seq_fwd(s, ndir, rows, h, xp, xst, hcur, hcur, 0, ru[0], ndir * rows * 2 * h, None, 0, cs[0],
                            ndir * rows * h, wgh, wch, ws, lengths=lengths, out=states, out_strides=seq)
            ops.gru_seq_bwd(s, ndir, rows, h, dh, d_out, seq, ru[0], ndir * rows * 2 * h, cs[0], ndir * rows * h, None,
                            states, seq, dxp, xst, wgh, wch, ws, lengths=lengths)
            torch.cuda.synchronize()
            if ops.gru_seq_failed(ws):
                print("iteration", it, "GAVE UP WAITING")
                bad += 1
            got = (hcur, states, dh, dxp)
            if first is None:
                first = [t.clone() for t in got]
            elif not all(torch.equal(a, b) for a, b in zip(got, first)):
                print("rows", rows, "ndir", ndir, "iteration", it, "DIFFERS from the first run:",
                      [float((a - b).abs().max()) for a, b in zip(got, first)])
                bad += 1
        print("rows {} x {} directions, H {}: {} iterations, forward + BPTT identical every time: {}".format(
            rows, ndir, h, iters, "yes" if bad == 0 else "NO"))
    # the one-stage loops of round 6 (ONE hand-off per step, two granule buffers that alternate): NematusGRU and LSTM
    for kind in ("nematus", "lstm"):
        for rows, ndir, h, s in ((128, 2, 512, 50), (37, 2, 256, 23), (16, 1, 384, 40)):
            rn = lambda *shape: torch.randn(*shape, device=dev, generator=g) * 0.2
            nb = 3 if kind == "nematus" else 4
            xp = rn(rows * s, ndir * nb * h)
            ug, uc, wh = rn(ndir, h, 2 * h) * 0.3, rn(ndir, h, h) * 0.3, rn(ndir, h, 4 * h) * 0.3
            lengths = torch.randint(1, s + 1, (rows,), device=dev, dtype=torch.int32, generator=g)
            d_out = rn(rows, s, ndir * h)
            floats = ops.nematus_seq_workspace_floats(rows, h, ndir) if kind == "nematus" else ops.lstm_seq_workspace_floats(rows, h, ndir)
            ws = torch.empty(floats, device=dev)
            xst, seq = (nb * h, s * ndir * nb * h, ndir * nb * h), (h, s * ndir * h, ndir * h)
            dst = (4 * h, s * ndir * 4 * h, ndir * 4 * h)
            first, bad_here = None, 0
            for it in range(iters):
                hzero, hcur = torch.zeros(ndir, rows, h, device=dev), torch.zeros(ndir, rows, h, device=dev)
                states = torch.zeros(rows, s, ndir * h, device=dev)
                gates = torch.empty(s, ndir, rows, (2 if kind == "nematus" else 4) * h, device=dev)
                cs, sc = torch.empty(s, ndir, rows, h, device=dev), torch.empty(s, ndir, rows, h, device=dev)
                dh, dxp = torch.zeros(ndir, rows, h, device=dev), torch.zeros(rows * s, ndir * 4 * h, device=dev)
                if it % 3:
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        for _ in range(1 + it % 4):
                            ops.gemm(mats[it % 3], mats[it % 3])
                if kind == "nematus":
                    ops.nematus_seq_fwd(s, ndir, rows, h, xp, xst, hzero, hcur, 0, gates[0], ndir * rows * 2 * h, sc[0],
                                        ndir * rows * h, cs[0], ndir * rows * h, ug, uc, ws, lengths=lengths, out=states,
                                        out_strides=seq)
                    ops.nematus_seq_bwd(s, ndir, rows, h, dh, d_out, seq, gates[0], ndir * rows * 2 * h, sc[0],
                                        ndir * rows * h, cs[0], ndir * rows * h, None, states, seq, dxp, dst, ug, uc, ws,
                                        lengths=lengths)
                else:
                    ops.lstm_seq_fwd(s, ndir, rows, h, xp, xst, hzero, hcur, 0, gates[0], ndir * rows * 4 * h, cs[0],
                                     ndir * rows * h, wh, ws, lengths=lengths, out=states, out_strides=seq)
                    ops.lstm_seq_bwd(s, ndir, rows, h, dh, d_out, seq, gates[0], ndir * rows * 4 * h, cs[0], ndir * rows * h,
                                     dxp, dst, wh, ws, lengths=lengths)
                torch.cuda.synchronize()
                if ops.gru_seq_failed(ws):
                    print(kind, "iteration", it, "GAVE UP WAITING")
                    bad_here += 1
                got = (hcur, states, dh, dxp)
                if first is None:
                    first = [t.clone() for t in got]
                elif not all(torch.equal(a, b) for a, b in zip(got, first)):
                    print(kind, "rows", rows, "ndir", ndir, "iteration", it, "DIFFERS from the first run:",
                          [float((a - b).abs().max()) for a, b in zip(got, first)])
                    bad_here += 1
            bad += bad_here
            print("{}: rows {} x {} directions, H {}: {} iterations, forward + BPTT identical every time: {}".format(
                kind, rows, ndir, h, iters, "yes" if bad_here == 0 else "NO"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
