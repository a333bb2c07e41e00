"""A taped decoder step's attention backward up to the query: nm_attn_step_bwd (one launch) against the three launches
it replaces (batched M = 1 product, nm_attn_softmax_bwd, nm_attn_energy_bwd in its query-only mode), event-timed over
back-to-back launches on rotating operand sets.   python tools/attn_step_bwd_bench.py [B S C A]"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from neuralmonkey_amd import ops
    b, s, c, a = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (64, 50, 1024, 512)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    mk = lambda *shape: torch.randn(*shape, generator=g).to(dev)
    nset = 8
    sets = [dict(dctx=mk(b, c), st=mk(b, s, c), e=mk(b, s), hf=mk(b, s, a), y=mk(b, a), v=mk(a), de=mk(1, b, s), dy=mk(b, a),
                 dw=mk(b, 1, s)) for _ in range(nset)]
    mask = torch.ones(b, s, device=dev)

    def three(z):
        ops.gemm(z["dctx"].view(b, 1, c), z["st"], out=z["dw"], trans_b=True)
        ops.attn_softmax_bwd(z["dw"].view(1, b, s), z["e"].view(1, b, s), mask, z["de"], b)
        ops.attn_energy_bwd(z["de"], z["hf"], z["y"].view(1, b, a), z["v"], None, None, z["dy"].view(1, b, a))

    def one(z):
        ops.attn_step_bwd(z["dctx"], z["st"], z["e"], mask, z["hf"], z["y"], z["v"], z["de"][0], z["dy"])

    for name, fn in (("three launches", three), ("nm_attn_step_bwd", one)):
        for i in range(20):
            fn(sets[i % nset])
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 400
        t0.record()
        for i in range(n):
            fn(sets[i % nset])
        t1.record()
        torch.cuda.synchronize()
        print("B=%d S=%d C=%d A=%d  %-18s %7.2f us per step" % (b, s, c, a, name, t0.elapsed_time(t1) * 1e3 / n))


if __name__ == "__main__":
    main()
