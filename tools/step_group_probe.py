"""The four GEMM groups of ONE 640-row (beam) decoder step, microseconds per launch: the register-staged 32x32 / 64x64
kernels (NM_STEP_DMA=0) against the LDS-DMA 64x64 tiles (csrc/nm_step.hip: step_group_dma_kernel)."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def one():
    from neuralmonkey_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    m, h = 640, 512
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
    groups = {"gates   N=1024 K=512": [(1024, 512)], "cand    N=512  K=512": [(512, 512)],
              "q + out N=1024+512 K=512": [(1024, 512), (512, 512)], "out_c   N=512  K=1024": [(512, 1024)]}
    for name, probs in groups.items():
        specs = []
        for n, k in probs:
            specs.append(dict(A=rnd(m, k), lda=k, Bt=rnd(n, k), ldb=k, N=n, K=k, epilogue=0, bias=rnd(n), C=torch.empty(m, n, device=dev), ldc=n))
        grp = ops.StepGroup(m, specs)
        for _ in range(5):
            grp.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            grp.launch()
        e1.record()
        torch.cuda.synchronize()
        ref = specs[0]["A"].double() @ specs[0]["Bt"].double().t() + specs[0]["bias"].double()
        err = float((specs[0]["C"].double() - ref).abs().max() / ref.abs().max())
        print("  {:28s} {:7.2f} us   err {:.1e}".format(name, e0.elapsed_time(e1) * 1e3 / 50, err), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for tag, env in (("register-staged (default)", {"NM_STEP_DMA": "0"}), ("LDS-DMA 64x64 (NM_STEP_DMA=1)", {"NM_STEP_DMA": "1"})):
            print(tag, flush=True)
            e = dict(os.environ)
            e.update(env)
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=e, timeout=300)
