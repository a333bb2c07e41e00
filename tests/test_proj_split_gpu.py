"""OPT-IN path (NM_PROJ_SPLIT=1): the decoding steps' vocabulary projection on the bf16 matrix cores -- both operands
split three ways into bf16 (24 mantissa bits), six products, fp32 accumulate (csrc/nm_gemm_bf16x3.hip) -- against
float64 and against the exact-fp32 kernel it may stand in for (tf.matmul(state, decoding_w) + bias with the argmax /
log-softmax statistics of decoders/autoregressive.py:450-470).  Gate: its error against float64 is no more than twice
the exact kernel's (both are rounding noise of fp32 sums over K), the statistics describe its own logits exactly, and
decoding parity -- symbols, beams, scores against the oracle and the reference-executed fixtures -- holds under the
switch (a fresh process runs those tests with NM_PROJ_SPLIT=1; marked slow)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n,k,trans_b", [
    (128, 32000, 512, False),        # greedy step of the headline model
    (640, 32000, 512, False),        # beam step
    (640, 32000, 512, True),         # tied embeddings: W is [V, K]
    (77, 1000, 64, False),           # rows and columns that do not fill their tiles
    (5, 300, 32, True),
])
def test_split_projection_against_float64_and_the_exact_kernel(dev, m, n, k, trans_b):
    from neuralmonkey_amd import ops
    g = torch.Generator(device=dev).manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev, generator=g)
    w = torch.randn((n, k) if trans_b else (k, n), device=dev, generator=g) * 0.05
    bias = torch.randn(n, device=dev, generator=g) * 0.1
    want = a.double() @ (w.double().t() if trans_b else w.double()) + bias.double()
    stats32, stats3 = ops.logits_stats_buffer(m, n, dev), ops.logits_stats_buffer(m, n, dev)
    out32, out3 = torch.empty(m, n, device=dev), torch.full((m, n), float("nan"), device=dev)
    ops.logits_stats_gemm(a, w, bias, stats32, out=out32, trans_b=trans_b)
    planes = ops.proj_split_prepare(w, trans_b=trans_b)
    try:
        ops.logits_stats_gemm(a, w, bias, stats3, out=out3, trans_b=trans_b)
        stats_only = ops.logits_stats_buffer(m, n, dev)
        ops.logits_stats_gemm(a, w, bias, stats_only, trans_b=trans_b)           # greedy decoding: no logits stored
    finally:
        ops.proj_split_forget(w)
    assert planes is not None
    scale = float(want.abs().max())
    err32 = float((out32.double() - want).abs().max()) / scale
    err3 = float((out3.double() - want).abs().max()) / scale
    assert err3 <= 2.0 * err32 + 1e-7, (err3, err32)
    assert not torch.equal(out3, out32), "the split kernel did not run"
    # the statistics are those of the kernel's own logits: per 128-column tile max, first argmax, sum exp(x - max)
    tiles = (n + 127) // 128
    rec = stats3.view(m, tiles, 4)
    assert torch.equal(rec, stats_only.view(m, tiles, 4))
    pad = torch.full((m, tiles * 128), float("-inf"), device=dev)
    pad[:, :n] = out3
    t = pad.view(m, tiles, 128)
    tmax, targ = t.max(dim=2)
    assert torch.equal(rec[:, :, 0], tmax)
    first = (t == tmax[:, :, None]).float().argmax(dim=2) + torch.arange(tiles, device=dev)[None, :] * 128
    assert torch.equal(rec[:, :, 2].contiguous().view(torch.int32), first.int())
    sums = torch.exp(t.double() - tmax.double()[:, :, None]).sum(2)
    assert float(((rec[:, :, 1].double() - sums).abs() / sums).max()) < 1e-5
    # after forget() the exact kernel is back
    again = torch.empty(m, n, device=dev)
    ops.logits_stats_gemm(a, w, bias, stats32, out=again, trans_b=trans_b)
    assert torch.equal(again, out32)


@pytest.mark.slow
def test_decoding_parity_holds_under_the_split_projection():
    env = dict(os.environ, NM_PROJ_SPLIT="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "tests/test_fullsize_parity_gpu.py",
                          "tests/test_reference_exec_gpu.py", "tests/test_engine_gpu.py", "tests/test_fullsize_gpu.py",
                          "-k", "not training_step"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
