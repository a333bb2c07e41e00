"""nm_beam_topk_step_fused (one register-resident scan of every hypothesis row: max / lse /
candidate filter / exact top-k) against nm_beam_topk_step (row statistics + slice-wise insertion
lists, itself checked against the oracle in test_kernels_gpu.py): every output bit-identical, on
inputs built to stress the candidate filter -- exact ties inside and across the top k, logits one
ulp apart, more equal maxima than the candidate list holds, the -1e9 log-prob sums of the first
beam step (all scores of a row collapse to one value), finished rows, row lengths at the register
tiling boundaries."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

END = 2


def _run_both(dev, logits, k, lps, lens, fin, alpha=0.6):
    from neuralmonkey_amd import ops
    rows, v = logits.shape
    b = rows // k
    T = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    ld, lpsd, lensd, find = T(logits), T(lps), T(lens, torch.int32), T(fin.astype(np.int32), torch.int32)
    pen = ops.length_penalty_table(64, alpha, dev)
    i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)

    def outputs():
        return [torch.empty((b, k), device=dev), i32(b, k), i32(b, k), torch.empty((b, k), device=dev), i32(b, k),
                i32(b, k), i32(b, k)]
    ws = ops.beam_workspace(b, k, v, dev)
    mx, lse = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.row_stats(ld, mx, lse, None)
    ref = outputs()
    ops.beam_topk_step(ld, b, k, mx, lse, lpsd, lensd, find, pen, END, *ref, ws)
    got = outputs()
    mx2, lse2 = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.beam_topk_step_fused(ld, b, k, lpsd, lensd, find, pen, END, *got, ws, mx2, lse2)
    live = ~fin.reshape(-1)
    assert torch.equal(mx2.cpu()[live], mx.cpu()[live]) and torch.equal(lse2.cpu()[live], lse.cpu()[live])
    names = ["score", "word", "beam", "logprob_sum", "lengths", "finished", "src_row"]
    for name, g, r in zip(names, got, ref):
        assert torch.equal(g.cpu(), r.cpu()), name
    return [g.cpu().numpy() for g in got]


def _state(rng, b, k, first_step=False, finished_frac=0.0):
    lps = (-rng.random((b, k)) * 30).astype(np.float32)
    if first_step:
        lps[:, 0] = 0.0
        lps[:, 1:] = -1e9
    lens = rng.integers(0, 40, size=(b, k)).astype(np.int32)
    fin = rng.random((b, k)) < finished_frac
    return lps, lens, fin


@pytest.mark.parametrize("b,k,v", [(16, 5, 32000), (4, 8, 32000), (3, 3, 4096), (2, 5, 32768), (2, 4, 65536),
                                   (5, 2, 2048), (2, 5, 131072)])
def test_fused_equals_two_pass_on_random_rows(dev, b, k, v):
    rng = np.random.default_rng(b * 7 + k + v)
    logits = (rng.standard_normal((b * k, v)) * 4).astype(np.float32)
    _run_both(dev, logits, k, *_state(rng, b, k, finished_frac=0.25))
    _run_both(dev, logits, k, *_state(rng, b, k, first_step=True))


def test_fused_exact_ties_and_one_ulp_neighbours(dev):
    rng = np.random.default_rng(5)
    b, k, v = 6, 5, 32000
    logits = (rng.standard_normal((b * k, v)) * 2).astype(np.float32)
    for r in range(b * k):
        top = np.float32(9.0 + r * 0.01)
        idx = rng.choice(v, size=12, replace=False)
        logits[r, idx[:4]] = top                                   # four exact ties for the row maximum
        logits[r, idx[4:8]] = np.nextafter(top, np.float32(0))     # four more one ulp below
        logits[r, idx[8:]] = np.nextafter(np.nextafter(top, np.float32(0)), np.float32(0))
    lps, lens, fin = _state(rng, b, k)
    lps[:] = lps[:, :1]                                            # equal sums: ties also ACROSS the beams of a sentence
    lens[:] = lens[:, :1]
    out = _run_both(dev, logits, k, lps, lens, fin)
    assert (out[1] >= 0).all()


def test_fused_overflowing_candidate_list_takes_the_full_path(dev):
    rng = np.random.default_rng(6)
    b, k, v = 3, 5, 32000
    logits = (rng.standard_normal((b * k, v))).astype(np.float32)
    logits[0, :] = 0.0                                             # every logit equal: 32000 candidates
    logits[1, rng.choice(v, size=700, replace=False)] = 30.0       # 700 equal maxima (> the 256-entry list)
    logits[2, rng.choice(v, size=256, replace=False)] = 30.0       # exactly the list size
    logits[3, rng.choice(v, size=257, replace=False)] = 30.0
    logits[4] = np.float32(1e-3) * np.arange(v, dtype=np.float32)  # a ramp: neighbours far closer than the margin
    lps, lens, fin = _state(rng, b, k)
    out = _run_both(dev, logits, k, lps, lens, fin)
    # sentence 0, beam 0 (all-equal row) and its neighbours: the lowest word ids of the best row win ties
    assert (np.diff(out[0], axis=1) <= 0).all()


def test_fused_huge_and_tiny_magnitudes(dev):
    rng = np.random.default_rng(8)
    b, k, v = 4, 5, 32000
    logits = (rng.standard_normal((b * k, v))).astype(np.float32)
    logits[:5] *= 1e4                                              # huge dynamic range
    logits[5:10] *= 1e-6                                           # all logits nearly equal
    logits[10:15] += 1e5                                           # large common offset (few mantissa bits left)
    lps, lens, fin = _state(rng, b, k)
    lps[1] = np.float32(-3e4)
    _run_both(dev, logits, k, lps, lens, fin)


def test_backtrace_equals_the_per_step_history_gather(dev):
    """nm_beam_backtrace (one walk over the back-pointers at the end) == nm_beam_reorder_tokens applied at every
    step (the reference's per-step gather of the whole token history, beam_search_decoder.py:546-551)."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(11)
    bsz, k, steps = 7, 5, 13
    rows = bsz * k
    parent = np.stack([(np.arange(bsz)[:, None] * k + rng.integers(0, k, size=(bsz, k))).reshape(-1)
                       for _ in range(steps)]).astype(np.int32)                 # a parent inside the same sentence
    word = rng.integers(0, 1000, size=(steps, rows)).astype(np.int32)
    first = rng.integers(0, 1000, size=rows).astype(np.int32)
    dt = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
    tok = torch.zeros((2, steps + 1, rows), dtype=torch.int32, device=dev)
    tok[0, 0].copy_(dt(first))
    for s in range(steps):
        ops.beam_reorder_tokens(tok[s & 1], dt(parent[s]), dt(word[s]), tok[(s & 1) ^ 1], s + 1, rows)
    want = tok[steps & 1].cpu().numpy()
    out = torch.full((steps + 1, rows), -1, dtype=torch.int32, device=dev)
    ops.beam_backtrace(dt(parent), dt(word), dt(first), out, steps)
    assert np.array_equal(out.cpu().numpy(), want)
    # NumPy restatement of the walk
    def walk(n):
        ref = np.zeros((n + 1, rows), np.int32)
        for r in range(rows):
            cur = r
            for t in range(n - 1, -1, -1):
                ref[t + 1, r] = word[t, cur]
                cur = parent[t, cur]
            ref[0, r] = first[cur]
        return ref
    assert np.array_equal(want, walk(steps))
    short = torch.full((steps + 1, rows), -1, dtype=torch.int32, device=dev)
    ops.beam_backtrace(dt(parent), dt(word), dt(first), short, 4)                # a search that stopped after 4 bodies
    assert np.array_equal(short[:5].cpu().numpy(), walk(4)) and int(short[5:].max()) == -1
