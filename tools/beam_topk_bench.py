"""Time nm_beam_topk_step (B=128, k=5, V=32000) under the NM_BEAM_NS slice-count override."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from neuralmonkey_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
b, k, v = 128, 5, 32000
rows = b * k
logits = torch.randn((rows, v), device=dev)
rmax, rlse = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
argmax = torch.empty(rows, dtype=torch.int32, device=dev)
ops.row_stats(logits, rmax, rlse, argmax)
lps = torch.zeros((b, k), device=dev)
lens = torch.zeros((b, k), dtype=torch.int32, device=dev)
fin = torch.zeros((b, k), dtype=torch.int32, device=dev)
penalty = ops.length_penalty_table(64, 0.6, dev)
scores = torch.empty((b, k), device=dev)
word, beam, src = (torch.empty((b, k), dtype=torch.int32, device=dev) for _ in range(3))
lps2, lens2, fin2 = torch.empty_like(lps), torch.empty_like(lens), torch.empty_like(fin)
ws = ops.beam_workspace(b, k, v, dev)
allfin = torch.ones(1, dtype=torch.int32, device=dev)


def run(n):
    for _ in range(n):
        ops.beam_topk_step(logits, b, k, rmax, rlse, lps, lens, fin, penalty, 2, scores, word, beam, lps2, lens2, fin2,
                           src, ws, allfin)


run(5)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(200)
torch.cuda.synchronize()
print("beam_topk_step: {:.1f} us".format((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for _ in range(200):
    ops.row_stats(logits, rmax, rlse, None)
torch.cuda.synchronize()
print("row_stats: {:.1f} us".format((time.perf_counter() - t0) / 200 * 1e6))
