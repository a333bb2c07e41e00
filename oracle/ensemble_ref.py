"""CPU restatement of beam-search ensembling (runners/beamsearch_runner.py:38-82 around
decoders/beam_search_decoder.py:394-556): every model advances its own decoder on the same
hypotheses, the step distributions are averaged in log space
(``scipy.special.logsumexp(prev_logprobs, 0) - log(num_sessions)``, :50-55) and the beam body
selects on the average.

TEST INFRASTRUCTURE ONLY.  PARITY PINNED to the reference's own code (see oracle/nm_oracle.py): fixture
``ensemble`` (BeamSearchRunner over three sessions, call by call)."""
import math
from typing import List

import numpy as np
import torch

from .general_ref import END, INF, PAD, START, GeneralModel


def beam_ensemble(models: List[GeneralModel], src_ids, k: int, max_steps: int, alpha: float):
    with torch.no_grad():
        dt = models[0].dtype
        bsz = src_ids.shape[0]
        rows = bsz * k
        setups = [m._decode_setup(src_ids, k) for m in models]          # (st, hf, mask, state) per model
        states = [list(s[3]) for s in setups]

        def step_all(emb_ids, t):
            lps = []
            for i, m in enumerate(models):
                st, hf, mask, _ = setups[i]
                table = m.p[m.cfg.dec_name + "/word_embeddings"]
                out, states[i], _ = m.decoder_step(table[emb_ids], states[i], st, hf, mask, False, t)
                lps.append(torch.log_softmax(m.logits(out), -1))
            return lps

        lps = step_all(torch.full((rows,), START), 0)
        vsz = lps[0].shape[1]
        token_ids = lps[0].argmax(1).view(1, bsz, k)       # parent symbol of the first model (dropped later)
        mean_lp = lambda xs: (torch.logsumexp(torch.stack(xs).double(), 0) - math.log(len(xs))).to(dt)
        prev_lp = mean_lp(lps).view(bsz, k, vsz)
        logprob_sum = torch.tensor([0.0] + [-INF] * (k - 1), dtype=dt).repeat(bsz, 1)
        lengths = torch.zeros(bsz, k, dtype=torch.int64)
        finished = torch.zeros(bsz, k, dtype=torch.bool)
        scores = torch.zeros(bsz, k, dtype=dt)
        fin_row = torch.full((vsz,), -INF, dtype=dt)
        fin_row[PAD] = 0.0
        bidx = torch.arange(bsz).view(-1, 1)
        step, min_gap = 1, float("inf")
        while (step - 1) < max_steps and not bool(finished.all()):
            fm = finished.to(dt).unsqueeze(-1)
            lp = (1.0 - fm) * prev_lp + fm * fin_row
            hyp = logprob_sum.unsqueeze(-1) + lp
            hyp_len = lengths + 1 - finished.to(torch.int64)
            pen = ((5.0 + hyp_len.to(dt)) / 6.0) ** alpha
            flat = (hyp / pen.unsqueeze(-1)).reshape(bsz, k * vsz)
            order = torch.argsort(-flat, dim=1, stable=True)[:, :k + 1]
            top = torch.gather(flat, 1, order)
            live = ~finished.all(1)
            if order.shape[1] > k and bool(live.any()):
                gap = (top[live, k - 1] - top[live, k]) / top[live, k - 1].abs().clamp_min(1e-30)
                min_gap = min(min_gap, float(gap.min()))
            idx, top = order[:, :k], top[:, :k]
            word, beam = idx % vsz, idx // vsz
            lengths = hyp_len[bidx, beam]
            logprob_sum = hyp.reshape(bsz, k * vsz)[bidx, idx]
            finished = finished[bidx, beam] | (word == END)
            src = (bidx * k + beam).reshape(-1)
            for i in range(len(models)):
                states[i] = [s[src] for s in states[i]]
            prev_lp = mean_lp(step_all(word.reshape(-1), step)).view(bsz, k, vsz)
            token_ids = torch.cat([token_ids[:, bidx, beam], word.unsqueeze(0)], 0)
            scores = top
            step += 1
        return token_ids.numpy(), scores.numpy(), min_gap


def beam_ensemble_transformer(models, src_ids, k: int, max_steps: int, alpha: float, follow=None):
    """The same protocol over ``transformer_ref.TransformerModel`` replicas (the only parent decoder the reference's
    runner can ensemble at this commit: over an RNN ``Decoder`` its feed dictionary is keyed by a structure that
    holds lists and raises TypeError, tests/golden/ref_exec/defects.npz).  Every model re-runs its stack over the
    whole prefix of every hypothesis (decoders/transformer.py:487-516); the step distributions are averaged in log
    space (float64 logsumexp as scipy's, beamsearch_runner.py:50-55) and one beam body (``beam_search_core``) selects
    on the average -- the averaged log-probabilities are themselves normalised, so handing them to the core as
    "logits" leaves them unchanged up to one rounding."""
    from . import nm_oracle as O
    with torch.no_grad():
        dt = models[0].dtype
        bsz = src_ids.shape[0]
        rows = bsz * k
        enc = []
        for m in models:
            states, masks = m.encode_all(src_ids, False)
            enc.append(([e.repeat_interleave(k, 0) for e in states], [x.repeat_interleave(k, 0) for x in masks]))
        tables = [m.target_embeddings() for m in models]
        seqs = [t[torch.full((rows,), START)].unsqueeze(1) for t in tables]
        box = {"seqs": seqs, "mask": torch.ones(rows, 1, dtype=dt)}

        def mean_lp():
            lps = [torch.log_softmax(m.logits(m._next_output(s, box["mask"], es, em)), -1)      # pylint: disable=protected-access
                   for m, s, (es, em) in zip(models, box["seqs"], enc)]
            return (torch.logsumexp(torch.stack(lps).double(), 0) - math.log(len(lps))).to(dt).numpy()

        finished = {"f": np.zeros(rows, dtype=bool)}

        def step_fn(src_rows, words):
            src = torch.as_tensor(np.asarray(src_rows, dtype=np.int64))
            w = torch.as_tensor(np.asarray(words, dtype=np.int64))
            finished["f"] = finished["f"][np.asarray(src_rows)] | (np.asarray(words) == END)
            box["seqs"] = [torch.cat([s[src], t[w].unsqueeze(1)], 1) for s, t in zip(box["seqs"], tables)]
            live = torch.as_tensor((~finished["f"]).astype(np.float64)).to(dt).reshape(-1, 1)
            box["mask"] = torch.cat([box["mask"][src], live], 1)
            return mean_lp()
        return O.beam_search_core(mean_lp(), step_fn, bsz, k, max_steps, alpha, follow=follow)
