#!/usr/bin/env python
"""Benchmark of the attention-decoder hot path on MI355X (BASELINE.json metric).

One "step" = one optimizer step (forward + hand-written backward + clip + Adam)
of the translation.ini-shape model (biGRU-512 encoder, Bahdanau attention,
GRU-512 decoder, V=32000) on one synthetic batch of B=128 sentences per GPU,
src_len=tgt_len=50, driven through ``TensorFlowManager.execute`` exactly as the
reference's training loop does.  ``value`` = target tokens (non-pad target
positions incl. </s>) per second over all GPUs; inputs are resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) with ``roofline`` (fused attention-step kernel,
HBM-bound, timed live with HIP events on its stream) and ``cpu_baseline``
(torch-CPU restatement of the reference's step, NOT TF 1.12 -- TF cannot be
installed here).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="sentences per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch sentences per GPU (default, what the driver runs); strong: --batch "
                         "sentences per optimizer step in total, sharded over the GPUs (BASELINE.json's "
                         "'shards minibatches': 16 per GPU at N=8)")
    ap.add_argument("--len", type=int, default=50, dest="length")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--beam-batches", type=int, default=4, help="beam-5 decode batches to time (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def attention_step_bytes(b, s, a, c):
    """Algorithmic bytes of one fused attention step."""
    return 4 * (b * s * a + b * s * c + 2 * b * s + b * a + b * c)


def cpu_baseline(args, ds, tokens_per_step):
    """torch-CPU restatement of one training step at the reference's op
    granularity (per-step cell / attention / projection / logits), all host cores."""
    from oracle import nm_oracle as O
    from oracle import torch_ref as TR
    h = args.hidden
    params = O.init_params(seed=1234, vocab_src=args.vocab, vocab_tgt=args.vocab, emb=h, rnn=h)
    src = O.pad_ids([list(s) for s in ds.get_series("source")], args.length)
    tgt = O.pad_ids([list(s) for s in ds.get_series("target")], args.length, add_end_symbol=True)
    tgt_tb = np.ascontiguousarray(tgt.T)
    tp = TR.to_torch(params)
    m = {k: torch.zeros_like(v) for k, v in tp.items()}
    v = {k: torch.zeros_like(x) for k, x in tp.items()}
    times = []
    for step in range(1, args.cpu_steps + 2):
        t0 = time.perf_counter()
        _, _, _, grads = TR.train_step_grads(tp, src, tgt_tb, l1_weight=0.0, l2_weight=1e-8)
        TR.clip_and_adam(tp, grads, m, v, step, 1.0)
        times.append(time.perf_counter() - t0)
    best = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {"value": tokens_per_step / best, "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "{} training steps (1 warm-up) of the same B={} len={} V={} batch, torch-CPU fp32 "
                      "restatement of the reference step, not TF 1.12".format(
                          args.cpu_steps, args.batch, args.length, args.vocab),
            "sec_per_step": best}


def main():
    args = parse()
    from neuralmonkey_amd import _lib, distributed, synthetic
    dp = distributed.init_from_env()
    rank = dp.rank if dp else 0
    world = dp.world_size if dp else 1
    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    # one rank per GPU; ranks beyond the visible devices wrap around (only meaningful with
    # NM_DIST_BACKEND=gloo, the single-GPU smoke test of the multi-rank code path)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = "cuda:{}".format(local)
    lib = _lib.load()
    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit("--scaling strong: --batch {} is not divisible by {} GPUs".format(args.batch, world))
        args.batch //= world                  # from here on: sentences per GPU

    h = args.hidden
    model = synthetic.build_translation_model(vocab_src=args.vocab, vocab_tgt=args.vocab, emb=h, rnn=h,
                                              max_len=args.length, beam_size=5, max_steps=args.length,
                                              length_normalization=0.6, l2_weight=1e-8, clip_norm=1.0,
                                              device=dev, seed=1234)
    store = model.tf_manager.sessions[0].store
    if dp:
        dp.broadcast_parameters(store)
    ds = synthetic.synthetic_dataset(seed=1234 + rank, batch=args.batch, src_len=args.length,
                                     tgt_len=args.length, vocab=args.vocab, ragged=False)
    tokens_local = args.batch * args.length               # every target has len-1 tokens + </s>
    tokens_global = tokens_local * world
    tfm, trainer = model.tf_manager, model.trainer

    def step():
        return tfm.execute(ds, trainer.feedables, [trainer], train=True)[0]

    def barrier():
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- beam-5 decode throughput (emitted rank-1 tokens up to and incl. </s>)
    beam_tok_s = beam_ms_per_batch = None
    if args.beam_batches > 0:
        runner = model.beam_runner
        # synthetic decode workload: </s> is made unreachable so that every hypothesis runs the
        # full max_steps=len steps (6400 emitted rank-1 tokens per 128-sentence batch)
        logit_b = store["decoder/state_to_word_b"]
        saved_end_bias = float(logit_b[2].item())
        logit_b[2] = -1e9
        dsb = synthetic.synthetic_dataset(seed=99 + rank, batch=args.batch, src_len=args.length,
                                          tgt_len=args.length, vocab=args.vocab, with_target=False)
        for _ in range(2):                  # warm-up: eager pass (allocations) + HIP-graph capture pass
            out = tfm.execute(dsb, runner.feedables, [runner], compute_losses=False)[0]
        barrier()
        tb = time.perf_counter()
        emitted = 0
        for _ in range(args.beam_batches):
            out = tfm.execute(dsb, runner.feedables, [runner], compute_losses=False)[0]
            emitted += sum(min(len(s) + 1, args.length) for s in out.outputs[runner.output_series])
        barrier()
        tb = time.perf_counter() - tb
        if dp:
            t = torch.tensor([tb, float(emitted)], dtype=torch.float64, device=dev)
            tmax = t.clone()
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
            tb, emitted = float(tmax[0].item()), float(t[1].item())
        beam_tok_s = emitted / tb
        beam_ms_per_batch = tb / args.beam_batches * 1e3
        logit_b[2] = saved_end_bias

    for _ in range(max(args.warmup, 2)):      # >= 2: eager pass (allocations) + HIP-graph capture pass
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0

    # roofline kernel = the fused attention step of autoregressive decoding (greedy here:
    # one query per sentence, B=128, S=50, A=C=1024 -- the "attention-decoder step" of the
    # north star).  Training batches all T steps into one launch, so the step kernel is timed
    # where it really runs once per step: a greedy decode of one 128-sentence batch, 50 steps,
    # launched eagerly with HIP events around every attn_partial launch on its stream.
    logit_b = store["decoder/state_to_word_b"]
    saved_end_bias = float(logit_b[2].item())
    logit_b[2] = -1e9                       # </s> unreachable: all 50 steps run
    dsg = synthetic.synthetic_dataset(seed=77 + rank, batch=args.batch, src_len=args.length,
                                      tgt_len=args.length, vocab=args.vocab, with_target=False)
    grunner = model.greedy_runner
    for _ in range(2):                      # warm-up: eager pass + HIP-graph capture pass
        tfm.execute(dsg, grunner.feedables, [grunner], compute_losses=False)
    barrier()
    tg = time.perf_counter()
    for _ in range(4):
        tfm.execute(dsg, grunner.feedables, [grunner], compute_losses=False)
    barrier()
    greedy_ms = (time.perf_counter() - tg) * 1e3 / 4
    # the same decode once more with graph replay off, HIP events around every attn_partial launch
    sess0 = tfm.sessions[0]
    graphs_were = sess0.use_graphs
    sess0.use_graphs = False
    lib.nm_prof_enable(1)
    tfm.execute(dsg, grunner.feedables, [grunner], compute_losses=False)
    barrier()
    lib.nm_prof_enable(0)
    sess0.use_graphs = graphs_were
    logit_b[2] = saved_end_bias
    tot_ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
    lib.nm_prof_attn_partial(ctypes.byref(tot_ms), ctypes.byref(cnt))
    if dp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        a = c = 2 * h
        step_bytes = attention_step_bytes(args.batch, args.length, a, c)
        avg_us = (tot_ms.value * 1e3 / cnt.value) if cnt.value else None
        achieved = (step_bytes / (avg_us * 1e-6) / 1e9) if avg_us else None
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "attn_partial_pmc.json")
        if os.path.exists(pmc):
            with open(pmc) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch")
        line = {
            "metric": "target tokens/sec/node (train), 512-hid GRU+attn",
            "value": tokens_global * args.steps / elapsed, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "translation.ini-shape: biGRU-{h} enc + Bahdanau attn + GRU-{h} dec, "
                                   "B={b}/GPU, src_len=tgt_len={l}, V={v}, CrossEntropyTrainer(l2=1e-8, "
                                   "clip_norm=1.0) + Adam(1e-4), one optimizer step per step".format(
                                       h=h, b=args.batch, l=args.length, v=args.vocab),
                       "global_batch": args.batch * world, "seq_len": args.length,
                       "parallelism": "dp{}".format(world)},
            "loss": res.losses["decoder - cost"],
            "beam5_decode_tok_s": beam_tok_s, "beam5_ms_per_batch": beam_ms_per_batch,
            "greedy_decode_tok_s": tokens_local / (greedy_ms * 1e-3), "greedy_ms_per_batch": greedy_ms,
            "roofline": {"kernel": "attn_partial_fast (fused Bahdanau score+softmax+context of one decoding step)",
                         "measured_in": "greedy decode of one B={} batch, {} steps".format(args.batch, args.length),
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                         "avg_launch_us": avg_us, "launches": cnt.value,
                         "algorithmic_bytes_per_launch": step_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args, ds, tokens_local)
            except Exception as exc:                      # pragma: no cover  (never hide the GPU number)
                line["cpu_baseline"] = {"value": None, "error": repr(exc)}
        print(json.dumps(line), flush=True)
    if dp:
        distributed.shutdown()


if __name__ == "__main__":
    main()
