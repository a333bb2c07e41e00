"""Training throughput on FRESH batches of token strings (what a real experiment feeds), headline
model shape (B=128, len 50, H=512, V=32000):

    strings      every batch indexed from strings on the step thread (the reference's feeding model)
    preindexed   dataset mapped to int32 ids once, batches sliced from integers
    +prefetch    worker thread: feed dicts built and uploaded (pinned memory, copy stream) while the
                 previous step runs

bench.py re-feeds one resident batch (the metric is defined on device-resident inputs); this tool
measures what the host side adds on top and how much of it the pipeline hides.

    python tools/input_pipeline_bench.py [--batches 40]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--len", type=int, default=50, dest="length")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--vocab", type=int, default=32000)
    args = ap.parse_args()
    from neuralmonkey_amd import synthetic
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    from neuralmonkey_amd.input_pipeline import Prefetcher, preindex
    model = synthetic.build_translation_model(vocab_src=args.vocab, vocab_tgt=args.vocab, emb=args.hidden,
                                              rnn=args.hidden, max_len=args.length, beam_size=0, device="cuda:0")
    tfm, trainer = model.tf_manager, model.trainer
    feedables = trainer.feedables
    words = model.src_vocab.index_to_word
    rng = np.random.default_rng(0)
    n = args.batches * args.batch

    def sentences(length):
        ids = rng.integers(4, args.vocab, size=(n, length))
        return [[words[i] for i in row] for row in ids]
    ds = Dataset("fresh", {"source": sentences(args.length), "target": sentences(args.length - 1)},
                 BatchingScheme(batch_size=args.batch))
    tokens = n * args.length

    def epoch(batches):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for batch in batches:
            tfm.execute(batch, feedables, [trainer], train=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    warm = ds.subset(0, 4 * args.batch)
    epoch(warm.batches())                                        # graphs captured, allocator warm
    fresh = lambda d: Dataset(d.name, {k: list(d.get_series(k)) for k in d.series}, d.batching)   # no caches
    t_str = epoch(fresh(ds).batches())
    t_str_pf = epoch(Prefetcher(tfm, feedables, train=True, depth=3).iterate(fresh(ds).batches()))
    t0 = time.perf_counter()
    pre = preindex(ds, feedables)
    t_index = time.perf_counter() - t0
    t_pre = epoch(fresh(pre).batches())
    t_pf = epoch(Prefetcher(tfm, feedables, train=True, depth=3).iterate(fresh(pre).batches()))
    resident = next(pre.batches())
    tfm.execute(resident, feedables, [trainer], train=True)
    t_res = epoch([resident] * args.batches)
    for name, t in (("strings", t_str), ("strings+prefetch", t_str_pf), ("preindexed", t_pre),
                    ("preindexed+prefetch", t_pf), ("resident batch", t_res)):
        print("{:20s} {:7.2f} ms/step  {:9.0f} tok/s".format(name, t / args.batches * 1e3, tokens / t))
    print("one-off pre-indexing of the dataset: {:.1f} ms ({:.2f} ms per batch)".format(
        t_index * 1e3, t_index / args.batches * 1e3))


if __name__ == "__main__":
    main()
