"""Training steps only, headline shape (B=128, len 50, H=512, V=32000) -- the workload of
bench.py's train leg without the decode legs, for per-kernel profiles:

    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- \
        python /root/repo/tools/train_profile.py --steps 10
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--len", type=int, default=50, dest="length")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--vocab", type=int, default=32000)
    args = ap.parse_args()
    from neuralmonkey_amd import synthetic
    if os.environ.get("NM_MAIN_PRIO"):        # experiment: the step's own stream above the side (leaf-GEMM) stream
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    model = synthetic.build_translation_model(vocab_src=args.vocab, vocab_tgt=args.vocab, emb=args.hidden,
                                              rnn=args.hidden, max_len=args.length, beam_size=0, device="cuda:0")
    synthetic.load_baseline_weights(model.tf_manager.sessions[0].store)
    ds = synthetic.synthetic_dataset(seed=1234, batch=args.batch, src_len=args.length, tgt_len=args.length,
                                     vocab=args.vocab)
    tfm, trainer = model.tf_manager, model.trainer
    for _ in range(3):
        tfm.execute(ds, trainer.feedables, [trainer], train=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tfm.execute(ds, trainer.feedables, [trainer], train=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print("train: {:.2f} ms/step  {:.0f} tok/s".format(dt * 1e3, args.batch * args.length / dt))


if __name__ == "__main__":
    main()
