"""Which kernels of ONE training step (headline model, small sizes by default) are torch's own rather than libnmhip's?
torch.profiler lists the device activities of a replayed step; everything named at::native / rocclr comes from a torch
tensor operation on the step's path (fills, copies, stacks), with the Python stack that issued it.

    python tools/torch_kernels_in_step.py [--full]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from neuralmonkey_amd import synthetic
    full = "--full" in sys.argv
    kw = dict(vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512, max_len=50) if full else \
        dict(vocab_src=2000, vocab_tgt=2000, emb=256, rnn=256, max_len=12)
    model = synthetic.build_translation_model(beam_size=0, device="cuda:0", **kw)
    batch, length = (128, 50) if full else (24, 10)
    ds = synthetic.synthetic_dataset(seed=1, batch=batch, src_len=length, tgt_len=length, vocab=kw["vocab_src"])
    tfm = model.tf_manager
    for _ in range(4):
        tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)
        torch.cuda.synchronize()
    foreign = {}
    total = 0
    for ev in prof.events():
        if ev.device_type is not None and "cuda" in str(ev.device_type).lower():
            total += 1
            name = ev.name
            if "at::native" in name or "rocclr" in name or "Memcpy" in name or "Memset" in name:
                foreign[name[:90]] = foreign.get(name[:90], 0) + 1
    print("device activities in the step:", total)
    for name, n in sorted(foreign.items(), key=lambda kv: -kv[1]):
        print("  {:3d} x {}".format(n, name))
    # where do they come from: CPU-side ops with their stacks
    seen = set()
    for ev in prof.events():
        if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::stack", "aten::cat", "aten::add",
                       "aten::mul", "aten::div", "aten::to", "aten::_to_copy", "aten::where", "aten::sum"):
            stack = [fr for fr in (ev.stack or []) if "neuralmonkey_amd" in fr][:2]
            key = (ev.name, tuple(stack))
            if key not in seen:
                seen.add(key)
                print(ev.name, "<-", " | ".join(s.split("neuralmonkey_amd/")[-1] for s in stack))


if __name__ == "__main__":
    main()
