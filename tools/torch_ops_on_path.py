"""Which torch tensor operations (each one a torch kernel launch or a runtime copy / fill) does a step of the engine
issue, and from which line?  A TorchFunctionMode logs every torch function called on a CUDA tensor while ONE step runs
eagerly (graphs off: NM_GRAPHS=0 semantics are the same launches, just not captured), with the innermost
neuralmonkey_amd frame that called it.

    python tools/torch_ops_on_path.py [train|greedy|beam|transformer_train] ..."""
import collections
import os
import sys
import traceback

import torch
from torch.overrides import TorchFunctionMode

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

QUIET = {"size", "dim", "stride", "data_ptr", "numel", "is_contiguous", "view", "__getitem__", "as_strided", "shape",
         "device", "dtype", "is_cuda", "element_size", "storage_offset", "transpose", "t", "unsqueeze", "squeeze",
         "__get__", "narrow", "select", "expand", "permute", "detach", "requires_grad", "grad", "is_floating_point",
         "dim_order", "untyped_storage", "record_stream", "__len__", "__repr__", "__format__", "view_as", "unbind",
         "ndim", "nelement", "is_pinned", "__bool__", "layout", "is_sparse", "is_quantized", "names",
         "_is_view", "_base", "is_leaf", "T", "mT", "real", "__hash__", "__eq__", "type", "is_meta", "is_complex", "__class__"}


class Log(TorchFunctionMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", str(func))
        cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list(kwargs.values()))
        if name in ("empty", "zeros", "ones", "full", "tensor", "arange", "empty_like", "zeros_like"):
            dev = str(kwargs.get("device", ""))
            cuda = cuda or "cuda" in dev
        if cuda and name in ("reshape", "contiguous", "flatten"):
            t = args[0]
            cuda = isinstance(t, torch.Tensor) and not t.is_contiguous()        # only when it has to copy
        if cuda and name not in QUIET:
            if torch.cuda.is_current_stream_capturing():
                name = name + " [captured]"
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "neuralmonkey_amd" in fr.filename:
                    where = "{}:{} {}".format(fr.filename.split("neuralmonkey_amd/")[-1], fr.lineno, fr.name)
                    break
            self.hits[(name, where)] += 1
        return func(*args, **kwargs)


def run(what):
    from neuralmonkey_amd import synthetic
    os.environ.setdefault("NM_GRAPHS", "0")
    if what in ("train", "greedy", "beam"):
        model = synthetic.build_translation_model(beam_size=5 if what == "beam" else 0, device="cuda:0", vocab_src=2000,
                                                  vocab_tgt=2000, emb=256, rnn=256, max_len=12)
        ds = synthetic.synthetic_dataset(seed=1, batch=24, src_len=10, tgt_len=10, vocab=2000, with_target=what == "train")
        tfm = model.tf_manager
        if what == "train":
            step = lambda: tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)
        else:
            runner = model.beam_runner if what == "beam" else model.greedy_runner
            step = lambda: tfm.execute(ds, runner.feedables, [runner], compute_losses=False)
    elif what == "transformer_train":
        m = synthetic.build_transformer_model(vocab=2000, dim=128, depth=2, heads=4, ff=256, max_len=12, max_steps=12,
                                              beam_size=0, device="cuda:0")
        ds = synthetic.synthetic_dataset(seed=1, batch=24, src_len=10, tgt_len=10, vocab=2000)
        tfm = m.tf_manager
        step = lambda: tfm.execute(ds, m.trainer.feedables, [m.trainer], train=True)
    else:
        raise SystemExit("unknown step " + what)
    log = Log()
    if os.environ.get("NM_GRAPHS") == "1":           # graphs on: what gets CAPTURED matters (it is replayed ever after)
        with log:
            for _ in range(4):
                step()
    else:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        with log:
            step()
    torch.cuda.synchronize()
    print("==== {}: {} torch calls on CUDA tensors in one step".format(what, sum(log.hits.values())))
    for (name, where), n in sorted(log.hits.items(), key=lambda kv: (-kv[1], kv[0])):
        print("  {:4d} x {:14s} {}".format(n, name, where))


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["train"]):
        run(w)
