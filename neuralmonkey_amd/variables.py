"""Variable store: every model variable is a view into one flat fp32 buffer.

The reference keeps variables in the TF graph under ``<part-name>/...`` names
(model/parameterized.py:68-72, SURVEY section 9 "Variable naming"); we keep the
same names but lay all of them out in a single contiguous device buffer so that
the optimizer step, the L1/L2 terms and the data-parallel all-reduce each touch
one allocation (sized for 288 GB HBM: no per-tensor launches or collectives).
"""
from collections import OrderedDict
from typing import List, Callable, Dict, Optional, Tuple

import numpy as np
import torch

Initializer = Callable[[np.random.Generator, Tuple[int, ...]], np.ndarray]


# ---- initializers (tf.* look-alikes used by model parts and INI files) ------
def random_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None) -> Initializer:
    return lambda rng, shape: (rng.standard_normal(shape) * stddev + mean).astype(np.float32)


def random_uniform_initializer(minval=0.0, maxval=None, seed=None, dtype=None) -> Initializer:
    hi = 1.0 if maxval is None else maxval
    return lambda rng, shape: rng.uniform(minval, hi, size=shape).astype(np.float32)


def zeros_initializer() -> Initializer:
    return lambda rng, shape: np.zeros(shape, np.float32)


def ones_initializer() -> Initializer:
    return lambda rng, shape: np.ones(shape, np.float32)


def constant_initializer(value) -> Initializer:
    return lambda rng, shape: np.full(shape, value, np.float32)


def orthogonal_initializer(gain=1.0) -> Initializer:
    """tf.orthogonal_initializer: QR of a normal matrix flattened to 2-D."""
    def init(rng, shape):
        rows = int(np.prod(shape[:-1]))
        cols = int(shape[-1])
        a = rng.standard_normal((max(rows, cols), min(rows, cols)))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        if rows < cols:
            q = q.T
        return (gain * q.reshape(shape)).astype(np.float32)
    return init


def glorot_uniform_initializer() -> Initializer:
    def init(rng, shape):
        fan_in, fan_out = (shape[0], shape[-1]) if len(shape) > 1 else (shape[0], shape[0])
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    return init


class VarSpec:
    __slots__ = ("name", "shape", "init", "trainable", "offset", "size")

    def __init__(self, name, shape, init, trainable):
        self.name, self.shape, self.init, self.trainable = name, tuple(int(s) for s in shape), init, trainable
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.offset = -1


def find_slot_suffixes(keys, names, preferred):
    """The names under which a checkpoint holds the optimizer's two slots of every variable: ``preferred`` when
    present, else whatever pair ``<var>/<opt>``, ``<var>/<opt>_1`` all of ``names`` carry (a checkpoint is read
    before the trainer that owns the slots has taken its first step and named them)."""
    names = list(names)
    if not names or all(n + preferred[0] in keys and n + preferred[1] in keys for n in names):
        return preferred
    first = names[0] + "/"
    for key in keys:
        if key.startswith(first) and "/" not in key[len(first):] and key + "_1" in keys:
            pair = ("/" + key[len(first):], "/" + key[len(first):] + "_1")
            if all(n + pair[0] in keys and n + pair[1] in keys for n in names):
                return pair
    return preferred


class VariableStore:
    """Named variables as views into flat ``theta`` / ``grad`` / Adam buffers."""

    ALIGN = 4          # floats: every variable starts 16-byte aligned

    def __init__(self, device, seed: Optional[int] = None):
        self.device = torch.device(device)
        self.seed = seed
        self.specs: "OrderedDict[str, VarSpec]" = OrderedDict()
        self.theta: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.adam_m: Optional[torch.Tensor] = None      # the optimizer's two slots per variable (Adam: m, v;
        self.adam_v: Optional[torch.Tensor] = None      # Adadelta: accum, accum_update) ...
        self.slot_suffixes = ("/Adam", "/Adam_1")       # ... and their names in a checkpoint (Optimizer.slot_suffixes)
        self.total = 0
        self.epoch = 0          # bumped by whoever writes the variables through raw pointers (Session.variables_changed)
        self._views: Dict[str, torch.Tensor] = {}
        self._gviews: Dict[str, torch.Tensor] = {}
        # variables the reference's graph creates and no computation reads (declare_checkpoint_only)
        self.checkpoint_only: "OrderedDict[str, VarSpec]" = OrderedDict()
        self._checkpoint_values: Dict[str, np.ndarray] = {}

    # -- declaration ---------------------------------------------------------
    def declare(self, name: str, shape, init: Initializer, trainable: bool = True) -> None:
        if self.theta is not None:
            raise RuntimeError(f"store already finalized; cannot declare {name}")
        spec = VarSpec(name, shape, init, trainable)
        old = self.specs.get(name)
        if old is not None:
            if old.shape != spec.shape:
                raise ValueError(f"variable {name} re-declared with shape {spec.shape} != {old.shape}")
            return
        self.specs[name] = spec

    def declare_checkpoint_only(self, name: str, shape, init: Initializer) -> None:
        """A variable that exists in the reference's checkpoints and that nothing ever reads (TensorFlow's
        ``GRUCell.build`` creates ``gates/kernel`` ... under ``NematusGRUCell``, which overrides ``call`` only:
        nn/ortho_gru_cell.py:57-105).  It takes no room in the flat buffers and no part in a step; checkpoints carry
        it -- written so that a Saver of the reference finds every key it restores, read back when a file has it."""
        if name in self.specs:
            raise ValueError(f"{name} is a variable of the model already")
        self.checkpoint_only.setdefault(name, VarSpec(name, shape, init, True))

    def checkpoint_only_values(self) -> Dict[str, np.ndarray]:
        """name -> array of the checkpoint-only variables: what a checkpoint brought, else their initial values (drawn
        from a generator of their own, so that they leave the model's initialisation stream alone)."""
        rng = np.random.default_rng([0 if self.seed is None else int(self.seed), 0xC0FFEE])
        for name, spec in self.checkpoint_only.items():
            fresh = np.asarray(spec.init(rng, spec.shape), dtype=np.float32).reshape(spec.shape)
            self._checkpoint_values.setdefault(name, fresh)
        return {name: self._checkpoint_values[name] for name in self.checkpoint_only}

    def take_checkpoint_only(self, values) -> List[str]:
        """Keep the checkpoint-only variables found in ``values`` (name -> array); returns their names."""
        taken = []
        for name, spec in self.checkpoint_only.items():
            if name in values:
                arr = np.asarray(values[name], dtype=np.float32)
                if arr.size != spec.size:
                    raise ValueError(f"{name}: checkpoint shape {arr.shape} != {spec.shape}")
                self._checkpoint_values[name] = arr.reshape(spec.shape).copy()
                taken.append(name)
        return taken

    def finalize(self) -> None:
        off = 0
        for spec in self.specs.values():
            spec.offset = off
            off += (spec.size + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        host = np.zeros(off, np.float32)
        rng = np.random.default_rng(self.seed)
        for spec in self.specs.values():
            val = np.asarray(spec.init(rng, spec.shape), dtype=np.float32)
            host[spec.offset:spec.offset + spec.size] = val.reshape(-1)
        self.theta = torch.from_numpy(host).to(self.device)
        self._views = {n: self._view(self.theta, s) for n, s in self.specs.items()}

    def _view(self, flat, spec):
        return flat[spec.offset:spec.offset + spec.size].view(spec.shape)

    # -- access --------------------------------------------------------------
    def __contains__(self, name):
        return name in self.specs

    def __getitem__(self, name) -> torch.Tensor:
        return self._views[name]

    def names(self):
        return list(self.specs)

    def trainable_names(self):
        return [n for n, s in self.specs.items() if s.trainable]

    def offset(self, name) -> int:
        return self.specs[name].offset

    def ensure_grad(self) -> torch.Tensor:
        if self.grad is None:
            self.grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self._gviews = {n: self._view(self.grad, s) for n, s in self.specs.items()}
        return self.grad

    def g(self, name) -> torch.Tensor:
        self.ensure_grad()
        return self._gviews[name]

    def ensure_adam(self):
        if self.adam_m is None:
            self.adam_m = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self.adam_v = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        return self.adam_m, self.adam_v

    # -- (de)serialisation: npz keyed by the TF-style variable names ---------
    def state_dict(self) -> Dict[str, np.ndarray]:
        return {n: self[n].detach().cpu().numpy().copy() for n in self.specs}

    def load_state_dict(self, values: Dict[str, np.ndarray], strict: bool = True) -> None:
        for n, spec in self.specs.items():
            if n not in values:
                if strict:
                    raise KeyError(f"checkpoint lacks variable {n}")
                continue
            arr = np.asarray(values[n], dtype=np.float32)
            if tuple(arr.shape) != spec.shape:
                # size-one axes may differ: scalar () <-> (1,), TF's [1,1,in,out] 1x1 conv filters <-> [in,out]
                squeeze = lambda shape: tuple(d for d in shape if d != 1)
                if arr.size != spec.size or squeeze(arr.shape) != squeeze(spec.shape):
                    raise ValueError(f"{n}: checkpoint shape {arr.shape} != {spec.shape}")
                arr = arr.reshape(spec.shape)
            self[n].copy_(torch.from_numpy(arr).to(self.device))

    def save(self, path: str, fmt: str = "npz", global_step: Optional[int] = None) -> None:
        """``fmt`` "npz" (one file) or "tf" (TensorFlow tensor bundle ``path``.index / .data-00000-of-00001,
        what the reference's tf.train.Saver writes: tf_manager.py:274-277).  Like the Saver over all global
        variables, a checkpoint carries the optimizer state too: the Adam slots (``<var>/Adam``,
        ``<var>/Adam_1``) once an optimizer has created them, and ``global_step`` when given."""
        if fmt == "tf":
            from . import tf_bundle
            tf_bundle.export_store(self, path, global_step=global_step, with_adam=self.adam_m is not None)
            return
        arrays = {k.replace("/", "|"): v for k, v in self.state_dict().items()}
        arrays.update({k.replace("/", "|"): v for k, v in self.checkpoint_only_values().items()})
        if self.adam_m is not None:
            m, v = self.adam_m.cpu().numpy(), self.adam_v.cpu().numpy()
            for name, spec in self.specs.items():
                arrays[(name + self.slot_suffixes[0]).replace("/", "|")] = m[spec.offset:spec.offset + spec.size].reshape(spec.shape)
                arrays[(name + self.slot_suffixes[1]).replace("/", "|")] = v[spec.offset:spec.offset + spec.size].reshape(spec.shape)
        if global_step is not None:
            arrays["global_step"] = np.int64(global_step)
        np.savez(path, **arrays)

    def load(self, path: str, strict: bool = True) -> Dict[str, object]:
        """A TensorFlow checkpoint prefix (``path``.index exists) or an .npz file.  Variables, and -- when the
        file has them -- the Adam slots; returns {"global_step": int or None}."""
        import os
        if os.path.exists(path + ".index"):
            from . import tf_bundle
            return tf_bundle.import_store(self, path, strict)
        if not path.endswith(".npz"):
            path = path + ".npz"
        with np.load(path) as data:
            values = {k.replace("|", "/"): data[k] for k in data.files}
        self.load_state_dict(values, strict)
        self.take_checkpoint_only(values)
        names = [n for n in self.specs if n in values]
        s0, s1 = self.slot_suffixes = find_slot_suffixes(values, names, self.slot_suffixes)
        if names and all(n + s0 in values and n + s1 in values for n in names):
            m, v = self.ensure_adam()
            for n in names:
                spec = self.specs[n]
                m[spec.offset:spec.offset + spec.size] = torch.from_numpy(
                    np.asarray(values[n + s0], np.float32).reshape(-1)).to(m.device)
                v[spec.offset:spec.offset + spec.size] = torch.from_numpy(
                    np.asarray(values[n + s1], np.float32).reshape(-1)).to(v.device)
        step = values.get("global_step")
        return {"global_step": None if step is None else int(step)}
