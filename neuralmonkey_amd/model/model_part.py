"""Model-part protocol (mirror of neuralmonkey/model/{model_part,parameterized,
feedable}.py): dependency collection by attribute name, per-part variable
scopes, feed dictionaries with ``train_mode`` / ``batch_size``."""
from typing import Any, Callable, Dict, Iterable, List, Optional, Set, Tuple

from ..checking import check_constructor_chain
from ..runtime import Placeholder, register_part
from ..variables import Initializer, random_normal_initializer

InitializerSpecs = List[Tuple[str, Callable]]
FeedDict = Dict[Placeholder, Any]


class Feedable:
    """model/feedable.py:17-66."""

    def __init__(self) -> None:
        self.train_mode = Placeholder("train_mode")
        self.batch_size = Placeholder("batch_size")
        self._dataset = None

    def feed_dict(self, dataset, train: bool = True) -> FeedDict:
        return {self.train_mode: train, self.batch_size: len(dataset)}

    def stage_inputs(self, ctx) -> None:
        """Evaluate the fetches that copy this part's fed data into persistent device buffers.  The
        trainer calls it on every feedable before a (possibly HIP-graph replayed) training step: the
        host-to-device copies cannot be part of a captured graph and must happen on every step."""

    @property
    def input_types(self) -> Dict[str, Any]:
        return {}

    @property
    def input_shapes(self) -> Dict[str, Any]:
        return {}

    @property
    def dataset(self):
        if self._dataset is None:
            raise RuntimeError("Getting dataset before registering it.")
        return self._dataset

    def register_input(self, dataset) -> None:
        self._dataset = dataset


class Parameterized:
    """model/parameterized.py:15-125: named scope, variable sharing via
    ``reuse``, per-variable initializer overrides, per-part checkpoints."""

    def __init__(self, name: str, reuse: "Parameterized" = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        check_constructor_chain(self)                       # the reference's check_argument_types() calls
        self._name = name
        self._save_checkpoint = save_checkpoint
        self._load_checkpoint = load_checkpoint
        self._reuse = reuse is not None
        self._default_initializer: Initializer = random_normal_initializer(stddev=0.001)
        self._initializer_overrides: Dict[str, Callable] = {}
        if reuse is not None:
            if initializers is not None:
                raise ValueError("Cannot use initializers in model part '{}' that reuses variables "
                                 "from '{}'.".format(name, reuse.name))
            self._scope = reuse._scope                      # pylint: disable=protected-access
            self._initializer_overrides = reuse._initializer_overrides
        else:
            self._scope = name
            for var_name, init in (initializers or []):
                self._initializer_overrides[f"{self._scope}/{var_name}"] = init
        register_part(self)

    @property
    def name(self) -> str:
        return self._name

    @property
    def scope(self) -> str:
        return self._scope

    @property
    def shares_variables(self) -> bool:
        """Do several model parts live in this part's variable scope (``reuse=``)?  Their gradient passes
        then ADD into the same gradient tensors (tf.gradients sums the contributions of every use of a
        variable); a part alone in its scope may overwrite (the buffer is zeroed at the start of the step,
        but an overwrite saves a read of the weight-sized gradient)."""
        from ..runtime import registered_parts
        return sum(1 for p in registered_parts() if getattr(p, "_scope", None) == self._scope) > 1

    def __str__(self) -> str:
        return self.name

    # -- variables -------------------------------------------------------------
    def set_default_initializer(self, init: Initializer) -> None:
        self._default_initializer = init

    def var_name(self, local: str) -> str:
        return f"{self._scope}/{local}"

    def declare(self, store, local: str, shape, initializer: Optional[Initializer] = None,
                trainable: bool = True) -> None:
        full = self.var_name(local)
        init = self._initializer_overrides.get(full, initializer or self._default_initializer)
        if callable(init) and not _is_rng_init(init):
            init = _wrap_foreign_init(init)
        store.declare(full, shape, init, trainable)

    def declare_checkpoint_only(self, store, local: str, shape, initializer: Optional[Initializer] = None) -> None:
        """A variable of this part's scope that only checkpoints know (VariableStore.declare_checkpoint_only)."""
        full = self.var_name(local)
        init = self._initializer_overrides.get(full, initializer or self._default_initializer)
        if callable(init) and not _is_rng_init(init):
            init = _wrap_foreign_init(init)
        store.declare_checkpoint_only(full, shape, init)

    def declare_variables(self, store) -> None:
        """Declare every variable of this part in ``store`` (overridden)."""

    def var(self, ctx_or_store, local: str):
        store = getattr(ctx_or_store, "store", ctx_or_store)
        return store[self.var_name(local)]

    def variable_names(self, store) -> List[str]:
        prefix = self._scope + "/"
        return [n for n in store.names() if n.startswith(prefix)]

    # -- per-part checkpoints (parameterized.py:101-125) -------------------------
    def save(self, session) -> None:
        """The variables of this part's scope as a TensorFlow tensor bundle (what the reference's
        per-part ``tf.train.Saver(var_list=...)`` writes: ``<path>.index`` + ``<path>.data-00000-of-00001``),
        or one .npz file when the path says so."""
        if not self._save_checkpoint:
            return
        import numpy as np
        vals = {n: session.store[n].detach().cpu().numpy() for n in self.variable_names(session.store)}
        if self._save_checkpoint.endswith(".npz"):
            np.savez(self._save_checkpoint, **{n.replace("/", "|"): v for n, v in vals.items()})
            return
        from .. import tf_bundle
        tf_bundle.write_bundle(self._save_checkpoint, {
            n: np.asarray(v, np.float32).reshape(tf_bundle.tf_shape(n, v.shape)) for n, v in vals.items()})

    def load(self, session) -> None:
        """Restore this part's variables from ``load_checkpoint`` -- a bundle the reference (or ``save``)
        wrote, e.g. a pre-trained encoder; every variable of the scope must be there, as with
        ``Saver.restore``."""
        if not self._load_checkpoint:
            return
        import os
        import numpy as np
        path = self._load_checkpoint
        mine = self.variable_names(session.store)
        if os.path.exists(path + ".index"):
            from .. import tf_bundle
            bundle = tf_bundle.read_bundle(path)
            vals = {}
            for n in mine:
                if n not in bundle:
                    raise KeyError("checkpoint '{}' lacks variable '{}' of model part '{}'".format(path, n, self.name))
                shape = session.store.specs[n].shape
                if [d for d in bundle[n].shape if d != 1] != [d for d in shape if d != 1]:
                    raise ValueError("shape of '{}' in checkpoint '{}' is {}, the model part declares {}"
                                     .format(n, path, bundle[n].shape, shape))
                vals[n] = np.asarray(bundle[n], np.float32).reshape(shape)
        else:
            if not path.endswith(".npz") and not os.path.exists(path):
                path += ".npz"
            with np.load(path) as data:
                vals = {k.replace("|", "/"): data[k] for k in data.files}
            vals = {k: v for k, v in vals.items() if k in set(mine)}
        session.store.load_state_dict(vals, strict=False)


def _is_rng_init(init) -> bool:
    import inspect
    try:
        return len(inspect.signature(init).parameters) == 2
    except (TypeError, ValueError):
        return False


def _wrap_foreign_init(init):
    """Accept ``init(shape)``-style callables from user configs."""
    return lambda rng, shape: init(shape)


class GenericModelPart:
    """model/model_part.py:9-83: recursive dependency collection."""

    @property
    def dependencies(self) -> List[str]:
        return ["encoder", "parent_decoder", "input_sequence", "attentions", "encoders"]

    def get_dependencies(self) -> Tuple[Set[Feedable], Set[Parameterized]]:
        feedables: Set[Feedable] = set()
        parameterizeds: Set[Parameterized] = set()
        if isinstance(self, Feedable):
            feedables.add(self)
        if isinstance(self, Parameterized):
            parameterizeds.add(self)
        for attr in self.dependencies:
            val = getattr(self, attr, None)
            if val is None:
                continue
            if isinstance(val, GenericModelPart):
                deps = [val]
            elif isinstance(val, Iterable) and not isinstance(val, (str, bytes)):
                deps = [v for v in val if isinstance(v, GenericModelPart)]
            else:
                deps = []
            for dep in deps:
                feeds, params = dep.get_dependencies()
                feedables |= feeds
                parameterizeds |= params
        return feedables, parameterizeds


class ModelPart(Parameterized, GenericModelPart, Feedable):
    """model/model_part.py:86-103."""

    def __init__(self, name: str, reuse: "ModelPart" = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        Parameterized.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        GenericModelPart.__init__(self)
        Feedable.__init__(self)
