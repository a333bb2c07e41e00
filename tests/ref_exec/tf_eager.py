"""NumPy-eager stand-in for the slice of the TensorFlow 1.12 API that the reference's hot-path modules call.

TEST INFRASTRUCTURE ONLY.  Nothing under ``neuralmonkey_amd/`` or in ``bench.py``'s timed region may import this
file (``tests/test_abi.py`` enforces it).  Its single purpose: ``tests/golden/make_reference_exec_golden.py`` installs
it as ``sys.modules["tensorflow"]``, imports the model parts FROM ``/root/reference`` and runs them on seeded inputs,
so that the committed fixtures under ``tests/golden/ref_exec/`` hold numbers produced by the reference's OWN Python
(``attention/feed_forward.py``, ``decoders/{autoregressive,decoder,beam_search_decoder,transformer}.py``,
``attention/scaled_dot_product.py``, ``encoders/{recurrent,transformer}.py``, ``tf_utils.py``, ``nn/*``,
``decoders/{output,encoder}_projection.py``, ``vocabulary.py``, ``runners/beamsearch_runner.py``) and the oracle
(``oracle/*.py``) is pinned to them by ``tests/test_reference_exec.py``.

What is emulated, and how faithfully:

* **Eager, not a graph.**  A ``Tensor`` wraps a NumPy array; every ``tf.*`` call computes at once in the array's
  dtype (float32 stays float32: the same IEEE operations TF's CPU kernels perform element-wise; matmul accumulation
  order and libm ulps differ -- fixtures are compared at 1e-6).  ``tf.placeholder`` returns a tensor whose value is
  looked up in the active feed dictionary (``feeding(fd)``) when it is first USED; the reference's ``@tensor``
  properties are lazy, so feeding before the first access of a computed property is the whole protocol.
  ``tf.while_loop`` is a Python loop (condition evaluated before every iteration on the previous iteration's loop
  state, as TF does), ``tf.cond`` a Python branch.
* **Variable scopes follow TF 1.12's rules** (``variable_scope.py``): name nesting, re-entry through a captured scope
  object (absolute name, the object's reuse / initializer), reuse inheritance, ``AUTO_REUSE``, ``default_name``
  uniquification through the per-store scope counts INCLUDING their reset when a scope is left -- which is what makes
  an unnamed ``tf.layers.dense`` inside a loop body resolve to ``dense`` in every iteration and in both the train and
  the runtime loop.  ``tf.layers`` / RNN cells open their scopes the way ``layers/base.py`` and ``rnn_cell_impl.py``
  do, so the variable NAMES that appear are the checkpoint names of the reference.
* **TF-internal arithmetic is restated here, not executed** (there is no TensorFlow in this image):
  ``GRUCell`` / ``LSTMCell`` / ``dynamic_rnn`` / ``bidirectional_dynamic_rnn`` / ``sequence_loss`` /
  ``softmax_cross_entropy`` / ``top_k`` tie order / ``layers.dense`` follow SURVEY.md section 9.  Everything the
  reference AUTHORS wrote on top of those ops is executed from their files.
* Values of variables come from ``VARIABLE_FACTORY(name, shape, dtype, initializer)`` (set by the fixture generator:
  seeded by the variable's name), not from TF's initializers' random streams.
"""
import builtins as _builtins
import contextlib
import copy
import math
import re
import sys
import types
import zlib

import numpy as np

_range = _builtins.range      # this module defines tf.range / tf.abs / tf.pow / tf.shape ... under their TF names

# --------------------------------------------------------------------------------------------------------------------
# dtypes and shapes
# --------------------------------------------------------------------------------------------------------------------


class DType:
    def __init__(self, name, np_dtype):
        self.name = name
        self._np = np_dtype

    @property
    def as_numpy_dtype(self):
        return self._np

    @property
    def base_dtype(self):
        return self

    @property
    def is_floating(self):
        return self._np in (np.float32, np.float64, np.float16)

    @property
    def is_integer(self):
        return self._np in (np.int32, np.int64)

    def __repr__(self):
        return "tf." + self.name

    def __eq__(self, other):
        return isinstance(other, DType) and other.name == self.name

    def __hash__(self):
        return hash(self.name)


float32 = DType("float32", np.float32)
float64 = DType("float64", np.float64)
int32 = DType("int32", np.int32)
int64 = DType("int64", np.int64)
bool_ = DType("bool", np.bool_)
string = DType("string", np.object_)
_DTYPES = [float32, float64, int32, int64, bool_, string]


def _np_dtype(dt):
    if dt is None:
        return None
    if isinstance(dt, DType):
        return dt.as_numpy_dtype
    return np.dtype(dt).type


def _tf_dtype(np_dt):
    np_dt = np.dtype(np_dt)
    if np_dt.kind in "OUS":
        return string
    for d in _DTYPES:
        if d is not string and np.dtype(d.as_numpy_dtype) == np_dt:
            return d
    raise TypeError("no tf dtype for {}".format(np_dt))


class Dimension:
    def __init__(self, value):
        self._value = None if value is None else int(value)

    @property
    def value(self):
        return self._value

    def __int__(self):
        return self._value

    __index__ = __int__

    def __eq__(self, other):
        o = other.value if isinstance(other, Dimension) else other
        return self._value == o

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash(self._value)

    def __repr__(self):
        return "Dimension({})".format(self._value)

    def __str__(self):
        return "?" if self._value is None else str(self._value)

    def __mul__(self, other):
        return Dimension(self._value * int(other))

    __rmul__ = __mul__

    def __add__(self, other):
        return Dimension(self._value + int(other))

    __radd__ = __add__


class TensorShape:
    def __init__(self, dims):
        if isinstance(dims, TensorShape):
            dims = dims.as_list()
        self._dims = None if dims is None else [d if isinstance(d, Dimension) else Dimension(d) for d in dims]

    @property
    def dims(self):
        return self._dims

    @property
    def ndims(self):
        return None if self._dims is None else len(self._dims)

    def as_list(self):
        return [d.value for d in self._dims]

    def __len__(self):
        return len(self._dims)

    def __iter__(self):
        return iter(self._dims)

    def __getitem__(self, key):
        if isinstance(key, slice):
            return TensorShape(self._dims[key])
        return self._dims[key]

    def __eq__(self, other):
        try:
            return self.as_list() == TensorShape(other).as_list()
        except TypeError:
            return NotImplemented

    def __repr__(self):
        return "TensorShape({})".format(self._dims)


# --------------------------------------------------------------------------------------------------------------------
# tensors
# --------------------------------------------------------------------------------------------------------------------
_FEEDS = [{}]       # stack of active feed dictionaries: id(placeholder) -> numpy value


@contextlib.contextmanager
def feeding(feed_dict):
    """Make ``feed_dict`` ({placeholder: value}) the active feeds (what Session.run(feed_dict=...) does)."""
    frame = dict(_FEEDS[-1])
    for key, val in feed_dict.items():
        # a key may be a nested structure of tensors fed with the same structure of values (TF flattens both)
        keys, vals = (nest.flatten(key), nest.flatten(val)) if nest.is_sequence(key) else ([key], [val])
        if len(keys) != len(vals):
            raise ValueError("feed structure mismatch for {!r}".format(key))
        for k, v in zip(keys, vals):
            if not isinstance(k, Tensor):
                raise TypeError("feed key is not a tensor: {!r}".format(k))
            frame[id(k)] = k._coerce_feed(v)
    _FEEDS.append(frame)
    try:
        yield
    finally:
        _FEEDS.pop()


def _convert(x, dtype=None):
    """``ops.convert_to_tensor`` on plain values: python float -> float32, int -> int32, bool, str -> object."""
    if isinstance(x, Tensor):
        v = x.numpy()
        if dtype is not None and v.dtype != np.dtype(_np_dtype(dtype)):
            raise TypeError("tensor of {} where {} is expected".format(v.dtype, dtype))
        return v
    if isinstance(x, Dimension):
        x = x.value
    if isinstance(x, np.ndarray) or isinstance(x, np.generic):
        v = np.asarray(x)
        if dtype is not None:
            v = v.astype(_np_dtype(dtype))
        return v
    if isinstance(x, (list, tuple)):
        if any(isinstance(e, (Tensor, Dimension)) for e in _flat_list(x)):
            x = _map_list(lambda e: e.numpy() if isinstance(e, Tensor) else (e.value if isinstance(e, Dimension) else e), x)
            v = np.asarray(x)
            if dtype is not None:
                v = v.astype(_np_dtype(dtype))
            elif v.dtype == np.float64 and not any(isinstance(e, np.ndarray) and e.dtype == np.float64
                                                   for e in _flat_list(x)):
                v = v.astype(np.float32)
            elif v.dtype == np.int64 and not any(isinstance(e, (np.ndarray, np.generic)) and e.dtype == np.int64
                                                 for e in _flat_list(x)):
                v = v.astype(np.int32)
            return v
    v = np.asarray(x)
    if dtype is not None:
        return v.astype(_np_dtype(dtype)) if v.dtype.kind not in "US" else v.astype(object)
    if v.dtype == np.float64:
        return v.astype(np.float32)
    if v.dtype == np.int64:
        return v.astype(np.int32)
    if v.dtype.kind in "US":
        return v.astype(object)
    return v


def _flat_list(x):
    for e in x:
        if isinstance(e, (list, tuple)):
            yield from _flat_list(e)
        else:
            yield e


def _map_list(fn, x):
    return [(_map_list(fn, e) if isinstance(e, (list, tuple)) else fn(e)) for e in x]


def _binary_operands(a, b):
    """Python scalars adopt the tensor operand's dtype (what TF's operator overloads do via convert_to_tensor)."""
    def adopt(plain, dt):
        if isinstance(plain, (np.ndarray, np.generic)):
            return np.asarray(plain)
        if dt.kind in "iu" and isinstance(plain, float) and plain != int(plain):
            raise TypeError("float constant {} against an integer tensor".format(plain))
        return _convert(plain, dtype=dt)
    if isinstance(a, Tensor) and not isinstance(b, Tensor):
        av = a.numpy()
        return av, adopt(b, av.dtype)
    if isinstance(b, Tensor) and not isinstance(a, Tensor):
        bv = b.numpy()
        return adopt(a, bv.dtype), bv
    return _convert(a), _convert(b)


def _same_dtype(av, bv, what):
    if av.dtype != bv.dtype:
        raise TypeError("{}: operands of {} and {} (TF does not promote)".format(what, av.dtype, bv.dtype))


class Tensor:
    __array_priority__ = 1000

    def __init__(self, value, name=None):
        self._value = None if value is None else np.asarray(value)
        self.name = name or "Tensor:0"

    # -- value -----------------------------------------------------------------------------------------------------
    def numpy(self):
        feeds = _FEEDS[-1]
        if feeds and id(self) in feeds:          # Session.run(feed_dict=...) may override ANY tensor, not only placeholders
            return feeds[id(self)]
        return self._value

    def _coerce_feed(self, v):
        mine = self._value
        v = np.asarray(v)
        return v.astype(mine.dtype) if mine is not None and mine.dtype.kind != "O" else v

    # -- static information ------------------------------------------------------------------------------------------
    @property
    def shape(self):
        return TensorShape(self.numpy().shape)

    def get_shape(self):
        return self.shape

    def set_shape(self, shape):
        dims = TensorShape(shape).as_list()
        mine = self.numpy().shape
        if len(dims) != len(mine) or any(d is not None and d != m for d, m in zip(dims, mine)):
            raise ValueError("set_shape {} on a tensor of shape {}".format(dims, mine))

    @property
    def dtype(self):
        return _tf_dtype(self.numpy().dtype)

    @property
    def op(self):
        return types.SimpleNamespace(name=self.name.split(":")[0])

    def eval(self, feed_dict=None, session=None):
        return self.numpy()

    def __repr__(self):
        return "<tf_eager.Tensor {} shape={} dtype={}>".format(self.name, self.numpy().shape, self.numpy().dtype)

    def __bool__(self):
        raise TypeError("Using a `tf.Tensor` as a Python `bool` is not allowed (graph-mode TF raises here too).")

    def __iter__(self):
        v = self.numpy()
        if v.ndim == 0:
            raise TypeError("Tensor objects are only iterable when they have at least one dimension")
        return iter([Tensor(v[i]) for i in _range(v.shape[0])])

    def __len__(self):
        raise TypeError("len() of a tf.Tensor")

    __hash__ = object.__hash__          # TF1 tensors hash / compare by identity (they are feed_dict keys)

    def __eq__(self, other):
        return self is other

    def __ne__(self, other):
        return self is not other

    # -- operators ------------------------------------------------------------------------------------------------
    def __add__(self, o):
        return add(self, o)

    def __radd__(self, o):
        return add(o, self)

    def __sub__(self, o):
        return subtract(self, o)

    def __rsub__(self, o):
        return subtract(o, self)

    def __mul__(self, o):
        return multiply(self, o)

    def __rmul__(self, o):
        return multiply(o, self)

    def __truediv__(self, o):
        return truediv(self, o)

    def __rtruediv__(self, o):
        return truediv(o, self)

    def __floordiv__(self, o):
        a, b = _binary_operands(self, o)
        return Tensor(np.floor_divide(a, b))

    def __mod__(self, o):
        return mod(self, o)

    def __pow__(self, o):
        a, b = _binary_operands(self, o)
        _same_dtype(a, b, "pow")
        return Tensor(np.power(a, b))

    def __rpow__(self, o):
        a, b = _binary_operands(o, self)
        return Tensor(np.power(a, b))

    def __neg__(self):
        return Tensor(-self.numpy())

    def __abs__(self):
        return Tensor(np.abs(self.numpy()))

    def __lt__(self, o):
        return less(self, o)

    def __le__(self, o):
        a, b = _binary_operands(self, o)
        return Tensor(a <= b)

    def __gt__(self, o):
        a, b = _binary_operands(self, o)
        return Tensor(a > b)

    def __ge__(self, o):
        a, b = _binary_operands(self, o)
        return Tensor(a >= b)

    def __invert__(self):
        return logical_not(self)

    def __and__(self, o):
        return logical_and(self, o)

    def __or__(self, o):
        return logical_or(self, o)

    def __getitem__(self, key):
        def conv(k):
            if isinstance(k, Tensor):
                kv = k.numpy()
                if kv.ndim != 0:
                    raise TypeError("only scalar tensors index a tensor (strided_slice)")
                return int(kv)
            if isinstance(k, Dimension):
                return k.value
            if isinstance(k, slice):
                return slice(conv(k.start) if k.start is not None else None,
                             conv(k.stop) if k.stop is not None else None,
                             conv(k.step) if k.step is not None else None)
            return k
        if isinstance(key, tuple):
            key = tuple(conv(k) for k in key)
        else:
            key = conv(key)
        return Tensor(self.numpy()[key])


class Placeholder(Tensor):
    def __init__(self, dtype, shape=None, name=None, default=None):
        super().__init__(None, name=(name or "Placeholder") + ":0")
        self._dtype = dtype
        self._static = shape
        self._default = default

    def _coerce_feed(self, v):
        if self._dtype == string:
            return np.asarray(v, dtype=object)
        return np.asarray(v).astype(self._dtype.as_numpy_dtype)

    def numpy(self):
        feeds = _FEEDS[-1]
        if id(self) in feeds:
            return feeds[id(self)]
        if self._default is not None:
            return _convert(self._default)
        raise RuntimeError("placeholder {} is used but was not fed".format(self.name))

    @property
    def dtype(self):
        return self._dtype


class Variable(Tensor):
    def __init__(self, value, name, trainable=True):
        super().__init__(value, name=name + ":0")
        self.trainable = trainable

    def initialized_value(self):
        return self

    def read_value(self):
        return Tensor(self.numpy())

    def assign(self, value):
        self._value = _convert(value).astype(self._value.dtype)
        return self

    @property
    def initializer(self):
        return no_op()


def _t(x):
    return x if isinstance(x, Tensor) else Tensor(_convert(x))


def _v(x, dtype=None):
    return _convert(x, dtype)


def _int(x):
    """A Python int out of an int / Dimension / scalar int tensor (shape and axis arguments)."""
    if isinstance(x, Tensor):
        v = x.numpy()
        if v.ndim != 0:
            raise TypeError("expected a scalar")
        return int(v)
    if isinstance(x, Dimension):
        return x.value
    return int(x)


def _shape_arg(shape):
    if isinstance(shape, Tensor):
        return tuple(int(d) for d in shape.numpy().reshape(-1))
    if isinstance(shape, TensorShape):
        return tuple(shape.as_list())
    if isinstance(shape, (int, np.integer, Dimension)):
        return (_int(shape),)
    return tuple(_int(d) for d in shape)


# --------------------------------------------------------------------------------------------------------------------
# variable scopes (tensorflow/python/ops/variable_scope.py of TF 1.12)
# --------------------------------------------------------------------------------------------------------------------
class _AutoReuse:
    def __repr__(self):
        return "tf.AUTO_REUSE"

    def __bool__(self):
        return True


AUTO_REUSE = _AutoReuse()


class VariableScope:
    def __init__(self, reuse, name="", initializer=None, custom_getter=None):
        self._name = name
        self._reuse = reuse
        self._initializer = initializer
        self._custom_getter = custom_getter

    @property
    def name(self):
        return self._name

    @property
    def original_name_scope(self):
        return self._name + "/" if self._name else ""

    @property
    def reuse(self):
        return self._reuse

    @property
    def initializer(self):
        return self._initializer

    def set_initializer(self, initializer):
        self._initializer = initializer

    def reuse_variables(self):
        self._reuse = True

    def global_variables(self):
        return [v for n, v in _STORE.vars.items() if n.startswith(self._name + "/")]

    trainable_variables = global_variables


class _Store:
    def __init__(self):
        self.vars = {}              # full name -> Variable   (creation order preserved)
        self.scope = VariableScope(None, "")
        self.counts = {}            # variable_scopes_count

    def open_scope(self, name):
        self.counts[name] = self.counts.get(name, 0) + 1

    def close_subscopes(self, name):
        for k in list(self.counts):
            if name is None or k.startswith(name + "/"):
                self.counts[k] = 0

    def count(self, name):
        return self.counts.get(name, 0)


_STORE = _Store()
VARIABLE_FACTORY = None         # callable(name, shape, np_dtype, initializer) -> ndarray, set by the fixture generator
CREATION_LOG = []               # (full variable name, shape) in creation order


def reset_default_graph():
    global _STORE
    _STORE = _Store()
    del CREATION_LOG[:]
    _LAYER_UIDS.clear()


@contextlib.contextmanager
def session_store(store=None, factory=None):
    """Run the enclosed code against another variable store (and variable factory): one per tf.Session that holds
    its own values of the same variable names (ensembles: runners/beamsearch_runner.py, tf_manager.py:158-185)."""
    global _STORE, VARIABLE_FACTORY
    saved = (_STORE, VARIABLE_FACTORY)
    _STORE = store if store is not None else _Store()
    if factory is not None:
        VARIABLE_FACTORY = factory
    try:
        yield _STORE
    finally:
        _STORE, VARIABLE_FACTORY = saved


def get_variable_scope():
    return _STORE.scope


def _unique_scope_name(prefix):
    cur = _STORE.scope.name
    name = cur + "/" + prefix if cur else prefix
    if _STORE.count(name) == 0:
        return prefix
    idx = 1
    while _STORE.count(name + "_%d" % idx) > 0:
        idx += 1
    return prefix + "_%d" % idx


class variable_scope:       # noqa: N801 (TF's spelling)
    def __init__(self, name_or_scope, default_name=None, values=None, initializer=None, reuse=None,
                 custom_getter=None, auxiliary_name_scope=True, dtype=None, regularizer=None, caching_device=None,
                 partitioner=None, use_resource=None, constraint=None):
        if name_or_scope is None and default_name is None:
            raise TypeError("If default_name is None then name_or_scope is required")
        if reuse is False:
            reuse = None            # "We don't allow non-inheriting scopes, False = None here."
        self._name_or_scope = name_or_scope
        self._default_name = default_name
        self._initializer = initializer
        self._reuse = reuse
        self._custom_getter = custom_getter

    def __enter__(self):
        store = _STORE
        self._old = store.scope
        nos = self._name_or_scope
        if nos is None:
            nos = _unique_scope_name(self._default_name)
        if isinstance(nos, VariableScope):
            self._entered_by_object = True
            self._new_name = nos.name
            store.open_scope(self._new_name)
            self._old_counts = copy.copy(store.counts)
            new = VariableScope(nos.reuse if not self._reuse else self._reuse, self._new_name,
                                initializer=nos.initializer, custom_getter=nos._custom_getter)
        else:
            self._entered_by_object = False
            self._new_name = self._old.name + "/" + nos if self._old.name else nos
            new = VariableScope(self._reuse or self._old.reuse, self._new_name,
                                initializer=self._old.initializer, custom_getter=self._old._custom_getter)
            store.open_scope(self._new_name)
        if self._initializer is not None:
            new.set_initializer(self._initializer)
        store.scope = new
        return new

    def __exit__(self, *exc):
        store = _STORE
        if self._entered_by_object:
            store.counts = self._old_counts
        else:
            store.close_subscopes(self._new_name)
        store.scope = self._old
        return False


@contextlib.contextmanager
def name_scope(name, default_name=None, values=None):
    yield name


def get_variable(name, shape=None, dtype=None, initializer=None, regularizer=None, trainable=True,
                 collections=None, **_):
    scope = _STORE.scope
    full = scope.name + "/" + name if scope.name else name
    reuse = scope.reuse
    if full in _STORE.vars:
        if reuse is None or reuse is False:
            raise ValueError("Variable {} already exists, disallowed. Did you mean to set reuse=True or "
                             "reuse=tf.AUTO_REUSE in VarScope?".format(full))
        var = _STORE.vars[full]
        if shape is not None and tuple(_shape_arg(shape)) != var.numpy().shape:
            raise ValueError("Trying to share variable {}, but specified shape {} and found shape {}."
                             .format(full, _shape_arg(shape), var.numpy().shape))
        return var
    if reuse is True:
        raise ValueError("Variable {} does not exist, or was not created with tf.get_variable().".format(full))
    if initializer is None:
        initializer = scope.initializer
    np_dt = _np_dtype(dtype) or np.float32
    if isinstance(initializer, Tensor) or isinstance(initializer, np.ndarray):
        value = _convert(initializer)
        shp = value.shape
    else:
        shp = _shape_arg(shape)
        if initializer is None:
            initializer = glorot_uniform_initializer() if np.dtype(np_dt).kind == "f" else zeros_initializer()
        value = None
    if VARIABLE_FACTORY is not None:
        value = np.asarray(VARIABLE_FACTORY(full, shp, np_dt, initializer), dtype=np_dt)
    elif value is None:
        value = _convert(initializer(list(shp), dtype=_tf_dtype(np_dt))).astype(np_dt)
    if value.shape != tuple(shp):
        raise ValueError("initial value of {} has shape {} instead of {}".format(full, value.shape, shp))
    var = Variable(value, full, trainable=trainable)
    _STORE.vars[full] = var
    CREATION_LOG.append((full, tuple(shp)))
    return var


def global_variables():
    return list(_STORE.vars.values())


def trainable_variables():
    return [v for v in _STORE.vars.values() if v.trainable]


def get_collection(key, scope=None):
    if key in (GraphKeys.GLOBAL_VARIABLES, GraphKeys.TRAINABLE_VARIABLES):
        vs = trainable_variables() if key == GraphKeys.TRAINABLE_VARIABLES else global_variables()
        if scope is not None:
            vs = [v for v in vs if re.match(scope, v.name)]
        return vs
    return []


class GraphKeys:
    GLOBAL_VARIABLES = "variables"
    TRAINABLE_VARIABLES = "trainable_variables"
    UPDATE_OPS = "update_ops"


# --------------------------------------------------------------------------------------------------------------------
# initializers (only asked for values when no VARIABLE_FACTORY is installed)
# --------------------------------------------------------------------------------------------------------------------
class _Initializer:
    kind = "?"

    def __init__(self, **kw):
        self.kw = kw

    def __call__(self, shape, dtype=None, partition_info=None):
        rng = np.random.default_rng(zlib.crc32(repr((self.kind, sorted(self.kw.items()), tuple(shape))).encode()))
        return Tensor(self._make(rng, tuple(_shape_arg(shape)), _np_dtype(dtype) or np.float32))

    def __repr__(self):
        return "<{} {}>".format(self.kind, self.kw)


class zeros_initializer(_Initializer):      # noqa: N801
    kind = "zeros"

    def _make(self, rng, shape, dt):
        return np.zeros(shape, dt)


class ones_initializer(_Initializer):       # noqa: N801
    kind = "ones"

    def _make(self, rng, shape, dt):
        return np.ones(shape, dt)


class constant_initializer(_Initializer):   # noqa: N801
    kind = "constant"

    def __init__(self, value=0, dtype=None):
        super().__init__(value=value)

    def _make(self, rng, shape, dt):
        return np.full(shape, self.kw["value"], dt)


class random_normal_initializer(_Initializer):      # noqa: N801
    kind = "random_normal"

    def __init__(self, mean=0.0, stddev=1.0, seed=None, dtype=None):
        super().__init__(mean=mean, stddev=stddev)

    def _make(self, rng, shape, dt):
        return rng.normal(self.kw["mean"], self.kw["stddev"], shape).astype(dt)


class random_uniform_initializer(_Initializer):     # noqa: N801
    kind = "random_uniform"

    def __init__(self, minval=0.0, maxval=None, seed=None, dtype=None):
        super().__init__(minval=minval, maxval=1.0 if maxval is None else maxval)

    def _make(self, rng, shape, dt):
        return rng.uniform(self.kw["minval"], self.kw["maxval"], shape).astype(dt)


class variance_scaling_initializer(_Initializer):   # noqa: N801
    kind = "variance_scaling"

    def __init__(self, scale=1.0, mode="fan_in", distribution="truncated_normal", seed=None, dtype=None):
        super().__init__(scale=scale, mode=mode, distribution=distribution)

    def _make(self, rng, shape, dt):
        fan_in = shape[0] if len(shape) == 2 else (int(np.prod(shape[:-1])) if shape else 1)
        fan_out = shape[-1] if shape else 1
        n = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2.0}[self.kw["mode"]]
        s = self.kw["scale"] / max(1.0, n)
        if self.kw["distribution"] == "uniform":
            lim = math.sqrt(3.0 * s)
            return rng.uniform(-lim, lim, shape).astype(dt)
        return (rng.normal(0, 1, shape) * math.sqrt(s)).astype(dt)


def glorot_uniform_initializer(seed=None, dtype=None):
    return variance_scaling_initializer(scale=1.0, mode="fan_avg", distribution="uniform")


class orthogonal_initializer(_Initializer):     # noqa: N801
    kind = "orthogonal"

    def __init__(self, gain=1.0, seed=None, dtype=None):
        super().__init__(gain=gain)

    def _make(self, rng, shape, dt):
        rows, cols = int(np.prod(shape[:-1])), shape[-1]
        a = rng.normal(0, 1, (max(rows, cols), min(rows, cols)))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        if rows < cols:
            q = q.T
        return (self.kw["gain"] * q.reshape(shape)).astype(dt)


# --------------------------------------------------------------------------------------------------------------------
# element-wise and shape ops
# --------------------------------------------------------------------------------------------------------------------
def convert_to_tensor(value, dtype=None, name=None, preferred_dtype=None):
    if isinstance(value, Tensor) and dtype is None:
        return value
    return Tensor(_convert(value, dtype))


def constant(value, dtype=None, shape=None, name=None):
    v = _convert(value, dtype)
    if shape is not None:
        v = np.broadcast_to(v, _shape_arg(shape)).copy()
    return Tensor(v)


def identity(x, name=None):
    return _t(x)


def stop_gradient(x, name=None):
    return _t(x)


def Print(x, data, message=None, **_):      # noqa: N802
    return _t(x)


def assert_equal(x, y, data=None, summarize=None, message=None, name=None):
    a, b = _binary_operands(x, y)
    if not np.array_equal(a, b):
        raise AssertionError("tf.assert_equal failed: {} {!r} vs {!r}".format(message or "", a, b))
    return no_op()


def no_op(name=None):
    return types.SimpleNamespace(name=name or "NoOp", run=lambda *a, **k: None)


def group(*ops, **_):
    return no_op()


@contextlib.contextmanager
def control_dependencies(ops):
    yield


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name)


def placeholder_with_default(input, shape, name=None):      # noqa: A002
    if isinstance(input, Tensor):
        default = input
        dt = input.dtype
    else:
        default = Tensor(_convert(input))
        dt = default.dtype
    p = Placeholder(dt, shape, name or "PlaceholderWithDefault", default=default)
    return p


def _binop(fn, what, same=True):
    def op(x, y, name=None):
        a, b = _binary_operands(x, y)
        if same:
            _same_dtype(a, b, what)
        return Tensor(fn(a, b))
    op.__name__ = what
    return op


add = _binop(np.add, "add")
subtract = _binop(np.subtract, "subtract")
multiply = _binop(np.multiply, "multiply")
maximum = _binop(np.maximum, "maximum")
minimum = _binop(np.minimum, "minimum")
less = _binop(np.less, "less")
equal = _binop(np.equal, "equal")
not_equal = _binop(np.not_equal, "not_equal")
greater = _binop(np.greater, "greater")
logical_and = _binop(np.logical_and, "logical_and")
logical_or = _binop(np.logical_or, "logical_or")


def truediv(x, y, name=None):
    a, b = _binary_operands(x, y)
    _same_dtype(a, b, "truediv")
    if a.dtype.kind in "iu":
        return Tensor(a.astype(np.float64) / b.astype(np.float64))
    return Tensor(a / b)


def div(x, y, name=None):
    """tf.div: Python-2 style -- floor division for integers, true division for floats."""
    a, b = _binary_operands(x, y)
    _same_dtype(a, b, "div")
    if a.dtype.kind in "iu":
        return Tensor(np.floor_divide(a, b))
    return Tensor(a / b)


def mod(x, y, name=None):
    a, b = _binary_operands(x, y)
    _same_dtype(a, b, "mod")
    return Tensor(np.mod(a, b))         # floormod, like tf.mod


floormod = mod


def negative(x, name=None):
    return Tensor(-_v(x))


def logical_not(x, name=None):
    return Tensor(np.logical_not(_v(x)))


def _unary(fn):
    def op(x, name=None):
        return Tensor(fn(_v(x)))
    return op


def _sigmoid(v):
    return (1.0 / (1.0 + np.exp(-v))).astype(v.dtype)


tanh = _unary(np.tanh)
exp = _unary(np.exp)
log = _unary(np.log)
sin = _unary(np.sin)
cos = _unary(np.cos)
sqrt = _unary(np.sqrt)
square = _unary(np.square)
sign = _unary(np.sign)
sigmoid = _unary(_sigmoid)
floor = _unary(np.floor)


def abs(x, name=None):      # noqa: A001
    return Tensor(np.abs(_v(x)))


def rsqrt(x, name=None):
    v = _v(x)
    return Tensor((1.0 / np.sqrt(v)).astype(v.dtype))


def pow(x, y, name=None):       # noqa: A001
    return _t(x) ** y


def cast(x, dtype, name=None):
    return Tensor(_v(x).astype(_np_dtype(dtype)))


def to_float(x, name=None):
    return cast(x, float32)


def to_int32(x, name=None):
    return cast(x, int32)      # float -> int truncates toward zero, as TF's cast


def to_int64(x, name=None):
    return cast(x, int64)


def shape(x, name=None, out_type=None):
    return Tensor(np.asarray(_v(x).shape, dtype=np.int32))


def size(x, name=None):
    return Tensor(np.asarray(_v(x).size, dtype=np.int32))


def rank(x, name=None):
    return Tensor(np.asarray(_v(x).ndim, dtype=np.int32))


def _zeros_like_dtype(dtype):
    np_dt = _np_dtype(dtype)
    return np_dt


def zeros(shape, dtype=float32, name=None):
    return Tensor(np.zeros(_shape_arg(shape), _np_dtype(dtype)))


def ones(shape, dtype=float32, name=None):
    return Tensor(np.ones(_shape_arg(shape), _np_dtype(dtype)))


def zeros_like(x, dtype=None, name=None):
    v = _v(x)
    return Tensor(np.zeros(v.shape, _np_dtype(dtype) or v.dtype))


def ones_like(x, dtype=None, name=None):
    v = _v(x)
    return Tensor(np.ones(v.shape, _np_dtype(dtype) or v.dtype))


def fill(dims, value, name=None):
    v = _convert(value)
    return Tensor(np.full(_shape_arg(dims), v, dtype=v.dtype))


def range(start, limit=None, delta=1, dtype=None, name=None):      # noqa: A001
    if limit is None:
        start, limit = 0, start
    vals = [start, limit, delta]
    is_float = any(isinstance(_convert(v), np.ndarray) and _convert(v).dtype.kind == "f" for v in vals)
    np_dt = _np_dtype(dtype) or (np.float32 if is_float else np.int32)
    a, b, c = (_convert(v).item() for v in vals)
    return Tensor(np.arange(a, b, c).astype(np_dt))


def expand_dims(x, axis=None, name=None, dim=None):
    axis = dim if axis is None else axis
    return Tensor(np.expand_dims(_v(x), _int(axis)))


def squeeze(x, axis=None, name=None, squeeze_dims=None):
    axis = squeeze_dims if axis is None else axis
    if axis is None:
        return Tensor(np.squeeze(_v(x)))
    axis = tuple(axis) if isinstance(axis, (list, tuple)) else (axis,)
    return Tensor(np.squeeze(_v(x), axis=tuple(_int(a) for a in axis)))


def reshape(x, shape, name=None):
    return Tensor(np.reshape(_v(x), _shape_arg(shape)))


def transpose(x, perm=None, name=None):
    return Tensor(np.transpose(_v(x), None if perm is None else [_int(p) for p in perm]))


def concat(values, axis, name=None):
    arrs = [_v(x) for x in values]
    for a in arrs[1:]:
        _same_dtype(arrs[0], a, "concat")
    return Tensor(np.concatenate(arrs, axis=_int(axis)))


def stack(values, axis=0, name=None):
    arrs = [_v(x) for x in values]
    for a in arrs[1:]:
        _same_dtype(arrs[0], a, "stack")
    return Tensor(np.stack(arrs, axis=_int(axis)))


def unstack(value, num=None, axis=0, name=None):
    v = _v(value)
    return [Tensor(a) for a in np.moveaxis(v, axis, 0)]


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    v = _v(value)
    axis = _int(axis)
    if isinstance(num_or_size_splits, (list, tuple)):
        idx = np.cumsum([_int(s) for s in num_or_size_splits])[:-1]
        return [Tensor(a) for a in np.split(v, idx, axis=axis)]
    n = _int(num_or_size_splits)
    if v.shape[axis] % n:
        raise ValueError("split: dimension {} not divisible by {}".format(v.shape[axis], n))
    return [Tensor(a) for a in np.split(v, n, axis=axis)]


def tile(x, multiples, name=None):
    v = _v(x)
    m = _shape_arg(multiples)
    if len(m) != v.ndim:
        raise ValueError("tile: {} multiples for a rank-{} tensor".format(len(m), v.ndim))
    return Tensor(np.tile(v, m))


def pad(x, paddings, mode="CONSTANT", name=None, constant_values=0):
    p = [[_int(a), _int(b)] for a, b in paddings]
    return Tensor(np.pad(_v(x), p, mode="constant", constant_values=constant_values))


def gather(params, indices, axis=0, name=None):
    return Tensor(np.take(_v(params), _v(indices), axis=_int(axis)))


def gather_nd(params, indices, name=None):
    p, idx = _v(params), _v(indices)
    return Tensor(p[tuple(np.moveaxis(idx, -1, 0))])


def one_hot(indices, depth, on_value=None, off_value=None, axis=None, dtype=None, name=None):
    idx = _v(indices)
    if dtype is None:
        if on_value is not None:
            dtype = _tf_dtype(_convert(on_value).dtype)
        elif off_value is not None:
            dtype = _tf_dtype(_convert(off_value).dtype)
        else:
            dtype = float32
    np_dt = _np_dtype(dtype)
    on = np_dt(1) if on_value is None else np_dt(_convert(on_value))
    off = np_dt(0) if off_value is None else np_dt(_convert(off_value))
    depth = _int(depth)
    out = np.full(idx.shape + (depth,), off, dtype=np_dt)
    valid = (idx >= 0) & (idx < depth)
    it = np.nonzero(valid) if idx.ndim else None
    if idx.ndim == 0:
        if valid:
            out[int(idx)] = on
    else:
        out[it + (idx[valid],)] = on
    return Tensor(out)


def where(condition, x=None, y=None, name=None):
    c = _v(condition)
    if x is None:
        return Tensor(np.argwhere(c).astype(np.int64))
    if c.ndim == 0:                 # scalar predicate (train_mode): only the chosen side is ever evaluated
        return _t(x) if bool(c) else _t(y)
    xv, yv = _v(x), _v(y)
    _same_dtype(xv, yv, "where")
    if c.ndim == 0 or c.shape == xv.shape:
        return Tensor(np.where(c, xv, yv))
    if c.ndim == 1 and c.shape[0] == xv.shape[0]:       # TF's select: a vector condition picks rows
        return Tensor(np.where(c.reshape((-1,) + (1,) * (xv.ndim - 1)), xv, yv))
    raise ValueError("where: condition {} against {}".format(c.shape, xv.shape))


def matrix_band_part(x, num_lower, num_upper, name=None):
    v = _v(x)
    m, n = v.shape[-2:]
    i = np.arange(m)[:, None]
    j = np.arange(n)[None, :]
    lo, up = _int(num_lower), _int(num_upper)
    keep = ((lo < 0) | ((i - j) <= lo)) & ((up < 0) | ((j - i) <= up))
    return Tensor(np.where(keep, v, np.zeros((), v.dtype)))


def reverse_sequence(x, seq_lengths, seq_axis=None, batch_axis=None, name=None, seq_dim=None, batch_dim=None):
    seq_axis = seq_dim if seq_axis is None else seq_axis
    batch_axis = (batch_dim if batch_axis is None else batch_axis) or 0
    v = _v(x)
    lens = _v(seq_lengths)
    out = v.copy()
    vm = np.moveaxis(v, (batch_axis, seq_axis), (0, 1))
    om = np.moveaxis(out, (batch_axis, seq_axis), (0, 1))
    for b, n in enumerate(lens):
        n = int(n)
        om[b, :n] = vm[b, :n][::-1]
    return Tensor(out)


def _axes(axis, ndim):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return tuple(_int(a) for a in axis)
    if isinstance(axis, Tensor):
        a = axis.numpy()
        return tuple(int(i) for i in a.reshape(-1)) if a.ndim else int(a)
    return int(axis)


def _reduce(fn):
    def op(x, axis=None, keepdims=None, name=None, reduction_indices=None, keep_dims=None):
        axis = reduction_indices if axis is None else axis
        keepdims = bool(keepdims if keepdims is not None else keep_dims)
        v = _v(x)
        out = fn(v, axis=_axes(axis, v.ndim), keepdims=keepdims)
        return Tensor(np.asarray(out).astype(v.dtype) if v.dtype.kind == "f" else np.asarray(out))
    return op


reduce_sum = _reduce(np.sum)
reduce_mean = _reduce(np.mean)
reduce_max = _reduce(np.max)
reduce_min = _reduce(np.min)
reduce_all = _reduce(np.all)
reduce_any = _reduce(np.any)
reduce_prod = _reduce(np.prod)


def argmax(x, axis=None, name=None, dimension=None, output_type=int64):
    axis = dimension if axis is None else axis
    return Tensor(np.argmax(_v(x), axis=0 if axis is None else _int(axis)).astype(_np_dtype(output_type)))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    av, bv = _v(a), _v(b)
    _same_dtype(av, bv, "matmul")
    if transpose_a:
        av = np.swapaxes(av, -1, -2)
    if transpose_b:
        bv = np.swapaxes(bv, -1, -2)
    if av.ndim != bv.ndim:
        raise ValueError("matmul: ranks {} and {} (TF 1.12 does not broadcast batch dimensions)"
                         .format(av.ndim, bv.ndim))
    return Tensor(np.matmul(av, bv))


def tensordot(a, b, axes, name=None):
    return Tensor(np.tensordot(_v(a), _v(b), axes))


def clip_by_norm(t, clip_norm, axes=None, name=None):
    """``t * clip_norm / max(l2norm(t), clip_norm)`` (clip_ops.py)."""
    v = _v(t)
    c = v.dtype.type(_convert(clip_norm))
    l2 = np.sqrt(np.sum(v * v, axis=_axes(axes, v.ndim), keepdims=True))
    return Tensor((v * c / np.maximum(l2, c)).astype(v.dtype))


def random_normal(shape, mean=0.0, stddev=1.0, dtype=float32, seed=None, name=None):
    rng = np.random.default_rng(zlib.crc32(repr(("random_normal", tuple(_shape_arg(shape)))).encode()))
    return Tensor(rng.normal(mean, stddev, _shape_arg(shape)).astype(_np_dtype(dtype)))


def svd(tensor, full_matrices=False, compute_uv=True, name=None):
    u, s, vt = np.linalg.svd(_v(tensor), full_matrices=full_matrices)
    return Tensor(s), Tensor(u), Tensor(np.swapaxes(vt, -1, -2))


def multinomial(logits, num_samples, seed=None, name=None, output_dtype=None):
    raise NotImplementedError("tf.multinomial: sampling has no reproducible reference stream")


def py_func(func, inp, Tout, stateful=True, name=None):     # noqa: N803
    out = func(*[_v(i) for i in inp])
    return [Tensor(np.asarray(o)) for o in (out if isinstance(out, (list, tuple)) else [out])]


# --------------------------------------------------------------------------------------------------------------------
# control flow
# --------------------------------------------------------------------------------------------------------------------
def _pred(p):
    v = _v(p)
    if v.ndim != 0:
        raise ValueError("predicate must be a scalar")
    return bool(v)


def cond(pred, true_fn=None, false_fn=None, name=None, strict=False):
    return true_fn() if _pred(pred) else false_fn()


def case(pred_fn_pairs, default=None, exclusive=False, strict=False, name="case"):      # noqa: A002
    """tf.case: the function of the FIRST predicate that holds (a list keeps its order), else ``default``."""
    pairs = list(pred_fn_pairs.items()) if isinstance(pred_fn_pairs, dict) else list(pred_fn_pairs)
    hits = [fn for pred, fn in pairs if _pred(pred)]
    if exclusive and len(hits) > 1:
        raise ValueError("tf.case(exclusive=True): more than one predicate holds")
    if hits:
        return hits[0]()
    if default is None:
        raise ValueError("tf.case: no predicate holds and there is no default")
    return default()


def _global_step():
    """tf.train.get_or_create_global_step: the graph's ``global_step`` variable (a collection entry in TF, outside
    every variable scope), made on first use and handed back ever after."""
    existing = _STORE.vars.get("global_step")
    if existing is None:
        existing = Variable(np.zeros([], np.int64), "global_step", trainable=False)
        _STORE.vars["global_step"] = existing
    return existing


def while_loop(cond, body, loop_vars, shape_invariants=None, parallel_iterations=10, back_prop=True,      # noqa: A002
               swap_memory=False, name=None, maximum_iterations=None, return_same_structure=False):
    """Python loop: ``cond`` is evaluated BEFORE each iteration on the loop variables the previous iteration
    produced; the structure of ``loop_vars`` is kept (TF packs the body's result back into it)."""
    state = loop_vars
    is_seq = isinstance(loop_vars, (list, tuple))
    n = 0
    while True:
        if maximum_iterations is not None and n >= maximum_iterations:
            break
        keep = cond(*state) if is_seq else cond(state)
        if not _pred(keep):
            break
        new = body(*state) if is_seq else body(state)
        nest.assert_same_structure(state, new)
        state = nest.pack_sequence_as(loop_vars, nest.flatten(new))
        n += 1
    return state


# --------------------------------------------------------------------------------------------------------------------
# nest (tf.contrib.framework.nest): None and tensors are leaves; tuples / namedtuples / lists / dicts are structure
# --------------------------------------------------------------------------------------------------------------------
class _Nest:
    @staticmethod
    def is_sequence(x):
        return isinstance(x, (list, tuple, dict)) and not isinstance(x, (str, bytes))

    def flatten(self, s):
        if not self.is_sequence(s):
            return [s]
        out = []
        items = [s[k] for k in sorted(s)] if isinstance(s, dict) else s
        for e in items:
            out.extend(self.flatten(e))
        return out

    def pack_sequence_as(self, structure, flat):
        flat = list(flat)

        def build(s):
            if not self.is_sequence(s):
                return flat.pop(0)
            if isinstance(s, dict):
                return type(s)((k, build(s[k])) for k in sorted(s))
            items = [build(e) for e in s]
            if isinstance(s, tuple) and hasattr(s, "_fields"):
                return type(s)(*items)
            return type(s)(items)
        out = build(structure)
        if flat:
            raise ValueError("pack_sequence_as: {} values left over".format(len(flat)))
        return out

    def assert_same_structure(self, a, b, check_types=True):
        if self.is_sequence(a) != self.is_sequence(b):
            raise ValueError("structures differ: {!r} vs {!r}".format(type(a), type(b)))
        if not self.is_sequence(a):
            return
        if len(a) != len(b):
            raise ValueError("structures differ in length: {} vs {}".format(len(a), len(b)))
        ia = [a[k] for k in sorted(a)] if isinstance(a, dict) else a
        ib = [b[k] for k in sorted(b)] if isinstance(b, dict) else b
        for x, y in zip(ia, ib):
            self.assert_same_structure(x, y)

    def map_structure(self, func, *structures, **kw):
        for s in structures[1:]:
            self.assert_same_structure(structures[0], s)
        flats = [self.flatten(s) for s in structures]
        return self.pack_sequence_as(structures[0], [func(*xs) for xs in zip(*flats)])


nest = _Nest()


# --------------------------------------------------------------------------------------------------------------------
# tf.nn
# --------------------------------------------------------------------------------------------------------------------
def _softmax(x, axis=-1, name=None, dim=None):
    v = _v(x)
    axis = dim if dim is not None else axis
    e = np.exp(v - v.max(axis=axis, keepdims=True))
    return Tensor((e / e.sum(axis=axis, keepdims=True)).astype(v.dtype))


def _log_softmax(x, axis=-1, name=None, dim=None):
    v = _v(x)
    axis = dim if dim is not None else axis
    s = v - v.max(axis=axis, keepdims=True)
    return Tensor((s - np.log(np.exp(s).sum(axis=axis, keepdims=True))).astype(v.dtype))


def _relu(x, name=None):
    v = _v(x)
    return Tensor(np.maximum(v, np.zeros((), v.dtype)))


def _top_k(x, k=1, sorted=True, name=None):     # noqa: A002
    """Values descending; among equal values the LOWER index first (TF's TopK kernel)."""
    v = _v(x)
    k = _int(k)
    order = np.argsort(-v, axis=-1, kind="stable")[..., :k]
    return Tensor(np.take_along_axis(v, order, axis=-1)), Tensor(order.astype(np.int32))


def _embedding_lookup(params, ids, partition_strategy="mod", name=None, validate_indices=True, max_norm=None):
    p, i = _v(params), _v(ids)
    if i.size and (i.min() < 0 or i.max() >= p.shape[0]):
        raise IndexError("embedding_lookup: index out of range")
    return Tensor(p[i])


class _Unreplayable(Tensor):
    """Result of tf.nn.dropout: TF's Philox stream cannot be replayed, so the value must never be looked at.  The
    reference selects it with ``tf.where(train_mode, dropped, x)`` (nn/utils.py:21-22); with train_mode False the
    scalar-predicate ``where`` above returns ``x`` without touching this object."""

    def __init__(self, like):
        super().__init__(None, name="dropout:0")
        self._like = like

    def numpy(self):
        raise NotImplementedError("value of tf.nn.dropout requested (train_mode True with keep_prob < 1)")

    @property
    def shape(self):
        return self._like.shape

    @property
    def dtype(self):
        return self._like.dtype


def _dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    return _Unreplayable(_t(x))


def _conv2d(input, filter, strides, padding, use_cudnn_on_gpu=True, data_format="NHWC", dilations=None, name=None):   # noqa: A002
    x, w = _v(input), _v(filter)
    if w.shape[:2] != (1, 1) or list(strides) != [1, 1, 1, 1]:
        raise NotImplementedError("conv2d: only the 1x1 / stride-1 case the reference uses")
    _same_dtype(x, w, "conv2d")
    return Tensor(np.matmul(x, w[0, 0]))


def _max_pool(value, ksize, strides, padding, data_format="NHWC", name=None):
    v = _v(value)
    if list(ksize) != [1, 1, 2, 1] or list(strides) != [1, 1, 2, 1] or v.shape[2] != 2:
        raise NotImplementedError("max_pool: only the maxout window of nn/projection.py")
    return Tensor(v.max(axis=2, keepdims=True))


def _sparse_xent(labels=None, logits=None, name=None, _sentinel=None):
    lg, lb = _v(logits), _v(labels)
    lp = _log_softmax(lg).numpy()
    return Tensor(-np.take_along_axis(lp, lb[..., None].astype(np.int64), axis=-1)[..., 0])


# --------------------------------------------------------------------------------------------------------------------
# tf.layers (layers/base.py + layers/core.py of TF 1.12)
# --------------------------------------------------------------------------------------------------------------------
_LAYER_UIDS = {}


def _snake(name):
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    s = re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()
    return "private" + s if s[0] == "_" else s


class Layer:
    def __init__(self, trainable=True, name=None, dtype=None, _scope=None, _reuse=None, **kwargs):
        self.trainable = trainable
        self.built = False
        self._dtype = dtype
        if isinstance(name, VariableScope):
            base_name = name.name
        else:
            base_name = name
        self._name = name
        if not name:
            base_name = _snake(self.__class__.__name__)
            _LAYER_UIDS[base_name] = _LAYER_UIDS.get(base_name, 0) + 1
            n = _LAYER_UIDS[base_name]
            self._name = base_name if n == 1 else "{}_{}".format(base_name, n - 1)
        self._base_name = base_name
        self._scope = None
        self._reuse = _reuse
        self._weights = []
        if _scope:
            with variable_scope(_scope) as captured:
                self._scope = captured

    @property
    def name(self):
        return self._name

    @property
    def scope_name(self):
        return self._scope.name

    @property
    def variables(self):
        return list(self._weights)

    weights = trainable_variables = trainable_weights = variables

    def _set_scope(self, scope=None):
        if self._scope is None:
            if self._reuse:
                with variable_scope(scope if scope is not None else self._base_name) as captured:
                    self._scope = captured
            else:
                with variable_scope(scope, default_name=self._base_name) as captured:
                    self._scope = captured

    def add_variable(self, name, shape, dtype=None, initializer=None, trainable=True, **_):
        v = get_variable(name, shape=shape, dtype=dtype or float32, initializer=initializer,
                         trainable=trainable and self.trainable)
        self._weights.append(v)
        return v

    add_weight = add_variable

    def build(self, input_shape):
        self.built = True

    def call(self, inputs, *args, **kwargs):
        return inputs

    def __call__(self, inputs, *args, **kwargs):
        self._set_scope(kwargs.pop("scope", None))
        with variable_scope(self._scope, reuse=(self.built or self._reuse), auxiliary_name_scope=False):
            if not self.built:
                shapes = nest.map_structure(lambda x: _t(x).get_shape(), inputs)
                self.build(shapes)
                self.built = True
            return self.call(inputs, *args, **kwargs)

    apply = __call__


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer=None,
                 bias_initializer=None, trainable=True, name=None, **kwargs):
        super().__init__(trainable=trainable, name=name, **kwargs)
        self.units = _int(units)
        self.activation = activation
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer if bias_initializer is not None else zeros_initializer()

    def build(self, input_shape):
        last = TensorShape(input_shape)[-1].value
        self.kernel = self.add_variable("kernel", [last, self.units], initializer=self.kernel_initializer)
        self.bias = self.add_variable("bias", [self.units], initializer=self.bias_initializer) \
            if self.use_bias else None
        self.built = True

    def call(self, inputs):
        x = _v(inputs)
        w = self.kernel.numpy()
        _same_dtype(x, w, "dense")
        # core.Dense: rank > 2 -> tensordot over the last axis, rank 2 -> mat_mul; bias_add; activation
        out = np.matmul(x, w) if x.ndim == 2 else np.tensordot(x, w, [[x.ndim - 1], [0]])
        if self.use_bias:
            out = out + self.bias.numpy()
        out = Tensor(out)
        return self.activation(out) if self.activation is not None else out


class Conv2D(Layer):
    """layers.Conv2D for the 1x1 / stride-1 case of encoders/numpy_stateful_filler.py:219-230."""

    def __init__(self, filters, kernel_size, activation=None, use_bias=True, kernel_initializer=None,
                 bias_initializer=None, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        ks = kernel_size if isinstance(kernel_size, (list, tuple)) else (kernel_size, kernel_size)
        if tuple(ks) != (1, 1):
            raise NotImplementedError("Conv2D: only 1x1 kernels")
        self.filters, self.activation, self.use_bias = _int(filters), activation, use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer if bias_initializer is not None else zeros_initializer()

    def build(self, input_shape):
        depth = TensorShape(input_shape)[-1].value
        self.kernel = self.add_variable("kernel", [1, 1, depth, self.filters], initializer=self.kernel_initializer)
        self.bias = self.add_variable("bias", [self.filters], initializer=self.bias_initializer) \
            if self.use_bias else None
        self.built = True

    def call(self, inputs):
        out = _conv2d(inputs, self.kernel, [1, 1, 1, 1], "VALID")
        if self.use_bias:
            out = out + self.bias
        return self.activation(out) if self.activation is not None else out


def _layers_conv2d(inputs, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, use_bias=True,
                   kernel_initializer=None, bias_initializer=None, name=None, reuse=None, **_):
    layer = Conv2D(filters, kernel_size, activation=activation, use_bias=use_bias,
                   kernel_initializer=kernel_initializer, bias_initializer=bias_initializer, name=name,
                   _scope=name, _reuse=reuse)
    return layer.apply(inputs)


def _dense(inputs, units, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
           kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
           bias_constraint=None, trainable=True, name=None, reuse=None):
    layer = Dense(units, activation=activation, use_bias=use_bias, kernel_initializer=kernel_initializer,
                  bias_initializer=bias_initializer, trainable=trainable, name=name, _scope=name, _reuse=reuse)
    return layer.apply(inputs)


# --------------------------------------------------------------------------------------------------------------------
# RNN cells and dynamic_rnn (rnn_cell_impl.py, rnn.py of TF 1.12) -- TF-internal arithmetic, RESTATED
# --------------------------------------------------------------------------------------------------------------------
class LSTMStateTuple(tuple):
    __slots__ = ()
    _fields = ("c", "h")

    def __new__(cls, c, h):
        return tuple.__new__(cls, (c, h))

    c = property(lambda self: self[0])
    h = property(lambda self: self[1])

    def _replace(self, **kw):
        return LSTMStateTuple(kw.get("c", self[0]), kw.get("h", self[1]))


class RNNCell(Layer):
    def __call__(self, inputs, state, scope=None):
        if scope is not None:
            with variable_scope(scope) as sc:
                return Layer.__call__(self, inputs, state, scope=sc)
        sc = getattr(self, "rnncell_scope", None)
        if sc is None:
            sc = self.rnncell_scope = get_variable_scope()
        with variable_scope(sc):
            return Layer.__call__(self, inputs, state)

    def build(self, inputs_shape):
        self.built = True

    def zero_state(self, batch_size, dtype):
        n = _int(batch_size)
        return nest.map_structure(lambda s: zeros([n, s], dtype), self.state_size)


class GRUCell(RNNCell):
    """rnn_cell_impl.GRUCell: gates = sigmoid([x, h] Wg + bg), r, u = split(gates); c = act([x, r*h] Wc + bc);
    h' = u*h + (1-u)*c.  gates/bias initialised to 1.0, candidate/bias to 0."""

    def __init__(self, num_units, activation=None, reuse=None, kernel_initializer=None, bias_initializer=None,
                 name=None, dtype=None):
        super().__init__(name=name, _reuse=reuse, dtype=dtype)
        self._num_units = _int(num_units)
        self._activation = activation or tanh
        self._kernel_initializer = kernel_initializer
        self._bias_initializer = bias_initializer

    state_size = property(lambda self: self._num_units)
    output_size = property(lambda self: self._num_units)

    def build(self, inputs_shape):
        shp = inputs_shape[0] if isinstance(inputs_shape, (list, tuple)) else inputs_shape
        depth = TensorShape(shp)[1].value
        n = self._num_units
        self._gate_kernel = self.add_variable("gates/kernel", [depth + n, 2 * n],
                                              initializer=self._kernel_initializer)
        self._gate_bias = self.add_variable(
            "gates/bias", [2 * n],
            initializer=self._bias_initializer if self._bias_initializer is not None else constant_initializer(1.0))
        self._candidate_kernel = self.add_variable("candidate/kernel", [depth + n, n],
                                                   initializer=self._kernel_initializer)
        self._candidate_bias = self.add_variable(
            "candidate/bias", [n],
            initializer=self._bias_initializer if self._bias_initializer is not None else zeros_initializer())
        self.built = True

    def call(self, inputs, state):
        gate_inputs = matmul(concat([inputs, state], 1), self._gate_kernel) + self._gate_bias
        value = sigmoid(gate_inputs)
        r, u = split(value, 2, axis=1)
        r_state = r * state
        candidate = matmul(concat([inputs, r_state], 1), self._candidate_kernel) + self._candidate_bias
        c = self._activation(candidate)
        new_h = u * state + (1 - u) * c
        return new_h, new_h


class LSTMCell(RNNCell):
    """rnn_cell_impl.LSTMCell without peepholes / projection: z = [x, h] W + b; i, j, f, o = split(z, 4);
    c' = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j); h' = sigmoid(o) tanh(c')."""

    def __init__(self, num_units, use_peepholes=False, cell_clip=None, initializer=None, num_proj=None,
                 forget_bias=1.0, state_is_tuple=True, activation=None, reuse=None, name=None, dtype=None):
        super().__init__(name=name, _reuse=reuse, dtype=dtype)
        if use_peepholes or cell_clip is not None or num_proj is not None or not state_is_tuple:
            raise NotImplementedError("LSTMCell options the reference never sets")
        self._num_units = _int(num_units)
        self._initializer = initializer
        self._forget_bias = forget_bias
        self._activation = activation or tanh

    state_size = property(lambda self: LSTMStateTuple(self._num_units, self._num_units))
    output_size = property(lambda self: self._num_units)

    def build(self, inputs_shape):
        shp = inputs_shape[0] if isinstance(inputs_shape, (list, tuple)) and not isinstance(inputs_shape, TensorShape) \
            else inputs_shape
        depth = TensorShape(shp)[1].value
        n = self._num_units
        self._kernel = self.add_variable("kernel", [depth + n, 4 * n], initializer=self._initializer)
        self._bias = self.add_variable("bias", [4 * n], initializer=zeros_initializer())
        self.built = True

    def call(self, inputs, state):
        c_prev, m_prev = state
        z = matmul(concat([inputs, m_prev], 1), self._kernel) + self._bias
        i, j, f, o = split(z, 4, axis=1)
        fb = _convert(self._forget_bias, dtype=_v(f).dtype)
        c = sigmoid(f + Tensor(fb)) * c_prev + sigmoid(i) * self._activation(j)
        m = sigmoid(o) * self._activation(c)
        return m, LSTMStateTuple(c, m)


def _dynamic_rnn(cell, inputs, sequence_length=None, initial_state=None, dtype=None, parallel_iterations=None,
                 swap_memory=False, time_major=False, scope=None):
    """rnn.dynamic_rnn: beyond a row's length the output row is zero and the state is copied through."""
    with variable_scope(scope or "rnn"):
        x = _v(inputs)
        if time_major:
            x = np.swapaxes(x, 0, 1)
        bsz, steps = x.shape[0], x.shape[1]
        state = initial_state if initial_state is not None else cell.zero_state(bsz, dtype or float32)
        lens = None if sequence_length is None else _v(sequence_length).astype(np.int64)
        outs = []
        for t in _range(steps):
            out, new_state = cell(Tensor(x[:, t]), state)
            if lens is not None:
                live = (t < lens)[:, None]
                out = Tensor(np.where(live, _v(out), np.zeros((), _v(out).dtype)))
                new_state = nest.map_structure(
                    lambda n, o: Tensor(np.where(live, _v(n), _v(o))), new_state, state)
            outs.append(_v(out))
            state = new_state
        if outs:
            y = np.stack(outs, axis=1)
        else:
            y = np.zeros((bsz, 0, cell.output_size), x.dtype)
        if time_major:
            y = np.swapaxes(y, 0, 1)
        return Tensor(y), state


def _bidirectional_dynamic_rnn(cell_fw, cell_bw, inputs, sequence_length=None, initial_state_fw=None,
                               initial_state_bw=None, dtype=None, parallel_iterations=None, swap_memory=False,
                               time_major=False, scope=None):
    if time_major:
        raise NotImplementedError
    with variable_scope(scope or "bidirectional_rnn"):
        with variable_scope("fw") as fw_scope:
            out_fw, st_fw = _dynamic_rnn(cell_fw, inputs, sequence_length, initial_state_fw, dtype, scope=fw_scope)
        with variable_scope("bw") as bw_scope:
            rev = reverse_sequence(inputs, sequence_length, seq_axis=1, batch_axis=0)
            tmp, st_bw = _dynamic_rnn(cell_bw, rev, sequence_length, initial_state_bw, dtype, scope=bw_scope)
        out_bw = reverse_sequence(tmp, sequence_length, seq_axis=1, batch_axis=0)
    return (out_fw, out_bw), (st_fw, st_bw)


# --------------------------------------------------------------------------------------------------------------------
# losses (contrib/seq2seq/python/ops/loss.py, losses/losses_impl.py) -- RESTATED
# --------------------------------------------------------------------------------------------------------------------
def _sequence_loss(logits, targets, weights, average_across_timesteps=True, average_across_batch=True,
                   softmax_loss_function=None, name=None):
    lg, tg, w = _v(logits), _v(targets), _v(weights)
    if lg.ndim != 3 or tg.ndim != 2 or w.ndim != 2:
        raise ValueError("sequence_loss: ranks")
    num_classes = lg.shape[2]
    flat = lg.reshape(-1, num_classes)
    tflat = tg.reshape(-1)
    if softmax_loss_function is None:
        crossent = _sparse_xent(labels=Tensor(tflat), logits=Tensor(flat)).numpy()
    else:
        crossent = _v(softmax_loss_function(labels=Tensor(tflat), logits=Tensor(flat)))
    crossent = crossent * w.reshape(-1)
    if average_across_timesteps and average_across_batch:
        return Tensor((crossent.sum() / (w.sum() + 1e-12)).astype(lg.dtype))
    crossent = crossent.reshape(lg.shape[:2])
    if average_across_timesteps and not average_across_batch:
        return Tensor((crossent.sum(1) / (w.sum(1) + 1e-12)).astype(lg.dtype))
    if not average_across_timesteps and average_across_batch:
        return Tensor((crossent.sum(0) / (w.sum(0) + 1e-12)).astype(lg.dtype))
    return Tensor(crossent.astype(lg.dtype))


def _softmax_cross_entropy(onehot_labels, logits, weights=1.0, label_smoothing=0, scope=None, **_):
    """losses.softmax_cross_entropy with the default SUM_BY_NONZERO_WEIGHTS reduction: ONE scalar."""
    lb, lg = _v(onehot_labels).astype(_v(logits).dtype), _v(logits)
    if label_smoothing:
        n = lb.shape[-1]
        ls = lg.dtype.type(label_smoothing)
        lb = lb * (lg.dtype.type(1.0) - ls) + ls / lg.dtype.type(n)
    losses = -(lb * _log_softmax(lg).numpy()).sum(-1)
    w = np.broadcast_to(_convert(weights, dtype=lg.dtype), losses.shape)
    present = (w != 0).sum()
    return Tensor((np.sum(losses * w) / max(present, 1)).astype(lg.dtype))


# --------------------------------------------------------------------------------------------------------------------
# lookup tables (contrib/lookup)
# --------------------------------------------------------------------------------------------------------------------
class _IndexTable:
    def __init__(self, mapping, num_oov_buckets=0, default_value=-1):
        if num_oov_buckets:
            raise NotImplementedError("oov buckets")
        self._map = {w: i for i, w in enumerate(mapping)}
        self._default = default_value

    def lookup(self, keys):
        k = _v(keys)
        out = np.empty(k.shape, dtype=np.int64)
        flat = k.reshape(-1)
        of = out.reshape(-1)
        for i, w in enumerate(flat):
            if isinstance(w, bytes):
                w = w.decode("utf-8")
            of[i] = self._map.get(w, self._default)
        return Tensor(out)


class _StringTable:
    def __init__(self, mapping, default_value="UNK"):
        self._words = list(mapping)
        self._default = default_value

    def lookup(self, ids):
        i = _v(ids)
        out = np.empty(i.shape, dtype=object)
        of, fl = out.reshape(-1), i.reshape(-1)
        for n, j in enumerate(fl):
            of[n] = self._words[j] if 0 <= j < len(self._words) else self._default
        return Tensor(out)


# --------------------------------------------------------------------------------------------------------------------
# sessions / summaries / savers: inert
# --------------------------------------------------------------------------------------------------------------------
class Session:
    def __init__(self, *a, **k):
        pass

    def run(self, fetches, feed_dict=None):
        with feeding(feed_dict or {}):
            return nest.map_structure(lambda t: t.numpy() if isinstance(t, Tensor) else None, fetches)

    def close(self):
        pass


class Summary:
    class Value:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    def __init__(self, value=None):
        self.value = value or []


class _Saver:
    def __init__(self, var_list=None, **_):
        self.var_list = var_list

    def save(self, *a, **k):
        raise NotImplementedError

    def restore(self, *a, **k):
        raise NotImplementedError


class _Optimizer:
    def __init__(self, *a, **k):
        self.args, self.kw = a, k


# --------------------------------------------------------------------------------------------------------------------
# module assembly
# --------------------------------------------------------------------------------------------------------------------
class _Missing:
    """A tf name this stand-in does not provide: harmless in annotations / default arguments of reference modules
    that are merely imported by a package ``__init__``, an error the moment it is called or touched."""

    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **k):
        raise NotImplementedError("tf stand-in has no " + self._name)

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Missing(self._name + "." + item)

    def __repr__(self):
        return "<missing {}>".format(self._name)


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)

    def fallback(attr, _name=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Missing(_name + "." + attr)
    m.__getattr__ = fallback
    return m


def build_modules():
    me = sys.modules[__name__]
    tf = _module("tensorflow")
    for k, v in vars(me).items():
        if not k.startswith("_"):
            setattr(tf, k, v)
    tf.bool = bool_
    tf.float32, tf.float64, tf.int32, tf.int64, tf.string = float32, float64, int32, int64, string
    tf.DType, tf.Tensor, tf.Variable, tf.TensorShape, tf.Dimension = DType, Tensor, Variable, TensorShape, Dimension
    tf.Operation = type(no_op())
    tf.AUTO_REUSE = AUTO_REUSE
    tf.newaxis = None
    tf.__version__ = "1.12.0-numpy-eager-standin"

    nn = _module("tensorflow.nn", softmax=_softmax, log_softmax=_log_softmax, relu=_relu, top_k=_top_k,
                 embedding_lookup=_embedding_lookup, dropout=_dropout, conv2d=_conv2d, max_pool=_max_pool,
                 tanh=tanh, sigmoid=sigmoid, dynamic_rnn=_dynamic_rnn,
                 bidirectional_dynamic_rnn=_bidirectional_dynamic_rnn,
                 sparse_softmax_cross_entropy_with_logits=_sparse_xent)
    rnn_cell = _module("tensorflow.nn.rnn_cell", RNNCell=RNNCell, GRUCell=GRUCell, LSTMCell=LSTMCell,
                       LSTMStateTuple=LSTMStateTuple)
    nn.rnn_cell = rnn_cell
    layers = _module("tensorflow.layers", dense=_dense, Dense=Dense, Layer=Layer, conv2d=_layers_conv2d, Conv2D=Conv2D)
    losses = _module("tensorflow.losses", softmax_cross_entropy=_softmax_cross_entropy)
    summary = _module("tensorflow.summary", scalar=lambda *a, **k: None, image=lambda *a, **k: None,
                      histogram=lambda *a, **k: None, merge=lambda *a, **k: None,
                      FileWriter=lambda *a, **k: None)
    train = _module("tensorflow.train", Saver=_Saver, Optimizer=_Optimizer, AdamOptimizer=_Optimizer,
                    get_or_create_global_step=_global_step)

    contrib = _module("tensorflow.contrib")
    contrib.rnn = _module("tensorflow.contrib.rnn", RNNCell=RNNCell, GRUCell=GRUCell, LSTMCell=LSTMCell,
                          LSTMStateTuple=LSTMStateTuple)
    contrib.framework = _module("tensorflow.contrib.framework", nest=nest)
    contrib.seq2seq = _module("tensorflow.contrib.seq2seq", sequence_loss=_sequence_loss)
    contrib.lookup = _module(
        "tensorflow.contrib.lookup",
        index_table_from_tensor=lambda mapping, num_oov_buckets=0, default_value=-1, **_:
            _IndexTable(mapping, num_oov_buckets, default_value),
        index_to_string_table_from_tensor=lambda mapping, default_value="UNK", **_:
            _StringTable(mapping, default_value))
    contrib.tensorboard = _module("tensorflow.contrib.tensorboard")
    contrib.tensorboard.plugins = _module("tensorflow.contrib.tensorboard.plugins", projector=_module("projector"))

    python = _module("tensorflow.python")
    python.framework = _module("tensorflow.python.framework", ops=_module("tensorflow.python.framework.ops",
                                                                          Tensor=Tensor))
    python.debug = _module("tensorflow.python.debug")

    tf.nn, tf.layers, tf.losses, tf.summary, tf.train, tf.contrib, tf.python = \
        nn, layers, losses, summary, train, contrib, python
    mods = {
        "tensorflow": tf, "tensorflow.nn": nn, "tensorflow.nn.rnn_cell": rnn_cell, "tensorflow.layers": layers,
        "tensorflow.losses": losses, "tensorflow.summary": summary, "tensorflow.train": train,
        "tensorflow.contrib": contrib, "tensorflow.contrib.rnn": contrib.rnn,
        "tensorflow.contrib.framework": contrib.framework, "tensorflow.contrib.seq2seq": contrib.seq2seq,
        "tensorflow.contrib.lookup": contrib.lookup, "tensorflow.contrib.tensorboard": contrib.tensorboard,
        "tensorflow.contrib.tensorboard.plugins": contrib.tensorboard.plugins,
        "tensorflow.python": python, "tensorflow.python.framework": python.framework,
        "tensorflow.python.framework.ops": python.framework.ops, "tensorflow.python.debug": python.debug,
    }
    return mods


def _check_type(argname, value, expected_type, memo=None):
    """``typeguard.check_type`` for the annotations the reference hands to it through ``util/match_type.py`` (the
    dataset series specifications of dataset.py:24-52): classes, Any, Union, List / Tuple / Dict / Iterable /
    Callable with or without parameters.  Raises TypeError on a mismatch, like typeguard 2."""
    import collections.abc as abc
    import typing

    def matches(val, tp):
        if tp is typing.Any or tp is None and val is None:
            return True
        if tp is None:
            return val is None
        origin = typing.get_origin(tp)
        args = typing.get_args(tp)
        if origin is typing.Union:
            return any(matches(val, a) for a in args)
        if origin in (list, abc.Sequence, abc.Iterable, abc.MutableSequence):
            want = list if origin is list else abc.Iterable
            if not isinstance(val, want) or (origin is not list and isinstance(val, str)):
                return isinstance(val, want) and origin is abc.Iterable
            return not args or all(matches(v, args[0]) for v in val)
        if origin is tuple:
            if not isinstance(val, tuple):
                return False
            if not args:
                return True
            if len(args) == 2 and args[1] is Ellipsis:
                return all(matches(v, args[0]) for v in val)
            return len(val) == len(args) and all(matches(v, a) for v, a in zip(val, args))
        if origin is dict:
            return isinstance(val, dict) and (not args or all(matches(k, args[0]) and matches(v, args[1])
                                                              for k, v in val.items()))
        if origin is abc.Callable:
            return callable(val)
        if origin is not None:
            return isinstance(val, origin)
        if isinstance(tp, type):
            return isinstance(val, tp) and not (tp is int and isinstance(val, bool))
        return True                      # TypeVars, forward references: not checked
    if not matches(value, expected_type):
        raise TypeError("type of {} must be {}; got {} instead".format(argname, expected_type, type(value).__name__))


def install():
    """Put the stand-in (plus a minimal ``typeguard`` and a no-op ``termcolor``) into ``sys.modules``.  Refuses to
    shadow a real TensorFlow."""
    if "tensorflow" in sys.modules and not getattr(sys.modules["tensorflow"], "__version__", "").endswith("standin"):
        raise RuntimeError("a real tensorflow is already imported")
    mods = build_modules()
    sys.modules.update(mods)
    if "typeguard" not in sys.modules:
        sys.modules["typeguard"] = _module("typeguard", check_argument_types=lambda *a, **k: True,
                                           check_type=_check_type)
    if "termcolor" not in sys.modules:
        sys.modules["termcolor"] = _module("termcolor", colored=lambda text, *a, **k: text)
    return mods["tensorflow"]
