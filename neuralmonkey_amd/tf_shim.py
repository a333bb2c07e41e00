"""The ``tf.*`` symbols that Neural Monkey INI files name
(tests/small.ini:68-80, tests/transformer.ini:97-102, tests/rl.ini),
mapped onto this engine's initializers / optimizers / activations."""
from types import SimpleNamespace

from . import optimizers as _opt
from . import variables as _var

random_uniform_initializer = _var.random_uniform_initializer
random_normal_initializer = _var.random_normal_initializer
zeros_initializer = _var.zeros_initializer
ones_initializer = _var.ones_initializer
constant_initializer = _var.constant_initializer
orthogonal_initializer = _var.orthogonal_initializer
glorot_uniform_initializer = _var.glorot_uniform_initializer


def _activation(name):
    def fn(*_args, **_kwargs):
        raise RuntimeError("tf.{} is a symbolic activation name in this engine".format(name))
    fn.nm_name = name
    fn.__name__ = name
    return fn


tanh = _activation("tanh")
identity = _activation("identity")
nn = SimpleNamespace(relu=_activation("relu"), tanh=tanh)

train = SimpleNamespace(AdamOptimizer=_opt.AdamOptimizer, AdadeltaOptimizer=_opt.AdadeltaOptimizer)
contrib = SimpleNamespace(opt=SimpleNamespace(LazyAdamOptimizer=_opt.LazyAdamOptimizer))
initializers = SimpleNamespace(random_uniform=random_uniform_initializer,
                               random_normal=random_normal_initializer, zeros=zeros_initializer,
                               ones=ones_initializer, orthogonal=orthogonal_initializer)
