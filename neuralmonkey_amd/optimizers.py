"""Optimizer descriptions (stand-ins for the tf.train.* objects INI files name).

The update itself is a fused clip + Adam / Adadelta HIP kernel over the flat parameter
buffer (csrc/nm_optim.hip); these classes only carry hyper-parameters and the names of
the two slot variables a TensorFlow checkpoint holds per variable (``slot_suffixes``)."""
import math
from typing import Callable, Union


class Optimizer:
    slot_suffixes = ("/Adam", "/Adam_1")


class AdamOptimizer(Optimizer):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, learning_rate: Union[float, Callable[[int], float]] = 0.001, beta1: float = 0.9,
                 beta2: float = 0.999, epsilon: float = 1e-8, use_locking: bool = False,
                 name: str = "Adam") -> None:
        self._lr = learning_rate
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon

    def learning_rate(self, step: int) -> float:
        """Learning rate of update number ``step`` (1-based).  A schedule is evaluated at the
        global step before the update, as the TF graph does (functions.py)."""
        return float(self._lr(step - 1)) if callable(self._lr) else float(self._lr)

    def lr_t(self, step: int, applied: int = None) -> float:
        """``step`` counts from 1 (the value of global_step after this update) and drives the learning
        rate schedule; ``applied`` is the number of updates this optimizer has applied including this
        one (TF keeps beta1_power / beta2_power per optimizer) -- equal to ``step`` with one trainer."""
        t = step if applied is None else applied
        return self.learning_rate(step) * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)


class LazyAdamOptimizer(AdamOptimizer):
    """tf.contrib.opt.LazyAdamOptimizer.  In Neural Monkey the L1/L2 terms make
    every embedding gradient dense (SURVEY section 9 "Gradient density"), so it
    behaves exactly like Adam."""


class AdadeltaOptimizer(Optimizer):
    """tf.train.AdadeltaOptimizer (tests/bpe.ini:102-108, tests/str.ini:100-106).  TF 1.12's ApplyAdadelta:
    accum = rho*accum + (1-rho)*g^2;  update = sqrt(accum_update + eps) * rsqrt(accum + eps) * g;
    var -= lr*update;  accum_update = rho*accum_update + (1-rho)*update^2.  The slots are created under the
    optimizer's ``name``: ``<var>/<name>`` (accum) and ``<var>/<name>_1`` (accum_update)."""

    def __init__(self, learning_rate: Union[float, Callable[[int], float]] = 0.001, rho: float = 0.95,
                 epsilon: float = 1e-8, use_locking: bool = False, name: str = "Adadelta") -> None:
        self._lr = learning_rate
        self.rho, self.epsilon = rho, epsilon
        self.slot_suffixes = ("/" + name, "/" + name + "_1")

    def learning_rate(self, step: int) -> float:
        """As ``AdamOptimizer.learning_rate``: a schedule is evaluated at the global step before the update."""
        return float(self._lr(step - 1)) if callable(self._lr) else float(self._lr)
