"""Optimizer descriptions (stand-ins for the tf.train.* objects INI files name).

The update itself is the fused clip+Adam HIP kernel over the flat parameter
buffer (csrc/nm_optim.hip); these classes only carry hyper-parameters."""
import math
from typing import Callable, Union


class Optimizer:
    pass


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
