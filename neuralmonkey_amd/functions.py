"""Schedules INI files name (mirror of neuralmonkey/functions.py).  The reference returns TF
tensors evaluated against the global step; here every schedule is a callable of the global step
*before* the update (what the TF graph reads when ``apply_gradients`` runs)."""
import math
from typing import Any, Callable, List, Optional


def noam_decay(learning_rate: float, model_dimension: int, warmup_steps: int) -> Callable[[int], float]:
    """functions.py:58-80: lr * d_model^-0.5 * min(step^-0.5, step * warmup^-1.5)."""
    inv_sq_dim = 1.0 / math.sqrt(model_dimension)
    inv_sq3_warmup = math.pow(warmup_steps, -1.5)

    def schedule(step: int) -> float:
        step = float(step)
        inv_sq_step = 1.0 / math.sqrt(step) if step > 0 else float("inf")
        return learning_rate * inv_sq_dim * min(inv_sq_step, step * inv_sq3_warmup)
    return schedule


def inverse_sigmoid_decay(param: Callable[[int], float], rate: float, min_value: float = 0.0,
                          max_value: float = 1.0, name: Optional[str] = None,
                          dtype: Any = None) -> Callable[[int], float]:
    """functions.py:9-28: k/(k+exp(x/k)) scaled to (min_value, max_value).  (``name`` / ``dtype`` name the TensorFlow
    op and its type in the reference; a host schedule has neither -- accepted so that a call written for the reference
    goes through.)"""
    del name, dtype
    def schedule(step: int) -> float:
        x = param(step) if callable(param) else float(param)
        return rate / (rate + math.exp(x / rate)) * (max_value - min_value) + min_value
    return schedule


def piecewise_function(param: Callable[[int], float], values: List[float], changepoints: List[float],
                       name: Optional[str] = None, dtype: Any = None) -> Callable[[int], float]:
    """functions.py:31-55.  (``name`` / ``dtype``: as above.)"""
    del name, dtype
    if len(changepoints) != len(values) - 1:
        raise ValueError("changepoints has length {}, expected {} (values has length {})"
                         .format(len(changepoints), len(values) - 1, len(values)))

    def schedule(step: int) -> float:
        x = param(step) if callable(param) else float(param)
        for point, value in zip(changepoints, values):
            if x < point:
                return value
        return values[-1]
    return schedule
