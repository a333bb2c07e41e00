"""DelayedUpdateTrainer (mirror of neuralmonkey/trainers/delayed_update_trainer.py).

The reference runs three session calls per batch (accumulate -> [apply the mean gradient] ->
[reset], :27-82) over TF gradient-buffer variables.  Here the buffer is one flat fp32 tensor the
size of the parameter buffer: every ``execute`` call adds the batch's objective gradients to it;
the ``batches_per_update``-th call turns it into the mean, applies the usual regulariser / clip /
Adam step of GenericTrainer on it and clears it.  (The regulariser gradient is the same for every
accumulated batch, so adding it once to the mean equals averaging per-batch sums, :171-178.)
With data parallelism the all-reduce runs once per update, on the accumulated buffer.
"""
from typing import List, Sequence

from .. import ops
from ..optimizers import Optimizer
from ..runtime import tensor
from .generic_trainer import GenericTrainer
from .objective import Objective


# pylint: disable=too-many-arguments
class DelayedUpdateTrainer(GenericTrainer):
    def __init__(self, batches_per_update: int, objectives: Sequence[Objective], l1_weight: float = 0.0,
                 l2_weight: float = 0.0, clip_norm: float = None, optimizer: Optimizer = None,
                 var_scopes: List[str] = None, var_collection: str = None) -> None:
        GenericTrainer.__init__(self, objectives, l1_weight, l2_weight, clip_norm, optimizer, var_scopes,
                                var_collection)
        if batches_per_update < 1:
            raise ValueError("batches_per_update must be a positive integer")
        self.batches_per_update = batches_per_update
        self._counter = 0                      # cumulator_counter (:138-140)

    # the accumulation buffer lives across steps: a step's error flag is read before the NEXT step starts (the host
    # stays one step ahead at most), so at most one garbage gradient -- zeroed on the device, below -- ever meets it
    GUARD_LOOKBACK = 1

    def snapshot_counters(self, sess):
        return (GenericTrainer.snapshot_counters(self, sess), self._counter)

    def restore_counters(self, sess, snap) -> None:
        GenericTrainer.restore_counters(self, sess, snap[0])
        self._counter = snap[1]

    @tensor
    def train_op(self, ctx) -> int:
        store = ctx.store
        ctx.session.begin_guarded_step(self, ctx.feed, self.GUARD_LOOKBACK)
        self._objective_gradients(ctx)
        grad = store.ensure_grad()
        if ctx.session.device.type == "cuda":
            ops.zero_if(ctx.session.error_word(), grad)       # a given-up time loop's gradient never enters the buffer
        accum = ctx.buffer((id(self), "gradient_buffer"), (store.total,))
        if self._counter == 0:
            ops.copy(accum, grad)                                  # first batch after a reset (:160-168)
        else:
            ops.ew("copy", grad, None, accum, accumulate=True)     # tf.assign_add (:146-150)
        self._counter += 1
        if self._counter < self.batches_per_update:
            return ctx.session.global_step
        ops.ew("scale", accum, None, grad, alpha=1.0 / self._counter)   # averaged gradients (:171-178)
        self._counter = 0
        return self._apply_gradients(ctx)
