"""GenericTrainer (mirror of neuralmonkey/trainers/generic_trainer.py).

train_op = forward (teacher forcing) -> hand-written backward into the flat
gradient buffer -> [data parallel: all-reduce over RCCL] -> regulariser terms +
per-tensor clip_by_norm + Adam as three fused launches over the flat buffers.
Fetch names and the ``ExecutionResult`` loss keys ("<decoder> - cost", "L1",
"L2") follow generic_trainer.py:39-51,245-250.
"""
import os
import re
from typing import Any, Dict, List, Sequence

import numpy as np
import torch

from .. import ops
from ..model.model_part import Feedable
from ..optimizers import AdadeltaOptimizer, AdamOptimizer, Optimizer
from ..runners.base_runner import GraphExecutor, LazyLosses, NextExecute
from ..runtime import HostPending, RunContext, tensor
from .objective import Objective

BIAS_REGEX = re.compile(r"[Bb]ias")
DEFER_LOSSES = os.environ.get("NM_DEFER_LOSSES", "1") != "0"


# pylint: disable=too-few-public-methods,too-many-arguments
class GenericTrainer(GraphExecutor, Feedable):
    class Executable(GraphExecutor.Executable):
        def __init__(self, executor: "GenericTrainer", compute_losses: bool, summaries: bool,
                     num_sessions: int) -> None:
            assert compute_losses
            if num_sessions != 1:
                raise ValueError("Trainer only supports execution in a single session")
            super().__init__(executor, compute_losses, summaries, num_sessions)

        def next_to_execute(self) -> NextExecute:
            return self.executor.fetches, []

        def collect_results(self, results: List[Dict]) -> None:
            assert len(results) == 1
            result = results[0]
            objective_names = [obj.name for obj in self.executor.objectives] + ["L1", "L2"]
            values = result["losses"]
            if isinstance(values, HostPending):       # still on their way to the host: read on first access
                losses = LazyLosses(objective_names, values)
            else:
                losses = dict(zip(objective_names, [float(x) for x in values]))
            self.set_result({}, losses, int(result["batch_size"]), [])

    @staticmethod
    def default_optimizer() -> Optimizer:
        return AdamOptimizer(learning_rate=1e-4)

    def __init__(self, objectives: Sequence[Objective], l1_weight: float = 0.0, l2_weight: float = 0.0,
                 clip_norm: float = None, optimizer: Optimizer = None, var_scopes: List[str] = None,
                 var_collection: str = None) -> None:
        GraphExecutor.__init__(self, {obj.decoder for obj in objectives})
        Feedable.__init__(self)
        self.objectives = objectives
        self.l1_weight = l1_weight
        self.l2_weight = l2_weight
        self.clip_norm = clip_norm
        self.var_scopes = var_scopes
        self.var_collection = var_collection
        self.optimizer = optimizer if optimizer is not None else self.default_optimizer()
        if not isinstance(self.optimizer, (AdamOptimizer, AdadeltaOptimizer)):
            raise NotImplementedError("the HIP trainer implements Adam / LazyAdam / Adadelta; got {}"
                                      .format(type(self.optimizer).__name__))
        if clip_norm is not None and clip_norm <= 0.0:
            raise ValueError("clip_norm must be positive")
        for obj in objectives:
            if obj.gradients is not None:
                raise NotImplementedError("objectives with explicit gradients are not supported")
        self._tables = {}

    # -- variable bookkeeping -------------------------------------------------------------
    def var_list(self, store) -> List[str]:
        names = store.trainable_names()
        if self.var_scopes is None:
            return names
        return [n for n in names if any(n.startswith(scope) for scope in self.var_scopes)]

    @staticmethod
    def regularizable(store) -> List[str]:
        """generic_trainer.py:87-91: trainables whose name lacks [Bb]ias."""
        return [n for n in store.trainable_names() if not BIAS_REGEX.findall(n)
                and not n.startswith(("vgg", "Inception", "resnet"))]

    def _optim_tables(self, store) -> ops.OptimizerTables:
        key = id(store)
        if key not in self._tables:
            from .. import distributed as dist
            dp = dist.current()
            # data parallel: chunks end where the ranks' slices of the flat buffers end (sharded optimizer); the table
            # is the same whether the update is sharded or replicated, so both add the same partial sums in one order
            cuts = dp.optimizer_cuts(store) if dp is not None and (dp.world_size > 1 or dp.forced) else ()
            self._tables[key] = ops.OptimizerTables(store, set(self.regularizable(store)), set(self.var_list(store)),
                                                    cuts=cuts)
        return self._tables[key]

    # -- the training step --------------------------------------------------------------------
    def _objective_gradients(self, outer) -> None:
        """Forward + backward of every objective into the (zeroed) flat gradient buffer.

        The pass runs in a run context of its own: an experiment may list several trainers that update
        on the same batch (tests/bahdanau.ini), and each must see activations computed from the CURRENT
        variables and own the backward tapes of the encoders it reads -- TensorFlow gives every trainer
        its own gradient subgraph.  The decoders' train results are handed back to the caller's context."""
        from .. import distributed as dist
        from ..runtime import RunContext
        ctx = RunContext(outer.session, outer.feed)
        ctx.memo["dp_overlap"] = bool(outer.memo.get("dp_overlap", False))
        ctx.memo["want_backward"] = True     # encoders keep their per-step state even when train_mode is fed False
        sess, store = ctx.session, ctx.store
        grad = store.ensure_grad()
        ops.zero(grad)
        dp = dist.current()
        if dp is not None:
            dp.begin_step()
        sess.step_tensor()                   # device copy of global_step (dropout salts), outside any capture
        for part in self.feedables:          # host -> device copies of the fed batch, into persistent buffers
            part.stage_inputs(ctx)
        train = bool(ctx.fed(self.train_mode)) if ctx.is_fed(self.train_mode) else True
        decoders, scales, counts = [], [], []
        for i, obj in enumerate(self.objectives):
            dec = obj.decoder
            if dec in decoders:
                raise NotImplementedError("two objectives over the decoder '{}' in one trainer".format(dec.name))
            weight = 1.0 if obj.weight is None else float(obj.weight)
            # loss = sum(xent) / sum(mask): with data parallelism the denominator is the
            # GLOBAL token count and gradients are summed over ranks
            count = dec.train_token_count(ctx)
            scale = ctx.buffer((id(self), "gscale", i), (1,))
            if dp is not None:
                dp.scale_by_global_count(count, weight, scale)               # on the device: no host exchange
            else:
                ops.fill(scale, weight / count if count else 0.0)               # a batch without target tokens: no gradient
            decoders.append(dec)
            scales.append(scale)
            counts.append(count)

        def forward_backward():
            """Every objective's forward + backward; encoders shared by several decoders run their
            backward pass once, on the summed gradient (RunContext.defer_backward)."""
            ctx.memo["backward_deferred"] = True
            results = []
            for dec, scale in zip(decoders, scales):
                res = dec._train_loop(ctx, want_grad=True, grad_scale=scale)     # pylint: disable=protected-access
                dec.backward(ctx, res)
                results.append(res)
            ctx.flush_backward()
            ctx.memo["backward_deferred"] = False
            return results
        if sess.use_step_graphs and all(getattr(d, "graph_safe_training", lambda t: False)(train) for d in decoders):
            # Taped (general-path) models launch hundreds of small kernels per step from Python:
            # the whole forward + backward becomes one HIP graph per batch shape.
            shapes = tuple(sorted((ph.name, tuple(np.shape(val))) for ph, val in ctx.feed.items()
                                  if hasattr(val, "shape")))
            results = sess.graphed_call((id(self), train, shapes), forward_backward)
            results = [res._replace(token_count=count) for res, count in zip(results, counts)]
        else:
            results = forward_backward()
        for dec, res in zip(decoders, results):
            outer.memo[dec.train_loop_result.key] = res
        sess.join_side()

    def _apply_gradients(self, ctx) -> int:
        """[gradient exchange] -> L1/L2 terms -> per-tensor clip_by_norm -> Adam / Adadelta -> [parameter all-gather]
        -> global_step += 1.  With data parallelism the exchange and the update are distributed.DataParallel's
        (replicated, or sharded over the ranks' slices of the flat buffers)."""
        from .. import distributed as dist
        sess, store = ctx.session, ctx.store
        grad = store.ensure_grad()
        dp = dist.current()
        tables = self._optim_tables(store)
        state = self._adam_state(sess, store)
        sess.global_step += 1
        state["applied"] += 1
        opt = self.optimizer
        if isinstance(opt, AdadeltaOptimizer):
            kind, params = 1, (opt.learning_rate(sess.global_step), opt.rho, opt.epsilon, 0.0)
        else:
            kind, params = 0, (opt.lr_t(sess.global_step, state["applied"]), opt.beta1, opt.beta2, opt.epsilon)
        # the update is a no-op on the device while the session's error word is set (a time loop of this step gave up:
        # the gradient is garbage, Session.recover_training runs the step again)
        skip = sess.error_word() if sess.device.type == "cuda" else None
        if dp is not None and (dp.world_size > 1 or dp.forced):
            l1l2 = dp.optimizer_step(store, tables, kind, state["m"], state["v"], self.l1_weight, self.l2_weight,
                                     self.clip_norm, params, skip=skip)
        else:
            l1l2 = tables.regularize_and_norms(store.theta, grad, self.l1_weight, self.l2_weight)
            tables.apply(kind, store.theta, grad, state["m"], state["v"], self.clip_norm, params, skip=skip)
        # (the norms live in the optimizer's workspace; the copy a step's losses read is one of four persistent slots)
        keep = ctx.buffer((id(self), "l1l2_kept", sess.global_step % 4), tuple(l1l2.shape))
        ctx.memo[(id(self), "l1l2")] = ops.copy(keep, l1l2)
        sess.variables_changed()
        return sess.global_step

    # -- what Session.recover_training puts back before it runs steps again -----------------------------------
    GUARD_LOOKBACK = 2            # steps the host may run ahead of the oldest unchecked error flag

    def snapshot_counters(self, sess):
        state = self.__dict__.get("_adam", {}).get(id(sess.store))
        return None if state is None else state["applied"]

    def restore_counters(self, sess, snap) -> None:
        state = self.__dict__.get("_adam", {}).get(id(sess.store))
        if state is not None:
            state["applied"] = snap if snap is not None else sess.global_step

    def _adam_state(self, sess, store):
        """The optimizer's two slots per variable (Adam: m, v; Adadelta: accum, accum_update) and the number of
        updates THIS optimizer has applied.  TensorFlow keeps one set of
        slots and one pair of beta powers per optimizer, while ``global_step`` is shared: an experiment
        with two trainers (tests/bahdanau.ini: ``trainer=[<mt_trainer>, <greedy_trainer>]``) advances the
        global step twice per batch but each optimizer's bias correction once.  The first trainer to
        update a store uses the store's own slots (the ones checkpoints carry) and continues from its
        global step; any further trainer gets private slots starting at zero."""
        key = id(store)
        state = self.__dict__.setdefault("_adam", {}).get(key)
        if state is None:
            owners = sess.__dict__.setdefault("_adam_owner", {})
            if owners.setdefault(key, self) is self:
                had_slots = store.adam_m is not None
                m, v = store.ensure_adam()
                mine = tuple(self.optimizer.slot_suffixes)
                if had_slots and tuple(store.slot_suffixes) != mine:
                    # a checkpoint written by ANOTHER optimizer (Adam's signed first moment is not an Adadelta
                    # accumulator: sqrt(accum + eps) would go NaN).  TensorFlow's Saver refuses such a restore (the
                    # slot variables it looks for are missing); here the variables stay and the slots start afresh.
                    import warnings
                    warnings.warn("optimizer slots {} of the restored checkpoint do not belong to {} (slots {}): they "
                                  "are reset to zero".format(store.slot_suffixes, type(self.optimizer).__name__, mine))
                    ops.zero(m)
                    ops.zero(v)
                store.slot_suffixes = mine                                    # the slots' names in checkpoints
                applied = sess.global_step
            else:
                m, v = torch.zeros_like(store.theta), torch.zeros_like(store.theta)
                applied = 0
            state = self._adam[key] = {"m": m, "v": v, "applied": applied}
        return state

    @tensor
    def train_op(self, ctx) -> int:
        # one optimizer step per batch: gradient slices that are final early in the backward pass may start
        # their all-reduce right away (distributed.DataParallel.all_reduce_early)
        ctx.session.begin_guarded_step(self, ctx.feed, self.GUARD_LOOKBACK)
        ctx.memo["dp_overlap"] = True
        self._objective_gradients(ctx)
        return self._apply_gradients(ctx)

    @tensor
    def regularization_losses(self, ctx):
        key = (id(self), "l1l2")
        if key not in ctx.memo:        # losses requested without a training step
            store = ctx.store
            tables = self._optim_tables(store)
            scratch = ctx.buffer((id(self), "zero_grad"), (store.total,), zero=True)
            ctx.memo[key] = tables.regularize_and_norms(store.theta, scratch, 0.0, 0.0).clone()
        return ctx.memo[key]

    @tensor
    def objective_values(self, ctx) -> List[Any]:
        losses = [o.loss(ctx) for o in self.objectives]
        l1l2 = self.regularization_losses(ctx)
        values = losses + [l1l2[0], l1l2[1]]
        # The scalars travel to the host asynchronously (Session.to_host_async) and ExecutionResult.losses reads
        # them on first access: the training loop only looks at them when it logs, and a blocking read-back per
        # step keeps the host from enqueuing the next step while this one runs.  NM_DEFER_LOSSES=0: read at once.
        # (behind them travels the session's device error word: a time loop that gave up makes the step garbage)
        if DEFER_LOSSES and all(isinstance(v, torch.Tensor) and v.is_cuda for v in values):
            # (gathered with 4-byte device copies into one of four persistent slots; the error word travels as its
            # raw int32 bits: zero stays 0.0, one reads as a denormal -- the host only asks "is it zero")
            sess = ctx.session
            vals = ctx.buffer((id(self), "loss_vals", sess.global_step % 4), (len(values) + 1,))
            for i, v in enumerate(values):
                ops.copy(vals[i:i + 1], v.detach().reshape(1))
            ops.copy(vals[len(values):].view(torch.int32), sess.error_word()[0:1])
            pending = sess.to_host_async(vals)
            ctx.session.attach_guarded_losses(self, pending)
            return pending
        if ctx.session.cluster_failure():
            # read at once: run the step again on the per-step path and hand out ITS losses
            sess = ctx.session
            guard = sess.__dict__.get("_train_guard") or []
            if getattr(sess, "_recovering", False) or not guard or guard[-1]["trainer"] is not self:
                sess.raise_device_error()
            sess.recover_training(len(guard) - 1)
            again = RunContext(sess, ctx.feed)
            losses = [o.loss(again) for o in self.objectives]
            l1l2 = self.regularization_losses(again)
            return losses + [l1l2[0], l1l2[1]]
        return values

    @property
    def fetches(self) -> Dict[str, Any]:
        return {"train_op": self.train_op, "losses": self.objective_values, "batch_size": self._batch_size_fetch}

    @tensor
    def _batch_size_fetch(self, ctx) -> int:
        return int(ctx.fed(self.batch_size))
