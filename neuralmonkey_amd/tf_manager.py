"""Executor (mirror of neuralmonkey/tf_manager.py): sessions, ``execute()``,
variable saving / n-best checkpoint rotation.

``TensorFlowManager.execute(batch, feedables, runners, train, compute_losses,
summaries) -> List[ExecutionResult]`` keeps the reference's call shape
(tf_manager.py:188-225).  A "session" is one variable set on this process's
GPU (``runtime.Session``); ``num_threads`` and the GPU memory options are
accepted for config compatibility and ignored (one process per GPU, HIP
streams instead of TF thread pools).
"""
import os
from typing import Dict, List, Optional, Sequence, Set, Union

import numpy as np
import torch

from .model.model_part import Feedable
from .runners.base_runner import ExecutionResult, GraphExecutor
from .runtime import Session, registered_parts, RunContext, _to_host


def default_device() -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    return torch.device("cpu")      # host-side plumbing only; any kernel call will fail loudly


# pylint: disable=too-many-instance-attributes
class TensorFlowManager:
    # pylint: disable=too-many-arguments
    def __init__(self, num_sessions: int, num_threads: int, save_n_best: int = 1,
                 minimize_metric: bool = False, gpu_allow_growth: bool = True,
                 per_process_gpu_memory_fraction: float = 1.0, enable_tf_debug: bool = False,
                 device: Optional[str] = None, seed: Optional[int] = None) -> None:
        if save_n_best < 1:
            raise Exception("save_n_best parameter must be greater than zero")
        self.saver_max_to_keep = save_n_best
        self.minimize_metric = minimize_metric
        self.num_sessions = num_sessions
        # "npz" (one file per checkpoint) or "tf": TensorFlow tensor bundles, the format the reference's
        # tf.train.Saver writes (tf_manager.py:274-277); restore() recognises either by the files present
        self.checkpoint_format = os.environ.get("NM_CHECKPOINT_FORMAT", "npz")
        self.num_threads = num_threads
        self.device = torch.device(device) if device is not None else default_device()
        self.seed = seed
        self.sessions = [Session(self.device, None if seed is None else seed + i)
                         for i in range(num_sessions)]
        self.saver = None
        self.best_score_index: Optional[int] = None
        self.best_score_epoch = 0
        self.best_score_batch = 0
        init_score = np.inf if self.minimize_metric else -np.inf
        self.saved_scores = [init_score for _ in range(self.saver_max_to_keep)]
        self.best_score = init_score
        self.variables_files: List[str] = []
        self._best_vars_file: Optional[str] = None

    # -- n-best bookkeeping (tf_manager.py:96-155) ---------------------------------
    @property
    def best_vars_file(self) -> str:
        if self._best_vars_file is None:
            raise RuntimeError("Saving not initialized yet.")
        return self._best_vars_file

    def _is_better(self, score1: float, score2: float) -> bool:
        return score1 < score2 if self.minimize_metric else score1 > score2

    def _argworst(self, scores: List[float]) -> int:
        return int(np.argmax(scores)) if self.minimize_metric else int(np.argmin(scores))

    def _update_best_vars(self, var_index: int) -> None:
        with open(self.best_vars_file, "w") as var_file:
            var_file.write(os.path.basename(self.variables_files[var_index]))

    def init_saving(self, vars_prefix: str) -> None:
        if self.saver_max_to_keep == 1:
            self.variables_files = [vars_prefix]
        else:
            self.variables_files = ["{}.{}".format(vars_prefix, i) for i in range(self.saver_max_to_keep)]
        self._best_vars_file = "{}.best".format(vars_prefix)

    def validation_hook(self, score: float, epoch: int, batch: int) -> None:
        if self._is_better(score, self.best_score):
            self.best_score, self.best_score_epoch, self.best_score_batch = score, epoch, batch
        worst_index = self._argworst(self.saved_scores)
        if self._is_better(score, self.saved_scores[worst_index]):
            self.save(self.variables_files[worst_index])
            self.saved_scores[worst_index] = score
            if self.best_score == score:
                self._update_best_vars(worst_index)
                self.best_score_index = worst_index

    # -- execution (tf_manager.py:158-225) -----------------------------------------------
    def _run_executables(self, feed_dict: Dict, executables: List[GraphExecutor.Executable], ahead=None) -> None:
        all_fetches = {}
        feed_dicts: List[Dict] = [{} for _ in self.sessions]
        pending = [ex for ex in executables if ex.result is None]
        for executable in pending:
            fetches, add_feed_dicts = executable.next_to_execute()
            all_fetches[executable] = fetches
            if add_feed_dicts:
                for fdict, add_fd in zip(feed_dicts, add_feed_dicts):
                    fdict.update(add_fd)
        for fdict in feed_dicts:
            fdict.update(feed_dict)
        # executables that combine the models of all sessions on the device (beam-search ensembles)
        ensemble_results = {}
        for executable in pending:
            if getattr(executable, "ensemble", False):
                ctxs = [RunContext(sess, dict(fd)) for sess, fd in zip(self.sessions, feed_dicts)]
                with torch.no_grad():
                    ensemble_results[executable] = _to_host(executable.run_ensemble(ctxs))
        session_results = [sess.run(all_fetches, feed_dict=fd, ahead=ahead if len(self.sessions) == 1 else None)
                           for sess, fd in zip(self.sessions, feed_dicts)]
        for executable in pending:
            if executable in ensemble_results:
                executable.collect_results([ensemble_results[executable] for _ in self.sessions])
            else:
                executable.collect_results([res[executable] for res in session_results])

    def execute(self, batch, feedables: Set[Feedable], runners: Sequence[GraphExecutor],
                train: bool = False, compute_losses: bool = True,
                summaries: bool = True, lookahead=None) -> List[ExecutionResult]:
        """tf_manager.py:188-225.  ``lookahead`` (not in the reference) names the batch that will be executed
        NEXT with the same runners: its encoder side (what ``GraphExecutor.ahead_fetches`` lists: encoder states,
        attention keys, initial decoder states) is evaluated on a second stream while this batch decodes -- both
        are latency-bound and leave most CUs idle (runtime.Session.run).  Inference with one session only."""
        default_feed_dict = _feed_dicts(batch, feedables, train=train)
        executables = [runner.get_executable(compute_losses=compute_losses, summaries=summaries,
                                             num_sessions=len(self.sessions)) for runner in runners]
        ahead = None
        if lookahead is not None and not train and len(self.sessions) == 1:
            fetches = [f for runner in runners for f in getattr(runner, "ahead_fetches", lambda: [])()]
            if fetches:
                ahead = (fetches, _feed_dicts(lookahead, feedables, train=False))
        while True:
            while not all(ex.result is not None for ex in executables):
                self._run_executables(default_feed_dict, executables, ahead)
                ahead = None
            if train:                # (a training step's error word travels with its losses: Session.recover_training)
                break
            # inference: the results are on the host, the stream is idle -- reading the sessions' error words costs a
            # few microseconds.  A set word = a GRU time loop launched as one cluster kernel gave up (something else
            # held compute units): that session continues on the per-step path (one warning) and the batch is run
            # again; only a failure of the fallback itself raises.
            failed = [sess for sess in self.sessions if sess.cluster_failure()]
            if not failed:
                break
            for sess in failed:
                sess.demote_cluster_loops()
            executables = [runner.get_executable(compute_losses=compute_losses, summaries=summaries,
                                                 num_sessions=len(self.sessions)) for runner in runners]
        return [ex.result for ex in executables]

    # -- variables ---------------------------------------------------------------------------
    def initialize_sessions(self) -> None:
        """Create and initialise every variable of every registered model part
        (== global_variables_initializer + Saver over all globals)."""
        parts = registered_parts()
        from . import ops
        if ops.PROJ_SPLIT and self.sessions and self.sessions[0].device.type == "cuda":
            ops.proj_split_forget()          # a new model: no weight matrix of an earlier one stays registered
        for sess in self.sessions:
            for part in parts:
                part.declare_variables(sess.store)
            sess.store.finalize()
        self.saver = True

    def initialize_model_parts(self, runners: Sequence[GraphExecutor]) -> None:
        if any(not hasattr(r, "parameterizeds") for r in runners):
            raise TypeError("Args to initialize_model_parts must be trainers or runners")
        parameterizeds = set.union(*[rnr.parameterizeds for rnr in runners])
        for coder in parameterizeds:
            for session in self.sessions:
                coder.load(session)

    def save(self, variable_files: Union[str, List[str]]) -> None:
        if self.saver is None:
            raise RuntimeError("Saver uninitialized")
        from . import distributed as dist
        dp = dist.current()
        for sess in self.sessions:
            # nothing is written while a training step's error flag is still unread: a step whose time loop gave up
            # is run again first (its update was skipped on the device, so the variables are clean either way -- but
            # global_step and the steps enqueued since belong to the checkpoint too)
            sess.settle_training()
            if dp is not None and dp.sharded_active() and sess.store.adam_m is not None:
                # sharded optimizer: every rank holds the slots of its own slices; a checkpoint carries all of them
                # (a collective: every rank calls save, as every rank runs every training step)
                dp.gather_optimizer_slots(sess.store, sess.store.adam_m, sess.store.adam_v)
        if isinstance(variable_files, str) and len(self.sessions) == 1:
            self.sessions[0].store.save(variable_files, fmt=self.checkpoint_format,
                                        global_step=self.sessions[0].global_step)
            return
        if isinstance(variable_files, str):
            variable_files = ["{}.{}".format(variable_files, i) for i in range(len(self.sessions))]
        if len(variable_files) != len(self.sessions):
            raise Exception("Provided {} files for saving {} sessions.".format(
                len(variable_files), len(self.sessions)))
        for sess, file_name in zip(self.sessions, variable_files):
            sess.store.save(file_name, fmt=self.checkpoint_format, global_step=sess.global_step)

    def restore(self, variable_files: Union[str, List[str]]) -> None:
        if self.saver is None:
            raise RuntimeError("Saver uninitialized")
        if isinstance(variable_files, str):
            variable_files = [variable_files]
        if len(variable_files) != len(self.sessions):
            raise Exception("Provided {} files for restoring {} sessions.".format(
                len(variable_files), len(self.sessions)))
        for sess, file_name in zip(self.sessions, variable_files):
            info = sess.store.load(file_name) or {}
            step = info.get("global_step")
            if step is not None:
                # the Saver restores ALL global variables (tf_manager.py:257-261): global_step and the optimizer's
                # slots / beta powers continue where the checkpoint left them (learning-rate schedules, Adam's
                # bias correction, dropout salts)
                sess.global_step = step
                owner = sess.__dict__.get("_adam_owner", {}).get(id(sess.store))
                state = getattr(owner, "_adam", {}).get(id(sess.store)) if owner is not None else None
                if state is not None:
                    state["applied"] = step

    def restore_best_vars(self) -> None:
        assert self.best_score_index is not None
        self.restore(self.variables_files[self.best_score_index])


def _feed_dicts(dataset, coders: Set[Feedable], train: bool = False) -> Dict:
    """Merged feed dict of ``coders`` for one batch, remembered on the batch object: a batch that was
    prepared ahead of time (input_pipeline.Prefetcher) or is executed again hands back the SAME host
    arrays, which ``Session.to_device`` recognises as already resident on the device."""
    cache = getattr(dataset, "__dict__", {}).setdefault("_feed_cache", {}) if hasattr(dataset, "__dict__") else {}
    ordered = sorted(coders, key=id)
    key = (train, tuple(id(c) for c in ordered))
    hit = cache.get(key)
    if hit is None or len(hit[0]) != len(ordered) or any(a is not b for a, b in zip(hit[0], ordered)):
        res: Dict = {}
        for coder in coders:
            res.update(coder.feed_dict(dataset, train=train))
        if len(cache) >= 8:
            cache.clear()
        hit = cache[key] = (ordered, res)       # the parts are kept alive with the entry: ids cannot be recycled
    return dict(hit[1])


def get_default_tf_manager() -> TensorFlowManager:
    return TensorFlowManager(num_sessions=1, num_threads=4)
