"""GreedyRunner (mirror of neuralmonkey/runners/runner.py:18-90).

The reference ships the [T,B,V] log-prob tensor to the host every batch and
arg-maxes it there (819 MB at the benchmark shape).  For one
session that equals the on-device argmax of the logits, which is what is
fetched here; with several sessions (ensembles) the log-prob tensors are
fetched and combined with ``np.logaddexp`` exactly as the reference does."""
from typing import Any, Callable, Dict, List

import numpy as np

from .base_runner import BaseRunner, NextExecute

Postprocessor = Callable[[List[List[str]]], List[List[str]]]


def ensemble_logprobs(per_session: List[np.ndarray]) -> np.ndarray:
    """[T,B,V] log-probabilities of several sessions -> log of their summed probabilities, step by step along the
    loop of the FIRST session (runners/runner.py:37-49; pinned by the reference-executed fixture
    "greedy_runner_ensemble" under tests/golden).  Every session's loop stops on its own: one that stopped earlier
    contributes to its own steps only; one that ran LONGER than the first makes the reference index past the end of
    its per-step list -- the same IndexError is raised here, before anything is combined."""
    steps = per_session[0].shape[0]
    if any(logprobs.shape[0] > steps for logprobs in per_session):
        raise IndexError("list index out of range")
    combined = np.full(per_session[0].shape, -np.inf, dtype=per_session[0].dtype)
    for logprobs in per_session:
        upto = logprobs.shape[0]
        combined[:upto] = np.logaddexp(combined[:upto], logprobs)
    return combined


class GreedyRunner(BaseRunner):
    class Executable(BaseRunner.Executable):
        def next_to_execute(self) -> NextExecute:
            dec = self.executor.decoder
            if self.num_sessions == 1:
                fetches = {"decoded_symbols": dec.decoded_symbols}
            else:
                fetches = {"decoded_logprobs": dec.runtime_logprobs}
            if self.compute_losses:
                fetches["train_xent"] = dec.train_loss
                fetches["runtime_xent"] = dec.runtime_loss
            else:
                fetches["train_xent"] = 0.0
                fetches["runtime_xent"] = 0.0
            return fetches, []

        def collect_results(self, results: List[Dict]) -> None:
            losses = [sum(float(session[key]) for session in results) for key in ("train_xent", "runtime_xent")]
            if self.num_sessions == 1:
                steps = list(results[0]["decoded_symbols"])
            else:
                steps = list(np.argmax(ensemble_logprobs([s["decoded_logprobs"] for s in results]), axis=2))
            sentences = self.executor.vocabulary.vectors_to_sentences(steps)
            if self.executor.postprocess is not None:
                sentences = self.executor.postprocess(sentences)
            self.set_runner_result(outputs=sentences, losses=losses, summaries=None)

    def __init__(self, output_series: str, decoder, postprocess: Postprocessor = None) -> None:
        super().__init__(output_series, decoder)
        self.postprocess = postprocess
        self.vocabulary = self.decoder.vocabulary

    def ahead_fetches(self) -> List[Any]:
        from .base_runner import encoder_side_fetches
        return encoder_side_fetches(self.decoder)

    @property
    def fetches(self) -> Dict[str, Any]:
        return {"decoded_symbols": self.decoder.decoded_symbols,
                "train_xent": self.decoder.train_loss,
                "runtime_xent": self.decoder.runtime_loss}

    @property
    def loss_names(self) -> List[str]:
        return ["train_xent", "runtime_xent"]
