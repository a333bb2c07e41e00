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
            train_loss = 0.0
            runtime_loss = 0.0
            for sess_result in results:
                train_loss += float(sess_result["train_xent"])
                runtime_loss += float(sess_result["runtime_xent"])
            if self.num_sessions == 1:
                argmaxes = list(results[0]["decoded_symbols"])
            else:
                # runners/runner.py:37-49, to the letter: one entry per step of SESSION 0'S loop; every session's
                # loop stops on its own, so a session that stopped earlier contributes to its own steps only, and one
                # that ran longer makes the list access fail with the reference's IndexError (pinned by
                # the reference-executed fixture "greedy_runner_ensemble" under tests/golden)
                summed = [-np.inf for _ in range(results[0]["decoded_logprobs"].shape[0])]
                for sess_result in results:
                    for i, logprob in enumerate(sess_result["decoded_logprobs"]):
                        summed[i] = np.logaddexp(summed[i], logprob)
                argmaxes = [np.argmax(l, axis=1) for l in summed]
            decoded_tokens = self.executor.vocabulary.vectors_to_sentences(argmaxes)
            if self.executor.postprocess is not None:
                decoded_tokens = self.executor.postprocess(decoded_tokens)
            self.set_runner_result(outputs=decoded_tokens, losses=[train_loss, runtime_loss], summaries=None)

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
