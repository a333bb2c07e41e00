"""BeamSearchRunner (mirror of neuralmonkey/runners/beamsearch_runner.py).

Single session: the whole search runs on the device in one ``Session.run``.
Several sessions = an ensemble (beamsearch_runner.py:38-82: the reference does
one Session.run per step per model and averages the step distributions in log
space on the host): here ``BeamSearchDecoder.ensemble_outputs`` runs the whole
ensemble search on the device, one decoder state per model, one shared beam."""
from typing import Any, Callable, Dict, List

import numpy as np

from ..decoders.beam_search_decoder import BeamSearchDecoder
from ..vocabulary import END_TOKEN_INDEX
from .base_runner import BaseRunner, NextExecute


class BeamSearchRunner(BaseRunner):
    class Executable(BaseRunner.Executable):
        def __init__(self, executor: "BeamSearchRunner", compute_losses: bool, summaries: bool,
                     num_sessions: int) -> None:
            super().__init__(executor, compute_losses, summaries, num_sessions)
            self.rank = executor.rank
            self.decoder = executor.decoder
            self.postprocess = executor.postprocess
            self.ensemble = num_sessions > 1     # TensorFlowManager then calls run_ensemble once

        def next_to_execute(self) -> NextExecute:
            if self.ensemble:
                return {}, []
            return {"bs_outputs": self.decoder.outputs}, [{}]

        def run_ensemble(self, ctxs) -> Dict[str, Any]:
            """All sessions' run contexts at once (same device): the ensemble search."""
            return {"bs_outputs": self.decoder.ensemble_outputs(ctxs)}

        def collect_results(self, results: List[Dict]) -> None:
            self.prepare_results(results[0]["bs_outputs"].last_search_step_output)

        def prepare_results(self, output) -> None:
            """beamsearch_runner.py:84-106: hypothesis ``rank``, first token
            dropped, cut at </s>; loss = mean(score) * batch."""
            bs_scores = [s[self.rank - 1] for s in output.scores]
            # hypothesis `rank` of every sentence, time-major without the parent's first symbol; the word lookup
            # and the cut at </s> are vectorised in Vocabulary.vectors_to_sentences
            tok_tb = np.asarray(output.token_ids)[1:, :, self.rank - 1]
            if tok_tb.shape[0] == 0:
                decoded_tokens = [[] for _ in range(tok_tb.shape[1])]
            else:
                decoded_tokens = self.decoder.vocabulary.vectors_to_sentences(tok_tb)
            if self.postprocess is not None:
                decoded_tokens = self.postprocess(decoded_tokens)
            self.set_runner_result(outputs=decoded_tokens,
                                   losses=[float(np.mean(bs_scores) * len(bs_scores))])

    def __init__(self, output_series: str, decoder: BeamSearchDecoder, rank: int = 1,
                 postprocess: Callable[[List[str]], List[str]] = None) -> None:
        super().__init__(output_series, decoder)
        if rank < 1 or rank > decoder.beam_size:
            raise ValueError("Rank of output hypothesis must be between 1 and the beam size ({}), "
                             "was {}.".format(decoder.beam_size, rank))
        self.rank = rank
        self.postprocess = postprocess

    def ahead_fetches(self) -> List[Any]:
        from .base_runner import encoder_side_fetches
        return encoder_side_fetches(self.decoder.parent_decoder)

    @property
    def fetches(self) -> Dict[str, Any]:
        return {"bs_outputs": self.decoder.outputs}

    @property
    def loss_names(self) -> List[str]:
        return ["beam_search_score"]


def beam_search_runner_range(output_series: str, decoder: BeamSearchDecoder, max_rank: int = None,
                             postprocess: Callable[[List[str]], List[str]] = None) -> List[BeamSearchRunner]:
    """beamsearch_runner.py:154-190."""
    if max_rank is None:
        max_rank = decoder.beam_size
    if max_rank > decoder.beam_size:
        raise ValueError("The maximum rank ({}) cannot be bigger than beam size {}."
                         .format(max_rank, decoder.beam_size))
    return [BeamSearchRunner("{}.rank{:03d}".format(output_series, r), decoder, r, postprocess)
            for r in range(1, max_rank + 1)]
