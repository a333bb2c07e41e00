"""``PlainRunner``: token lists from the decoder's ``decoded`` tensor (interface of
neuralmonkey/runners/plain_runner.py:19-62).

``decoded`` is the argmax over the runtime logits with <pad> excluded (autoregressive.py:341-349); the
vocabulary row-scan kernel computes it on the device, so only [T,B] int32 symbols cross to the host,
not the logits.  Reported losses: ``train_loss`` and ``runtime_loss`` of the decoder.
"""
from typing import Any, Callable, Dict, List

from .base_runner import BaseRunner

Postprocessor = Callable[[List[List[str]]], List[List[str]]]
LOSSES = ("train_loss", "runtime_loss")


class PlainRunner(BaseRunner):
    def __init__(self, output_series: str, decoder, postprocess: Postprocessor = None) -> None:
        super().__init__(output_series, decoder)
        self.postprocess = postprocess

    @property
    def loss_names(self) -> List[str]:
        return list(LOSSES)

    @property
    def fetches(self) -> Dict[str, Any]:
        wanted = ("decoded",) + LOSSES
        return {key: getattr(self.decoder, key) for key in wanted}

    class Executable(BaseRunner.Executable):
        def collect_results(self, results: List[Dict]) -> None:
            if len(results) != 1:
                raise ValueError("PlainRunner needs exactly 1 execution result, got {}".format(len(results)))
            (fetched,), runner = results, self.executor
            sentences = runner.decoder.vocabulary.vectors_to_sentences(list(fetched["decoded"]))
            if runner.postprocess is not None:
                sentences = runner.postprocess(sentences)
            self.set_runner_result(outputs=sentences, losses=[float(fetched[name]) for name in LOSSES])
