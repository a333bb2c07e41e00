"""PlainRunner (mirror of neuralmonkey/runners/plain_runner.py:19-62): the decoder's ``decoded``
tensor -- argmax over the runtime logits with <pad> excluded (autoregressive.py:341-349), computed
on the device by the vocabulary row-scan kernel -- turned into token lists."""
from typing import Any, Callable, Dict, List

from .base_runner import BaseRunner

Postprocessor = Callable[[List[List[str]]], List[List[str]]]


class PlainRunner(BaseRunner):
    class Executable(BaseRunner.Executable):
        def collect_results(self, results: List[Dict]) -> None:
            if len(results) != 1:
                raise ValueError("PlainRunner needs exactly 1 execution result, got {}".format(len(results)))
            vocabulary = self.executor.decoder.vocabulary
            decoded_tokens = vocabulary.vectors_to_sentences(list(results[0]["decoded"]))
            if self.executor.postprocess is not None:
                decoded_tokens = self.executor.postprocess(decoded_tokens)
            self.set_runner_result(outputs=decoded_tokens,
                                   losses=[float(results[0]["train_loss"]), float(results[0]["runtime_loss"])])

    def __init__(self, output_series: str, decoder, postprocess: Postprocessor = None) -> None:
        super().__init__(output_series, decoder)
        self.postprocess = postprocess

    @property
    def fetches(self) -> Dict[str, Any]:
        return {"decoded": self.decoder.decoded, "train_loss": self.decoder.train_loss,
                "runtime_loss": self.decoder.runtime_loss}

    @property
    def loss_names(self) -> List[str]:
        return ["train_loss", "runtime_loss"]
