"""XentRunner (mirror of neuralmonkey/runners/xent_runner.py:16-40): per-sentence, per-position
cross entropies of the teacher-forced pass ([B,T], zero on padding), averaged over sessions."""
from typing import Any, Dict, List

import numpy as np

from .base_runner import BaseRunner


class XentRunner(BaseRunner):
    class Executable(BaseRunner.Executable):
        def collect_results(self, results: List[Dict]) -> None:
            xents = np.mean([np.asarray(res["xents"]) for res in results], axis=0)
            self.set_runner_result(outputs=xents.tolist(), losses=[float(np.mean(xents))])

    def __init__(self, output_series: str, decoder) -> None:
        super().__init__(output_series, decoder)

    @property
    def fetches(self) -> Dict[str, Any]:
        return {"xents": self.decoder.train_xents}

    @property
    def loss_names(self) -> List[str]:
        return ["xent"]
