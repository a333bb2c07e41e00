"""XentRunner: the cross entropy of every target position of the teacher-forced pass ([B,T], zero on padding) as a
runner output -- what neuralmonkey/runners/xent_runner.py:16-40 fetches from ``decoder.train_xents``.  With several
sessions the tables are averaged element by element; the reported loss is the mean of that table."""
from typing import Any, Dict, List

import numpy as np

from .base_runner import BaseRunner

XENT_FETCH = "xents"


class XentRunner(BaseRunner):
    loss_names = ["xent"]

    class Executable(BaseRunner.Executable):
        def collect_results(self, results: List[Dict]) -> None:
            table = np.stack([np.asarray(session[XENT_FETCH]) for session in results]).mean(axis=0)
            self.set_runner_result(outputs=table.tolist(), losses=[float(table.mean())])

    def __init__(self, output_series: str, decoder) -> None:
        BaseRunner.__init__(self, output_series, decoder)

    @property
    def fetches(self) -> Dict[str, Any]:
        return {XENT_FETCH: self.decoder.train_xents}
