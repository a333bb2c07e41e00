"""Average the variables of several checkpoints into a new one (the reference's
scripts/avg_checkpoints.py, without TensorFlow): TensorFlow tensor bundles (``<prefix>.index`` +
``<prefix>.data-00000-of-00001``, read and written by neuralmonkey_amd/tf_bundle.py) or the engine's
``.npz`` checkpoints.  ``global_step`` is not averaged: the output carries step 0, as the reference's
does (scripts/avg_checkpoints.py:21,68-70).

    python tools/avg_checkpoints.py ckpt.0 ckpt.1 ckpt.2 averaged
"""
import argparse
import os
import re
import sys
from typing import Dict, List

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

IGNORED_PATTERNS = ["global_step"]


def read_checkpoint(path: str) -> Dict[str, np.ndarray]:
    from neuralmonkey_amd import tf_bundle
    if os.path.exists(path + ".index"):
        return tf_bundle.read_bundle(path)
    npz = path if path.endswith(".npz") else path + ".npz"
    if os.path.exists(npz):
        with np.load(npz) as data:
            return {name.replace("|", "/"): data[name] for name in data.files}
    raise ValueError("Provided checkpoints do not exist: {}".format(path))


def average(checkpoints: List[str]) -> Dict[str, np.ndarray]:
    sums: Dict[str, np.ndarray] = {}
    dtypes = {}
    first = read_checkpoint(checkpoints[0])
    names = [n for n in first if not any(re.match(pat, n) for pat in IGNORED_PATTERNS)]
    for path in checkpoints:
        values = first if path == checkpoints[0] else read_checkpoint(path)
        for name in names:
            if name not in values:
                raise KeyError("variable '{}' is missing from checkpoint {}".format(name, path))
            tensor = np.asarray(values[name])
            if name in sums and tensor.shape != sums[name].shape:
                raise ValueError("variable '{}' has shape {} in {} and {} before".format(
                    name, tensor.shape, path, sums[name].shape))
            dtypes[name] = tensor.dtype
            sums[name] = sums.get(name, np.zeros(tensor.shape, np.float64)) + tensor
    return {name: (total / len(checkpoints)).astype(dtypes[name]) for name, total in sums.items()}


def write_checkpoint(path: str, values: Dict[str, np.ndarray], as_bundle: bool) -> None:
    if as_bundle:
        from neuralmonkey_amd import tf_bundle
        tf_bundle.write_bundle(path, dict(values, global_step=np.asarray(0, dtype=np.int64)))
    else:
        np.savez(path if path.endswith(".npz") else path + ".npz",
                 **{name.replace("/", "|"): value for name, value in values.items()})


def main() -> None:
    parser = argparse.ArgumentParser(description=__doc__)
    parser.add_argument("checkpoints", type=str, nargs="+", help="checkpoints to average")
    parser.add_argument("output_path", type=str, help="where the averaged checkpoint goes")
    args = parser.parse_args()
    as_bundle = os.path.exists(args.checkpoints[0] + ".index")
    write_checkpoint(args.output_path, average(args.checkpoints), as_bundle)
    print("Averaged checkpoints saved in {}".format(args.output_path))


if __name__ == "__main__":
    main()
