"""tools/avg_checkpoints.py (scripts/avg_checkpoints.py of the reference) on TensorFlow tensor bundles
and on .npz checkpoints.  Host-side only."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("avg_checkpoints", os.path.join(ROOT, "tools", "avg_checkpoints.py"))
avg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(avg)


def _values(rng, shift):
    return {"encoder/rnn/kernel": (rng.standard_normal((5, 7)) + shift).astype(np.float32),
            "decoder/state_to_word_b": (rng.standard_normal(9) + shift).astype(np.float32),
            "attention/attn_bias": np.asarray(shift, dtype=np.float32)}


@pytest.mark.parametrize("bundle", [True, False])
def test_average_of_three_checkpoints(tmp_path, bundle):
    from neuralmonkey_amd import tf_bundle
    rng = np.random.default_rng(3)
    paths, all_values = [], []
    for i in range(3):
        vals = _values(rng, float(i))
        path = str(tmp_path / "variables.{}".format(i))
        if bundle:
            tf_bundle.write_bundle(path, dict(vals, global_step=np.asarray(100 + i, dtype=np.int64)))
        else:
            np.savez(path + ".npz", **{k.replace("/", "|"): v for k, v in vals.items()})
        paths.append(path)
        all_values.append(vals)
    out = str(tmp_path / "averaged")
    avg.write_checkpoint(out, avg.average(paths), bundle)
    got = avg.read_checkpoint(out)
    for name in all_values[0]:
        want = np.mean([v[name].astype(np.float64) for v in all_values], axis=0)
        assert got[name].dtype == np.float32 and got[name].shape == all_values[0][name].shape
        assert np.abs(got[name] - want).max() < 1e-6
    if bundle:
        assert int(got["global_step"]) == 0               # not averaged: the output starts at step 0
    with pytest.raises(ValueError, match="do not exist"):
        avg.average([paths[0], str(tmp_path / "nope")])
