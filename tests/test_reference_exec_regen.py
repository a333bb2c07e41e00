"""Where the reference tree is present (the build container), the committed fixtures under ``tests/golden/ref_exec``
ARE what ``tests/golden/make_reference_exec_golden.py`` produces from it today: the generator is re-run into a scratch
directory and every array is compared bit for bit.  Skipped where ``/root/reference`` does not exist (the GPU box)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
CASES = ["functions", "beam_body", "rnn_gru", "ms_hier_share_sentinel", "transformer", "transformer_ms_hier",
         "fd_gradients_rnn_gru", "defects"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "neuralmonkey")), reason="no reference tree here")
def test_committed_fixtures_are_what_the_reference_produces(tmp_path):
    env = dict(os.environ, NM_REF_EXEC_OUT=str(tmp_path))
    gen = os.path.join(HERE, "golden", "make_reference_exec_golden.py")
    subprocess.run([sys.executable, gen] + CASES, check=True, env=env, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=600)
    made = sorted(glob.glob(os.path.join(str(tmp_path), "*.npz")))
    assert [os.path.basename(p)[:-4] for p in made if not os.path.basename(p).startswith("_")] == sorted(CASES)
    for path in made:
        name = os.path.basename(path)
        if name.startswith("_"):
            continue
        new, old = np.load(path), np.load(os.path.join(HERE, "golden", "ref_exec", name))
        assert sorted(new.files) == sorted(old.files), name
        for key in new.files:
            assert np.array_equal(new[key], old[key]), "{}: {} differs from the committed fixture".format(name, key)
