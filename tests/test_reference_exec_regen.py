"""Where the reference tree is present (the build container), the committed fixtures under ``tests/golden/ref_exec``
ARE what ``tests/golden/make_reference_exec_golden.py`` produces from it today: the generator is re-run into a scratch
directory -- every one of the fixtures -- and every array is compared bit for bit.  Skipped where ``/root/reference`` does not exist (the GPU box)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "neuralmonkey")), reason="no reference tree here")
def test_committed_fixtures_are_what_the_reference_produces(tmp_path):
    """ALL fixtures (the generator without arguments: ~30 s), not a sample of them."""
    env = dict(os.environ, NM_REF_EXEC_OUT=str(tmp_path))
    gen = os.path.join(HERE, "golden", "make_reference_exec_golden.py")
    subprocess.run([sys.executable, gen], check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   timeout=900)
    made = sorted(os.path.basename(p) for p in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    committed = sorted(os.path.basename(p) for p in glob.glob(os.path.join(HERE, "golden", "ref_exec", "*.npz")))
    assert made == committed and len(made) >= 57, sorted(set(made) ^ set(committed))
    for name in made:
        new, old = np.load(os.path.join(str(tmp_path), name)), np.load(os.path.join(HERE, "golden", "ref_exec", name))
        assert sorted(new.files) == sorted(old.files), name
        for key in new.files:
            assert np.array_equal(new[key], old[key]), "{}: {} differs from the committed fixture".format(name, key)
