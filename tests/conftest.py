import os
import sys

# Two thread pools share the host's cores in these tests: torch's OpenMP threads (the oracles) and NumPy's BLAS threads
# (the reference executed on the NumPy-eager TensorFlow stand-in).  Idle OpenMP threads that spin for work starve the
# other pool: the same suite took 70 s or 480 s from run to run.  Waiting threads sleep instead (set before either
# library is imported; an explicit setting in the environment wins).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("KMP_BLOCKTIME", "0")

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# GPU tests that spend more than ~10 s each, nearly all of it in the CPU oracle at the benchmarked sizes
# (profiles/r05_gpu_suite_durations.txt: 290 s of the suite's 404 s).  Iterating on a kernel runs
# `pytest -m "gpu and not slow"` (~2 minutes); the full `-m gpu` suite is for the commits that end a piece of work.
SLOW = (
    "test_transformer_fullsize_gpu.py::test_greedy_and_beam_through_the_cache_match_the_prefix_recompute",
    "test_transformer_fullsize_gpu.py::test_training_step_loss_and_every_gradient",
    "test_transformer_fullsize_gpu.py::test_logits_on_the_benchmarked_weights_meet_1e_4_outright",
    "test_fullsize_parity_gpu.py::test_training_step_gradients_and_adam_match_the_oracle",
    "test_fullsize_parity_gpu.py::test_beam_search_every_selection_of_all_50_steps_is_accounted_for",
    "test_captioning_fullsize_gpu.py::test_every_beam_selection_of_all_50_steps_is_accounted_for",
    "test_transformer_gpu.py::test_transformer_base_width_matches_the_oracle",
    "test_proj_split_gpu.py::test_decoding_parity_holds_under_the_split_projection",
)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: GPU tests dominated by the CPU oracle at full size (see SLOW)")


def pytest_collection_modifyitems(items):
    for item in items:
        if any(name in item.nodeid for name in SLOW):
            item.add_marker(pytest.mark.slow)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from neuralmonkey_amd import _lib
    _lib.load()            # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")
