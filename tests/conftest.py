import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle's CPU convolutions run on torch's native kernels in EVERY test run.  test_oracle_golden.py used to switch
    # oneDNN off at import, so whether the oracle (and the targets it builds for the GPU tests) used oneDNN depended on which
    # test modules had been collected -- and an fp16 Deep-Image-Prior trajectory turns a last-bit difference of its target image
    # into 2 % of the loss within three Adam steps (test_dip_step_vs_oracle passed or failed with the selection).
    import torch
    torch.backends.mkldnn.enabled = False


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load
