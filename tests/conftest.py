import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def state_dict():
    from playaid_core_amd import synth

    return synth.make_state_dict(seed=1234)


@pytest.fixture(scope="session")
def engine(state_dict):
    """One engine for the whole GPU session (pa_create allocates ~2 GB)."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from playaid_core_amd.engine import Engine

    eng = Engine(state_dict, max_batch_frames=64, max_clip_frames=512)
    yield eng
    eng.close()
