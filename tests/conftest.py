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


# Every GPU parity test that takes the shared engine runs under both arithmetic choices of the fp32 path: the exact kernels
# (compute_dtype "f32", the default and the one bench.py's `value` is measured on) and PA_DTYPE_EMULATED_F32 (fp32 results from
# the bf16 matrix cores, csrc/psgemm.hip) -- same oracle, same tolerances.
FP32_DTYPES = ["f32", "emulated_f32"]


@pytest.fixture(scope="session", params=FP32_DTYPES)
def engine(state_dict, request):
    """One engine per compute dtype for the whole GPU session (pa_create allocates ~2 GB)."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from playaid_core_amd.engine import Engine

    eng = Engine(state_dict, max_batch_frames=64, max_clip_frames=512, compute_dtype=request.param)
    yield eng
    eng.close()
