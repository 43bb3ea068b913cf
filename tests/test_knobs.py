"""The library's A/B knobs change HOW a result is computed, never what it is: every setting measured
this round must reproduce the default process's outputs -- bit for bit where the knob only moves work between streams, within
fp32 rounding (1e-5 on log-probabilities, 2e-4 on scores, 4e-2 px on boxes: the paths' own bars) where it swaps a kernel."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KNOBS = [
    ({"PA_DET_LANES": "2"}, True),        # the detector's batch as two halves on two streams: same kernels, same images
    ({"PA_DS_SIDE": "1"}, True),          # ResNet's branch GEMMs on the side stream
    ({"PA_PS_STAGES": "2"}, True),        # a two-stage LDS ring in the emulated kernel: same k order
    ({"PA_DET_EMU_S1": "1"}, False),      # the detector's stride-1 3x3 layers as emulated implicit GEMMs instead of exact Winograd
    ({"PA_DET_EMU_STEM": "0"}, False),    # the emulated detector's stem on the exact fp32 kernel instead of the integer-pixel bf16 one
    ({"PA_PS_RES128": "0", "PA_DET_EMU_S1": "1"}, False),   # ... with 64-channel residual tiles
    ({"PA_DET_UP_FUSE": "0"}, True),      # up-sampling layers as passes of their own
    ({"PA_DET_BLOCK": "8"}, False),       # the large-map layers over 8 images at a time (Winograd's split K follows the batch size)
]


def _run(env_extra, path):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "knob_worker.py"), path], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


@pytest.fixture(scope="module")
def default_outputs(tmp_path_factory):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _run({}, str(tmp_path_factory.mktemp("knobs") / "default.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("knob,bitwise", KNOBS, ids=[",".join(f"{k}={v}" for k, v in kn.items()) for kn, _ in KNOBS])
def test_knob_reproduces_the_default_outputs(default_outputs, tmp_path, knob, bitwise):
    got = _run(knob, str(tmp_path / "knob.npz"))
    for key in ("rows_f32", "rows_emulated_f32", "logp_f32", "logp_emulated_f32"):
        a, b = got[key], default_outputs[key]
        assert a.shape == b.shape
        if bitwise:
            assert np.array_equal(a, b), (knob, key, float(np.abs(a - b).max()))
        elif key.startswith("rows"):
            assert np.abs(a[..., :4] - b[..., :4]).max() <= 4e-2 and np.abs(a[..., 4:] - b[..., 4:]).max() <= 2e-4, (knob, key)
        else:
            assert np.abs(a - b).max() <= 1e-5, (knob, key)
