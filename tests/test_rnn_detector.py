"""f4: the reference's RNNActionDetector (models/rnn_action_detector.py:55-95).
CPU: the oracle's reading of nn.LSTM without batch_first against a literal numpy recurrence, blob packing.
GPU: pa_backbone_windows + pa_lstm_forward against the oracle, through the host mirror."""
import numpy as np
import pytest
import torch

from oracle import rnn as oracle_rnn
from playaid_core_amd import synth

TOL = 1e-4  # fp32 operator, same bar as the Conv1d model (north_star)


def _inputs(b, s, seed=5):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.integers(0, 256, size=(b, s, 3, 128, 128)).astype(np.float32) / 255.0))


def test_oracle_lstm_runs_over_windows_like_the_reference():
    sd = synth.make_rnn_state_dict(seed=11, num_actions=9)
    rng = np.random.default_rng(3)
    feats = rng.standard_normal((5, 7, 300))
    lit = oracle_rnn.lstm_literal(feats, sd)
    mod = oracle_rnn._lstm_module(sd, torch.float64)
    with torch.no_grad():
        live, _ = mod(torch.from_numpy(feats), None)
    assert live.shape == (5, 7, 512)
    np.testing.assert_allclose(live.numpy(), lit, rtol=0, atol=1e-12)
    # the state really crosses windows: window 1 alone differs from window 1 after window 0
    with torch.no_grad():
        alone, _ = mod(torch.from_numpy(feats[1:2]), None)
    assert np.abs(alone.numpy()[0] - lit[1]).max() > 1e-3


def test_oracle_forward_shape_and_normalisation():
    sd = synth.make_rnn_state_dict(seed=11, num_actions=9)
    lp = oracle_rnn.forward(_inputs(2, 3), sd)
    assert lp.shape == (6, 9)
    np.testing.assert_allclose(torch.exp(lp).sum(dim=1).numpy(), 1.0, atol=1e-5)


def test_lstm_blob_layout():
    from playaid_core_amd import _lib
    from playaid_core_amd.rnn_action_detector import backbone_state_dict, pack_lstm_blob

    sd = synth.make_rnn_state_dict(seed=11, num_actions=9)
    blob = pack_lstm_blob(sd, 9)
    hdr = blob[:32].view(np.int32)
    assert list(hdr[:6]) == [_lib.PA_LSTM_MAGIC, 1, 300, 512, 3, 9]
    floats = blob[32:].view(np.float32)
    n0 = 2048 * 300
    np.testing.assert_array_equal(floats[:n0], sd["lstm.weight_ih_l0"].reshape(-1))
    np.testing.assert_array_equal(floats[-9:], sd["action_decoder.2.bias"])
    assert floats.size == 3 * (2048 * 512 + 2 * 2048) + 2048 * 300 + 2 * 2048 * 512 + 128 * 512 + 128 + 9 * 128 + 9
    bsd = backbone_state_dict(sd)
    assert bsd["model.cnn2d.fc.weight"].shape == (1000, 512)
    np.testing.assert_array_equal(bsd["model.cnn2d.fc.weight"][:300], sd["resnet.fc.0.weight"])
    assert not bsd["model.cnn2d.fc.weight"][300:].any() and not bsd["model.cnn2d.fc.bias"][300:].any()
    with pytest.raises(ValueError):
        bad = dict(sd)
        bad["lstm.weight_hh_l1"] = np.zeros((2048, 300), np.float32)
        pack_lstm_blob(bad, 9)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "emulated_f32"])
def test_rnn_detector_matches_oracle(dtype):
    from playaid_core_amd.rnn_action_detector import RNNActionDetector

    actions = [f"a{i}" for i in range(63)]
    sd = synth.make_rnn_state_dict(seed=4321, num_actions=63)
    model = RNNActionDetector("byleth", actions, state_dict=sd, max_rows=64, compute_dtype=dtype).eval()
    try:
        for b, s in ((1, 7), (4, 7), (3, 4)):   # the vis script's shape (B = 1), several windows, another S
            x = _inputs(b, s, seed=b * 10 + s)
            want = oracle_rnn.forward(x, sd).numpy()
            got = model(x).numpy()
            assert got.shape == (b * s, 63)
            assert np.abs(got - want).max() <= TOL, np.abs(got - want).max()
            assert (got.argmax(1) == want.argmax(1)).all()
        # the recurrence over windows is live: window 1 of a 2-window call != the same window alone
        x = _inputs(2, 7, seed=99)
        both = model(x).numpy()
        alone = model(x[1:2]).numpy()
        assert np.abs(both[7:] - alone).max() > 1e-4
        with pytest.raises(ValueError):
            model(_inputs(10, 7))   # 70 rows > max_rows
    finally:
        model.close()


@pytest.mark.gpu
def test_lstm_barrier_timeout_is_reported_once_and_the_handle_recovers():
    """The failure path of the per-layer LSTM kernel (``csrc/lstm.hip``), driven by the debug knob ``PA_LSTM_FORCE_TIMEOUT=1`` (one
    workgroup withholds its granules of step 0, so every other one gives up after 20 ms): that call's log-probabilities are NaN,
    ``check()`` raises ONCE with the status ``pa_lstm_last_status`` reports (read only behind the forward's own copies), and
    the handle then launches one kernel per time step -- results equal to a handle made with ``PA_LSTM_STEPS=1``. Runs in a
    child process: both knobs are read once per process."""
    import json
    import os
    import subprocess
    import sys

    code = r"""
import json, sys
import numpy as np, torch
sys.path.insert(0, %r)
from playaid_core_amd import synth
from playaid_core_amd.engine import EngineError
from playaid_core_amd.rnn_action_detector import RNNActionDetector
actions = [f"a{i}" for i in range(63)]
sd = synth.make_rnn_state_dict(seed=4321, num_actions=63)
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.integers(0, 256, (2, 7, 3, 128, 128)).astype(np.float32) / 255)
m = RNNActionDetector("byleth", actions, state_dict=sd, max_rows=64).eval()
out = {}
try:
    first = m(x.cuda())          # device input: enqueued only, nothing raised here
    torch.cuda.synchronize()
    out["first_nan"] = bool(torch.isnan(first).all())
    try:
        m.check(); out["raised"] = False
    except EngineError as e:
        out["raised"] = True; out["msg"] = str(e)
    try:
        m.check(); out["raised_twice"] = False
    except EngineError:
        out["raised_twice"] = True
    out["second"] = m(x).numpy().tolist()   # per-step launches from here on
finally:
    m.close()
print("RESULT" + json.dumps(out))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1][6:])

    forced = run({"PA_LSTM_FORCE_TIMEOUT": "1"})
    assert forced["first_nan"], "the timed-out call must return NaN rows, not stale or partial numbers"
    assert forced["raised"] and "timed out" in forced["msg"] and not forced["raised_twice"]
    steps = run({"PA_LSTM_STEPS": "1"})
    assert not steps["raised"] and not steps["first_nan"]
    assert np.array_equal(np.asarray(forced["second"]), np.asarray(steps["second"]))


@pytest.mark.gpu
def test_lstm_forward_beside_a_busy_stream_is_correct_or_reported():
    """the per-layer LSTM kernel is an ordinary launch whose workgroups wait for each other; the check that they
    fit looks at an empty device. Here a second stream keeps the chip busy with large fp32 matrix products while the forward runs.
    The contract: either the result equals the quiet run's, or the rows are NaN AND ``check()`` reports the timeout (after which
    the handle launches per time step and is valid again) -- never a silent wrong row."""
    from playaid_core_amd.engine import EngineError
    from playaid_core_amd.rnn_action_detector import RNNActionDetector

    actions = [f"a{i}" for i in range(63)]
    sd = synth.make_rnn_state_dict(seed=4321, num_actions=63)
    model = RNNActionDetector("byleth", actions, state_dict=sd, max_rows=64).eval()
    try:
        x = _inputs(4, 7, seed=77).cuda()
        quiet = model(x)
        torch.cuda.synchronize()
        model.check()
        quiet = quiet.cpu().numpy()
        assert np.isfinite(quiet).all()
        side = torch.cuda.Stream()
        a = torch.randn(8192, 8192, device="cuda")
        busy_out = torch.empty_like(a)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6):      # ~60 ms of chip-filling work on the other stream
                torch.mm(a, a, out=busy_out)
        got = model(x)              # enqueued beside it
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        try:
            model.check()
            reported = False
        except EngineError:
            reported = True
        if reported:
            assert np.isnan(got).all(), "a reported timeout must leave NaN rows, not partial results"
            again = model(x)
            torch.cuda.synchronize()
            model.check()
            assert np.abs(again.cpu().numpy() - quiet).max() <= 1e-5   # (per-step launches from here on: same sums)
        else:
            assert np.isfinite(got).all(), "NaN rows without a reported status"
            assert np.array_equal(got, quiet)
    finally:
        model.close()
