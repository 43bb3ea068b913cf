"""f4: the reference's ResnetTransformerDetector
(models/resnet_transformer_detector.py:26-141). CPU: the oracle's reading of the un-batch_first encoder against a
literal numpy restatement, the time encoding, the layer table and blob packing. GPU: the ResNet-50 table on the
engine's convolution kernels and the encoder head against the oracle, through the host mirror."""
import numpy as np
import pytest
import torch

from oracle import resformer as oracle_rf
from playaid_core_amd import synth

TOL = 1e-4  # fp32 operator (north_star)


@pytest.fixture(scope="module")
def rf_sd():
    return synth.make_resformer_state_dict(seed=2468, num_actions=63, sequence_length=7)


def _inputs(b, s, seed=5):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.integers(0, 256, size=(b, s, 3, 128, 128)).astype(np.float32) / 255.0)


def test_time_encoding_matches_the_reference_formula():
    from playaid_core_amd.resnet_transformer_detector import time_encoding

    enc = time_encoding(7)
    assert enc.shape == (7, 9) and enc.dtype == np.float32
    x = np.linspace(0, 1, 7)
    np.testing.assert_allclose(enc[:, 0], x, atol=1e-7)
    for i in range(4):
        np.testing.assert_allclose(enc[:, 1 + 2 * i], np.cos(np.pi * x * 2 ** i), atol=2e-6)
        np.testing.assert_allclose(enc[:, 2 + 2 * i], np.sin(np.pi * x * 2 ** i), atol=2e-6)


def test_oracle_encoder_attends_across_windows(rf_sd):
    rng = np.random.default_rng(3)
    feats = rng.standard_normal((4, 7, 256))
    enc = oracle_rf._encoder(rf_sd, torch.float64)
    with torch.no_grad():
        live = enc(torch.from_numpy(feats)).numpy()
    np.testing.assert_allclose(live, oracle_rf.encoder_literal(feats, rf_sd), rtol=0, atol=1e-10)
    with torch.no_grad():
        alone = enc(torch.from_numpy(feats[1:2])).numpy()
    assert np.abs(alone[0] - live[1]).max() > 1e-3   # window 1 alone != window 1 among four: dimension 0 is the sequence


def test_table_and_blob_layout(rf_sd):
    from playaid_core_amd import _lib
    from playaid_core_amd.resnet_transformer_detector import build_resnet50_table, pack_encoder_blob

    descs, bufs, weights, dim = build_resnet50_table(rf_sd)
    assert dim == 2048 and len(descs) == 1 + 16 * 3 + 4 + 1
    assert descs[0]["kind"] == 1 and descs[-1]["kind"] == 2 and all(d["kind"] == 0 for d in descs[1:-1])
    # every bordered buffer has one geometry; 3x3 convolutions read bordered buffers only
    seen = {}
    for d in descs[1:-1]:
        if d["ksize"] == 3:
            assert d["in_pad"] == 1
        if d["in_pad"]:
            assert seen.setdefault(d["in_buf"], (d["in_hw"], d["cin"])) == (d["in_hw"], d["cin"])
    # folded first bottleneck conv: w * gamma / sqrt(var + eps), [cout][ky][kx][cin]
    d = descs[1]
    w = rf_sd["model.resnet.layer1.0.conv1.weight"].astype(np.float64)
    g, v = rf_sd["model.resnet.layer1.0.bn1.weight"].astype(np.float64), rf_sd["model.resnet.layer1.0.bn1.running_var"].astype(np.float64)
    want = (w * (g / np.sqrt(v + 1e-5))[:, None, None, None]).transpose(0, 2, 3, 1).reshape(-1).astype(np.float32)
    np.testing.assert_array_equal(weights[d["w_off"]:d["w_off"] + want.size], want)
    blob = pack_encoder_blob(rf_sd, 63, 7)
    hdr = blob[:64].view(np.int32)
    assert list(hdr[:10]) == [_lib.PA_ENCODER_MAGIC, 1, 2048, 247, 7, 9, 8, 3, 2048, 63]
    floats = blob[64:].view(np.float32)
    np.testing.assert_array_equal(floats[:247 * 2048], rf_sd["model.resnet_ffn.weight"].reshape(-1))
    np.testing.assert_array_equal(floats[-63:], rf_sd["model.classifier.bias"])
    with pytest.raises(KeyError):
        build_resnet50_table({k: v for k, v in rf_sd.items() if "layer3.4.conv2" not in k})


def test_oracle_forward_shape(rf_sd):
    lp = oracle_rf.forward(_inputs(2, 7), rf_sd)
    assert lp.shape == (2, 7, 63)
    np.testing.assert_allclose(torch.exp(lp).sum(dim=2).numpy(), 1.0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "emulated_f32"])
def test_resformer_detector_matches_oracle(rf_sd, dtype):
    from playaid_core_amd.resnet_transformer_detector import ResnetTransformerDetector

    actions = [f"a{i}" for i in range(63)]
    model = ResnetTransformerDetector(actions, sequence_length=7, state_dict=rf_sd, max_rows=70, compute_dtype=dtype).eval()
    try:
        # backbone alone: pooled ResNet-50 features
        x = _inputs(3, 7, seed=1)
        want_f = oracle_rf.resnet50_features(x.reshape(21, 3, 128, 128), rf_sd).numpy()
        got_f = model._net.forward(x.reshape(21, 3, 128, 128)).cpu().numpy()
        assert np.abs(got_f - want_f).max() <= 1e-4 * max(1.0, np.abs(want_f).max())
        for b in (1, 3, 10):
            x = _inputs(b, 7, seed=20 + b)
            want = oracle_rf.forward(x, rf_sd).numpy()
            got = model(x).numpy()
            assert got.shape == (b, 7, 63)
            assert np.abs(got - want).max() <= TOL, np.abs(got - want).max()
            assert (got.argmax(2) == want.argmax(2)).all()
        # attention across windows is live
        x = _inputs(2, 7, seed=99)
        assert np.abs(model(x).numpy()[1] - model(x[1:2]).numpy()[0]).max() > 1e-4
        with pytest.raises(ValueError):
            model(_inputs(2, 5))    # the checkpoint encodes 7 frame slots
        with pytest.raises(ValueError):
            model(_inputs(11, 7))   # 77 rows > max_rows
    finally:
        model.close()
