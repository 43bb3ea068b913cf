"""The Winograd F(2x2, 3x3) convolution kernel (``csrc/wino.hip``) through the C ABI against torch's CPU ``conv2d`` in
float64 -- the operator ``torchvision.resnet18`` / YOLOv5's Bottleneck apply (``cnn_action_detector.py:16,32``;
``ai_runner.py:191-224``). Tolerance: 2e-5 of the layer's largest output (fp32 rounding of a K = 9 cin sum and of the
transforms; the end-to-end bars stay the path's 1e-4 on log-probabilities / scores)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _ref_transform(w):
    """G g G^T in float64, [cout, cin, 4, 4]."""
    G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
    return np.einsum("ij,ocjk,lk->ocil", G, w.astype(np.float64), G)


def test_weight_transform_layout():
    """Host side only: every transformed value sits where the kernel's stage image expects it."""
    from playaid_core_amd import wino

    rng = np.random.default_rng(3)
    for cin, cout in ((8, 32), (16, 64), (24, 96), (64, 128)):
        w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
        bn = 64 if cout % 64 == 0 else 32
        if (cin, cout) == (64, 128):
            bn = 32   # the four-wave form of a layer that also has the 64-channel one
        ug = wino.transform_weights(w, bn=bn)
        u = _ref_transform(w).astype(np.float32)
        gi_n, n_chunks = bn // 16, cin // 8
        img = ug.reshape(cout // bn, n_chunks, 16, gi_n, 16, 8)
        for co in rng.integers(0, cout, 12):
            for ci in rng.integers(0, cin, 6):
                r = co % 16
                got = img[co // bn, ci // 8, :, (co % bn) // 16, r, ((ci % 8) + 4 * (r >> 3)) & 7]
                assert np.array_equal(got, u[co, ci].reshape(16)), (cin, cout, co, ci)
        assert ug.size == 16 * cin * cout


def test_transform_rejects_bad_shapes():
    from playaid_core_amd import _lib

    lib = _lib.load()
    assert lib.pa_wino_weight_floats(12, 64) == 0 and lib.pa_wino_weight_floats(8, 48) == 0
    buf = np.zeros(16, dtype=np.float32)
    assert lib.pa_wino_transform_weights(buf.ctypes.data_as(ctypes.c_void_p), 12, 64, 64, buf.ctypes.data_as(ctypes.c_void_p)) != 0
    assert lib.pa_wino_transform_weights(buf.ctypes.data_as(ctypes.c_void_p), 8, 32, 64, buf.ctypes.data_as(ctypes.c_void_p)) != 0   # 32 channels, 64-wide workgroups
    assert lib.pa_wino_transform_weights(buf.ctypes.data_as(ctypes.c_void_p), 8, 64, 48, buf.ctypes.data_as(ctypes.c_void_p)) != 0


def _case(n, h, w, cin, cout, seed, act=0, residual=False, res_after=False, in_extra=0, out_extra=0, out_pad=1, bn=0, split_k=None):
    from playaid_core_amd import wino

    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((n, cout, h, w)).astype(np.float32) if residual else None
    dev = torch.device("cuda:0")
    xp = torch.zeros((n, h + 2, w + 2, cin + in_extra), dtype=torch.float32)
    xp[:, 1:-1, 1:-1, :cin] = torch.from_numpy(x).permute(0, 2, 3, 1)
    if in_extra:
        xp[:, :, :, cin:] = 7.0  # channels of a wider buffer the kernel must not read
    ops = cout + out_extra
    out = torch.full((n, h + 2 * out_pad, w + 2 * out_pad, ops), -3.0, dtype=torch.float32)
    resp = None
    if residual:
        resp = torch.zeros_like(out)
        resp[:, out_pad:out_pad + h, out_pad:out_pad + w, :cout] = torch.from_numpy(res).permute(0, 2, 3, 1)
    ug = torch.from_numpy(wino.transform_weights(wt, bn=bn)).to(dev)
    got = wino.conv3x3(xp.to(dev), ug, cin, cout, bias=torch.from_numpy(b).to(dev), residual=resp.to(dev) if residual else None,
                       out=out.to(dev), out_pad=out_pad, act=act, res_after=res_after, bn=bn, split_k=split_k).cpu()
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1)
    if residual and not res_after:
        ref = ref + torch.from_numpy(res).double()
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.silu(ref)
    if residual and res_after:
        ref = ref + torch.from_numpy(res).double()
    inner = got[:, out_pad:out_pad + h, out_pad:out_pad + w, :cout].permute(0, 3, 1, 2).double()
    err = float((inner - ref).abs().max() / ref.abs().max())
    # nothing outside the interior / the layer's channels is written
    mask = torch.ones_like(got, dtype=torch.bool)
    mask[:, out_pad:out_pad + h, out_pad:out_pad + w, :cout] = False
    assert bool((got[mask] == -3.0).all()), "the kernel wrote outside its output"
    return err


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [
    (3, 32, 32, 64, 64),      # ResNet-18 layer1 (48 sub-blocks per image row pair)
    (5, 16, 16, 128, 128),    # layer2
    (7, 8, 8, 256, 256),      # layer3: partial last workgroup
    (9, 4, 4, 512, 512),      # layer4
    (2, 24, 40, 128, 128),    # detector P4
    (1, 12, 20, 256, 256),    # detector P5: 15 sub-blocks, one partial workgroup
    (2, 48, 80, 64, 64),
    (1, 96, 160, 32, 32),     # the 32-channel Bottleneck: four-wave workgroups of 32 channels
    (1, 4, 4, 8, 32),
])
def test_wino_conv_matches_conv2d(shape):
    n, h, w, cin, cout = shape
    err = _case(n, h, w, cin, cout, seed=h * 131 + cin)
    assert err <= 2e-5, (shape, err)


def test_workgroup_width_rule():
    """64-channel workgroups where they fill the chip, 32-channel ones where they would leave CUs idle (host-side rule)."""
    from playaid_core_amd import wino

    assert wino.channels_per_workgroup(64, 128 * 64) == 64       # ResNet-18 layer 1 at 128 crops: 512 workgroups
    assert wino.channels_per_workgroup(256, 128 * 4) == 32       # layer 3: 128 workgroups of 64 channels -> 256 of 32
    assert wino.channels_per_workgroup(256, 64 * 15) == 64       # the detector's 12 x 20 map at 64 frames: 240 workgroups
    assert wino.channels_per_workgroup(32, 64 * 24 * 40) == 32   # a 32-channel layer has no 64-channel form
    # (the engine, which launches with split-K scratch, keeps 64-channel workgroups on layers 3 / 4 and splits the input channels)
    with pytest.raises(ValueError):
        wino.channels_per_workgroup(48, 100)


@pytest.mark.gpu
def test_wino_conv_four_wave_workgroups():
    """The 32-channel, four-wave workgroup form on layers that also have the 64-channel one (what the engine picks for small
    launches): same results to the same bar, partial last workgroup included."""
    assert _case(7, 8, 8, 256, 256, 11, bn=32) <= 2e-5
    assert _case(3, 16, 16, 128, 128, 12, act=1, residual=True, bn=32) <= 2e-5
    assert _case(1, 12, 20, 64, 64, 13, act=2, residual=True, res_after=True, bn=32) <= 2e-5


@pytest.mark.gpu
def test_wino_conv_epilogues_and_channel_slices():
    # ResNet BasicBlock tail: residual before the ReLU
    assert _case(2, 16, 16, 64, 64, 1, act=1, residual=True) <= 2e-5
    # YOLOv5 Bottleneck: SiLU, residual after it; input and output are halves of wider [cv1 | cv2] buffers
    assert _case(2, 24, 40, 64, 64, 2, act=2, residual=True, res_after=True, in_extra=64, out_extra=64) <= 2e-5
    assert _case(1, 8, 12, 32, 32, 3, act=2, in_extra=32, out_extra=32, out_pad=0) <= 2e-5
    assert _case(1, 8, 8, 16, 96, 4, act=0, out_pad=2) <= 2e-5


@pytest.mark.gpu
def test_wino_conv_is_bitwise_repeatable_and_batch_independent():
    from playaid_core_amd import wino

    rng = np.random.default_rng(9)
    dev = torch.device("cuda:0")
    n, h, w, c = 6, 16, 16, 128
    xp = torch.zeros((n, h + 2, w + 2, c))
    xp[:, 1:-1, 1:-1] = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32))
    wt = (rng.standard_normal((c, c, 3, 3)) / 30).astype(np.float32)
    ug = torch.from_numpy(wino.transform_weights(wt)).to(dev)
    xd = xp.to(dev)
    a = wino.conv3x3(xd, ug, c, c).cpu()
    b = wino.conv3x3(xd, ug, c, c).cpu()
    assert torch.equal(a, b)
    one = wino.conv3x3(xd[2:3].contiguous(), ug, c, c).cpu()
    assert torch.equal(one[0], a[2]), "an image's result must not depend on the batch around it"


@pytest.mark.gpu
def test_wino_conv_split_k():
    """Few tiles, many channels (ResNet-18's layers 3 and 4 at 128 crops): several workgroups per tile, each over a run of input
    channels, the partial tiles summed in split order by the last to arrive. Same bar as the unsplit launch, the tickets back at
    zero, bitwise repeatable (arrival order must not matter), epilogues and partial tiles included."""
    from playaid_core_amd import wino

    dev = torch.device("cuda:0")
    sk = wino.SplitKScratch(dev)
    assert _case(128, 4, 4, 512, 512, 21, act=1, residual=True, split_k=sk) <= 2e-5          # layer 4: 64 tiles x 4 splits
    assert _case(128, 4, 4, 512, 512, 22, act=1, bn=32, split_k=sk) <= 2e-5                  # ... as 128 four-wave tiles x 4
    assert _case(37, 8, 8, 256, 256, 23, act=1, residual=True, split_k=sk) <= 2e-5           # layer 3 shape, partial last tile
    assert _case(9, 4, 4, 512, 512, 24, act=2, residual=True, res_after=True, out_extra=32, split_k=sk) <= 2e-5   # 8 splits of 8 chunks
    assert _case(2, 16, 16, 64, 64, 25, split_k=sk) <= 2e-5                                  # 8 chunks: too few to split
    assert int(sk.tickets.abs().sum()) == 0
    rng = np.random.default_rng(5)
    n, h, w, c = 128, 4, 4, 512
    xp = torch.zeros((n, h + 2, w + 2, c))
    xp[:, 1:-1, 1:-1] = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32))
    wt = (rng.standard_normal((c, c, 3, 3)) / 60).astype(np.float32)
    ug = torch.from_numpy(wino.transform_weights(wt)).to(dev)
    xd = xp.to(dev)
    first = wino.conv3x3(xd, ug, c, c, split_k=sk).cpu()
    for _ in range(20):
        assert torch.equal(wino.conv3x3(xd, ug, c, c, split_k=sk).cpu(), first)
    plain = wino.conv3x3(xd, ug, c, c).cpu()
    assert float((plain - first).abs().max()) <= 2e-5 * float(plain.abs().max())   # (a different summation order, not a different sum)
    # a scratch too small for any split, or with too few tickets, runs the unsplit launch
    tiny = wino.SplitKScratch(dev, slab_floats=1024, n_tickets=4)
    assert torch.equal(wino.conv3x3(xd, ug, c, c, split_k=tiny).cpu(), plain)


@pytest.mark.gpu
def test_wino_split_k_results_move_by_rounding_only_across_batch_sizes():
    """the launcher picks ksplit from the number of tiles, i.e. from the batch -- ResNet-18's layers 3 / 4 sum
    their input channels in a different grouping at 128 crops (split) than at 6 (the tiles alone fill less of the chip: another
    split, or none). An image's values are then NOT bitwise batch-independent; they move by fp32 rounding, bounded here at 2e-5 of
    the layer's largest output. (Without split-K scratch the kernel never splits and batch independence is exact:
    test_wino_conv_is_bitwise_repeatable_and_batch_independent.)"""
    from playaid_core_amd import wino

    dev = torch.device("cuda:0")
    sk = wino.SplitKScratch(dev)
    rng = np.random.default_rng(31)
    for (h, w, c) in ((4, 4, 512), (8, 8, 256)):
        n = 128
        xp = torch.zeros((n, h + 2, w + 2, c))
        xp[:, 1:-1, 1:-1] = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32))
        wt = (rng.standard_normal((c, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32)
        ug = torch.from_numpy(wino.transform_weights(wt)).to(dev)
        xd = xp.to(dev)
        full = wino.conv3x3(xd, ug, c, c, split_k=sk).cpu()
        scale = float(full.abs().max())
        for lo, cnt in ((0, 6), (40, 1), (64, 64)):
            part = wino.conv3x3(xd[lo:lo + cnt].contiguous(), ug, c, c, split_k=sk).cpu()
            assert float((part - full[lo:lo + cnt]).abs().max()) <= 2e-5 * scale, (h, w, c, lo, cnt)
    assert int(sk.tickets.abs().sum()) == 0
