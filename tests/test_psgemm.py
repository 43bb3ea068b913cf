"""The "emulated fp32" persistent convolution kernel (``csrc/psgemm.hip``: six bf16 matrix instructions per fp32 product, three
bf16 slices per operand) through the C ABI against torch's CPU ``conv2d`` in float64 -- the operator YOLOv5's ``Conv`` /
``torchvision.resnet18``'s 3x3 layers apply (``ai_runner.py:191-224``; ``cnn_action_detector.py:16,32``). Bars: 2e-5 of the
layer's largest output (test_wino.py's bar for the exact kernels), AND no further from the float64 result than 1.5x the exact
fp32 kernel (``csrc/pigemm.hip``) on the same inputs where that kernel takes the layer."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _bf16(x):
    """float32 array -> (uint16 bf16 bits rounded to nearest even, its float32 value)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    h = u.astype(np.uint16)
    return h, (h.astype(np.uint32) << 16).view(np.float32)


def test_weight_slices_are_an_exact_decomposition_in_the_stage_image_order():
    """Host side only: the three bf16 planes sum back to the fp32 weight exactly (24 significand bits = 3 x 8) and every value sits
    where the kernel's LDS stage image expects it: [tile_n][k-step][plane][row][chunk ^ ((row >> 2) & 3)][8]."""
    from playaid_core_amd import conv

    rng = np.random.default_rng(7)
    for cin, cout, k, res in ((32, 32, 1, False), (64, 64, 1, False), (64, 128, 3, False), (128, 128, 3, True), (256, 96, 1, False)):
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(k * k * cin)).astype(np.float32)
        img = conv.pack_weights(w, "emulated_f32", has_residual=res).view(np.uint16)
        bn = 128 if cout % 128 == 0 else (64 if cout % 64 == 0 else 32)   # (with a residual too: PA_PS_RES128 defaults to 1)
        pieces = 8 if bn == 32 else bn * 3 // 16
        ktot, nk = k * k * cin, k * k * cin // 32
        assert img.size == (cout // bn) * nk * pieces * 512
        img = img.reshape(cout // bn, nk, pieces * 512)
        wk = np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(cout, ktot)   # K order: (ky, kx, cin)
        planes = []
        rem = wk.copy()
        for _ in range(3):
            h, f = _bf16(rem)
            planes.append(h)
            rem = rem - f
        assert np.all(rem == 0), "three bf16 slices must reproduce an fp32 value exactly"
        for co in rng.integers(0, cout, 10):
            for kk in rng.integers(0, ktot, 10):
                tn, r, ks, c, j = co // bn, co % bn, kk // 32, (kk % 32) // 8, kk % 8
                for s in range(3):
                    got = img[tn, ks, ((s * bn + r) * 4 + (c ^ ((r >> 2) & 3))) * 8 + j]
                    assert got == planes[s][co, kk], (cin, cout, k, co, kk, s)


def _case(n, h, w, cin, cout, k, stride, seed, act=0, residual=False, res_after=False, in_extra=0, out_extra=0, out_pad=0, in_pad=None,
          dtype="emulated_f32", in_place=False, scale=1.0):
    from playaid_core_amd import conv

    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n, cin, h, w)) * scale).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    oh, ow = h // stride, w // stride
    res = rng.standard_normal((n, cout, oh, ow)).astype(np.float32) if residual else None
    dev = torch.device("cuda:0")
    ip = (k - 1) // 2 if in_pad is None else in_pad
    xp = torch.zeros((n, h + 2 * ip, w + 2 * ip, cin + in_extra), dtype=torch.float32)
    xp[:, ip:ip + h, ip:ip + w, :cin] = torch.from_numpy(x).permute(0, 2, 3, 1)
    if in_extra:
        xp[:, :, :, cin:] = 7.0   # channels of a wider buffer the kernel must not read
    out = torch.full((n, oh + 2 * out_pad, ow + 2 * out_pad, cout + out_extra), -3.0, dtype=torch.float32)
    resp = None
    if residual:
        resp = torch.zeros_like(out) if not in_place else out
        resp[:, out_pad:out_pad + oh, out_pad:out_pad + ow, :cout] = torch.from_numpy(res).permute(0, 2, 3, 1)
    wp = torch.from_numpy(conv.pack_weights(wt, dtype, has_residual=residual)).to(dev)
    out_d = out.to(dev)
    res_d = None if not residual else (out_d if in_place else resp.to(dev))
    got = conv.conv2d(xp.to(dev), wp, cin, cout, k, stride, in_pad=ip, bias=torch.from_numpy(b).to(dev), residual=res_d, out=out_d, out_pad=out_pad,
                      act=act, res_after=res_after, compute_dtype=dtype)
    torch.cuda.synchronize()
    got = got.cpu()
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), stride=stride, padding=(k - 1) // 2)
    if residual and not res_after:
        ref = ref + torch.from_numpy(res).double()
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.silu(ref)
    if residual and res_after:
        ref = ref + torch.from_numpy(res).double()
    inner = got[:, out_pad:out_pad + oh, out_pad:out_pad + ow, :cout].permute(0, 3, 1, 2).double()
    err = float((inner - ref).abs().max() / ref.abs().max())
    mask = torch.ones_like(got, dtype=torch.bool)
    mask[:, out_pad:out_pad + oh, out_pad:out_pad + ow, :cout] = False
    assert bool((got[mask] == -3.0).all()), "the kernel wrote outside its output"
    return err


SHAPES = [
    # n, h, w, cin, cout, k, stride
    (2, 48, 80, 128, 128, 1, 1),     # detector P3 C3.cv3: 128-channel tiles
    (1, 96, 160, 64, 64, 1, 1),      # two k-steps per tile
    (1, 96, 160, 32, 32, 1, 1),      # ONE k-step per tile, 32-channel tiles (6 + 2 padded weight pieces)
    (3, 24, 40, 256, 256, 1, 1),     # two 128-channel columns
    (5, 12, 20, 512, 512, 1, 1),     # 1200 pixels: partial last tile (1200 = 9 x 128 + 48)
    (2, 12, 20, 1024, 512, 1, 1),    # 32 k-steps
    (2, 24, 40, 256, 64, 1, 1),      # the Detect convolutions' shape class
    (1, 96, 160, 64, 128, 3, 2),     # stride-2 3x3
    (2, 24, 40, 256, 512, 3, 2),
    (1, 192, 320, 32, 64, 3, 2),
    (3, 16, 16, 128, 128, 3, 1),     # stride-1 3x3 as an implicit GEMM (ResNet-18 layer 2)
    (7, 4, 4, 512, 512, 3, 1),       # layer 4: 112 pixels, one partial tile
    (1, 8, 12, 96, 96, 1, 1),        # 96 channels: three 32-channel columns, 96 pixels: less than one tile
    (1, 8, 8, 32, 64, 1, 1),         # ONE tile of ONE k-step: fewer stages than the ring holds (the prologue's counted waits)
    (1, 8, 16, 64, 128, 1, 1),       # one tile, two k-steps
    (1, 16, 24, 96, 128, 1, 1),      # three tiles of three k-steps on one workgroup column
    (33, 12, 20, 128, 256, 1, 1),    # 7920 pixels = 61.9 tiles: every XCD's share is ragged
    # workgroups with MORE THAN ONE tile (the ring runs on across tile boundaries; the shapes above give every workgroup one)
    (26, 48, 80, 64, 64, 1, 1),      # 780 tiles of two k-steps, up to four per workgroup
    (10, 48, 80, 128, 128, 1, 1),    # 300 tiles of four k-steps on 256 workgroups: one or two each (ragged)
    (16, 48, 80, 96, 128, 1, 1),     # three k-steps
    (20, 24, 40, 256, 256, 1, 1),    # two 128-channel columns, 150 pixel tiles
    (12, 96, 160, 64, 128, 3, 2),    # stride-2 3x3, 18 k-steps
    (9, 96, 160, 32, 32, 3, 1),      # 32-channel tiles, nine k-steps, 1080 tiles
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_emulated_conv_matches_conv2d_in_float64(shape):
    n, h, w, cin, cout, k, stride = shape
    err = _case(n, h, w, cin, cout, k, stride, seed=h * 131 + cin + k)
    assert err <= 2e-5, (shape, err)


@pytest.mark.gpu
def test_emulated_conv_is_as_close_to_float64_as_the_exact_kernel():
    """The same layers on the exact fp32 kernel (v_mfma_f32_32x32x2_f32, pigemm.hip) and on the emulated one: the emulated result
    may be at most 1.5x as far from the float64 convolution (measured: about equal, often closer -- its products are exact and
    only the fp32 accumulation rounds)."""
    for shape in ((2, 48, 80, 128, 128, 1, 1), (5, 12, 20, 512, 512, 1, 1), (1, 96, 160, 64, 128, 3, 2), (2, 12, 20, 1024, 512, 1, 1)):
        n, h, w, cin, cout, k, stride = shape
        e_emu = _case(n, h, w, cin, cout, k, stride, seed=11, act=2)
        e_f32 = _case(n, h, w, cin, cout, k, stride, seed=11, act=2, dtype="f32")
        print(f"{shape}: emulated {e_emu:.2e}, exact fp32 {e_f32:.2e}")
        assert e_emu <= 1.5 * e_f32 + 1e-7, (shape, e_emu, e_f32)


@pytest.mark.gpu
def test_emulated_conv_epilogues_channel_slices_and_borders():
    # SiLU into a slice of a wider [cv1 | cv2] buffer, reading a slice of a wider buffer (YOLOv5 C3)
    assert _case(2, 24, 40, 64, 64, 1, 1, 1, act=2, in_extra=64, out_extra=64, out_pad=1) <= 2e-5
    # ReLU, input with a wider border than the kernel needs (a 1x1 reading a 3x3 layer's zero-bordered buffer)
    assert _case(3, 16, 16, 128, 64, 1, 1, 2, act=1, in_pad=1, out_pad=1) <= 2e-5
    # ResNet BasicBlock tail: residual BEFORE the ReLU (64-channel tiles with a residual)
    assert _case(4, 16, 16, 128, 128, 3, 1, 3, act=1, residual=True, out_pad=1) <= 2e-5
    # YOLOv5 Bottleneck: SiLU, residual AFTER it, added IN PLACE (the output buffer is the residual)
    assert _case(2, 24, 40, 64, 64, 3, 1, 4, act=2, residual=True, res_after=True, in_place=True, out_extra=64, out_pad=1) <= 2e-5
    assert _case(1, 96, 160, 32, 32, 3, 1, 5, act=2, residual=True, res_after=True, in_place=True, in_extra=32, out_extra=32, out_pad=1) <= 2e-5
    # no activation, stride 2 with a residual of the output's geometry
    assert _case(2, 16, 16, 64, 128, 3, 2, 6, act=0, residual=True, out_pad=1) <= 2e-5
    # the same slices and borders with several tiles per workgroup: SiLU, ReLU, none
    assert _case(12, 48, 80, 64, 64, 1, 1, 7, act=2, in_extra=64, out_extra=64, out_pad=1) <= 2e-5
    assert _case(10, 48, 80, 128, 128, 1, 1, 8, act=1, in_pad=1, out_pad=1, out_extra=32) <= 2e-5
    assert _case(12, 48, 80, 128, 64, 1, 1, 9, act=0, out_pad=2) <= 2e-5


@pytest.mark.gpu
def test_emulated_conv_keeps_small_and_large_magnitudes():
    """The three slices follow the operand's exponent: inputs of 1e-3 and of 1e3 keep the relative bar (nothing is quantised to a
    fixed bf16 grid)."""
    assert _case(2, 24, 40, 128, 128, 1, 1, 21, scale=1e-3) <= 2e-5
    assert _case(2, 24, 40, 128, 128, 1, 1, 22, scale=1e3) <= 2e-5


@pytest.mark.gpu
def test_emulated_conv_is_bitwise_repeatable_and_batch_independent():
    from playaid_core_amd import conv

    rng = np.random.default_rng(9)
    dev = torch.device("cuda:0")
    n, h, w, c = 6, 24, 40, 256
    x = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(dev)
    wt = (rng.standard_normal((c, c, 1, 1)) / 16).astype(np.float32)
    wp = torch.from_numpy(conv.pack_weights(wt)).to(dev)
    a = conv.conv2d(x, wp, c, c, 1).cpu()
    for _ in range(5):
        assert torch.equal(conv.conv2d(x, wp, c, c, 1).cpu(), a)
    one = conv.conv2d(x[2:3].contiguous(), wp, c, c, 1).cpu()
    assert torch.equal(one[0], a[2]), "an image's result must not depend on the batch around it (no split-K, fixed k order)"


@pytest.mark.gpu
def test_conv_operator_rejects_what_the_kernels_do_not_take():
    from playaid_core_amd import conv

    dev = torch.device("cuda:0")
    with pytest.raises(ValueError):
        conv.pack_weights(np.zeros((48, 40, 1, 1), np.float32))          # cin % 32
    with pytest.raises(ValueError):
        conv.pack_weights(np.zeros((64, 64, 5, 5), np.float32))          # 5 x 5
    x = torch.zeros((1, 8, 8, 64), device=dev)
    wp = torch.from_numpy(conv.pack_weights(np.zeros((64, 64, 1, 1), np.float32), "f32")).to(dev)
    with pytest.raises(ValueError):
        conv.conv2d(x, wp, 64, 64, 1, residual=torch.zeros((1, 8, 8, 64), device=dev), compute_dtype="f32")   # the exact kernel has no residual epilogue
    with pytest.raises(ValueError):
        conv.conv2d(x, wp, 64, 64, 3, in_pad=0, compute_dtype="f32")                                            # a 3x3 needs a border
    # 16-byte units: pixel strides in multiples of four floats, 16-byte aligned tensors
    we = torch.from_numpy(conv.pack_weights(np.zeros((64, 64, 1, 1), np.float32))).to(dev)
    with pytest.raises(ValueError):
        conv.conv2d(torch.zeros((1, 8, 8, 66), device=dev), we, 64, 64, 1)
    with pytest.raises(ValueError):
        conv.conv2d(x, we, 64, 64, 1, out=torch.zeros((1, 8, 8, 70), device=dev))
    with pytest.raises(ValueError):
        conv.conv2d(x, we, 64, 64, 1, out=torch.zeros(8 * 8 * 64 + 4, device=dev)[1:1 + 8 * 8 * 64].view(1, 8, 8, 64))
    assert conv.conv2d(x, we, 64, 64, 1).abs().max().item() == 0.0
