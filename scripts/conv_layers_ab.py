"""Per-layer A/B of the convolution kernels on the detection network's and the ResNet-18's layer shapes (64 x 1080p frames / 128 crops):
exact fp32 (pigemm.hip for 1x1 and stride-2 3x3, wino.hip for stride-1 3x3) against emulated fp32 (psgemm.hip). Interleaved rounds in
one process, median of the rounds; error of both against a float64 convolution on a sample of the outputs.
    python scripts/conv_layers_ab.py [--rounds 7] > gpurun_out/r06_pgemm_split_layers.txt"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from playaid_core_amd import conv, wino  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--only", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
N_DET, N_RES = 64, 128
# (name, n, h, w, cin, cout, k, stride, act, residual, count in the network)
LAYERS = [
    ("det 1x1  96x160  64-> 64", N_DET, 96, 160, 64, 64, 1, 1, 2, False, 2),
    ("det 1x1  96x160  32-> 32", N_DET, 96, 160, 32, 32, 1, 1, 2, False, 1),
    ("det 1x1  48x 80 128->128", N_DET, 48, 80, 128, 128, 1, 1, 2, False, 3),
    ("det 1x1  48x 80  64-> 64", N_DET, 48, 80, 64, 64, 1, 1, 2, False, 3),
    ("det 1x1  48x 80 256->128", N_DET, 48, 80, 256, 128, 1, 1, 2, False, 1),
    ("det 1x1  48x 80 128-> 64", N_DET, 48, 80, 128, 64, 1, 1, 0, False, 1),
    ("det 1x1  24x 40 256->256", N_DET, 24, 40, 256, 256, 1, 1, 2, False, 5),
    ("det 1x1  24x 40 128->128", N_DET, 24, 40, 128, 128, 1, 1, 2, False, 5),
    ("det 1x1  24x 40 512->256", N_DET, 24, 40, 512, 256, 1, 1, 2, False, 1),
    ("det 1x1  24x 40 256->128", N_DET, 24, 40, 256, 128, 1, 1, 2, False, 1),
    ("det 1x1  24x 40 256-> 64", N_DET, 24, 40, 256, 64, 1, 1, 0, False, 1),
    ("det 1x1  12x 20 512->512", N_DET, 12, 20, 512, 512, 1, 1, 2, False, 4),
    ("det 1x1  12x 20 256->256", N_DET, 12, 20, 256, 256, 1, 1, 2, False, 2),
    ("det 1x1  12x 20 512->256", N_DET, 12, 20, 512, 256, 1, 1, 2, False, 2),
    ("det 1x1  12x 20 1024->512", N_DET, 12, 20, 1024, 512, 1, 1, 2, False, 1),
    ("det 1x1  12x 20 512-> 64", N_DET, 12, 20, 512, 64, 1, 1, 0, False, 1),
    ("det 3x3/2 192x320  32-> 64", N_DET, 192, 320, 32, 64, 3, 2, 2, False, 1),
    ("det 3x3/2  96x160  64->128", N_DET, 96, 160, 64, 128, 3, 2, 2, False, 1),
    ("det 3x3/2  48x 80 128->256", N_DET, 48, 80, 128, 256, 3, 2, 2, False, 1),
    ("det 3x3/2  24x 40 256->512", N_DET, 24, 40, 256, 512, 3, 2, 2, False, 1),
    ("det 3x3/2  48x 80 128->128", N_DET, 48, 80, 128, 128, 3, 2, 2, False, 1),
    ("det 3x3/2  24x 40 256->256", N_DET, 24, 40, 256, 256, 3, 2, 2, False, 1),
    ("det 3x3    96x160  32-> 32 +res", N_DET, 96, 160, 32, 32, 3, 1, 2, True, 1),
    ("det 3x3    48x 80  64-> 64 +res", N_DET, 48, 80, 64, 64, 3, 1, 2, True, 3),
    ("det 3x3    24x 40 128->128 +res", N_DET, 24, 40, 128, 128, 3, 1, 2, True, 5),
    ("det 3x3    12x 20 256->256 +res", N_DET, 12, 20, 256, 256, 3, 1, 2, True, 2),
    ("res 3x3    32x 32  64-> 64 +res", N_RES, 32, 32, 64, 64, 3, 1, 1, True, 4),
    ("res 3x3    16x 16 128->128 +res", N_RES, 16, 16, 128, 128, 3, 1, 1, True, 3),
    ("res 3x3     8x  8 256->256 +res", N_RES, 8, 8, 256, 256, 3, 1, 1, True, 3),
    ("res 3x3     4x  4 512->512 +res", N_RES, 4, 4, 512, 512, 3, 1, 1, True, 3),
    ("res 3x3/2  32x 32  64->128", N_RES, 32, 32, 64, 128, 3, 2, 1, False, 1),
    ("res 3x3/2  16x 16 128->256", N_RES, 16, 16, 128, 256, 3, 2, 1, False, 1),
    ("res 3x3/2   8x  8 256->512", N_RES, 8, 8, 256, 512, 3, 2, 1, False, 1),
]


def timed(fn, reps=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


print(f"{'layer':34s} {'GFLOP':>7s} | {'exact us':>9s} {'TF':>6s} | {'emul us':>8s} {'TF-eq':>6s} | speed-up | err vs f64: exact   emulated | memory floor us (5 TB/s)")
tot_e = tot_s = 0.0
sk = wino.SplitKScratch(dev)
for name, n, h, w, cin, cout, k, stride, act, res, count in LAYERS:
    if args.only and args.only not in name:
        continue
    rng = np.random.default_rng(1)
    pad = 1
    oh, ow = h // stride, w // stride
    x = torch.zeros((n, h + 2, w + 2, cin), dtype=torch.float32, device=dev)
    x[:, 1:-1, 1:-1] = torch.randn((n, h, w, cin), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(k * k * cin)).astype(np.float32)
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    resid = torch.zeros((n, oh + 2, ow + 2, cout), dtype=torch.float32, device=dev) if res else None
    if res:
        resid[:, 1:-1, 1:-1] = torch.randn((n, oh, ow, cout), device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    res_after = act == 2
    w_emu = torch.from_numpy(conv.pack_weights(wt, "emulated_f32", has_residual=res)).to(dev)
    out_e = torch.zeros((n, oh + 2, ow + 2, cout), dtype=torch.float32, device=dev)
    out_s = torch.zeros_like(out_e)
    if k == 3 and stride == 1:
        ug = torch.from_numpy(wino.transform_weights(wt)).to(dev)
        exact = lambda: wino.conv3x3(x, ug, cin, cout, bias=bias, residual=resid, out=out_e, out_pad=1, act=act, res_after=res_after, split_k=sk)
    else:
        w_f32 = torch.from_numpy(conv.pack_weights(wt, "f32")).to(dev)
        exact = lambda: conv.conv2d(x, w_f32, cin, cout, k, stride, in_pad=1, bias=bias, out=out_e, out_pad=1, act=act, compute_dtype="f32")
    emu = lambda: conv.conv2d(x, w_emu, cin, cout, k, stride, in_pad=1, bias=bias, residual=resid, out=out_s, out_pad=1, act=act, res_after=res_after,
                              compute_dtype="emulated_f32")
    exact(); emu()
    torch.cuda.synchronize()
    te, ts = [], []
    for _ in range(args.rounds):
        te.append(timed(exact))
        ts.append(timed(emu))
    te, ts = float(np.median(te)), float(np.median(ts))
    # error on the first image against float64 (CPU)
    xi = x[:1, 1:-1, 1:-1].permute(0, 3, 1, 2).double().cpu()
    ref = F.conv2d(xi, torch.from_numpy(wt).double(), bias.double().cpu(), stride=stride, padding=(k - 1) // 2)
    r0 = resid[:1, 1:-1, 1:-1].permute(0, 3, 1, 2).double().cpu() if res else 0
    if res and not res_after:
        ref = ref + r0
    ref = F.relu(ref) if act == 1 else (F.silu(ref) if act == 2 else ref)
    if res and res_after:
        ref = ref + r0
    ee = float((out_e[:1, 1:-1, 1:-1].permute(0, 3, 1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
    es = float((out_s[:1, 1:-1, 1:-1].permute(0, 3, 1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
    gf = 2.0 * n * oh * ow * cout * k * k * cin / 1e9
    floor = (n * oh * ow * (cin * (1 if k == 1 else (1.0 if stride == 2 else 1.0)) + cout * (2 if res else 1)) * 4 + wt.size * 4) / 5e12 * 1e6
    print(f"{name:34s} {gf:7.2f} | {te:9.1f} {gf / te * 1e3:6.1f} | {ts:8.1f} {gf / ts * 1e3:6.1f} | {te / ts:7.2f}x | {ee:18.2e} {es:10.2e} | {floor:6.1f}   x{count}", flush=True)
    tot_e += te * count
    tot_s += ts * count
    del x, out_e, out_s, resid
print(f"sum over the networks' layers (x count): exact {tot_e:.0f} us, emulated {tot_s:.0f} us")
