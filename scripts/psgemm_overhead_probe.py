"""Fixed cost against per-tile cost of the emulated-fp32 kernel: one layer shape at batch sizes that give every workgroup exactly
1, 2, 3, 4, 6 tiles (256 workgroups; stride-2 3x3 24x40 256->512 and 1x1 12x20 512->512: 4 channel columns, so n images give n x px / 128 x 4 tiles)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from playaid_core_amd import conv
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for name, h, w, cin, cout, k, stride in (("3x3/2 24x40 256->512", 24, 40, 256, 512, 3, 2), ("1x1 24x40 256->256", 24, 40, 256, 256, 1, 1), ("3x3 24x40 128->128", 24, 40, 128, 128, 3, 1)):
    oh, ow = h // stride, w // stride
    bn = 128
    cols = cout // bn
    rng = np.random.default_rng(1)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(k * k * cin)).astype(np.float32)
    wp = torch.from_numpy(conv.pack_weights(wt)).to(dev)
    bias = torch.zeros(cout, device=dev)
    print(name)
    for rounds in (1, 2, 3, 4, 6, 8):
        tiles_m = 256 * rounds // cols
        n = max(1, tiles_m * 128 // (oh * ow))
        x = torch.randn((n, h + 2, w + 2, cin), device=dev)
        out = torch.zeros((n, oh + 2, ow + 2, cout), device=dev)
        f = lambda: conv.conv2d(x, wp, cin, cout, k, stride, in_pad=1, bias=bias, out=out, out_pad=1, act=2)
        f(); torch.cuda.synchronize()
        t = np.median([timed(f) for _ in range(5)])
        tiles = ((n * oh * ow + 127) // 128) * cols
        ksteps = k * k * cin // 32
        print(f"  n {n:4d}  tiles {tiles:5d} ({tiles / 256:.2f} per workgroup)  {t:7.1f} us   {t / (tiles / 256):7.1f} us per round   matrix time per round at 2.0 GHz: {ksteps * 1536 / 2000:.1f} us")
