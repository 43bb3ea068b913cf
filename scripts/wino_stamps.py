"""Timeline of one Winograd launch from its per-wave s_memtime stamps (PA_WINO_ABL=8, PA_WINO_STAMP_FILE): where a workgroup's
life goes -- prologue, per chunk: barrier wait, raw reads + input transform, matrix phase -- and the epilogue.
  PA_WINO_ABL=8 PA_WINO_STAMP_FILE=/tmp/st.bin PA_WINO_STAMP_CALL=6 python scripts/wino_stamps.py <shape index>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from playaid_core_amd import wino

SHAPES = [(128, 32, 32, 64, 64), (128, 16, 16, 128, 128), (64, 48, 80, 64, 64), (64, 24, 40, 128, 128)]
n, h, w, cin, cout = SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
xp = torch.zeros((n, h + 2, w + 2, cin), device=dev)
if not os.environ.get("PA_WINO_ZERO_DATA"):   # (all-zero operands: the clock the matrix pipe runs at then shows what random data costs in power)
    xp[:, 1:-1, 1:-1] = torch.randn((n, h, w, cin), device=dev)
wts = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
if os.environ.get("PA_WINO_ZERO_DATA"):
    wts[:] = 0
ug = torch.from_numpy(wino.transform_weights(wts)).to(dev)
out = torch.zeros((n, h + 2, w + 2, cout), device=dev)
for _ in range(10):
    wino.conv3x3(xp, ug, cin, cout, out=out, act=1)
torch.cuda.synchronize()
raw = open(os.environ["PA_WINO_STAMP_FILE"], "rb").read()
grid, nw, cin_, n_sb = np.frombuffer(raw[:16], dtype=np.int32)
st = np.frombuffer(raw[16:], dtype=np.uint64).reshape(grid, nw, 64).astype(np.int64)
nch = cin_ // 8
t0 = st[..., 0].min()
print(f"shape n={n} {h}x{w} cin {cin} cout {cout}: grid {grid}, {nw} waves, {nch} chunks; values below are s_memtime ticks / 100 (shader clock: 100 ticks = ~42 ns at 2.4 GHz)")
life = st[..., 63] - st[..., 0]
print(f"launch span {(st[..., 63].max() - t0) / 100:.1f} us; wave lifetime median {np.median(life) / 100:.2f} us, max {life.max() / 100:.2f} us")
start = (st[..., 0] - t0) / 100
print(f"wave start: p10 {np.percentile(start, 10):.1f} median {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f} us")
pro = (st[..., 1] - st[..., 0]) / 100
print(f"prologue (entry -> first barrier passed): median {np.median(pro):.2f} us, p90 {np.percentile(pro, 90):.2f}")
bw, tr, mm = [], [], []
for c in range(nch):
    b, t, m = st[..., 1 + 3 * c], st[..., 2 + 3 * c], st[..., 3 + 3 * c]
    tr.append(np.median(t - b) / 100)
    mm.append(np.median(m - t) / 100)
    if c + 1 < nch:
        bw.append(np.median(st[..., 4 + 3 * c] - m) / 100)
print("per chunk, median us: transform phase (barrier -> V ready)", " ".join(f"{x:.2f}" for x in tr))
print("                      matrix phase (V ready -> last mfma issued)", " ".join(f"{x:.2f}" for x in mm))
print("                      wait (last mfma issued -> next barrier passed)", " ".join(f"{x:.2f}" for x in bw))
ep = (st[..., 63] - st[..., 3 * nch]) / 100
print(f"epilogue (last mfma issued -> stores issued): median {np.median(ep):.2f} us, p90 {np.percentile(ep, 90):.2f}")
print(f"sum of medians: prologue {np.median(pro):.2f} + transform {sum(tr):.2f} + matrix {sum(mm):.2f} + wait {sum(bw):.2f} + epilogue {np.median(ep):.2f} us")
