#!/usr/bin/env python3
"""Per-LAYER HBM traffic of the conv3x3 family: the dispatches of one bench clip in launch order, FETCH_SIZE (x2: gfx950
wide-read correction) and WRITE_SIZE from two separate rocprofv3 --pmc passes, next to each layer's algorithmic bytes.
  python scripts/pmc_per_layer.py <dir with pmc_fetch/ and pmc_write/> [crops]"""
import collections, csv, glob, sys

d = sys.argv[1]
crops = int(sys.argv[2]) if len(sys.argv) > 2 else 128


def load(path, counter):
    rows = list(csv.DictReader(open(glob.glob(path + "/*/*counter_collection.csv")[0])))
    out = collections.OrderedDict()
    for r in rows:
        if r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0)
    return [out[k] for k in sorted(out)]


f, w = load(d + "/pmc_fetch", "FETCH_SIZE"), load(d + "/pmc_write", "WRITE_SIZE")
conv = lambda n: ("conv3x3_patch_kernel" in n) or ("igemm_f32_kernel" in n and ", true," not in n) or "stem_pool" in n
fi = [(n, v) for n, v in f if conv(n)]
wi = [(n, v) for n, v in w if conv(n)]
per = 1 + 16 + 1  # stem, 16 convs, fc
assert len(fi) % per == 0 and len(fi) == len(wi), (len(fi), len(wi))
names = ["stem+pool"] + [f"layer{l}.{b}.conv{c}" for l in (1, 2, 3, 4) for b in (0, 1) for c in (1, 2)] + ["fc"]
# algorithmic bytes (fp32): in + out (+ residual) + weights (+ second source)
hw = {1: 32, 2: 16, 3: 8, 4: 4}
ch = {1: 64, 2: 128, 3: 256, 4: 512}
alg = [crops * (134 * 134 * 4 * 4) + crops * 32 * 32 * 64 * 4]
for l in (1, 2, 3, 4):
    for b in (0, 1):
        for c in (1, 2):
            cin = ch[l] if not (b == 0 and c == 1 and l > 1) else ch[l - 1]
            hin = hw[l] * (2 if (b == 0 and c == 1 and l > 1) else 1)
            a = crops * (hin * hin * cin + hw[l] * hw[l] * ch[l]) * 4 + ch[l] * cin * 9 * 4
            if c == 2 and (b == 1 or l == 1):
                a += crops * hw[l] * hw[l] * ch[l] * 4  # residual
            if c == 2 and b == 0 and l > 1:
                a += crops * (2 * hw[l]) ** 2 * ch[l - 1] * 4 / 4 + ch[l] * ch[l - 1] * 4  # strided second source + its weights
            alg.append(a)
alg.append(crops * 512 * 4 + 1024 * 512 * 4 + crops * 1024 * 4)
steps = len(fi) // per
print(f"{'layer':18s} {'fetch MB':>9s} {'write MB':>9s} {'total':>8s} {'algorithmic':>12s} {'ratio':>6s}   kernel")
tf = ta = 0
for i, nm in enumerate(names):
    fv = sum(fi[s * per + i][1] for s in range(steps)) / steps * 2.0
    wv = sum(wi[s * per + i][1] for s in range(steps)) / steps
    if nm.startswith("layer"):
        tf += fv + wv
        ta += alg[i]
    print(f"{nm:18s} {fv / 1e6:9.1f} {wv / 1e6:9.1f} {(fv + wv) / 1e6:8.1f} {alg[i] / 1e6:12.1f} {(fv + wv) / alg[i]:6.2f}   {fi[i][0][:60]}")
print(f"conv family per launch: measured {tf / 16e6:.1f} MB, algorithmic {ta / 16e6:.1f} MB, ratio {tf / ta:.2f}")
