#!/usr/bin/env python3
"""HBM traffic of the conv3x3 kernel family from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes).

  python scripts/pmc_traffic.py <dir with pmc_fetch/ and pmc_write/> <out.json> [f32|bf16] [frames height width]

Units: both counters are in KiB. On gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced read stream (guide, section HBM), so it is doubled; WRITE_SIZE is taken as is.
fp32: the family = every wino3x3_kernel / conv3x3_patch_kernel launch (thirteen stride-1 convs per step), every pgemm_kernel
launch (the stride-2 openers of layers 2 and 3) plus the first four of each five non-gather igemm_f32_kernel launches (layer 4's opener,
which splits K, and the 1x1/2 branch GEMMs of the Winograd block-0 layers; the fifth is the fc): 19 per step. bf16: every conv3x3_bf16_patch_kernel and igemm_bf16_kernel launch (16 per step; 19 with PA_BF16_DS_FUSE=0)."""
import collections, csv, glob, json, sys


def dispatches(path, counter):
    rows = list(csv.DictReader(open(glob.glob(path + "/*/*counter_collection.csv")[0])))
    out = collections.OrderedDict()
    for r in rows:
        if r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return [out[k] for k in sorted(out)]


def conv3x3_values(disp, dtype):
    if dtype == "bf16":  # 16 launches per step: patch kernels and the stride-2 openers (which carry the 1x1/2 downsample branch)
        return [v for (name, v) in disp if "igemm_bf16_kernel" in name or "conv3x3_bf16_patch_kernel" in name]
    vals = [v for (name, v) in disp if "conv3x3_patch_kernel" in name or "wino3x3_kernel" in name or "pgemm_kernel" in name]
    ig = [v for (name, v) in disp if "igemm_f32_kernel" in name and ", true," not in name]
    per_step = 4 + 1  # two 1x1/2 branch GEMMs, layer 4's opener, its branch GEMM, then the fc
    assert len(ig) % per_step == 0, (len(ig), per_step)
    for s in range(len(ig) // per_step):
        vals += ig[s * per_step:s * per_step + per_step - 1]
    return vals


d = sys.argv[1]
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
f = conv3x3_values(dispatches(d + "/pmc_fetch", "FETCH_SIZE"), dtype)
w = conv3x3_values(dispatches(d + "/pmc_write", "WRITE_SIZE"), dtype)
fetch = 2.0 * sum(f) / len(f) * 1024.0
write = sum(w) / len(w) * 1024.0
# the workload the passes ran (bench.py replays the figure only for this shape): defaults = the two profiled configs
shape = [int(v) for v in sys.argv[4:7]] if len(sys.argv) >= 7 else ([256, 720, 1280] if dtype == "bf16" else [64, 1080, 1920])
out = {
    "kernel": "igemm_conv3x3",
    "dtype": dtype,
    "workload": {"frames": shape[0], "height": shape[1], "width": shape[2]},
    "launches_sampled": len(f),
    "fetch_bytes_per_launch": round(fetch),
    "write_bytes_per_launch": round(write),
    "traffic_bytes_per_launch": round(fetch + write),
    "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes, mean over the conv3x3-family launches of each step (19 in fp32, 16 in bf16)",
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
