#!/usr/bin/env python3
"""MFMA-pipe busy fraction, LDS bank conflicts and clock of the conv3x3 launches from one
rocprofv3 SQ pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE, with --kernel-trace only).

  python scripts/pmc_sq.py <dir>/pmc_sq <out.json> [f32|bf16]

SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMD pipes and GRBM_GUI_ACTIVE over the 8
XCDs: busy = counter / (4 pipes x 256 CUs x GRBM_GUI_ACTIVE / 8); clock = (GRBM_GUI_ACTIVE / 8) / duration."""
import collections, csv, glob, json, sys

path, outp = sys.argv[1], sys.argv[2]
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
rows = list(csv.DictReader(open(glob.glob(path + "/*/*counter_collection.csv")[0])))
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ds = [disp[k] for k in sorted(disp)]
if dtype == "bf16":
    conv = [d for d in ds if "igemm_bf16_kernel" in d["name"] or "conv3x3_bf16_patch_kernel" in d["name"]]
else:
    ig = [d for d in ds if "igemm_f32_kernel" in d["name"] and ", true," not in d["name"]]
    assert len(ig) % 5 == 0, len(ig)   # per step: layer 4's opener + three 1x1/2 branch GEMMs, then the fc (layers 2 / 3's openers are pgemm_kernel launches)
    conv = [d for d in ds if "conv3x3_patch_kernel" in d["name"] or "wino3x3_kernel" in d["name"] or "pgemm_kernel" in d["name"]] + [d for s in range(len(ig) // 5) for d in ig[s * 5:s * 5 + 4]]
n = len(conv)
mean = lambda k: sum(d.get(k, 0.0) for d in conv) / n
gui = mean("GRBM_GUI_ACTIVE") / 8.0  # the counter comes back summed over the 8 XCDs
dur = sum(d["dur"] for d in conv) / n
out = {
    "kernel": "igemm_conv3x3",
    "dtype": dtype,
    "launches_sampled": n,
    "mfma_busy_fraction": round(mean("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * 256 * gui), 4) if gui else None,
    "lds_bank_conflict_cycles": mean("SQ_LDS_BANK_CONFLICT"),
    "approx_clock_GHz_from_GRBM_GUI_ACTIVE": round(gui / dur, 3) if dur else None,
    "avg_duration_us": round(dur / 1e3, 2),
}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out))
