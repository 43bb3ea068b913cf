#!/usr/bin/env python3
"""Average duration per kernel FAMILY from a rocprofv3 --kernel-trace CSV of bench.py.

rocprofv3 --stats groups by kernel symbol. The conv3x3 family of bench.py's `roofline` is served
by three symbols (round 5): wino3x3_kernel<...> (the thirteen stride-1 3x3 convs, Winograd; conv3x3_patch_kernel<...> under
PA_WINO=0 / PA_WINO_MIN_HW=8), pgemm_kernel<...> (the stride-2 openers of layers 2 and 3) and igemm_f32_kernel<...> (layer 4's
opener, which splits K, and the three 1x1/2 branch GEMMs of layers 2-4, whose 3x3 runs as Winograd), and igemm_f32_kernel also runs
the fc. The non-gather igemm launches of a bench step come in a fixed order (layer2.0 branch, layer3.0 branch, layer4.0.conv1,
layer4.0 branch, fc; the gather-mode Conv1d is a different instantiation), which this script uses to split them, so that bench.py's roofline.avg_launch_ms can be checked against the
profiler: 19 launches per step. Earlier sets: `f32-igemm-s2` (the openers on igemm_f32_kernel: seven non-gather igemm launches per
step), `f32-wino8` (PA_WINO_MIN_HW=8, mid round 5: 18 launches, six), `f32-direct` (PA_WINO=0: 16, four).

  python scripts/rocprof_families.py <dir>/<host>/<pid>_kernel_trace.csv out.json [f32|bf16]

bf16 (configs[2]): the conv3x3 family = every conv3x3_bf16_patch_kernel launch (the stride-1 convs) + every
igemm_bf16_kernel launch (the stride-2 convs and the three 1x1/2 downsample GEMMs that feed the patch kernel as
its residual): 16 launches per step (the stride-2 openers carry the 1x1/2 branch; 19 with PA_BF16_DS_FUSE=0); igemm_f32_kernel then only runs the fc."""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fam = collections.defaultdict(list)
ig = []
for r in rows:
    name = r["Kernel_Name"]
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if "igemm_f32_kernel" in name:
        if ", true," in name:
            fam["igemm_conv1d_head"].append(dur)
        else:
            ig.append(dur)
    elif name.startswith("pa::") or " pa::" in name:
        fam[name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("pa::", "")].append(dur)
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
per = {"f32": 5, "f32-igemm-s2": 7, "f32-wino8": 6, "f32-direct": 4}.get(dtype, 1)
assert len(ig) % per == 0, len(ig)
for s in range(len(ig) // per):
    step = ig[s * per:(s + 1) * per]
    fam["igemm_conv3x3"] += step[0:per - 1]
    fam["igemm_fc"].append(step[per - 1])
for k in [k for k in fam if k.startswith(("conv3x3_patch_kernel", "conv3x3_bf16_patch_kernel", "igemm_bf16_kernel", "pgemm_kernel")) or "wino3x3_kernel" in k]:
    fam["igemm_conv3x3"] += fam.pop(k)
out = {k: {"launches": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 2), "total_ms": round(sum(v) / 1e6, 3)} for k, v in sorted(fam.items())}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(f"{k:28s} {v['launches']:5d} launches  avg {v['avg_us']:9.2f} us")
