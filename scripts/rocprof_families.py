#!/usr/bin/env python3
"""Average duration per kernel FAMILY from a rocprofv3 --kernel-trace CSV of bench.py.

rocprofv3 --stats groups by kernel symbol, and one symbol (igemm_f32_kernel<...>) serves the
sixteen 3x3 convolutions and the fc (the stem has its own kernel, stem7x7_kernel). The launches
of a bench step come in a fixed order (16 x conv3x3, fc, then the gather-mode Conv1d), which this script uses to split
them, so that bench.py's roofline.avg_launch_ms can be checked against the profiler.

  python scripts/rocprof_families.py <dir>/runc/<pid>_kernel_trace.csv out.json"""
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
        fam[name.split("(")[0].replace("void ", "").replace("pa::", "")].append(dur)
per = 17
assert len(ig) % per == 0, len(ig)
for s in range(len(ig) // per):
    step = ig[s * per:(s + 1) * per]
    fam["igemm_conv3x3"] += step[0:16]
    fam["igemm_fc"].append(step[16])
out = {k: {"launches": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 2), "total_ms": round(sum(v) / 1e6, 3)} for k, v in sorted(fam.items())}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(f"{k:28s} {v['launches']:5d} launches  avg {v['avg_us']:9.2f} us")
