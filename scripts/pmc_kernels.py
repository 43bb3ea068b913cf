#!/usr/bin/env python3
"""Per-kernel means of every counter in one rocprofv3 --pmc pass (kernel names shortened).
  python scripts/pmc_kernels.py <dir> [substring filter]"""
import collections, csv, glob, sys
path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(glob.glob(path + "/*/*counter_collection.csv")[0])))
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                               "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size"), "lds": r.get("LDS_Block_Size"), "vgpr": r.get("VGPR_Count")})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
groups = collections.OrderedDict()
for d in disp.values():
    if filt and filt not in d["name"]:
        continue
    key = (d["name"][:70], d["grid"], d["wg"])
    groups.setdefault(key, []).append(d)
for key, ds in groups.items():
    n = len(ds)
    ctrs = sorted({k for d in ds for k in d if k not in ("name", "dur", "grid", "wg", "lds", "vgpr")})
    print(f"{key[0]} grid={key[1]} wg={key[2]} lds={ds[0]['lds']} vgpr={ds[0]['vgpr']} n={n} dur_us={sum(d['dur'] for d in ds) / n / 1e3:.1f}")
    for c in ctrs:
        print(f"    {c:32s} {sum(d.get(c, 0.0) for d in ds) / n:16.1f}")
