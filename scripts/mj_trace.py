#!/usr/bin/env python3
"""Per-kernel timeline of the LAST pa_mjpeg_decode call in a rocprofv3 --kernel-trace CSV (mjpeg kernels + runtime copies)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sel = [r for r in rows if "pa::mj" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
# last call = from the last unstuff_count_kernel on
last = max(i for i, r in enumerate(sel) if "unstuff_count" in r["Kernel_Name"])
sel = sel[last:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print(f"{(a - t0) / 1e3:9.1f} us  {(b - a) / 1e3:8.1f} us  {name[:48]:48s} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
print(f"total {(int(sel[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
