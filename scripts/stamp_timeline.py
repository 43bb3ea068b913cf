"""Read the per-workgroup timeline stamps of one patchconv launch (build/libpa_stamp.so,
-DPA_STAMP_BUILD; PA_STAMP_FILE / PA_STAMP_CALL) and print where the time goes."""
import struct, sys, collections
import numpy as np
raw = open(sys.argv[1], "rb").read()
grid, bm, M, chunk = struct.unpack("4i", raw[:16])
a = np.frombuffer(raw[16:], dtype=np.uint64).reshape(grid, 6).astype(np.int64)
t0 = a[:, 0].min()
start, loop0, loop1, end = [(a[:, i] - t0) * 0.01 for i in range(4)]  # 100 MHz -> us
hw, xcc = a[:, 4], a[:, 5] & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
print(f"grid {grid} bm {bm} M {M} Cin {chunk}: kernel span {end.max():.1f} us")
print(f"  start     min/med/max {start.min():.1f} {np.median(start):.1f} {start.max():.1f}")
print(f"  prologue  med {np.median(loop0 - start):.2f} us  (max {np.max(loop0 - start):.2f})")
print(f"  main loop med {np.median(loop1 - loop0):.2f} us  (min {np.min(loop1 - loop0):.2f} max {np.max(loop1 - loop0):.2f})")
print(f"  epilogue  med {np.median(end - loop1):.2f} us  (max {np.max(end - loop1):.2f})")
print(f"  lifetime  med {np.median(end - start):.2f} us")
key = xcc * 1000 + se * 100 + sh * 50 + cu
per = collections.Counter(key.tolist())
print(f"  distinct (xcc,se,sh,cu) {len(per)}; workgroups per CU min/max {min(per.values())}/{max(per.values())}")
# concurrency: how many WGs alive at sampled times
ts = np.linspace(0, end.max(), 41)
alive = [(int(((start <= t) & (end > t)).sum()), int(((loop0 <= t) & (loop1 > t)).sum())) for t in ts]
print("  t(us): alive / in-loop")
for t, (al, il) in zip(ts, alive):
    print(f"   {t:6.1f}: {al:4d} {il:4d}")
