"""Per-workgroup timeline of one crop_fused_kernel launch (build with -DPA_STAMP_BUILD,
PA_CROP_STAMP_FILE=...): lifetimes, rounds, tail."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
st, en, d = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01, a[:, 2]
life = en - st
print(f"{len(a)} workgroups, span {en.max():.1f} us; lifetime med {np.median(life):.1f} min {life.min():.1f} max {life.max():.1f}")
print(f"start med {np.median(st):.1f} max {st.max():.1f}; d range {d.min()}..{d.max()}")
for lo in range(0, int(en.max()) + 1, 10):
    alive = int(((st <= lo) & (en > lo)).sum())
    print(f"  t={lo:4d} us alive {alive}")
c = np.corrcoef(d, life)[0, 1]
print("corr(d, lifetime) =", round(float(c), 3))
