"""Per-workgroup timeline of one stem7x7_kernel launch (-DPA_STAMP_BUILD, PA_STEM_STAMP_FILE)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 6).astype(np.int64)
t0 = a[:, 0].min()
st, pro, en = (a[:, 0] - t0) * 0.01, (a[:, 1] - a[:, 0]) * 0.01, (a[:, 2] - t0) * 0.01
ep, nt = a[:, 3] * 0.01, a[:, 4]
print(f"{len(a)} workgroups, span {en.max():.1f} us, tiles per workgroup {nt.min()}..{nt.max()}")
print(f"start max {st.max():.2f}; prologue (weights + first patch) med {np.median(pro):.2f} max {pro.max():.2f}")
life = en - st
print(f"lifetime med {np.median(life):.1f} min {life.min():.1f} max {life.max():.1f}")
print(f"epilogue per tile med {np.median(ep / nt):.2f} us; per-tile total med {np.median((life - pro) / nt):.2f} us")
