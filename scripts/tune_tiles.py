"""Sweep (tile, split-K) for every conv layer: one process per candidate."""
import os, subprocess, sys, collections
best = collections.defaultdict(lambda: (1e9, None))
for tile in (0, 1, 2, 3, 4):
    for sk in (1, 2, 4):
        env = dict(os.environ, PA_FORCE_TILE=str(tile), PA_FORCE_SPLITK=str(sk))
        out = subprocess.run([sys.executable, "scripts/layer_times.py"], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            f = line.split()
            if len(f) >= 3 and (f[0].startswith("layer") or f[0].startswith("igemm_conv7")):
                us = float(f[1])
                if us < best[f[0]][0]:
                    best[f[0]] = (us, (tile, sk))
        print("tile", tile, "splitk", sk, [l for l in out.splitlines() if l.startswith("total")], flush=True)
tot = 0
for k, (us, cfg) in best.items():
    print(f"{k:24s} {us:8.1f} us  tile={cfg[0]} splitk={cfg[1]}")
    tot += us
print("sum of best", tot)
