"""Side-by-side of two scripts/detect_layer_times.py outputs: python scripts/cmp_layers.py before.txt after.txt"""
import re, sys
def rd(p):
    out = []
    for l in open(p):
        m = re.match(r"\s*(\d+) (\w+)\s+k(\d) s(\d)\s+(\d+)x\s*(\d+) cin\s+(\d+) cout\s+(\d+) M\s+(\d+)\s+([\d.]+) GF\s+([\d.]+) us", l)
        if m:
            out.append((m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)), int(m.group(8)), float(m.group(10)), float(m.group(11))))
    return out
a, b = rd(sys.argv[1]), rd(sys.argv[2])
ta = tb = 0
for i, (x, y) in enumerate(zip(a, b)):
    ta += x[8]; tb += y[8]
    print(f"{i:3d} {x[0]:6s} k{x[1]} s{x[2]} {x[3]:3d}x{x[4]:3d} {x[5]:4d}->{x[6]:4d}  {x[8]:7.1f} -> {y[8]:7.1f} us  {y[7] / y[8] * 1e3 if y[8] else 0:6.1f} TF  {'' if abs(y[8]-x[8]) < 0.05*x[8] else ('+' if y[8] > x[8] else '-')}")
print(f"total {ta:.0f} -> {tb:.0f} us")
