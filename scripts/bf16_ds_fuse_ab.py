"""A/B of the bf16 downsample branch: PA_BF16_DS_FUSE=2 (its own 1x1/2 GEMM, the opener launched with the fused
path's tile and K split) against =1 (on the centre tap of the block's
stride-2 opener, igemm_bf16.hip DS), each in its own process (the knob is read once). The two accumulate the same products
in the same order, so the log-probs must be BIT-IDENTICAL; sizes cover partial tiles and both tile shapes."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine
n, mb, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
sd = synth.make_state_dict(seed=1234)
e = Engine(sd, max_batch_frames=mb, max_clip_frames=max(n, 64), max_frame_height=720, max_frame_width=1280, compute_dtype="bf16")
r = e.infer_clip(synth.make_frames(n, 720, 1280), synth.make_boxes(n, 720, 1280))
np.save(out, r["logp"])
e.close()
''' % ROOT
tmp = tempfile.mkdtemp()
ok = True
for n, mb in ((21, 21), (40, 32), (70, 64), (256, 256)):
    res = {}
    for tag in ("2", "1"):
        out = os.path.join(tmp, f"{tag}_{n}.npy")
        subprocess.run([sys.executable, "-c", CHILD, str(n), str(mb), out], check=True, env=dict(os.environ, PA_BF16_DS_FUSE=tag), timeout=600)
        res[tag] = np.load(out)
    same = np.array_equal(res["2"], res["1"])
    print(f"n={n} mb={mb}: identical={same}  max|d|={np.abs(res['2'] - res['1']).max():.3e}  finite={np.isfinite(res['1']).all()}", flush=True)
    ok &= same and bool(np.isfinite(res["1"]).all())
print("AB_OK" if ok else "AB_FAIL")
sys.exit(0 if ok else 1)
