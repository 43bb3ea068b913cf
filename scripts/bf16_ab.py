"""A/B of the two bf16 conv kernels (PA_BF16_PATCH=0: igemm_bf16.hip everywhere; =1: patchconv_bf16.hip for the
stride-1 3x3 convs without a second source), each in its own process (the knob is read once), on clips whose crop
counts exercise full and partial tiles; prints the log-prob differences between the two and against the fp32 engine."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine
n, mb, dt, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
sd = synth.make_state_dict(seed=1234)
e = Engine(sd, max_batch_frames=mb, max_clip_frames=max(n, 64), max_frame_height=720, max_frame_width=1280, compute_dtype=dt)
r = e.infer_clip(synth.make_frames(n, 720, 1280), synth.make_boxes(n, 720, 1280))
np.save(out, r["logp"])
e.close()
''' % ROOT
tmp = tempfile.mkdtemp()
ok = True
for n, mb in ((21, 21), (40, 32), (70, 64)):
    res = {}
    for tag, env, dt in (("old", {"PA_BF16_PATCH": "0"}, "bf16"), ("new", {"PA_BF16_PATCH": "1"}, "bf16"), ("f32", {}, "f32")):
        out = os.path.join(tmp, f"{tag}_{n}.npy")
        subprocess.run([sys.executable, "-c", CHILD, str(n), str(mb), dt, out], check=True, env=dict(os.environ, **env), timeout=600)
        import numpy as np
        res[tag] = np.load(out)
    d_on = np.abs(res["new"] - res["old"]).max()
    d_nf = np.abs(res["new"] - res["f32"]).max()
    d_of = np.abs(res["old"] - res["f32"]).max()
    print(f"n={n} mb={mb}: |new-old| {d_on:.4f}  |new-f32| {d_nf:.4f}  |old-f32| {d_of:.4f}", flush=True)
    ok &= d_nf <= 5e-2 and np.isfinite(res["new"]).all()
print("AB_OK" if ok else "AB_FAIL")
sys.exit(0 if ok else 1)
