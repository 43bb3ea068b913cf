"""Per-layer HIP-event timings of the detection network (pa_detector_forward_timed): 64 x 1080p frames, YOLOv5s at 384 x 640.
Prints one row per layer of the table: kind, geometry, executed GFLOP, microseconds (median of 5 calls), TFLOP/s."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from playaid_core_amd import synth  # noqa: E402
from playaid_core_amd.yolov5 import YoloV5Detector, build_yolov5s_table  # noqa: E402

n, H, W = int(os.environ.get("N", 64)), 1080, 1920
dev = torch.device("cuda:0")
sd = synth.make_yolov5s_state_dict()
DTYPE = os.environ.get("DTYPE", "f32")   # f32 | emulated_f32
det = YoloV5Detector(sd, 6, (384, 640), max_images=n, device="cuda:0", compute_dtype=DTYPE)
layers = build_yolov5s_table(sd, (384, 640), 6)[0]
frames = torch.from_numpy(synth.make_frames(4, H, W)).to(dev).repeat((n + 3) // 4, 1, 1, 1)[:n].contiguous()
pred = torch.empty((n, det.rows, 11), dtype=torch.float32, device=dev)
us = np.zeros((7, len(layers)), np.float32)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for it in range(7):
    rc = det._lib.pa_detector_forward_timed(det._h, C.c_void_p(frames.data_ptr()), n, H, W, C.c_void_p(pred.data_ptr()), stream,
                                            us[it].ctypes.data_as(C.c_void_p), len(layers))
    assert rc == 0, det._lib.pa_detector_last_error(det._h)
med = np.median(us[2:], axis=0)
tot_us, tot_gf = 0.0, 0.0
names = {0: "conv", 3: "stem", 4: "pool5", 5: "up2", 6: "decode"}
for i, L in enumerate(layers):
    oh, ow = L.in_h // max(L.stride, 1), L.in_w // max(L.stride, 1)
    gf = 0.0
    if L.kind == 0:
        gf = 2.0 * n * oh * ow * L.cout * L.ksize * L.ksize * L.cin / 1e9
    elif L.kind == 3:
        gf = 2.0 * n * oh * ow * L.cout * 108 / 1e9
    tot_us += med[i]
    tot_gf += gf
    print(f"{i:3d} {names[L.kind]:6s} k{L.ksize} s{L.stride} {L.in_h:3d}x{L.in_w:3d} cin {L.cin:4d} cout {L.cout:4d} M {n * oh * ow:8d} "
          f"{gf:7.2f} GF {med[i]:8.1f} us {gf / med[i] * 1e3 if med[i] > 0 else 0:6.1f} TF  res {int(L.res_buf >= 0)}")
print(f"compute_dtype {DTYPE}")
print(f"total {tot_gf:.1f} GFLOP executed in {tot_us:.0f} us (events between layers) = {tot_gf / tot_us * 1e3:.1f} TFLOP/s")
