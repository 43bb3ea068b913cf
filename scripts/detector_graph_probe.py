"""Probe: the detection network's 60 launches replayed from a captured graph against eager launches (64 x 1080p)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from playaid_core_amd import synth
from playaid_core_amd.yolov5 import YoloV5Detector

dev = torch.device("cuda:0")
n, H, W = 64, 1080, 1920
frames = torch.from_numpy(synth.make_frames(4, H, W)).to(dev).repeat(16, 1, 1, 1).contiguous()
for dt in ("f32", "emulated_f32"):
    det = YoloV5Detector(synth.make_yolov5s_state_dict(), 6, (384, 640), max_images=n, device="cuda:0", compute_dtype=dt)
    out = torch.empty((n, det.rows, 11), dtype=torch.float32, device=dev)
    import ctypes as C
    def run():
        rc = det._lib.pa_detector_forward(det._h, C.c_void_p(frames.data_ptr()), n, H, W, C.c_void_p(out.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        assert rc == 0
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            run()
        s.synchronize()
        def timed(fn, reps=20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            for _ in range(reps):
                fn()
            b.record(s)
            s.synchronize()
            return a.elapsed_time(b) / reps
        eager = timed(run)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s):
                r = run()
            graph = timed(g.replay)
            print(f"{dt}: eager {eager:.3f} ms, graph replay {graph:.3f} ms per 64 frames")
        except Exception as e:
            print(f"{dt}: eager {eager:.3f} ms, capture failed: {type(e).__name__}: {str(e)[:200]}")
    det.close()
