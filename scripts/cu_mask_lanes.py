"""Experiment: the two lanes of ClipLanes on CU-MASKED streams (hipExtStreamCreateWithCUMask), each lane on its own half of
the chip, against the committed lanes (plain streams on distinct hardware queues, the whole chip shared). Prints frames/s
for the headline shape (64 x 1080p clips, fp32) per mask layout."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine
from playaid_core_amd.parallel import ClipLanes

n = 64
sd = synth.make_state_dict()
eng = Engine(sd, max_batch_frames=n, max_clip_frames=n)
frames = torch.from_numpy(synth.make_frames(8, 1080, 1920)).cuda().repeat(n // 8, 1, 1, 1).contiguous()
boxes = torch.from_numpy(synth.make_boxes(n, 1080, 1920)).cuda()
lanes = ClipLanes(eng, 7, 3, lanes=2)


def rate(clips=240):
    lanes._cold = True
    for _ in range(8):
        lanes.submit(frames, boxes, n)
    torch.cuda.synchronize()
    lanes.idle()
    t0 = time.perf_counter()
    for _ in range(clips):
        lanes.submit(frames, boxes, n)
    torch.cuda.synchronize()
    lanes.idle()
    return n * clips / (time.perf_counter() - t0)


lanes.calibrate(frames, boxes, n)
print("plain streams (calibrated):", " ".join(f"{rate():.0f}" for _ in range(3)), flush=True)

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value, device=eng.device)


layouts = {
    "low 128 / high 128": (range(0, 128), range(128, 256)),
    "even / odd": (range(0, 256, 2), range(1, 256, 2)),
    "i % 8 < 4 / >= 4": ([i for i in range(256) if i % 8 < 4], [i for i in range(256) if i % 8 >= 4]),
    "(i // 8) % 2": ([i for i in range(256) if (i // 8) % 2 == 0], [i for i in range(256) if (i // 8) % 2 == 1]),
    "all / all (masked API, full masks)": (range(256), range(256)),
    "192 / 192 overlapping": (range(0, 192), range(64, 256)),
}
for name, (a, b) in layouts.items():
    lanes.streams = [masked_stream(a), masked_stream(b)]
    print(f"{name}:", " ".join(f"{rate():.0f}" for _ in range(3)), flush=True)
lanes.close()
eng.close()
