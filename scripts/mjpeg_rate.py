"""Decode rate of pa_mjpeg_decode on one GPU: 64-frame 1080p clips, compressed bytes in pinned host memory.
usage: python scripts/mjpeg_rate.py [--frames 64] [--quality 95] [--height 1080 --width 1920] [--variants none,rows1,blk8]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from playaid_core_amd import synth, video  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--quality", type=int, default=95)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--variants", default="none,rows1,blk30,blk8,blk2")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--noise-mask", type=int, default=31, help="amplitude mask of the clip's per-sample noise (31 = the headline's clip, 7 = camera-like)")
ap.add_argument("--host-synth", action="store_true", help="generate the clip with numpy on the host (rocprofv3 counter passes do not get through the torch integer kernels of the device generator)")
ap.add_argument("--streams", type=int, default=1, help="decoders working at once, each on a stream of its own (calls alternate)")
ap.add_argument("--rounds", type=int, default=-1, help="verify passes enqueued per call (default: the library's)")
args = ap.parse_args()
n, h, w = args.frames, args.height, args.width
if not args.host_synth and (any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")):
    # profiles/README.md (round 3): under rocprofv3 --pmc the torch integer kernels of the device frame generator never came
    # back (three passes, 900 s). It is third-party code off the product path; counter passes take the host generator.
    sys.exit("mjpeg_rate.py under rocprofv3 needs --host-synth (the device frame generator hangs under counter collection)")
if args.host_synth:
    assert args.noise_mask == 31, "the host generator makes the headline's clip only"
    frames = synth.make_frames(n, h, w)
else:
    frames = synth.make_frames_torch(n, h, w, device="cuda", noise_mask=args.noise_mask, fine_mask=min(args.noise_mask, 15)).cpu().numpy()
for var in args.variants.split(","):
    kw = {}
    if var.startswith("rows"):
        kw["restart_marker_rows"] = int(var[4:])
    elif var.startswith("blk"):
        kw["restart_marker_blocks"] = int(var[3:])
    t0 = time.time()
    blobs = synth.encode_jpeg_frames(frames, quality=args.quality, **kw)
    enc_s = time.time() - t0
    sizes = np.array([len(b) for b in blobs])
    ends = np.cumsum(sizes)
    spans = np.stack([ends - sizes, ends], axis=1)
    data = torch.from_numpy(np.frombuffer(b"".join(blobs), np.uint8).copy()).pin_memory()
    dec = video.MjpegDecoder(n, h, w, int(ends[-1]) + 4096)
    if args.rounds >= 0:
        dec.set_sync_rounds(args.rounds)
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    st = torch.zeros(n, dtype=torch.int32, device="cuda")
    dec.decode(data, spans, h, w, out=out, status=st)
    torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    th = time.perf_counter()
    dec.decode(data, spans, h, w, out=out)  # the device is idle: nothing holds the host back but its own work
    host_ms = (time.perf_counter() - th) * 1e3
    torch.cuda.synchronize()
    if args.streams > 1:
        decs = [dec] + [video.MjpegDecoder(n, h, w, int(ends[-1]) + 4096) for _ in range(args.streams - 1)]
        outs = [out] + [torch.empty_like(out) for _ in range(args.streams - 1)]
        strs = [torch.cuda.Stream() for _ in range(args.streams)]
        for d_, o_, s_ in zip(decs, outs, strs):
            with torch.cuda.stream(s_):
                d_.decode(data, spans, h, w, out=o_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(args.reps * args.streams):
            with torch.cuda.stream(strs[r % args.streams]):
                decs[r % args.streams].decode(data, spans, h, w, out=outs[r % args.streams])
        torch.cuda.synchronize()
        ms_multi = (time.perf_counter() - t0) * 1e3 / (args.reps * args.streams)
        print(f"{var:8s} {args.streams} decoders on {args.streams} streams: {ms_multi:.3f} ms per {n} frames = {n / ms_multi * 1e3:.0f} frames/s", flush=True)
        for d_ in decs[1:]:
            d_.close()
    e0.record()
    for _ in range(args.reps):
        dec.decode(data, spans, h, w, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    print(f"{var:8s} q{args.quality} {sizes.mean() / 1e6:.3f} MB/frame  {ms:.3f} ms per {n} frames = {n / ms * 1e3:.0f} frames/s "
          f"({ends[-1] / ms / 1e6:.2f} GB/s compressed; host encode {enc_s / n * 1e3:.1f} ms/frame; the call returns after {host_ms:.2f} ms on the host)", flush=True)
    import ctypes as C
    from playaid_core_amd import _lib
    c = (C.c_ulonglong * 16)()
    _lib.load().pa_mjpeg_debug_counters(c)
    for m, nm in enumerate(("spec", "verify", "final")):
        cyc, ticks, syms = c[2 * m], c[2 * m + 1] >> 32, c[2 * m + 1] & 0xffffffff
        if syms:
            print(f"   {nm}: wave 0 walked {syms} symbols in {cyc} shader cycles = {cyc / syms:.0f} cycles/symbol, {ticks / 100:.1f} us "
                  f"-> {cyc / max(ticks, 1) * 100 / 1e3:.2f} GHz; slow steps {c[9 + 2 * m]} taking {c[8 + 2 * m]} cycles", flush=True)
    dec.close()
