#!/usr/bin/env python3
"""Headline benchmark: 1080p frames/s, raw BGR frame -> per-frame labels, on N
MI355X (BASELINE.json `metric`).

Two workloads, both with the frames already resident in HBM when the timed
region starts (no decode, no host->device copy inside it; the PCIe-inclusive
rate is measured separately and reported next to `value`, never as it):

* N = 1 (default): configs[1], batch = 64 x 1080p frames, fp32. One "clip" is
  one pass of the hot path over those 64 frames: square-crop + resample + /255
  (HIP), ResNet-18 on the 128 crops (fp32 MFMA implicit GEMM), temporal
  Conv1d/MLP head + log-softmax + argmax (HIP) -> pa_record per (frame,
  fighter). A timed step runs `--inner-repeat` such clips back to back so that
  the driver's 20 steps time ~0.6 s instead of 0.03 s; `ms_per_step` stays the
  time of ONE 64-frame clip (timed region / (steps x inner_repeat)).
* N > 1 (or `--clip-frames F`): configs[3], ONE F = 8192-frame clip sharded
  frame-parallel over the ranks (strong scaling): each rank holds, crops and
  runs the backbone on its 8192/N frames in chunks of 64, the 27-frame feature
  halo goes to the neighbouring rank(s) by RCCL send/recv underneath the head of
  the interior frames, one all-gather collects the records. A step is one pass
  over the whole clip.

Prints ONE JSON line on rank 0 (contract in the task prompt) with `roofline`
(dominant kernel family, timed with HIP events on the launch stream in a second
pass right after the timed region) and `cpu_baseline` (the CPU oracle,
reference-literal shape, on a bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from playaid_core_amd import synth
from playaid_core_amd.engine import Engine
from playaid_core_amd.parallel import ClipLanes, FrameParallelClip, broadcast_engine, halo_plan, shard_range

PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = fp32 vector peak
PEAK_HBM_GBS = 8000.0
TRAFFIC_FILES = {"f32": "r06_traffic.json", "bf16": "r04_cfg2_bf16_traffic.json"}  # (the bf16 conv path is round 4's: its counter passes were not repeated)
TRAFFIC_FALLBACK = {"f32": "r05_traffic.json", "bf16": "r01_cfg2_bf16_traffic.json"}


def _pmc_traffic(kernel_name, dtype, frames, height, width):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (produced by scripts/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same
    command -- counters cannot be read from inside this process, so the figure is REPLAYED from the
    file named in `traffic_source`, not measured in this run). (None, None) when no measurement of
    that kernel on that workload shape is on file."""
    extra = ("r02_clipbatch4_traffic.json",) if dtype == "f32" else ()   # 256-frame backbone batches (clip batches, long clip)
    for name in (TRAFFIC_FILES[dtype],) + extra + (TRAFFIC_FALLBACK[dtype],):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
        except Exception:
            continue
        if t.get("kernel") == kernel_name and t.get("workload") == {"frames": frames, "height": height, "width": width}:
            return t["traffic_bytes_per_launch"], f"profiles/{name} (committed rocprofv3 PMC pass, replayed)"
    return None, None


_ORACLE_SAMPLE = {}   # cpu_baseline's sample and the oracle's log-probabilities on it (see there)


def cpu_baseline(sd, height, width, sample_frames):
    """Reference-literal CPU path (oracle = "port"): batch 1 per (frame,
    fighter), 7 backbone forwards per window, crops through the PIL/cv
    restatement -- the shape of ai_runner.py:493-520."""
    from oracle import pipeline  # checker / baseline only

    frames = synth.make_frames(sample_frames, height, width)
    boxes = synth.make_boxes(sample_frames, height, width)
    t0 = time.perf_counter()
    crops, ok = pipeline.crops_for_clip(frames, boxes)
    t_crop = time.perf_counter() - t0
    pipeline.run_action_recognition(frames, boxes, sd, mode="literal", crops_rgb=crops)
    dt = time.perf_counter() - t0
    # SURVEY 8d's second CPU mode: the GPU path's own algorithm (feature cache, batched backbone)
    t1 = time.perf_counter()
    cached = pipeline.run_action_recognition(frames, boxes, sd, mode="cached", crops_rgb=crops)
    dt_batched = time.perf_counter() - t1 + t_crop
    # (the one place bench.py runs the oracle: its log-probabilities on this sample are also the CHECK of the exact and the
    # emulated-fp32 engine in the `emulated_fp32` block -- checker, never the thing measured)
    _ORACLE_SAMPLE.update(frames=frames, boxes=boxes, logp=np.asarray(cached["logp"], dtype=np.float64))
    return {
        "value": round(sample_frames / dt, 3),
        "unit": "frames/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{sample_frames} synthetic {height}x{width} frames ({2 * (sample_frames - 1)} windows, "
        f"reference-literal: batch 1, 7 ResNet-18 forwards per window), {dt:.1f} s",
        "batched_value": round(sample_frames / dt_batched, 3),
        "batched_note": "same sample through the feature-cached, batch-32 formulation the GPU path uses (not the reference's shape)",
    }


def pcie_inclusive(eng, frames_dev, boxes_dev, steps=8):
    """The same 64-frame clip with the frames starting in pinned HOST memory: every clip's 398 MB
    cross PCIe on a side stream into one of two device buffers while the previous clip computes
    (DESIGN.md section 6). Reported beside `value`, never as it."""
    dev = eng.device
    n = frames_dev.shape[0]
    host = frames_dev.cpu().pin_memory()
    bufs = [torch.empty_like(frames_dev), torch.empty_like(frames_dev)]
    rec = eng.alloc_records(n - 1)
    side, main = torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    free = [torch.cuda.Event(), torch.cuda.Event()]

    def run(k_steps):
        for e in free:
            e.record(main)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(k_steps + 1):
            if k < k_steps:
                with torch.cuda.stream(side):
                    side.wait_event(free[k & 1])
                    bufs[k & 1].copy_(host, non_blocking=True)
                    ready[k & 1].record(side)
            if k > 0:
                j = (k - 1) & 1
                main.wait_event(ready[j])
                eng.infer_clip_device(bufs[j], boxes_dev, rec)
                free[j].record(main)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / k_steps

    run(2)
    dt = run(steps)
    return {
        "value": round(n / dt, 1),
        "unit": "frames/s",
        "ms_per_clip": round(dt * 1e3, 3),
        "h2d_MB_per_clip": round(host.numel() / 1e6, 1),
        "method": f"{steps} clips, frames in pinned host memory, whole-frame H2D copy of clip k+1 on a side stream under the "
        "compute of clip k (two device buffers); no decode",
    }


def compact_line(result):
    """The ONE stdout line: what the round is judged on, under 1800 characters (the driver keeps the last 2000 of a run's
    output). Everything else -- notes, per-kernel table, calibration, side measurements in full -- goes to bench_details.json
    (and to stderr, before this line)."""
    cfg = result.get("config", {})
    out = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data")}
    out["metric"] = out["metric"].split(" (")[0] if out.get("metric") else None
    out["config"] = {"workload": (cfg.get("workload") or "").split(", 2 fighters")[0], "parallelism": cfg.get("parallelism"),
                     "lanes": cfg.get("lanes"), "ingest": "raw BGR frames resident in HBM"}
    if cfg.get("exchange"):
        out["config"]["exchange"] = cfg["exchange"]
    r = result.get("roofline")
    if r:
        out["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "algorithmic_tflops", "algorithmic_frac",
                                                  "executed_frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms") if k in r}
    c = result.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
                               "sample": c["sample"].split(" (")[0] + ", reference-literal", "batched_value": c.get("batched_value")}
    def side(name, short):   # a side measurement's rate, or its error text
        d = result.get(name)
        if isinstance(d, dict):
            out[short] = d.get("value", d.get("error"))
    ch = result.get("chain_inclusive")
    if isinstance(ch, dict) and "stage_ms_per_clip_alone" in ch:
        st = ch["stage_ms_per_clip_alone"]
        out["chain_frames_per_s"] = ch["value"]
        out["chain_stage_ms"] = {"decode": st["mjpeg_decode"], "detector": st["detector_network"], "nms_repair": st["nms_and_label_repair"],
                                 "crops_cnn_head": st["detector_crops_jpeg_runner_inputs_cnn_head"]}
    else:
        side("chain_inclusive", "chain_frames_per_s")
    side("chain_inclusive_camera_like", "chain_camera_like_frames_per_s")
    side("decode_inclusive", "decode_inclusive_frames_per_s")
    side("decode_inclusive_camera_like", "decode_camera_like_frames_per_s")
    side("pcie_inclusive_windows", "pcie_inclusive_frames_per_s")
    e = result.get("emulated_fp32")
    if isinstance(e, dict):
        out["emulated_fp32"] = {k: e.get(k) for k in ("frames_per_s", "frames_per_s_exact_f32", "detector_ms", "detector_ms_exact_f32", "chain_frames_per_s", "chain_camera_like_frames_per_s", "max_dlogp_vs_oracle",
                                                      "max_dlogp_vs_oracle_exact_f32", "max_dlogp_emulated_vs_exact", "error") if k in e}
        if isinstance(e.get("roofline"), dict):
            out["emulated_fp32"]["roofline"] = {k: e["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "fp32_equivalent_tflops")}
    out["details"] = "bench_details.json"
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > 1800:   # never let a long note push the headline out of the driver's tail
        for k in ("emulated_fp32", "chain_stage_ms", "cpu_baseline"):
            if k == "cpu_baseline" and k in out:
                out[k].pop("sample", None)
            else:
                out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= 1800:
                break
    return line


def emit(result):
    """Full record -> bench_details.json (+ gpurun_out/ when present) and one stderr line; the compact record -> the LAST stdout line."""
    full = json.dumps(result)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_details.json"), "w") as f:
                    f.write(full + "\n")
        except OSError:
            pass
    print(full, file=sys.stderr, flush=True)
    print(compact_line(result), flush=True)


def emulated_fp32_side(sd, frames, boxes, n_clip, height, width, S, DELTA, device, quality):
    """the fp32 path with its convolutions' products on the bf16 matrix cores (compute_dtype
    "emulated_f32" = PA_DTYPE_EMULATED_F32: three bf16 slices per fp32 operand, six bf16 matrix instructions per fp32 product,
    fp32 accumulation; csrc/psgemm.hip). Never `value`: the headline and `dtype` stay on the exact fp32 kernels. Reports the
    headline shape's rate, the detector stage, the chain, and the error of BOTH paths against the oracle's log-probabilities on the
    CPU baseline's sample (the error against a float64 run of the detection network: profiles/r06_yolov5_parity.txt)."""
    from playaid_core_amd.yolov5 import YoloV5Detector

    out = {"what": "compute_dtype emulated_f32: fp32 in / fp32 out, fp32-accurate sums; the detector's 1x1 and stride-2 3x3 convolutions and the "
                   "ResNet-18's stride-2 openers + 1x1/2 branch GEMMs on v_mfma_f32_32x32x16_bf16 (csrc/psgemm.hip); the stride-1 3x3 layers "
                   "keep their exact Winograd kernel (measured equal per layer, profiles/r06_pgemm_split_layers.txt); the detector's 6x6 stem on integer "
                   "pixels as one exact bf16 value each (stem6x6_bf16_kernel); the ResNet stem and the heads exact"}
    def two_lane_rate(e):   # the headline's shape, this block's own protocol (8 clips of warm-up, 100 timed) -- for BOTH dtypes, side by side
        lanes = ClipLanes(e, S, DELTA, lanes=2)
        try:
            for _ in range(8):
                lanes.submit(frames, boxes, n_clip)
            torch.cuda.synchronize(device)
            lanes.idle()
            k = 100
            t0 = time.perf_counter()
            for _ in range(k):
                lanes.submit(frames, boxes, n_clip)
            torch.cuda.synchronize(device)
            return round(n_clip * k / (time.perf_counter() - t0), 1)
        finally:
            lanes.close()

    eng_x = Engine(sd, device=str(device), max_batch_frames=n_clip, max_clip_frames=max(n_clip, 64), max_frame_height=height, max_frame_width=width)
    try:
        out["frames_per_s_exact_f32"] = two_lane_rate(eng_x)
    finally:
        eng_x.close()
    eng = Engine(sd, device=str(device), max_batch_frames=n_clip, max_clip_frames=max(n_clip, 64), max_frame_height=height, max_frame_width=width,
                 compute_dtype="emulated_f32")
    try:
        out["frames_per_s"] = two_lane_rate(eng)
        # both arithmetic choices on cpu_baseline's sample: against each other always, against the oracle's log-probabilities where
        # the CPU baseline ran (its leg is the only place this file executes the oracle)
        fs, bs = _ORACLE_SAMPLE.get("frames"), _ORACLE_SAMPLE.get("boxes")
        if fs is None:
            fs, bs = synth.make_frames(8, height, width), synth.make_boxes(8, height, width)
        got = {}
        for dt_ in ("f32", "emulated_f32"):
            e8 = Engine(sd, device=str(device), max_batch_frames=min(len(fs), 64), max_clip_frames=max(len(fs), 64), max_frame_height=height,
                        max_frame_width=width, compute_dtype=dt_)
            try:
                got[dt_] = e8.infer_clip(fs, bs)["logp"].astype(np.float64)
            finally:
                e8.close()
        out["max_dlogp_emulated_vs_exact"] = float(f"{np.abs(got['emulated_f32'] - got['f32']).max():.3e}")
        if "logp" in _ORACLE_SAMPLE:
            out["max_dlogp_vs_oracle"] = float(f"{np.abs(got['emulated_f32'] - _ORACLE_SAMPLE['logp']).max():.3e}")
            out["max_dlogp_vs_oracle_exact_f32"] = float(f"{np.abs(got['f32'] - _ORACLE_SAMPLE['logp']).max():.3e}")
            out["oracle_sample"] = f"cpu_baseline's {len(fs)} frames ({2 * (len(fs) - 1)} windows)"
        # detector stage alone (64 frames, events on one stream), both dtypes
        for dt_, key in (("f32", "detector_ms_exact_f32"), ("emulated_f32", "detector_ms")):
            det = YoloV5Detector(synth.make_yolov5s_state_dict(), 6, (384, 640), max_images=n_clip, device=str(device), compute_dtype=dt_)
            try:
                ts = []
                for _ in range(6):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    det(frames)
                    b.record()
                    torch.cuda.synchronize(device)
                    ts.append(a.elapsed_time(b))
                out[key] = round(float(np.median(ts[2:])), 3)
                if dt_ == "emulated_f32":
                    # roofline of the emulated kernel itself: the layers that run on psgemm.hip (every 1x1 / stride-2 3x3 convolution), timed
                    # live with a HIP event between the table's layers (pa_detector_forward_timed), against the bf16 matrix peak: every fp32
                    # multiply-add is six bf16 ones, so fp32-equivalent FLOP/s x 6 = the bf16 FLOP/s the matrix cores execute
                    import ctypes as C
                    from playaid_core_amd.yolov5 import build_yolov5s_table

                    layers = build_yolov5s_table(synth.make_yolov5s_state_dict(), (384, 640), 6)[0]
                    pred = torch.empty((n_clip, det.rows, 11), dtype=torch.float32, device=device)
                    us = np.zeros((5, len(layers)), np.float32)
                    for it in range(5):
                        rc = det._lib.pa_detector_forward_timed(det._h, C.c_void_p(frames.data_ptr()), n_clip, height, width, C.c_void_p(pred.data_ptr()),
                                                                C.c_void_p(torch.cuda.current_stream(device).cuda_stream), us[it].ctypes.data_as(C.c_void_p), len(layers))
                        assert rc == 0
                    med = np.median(us[1:], axis=0)
                    gf = t_us = 0.0
                    for i, L in enumerate(layers):
                        if L.kind == 0 and not (L.ksize == 3 and L.stride == 1):
                            gf += 2.0 * n_clip * (L.in_h // L.stride) * (L.in_w // L.stride) * L.cout * L.ksize * L.ksize * L.cin / 1e9
                            t_us += float(med[i])
                    tf_eq = gf / t_us * 1e3
                    out["roofline"] = {"kernel": "psgemm_kernel (40 layers of the detector: 1x1 and stride-2 3x3)", "bound": "mfma",
                                       "achieved": round(tf_eq * 6, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf_eq * 6 / 2500.0, 4),
                                       "fp32_equivalent_tflops": round(tf_eq, 1), "gflop_fp32": round(gf, 1), "total_ms": round(t_us / 1e3, 3),
                                       "note": "bf16 FLOP/s executed (six per fp32 multiply-add) against the 2.5 PFLOP/s dense bf16 peak at 2.4 GHz; in-kernel stamps "
                                               "(profiles/r06_pgemm_split_stamps.txt): the loop runs at 0.96 of the matrix pipe's cycles, the chip holds 1.53-1.57 GHz in it"}
            finally:
                det.close()
        ch = chain_inclusive(eng, sd, frames, boxes, quality=quality, compute_dtype="emulated_f32")
        out["chain_frames_per_s"] = ch["value"]
        out["chain_stage_ms"] = ch["stage_ms_per_clip_alone"]
        # ... and on the content it will meet (three bits of noise per sample: the compressed size of camera / game footage)
        quiet = synth.make_frames_torch(n_clip, height, width, first_frame=0, device=device, noise_mask=7, fine_mask=7)
        chq = chain_inclusive(eng, sd, quiet, boxes, quality=quality, compute_dtype="emulated_f32")
        out["chain_camera_like_frames_per_s"] = chq["value"]
        out["chain_camera_like_stage_ms"] = chq["stage_ms_per_clip_alone"]
        del quiet
    finally:
        eng.close()
    return out


_ENCODED = {}


def _encoded_clip(frames_dev, quality, restart_blocks):
    """The clip as a Motion-JPEG stream in pinned host memory (libjpeg-turbo via Pillow, outside every timed region); encoded
    once per (clip, quality) and shared by the side measurements."""
    key = (frames_dev.data_ptr(), tuple(frames_dev.shape), quality, restart_blocks)
    if key not in _ENCODED:
        t0 = time.perf_counter()
        kw = {"restart_marker_blocks": restart_blocks} if restart_blocks else {}
        blobs = synth.encode_jpeg_frames(frames_dev.cpu().numpy(), quality=quality, **kw)
        enc_s = time.perf_counter() - t0
        sizes = np.array([len(b) for b in blobs], dtype=np.int64)
        ends = np.cumsum(sizes)
        spans = np.stack([ends - sizes, ends], axis=1)
        data = torch.from_numpy(np.frombuffer(b"".join(blobs), np.uint8).copy()).pin_memory()
        _ENCODED.clear()   # (one clip at a time: the pinned bytes of the previous one are released)
        _ENCODED[key] = (data, spans, sizes, ends, enc_s)
    return _ENCODED[key]


def decode_inclusive(eng, frames_dev, boxes_dev, steps=24, quality=95, restart_blocks=0, content=None):
    """decode -> labels: the clip as a Motion-JPEG stream (one baseline JPEG per frame, written by libjpeg-turbo at
    OpenCV's defaults: quality 95, 4:2:0) in pinned HOST memory; per clip the compressed bytes cross PCIe and are decoded
    on the device (pa_mjpeg_decode: un-stuffing, self-synchronising Huffman decoding, IDCT, up-sampling, colour
    conversion) on a side stream into a frame buffer of its own while an earlier clip runs crops + CNN + head; three
    decoders work at once. Replaces cv2.VideoCapture.read (ai_runner.py:153,404-405). Reported beside `value`, never as it."""
    from playaid_core_amd import video

    dev = eng.device
    n, h, w, _ = frames_dev.shape
    data, spans, sizes, ends, enc_s = _encoded_clip(frames_dev, quality, restart_blocks)
    # ND decoders at once, each on a stream of its own with a frame buffer of its own: the entropy passes are bound by
    # instruction issue and single-wave latency, so calls whose phases are out of step fill each other's idle time
    ND = 4
    decs = [video.MjpegDecoder(n, h, w, int(ends[-1]) + 4096, device=str(dev)) for _ in range(ND)]
    for d_ in decs:
        d_.set_groups(1)
    bufs = [torch.empty_like(frames_dev) for _ in range(ND)]
    st = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(ND)]
    rec = eng.alloc_records(n - 1)
    sides, main = [torch.cuda.Stream(dev) for _ in range(ND)], torch.cuda.current_stream(dev)
    ready = [torch.cuda.Event() for _ in range(ND)]
    free = [torch.cuda.Event() for _ in range(ND)]

    def run(k_steps, compute=True):
        for e in free:
            e.record(main)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(k_steps + ND - 1):
            if k < k_steps:
                i = k % ND
                with torch.cuda.stream(sides[i]):
                    sides[i].wait_event(free[i])
                    decs[i].decode(data, spans, h, w, out=bufs[i], status=st[i])
                    ready[i].record(sides[i])
            if k >= ND - 1:
                j = (k - (ND - 1)) % ND
                main.wait_event(ready[j])
                if compute:
                    eng.infer_clip_device(bufs[j], boxes_dev, rec)
                free[j].record(main)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / k_steps

    try:
        run(ND)
        bad = int(sum(int((s != 0).sum()) for s in st))
        dt = run(steps)
        dt_dec = run(steps, compute=False)
        psnr = None
        if bad == 0:
            d = (bufs[0][:4].float() - frames_dev[:4].float())
            psnr = round(float(10 * torch.log10(255.0 ** 2 / (d * d).mean())), 2)
    finally:
        for d_ in decs:
            d_.close()
    return {
        "value": round(n / dt, 1),
        "unit": "frames/s",
        "ms_per_clip": round(dt * 1e3, 3),
        "decode_only_frames_per_s": round(n / dt_dec, 1),
        "decode_only_ms_per_clip": round(dt_dec * 1e3, 3),
        "compressed_MB_per_clip": round(float(ends[-1]) / 1e6, 2),
        "compressed_MB_per_frame": round(float(sizes.mean()) / 1e6, 3),
        "jpeg": {"quality": quality, "subsampling": "4:2:0", "restart_interval_mcus": restart_blocks or None,
                 "encoder": "libjpeg-turbo via Pillow (host, outside the timed region)", "host_encode_ms_per_frame": round(enc_s / n * 1e3, 1),
                 "psnr_vs_raw_dB": psnr},
        "frames_with_decode_errors": bad,
        "content": content or "the headline's clip (five bits of white noise on every sample: the hardest input a JPEG coder meets)",
        "method": f"{steps} clips; compressed frames in pinned host memory; H2D copy + device Motion-JPEG decode of clip k+1 on a "
        "side stream under crops + CNN + head of an earlier clip on its decoded frames; four decoders, each with a side stream and "
        "a frame buffer of its own, work at once (clips k+1 .. k+4)",
    }


def synthetic_head_rows(boxes_dev, rows, nc, height, width, net_hw=(384, 640)):
    """Head rows float32[n, rows, 5 + nc] (device) as pa_detector_forward lays them out, carrying the clip's TRUE fighter boxes:
    three near-duplicate candidates per fighter (classes 2 and 3, objectness 0.95 / 0.85 / 0.75) among clutter below the
    objectness gate -- what a trained detector would hand to NMS. Seeded random-init weights detect nothing, so the chain's
    post-processing consumes these while the network's own rows (computed in full) are discarded."""
    n = boxes_dev.shape[0]
    dev = boxes_dev.device
    gain = min(net_hw[0] / height, net_hw[1] / width)
    pad_x, pad_y = (net_hw[1] - width * gain) / 2, (net_hw[0] - height * gain) / 2
    g = torch.Generator(device="cpu").manual_seed(99)
    pred = torch.zeros((n, rows, 5 + nc), dtype=torch.float32)
    pred[:, :, 4] = torch.rand((n, rows), generator=g) * 0.2
    pred[:, :, :4] = 10 + torch.rand((n, rows, 4), generator=g) * 290
    pred = pred.to(dev)
    b = boxes_dev.to(torch.float32)
    for p in range(2):
        for k in range(3):
            r = 10 * p + k
            pred[:, r, 0] = b[:, p, 0] * width * gain + pad_x + k
            pred[:, r, 1] = b[:, p, 1] * height * gain + pad_y - k
            pred[:, r, 2] = b[:, p, 2] * width * gain
            pred[:, r, 3] = b[:, p, 3] * height * gain
            pred[:, r, 4] = 0.95 - 0.1 * k
            pred[:, r, 5:] = 0.0
            pred[:, r, 5 + 2 + p] = 0.9
    if n > 8:
        pred[5:8, 10:13, 4] = 0.0   # the detector loses fighter 1 in frames 6-8: three square_crop repairs per clip
    return pred


def chain_inclusive(eng, sd, frames_dev, boxes_dev, steps=12, quality=95, compute_dtype="f32"):
    """The path north_star names, chained (ai_runner.py:181-189 run_detection_setup -> :191-224 YOLO -> :226-424 repair -> :426-520
    windows -> CNN -> labels): the clip as Motion-JPEG bytes in pinned host memory -> pa_mjpeg_decode -> pa_detector_forward
    (YOLOv5s) -> pa_detect_postprocess (NMS, --max-det 2 --classes 2 3) -> pa_clean_detections -> pa_save_one_box_crops (+ the
    crops' 4:4:4 JPEG write / read; square_crop repairs where the detector lost a fighter) -> pa_backbone_crop_images (runner
    resize / letterbox + ResNet-18) -> temporal head -> records. Clips are pipelined three deep: decode of clip k on a decoder
    stream, detector + NMS + repair of clip k-1 on a second stream, crops + CNN + head of clip k-2 on a third; the host waits
    once per clip, for the repair's five words. Reported beside `value`, never as it."""
    from playaid_core_amd import detector_path, video
    from playaid_core_amd.yolov5 import YoloV5Detector

    dev = eng.device
    n, h, w, _ = frames_dev.shape
    data, spans, sizes, ends, _ = _encoded_clip(frames_dev, quality, 0)
    ND = 3
    decs = [video.MjpegDecoder(n, h, w, int(ends[-1]) + 4096, device=str(dev)) for _ in range(ND)]
    bufs = [torch.empty_like(frames_dev) for _ in range(ND)]
    st = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(ND)]
    det = YoloV5Detector(synth.make_yolov5s_state_dict(), 6, (384, 640), max_images=n, device=str(dev), compute_dtype=compute_dtype)
    front_eng = Engine(sd, device=str(dev), max_batch_frames=8, max_clip_frames=max(n, 64), max_frame_height=h, max_frame_width=w)
    pred_syn = synthetic_head_rows(boxes_dev, det.rows, 6, h, w)
    pred_net = torch.empty((n, det.rows, 11), dtype=torch.float32, device=dev)
    # the stages' streams are probed against each other like the lanes' (parallel._concurrent_streams): the HIP runtime multiplexes
    # streams onto a few hardware queues, and when the detector's and the CNN's stream happened to share one the three-deep
    # pipeline ran SLOWER than the stages one after the other (5.8 k against 7.0 k frames/s from one process to the next)
    from playaid_core_amd.parallel import _concurrent_streams
    picked = _concurrent_streams(eng, 2 + ND)
    s_front, s_back = picked[0], picked[1]
    if os.environ.get("PA_CHAIN_BACK_PRIO"):   # A/B: the oldest clip's stage (crops + CNN + head) on a stream of the greatest priority
        s_back = torch.cuda.Stream(dev, priority=-1)
    if os.environ.get("PA_CHAIN_FRONT_PRIO"):
        s_front = torch.cuda.Stream(dev, priority=-1)
    s_dec = picked[2:2 + ND]
    ready = [torch.cuda.Event() for _ in range(ND)]
    free = [torch.cuda.Event() for _ in range(ND)]
    import ctypes as C

    def detector(frames):
        rc = det._lib.pa_detector_forward(det._h, C.c_void_p(frames.data_ptr()), n, h, w, C.c_void_p(pred_net.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        assert rc == 0, det._lib.pa_detector_last_error(det._h)

    def front(i):
        detector(bufs[i])
        dets, counts = front_eng.detect_postprocess(pred_syn, det.net_hw, (h, w))
        return detector_path.begin(front_eng, bufs[i], dets, counts)

    def run(k_steps):
        keep, tickets, last = [], {}, None
        for e in free:
            e.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(k_steps + 2):
            if k >= 2:   # crops + CNN + head of clip k - 2
                c = k - 2
                with torch.cuda.stream(s_back):
                    t = tickets.pop(c)
                    s_back.wait_event(t.event)
                    last = detector_path.finish(eng, t, jpeg_quality=95, device_results=True)
                    free[c % ND].record(s_back)
                    keep.append((t, last))   # (tensors of one stream read on another: alive until the run's final synchronisation)
            if 1 <= k <= k_steps:   # detector + NMS + repair of clip k - 1
                c = k - 1
                with torch.cuda.stream(s_front):
                    s_front.wait_event(ready[c % ND])
                    tickets[c] = front(c % ND)
            if k < k_steps:   # decode of clip k
                i = k % ND
                with torch.cuda.stream(s_dec[i]):
                    s_dec[i].wait_event(free[i])
                    decs[i].decode(data, spans, h, w, out=bufs[i], status=st[i])
                    ready[i].record(s_dec[i])
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / k_steps, last

    def stage_ms(fn, reps=3):
        ts = []
        for _ in range(reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            a.record()
            out = fn()
            b.record()
            torch.cuda.synchronize(dev)
            ts.append(a.elapsed_time(b))
        return float(np.median(ts[1:])), out

    try:
        run(3)
        bad = int(sum(int((s_ != 0).sum()) for s_ in st))
        dt, last = run(steps)
        front_eng.check_device_errors()
        eng.check_device_errors()
        rec = Engine.decode_records(last["records"])
        finite = bool(torch.isfinite(last["logp"]).all())
        # per-stage times, one clip alone on one stream (nothing overlaps; the pipelined rate above is what counts)
        ms_decode, _ = stage_ms(lambda: decs[0].decode(data, spans, h, w, out=bufs[0], status=st[0]))
        ms_det, _ = stage_ms(lambda: detector(bufs[0]))

        def nms_and_repair():   # its own event pair (round 4 subtracted two medians and printed a negative time)
            dets, counts = front_eng.detect_postprocess(pred_syn, det.net_hw, (h, w))
            return detector_path.begin(front_eng, bufs[0], dets, counts)

        ms_nms, tk = stage_ms(nms_and_repair)
        tk.event.synchronize()

        def back():
            return detector_path.finish(eng, tk, jpeg_quality=95, device_results=True)

        ms_back, _ = stage_ms(back)
        ms_cnn, _ = stage_ms(lambda: eng.infer_clip_device(bufs[0], boxes_dev, eng.alloc_records(n - 1)))
    finally:
        for d_ in decs:
            d_.close()
        det.close()
        front_eng.close()
    stages = {"mjpeg_decode": round(ms_decode, 3), "detector_network": round(ms_det, 3), "nms_and_label_repair": round(ms_nms, 3),
              "detector_crops_jpeg_runner_inputs_cnn_head": round(ms_back, 3)}
    return {
        "value": round(n / dt, 1),
        "unit": "frames/s",
        "ms_per_clip": round(dt * 1e3, 3),
        "stage_ms_per_clip_alone": stages,
        "stage_sum_ms": round(sum(stages.values()), 3),
        "bound_by": max(stages, key=stages.get),
        "crops_plus_cnn_of_the_headline_formulation_ms": round(ms_cnn, 3),
        "compressed_MB_per_clip": round(float(ends[-1]) / 1e6, 2),
        "compute_dtype": compute_dtype,
        "frames_with_decode_errors": bad,
        "labels_finite": finite,
        "actions_in_last_clip": int(len(np.unique(rec["action_id"]))),
        "detections": "the detection NETWORK runs in full on the decoded frames (YOLOv5s v7.0, 384 x 640, seeded random-init weights: its "
                      "rows detect nothing and are discarded); NMS, repair, crops and labels consume synthetic head rows of the same shape "
                      "carrying the clip's true fighter boxes (bench.py::synthetic_head_rows)",
        "method": f"{steps} clips of {n} frames, three in flight: decode (clip k) | detector + NMS + repair (k - 1) | save_one_box crops + "
                  "4:4:4 JPEG + runner inputs + ResNet-18 + head (k - 2), each on its own stream; one host wait per clip (five words of "
                  "the repair); compressed bytes cross PCIe inside the decode stage",
    }


def pcie_inclusive_windows(eng, frames_dev, boxes_dev, steps=8):
    """PCIe-inclusive again, but only the crops' source slices cross the link (pa_upload_crop_windows: ~0.42 MB per
    1080p crop instead of 6.2 MB per frame), double buffered like above."""
    dev = eng.device
    n, h, w, _ = frames_dev.shape
    host = frames_dev.cpu().pin_memory()
    boxes_host = boxes_dev.cpu().numpy()
    stages = [eng.make_window_stage(n), eng.make_window_stage(n)]
    rec = eng.alloc_records(n - 1)
    side, main = torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    free = [torch.cuda.Event(), torch.cuda.Event()]

    def run(k_steps):
        for e in free:
            e.record(main)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(k_steps + 1):
            if k < k_steps:
                with torch.cuda.stream(side):
                    side.wait_event(free[k & 1])
                    eng.upload_crop_windows(host, boxes_host, stages[k & 1])
                    ready[k & 1].record(side)
            if k > 0:
                j = (k - 1) & 1
                main.wait_event(ready[j])
                eng.clip_begin(n)
                eng.preprocess_windows(stages[j], n, h, w, boxes_dev, 0)
                free[j].record(main)  # the windows are consumed once the crop stage has run
                eng.backbone_slot(0, n, 0)
                eng.head_frames(1, n, rec, None)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / k_steps

    run(2)
    dt = run(steps)
    return {
        "value": round(n / dt, 1),
        "unit": "frames/s",
        "ms_per_clip": round(dt * 1e3, 3),
        "h2d_MB_per_clip": round(stages[0]["used"] / 1e6, 1),
        "method": f"{steps} clips, frames in pinned host memory; only every crop's source slice crosses PCIe, read straight out of "
        "the host frames by one upload kernel per clip (slice geometry computed on the host with the device plan's arithmetic), "
        "uploads of clip k+1 on a side stream under the compute of clip k; no decode",
    }


def bench_f4(args):
    """the reference's alternative temporal models on the same conv kernels, timed per call
    (windows in, log-probabilities out, input resident in HBM). rnn = RNNActionDetector (ResNet-18 -> 3-layer LSTM,
    models/rnn_action_detector.py:74-95), resformer = ResnetTransformerDetector (timm ResNet-50 -> 3-layer transformer
    encoder, models/resnet_transformer_detector.py:65-93). Single GPU, own metric (never the headline)."""
    from playaid_core_amd.anim_ontology import ACTIONS

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    b, s_len = args.f4_windows, 7
    f4_dtype = "emulated_f32" if args.dtype == "emulated_f32" else "f32"
    if args.workload == "rnn":
        from playaid_core_amd.rnn_action_detector import RNNActionDetector
        model = RNNActionDetector("Joker", ACTIONS, state_dict=synth.make_rnn_state_dict(), max_rows=b * s_len, compute_dtype=f4_dtype)
        backbone_gflop_per_crop = 1.1843  # ResNet-18 at 128 x 128 with fc -> 300 (SURVEY 8a7 minus the unused fc rows)
        name = "RNNActionDetector: ResNet-18 (fc 300) -> LSTM(300, 512, 3 layers) -> 512-128-A"
    else:
        from playaid_core_amd.resnet_transformer_detector import ResnetTransformerDetector
        model = ResnetTransformerDetector(ACTIONS, sequence_length=s_len, state_dict=synth.make_resformer_state_dict(sequence_length=s_len),
                                          max_rows=b * s_len, compute_dtype=f4_dtype)
        backbone_gflop_per_crop = 2.0 * 1.339  # timm resnet50 at 128 x 128: 1.339 GMAC
        name = "ResnetTransformerDetector: ResNet-50 -> Linear(2048, 247) + time encoding -> 3 encoder layers (d 256, 8 heads, ff 2048) -> A"
    x = (torch.rand((b, s_len, 3, 128, 128), device=device) * 255).round() / 255
    for _ in range(args.warmup):
        out = model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    tf = backbone_gflop_per_crop * b * s_len / dt / 1e3
    print(json.dumps({
        "metric": f"windows/sec, {args.workload} temporal model (7 x 128 x 128 windows resident in HBM -> per-frame log-probabilities)",
        "value": round(b / dt, 1), "unit": "windows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": f4_dtype,
        "data": "synthetic (seeded random-init weights, random windows)",
        "config": {"workload": f"{args.workload}: {name}; {b} windows ({b * s_len} crops) per call", "windows_per_call": b},
        "backbone_tflops_if_all_time_were_backbone": round(tf, 2),
        "backbone_frac_of_fp32_matrix_peak_lower_bound": round(tf / PEAK_FP32_MATRIX_TFLOPS, 4),
        "finite": bool(torch.isfinite(out).all()),
    }), flush=True)
    model.close()


def bench_detect(args):
    """the detection stage the reference runs as a YOLOv5 subprocess (ai_runner.py:191-224) --
    frames resident in HBM -> letterbox -> YOLOv5s (seeded random-init weights, 6 classes) -> Detect decode -> confidence gates
    + class-aware NMS (--max-det 2 --classes 2 3) -> label rows. Single GPU, own metric (never the headline)."""
    from playaid_core_amd.yolov5 import YoloV5Detector

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    n = args.frames
    # (the host generator: rocprofv3 counter passes do not survive the torch integer kernels of the device one -- under
    # --pmc this workload crashed inside the next hipMemcpy)
    frames = torch.from_numpy(synth.make_frames(n, args.height, args.width)).to(device)
    eng = Engine(synth.make_state_dict(seed=1234), device=str(device), max_batch_frames=8, max_clip_frames=64, max_frame_height=args.height,
                 max_frame_width=args.width)
    det = YoloV5Detector(synth.make_yolov5s_state_dict(), 6, (384, 640), max_images=n, device=str(device),
                         compute_dtype="emulated_f32" if args.dtype == "emulated_f32" else "f32")
    for _ in range(args.warmup):
        dets, counts = det.detections(eng, frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dets, counts = det.detections(eng, frames)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    tf = det.flops_per_image * n / dt / 1e12
    print(json.dumps({
        "metric": f"{args.height}p frames/sec, detection stage (frames resident in HBM -> YOLOv5s -> NMS -> label rows)",
        "value": round(n / dt, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": det.compute_dtype,
        "data": "synthetic (seeded random-init weights, synthetic frames)",
        "config": {"workload": f"detect: {n} x {args.height}x{args.width} frames per step, YOLOv5s v7.0 at 384 x 640, 6 classes",
                   "layers": det.n_layers, "rows_per_image": det.rows},
        "gflop_per_image_executed": round(det.flops_per_image / 1e9, 3),
        "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / PEAK_FP32_MATRIX_TFLOPS, 4), "traffic": None,
                     "note": "whole stage (letterbox, stem, 58 GEMM convolutions, pools, decode, NMS) against the fp32 matrix peak"},
    }), flush=True)
    det.close()
    eng.close()


def bench_chain(args):
    """The chain north_star names as its own line (the default run reports it as the side measurement `chain_inclusive`):
    Motion-JPEG bytes in pinned host memory -> decode -> YOLOv5s -> NMS -> repair -> detector crops -> CNN -> labels."""
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    n = args.frames
    sd = synth.make_state_dict(seed=1234)
    dt_ = "emulated_f32" if args.dtype == "emulated_f32" else "f32"
    eng = Engine(sd, device=str(device), max_batch_frames=n, max_clip_frames=max(n, 64), max_frame_height=args.height, max_frame_width=args.width,
                 compute_dtype=dt_)
    if args.camera_like:   # three bits of noise per sample instead of five: the compressed size of camera / game footage
        frames = synth.make_frames_torch(n, args.height, args.width, first_frame=0, device=device, noise_mask=7, fine_mask=7)
    else:
        frames = torch.from_numpy(synth.make_frames(n, args.height, args.width)).to(device)
    boxes = torch.from_numpy(synth.make_boxes(n, args.height, args.width)).to(device)
    r = chain_inclusive(eng, sd, frames, boxes, steps=max(args.steps, 3), quality=args.jpeg_quality, compute_dtype=dt_)
    print(json.dumps({
        "metric": f"{args.height}p frames/sec, Motion-JPEG bytes -> decode -> detector -> NMS -> repair -> crops -> CNN -> labels (all on the device)",
        "value": r["value"], "unit": "frames/s", "n_gpus": 1, "steps": max(args.steps, 3), "warmup": 3, "ms_per_step": r["ms_per_clip"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt_, "data": "synthetic",
        "config": {"workload": f"chain: {n} x {args.height}x{args.width} frames per clip, quality-{args.jpeg_quality} 4:2:0 Motion-JPEG, YOLOv5s at 384 x 640, "
                               "ResNet-18 action CNN; three clips in flight"},
        "chain": r}), flush=True)
    eng.close()


def bench_mixed(args):
    """BASELINE.json configs[4]: a 1080p/720p interleaved clip, frames resident in HBM per resolution bucket, fixed-size
    batches through hipGraph-captured "crop + backbone + scatter into the feature cache" sequences, one head pass per
    clip. Reports the graph-replay rate with the eager rate of the same launches beside it. Single GPU, own metric."""
    from playaid_core_amd.stream_runner import MixedResolutionRunner

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    n, batch = 192, args.mixed_batch  # 128 frames at 1080p + 64 at 720p: with 32-frame batches 4 + 2 graph replays per clip
    res = [(1080, 1920) if (i % 3) != 1 else (720, 1280) for i in range(n)]
    sd = synth.make_state_dict(seed=1234)
    eng = Engine(sd, device=str(device), max_batch_frames=batch, max_clip_frames=n, compute_dtype=args.dtype)
    buckets = {}
    for shape in sorted(set(res)):
        idx = [i for i in range(n) if res[i] == shape]
        assert len(idx) % batch == 0
        fr = torch.stack([torch.from_numpy(synth.make_frame(i, *shape)) for i in idx]).to(device)
        bx = torch.from_numpy(np.stack([[synth.fighter_box(i, p, *shape) for p in range(2)] for i in idx]).astype(np.float64)).to(device)
        buckets[shape] = (fr, bx, torch.tensor(idx, dtype=torch.int32, device=device))
    rec, lp = eng.alloc_records(n - 1), eng.alloc_logp(n - 1)
    out = {}
    for mode, use_graphs in (("graph", True), ("eager", False)):
        runner = MixedResolutionRunner(eng, batch_frames=batch, use_graphs=use_graphs)
        for _ in range(max(args.warmup, 2)):  # two passes: both staging slots of every bucket get captured
            runner.run_resident(buckets, n, rec, lp)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            runner.run_resident(buckets, n, rec, lp)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        assert torch.isfinite(lp).all()
        out[mode] = {"frames_per_s": round(n * args.steps / dt, 1), "ms_per_clip": round(1e3 * dt / args.steps, 3),
                     "graph_captures": runner.captures, "graph_replays": runner.replays}
    eng.check_device_errors()
    result = {
        "metric": "mixed-resolution frames/sec (1080p + 720p interleaved), frames resident in HBM -> per-frame labels",
        "value": out["graph"]["frames_per_s"], "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 2),
        "ms_per_step": out["graph"]["ms_per_clip"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"configs[4]: {n}-frame clip, 2/3 at 1920x1080 and 1/3 at 1280x720 interleaved, bucketed by resolution into "
                   f"batches of {batch} frames, each batch one hipGraph replay (crop stage + backbone + "
                   "scatter into the feature cache), one temporal-head pass per clip; seeded weights",
                   "ingest": "frames resident in HBM per bucket; no decode"},
        "hipgraph": out["graph"], "eager": out["eager"],
    }
    print(json.dumps(result), flush=True)
    eng.close()


def clip_batch_side(eng, n_clip, kb, height, width, S, delta, lanes_n, clips=160):
    """Side measurement (not `value`): the same 64-frame clips, `kb` of them concatenated per backbone batch
    (pa_clip_begin_batch: every window stays inside its own clip; results equal the per-clip ones to fp32 rounding,
    tests/test_gpu_contract.py::test_clip_batch_equals_separate_clips), two lanes as in the headline. More crops per
    launch amortise the ~8 us a convolution launch costs outside its steady state."""
    big = eng.clone(max_batch_frames=n_clip * kb, max_clip_frames=max(n_clip * kb, 64))
    lanes = None
    try:
        n = n_clip * kb
        frames = torch.from_numpy(synth.make_frames(n, height, width)).to(eng.device)
        boxes = torch.from_numpy(synth.make_boxes(n, height, width)).to(eng.device)
        lanes = ClipLanes(big, S, delta, lanes=lanes_n)
        lanes.calibrate(frames, boxes, n, batch_of=kb)
        calls = max(clips // kb, 4)
        for _ in range(4):
            lanes.submit(frames, boxes, n, kb)
        torch.cuda.synchronize(eng.device)
        lanes.idle()
        t0 = time.perf_counter()
        for _ in range(calls):
            lanes.submit(frames, boxes, n, kb)
        torch.cuda.synchronize(eng.device)
        dt = time.perf_counter() - t0
        # kernel pass on one stream
        runner = FrameParallelClip(big, S, delta)
        big.profile_enable(True)
        for _ in range(6):
            runner.run(frames, boxes, n, gather=False, pipeline=False, reuse_buffers=True, batch_of=kb)
        torch.cuda.synchronize(eng.device)
        big.profile_enable(False)
        stats = big.profile_read()
        dom = max(stats, key=lambda s: s["total_ms"])
        tf = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        return {
            "clips_per_backbone_batch": kb,
            "crops_per_launch": n * 2,
            "lanes": lanes_n,
            "value": round(n * calls / dt, 2),
            "unit": "frames/s",
            "ms_per_clip": round(1000.0 * dt / (calls * kb), 4),
            "roofline_frac": round(tf / PEAK_FP32_MATRIX_TFLOPS, 4) if big.compute_dtype == "f32" else None,
            "dominant_family_tflops": round(tf, 2),
            "dominant_family_executed_tflops": round(dom.get("flops_executed", dom["flops"]) / (dom["total_ms"] * 1e-3) / 1e12, 2),
            "roofline_frac_is": "algorithmic (direct-form) FLOP/s over the fp32 matrix peak; the Winograd launches execute 4/9 of theirs",
            "avg_launch_ms": round(dom["total_ms"] / max(dom["launches"], 1), 5),
            "note": "side measurement, NOT `value`: the headline keeps the batch size BASELINE.json names (one clip per backbone pass)",
        }
    finally:
        if lanes is not None:
            lanes.close()
        big.close()


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(argv, gpus, run=None):
    """`python bench.py --gpus N` without torch.distributed.run around it: start the N ranks as a CHILD process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), wait, and return its exit
    code. Called before this process has made any GPU call -- it never execs, and it never touches the device itself; rank 0 of
    the child prints the one JSON line on the inherited stdout."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return (run or subprocess.call)(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="frames per backbone batch (and per clip in the configs[1]/[2] workloads)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--clip-frames", type=int, default=None,
                    help="configs[3]: ONE clip of this many frames sharded over the ranks (strong scaling); default 8192 when "
                    "--gpus > 1, off (configs[1]: a 64-frame clip per step) when --gpus 1; 0 = a --frames clip per rank (weak scaling)")
    ap.add_argument("--inner-repeat", type=int, default=20,
                    help="configs[1]/[2] only: clips per timed step (ms_per_step stays per clip)")
    ap.add_argument("--f4-windows", type=int, default=64, help="--workload rnn | resformer: windows per call")
    ap.add_argument("--workload", default="clip", choices=["clip", "mixed", "rnn", "resformer", "detect", "chain"],
                    help="clip = the headline / configs[1-3] workloads; mixed = BASELINE.json configs[4] (mixed-resolution stream, "
                    "bucketing + hipGraph replay; single GPU, reported under its own metric)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="configs[1]/[2] at N = 1: independent clips alternate over this many engines / streams (ClipLanes); "
                    "1 = one engine with the crop stage of clip k+1 under the backbone of clip k")
    ap.add_argument("--clips-per-batch", type=int, default=1,
                    help="configs[1]/[2] at N = 1: this many independent clips go through the backbone as ONE batch "
                    "(pa_clip_begin_batch: windows stay inside their own clip, results equal the per-clip ones) -- more crops per launch")
    ap.add_argument("--mixed-batch", type=int, default=64, help="--workload mixed: frames per resolution bucket batch (divides 64)")
    ap.add_argument("--clip-batch-side", type=int, default=4,
                    help="side measurement `clip_batches` of the default run: this many clips per backbone batch (0 / 1: skip; also skipped "
                    "with --no-pcie)")
    ap.add_argument("--long-clip-batch", type=int, default=256,
                    help="configs[3]: frames per backbone batch of the long clip (the 64-frame batch is configs[1]'s workload, "
                    "not a property of the 8192-frame clip)")
    ap.add_argument("--cpu-sample-frames", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive side measurement")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode-inclusive (Motion-JPEG) side measurement")
    ap.add_argument("--jpeg-quality", type=int, default=95, help="decode_inclusive: quality of the synthetic Motion-JPEG clip (OpenCV's default)")
    ap.add_argument("--jpeg-restart-blocks", type=int, default=0, help="decode_inclusive: restart interval in MCUs (0 = none, like OpenCV / FFmpeg writers)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse ranks > GPUs)")
    ap.add_argument("--no-pipeline", action="store_true", help="crop stage and backbone on one stream (no overlap across steps)")
    ap.add_argument("--camera-like", action="store_true", help="--workload chain: the clip with three bits of noise per sample (0.3-0.6 MB per 1080p frame at quality 95)")
    ap.add_argument("--no-emulated", action="store_true", help="skip the emulated-fp32 side block (compute_dtype emulated_f32)")
    ap.add_argument("--calibrate", action="store_true",
                    help="time every pair of candidate streams before the timed region and keep the fastest (round 5's default; round 6 measured "
                    "no difference between the pairs once warm: profiles/r06_hw_queues.txt)")
    ap.add_argument("--no-calibrate", action="store_true", help="(default since round 6; kept so that scripts/hw_queues.sh's command lines still parse)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "emulated_f32"],
                    help="f32 = the headline (reference arithmetic); bf16 = BASELINE.json configs[2]'s conv path, reported under its own dtype, never as the headline")
    args = ap.parse_args()

    if args.workload == "mixed":
        return bench_mixed(args)
    if args.workload in ("rnn", "resformer"):
        return bench_f4(args)
    if args.workload == "detect":
        return bench_detect(args)
    if args.workload == "chain":
        return bench_chain(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(sys.argv[1:], args.gpus))   # (nothing above has touched the GPU)
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} inside a torch.distributed.run of {world} rank(s): the two must agree")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)  # (== local_rank on a real N-GPU node)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:  # rehearsal of the multi-rank path on a box with fewer GPUs than ranks
            dist.init_process_group(args.backend)

    F, S, A, DELTA = 2, 7, 63, 3
    clip_frames = args.clip_frames if args.clip_frames is not None else (8192 if world > 1 else 0)
    long_clip = clip_frames > 0
    kb = max(args.clips_per_batch, 1) if (world == 1 and not long_clip) else 1   # independent clips per backbone batch
    n_clip = args.frames                                                           # frames of one clip (short-clip workloads)
    n_batch = args.long_clip_batch if long_clip else n_clip * kb
    n_total = clip_frames if long_clip else n_batch * world
    lo, hi = shard_range(n_total, world, rank)
    repeat = 1 if (long_clip or world > 1) else max(args.inner_repeat // kb, 1)   # step() calls per timed step

    # weights: rank 0 folds them once; the prepared device arena crosses xGMI in one RCCL broadcast
    sd = synth.make_state_dict(seed=1234) if rank == 0 else None

    def make_engine(w):
        return Engine(w, device=str(device), max_batch_frames=n_batch, max_clip_frames=max(n_total, 64),
                      max_frame_height=args.height, max_frame_width=args.width, compute_dtype=args.dtype)

    eng = broadcast_engine(make_engine, sd, device) if world > 1 else make_engine(sd)
    # this rank's shard of the synthetic clip, generated on the device it will be read from (bit-identical to
    # synth.make_frames) and resident in HBM before timing
    if long_clip or os.environ.get("PA_BENCH_DEVICE_SYNTH") == "1":
        frames = synth.make_frames_torch(hi - lo, args.height, args.width, first_frame=lo, device=device)
    else:  # short clips: the host generator (rocprofv3 counter passes crash in the torch integer kernels of the device one)
        frames = torch.from_numpy(synth.make_frames(hi - lo, args.height, args.width, first_frame=lo)).to(device)
    boxes = torch.from_numpy(synth.make_boxes(hi - lo, args.height, args.width, first_frame=lo)).to(device)
    runner = FrameParallelClip(eng, S, DELTA)
    lanes = None
    if world == 1 and not long_clip and args.lanes > 1 and not args.no_pipeline:
        lanes = ClipLanes(eng, S, DELTA, lanes=args.lanes)
        if args.calibrate and not args.no_calibrate:
            lanes.calibrate(frames, boxes, n_total, batch_of=kb if kb > 1 else 0)   # untimed: which of the concurrent streams overlap best on this shape
    batch_of = kb if kb > 1 else 0

    def step(pipeline=None):
        if lanes is not None and pipeline is None:
            return lanes.submit(frames, boxes, n_total, batch_of)[1:]
        pipeline = (not args.no_pipeline) if pipeline is None else pipeline
        return runner.run(frames, boxes, n_total, gather=True, pipeline=pipeline, reuse_buffers=True, batch_of=batch_of)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        if lanes is not None:
            lanes.idle()   # the device has drained: the next clips of the lanes start aligned (ClipLanes._aligned_start)

    for _ in range(args.warmup * repeat):
        step()
    fence()
    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _ in range(repeat):
            rec, lp = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_calls = args.steps * repeat       # step() calls in the timed region
    n_clips = n_calls * kb              # clips they processed
    # ---- kernel pass: clips again with every launch bracketed by HIP events on the launch stream
    # (pa_profile_enable). Kept out of the timed region because the event pairs serialise
    # neighbouring kernels and cost ~10 % throughput; its own wall time is reported as
    # profiled_ms_per_step.
    stats, dt_prof, prof_clips = [], None, 0
    if not args.no_profile:
        prof_clips = args.steps if long_clip else min(n_calls, 40)   # (step() calls of the kernel pass)
        if long_clip:
            prof_clips = min(prof_clips, 2)
        eng.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(prof_clips):
            step(pipeline=False)  # one stream: kernel durations free of cross-stream overlap
        fence()
        dt_prof = time.perf_counter() - t1
        eng.profile_enable(False)
        stats = eng.profile_read()

    if rank == 0:
        # sanity: results are finite and complete
        assert rec.shape[0] == n_total - 1 and torch.isfinite(lp).all()
        fps = n_total * n_calls / dt
        shape = (n_clip, args.height, args.width, args.dtype)
        if long_clip:
            cfg_name = "configs[3]" if (clip_frames, args.height, args.width, args.dtype) == (8192, 1080, 1920, "f32") else "custom long clip"
            what = (f"ONE {clip_frames}-frame {args.height}x{args.width} BGR clip sharded frame-parallel over {world} rank(s) "
                    f"({hi - lo} frames on rank 0, backbone batches of {n_batch})")
        else:
            cfg_name = {(64, 1080, 1920, "f32"): "configs[1]", (256, 720, 1280, "bf16"): "configs[2]"}.get(shape, "custom shape")
            what = f"{n_clip} x {args.height}x{args.width} BGR frames per GPU per clip"
        reach = DELTA * (S // 2) ** 2
        result = {
            "metric": f"{args.height}p frames/sec end-to-end (decode->labels; 'decode' here = ingest of raw BGR frames already resident in HBM)",
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / n_clips, 4),
            "ms_per_timed_step": round(1000.0 * dt / args.steps, 4),
            "profiled_ms_per_step": round(1000.0 * dt_prof / prof_clips, 4) if dt_prof else None,
            "higher_is_better": True,
            "scaling": "strong" if long_clip else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{cfg_name}: {what}, 2 fighters/frame, S=7 delta=3 window, "
                f"{'fp32' if args.dtype == 'f32' else 'bf16-conv (3x3 stack in bf16, fp32 accumulate; stem, fc, head fp32)'} "
                f"CNNActionDetector (ResNet-18 + Conv1d/MLP head, 63 actions), seeded weights",
                "ingest": "frames resident in HBM before the timed region; no decode (BASELINE metric's 'decode' = ingest of raw BGR frames, SURVEY.md 8a1)",
                "clip_frames": n_total if long_clip else n_clip,
                "clips_per_backbone_batch": kb,
                "clips_per_backbone_batch_note": ("independent clips concatenated for the backbone launches; every window is clamped to its own "
                                                  "clip (pa_clip_begin_batch), results equal the per-clip ones to fp32 rounding "
                                                  "(tests/test_gpu_contract.py::test_clip_batch_equals_separate_clips)") if kb > 1 else None,
                "frames_per_backbone_batch": n_batch,
                "crops_per_backbone_batch": n_batch * F,
                "inner_repeat": repeat * kb,
                "inner_repeat_note": "clips per timed step; ms_per_step = timed region / (steps x inner_repeat) = one clip",
                "parallelism": f"frame-parallel x{world}" if world > 1 else "single GPU",
                "pipeline": (f"{args.lanes} lanes: independent clips alternate over {args.lanes} engines (own activation buffers, own HIP stream), "
                             "so one clip's launch ramps / kernel tails run under another clip's steady state; inside a lane the crop stage "
                             "and the backbone run in stream order") if lanes is not None
                else ("crop stage of batch k+1 overlaps the backbone of batch k (2 streams, 2 input slots)" if not args.no_pipeline else "none"),
                "lanes": args.lanes if lanes is not None else 1,
                "lane_stream_calibration": lanes.calibration if lanes is not None else None,
                "lane_start": ("after every device synchronisation the lanes' first clips are held behind ONE gate -- a one-thread kernel polling "
                               "a word of pinned memory that the host opens once every lane has its first clip enqueued -- and then start "
                               "together (inside the timed region: once, at its start)") if lanes is not None else None,
            },
        }
        if world > 1:
            recvs, sends = halo_plan(n_total, world, 0, reach)
            result["config"]["exchange"] = {
                "halo_rows_sent_rank0": sum(c for _, _, c in sends) * F,
                "halo_bytes_per_edge": reach * F * 4096,
                "gather_bytes_per_rank": max(hi - lo, 1) * F * (4 + A) * 4,
                "weights_broadcast_bytes_once": int(eng._lib.pa_weights_arena_bytes(eng._h)),
            }
        # whole-path fractions per SURVEY 8d: algorithmic bytes / FLOPs per frame (feature-cached
        # formulation, F frames per batch) x measured frames/s against the two chip roofs
        if (args.height, args.width) == (1080, 1920) and args.dtype == "f32":
            bytes_frame = 6220800 + 2 * (49152 * 2) + 2 * 8388608 + 61391260 / n_batch
            flops_frame = 2 * (1185390592 + 2 * 3584000 + 2 * 73600)
            per_gpu_fps = fps / world
            result["roofline_path"] = {
                "bytes_per_frame": round(bytes_frame),
                "flops_per_frame": flops_frame,
                "hbm_frac": round(bytes_frame * per_gpu_fps / (PEAK_HBM_GBS * 1e9), 4),
                "fp32_matrix_frac": round(flops_frame * per_gpu_fps / (PEAK_FP32_MATRIX_TFLOPS * 1e12), 4),
                "binding_roof": "fp32 MFMA (compute floor 15.2 us/frame vs HBM floor 3.0 us/frame)",
            }
        if stats:
            by = {s["name"]: s for s in stats}
            dom = max(stats, key=lambda s: s["total_ms"])
            tf = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12 if dom["total_ms"] > 0 else 0.0
            traffic, traffic_src = _pmc_traffic(dom["name"], args.dtype, n_batch, args.height, args.width)
            if args.dtype == "f32":
                tf_exec = dom.get("flops_executed", dom["flops"]) / (dom["total_ms"] * 1e-3) / 1e12 if dom["total_ms"] > 0 else 0.0
                result["roofline"] = {
                    "kernel": dom["name"],
                    "bound": "mfma",
                    "achieved": round(tf_exec, 3),
                    "peak": PEAK_FP32_MATRIX_TFLOPS,
                    "unit": "TFLOP/s",
                    "frac": round(tf_exec / PEAK_FP32_MATRIX_TFLOPS, 4),
                    "achieved_is": "EXECUTED FLOP/s of the family (pa_kernel_stat.flops_executed over its HIP-event time): what the matrix cores ran, a "
                                   "fraction of the chip's peak by construction. Thirteen of its launches run as Winograd F(2x2, 3x3) (csrc/wino.hip) and "
                                   "execute 4/9 of their direct-form multiply-adds; the direct-form count SURVEY.md 8d defines (2 x outputs x 9 x cin per 3x3 "
                                   "convolution) over the same time is `algorithmic_tflops` / `algorithmic_frac`, an effective rate that may pass the peak",
                    "algorithmic_tflops": round(tf, 3),
                    "algorithmic_frac": round(tf / PEAK_FP32_MATRIX_TFLOPS, 4),
                    "executed_tflops": round(tf_exec, 3),
                    "executed_frac": round(tf_exec / PEAK_FP32_MATRIX_TFLOPS, 4),
                    "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / max(dom["launches"], 1)),
                    "launches": dom["launches"],
                    "avg_launch_ms": round(dom["total_ms"] / max(dom["launches"], 1), 5),
                }
            else:
                # bf16 conv path: 141 FLOP/B (fp32 figure) becomes ~280 FLOP/B of bf16 traffic, below the
                # bf16 ridge (2.5 PFLOP/s / 8 TB/s ~ 315): HBM is the roof
                gbs = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9 if dom["total_ms"] > 0 else 0.0
                result["roofline"] = {
                    "kernel": dom["name"],
                    "bound": "hbm",
                    "achieved": round(gbs, 1),
                    "peak": PEAK_HBM_GBS,
                    "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / max(dom["launches"], 1)),
                    "launches": dom["launches"],
                    "avg_launch_ms": round(dom["total_ms"] / max(dom["launches"], 1), 5),
                    "tflops": round(tf, 2),
                }
            total_ms = sum(s["total_ms"] for s in stats)
            batches = prof_clips * max((hi - lo + n_batch - 1) // n_batch, 1)
            result["kernels"] = {
                s["name"]: {
                    "launches_per_batch": round(s["launches"] / batches, 3),
                    "ms_per_batch": round(s["total_ms"] / batches, 4),
                    "share": round(s["total_ms"] / total_ms, 4),
                    "tflops": round(s["flops"] / (s["total_ms"] * 1e-3) / 1e12, 2) if s["flops"] and s["total_ms"] else None,
                    "algo_GBs": round(s["bytes"] / (s["total_ms"] * 1e-3) / 1e9, 1) if s["total_ms"] else None,
                }
                for s in stats
            }
            result["kernels_note"] = ("HIP-event times of a separate pass on ONE stream (profiled_ms_per_step is that pass's wall time per clip); "
                                      "the timed region runs two lanes, whose kernels overlap across clips, so the families' sum may exceed ms_per_step"
                                      if lanes is not None else "HIP-event times of a separate pass on one stream")
            pre = by.get("preprocess_crops")
            if pre and pre["total_ms"] > 0:
                gbs = pre["bytes"] / (pre["total_ms"] * 1e-3) / 1e9
                result["roofline_preprocess"] = {
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "note": "the stage's compulsory bytes against HBM, as the contract asks; by the counters (profiles/r03_crop_pmc_per_kernel.txt, "
                            "committed rocprofv3 PMC pass at configs[1]) the fused crop kernel is bound by vector issue -- 0.64 of the VALU slots "
                            "over the whole launch -- and moves 0.86 TB/s",
                }
        if world == 1 and not long_clip and not args.no_pcie:
            result["pcie_inclusive"] = pcie_inclusive(eng, frames[:n_clip], boxes[:n_clip])
            result["pcie_inclusive_windows"] = pcie_inclusive_windows(eng, frames[:n_clip], boxes[:n_clip])
            if not args.no_decode:
                try:
                    result["decode_inclusive"] = decode_inclusive(eng, frames[:n_clip], boxes[:n_clip], quality=args.jpeg_quality,
                                                                  restart_blocks=args.jpeg_restart_blocks)
                except Exception as exc:  # a side measurement must never cost the line its headline
                    result["decode_inclusive"] = {"error": f"{type(exc).__name__}: {exc}"}
                try:
                    result["chain_inclusive"] = chain_inclusive(eng, sd, frames[:n_clip], boxes[:n_clip], quality=args.jpeg_quality)
                except Exception as exc:
                    result["chain_inclusive"] = {"error": f"{type(exc).__name__}: {exc}"}
                try:  # the same clip with three bits of noise: the compressed size of real 1080p footage at quality 95
                    quiet = synth.make_frames_torch(n_clip, args.height, args.width, first_frame=lo, device=device, noise_mask=7, fine_mask=7)
                    result["decode_inclusive_camera_like"] = decode_inclusive(
                        eng, quiet, boxes[:n_clip], quality=args.jpeg_quality, restart_blocks=args.jpeg_restart_blocks,
                        content="the headline's clip with three bits of noise per sample instead of five (0.3-0.6 MB per 1080p frame at "
                                "quality 95, the range of camera / game footage)")
                except Exception as exc:
                    quiet = None
                    result["decode_inclusive_camera_like"] = {"error": f"{type(exc).__name__}: {exc}"}
                try:  # the chain on the content it will meet (decode alone is ~2.3x faster there than on white noise)
                    if quiet is not None:
                        result["chain_inclusive_camera_like"] = chain_inclusive(eng, sd, quiet, boxes[:n_clip], quality=args.jpeg_quality)
                        result["chain_inclusive_camera_like"]["content"] = result["decode_inclusive_camera_like"].get("content")
                except Exception as exc:
                    result["chain_inclusive_camera_like"] = {"error": f"{type(exc).__name__}: {exc}"}
                quiet = None
        if world == 1 and not args.no_cpu_baseline:   # (before the emulated block: its sample is that block's oracle check)
            result["cpu_baseline"] = cpu_baseline(sd, args.height, args.width, args.cpu_sample_frames)
        if world == 1 and not long_clip and args.dtype == "f32" and not args.no_pcie and not args.no_decode and not args.no_emulated:
            try:
                result["emulated_fp32"] = emulated_fp32_side(sd, frames[:n_clip], boxes[:n_clip], n_clip, args.height, args.width, S, DELTA, device,
                                                             args.jpeg_quality)
            except Exception as exc:  # a side measurement must never cost the line its headline
                result["emulated_fp32"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not long_clip and kb == 1 and not args.no_pcie and not args.no_pipeline and args.clip_batch_side > 1:
            try:
                result["clip_batches"] = clip_batch_side(eng, n_clip, args.clip_batch_side, args.height, args.width, S, DELTA, max(args.lanes, 1))
            except Exception as exc:  # a side measurement must never cost the line its headline
                result["clip_batches"] = {"error": f"{type(exc).__name__}: {exc}"}
        emit(result)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if lanes is not None:
        lanes.close()
    eng.close()


if __name__ == "__main__":
    main()
