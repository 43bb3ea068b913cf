"""How many verify passes does the self-synchronising entropy decoder need? For a few encodings of one synthetic frame:
exact mode (pa_mjpeg_set_sync_rounds(h, 0)) at several subsequence sizes; prints rounds, status and parity with the oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import jpeg  # noqa: E402  (checker only)
from playaid_core_amd import synth, video  # noqa: E402

h, w = int(os.environ.get("H", 1080)), int(os.environ.get("W", 1920))
fr = synth.make_frame(5, 1080, 1920)[:h, :w]
variants = {"q95": dict(quality=95), "q95opt": dict(quality=95, optimize=True), "q75": dict(quality=75), "q75opt": dict(quality=75, optimize=True),
            "q30": dict(quality=30), "q95rows1": dict(quality=95, restart_marker_rows=1)}
flat = np.zeros((h, w, 3), np.uint8)
flat[: h // 2] = 40
for name, kw in variants.items():
    for img, tag in ((fr, ""), (flat, "-flat")):
        if tag and name != "q95":
            continue
        blob = synth.encode_jpeg_frames([img], **kw)[0]
        want = jpeg.decode_bgr(blob)
        for sh in (0, 7, 8, 9, 10, 11):
            if sh:
                os.environ["PA_MJPEG_SUB_SHIFT"] = str(sh)
            else:
                os.environ.pop("PA_MJPEG_SUB_SHIFT", None)
            dec = video.MjpegDecoder(1, h, w, len(blob) + 4096)
            data = np.frombuffer(blob, np.uint8)
            res = []
            for rounds in (0, 1, 2, 3, 8):
                dec.set_sync_rounds(rounds)
                st = torch.zeros(1, dtype=torch.int32, device="cuda")
                out = dec.decode(data, np.array([[0, len(blob)]]), h, w, status=st)
                torch.cuda.synchronize()
                ok = bool(np.array_equal(out.cpu().numpy()[0], want))
                res.append(f"r{rounds}:{'ok' if ok else 'BAD'}/st{int(st[0])}" + (f"/ran{dec.last_sync_rounds()}" if rounds == 0 else ""))
            print(f"{name + tag:10s} {len(blob):8d} B  shift {sh:2d}: " + "  ".join(res), flush=True)
            dec.close()
