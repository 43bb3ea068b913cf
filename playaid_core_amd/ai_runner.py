"""``AIRunner`` -- the runner boundary (b2 in SURVEY.md section 8b).

Mirror of the hot loop of ``playaid/ai_runner.py:136-608``:
``action_recognition(frame_num, fighter)`` (``:466-491``),
``run_action_recognition(overwrite=False)`` (``:493-520``), ``write_output()``
(``:606-608``) and ``load_ai_output()`` (``:592-604``), with the same names,
argument meaning, 1-indexed frame numbers and result layout.

What differs, and why:

* Input. The reference opens a video with OpenCV and shells out to a YOLOv5
  checkout (``:153,191-224``). ``input_video_path`` here names a Motion-JPEG
  video (``.avi`` / ``.mjpeg`` / a directory of ``.jpg`` frames), which is decoded on
  the device (``video.py``, ``pa_mjpeg_decode``; the detector's label files are
  read from ``<output_dir>/labels/<video>_<n>.txt`` where ``detect.py --save-txt
  --save-conf`` leaves them), or a clip archive (``.npz`` with ``frames``
  uint8[N,H,W,3] BGR and ``labels``: one YOLO text block per frame, lines
  ``"cls cx cy w h conf"``), or is a ``ClipSource`` already in memory. Other
  codecs (H.264 ...) need a decoder this stack does not have.
* Work. Crops, backbone, head and argmax run on the MI355X for the whole clip
  at once (each crop through ResNet-18 once, not once per window);
  ``action_recognition`` then serves single (frame, fighter) queries from
  those results. Numerically this equals the reference's per-call path to
  fp32 rounding (eval-mode BatchNorm makes crops independent).
* ``clean_yolo_crops`` (``:226-424``) runs on the label text in memory
  (``label_cleaning.py``): duplicate resolution, gap interpolation and tail
  duplication decide which box and which decoded frame every crop is cut from;
  the crop pixels are produced by the HIP crop stage (``square_crop`` geometry,
  ``:417-418``) for every frame, not only for the repaired ones.
* ``run_damage_detection`` (PaddleOCR) is out of scope.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch
import yaml

from . import constants
from .anim_ontology import MOVE_TO_CLASS_ID
from .cnn_action_detector import CNNActionDetector
from .dataset_utils import action_sample_from_frame_middle_out
from .fighter import YoloCrop
from .label_cleaning import clean_yolo_labels


def read_fighter_yolo_crop_text(label_text: str, fighter: str, where: str = "<memory>") -> Optional[YoloCrop]:
    """``read_fighter_yolo_crop`` (``ai_runner.py:53-71``) on the text of one label file."""
    for line in label_text.splitlines():
        if not line:
            continue
        assert len(line.split(" ")) == 6, f"Too much data for line: {line} in label {where}"
        class_id, center_x, center_y, width, height, confidence = line.split(" ")
        if int(class_id) == constants.CHAR_LIST.index(fighter):
            return YoloCrop(
                float(center_x), float(center_y), float(width), float(height),
                confidence=float(confidence), class_id=int(class_id),
            )
    return None


class ClipSource:
    """Decoded frames + per-frame YOLO label text (stand-in for VideoCapture +
    the ``labels/<video>_<n>.txt`` files, 1-indexed like them)."""

    def __init__(self, frames: np.ndarray, labels: List[str], name: str = "clip", crop_images=None):
        """``crop_images`` (optional): ``crop_images[i][p]`` = the BGR crop image YOLOv5 ``--save-crop`` wrote for
        fighter slot p (sorted class ids) of frame i (``crops/<Fighter>/<video>_<i+1>.jpg`` after
        ``cv2.imread``), any size, or None where the detector saved none. With them the runner takes the
        reference's own input branch (``ai_runner.py:446-459``) from the images instead of cutting crops from
        ``frames``; ``frames`` may then be an empty ``uint8[n, 0, 0, 3]`` array."""
        # frames: uint8[n, H, W, 3] BGR, a numpy array or a tensor already in HBM (a decoded video stays where it was decoded)
        assert frames.ndim == 4 and frames.shape[3] == 3 and str(frames.dtype).endswith("uint8")
        assert len(labels) == frames.shape[0]
        assert crop_images is None or len(crop_images) == len(labels)
        self.frames = frames
        self.labels = list(labels)
        self.name = name
        self.crop_images = crop_images

    VIDEO_EXTENSIONS = (".avi", ".mjpeg", ".mjpg")

    @classmethod
    def load(cls, path: str, labels_dir: str = None) -> "ClipSource":
        """``.npz`` clip archive, or a video the device can decode (Motion-JPEG ``.avi`` / ``.mjpeg`` or a directory of
        ``.jpg`` frames, ``playaid_core_amd/video.py``) plus the detector's label files."""
        if os.path.isdir(path) or path.lower().endswith(cls.VIDEO_EXTENSIONS):
            return cls.from_video(path, labels_dir)
        z = np.load(path, allow_pickle=False)
        name = os.path.splitext(os.path.basename(path))[0]
        return cls(z["frames"], [str(s) for s in z["labels"]], name)

    @classmethod
    def from_video(cls, path: str, labels_dir: str = None, labels: List[str] = None, batch_frames: int = 64) -> "ClipSource":
        """What the reference does with ``cv2.VideoCapture(input_video_path)`` + the YOLOv5 output directory
        (``ai_runner.py:153,156-159``): every frame decoded ON THE DEVICE (``video.VideoCapture.read_frames``; the
        frames stay in HBM), the label text of frame n (1-indexed) read from ``<labels_dir>/<video>_<n>.txt``
        (default ``<AI_CACHE>/<video>/labels``, where ``detect.py --save-txt`` puts it; a frame without detections
        has no file, ``:255-262``)."""
        import torch

        from . import video

        name = os.path.splitext(os.path.basename(os.path.normpath(path)))[0]
        cap = video.VideoCapture(path, batch_frames=batch_frames)
        if not cap.isOpened():
            raise FileNotFoundError(f"cannot open {path} as a Motion-JPEG stream")
        n = cap.frame_count()
        frames = torch.empty((n, cap.height, cap.width, 3), dtype=torch.uint8, device=cap._device)
        status = torch.zeros(n, dtype=torch.int32, device=cap._device)
        for j0 in range(0, n, batch_frames):
            cnt = min(batch_frames, n - j0)
            cap.read_frames(j0, cnt, out=frames[j0:j0 + cnt], status=status[j0:j0 + cnt])
        torch.cuda.synchronize()
        bad = torch.nonzero(status).flatten().tolist()
        for j in bad:  # a frame whose decoder states had not settled in the enqueued passes: decode it again, exactly
            cap.read_frames(j, 1, out=frames[j:j + 1], status=status[j:j + 1], exact=True)
        torch.cuda.synchronize()
        still = torch.nonzero(status).flatten().tolist()
        cap.release()
        if still:
            raise ValueError(f"{path}: frame {still[0] + 1} does not decode (status {int(status[still[0]])})")
        if labels is None:
            labels_dir = labels_dir or os.path.join(constants.AI_CACHE, name, "labels")
            labels = []
            for i in range(n):
                fp = os.path.join(labels_dir, f"{name}_{i + 1}.txt")
                labels.append(open(fp).read() if os.path.exists(fp) else "")
        return cls(frames, labels, name)

    def save(self, path: str):
        frames = self.frames if isinstance(self.frames, np.ndarray) else self.frames.cpu().numpy()
        np.savez(path, frames=frames, labels=np.array(self.labels))

    @classmethod
    def synthetic(cls, n: int, height: int, width: int, seed: int = 7, name: str = "synthetic") -> "ClipSource":
        from . import synth

        frames = synth.make_frames(n, height, width, seed)
        boxes = synth.make_boxes(n, height, width)
        labels = []
        for i in range(n):
            lines = [
                str(YoloCrop(*boxes[i, p], confidence=1.0, class_id=synth.FIGHTER_CLASS_IDS[p]))
                for p in range(boxes.shape[1])
            ]
            labels.append("\n".join(lines) + "\n")
        return cls(frames, labels, name)


class _Rec(dict):
    """Tiny stand-in for ``addict.Dict`` (attribute access, auto-vivification)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            return None

    def __setattr__(self, k, v):
        self[k] = v

    def __missing__(self, k):
        v = _Rec()
        self[k] = v
        return v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, _Rec) else v) for k, v in self.items()}


class AIRunner:
    """Runs action recognition end to end (tracking boxes come with the clip)."""

    def __init__(self, input_video_path, debug: bool = False, model: CNNActionDetector = None,
                 checkpoint_path: str = None, output_dir: str = None, crop_jpeg_quality: int = 95, crop_mode: str = "yolo",
                 **dataset_args):
        """``crop_jpeg_quality``: the reference never shows the CNN a crop as cut -- every crop is written as a JPEG
        (YOLOv5 ``--save-crop``, ``cv2.imwrite`` at ``ai_runner.py:420``) and read back (``:446``). 95 (OpenCV's
        default quality, and the default here) makes the crops this runner cuts from frames take the same write +
        read on the device (``pa_set_crop_jpeg_quality``); 0 opts out and feeds the exact resampler output, which
        differs from what the reference's CNN sees by more than 1e-3 on the log-probabilities. Crop IMAGES handed in
        through ``ClipSource.crop_images`` are decoded JPEGs already and are never re-coded.

        ``crop_mode``: "yolo" (default) gives the CNN what the reference's gets -- for every (frame, fighter) the
        detector reported, YOLOv5's own ``--save-crop`` image (``save_one_box``: the box x 1.02 + 10 px, not square, any
        size, written as a 4:4:4 JPEG; ``ai_runner.py:208``) through the runner's resize / letterbox (``:446-459``), and only
        for repaired gap frames the ``square_crop(128, padding=30)`` the reference cuts itself and writes with
        ``cv2.imwrite`` (4:2:0; ``:417-420``); all of it on the device (``pa_save_one_box_crops``,
        ``pa_square_crops``, ``pa_backbone_crop_images``). "square" cuts ``square_crop`` for every frame (one fused crop
        stage, the bench's formulation; what the reference's ground-truth data generator does,
        ``data_gen_scripts/gen_gt_action_detection.py:52-53``)."""
        if crop_mode not in ("yolo", "square"):
            raise ValueError("crop_mode is 'yolo' or 'square'")
        self.crop_mode = crop_mode
        if isinstance(input_video_path, ClipSource):
            self.clip = input_video_path
        else:
            # a video (decoded on the device) finds its detector output where the reference's run_yolo leaves it:
            # <yolo_output_dir>/labels/<video>_<n>.txt (ai_runner.py:156-159, 191-224)
            name = os.path.splitext(os.path.basename(os.path.normpath(input_video_path)))[0]
            ydir = output_dir or os.path.join(constants.AI_CACHE, name)
            self.clip = ClipSource.load(input_video_path, labels_dir=os.path.join(ydir, "labels"))
        self.input_video_path = getattr(input_video_path, "name", input_video_path)
        self.video_name = self.clip.name
        self.dataset_args = dataset_args
        self.debug = debug
        self.yolo_output_dir = output_dir or os.path.join(constants.AI_CACHE, self.video_name)
        self.ai_output_file = os.path.join(self.yolo_output_dir, "ai_output.yaml")
        if model is None:
            path = checkpoint_path or os.path.join(constants.SAVED_ACTION_MODELS, "four-chars-aug-4.ckpt")
            model = CNNActionDetector.load_from_checkpoint(path, actions=list(MOVE_TO_CLASS_ID.keys()))
        self.model = model
        self.model.eval()
        self.crop_jpeg_quality = int(crop_jpeg_quality)
        class_ids = sorted({int(l.split(" ")[0]) for t in self.clip.labels for l in t.splitlines() if l})
        if len(class_ids) != 2:
            # ai_runner.py:240-242 prints and exit()s here
            raise ValueError(f"expected exactly 2 fighters in the labels, found class ids {class_ids}")
        self.fighters = [constants.CHAR_LIST[c] for c in class_ids]
        self._class_ids = class_ids
        # ai_runner.py:226-424 (run from __init__ via run_yolo, :188-189): repair the labels;
        # max_frames is the number of the last label file (:244-245)
        self.cleaned = clean_yolo_labels(self.clip.labels, self.fighters, self.clip.frames.shape[0], self.video_name)
        self.max_frames = self.cleaned.max_frames
        if debug:
            for line in self.cleaned.log:
                print(line)
        res, self.ai_output_data = self.load_ai_output()
        self._results = None

    # -- geometry of the window (ai_runner.py:430-439) ---------------------------
    @property
    def num_frames_per_sample(self) -> int:
        return self.dataset_args.get("num_frames_per_sample", constants.NUM_FRAMES_PER_SAMPLE)

    @property
    def frame_delta(self) -> int:
        fd = self.dataset_args.get("frame_delta", constants.FRAME_DELTA)
        if isinstance(fd, (list, tuple)):
            if len(fd) != 1:
                raise ValueError("the batched path needs one frame_delta (the reference draws random.choice per call)")
            fd = fd[0]
        return int(fd)

    def _boxes(self):
        """-> (boxes float64[max_frames,2,4], src int32[max_frames,2], missing bool[max_frames,2]) from the
        repaired labels. An entry without a crop gets a stand-in box (never reported: the reference
        asserts when a window needs it, see ``_run_clip``)."""
        cl = self.cleaned
        n = self.max_frames
        self._crops = cl.label_crop
        missing = cl.pixel_frame < 0
        boxes = cl.pixel_box.copy()
        src = cl.pixel_frame.copy()
        for p in range(len(self.fighters)):
            ok = np.nonzero(~missing[:, p])[0]
            for i in np.nonzero(missing[:, p])[0]:
                boxes[i, p] = cl.pixel_box[ok[0], p]
                src[i, p] = min(i, self.clip.frames.shape[0] - 1)
        return boxes, src, missing

    def _run_clip(self):
        if self._results is not None:
            return self._results
        eng = self.model.engine
        if eng.S != self.num_frames_per_sample:
            raise ValueError(f"model was trained with sequence_length {eng.S}, runner asked for {self.num_frames_per_sample}")
        if eng.cfg.frame_delta != self.frame_delta or list(eng.cfg.fighter_class_ids)[:2] != self._class_ids:
            hs, ws = [self.clip.frames.shape[1], 128], [self.clip.frames.shape[2], 128]
            for row in self.clip.crop_images or []:
                for im in row:
                    if im is not None:
                        hs.append(im.shape[0])
                        ws.append(im.shape[1])
            eng = eng.reconfigured(frame_delta=self.frame_delta, fighter_class_ids=tuple(self._class_ids),
                                   max_clip_frames=max(self.max_frames, 64), max_frame_height=max(hs), max_frame_width=max(ws))
        boxes, src, missing = self._boxes()
        n = self.max_frames
        if self.clip.crop_images is not None:
            self._results = self._run_clip_from_crop_images(eng, boxes, src, missing)
            return self._results
        for p, fighter in enumerate(self.fighters):
            # every frame in [1, max_frames) is the middle of its own window (ai_runner.py:443-447)
            bad = np.nonzero(missing[: n - 1, p])[0]
            assert len(bad) == 0, f"Failed to get frame crops/{fighter}/{self.video_name}_{bad[0] + 1}.jpg"
        if self.crop_mode == "yolo":
            self._results = self._run_clip_yolo_crops(eng, boxes, src, missing)
            return self._results
        own = np.arange(n, dtype=np.int32)
        eng.set_crop_jpeg_quality(self.crop_jpeg_quality)   # for this clip only: the engine is the model's, others use it
        try:
            if all(np.array_equal(src[:, p], own) for p in range(src.shape[1])):
                out = eng.infer_clip(self.clip.frames[:n], boxes, want_crops=True)
            else:
                # repaired gaps are cut from VideoCapture position j, not j-1 (ai_runner.py:405-406), so a crop
                # may come from another decoded frame than its own, differently per fighter: the crop stage
                # takes a per-crop source index and the clip still runs once
                out = eng.infer_clip(self.clip.frames, boxes, want_crops=True, src=src)
        finally:
            eng.set_crop_jpeg_quality(0)
        st = out["crop_status"].copy()
        st[missing] = 0
        bad = np.argwhere(st != 0)
        assert len(bad) == 0, f"Failed to get square crop from frame {bad[0][0] + 1}"  # ai_runner.py:418
        self._results = out
        return out

    def _run_clip_yolo_crops(self, eng, boxes, src, missing):
        """The reference's own mix of crops, made on the device: ``save_one_box`` images (+ their 4:4:4 JPEG write /
        read) where the detector reported the fighter, ``square_crop(128, 30)`` (+ ``cv2.imwrite``'s 4:2:0 JPEG) where the
        label repair interpolated it, copies where the repair copied files -- then every image through the runner's
        input branch and the backbone in one pass."""
        cl = self.cleaned
        n, F = self.max_frames, len(self.fighters)
        dev = eng.device
        frames = self.clip.frames
        fd = frames if isinstance(frames, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
        q = self.crop_jpeg_quality
        kind = cl.crop_kind.copy()
        kind[missing] = 0
        rows = torch.from_numpy(np.ascontiguousarray(cl.crop_row)).to(dev)
        counts = torch.full((n,), F, dtype=torch.int32, device=dev)
        det_index = np.where(kind == 1, np.arange(F, dtype=np.int32)[None, :], -1).astype(np.int32)
        srcf = np.where(kind == 1, src, 0).astype(np.int32)
        step = eng.max_batch_frames
        parts, descs, base = [], [], 0
        for f0 in range(0, n, step):
            cnt = min(step, n - f0)
            images, desc = eng.save_one_box_crops(fd, rows[f0:f0 + cnt], counts[f0:f0 + cnt], det_index=det_index[f0:f0 + cnt],
                                                  jpeg_quality=q, src_frame=srcf[f0:f0 + cnt])
            used = int((desc[:, 0] + ((desc[:, 1] & 0xFFFFFFFF) * (desc[:, 1] >> 32) * 3 + 15) // 16 * 16).max().item()) if cnt else 0
            desc = desc.clone()
            desc[:, 0] += base
            parts.append(images[:used])
            descs.append(desc)
            base += used
        eng.check_device_errors()
        desc = torch.cat(descs)
        # the repaired gaps: square crops (BGR, as cv2.imwrite gets them) with the 4:2:0 write / read, as 128 x 128 images
        rep = np.argwhere(kind == 2)
        if len(rep):
            eng.set_crop_jpeg_quality(q)
            try:
                sq, st = eng.square_crops(fd[torch.from_numpy(src[rep[:, 0], rep[:, 1]].astype(np.int64)).to(dev)],
                                          np.repeat(boxes[rep[:, 0], rep[:, 1]][:, None, :], F, axis=1), padding=eng.cfg.crop_padding)
            finally:
                eng.set_crop_jpeg_quality(0)
            assert (st[:, 0] == 0).all(), f"Failed to get square crop from frame {rep[np.nonzero(st[:, 0])[0][0], 0] + 1}"  # ai_runner.py:418
            sqd = torch.from_numpy(np.ascontiguousarray(sq[:, 0])).to(dev).reshape(-1)
            parts.append(sqd)
            e = torch.from_numpy((rep[:, 0] * F + rep[:, 1]).astype(np.int64)).to(dev)
            desc[e, 0] = base + torch.arange(len(rep), device=dev, dtype=torch.int64) * (128 * 128 * 3)
            desc[e, 1] = (128 << 32) | 128
        images = torch.cat(parts + [torch.zeros(64, dtype=torch.uint8, device=dev)])
        out = eng.infer_clip_from_packed_crop_images(images, desc, n, want_crops=True)
        st = out["crop_status"].copy()
        st[missing | (kind == 0)] = 0
        bad = np.argwhere(st != 0)
        assert len(bad) == 0, f"Bad shape of crop image for frame {bad[0][0] + 1}"  # ai_runner.py:458
        return out

    def _run_clip_from_crop_images(self, eng, boxes, src, missing):
        """The clip's crops are the detector's saved crop images: every (frame, fighter) goes through the
        runner-input branch on the device (``pa_backbone_crop_images``). A crop that the label repair made up
        by duplicating the last detection (``ai_runner.py:270-289``) re-uses that detection's image, like the
        file copy the reference makes; an interpolated gap (``:389-418``) is re-cut from the video by the
        reference and therefore needs ``frames`` -- not available from crop images alone."""
        n = self.max_frames
        cl = self.cleaned
        images = []
        for i in range(n):
            row = []
            for p, fighter in enumerate(self.fighters):
                j = int(src[i, p])
                if missing[i, p]:
                    assert i == n - 1, f"Failed to get frame crops/{fighter}/{self.video_name}_{i + 1}.jpg"
                    j = int(np.nonzero(~missing[:, p])[0][-1])  # never reported (frame max_frames has no window)
                same = cl.pixel_frame[j, p] == j and np.array_equal(cl.pixel_box[j, p], boxes[i, p])
                img = self.clip.crop_images[j][p] if same or missing[i, p] else None
                if img is None:
                    raise NotImplementedError(
                        f"crop of {fighter} for frame {i + 1} was interpolated by the label repair; the reference re-cuts it "
                        "from the video (ai_runner.py:404-418): give the ClipSource its frames as well")
                row.append(img)
            images.append(row)
        out = eng.infer_clip_from_crop_images(images, want_crops=True)
        st = out["crop_status"].copy()
        st[missing] = 0
        bad = np.argwhere(st != 0)
        assert len(bad) == 0, f"Bad shape of crop image for frame {bad[0][0] + 1}"  # ai_runner.py:458
        return out

    def get_action_recognition_input_for_frame(self, frame: int, fighter: str):
        """``ai_runner.py:426-464``: -> (float32[1,S,3,128,128] /255, list of S uint8[128,128,3] RGB)."""
        res = self._run_clip()
        frame_nums = action_sample_from_frame_middle_out(
            frame, num_frames_per_sample=self.num_frames_per_sample, frame_delta=self.frame_delta,
            max_frames=self.max_frames, min_frame=1,
        )
        p = self.fighters.index(fighter)
        frames = [res["crops_rgb"][f - 1, p] for f in frame_nums]
        input_frames = torch.tensor(np.array(frames)).permute(0, 3, 1, 2).unsqueeze(0).float() / 255.0
        return input_frames, frames

    def action_recognition(self, frame_num: int, fighter: str):
        """``ai_runner.py:466-491``: frame_num is 1-indexed, in [1, max_frames)."""
        if not 1 <= frame_num < self.max_frames:
            raise IndexError(f"frame_num {frame_num} outside [1, {self.max_frames})")
        res = self._run_clip()
        p = self.fighters.index(fighter)
        input_frames, frames = self.get_action_recognition_input_for_frame(frame_num, fighter)
        predicted_action_id = int(res["action_id"][frame_num - 1, p])
        predicted_action = self.model.actions[predicted_action_id]
        confidence = float(res["prob"][frame_num - 1, p]) * 100.0
        crop = self._crops[frame_num - 1][p]
        return (
            input_frames,
            constants.CHAR_LIST.index(fighter),
            torch.tensor(predicted_action_id),
            {
                "char": fighter,
                "predicted_action": predicted_action,
                "confidence": confidence,
                "crop": crop,
                "frames": [np.array(f) for f in frames],
            },
        )

    def run_action_recognition(self, overwrite=False):
        """``ai_runner.py:493-520``."""
        res = self._run_clip()
        for p, fighter in enumerate(self.fighters):
            if not overwrite and self.ai_output_data[fighter][0].action:
                print(f"Already performed action recognition for {fighter}")
                continue
            for frame_num in range(1, self.max_frames):
                predicted_action = self.model.actions[int(res["action_id"][frame_num - 1, p])]
                confidence = float(res["prob"][frame_num - 1, p]) * 100.0
                frame_data = self.ai_output_data[fighter][frame_num - 1]  # YOLO is 1-indexed -> 0-indexed
                frame_data.crop = str(self._crops[frame_num - 1][p])
                frame_data.action = predicted_action
                frame_data.predicted_action_confidence = confidence

    def load_ai_output(self):
        if not os.path.exists(self.ai_output_file):
            return False, _Rec()
        with open(self.ai_output_file, "r") as f:
            try:
                data = yaml.safe_load(f)
            except Exception:
                return False, _Rec()
        out = _Rec()
        for fighter, frames in (data or {}).items():
            for idx, rec in frames.items():
                out[fighter][idx] = _Rec(rec)
        return True, out

    def write_output(self):
        os.makedirs(os.path.dirname(self.ai_output_file), exist_ok=True)
        with open(self.ai_output_file, "w") as f:
            yaml.dump(self.ai_output_data.to_dict(), f)
