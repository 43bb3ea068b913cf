"""``YoloCrop`` -- the bounding-box half of the output contract.

Host-side mirror of ``playaid/fighter.py:158-390``: the scalar methods
(pixel conversion, interpolation, string round trip) are plain Python as in
the reference; ``square_crop`` -- the pixel work -- runs the HIP preprocess
kernels through the engine instead of PIL/cv2.
"""
from __future__ import annotations

import numpy as np


class YoloCrop:
    def __init__(self, center_x, center_y, crop_width, crop_height, confidence=0, class_id=-1):
        """Normalised (0..1) YOLO box, as in ``fighter.py:159-169``."""
        self.center_x = center_x
        self.center_y = center_y
        self.crop_width = crop_width
        self.crop_height = crop_height
        self.confidence = confidence
        self.class_id = class_id

    @classmethod
    def from_pixel_yolo(cls, image_width, image_height, center_x, center_y, width, height):
        return cls(
            center_x / image_width, center_y / image_height, width / image_width, height / image_height
        )

    @classmethod
    def from_string(cls, yolo_string):
        """``"cls cx cy w h conf"`` -> YoloCrop (``fighter.py:204-218``)."""
        class_id, center_x, center_y, width, height, confidence = yolo_string.split(" ")
        return cls(
            float(center_x),
            float(center_y),
            float(width),
            float(height),
            confidence=float(confidence),
            class_id=int(class_id),
        )

    def interp(self, b, percent):
        """Linear interpolation towards ``b`` (``fighter.py:220-231``)."""
        assert self.class_id == b.class_id, "Interpolating between two different class ids"
        lerp = lambda p, q: p + (percent * (q - p))  # noqa: E731
        return YoloCrop(
            lerp(self.center_x, b.center_x),
            lerp(self.center_y, b.center_y),
            lerp(self.crop_width, b.crop_width),
            lerp(self.crop_height, b.crop_height),
            confidence=lerp(self.confidence, b.confidence),
            class_id=self.class_id,
        )

    def yolo_crop(self):
        return (self.center_x, self.center_y, self.crop_width, self.crop_height)

    def xyxy_norm(self):
        return (
            self.center_x - (self.crop_width / 2),
            self.center_y - (self.crop_height / 2),
            self.center_x + (self.crop_width / 2),
            self.center_y + (self.crop_height / 2),
        )

    def xyxy_pixels(self, image_width, image_height):
        (x1, y1, x2, y2) = self.xyxy_norm()
        return (
            max(0, int(x1 * image_width)),
            max(0, int(y1 * image_height)),
            min(image_width, int(x2 * image_width)),
            min(image_height, int(y2 * image_height)),
        )

    def center_pixels(self, image_width, image_height):
        return (int(self.center_x * image_width), int(self.center_y * image_height))

    def yolo_pixels(self, image_width, image_height):
        """Truncating pixel conversion (``fighter.py:305-314``)."""
        return (
            int(self.center_x * image_width),
            int(self.center_y * image_height),
            int(self.crop_width * image_width),
            int(self.crop_height * image_height),
        )

    def crop_img(self, image):
        (x1, y1, x2, y2) = self.xyxy_pixels(image.shape[1], image.shape[0])
        return image[y1:y2, x1:x2]

    def square_crop(self, image, output_size=128, padding=0, engine=None):
        """``(ok, uint8[128,128,3])`` like ``fighter.py:323-381`` -- same
        channel order as ``image`` -- computed by the HIP preprocess kernels.
        ``(False, None)`` for an empty / off-screen slice. ``engine`` defaults
        to the process-wide engine on ``cuda:0``."""
        if output_size != 128:
            raise ValueError("the HIP path produces 128x128 crops only (reference default)")
        from .engine import default_engine

        eng = engine if engine is not None else default_engine()
        frames = np.ascontiguousarray(image)[None]
        boxes = np.array([[[self.center_x, self.center_y, self.crop_width, self.crop_height]]], dtype=np.float64)
        crops, status = eng.square_crops(frames, boxes, padding=padding, swap_rb=False)
        if int(status[0, 0]) != 0:
            return False, None
        return True, crops[0, 0]

    def __str__(self):
        return (
            f"{self.class_id} {self.center_x} {self.center_y} {self.crop_width} "
            + f"{self.crop_height} {self.confidence}"
        )

    def __repr__(self):
        return str(self)
