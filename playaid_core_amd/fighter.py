"""``YoloCrop`` and ``Fighter`` -- the output contract of the path.

Host-side mirror of ``playaid/fighter.py``. ``YoloCrop`` (``:158-390``): the scalar
methods (pixel conversion, interpolation, string round trip) are plain Python as in
the reference; ``square_crop`` -- the pixel work -- runs the HIP preprocess kernels
through the engine instead of PIL/cv2. ``Fighter`` (``:394-612``) is the per-fighter
record the timeline feeds: scalar copy-through of a log / AI dict, the log camera
projection of the fighter's box (O(1) numpy per frame, as in the reference; the batched
form for a whole log is ``Engine.project_boxes``), ``motion_kind`` -> action through
``params_labels.csv``, and the frame-to-frame deltas of ``update``.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import anim_ontology
from .constants import CHAR_LIST

LOG_IMAGE_WIDTH, LOG_IMAGE_HEIGHT = 1280, 720  # the reference's hard-coded projection target (fighter.py:497,528-530)


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
    def from_pixel_coordinates(cls, image_width, image_height, x1, y1, x2, y2, x3, y3, x4, y4):
        """Axis-aligned box around four pixel points, normalised (``fighter.py:170-190``)."""
        xs, ys = (x1, x2, x3, x4), (y1, y2, y3, y4)
        return cls(
            sum(xs) / 4 / image_width,
            sum(ys) / 4 / image_height,
            (max(xs) - min(xs)) / image_width,
            (max(ys) - min(ys)) / image_height,
        )

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


class LogCamera:
    """The game log's camera for one frame: look-at pose + pinhole intrinsics, both exactly as
    ``calculate_lookat_matrix`` / ``calculate_intrinsic_matrix`` / ``project_point_to_pixel``
    (``fighter.py:31-155``) build them, including the general 4x4 ``np.linalg.inv`` of the pose
    (kept so that results round to the same pixels)."""

    def __init__(self, camera_position, target_position, fov_degrees, width=LOG_IMAGE_WIDTH, height=LOG_IMAGE_HEIGHT):
        eye = np.array(camera_position)
        back = eye - np.array(target_position)
        back /= np.linalg.norm(back)
        side = np.cross(np.array([0, 1, 0]), back)
        side /= np.linalg.norm(side)
        self.extrinsics = np.eye(4)
        self.extrinsics[0, :3] = side
        self.extrinsics[1, :3] = np.cross(back, side)
        self.extrinsics[2, :3] = -back
        self.extrinsics[:3, 3] = eye
        focal = width / (2 * np.tan(np.deg2rad(fov_degrees) / 2))
        self.intrinsics = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1]])
        self.height = height

    def pixel(self, point_world) -> np.ndarray:
        cam = np.linalg.inv(self.extrinsics) @ np.append(point_world, 1)
        px = self.intrinsics @ (cam[:3] / cam[2])
        px[1] = self.height - px[1]
        return np.round(px[:2]).astype(int)

    # corners of the fighter's box in world units around (pos_x, pos_y, 0) (fighter.py:507-526)
    BOX_CORNERS = ((-10, 20, 0), (10, 20, 0), (-10, -3, 0), (10, -3, 0))

    def fighter_crop(self, position_in_world) -> "YoloCrop":
        flat = []
        for corner in self.BOX_CORNERS:
            flat.extend(self.pixel(position_in_world + np.array(corner)))
        return YoloCrop.from_pixel_coordinates(LOG_IMAGE_WIDTH, LOG_IMAGE_HEIGHT, *flat)


class Fighter:
    """``playaid.fighter.Fighter``: constructor arguments, ``set_from_json`` and ``update`` as the
    reference (``fighter.py:394-612``). ``motion_kind`` -> ``action_string`` needs
    ``anim_ontology.load_hex_to_action(path)`` to have been called (the reference reads the CSV at
    import); with an empty table every ``action_string`` is ``""`` and the derived action
    ``"Undefined"``, exactly what the reference produces for an unknown hex."""

    def __init__(self, frame_num: int, fighter_name: str = "", char_class_id: int = -1, crop=None,
                 crop_confidence: float = -1.0, yolo_string: str = "", action: str = "", action_confidence: float = 0.0,
                 advantage_state: str = "", fighter_id: int = -1, data: Optional[Dict] = None):
        self.frame_num = frame_num
        self.char_class_id = char_class_id
        self.fighter_name = fighter_name
        self.fighter_id = fighter_id
        self.crop = crop
        self.crop_confidence = crop_confidence
        self.action = action
        self.action_confidence = action_confidence
        self.advantage_state = advantage_state
        self.damage = self.previous_damage = self.damage_delta = 0
        self.new_action = True
        self.num_frames_left = 25200
        self.previous_non_damaged_action = None
        self.frames_since_damaged = self.frames_since_hit = 0
        self.last_frame_in_tech_situation = self.last_frame_in_ledge_situation = -1
        self.hitstun_left = 0
        self.attack_connected = False
        self.status_kind = -1
        self.can_act = True
        self.previous_action = ""
        self.move_counter = 0
        self.raw_animation_frame_num = 0.0
        self.animation_frame_num = 1
        if yolo_string:
            box = YoloCrop.from_string(yolo_string)
            self.char_class_id = box.class_id
            self.fighter_name = CHAR_LIST[box.class_id]
            self.crop = YoloCrop(box.center_x, box.center_y, box.crop_width, box.crop_height)
            self.crop_confidence = box.confidence
        if data:
            self.set_from_json(data)
        assert self.crop, "No crop specified"
        assert self.fighter_name, "No fighter_name specified"

    # log keys copied to attributes of the same name (fighter.py:461-477); a missing one is a KeyError
    _COPIED = ("damage", "facing", "fighter_id", "motion_kind", "num_frames_left", "pos_x", "pos_y", "shield_size",
               "status_kind", "stock_count", "attack_connected")

    def set_from_json(self, data: Dict):
        for key in self._COPIED:
            setattr(self, key, data[key])
        self.position_in_world = [data["pos_x"], data["pos_y"], 0]
        self.can_act = data.get("can_act", True)
        self.raw_animation_frame_num = data.get("animation_frame_num", 0)
        self.stage_id = data["stage_id"] if data["stage_id"] in anim_ontology.STAGE_ENUM_TO_DATA else 0
        stage = anim_ontology.STAGE_ENUM_TO_DATA[self.stage_id]
        self.stage = stage["name"]
        self.fighter_name = anim_ontology.FIGHTER_ENUM_TO_NAME[data["fighter_name"]]
        # the logged camera_fov is ignored in favour of the stage table (fighter.py:484-488)
        camera = LogCamera(list(data["camera_position"].values()), list(data["camera_target_position"].values()), stage["fov"])
        self.extrinsics, self.intrinsics = camera.extrinsics, camera.intrinsics
        self.point_in_pixel = camera.pixel(self.position_in_world)
        if "crop" in data:  # AI-predicted data only
            self.crop = YoloCrop.from_string(data["crop"])
        else:
            self.crop = camera.fighter_crop(self.position_in_world)
        self.motion_hex = anim_ontology.motion_hex(self.motion_kind)
        self.action_string = anim_ontology.HEX_TO_ACTION.get(self.motion_hex, "")
        self.action = anim_ontology.anim_for_string_and_status_kind(self.action_string, self.status_kind)
        if "action" in data:  # AI-predicted data only
            self.action = data["action"]
        # (predicted_action_confidence is carried by the timeline dict but, as in the reference, not read here)
        self.hitstun_left = data["hitstun_left"]

    # attributes whose previous-frame value update() keeps as previous_<name> (fighter.py:567-584)
    _REMEMBERED = ("position_in_world", "damage", "facing", "fighter_id", "motion_kind", "num_frames_left", "pos_x",
                   "pos_y", "shield_size", "status_kind", "stock_count", "fighter_name", "crop", "motion_hex",
                   "action_string", "attack_connected", "action")

    def update(self, frame_number: int, data: Dict):
        self.frame_num = frame_number
        for name in self._REMEMBERED:
            setattr(self, "previous_" + name, getattr(self, name))
        self.set_from_json(data)
        # max(): a respawn resets damage to 0 and must not count as negative damage
        self.damage_delta = max(self.damage - self.previous_damage, 0)
        self.new_action = self.previous_action != self.action
        if self.new_action:
            self.move_counter += 1
        self.animation_frame_num = 1 if self.new_action else self.animation_frame_num + 1
        self.frames_since_damaged = 0 if self.damage_delta else self.frames_since_damaged + 1
        self.frames_since_hit = 0 if self.damage_delta else self.frames_since_hit + 1
        if self.previous_action != "Damaged":
            self.previous_non_damaged_action = self.previous_action
        if self.in_tech_situation:
            self.last_frame_in_tech_situation = frame_number
        if self.in_ledge_situation:
            self.last_frame_in_ledge_situation = frame_number

    @property
    def in_tech_situation(self) -> bool:
        return anim_ontology.OPTION_GROUP[self.action] == "tech"  # KeyError for an action outside the ontology, as the reference

    @property
    def in_ledge_situation(self) -> bool:
        return anim_ontology.OPTION_GROUP[self.action] == "ledge"

    @property
    def time_remaining(self) -> str:
        minutes, seconds = divmod(self.num_frames_left / 60, 60)
        seconds, fraction = divmod(seconds, 1)
        return f"{int(minutes)}:{int(seconds):02d}.{round(fraction * 100):02d}"

    def __str__(self):
        return (f"<{self.fighter_name}@{self.action} | {self.advantage_state} | "
                f"{self.crop_confidence:.2f}%  {self.crop.center_x:.2f}x{self.crop.center_y:.2f}y />")
