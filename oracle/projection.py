"""ORACLE (test infrastructure, never shipped, never on the product path).

Literal numpy restatement of the log-projection box path of the reference:
``calculate_focal_length`` / ``calculate_intrinsic_matrix`` / ``calculate_lookat_matrix`` /
``project_point_to_pixel`` (``playaid/fighter.py:31-155``), the four corner offsets of
``Fighter.set_from_json`` (``:507-526``) and ``YoloCrop.from_pixel_coordinates``
(``:170-190``), including ``np.linalg.inv`` on the 4x4 pose.

Pinning: the reference's only test that builds a ``Fighter`` (``fighter_test.py``) is stale
and asserts nothing about the crop, so this is "parity unpinned" (numpy is the reference's
own dependency and is what runs here).
"""
import numpy as np


def project_box(pos_x, pos_y, camera_position, target_position, fov, image_width=1280, image_height=720):
    fov_rad = np.deg2rad(fov)
    f = image_width / (2 * np.tan(fov_rad / 2))
    K = np.array([[f, 0, image_width / 2], [0, f, image_height / 2], [0, 0, 1]])
    forward = np.array(camera_position, dtype=np.float64) - np.array(target_position, dtype=np.float64)
    forward /= np.linalg.norm(forward)
    up = np.array([0, 1, 0])
    right = np.cross(up, forward)
    right /= np.linalg.norm(right)
    up = np.cross(forward, right)
    pose = np.eye(4)
    pose[0, :3] = right
    pose[1, :3] = up
    pose[2, :3] = -forward
    pose[:3, 3] = camera_position
    inv = np.linalg.inv(pose)
    world = np.array([pos_x, pos_y, 0], dtype=np.float64)
    pts = []
    for off in ([-10, 20, 0], [10, 20, 0], [-10, -3, 0], [10, -3, 0]):
        ph = np.append(world + np.array(off), 1)
        pc = inv @ ph
        pn = pc[:3] / pc[2]
        px = K @ pn
        px[1] = image_height - px[1]
        pts.append(np.round(px[:2]).astype(int))
    xs = [int(p[0]) for p in pts]
    ys = [int(p[1]) for p in pts]
    cx = sum(xs) / 4 / image_width
    cy = sum(ys) / 4 / image_height
    return (cx, cy, (max(xs) - min(xs)) / image_width, (max(ys) - min(ys)) / image_height)
