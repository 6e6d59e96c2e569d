"""The reference's default scene rendered with the reference's own tag images (tests/golden/tag_textures.npz)
and its committed trajectory (tests/golden/reference_trajectory.json).  Test infrastructure."""
import contextlib
import io
import json
import os

import numpy as np

from aprilslam_amd import synth
from aprilslam_amd.slam import SLAM

HERE = os.path.dirname(os.path.abspath(__file__))
TRAJ = json.load(open(os.path.join(HERE, "golden", "reference_trajectory.json")))["rows"]
# rows beyond this one have tag 0 clipped by the image border and the graph running on stale world
# transforms accumulated by the legacy loop; the detector/PnP comparison uses the rows before it
N_TAG0_ROWS = 60

_TEX = None


def textures():
    """id -> (354, 354, 3) uint8 RGB, as the reference's renderer uploads them (renderer.py:160-171)."""
    global _TEX
    if _TEX is None:
        t = np.load(os.path.join(HERE, "golden", "tag_textures.npz"))["textures"]
        _TEX = {i: np.repeat(t[i][:, :, None], 3, axis=2) for i in range(t.shape[0])}
    return _TEX


SCENE = synth.default_scene()
W, H = SCENE["display_width"], SCENE["display_height"]
K = synth.camera_matrix(W, H, SCENE["fov_y"])
TAG_SIZE = SCENE["tag_size_inner"] * SCENE["size_scale"]


def render(cam_position, cam_rotation_deg=(0, 0, 0)):
    return synth.render_frame(W, H, SCENE["tags"], SCENE["tag_size_outer"] * SCENE["size_scale"],
                              cam_position=cam_position, cam_rotation_deg=cam_rotation_deg, textures=textures())


def camera_position(row):
    x, y, z = row["gt_xyz"]
    return (x, y, z - 50.0)


class Log:
    def __init__(self):
        self.lines = []

    def info(self, m):
        self.lines.append(str(m))


def new_slam(log=None):
    return SLAM(log or Log(), {"camera_matrix": K, "dist_coeffs": np.zeros((4, 1))}, detector=object())


def feed(slam, ids, T):
    with contextlib.redirect_stdout(io.StringIO()):
        return slam.process_observations(list(ids), T)
