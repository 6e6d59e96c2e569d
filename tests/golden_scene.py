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
# the first 60 rows are the poses round 2 pinned; tag 0 stays in view up to row 77 (3, then 4, then 5 nodes), and in rows
# 78-88 it has left the image: the visible tags 2, 3, 4 keep the world transforms they got while it was in view
# (branch C2, slam_graph.py:50-53)
N_TAG0_ROWS = 60
N_TAG0_VISIBLE = 78
# The bars the oracle and the HIP path are held to against the reference's own logged estimate (est_xyz, est_rpy),
# tests/golden/README.md has the derivation; observed values of this build in brackets
LEAD_ROWS = 33           # rows before the reference's first broken edge
LEAD_POS, LEAD_RPY = 0.025, 0.4e-3       # units, rad   [0.0207, 0.25e-3]
CLEAN_POS = 0.05         # wherever the reference itself is within 0.1 units of ground truth   [0.032]
ALL_POS, ALL_RPY = 0.25, 2.0e-3          # every row with tag 0 in view   [0.229, 1.95e-3]
ABOVE_018 = [38, 40, 63, 67, 68, 71, 73]  # the rows further than 0.18 units (1 mm) from the reference: all on frames where
#                                           the reference itself is 0.49 .. 1.75 units off ground truth

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


def wrap(a):
    return (np.asarray(a) + np.pi) % (2 * np.pi) - np.pi


def check_trajectory(poses, ids, nodes):
    """poses / ids / node counts of the 89 rows (one run through one SLAM object) against the reference's log."""
    from aprilslam_amd.harness import rotation_matrix_to_euler
    est = np.array([r["est_xyz"] for r in TRAJ]); gt = np.array([r["gt_xyz"] for r in TRAJ])
    ours = np.array([p[:3, 3] for p in poses])
    d_ref = np.linalg.norm(ours - est, axis=1)
    e_ref, e_ours = np.linalg.norm(est - gt, axis=1), np.linalg.norm(ours - gt, axis=1)
    d_rpy = np.abs(wrap(np.array([rotation_matrix_to_euler(p[:3, :3]) for p in poses]) - np.array([r["est_rpy"] for r in TRAJ]))).max(axis=1)
    assert [int(n) for n in nodes] == [r["num_nodes"] for r in TRAJ]
    assert all(i[0] == 0 for i in ids[:N_TAG0_VISIBLE]) and all(list(i) == [2, 3, 4] for i in ids[N_TAG0_VISIBLE:])
    v = slice(0, N_TAG0_VISIBLE)
    assert d_ref[:LEAD_ROWS].max() < LEAD_POS and d_rpy[:LEAD_ROWS].max() < LEAD_RPY, (d_ref[:LEAD_ROWS].max(), d_rpy[:LEAD_ROWS].max())
    assert d_ref[v].max() < ALL_POS and d_rpy.max() < ALL_RPY, (d_ref[v].max(), d_rpy.max())
    clean = e_ref[v] < 0.1
    assert d_ref[v][clean].max() < CLEAN_POS and (e_ours[v][clean] < 0.1).all() and (e_ours[v][~clean] > 0.1).all()
    assert [int(k) for k in np.nonzero(d_ref[v] > 0.18)[0]] == ABOVE_018, np.nonzero(d_ref[v] > 0.18)[0]
    assert (d_ref[:N_TAG0_ROWS] < 0.05).sum() >= 42
    rm_o, rm_r = np.sqrt((e_ours[:N_TAG0_ROWS] ** 2).mean()), np.sqrt((e_ref[:N_TAG0_ROWS] ** 2).mean())
    assert abs(rm_o - rm_r) < 0.15 * rm_r, (rm_o, rm_r)
    # Tag 0 out of view: the position rides on world transforms that tags 2-4 received in some earlier frame.  The
    # reference's run had 570 logged frames (and unlogged ones between them), the fixture keeps 89, so the frame that
    # re-anchored them last is not the same and the estimate is not reproducible: what is held is the rotation (which
    # does not depend on it), an error vs ground truth no larger than the reference's own, and a distance from the
    # reference's estimate no larger than its distance from ground truth.
    t = slice(N_TAG0_VISIBLE, len(TRAJ))
    assert (e_ours[t] <= e_ref[t] + 0.05).all() and (d_ref[t] <= e_ref[t]).all(), (e_ours[t], d_ref[t], e_ref[t])
    return d_ref, d_rpy
