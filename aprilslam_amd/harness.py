"""Headless driver for the reference's per-frame loop.

The reference harness (src/simulation/simulation_engine.py:145-300) needs pygame + OpenGL + a GUI; what it does
per frame is small and is reproduced here without them:

    frame = render(camera)                        renderer.py:197-274        -> aprilslam_amd.synth.render_frame
    detections = slam.detect(frame)               simulation_engine.py:219
    for d in detections: slam.get_pose(d)         simulation_engine.py:222-223
    pose = slam.my_pose()                         simulation_engine.py:232
    errors vs analytic ground truth, CSV row      simulation_engine.py:240-300, data_logger.py:110-183

Ground truth and error metrics follow src/simulation/ground_truth.py:146-188 (camera pose in the world tag's
frame, OpenGL->OpenCV flip diag(1,-1,-1)), :214-239 (ZYX Euler) and :274-300 (|dt|, Frobenius |dR|).
The CSV has the reference's 17 columns in the reference's order, so src/analysis/* can read it unchanged.
"""
import csv
import json
import os
import time

import numpy as np

from . import synth
from .slam import SLAM

MAIN_CSV_HEADER = ['Time', 'Number_of_Nodes', 'Average_Distance', 'Est_X', 'Est_Y', 'Est_Z', 'Est_Roll', 'Est_Pitch',
                   'Est_Yaw', 'GT_X', 'GT_Y', 'GT_Z', 'GT_Roll', 'GT_Pitch', 'GT_Yaw', 'Translation_Difference',
                   'Rotation_Difference']

_FLIP = np.diag([1.0, -1.0, -1.0])


def euler_to_rotation_matrix(euler_deg):
    """[roll(x), pitch(y), yaw(z)] degrees -> Rz Ry Rx (ground_truth.py:241-272)."""
    r, p, y = np.radians(np.asarray(euler_deg, dtype=np.float64))
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def rotation_matrix_to_euler(R):
    """3x3 -> [roll, pitch, yaw] radians, ZYX (ground_truth.py:214-239)."""
    sy = np.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    if sy >= 1e-6:
        return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])])
    return np.array([np.arctan2(-R[1, 2], R[1, 1]), np.arctan2(-R[2, 0], sy), 0.0])


def calculate_pose_error(estimated, ground_truth):
    """(|dt|, Frobenius |dR|) (ground_truth.py:274-300)."""
    return (float(np.linalg.norm(estimated[:3, 3] - ground_truth[:3, 3])),
            float(np.linalg.norm(estimated[:3, :3] - ground_truth[:3, :3], 'fro')))


class GroundTruth:
    """Analytic ground truth for a static-orientation camera, as the reference computes it."""

    def __init__(self, tags):
        self.tags = {int(t["id"]): t for t in tags}

    def camera_to_tag(self, tag_id, camera_position):
        t = self.tags[tag_id]
        rel = np.asarray(t["position"], dtype=np.float64) - np.asarray(camera_position, dtype=np.float64)
        rel[1:] = -rel[1:]
        T = np.eye(4)
        T[:3, :3] = _FLIP @ euler_to_rotation_matrix(t["rotation"])
        T[:3, 3] = rel
        return T

    def inverse_transform(self, tag_id, camera_position):
        """camera pose in tag `tag_id`'s frame (what SLAM.my_pose estimates when that tag is the world)."""
        T = self.camera_to_tag(tag_id, camera_position)
        out = np.eye(4)
        out[:3, :3] = T[:3, :3].T
        out[:3, 3] = -T[:3, :3].T @ T[:3, 3]
        return out


class HeadlessSimulation:
    def __init__(self, config, logger, output_dir=None, device=0, slam=None):
        if isinstance(config, str):
            with open(config) as f:
                config = json.load(f)
        self.config = config
        self.width, self.height = int(config["display_width"]), int(config["display_height"])
        self.scale = float(config.get("size_scale", 1))
        self.tag_size_inner = config["tag_size_inner"] * self.scale
        self.tag_size_outer = config["tag_size_outer"] * self.scale
        self.mm_per_unit = float(config.get("actual_size_in_mm", 0)) / self.tag_size_inner if config.get("actual_size_in_mm") else None
        self.tags = config["tags"]
        # simulation_engine.py:124-134
        self.camera_matrix = synth.camera_matrix(self.width, self.height, config.get("fov_y", 45))
        params = {"camera_matrix": self.camera_matrix, "dist_coeffs": np.zeros((4, 1))}
        self.slam = slam if slam is not None else SLAM(logger, params, tag_size=self.tag_size_inner, device=device)
        self.ground_truth = GroundTruth(self.tags)
        self.rows = []
        self.start_time = time.time()
        self._file = None
        self._writer = None
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
            self._file = open(os.path.join(output_dir, "slam_simulation_data.csv"), "w", newline="")
            self._writer = csv.writer(self._file)
            self._writer.writerow(MAIN_CSV_HEADER)

    def step(self, camera_position, camera_rotation=(0.0, 0.0, 0.0)):
        """One iteration of the reference's main loop.  The reference's ground truth ignores camera rotation
        (ground_truth.py:146-188), so error columns are only meaningful for camera_rotation == 0."""
        frame, _ = synth.render_frame(self.width, self.height, self.tags, self.tag_size_outer, cam_position=camera_position,
                                      cam_rotation_deg=camera_rotation, fov_y_deg=self.config.get("fov_y", 45))
        detections = self.slam.detect(frame)
        for d in detections:
            self.slam.get_pose(d)
        pose = self.slam.my_pose()
        if pose is None:
            return None
        gt = self.ground_truth.inverse_transform(self.slam.coordinate_id, camera_position)
        dt, dr = calculate_pose_error(pose, gt)
        est_e, gt_e = rotation_matrix_to_euler(pose[:3, :3]), rotation_matrix_to_euler(gt[:3, :3])
        row = [time.time() - self.start_time, len(self.slam.graph.get_nodes()), self.slam.average_distance_to_nodes(),
               pose[0, 3], pose[1, 3], pose[2, 3], est_e[0], est_e[1], est_e[2],
               gt[0, 3], gt[1, 3], gt[2, 3], gt_e[0], gt_e[1], gt_e[2], dt, dr]
        self.rows.append(row)
        if self._writer:
            self._writer.writerow(row)
        return {"pose": pose, "ground_truth": gt, "translation_error": dt, "rotation_error": dr, "ids": [d["id"] for d in detections]}

    def statistics(self):
        """frames/s over logged frames (data_logger.py:266-286) plus error RMSE in units and, if known, mm."""
        if not self.rows:
            return {"frames": 0}
        a = np.array([[r[15], r[16]] for r in self.rows])
        runtime = self.rows[-1][0] if self.rows[-1][0] > 0 else float("nan")
        out = {"frames": len(self.rows), "fps": len(self.rows) / runtime, "translation_rmse_units": float(np.sqrt((a[:, 0] ** 2).mean())),
               "rotation_fro_rmse": float(np.sqrt((a[:, 1] ** 2).mean()))}
        if self.mm_per_unit:
            out["translation_rmse_mm"] = out["translation_rmse_units"] * self.mm_per_unit
        return out

    def close(self):
        if self._file:
            self._file.close()
            self._file = None
