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
The three CSV files have the reference's names, columns and column order (data_logger.py:105-147: 17-column
slam_simulation_data.csv, 22-column error_analysis.csv, 8-column covariance_analysis.csv; rows as
simulation_engine.py:240-356 fills them), so src/analysis/* can read them unchanged.
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

ERROR_CSV_HEADER = ['Number_of_Jumps', 'Est_X_Local', 'Est_Y_Local', 'Est_Z_Local', 'Est_Roll_Local', 'Est_Pitch_Local',
                    'Est_Yaw_Local', 'Est_X_World', 'Est_Y_World', 'Est_Z_World', 'Est_Roll_World', 'Est_Pitch_World',
                    'Est_Yaw_World', 'Tag_Est_X', 'Tag_Est_Y', 'Tag_Est_Z', 'Tag_Est_Roll', 'Tag_Est_Pitch', 'Tag_Est_Yaw',
                    'Error_World', 'Error_Local', 'Translation_Error']
COVARIANCE_CSV_HEADER = ['Number_of_Jumps', 'Tag_Est_X', 'Tag_Est_Y', 'Tag_Est_Z', 'Tag_Est_Roll', 'Tag_Est_Pitch', 'Tag_Est_Yaw',
                         'Translation_Error']

_FLIP = np.diag([1.0, -1.0, -1.0])


def euler_to_rotation_matrix(euler_deg):
    """[roll(x), pitch(y), yaw(z)] degrees -> Rz Ry Rx (ground_truth.py:241-272).  Like the reference it computes in the
    precision it is handed: the renderer's float32 tag rotations give a float32 matrix."""
    r, p, y = np.radians(euler_deg)
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
        # the reference's renderer keeps tag positions and rotations as float32 (renderer.py:110-111)
        self.tags = {int(t["id"]): {"position": np.array(t["position"], dtype=np.float32), "rotation": np.array(t["rotation"], dtype=np.float32)}
                     for t in tags}

    def camera_to_tag(self, tag_id, camera_position):
        """ground_truth.py:48-90"""
        if tag_id not in self.tags:
            raise ValueError(f"Tag {tag_id} not found in configuration")
        t = self.tags[tag_id]
        rel = t["position"] - np.asarray(camera_position)
        rel[1:] = -rel[1:]
        T = np.eye(4)
        T[:3, :3] = _FLIP @ euler_to_rotation_matrix(t["rotation"])
        T[:3, 3] = rel
        return T

    def tag_world_transform(self, tag_id, camera_position, coordinate_frame_tag_id):
        """ground_truth.py:92-114 (the product of the two camera<-tag transforms, as the reference forms it)"""
        return self.camera_to_tag(tag_id, camera_position) @ self.camera_to_tag(coordinate_frame_tag_id, camera_position)

    def tag_to_tag_distance(self, tag1_id, tag2_id, camera_position):
        """ground_truth.py:116-144"""
        if tag1_id not in self.tags or tag2_id not in self.tags:
            raise ValueError(f"One or both tags not found: {tag1_id}, {tag2_id}")
        cam = np.asarray(camera_position)
        return np.linalg.norm((self.tags[tag1_id]["position"] - cam) - (self.tags[tag2_id]["position"] - cam))

    def inverse_transform(self, tag_id, camera_position):
        """camera pose in tag `tag_id`'s frame (what SLAM.my_pose estimates when that tag is the world), ground_truth.py:146-188.
        Same expressions on arrays of the same memory layout as the reference's (the product with a transposed view takes
        its own path through BLAS), so the result is bit-identical."""
        if tag_id not in self.tags:
            raise ValueError(f"Tag {tag_id} not found in configuration")
        t = self.tags[tag_id]
        rel = t["position"] - np.asarray(camera_position)
        rel[1:] = -rel[1:]
        rotation = _FLIP @ euler_to_rotation_matrix(t["rotation"])
        inverse_rotation = rotation.T
        out = np.eye(4)
        out[:3, :3] = inverse_rotation
        out[:3, 3] = -inverse_rotation @ rel
        return out


class HeadlessSimulation:
    def __init__(self, config, logger, output_dir=None, device=0, slam=None, textures=None):
        """textures: {tag id: (h, w, 3) uint8} tag images as the reference's renderer uploads them (renderer.py:160-171:
        assets/tags/tag<id>.png); None = the synthetic 40-texel-per-cell bitmaps of aprilslam_amd.synth."""
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
        self.textures = textures
        self.rows, self.error_rows, self.covariance_rows = [], [], []
        self.start_time = time.time()
        self._files, self._writer, self._error_writer, self._covariance_writer = [], None, None, None
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
            writers = []
            for name, header in (("slam_simulation_data.csv", MAIN_CSV_HEADER), ("error_analysis.csv", ERROR_CSV_HEADER),
                                 ("covariance_analysis.csv", COVARIANCE_CSV_HEADER)):
                f = open(os.path.join(output_dir, name), "w", newline="")
                self._files.append(f)
                w = csv.writer(f)
                w.writerow(header)
                writers.append(w)
            self._writer, self._error_writer, self._covariance_writer = writers

    def step(self, camera_position, camera_rotation=(0.0, 0.0, 0.0)):
        """One iteration of the reference's main loop.  The reference's ground truth ignores camera rotation
        (ground_truth.py:146-188), so error columns are only meaningful for camera_rotation == 0."""
        frame, _ = synth.render_frame(self.width, self.height, self.tags, self.tag_size_outer, cam_position=camera_position,
                                      cam_rotation_deg=camera_rotation, fov_y_deg=self.config.get("fov_y", 45), textures=self.textures)
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
        self._log_node_analysis(camera_position)
        return {"pose": pose, "ground_truth": gt, "translation_error": dt, "rotation_error": dr, "ids": [d["id"] for d in detections]}

    def _log_node_analysis(self, camera_position):
        """One row per visible node in error_analysis.csv and covariance_analysis.csv (simulation_engine.py:302-356)."""
        for tag_id, node in self.slam.graph.get_nodes().items():
            if not node.visible:
                continue
            gt_local = self.ground_truth.camera_to_tag(tag_id, camera_position)
            gt_world_distance = self.ground_truth.tag_to_tag_distance(tag_id, self.slam.coordinate_id, camera_position)
            lt, le = node.local[:3, 3], rotation_matrix_to_euler(node.local[:3, :3])
            wt, we = node.world[:3, 3], rotation_matrix_to_euler(node.world[:3, :3])
            gtt, ge = gt_local[:3, 3], rotation_matrix_to_euler(gt_local[:3, :3])
            local_error = abs(np.linalg.norm(lt) - np.linalg.norm(gtt))
            world_error = abs(np.linalg.norm(wt) - gt_world_distance)
            translation_error = np.linalg.norm(lt - gtt)
            erow = [node.weight, lt[0], lt[1], lt[2], le[0], le[1], le[2], wt[0], wt[1], wt[2], we[0], we[1], we[2],
                    gtt[0], gtt[1], gtt[2], ge[0], ge[1], ge[2], world_error, local_error, translation_error]
            crow = [node.weight, lt[0], lt[1], lt[2], le[0], le[1], le[2], translation_error]  # the reference logs the LOCAL pose here
            self.error_rows.append(erow)
            self.covariance_rows.append(crow)
            if self._error_writer:
                self._error_writer.writerow(erow)
                self._covariance_writer.writerow(crow)

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
        for f in self._files:
            f.close()
        self._files = []
