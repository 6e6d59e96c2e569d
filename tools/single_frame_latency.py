#!/usr/bin/env python3
"""Latency of the drop-in per-frame calls (host numpy frame in, Python objects out), as the reference harness makes
them: SLAM.detect(frame) + SLAM.get_pose(d) per tag + SLAM.my_pose() (simulation_engine.py:219-232)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aprilslam_amd import synth  # noqa: E402
from aprilslam_amd.slam import SLAM
from aprilslam_amd.tag_detector import TagDetector  # noqa: E402


class _Log:
    def info(self, m):
        pass


def main():
    out = {}
    for name, (w, h, ntags) in {"default scene 1000x1000": (1000, 1000, 0), "1280x720, 20 tags": (1280, 720, 20)}.items():
        if ntags == 0:
            sc = synth.default_scene()
            frame, _ = synth.render_frame(w, h, sc["tags"], 18.0, cam_position=(-0.7, -0.4, 1.1), cam_rotation_deg=(0.5, -1.0, -0.7))
        else:
            rng = np.random.default_rng(20250620 + 1)
            frame, _ = synth.render_frame(w, h, synth.random_scene(w, h, ntags, rng), 18.0)
        K = synth.camera_matrix(w, h)
        cp = {"camera_matrix": K, "dist_coeffs": np.zeros((4, 1))}
        # the synthetic 20-tag scene uses ids beyond the five the reference pins: open the whole table for it
        slam = SLAM(_Log(), cp, tag_size=10.0, detector=None if ntags == 0 else TagDetector(cp, tag_size=10.0, id_limit=0))
        n = 0
        for it in range(60):
            if it == 10:
                t0 = time.perf_counter()
            dets = slam.detect(frame)
            for d in dets:
                slam.get_pose(d)
            slam.my_pose()
            n = len(dets)
        dt = (time.perf_counter() - t0) / 50
        t1 = time.perf_counter()
        for it in range(50):
            slam.detect(frame)
        dd = (time.perf_counter() - t1) / 50
        out[name] = {"tags": n, "ms_per_frame_detect_pose_graph": round(dt * 1e3, 3), "ms_per_frame_detect_only": round(dd * 1e3, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
