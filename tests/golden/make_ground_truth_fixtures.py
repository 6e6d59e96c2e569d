#!/usr/bin/env python3
"""Generate tests/golden/ground_truth_fixtures.json by RUNNING the reference's own ground-truth code.

Runs only in the build container (needs /root/reference).  src/simulation/ground_truth.py is imported unmodified;
its package pulls in pygame / OpenGL / cv2 / termcolor at import time only (window, GL calls, colours), so empty
stand-in modules are registered first -- the functions exercised here are pure NumPy.  The fixture holds inputs and
expected outputs only (float64 as hex strings, bit-exact round trip); no reference source text.

Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ground_truth_fixtures.py
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)


class _Nothing:
    """stands for any constant, class or function of the GUI stack; none of them is used by the code exercised here"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Nothing()

    def __call__(self, *a, **k):
        return _Nothing()


class _Anything(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Nothing()


for name in ("pygame", "pygame.locals", "OpenGL", "OpenGL.GL", "OpenGL.GLU", "cv2", "apriltag", "termcolor"):
    m = _Anything(name)
    m.__all__ = []
    sys.modules[name] = m
sys.modules["apriltag"].apriltag = lambda *a, **k: None
sys.modules["termcolor"].colored = lambda s, *a, **k: s

from src.simulation.config_manager import SimulationConfig  # noqa: E402
from src.simulation.ground_truth import GroundTruthCalculator  # noqa: E402
from src.simulation.renderer import TagData  # noqa: E402


def hexm(a):
    return [float(x).hex() for x in np.asarray(a, dtype=np.float64).ravel()]


config = SimulationConfig(os.path.join(REF, "config", "sim_settings.json"))
tags = [TagData(t["id"], 0, np.array(t["position"], dtype=np.float32), np.array(t["rotation"], dtype=np.float32)) for t in config.tags]
gt = GroundTruthCalculator(config, tags)
rng = np.random.default_rng(20250620)
cams = [np.zeros(3), np.array([2.0, -2.0, 34.0])] + [rng.uniform([-15, -5, -5], [60, 5, 70]) for _ in range(6)]
cases = []
for cam in cams:
    c = {"camera_position": hexm(cam), "camera_to_tag": {}, "inverse": {}, "tag_to_tag": {}, "tag_world": {}}
    for t in config.tags:
        i = t["id"]
        c["camera_to_tag"][str(i)] = hexm(gt.get_camera_to_tag_transform(i, cam.copy()))
        c["inverse"][str(i)] = hexm(gt.get_inverse_transform(i, cam.copy()))
        c["tag_to_tag"][str(i)] = float(gt.get_tag_to_tag_distance(i, 0, cam.copy())).hex()
        c["tag_world"][str(i)] = hexm(gt.get_tag_world_transform(i, cam.copy(), 0))
    cases.append(c)
euler = []
for _ in range(12):
    e = rng.uniform(-170, 170, 3)
    R = GroundTruthCalculator._euler_to_rotation_matrix(e)
    euler.append({"euler_deg": hexm(e), "R": hexm(R), "back": hexm(GroundTruthCalculator.rotation_matrix_to_euler(R))})
# gimbal lock branch
R = GroundTruthCalculator._euler_to_rotation_matrix(np.array([30.0, 90.0, 0.0]))
euler.append({"euler_deg": hexm([30.0, 90.0, 0.0]), "R": hexm(R), "back": hexm(GroundTruthCalculator.rotation_matrix_to_euler(R))})
errors = []
for _ in range(6):
    A, B = np.eye(4), np.eye(4)
    A[:3, :3] = GroundTruthCalculator._euler_to_rotation_matrix(rng.uniform(-40, 40, 3)); A[:3, 3] = rng.uniform(-50, 50, 3)
    B[:3, :3] = GroundTruthCalculator._euler_to_rotation_matrix(rng.uniform(-40, 40, 3)); B[:3, 3] = rng.uniform(-50, 50, 3)
    dt, dr = gt.calculate_pose_error(A, B)
    errors.append({"estimated": hexm(A), "ground_truth": hexm(B), "translation": float(dt).hex(), "rotation": float(dr).hex()})
out = {
    "source": "reference src/simulation/ground_truth.py run on config/sim_settings.json (tags as the renderer loads them: float32)",
    "tags": [{"id": t["id"], "position": t["position"], "rotation": t["rotation"]} for t in config.tags],
    "sizes": {"tag_size_inner": config.tag_size_inner, "tag_size_outer": config.tag_size_outer,
              "mm_of_1_unit": float(gt.convert_simulation_to_mm(1.0)), "units_of_1_mm": float(gt.convert_mm_to_simulation(1.0))},
    "cases": cases, "euler": euler, "pose_errors": errors,
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ground_truth_fixtures.json")
json.dump(out, open(path, "w"), indent=0)
print(len(cases), "camera positions,", len(euler), "Euler cases ->", os.path.getsize(path), "bytes")
