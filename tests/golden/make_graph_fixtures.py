#!/usr/bin/env python3
"""Generate tests/golden/graph_fixtures.json by RUNNING the reference's own graph code.

Runs only in the build container (needs /root/reference).  The reference modules
src/core/slam_graph.py and src/core/slam.py are imported unmodified; slam.py needs the
absent native packages `cv2` and `apriltag` only at import time, so empty stand-in module
objects are registered first and poses are injected exactly as
SLAM.get_pose does (slam.py:30-31).  The fixture holds inputs and expected outputs only
(float64 numbers as hex strings so they round-trip bit-exactly); no reference source text.

Usage: PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_graph_fixtures.py
"""
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)
for name in ("cv2", "apriltag"):
    m = types.ModuleType(name)
    if name == "apriltag":
        m.apriltag = lambda *a, **k: None
    sys.modules[name] = m

from src.core.slam import SLAM  # noqa: E402
from src.core.slam_graph import SLAMGraph  # noqa: E402


class _Log:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(str(msg))


def hexm(a):
    return [float(x).hex() for x in np.asarray(a, dtype=np.float64).ravel()]


def rand_pose(rng, scale=100.0):
    w = rng.normal(size=3)
    w = w / np.linalg.norm(w) * rng.uniform(0, np.pi)
    th = np.linalg.norm(w)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-scale, scale, size=3)
    return T


def node_state(graph):
    out = {}
    for tid, n in sorted(graph.get_nodes().items()):
        out[str(tid)] = {"local": hexm(n.local), "world": hexm(n.world), "reference": int(n.reference),
                         "weight": int(n.weight), "updated": bool(n.updated), "visible": bool(n.visible)}
    return out


def run_scenario(name, frames, rng):
    """frames: list of lists of visible tag ids (ascending, as TagDetector.detect sorts them)."""
    log = _Log()
    slam = SLAM.__new__(SLAM)  # skip __init__: it would construct TagDetector/SLAMVisualizer (cv2, figures)
    slam.logger = log
    slam.graph = SLAMGraph(log)
    slam.visible_tags = []
    rec = {"name": name, "frames": []}
    for ids in frames:
        obs = [(int(t), rand_pose(rng)) for t in ids]
        slam.visible_tags = [t for t, _ in obs]
        buf = io.StringIO()
        with redirect_stdout(buf):
            for t, T in obs:
                slam.graph.add_or_update_node(t, T, slam.visible_tags)  # slam.py:30-31
            pose = slam.my_pose()
            avg = slam.average_distance_to_nodes()
        rec["frames"].append({
            "visible": [t for t, _ in obs],
            "T": {str(t): hexm(T) for t, T in obs},
            "coordinate_id": int(slam.coordinate_id),
            "nodes": node_state(slam.graph),
            "my_pose": None if pose is None else hexm(pose),
            "estimated_pose": hexm(slam.graph.get_estimated_pose()),
            "avg_distance": float(avg).hex(),
            "stdout": buf.getvalue().splitlines(),
        })
    return rec


def main():
    rng = np.random.default_rng(20250620)
    scenarios = [
        # A then C1 (reference == coordinate id visible)
        ("A_C1_basic", [[0], [0, 1, 2], [0, 1, 2, 3]]),
        # C2: known node whose reference is the coordinate id, seen without the coordinate tag
        ("C2_keep_world", [[0, 1], [1], [1, 2]]),
        # C3: chain through a non-coordinate reference; weights grow; updated flag inherited
        ("C3_chain", [[0, 1], [1, 2], [2, 3], [3, 4], [1, 4]]),
        # B: lower id appears later -> coordinate switch, stale worlds kept, "No world update"
        ("B_switch", [[3, 4], [4, 5], [1, 3, 4], [1, 5], [0, 5]]),
        # C4: reference unknown to the graph -> "Cannot find world reference"
        ("C4_unknown_ref", [[0, 1], [2, 3], [3], [2, 3, 4]]),
        # empty frames and re-observation of the coordinate tag alone
        ("empty_and_repeat", [[], [2], [], [2], [2, 7], [7], []]),
        # long random walk over 12 tags
        ("random_walk", None),
    ]
    out = []
    for name, frames in scenarios:
        if frames is None:
            frames = []
            for _ in range(40):
                k = int(rng.integers(0, 5))
                frames.append(sorted(int(x) for x in rng.choice(12, size=k, replace=False)))
        out.append(run_scenario(name, frames, rng))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_fixtures.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_graph_fixtures.py", "reference": "mikostrzewa/AprilSLAM src/core/slam_graph.py + slam.py",
                   "numpy": np.__version__, "scenarios": out}, f)
    print("wrote", path, os.path.getsize(path), "bytes;", sum(len(s["frames"]) for s in out), "frames")


if __name__ == "__main__":
    main()
