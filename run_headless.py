#!/usr/bin/env python3
"""Headless counterpart of the reference's `python run_simulation.py --no-movement` (run_simulation.py:88-206):
the default scene, Monte-Carlo camera positions (camera_controller.py:105-121 draws uniform positions), the same
per-frame SLAM calls, the same CSV.  Needs a GPU (the detector has no CPU path)."""
import argparse
import json
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from aprilslam_amd import synth  # noqa: E402
from aprilslam_amd.harness import HeadlessSimulation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", "-c", default=None, help="sim_settings.json (default: the reference's default scene)")
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--seed", type=int, default=20250620)
    ap.add_argument("--out", default="gpurun_out/headless")
    args = ap.parse_args()
    logging.basicConfig(level=logging.WARNING)
    config = json.load(open(args.config)) if args.config else synth.default_scene()
    sim = HeadlessSimulation(config, logging, output_dir=args.out)
    rng = np.random.default_rng(args.seed)
    for _ in range(args.frames):
        sim.step(rng.uniform([-8, -8, -10], [8, 8, 15]))
    sim.close()
    print(json.dumps(sim.statistics()))


if __name__ == "__main__":
    main()
