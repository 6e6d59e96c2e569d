#!/usr/bin/env python3
"""Randomised parity sweep on cluttered frames, GPU against the CPU restatement (test infrastructure, like tests/):
tag scenes overlaid with what a real camera sees and a renderer does not -- rectangles of arbitrary gray, blocky noise at
1-8 pixel scale, stripes, checkerboards, gradients -- which fill the cluster table, the point pool and the size classes the
clean scenes never reach (buffers must grow and the batch re-run, never truncate).  usage: fuzz_clutter.py [n_frames] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from aprilslam_amd import _lib, synth  # noqa: E402
from aprilslam_amd.families import get_family  # noqa: E402


def clutter(frame, rng):
    h, w = frame.shape[:2]
    out = frame.copy()
    for _ in range(int(rng.integers(1, 12))):
        kind = int(rng.integers(0, 6))
        x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
        x1, y1 = min(w, x0 + int(rng.integers(8, w // 2))), min(h, y0 + int(rng.integers(8, h // 2)))
        hh, ww = y1 - y0, x1 - x0
        if kind == 0:
            patch = np.full((hh, ww), int(rng.integers(0, 256)), np.uint8)
        elif kind == 1:
            s = int(rng.integers(1, 9))
            patch = (rng.integers(0, 2, (hh // s + 1, ww // s + 1), dtype=np.uint8) * int(rng.integers(60, 256))).repeat(s, 0).repeat(s, 1)[:hh, :ww]
        elif kind == 2:
            s = int(rng.integers(1, 7))
            patch = ((((np.arange(hh) // s) & 1)[:, None] * 255) * np.ones((1, ww), np.uint8)).astype(np.uint8)
        elif kind == 3:
            s = int(rng.integers(2, 16))
            yy, xx = np.mgrid[0:hh, 0:ww]
            patch = ((((yy // s) + (xx // s)) & 1) * 255).astype(np.uint8)
        elif kind == 4:
            patch = np.linspace(0, 255, ww)[None, :].repeat(hh, 0).astype(np.uint8)
        else:
            patch = rng.integers(0, 256, (hh, ww), dtype=np.uint8)
        out[y0:y1, x0:x1] = patch[:, :, None]
    return out


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    fam = get_family()
    det = _lib.Detector("tagStandard41h12", id_limit=0)
    t0 = time.time()
    ntags = 0
    for i in range(nf):
        w, h = int(rng.integers(160, 1400)), int(rng.integers(120, 900))
        tags = synth.random_scene(w, h, int(rng.integers(1, 6)), rng)
        f, _ = synth.render_frame(w, h, tags, 18.0, cam_position=tuple(rng.uniform(-2, 2, 3)), cam_rotation_deg=tuple(rng.uniform(-3, 3, 3)))
        f = clutter(f, rng)
        d = det.detect_host(f[None])[0]
        ref = O.detect_gray(O.bgr2gray(f), fam, 2)
        ok = [int(x["id"]) for x in d] == [r["id"] for r in ref] and all(
            int(x["hamming"]) == r["hamming"] and np.abs(x["corners"] - r["corners"]).max() <= 1e-9 for x, r in zip(d, ref))
        if not ok:
            print("MISMATCH frame %d: %dx%d seed %d" % (i, w, h, seed))
            print(" gpu", [(int(x["id"]), int(x["hamming"])) for x in d]); print(" cpu", [(r["id"], r["hamming"]) for r in ref])
            sys.exit(1)
        ntags += len(ref)
        c = det.debug_counters()
        if c[11] or c[12] or c[13] or c[14]:
            print("overflow left after frame %d:" % i, c[11:15]); sys.exit(1)
    print("OK: %d cluttered frames, %d tags identical, %.0f s" % (nf, ntags, time.time() - t0))


if __name__ == "__main__":
    main()
