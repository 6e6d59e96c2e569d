#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle shares of the per-cluster / per-quad kernels.
Needs a library built with -DASL_PHASE_TIMING:  ASL_LIB=build/libaprilslam_timing.so python tools/phase_timing.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from aprilslam_amd import _lib, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = bench.make_frames(16)
t = torch.from_numpy(frames).to("cuda:0").repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
det = _lib.Detector(id_limit=0)
det.set_profiling(True)
K = synth.camera_matrix(bench.W, bench.H)
for it in range(3):
    det.phase_cycles(reset=True)
    det.detect_device(t.data_ptr(), B, 3, bench.W, bench.H, K=K, dist=np.zeros(4), tag_size=10.0)
cyc = det.phase_cycles()
names = {0: "fit: bbox/polarity", 1: "fit: keys+sort", 2: "fit: dedup compact", 3: "fit: weights", 4: "fit: moment scan",
         5: "fit: errs+smooth", 6: "fit: maxima select", 7: "fit: combos", 8: "fit: final",
         16: "ref: load", 17: "ref: edges", 18: "dec: load H", 19: "dec: border+graymodel", 20: "dec: bits+sharpen",
         21: "dec: code book", 22: "ref: homography", 23: "dec: emit", 24: "ref: normals", 25: "ref: probes fetch+park", 26: "ref: steps+centroid", 27: "ref: ordered line fit", 32: "pts: masks+run list", 33: "pts: run labels", 34: "pts: items -> tile table", 37: "pts: site masks + item list", 35: "pts: global table",
         36: "pts: write out", 50: "tile: threshold", 56: "tile: masks+edges out", 51: "tile: links+runs listed", 57: "tile: unions", 58: "tile: finds, sizes, roots", 59: "tile: parents out", 60: "tile: links (count)", 61: "tile: runs (count)", 62: "tile: tiles with contrast (count)"}
for grp in ((0, 9), (16, 18), (24, 28), (22, 23), (18, 22), (23, 24), (32, 38), (50, 52), (56, 60)):
    tot = float(cyc[grp[0]:grp[1]].sum()) or 1.0
    for k in range(*grp):
        print("%-24s %14d cycles  %5.1f%%" % (names.get(k, str(k)), int(cyc[k]), 100.0 * float(cyc[k]) / tot))
for k in (60, 61, 62):
    print("%-32s %14d" % (names[k], int(cyc[k])))
print("counters", det.debug_counters().tolist())
print({k: round(v, 3) for k, v in det.stage_times().items()})
