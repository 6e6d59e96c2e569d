#!/usr/bin/env python3
"""Diagnostic: frames/s of the host-frame entry (asl_detect_batch_pose_u8) against the bare PCIe copy, for a few chunk sizes
(ASL_HOST_CHUNK is read per call)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from aprilslam_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
det = _lib.Detector(id_limit=0)
d_frames, _, _ = bench.render_stream_device(det, n, dev)
K = synth.camera_matrix(bench.W, bench.H)
host = d_frames.cpu().pin_memory()
a = host.numpy()
dst = torch.empty_like(d_frames)
for rep in range(2):
    dst.copy_(host, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); dst.copy_(host, non_blocking=True); torch.cuda.synchronize(); tc = time.perf_counter() - t0
print("bare copy %.2f ms  %.1f GB/s" % (1e3 * tc, a.nbytes / tc / 1e9))
for chunk in (10 ** 9, 32, 64, 128, 256):
    os.environ["ASL_HOST_CHUNK"] = str(chunk)
    det.detect_host(a, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        dets, poses, npf = det.detect_host(a, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print("chunk %10d: %.2f ms  %.0f frames/s  frac_of_link %.3f  dets %d" % (chunk, 1e3 * t, n / t, tc / t, len(dets)))
