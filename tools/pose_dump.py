#!/usr/bin/env python3
"""Diagnostic: rvec | tvec of every tag of a 256-frame bench batch to an .npy file (two builds can be compared number by number):
    ASL_LIB=build/libaprilslam_x.so python tools/pose_dump.py gpurun_out/x.npy"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from aprilslam_amd import _lib, synth
det = _lib.Detector(id_limit=0, decimate=2.0)
dev = torch.device("cuda", 0)
B = 256
d_frames, _, _ = bench.render_stream_device(det, B, dev)
K = synth.camera_matrix(bench.W, bench.H)
st = torch.cuda.current_stream(dev).cuda_stream
det.submit_device(d_frames.data_ptr(), B, 3, bench.W, bench.H, stream=st, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
dets, poses, npf = det.collect(max_per_frame=bench.MAXDET)
np.save(sys.argv[1], np.concatenate([poses["rvec"], poses["tvec"]], axis=1))
