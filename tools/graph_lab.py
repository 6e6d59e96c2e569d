#!/usr/bin/env python3
"""Diagnostic: time of asl_graph_frames_device (k_graph_frames + k_graph_pick) on a gathered block of `world` ranks,
one rank's records replicated, HIP events on the stream."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from aprilslam_amd import _lib, synth  # noqa: E402
from aprilslam_amd import dist as adist  # noqa: E402

B = 1024
MT = 24
dev = torch.device("cuda", 0)
det = _lib.Detector(id_limit=0)
d_frames, _, _ = bench.render_stream_device(det, B, dev)
K = synth.camera_matrix(bench.W, bench.H)
st = torch.cuda.current_stream(dev).cuda_stream
obs = torch.empty((B, MT, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
det.submit_device(d_frames.data_ptr(), B, 3, bench.W, bench.H, stream=st, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
det.pack_observations_device(obs.data_ptr(), MT, stream=st)
det.collect_view()
for world in (1, 2, 4, 8):
    block = obs[None].repeat(world, 1, 1, 1).contiguous()
    pose = torch.zeros((world * B, 16), dtype=torch.float64, device=dev)
    status = torch.zeros(world * B, dtype=torch.uint8, device=dev)
    last = torch.zeros(adist.MAX_IDS, dtype=torch.int32, device=dev)
    picks = torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    ts = []
    for rep in range(6):
        last.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        det.graph_frames_device(block.data_ptr(), world, B, MT, 0, pose.data_ptr(), status.data_ptr(), last.data_ptr(), adist.MAX_IDS,
                                picks_ptr=picks.data_ptr(), stream=st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("world %d: %d frames, graph_frames_device %.3f ms (min of 5 after warm-up), steady frames %d" % (world, world * B, min(ts[1:]), int((status == 0).sum())))
