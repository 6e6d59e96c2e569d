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

# the read-back of a world = 8 step, piece by piece (wall clock around a synchronize each)
import time  # noqa: E402

world = 8
block = obs[None].repeat(world, 1, 1, 1).contiguous()
pose = torch.zeros((world * B, 16), dtype=torch.float64, device=dev)
status = torch.zeros(world * B, dtype=torch.uint8, device=dev)
last = torch.zeros(adist.MAX_IDS, dtype=torch.int32, device=dev)
picks = torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
h_pose, h_status = torch.zeros((world * B, 16), dtype=torch.float64).pin_memory(), torch.zeros(world * B, dtype=torch.uint8).pin_memory()
h_last = torch.zeros(adist.MAX_IDS, dtype=torch.int32).pin_memory()
h_picks = torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
h_tail = torch.zeros((world, MT, adist.OBS_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
print("bytes: pose %d status %d last %d picks %d tail %d" % (h_pose.numel() * 8, h_status.numel(), h_last.numel() * 4, h_picks.numel(), h_tail.numel()))


def timed(name, fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print("  %-28s %.3f ms" % (name, min(ts)))


timed("last.zero_", lambda: last.zero_())
timed("graph_frames_device", lambda: det.graph_frames_device(block.data_ptr(), world, B, MT, 0, pose.data_ptr(), status.data_ptr(), last.data_ptr(), adist.MAX_IDS,
                                                             picks_ptr=picks.data_ptr(), stream=st))
timed("pose -> host", lambda: h_pose.copy_(pose, non_blocking=True))
timed("status -> host", lambda: h_status.copy_(status, non_blocking=True))
timed("last -> host", lambda: h_last.copy_(last, non_blocking=True))
timed("picks -> host", lambda: h_picks.copy_(picks, non_blocking=True))
timed("tail -> host", lambda: h_tail.copy_(block[:, B - 1], non_blocking=True))
