#!/usr/bin/env python3
"""Diagnostic: isolated per-kernel times of one detector batch on the bench scene, median over several
synchronous batches, plus a checksum of the results (so that two builds can be compared for identical output).

    ASL_LIB=build/libaprilslam_x.so python tools/stage_lab.py [--batch 1024] [--reps 9] [--decimate 2] [--tag NAME]

Prints one JSON line.  Nothing else runs on the GPU meanwhile, so the figures are the `*_isolated` ones of bench.py."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from aprilslam_amd import _lib, synth  # noqa: E402

SEG = ("k_hash_clear", "k_decimate_minmax", "k_tile_cut", "k_seg_tile", "k_seg_border", "k_seg_roots", "k_seg_points", "k_cluster_filter",
       "k_point_place")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--decimate", type=int, default=2)
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("ASL_LIB", "libaprilslam.so")))
    ap.add_argument("--blank", action="store_true", help="uniform frames: the floor of every kernel (no contrast anywhere)")
    ap.add_argument("--phases", action="store_true", help="print the phase-cycle shares (needs a -DASL_PHASE_TIMING build)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = args.batch
    det = _lib.Detector(id_limit=0, decimate=float(args.decimate))
    d_frames, gts, _ = bench.render_stream_device(det, B, dev)
    if args.blank:
        d_frames.fill_(128)
    det.set_profiling(True)
    K = synth.camera_matrix(bench.W, bench.H)
    st = torch.cuda.current_stream(dev).cuda_stream
    times = {}
    digest = None
    for it in range(args.reps + 2):
        if args.phases and it == 2:
            det.phase_cycles(reset=True)
        det.submit_device(d_frames.data_ptr(), B, 3, bench.W, bench.H, stream=st, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
        dets, poses, npf = det.collect(max_per_frame=bench.MAXDET)
        if it < 2:
            continue
        for k, v in det.stage_times().items():
            times.setdefault(k, []).append(v)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(dets).tobytes()); h.update(np.ascontiguousarray(npf).tobytes())
        hp = hashlib.sha256(np.ascontiguousarray(poses).tobytes()).hexdigest()[:16]
        d = h.hexdigest()[:16]
        if digest is None:
            digest = (d, hp)
        elif digest != (d, hp):
            digest = (digest[0] + "!" + d, digest[1] + "!" + hp)  # run-to-run difference: a race
    med = {k: float(np.median(v)) for k, v in times.items()}
    mn = {k: float(np.min(v)) for k, v in times.items()}
    kern = {k: v for k, v in med.items() if k.startswith("k_")}
    seg = sum(med.get(k, 0.0) for k in SEG)
    seg_bytes = bench.stage_algorithmic_read_bytes(bench.W, bench.H, 3, args.decimate)
    out = {"tag": args.tag, "batch": B, "decimate": args.decimate, "n_dets": int(len(dets)), "digest_dets": digest[0], "digest_poses": digest[1],
           "stage_ms": round(seg, 4), "stage_frac": round(seg_bytes * B / (seg * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4),
           "all_kernels_ms": round(sum(kern.values()), 4),
           "median_ms": {k: round(v, 4) for k, v in med.items()}, "min_ms": {k: round(v, 4) for k, v in mn.items()},
           "counters": det.debug_counters().tolist()}
    print(json.dumps(out))
    if args.phases:
        cyc = det.phase_cycles()
        nz = {int(i): int(c) for i, c in enumerate(cyc) if c}
        print(json.dumps({"phase_cycles": nz}))


if __name__ == "__main__":
    main()
