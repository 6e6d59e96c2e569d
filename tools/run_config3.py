#!/usr/bin/env python3
"""BASELINE.json configs[2]: 1920x1080, 50 tags/frame, detection + PnP + pose-graph Gauss-Newton on one MI355X.

Renders a seeded stream (moving camera), runs detect + PnP on the GPU, chains the tag graph exactly like the
reference (SLAMGraph, world = lowest tag id) to get initial tag/camera poses, then refines everything with the
device back-end (asl_gn_solve) and reports map / trajectory error against the renderer's ground truth before
and after.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from aprilslam_amd import _lib, synth  # noqa: E402
from aprilslam_amd.slam import SLAM  # noqa: E402


class _Log:
    def info(self, m):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tags", type=int, default=50)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    W, H, NT, P = args.width, args.height, args.tags, args.frames
    rng = np.random.default_rng(20250620 + 2)
    tags = synth.random_scene(W, H, NT, rng)
    K = synth.camera_matrix(W, H)
    frames, cam_gt = [], []
    for i in range(P):
        a = 2 * np.pi * i / P
        pos = (4 * np.cos(a), 3 * np.sin(a), 5 * np.sin(2 * a))
        rot = (1.5 * np.sin(a), 2.0 * np.cos(a), 1.0 * np.sin(3 * a))
        f, _ = synth.render_frame(W, H, tags, 18.0, cam_position=pos, cam_rotation_deg=rot)
        frames.append(f)
        V = synth.view_matrix(pos, rot)
        Wf = np.eye(4)
        Wf[:3, :3] = synth._FLIP @ V[:3, :3]
        Wf[:3, 3] = synth._FLIP @ V[:3, 3]
        cam_gt.append(np.linalg.inv(Wf))  # GL-world <- camera (OpenCV camera axes)
    frames = np.stack(frames)
    tag_gt = np.array([synth.tag_model_matrix(t["position"], t["rotation"]) for t in tags])  # GL-world <- tag
    # express ground truth in the frame of tag 0 (the SLAM world)
    G0i = np.linalg.inv(tag_gt[0])
    tag_gt0 = np.array([G0i @ T for T in tag_gt])
    cam_gt0 = np.array([G0i @ T for T in cam_gt])

    det = _lib.Detector(id_limit=0)
    d_frames = torch.from_numpy(frames).to("cuda:0")
    det.detect_device(d_frames.data_ptr(), P, 3, W, H, K=K, dist=np.zeros(4), tag_size=10.0)  # warm-up / allocation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dets, poses, npf = det.detect_device(d_frames.data_ptr(), P, 3, W, H, K=K, dist=np.zeros(4), tag_size=10.0)
    t_det = time.perf_counter() - t0
    dets, poses, npf = dets.copy(), poses.copy(), npf.copy()

    # reference-style graph initialisation (SLAMGraph), frame by frame
    slam = SLAM(_Log(), {"camera_matrix": K, "dist_coeffs": np.zeros(4)}, detector=object())
    cam0, start = [], 0
    obs_cam, obs_tag, obs_corners = [], [], []
    for f in range(P):
        n = int(npf[f])
        ids = [int(x) for x in dets["id"][start:start + n]]
        pose = slam.process_observations(ids, poses["T"][start:start + n], poses["ok"][start:start + n])
        cam0.append(pose if pose is not None else np.eye(4))
        for k in range(n):
            obs_cam.append(f); obs_tag.append(ids[k]); obs_corners.append(dets["corners"][start + k])
        start += n
    nodes = slam.graph.get_nodes()
    tag0 = np.array([nodes[j].world if j in nodes else np.eye(4) for j in range(NT)])
    cam0 = np.array(cam0)
    # the reference's element-wise matrix average is not a rotation: project to SO(3) before refining
    for T in cam0:
        U, _, Vt = np.linalg.svd(T[:3, :3])
        T[:3, :3] = U @ Vt
        T[3] = [0, 0, 0, 1]

    def rmse(est, gt, idx):
        dt = [np.linalg.norm(est[i][:3, 3] - gt[i][:3, 3]) for i in idx]
        ang = [np.arccos(np.clip((np.trace(est[i][:3, :3] @ gt[i][:3, :3].T) - 1) / 2, -1, 1)) for i in idx]
        return float(np.sqrt(np.mean(np.square(dt)))), float(np.sqrt(np.mean(np.square(ang))) * 1e3)

    seen = sorted(nodes.keys())
    before_tag, before_cam = rmse(tag0, tag_gt0, seen), rmse(cam0, cam_gt0, range(P))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cam1, tag1, st = det.gn_solve(cam0, tag0, obs_cam, obs_tag, np.array(obs_corners), K, 10.0, fixed_tag=slam.coordinate_id, iters=args.iters)
    t_gn = time.perf_counter() - t0
    after_tag, after_cam = rmse(tag1, tag_gt0, seen), rmse(cam1, cam_gt0, range(P))
    mm = 5.56
    print(json.dumps({
        "workload": "configs[2]: %dx%d, %d tags/frame, detect + PnP + pose-graph LM, %d frames" % (W, H, NT, P),
        "detect_pnp_frames_per_s": P / t_det, "tags_found": int(len(dets)), "observations": len(obs_cam),
        "gn_iterations": args.iters, "gn_ms_per_iteration": 1e3 * t_gn / max(args.iters, 1), "gn_cost": [st[0], st[1]], "gn_steps_accepted": int(st[2]),
        "tag_pose_rmse_before": {"translation_mm": before_tag[0] * mm, "rotation_mrad": before_tag[1]},
        "tag_pose_rmse_after": {"translation_mm": after_tag[0] * mm, "rotation_mrad": after_tag[1]},
        "camera_pose_rmse_before": {"translation_mm": before_cam[0] * mm, "rotation_mrad": before_cam[1]},
        "camera_pose_rmse_after": {"translation_mm": after_cam[0] * mm, "rotation_mrad": after_cam[1]},
    }))


if __name__ == "__main__":
    main()
