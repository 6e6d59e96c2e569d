#!/usr/bin/env python3
"""BASELINE.json configs[2]: 1920x1080, 50 tags/frame, detection + PnP + pose-graph Gauss-Newton on one MI355X.

Renders a seeded stream (moving camera), runs detect + PnP on the GPU, chains the per-observation poses from the
lowest tag id (the reference's world convention; aprilslam_amd.map_init) to get initial tag/camera poses, lets every
camera / tag pick the most consistent pose its observations imply, then refines everything with the device back-end (asl_gn_solve) and reports map / trajectory error against the renderer's ground truth before
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

from aprilslam_amd import _lib, map_init, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tags", type=int, default=50)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--verbose", type=int, default=0, help="per-tag errors on stderr")
    ap.add_argument("--both-minima", type=int, default=1, help="PnP tries the mirrored start too (asl_detector_set_pnp_both_minima)")
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

    det = _lib.Detector(id_limit=0)
    det.set_pnp_both_minima(bool(args.both_minima))
    d_frames = torch.from_numpy(frames).to("cuda:0")
    det.detect_device(d_frames.data_ptr(), P, 3, W, H, K=K, dist=np.zeros(4), tag_size=10.0)  # warm-up / allocation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dets, poses, npf = det.detect_device(d_frames.data_ptr(), P, 3, W, H, K=K, dist=np.zeros(4), tag_size=10.0)
    t_det = time.perf_counter() - t0
    dets, poses, npf = dets.copy(), poses.copy(), npf.copy()

    # starting values: chain the per-observation PnP poses from the lowest tag id (map_init; the reference's own graph
    # keeps stale nodes after a world switch), then let every camera / tag pick the most consistent of the poses its
    # observations imply
    per_frame, start = [], 0
    obs_cam, obs_tag, obs_corners, obs_T = [], [], [], []
    for f in range(P):
        n = int(npf[f])
        fr = []
        for k in range(start, start + n):
            if poses["ok"][k]:
                fr.append((int(dets["id"][k]), poses["T"][k].reshape(4, 4), dets["corners"][k].reshape(4, 2)))
                obs_cam.append(f); obs_tag.append(int(dets["id"][k])); obs_corners.append(dets["corners"][k]); obs_T.append(poses["T"][k].reshape(4, 4))
        per_frame.append(fr)
        start += n
    world, placed, cams = map_init.chain_initial_map(per_frame)
    tag0 = np.array([placed.get(j, np.eye(4)) for j in range(NT)])
    cam0 = np.array([c if c is not None else np.eye(4) for c in cams])
    seen = sorted(placed.keys())
    G0w = np.linalg.inv(tag_gt[world])
    tag_gt0 = np.array([G0w @ T for T in tag_gt])
    cam_gt0 = np.array([G0w @ T for T in cam_gt])
    chain_tag, chain_cam = None, None

    def rmse(est, gt, idx):
        dt = [np.linalg.norm(est[i][:3, 3] - gt[i][:3, 3]) for i in idx]
        ang = [np.arccos(np.clip((np.trace(est[i][:3, :3] @ gt[i][:3, :3].T) - 1) / 2, -1, 1)) for i in idx]
        return float(np.sqrt(np.mean(np.square(dt)))), float(np.sqrt(np.mean(np.square(ang))) * 1e3)

    def aligned(tag_est, cam_est):
        """The gauge (tag `world` = identity) puts the world tag's own pose error into every other pose; the usual
        trajectory metric removes it with the best rigid fit of the estimated tag centres onto the true ones."""
        A = np.array([tag_est[j][:3, 3] for j in seen]); B = np.array([tag_gt0[j][:3, 3] for j in seen])
        ca, cb = A.mean(0), B.mean(0)
        U, _, Vt = np.linalg.svd((B - cb).T @ (A - ca))
        D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
        M = np.eye(4); M[:3, :3] = U @ D @ Vt; M[:3, 3] = cb - M[:3, :3] @ ca
        return np.array([M @ T for T in tag_est]), np.array([M @ T for T in cam_est])

    def med_rot(est, gt, idx):
        return float(np.median([np.arccos(np.clip((np.trace(est[i][:3, :3] @ gt[i][:3, :3].T) - 1) / 2, -1, 1)) for i in idx]) * 1e3)

    chain_tag, chain_cam = rmse(tag0, tag_gt0, seen), rmse(cam0, cam_gt0, range(P))
    t0 = time.perf_counter()
    cam0, tag0 = map_init.reseed_poses(cam0, tag0, obs_cam, obs_tag, obs_T, obs_corners, K, 10.0, fixed_tag=world)
    t_seed = time.perf_counter() - t0
    before_tag, before_cam = rmse(tag0, tag_gt0, seen), rmse(cam0, cam_gt0, range(P))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cam1, tag1, st = det.gn_solve(cam0, tag0, obs_cam, obs_tag, np.array(obs_corners), K, 10.0, fixed_tag=world, iters=args.iters)
    t_gn = time.perf_counter() - t0
    # tags that sit in the mirrored planar-PnP minimum in every view: test the mirrored pose against all views, polish again
    t0 = time.perf_counter()
    tag1f, flipped = map_init.flip_test_tags(cam1, tag1, obs_cam, obs_tag, obs_corners, K, 10.0, fixed_tag=world)
    t_flip = time.perf_counter() - t0
    st2 = st
    if flipped:
        cam1, tag1, st2 = det.gn_solve(cam1, tag1f, obs_cam, obs_tag, np.array(obs_corners), K, 10.0, fixed_tag=world, iters=args.iters)
    after_tag, after_cam = rmse(tag1, tag_gt0, seen), rmse(cam1, cam_gt0, range(P))
    tag1a, cam1a = aligned(tag1, cam1)
    al_tag, al_cam = rmse(tag1a, tag_gt0, seen), rmse(cam1a, cam_gt0, range(P))
    mm = 5.56
    if args.verbose:
        nobs = np.bincount(np.array(obs_tag), minlength=NT)
        rows = []
        for j in seen:
            ang = np.arccos(np.clip((np.trace(tag1a[j][:3, :3] @ tag_gt0[j][:3, :3].T) - 1) / 2, -1, 1))
            ang0 = np.arccos(np.clip((np.trace(tag0[j][:3, :3] @ tag_gt0[j][:3, :3].T) - 1) / 2, -1, 1))
            rows.append((ang * 1e3, ang0 * 1e3, np.linalg.norm(tag1a[j][:3, 3] - tag_gt0[j][:3, 3]) * mm, int(nobs[j]), j))
        rows.sort(reverse=True)
        for r in rows[:12]:
            print("tag %3d  obs %2d  rot after %8.2f mrad (before %8.2f)  trans %7.2f mm" % (r[4], r[3], r[0], r[1], r[2]), file=sys.stderr)
        a = np.array([r[0] for r in rows])
        print("rot mrad percentiles 50/90/99/max:", np.percentile(a, [50, 90, 99, 100]), file=sys.stderr)
        cr = [np.arccos(np.clip((np.trace(cam1a[i][:3, :3] @ cam_gt0[i][:3, :3].T) - 1) / 2, -1, 1)) * 1e3 for i in range(P)]
        ct = [np.linalg.norm(cam1a[i][:3, 3] - cam_gt0[i][:3, 3]) * mm for i in range(P)]
        print("camera rot mrad:", np.round(cr, 2), file=sys.stderr)
        print("camera trans mm:", np.round(ct, 2), file=sys.stderr)
    print(json.dumps({
        "workload": "configs[2]: %dx%d, %d tags/frame, detect + PnP + pose-graph LM, %d frames" % (W, H, NT, P),
        "detect_pnp_frames_per_s": P / t_det, "tags_found": int(len(dets)), "observations": len(obs_cam),
        "gn_iterations": args.iters, "gn_ms_per_iteration": 1e3 * t_gn / max(args.iters, 1), "gn_cost": [st[0], st[1]], "gn_steps_accepted": int(st[2]),
        "flip_test": {"tags_flipped": [int(j) for j in flipped], "host_ms": 1e3 * t_flip, "gn_cost_after": float(st2[1])},
        "world_tag": int(world), "reseed_ms": 1e3 * t_seed,
        "tag_pose_rmse_chained": {"translation_mm": chain_tag[0] * mm, "rotation_mrad": chain_tag[1]},
        "camera_pose_rmse_chained": {"translation_mm": chain_cam[0] * mm, "rotation_mrad": chain_cam[1]},
        "tag_pose_rmse_before": {"translation_mm": before_tag[0] * mm, "rotation_mrad": before_tag[1]},
        "tag_pose_rmse_after": {"translation_mm": after_tag[0] * mm, "rotation_mrad": after_tag[1]},
        "camera_pose_rmse_before": {"translation_mm": before_cam[0] * mm, "rotation_mrad": before_cam[1]},
        "camera_pose_rmse_after": {"translation_mm": after_cam[0] * mm, "rotation_mrad": after_cam[1]},
        "tag_pose_rmse_after_aligned": {"translation_mm": al_tag[0] * mm, "rotation_mrad": al_tag[1], "rotation_median_mrad": med_rot(tag1a, tag_gt0, seen)},
        "camera_pose_rmse_after_aligned": {"translation_mm": al_cam[0] * mm, "rotation_mrad": al_cam[1]},
    }))


if __name__ == "__main__":
    main()
