#!/usr/bin/env python3
"""Headline benchmark: frames/s of the AprilTag hot path at 1280x720, 20 tags/frame
(BASELINE.json configs[1]: detection + batched PnP on one MI355X), frames resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--batch B]

A step = one batch of B synthetic BGR frames through asl_detect_batch_device (threshold -> components ->
clusters -> quads -> decode -> PnP -> de-duplication, results copied back).  With --gpus N > 1 the script starts N
ranks itself (or runs as one rank of a torchrun launch); every rank runs its own stream of B frames per step (weak
scaling), and inside the timed region the ranks all-gather their observation records (packed on the device, one RCCL
all_gather_into_tensor), apply the gathered block to the tag graph (k_graph_frames + the last-sighting update) and
every --gn-every steps run the pose-graph LM on a window -- the serial term the scaling curve pays for.
Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NTAGS = 1280, 720, 20  # --workload replaces them (main)
# BASELINE.json configs the harness can run; "configs1" is the one the metric is quoted on (and the default)
WORKLOADS = {
    "configs1": {"w": 1280, "h": 720, "tags": 20, "batch": 1024, "partition": "streams", "exchange": False, "gn_every": 50,
                 "name": "configs[1]: 1280x720 BGR stream, 20 tags/frame, detect + PnP"},
    "configs2": {"w": 1920, "h": 1080, "tags": 50, "batch": 384, "partition": "streams", "exchange": True, "gn_every": 8,
                 "name": "configs[2]: 1920x1080, 50 tags/frame, detect + PnP + pose-graph LM"},
    "configs3": {"w": 1280, "h": 720, "tags": 20, "batch": 1024, "partition": "streams", "exchange": True, "gn_every": 50,
                 "name": "configs[3]: one 1280x720 stream per GPU, all-gather of observations before the global solve"},
    "configs4": {"w": 3840, "h": 2160, "tags": 200, "batch": 128, "partition": "frames", "exchange": True, "gn_every": 8,
                 "name": "configs[4]: 3840x2160 dense 200-tag scene, frame i on GPU i mod N, graph update + pose-graph LM"},
}
TAG_OUTER, TAG_INNER = 18.0, 10.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MAXDET = 64  # detections per frame the result buffers hold (set from the workload)


def make_frames(n_distinct, seed=20250620 + 1, with_gt=False, phase=0.0):
    """n_distinct frames of one seeded 20-tag scene seen from a smooth camera trajectory.  `phase` shifts the camera
    along the trajectory: the streams of a multi-GPU run look at the SAME tags from different places, so that the
    gathered observations describe one map."""
    from aprilslam_amd import synth
    rng = np.random.default_rng(seed)
    tags = synth.random_scene(W, H, NTAGS, rng, tag_size_outer=TAG_OUTER)
    frames, gts = [], []
    for i in range(n_distinct):
        a = 2 * np.pi * i / max(n_distinct, 1) + phase
        pos = (1.5 * np.cos(a), 1.0 * np.sin(a), 2.0 * np.sin(2 * a))
        rot = (0.6 * np.sin(a), 0.8 * np.cos(a), 0.5 * np.sin(3 * a))
        f, gt = synth.render_frame(W, H, tags, TAG_OUTER, cam_position=pos, cam_rotation_deg=rot)
        frames.append(f)
        gts.append(gt)
    if with_gt:
        return np.stack(frames), gts
    return np.stack(frames)


def camera_trajectory(n, phase=0.0):
    """n camera poses (position, rotation in degrees) along a smooth closed curve"""
    cams = []
    for i in range(n):
        a = 2 * np.pi * i / max(n, 1) + phase
        cams.append(((1.5 * np.cos(a), 1.0 * np.sin(a), 2.0 * np.sin(2 * a)), (0.6 * np.sin(a), 0.8 * np.cos(a), 0.5 * np.sin(3 * a))))
    return cams


def render_stream_device(det, n_frames, dev, phase=0.0, seed=20250620 + 1, take=None):
    """n_frames DISTINCT frames of the seeded 20-tag scene, rendered on the device (asl_render_frames_device: the
    reference renderer's image formation as a kernel, byte-identical to aprilslam_amd.synth.render_frame) straight into
    HBM.  Returns (uint8 tensor (n_frames, H, W, 3), per-frame ground truth {id: camera<-tag})."""
    import torch
    from aprilslam_amd import synth
    rng = np.random.default_rng(seed)
    tags = synth.random_scene(W, H, NTAGS, rng, tag_size_outer=TAG_OUTER)
    # take = (rank, world): this rank's frames are every world-th pose of ONE stream (frame i on GPU i mod N)
    cams = camera_trajectory(n_frames, phase) if take is None else camera_trajectory(n_frames * take[1])[take[0]::take[1]]
    planes, gts = synth.render_planes(W, H, tags, TAG_OUTER, cams)
    tex = synth.gray_textures([int(t["id"]) for t in tags])
    d_tex = torch.from_numpy(tex).to(dev)
    d_planes = torch.from_numpy(planes.view(np.uint8).reshape(planes.shape + (-1,))).to(dev)
    out = torch.empty((n_frames, H, W, 3), dtype=torch.uint8, device=dev)
    det.render_frames_device(out.data_ptr(), n_frames, W, H, d_planes.data_ptr(), planes.shape[1], d_tex.data_ptr(), tex.shape[2], tex.shape[1],
                             0.5 * TAG_OUTER, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)

    def rerender(det2, stream):
        """the same frames again, enqueued on `stream` (the `render_included` leg: frames produced on the device per step)"""
        det2.render_frames_device(out.data_ptr(), n_frames, W, H, d_planes.data_ptr(), planes.shape[1], d_tex.data_ptr(), tex.shape[2], tex.shape[1],
                                  0.5 * TAG_OUTER, stream=stream)
    return out, gts, rerender


def algorithmic_bytes(kernel, w, h, ch, f, runs=7200, points=18600):
    """Compulsory HBM bytes of one launch PER FRAME (reads + writes of the kernel's operands, each counted once).
    DESIGN.md section 'Kernels' derives these.  `runs` / `points` = runs of one colour inside a 64-pixel word and staged
    boundary points of the bench scene (per frame)."""
    sw, sh = 1 + (w - 1) // f, 1 + (h - 1) // f
    npix = sw * sh
    ntile = (sw // 4) * (sh // 4)
    rows_read = sh * w * ch  # only every f-th row of the input is touched (whole rows: 64-byte lines)
    table = {
        "k_decimate_minmax": rows_read + npix + 2 * ntile,
        "k_tile_cut": 2 * ntile + ntile,
        "k_seg_tile": npix + ntile + 2 * npix // 8 + npix // 8 + 4 * runs,              # gray + cuts in; two bit masks + root mask + run labels out
        "k_seg_points": 2 * npix // 8 + 12 * runs + 12 * points,                        # masks + (label, root, size) per run in; 12 B per staged point out
        "k_point_place": 12 * points + 8 * points,
    }
    return table.get(kernel)


def stage_algorithmic_read_bytes(w, h, ch, f):
    """R = W*H*(1 + 5/d^2) (+ 2*W*H when BGR->gray is inside the timed region)."""
    return w * h * (1 + 5.0 / (f * f)) + (2 * w * h if ch == 3 else 0)


MM_PER_UNIT = 55.6 / 10.0  # reference config: actual_size_in_mm 55.6 for tag_size_inner 5 * size_scale 2 (config_manager.py:199-209)


def pose_errors(T_est, T_ref):
    """translation error in sim units and geodesic rotation angle in rad between 4x4 transforms"""
    dt = np.linalg.norm(T_est[:3, 3] - T_ref[:3, 3])
    R = T_est[:3, :3] @ T_ref[:3, :3].T
    ang = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    return dt, ang


def pose_rmse_vs_ground_truth(dets, poses, npf, gts):
    """camera<-tag pose RMSE of the first len(gts) frames of a batch against the renderer's analytic ground truth
    (reference src/simulation/ground_truth.py:48-90)."""
    dt, da, start, missing = [], [], 0, 0
    for f in range(len(gts)):
        n = int(npf[f])
        seen = set()
        for k in range(start, start + n):
            tid = int(dets["id"][k])
            if tid in gts[f] and poses["ok"][k]:
                a, b = pose_errors(poses["T"][k], gts[f][tid])
                dt.append(a); da.append(b); seen.add(tid)
        missing += len(set(gts[f]) - seen)
        start += n
    dt, da = np.array(dt), np.array(da)
    return {"translation_mm": float(np.sqrt((dt ** 2).mean()) * MM_PER_UNIT), "rotation_mrad": float(np.sqrt((da ** 2).mean()) * 1e3),
            "tags": int(len(dt)), "tags_missed": int(missing),
            "note": "vs analytic ground truth of the synthetic renderer; includes the detector's corner noise"}


CPU_FLAGS = "-O3 -march=native -ffp-contract=off (oracle/Makefile: native)"


def build_native_oracle():
    """The CPU baseline's own build of oracle/ (-O3 -march=native), made on THIS host, outside the
    tree; oracle_lib loads it through ASO_SO.  Falls back to the -O2 parity build (and says so) if the compiler is missing."""
    import subprocess
    import tempfile
    global CPU_FLAGS
    if os.environ.get("ASO_SO"):
        return
    out = os.path.join(tempfile.gettempdir(), "liboracle_native_%d.so" % os.getuid())
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "NATIVE_OUT=" + out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["ASO_SO"] = out  # inherited by the worker processes of the all-cores leg
    except (OSError, subprocess.CalledProcessError):
        CPU_FLAGS = "-O2 -ffp-contract=off (oracle/liboracle.so; the native build failed on this host)"


def cpu_baseline(frames, K, budget_s=12.0, gpu=None):
    """The CPU restatement (oracle/, 1 thread) timed on this host on a bounded sample of the same workload.
    With gpu=(dets, poses, npf) of the same frames it also reports how far the GPU results are from it."""
    build_native_oracle()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from aprilslam_amd.families import get_family
    fam = get_family()
    n = 0
    t0 = time.perf_counter()
    while True:
        fr = frames[n % len(frames)]
        dets = O.detect_bgr(fr, fam)
        if dets:
            O.solve_pnp(np.stack([d["corners"] for d in dets]), K, np.zeros(4), TAG_INNER)
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 8:
            break
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same %dx%dx%d-tag stream, oracle/ detect_bgr + solve_pnp built %s, 1 thread, %.1f s" % (n, W, H, NTAGS, CPU_FLAGS, dt)}
    if gpu is not None:
        dets, poses, npf = gpu
        start, ids_equal, dcorner, dtr, drot = 0, True, 0.0, [], []
        for f in range(len(frames)):
            ref = O.detect_bgr(frames[f], fam)
            m = int(npf[f])
            mine = dets[start:start + m]
            if [int(x) for x in mine["id"]] != [r["id"] for r in ref]:
                ids_equal = False
            else:
                if ref:
                    rv, tv, T, ok = O.solve_pnp(np.stack([r["corners"] for r in ref]), K, np.zeros(4), TAG_INNER)
                for k, r in enumerate(ref):
                    dcorner = max(dcorner, float(np.abs(mine["corners"][k] - r["corners"]).max()))
                    a, b = pose_errors(poses["T"][start + k], T[k])
                    dtr.append(a); drot.append(b)
            start += m
        out["gpu_vs_cpu"] = {"frames": len(frames), "ids_identical": bool(ids_equal), "max_corner_diff_px": dcorner,
                             "pose_rmse_translation_mm": float(np.sqrt(np.mean(np.square(dtr))) * MM_PER_UNIT) if dtr else None,
                             "pose_rmse_rotation_mrad": float(np.sqrt(np.mean(np.square(drot))) * 1e3) if drot else None}
    return out


def spawn_ranks(n, argv, timeout_s=1500.0):
    """--gpus N without a launcher: N fresh children, one per GPU, started BEFORE this process touches the GPU; rank 0's
    stdout (the JSON line) is passed through, every rank's stderr (and the other ranks' stdout) goes to this process's
    stderr.  All children are watched: when one of them fails, or the time runs out, the others are terminated (a rank
    that dies before its first collective would otherwise leave its peers waiting in RCCL until their own timeout) and
    the exit code is non-zero."""
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t0, failed = time.monotonic(), None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % bad[0]
        elif time.monotonic() - t0 > timeout_s:
            failed = "no result after %.0f s" % timeout_s
        if failed or all(c is not None for c in codes):
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write("bench.py: %s; the other ranks were stopped\n" % failed)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode(errors="replace"))
    sys.stdout.flush()
    return 1 if failed else 0


def _cpu_worker(args):
    frames, K, budget = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from aprilslam_amd.families import get_family
    fam = get_family()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        dets = O.detect_bgr(frames[n % len(frames)], fam)
        if dets:
            O.solve_pnp(np.stack([d["corners"] for d in dets]), K, np.zeros(4), TAG_INNER)
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline_all_cores(frames, K, budget_s=10.0):
    """The CPU restatement on every host core at once, frame-parallel (one process per core, each its own frames)."""
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))  # the cores this process may use, not the ones the machine has
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    ctx = mp.get_context("spawn")  # fresh interpreters: no forked HIP state
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(frames[i % len(frames):] if len(frames) > 1 else frames, K, budget_s) for i in range(cores)])
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    rate = sum(r[0] / r[1] for r in res)
    return {"value": rate, "unit": "frames/s", "cores": cores, "cpu": model, "kind": "port",
            "sample": "%d frames in %.1f s wall (%d processes x %.0f s of the same %dx%dx%d-tag stream, oracle/ detect_bgr + solve_pnp built %s)" % (total, wall, cores, budget_s, W, H, NTAGS, CPU_FLAGS)}


def h2d_included(det, d_frames, K, n=512, reps=3):
    """frames/s when the frames start in (page-locked) HOST memory: asl_detect_batch_pose_u8 copies them over PCIe in chunks
    and runs each chunk's kernels under the next chunk's transfer.  Reported next to `value`, never as `value`.  The bare
    transfer of the same bytes is timed beside it: the link is the bound, `frac_of_link` says how close the call comes."""
    import torch
    n = min(n, d_frames.shape[0])
    host = d_frames[:n].cpu().pin_memory()
    a = host.numpy()
    det.detect_host(a, K=K, dist=np.zeros(4), tag_size=TAG_INNER)  # staging buffer allocation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        det.detect_host(a, K=K, dist=np.zeros(4), tag_size=TAG_INNER)
    dt = (time.perf_counter() - t0) / reps
    dst = torch.empty_like(d_frames[:n])
    dst.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dst.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    dt_copy = (time.perf_counter() - t0) / reps
    nbytes = n * W * H * 3
    return {"value": n / dt, "unit": "frames/s", "frames_per_call": n, "ms_per_call": 1e3 * dt,
            "host_to_device_GBs": nbytes / dt / 1e9, "link_GBs": nbytes / dt_copy / 1e9, "link_limited_frames_per_s": n / dt_copy,
            "frac_of_link": dt_copy / dt,
            "note": "asl_detect_batch_pose_u8 from page-locked host frames: PCIe copy in chunks of 64 frames, each chunk's kernels under the next chunk's transfer; link_GBs = one plain pinned-to-device copy of the same bytes"}


def stage_at_decimate_1(d_frames, K, nframes=256):
    """SURVEY.md 8(d) asks for the threshold + segmentation stage at decimate = 1 as well (R = 6 W H, + 2 W H for BGR in the
    timed region): one synchronous batch on a detector built with decimate = 1 (parity-tested: tests/test_gpu_parity.py)."""
    from aprilslam_amd import _lib
    n = min(nframes, d_frames.shape[0])
    det = _lib.Detector("tagStandard41h12", decimate=1.0, id_limit=0)
    try:
        det.set_profiling(True)
        for _ in range(2):  # the first call allocates the workspace
            det.submit_device(d_frames.data_ptr(), n, 3, W, H, stream=0, K=K, dist=np.zeros(4), tag_size=TAG_INNER)
            det.collect(max_per_frame=MAXDET)
        t = det.stage_times()
    finally:
        det.close()
    names = ("k_hash_clear", "k_decimate_minmax", "k_tile_cut", "k_seg_tile", "k_seg_border", "k_seg_roots", "k_seg_points", "k_cluster_filter", "k_point_place")
    ms = sum(t.get(k, 0.0) for k in names)
    rb = stage_algorithmic_read_bytes(W, H, 3, 1)
    gbs = rb * n / (ms * 1e-3) / 1e9
    return {"decimate": 1, "frames": n, "ms_per_batch_isolated": ms, "algorithmic_read_bytes_per_frame": rb, "achieved_GBs": gbs,
            "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "members_ms": {k: t.get(k, 0.0) for k in names}}


def project_serial_term(det, d_frames, K, dev, max_tags, world=8, reps=5):
    """The serial (Amdahl) part of a `world`-GPU step, measured on ONE GPU: this rank's packed observation block is
    replicated `world` times on the device (what the all-gather would deliver, minus the transfer itself), then the
    per-step work every rank does on the gathered block runs on it as in the multi-GPU step: asl_graph_frames_device over
    world x B frames, the read-back of its results, and dist.apply_block on the host.  A PROJECTION: the collective's own
    time (3.3 MB per rank over xGMI) is not in it."""
    import torch
    from aprilslam_amd import dist as adist
    from aprilslam_amd.slam import SLAM

    class _Log:
        def info(self, m):
            pass
    B = d_frames.shape[0]
    st = torch.cuda.current_stream(dev).cuda_stream
    obs = torch.empty((B, max_tags, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    det.submit_device(d_frames.data_ptr(), B, 3, W, H, stream=st, K=K, dist=np.zeros(4), tag_size=TAG_INNER)
    det.pack_observations_device(obs.data_ptr(), max_tags, stream=st)
    det.collect_view()
    block = obs[None].repeat(world, 1, 1, 1).contiguous()
    pose = torch.zeros((world * B, 16), dtype=torch.float64, device=dev)
    status = torch.zeros(world * B, dtype=torch.uint8, device=dev)
    last = torch.zeros(adist.MAX_IDS, dtype=torch.int32, device=dev)
    picks = torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    h_pose, h_status = torch.zeros((world * B, 16), dtype=torch.float64).pin_memory(), torch.zeros(world * B, dtype=torch.uint8).pin_memory()
    h_last = torch.zeros(adist.MAX_IDS, dtype=torch.int32).pin_memory()
    h_picks = torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
    h_tail = torch.zeros((world, max_tags, adist.OBS_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
    slam = SLAM(_Log(), {"camera_matrix": K, "dist_coeffs": np.zeros(4)}, tag_size=TAG_INNER, detector=object())
    t_dev, t_host, nseq = [], [], 0
    for it in range(reps + 2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res = None
        if slam.coordinate_id != -1:
            last.zero_()
            det.graph_frames_device(block.data_ptr(), world, B, max_tags, slam.coordinate_id, pose.data_ptr(), status.data_ptr(), last.data_ptr(),
                                    adist.MAX_IDS, picks_ptr=picks.data_ptr(), stream=st)
            h_pose.copy_(pose, non_blocking=True); h_status.copy_(status, non_blocking=True)
            h_last.copy_(last, non_blocking=True); h_picks.copy_(picks, non_blocking=True)
            h_tail.copy_(block[:, B - 1], non_blocking=True)
            torch.cuda.synchronize(dev)
            res = (h_pose.numpy(), h_status.numpy(), h_last.numpy().view(np.uint32))
        t1 = time.perf_counter()
        blk = adist.ObsBlock(block)
        if res is not None:
            pk = h_picks.numpy().reshape(-1).view(adist.OBS_DTYPE)
            tail = h_tail.numpy().reshape(-1).view(adist.OBS_DTYPE).reshape(world, max_tags)
            _, n = adist.apply_block(slam, blk, res, picks=pk, tail=tail)
        else:
            _, n = adist.apply_block(slam, blk)
        t2 = time.perf_counter()
        if it > 1:  # the first pass starts without a world tag (sequential start-up), the second is the first to run the graph kernel and the read-backs (16 ms once)
            t_dev.append(t1 - t0); t_host.append(t2 - t1); nseq += n
    return {"world": world, "frames_per_step": world * B, "graph_kernel_and_readback_ms": 1e3 * float(np.median(t_dev)),
            "graph_update_host_ms": 1e3 * float(np.median(t_host)), "serial_ms_per_step": 1e3 * float(np.median(t_dev) + np.median(t_host)),
            "frames_through_sequential_update": nseq, "graph_kernel_and_readback_ms_each": [round(1e3 * t, 3) for t in t_dev],
            "note": "PROJECTED on one GPU: this rank's block replicated %d x on the device, graph kernel + read-back + host apply of a %d-GPU step; the all-gather itself is not included" % (world, world)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs1", help="BASELINE.json config to run (default: the one the metric is quoted on)")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (0 = the workload's default)")
    ap.add_argument("--distinct", type=int, default=0, help="distinct rendered frames (tiled to --batch); 0 = every frame of the batch is distinct")
    ap.add_argument("--pipeline", type=int, default=3, help="detector workspaces/streams per GPU; the batch is split among them so one part's host post-processing and latency-bound tail kernels overlap the other part's bulk kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="only the warm-up, the timed steps and the one isolated batch: no h2d / render / both-minima / projection / CPU legs (for rocprofv3 passes: every detector launch then has the full batch size)")
    ap.add_argument("--rehearse", action="store_true", help="multi-rank rehearsal on a one-GPU box: gloo backend, every rank on cuda:0")
    ap.add_argument("--max-tags", type=int, default=0, help="tag slots per frame in the exchanged observation records (0 = tags per frame + 4)")
    ap.add_argument("--gn-every", type=int, default=-1, help="pose-graph LM on a window every this many steps of the exchange (0 = never, -1 = the workload's default)")
    ap.add_argument("--gn-frames", type=int, default=8, help="frames per stream in the LM window")
    ap.add_argument("--exchange", action="store_true", help="N = 1: run the exchange + graph update + LM of the multi-GPU step anyway (the collective degenerates to a copy)")
    args = ap.parse_args()
    global W, H, NTAGS
    wl = WORKLOADS[args.workload]
    W, H, NTAGS = wl["w"], wl["h"], wl["tags"]
    if args.batch <= 0:
        args.batch = wl["batch"]
    if args.max_tags <= 0:
        args.max_tags = NTAGS + 4
    if args.gn_every < 0:
        args.gn_every = wl["gn_every"]
    args.gn_frames = max(1, min(args.gn_frames, args.batch - 1))  # the LM window and the frame before it come out of one block
    global MAXDET
    MAXDET = max(64, 2 * NTAGS)
    args.exchange = args.exchange or wl["exchange"]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus, or let --gpus start the ranks)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse:
        local_rank = 0
    # a launcher-started single rank with the exchange on still goes through the process group (RCCL with one rank)
    use_pg = world > 1 or (args.exchange and "RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from aprilslam_amd import _lib, synth
    from aprilslam_amd import dist as adist

    K = synth.camera_matrix(W, H)
    xchg = world > 1 or args.exchange  # the part of a step that needs every rank
    B = args.batch
    ndist = B if args.distinct <= 0 else min(args.distinct, B)
    # every rank = its own camera on the trajectory, one scene: the streams see the same tags from different places
    render_det = _lib.Detector("tagStandard41h12", device=local_rank, id_limit=0)
    if wl["partition"] == "frames":
        d_distinct, distinct_gt, rerender = render_stream_device(render_det, ndist, dev, take=(rank, world))
    else:
        d_distinct, distinct_gt, rerender = render_stream_device(render_det, ndist, dev, phase=np.pi * rank / max(world, 1) / max(ndist, 1))
    render_det.close()
    d_frames = d_distinct if ndist == B else d_distinct.repeat((B + ndist - 1) // ndist, 1, 1, 1)[:B].contiguous()
    NCHK = min(32, ndist)  # frames checked against ground truth / handed to the CPU baseline
    distinct = d_distinct[:NCHK].cpu().numpy()
    # P detector workspaces used round-robin: batch i is submitted before batch i-1 is collected, so the host
    # post-processing (dedup/sort/copy-out) and the latency-bound tail kernels of one batch overlap the bulk
    # kernels of the next.  P = 1 is the plain synchronous call.
    P = max(1, args.pipeline)
    detectors = [_lib.Detector("tagStandard41h12", device=local_rank, id_limit=0) for _ in range(P)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    for dd in detectors:
        dd.set_profiling(True)
    det = detectors[0]
    zeros4 = np.zeros(4)
    state = {"i": 0, "inflight": []}

    MT = args.max_tags
    slam = None
    xb = []  # per pipeline part: the buffers of the exchange
    if xchg:
        from aprilslam_amd.slam import SLAM

        class _Log:
            def info(self, m):
                pass
        slam = SLAM(_Log(), {"camera_matrix": K, "dist_coeffs": np.zeros(4)}, tag_size=TAG_INNER, detector=object())
        for _ in range(P):
            xb.append({"obs": torch.empty((B, MT, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev),
                       "pose": torch.zeros((world * B, 16), dtype=torch.float64, device=dev),
                       "status": torch.zeros(world * B, dtype=torch.uint8, device=dev),
                       "last": torch.zeros(adist.MAX_IDS, dtype=torch.int32, device=dev),
                       "h_pose": torch.zeros((world * B, 16), dtype=torch.float64).pin_memory(),
                       "h_status": torch.zeros(world * B, dtype=torch.uint8).pin_memory(),
                       "h_last": torch.zeros(adist.MAX_IDS, dtype=torch.int32).pin_memory(),
                       "picks": torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev),
                       "h_picks": torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8).pin_memory(),
                       "h_tail": torch.zeros((world, MT + args.gn_frames * MT, adist.OBS_DTYPE.itemsize), dtype=torch.uint8).pin_memory(),
                       "block": None, "cid": -2})
    serial = {"gather": [], "graph": [], "gn": [], "seq_frames": 0, "gn_runs": 0}

    def enqueue_exchange(k):
        """Stream-ordered behind batch k, no host wait: pack the records on the device, all-gather them, run the per-frame
        graph kernel for the current world tag and start the read-back of its (small) results."""
        x = xb[k]
        detectors[k].pack_observations_device(x["obs"].data_ptr(), MT, stream=streams[k].cuda_stream)
        if args.rehearse:
            return  # gloo gathers host memory: done after the batch has been collected
        with torch.cuda.stream(streams[k]):
            x["block"] = adist.all_gather_observations(x["obs"])
            x["cid"] = slam.coordinate_id
            if x["cid"] != -1:
                x["last"].zero_()
                detectors[k].graph_frames_device(x["block"].data_ptr(), world, B, MT, x["cid"], x["pose"].data_ptr(), x["status"].data_ptr(),
                                                 x["last"].data_ptr(), adist.MAX_IDS, picks_ptr=x["picks"].data_ptr(), stream=streams[k].cuda_stream)
                x["h_pose"].copy_(x["pose"], non_blocking=True)
                x["h_status"].copy_(x["status"], non_blocking=True)
                x["h_last"].copy_(x["last"], non_blocking=True)
                x["h_picks"].copy_(x["picks"], non_blocking=True)
            # the last frames of every stream: the tail of the update and the LM window
            nt = 1 + args.gn_frames
            x["h_tail"][:, :nt * MT].copy_(x["block"][:, B - nt:].reshape(world, nt * MT, -1), non_blocking=True)

    def update_graph(k):
        """Host part, after batch k has been collected (its stream has drained): apply the gathered block to the graph."""
        x = xb[k]
        t0 = time.perf_counter()
        res = None
        if args.rehearse:
            # gloo gathers host memory: the block goes back to the device for the same graph kernels as the RCCL path
            gathered = adist.all_gather_observations(x["obs"].cpu().numpy().reshape(-1).view(adist.OBS_DTYPE).reshape(B, MT))
            block = adist.ObsBlock(gathered)
            cid = slam.coordinate_id
            if cid != -1:
                d_block = torch.from_numpy(np.ascontiguousarray(gathered).view(np.uint8).reshape(world, B, MT, -1)).to(dev)
                x["d_block"] = d_block
                x["last"].zero_()
                detectors[k].graph_frames_device(d_block.data_ptr(), world, B, MT, cid, x["pose"].data_ptr(), x["status"].data_ptr(),
                                                 x["last"].data_ptr(), adist.MAX_IDS, picks_ptr=x["picks"].data_ptr(),
                                                 stream=torch.cuda.current_stream(dev).cuda_stream)
                x["h_pose"].copy_(x["pose"]); x["h_status"].copy_(x["status"]); x["h_last"].copy_(x["last"]); x["h_picks"].copy_(x["picks"])
                nt = 1 + args.gn_frames
                x["h_tail"][:, :nt * MT].copy_(d_block[:, B - nt:].reshape(world, nt * MT, -1))
                torch.cuda.synchronize(dev)
                res = (x["h_pose"].numpy(), x["h_status"].numpy(), x["h_last"].numpy().view(np.uint32))
        else:
            block = adist.ObsBlock(x["block"])
            if x["cid"] != -1 and x["cid"] == slam.coordinate_id:  # the kernel ran for the world tag the graph still has
                res = (x["h_pose"].numpy(), x["h_status"].numpy(), x["h_last"].numpy().view(np.uint32))
        picks = tail_frames = None
        if res is not None:
            nt = 1 + args.gn_frames
            picks = x["h_picks"].numpy().reshape(-1).view(adist.OBS_DTYPE)
            tail_frames = x["h_tail"].numpy()[:, :nt * MT].reshape(-1).view(adist.OBS_DTYPE).reshape(world, nt, MT)
        t1 = time.perf_counter()
        def picker(lo, hi):
            """last sightings of a stretch of the block on the device (a few frames needed the sequential update)"""
            src = x["block"] if not args.rehearse else x["d_block"]
            detectors[k].graph_picks_device(src.data_ptr(), world, B, MT, x["status"].data_ptr(), lo, hi, x["last"].data_ptr(), adist.MAX_IDS,
                                            x["picks"].data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
            x["h_last"].copy_(x["last"]); x["h_picks"].copy_(x["picks"])
            torch.cuda.synchronize(dev)
            return x["h_last"].numpy().view(np.uint32).copy(), x["h_picks"].numpy().reshape(-1).view(adist.OBS_DTYPE).copy()
        poses, nseq = adist.apply_block(slam, block, res, picks=picks, tail=None if tail_frames is None else tail_frames[:, -1],
                                        picker=picker if res is not None else None)
        t2 = time.perf_counter()
        serial["gather"].append(t1 - t0); serial["graph"].append(t2 - t1); serial["seq_frames"] += nseq
        state["blocks"] = state.get("blocks", 0) + 1
        if args.gn_every and (state["blocks"] % args.gn_every == 0 or state["blocks"] == 2):  # block 2: warm-up run (allocates the LM workspace)
            # window: the last frames of every stream in this block, initial guesses from the graph
            pairs = [(s_, f_) for f_ in range(max(0, B - args.gn_frames), B) for s_ in range(world)]
            recs = block.frames(pairs) if tail_frames is None else [tail_frames[s_, f_ - (B - 1 - args.gn_frames)] for s_, f_ in pairs]
            cams, obs = [], []
            for (s_, f_), rec in zip(pairs, recs):
                rec = rec[(rec["flags"] & 3) == 3]
                if len(rec) and not np.isnan(poses[s_, f_]).any():
                    cams.append(poses[s_, f_])
                    # (id, corners, single-view PnP pose): the poses let the solve pick consistent starting values
                    obs.append([(int(i), c.reshape(4, 2).astype(np.float64), T_) for i, c, T_ in zip(rec["id"], rec["corners"], adist._full(rec["T"]))])
            if cams:
                state["gn_last"] = slam.optimize_window(cams, obs, iters=2, backend=detectors[k])
                serial["gn_runs"] += 1
            serial["gn"].append(time.perf_counter() - t2)

    def finish(k):
        dets, poses, npf = detectors[k].collect_view()  # views of the detector's page-locked result buffers: no host copy
        if xchg:
            update_graph(k)
        for kk, v in detectors[k].stage_times().items():
            kernel_ms.setdefault(kk, []).append(v)
        return dets, npf

    def step():
        """Submit one batch of B frames; collect the oldest batch once P are in flight."""
        k = state["i"] % P
        state["i"] += 1
        if len(state["inflight"]) == P:
            finish(state["inflight"].pop(0))
        detectors[k].submit_device(d_frames.data_ptr(), B, 3, W, H, stream=streams[k].cuda_stream, K=K, dist=zeros4, tag_size=TAG_INNER)
        if xchg:
            enqueue_exchange(k)
        state["inflight"].append(k)

    def drain():
        out = None
        while state["inflight"]:
            out = finish(state["inflight"].pop(0))
        return out

    kernel_ms = {}
    lm_bad = False
    for k in range(P):  # set-up, not a step: first use allocates each workspace (hipMalloc of several GB)
        detectors[k].submit_device(d_frames.data_ptr(), B, 3, W, H, stream=streams[k].cuda_stream, K=K, dist=zeros4, tag_size=TAG_INNER)
        detectors[k].collect(max_per_frame=MAXDET)
    for _ in range(args.warmup):
        step()
    res = drain()
    n_found = int(len(res[0])) if res else -1

    kernel_ms.clear()
    for v in serial.values():
        if isinstance(v, list):
            v.clear()
    serial["seq_frames"] = 0; serial["gn_runs"] = 0
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize(dev)
    if use_pg:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_pg:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one extra synchronous batch (nothing else on the GPU) for un-overlapped kernel durations
    detectors[0].submit_device(d_frames.data_ptr(), B, 3, W, H, stream=streams[0].cuda_stream, K=K, dist=zeros4, tag_size=TAG_INNER)
    last = detectors[0].collect(max_per_frame=MAXDET)
    last = (last[0].copy(), last[1].copy(), last[2].copy())
    isolated = detectors[0].stage_times()

    if rank == 0:
        avg = {k: float(np.mean(v)) for k, v in kernel_ms.items()}
        kernels_only = {k: v for k, v in avg.items() if k.startswith("k_")}
        # the slowest kernel of the step that has an HBM byte model (the per-cluster / per-quad kernels have none)
        if args.workload != "configs1":
            # the per-run / per-point terms of the byte models: measured points of this scene, runs in the bench scene's proportion
            pts_frame = float(detectors[0].debug_counters()[4]) / B
            _ab = algorithmic_bytes
            algorithmic_bytes_wl = lambda k_, w_, h_, c_, f_: _ab(k_, w_, h_, c_, f_, runs=int(0.39 * pts_frame), points=int(pts_frame))  # noqa: E731
        else:
            algorithmic_bytes_wl = algorithmic_bytes
        modelled = {k: v for k, v in kernels_only.items() if algorithmic_bytes_wl(k, W, H, 3, 2) is not None}
        dom = max(modelled or kernels_only, key=(modelled or kernels_only).get)
        ab = algorithmic_bytes_wl(dom, W, H, 3, 2)
        roof = None
        if ab is not None:
            achieved = ab * B / (avg[dom] * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg[dom],
                    "avg_launch_ms_isolated": isolated.get(dom), "algorithmic_bytes_per_frame": ab}
        else:
            # per-cluster / per-quad kernels: latency- and occupancy-bound work on L2-resident slabs; their
            # compulsory HBM bytes are the boundary-point keys (8 B read) plus the moment slab (64 B written)
            pts = 8.0 * int(detectors[0].debug_counters()[4]) / B
            achieved = (pts * 9) * B / (avg[dom] * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg[dom],
                    "avg_launch_ms_isolated": isolated.get(dom), "algorithmic_bytes_per_frame": pts * 9}
        # HBM traffic: not measurable live; taken from the committed rocprofv3 --pmc pass of the same build
        tr = None
        try:
            import glob
            tr = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_traffic.json")))[-1]))  # the latest round's counter passes
            if tr.get("batch_frames") != B:
                tr = None
        except Exception:
            tr = None
        if tr and dom in tr["bytes_per_launch"]:
            roof["traffic"] = tr["bytes_per_launch"][dom]
            roof["traffic_source"] = tr["source"]
        line_rooflines = []
        for kname, ms_ in sorted(kernels_only.items(), key=lambda kv: -kv[1]):
            ab_ = algorithmic_bytes_wl(kname, W, H, 3, 2)
            if ab_ is not None:
                gbs = ab_ * B / (ms_ * 1e-3) / 1e9
                line_rooflines.append({"kernel": kname, "avg_launch_ms": ms_, "avg_launch_ms_isolated": isolated.get(kname),
                                       "algorithmic_bytes_per_frame": ab_, "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS})
        seg_names = ("k_hash_clear", "k_decimate_minmax", "k_tile_cut", "k_seg_tile", "k_seg_border", "k_seg_roots", "k_seg_points", "k_cluster_filter",
                     "k_point_place")
        seg = sum(isolated.get(k, 0.0) for k in seg_names)
        seg_bytes = stage_algorithmic_read_bytes(W, H, 3, 2)
        seg_gbs = seg_bytes * B / (seg * 1e-3) / 1e9 if seg > 0 else 0.0
        # The headline roofline is the stage SURVEY 8(d) defines the algorithmic bytes for: threshold + segmentation is a
        # chain of launches (K1-K9), each of them processes the same B frames, so "the kernel" is the chain and its launch
        # duration the sum of its members' HIP-event durations inside the timed region (where the two pipeline parts share
        # the GPU, so each member runs longer than alone; the isolated sum is given beside it).  The slowest single kernel
        # with a byte model of its own is kept as `dominant_kernel`.
        seg_live = sum(avg.get(k, 0.0) for k in seg_names)
        live_gbs = seg_bytes * B / (seg_live * 1e-3) / 1e9 if seg_live > 0 else 0.0
        stage_roof = {"kernel": "threshold + segmentation stage (k_hash_clear, k_decimate_minmax, k_tile_cut, k_seg_tile, k_seg_border, "
                                "k_seg_roots, k_seg_points, k_cluster_filter, k_point_place)",
                      "bound": "hbm", "achieved": live_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": live_gbs / HBM_PEAK_GBS,
                      "traffic": (sum(tr["bytes_per_launch"].get(k, 0.0) for k in seg_names) if tr else None),
                      "traffic_source": tr["source"] if tr else None,
                      "avg_launch_ms": seg_live, "avg_launch_ms_isolated": seg, "achieved_isolated": seg_gbs, "frac_isolated": seg_gbs / HBM_PEAK_GBS,
                      "algorithmic_bytes_per_frame": seg_bytes, "units_per_launch": B,
                      "members_ms": {k: avg.get(k, 0.0) for k in seg_names}, "dominant_kernel": roof}
        line = {
            "metric": "frames/sec at %dx%d, %d tags/frame (detection + batched PnP%s)" % (W, H, NTAGS, ", graph update + pose-graph LM" if xchg else ""),
            "value": world * B * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/f64",
            "data": "synthetic: %d distinct frames of a seeded %d-tag scene along a camera trajectory, rendered on the device into HBM (%d frames per step)" % (ndist, NTAGS, B),
            "config": {"workload": wl["name"], "batch_frames": B,
                       "decimate": 2, "tags_found_per_batch": n_found, "pipeline_parts": P,
                       "parallelism": "frame i of one stream on GPU i mod N" if wl["partition"] == "frames" else "1 video stream per GPU"},
            "multi_gpu": None if not xchg else {
                "exchange": "one all_gather_into_tensor per step of %d x %d asl_obs records (136 B) per rank = %.1f MB per rank, packed on the device" % (B, MT, B * MT * 136 / 1e6),
                "backend": "gloo (rehearsal)" if args.rehearse else "nccl (RCCL)",
                "serial_ms_per_step": {"gather_host_side": 1e3 * float(np.mean(serial["gather"])), "graph_update_host": 1e3 * float(np.mean(serial["graph"])),
                                       "pose_graph_lm_per_run": (1e3 * float(np.mean(serial["gn"])) if serial["gn"] else None)},
                "frames_through_sequential_update": serial["seq_frames"], "lm_runs": serial["gn_runs"],
                "lm_last": {k_: v_ for k_, v_ in (state.get("gn_last") or {}).items() if k_ != "camera_poses"} or None,
                "graph_nodes": len(slam.graph.get_nodes()), "world_tag": slam.coordinate_id},
            "roofline": stage_roof,
            "stage_threshold_segmentation": {"ms_per_batch_isolated": seg, "algorithmic_read_bytes_per_frame": seg_bytes,
                                             "achieved_GBs": seg_gbs, "frac_of_hbm_peak": seg_gbs / HBM_PEAK_GBS},
            "kernel_rooflines_hbm": line_rooflines,
            "kernel_ms_per_batch": avg,
            "kernel_ms_per_batch_isolated": isolated,
        }
        nchk = NCHK
        line["pose_rmse"] = pose_rmse_vs_ground_truth(last[0], last[1], last[2], distinct_gt[:nchk])
        if world == 1 and not args.timed_only:
            # the same frames with the better of the two planar poses per tag (not what the reference computes: reported apart)
            detectors[0].set_pnp_both_minima(True)
            detectors[0].submit_device(d_frames.data_ptr(), B, 3, W, H, stream=streams[0].cuda_stream, K=K, dist=zeros4, tag_size=TAG_INNER)
            alt = detectors[0].collect(max_per_frame=MAXDET)
            detectors[0].set_pnp_both_minima(False)
            line["pose_rmse_both_minima"] = pose_rmse_vs_ground_truth(alt[0], alt[1], alt[2], distinct_gt[:nchk])
            line["pose_rmse_both_minima"]["k_pnp_dets_ms"] = detectors[0].stage_times().get("k_pnp_dets")
            line["h2d_included"] = h2d_included(detectors[0], d_frames, K)
            line["multi_gpu_projection"] = project_serial_term(detectors[0], d_frames, K, dev, MT)
            if args.workload == "configs1":
                line["stage_threshold_segmentation_decimate1"] = stage_at_decimate_1(d_frames, K)
            if ndist == B:
                # the producer inside the step: every step first renders its B frames on the device, then detects
                def rstep(k_):
                    rerender(detectors[k_], streams[k_].cuda_stream)
                    detectors[k_].submit_device(d_frames.data_ptr(), B, 3, W, H, stream=streams[k_].cuda_stream, K=K, dist=zeros4, tag_size=TAG_INNER)
                nrs = max(4, args.steps // 2)
                torch.cuda.synchronize(dev)
                tr0 = time.perf_counter()
                for i_ in range(nrs):  # one workspace: the next render may not overwrite frames a running batch still reads
                    rstep(0)
                    detectors[0].collect(max_per_frame=MAXDET)
                torch.cuda.synchronize(dev)
                tr = time.perf_counter() - tr0
                line["render_included"] = {"value": B * nrs / tr, "unit": "frames/s", "steps": nrs, "ms_per_step": 1e3 * tr / nrs,
                                           "note": "asl_render_frames_device (k_render) of the step's 1024 frames + the same batch, one workspace, synchronous; not the headline value"}
        if not args.no_cpu_baseline and not args.timed_only and world == 1:  # reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(distinct[:nchk], K, gpu=last)
            line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(distinct[:nchk], K)
        print(json.dumps(line))
        lm = state.get("gn_last")
        if xchg and lm is not None:
            # a pose-graph solve inside the timed region must have solved something: a start behind the image plane or a
            # diverged step shows as a cost of 1e12 and more per observation
            per_obs = lm["cost"] / max(lm["observations"], 1)
            if not (lm["cost"] <= lm["cost0"] and per_obs < 50.0):
                sys.stderr.write("bench.py: the pose-graph LM of the last window is unhealthy (cost0 %.3g -> cost %.3g over %d observations)\n"
                                 % (lm["cost0"], lm["cost"], lm["observations"]))
                lm_bad = True
    if use_pg:
        dist.destroy_process_group()
    if lm_bad:
        sys.exit(3)


if __name__ == "__main__":
    main()
