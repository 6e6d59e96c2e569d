"""Multi-rank path on CPU: world_size-2 gloo all-gather of observation records and the deterministic
ordered graph update (the N>1 path of bench.py / SURVEY.md section 8e)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_step(rank, n_frames=3, ntags=4):
    """Deterministic fake detections/poses of one rank (stream)."""
    from aprilslam_amd import _lib
    rng = np.random.default_rng(100 + rank)
    npf = np.array([ntags, 0, ntags - 1][:n_frames])
    n = int(npf.sum())
    dets = np.zeros(n, dtype=_lib.DET_DTYPE)
    poses = np.zeros(n, dtype=_lib.POSE_DTYPE)
    k = 0
    for f in range(n_frames):
        for t in range(int(npf[f])):
            dets["id"][k] = t + rank  # streams see overlapping tag sets
            dets["frame"][k] = f
            dets["corners"][k] = rng.uniform(0, 100, (4, 2))
            T = np.eye(4)
            a = rng.uniform(-0.3, 0.3)
            T[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            T[:3, 3] = rng.uniform(-50, 50, 3)
            poses["T"][k] = T
            poses["ok"][k] = 1
            k += 1
    return dets, poses, npf


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aprilslam_amd import dist as adist
    dets, poses, npf = _fake_step(rank)
    obs = adist.pack_observations(dets, poses, npf, rank, max_tags=6)
    gathered = adist.all_gather_observations(obs)
    out[rank] = gathered
    dist.destroy_process_group()


def test_all_gather_and_ordered_update_two_ranks():
    from aprilslam_amd import dist as adist
    from aprilslam_amd.slam import SLAM
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, g1 = out[0], out[1]
    assert g0.shape == (2, 3, 6, adist.OBS_WIDTH)
    assert np.array_equal(g0, g1), "ranks disagree on the gathered observations"
    # the gathered block equals what each rank packed locally
    for r in range(world):
        dets, poses, npf = _fake_step(r)
        assert np.array_equal(g0[r], adist.pack_observations(dets, poses, npf, r, max_tags=6))

    class Log:
        def info(self, m):
            pass

    res = []
    for _ in range(2):  # identical input -> identical graph on "every rank"
        slam = SLAM(Log(), {"camera_matrix": np.eye(3), "dist_coeffs": np.zeros(4)}, detector=object())
        poses = adist.apply_observations(slam, g0)
        res.append((poses, {k: (v.world.copy(), v.weight, v.reference) for k, v in slam.graph.get_nodes().items()}))
    assert len(res[0][0]) == 6
    for a, b in zip(res[0][0], res[1][0]):
        assert (a is None and b is None) or np.array_equal(a, b)
    assert res[0][1].keys() == res[1][1].keys()
    for k in res[0][1]:
        assert np.array_equal(res[0][1][k][0], res[1][1][k][0])


def test_shard_frames():
    from aprilslam_amd import dist as adist
    shards = [adist.shard_frames(10, r, 4) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(10))
    assert shards[1] == [1, 5, 9]
