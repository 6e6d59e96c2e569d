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
    obs = adist.pack_observations(dets, poses, npf, max_tags=6)
    gathered = adist.all_gather_observations(obs)        # numpy records -> gloo all_gather_into_tensor on their bytes
    out[rank] = gathered.view(np.uint8).reshape(gathered.shape + (-1,))
    dist.destroy_process_group()


def _blocks(rng, world, n_frames, max_tags, n_blocks, world_tag=0):
    """Random gathered blocks: most frames self-contained (world tag present, every PnP ok), with an empty frame, a frame
    that misses the world tag and one with a failed PnP thrown in."""
    from aprilslam_amd import dist as adist, synth
    out = []
    for b in range(n_blocks):
        obs = np.zeros((world, n_frames, max_tags), dtype=adist.OBS_DTYPE)
        obs["id"] = -1
        for s in range(world):
            for f in range(n_frames):
                ids = [world_tag] + sorted(rng.choice(np.arange(1, 9), size=rng.integers(1, max_tags - 1), replace=False).tolist())
                if (b, s, f) == (2, 1, 2):
                    ids = ids[1:]          # world tag out of view
                if (b, s, f) == (1, 0, 3):
                    ids = []               # nothing detected
                for j, i in enumerate(ids):
                    T = synth.camera_from_tag([rng.uniform(-20, 20), rng.uniform(-10, 10), -rng.uniform(40, 90)], rng.uniform(-20, 20, 3))
                    obs["id"][s, f, j] = i
                    obs["flags"][s, f, j] = 1 if (b, s, f, j) == (3, 0, 1, 1) else 3
                    obs["corners"][s, f, j] = rng.uniform(0, 700, 8)
                    obs["T"][s, f, j] = T[:3].ravel()
        out.append(obs)
    return out


def _new_slam():
    from aprilslam_amd.slam import SLAM

    class Log:
        def info(self, m):
            pass

    return SLAM(Log(), {"camera_matrix": np.eye(3), "dist_coeffs": np.zeros(4)}, detector=object())


def test_block_update_equals_the_sequential_reference_update():
    """apply_block (self-contained frames in bulk, the rest through the mirror) reaches exactly the state of the
    reference's one-observation-at-a-time update in (frame, stream) order: node matrices bit for bit, per-frame
    poses to 1e-12 (the bulk path sums the same votes in the same order, with batched inverses)."""
    import contextlib
    import io
    from aprilslam_amd import dist as adist
    rng = np.random.default_rng(42)
    a, b = _new_slam(), _new_slam()
    seq_counts = []
    for obs in _blocks(rng, 2, 5, 7, 5):
        world, n_frames, _ = obs.shape
        pa, nseq = adist.apply_block(a, obs)
        seq_counts.append(nseq)
        order = [(s, f) for f in range(n_frames) for s in range(world)]
        with contextlib.redirect_stdout(io.StringIO()):
            pb = adist._sequential(b, obs, order)
        for (s, f), p in zip(order, pb):
            assert np.isnan(pa[s, f]).all() if p is None else np.abs(pa[s, f] - p).max() < 1e-12
        ga, gb = a.graph.get_nodes(), b.graph.get_nodes()
        assert sorted(ga) == sorted(gb) and a.coordinate_id == b.coordinate_id
        for k in ga:
            assert np.array_equal(ga[k].local, gb[k].local) and np.array_equal(ga[k].world, gb[k].world)
            assert (ga[k].reference, ga[k].weight, ga[k].updated, ga[k].visible) == (gb[k].reference, gb[k].weight, gb[k].updated, gb[k].visible)
        assert np.array_equal(a.graph.estimated_pose, b.graph.estimated_pose) and a.visible_tags == b.visible_tags
    # block 0 starts without a world tag, blocks 2 and 3 hold a frame that is not self-contained; 1 and 4 run in bulk
    assert seq_counts[1] == 0 and seq_counts[4] == 0 and seq_counts[0] > 0 and seq_counts[2] > 0 and seq_counts[3] > 0


def test_all_gather_and_ordered_update_two_ranks():
    from aprilslam_amd import dist as adist
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0 = out[0].reshape(-1).view(adist.OBS_DTYPE).reshape(out[0].shape[:-1])
    g1 = out[1].reshape(-1).view(adist.OBS_DTYPE).reshape(out[1].shape[:-1])
    assert g0.shape == (2, 3, 6)
    assert np.array_equal(out[0], out[1]), "ranks disagree on the gathered observations"
    # the gathered block equals what each rank packed locally
    for r in range(world):
        dets, poses, npf = _fake_step(r)
        assert np.array_equal(g0[r], adist.pack_observations(dets, poses, npf, max_tags=6))
    assert g0["id"][0, 0, :4].tolist() == [0, 1, 2, 3] and g0["id"][0, 1, 0] == -1 and g0["flags"][0, 0, 0] == 3

    res = []
    for g in (g0, g1):  # identical input -> identical graph on "every rank"
        slam = _new_slam()
        poses, _ = adist.apply_block(slam, g)
        res.append((poses, {k: (v.local.copy(), v.world.copy(), v.weight, v.reference) for k, v in slam.graph.get_nodes().items()}))
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True)
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) > 0
    for k in res[0][1]:
        assert all(np.array_equal(x, y) for x, y in zip(res[0][1][k], res[1][1][k]))


def test_shard_frames():
    from aprilslam_amd import dist as adist
    shards = [adist.shard_frames(10, r, 4) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(10))
    assert shards[1] == [1, 5, 9]


def test_segmented_update_randomised():
    """Blocks with several frames that need the sequential update (world tag out of view, failed PnP), empty frames also at
    the end of a stretch, and world-tag switches in the middle of a block: the segmented bulk path (last sightings of every
    self-contained stretch + the mirror for the frames in between) ends in exactly the state of the sequential update."""
    import contextlib
    import io
    from aprilslam_amd import dist as adist, synth
    rng = np.random.default_rng(7)
    a, b = _new_slam(), _new_slam()
    world, n_frames, max_tags = 3, 12, 7
    bulk_frames = 0
    for blk_i in range(40):
        obs = np.zeros((world, n_frames, max_tags), dtype=adist.OBS_DTYPE)
        obs["id"] = -1
        low = 3 if blk_i < 6 else (1 if blk_i < 14 else 0)  # lower ids appear later: the world tag switches twice
        for s in range(world):
            for f in range(n_frames):
                u = rng.random()
                pool = np.arange(low, 9)
                ids = sorted(rng.choice(pool, size=rng.integers(1, min(max_tags, len(pool)) + 1), replace=False).tolist())
                if u < 0.10:
                    ids = []
                elif u < 0.85 and a.coordinate_id >= 0 and a.coordinate_id not in ids and len(ids) < max_tags and a.coordinate_id >= low:
                    ids = sorted(ids + [a.coordinate_id])[:max_tags]
                    if a.coordinate_id not in ids:
                        ids = sorted([a.coordinate_id] + ids[:-1])
                for j, i in enumerate(ids):
                    T = synth.camera_from_tag([rng.uniform(-20, 20), rng.uniform(-10, 10), -rng.uniform(40, 90)], rng.uniform(-20, 20, 3))
                    obs["id"][s, f, j] = i
                    obs["flags"][s, f, j] = 1 if rng.random() < 0.02 else 3
                    obs["T"][s, f, j] = T[:3].ravel()
        pa, nseq = adist.apply_block(a, obs)
        bulk_frames += world * n_frames - nseq
        order = [(s, f) for f in range(n_frames) for s in range(world)]
        with contextlib.redirect_stdout(io.StringIO()):
            pb = adist._sequential(b, obs, order)
        for (s, f), p in zip(order, pb):
            assert np.isnan(pa[s, f]).all() if p is None else np.abs(pa[s, f] - p).max() < 1e-12, (blk_i, s, f)
        ga, gb = a.graph.get_nodes(), b.graph.get_nodes()
        assert sorted(ga) == sorted(gb) and a.coordinate_id == b.coordinate_id, blk_i
        for k in ga:
            assert np.array_equal(ga[k].local, gb[k].local) and np.array_equal(ga[k].world, gb[k].world), (blk_i, k)
            assert (ga[k].reference, ga[k].weight, ga[k].updated, ga[k].visible) == (gb[k].reference, gb[k].weight, gb[k].updated, gb[k].visible), (blk_i, k)
        assert np.array_equal(a.graph.estimated_pose, b.graph.estimated_pose) and a.visible_tags == b.visible_tags, blk_i
    assert bulk_frames > 0.5 * 40 * world * n_frames  # most frames did take the bulk path
    assert a.coordinate_id == 0


def test_spawn_ranks_stops_the_peers_of_a_failed_rank(tmp_path, capfd):
    """bench.py --gpus N starts its own ranks: rank 0's stdout is passed through, and a rank that dies takes the others
    down with a non-zero exit code instead of leaving them waiting in a collective."""
    import textwrap
    import time
    import bench
    fake = tmp_path / "fake_rank.py"
    fake.write_text(textwrap.dedent('''
        import os, sys, time
        r = int(os.environ["RANK"])
        assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        if "--fail" in sys.argv and r == 2:
            sys.exit(7)
        if r == 0:
            time.sleep(60 if "--fail" in sys.argv else 0.2)
            print('{"n_gpus": 3}')
        elif "--fail" in sys.argv:
            time.sleep(60)
    '''))
    real = bench.__file__
    bench.__file__ = str(fake)
    try:
        assert bench.spawn_ranks(3, []) == 0
        assert '{"n_gpus": 3}' in capfd.readouterr().out
        t0 = time.monotonic()
        assert bench.spawn_ranks(3, ["--fail"]) == 1
        assert time.monotonic() - t0 < 20
        assert "rank 2 exited with code 7" in capfd.readouterr().err
    finally:
        bench.__file__ = real
