"""Randomised end-to-end sweep (GPU vs CPU oracle): many small seeded scenes of assorted sizes, tag counts, noise
levels, camera poses and decimations, in ragged batches.  Complements the stage-by-stage tests in test_gpu_parity.py:
same bars (ids, hamming, corner order exact; corners/centres to 1e-9 px; PnP to 1e-6)."""
import numpy as np
import pytest

import oracle_lib as O
from aprilslam_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _scene(w, h, ntags, seed, noise):
    rng = np.random.default_rng(seed)
    tags = synth.random_scene(w, h, ntags, rng)
    pos = tuple(rng.uniform(-3, 3, 3))
    rot = tuple(rng.uniform(-2, 2, 3))
    frame, _ = synth.render_frame(w, h, tags, 18.0, cam_position=pos, cam_rotation_deg=rot, noise_sigma=noise, rng=rng)
    return frame


@pytest.mark.parametrize("decimate", [1, 2, 3])
def test_random_scenes_end_to_end(family, decimate):
    sizes = [(640, 360), (642, 362), (801, 601), (1280, 720), (320, 200)]
    det = _lib.Detector("tagStandard41h12", decimate=float(decimate), id_limit=0)
    try:
        seed = 1000 * decimate
        for (w, h) in sizes:
            nb = 1 + (w * h) % 3  # ragged batch sizes 1..3
            frames = []
            for b in range(nb):
                seed += 1
                frames.append(_scene(w, h, 2 + seed % 7, seed, noise=float(seed % 3)))
            frames = np.stack(frames)
            K = synth.camera_matrix(w, h)
            dets, poses, npf = det.detect_host(frames, K=K, dist=np.zeros(4), tag_size=10.0)
            start = 0
            for b in range(nb):
                ref = O.detect_gray(O.bgr2gray(frames[b]), family, decimate)
                mine = dets[start:start + npf[b]]
                mp = poses[start:start + npf[b]]
                start += npf[b]
                assert [int(d["id"]) for d in mine] == [r["id"] for r in ref], (w, h, b, decimate)
                for d, r in zip(mine, ref):
                    assert int(d["hamming"]) == r["hamming"]
                    assert np.abs(d["corners"] - r["corners"]).max() <= 1e-9
                    assert np.abs(d["center"] - r["center"]).max() <= 1e-9
                if len(mine):
                    c32 = np.stack([np.asarray(d["corners"], dtype=np.float32) for d in mine])
                    orv, otv, oT, ook = O.solve_pnp(c32, K, np.zeros(4), 10.0)
                    assert np.array_equal(mp["ok"].astype(bool), ook.astype(bool))
                    assert np.abs(mp["rvec"] - orv).max() <= 1e-6 and np.abs(mp["tvec"] - otv).max() <= 1e-6
    finally:
        det.close()


def test_device_observation_records_and_graph_frames(family):
    """asl_pack_observations_device against the host packing of the same results, and asl_graph_frames_device against
    its numpy mirror (status and last-seen table exact, poses to 1e-9) on those records."""
    import torch
    from aprilslam_amd import dist as adist
    det = _lib.Detector("tagStandard41h12", id_limit=0)
    try:
        w, h, nb, max_tags = 640, 360, 6, 8
        frames = np.stack([_scene(w, h, 3 + b % 4, 500 + b, noise=0.0) for b in range(nb)])
        frames[4] = 128  # a frame without detections
        K = synth.camera_matrix(w, h)
        dev = torch.device("cuda", 0)
        t = torch.from_numpy(frames).to(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        det.submit_device(t.data_ptr(), nb, 3, w, h, stream=st, K=K, dist=np.zeros(4), tag_size=10.0)
        obs_dev = torch.empty((nb, max_tags, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        det.pack_observations_device(obs_dev.data_ptr(), max_tags, stream=st)
        dets, poses, npf = det.collect()
        host = adist.pack_observations(dets, poses, npf, max_tags)
        got = obs_dev.cpu().numpy().reshape(-1).view(adist.OBS_DTYPE).reshape(nb, max_tags)
        assert npf[4] == 0 and npf.sum() > 10
        for name in ("id", "flags", "corners", "T"):
            assert np.array_equal(got[name], host[name]), name
        # two "streams": the block and a copy with one frame's world tag removed and one PnP marked failed
        other = got.copy()
        c = int(got["id"][0, 0])
        other["id"][2, :-1] = got["id"][2, 1:]; other["flags"][2, :-1] = got["flags"][2, 1:]; other["T"][2, :-1] = got["T"][2, 1:]
        other["flags"][3, 1] = 1
        block = np.stack([got, other])
        # frames whose lowest id is not c are "not self-contained" by definition: both implementations must agree on that
        obs2 = torch.from_numpy(block.view(np.uint8).reshape(2, nb, max_tags, -1)).to(dev)
        pose = torch.zeros((2 * nb, 16), dtype=torch.float64, device=dev)
        status = torch.zeros(2 * nb, dtype=torch.uint8, device=dev)
        last = torch.zeros(adist.MAX_IDS, dtype=torch.int32, device=dev)
        picks = torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        det.graph_frames_device(obs2.data_ptr(), 2, nb, max_tags, c, pose.data_ptr(), status.data_ptr(), last.data_ptr(), adist.MAX_IDS,
                                picks_ptr=picks.data_ptr(), stream=st)
        torch.cuda.synchronize(dev)
        rp, rs, rl = adist.graph_frames_numpy(block, c)
        assert np.array_equal(status.cpu().numpy().reshape(2, nb), rs)
        assert np.array_equal(last.cpu().numpy().view(np.uint32), rl)
        assert (rs == 0).sum() >= 3 and (rs == 1).sum() >= 2 and (rs == 2).sum() == 2
        assert np.abs(pose.cpu().numpy().reshape(2, nb, 4, 4) - rp).max() < 1e-9
        # the picks are the records the last-seen table points at, and they finish the update like the block itself does
        pk = picks.cpu().numpy().reshape(-1).view(adist.OBS_DTYPE)
        for t in np.nonzero(rl)[0]:
            key = int(rl[t]) - 1
            slot, o = key % max_tags, key // max_tags
            assert pk[2 * t] == block[o % 2, o // 2, slot] and pk[2 * t + 1] == block[o % 2, o // 2, 0]
        import golden_scene as G
        a, b = G.new_slam(), G.new_slam()
        good = block[:, [0, 1, 5]]  # frames that are self-contained in both streams
        for sl in (a, b):
            adist.apply_block(sl, good)  # first block: sequential (no world tag yet)
        r2 = adist.graph_frames_numpy(good, c)
        adist.apply_block(a, good, r2)
        pk2 = np.zeros(2 * adist.MAX_IDS, dtype=adist.OBS_DTYPE)
        for t in np.nonzero(r2[2])[0]:
            key = int(r2[2][t]) - 1
            pk2[2 * t] = good[(key // max_tags) % 2, (key // max_tags) // 2, key % max_tags]
            pk2[2 * t + 1] = good[(key // max_tags) % 2, (key // max_tags) // 2, 0]
        adist.apply_block(b, adist.ObsBlock(good), r2, picks=pk2, tail=good[:, -1])
        for k_ in a.graph.get_nodes():
            assert np.array_equal(a.graph.get_nodes()[k_].world, b.graph.get_nodes()[k_].world)
            assert np.array_equal(a.graph.get_nodes()[k_].local, b.graph.get_nodes()[k_].local)
        assert np.array_equal(a.graph.estimated_pose, b.graph.estimated_pose)
    finally:
        det.close()


def test_full_size_batch_properties(family):
    """BASELINE.json's batch (1024 frames of 1280x720 BGR, 20 tags per frame, resident in HBM) through properties that do
    not depend on the size: every copy of a frame gives bit-identical results wherever it sits in the batch (frames do
    not interact: no leak through the shared hash table, point pool, cluster lists or counters), a second run of the
    same batch is bit-identical (atomics only decide WHERE things are stored, never what), and four of the frames
    match the CPU oracle."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    det = _lib.Detector("tagStandard41h12", id_limit=0)
    try:
        ndist, B = 32, 1024
        distinct, _gt, _ = bench.render_stream_device(det, ndist, dev)
        perm = np.random.default_rng(7).permutation(B) % ndist  # which distinct frame sits at each batch position
        frames = distinct[torch.from_numpy(perm).to(dev)].contiguous()
        K = synth.camera_matrix(bench.W, bench.H)
        runs = []
        for _ in range(2):
            d, p, n = det.detect_device(frames.data_ptr(), B, 3, bench.W, bench.H, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER, max_per_frame=64)
            runs.append((d.copy(), p.copy(), n.copy()))
        (d0, p0, n0), (d1, p1, n1) = runs
        assert np.array_equal(n0, n1) and d0.tobytes() == d1.tobytes() and p0.tobytes() == p1.tobytes()
        off = np.concatenate([[0], np.cumsum(n0)])
        first = {}
        for b in range(B):
            sl = slice(off[b], off[b + 1])
            cur = tuple(np.ascontiguousarray(d0[sl][f]).tobytes() for f in ("id", "hamming", "margin", "center", "corners")) + (p0[sl].tobytes(),)
            k = int(perm[b])
            if k in first:
                assert cur == first[k], "frame %d (copy of distinct frame %d) differs from its first copy" % (b, k)
            else:
                first[k] = cur
                assert n0[b] == bench.NTAGS
        host = distinct[:4].cpu().numpy()
        for k in range(4):
            b = int(np.flatnonzero(perm == k)[0])
            ref = O.detect_bgr(host[k], family)
            mine = d0[off[b]:off[b + 1]]
            assert [int(x["id"]) for x in mine] == [r["id"] for r in ref]
            for x, r in zip(mine, ref):
                assert np.abs(x["corners"] - r["corners"]).max() <= 1e-9
    finally:
        det.close()


def test_device_last_sightings_of_a_stretch(family):
    """asl_graph_picks_device against its numpy mirror: last sightings and picks over ranges of a gathered block that holds
    frames without the world tag and empty frames."""
    import torch
    from aprilslam_amd import dist as adist
    rng = np.random.default_rng(11)
    world, n_frames, max_tags = 3, 40, 6
    obs = np.zeros((world, n_frames, max_tags), dtype=adist.OBS_DTYPE)
    obs["id"] = -1
    for s in range(world):
        for f in range(n_frames):
            u = rng.random()
            ids = [] if u < 0.1 else sorted(rng.choice(np.arange(1, 9), size=rng.integers(1, max_tags - 1), replace=False).tolist())
            if ids and u > 0.25:
                ids = [0] + ids
            for j, i in enumerate(ids):
                obs["id"][s, f, j] = i
                obs["flags"][s, f, j] = 3
                obs["T"][s, f, j] = synth.camera_from_tag([rng.uniform(-20, 20), rng.uniform(-10, 10), -rng.uniform(40, 90)], rng.uniform(-20, 20, 3))[:3].ravel()
    dev = torch.device("cuda", 0)
    d_obs = torch.from_numpy(obs.view(np.uint8).reshape(world, n_frames, max_tags, -1)).to(dev)
    pose = torch.zeros((world * n_frames, 16), dtype=torch.float64, device=dev)
    status = torch.zeros(world * n_frames, dtype=torch.uint8, device=dev)
    last = torch.zeros(adist.MAX_IDS, dtype=torch.int32, device=dev)
    picks = torch.zeros((2 * adist.MAX_IDS, adist.OBS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    det = _lib.Detector("tagStandard41h12")
    try:
        det.graph_frames_device(d_obs.data_ptr(), world, n_frames, max_tags, 0, pose.data_ptr(), status.data_ptr(), last.data_ptr(), adist.MAX_IDS)
        torch.cuda.synchronize()
        st = status.cpu().numpy().reshape(world, n_frames)
        _, st_ref, _ = adist.graph_frames_numpy(obs, 0, adist.MAX_IDS)
        assert np.array_equal(st, st_ref) and (st == 1).sum() > 3 and (st == 2).sum() > 3
        for lo, hi in [(0, world * n_frames), (0, 17), (17, 58), (58, 59), (100, 120)]:
            det.graph_picks_device(d_obs.data_ptr(), world, n_frames, max_tags, status.data_ptr(), lo, hi, last.data_ptr(), adist.MAX_IDS, picks.data_ptr())
            torch.cuda.synchronize()
            l_ref, p_ref = adist.last_sightings_numpy(obs, st, lo, hi)
            l_dev = last.cpu().numpy().view(np.uint32)
            assert np.array_equal(l_dev, l_ref), (lo, hi)
            p_dev = picks.cpu().numpy().reshape(-1).view(adist.OBS_DTYPE)
            seen = np.nonzero(l_ref)[0]
            for t in seen:
                assert p_dev[2 * t].tobytes() == p_ref[2 * t].tobytes() and p_dev[2 * t + 1].tobytes() == p_ref[2 * t + 1].tobytes()
    finally:
        det.close()


def test_host_batch_in_chunks_equals_the_device_batch(family):
    """Frames handed over in host memory (asl_detect_batch_pose_u8 / asl_detect_batch_u8) are copied and processed in chunks
    of 64 so that the transfer of one chunk hides the kernels of the one before; the results, appended chunk by chunk,
    must be those of one batch over the same frames resident on the device: same order, same frame indices, same bits."""
    import torch
    w, h, nd = 320, 240, 12
    distinct = np.stack([_scene(w, h, 2 + s % 4, 9000 + s, noise=float(s % 2)) for s in range(nd)])
    n = 150  # three chunks, the last one ragged
    frames = np.ascontiguousarray(distinct[np.arange(n) % nd])
    frames[37] = 128  # a frame without detections in the middle of a chunk
    K = synth.camera_matrix(w, h)
    det = _lib.Detector("tagStandard41h12", id_limit=0)
    try:
        d_frames = torch.from_numpy(frames).to("cuda:0")
        rd, rp, rn = det.detect_device(d_frames.data_ptr(), n, 3, w, h, K=K, dist=np.zeros(4), tag_size=10.0)
        rd, rp, rn = rd.copy(), rp.copy(), rn.copy()
        hd, hp, hn = det.detect_host(frames, K=K, dist=np.zeros(4), tag_size=10.0)
        assert np.array_equal(hn, rn) and len(hd) == len(rd) > 2 * n
        assert hd.tobytes() == rd.tobytes() and hp.tobytes() == rp.tobytes()
        assert np.array_equal(hd["frame"], np.repeat(np.arange(n), rn))
        gd, gn = det.detect_host(frames)  # without poses: asl_detect_batch_u8, same chunks
        assert np.array_equal(gn, rn) and gd.tobytes() == rd.tobytes()
        small, sp, sn = det.detect_host(frames[:100], K=K, dist=np.zeros(4), tag_size=10.0)  # fewer than two chunks: one batch
        assert small.tobytes() == rd[:len(small)].tobytes() and np.array_equal(sn, rn[:100])
    finally:
        det.close()


def test_quad_fit_gives_the_same_quads_every_time():
    """The quad fit is a pure function of the clusters: run again and again on the buffers one batch left behind
    (asl_debug_fetch item 7) it must return the same quads.  Round 3 found a flag shared by three checks in k_fit_quads
    that let the two wavefronts of a workgroup fall one barrier apart: one quad lost in ~170 batches of 1024 frames,
    invisible to every comparison against the oracle (tools/race_hunt.py is the long form of this test)."""
    import torch

    import bench
    det = _lib.Detector("tagStandard41h12", decimate=2.0, id_limit=0)
    try:
        dev = torch.device("cuda", 0)
        B = 256
        d_frames, _, _ = bench.render_stream_device(det, B, dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        K = synth.camera_matrix(bench.W, bench.H)
        det.submit_device(d_frames.data_ptr(), B, 3, bench.W, bench.H, stream=st, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
        dets, _, _ = det.collect(max_per_frame=bench.MAXDET)
        assert len(dets) == 20 * B
        r = det.debug_refit(3000)
        assert int(r[0]) == 3000 and int(r[1]) == 0, r.tolist()
    finally:
        det.close()


def test_shared_reciprocal_division_is_the_ieee_quotient():
    """asl_common.h's div_by(a, recip_of(d)) -- the five quotients of a line fit over one denominator -- against the
    compiler's a / d, as compiled into the shipped library: bit for bit on 2^29 random pairs, exponents within +-100 and
    +-400 (the quad path's operands are within 2^+-80)."""
    det = _lib.Detector("tagStandard41h12")
    try:
        for lim in (100, 400):
            pairs, bad = det.debug_division_check(lim)
            assert pairs == 2048 * 256 * 1024 and bad == 0, (lim, pairs, bad)
    finally:
        det.close()


def test_a_batch_gives_the_same_bytes_every_time():
    """400 runs of one 256-frame batch through one detector: detections, poses and per-frame counts byte for byte the same
    (atomics decide where things are stored, never what).  The race test_quad_fit_gives_the_same_quads_every_time is about
    changed one run in ~170 at four times this batch size; tools/race_hunt.py reports such a difference stage by stage."""
    import torch

    import bench
    det = _lib.Detector("tagStandard41h12", decimate=2.0, id_limit=0)
    try:
        dev = torch.device("cuda", 0)
        B = 256
        d_frames, _, _ = bench.render_stream_device(det, B, dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        K = synth.camera_matrix(bench.W, bench.H)
        first = None
        for it in range(400):
            det.submit_device(d_frames.data_ptr(), B, 3, bench.W, bench.H, stream=st, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
            dets, poses, npf = det.collect(max_per_frame=bench.MAXDET)
            cur = (dets.tobytes(), poses.tobytes(), npf.tobytes())
            if first is None:
                first = cur
                assert len(dets) == 20 * B
            else:
                assert cur == first, "repetition %d differs from the first" % it
    finally:
        det.close()
