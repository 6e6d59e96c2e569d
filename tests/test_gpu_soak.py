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


def test_observation_gather_device_path(monkeypatch):
    """The device path of all_gather_observations (cached device and page-locked buffers, asynchronous copies) with
    the collective itself replaced by a local stand-in -- a one-GPU box cannot host two RCCL ranks.  The gloo tests
    cover the real collective on CPU tensors."""
    import torch
    import torch.distributed as dist
    from aprilslam_amd import dist as adist

    def fake_all_gather(outs, t):
        for r, o in enumerate(outs):
            o.copy_(t + r)

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda: 3)
    monkeypatch.setattr(dist, "all_gather", fake_all_gather)
    dev = torch.device("cuda", 0)
    buf = adist.pinned_observation_buffer(16, 24)
    rng = np.random.default_rng(5)
    for it in range(3):  # second and third call reuse the cached buffers
        buf[...] = rng.normal(size=buf.shape)
        got = adist.all_gather_observations(buf, device=dev)
        assert got.shape == (3,) + buf.shape
        for r in range(3):
            assert np.array_equal(got[r], buf + r)
