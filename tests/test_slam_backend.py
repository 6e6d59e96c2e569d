"""SLAM(window=N): starting values and triggers of the pose-graph back-end (NOT in the reference: slam_graph.py:72-76
is a stub, docs/api/core/SLAM.md:255-260 lists it as future work -- build-defined, parity unpinned).

The Levenberg-Marquardt step itself runs on the device (asl_gn_solve); here the CPU restatement oracle/gn_oracle.py
stands in for it through the same `gn_solve` interface, so the host logic (seeding through map_init, the guard against
tags behind the image plane, keyframes, the loop-closure trigger) is covered without a GPU.  The regression these tests
pin: configs[4] of BASELINE.json ran a window whose start had a tag behind a camera (the world tag's single-view PnP had
fallen into the mirrored planar minimum in the frame that saw that tag last): cost 1.1e16 -> 4.1e14 over 1,567
observations (profiles/r02_bench_configs4_n1.json)."""
import numpy as np

from gn_problem import G

from aprilslam_amd import map_init, synth
from aprilslam_amd.slam import SLAM

TAG = 10.0


class _Log:
    def __init__(self):
        self.lines = []

    def info(self, m):
        self.lines.append(m)


class OracleBackend:
    """gn_solve of _lib.Detector, answered by the NumPy restatement"""
    def __init__(self):
        self.calls = []

    def gn_solve(self, cam_T, tag_T, obs_cam, obs_tag, obs_corners, K, tag_size, fixed_tag=0, iters=10):
        self.calls.append((len(cam_T), len(tag_T), len(obs_cam)))
        return G.solve(np.asarray(cam_T), np.asarray(tag_T), list(obs_cam), list(obs_tag), np.asarray(obs_corners), K, tag_size, fixed_tag, iters)


def _project(K, T_cam_tag):
    X = map_init._corners_obj(TAG)
    p = (T_cam_tag[:3] @ X.T).T
    return np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1)


def _wall(n_tags, spacing=30.0, seed=0):
    """tags on a wall (z = 0 of the world tag), slightly tilted; tag 0 at the origin"""
    rng = np.random.default_rng(seed)
    tags = []
    for j in range(n_tags):
        T = np.eye(4)
        if j:
            w = rng.normal(size=3) * 0.15
            T[:3, :3] = G.exp_rot(w)
            T[:3, 3] = [spacing * j, rng.uniform(-5, 5), rng.uniform(-2, 2)]
        tags.append(T)
    return np.array(tags)


def _camera(x, rng):
    """camera in front of the wall at abscissa x, looking at it (CV axes: z forward)"""
    T = np.eye(4)
    T[:3, :3] = G.exp_rot(np.array([np.pi, 0.0, 0.0])) @ G.exp_rot(rng.normal(size=3) * 0.05)  # z towards -z of the world
    T[:3, 3] = [x + rng.uniform(-2, 2), rng.uniform(-3, 3), 140.0 + rng.uniform(-5, 5)]
    return T


def _slam(window, **kw):
    K = synth.camera_matrix(1280, 720)
    s = SLAM(_Log(), {"camera_matrix": K, "dist_coeffs": np.zeros(4)}, tag_size=TAG, detector=object(), window=window, **kw)
    s.lm_backend = OracleBackend()
    return s, K


def _feed(slam, K, tags, cam, ids, mirrored=()):
    Ts, corners = [], []
    for j in ids:
        T = np.linalg.inv(cam) @ tags[j]
        corners.append(_project(K, T))
        Ts.append(map_init.mirrored_pose(T) if j in mirrored else T)
    return slam.process_observations(list(ids), np.array(Ts), corners=np.array(corners))


def test_window_solve_survives_a_mirrored_world_tag():
    rng = np.random.default_rng(5)
    tags = _wall(6)
    slam, K = _slam(window=6)
    cams = [_camera(60.0 + 5 * f, rng) for f in range(6)]
    for f, cam in enumerate(cams):
        ids = [0, 1, 2, 3, 4] if f == 2 else [0, 1, 2, 3, 5] if f > 2 else [0, 1, 2, 3, 4, 5]
        # frame 2: the world tag's single-view pose is the mirrored planar minimum, and it is the last frame that sees tag 4
        _feed(slam, K, tags, cam, ids, mirrored=(0,) if f == 2 else ())
    world4 = slam.graph.get_nodes()[4].world
    assert np.abs(world4 - tags[4]).max() > 5.0  # the reference's graph keeps the pose chained through the bad frame
    frames = list(slam._frames)
    # without the seeding the solve starts from that map: tag 4 lies far from where four cameras saw it
    slam_plain, _ = _slam(window=6)
    slam_plain.graph = slam.graph
    unseeded = slam_plain.optimize_window([f[0] for f in frames], [[(o[0], o[1]) for o in f[1]] for f in frames], iters=0)
    assert unseeded["seeded"] is False and unseeded["cost0"] > 1e5
    res = slam.optimize(iters=8)
    assert res["seeded"] and res["observations_dropped_behind_camera"] == 0
    assert res["cost0"] < 1e-3 * unseeded["cost0"]
    assert res["cost"] / res["observations"] < 1e-9
    nodes = slam.graph.get_nodes()
    for j in range(1, 6):
        assert np.abs(nodes[j].world - tags[j]).max() < 1e-5, j
    assert np.array_equal(nodes[0].world, np.eye(4))


def test_observation_behind_the_camera_is_left_out_of_the_solve():
    rng = np.random.default_rng(7)
    tags = _wall(4)
    slam, K = _slam(window=3)
    for f in range(3):
        _feed(slam, K, tags, _camera(40.0 + 4 * f, rng), [0, 1, 2, 3])
    nodes = slam.graph.get_nodes()
    bad = nodes[3].world.copy()
    bad[:3, 3] = [45.0, 0.0, 400.0]  # behind every camera of the window
    nodes[3].world = bad
    frames = list(slam._frames)
    res = slam.optimize_window([f[0] for f in frames], [[(o[0], o[1]) for o in f[1]] for f in frames], iters=6)  # no PnP poses: no seeding
    assert res["seeded"] is False and res["observations_dropped_behind_camera"] == 3
    assert res["cost"] / res["observations"] < 1e-9 and res["observations"] == 9
    assert np.array_equal(nodes[3].world, bad)  # a tag without usable observations keeps its pose


def test_a_tag_that_comes_back_triggers_the_global_solve():
    rng = np.random.default_rng(11)
    tags = _wall(8)
    slam, K = _slam(window=3, keyframes=32, keyframe_every=2)
    xs = [0, 30, 60, 90, 120, 150, 180, 150, 120, 90, 60, 30, 0]  # out along the wall and back
    n_before = 0
    for f, x in enumerate(xs):
        near = sorted(j for j in range(8) if abs(30.0 * j - x) <= 45.0)
        if 0 not in near:
            near = [0] + near  # the world tag stays in view (large wall, wide lens) so that every frame is self-contained
        _feed(slam, K, tags, _camera(float(x), rng), near)
        if f == 6:
            assert slam.loop_closures == 0  # nothing has come back yet
            n_before = len(slam.lm_backend.calls)
    assert slam.loop_closures >= 1
    assert len(slam.lm_backend.calls) > n_before
    ncam, ntag, nobs = slam.lm_backend.calls[-1]
    assert ncam > slam.window and ntag == 8  # keyframes of the way out and the window of the way back, every tag of the map
    assert any("Loop closure" in ln for ln in slam.logger.lines)
    last = slam.last_optimize
    assert last["cost"] <= last["cost0"] and last["cost"] / last["observations"] < 1e-6
    nodes = slam.graph.get_nodes()
    for j in range(1, 8):
        assert np.abs(nodes[j].world - tags[j]).max() < 1e-4, j


def test_window_zero_is_the_reference():
    slam, K = _slam(window=0)
    tags = _wall(3)
    pose = _feed(slam, K, tags, _camera(20.0, np.random.default_rng(1)), [0, 1, 2])
    assert pose is not None and slam._frames is None and slam._keyframes is None and slam.loop_closures == 0
    try:
        slam.optimize()
    except RuntimeError:
        pass
    else:
        raise AssertionError("optimize() without a window must raise")
