"""`SLAM` facade: host-side mirror of reference src/core/slam.py:9-97.

Per-frame contract used by the reference harness (simulation_engine.py:219-238):
    detections = slam.detect(frame); for d in detections: slam.get_pose(d); pose = slam.my_pose()
`process_observations` is the batched entry the multi-GPU path uses after the all-gather.
The matplotlib visualiser of the reference (slam_visualizer.py) is out of scope; the three
plotting methods are kept callable and do nothing unless a visualiser object is supplied.
"""
import collections

import numpy as np

from .slam_graph import SLAMGraph
from .tag_detector import TagDetector


class _NullVisualizer:
    def slam_graph(self, *a, **k):
        pass

    def vis_slam(self, *a, **k):
        pass

    def error_graph(self, *a, **k):
        pass


def fuse_camera_pose(nodes, visible_ids):
    """Every visible tag that is already a node votes with world @ local (the camera pose it implies), weighted by
    1 / chain length; the votes are averaged ELEMENT-WISE (the result is in general not orthonormal -- that is the
    reference's estimator and it is reproduced bit for bit: same operand order, same accumulation order).
    Side effect, as in the reference: node.visible is refreshed for all nodes."""
    for tag in nodes.values():
        tag.visible = False
    acc = np.zeros((4, 4))
    norm = 0
    for tag_id in visible_ids:
        tag = nodes.get(tag_id)
        if tag is None:
            continue
        tag.visible = True
        vote = np.matmul(tag.world, tag.local)
        acc += vote / tag.weight
        norm += 1 / tag.weight
    if norm == 0:
        return None
    return acc / norm


def project_se3(T):
    """Nearest rigid transform to a 4x4 whose rotation block is only approximately orthonormal (my_pose() averages
    matrices element-wise, slam.py:36-63)."""
    out = np.eye(4)
    U, _, Vt = np.linalg.svd(np.asarray(T, dtype=np.float64)[:3, :3])
    if np.linalg.det(U @ Vt) < 0:
        U[:, 2] = -U[:, 2]
    out[:3, :3] = U @ Vt
    out[:3, 3] = T[:3, 3]
    return out


class SLAM:
    """Facade with the reference's constructor and methods (slam.py:9-97); detector and visualiser can be injected.

    `window` > 0 (NOT in the reference, off by default) keeps the observations of the last `window` frames and makes
    `optimize()` available: the pose-graph Levenberg-Marquardt back-end the reference lists as future work
    (slam_graph.py:72-76 `update_world` is a stub, docs/api/core/SLAM.md:255-260).  It also keeps up to `keyframes`
    keyframes (every `keyframe_every`-th frame) of the whole run: when a tag comes back into view after it has been out
    of the window (a loop closure), `optimize_global()` re-solves the map over the keyframes and the window.
    With window = 0 every method behaves exactly like the reference's."""

    def __init__(self, logger, camera_params, tag_type="tagStandard41h12", tag_size=0.06, detector=None,
                 visualizer=None, device=0, window=0, keyframes=64, keyframe_every=None):
        self.logger = logger
        self.logger.info("Initializing SLAM")
        self.detector = detector if detector is not None else TagDetector(camera_params, tag_type, tag_size, device=device)
        self.graph = SLAMGraph(logger)
        self.visualizer = visualizer if visualizer is not None else _NullVisualizer()
        self.visible_tags = []
        self.camera_matrix = np.asarray(camera_params['camera_matrix'], dtype=np.float64)
        self.tag_size = tag_size
        self.window = int(window)
        self._frames = collections.deque(maxlen=self.window) if self.window > 0 else None  # (pose, [(id, corners 4x2, T)])
        self._pending = []
        self._keyframes = collections.deque(maxlen=max(1, int(keyframes))) if self.window > 0 else None
        self._keyframe_every = int(keyframe_every) if keyframe_every else max(1, self.window)
        self._frame_no = 0
        self._last_seen = {}     # tag id -> number of the last frame that observed it
        self.loop_closures = 0   # times a tag came back after leaving the window (each one ran optimize_global)
        self.last_optimize = None
        self.lm_backend = None   # object with gn_solve(...) (a _lib.Detector); default: the TagDetector's own
        if self.window > 0:
            # a switch of the world frame (branch B of the update) is where the reference meant to re-express the map
            self.graph.world_updater = lambda: self.optimize()

    def detect(self, image):
        detections = self.detector.detect(image)
        self.visible_tags = [d['id'] for d in detections]
        return detections

    def get_pose(self, detection):
        retval, rvec, tvec, T = self.detector.get_pose(detection)
        if retval:
            self.graph.add_or_update_node(detection['id'], T, self.visible_tags)
            if self._frames is not None:
                self._pending.append((int(detection['id']), np.array(detection['lb-rb-rt-lt'], dtype=np.float64), np.array(T, dtype=np.float64)))
        return retval, rvec, tvec

    def process_observations(self, ids, transforms, oks=None, corners=None):
        """One frame's observations (ids ascending, T camera<-tag each) -> graph update + my_pose().  `corners`
        (n, 4, 2) are only needed when a window is kept for optimize()."""
        self.visible_tags = [int(i) for i in ids]
        for k, tag_id in enumerate(self.visible_tags):
            if oks is None or oks[k]:
                self.graph.add_or_update_node(tag_id, np.asarray(transforms[k], dtype=np.float64), self.visible_tags)
                if self._frames is not None and corners is not None:
                    self._pending.append((tag_id, np.array(corners[k], dtype=np.float64).reshape(4, 2), np.array(transforms[k], dtype=np.float64)))
        return self.my_pose()

    def my_pose(self):
        """Camera pose in the world frame of the graph, or None when no visible tag is in the graph yet.
        Same arithmetic as the reference (slam.py:36-63), see `fuse_camera_pose`."""
        seen = self.visible_tags
        if not seen:
            return None
        estimate = fuse_camera_pose(self.graph.get_nodes(), seen)
        if estimate is not None:
            self.graph.estimated_pose = estimate
        if self._frames is not None:
            self._end_of_frame(estimate)
        return estimate

    def _end_of_frame(self, estimate):
        """window / keyframe bookkeeping of one frame, and the loop-closure trigger"""
        pending, self._pending = self._pending, []
        if estimate is None or not pending:
            return
        self._frame_no += 1  # frames that carried observations
        entry = (estimate.copy(), pending)
        self._frames.append(entry)
        if not self._keyframes or self._frame_no - self._keyframes[-1][0] >= self._keyframe_every:
            self._keyframes.append((self._frame_no, entry))
        # a tag of the map that has not been observed for longer than the window holds frames: the camera is back at a
        # place it has seen before, and the drift accumulated since then can be distributed over the whole map
        back = [t for t, _, _ in pending if t in self._last_seen and self._frame_no - self._last_seen[t] > self.window]
        for t, _, _ in pending:
            self._last_seen[t] = self._frame_no
        if back:
            self.loop_closures += 1
            self.logger.info(f"Loop closure: tags {back} are back in view; global solve over {len(self._keyframes)} keyframes")
            self.optimize_global()

    # -- pose-graph back-end (not in the reference) -----------------------------------------------------------------
    def optimize(self, iters=10, backend=None):
        """Levenberg-Marquardt over the kept window on the device (asl_gn_solve): unknowns are the window's camera poses
        and the poses of the tags seen in it, residuals the pixel reprojection errors of the tag corners; the world tag
        stays fixed.  Refined tag poses replace `node.world`, the refined last camera pose `estimated_pose`.
        Returns {"cost0", "cost", "accepted", "cameras", "tags", "observations", ...} or None if there is nothing to do."""
        if self._frames is None:
            raise RuntimeError("SLAM(window=N) keeps no observations with N = 0: nothing to optimise")
        frames = list(self._frames)
        return self.optimize_window([f[0] for f in frames], [f[1] for f in frames], iters=iters, backend=backend)

    def optimize_global(self, iters=10, backend=None):
        """The same solve over the keyframes of the whole run plus the current window (loop closure): every tag the
        keyframes saw is an unknown, so a tag that comes back into view ties the two ends of the trajectory together.
        The mirrored-minimum test of the tags (map_init.flip_test_tags) runs after the first refinement."""
        if self._frames is None:
            raise RuntimeError("SLAM(window=N) keeps no observations with N = 0: nothing to optimise")
        in_window = {id(f) for f in self._frames}
        frames = [e for _, e in self._keyframes if id(e) not in in_window] + list(self._frames)
        return self.optimize_window([f[0] for f in frames], [f[1] for f in frames], iters=iters, backend=backend, flip_test=True)

    def optimize_window(self, cam_poses, frame_obs, iters=10, backend=None, seed=True, flip_test=False):
        """cam_poses: world<-camera 4x4 per frame (initial guesses, e.g. my_pose()); frame_obs: per frame a list of
        (tag id, corners 4x2) or (tag id, corners 4x2, T camera<-tag of the single-view PnP).  See optimize().

        Starting values: the graph's `world` poses are single chains through single-view PnP answers (stale after a world
        switch, mirrored whenever the world tag's PnP fell into the other planar minimum), so with seed=True and the
        per-observation poses at hand every camera and tag first picks the most consistent pose its observations imply
        (map_init.reseed_poses); an observation whose tag still lies behind its camera afterwards is left out of the
        solve (its residual is unbounded at the image plane and would swamp every other term)."""
        from . import map_init
        nodes = self.graph.get_nodes()
        c = self.coordinate_id
        if backend is None:
            backend = self.lm_backend
        if backend is None:
            backend = getattr(getattr(self.detector, "detector", None), "_det", None)
        if backend is None or not hasattr(backend, "gn_solve"):
            raise RuntimeError("optimize() needs the device back-end (a TagDetector, or backend=_lib.Detector)")
        tag_ids = sorted({o[0] for obs in frame_obs for o in obs if o[0] in nodes})
        if c not in tag_ids or len(tag_ids) < 2 or not frame_obs:
            return None
        index = {t: k for k, t in enumerate(tag_ids)}
        obs_cam, obs_tag, obs_corners, obs_T = [], [], [], []
        for f, obs in enumerate(frame_obs):
            for o in obs:
                if o[0] in index:
                    obs_cam.append(f); obs_tag.append(index[o[0]]); obs_corners.append(o[1])
                    if len(o) > 2 and o[2] is not None:
                        obs_T.append(o[2])
        obs_cam, obs_tag = np.asarray(obs_cam, dtype=np.int64), np.asarray(obs_tag, dtype=np.int64)
        obs_corners = np.asarray(obs_corners, dtype=np.float64).reshape(-1, 4, 2)
        cam0 = np.array([project_se3(T) for T in cam_poses])
        tag0 = np.array([project_se3(nodes[t].world) for t in tag_ids])
        seeded = bool(seed and len(obs_T) == len(obs_cam))
        if seeded:
            cam0, tag0 = map_init.reseed_poses(cam0, tag0, obs_cam, obs_tag, np.asarray(obs_T), obs_corners, self.camera_matrix,
                                               self.tag_size, fixed_tag=index[c], sweeps=2, max_cand=8)
        keep = ~map_init.behind_camera(cam0, tag0, obs_cam, obs_tag, self.tag_size)
        dropped = int((~keep).sum())
        if dropped:
            obs_cam, obs_tag, obs_corners = obs_cam[keep], obs_tag[keep], obs_corners[keep]
        if len(obs_cam) == 0:
            return None
        cam1, tag1, st = backend.gn_solve(cam0, tag0, obs_cam, obs_tag, obs_corners, self.camera_matrix, self.tag_size,
                                          fixed_tag=index[c], iters=iters)
        cost0, flipped = float(st[0]), []
        if flip_test and st[1] <= st[0]:
            tag1f, flipped = map_init.flip_test_tags(cam1, tag1, obs_cam, obs_tag, obs_corners, self.camera_matrix, self.tag_size, fixed_tag=index[c])
            if flipped:
                cam1, tag1, st2 = backend.gn_solve(cam1, tag1f, obs_cam, obs_tag, obs_corners, self.camera_matrix, self.tag_size,
                                                   fixed_tag=index[c], iters=iters)
                st = np.array([cost0, st2[1], st[2] + st2[2]])
        if st[1] < st[0] or (seeded and st[1] <= st[0]):  # a step was accepted (or the seeding alone repaired the map): take it
            seen = set(obs_tag.tolist())
            for t in tag_ids:
                if t != c and index[t] in seen:
                    nodes[t].world = tag1[index[t]].copy()
                    nodes[t].updated = True
            self.graph.estimated_pose = cam1[-1].copy()
        self.last_optimize = {"cost0": cost0, "cost": float(st[1]), "accepted": int(st[2]), "cameras": len(cam0), "tags": len(tag_ids),
                              "observations": int(len(obs_cam)), "observations_dropped_behind_camera": dropped, "seeded": seeded,
                              "tags_flipped": [int(tag_ids[j]) for j in flipped], "camera_poses": cam1}
        return self.last_optimize

    def average_distance_to_nodes(self):
        """Mean distance camera <-> tag over ALL nodes of the graph (0 for an empty graph), slam.py:65-80."""
        nodes = self.graph.get_nodes()
        if len(nodes) == 0:
            return 0
        dist_sum = 0
        for tag in nodes.values():
            dist_sum += np.linalg.norm(tag.local[:3, 3])
        return dist_sum / len(nodes)

    @property
    def coordinate_id(self):
        return self.graph.get_coordinate_id()

    def slam_graph(self):
        self.visualizer.slam_graph(self.graph.get_nodes())

    def vis_slam(self, ground_truth=None):
        self.visualizer.vis_slam(self.graph.get_nodes(), self.graph.get_estimated_pose(), ground_truth)

    def error_graph(self, ground_truth_graph):
        self.visualizer.error_graph(self.graph.get_nodes(), ground_truth_graph)
