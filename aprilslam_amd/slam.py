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
    (slam_graph.py:72-76 `update_world` is a stub, docs/api/core/SLAM.md:255-260).  With window = 0 every method
    behaves exactly like the reference's."""

    def __init__(self, logger, camera_params, tag_type="tagStandard41h12", tag_size=0.06, detector=None,
                 visualizer=None, device=0, window=0):
        self.logger = logger
        self.logger.info("Initializing SLAM")
        self.detector = detector if detector is not None else TagDetector(camera_params, tag_type, tag_size, device=device)
        self.graph = SLAMGraph(logger)
        self.visualizer = visualizer if visualizer is not None else _NullVisualizer()
        self.visible_tags = []
        self.camera_matrix = np.asarray(camera_params['camera_matrix'], dtype=np.float64)
        self.tag_size = tag_size
        self.window = int(window)
        self._frames = collections.deque(maxlen=self.window) if self.window > 0 else None  # (pose, [(id, corners 4x2)])
        self._pending = []
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
                self._pending.append((int(detection['id']), np.array(detection['lb-rb-rt-lt'], dtype=np.float64)))
        return retval, rvec, tvec

    def process_observations(self, ids, transforms, oks=None, corners=None):
        """One frame's observations (ids ascending, T camera<-tag each) -> graph update + my_pose().  `corners`
        (n, 4, 2) are only needed when a window is kept for optimize()."""
        self.visible_tags = [int(i) for i in ids]
        for k, tag_id in enumerate(self.visible_tags):
            if oks is None or oks[k]:
                self.graph.add_or_update_node(tag_id, np.asarray(transforms[k], dtype=np.float64), self.visible_tags)
                if self._frames is not None and corners is not None:
                    self._pending.append((tag_id, np.array(corners[k], dtype=np.float64).reshape(4, 2)))
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
            if estimate is not None and self._pending:
                self._frames.append((estimate.copy(), self._pending))
            self._pending = []
        return estimate

    # -- pose-graph back-end (not in the reference) -----------------------------------------------------------------
    def optimize(self, iters=10, backend=None):
        """Levenberg-Marquardt over the kept window on the device (asl_gn_solve): unknowns are the window's camera poses
        and the poses of the tags seen in it, residuals the pixel reprojection errors of the tag corners; the world tag
        stays fixed.  Refined tag poses replace `node.world`, the refined last camera pose `estimated_pose`.
        Returns {"cost0", "cost", "accepted", "cameras", "tags", "observations"} or None if there is nothing to do."""
        if self._frames is None:
            raise RuntimeError("SLAM(window=N) keeps no observations with N = 0: nothing to optimise")
        frames = list(self._frames)
        return self.optimize_window([f[0] for f in frames], [f[1] for f in frames], iters=iters, backend=backend)

    def optimize_window(self, cam_poses, frame_obs, iters=10, backend=None):
        """cam_poses: world<-camera 4x4 per frame (initial guesses, e.g. my_pose()); frame_obs: per frame a list of
        (tag id, corners 4x2).  See optimize()."""
        nodes = self.graph.get_nodes()
        c = self.coordinate_id
        if backend is None:
            backend = getattr(getattr(self.detector, "detector", None), "_det", None)
        if backend is None or not hasattr(backend, "gn_solve"):
            raise RuntimeError("optimize() needs the device back-end (a TagDetector, or backend=_lib.Detector)")
        tag_ids = sorted({t for obs in frame_obs for t, _ in obs if t in nodes})
        if c not in tag_ids or len(tag_ids) < 2 or not frame_obs:
            return None
        index = {t: k for k, t in enumerate(tag_ids)}
        obs_cam, obs_tag, obs_corners = [], [], []
        for f, obs in enumerate(frame_obs):
            for t, corners in obs:
                if t in index:
                    obs_cam.append(f); obs_tag.append(index[t]); obs_corners.append(corners)
        cam0 = np.array([project_se3(T) for T in cam_poses])
        tag0 = np.array([project_se3(nodes[t].world) for t in tag_ids])
        cam1, tag1, st = backend.gn_solve(cam0, tag0, obs_cam, obs_tag, np.array(obs_corners), self.camera_matrix, self.tag_size,
                                          fixed_tag=index[c], iters=iters)
        if st[1] < st[0]:  # a step was accepted: take the refined map
            for t in tag_ids:
                if t != c:
                    nodes[t].world = tag1[index[t]].copy()
                    nodes[t].updated = True
            self.graph.estimated_pose = cam1[-1].copy()
        return {"cost0": float(st[0]), "cost": float(st[1]), "accepted": int(st[2]), "cameras": len(cam0), "tags": len(tag_ids),
                "observations": len(obs_cam), "camera_poses": cam1}

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
