"""`SLAM` facade: host-side mirror of reference src/core/slam.py:9-97.

Per-frame contract used by the reference harness (simulation_engine.py:219-238):
    detections = slam.detect(frame); for d in detections: slam.get_pose(d); pose = slam.my_pose()
`process_observations` is the batched entry the multi-GPU path uses after the all-gather.
The matplotlib visualiser of the reference (slam_visualizer.py) is out of scope; the three
plotting methods are kept callable and do nothing unless a visualiser object is supplied.
"""
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


class SLAM:
    """Facade with the reference's constructor and methods (slam.py:9-97); detector and visualiser can be injected."""

    def __init__(self, logger, camera_params, tag_type="tagStandard41h12", tag_size=0.06, detector=None,
                 visualizer=None, device=0):
        self.logger = logger
        self.logger.info("Initializing SLAM")
        self.detector = detector if detector is not None else TagDetector(camera_params, tag_type, tag_size, device=device)
        self.graph = SLAMGraph(logger)
        self.visualizer = visualizer if visualizer is not None else _NullVisualizer()
        self.visible_tags = []

    def detect(self, image):
        detections = self.detector.detect(image)
        self.visible_tags = [d['id'] for d in detections]
        return detections

    def get_pose(self, detection):
        retval, rvec, tvec, T = self.detector.get_pose(detection)
        if retval:
            self.graph.add_or_update_node(detection['id'], T, self.visible_tags)
        return retval, rvec, tvec

    def process_observations(self, ids, transforms, oks=None):
        """One frame's observations (ids ascending, T camera<-tag each) -> graph update + my_pose()."""
        self.visible_tags = [int(i) for i in ids]
        for k, tag_id in enumerate(self.visible_tags):
            if oks is None or oks[k]:
                self.graph.add_or_update_node(tag_id, np.asarray(transforms[k], dtype=np.float64), self.visible_tags)
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
        return estimate

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
