"""Tag graph of the SLAM back-end: host-side float64 mirror of the reference interface.

Mirrors reference src/core/slam_graph.py:5-89 (`Node`, `SLAMGraph`): same class names,
attributes, method names, return values, log lines and printed messages, so the
reference harness (simulation_engine.py:219-238, 307-353) runs against it unchanged.
The update is sequential, order-dependent and a handful of 4x4 float64 products per tag
(SURVEY.md section 3.3), so it stays on the host; results are bit-identical to the
reference's (tests/test_graph_parity.py against fixtures produced by the reference code).

Semantics kept on purpose, including the reference's quirks:
  * `local` is always inv(T) (camera pose in the tag frame)               slam_graph.py:25-27
  * the lowest tag id ever seen becomes the world frame; switching to a lower id does
    NOT re-express existing nodes (`update_world` is a stub)              slam_graph.py:36-39,72-76
  * reference tag of a frame = min(visible ids)                           slam_graph.py:41
"""
import numpy as np


class Node:
    """One tag: `local` = tag<-camera 4x4, `world` = world<-tag 4x4, chain bookkeeping."""

    def __init__(self, local, world, reference, weight=1, updated=True, visible=False):
        self.local = local
        self.world = world
        self.reference = reference
        self.weight = weight
        self.updated = updated
        self.visible = visible


class SLAMGraph:
    def __init__(self, logger):
        self.logger = logger
        self.graph = {}
        self.visible_tags = []
        self.coordinate_id = -1
        self.estimated_pose = np.zeros((4, 4))
        self.world_updater = None  # set by SLAM(window=N): what the reference's update_world stub was meant to become

    # -- small helpers (same names as the reference) ---------------------------------
    def invert(self, T):
        return np.linalg.inv(T)

    def get_world(self, reference, T):
        return self.graph[reference].local @ T

    def find_world(self, reference, T):
        ref = self.graph[reference]
        world = np.matmul(ref.world, self.get_world(reference, T))
        return world, ref.weight + 1, ref.reference

    def update_world(self):
        # the reference leaves this unimplemented and announces it (slam_graph.py:72-76); with a back-end attached the
        # map is re-optimised instead
        if self.world_updater is None:
            print("No world update")
        else:
            self.world_updater()

    # -- the per-observation update --------------------------------------------------
    def add_or_update_node(self, tag_id, T, visible_tags):
        self.visible_tags = visible_tags
        cid = self.coordinate_id

        if cid == -1 or cid == tag_id:            # branch A: (re)observe the world tag
            self.coordinate_id = tag_id
            self.graph[tag_id] = Node(self.invert(T), np.eye(4), tag_id)
            return
        if tag_id < cid:                          # branch B: a lower id takes over as world
            self.coordinate_id = tag_id
            self.graph[tag_id] = Node(self.invert(T), np.eye(4), tag_id)
            self.update_world()
            return

        reference = min(self.visible_tags)
        if reference == cid:                      # C1: world tag is in view
            self.graph[tag_id] = Node(self.invert(T), self.get_world(reference, T), cid)
            length = np.linalg.norm(self.get_world(reference, T)[:3, 3])
            self.logger.info(f"Tag ID {tag_id} (reference: {reference}): World transform translation length = {length}")
        elif tag_id in self.graph and self.graph[tag_id].reference == cid:   # C2: keep the direct estimate
            old = self.graph[tag_id]
            self.logger.info(f"World not updated! Detection ID: {tag_id}, Node World: {old.world}")
            self.graph[tag_id] = Node(self.invert(T), old.world, cid, weight=old.weight, updated=False)
        elif reference != tag_id and reference in self.graph:                # C3: chain through the reference tag
            world, weight, new_reference = self.find_world(reference, T)
            self.graph[tag_id] = Node(self.invert(T), world, new_reference, weight,
                                      updated=self.graph[reference].updated)
        else:                                                                # C4
            print("Cannot find world reference")

    # -- getters ---------------------------------------------------------------------
    def get_nodes(self):
        return self.graph

    def get_coordinate_id(self):
        return self.coordinate_id

    def get_estimated_pose(self):
        return self.estimated_pose
