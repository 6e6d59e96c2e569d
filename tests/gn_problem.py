"""Seeded synthetic pose-graph problems shared by the GN tests (TEST INFRASTRUCTURE)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gn_oracle as G  # noqa: E402

from aprilslam_amd import synth  # noqa: E402


def make_problem(P=12, L=6, seed=0, noise=0.2, pert=0.05, width=1280, height=720):
    rng = np.random.default_rng(seed)
    K = synth.camera_matrix(width, height)
    tags = synth.random_scene(width, height, L, rng)
    tag_T = np.array([synth.tag_model_matrix(t["position"], t["rotation"]) for t in tags])  # world<-tag (GL world)
    cam_T = []
    for _ in range(P):
        V = synth.view_matrix(rng.uniform(-3, 3, 3), rng.uniform(-2, 2, 3))                   # GL camera<-world
        Wf = np.eye(4)
        Wf[:3, :3] = synth._FLIP @ V[:3, :3]
        Wf[:3, 3] = synth._FLIP @ V[:3, 3]                                                    # CV camera<-world
        cam_T.append(np.linalg.inv(Wf))
    cam_T = np.array(cam_T)
    X = G.corners_obj(10.0)
    oc, ot, ocorn = [], [], []
    for f in range(P):
        Wf = np.linalg.inv(cam_T[f])
        for j in range(L):
            c = []
            for k in range(4):
                p = Wf[:3, :3] @ (tag_T[j][:3, :3] @ X[k] + tag_T[j][:3, 3]) + Wf[:3, 3]
                c.append([K[0, 0] * p[0] / p[2] + K[0, 2], K[1, 1] * p[1] / p[2] + K[1, 2]])
            c = np.array(c)
            if (c[:, 0] > 0).all() and (c[:, 0] < width).all() and (c[:, 1] > 0).all() and (c[:, 1] < height).all() \
                    and (j == 0 or rng.uniform() < 0.85):
                oc.append(f); ot.append(j); ocorn.append(c + rng.normal(0, noise, (4, 2)))

    def perturb(T):
        d = np.concatenate([rng.normal(0, pert * 0.2, 3), rng.normal(0, pert * 20, 3)])
        return G.apply_update(T, d)

    cam0 = np.array([perturb(T) for T in cam_T])
    tag0 = np.array([tag_T[0]] + [perturb(T) for T in tag_T[1:]])
    return dict(K=K, cam_gt=cam_T, tag_gt=tag_T, cam0=cam0, tag0=tag0, obs_cam=np.array(oc, np.int32),
                obs_tag=np.array(ot, np.int32), obs_corners=np.array(ocorn))
