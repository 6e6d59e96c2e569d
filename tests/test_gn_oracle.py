"""The pose-graph back-end has no reference counterpart (slam_graph.py:72-76 is a stub): its CPU restatement
(oracle/gn_oracle.py) is checked against ground truth and against an independent SciPy solve."""
import numpy as np
from scipy.optimize import least_squares

from gn_problem import G, make_problem


def test_gn_oracle_converges_to_ground_truth():
    pr = make_problem(P=10, L=5, seed=1, noise=0.0)
    cam, tag, st = G.solve(pr["cam0"], pr["tag0"], pr["obs_cam"], pr["obs_tag"], pr["obs_corners"], pr["K"], 10.0, 0, iters=15)
    assert st[1] < 1e-12 * max(st[0], 1.0) and st[2] >= 5
    assert np.abs(tag - pr["tag_gt"]).max() < 1e-6
    assert np.abs(cam - pr["cam_gt"]).max() < 1e-6


def test_gn_oracle_matches_scipy_minimum():
    pr = make_problem(P=6, L=4, seed=2, noise=0.3)
    cam, tag, st = G.solve(pr["cam0"], pr["tag0"], pr["obs_cam"], pr["obs_tag"], pr["obs_corners"], pr["K"], 10.0, 0, iters=25)
    P, L = len(pr["cam0"]), len(pr["tag0"])
    W0 = [np.linalg.inv(T) for T in pr["cam0"]]

    def unpack(x):
        W = [G.apply_update(W0[f], x[6 * f:6 * f + 6]) for f in range(P)]
        Gs = [pr["tag0"][0]] + [G.apply_update(pr["tag0"][j], x[6 * P + 6 * (j - 1):6 * P + 6 * j]) for j in range(1, L)]
        return W, Gs

    def fun(x):
        W, Gs = unpack(x)
        return G.linearize(W, Gs, pr["obs_cam"], pr["obs_tag"], pr["obs_corners"], pr["K"], 10.0)[1].ravel()

    sol = least_squares(fun, np.zeros(6 * P + 6 * (L - 1)), method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    cost_scipy = float((sol.fun ** 2).sum())
    assert abs(st[1] - cost_scipy) < 1e-5 * cost_scipy and st[1] <= cost_scipy * (1 + 1e-9)
    Ws, Gs = unpack(sol.x)
    assert np.abs(np.array(Gs) - tag).max() < 2e-2  # SciPy stops a little short of the minimum (its cost is 1e-6 higher)
