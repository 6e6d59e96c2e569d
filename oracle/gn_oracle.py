"""oracle/gn_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (NumPy, float64) of the pose-graph Levenberg-Marquardt back-end behind
`asl_gn_solve` (include/aprilslam.h).  The reference has NO such back-end: `SLAMGraph.update_world`
is a stub (reference src/core/slam_graph.py:72-76) and its docs list bundle adjustment as TODO
(docs/api/core/SLAM.md:255-260), so this stage is build-defined and "parity unpinned"; it is judged
against simulation ground truth and this restatement.

Model: camera f has pose W_f (camera<-world), tag j has pose G_j (world<-tag); observation (f, j)
gives the 4 pixel corners of the tag (lb, rb, rt, lt = (-h,-h) (h,-h) (h,h) (-h,h), h = tag_size/2).
Residual = pinhole projection of W_f G_j X_k minus the observed corner (8 per observation).
Updates are left-multiplicative: W_f <- exp(d_f) W_f, G_j <- exp(e_j) G_j with d, e = (omega, v).
One LM step solves (H + lambda*diag(H)) x = -g by Schur complement on the cameras; a step is accepted
when the cost decreases (lambda *= 0.1) and rejected otherwise (lambda *= 10).
"""
import numpy as np


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def exp_rot(w):
    th = np.linalg.norm(w)
    if th < 1e-300:
        return np.eye(3)
    k = w / th
    K = hat(k)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def apply_update(T, d):
    """exp((omega, v)) * T with the first-order translation update used by the device code:
    R <- exp(omega) R, t <- exp(omega) t + v."""
    R = exp_rot(d[:3])
    out = np.eye(4)
    out[:3, :3] = R @ T[:3, :3]
    out[:3, 3] = R @ T[:3, 3] + d[3:]
    return out


def corners_obj(tag_size):
    h = float(np.float32(tag_size / 2))
    return np.array([[-h, -h, 0], [h, -h, 0], [h, h, 0], [-h, h, 0]], dtype=np.float64)


def linearize(W, G, obs_cam, obs_tag, obs_corners, K, tag_size):
    """Returns cost, per-observation residuals (M,8), J_cam (M,8,6), J_tag (M,8,6)."""
    X = corners_obj(tag_size)
    M = len(obs_cam)
    r = np.zeros((M, 8)); Jc = np.zeros((M, 8, 6)); Jt = np.zeros((M, 8, 6))
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    for m in range(M):
        Wf, Gj = W[obs_cam[m]], G[obs_tag[m]]
        for k in range(4):
            q = Gj[:3, :3] @ X[k] + Gj[:3, 3]
            p = Wf[:3, :3] @ q + Wf[:3, 3]
            iz = 1.0 / p[2]
            u = np.array([fx * p[0] * iz + cx, fy * p[1] * iz + cy])
            r[m, 2 * k:2 * k + 2] = u - obs_corners[m, k]
            Jp = np.array([[fx * iz, 0, -fx * p[0] * iz * iz], [0, fy * iz, -fy * p[1] * iz * iz]])
            dp_dd = np.hstack([-hat(p), np.eye(3)])                     # d p / d (omega, v) of the camera
            dp_de = Wf[:3, :3] @ np.hstack([-hat(q), np.eye(3)])         # d p / d (omega, v) of the tag
            Jc[m, 2 * k:2 * k + 2] = Jp @ dp_dd
            Jt[m, 2 * k:2 * k + 2] = Jp @ dp_de
    return float((r * r).sum()), r, Jc, Jt


def lm_step(W, G, obs_cam, obs_tag, obs_corners, K, tag_size, fixed_tag, lam):
    """One damped Gauss-Newton step via the camera Schur complement.  Returns (dW list, dG list)."""
    P, L = len(W), len(G)
    cost, r, Jc, Jt = linearize(W, G, obs_cam, obs_tag, obs_corners, K, tag_size)
    Hcc = np.zeros((P, 6, 6)); gc = np.zeros((P, 6)); Hll = np.zeros((L, 6, 6)); gl = np.zeros((L, 6))
    Wb = {}
    for m in range(len(obs_cam)):
        f, j = obs_cam[m], obs_tag[m]
        Hcc[f] += Jc[m].T @ Jc[m]; gc[f] += Jc[m].T @ r[m]
        Hll[j] += Jt[m].T @ Jt[m]; gl[j] += Jt[m].T @ r[m]
        Wb[(f, j)] = Jc[m].T @ Jt[m]
    for f in range(P):
        Hcc[f] += lam * np.diag(np.maximum(np.diag(Hcc[f]), 1e-12))
    n = 6 * L
    S = np.zeros((n, n)); b = np.zeros(n)
    for j in range(L):
        S[6 * j:6 * j + 6, 6 * j:6 * j + 6] = Hll[j] + lam * np.diag(np.maximum(np.diag(Hll[j]), 1e-12))
        b[6 * j:6 * j + 6] = -gl[j]
    Hinv = np.zeros_like(Hcc)
    seen = {f: [] for f in range(P)}
    for (f, j) in Wb:
        seen[f].append(j)
    for f in range(P):
        if not seen[f]:
            continue
        Hinv[f] = np.linalg.inv(Hcc[f])
        for j in seen[f]:
            b[6 * j:6 * j + 6] += Wb[(f, j)].T @ Hinv[f] @ gc[f]
            for j2 in seen[f]:
                S[6 * j:6 * j + 6, 6 * j2:6 * j2 + 6] -= Wb[(f, j)].T @ Hinv[f] @ Wb[(f, j2)]
    # gauge: the fixed tag does not move
    keep = np.ones(n, bool)
    keep[6 * fixed_tag:6 * fixed_tag + 6] = False
    # tags never observed: identity rows
    for j in range(L):
        if not np.any(Hll[j]):
            keep[6 * j:6 * j + 6] = False
    xl = np.zeros(n)
    xl[keep] = np.linalg.solve(S[np.ix_(keep, keep)], b[keep])
    dG = xl.reshape(L, 6)
    dW = np.zeros((P, 6))
    for f in range(P):
        if not seen[f]:
            continue
        rhs = -gc[f]
        for j in seen[f]:
            rhs = rhs - Wb[(f, j)] @ dG[j]
        dW[f] = Hinv[f] @ rhs
    return cost, dW, dG


def solve(cam_T, tag_T, obs_cam, obs_tag, obs_corners, K, tag_size, fixed_tag, iters=10):
    """cam_T: (P,4,4) world<-camera; tag_T: (L,4,4) world<-tag.  Returns (cam_T, tag_T, stats)."""
    W = [np.linalg.inv(T) for T in cam_T]
    G = [np.array(T, dtype=np.float64) for T in tag_T]
    lam = 1e-3
    cost0, _, _, _ = linearize(W, G, obs_cam, obs_tag, obs_corners, K, tag_size)
    cost = cost0
    accepted = 0
    for _ in range(iters):
        _, dW, dG = lm_step(W, G, obs_cam, obs_tag, obs_corners, K, tag_size, fixed_tag, lam)
        Wn = [apply_update(W[f], dW[f]) for f in range(len(W))]
        Gn = [apply_update(G[j], dG[j]) for j in range(len(G))]
        costn, _, _, _ = linearize(Wn, Gn, obs_cam, obs_tag, obs_corners, K, tag_size)
        if costn < cost:
            W, G, cost = Wn, Gn, costn
            lam = max(lam * 0.1, 1e-12)
            accepted += 1
        else:
            lam *= 10
    return np.array([np.linalg.inv(T) for T in W]), np.array(G), np.array([cost0, cost, accepted], dtype=np.float64)
