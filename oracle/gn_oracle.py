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


def _hat_many(v):
    """(..., 3) -> (..., 3, 3) cross-product matrices"""
    out = np.zeros(v.shape[:-1] + (3, 3))
    out[..., 0, 1] = -v[..., 2]; out[..., 0, 2] = v[..., 1]
    out[..., 1, 0] = v[..., 2]; out[..., 1, 2] = -v[..., 0]
    out[..., 2, 0] = -v[..., 1]; out[..., 2, 1] = v[..., 0]
    return out


def linearize(W, G, obs_cam, obs_tag, obs_corners, K, tag_size):
    """Returns cost, per-observation residuals (M,8), J_cam (M,8,6), J_tag (M,8,6).  All observations at once (the
    1,200-unknown system of the 4K / 200-tag shape takes seconds, not minutes); the formulas are the scalar ones."""
    X = corners_obj(tag_size)
    W = np.asarray(W, dtype=np.float64); G = np.asarray(G, dtype=np.float64)
    oc = np.asarray(obs_cam, dtype=np.int64); ot = np.asarray(obs_tag, dtype=np.int64)
    M = len(oc)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    Wf, Gj = W[oc], G[ot]                                                     # (M,4,4)
    q = np.einsum('mij,kj->mki', Gj[:, :3, :3], X) + Gj[:, None, :3, 3]       # (M,4,3) world
    p = np.einsum('mij,mkj->mki', Wf[:, :3, :3], q) + Wf[:, None, :3, 3]      # (M,4,3) camera
    iz = 1.0 / p[..., 2]
    u = np.stack([fx * p[..., 0] * iz + cx, fy * p[..., 1] * iz + cy], axis=-1)   # (M,4,2)
    r = (u - np.asarray(obs_corners, dtype=np.float64).reshape(M, 4, 2)).reshape(M, 8)
    Jp = np.zeros((M, 4, 2, 3))
    Jp[..., 0, 0] = fx * iz; Jp[..., 0, 2] = -fx * p[..., 0] * iz * iz
    Jp[..., 1, 1] = fy * iz; Jp[..., 1, 2] = -fy * p[..., 1] * iz * iz
    I3 = np.broadcast_to(np.eye(3), (M, 4, 3, 3))
    dp_dd = np.concatenate([-_hat_many(p), I3], axis=-1)                       # d p / d (omega, v) of the camera
    dp_de = np.einsum('mij,mkjl->mkil', Wf[:, :3, :3], np.concatenate([-_hat_many(q), I3], axis=-1))  # ... of the tag
    Jc = np.einsum('mkij,mkjl->mkil', Jp, dp_dd).reshape(M, 8, 6)
    Jt = np.einsum('mkij,mkjl->mkil', Jp, dp_de).reshape(M, 8, 6)
    return float((r * r).sum()), r, Jc, Jt


def lm_step(W, G, obs_cam, obs_tag, obs_corners, K, tag_size, fixed_tag, lam):
    """One damped Gauss-Newton step via the camera Schur complement.  Returns (cost, dW, dG)."""
    P, L = len(W), len(G)
    oc = np.asarray(obs_cam, dtype=np.int64); ot = np.asarray(obs_tag, dtype=np.int64)
    cost, r, Jc, Jt = linearize(W, G, oc, ot, obs_corners, K, tag_size)
    Hcc = np.zeros((P, 6, 6)); gc = np.zeros((P, 6)); Hll = np.zeros((L, 6, 6)); gl = np.zeros((L, 6))
    np.add.at(Hcc, oc, np.einsum('mki,mkj->mij', Jc, Jc)); np.add.at(gc, oc, np.einsum('mki,mk->mi', Jc, r))
    np.add.at(Hll, ot, np.einsum('mki,mkj->mij', Jt, Jt)); np.add.at(gl, ot, np.einsum('mki,mk->mi', Jt, r))
    Wb = np.einsum('mki,mkj->mij', Jc, Jt)                                     # per observation: camera x tag block
    d = np.arange(6)
    Hcc[:, d, d] += lam * np.maximum(Hcc[:, d, d], 1e-12)
    n = 6 * L
    S = np.zeros((n, n)); b = (-gl).reshape(n).copy()
    Hl = Hll.copy()
    Hl[:, d, d] += lam * np.maximum(Hll[:, d, d], 1e-12)
    for j in range(L):
        S[6 * j:6 * j + 6, 6 * j:6 * j + 6] = Hl[j]
    Hinv = np.zeros_like(Hcc)
    seen = [np.flatnonzero(oc == f) for f in range(P)]
    for f in range(P):
        idx = seen[f]
        if len(idx) == 0:
            continue
        Hinv[f] = np.linalg.inv(Hcc[f])
        Wf = Wb[idx]                                                            # (nj,6,6)
        js = ot[idx]
        HW = np.einsum('ij,njk->nik', Hinv[f], Wf)                              # Hinv Wb[j2]
        blocks = np.einsum('aji,bjk->abik', Wf, HW)                             # Wb[j]^T Hinv Wb[j2]
        rows = (6 * js[:, None] + d[None, :]).reshape(-1)
        S[np.ix_(rows, rows)] -= blocks.transpose(0, 2, 1, 3).reshape(len(rows), len(rows))
        b[rows] += np.einsum('nji,j->ni', Wf, Hinv[f] @ gc[f]).reshape(-1)
    # gauge: the fixed tag does not move; tags never observed: identity rows
    keep = np.ones(n, bool)
    keep[6 * fixed_tag:6 * fixed_tag + 6] = False
    unseen = ~np.any(Hll.reshape(L, 36) != 0, axis=1)
    keep[np.repeat(unseen, 6)] = False
    xl = np.zeros(n)
    xl[keep] = np.linalg.solve(S[np.ix_(keep, keep)], b[keep])
    dG = xl.reshape(L, 6)
    dW = np.zeros((P, 6))
    for f in range(P):
        idx = seen[f]
        if len(idx) == 0:
            continue
        rhs = -gc[f] - np.einsum('nij,nj->i', Wb[idx], dG[ot[idx]])
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
