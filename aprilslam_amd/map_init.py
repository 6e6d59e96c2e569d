"""Starting values for the pose-graph back-end (SLAM.optimize / asl_gn_solve).  Not in the reference: its graph chains
each tag once, keeps stale poses when the world tag changes (slam_graph.py:72-76, "No world update") and accepts
whichever of the two planar-PnP minima the solver lands in, all of which a Levenberg-Marquardt refinement inherits
as a bad start.  These helpers build a consistent start from the same per-observation PnP poses:

  chain_initial_map   breadth-first chaining over frames from the lowest tag id, always through the largest visible
                      tag already placed (no stale nodes, no order dependence)
  reseed_poses        for every camera / tag, try each pose its own observations imply (through the current map) and
                      keep the one with the smallest total reprojection error over ALL its observations: a mirrored
                      single-view PnP answer loses against any consistent one as soon as a second view exists

Host-side numpy on a few thousand 4x4s; the refinement itself runs on the device.
"""
import numpy as np


def _corners_obj(tag_size):
    h = float(np.float32(tag_size / 2))
    return np.array([[-h, -h, 0, 1], [h, -h, 0, 1], [h, h, 0, 1], [-h, h, 0, 1]], dtype=np.float64)


def _inv(T):
    out = np.zeros_like(T)
    R = np.swapaxes(T[..., :3, :3], -1, -2)
    out[..., :3, :3] = R
    out[..., :3, 3] = -np.einsum('...ij,...j->...i', R, T[..., :3, 3])
    out[..., 3, 3] = 1.0
    return out


def reprojection_cost(cam_from_tag, corners, K, tag_size):
    """Sum of squared pixel errors of (..., 4, 4) camera<-tag poses against (..., 4, 2) corners; points behind the
    camera cost a large constant."""
    X = _corners_obj(tag_size)
    # the corners lie in the tag's plane (z = 0): p = R[:, :2] (x, y) + t, as one small matrix product per pose
    p = cam_from_tag[..., :3, :2] @ X[:, :2].T + cam_from_tag[..., :3, 3:4]          # (..., 3, 4 corners)
    z = p[..., 2, :]
    ok = z > 1e-9
    zs = np.where(ok, z, 1.0)
    u = K[0, 0] * p[..., 0, :] / zs + K[0, 2]
    v = K[1, 1] * p[..., 1, :] / zs + K[1, 2]
    e = (u - corners[..., 0]) ** 2 + (v - corners[..., 1]) ** 2
    return np.where(ok, e, 1e12).sum(axis=-1)


def _area(c):
    x, y = c[:, 0], c[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def chain_initial_map(frames):
    """frames: per frame a list of (tag id, T camera<-tag 4x4, corners 4x2).  Returns (world id, {tag id: world<-tag},
    [world<-camera or None per frame])."""
    ids = [t for fr in frames for t, _, _ in fr]
    if not ids:
        return -1, {}, [None] * len(frames)
    world = min(ids)
    tags = {world: np.eye(4)}
    cams = [None] * len(frames)
    changed = True
    while changed:
        changed = False
        for f, fr in enumerate(frames):
            if cams[f] is None:
                placed = [(t, T, c) for t, T, c in fr if t in tags]
                if not placed:
                    continue
                t, T, _ = max(placed, key=lambda o: _area(np.asarray(o[2])))
                cams[f] = tags[t] @ _inv(np.asarray(T, dtype=np.float64))
                changed = True
            for t, T, _ in fr:
                if t not in tags:
                    tags[t] = cams[f] @ np.asarray(T, dtype=np.float64)
                    changed = True
    return world, tags, cams


def _padded_groups(key, n_groups):
    """indices of the observations of every group (camera or tag), padded with -1: (n_groups, widest group)"""
    key = np.asarray(key, dtype=np.int64)
    order = np.argsort(key, kind="stable")
    counts = np.bincount(key, minlength=n_groups)
    width = int(counts.max()) if len(key) else 0
    out = np.full((n_groups, max(width, 1)), -1, dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
    col = np.arange(len(key)) - np.repeat(starts, counts)
    out[key[order], col] = order
    return out


def _pick_best(current, cand_from_obs, cand_ok, rel_of, corners, obs_ok, K, tag_size):
    """current (G,4,4); cand_from_obs (G,C,4,4) with validity cand_ok (G,C); rel_of(cand (G,C+1,4,4)) -> camera<-tag poses
    (G,C+1,N,4,4) of every candidate against every observation of the group; corners (G,N,4,2), obs_ok (G,N).
    Returns the candidate (current first, then in observation order) with the smallest total reprojection error."""
    cand = np.concatenate([current[:, None], cand_from_obs], axis=1)
    cost = reprojection_cost(rel_of(cand), corners[:, None], K, tag_size)               # (G, C+1, N)
    cost = np.where(obs_ok[:, None, :], cost, 0.0).sum(axis=2)
    cost[:, 1:] = np.where(cand_ok, cost[:, 1:], np.inf)
    best = np.argmin(cost, axis=1)
    return cand[np.arange(len(cand)), best]


def reseed_poses(cam_T, tag_T, obs_cam, obs_tag, obs_T, obs_corners, K, tag_size, fixed_tag, sweeps=2, max_cand=None):
    """cam_T (P,4,4) world<-camera, tag_T (L,4,4) world<-tag, observations (camera index, tag index, PnP pose
    camera<-tag, corners 4x2).  Returns new (cam_T, tag_T).  The gauge is left free during the sweeps (a chain that
    started from a bad observation of the world tag is consistent everywhere except at that tag, and it is the world
    tag that has to give way); at the end the map is re-expressed so that tag `fixed_tag` sits at the identity again.
    max_cand: a camera / tag tries only the poses implied by its max_cand largest observations (by corner area: the most
    reliable single-view poses) instead of all of them; every candidate is still scored against ALL observations.
    All cameras (then all tags) of a sweep are evaluated at once: they depend only on the other kind."""
    cam = np.array(cam_T, dtype=np.float64)
    tag = np.array(tag_T, dtype=np.float64)
    oc = np.asarray(obs_cam, dtype=np.int64)
    ot = np.asarray(obs_tag, dtype=np.int64)
    oT = np.asarray(obs_T, dtype=np.float64).reshape(-1, 4, 4)
    oC = np.asarray(obs_corners, dtype=np.float64).reshape(-1, 4, 2)
    if len(oc) == 0:
        return cam, tag
    oTi = _inv(oT)
    x, y = oC[:, :, 0], oC[:, :, 1]
    area = 0.5 * np.abs((x * np.roll(y, -1, axis=1) - y * np.roll(x, -1, axis=1)).sum(axis=1))

    def groups(key, n):
        idx = _padded_groups(key, n)
        ok = idx >= 0
        if max_cand is not None and idx.shape[1] > max_cand:
            a = np.where(ok, area[np.maximum(idx, 0)], -1.0)
            sel = np.sort(np.argsort(-a, axis=1, kind="stable")[:, :max_cand], axis=1)     # the largest, in observation order
            cidx = np.take_along_axis(idx, sel, axis=1)
        else:
            cidx = idx
        return np.maximum(idx, 0), ok, np.maximum(cidx, 0), cidx >= 0

    ci, ci_ok, cc, cc_ok = groups(oc, len(cam))
    ti, ti_ok, tc, tc_ok = groups(ot, len(tag))
    for _ in range(sweeps):
        has = ci_ok.any(axis=1)
        new = _pick_best(cam, tag[ot[cc]] @ oTi[cc], cc_ok, lambda cand: _inv(cand)[:, :, None] @ tag[ot[ci]][:, None], oC[ci], ci_ok, K, tag_size)
        cam = np.where(has[:, None, None], new, cam)
        has = ti_ok.any(axis=1)
        new = _pick_best(tag, cam[oc[tc]] @ oT[tc], tc_ok, lambda cand: _inv(cam[oc[ti]])[:, None] @ cand[:, :, None], oC[ti], ti_ok, K, tag_size)
        tag = np.where(has[:, None, None], new, tag)
    M = _inv(tag[fixed_tag])
    cam, tag = M[None] @ cam, M[None] @ tag
    tag[fixed_tag] = np.eye(4)
    return cam, tag


def behind_camera(cam_T, tag_T, obs_cam, obs_tag, tag_size, margin=1e-6):
    """per observation: does any corner of the tag, as the map has it, lie on or behind the camera's image plane?"""
    X = _corners_obj(tag_size)
    rel = _inv(np.asarray(cam_T, dtype=np.float64))[np.asarray(obs_cam, dtype=np.int64)] @ np.asarray(tag_T, dtype=np.float64)[np.asarray(obs_tag, dtype=np.int64)]
    z = np.einsum('nj,kj->nk', rel[:, 2, :], X)
    return (z <= margin).any(axis=1)


def _exp_so3(w):
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / (th * th) * (Kx @ Kx)


def _tag_residuals(cam_inv, T, corners, K, tag_size):
    X = _corners_obj(tag_size)
    p = np.einsum('nij,kj->nki', (cam_inv @ T[None])[:, :3, :], X)
    z = np.where(p[..., 2] > 1e-9, p[..., 2], 1e-9)
    return np.stack([K[0, 0] * p[..., 0] / z + K[0, 2] - corners[..., 0], K[1, 1] * p[..., 1] / z + K[1, 2] - corners[..., 1]], -1).ravel()


def refine_tag(cam_inv, T, corners, K, tag_size, iters=6):
    """Gauss-Newton on one tag's pose (6 dof, world<-tag) with the cameras held fixed; cam_inv (N,4,4) camera<-world of
    its N observations, corners (N,4,2).  Returns (T, cost)."""
    T = np.array(T, dtype=np.float64)
    r = _tag_residuals(cam_inv, T, corners, K, tag_size)
    cost = float(r @ r)
    eps = 1e-6
    for _ in range(iters):
        J = np.empty((r.size, 6))
        for a in range(6):
            d = np.zeros(6); d[a] = eps
            Tp = T.copy()
            Tp[:3, :3] = T[:3, :3] @ _exp_so3(d[:3])      # perturb in the tag's own frame
            Tp[:3, 3] = T[:3, 3] + T[:3, :3] @ d[3:]
            J[:, a] = (_tag_residuals(cam_inv, Tp, corners, K, tag_size) - r) / eps
        H = J.T @ J
        step = np.linalg.solve(H + 1e-9 * np.trace(H) * np.eye(6), -J.T @ r)
        Tn = T.copy()
        Tn[:3, :3] = T[:3, :3] @ _exp_so3(step[:3])
        Tn[:3, 3] = T[:3, 3] + T[:3, :3] @ step[3:]
        rn = _tag_residuals(cam_inv, Tn, corners, K, tag_size)
        cn = float(rn @ rn)
        if not cn < cost:
            break
        T, r, cost = Tn, rn, cn
    return T, cost


def mirrored_pose(cam_from_tag):
    """The other minimum of planar PnP under weak perspective: the tag turned half a turn about the line of sight
    through its centre and half a turn about its own normal (corners land where they were, the normal is reflected
    about the line of sight)."""
    T = np.array(cam_from_tag, dtype=np.float64)
    s = T[:3, 3] / np.linalg.norm(T[:3, 3])
    Rs = 2.0 * np.outer(s, s) - np.eye(3)                 # half turn about s
    out = T.copy()
    out[:3, :3] = Rs @ T[:3, :3] @ np.diag([-1.0, -1.0, 1.0])
    return out


def flip_test_tags(cam_T, tag_T, obs_cam, obs_tag, obs_corners, K, tag_size, fixed_tag):
    """After a refinement: every tag tries its mirrored pose (built in the view where it appears largest, polished with
    refine_tag against all its views) and keeps it if the total reprojection error is lower.  Small, distant tags seen
    from a short baseline keep the wrong planar-PnP minimum in every single view, so no per-view candidate is right and
    Levenberg-Marquardt cannot leave the basin.  Returns (tag_T, ids flipped)."""
    cam_inv_all = _inv(np.asarray(cam_T, dtype=np.float64))
    tag = np.array(tag_T, dtype=np.float64)
    oc = np.asarray(obs_cam, dtype=np.int64)
    ot = np.asarray(obs_tag, dtype=np.int64)
    oC = np.asarray(obs_corners, dtype=np.float64).reshape(-1, 4, 2)
    flipped = []
    for j in range(len(tag)):
        idx = np.flatnonzero(ot == j)
        if j == fixed_tag or len(idx) == 0:
            continue
        ci = cam_inv_all[oc[idx]]
        T0, c0 = refine_tag(ci, tag[j], oC[idx], K, tag_size, iters=2)
        big = int(np.argmax([_area(c) for c in oC[idx]]))
        Tm = np.asarray(cam_T, dtype=np.float64)[oc[idx[big]]] @ mirrored_pose(ci[big] @ T0)
        T1, c1 = refine_tag(ci, Tm, oC[idx], K, tag_size, iters=8)
        if c1 < c0:
            tag[j] = T1
            flipped.append(j)
        else:
            tag[j] = T0
    return tag, flipped
