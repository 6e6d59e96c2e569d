"""Multi-GPU sharding of the hot path: one process per GPU, frames (or whole streams) are independent until the
graph update, which needs every observation in frame order.

The only exchange is one flat all-gather per step of fixed-size padded observation records (`asl_obs`, 136 bytes per
tag slot: id, flags, 8 float32 corners, 12 float64 of T).  The records are packed ON THE DEVICE from the detector's
results (asl_pack_observations_device) and gathered with `all_gather_into_tensor` -- RCCL over xGMI with backend
"nccl", gloo on CPU tensors in the tests; the payload is a few MB per rank and step, so a single flat collective,
never a ring of buckets.

After the gather every rank holds the same block and updates its own copy of the graph:
  * `graph_frames` (device kernel k_graph_frames / numpy mirror `graph_frames_numpy`): per frame, whether its update is
    self-contained (the world tag is its lowest id and no PnP failed: branches A / C1 of slam_graph.py:33-49), its
    camera pose, and per tag the last such frame in (frame, stream) order;
  * `apply_block` turns that into the state the sequential reference update would have reached: node values from the
    last frame that saw each tag (computed on the host with the reference's own float64 operations), and falls back
    to the sequential mirror, frame by frame, whenever a frame is not self-contained.
"""
import numpy as np

from . import _lib
from .slam_graph import Node

OBS_DTYPE = _lib.OBS_DTYPE
MAX_IDS = 4096  # size of the per-tag "last seen" table


def shard_frames(n_frames, rank, world_size):
    """Frame-round-robin shard (C5: one 4K stream over 8 GPUs): frame i -> rank i % world_size."""
    return list(range(rank, n_frames, world_size))


def pack_observations(dets, poses, n_per_frame, max_tags):
    """Host-side packing of one step's structured detection/pose arrays into (n_frames, max_tags) asl_obs records --
    the layout asl_pack_observations_device produces on the GPU (used by the CPU tests and host-only callers).
    Detections beyond max_tags per frame are dropped; `poses` is required (an observation without a transform cannot
    update the graph)."""
    if poses is None:
        raise ValueError("pack_observations needs the poses: a record without a transform cannot update the graph")
    npf = np.asarray(n_per_frame, dtype=np.int64)
    n_frames = len(npf)
    out = np.zeros((n_frames, max_tags), dtype=OBS_DTYPE)
    out["id"] = -1
    total = int(npf.sum())
    if total == 0:
        return out
    starts = np.concatenate(([0], np.cumsum(npf)[:-1]))
    frame = np.repeat(np.arange(n_frames), npf)
    slot = np.arange(total) - np.repeat(starts, npf)
    keep = slot < max_tags
    f, s = frame[keep], slot[keep]
    out["id"][f, s] = dets["id"][:total][keep]
    out["flags"][f, s] = 1 | (2 * (poses["ok"][:total][keep] != 0))
    out["corners"][f, s] = dets["corners"][:total][keep].reshape(-1, 8).astype(np.float32)
    out["T"][f, s] = poses["T"][:total][keep].reshape(-1, 16)[:, :12]
    return out


class _Roctx:
    """ROCTX ranges for `rocprofv3 --marker-trace` (SURVEY.md 8d asks for one around the all-gather): ASL_ROCTX=1 loads the
    marker library the way libaprilslam.so does for its stage groups; without it push / pop do nothing."""

    def __init__(self):
        self._push = self._pop = None
        import os
        if os.environ.get("ASL_ROCTX", "0") in ("", "0"):
            return
        import ctypes
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                lib = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                self._push, self._pop = lib.roctxRangePushA, lib.roctxRangePop
                return
            except (OSError, AttributeError):
                continue

    def push(self, name):
        if self._push:
            self._push(name.encode())

    def pop(self):
        if self._pop:
            self._pop()


_roctx = None


def roctx():
    global _roctx
    if _roctx is None:
        _roctx = _Roctx()
    return _roctx


def all_gather_observations(local_obs):
    """local_obs: (n_frames, max_tags) asl_obs records as a numpy structured array or as a torch uint8 tensor
    (n_frames, max_tags, 136) on the device.  Returns the same kind with a leading `world` axis, identical on every
    rank: ONE all_gather_into_tensor (also for a group of one rank: the same call path).  Without an initialised process
    group the block is returned with world = 1."""
    import torch
    import torch.distributed as dist

    is_np = isinstance(local_obs, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local_obs).view(np.uint8).reshape(local_obs.shape + (OBS_DTYPE.itemsize,))) if is_np else local_obs
    if not (dist.is_available() and dist.is_initialized()):
        out = t[None]
    else:
        world = dist.get_world_size()
        flat = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
        roctx().push("all-gather of observation records")
        dist.all_gather_into_tensor(flat, t.contiguous().view(-1))  # flat in, flat out: the form every backend accepts
        roctx().pop()
        out = flat.view((world,) + tuple(t.shape))
    if is_np:
        return out.numpy().view(OBS_DTYPE).reshape(out.shape[:-1])
    return out


class ObsBlock:
    """A gathered block obs[world][n_frames][max_tags] that lives either on the host (numpy asl_obs records) or on the
    device (torch uint8 tensor (world, n_frames, max_tags, 136)).  The bulk update touches only a handful of frames of
    it (the last sighting of every tag), so a device block is never copied to the host as a whole unless the sequential
    fall-back needs it."""

    def __init__(self, data):
        self.data = data
        self.on_host = isinstance(data, np.ndarray)
        self.shape = tuple(data.shape[:3])
        self._host = data if self.on_host else None

    def frames(self, pairs):
        """records (len(pairs), max_tags) of the (stream, frame) pairs, one copy"""
        if self._host is not None:
            return np.stack([self._host[s, f] for s, f in pairs]) if len(pairs) else np.zeros((0, self.shape[2]), OBS_DTYPE)
        import torch
        world, n_frames, max_tags = self.shape
        idx = torch.tensor([s * n_frames + f for s, f in pairs], dtype=torch.int64, device=self.data.device)
        rows = self.data.view(world * n_frames, max_tags, -1).index_select(0, idx).cpu().numpy()
        return rows.reshape(-1).view(OBS_DTYPE).reshape(len(pairs), max_tags)

    def host(self):
        if self._host is None:
            a = self.data.cpu().numpy()
            self._host = a.reshape(-1).view(OBS_DTYPE).reshape(self.shape)
        return self._host


def _full(T12):
    T = np.zeros(T12.shape[:-1] + (4, 4))
    T[..., :3, :] = T12.reshape(T12.shape[:-1] + (3, 4))
    T[..., 3, 3] = 1.0
    return T


def graph_frames_numpy(obs, coordinate_id, n_ids=MAX_IDS):
    """Host mirror of asl_graph_frames_device on obs[world][n_frames][max_tags]: (pose (world, n_frames, 4, 4), status
    (world, n_frames) uint8, last (n_ids,) uint32), same definitions (include/aprilslam.h)."""
    world, n_frames, max_tags = obs.shape
    used = (obs["flags"] & 1) != 0
    n = used.sum(axis=2)
    ok = (((obs["flags"] & 2) != 0) & (obs["id"] >= 0) & (obs["id"] < n_ids)) | ~used
    steady = (coordinate_id >= 0) & ok.all(axis=2) & (n > 0) & (obs["id"][:, :, 0] == coordinate_id)
    status = np.where(n == 0, 2, np.where(steady, 0, 1)).astype(np.uint8)
    pose = np.zeros((world, n_frames, 4, 4))
    last = np.zeros(n_ids, dtype=np.uint32)
    if steady.any():
        T = _full(obs["T"])                                   # (world, n_frames, max_tags, 4, 4)
        with np.errstate(all="ignore"):
            Ts = np.where(used[..., None, None], T, np.eye(4))
            local = np.linalg.inv(Ts[steady])                 # (S, max_tags, 4, 4)
        lc = local[:, :1]
        vote = (lc @ Ts[steady]) @ local                      # world_j @ local_j; slot 0 gives I @ inv(T_c)
        vote[:, 0] = local[:, 0]
        u = used[steady]
        acc = np.zeros((vote.shape[0], 4, 4))
        for j in range(max_tags):                             # the reference's accumulation order: slot by slot
            acc += np.where(u[:, j, None, None], vote[:, j], 0.0)
        pose[steady] = acc / n[steady][:, None, None]
        s_idx, f_idx = np.nonzero(steady)
        order = (f_idx * world + s_idx).astype(np.uint64) * max_tags
        ids = obs["id"][steady]
        for j in range(max_tags):
            sel = u[:, j]
            np.maximum.at(last, ids[sel, j], (order[sel] + j + 1).astype(np.uint32))
    return pose, status, last


def last_sightings_numpy(obs, status, lo, hi, n_ids=MAX_IDS):
    """Host mirror of asl_graph_picks_device: (last (n_ids,) uint32, picks (2 * n_ids,) asl_obs) over the status-0 frames
    at positions frame * world + stream in [lo, hi)."""
    world, n_frames, max_tags = obs.shape
    last = np.zeros(n_ids, dtype=np.uint32)
    picks = np.zeros(2 * n_ids, dtype=OBS_DTYPE)
    pos = np.arange(n_frames)[None, :] * world + np.arange(world)[:, None]
    sel = (np.asarray(status).reshape(world, n_frames) == 0) & (pos >= lo) & (pos < hi)
    s_idx, f_idx = np.nonzero(sel)
    if len(s_idx):
        rec = obs[s_idx, f_idx]
        used = (rec["flags"] & 1) != 0
        order = (f_idx * world + s_idx).astype(np.uint64) * max_tags
        for j in range(max_tags):
            u = used[:, j] & (rec["id"][:, j] >= 0) & (rec["id"][:, j] < n_ids)
            np.maximum.at(last, rec["id"][u, j], (order[u] + j + 1).astype(np.uint32))
        for t in np.nonzero(last)[0]:
            key = int(last[t]) - 1
            o, slot = key // max_tags, key % max_tags
            picks[2 * t] = obs[o % world, o // world, slot]
            picks[2 * t + 1] = obs[o % world, o // world, 0]
    return last, picks


MAX_SEQUENTIAL_FRAMES = 64  # more frames of a block than this needing the sequential update: the whole block takes it


def _sequential(slam, obs, frames):
    """The reference's own update, frame by frame, for `frames` = iterable of (stream, frame) in order."""
    poses = []
    for s, f in frames:
        rec = obs[s, f]
        rec = rec[(rec["flags"] & 1) != 0]
        if len(rec) == 0:
            slam.visible_tags = []
            poses.append(None)
            continue
        # every detected id is "visible" (slam.py:24); only tags whose PnP succeeded update the graph (slam.py:30-31)
        poses.append(slam.process_observations(rec["id"].tolist(), _full(rec["T"]), oks=(rec["flags"] & 2) != 0))
    return poses


def apply_block(slam, obs, frames_result=None, picks=None, tail=None, picker=None):
    """Apply one gathered block (numpy asl_obs records [world][n_frames][max_tags], or an ObsBlock) to `slam` in
    (frame, stream) order, deterministically and identically on every rank.  frames_result = (pose, status, last) of
    graph_frames (device kernel or numpy mirror) for this block and slam.coordinate_id; computed here with numpy if
    omitted.  picks (2 * n_ids asl_obs, from asl_graph_frames_device) and tail (the records of the last frame of every
    stream, (world, max_tags)) spare the bulk path every access to the block itself.
    picker(lo, hi) -> (last, picks) over the status-0 frames at positions frame * world + stream in [lo, hi)
    (asl_graph_picks_device; default: the numpy mirror on the host copy of the block) serves the case where a few frames
    need the sequential update: the self-contained stretches between them are still applied in bulk.
    Returns (poses (world, n_frames, 4, 4), NaN where my_pose() is None; frames that took the sequential path)."""
    blk = obs if isinstance(obs, ObsBlock) else ObsBlock(obs)
    world, n_frames, max_tags = blk.shape
    total = world * n_frames
    out = np.full((world, n_frames, 4, 4), np.nan)
    pos_all = np.arange(n_frames)[None, :] * world + np.arange(world)[:, None]
    lo, nseq = 0, 0
    given = frames_result is not None and slam.coordinate_id != -1

    def one_frame(p):
        s, f = p % world, p // world
        pose_p = _sequential(slam, blk.frames([(s, f)])[0][None, None], [(0, 0)])[0]
        if pose_p is not None:
            out[s, f] = pose_p

    while lo < total:
        c0 = slam.coordinate_id
        if c0 == -1:
            # nothing is self-contained before a world tag exists: frame by frame until one does (once per run)
            one_frame(lo)
            lo, nseq = lo + 1, nseq + 1
            continue
        if given and lo == 0:
            pose, status, last = frames_result
            seg_picker, whole_picks = picker, picks
        else:  # statuses for the world tag the graph has now (the block started without one, or it changed on the way)
            pose, status, last = graph_frames_numpy(blk.host(), c0)
            seg_picker, whole_picks = None, None
        pose = np.asarray(pose).reshape(world, n_frames, 4, 4)
        status = np.asarray(status).reshape(world, n_frames)
        todo = pos_all >= lo
        if lo == 0 and not (status == 1).any():
            _apply_steady(slam, blk, status, np.asarray(last), world, n_frames, max_tags, whole_picks, tail if whole_picks is not None else None)
            good = status == 0
            out[good] = pose[good]
            return out, nseq
        # Some frames need the reference's sequential update (the world tag is not in view, or a PnP failed): what they
        # read depends on everything before them.  If they are few, the self-contained stretches between them are applied
        # in bulk (last sightings of the stretch) and only those frames go through the mirror, in order.
        s_seq, f_seq = np.nonzero((status == 1) & todo)
        pos_seq = np.sort(f_seq * world + s_seq)
        if len(pos_seq) > MAX_SEQUENTIAL_FRAMES:
            od = [(q % world, q // world) for q in range(lo, total)]
            for (s, f), pq in zip(od, _sequential(slam, blk.host(), od)):
                if pq is not None:
                    out[s, f] = pq
            return out, nseq + len(od)
        if seg_picker is None:
            host = blk.host()
            seg_picker = lambda lo_, hi_, st_=status: last_sightings_numpy(host, st_, lo_, hi_)  # noqa: E731
        switched = False
        start = lo
        for p in list(pos_seq) + [total]:
            p = int(p)
            if p > lo:
                seg_status = np.where((pos_all >= lo) & (pos_all < p), status, 2).astype(np.uint8)
                if (seg_status == 0).any():
                    last_seg, picks_seg = seg_picker(lo, p)
                    _apply_steady(slam, blk, seg_status, np.asarray(last_seg), world, n_frames, max_tags, picks_seg, None, end_pos=p)
                elif status[(p - 1) % world, (p - 1) // world] == 2:
                    slam.visible_tags = []
            lo = p
            if p < total:
                one_frame(p)
                nseq += 1
                lo = p + 1
                if slam.coordinate_id != c0:  # the world tag changed: the statuses computed for the old one no longer hold
                    switched = True
                    break
        good = (status == 0) & (pos_all >= start) & (pos_all < lo)
        out[good] = pose[good]
        if not switched:
            break
    return out, nseq


def _apply_steady(slam, blk, status, last, world, n_frames, max_tags, picks=None, tail=None, end_pos=None):
    """State after a block of self-contained frames: every tag carries the values of the last frame that saw it, computed
    with the reference's operations (inv, @) on the host -- bit-identical to the sequential update."""
    c = slam.coordinate_id
    graph = slam.graph
    ids = np.nonzero(last)[0]
    keys = last[ids].astype(np.int64) - 1
    slots = keys % max_tags
    orders = keys // max_tags
    # the last frame with detections leaves its ids visible and its pose in estimated_pose (slam.py:36-63)
    nz = np.argwhere(status.T != 2)  # (frame, stream) pairs, frame-major
    tail_pair = (int(nz[-1][1]), int(nz[-1][0])) if len(nz) else None
    if picks is None:
        pairs = [(int(o % world), int(o // world)) for o in orders]
        recs = blk.frames(pairs)
        own = [recs[k][slots[k]] for k in range(len(ids))]
        ref = [recs[k][0] for k in range(len(ids))]
    else:
        own = [picks[2 * int(t)] for t in ids]
        ref = [picks[2 * int(t) + 1] for t in ids]
    for k, tag_id in enumerate(ids):
        T = _full(own[k]["T"])
        if tag_id == c:
            graph.graph[int(tag_id)] = Node(graph.invert(T), np.eye(4), int(tag_id))
        else:
            graph.graph[int(tag_id)] = Node(graph.invert(T), graph.invert(_full(ref[k]["T"])) @ T, c)
    if tail_pair is not None:
        if tail is not None and tail_pair[1] == n_frames - 1:
            rec = tail[tail_pair[0]]
        else:
            rec = blk.frames([tail_pair])[0]
        rec = rec[(rec["flags"] & 1) != 0]
        slam.visible_tags = rec["id"].tolist()
        graph.visible_tags = slam.visible_tags
        slam.my_pose()
    # a stretch that ends with a frame without detections leaves nothing visible (slam.py:21-25)
    end_pos = world * n_frames if end_pos is None else end_pos
    if end_pos > 0 and status[(end_pos - 1) % world, (end_pos - 1) // world] == 2:
        slam.visible_tags = []
