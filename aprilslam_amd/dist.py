"""Multi-GPU sharding of the hot path: one process per GPU, frames (or whole streams) are
independent until the graph update, which needs every observation in frame order
(SURVEY.md section 8e).  The only exchange is one all-gather of fixed-size padded observation
records per step -- a few KB per rank, latency-bound, so a single flat all-gather (RCCL's direct
algorithm over the fully connected xGMI mesh), never a ring of large buckets.

Record layout (float64, OBS_WIDTH values per tag slot):
    [valid, stream, frame, id, corners(8), T(16)]
`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU in the tests.
"""
import numpy as np

OBS_WIDTH = 4 + 8 + 16


def shard_frames(n_frames, rank, world_size):
    """Frame-round-robin shard (C5: one 4K stream over 8 GPUs): frame i -> rank i % world_size."""
    return list(range(rank, n_frames, world_size))


def pack_observations(dets, poses, n_per_frame, stream_id, max_tags, frame_offset=0, out=None):
    """Structured detection/pose arrays of one step -> (n_frames, max_tags, OBS_WIDTH) float64, zero padded.
    Vectorised (at 1024 frames x 20 tags a per-frame Python loop costs more than the GPU step it follows); pass the
    previous result as `out` to reuse its memory (a fresh 5 MB array costs milliseconds of page faults)."""
    npf = np.asarray(n_per_frame, dtype=np.int64)
    n_frames = len(npf)
    if out is not None and out.shape == (n_frames, max_tags, OBS_WIDTH) and out.dtype == np.float64:
        out.fill(0.0)
    else:
        out = np.zeros((n_frames, max_tags, OBS_WIDTH), dtype=np.float64)
    total = int(npf.sum())
    if total == 0:
        return out
    c = int(npf[0])
    if c <= max_tags and np.all(npf == c):                # every frame holds the same number of tags: plain slices
        out[:, :c, 0] = 1.0 if poses is None else poses["ok"][:total].reshape(n_frames, c)
        out[:, :c, 1] = stream_id
        out[:, :c, 2] = (frame_offset + np.arange(n_frames))[:, None]
        out[:, :c, 3] = dets["id"][:total].reshape(n_frames, c)
        out[:, :c, 4:12] = dets["corners"][:total].reshape(n_frames, c, 8)
        if poses is not None:
            out[:, :c, 12:28] = poses["T"][:total].reshape(n_frames, c, 16)
        return out
    starts = np.concatenate(([0], np.cumsum(npf)[:-1]))
    frame = np.repeat(np.arange(n_frames), npf)           # frame of every detection (detections are frame-ordered)
    slot = np.arange(total) - np.repeat(starts, npf)      # its position inside the frame
    keep = slot < max_tags
    sel = slice(None) if keep.all() else keep             # field-wise selection: no copy of whole records
    flat = (frame * max_tags + slot)[sel]
    o = out.reshape(n_frames * max_tags, OBS_WIDTH)
    o[flat, 0] = 1.0 if poses is None else poses["ok"][:total][sel]
    o[flat, 1] = stream_id
    o[flat, 2] = frame_offset + frame[sel]
    o[flat, 3] = dets["id"][:total][sel]
    o[flat, 4:12] = dets["corners"][:total][sel].reshape(-1, 8)
    if poses is not None:
        o[flat, 12:28] = poses["T"][:total][sel].reshape(-1, 16)
    return out


_gather_cache = {}


def pinned_observation_buffer(n_frames, max_tags):
    """A page-locked (n_frames, max_tags, OBS_WIDTH) float64 numpy array to pack into (pack_observations(out=...)):
    the H2D copy of the all-gather then runs at full PCIe speed and without a staging copy."""
    import torch

    t = torch.empty((n_frames, max_tags, OBS_WIDTH), dtype=torch.float64)
    if torch.cuda.is_available():
        t = t.pin_memory()
    _gather_cache[("in", t.data_ptr())] = t  # keep the storage alive as long as the module
    return t.numpy()


def all_gather_observations(local_obs, device=None, wait=True):
    """All ranks contribute an equally shaped record block; returns (world, n_frames, max_tags, OBS_WIDTH)
    as a numpy array, identical on every rank.  Falls back to the local block when not distributed.
    With a CUDA `device` the gathered block comes back through cached page-locked buffers (two, used alternately):
    the returned array is a view of one, valid until the call after next with the same shape.  wait=False returns
    (array, event) right after enqueueing the read-back; the array may be read once event.synchronize() returned."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_obs[None]
    t = torch.from_numpy(np.ascontiguousarray(local_obs))
    world = dist.get_world_size()
    if device is None:
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype)
        dist.all_gather([out[r] for r in range(world)], t)
        return out.numpy()
    key = (str(device), world) + tuple(t.shape)
    bufs = _gather_cache.get(key)
    if bufs is None:
        bufs = {"d_in": torch.empty(tuple(t.shape), dtype=t.dtype, device=device),
                "d_out": torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=device),
                "h_out": [torch.empty((world,) + tuple(t.shape), dtype=t.dtype).pin_memory() for _ in range(2)],
                "turn": 0}
        _gather_cache[key] = bufs
    d_in, d_out = bufs["d_in"], bufs["d_out"]
    h_out = bufs["h_out"][bufs["turn"]]
    bufs["turn"] ^= 1
    d_in.copy_(t, non_blocking=True)
    # one flat all-gather; the output rows are views of one contiguous block
    dist.all_gather([d_out[r] for r in range(world)], d_in)
    h_out.copy_(d_out, non_blocking=True)
    if not wait:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        return h_out.numpy(), ev
    torch.cuda.current_stream(device).synchronize()
    return h_out.numpy()


def apply_observations(slam, gathered):
    """Deterministic global update: observations applied in (frame, stream, id) order to one SLAM
    graph (identical on every rank).  Returns the list of my_pose() results per (frame, stream)."""
    world, n_frames, max_tags, _ = gathered.shape
    poses = []
    for f in range(n_frames):
        for s in range(world):
            rec = gathered[s, f]
            rec = rec[rec[:, 0] > 0]
            if len(rec) == 0:
                slam.visible_tags = []
                poses.append(None)
                continue
            order = np.argsort(rec[:, 3], kind="stable")
            rec = rec[order]
            poses.append(slam.process_observations(rec[:, 3].astype(int).tolist(), rec[:, 12:28].reshape(-1, 4, 4)))
    return poses
