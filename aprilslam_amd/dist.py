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


def pack_observations(dets, poses, n_per_frame, stream_id, max_tags, frame_offset=0):
    """Structured detection/pose arrays of one step -> (n_frames, max_tags, OBS_WIDTH) float64, zero padded."""
    n_frames = len(n_per_frame)
    out = np.zeros((n_frames, max_tags, OBS_WIDTH), dtype=np.float64)
    start = 0
    for f in range(n_frames):
        n = int(n_per_frame[f])
        k = min(n, max_tags)
        if k:
            d = dets[start:start + k]
            out[f, :k, 0] = 1.0 if poses is None else poses["ok"][start:start + k]
            out[f, :k, 1] = stream_id
            out[f, :k, 2] = frame_offset + f
            out[f, :k, 3] = d["id"]
            out[f, :k, 4:12] = d["corners"].reshape(k, 8)
            if poses is not None:
                out[f, :k, 12:28] = poses["T"][start:start + k].reshape(k, 16)
        start += n
    return out


def all_gather_observations(local_obs, device=None):
    """All ranks contribute an equally shaped record block; returns (world, n_frames, max_tags, OBS_WIDTH)
    as a numpy array, identical on every rank.  Falls back to the local block when not distributed."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_obs[None]
    t = torch.from_numpy(np.ascontiguousarray(local_obs))
    if device is not None:
        t = t.to(device)
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    # one flat all-gather; the output rows are views of one contiguous block
    dist.all_gather([out[r] for r in range(world)], t)
    return out.cpu().numpy()


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
