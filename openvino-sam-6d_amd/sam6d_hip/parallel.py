"""Multi-GPU sharding of the matching path: proposals are independent, so they are dealt round-robin to the
ranks (proposal b -> rank b % world), weights are replicated, and the only exchange is ONE all-gather of
(R 9 + t 3 + score 1) = 13 fp32 per proposal -- a latency-bound message far below the per-link xGMI bandwidth.
The reference has no collective at all (files / DataParallel only).

One process per GPU; `dist` is torch.distributed initialised by the caller (backend "nccl" = RCCL on ROCm, "gloo" in
the CPU tests).
"""
import torch


def shard_indices(n_total, rank, world):
    """Round-robin shard: global proposal ids handled by `rank`, padded by repeating the last id so that every rank holds
    the same count (the collective needs equal shapes).  Returns (ids, n_valid)."""
    ids = list(range(rank, n_total, world))
    per = (n_total + world - 1) // world
    n_valid = len(ids)
    while len(ids) < per:
        ids.append(ids[-1] if ids else 0)
    return torch.tensor(ids, dtype=torch.long), n_valid


def pack_poses(R, t, score):
    """(B,3,3), (B,3), (B,) -> (B,13) contiguous."""
    B = R.shape[0]
    return torch.cat([R.reshape(B, 9), t.reshape(B, 3), score.reshape(B, 1)], dim=1).contiguous()


def unpack_poses(p):
    return p[:, :9].reshape(-1, 3, 3), p[:, 9:12], p[:, 12]


def gather_poses(R, t, score, dist):
    """All-gather of every rank's (B,13) pose block; returns rank-major (world*B, ...) tensors on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return R, t, score
    mine = pack_poses(R, t, score)
    out = torch.empty((dist.get_world_size() * mine.shape[0], 13), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return unpack_poses(out)


def unshard(gathered, n_total, world):
    """Inverse of shard_indices on rank-major gathered rows: (world*per, k) -> (n_total, k) in global proposal order."""
    per = gathered.shape[0] // world
    out = torch.empty((n_total,) + tuple(gathered.shape[1:]), dtype=gathered.dtype, device=gathered.device)
    for r in range(world):
        ids = torch.arange(r, n_total, world, device=gathered.device)
        out[ids] = gathered[r * per:r * per + len(ids)]
    return out
