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


# ------------------------------------------------------------------------------------------------ ISM proposals
# The <= 200 SAM proposals of one image shard like the PEM proposals: proposal i -> rank i % world.  Every rank scores its shard -- the
# semantic / appearance scores, the visible ratio and the IoU of a proposal depend on that proposal alone (ISM/model/detector.py:260-322)
# -- and contributes one record per proposal
#   [valid, sem, appe, iou, vis, shard_all_positive, object id, x0, y0, x1, y1, global proposal id]        12 fp32
# (the class-token filter `score > confidence_thresh`, detector.py:289-296, drops proposals: `valid` marks the survivors, rows are padded
# to the shard size so that the collective has equal shapes).  ONE all-gather; then, on the merged list: the IoU quirk of
# ISM/utils/bbox_utils.py:214-220 (ANY pair with a non-positive overlap turns the whole geometric score into the scalar 0.0 -- the one
# image-wide term: every rank reports whether its shard was all-positive), the final score (detector.py:384) and the per-object NMS of
# Detections.apply_nms_per_object_id (ISM/model/utils.py:107-119).  The reference merges result files on rank 0 (detector.py:425-431);
# here every rank holds the gathered records, so any rank can merge.  Masks stay on their owner ranks: the survivors' global proposal
# ids say which ones to keep.
DET_FIELDS = 12


def pack_detections(sem, appe, iou, vis, all_positive, object_ids, boxes, global_ids, per):
    """This rank's surviving proposals -> (per, 12) fp32 records (padding rows: valid = 0).  sem / appe / iou / vis (k,) fp32,
    all_positive: bool (this shard had no non-positive overlap), object_ids (k,), boxes (k,4) xyxy, global_ids (k,); k <= per.
    Box coordinates and ids < 2^24 are exact in fp32."""
    k = int(sem.shape[0])
    if k > per:
        raise ValueError("pack_detections: %d records for a shard of %d" % (k, per))
    rec = torch.zeros((per, DET_FIELDS), dtype=torch.float32, device=sem.device)
    rec[:, 5] = 1.0 if all_positive else 0.0
    if k:
        rec[:k, 0] = 1.0
        rec[:k, 1] = sem.to(torch.float32)
        rec[:k, 2] = appe.to(torch.float32)
        rec[:k, 3] = iou.to(torch.float32) if torch.is_tensor(iou) else float(iou)
        rec[:k, 4] = vis.to(torch.float32)
        rec[:k, 6] = object_ids.to(torch.float32)
        rec[:k, 7:11] = boxes.to(torch.float32)
        rec[:k, 11] = global_ids.to(torch.float32)
    return rec.contiguous()


def gather_detections(rec, dist):
    """All-gather of the (per, 12) record blocks: (world * per, 12), rank-major, on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    out = torch.empty((dist.get_world_size() * rec.shape[0], DET_FIELDS), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec)
    return out


def merge_detections(gathered, final_fn, nms_fn, nms_thresh=0.25):
    """Gathered records -> the image's detections as the unsharded run would hold them after apply_nms_per_object_id.
    The valid records are put in global proposal order (an unsharded run keeps proposal order through its index selections);
    geo = iou, or None (the scalar-0.0 quirk) unless EVERY shard was all-positive; scores = final_fn(sem, appe, geo, vis)
    (sam6d_hip.ism.final_score on the GPU: detector.py:384); keep = nms_fn(boxes f32, scores, nms_thresh, object_ids)
    (sam6d_hip.ism.nms).  Returns a dict: scores, object_ids (int64), boxes (int64 xyxy), proposal_ids (int64) of the kept
    proposals, and all_scores / all_proposal_ids before NMS."""
    all_pos = bool((gathered[:, 5] > 0.5).all())
    g = gathered[gathered[:, 0] > 0.5]
    g = g[torch.argsort(g[:, 11], stable=True)]
    obj = g[:, 6].to(torch.int64).contiguous()
    boxes = g[:, 7:11].to(torch.int64).contiguous()
    pid = g[:, 11].to(torch.int64).contiguous()
    if g.shape[0] == 0:
        e = g[:, 1].contiguous()
        return dict(scores=e, object_ids=obj, boxes=boxes, proposal_ids=pid, all_scores=e, all_proposal_ids=pid)
    scores = final_fn(g[:, 1].contiguous(), g[:, 2].contiguous(), g[:, 3].contiguous() if all_pos else None, g[:, 4].contiguous())
    keep = nms_fn(boxes.to(torch.float32), scores, nms_thresh, obj).to(torch.int64)
    return dict(scores=scores[keep], object_ids=obj[keep], boxes=boxes[keep], proposal_ids=pid[keep], all_scores=scores,
                all_proposal_ids=pid)


def ism_sharded_detections(score_fn, n_proposals, final_fn, nms_fn, dist, nms_thresh=0.25, device=None):
    """The ISM scoring of one image over the ranks of `dist`.  score_fn(ids) -> dict(sel, sem, appe, iou, vis, all_positive,
    object_ids, boxes): given this rank's global proposal ids (int64, padding removed) it runs the per-proposal scoring path on them
    (semantic select, appearance, projection + IoU, visible ratio) and returns, for the positions `sel` (into ids) that pass the
    class-token filter, the four score terms, the object ids and the proposal boxes, plus the shard's all-positive IoU flag (iou may
    be the float 0.0 the reference returns for a shard with a non-positive overlap).  Returns merge_detections' dict (identical on
    every rank)."""
    on = dist is not None and dist.is_initialized()
    rank = dist.get_rank() if on else 0
    world = dist.get_world_size() if on else 1
    ids, n_valid = shard_indices(n_proposals, rank, world)
    per = ids.shape[0]
    ids = ids[:n_valid]
    if device is not None:
        ids = ids.to(device)
    r = score_fn(ids)
    gid = ids[r["sel"].to(torch.int64)] if n_valid else ids
    rec = pack_detections(r["sem"], r["appe"], r["iou"], r["vis"], r["all_positive"], r["object_ids"], r["boxes"], gid, per)
    return merge_detections(gather_detections(rec, dist), final_fn, nms_fn, nms_thresh)
