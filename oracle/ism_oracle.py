"""TEST INFRASTRUCTURE ONLY -- CPU oracle of SAM-6D's ISM template-scoring hot path.

Functional torch-CPU restatement; each function cites the reference file:line it follows
(ISM = SAM-6D/Instance_Segmentation_Model).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.  Pinned by tests/test_oracle_golden.py against vectors captured from the reference's
own model/loss.py, model/detector.py, utils/bbox_utils.py and utils/trimesh_utils.py (oracle/gen_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


# SAM-6D/Data/Example/camera.json (cam_K, depth_scale): the demo camera, used by the ISM fixtures
EXAMPLE_CAM_K = [572.4114, 0.0, 325.2611, 0.0, 573.57043, 242.04899, 0.0, 0.0, 1.0]


def caller_intrinsics(cam_K=EXAMPLE_CAM_K, depth_scale=1.0):
    """K (3,3) and depth_scale (1,) with the dtypes the reference's caller hands to the model: np.array of the JSON
    numbers -> float64 tensors (ISM/run_inference_custom.py:86-96, batch_input_data)."""
    return torch.from_numpy(np.array(cam_K).reshape((3, 3))), torch.from_numpy(np.array(depth_scale)).unsqueeze(0)


def pairwise_similarity(query, reference):
    """ISM/model/loss.py:27-44: cosine of every query (Nq,D) with every template (No,Nt,D) -> (Nq,No,Nt) in [0,1]."""
    Nq = query.shape[0]
    No, Nt = reference.shape[0], reference.shape[1]
    refs = F.normalize(reference.clone().unsqueeze(0).repeat(Nq, 1, 1, 1), dim=-1)
    qs = F.normalize(query.clone().unsqueeze(1).repeat(1, Nt, 1), dim=-1)
    sims = [F.cosine_similarity(qs, refs[:, o], dim=-1) for o in range(No)]
    return torch.stack(sims).permute(1, 0, 2).clamp(min=0.0, max=1.0)


def semantic_score(query, ref_desc, aggregation="avg_5", confidence_thresh=0.2):
    """ISM/model/detector.py:260-296 (+ best_template_pose :198-207).
    Returns idx_selected (K,), pred_idx_objects (K,), semantic_score (K,), best_template (K,)."""
    scores = pairwise_similarity(query, ref_desc)
    if aggregation == "mean":
        per_obj = scores.sum(-1) / scores.shape[-1]
    elif aggregation == "median":
        per_obj = torch.median(scores, dim=-1)[0]
    elif aggregation == "max":
        per_obj = torch.max(scores, dim=-1)[0]
    elif aggregation == "avg_5":
        per_obj = torch.topk(scores, k=5, dim=-1)[0].mean(-1)
    else:
        raise NotImplementedError
    score, obj = torch.max(per_obj, dim=-1)
    sel = torch.arange(len(score))[score > confidence_thresh]
    pred_obj = obj[sel]
    best_all = torch.max(scores[sel], dim=-1)[1]  # (K, No)
    best = torch.gather(best_all, 1, pred_obj[:, None].repeat(1, best_all.shape[1]))[:, 0]
    return sel, pred_obj, score[sel], best


def appearance_score(best_pose, pred_obj, q_appe, ref_appe):
    """ISM/model/detector.py:298-308 -> ISM/model/loss.py:52-62 (compute_straight)."""
    ref = ref_appe[pred_obj, best_pose]  # (K, P, D)
    sim = torch.matmul(q_appe, ref.permute(0, 2, 1))
    mx = torch.max(sim, dim=-1).values
    factor = torch.count_nonzero(q_appe.sum(dim=-1), dim=-1) + 1e-6
    return (mx.sum(-1) / factor).clamp(min=0.0, max=1.0), ref


def visible_ratio(q_appe, ref, thred=0.5):
    """ISM/model/loss.py:64-76 (compute_visible_ratio)."""
    sim = torch.matmul(q_appe, ref.permute(0, 2, 1)).max(1)[0]
    valid = torch.count_nonzero(sim, dim=(1,)) + 1e-6
    flt = sim * (sim > thred)
    return torch.count_nonzero(flt, dim=(1,)) / valid


def query_translation(masks, depth, K, depth_scale):
    """ISM/model/detector.py:234-246 + ISM/utils/trimesh_utils.py:77-105: mean back-projected masked depth."""
    md = masks * depth[None].repeat(masks.shape[0], 1, 1)
    H, W = md.shape[1], md.shape[2]
    u, v = torch.meshgrid(torch.arange(0, W), torch.arange(0, H), indexing="xy")
    Z = md * depth_scale / 1000
    X = (u - K[0, 2]) * Z / K[0, 0]
    Y = (v - K[1, 2]) * Z / K[1, 1]
    valid = Z > 0
    X, Y, Z = X * valid, Y * valid, Z * valid
    n = torch.count_nonzero(valid, dim=(1, 2)) + 1e-8
    t = torch.vstack((X.sum((1, 2)) / n, Y.sum((1, 2)) / n, Z.sum((1, 2)) / n)).permute(1, 0)
    return t.to(torch.float32)


def project_template_to_image(best_pose, pred_obj, poses, pointcloud, masks, depth, K, depth_scale):
    """ISM/model/detector.py:209-232 -> image_vu (K, Npc, 2) int32 (x then y, clamped to the image)."""
    R = poses[best_pose, 0:3, 0:3]
    pc = pointcloud[pred_obj]
    Nq, Np, _ = pc.shape
    posed = torch.matmul(R, pc.permute(0, 2, 1)).permute(0, 2, 1)
    posed = posed + query_translation(masks, depth, K, depth_scale)[:, None, :].repeat(1, Np, 1)
    Kq = K[None].repeat(Nq, 1, 1).to(torch.float32)
    homo = torch.bmm(Kq, posed.permute(0, 2, 1)).permute(0, 2, 1)
    vu = (homo / homo[:, :, -1][:, :, None])[:, :, 0:2].to(torch.int)
    H, W = depth.shape
    vu[:, :, 0].clamp_(min=0, max=W - 1)
    vu[:, :, 1].clamp_(min=0, max=H - 1)
    return vu


def compute_iou(bb_a, bb_b):
    """ISM/utils/bbox_utils.py:197-222 -- including the quirk: any non-positive overlap => scalar 0.0."""
    tl = torch.max(bb_a[:, 0:2], bb_b[:, 0:2])
    br = torch.min(bb_a[:, 2:4], bb_b[:, 2:4])
    wh_a = bb_a[:, 2:4] - bb_a[:, 0:2]
    wh_b = bb_b[:, 2:4] - bb_b[:, 0:2]
    wh = br - tl
    if (wh > 0).all():
        inter = wh[:, 0] * wh[:, 1]
        return inter / (wh_a[:, 0] * wh_a[:, 1] + wh_b[:, 0] * wh_b[:, 1] - inter)
    return 0.0


def geometric_score(image_uv, boxes, q_appe, ref, thred=0.5):
    """ISM/model/detector.py:310-322."""
    vis = visible_ratio(q_appe, ref, thred)
    xyxy = torch.concatenate((torch.min(image_uv, dim=1).values, torch.max(image_uv, dim=1).values), dim=-1)
    return compute_iou(xyxy, boxes), vis


def final_score(sem, appe, geo, vis):
    """ISM/model/detector.py:384 / ISM/run_inference_custom.py:255."""
    return (sem + appe + geo * vis) / (1 + 1 + vis)


def masked_patch_features(patch_features, masks, patch_size=14, validpatch_thresh=0.5):
    """ISM/model/dinov2.py:265-269 / 322-324 (patch_kernel = nn.AvgPool2d(patch_size), :139-141)."""
    keep = torch.nn.AvgPool2d(kernel_size=patch_size, stride=patch_size)(masks).flatten(-2) > validpatch_thresh
    keep = keep.unsqueeze(-1).repeat(1, 1, patch_features.shape[-1])
    return F.normalize(patch_features * keep, dim=-1)


# --------------------------------------------------------------- Detections bookkeeping (ISM/model/utils.py)
def small_detection_keep(boxes, masks, min_box_size, min_mask_size):
    """ISM/model/utils.py:96-102; torchvision's box_area is (x2 - x1) * (y2 - y1)."""
    img_area = masks.shape[1] * masks.shape[2]
    box_areas = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])) / img_area
    mask_areas = masks.sum(dim=(1, 2)) / img_area
    return torch.logical_and(box_areas > min_box_size ** 2, mask_areas > min_mask_size)


def nms(boxes, scores, thresh):
    """torchvision.ops.nms, restated from torchvision's published CPU kernel (torchvision 0.15.1 / 0.20.x pinned by the
    reference's environment files; the package is NOT installed here -> PARITY UNPINNED for this function): stable
    descending sort by score; walk the order, keep a box unless suppressed, suppress every later box whose
    IoU = inter / (area_i + area_j - inter) exceeds thresh.  Returns the kept indices in descending-score order."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy().astype(np.float32)
    n = b.shape[0]
    order = np.argsort(-s, kind="stable")
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(n, bool)
    keep = []
    for a in range(n):
        i = order[a]
        if dead[i]:
            continue
        keep.append(int(i))
        rest = order[a + 1:]
        w = np.maximum(np.float32(0), np.minimum(b[i, 2], b[rest, 2]) - np.maximum(b[i, 0], b[rest, 0]))
        h = np.maximum(np.float32(0), np.minimum(b[i, 3], b[rest, 3]) - np.maximum(b[i, 1], b[rest, 1]))
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        dead[rest[ovr > np.float32(thresh)]] = True
    return torch.tensor(keep, dtype=torch.int64)


def nms_per_object_id(boxes, scores, object_ids, thresh):
    """ISM/model/utils.py:107-117: NMS inside each object id, ids in torch.unique (ascending) order."""
    all_idx = torch.arange(len(object_ids))
    out = []
    for oid in torch.unique(object_ids):
        sel = object_ids == oid
        out.append(all_idx[sel][nms(boxes[sel].float(), scores[sel].float(), thresh)])
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.int64)


# --------------------------------------------------------------- detection_ism.json (ISM/model/utils.py)
LMO_OBJECT_IDS = np.array([1, 5, 6, 8, 9, 10, 11, 12])  # ISM/model/utils.py:8-22: LM-O's object ids (BOP numbering)


def force_binary_mask(mask, threshold=0.0):
    """ISM/utils/bbox_utils.py:190-192."""
    return np.where(np.asarray(mask) > threshold, 1, 0)


def mask_to_rle_loop(binary_mask):
    """ISM/model/utils.py:25-43, statement by statement (small masks only)."""
    counts = []
    last, run = 0, 0
    for elem in np.asarray(binary_mask).ravel(order="F"):
        if elem != last:
            counts.append(run)
            run = 0
            last = elem
        run += 1
    counts.append(run)
    return {"counts": counts, "size": list(np.asarray(binary_mask).shape)}


def mask_to_rle(binary_mask):
    """Same result as mask_to_rle_loop, vectorised: boundaries of the column-major sequence (value before the start = 0)."""
    m = np.asarray(binary_mask)
    flat = (m.ravel(order="F") != 0).astype(np.int8)
    prev = np.concatenate(([0], flat[:-1]))
    pos = np.flatnonzero(flat != prev)
    edges = np.concatenate(([0], pos, [flat.size]))
    return {"counts": np.diff(edges).tolist(), "size": list(m.shape)}


def rle_to_mask(rle):
    """ISM/segment_anything/utils/amg.py:138-150 (what pycocotools' decode returns for an uncompressed RLE)."""
    h, w = rle["size"]
    counts = np.asarray(rle["counts"], dtype=np.int64)
    vals = (np.arange(len(counts)) & 1).astype(bool)
    flat = np.repeat(vals, counts)
    out = np.zeros(h * w, dtype=bool)
    out[: min(flat.size, h * w)] = flat[: h * w]
    return out.reshape(w, h).transpose()


def rle_counts_to_string(counts):
    """pycocotools' COMPRESSED counts string (`rleToString`, cocoapi common/maskApi.c, pycocotools 2.0.x -- the package is NOT installed
    here and the reference checkout holds no such string: PARITY UNPINNED, restated from the published algorithm and pinned only by
    hand-worked strings in tests/test_oracle_golden.py).  Every count from the third on is stored as the difference to the count two
    places before; each value is written in 5-bit groups, least significant first, bit 5 = "more groups follow", bit 4 of the last
    group = the sign; every group + 48 is one ASCII character."""
    out = []
    for i, c in enumerate(counts):
        x = int(c) - (int(counts[i - 2]) if i > 2 else 0)
        more = True
        while more:
            g = x & 0x1F
            x >>= 5  # arithmetic shift
            more = (x != -1) if (g & 0x10) else (x != 0)
            if more:
                g |= 0x20
            out.append(chr(g + 48))
    return "".join(out)


def rle_counts_from_string(s):
    """Inverse of rle_counts_to_string (`rleFrString`, same source; what cocomask.decode does first with the string form that
    PEM/run_inference_custom_pytorch.py:312-317 falls back to when `frPyObjects` refuses the object)."""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            g = ord(s[p]) - 48
            x |= (g & 0x1F) << (5 * k)
            more = bool(g & 0x20)
            p += 1
            k += 1
            if not more and (g & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def xyxy_to_xywh(bbox):
    """ISM/utils/bbox_utils.py:129-138 (the 2-D branch has no +1; the 1-D branch has)."""
    bbox = np.asarray(bbox)
    if bbox.ndim == 1:
        x1, y1, x2, y2 = bbox
        return [x1, y1, x2 - x1 + 1, y2 - y1 + 1]
    if bbox.ndim == 2:
        return np.stack([bbox[:, 0], bbox[:, 1], bbox[:, 2] - bbox[:, 0], bbox[:, 3] - bbox[:, 1]], axis=1)
    raise ValueError("bbox must be a numpy array of shape (4,) or (N, 4)")


def detections_to_records(object_ids, scores, boxes_xyxy, masks, scene_id=0, frame_id=0, runtime=0, dataset_name="Custom"):
    """Detections.save_to_file (ISM/model/utils.py:153-173) followed by convert_npz_to_json (:199-216), without the file in between:
    the list that save_json_bop23 dumps as detection_ism.json."""
    object_ids = np.asarray(object_ids)
    cat = object_ids + 1 if dataset_name != "lmo" else LMO_OBJECT_IDS[object_ids]
    bbox = xyxy_to_xywh(np.asarray(boxes_xyxy))
    out = []
    for i in range(len(bbox)):
        out.append({"scene_id": int(scene_id), "image_id": int(frame_id), "category_id": int(cat[i]), "bbox": bbox[i].tolist(),
                    "score": float(np.asarray(scores)[i]), "time": float(runtime),
                    "segmentation": mask_to_rle(force_binary_mask(np.asarray(masks)[i]))})
    return out
