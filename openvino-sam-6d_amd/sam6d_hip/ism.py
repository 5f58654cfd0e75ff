"""Host side of the ISM template-scoring path on one MI355X: the functions behind
Instance_Segmentation_Model.compute_semantic_score / compute_appearance_score / project_template_to_image /
compute_geometric_score and model.loss.*  (ISM = SAM-6D/Instance_Segmentation_Model in the reference)."""
import torch

from . import _lib
from .pem import _empty, _p, _s, gemm, on_tensor_device

_MODES = {"avg_5": 0, "mean": 1, "max": 2}


@on_tensor_device
def pairwise_similarity(query, reference):
    """ISM/model/loss.py:27-44: query (Nq,D), reference (No,Nt,D) -> (Nq,No,Nt) in [0,1]."""
    Nq, D = query.shape
    No, Nt, _ = reference.shape
    out = _empty((Nq, No, Nt), query)
    _lib.call("sam6d_ism_cosine", _p(query), _p(reference), Nq, No, Nt, D, _p(out), _s())
    return out


@on_tensor_device
def semantic_select(scores, aggregation="avg_5", confidence_thresh=0.2):
    """ISM/model/detector.py:265-296 on precomputed scores (Nq,No,Nt).
    Returns idx_selected (K,) i64, pred_idx_objects (K,) i64, semantic_score (K,), best_template (K,) i64.
    (one host read-back of K, like the reference's boolean-mask indexing which also synchronises)"""
    if aggregation not in _MODES:
        raise NotImplementedError("aggregation_function %r (implemented: avg_5, mean, max)" % aggregation)
    Nq, No, Nt = scores.shape
    n = max(Nq, 1)
    ws_i = _empty((2 * n + 1,), scores, torch.int32)   # obj | best | count
    ws_f = _empty((2 * n,), scores)                    # sem (per query) | sem (compacted)
    out64 = _empty((3 * n,), scores, torch.int64)      # sel | obj | best, compacted
    _lib.call("sam6d_ism_semantic_compact", _p(scores), Nq, No, Nt, _MODES[aggregation], float(confidence_thresh), _p(ws_f), _p(ws_i),
              _p(ws_i, n), out64.data_ptr(), out64.data_ptr() + 8 * n, _p(ws_f, n), out64.data_ptr() + 16 * n, _p(ws_i, 2 * n), _s())
    k = int(ws_i[2 * n].item())  # the one host read-back of the pass (the survivors' count fixes every later shape)
    return out64[:k], out64[n:n + k], ws_f[n:n + k], out64[2 * n:2 * n + k]


@on_tensor_device
def patch_similarity(q_appe, ref_sel):
    """sim (Ns,P,P) = q_appe (Ns,P,D) @ ref_sel (Ns,P,D)^T on the matrix cores (ISM/model/loss.py:54,66)."""
    Ns, P, D = q_appe.shape
    sim = _empty((Ns, P, P), q_appe)
    gemm(q_appe, ref_sel, None, sim, P, P, D, D, D, P, batch=Ns, sA=P * D, sW=P * D, sC=P * P)
    return sim


@on_tensor_device
def patch_scores(sim, q_appe, thred=0.5):
    """appearance score (loss.py:52-62) and visible ratio (loss.py:64-76) from one similarity tensor."""
    Ns, P, D = q_appe.shape
    appe = _empty((Ns,), q_appe)
    vis = _empty((Ns,), q_appe)
    _lib.call("sam6d_ism_patch_scores", _p(sim), _p(q_appe), Ns, P, D, float(thred), _p(appe), _p(vis), _s())
    return appe, vis


class PatchScores:
    """Row / column maxima of the patch similarity of every selected proposal against its best template (the workspace of
    sam6d_ism_patch_fused): what compute_appearance_score and compute_geometric_score both need, without the (Ns,P,P) tensor."""
    __slots__ = ("ws", "Ns", "P", "q_ptr", "obj", "best", "ref")

    def scores(self, thred=0.5):
        """-> appearance score (Ns,), visible ratio (Ns,)   (ISM/model/loss.py:52-62, 64-76)"""
        dev = self.ws.device
        appe = torch.empty(self.Ns, dtype=torch.float32, device=dev)
        vis = torch.empty(self.Ns, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sam6d_ism_patch_fused_scores", self.ws.data_ptr(), self.Ns, self.P, float(thred), _p(appe), _p(vis), _s())
        return appe, vis

    def gathered_reference(self):
        """the (Ns,P,D) tensor the reference's compute_appearance_score returns (detector.py:303) -- built only when asked for"""
        return self.ref[self.obj, self.best, ...].contiguous()


@on_tensor_device
def patch_scores_fused(q_appe, ref_appe, pred_obj, best_pose, q_index=None):
    """q_appe (Nq,P,D) query patch descriptors (rows q_index when given, else Nq == Ns), ref_appe (No,Nt,P,D) ALL templates' patch
    descriptors, pred_obj / best_pose (Ns,) i64 -> PatchScores.  One GEMM launch whose tiles end in row / column maxima."""
    from .ops import _chk
    q_appe = q_appe.contiguous()
    ref_appe = ref_appe.contiguous()
    _chk(q_appe, "q_appe", torch.float32, 3)
    _chk(ref_appe, "ref_appe", torch.float32, 4)
    No, Nt, P, D = ref_appe.shape
    obj = pred_obj.to(torch.int64).contiguous()
    best = best_pose.to(torch.int64).contiguous()
    qi = q_index.to(torch.int64).contiguous() if q_index is not None else None
    Ns = obj.shape[0]
    if q_appe.shape[1:] != (P, D) or best.shape[0] != Ns or (qi is None and q_appe.shape[0] != Ns) or (qi is not None and qi.shape[0] != Ns):
        raise RuntimeError("patch_scores_fused: shapes do not match (q %s, ref %s, %d proposals)" % (tuple(q_appe.shape), tuple(ref_appe.shape), Ns))
    if P % 128 or D % 32:
        raise NotImplementedError("patch_scores_fused: P must be a multiple of 128 and D of 32 (got %d, %d)" % (P, D))
    nbytes = int(_lib.load().sam6d_ism_patch_fused_workspace_bytes(Ns, P))
    ps = PatchScores()
    ps.ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q_appe.device)
    ps.Ns, ps.P, ps.q_ptr, ps.obj, ps.best, ps.ref = Ns, P, q_appe.data_ptr(), obj, best, ref_appe
    _lib.call("sam6d_ism_patch_fused", _p(q_appe), _p(qi), _p(ref_appe), _p(obj), _p(best), Ns, Nt, P, D, ps.ws.data_ptr(), nbytes, _s())
    return ps


@on_tensor_device
def project_template_to_image(best_pose, pred_obj, poses, pointcloud, masks, depth, K, depth_scale, mask_index=None):
    """ISM/model/detector.py:209-246: -> image_vu (Ns,Npc,2) i32, xyxy (Ns,4) i32, translate (Ns,3).
    masks (Nq,H,W): uint8 / bool (SAM's binary proposals, one byte per pixel) or float32 (Detections.masks), read in place;
    mask_index (Ns,) i64: proposal i uses masks[mask_index[i]] (the class-token selection without a gathered copy), else Nq == Ns."""
    Nq, H, W = masks.shape
    Npc = pointcloud.shape[1]
    dev = masks.device
    if masks.dtype == torch.bool:
        masks = masks.view(torch.uint8)
    elif masks.dtype not in (torch.uint8, torch.float32):
        masks = masks.to(torch.float32)
    masks = masks.contiguous()
    depth = depth.to(torch.int32).contiguous()
    Kd = K.to(device=dev, dtype=torch.float64).contiguous()
    best = best_pose.to(torch.int32).contiguous()
    obj = pred_obj.to(torch.int32).contiguous()
    Ns = best.shape[0]
    mi = None
    if mask_index is not None:
        mi = mask_index.to(device=dev, dtype=torch.int64).contiguous()
        if mi.shape[0] != Ns:
            raise RuntimeError("project_template_to_image: mask_index must have one entry per proposal")
    elif Nq != Ns:
        raise RuntimeError("project_template_to_image: %d masks for %d proposals (pass mask_index)" % (Nq, Ns))
    vu = torch.empty(Ns, Npc, 2, dtype=torch.int32, device=dev)
    xyxy = torch.empty(Ns, 4, dtype=torch.int32, device=dev)
    tr = torch.empty(Ns, 3, dtype=torch.float32, device=dev)
    fast = W % 16 == 0 and float(depth_scale) > 0.0
    if not fast:  # general shapes: float32 masks, gathered
        masks = masks.to(torch.float32)
        if mi is not None:
            masks = take_rows(masks, mi)
        part = torch.empty(max(Ns, 1) * 64 * 4, dtype=torch.float64, device=dev)
        _lib.call("sam6d_ism_project", _p(masks), _p(depth), Kd.data_ptr(), float(depth_scale), _p(poses.contiguous()),
                  _p(pointcloud.contiguous()), _p(best), _p(obj), Ns, H, W, Npc, part.data_ptr(), vu.data_ptr(), xyxy.data_ptr(),
                  _p(tr), _s())
        return vu, xyxy, tr
    part = torch.empty(max(int(_lib.load().sam6d_ism_project_workspace_doubles(Ns, H, W)), 1), dtype=torch.float64, device=dev)
    _lib.call("sam6d_ism_project2", masks.data_ptr(), masks.element_size(), (mi.data_ptr() if mi is not None else None), _p(depth),
              Kd.data_ptr(), float(depth_scale), _p(poses.contiguous()), _p(pointcloud.contiguous()), _p(best), _p(obj), Ns, H, W, Npc,
              part.data_ptr(), vu.data_ptr(), xyxy.data_ptr(), _p(tr), _s())
    return vu, xyxy, tr


@on_tensor_device
def translate_masked_depth_maps(masked_depth, K, depth_scale):
    """ISM/utils/trimesh_utils.py:77-105 for all N already-masked depth maps (N,H,W) in one launch -> translate (N,3) f32."""
    N, H, W = masked_depth.shape
    dev = masked_depth.device
    md = masked_depth.to(torch.float32).contiguous()
    Kd = torch.as_tensor(K).to(device=dev, dtype=torch.float64).contiguous()
    part = torch.empty(max(N, 1) * 64 * 4, dtype=torch.float64, device=dev)
    tr = torch.empty(N, 3, dtype=torch.float32, device=dev)
    _lib.call("sam6d_ism_translate_maps", _p(md), Kd.data_ptr(), float(depth_scale), N, H, W, part.data_ptr(), _p(tr), _s())
    return tr


@on_tensor_device
def compute_iou(xyxy, boxes, return_flag=False):
    """ISM/utils/bbox_utils.py:197-222 incl. the quirk: any non-positive overlap => the python float 0.0.
    return_flag=True: (iou (Ns,), all_positive (1,) i32 device flag) without the host read-back -- final_score(..., all_positive=flag)
    applies the quirk on the device."""
    Ns = xyxy.shape[0]
    iou = torch.empty(Ns, dtype=torch.float32, device=xyxy.device)
    flag = torch.empty(1, dtype=torch.int32, device=xyxy.device)
    a = xyxy.to(torch.int32).contiguous()
    b = boxes.to(torch.int64).contiguous()
    _lib.call("sam6d_ism_iou", a.data_ptr(), b.data_ptr(), Ns, _p(iou), flag.data_ptr(), _s())
    if return_flag:
        return iou, flag
    return iou if int(flag.item()) == 1 else 0.0


@on_tensor_device
def final_score(sem, appe, geo, vis, all_positive=None):
    """ISM/model/detector.py:384.  geo: the IoU tensor, or the float 0.0 of the quirk; with all_positive (the device flag of
    compute_iou(return_flag=True)) the quirk is decided on the device."""
    Ns = appe.shape[0]
    out = _empty((Ns,), appe)
    g = geo if torch.is_tensor(geo) else None
    if all_positive is not None and g is not None:
        _lib.call("sam6d_ism_final_score_flag", _p(sem.contiguous()), _p(appe), _p(g), _p(vis), all_positive.data_ptr(), Ns, _p(out), _s())
        return out
    _lib.call("sam6d_ism_final_score", _p(sem.contiguous()), _p(appe), _p(g), _p(vis), None, Ns, _p(out), _s())
    return out


@on_tensor_device
def masked_patch_features(patch_features, masks, patch_size=14, validpatch_thresh=0.5):
    """CustomDINOv2's descriptor post-processing (ISM/model/dinov2.py:265-269, 322-324): zero the patches whose mask
    coverage (AvgPool2d(patch_size)) is <= validpatch_thresh and L2-normalise the rest.  patch_features (N,P,D),
    masks (N,H,W) -> (N,P,D)."""
    N, P, D = patch_features.shape
    H, W = masks.shape[1], masks.shape[2]
    from .ops import _chk
    f = patch_features.contiguous()
    m = masks.to(torch.float32).contiguous()
    _chk(f, "patch_features", torch.float32, 3)
    _chk(m, "masks", torch.float32, 3)
    out = _empty((N, P, D), f)
    _lib.call("sam6d_masked_patch_normalize", _p(f), _p(m), N, P, D, H, W, int(patch_size), float(validpatch_thresh), _p(out),
              _s())
    return out


# ------------------------------------------------------------------ Detections bookkeeping (ISM/model/utils.py:84-196)
@on_tensor_device
def small_detection_keep(boxes, masks, min_box_size, min_mask_size):
    """ISM/model/utils.py:96-102: bool (N,) -- box area and mask area (as fractions of the image) above the thresholds."""
    from .ops import _chk
    boxes = boxes.contiguous()
    masks = masks.to(torch.float32).contiguous()
    _chk(boxes, "boxes", torch.int64, 2)
    _chk(masks, "masks", torch.float32, 3)
    N, H, W = masks.shape
    keep = _empty((N,), masks, torch.uint8)
    _lib.call("sam6d_detections_small_keep", _p(boxes), _p(masks), N, H, W, float(min_box_size ** 2), float(min_mask_size),
              _p(keep), _s())
    return keep.bool()


@on_tensor_device
def mask_to_indices(keep):
    """nonzero(keep) as int64 indices (one host read-back of the count, like boolean-mask indexing in torch)."""
    k8 = keep.to(torch.uint8).contiguous()
    N = k8.shape[0]
    idx = _empty((max(N, 1),), k8, torch.int64)
    cnt = _empty((1,), k8, torch.int32)
    _lib.call("sam6d_mask_to_indices", _p(k8), N, _p(idx), _p(cnt), _s())
    return idx[: int(cnt.item())]


@on_tensor_device
def take_rows(src, idx):
    """src[idx] for an int64 index vector or a bool mask, any dtype (ISM/model/utils.py:105,119,126,190)."""
    if not src.is_cuda:
        raise RuntimeError("take_rows: src must be a HIP device tensor (no CPU path)")
    if idx.dtype == torch.bool or idx.dtype == torch.uint8:
        idx = mask_to_indices(idx)
    idx = idx.to(torch.int64).contiguous()
    src = src.contiguous()
    M = idx.shape[0]
    out = torch.empty((M,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    row_bytes = src.element_size()
    for d in src.shape[1:]:
        row_bytes *= d
    if M and row_bytes:
        _lib.call("sam6d_take_rows", _p(src), _p(idx), src.shape[0], M, row_bytes, _p(out), _s())
    return out


@on_tensor_device
def nms(boxes, scores, iou_threshold, object_ids=None):
    """torchvision.ops.nms as ISM/model/utils.py:107-126 calls it; with object_ids the per-id variant (ids ascending,
    survivors of each id in descending score order).  Returns int64 indices."""
    from .ops import _chk
    boxes = boxes.to(torch.float32).contiguous()
    scores = scores.to(torch.float32).contiguous()
    _chk(boxes, "boxes", torch.float32, 2)
    _chk(scores, "scores", torch.float32, 1)
    N = boxes.shape[0]
    grp = None
    if object_ids is not None:
        grp = object_ids.to(torch.int64).contiguous()
        _chk(grp, "object_ids", torch.int64, 1)
    keep = _empty((max(N, 1),), boxes, torch.int64)
    cnt = _empty((1,), boxes, torch.int32)
    nbytes = int(_lib.load().sam6d_nms_workspace_bytes(N))
    ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=boxes.device)
    _lib.call("sam6d_nms", _p(boxes), _p(scores), _p(grp), N, float(iou_threshold), _p(keep), _p(cnt), _p(ws), nbytes, _s())
    return keep[: int(cnt.item())]


def mask_rle_encode(masks):
    """Uncompressed COCO RLE of (masks > 0), column-major runs starting with the zero run
    (mask_to_rle(force_binary_mask(m)), ISM/model/utils.py:25-43, 211-213).  masks (N,H,W) on the HIP device (any real dtype).
    Returns (counts i32 (total,), offsets i64 (N+1,)) on the device: mask i's runs are counts[offsets[i]:offsets[i+1]]."""
    if not masks.is_cuda:
        raise RuntimeError("mask_rle_encode: masks must be a HIP device tensor (no CPU path)")
    if masks.dim() != 3:
        raise RuntimeError("mask_rle_encode: masks must be (N,H,W)")
    m = masks.to(torch.float32).contiguous()
    N, H, W = m.shape
    if N == 0:
        return torch.empty(0, dtype=torch.int32, device=m.device), torch.zeros(1, dtype=torch.int64, device=m.device)
    with torch.cuda.device(m.device):
        nruns = torch.empty(N, dtype=torch.int32, device=m.device)
        _lib.call("sam6d_mask_rle_count", _p(m), N, H, W, _p(nruns), _s())
        offsets = torch.zeros(N + 1, dtype=torch.int64, device=m.device)
        if N:
            offsets[1:] = torch.cumsum(nruns.to(torch.int64), 0)
        total = int(offsets[-1].item())
        counts = torch.empty(total, dtype=torch.int32, device=m.device)
        _lib.call("sam6d_mask_rle_encode", _p(m), N, H, W, _p(offsets), _p(counts), _s())
    return counts, offsets


def mask_to_rle(masks):
    """List of {"counts": [...], "size": [H, W]} dicts, one per mask: what convert_npz_to_json stores under "segmentation"
    (ISM/model/utils.py:199-216)."""
    counts, offsets = mask_rle_encode(masks)
    c = counts.cpu().tolist()
    o = offsets.cpu().tolist()
    H, W = int(masks.shape[1]), int(masks.shape[2])
    return [{"counts": c[o[i]:o[i + 1]], "size": [H, W]} for i in range(len(o) - 1)]


def rle_string_to_counts(s):
    """The counts of a pycocotools COMPRESSED RLE string (cocoapi maskApi.c rleFrString): 5-bit groups, least significant first, + 48 per
    character, bit 5 = continuation, bit 4 of the last group = sign; counts from the third on are differences to the count two places
    before.  Host-side text decoding (a few hundred characters per mask); the runs are expanded on the device (rle_to_mask)."""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p, n = [], 0, len(s)
    while p < n:
        x, k, more = 0, 0, True
        while more:
            if p >= n:
                raise RuntimeError("rle_string_to_counts: truncated string")
            g = ord(s[p]) - 48
            if g < 0 or g > 63:
                raise RuntimeError("rle_string_to_counts: character %r is outside the RLE alphabet" % s[p])
            x |= (g & 0x1F) << (5 * k)
            more = bool(g & 0x20)
            p += 1
            k += 1
            if not more and (g & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        if x < 0:
            raise RuntimeError("rle_string_to_counts: negative run length")
        counts.append(x)
    return counts


def rle_counts_to_string(counts):
    """Inverse of rle_string_to_counts (cocoapi maskApi.c rleToString)."""
    out = []
    for i, c in enumerate(counts):
        x = int(c) - (int(counts[i - 2]) if i > 2 else 0)
        more = True
        while more:
            g = x & 0x1F
            x >>= 5
            more = (x != -1) if (g & 0x10) else (x != 0)
            if more:
                g |= 0x20
            out.append(chr(g + 48))
    return "".join(out)


def rle_to_mask(rles, device):
    """Inverse of mask_to_rle for a list of RLE dicts of one size -> (N,H,W) uint8 on `device` (what cocomask.decode returns in
    PEM/run_inference_custom_pytorch.py:312-317): uncompressed counts lists as the ISM side writes them, or pycocotools' compressed
    counts strings (the form `rle = seg` takes there when frPyObjects refuses the object)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("rle_to_mask: needs a HIP device (no CPU path)")
    if len(rles) == 0:
        return torch.zeros(0, 0, 0, dtype=torch.uint8, device=device)
    H, W = (int(v) for v in rles[0]["size"])
    flat, offs = [], [0]
    for r in rles:
        if [int(v) for v in r["size"]] != [H, W]:
            raise RuntimeError("rle_to_mask: all masks must have one size")
        c = rle_string_to_counts(r["counts"]) if isinstance(r["counts"], (str, bytes)) else r["counts"]
        flat.extend(int(v) for v in c)
        offs.append(len(flat))
    N = len(rles)
    with torch.cuda.device(device):
        counts = torch.tensor(flat if flat else [0], dtype=torch.int32, device=device)
        offsets = torch.tensor(offs, dtype=torch.int64, device=device)
        ends = torch.empty_like(counts)
        out = torch.empty(N, H, W, dtype=torch.uint8, device=device)
        _lib.call("sam6d_mask_rle_decode", _p(counts), _p(offsets), N, H, W, _p(ends), _p(out), _s())
    return out
