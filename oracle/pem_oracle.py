"""TEST INFRASTRUCTURE ONLY -- CPU oracle of SAM-6D's PEM geometric-matching hot path.

A functional (no nn.Module) restatement, on torch-CPU float32 tensors, of the reference's algorithm for the
path BASELINE.json's north_star names.  Each function cites the reference file:line it follows
(PEM = SAM-6D/Pose_Estimation_Model, EXT = PEM/model/pointnet2/_ext_src).  The point-cloud primitives and the
K=3 pairwise distance come from oracle/pointops_oracle.c (plain C).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(openvino-sam-6d_amd/) never does.

Parity pin: tests/test_oracle_golden.py compares every function here with vectors captured from the reference's
own Python modules + its own compiled `_ext` (oracle/gen_golden.py, run in the build container only).

Weights are passed as a flat dict with the reference's state_dict keys.
"""
import math

import torch
import torch.nn.functional as F

from . import pointops

# ----------------------------------------------------------------------------------------------- config
DEFAULT_CFG = dict(  # PEM/config/base.yaml:16-54
    coarse_npoint=196, fine_npoint=2048, hidden_dim=256, num_heads=4,
    sigma_d=0.2, sigma_a=15, angle_k=3, nblock=3, temp=0.1,
    nproposal1=6000, nproposal2=300, pe_radius1=0.1, pe_radius2=0.2, pe_nsample1=32, pe_nsample2=64,
    focusing_factor=3, dis_thres=0.15,
)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


# ------------------------------------------------------------------------------------- sampling (a1, a2)
def sample_pts_feats(pts, feats, npoint):
    """PEM/utils/model_utils.py:70-84: FPS + two gathers.  pts (B,N,3), feats (B,N,C)."""
    idx = pointops.furthest_point_sampling(pts.contiguous(), npoint)
    p = pointops.gather_points(pts.transpose(1, 2).contiguous(), idx).transpose(1, 2)
    f = pointops.gather_points(feats.transpose(1, 2).contiguous(), idx).transpose(1, 2)
    return p, f, idx


# ---------------------------------------------------------------------------------------- distances (a6)
def pairwise_distance(x, y):
    """PEM/utils/model_utils.py:101-128 for 3-channel points (bit recipe in pointops_oracle.c)."""
    return pointops.pairwise_distance(x, y)


# ------------------------------------------------------------------------------- geometric embedding (a7)
def sinusoidal_embedding(idx, div_term):
    """PEM/model/transformer.py:259-285: interleaved [sin(w_i x), cos(w_i x)]."""
    shp = idx.shape
    om = idx.reshape(-1, 1, 1) * div_term.view(1, -1, 1)
    emb = torch.cat([torch.sin(om), torch.cos(om)], dim=2)
    return emb.view(*shp, div_term.numel() * 2)


def geo_embedding_indices(points, sigma_d=0.2, sigma_a=15, angle_k=3):
    """PEM/model/transformer.py:306-341 (get_embedding_indices)."""
    B, N, _ = points.shape
    dist = torch.sqrt(pairwise_distance(points, points))
    d_idx = dist / sigma_d
    k = angle_k
    knn = dist.topk(k=k + 1, dim=2, largest=False)[1][:, :, 1:]  # (B,N,k)
    knn_pts = torch.gather(points.unsqueeze(1).expand(B, N, N, 3), 2, knn.unsqueeze(3).expand(B, N, k, 3))
    ref = knn_pts - points.unsqueeze(2)  # (B,N,k,3)
    anc = points.unsqueeze(1) - points.unsqueeze(2)  # (B,N,N,3)
    ref = ref.unsqueeze(2).expand(B, N, N, k, 3)
    anc = anc.unsqueeze(3).expand(B, N, N, k, 3)
    sin_v = torch.linalg.norm(torch.cross(ref, anc, dim=-1), dim=-1)
    cos_v = torch.sum(ref * anc, dim=-1)
    sin_v = torch.clamp(sin_v, min=1e-8)
    cos_v = torch.clamp(cos_v, min=-1.0 + 1e-8, max=1.0 - 1e-8)
    a_idx = torch.atan2(sin_v, cos_v) * (180.0 / (sigma_a * math.pi))
    return d_idx, a_idx, knn


def geo_embedding(points, sd, p="geo_embedding", sigma_d=0.2, sigma_a=15, angle_k=3):
    """PEM/model/transformer.py:343-363 (reduction_a == 'max').  points (B,N,3) -> (B,N,N,C)."""
    d_idx, a_idx, _ = geo_embedding_indices(points, sigma_d, sigma_a, angle_k)
    div = sd[p + ".embedding.div_term"]
    d = _lin(sinusoidal_embedding(d_idx, div), sd, p + ".proj_d")
    a = _lin(sinusoidal_embedding(a_idx, div), sd, p + ".proj_a").max(dim=3)[0]
    return d + a


# ----------------------------------------------------------------------------------- transformer (a8, a13)
def _heads(x, h):
    B, N, C = x.shape
    return x.view(B, N, h, C // h).permute(0, 2, 1, 3)  # b h n c


def _attn_output(x, sd, p):
    """PEM/model/transformer.py:184-199 AttentionOutput."""
    h = _lin(F.relu(_lin(x, sd, p + ".expand")), sd, p + ".squeeze")
    return F.layer_norm(x + h, (x.shape[-1],), sd[p + ".norm.weight"], sd[p + ".norm.bias"])


def rpe_transformer_layer(x, mem, emb, sd, p, H=4):
    """PEM/model/transformer.py:366-479 RPE self layer: softmax(((q.k)+(q.proj_p(E)))/sqrt(c)) v -> linear ->
    +res -> LN -> FFN -> +res -> LN."""
    a = p + ".attention.attention"
    q = _heads(_lin(x, sd, a + ".proj_q"), H)
    k = _heads(_lin(mem, sd, a + ".proj_k"), H)
    v = _heads(_lin(mem, sd, a + ".proj_v"), H)
    B, N, M, C = emb.shape
    pe = _lin(emb, sd, a + ".proj_p").view(B, N, M, H, C // H).permute(0, 3, 1, 2, 4)  # b h n m c
    s_p = torch.einsum("bhnc,bhnmc->bhnm", q, pe)
    s_e = torch.einsum("bhnc,bhmc->bhnm", q, k)
    s = (s_e + s_p) / (C // H) ** 0.5
    s = F.softmax(s, dim=-1)
    hid = torch.matmul(s, v).permute(0, 2, 1, 3).reshape(B, N, C)
    hid = _lin(hid, sd, p + ".attention.linear")
    out = F.layer_norm(hid + x, (C,), sd[p + ".attention.norm.weight"], sd[p + ".attention.norm.bias"])
    return _attn_output(out, sd, p + ".output")


def transformer_layer(x, mem, sd, p, H=4):
    """PEM/model/transformer.py:95-226 vanilla cross layer."""
    a = p + ".attention.attention"
    B, N, C = x.shape
    q = _heads(_lin(x, sd, a + ".proj_q"), H)
    k = _heads(_lin(mem, sd, a + ".proj_k"), H)
    v = _heads(_lin(mem, sd, a + ".proj_v"), H)
    s = torch.einsum("bhnc,bhmc->bhnm", q, k) / (C // H) ** 0.5
    s = F.softmax(s, dim=-1)
    hid = torch.matmul(s, v).permute(0, 2, 1, 3).reshape(B, N, C)
    hid = _lin(hid, sd, p + ".attention.linear")
    out = F.layer_norm(hid + x, (C,), sd[p + ".attention.norm.weight"], sd[p + ".attention.norm.bias"])
    return _attn_output(out, sd, p + ".output")


def geometric_transformer(f0, e0, f1, e1, sd, p, H=4):
    """PEM/model/transformer.py:483-527, blocks=['self','cross'], parallel=False (sequential cross)."""
    f0 = rpe_transformer_layer(f0, f0, e0, sd, p + ".layers.0", H)
    f1 = rpe_transformer_layer(f1, f1, e1, sd, p + ".layers.0", H)
    f0 = transformer_layer(f0, f1, sd, p + ".layers.1", H)
    f1 = transformer_layer(f1, f0, sd, p + ".layers.1", H)
    return f0, f1


def linear_attention(xq, xm, sd, p, H=4, focusing_factor=3):
    """PEM/model/transformer.py:532-578 focused linear attention (kv path)."""
    q = _lin(xq, sd, p + ".proj_q")
    k = _lin(xm, sd, p + ".proj_k")
    v = _lin(xm, sd, p + ".proj_v")
    scale = F.softplus(sd[p + ".scale"])
    q = F.relu(q) + 1e-6
    k = F.relu(k) + 1e-6
    q = q / scale
    k = k / scale
    qn = q.norm(dim=-1, keepdim=True)
    kn = k.norm(dim=-1, keepdim=True)
    q = q ** focusing_factor
    k = k ** focusing_factor
    q = (q / q.norm(dim=-1, keepdim=True)) * qn
    k = (k / k.norm(dim=-1, keepdim=True)) * kn
    B, I, C = q.shape
    J = k.shape[1]
    c = C // H

    def sp(x):
        return x.view(B, -1, H, c).permute(0, 2, 1, 3).reshape(B * H, -1, c)

    q, k, v = sp(q), sp(k), sp(v)
    z = 1 / (torch.einsum("bic,bc->bi", q, k.sum(dim=1)) + 1e-6)
    if I * J * (c + c) > c * c * (I + J):
        kv = torch.einsum("bjc,bjd->bcd", k, v)
        x = torch.einsum("bic,bcd,bi->bid", q, kv, z)
    else:
        qk = torch.einsum("bic,bjc->bij", q, k)
        x = torch.einsum("bij,bjd,bi->bid", qk, v, z)
    return x.view(B, H, I, c).permute(0, 2, 1, 3).reshape(B, I, C)


def linear_transformer_layer(x, mem, sd, p, H=4, focusing_factor=3):
    """PEM/model/transformer.py:581-622."""
    hid = linear_attention(x, mem, sd, p + ".attention.attention", H, focusing_factor)
    hid = _lin(hid, sd, p + ".attention.linear")
    out = F.layer_norm(hid + x, (x.shape[-1],), sd[p + ".attention.norm.weight"], sd[p + ".attention.norm.bias"])
    return _attn_output(out, sd, p + ".output")


def _sample_feats(dense, fps_idx):
    """PEM/model/transformer.py:667-705: gather on the tensor that still holds the bg token at index 0
    (so fps_idx=i picks dense point i-1 and fps_idx=0 picks the bg token) -- reproduced, not fixed."""
    bg = dense[:, 0:1, :]
    g = pointops.gather_points(dense.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2)
    return torch.cat([bg, g], dim=1)


def sparse_to_dense_transformer(d0, e0, i0, d1, e1, i1, sd, p, H=4, focusing_factor=3):
    """PEM/model/transformer.py:627-720 (with_bg_token, replace_bg_token)."""
    s0 = _sample_feats(d0, i0)
    s1 = _sample_feats(d1, i1)
    s0, s1 = geometric_transformer(s0, e0, s1, e1, sd, p + ".sparse_layer", H)

    def lift(d, s):
        x = linear_transformer_layer(d[:, 1:, :].contiguous(), s[:, 1:, :].contiguous(), sd, p + ".dense_layer", H,
                                     focusing_factor)
        return torch.cat([s[:, 0:1, :], x], dim=1)

    return lift(d0, s0), lift(d1, s1)


# --------------------------------------------------------------------------- positional encoding (a3-a5)
def _shared_mlp(x, sd, p, nlayer=3, eps=1e-5):
    """PEM/model/pointnet2/pytorch_utils.py:25-50: Conv2d 1x1 (no bias) + BatchNorm2d(eval) + ReLU, x3."""
    for i in range(nlayer):
        q = "%s.layer%d" % (p, i)
        x = F.conv2d(x, sd[q + ".conv.weight"])
        x = F.batch_norm(x, sd[q + ".normlayer.bn.running_mean"], sd[q + ".normlayer.bn.running_var"],
                         sd[q + ".normlayer.bn.weight"], sd[q + ".normlayer.bn.bias"], False, 0.0, eps)
        x = F.relu(x)
    return x


def query_and_group(xyz, new_xyz, radius, nsample):
    """PEM/model/pointnet2/pointnet2_utils.py:326-403 with features = xyz^T, use_xyz=True -> (B,6,M,S)."""
    idx = pointops.ball_query(new_xyz.contiguous(), xyz.contiguous(), radius, nsample)
    xt = xyz.transpose(1, 2).contiguous()
    g = pointops.group_points(xt, idx)
    rel = g - new_xyz.transpose(1, 2).unsqueeze(-1)
    return torch.cat([rel, g], dim=1), idx


def positional_encoding(pts, sd, p, r1=0.1, r2=0.2, ns1=32, ns2=64):
    """PEM/model/fine_point_matching.py:102-144.  pts (B,N,3) -> (B,N,256)."""
    pts = pts.contiguous()
    q = pts + 0.00000001
    f1, _ = query_and_group(pts, q, r1, ns1)
    f1 = torch.amax(_shared_mlp(f1, sd, p + ".mlp1"), dim=3, keepdim=True)
    f2, _ = query_and_group(pts, q, r2, ns2)
    f2 = torch.amax(_shared_mlp(f2, sd, p + ".mlp2"), dim=3, keepdim=True)
    f = torch.cat([f1, f2], dim=1).squeeze(-1)
    f = F.conv1d(f, sd[p + ".mlp3.conv.weight"], sd[p + ".mlp3.conv.bias"])
    return f.transpose(1, 2)


# -------------------------------------------------------------------------------------- similarity (a9)
def feature_similarity(f1, f2, temp=0.1):
    """PEM/utils/model_utils.py:131-153 (cosine, normalize_feat=True)."""
    f1 = F.normalize(f1, p=2, dim=2)
    f2 = F.normalize(f2, p=2, dim=2)
    return (f1 @ f2.transpose(1, 2)) / temp


# ------------------------------------------------------------------------------------- procrustes (a12)
def weighted_procrustes(src, ref, weights=None, weight_thresh=0.0, eps=1e-5):
    """PEM/utils/model_utils.py:343-436: rigid transform src -> ref by weighted SVD (torch.svd + torch.det)."""
    B = src.shape[0]
    if weights is None:
        weights = torch.ones_like(src[:, :, 0])
    weights = torch.where(weights < weight_thresh, torch.zeros_like(weights), weights)
    weights = (weights / (weights.sum(dim=1, keepdim=True) + eps)).unsqueeze(2)
    sc = (src * weights).sum(dim=1, keepdim=True)
    rc = (ref * weights).sum(dim=1, keepdim=True)
    Hm = (src - sc).permute(0, 2, 1) @ (weights * (ref - rc))
    U, _, V = torch.svd(Hm)
    Ut = U.transpose(1, 2)
    eye = torch.eye(3).unsqueeze(0).repeat(B, 1, 1)
    eye[:, -1, -1] = torch.sign(torch.det(V @ Ut))
    R = V @ eye @ Ut
    t = (rc.permute(0, 2, 1) - R @ sc.permute(0, 2, 1)).squeeze(2)
    return R, t


# -------------------------------------------------------------------------------- coarse pose (a10)
def soft_assignment(atten):
    """Shared head of compute_coarse_Rt / compute_fine_Rt (PEM/utils/model_utils.py:229-238, 320-325)."""
    S = torch.softmax(atten, dim=2) * torch.softmax(atten, dim=1)
    l1 = torch.max(S[:, 1:, :], dim=2)[1]
    l2 = torch.max(S[:, :, 1:], dim=1)[1]
    return S, l1, l2


def coarse_sampling_weights(atten):
    """PEM/utils/model_utils.py:229-238: (B, N1*N2) un-normalised sampling weights and the fg mask w1."""
    B = atten.shape[0]
    S, l1, l2 = soft_assignment(atten)
    w1 = (l1 > 0).float()
    w2 = (l2 > 0).float()
    S = S[:, 1:, 1:].contiguous() * w1.unsqueeze(2) * w2.unsqueeze(1)
    return S.reshape(B, -1) ** 1.5, w1


def weighted_sampling(weights, rand, faithful=False, chunk=256):
    """PEM/utils/model_utils.py:241-242 + 277-305: cumsum (double accumulate) / (last + 1e-8), then
    idx = argmax((cum >= u).float()).  `rand` (B, ns) replaces the reference's torch.rand draw.
    faithful=True materialises the (chunk, N) comparison like the reference does (same cost class);
    otherwise an equivalent binary search (first i with cum[i] >= u, 0 if none)."""
    cum = torch.cumsum(weights, dim=1)
    cum = cum / (cum[:, -1].unsqueeze(1).contiguous() + 1e-8)
    if not faithful:
        return pointops.first_ge(cum.contiguous(), rand.contiguous())
    B, ns = rand.shape
    out = torch.empty(B, ns, dtype=torch.int64)
    for s in range(0, ns, chunk):
        cmp = cum.unsqueeze(1) >= rand[:, s:s + chunk].unsqueeze(2)
        out[:, s:s + chunk] = torch.argmax(cmp.float(), dim=2)
    return out


def coarse_hypotheses(idx, pts1, pts2, n_proposal1):
    """PEM/utils/model_utils.py:244-257: triples -> 3-point Procrustes -> mean residual per hypothesis."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    idx1 = torch.clamp(idx.div(N2, rounding_mode="floor"), max=N1 - 1).unsqueeze(2).repeat(1, 1, 3)
    idx2 = torch.clamp(idx % N2, max=N2 - 1).unsqueeze(2).repeat(1, 1, 3)
    p1 = torch.gather(pts1, 1, idx1).reshape(B * n_proposal1, 3, 3)
    p2 = torch.gather(pts2, 1, idx2).reshape(B * n_proposal1, 3, 3)
    Rs, ts = weighted_procrustes(p2, p1, None, weight_thresh=0.5)
    Rs = Rs.reshape(B, n_proposal1, 3, 3)
    ts = ts.reshape(B, n_proposal1, 1, 3)
    p1 = p1.reshape(B, n_proposal1, 3, 3)
    p2 = p2.reshape(B, n_proposal1, 3, 3)
    dis = torch.norm((p1 - ts) @ Rs - p2, dim=3).mean(2)
    return Rs, ts, dis


def coarse_select(Rs, ts, dis, w1, pts1, model_pts, n_proposal2):
    """PEM/utils/model_utils.py:258-275: keep the n_proposal2 lowest-residual hypotheses, score each against
    the CAD points, return the argmax pose."""
    B = pts1.shape[0]
    idx = torch.topk(dis, n_proposal2, dim=1, largest=False)[1]
    Rs = torch.gather(Rs, 1, idx.reshape(B, n_proposal2, 1, 1).repeat(1, 1, 3, 3))
    ts = torch.gather(ts, 1, idx.reshape(B, n_proposal2, 1, 1).repeat(1, 1, 1, 3))
    tp = ((pts1.unsqueeze(1) - ts) @ Rs).reshape(B * n_proposal2, -1, 3)
    em = model_pts.unsqueeze(1).repeat(1, n_proposal2, 1, 1).reshape(B * n_proposal2, -1, 3)
    d = torch.sqrt(pairwise_distance(tp, em)).min(2)[0].reshape(B, n_proposal2, -1)
    scores = w1.unsqueeze(1).sum(2) / ((d * w1.unsqueeze(1)).sum(2) + 1e-8)
    best = scores.max(1)[1]
    R = torch.gather(Rs, 1, best.reshape(B, 1, 1, 1).repeat(1, 1, 3, 3)).squeeze(1)
    t = torch.gather(ts, 1, best.reshape(B, 1, 1, 1).repeat(1, 1, 1, 3)).squeeze(2).squeeze(1)
    return R, t, idx, scores


def compute_coarse_Rt(atten, pts1, pts2, model_pts, rand, n_proposal1=6000, n_proposal2=300, faithful=False,
                      return_aux=False):
    """PEM/utils/model_utils.py:204-275.  `rand` (B, 3*n_proposal1) are the hypothesis uniforms the reference
    draws from torch's global CPU generator (model_utils.py:292)."""
    weights, w1 = coarse_sampling_weights(atten)
    idx = weighted_sampling(weights, rand, faithful)
    Rs, ts, dis = coarse_hypotheses(idx, pts1, pts2, n_proposal1)
    R, t, top, scores = coarse_select(Rs, ts, dis, w1, pts1, model_pts, n_proposal2)
    if return_aux:
        return R, t, dict(weights=weights, w1=w1, idx=idx, dis=dis, top=top, scores=scores)
    return R, t


# ---------------------------------------------------------------------------------- fine pose (a11)
def compute_fine_Rt(atten, pts1, pts2, model_pts, dis_thres=0.15):
    """PEM/utils/model_utils.py:308-341."""
    S, l1, l2 = soft_assignment(atten)
    A = S[:, 1:, 1:] * (l1 > 0).float().unsqueeze(2) * (l2 > 0).float().unsqueeze(1)
    pred = (A / (A.sum(2, keepdim=True) + 1e-6)) @ pts2
    R, t = weighted_procrustes(pred, pts1, A.sum(2), weight_thresh=0.0)
    pp = (pts1 - t.unsqueeze(1)) @ R
    dis = torch.sqrt(pairwise_distance(pp.contiguous(), model_pts.contiguous())).min(2)[0]
    mask = (l1 > 0).float()
    score = ((dis < dis_thres).float() * mask).sum(1) / (mask.sum(1) + 1e-8)
    return R, t, score * mask.mean(1)


# ---------------------------------------------------------------------------- next rows
def radius_normalize(pts, dense_po):
    """PEM/model/feature_extraction.py:133-137 (ViTEncoder.forward): radius = max point norm of the template cloud."""
    radius = torch.norm(dense_po, dim=2).max(1)[0]
    return pts / (radius.reshape(-1, 1, 1) + 1e-6), dense_po / (radius.reshape(-1, 1, 1) + 1e-6), radius


def depth_to_cloud(depth, K, bbox=None):
    """PEM/utils/data_utils.py:92-110 (get_point_cloud_from_depth), fp32 throughout as under the numpy 1.x value-based
    casting the reference environment pins (K's float64 scalars do not promote the float32 maps)."""
    import numpy as np
    fx, fy, cx, cy = [np.float32(v) for v in (K[0][0], K[1][1], K[0][2], K[1][2])]
    H, W = depth.shape
    xmap = np.tile(np.arange(W, dtype=np.float32)[None, :], (H, 1))
    ymap = np.tile(np.arange(H, dtype=np.float32)[:, None], (1, W))
    if bbox is not None:
        rmin, rmax, cmin, cmax = bbox
        depth, xmap, ymap = depth[rmin:rmax, cmin:cmax], xmap[rmin:rmax, cmin:cmax], ymap[rmin:rmax, cmin:cmax]
    pt2 = depth.astype(np.float32)
    pt0 = (xmap - cx) * pt2 / fx
    pt1 = (ymap - cy) * pt2 / fy
    return np.stack([pt0, pt1, pt2]).transpose((1, 2, 0))


def get_bbox(label):
    """PEM/utils/data_utils.py:125-160."""
    import numpy as np
    H, Wd = label.shape
    rows = np.any(label, axis=1)
    cols = np.any(label, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    rmax += 1
    cmax += 1
    b = min(max(rmax - rmin, cmax - cmin), min(H, Wd))
    center = [int((rmin + rmax) / 2), int((cmin + cmax) / 2)]
    rmin, rmax = center[0] - int(b / 2), center[0] + int(b / 2)
    cmin, cmax = center[1] - int(b / 2), center[1] + int(b / 2)
    if rmin < 0:
        rmax, rmin = rmax - rmin, 0
    if cmin < 0:
        cmax, cmin = cmax - cmin, 0
    if rmax > H:
        rmin, rmax = rmin - (rmax - H), H
    if cmax > Wd:
        cmin, cmax = cmin - (cmax - Wd), Wd
    return [int(rmin), int(rmax), int(cmin), int(cmax)]


def get_resize_rgb_choose(choose, bbox, img_size):
    """PEM/utils/data_utils.py:113-123."""
    import numpy as np
    rmin, rmax, cmin, cmax = bbox
    crop_h, crop_w = rmax - rmin, cmax - cmin
    ratio_h, ratio_w = img_size / crop_h, img_size / crop_w
    row_idx, col_idx = choose // crop_w, choose % crop_w
    return (np.floor(row_idx * ratio_h) * img_size + np.floor(col_idx * ratio_w)).astype(np.int64)


def proposal_geometry(mask, depth, K, radius):
    """One proposal of get_test_data up to the radius filter (PEM/run_inference_custom_pytorch.py:316-337).  `radius * 1.2` is
    evaluated as under numpy 1.26, the version the reference pins: np.float32 scalar * Python float = float64, compared with the
    float32 norms at its float32 value.  Returns None where the reference `continue`s."""
    import numpy as np
    whole_pts = depth_to_cloud(depth, K)
    mask = np.logical_and(mask > 0, depth > 0)
    if np.sum(mask) <= 32:
        return None
    bbox = get_bbox(mask)
    y1, y2, x1, x2 = bbox
    m = mask[y1:y2, x1:x2]
    choose = m.astype(np.float32).flatten().nonzero()[0]
    cloud = whole_pts.copy()[y1:y2, x1:x2, :].reshape(-1, 3)[choose, :]
    center = np.mean(cloud, axis=0)
    tmp = cloud - center[None, :]
    thr = np.float32(np.float64(np.float32(radius)) * 1.2)
    flag = np.linalg.norm(tmp, axis=1) < thr
    if np.sum(flag) < 4:
        return None
    return dict(bbox=bbox, count=int(np.sum(mask)), choose=choose[flag], cloud=cloud[flag], center=center)


# -------------------------------------------------------------------------------- modules (a14)
def coarse_point_matching(p1, f1, g1, p2, f2, g2, radius, model, sd, rand, cfg=DEFAULT_CFG, faithful=False,
                          return_aux=False, p="coarse_point_matching"):
    """PEM/model/coarse_point_matching.py:32-63 (eval)."""
    B = f1.shape[0]
    bg = sd[p + ".bg_token"].repeat(B, 1, 1)
    f1 = torch.cat([bg, _lin(f1, sd, p + ".in_proj")], dim=1)
    f2 = torch.cat([bg, _lin(f2, sd, p + ".in_proj")], dim=1)
    for i in range(cfg["nblock"]):
        f1, f2 = geometric_transformer(f1, g1, f2, g2, sd, "%s.transformers.%d" % (p, i), cfg["num_heads"])
    atten = feature_similarity(_lin(f1, sd, p + ".out_proj"), _lin(f2, sd, p + ".out_proj"), cfg["temp"])
    mp = model / (radius.reshape(-1, 1, 1) + 1e-6)
    out = compute_coarse_Rt(atten, p1, p2, mp, rand, cfg["nproposal1"], cfg["nproposal2"], faithful, return_aux)
    if return_aux:
        out[2]["atten"] = atten
    return out


def fine_point_matching(p1, f1, g1, i1, p2, f2, g2, i2, radius, model, init_R, init_t, sd, cfg=DEFAULT_CFG,
                        return_aux=False, p="fine_point_matching"):
    """PEM/model/fine_point_matching.py:42-79 (eval)."""
    B = p1.shape[0]
    p1_ = (p1 - init_t.unsqueeze(1)) @ init_R
    pe = lambda x: positional_encoding(x, sd, p + ".PE", cfg["pe_radius1"], cfg["pe_radius2"],
                                       cfg["pe_nsample1"], cfg["pe_nsample2"])
    bg = sd[p + ".bg_token"].repeat(B, 1, 1)
    f1 = torch.cat([bg, _lin(f1, sd, p + ".in_proj") + pe(p1_)], dim=1)
    f2 = torch.cat([bg, _lin(f2, sd, p + ".in_proj") + pe(p2)], dim=1)
    for i in range(cfg["nblock"]):
        f1, f2 = sparse_to_dense_transformer(f1, g1, i1, f2, g2, i2, sd, "%s.transformers.%d" % (p, i),
                                             cfg["num_heads"], cfg["focusing_factor"])
    atten = feature_similarity(_lin(f1, sd, p + ".out_proj"), _lin(f2, sd, p + ".out_proj"), cfg["temp"])
    mp = model / (radius.reshape(-1, 1, 1) + 1e-6)
    R, t, score = compute_fine_Rt(atten, p1, p2, mp, cfg["dis_thres"])
    t = t * (radius.reshape(-1, 1) + 1e-6)
    if return_aux:
        return R, t, score, dict(atten=atten)
    return R, t, score


def pem_match(dense_pm, dense_fm, dense_po, dense_fo, radius, model, sd, rand, cfg=DEFAULT_CFG, faithful=False,
              return_aux=False):
    """PEM/model/pose_estimation_model.py:29-55 -- Net.forward after feature extraction (the seam SURVEY 8c
    names): FPS x2 -> geo-embedding x2 -> coarse -> fine -> (R, t, score)."""
    B = dense_pm.shape[0]
    bgp = torch.ones(B, 1, 3) * 100
    n = cfg["coarse_npoint"]
    spm, sfm, im = sample_pts_feats(dense_pm, dense_fm, n)
    gm = geo_embedding(torch.cat([bgp, spm], 1), sd, "geo_embedding", cfg["sigma_d"], cfg["sigma_a"], cfg["angle_k"])
    spo, sfo, io = sample_pts_feats(dense_po, dense_fo, n)
    go = geo_embedding(torch.cat([bgp, spo], 1), sd, "geo_embedding", cfg["sigma_d"], cfg["sigma_a"], cfg["angle_k"])
    c = coarse_point_matching(spm, sfm, gm, spo, sfo, go, radius, model, sd, rand, cfg, faithful, return_aux)
    R0, t0 = c[0], c[1]
    f = fine_point_matching(dense_pm, dense_fm, gm, im, dense_po, dense_fo, go, io, radius, model, R0, t0, sd, cfg,
                            return_aux)
    if return_aux:
        aux = dict(coarse=c[2], fine=f[3], init_R=R0, init_t=t0, fps_idx_m=im, fps_idx_o=io)
        return f[0], f[1], f[2], aux
    return f
