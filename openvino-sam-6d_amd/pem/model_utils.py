"""Drop-in for the hot-path functions of PEM/utils/model_utils.py (same names and argument meaning), running on the
HIP kernels.  Two deliberate, documented differences of the host interface:
  * compute_coarse_Rt takes the hypothesis uniforms explicitly (`rand`, (B, 3*n_proposal1)); when omitted they are
    drawn with torch.rand on the device like the reference does (model_utils.py:292).
  * `model_pts` is taken exactly as the reference takes it (already divided by radius + 1e-6 by the caller,
    coarse_point_matching.py:60, fine_point_matching.py:76); no further scaling is applied here.
"""
import torch
import torch.nn as nn

from pointnet2_utils import furthest_point_sample, gather_operation
from sam6d_hip import pem as _pem


def sample_pts_feats(pts, feats, npoint=2048, return_index=False):
    """model_utils.py:70-84."""
    sp, sf, idx = _pem.sample_pts_feats(pts.contiguous(), feats.contiguous(), npoint)
    return (sp, sf, idx) if return_index else (sp, sf)


def pairwise_distance(x, y, normalized=False, channel_first=False):
    """model_utils.py:101-128 for 3-d points (the only use on the path): squared distances with the bits of the reference's CPU
    evaluation (sam6d_pairwise_distance)."""
    if normalized:
        raise NotImplementedError("pairwise_distance: the normalized form is not used by the inference path")
    if channel_first:
        x, y = x.transpose(-1, -2), y.transpose(-1, -2)
    if x.shape[-1] != 3 or y.shape[-1] != 3 or x.shape[:-2] != y.shape[:-2]:
        raise NotImplementedError("pairwise_distance: (*, N, 3) x (*, M, 3) point clouds with equal batch dimensions")
    lead = x.shape[:-2]
    return _pem.pairwise_distance(x.reshape(-1, x.shape[-2], 3).contiguous(), y.reshape(-1, y.shape[-2], 3).contiguous()).reshape(
        *lead, x.shape[-2], y.shape[-2])


def compute_feature_similarity(feat1, feat2, type='cosine', temp=1.0, normalize_feat=True):
    """model_utils.py:131-153 (cosine with normalisation, the configuration PEM uses)."""
    if type != 'cosine' or not normalize_feat or feat1.shape[1] != feat2.shape[1]:
        raise NotImplementedError("only the cosine / normalize_feat=True / N==M form used by PEM is implemented")
    B, n, Cc = feat1.shape
    ident = _pem.Linear(torch.eye(Cc, device=feat1.device), torch.zeros(Cc, device=feat1.device))
    return _pem.feature_similarity(torch.cat([feat1, feat2], 0).contiguous(), B, n, ident, temp)


def _unit_radius(B, dev):
    # the kernels divide model points by (radius + 1e-6); radius = 1 - 1e-6 rounds that divisor to exactly 1.0f
    return torch.full((B,), 1.0 - 1e-6, device=dev)


def compute_coarse_Rt(atten, pts1, pts2, model_pts=None, n_proposal1=6000, n_proposal2=300, rand=None):
    """model_utils.py:204-275."""
    B = atten.shape[0]
    if model_pts is None:
        model_pts = pts2
    if rand is None:
        rand = torch.rand(B, n_proposal1 * 3, device=atten.device)
    return _pem.compute_coarse_Rt(atten.contiguous(), pts1.contiguous(), pts2.contiguous(), model_pts.contiguous(),
                                  _unit_radius(B, atten.device), rand.contiguous(), n_proposal1, n_proposal2)


def compute_fine_Rt(atten, pts1, pts2, model_pts=None, dis_thres=0.15):
    """model_utils.py:308-341 (translation NOT rescaled here: radius = 1)."""
    B = atten.shape[0]
    if model_pts is None:
        model_pts = pts2
    R, t, s = _pem.compute_fine_Rt(atten.contiguous(), pts1.contiguous(), pts2.contiguous(), model_pts.contiguous(),
                                   _unit_radius(B, atten.device), dis_thres)
    return R, t, s


def weighted_procrustes(src_points, ref_points, weights=None, weight_thresh=0.0, eps=1e-5, return_transform=False,
                        src_centroid=None, ref_centroid=None):
    """model_utils.py:343-436."""
    if src_centroid is not None or ref_centroid is not None:
        raise NotImplementedError("explicit centroids are not used by the inference path")
    squeeze = src_points.ndim == 2
    if squeeze:
        src_points, ref_points = src_points.unsqueeze(0), ref_points.unsqueeze(0)
        weights = weights.unsqueeze(0) if weights is not None else None
    R, t = _pem.weighted_procrustes(src_points.contiguous(), ref_points.contiguous(),
                                    weights.contiguous() if weights is not None else None, weight_thresh, eps)
    if return_transform:
        T = torch.eye(4, device=R.device).unsqueeze(0).repeat(R.shape[0], 1, 1)
        T[:, :3, :3] = R
        T[:, :3, 3] = t
        return T.squeeze(0) if squeeze else T
    return (R.squeeze(0), t.squeeze(0)) if squeeze else (R, t)


class WeightedProcrustes(nn.Module):
    def __init__(self, weight_thresh=0.5, eps=1e-5, return_transform=False):
        super().__init__()
        self.weight_thresh, self.eps, self.return_transform = weight_thresh, eps, return_transform

    def forward(self, src_points, tgt_points, weights=None, src_centroid=None, ref_centroid=None):
        return weighted_procrustes(src_points, tgt_points, weights, self.weight_thresh, self.eps, self.return_transform,
                                   src_centroid, ref_centroid)


def get_chosen_pixel_feats(img, choose):
    """model_utils.py:86-98 (feature-extraction side, plain torch: outside the matching path)."""
    B, Cc, H, W = img.size()
    img = img.reshape(B, Cc, H * W)
    choose = choose.unsqueeze(1).repeat(1, Cc, 1)
    return torch.gather(img, 2, choose).transpose(1, 2).contiguous()
