"""Drop-in for PEM/model/pointnet2/pointnet2_utils.py (inference subset): same callable names and argument order,
backed by the HIP kernels through `pointnet2._ext`.  Inference only -- nothing here is differentiable."""
import torch
import torch.nn as nn

import pointnet2._ext as _ext


def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3); `npoint` is a tensor whose LENGTH is the sample count (pointnet2_utils.py:59-80) or an int."""
    n = int(npoint.shape[0]) if torch.is_tensor(npoint) else int(npoint)
    return _ext.furthest_point_sampling(xyz.contiguous(), n)


def gather_operation(features, idx):
    """features (B,C,N), idx (B,npoint) i32 -> (B,C,npoint)   (pointnet2_utils.py:94-118)."""
    return _ext.gather_points(features.contiguous(), idx.contiguous())


def grouping_operation(features, idx):
    """features (B,C,N), idx (B,npoint,nsample) i32 -> (B,C,npoint,nsample)   (pointnet2_utils.py:236-262)."""
    return _ext.group_points(features.contiguous(), idx.contiguous())


def ball_query(new_xyz, xyz, radius, nsample):
    """note the argument order: new_xyz first (pointnet2_utils.py:289-293)."""
    return _ext.ball_query(new_xyz.contiguous(), xyz.contiguous(), radius, nsample)


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:305-403 with the options PEM uses (use_xyz=True, no normalisation / uniform sampling)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False, sample_uniformly=False,
                 ret_unique_cnt=False):
        super().__init__()
        if normalize_xyz or sample_uniformly or ret_unique_cnt:
            raise NotImplementedError("QueryAndGroup: only the options PositionalEncoding uses are implemented")
        self.radius, self.nsample, self.use_xyz, self.ret_grouped_xyz = radius, nsample, use_xyz, ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(new_xyz, xyz, self.radius, self.nsample)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            g = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, g], dim=1) if self.use_xyz else g
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features
