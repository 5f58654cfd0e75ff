"""Drop-in for PEM/model/pose_estimation_model.py: `importlib.import_module("pose_estimation_model").Net(cfg.model)`
(PEM/run_inference_custom_pytorch.py:383-386) builds this class; same forward signature, same state_dict keys."""
import torch
import torch.nn as nn

from feature_extraction import ViTEncoder
from coarse_point_matching import CoarsePointMatching
from fine_point_matching import FinePointMatching
from transformer import GeometricStructureEmbedding
from model_utils import sample_pts_feats


class Net(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.coarse_npoint = cfg.coarse_npoint
        self.fine_npoint = cfg.fine_npoint
        self.feature_extraction = ViTEncoder(cfg.feature_extraction, self.fine_npoint)
        self.geo_embedding = GeometricStructureEmbedding(cfg.geo_embedding)
        self.coarse_point_matching = CoarsePointMatching(cfg.coarse_point_matching)
        self.fine_point_matching = FinePointMatching(cfg.fine_point_matching)

    def forward(self, pts, rgb, rgb_choose, model, dense_po, dense_fo):
        """pose_estimation_model.py:25-56."""
        dense_pm, dense_fm, dense_po, dense_fo, radius = self.feature_extraction(pts, rgb, rgb_choose, dense_po, dense_fo)
        return self.match(dense_pm, dense_fm, dense_po, dense_fo, radius, model)

    def match(self, dense_pm, dense_fm, dense_po, dense_fo, radius, model):
        """The post-feature-extraction seam (pose_estimation_model.py:29-55): the hot path proper."""
        bg_point = torch.ones(dense_pm.size(0), 1, 3, device=dense_pm.device) * 100
        sparse_pm, sparse_fm, fps_idx_m = sample_pts_feats(dense_pm, dense_fm, self.coarse_npoint, return_index=True)
        geo_m = self.geo_embedding(torch.cat([bg_point, sparse_pm], dim=1))
        sparse_po, sparse_fo, fps_idx_o = sample_pts_feats(dense_po, dense_fo, self.coarse_npoint, return_index=True)
        geo_o = self.geo_embedding(torch.cat([bg_point, sparse_po], dim=1))
        init_R, init_t = self.coarse_point_matching(sparse_pm, sparse_fm, geo_m, sparse_po, sparse_fo, geo_o, radius, model)
        return self.fine_point_matching(dense_pm, dense_fm, geo_m, fps_idx_m, dense_po, dense_fo, geo_o, fps_idx_o,
                                        radius, model, init_R, init_t)
