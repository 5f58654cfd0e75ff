"""Drop-in for PEM/model/fine_point_matching.py (inference)."""
import torch
import torch.nn as nn

from transformer import SparseToDenseTransformer, _Packed
from pointnet2_utils import QueryAndGroup
from pytorch_utils import SharedMLP, Conv1d
from coarse_point_matching import _cfg_dict
from sam6d_hip import pem as _pem


class PositionalEncoding(_Packed):
    """fine_point_matching.py:102-144.  forward(pts1 (B,N,3)) -> (B,N,out_dim)."""

    def __init__(self, out_dim, r1=0.1, r2=0.2, nsample1=32, nsample2=64, use_xyz=True, bn=True):
        super().__init__()
        if not (use_xyz and bn and out_dim == 256):
            raise ValueError("only use_xyz=True, bn=True, out_dim=256 is implemented")
        self.r1, self.r2, self.ns1, self.ns2 = r1, r2, nsample1, nsample2
        self.group1 = QueryAndGroup(r1, nsample1, use_xyz=use_xyz)
        self.group2 = QueryAndGroup(r2, nsample2, use_xyz=use_xyz)
        self.mlp1 = SharedMLP([6, 32, 64, 128], bn=bn)
        self.mlp2 = SharedMLP([6, 32, 64, 128], bn=bn)
        self.mlp3 = Conv1d(256, out_dim, 1, activation=None, bn=None)

    def forward(self, pts1, pts2=None):
        if pts2 is not None:
            raise NotImplementedError("PositionalEncoding is only called with one cloud (fine_point_matching.py:47,50)")
        if self.training:
            raise RuntimeError("inference only: BatchNorm runs in eval mode")
        W = self._packed(lambda sd, dev: _pem.pack_pe({"PE." + k: v for k, v in sd.items()}, dev, "PE"))
        B, N, _ = pts1.shape
        out = torch.zeros(B, N, 256, device=pts1.device)
        _pem.positional_encoding_add(pts1.contiguous(), W, out, 0, N * 256, self.r1, self.r2, self.ns1, self.ns2)
        return out


class FinePointMatching(_Packed):
    """fine_point_matching.py:16-79.  forward(p1, f1, geo1, fps_idx1, p2, f2, geo2, fps_idx2, radius, model, init_R,
    init_t) -> (pred_R, pred_t, pred_pose_score)."""

    def __init__(self, cfg, return_feat=False):
        super().__init__()
        self.cfg, self.return_feat, self.nblock = cfg, return_feat, cfg.nblock
        self.in_proj = nn.Linear(cfg.input_dim, cfg.hidden_dim)
        self.out_proj = nn.Linear(cfg.hidden_dim, cfg.out_dim)
        self.bg_token = nn.Parameter(torch.randn(1, 1, cfg.hidden_dim) * .02)
        self.PE = PositionalEncoding(cfg.hidden_dim, r1=cfg.pe_radius1, r2=cfg.pe_radius2)
        self.transformers = nn.ModuleList([
            SparseToDenseTransformer(cfg.hidden_dim, num_heads=4, sparse_blocks=['self', 'cross'], dropout=None,
                                     activation_fn='ReLU', focusing_factor=cfg.focusing_factor, with_bg_token=True,
                                     replace_bg_token=True) for _ in range(self.nblock)])

    def _build(self, sd, dev):
        sd = {"fine_point_matching." + k: v for k, v in sd.items()}
        g = _pem._getter(sd, dev)
        W = _pem.pack_pe(sd, dev, "fine_point_matching.PE")
        W.fine = _pem.PemWeights._matching(g, "fine_point_matching")
        W.fine["blocks"] = [_pem.pack_sparse_to_dense(sd, dev, "fine_point_matching.transformers.%d" % i)
                            for i in range(self.nblock)]
        return W

    def forward(self, p1, f1, geo1, fps_idx1, p2, f2, geo2, fps_idx2, radius, model, init_R, init_t):
        if self.training:
            raise RuntimeError("inference only: call .eval()")
        W = self._packed(self._build)
        dp = torch.cat([p1, p2], 0).contiguous()
        df = torch.cat([f1, f2], 0).contiguous()
        E = torch.cat([geo1, geo2], 0).contiguous()
        idx = torch.cat([fps_idx1, fps_idx2], 0).contiguous()
        return _pem.fine_point_matching(dp, df, E, idx, radius.reshape(-1).contiguous(), model.contiguous(),
                                        init_R.contiguous(), init_t.contiguous(), W, _cfg_dict(self.cfg))
