"""Drop-in for PEM/model/pose_estimation_model.py: `importlib.import_module("pose_estimation_model").Net(cfg.model)`
(PEM/run_inference_custom_pytorch.py:383-386) builds this class; same forward signature, same state_dict keys."""
import torch
import torch.nn as nn

from feature_extraction import ViTEncoder
from coarse_point_matching import CoarsePointMatching
from fine_point_matching import FinePointMatching
from transformer import GeometricStructureEmbedding, _sig
from model_utils import sample_pts_feats
from sam6d_hip import pem as _pem
from coarse_point_matching import _cfg_dict


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

    # fused = True: the whole seam runs as one libsam6d_hip pipeline (pem.pem_match: scene / template clouds stacked, RPE
    # attention without the (B,197,197,256) embedding tensors).  False: module by module through the reference's own call
    # graph below (geo embeddings materialised and handed from module to module), same results to ~1e-6.
    fused = True
    # None: detect per call whether all proposals share one template cloud; True / False: the caller's promise
    shared_template = None

    def _whole_weights(self):
        sig = _sig(self)
        if getattr(self, "_pack_sig", None) != sig:
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith("feature_extraction.")}
            dev = self.coarse_point_matching.in_proj.weight.device
            if dev.type != "cuda":
                raise RuntimeError("Net: parameters must live on a HIP device (model.to('cuda')); there is no CPU path")
            object.__setattr__(self, "_pack", _pem.PemWeights(sd, dev, nblock=self.coarse_point_matching.nblock))
            object.__setattr__(self, "_pack_sig", sig)
        return self._pack

    def match(self, dense_pm, dense_fm, dense_po, dense_fo, radius, model, template_ids=None):
        """The post-feature-extraction seam (pose_estimation_model.py:29-55): the hot path proper.
        template_ids (B,): a multi-object batch -- dense_po / dense_fo hold the T unique templates (the `all_dense_po[obj]` form of the
        BOP provider, PEM/provider/bop_test_dataset.py:107,156, before it is indexed per instance) and proposal b uses template
        template_ids[b]; bit-identical to the per-instance repeated form."""
        if self.fused and self.coarse_npoint == 196:
            if self.training:
                raise RuntimeError("inference only: call .eval() (the reference fork is inference-only too, README.md:78-84)")
            W = self._whole_weights()
            rand = self.coarse_point_matching.hypothesis_rand
            if rand is None:
                rand = torch.rand(dense_pm.shape[0], self.coarse_point_matching.cfg.nproposal1 * 3, device=dense_pm.device)
            cfg = _cfg_dict(self.coarse_point_matching.cfg, _cfg_dict(self.fine_point_matching.cfg))
            cfg.update(coarse_npoint=self.coarse_npoint, sigma_d=self.geo_embedding.sigma_d, sigma_a=self.geo_embedding.sigma_a,
                       angle_k=self.geo_embedding.angle_k)
            # the reference's caller repeats one object's template tensors per instance (run_inference_custom_pytorch.py:445-446):
            # recognised here (bitwise comparison on the device), their pose-independent work then runs once
            if template_ids is not None:
                return _pem.pem_match(dense_pm.contiguous(), dense_fm.contiguous(), dense_po.contiguous(), dense_fo.contiguous(),
                                      radius.reshape(-1).contiguous(), model.contiguous(), W, rand.contiguous(), cfg=cfg,
                                      template_ids=template_ids)
            shared = _pem.template_is_shared(dense_po, dense_fo) if self.shared_template is None else bool(self.shared_template)
            return _pem.pem_match(dense_pm.contiguous(), dense_fm.contiguous(), dense_po.contiguous(), dense_fo.contiguous(),
                                  radius.reshape(-1).contiguous(), model.contiguous(), W, rand.contiguous(), cfg=cfg,
                                  shared_template=shared)
        if template_ids is not None:  # module-by-module path: the repeated form
            ids = template_ids.to(dense_po.device).long()
            dense_po, dense_fo = dense_po[ids], dense_fo[ids]
        bg_point = torch.ones(dense_pm.size(0), 1, 3, device=dense_pm.device) * 100
        sparse_pm, sparse_fm, fps_idx_m = sample_pts_feats(dense_pm, dense_fm, self.coarse_npoint, return_index=True)
        geo_m = self.geo_embedding(torch.cat([bg_point, sparse_pm], dim=1))
        sparse_po, sparse_fo, fps_idx_o = sample_pts_feats(dense_po, dense_fo, self.coarse_npoint, return_index=True)
        geo_o = self.geo_embedding(torch.cat([bg_point, sparse_po], dim=1))
        init_R, init_t = self.coarse_point_matching(sparse_pm, sparse_fm, geo_m, sparse_po, sparse_fo, geo_o, radius, model)
        return self.fine_point_matching(dense_pm, dense_fm, geo_m, fps_idx_m, dense_po, dense_fo, geo_o, fps_idx_o,
                                        radius, model, init_R, init_t)
