"""Drop-in for the scoring methods of ISM/model/detector.py::Instance_Segmentation_Model.

Constructor arguments, attribute names (`ref_data`, `matching_config`, `visible_thred`, ...) and method signatures are
the reference's.  The proposal generators and the DINOv2 descriptor model are outside the hot path: they are accepted
and stored, never touched here.  Lightning is optional (plain nn.Module when pytorch_lightning is absent)."""
import torch
import torch.nn as nn

from sam6d_hip import ism as _ism

try:  # the reference derives from pl.LightningModule (detector.py:25)
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # not installed here: a plain Module has everything the scoring path needs
    _Base = nn.Module


class RefAux:
    """Stand-in for the gathered template descriptors compute_appearance_score returns in the reference (detector.py:303): holds the
    reduced similarity of the fused launch; `.tensor()` builds the reference's (N,P,D) tensor on demand."""

    def __init__(self, ps):
        self.ps = ps

    def tensor(self):
        return self.ps.gathered_reference()

    @property
    def shape(self):
        return (self.ps.Ns, self.ps.P, self.ps.ref.shape[-1])


class Instance_Segmentation_Model(_Base):
    def __init__(self, segmentor_model, descriptor_model, onboarding_config, matching_config, post_processing_config,
                 log_interval, log_dir, visible_thred, pointcloud_sample_num, **kwargs):
        super().__init__()
        self.segmentor_model = segmentor_model
        self.descriptor_model = descriptor_model
        self.onboarding_config = onboarding_config
        self.matching_config = matching_config
        self.post_processing_config = post_processing_config
        self.log_interval, self.log_dir = log_interval, log_dir
        self.visible_thred = visible_thred
        self.pointcloud_sample_num = pointcloud_sample_num
        self.ref_data = {}

    # -- detector.py:198-207 ----------------------------------------------------------------------------------
    def best_template_pose(self, scores, pred_idx_objects):
        _, best_template_idxes = torch.max(scores, dim=-1)
        assert scores.shape[0] == pred_idx_objects.shape[0], "Prediction num != Query num"
        return torch.gather(best_template_idxes, 1, pred_idx_objects[:, None])[:, 0]

    # -- detector.py:260-296 ----------------------------------------------------------------------------------
    def compute_semantic_score(self, proposal_decriptors):
        scores = self.matching_config.metric(proposal_decriptors, self.ref_data["descriptors"])
        return _ism.semantic_select(scores, self.matching_config.aggregation_function, self.matching_config.confidence_thresh)

    # -- detector.py:298-308 ----------------------------------------------------------------------------------
    def compute_appearance_score(self, best_pose, pred_objects_idx, qurey_appe_descriptors):
        """-> (appe_scores, ref_aux_descriptor).  The reference gathers ref_data["appe_descriptors"][pred_objects_idx, best_pose] into a
        new (N,P,D) tensor, multiplies, and hands the gathered tensor back for compute_geometric_score to multiply AGAIN.  Here one
        launch reads the chosen templates in place and keeps the row / column maxima both scores need; the second return value is a
        RefAux handle (a stand-in for the gathered tensor: pass it on to compute_geometric_score as the reference's caller does,
        ISM/run_inference_custom.py:236-250; `.tensor()` materialises the reference's tensor for any other use)."""
        q = qurey_appe_descriptors.contiguous()
        ref_all = self.ref_data["appe_descriptors"]
        P, D = q.shape[1], q.shape[2]
        if ref_all.dim() == 4 and P % 128 == 0 and D % 32 == 0 and ref_all.dtype == torch.float32 and q.dtype == torch.float32:
            ps = _ism.patch_scores_fused(q, ref_all, pred_objects_idx, best_pose)
            return ps.scores()[0], RefAux(ps)
        ref = ref_all[pred_objects_idx, best_pose, ...].contiguous()  # (N, P, D) gather: shapes the fused kernel is not built for
        self._sim_cache = (_ism.patch_similarity(q, ref), q.data_ptr(), ref.data_ptr())
        return _ism.patch_scores(self._sim_cache[0], q)[0], ref

    # -- detector.py:209-246 ----------------------------------------------------------------------------------
    def project_template_to_image(self, best_pose, pred_object_idx, batch, proposals):
        vu, xyxy, _ = _ism.project_template_to_image(best_pose, pred_object_idx, self.ref_data["poses"],
                                                     self.ref_data["pointcloud"], proposals.squeeze_(), batch["depth"][0],
                                                     batch["cam_intrinsic"][0],
                                                     float(torch.as_tensor(batch["depth_scale"]).reshape(-1)[0]))
        self._xyxy_cache = (vu.data_ptr(), xyxy)  # the projected bounding box comes out of the same kernel
        return vu

    def Calculate_the_query_translation(self, proposal, depth, cam_intrinsic, depth_scale):
        N = proposal.shape[0]
        dev = proposal.device
        z = torch.zeros(N, dtype=torch.int64, device=dev)
        _, _, tr = _ism.project_template_to_image(z, z, torch.eye(4, device=dev).unsqueeze(0), torch.zeros(1, 1, 3, device=dev),
                                                  proposal, depth, cam_intrinsic,
                                                  float(torch.as_tensor(depth_scale).reshape(-1)[0]))
        return tr

    # -- detector.py:310-322 ----------------------------------------------------------------------------------
    def compute_geometric_score(self, image_uv, proposals, appe_descriptors, ref_aux_descriptor, visible_thred=0.5):
        q = appe_descriptors.contiguous()
        if isinstance(ref_aux_descriptor, RefAux) and ref_aux_descriptor.ps.q_ptr == q.data_ptr():
            visible_ratio = ref_aux_descriptor.ps.scores(visible_thred)[1]  # the maxima of compute_appearance_score's launch
        else:
            ref = (ref_aux_descriptor.tensor() if isinstance(ref_aux_descriptor, RefAux) else ref_aux_descriptor).contiguous()
            cache = getattr(self, "_sim_cache", None)
            if cache is not None and cache[1] == q.data_ptr() and cache[2] == ref.data_ptr():
                sim = cache[0]  # the reference recomputes the same matmul: reuse it
            else:
                sim = _ism.patch_similarity(q, ref)
            visible_ratio = _ism.patch_scores(sim, q, visible_thred)[1]
        bc = getattr(self, "_xyxy_cache", None)
        if bc is not None and bc[0] == image_uv.data_ptr():
            xyxy = bc[1]
        else:  # image_uv did not come from project_template_to_image: plain min/max (detector.py:316-318)
            xyxy = torch.cat((torch.min(image_uv, dim=1).values, torch.max(image_uv, dim=1).values), dim=-1)
        return _ism.compute_iou(xyxy, proposals.boxes), visible_ratio

    @staticmethod
    def final_score(semantic_score, appe_scores, geometric_score, visible_ratio):
        """detector.py:384 / run_inference_custom.py:255"""
        return _ism.final_score(semantic_score, appe_scores, geometric_score, visible_ratio)
