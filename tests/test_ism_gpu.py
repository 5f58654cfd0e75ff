"""GPU parity of the ISM template-scoring path through the drop-in Instance_Segmentation_Model /
model.loss call sites, against the reference's outputs (tests/golden/ism.npz) and the CPU oracle."""
import importlib
import sys

import numpy as np
import pytest
import torch

from tests._util import golden, PKG
from tests.test_oracle_golden import ism_inputs, ism_masks

pytestmark = pytest.mark.gpu


class _Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.fixture(scope="module")
def ism_model(dev):
    import os
    p = os.path.join(PKG, "ism")
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.") or k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    sys.path.insert(0, p)
    loss = importlib.import_module("model.loss")
    det = importlib.import_module("model.detector")
    m = det.Instance_Segmentation_Model(segmentor_model=None, descriptor_model=None, onboarding_config=None,
                                        matching_config=_Cfg(metric=loss.PairwiseSimilarity("cosine", 16),
                                                             aggregation_function="avg_5", confidence_thresh=0.2),
                                        post_processing_config=None, log_interval=5, log_dir=".", visible_thred=0.5,
                                        pointcloud_sample_num=2048)
    return m, loss


def _close(got, want, atol, what):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    d = np.abs(got.astype(np.float64) - np.asarray(want, dtype=np.float64)).max()
    assert d <= atol, "%s: max abs diff %.3e > %.1e" % (what, d, atol)


def test_ism_pipeline_golden(dev, ism_model):
    from oracle import ism_oracle as IO
    m, loss = ism_model
    g = golden("ism")
    d = ism_inputs(int(g["seed"]))
    q, ref = d["q"].to(dev), d["ref"].to(dev)
    m.ref_data = {"descriptors": ref, "appe_descriptors": d["r_appe"].to(dev), "poses": d["poses"].to(dev),
                  "pointcloud": d["pc"].to(dev)}
    sim = loss.PairwiseSimilarity("cosine", 16)(q, ref)
    _close(sim[:8], g["sim_rows"], 2e-6, "pairwise similarity")
    sel, obj, sem, best = m.compute_semantic_score(q)
    assert np.array_equal(sel.cpu().numpy().astype(np.int32), g["sel"]), "selected proposals"
    assert np.array_equal(obj.cpu().numpy().astype(np.int32), g["obj"])
    assert np.array_equal(best.cpu().numpy().astype(np.int32), g["best"]), "best template per proposal"
    _close(sem, g["sem"], 2e-6, "semantic score")
    qa = d["q_appe"].to(dev)[sel]
    appe, ref_sel = m.compute_appearance_score(best, obj, qa)
    _close(appe, g["appe"], 5e-6, "appearance score")
    masks, boxes = ism_masks(d["gen"], len(sel))
    assert np.array_equal(boxes.numpy().astype(np.int32), g["boxes"])
    # dtypes as the reference's caller hands them over (ISM/run_inference_custom.py:86-96): K and depth_scale float64, depth int32
    assert d["K"].dtype == torch.float64 and d["depth_scale"].dtype == torch.float64 and d["depth"].dtype == torch.int32
    batch = {"depth": d["depth"][None].to(dev), "cam_intrinsic": d["K"][None].to(dev), "depth_scale": d["depth_scale"].to(dev)}
    tr = m.Calculate_the_query_translation(masks.to(dev), batch["depth"][0], batch["cam_intrinsic"][0], batch["depth_scale"])
    assert tr.dtype == torch.float32
    assert np.array_equal(tr.cpu().numpy(), g["translate"]), "query translation: float64 sums rounded once to float32, bit-exact"
    vu = m.project_template_to_image(best, obj, batch, masks.to(dev))
    # integer work: bit-exact against the reference's own image_vu (north_star: index/integer outputs bit-exact)
    want_vu = g["vu"].astype(np.int32)
    nbad = int((vu.cpu().numpy() != want_vu).sum())
    assert nbad == 0, "image_vu: %d of %d coordinates differ from the reference" % (nbad, want_vu.size)

    class Det:
        pass
    dets = Det(); dets.boxes = boxes.to(dev)
    iou, vis = m.compute_geometric_score(vu, dets, qa, ref_sel, visible_thred=0.5)
    _close(vis, g["vis"], 2e-6, "visible ratio")
    assert torch.is_tensor(iou)
    _close(iou, g["iou"], 1e-6, "IoU")
    fin = m.final_score(sem, appe, iou, vis)
    _close(fin, g["final"], 5e-6, "final score")
    # scalar-0.0 quirk of compute_iou (bbox_utils.py:214-220)
    bq = boxes.clone(); bq[3] = torch.tensor([0, 0, 2, 2]); dets.boxes = bq.to(dev)
    iou_q, _ = m.compute_geometric_score(vu, dets, qa, ref_sel, visible_thred=0.5)
    assert isinstance(iou_q, float) and iou_q == 0.0
    fin_q = m.final_score(sem, appe, iou_q, vis)
    want_q = (sem.cpu() + appe.cpu()) / (2 + vis.cpu())
    _close(fin_q, want_q, 1e-6, "final score with the 0.0 IoU quirk")


def test_fused_patch_scores_equal_materialised_path(dev, ism_model):
    """compute_appearance_score / compute_geometric_score through the fused launch (templates read in place, row / column maxima kept,
    no (N,256,256) similarity tensor) against the materialised path (gather + GEMM + reduction) and the oracle; the RefAux handle the
    drop-in returns in place of the gathered tensor materialises to exactly the reference's gather; q_index reads the query patches
    in place as well."""
    from oracle import ism_oracle as IO
    from sam6d_hip import ism
    m, loss = ism_model
    g = golden("ism")
    d = ism_inputs(int(g["seed"]))
    r_appe = d["r_appe"].to(dev)
    m.ref_data = {"descriptors": d["ref"].to(dev), "appe_descriptors": r_appe, "poses": d["poses"].to(dev), "pointcloud": d["pc"].to(dev)}
    sel, obj, sem, best = m.compute_semantic_score(d["q"].to(dev))
    q_all = d["q_appe"].to(dev)
    qa = q_all[sel].contiguous()
    appe, aux = m.compute_appearance_score(best, obj, qa)
    assert type(aux).__name__ == "RefAux", "the fused path must be what the drop-in runs at these shapes"
    ref_sel = r_appe[obj, best].contiguous()
    assert torch.equal(aux.tensor(), ref_sel)
    sim = ism.patch_similarity(qa, ref_sel)
    a2, v2 = ism.patch_scores(sim, qa, 0.5)
    ps = ism.patch_scores_fused(q_all, r_appe, obj, best, q_index=sel)
    a3, v3 = ps.scores(0.5)
    assert torch.equal(a3, appe), "q_index path differs from the gathered-query path"
    _close(appe, a2.cpu(), 2e-6, "appearance: fused vs materialised")
    _close(v3, v2.cpu(), 1e-6, "visible ratio: fused vs materialised")
    _close(appe, g["appe"], 5e-6, "appearance vs the reference")
    _close(v3, g["vis"], 2e-6, "visible ratio vs the reference")
    for thr in (0.3, 0.7):
        _close(ps.scores(thr)[1], ism.patch_scores(sim, qa, thr)[1].cpu(), 1e-6, "visible ratio at threshold %g" % thr)


def test_translation_matches_fp64_reference_path(dev, ism_model):
    """K / depth_scale arrive as float64 from the reference's caller (run_inference_custom.py:87-94): the whole
    translation is then float64 in the reference; the kernel must agree to fp32 rounding."""
    from oracle import ism_oracle as IO
    m, _ = ism_model
    d = ism_inputs(0)
    gen = torch.Generator().manual_seed(77)
    masks, _ = ism_masks(gen, 12)
    K64 = d["K"].double()
    want = IO.query_translation(masks, d["depth"], K64, d["depth_scale"])
    got = m.Calculate_the_query_translation(masks.to(dev), d["depth"].to(dev), K64.to(dev), d["depth_scale"].to(dev))
    assert torch.equal(got.cpu(), want), "query translation (float64 inside, rounded once): max diff %g" % (got.cpu() - want).abs().max()


@pytest.mark.parametrize("mode", ["mean", "max", "avg_5"])
def test_semantic_aggregations_vs_oracle(dev, mode):
    from oracle import ism_oracle as IO
    from sam6d_hip import ism
    gen = torch.Generator().manual_seed(5)
    q = torch.randn(64, 256, generator=gen)
    base = torch.randn(3, 1, 256, generator=gen)
    ref = base + 0.7 * torch.randn(3, 42, 256, generator=gen)
    q[:40] = base[torch.randint(0, 3, (40,), generator=gen), 0] + 0.8 * q[:40]
    sel, obj, sem, best = IO.semantic_score(q, ref, mode, 0.2)
    s = ism.pairwise_similarity(q.to(dev), ref.to(dev))
    gsel, gobj, gsem, gbest = ism.semantic_select(s, mode, 0.2)
    assert torch.equal(gsel.cpu(), sel) and torch.equal(gobj.cpu(), obj) and torch.equal(gbest.cpu(), best)
    _close(gsem, sem, 2e-6, "semantic score " + mode)


def test_empty_selection(dev):
    from sam6d_hip import ism
    s = torch.zeros(10, 1, 42, device=dev)
    sel, obj, sem, best = ism.semantic_select(s, "avg_5", 0.2)
    assert sel.numel() == 0 and obj.numel() == 0 and sem.numel() == 0 and best.numel() == 0


def test_trimesh_utils_dropin_translation_one_launch(dev, ism_model):
    """The stand-alone drop-in utils.trimesh_utils.depth_image_to_pointcloud_translate_torch (ISM/utils/trimesh_utils.py:77-105) on ALL
    masked depth maps in one launch (round 3 looped over the proposals in Python and built an (N,H,W) ones tensor): equal to the
    reference's captured translation bit for bit, with the reference caller's dtypes (mask * int32 depth, float64 K / depth_scale)."""
    tu = importlib.import_module("utils.trimesh_utils")
    g = golden("ism")
    d = ism_inputs(int(g["seed"]))
    masks, _ = ism_masks(d["gen"], len(g["sel"]))
    masked = masks.to(dev) * d["depth"][None].to(dev)  # detector.py:243
    tr = tu.depth_image_to_pointcloud_translate_torch(masked, d["depth_scale"].to(dev), d["K"].to(dev))
    assert tr.shape == (len(g["sel"]), 3)
    assert np.array_equal(tr.to(torch.float32).cpu().numpy(), g["translate"])
    # an empty mask: the reference's sum / (0 + 1e-8) = 0
    z = tu.depth_image_to_pointcloud_translate_torch(torch.zeros(2, 48, 64, device=dev), 1.0, d["K"].to(dev))
    assert float(z.abs().max()) == 0.0


@pytest.mark.parametrize("quirk", [False, True])
def test_ism_proposals_sharded_equals_unsharded_on_one_gpu(dev, quirk):
    """SURVEY 8e for the ISM leg through the HIP kernels: the 200 proposals dealt round-robin to 8 shards (what 8 ranks would score),
    their 12-float records concatenated rank-major (= the all-gather's result) and merged with the device NMS -- the same detections
    (proposal ids, object ids, boxes, final scores bit for bit) as one unsharded pass; `quirk`: one proposal with a non-positive
    overlap zeroes the geometric term image-wide, also for proposals of other shards."""
    from sam6d_hip import ism, parallel
    d = ism_inputs(0)
    masks, _ = ism_masks(d["gen"], 200)
    boxes = torch.tensor([0, 0, 640, 480]).repeat(200, 1)
    if quirk:
        boxes[7] = torch.tensor([0, 0, 1, 1])
    g = {k: (v.to(dev).contiguous() if torch.is_tensor(v) else v) for k, v in d.items()}
    masks, boxes = masks.to(dev), boxes.to(dev)

    def score_fn(ids):
        sim = ism.pairwise_similarity(g["q"][ids].contiguous(), g["ref"])
        sel, obj, sem, best = ism.semantic_select(sim, "avg_5", 0.2)
        appe, vis = ism.patch_scores_fused(g["q_appe"], g["r_appe"], obj, best, q_index=ids[sel]).scores(0.5)
        bx = boxes[ids][sel]
        vu, xyxy, tr = ism.project_template_to_image(best, obj, g["poses"], g["pc"], masks[ids][sel], g["depth"], g["K"], g["depth_scale"])
        iou = ism.compute_iou(xyxy, bx)
        return dict(sel=sel, sem=sem, appe=appe, iou=iou, vis=vis, all_positive=torch.is_tensor(iou), object_ids=obj, boxes=bx)

    final_fn = lambda sem, appe, geo, vis: ism.final_score(sem, appe, geo if geo is not None else 0.0, vis)
    nms_fn = lambda b, s, th, o: ism.nms(b, s, th, object_ids=o)
    # unsharded
    r = score_fn(torch.arange(200, device=dev))
    assert r["all_positive"] == (not quirk)
    fin = ism.final_score(r["sem"], r["appe"], r["iou"], r["vis"])
    # the same with the quirk decided on the device (no host read-back between IoU and final score)
    sim_all = ism.pairwise_similarity(g["q"], g["ref"])
    sel_a, obj_a, sem_a, best_a = ism.semantic_select(sim_all, "avg_5", 0.2)
    _, xyxy_a, _ = ism.project_template_to_image(best_a, obj_a, g["poses"], g["pc"], masks, g["depth"], g["K"], g["depth_scale"], mask_index=sel_a)
    iou_d, flag = ism.compute_iou(xyxy_a, boxes[sel_a], return_flag=True)
    assert int(flag.item()) == (0 if quirk else 1)
    assert torch.equal(ism.final_score(r["sem"], r["appe"], iou_d, r["vis"], all_positive=flag), fin)
    keep = ism.nms(r["boxes"].float(), fin, 0.25, object_ids=r["object_ids"])
    want = dict(scores=fin[keep], object_ids=r["object_ids"][keep], boxes=r["boxes"][keep], proposal_ids=r["sel"][keep])
    assert 0 < len(keep) < len(r["sel"]) < 200
    # 8 shards, one after the other on this GPU
    world, recs = 8, []
    for rank in range(world):
        ids, nv = parallel.shard_indices(200, rank, world)
        ids = ids[:nv].to(dev)
        s = score_fn(ids)
        recs.append(parallel.pack_detections(s["sem"], s["appe"], s["iou"], s["vis"], s["all_positive"], s["object_ids"], s["boxes"],
                                             ids[s["sel"]], 25))
    got = parallel.merge_detections(torch.cat(recs, 0), final_fn, nms_fn, 0.25)
    for k in ("proposal_ids", "object_ids", "boxes", "scores"):
        assert torch.equal(got[k].cpu(), want[k].cpu()), "%s differs between the sharded and the unsharded ISM pass" % k


def test_project_uint8_masks_read_in_place_through_the_selection(dev):
    """sam6d_ism_project2: uint8 / bool masks (SAM's binary proposals, one byte per pixel) read in place through the selection index --
    same image_vu / box / translation as the float32 masks gathered first (the reference's Detections.filter + mask * depth,
    ISM/model/detector.py:243), equal to the reference's capture; also the general fallback (W not a multiple of 16)."""
    from sam6d_hip import ism
    g = golden("ism")
    d = ism_inputs(int(g["seed"]))
    masks, _ = ism_masks(d["gen"], len(g["sel"]))
    sel = torch.as_tensor(g["sel"]).long()
    best, obj = torch.as_tensor(g["best"]).long().to(dev), torch.as_tensor(g["obj"]).long().to(dev)
    # scatter the 150 selected proposals' masks into a (200, H, W) uint8 tensor at their original positions
    all_u8 = torch.zeros(200, 480, 640, dtype=torch.uint8)
    all_u8[sel] = (masks > 0).to(torch.uint8)
    args = (d["poses"].to(dev), d["pc"].to(dev))
    K, ds, depth = d["K"].to(dev), d["depth_scale"].to(dev), d["depth"].to(dev)
    vu_f, xy_f, tr_f = ism.project_template_to_image(best, obj, *args, masks.to(dev), depth, K, ds)
    vu_u, xy_u, tr_u = ism.project_template_to_image(best, obj, *args, all_u8.to(dev), depth, K, ds, mask_index=sel.to(dev))
    vu_b, xy_b, tr_b = ism.project_template_to_image(best, obj, *args, all_u8.to(dev).bool(), depth, K, ds, mask_index=sel.to(dev))
    for a, b in ((vu_f, vu_u), (xy_f, xy_u), (tr_f, tr_u), (vu_f, vu_b), (tr_f, tr_b)):
        assert torch.equal(a, b)
    assert np.array_equal(tr_u.cpu().numpy(), g["translate"]) and np.array_equal(vu_u.cpu().numpy(), g["vu"].astype(np.int32))
    assert np.array_equal(xy_u.cpu().numpy(), g["xyxy"])
    # general shapes (W % 16 != 0) take the scalar kernel; compare with the oracle
    from oracle import ism_oracle as IO
    gen = torch.Generator().manual_seed(3)
    m2 = (torch.rand(5, 50, 70, generator=gen) > 0.6).float()
    dp = (700 + 300 * torch.rand(50, 70, generator=gen)).to(torch.int32)
    b2, o2 = torch.tensor([3, 1, 0, 7, 2]), torch.zeros(5, dtype=torch.long)
    want = IO.project_template_to_image(b2, o2, d["poses"], d["pc"], m2, dp, d["K"], d["depth_scale"])
    got, _, _ = ism.project_template_to_image(b2.to(dev), o2.to(dev), *args, m2.to(dev).bool(), dp.to(dev), K, ds)
    assert np.array_equal(got.cpu().numpy(), want.numpy().astype(np.int32))
