"""Pins the CPU oracle (oracle/) to golden vectors captured from the REFERENCE ITSELF (oracle/gen_golden.py imported
the reference's Python modules and its own compiled `_ext` in the build container).  CPU-only, no GPU, and nothing
here reads /root/reference.  Integer/index outputs are bit-exact; float outputs that go through torch ops whose
kernel choice can vary with the host CPU (GEMMs, softmax, sin/cos) use the tolerance stated per test."""
import hashlib

import numpy as np
import pytest
import torch

from tests._util import golden
from oracle import ism_oracle as IO
from oracle import pem_oracle as O
from oracle import pointops as P
from sam6d_hip import synth


def _sha(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def sd():
    return synth.make_pem_weights(1)


def test_param_inventory(sd):
    # 5 + 107 + 196 tensors; gen_golden.py loaded exactly these keys into the reference modules strict=True
    keys = list(sd)
    assert len([k for k in keys if k.startswith("geo_embedding.")]) == 5
    assert len([k for k in keys if k.startswith("coarse_point_matching.")]) == 107
    assert len([k for k in keys if k.startswith("fine_point_matching.")]) == 196


def test_pointops_bit_exact():
    g = golden("pointops")
    xyz = _t(g["fps_xyz"])
    assert np.array_equal(P.furthest_point_sampling(xyz, 196).numpy(), g["fps_idx"])
    pts = _t(g["bq_pts"])
    q = (pts + 0.00000001).contiguous()
    i1 = P.ball_query(q, pts, 0.1, 32)
    assert np.array_equal(i1.numpy(), g["bq_r1"].astype(np.int32))
    assert np.array_equal(P.ball_query(q, pts, 0.2, 64).numpy(), g["bq_r2"].astype(np.int32))
    assert _sha(P.group_points(pts.transpose(1, 2).contiguous(), i1)) == str(g["group_hot_sha"])
    gen = torch.Generator().manual_seed(11)
    _ = torch.rand(2, 2048, 3, generator=gen)
    feats = torch.randn(16, 128, 256, generator=gen)
    out = P.gather_points(feats, _t(g["gather_idx"]))
    assert _sha(out) == str(g["gather_out_sha"])


def test_fps_reference_test_shape():
    g = golden("pointops")
    rs = np.random.RandomState(324)
    big = torch.from_numpy(rs.randn(1, 21000, 3).astype(np.float32))
    assert np.array_equal(P.furthest_point_sampling(big, 2048).numpy(), g["fps_big_idx"])
    nx = torch.from_numpy(rs.randn(1, 1024, 3).astype(np.float32))
    xx = torch.from_numpy(rs.randn(1, 256, 3).astype(np.float32))
    assert np.array_equal(P.ball_query(nx, xx, 0.1, 64).numpy(), g["bq_ref_test"].astype(np.int32))


def test_pairwise_distance_bit_recipe():
    g = golden("pairwise")
    pts = _t(g["pts"])
    assert np.array_equal(O.pairwise_distance(pts, pts).numpy(), g["pd"])
    pd2 = O.pairwise_distance(_t(g["a"]), _t(g["b"]))
    assert _sha(pd2) == str(g["pd2_sha"])


def test_geo_embedding(sd):
    g = golden("geo_embedding")
    pts = _t(g["pts"])
    d_idx, a_idx, knn = O.geo_embedding_indices(pts)
    assert np.array_equal(d_idx.numpy(), g["d_idx"])  # sqrt + division: exact
    assert np.array_equal(knn.numpy(), g["knn"].astype(np.int64))
    np.testing.assert_allclose(a_idx.numpy(), g["a_idx"], rtol=0, atol=2e-5)  # atan2/cross: libm-dependent
    out = O.geo_embedding(pts, sd)
    rows = g["rows"]
    np.testing.assert_allclose(out[:, rows].numpy(), g["out_rows"], rtol=0, atol=2e-5)


def _layer_inputs(seed, B=1, n=197):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, n, 256, generator=gen)
    y = torch.randn(B, n, 256, generator=gen)
    e0 = 0.5 * torch.randn(B, n, n, 256, generator=gen)
    e1 = 0.5 * torch.randn(B, n, n, 256, generator=gen)
    return x, y, e0, e1


def test_transformer_layers(sd):
    g = golden("transformer")
    x, y, e0, e1 = _layer_inputs(int(g["seed"]))
    p = "coarse_point_matching.transformers.0"
    np.testing.assert_allclose(O.rpe_transformer_layer(x, x, e0, sd, p + ".layers.0").numpy(), g["rpe"], atol=3e-5, rtol=0)
    np.testing.assert_allclose(O.transformer_layer(x, y, sd, p + ".layers.1").numpy(), g["cross"], atol=3e-5, rtol=0)
    f0, f1 = O.geometric_transformer(x, e0, y, e1, sd, p)
    np.testing.assert_allclose(f0.numpy(), g["f0"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(f1.numpy(), g["f1"], atol=5e-5, rtol=0)


def test_sparse_to_dense(sd):
    g = golden("sparse_to_dense")
    gen = torch.Generator().manual_seed(int(g["seed_dense"]))
    d0 = torch.randn(1, 2049, 256, generator=gen)
    d1 = torch.randn(1, 2049, 256, generator=gen)
    _, _, e0, e1 = _layer_inputs(int(g["seed_emb"]))
    p = "fine_point_matching.transformers.0"
    lin = O.linear_transformer_layer(d0[:, 1:].contiguous(), d1[:, 1:197].contiguous(), sd, p + ".dense_layer")
    np.testing.assert_allclose(lin[:, ::8].numpy(), g["lin_rows"], atol=5e-5, rtol=0)
    o0, o1 = O.sparse_to_dense_transformer(d0, e0, _t(g["idx0"]), d1, e1, _t(g["idx1"]), sd, p)
    np.testing.assert_allclose(o0[:, ::8].numpy(), g["out0_rows"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(o1[:, ::8].numpy(), g["out1_rows"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(o0[:, :4].numpy(), g["out0_head"], atol=1e-4, rtol=0)


def test_positional_encoding(sd):
    g = golden("pos_encoding")
    out = O.positional_encoding(_t(g["pts"]), sd, "fine_point_matching.PE")
    np.testing.assert_allclose(out[:, ::8].numpy(), g["out_rows"], atol=3e-5, rtol=0)


def test_feature_similarity():
    g = golden("similarity")
    gen = torch.Generator().manual_seed(int(g["seed"]))
    a = torch.randn(2, 197, 256, generator=gen)
    b = torch.randn(2, 197, 256, generator=gen)
    np.testing.assert_allclose(O.feature_similarity(a, b, 0.1).numpy(), g["out"], atol=1e-5, rtol=0)


def test_coarse_rt_known_answer_and_flat():
    g = golden("coarse_rt")
    p1, p2, model, u = _t(g["p1"]), _t(g["p2"]), _t(g["model"]), _t(g["u"])
    for tag in ("", "2"):
        att = _t(g["att" + tag])
        R, t, aux = O.compute_coarse_Rt(att, p1, p2, model, u, return_aux=True)
        # discrete stages: bit-exact hypothesis indices and foreground mask
        assert np.array_equal(aux["idx"].numpy().astype(np.int32), g["idx" + tag])
        assert np.array_equal(aux["w1"].numpy(), g["w1" + ("_2" if tag else "")])
        assert set(aux["top"][0].tolist()) == set(g["top" + tag][0].tolist())
        np.testing.assert_allclose(R.numpy(), g["R" + tag], atol=1e-5, rtol=0)
        np.testing.assert_allclose(t.numpy(), g["t" + tag], atol=1e-5, rtol=0)
    # the known-answer scene recovers the ground-truth motion (SURVEY 8c KAT)
    R, t = O.compute_coarse_Rt(_t(g["att"]), p1, p2, model, u)
    np.testing.assert_allclose(R.numpy(), g["R_gt"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(t.numpy(), g["t_gt"], atol=2e-5, rtol=0)


def test_weighted_sampling_dense_compare_equals_binary_search():
    gen = torch.Generator().manual_seed(7)
    w = torch.rand(2, 5000, generator=gen) ** 8
    w[1] = 0  # all-false row: cum/(0+1e-8) = 0 -> u > 0 never reached -> index 0
    u = torch.rand(2, 300, generator=gen)
    a = O.weighted_sampling(w, u, faithful=True)
    b = O.weighted_sampling(w, u, faithful=False)
    assert torch.equal(a, b)
    assert (b[1] == 0).all()
    cum = torch.cumsum(w, 1)
    assert torch.equal(P.cumsum_f32(w), cum)  # double-accumulate recipe


def _kat_atten(p1, p2, sharp, bg):
    B, n, _ = p1.shape
    d = torch.cdist(p1, p2)
    a = torch.full((B, n + 1, p2.shape[1] + 1), float(bg))
    a[:, 1:, 1:] = torch.clamp(1 - sharp * d, min=-1) / 0.1
    return a


def test_fine_rt():
    g = golden("fine_rt")
    p1, p2, Rg, tg = _t(g["p1"]), _t(g["p2"]), _t(g["R_gt"]), _t(g["t_gt"])
    model = p2[:, :1024].contiguous()
    att = _kat_atten((p1 - tg.unsqueeze(1)) @ Rg, p2, float(g["sharp"]), float(g["bg"]))
    R, t, s = O.compute_fine_Rt(att, p1, p2, model)
    np.testing.assert_allclose(R.numpy(), g["R"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(t.numpy(), g["t"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(s.numpy(), g["score"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(R.numpy(), g["R_gt"], atol=2e-3, rtol=0)
    att2 = torch.randn(1, 2049, 2049, generator=torch.Generator().manual_seed(int(g["att2_seed"]))) * float(g["att2_scale"])
    R2, t2, s2 = O.compute_fine_Rt(att2, p1, p2, model)
    np.testing.assert_allclose(R2.numpy(), g["R2"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(t2.numpy(), g["t2"], atol=1e-4, rtol=0)


def test_procrustes_batch():
    g = golden("procrustes")
    src, ref = _t(g["src"]), _t(g["ref"])
    R, t = O.weighted_procrustes(src, ref, None, weight_thresh=0.5)
    ok = np.ones(len(src), bool)
    ok[400:500] = False  # rank-1 (duplicate-sample) cases: LAPACK-dependent, SURVEY 7 'hard parts'
    np.testing.assert_allclose(R.numpy()[ok], g["R"][ok], atol=2e-4, rtol=0)
    np.testing.assert_allclose(t.numpy()[ok], g["t"][ok], atol=5e-4, rtol=0)
    assert np.isfinite(R.numpy()).all()


def test_pem_end_to_end_seam(sd):
    """Whole matching path at the post-feature-extraction seam on the known-answer scene, B=2."""
    g = golden("pem_e2e")
    inp = synth.kat_inputs(B=2, seed=int(g["kat_seed"]))
    R, t, s, aux = O.pem_match(inp["dense_pm"], inp["dense_fm"], inp["dense_po"], inp["dense_fo"], inp["radius"],
                               inp["model"], sd, inp["rand"], return_aux=True)
    assert np.array_equal(aux["fps_idx_m"].numpy().astype(np.int16), g["kat_fps_m"])
    assert np.array_equal(aux["fps_idx_o"].numpy().astype(np.int16), g["kat_fps_o"])
    np.testing.assert_allclose(R.numpy(), g["kat_R"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(t.numpy(), g["kat_t"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(s.numpy(), g["kat_score"], atol=1e-4, rtol=0)


def config1_inputs(g):
    """Seam inputs of SURVEY 8d config 1 from tests/golden/config1.npz (same generator walk as oracle/gen_golden.py:fx_config1):
    geometry from the fixture, features N(0,1) / uniforms from seed `feat_seed`; dense_fo = the template features at the FPS picks."""
    gen = torch.Generator().manual_seed(int(g["feat_seed"]))
    tem_feat = torch.randn(1, 5000, 256, generator=gen)
    dense_fm = torch.randn(1, 2048, 256, generator=gen)
    rand = torch.rand(1, 18000, generator=gen)
    dense_fo = tem_feat[:, torch.from_numpy(g["tem_idx"].astype(np.int64))[0]].contiguous()
    return dict(dense_pm=torch.from_numpy(g["dense_pm"]), dense_fm=dense_fm, dense_po=torch.from_numpy(g["dense_po"]), dense_fo=dense_fo,
                radius=torch.from_numpy(g["radius"]), model=torch.from_numpy(g["model"])[None].contiguous(), rand=rand)


def test_config1_example_seam(sd):
    """SURVEY 8d config 1 (demo Example: real depth map, camera, CAD model): the oracle against the reference's outputs, plus the
    data-preparation steps that feed the seam (template FPS, radius normalisation, get_test_data geometry)."""
    g = golden("config1")
    inp = config1_inputs(g)
    R, t, s, aux = O.pem_match(inp["dense_pm"], inp["dense_fm"], inp["dense_po"], inp["dense_fo"], inp["radius"], inp["model"], sd,
                               inp["rand"], return_aux=True)
    assert np.array_equal(aux["fps_idx_m"].numpy().astype(np.int16), g["fps_m"])
    assert np.array_equal(aux["fps_idx_o"].numpy().astype(np.int16), g["fps_o"])
    np.testing.assert_allclose(aux["init_R"].numpy(), g["R0"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(R.numpy(), g["R"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(t.numpy(), g["t"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(s.numpy(), g["score"], atol=1e-4, rtol=0)
    # template FPS on the CAD surface sample, radius normalisation
    tem = torch.from_numpy(g["tem_pts"])[None]
    idx = P.furthest_point_sampling(tem, 2048)
    assert np.array_equal(idx.numpy().astype(np.int16), g["tem_idx"])
    pm, po, radius = O.radius_normalize(torch.from_numpy(g["pts"])[None], tem[:, idx[0].long()])
    assert np.array_equal(pm.numpy(), g["dense_pm"]) and np.array_equal(po.numpy(), g["dense_po"]) and np.array_equal(radius.numpy(), g["radius"])
    # get_test_data geometry on the real depth map
    depth = (g["depth_u16"].astype(np.float32) * np.float32(g["depth_scale"]) / np.float32(1000.0)).astype(np.float32)
    mask = np.unpackbits(g["mask_bits"])[: 480 * 640].reshape(480, 640)
    o = O.proposal_geometry(mask, depth, g["K"], np.float32(g["model_radius"]))
    assert o["bbox"] == [int(v) for v in g["bbox"]] and len(o["choose"]) == int(g["n_keep"]) and np.array_equal(o["center"], g["center"])
    assert np.array_equal(o["cloud"][g["choose_idx"]], g["pts"])
    assert np.array_equal(O.get_resize_rgb_choose(o["choose"][g["choose_idx"]], o["bbox"], 224), g["rgb_choose"])


# ------------------------------------------------------------------------------------------------ ISM (a15-a18)
def ism_inputs(seed=0):
    """Same generator walk as oracle/gen_golden.py:fx_ism."""
    gen = torch.Generator().manual_seed(seed)
    Nq, Nt, D, Pn = 200, 42, 1024, 256
    q = torch.randn(Nq, D, generator=gen)
    base = torch.randn(D, generator=gen)
    ref = (base + 0.8 * torch.randn(1, Nt, D, generator=gen))
    q[:150] = base + 0.5 * ref[0, torch.randint(0, Nt, (150,), generator=gen)] + 0.9 * q[:150]
    q_appe = torch.nn.functional.normalize(torch.randn(Nq, Pn, D, generator=gen), dim=-1)
    r_appe = torch.nn.functional.normalize(torch.randn(1, Nt, Pn, D, generator=gen), dim=-1)
    q_appe = q_appe * (torch.rand(Nq, Pn, 1, generator=gen) > 0.3)
    r_appe = r_appe * (torch.rand(1, Nt, Pn, 1, generator=gen) > 0.3)
    for i in range(0, Nq, 2):
        tsel = int(torch.randint(0, Nt, (1,), generator=gen))
        q_appe[i, :128] = 0.9 * r_appe[0, tsel, :128] + 0.1 * q_appe[i, :128]
    poses = torch.eye(4).repeat(Nt, 1, 1)
    for i in range(Nt):
        poses[i, :3, :3] = synth.random_rotation(gen)
    poses[:, :3, 3] = torch.randn(Nt, 3, generator=gen) * 0.4
    pc = (torch.rand(1, 2048, 3, generator=gen) - 0.5) * 0.2
    H, W = 480, 640
    K, depth_scale = IO.caller_intrinsics()  # float64, as the reference's caller passes them
    depth = (800 + 200 * torch.rand(H, W, generator=gen)).to(torch.int32)
    depth[torch.rand(H, W, generator=gen) < 0.1] = 0
    return dict(q=q, ref=ref, q_appe=q_appe, r_appe=r_appe, poses=poses, pc=pc, K=K, depth_scale=depth_scale, depth=depth, gen=gen,
                H=H, W=W)


def ism_masks(gen, Ns, H=480, W=640):
    masks = torch.zeros(Ns, H, W)
    boxes = torch.zeros(Ns, 4, dtype=torch.long)
    for i in range(Ns):
        x0 = int(torch.randint(0, W - 120, (1,), generator=gen)); y0 = int(torch.randint(0, H - 120, (1,), generator=gen))
        w = int(torch.randint(40, 120, (1,), generator=gen)); h = int(torch.randint(40, 120, (1,), generator=gen))
        masks[i, y0:y0 + h, x0:x0 + w] = 1
        boxes[i] = torch.tensor([x0, y0, x0 + w, y0 + h])
    return masks, boxes


def test_ism_scoring():
    g = golden("ism")
    d = ism_inputs(int(g["seed"]))
    sim = IO.pairwise_similarity(d["q"], d["ref"])
    np.testing.assert_allclose(sim[:8].numpy(), g["sim_rows"], atol=1e-6, rtol=0)
    sel, obj, sem, best = IO.semantic_score(d["q"], d["ref"])
    assert np.array_equal(sel.numpy().astype(np.int32), g["sel"])
    assert np.array_equal(obj.numpy().astype(np.int32), g["obj"])
    assert np.array_equal(best.numpy().astype(np.int32), g["best"])
    np.testing.assert_allclose(sem.numpy(), g["sem"], atol=1e-6, rtol=0)
    qa = d["q_appe"][sel]
    appe, ref_sel = IO.appearance_score(best, obj, qa, d["r_appe"])
    np.testing.assert_allclose(appe.numpy(), g["appe"], atol=2e-6, rtol=0)
    masks, boxes = ism_masks(d["gen"], len(sel))
    assert np.array_equal(boxes.numpy().astype(np.int32), g["boxes"])
    vu = IO.project_template_to_image(best, obj, d["poses"], d["pc"], masks, d["depth"], d["K"], d["depth_scale"])
    assert _sha(vu) == str(g["vu_sha"]) and np.array_equal(vu.numpy(), g["vu"].astype(np.int32))
    tr = IO.query_translation(masks, d["depth"], d["K"], d["depth_scale"])
    assert tr.dtype == torch.float32 and np.array_equal(tr.numpy(), g["translate"])  # float64 inside, one rounding at the end
    iou, vis = IO.geometric_score(vu, boxes, qa, ref_sel, 0.5)
    np.testing.assert_allclose(vis.numpy(), g["vis"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(iou.numpy(), g["iou"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(IO.final_score(sem, appe, iou, vis).numpy(), g["final"], atol=2e-6, rtol=0)
    bq = boxes.clone()
    bq[3] = torch.tensor([0, 0, 2, 2])
    xyxy = torch.cat((vu.min(1).values, vu.max(1).values), -1)
    assert IO.compute_iou(xyxy, bq) == 0.0 and float(g["iou_quirk"]) == 0.0


# ------------------------------------------------------------------------------------------ next rows
def test_depth_to_cloud_matches_reference_capture():
    """get_point_cloud_from_depth (PEM/utils/data_utils.py:92-110) captured from the reference (oracle/gen_golden.py)."""
    g = golden("depth_cloud")
    full = O.depth_to_cloud(g["depth"], g["K"])
    assert _sha(torch.from_numpy(np.ascontiguousarray(full))) == str(g["full_sha"])
    crop = O.depth_to_cloud(g["depth"], g["K"], [int(v) for v in g["bbox"]])
    assert np.array_equal(crop, g["crop"])


def test_nms_oracle_known_answers():
    """torchvision.ops.nms restated (PARITY UNPINNED: torchvision is not installed); known answers worked by hand."""
    b = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0], [0, 0, 10, 5]])
    s = torch.tensor([0.9, 0.8, 0.7, 0.9, 0.95])
    # box 4 (score .95) overlaps 0 and 3 with IoU exactly 0.5 -> NOT suppressed at thresh 0.5 (strict >); 0 kills 3 (tie, stable) and 1
    assert IO.nms(b, s, 0.5).tolist() == [4, 0, 2]
    assert IO.nms(b, s, 0.49).tolist() == [4, 1, 2]
    assert IO.nms_per_object_id(b, s, torch.tensor([1, 0, 1, 1, 0]), 0.5).tolist() == [4, 1, 0, 2]
    assert IO.nms(b[:0], s[:0], 0.5).tolist() == []


def test_small_detection_keep_oracle():
    boxes = torch.tensor([[0, 0, 10, 10], [0, 0, 3, 3], [5, 5, 40, 40]])
    masks = torch.zeros(3, 50, 50)
    masks[0, :10, :10] = 1
    masks[1, :3, :3] = 1
    masks[2, 5:8, 5:8] = 1
    keep = IO.small_detection_keep(boxes, masks, 0.05, 3e-4)  # 0.05**2 = 6.25 px of 2500 -> box area > 6.25; mask > 0.75 px
    assert keep.tolist() == [True, True, True]
    keep = IO.small_detection_keep(boxes, masks, 0.1, 5e-3)  # box area > 25 px, mask area > 12.5 px
    assert keep.tolist() == [True, False, False]


def test_test_data_geometry_matches_reference_capture():
    """get_test_data's per-proposal geometry (PEM/run_inference_custom_pytorch.py:316-355) composed from the reference's own
    get_bbox / get_point_cloud_from_depth / get_resize_rgb_choose (oracle/gen_golden.py::fx_test_data)."""
    g = golden("test_data")
    kept = set(int(i) for i in g["kept"])
    for i in range(g["masks"].shape[0]):
        o = O.proposal_geometry(g["masks"][i], g["depth"], g["K"], g["radius"])
        assert (o is not None) == (i in kept)
        if o is None:
            continue
        assert o["bbox"] == [int(v) for v in g["p%d_bbox" % i]]
        assert len(o["choose"]) == int(g["p%d_n_keep" % i])
        assert np.array_equal(o["center"], g["p%d_center" % i])
        assert _sha(torch.from_numpy(o["choose"].astype(np.int32))) == str(g["p%d_choose_sha" % i])
        assert _sha(torch.from_numpy(np.ascontiguousarray(o["cloud"]))) == str(g["p%d_cloud_sha" % i])
        sel = g["p%d_sel" % i]
        assert np.array_equal(o["cloud"][sel], g["p%d_pts" % i])
        assert np.array_equal(O.get_resize_rgb_choose(o["choose"][sel], o["bbox"], int(g["img_size"])), g["p%d_rgb_choose" % i])


def test_mask_rle_oracle_matches_reference_capture_and_known_answers():
    """detection_ism.json `segmentation` (ISM/model/utils.py:25-43, 199-216).  tests/golden/rle.npz holds the RLE the reference's
    own segment_anything/utils/amg.py produced for seeded masks (same uncompressed COCO format; oracle/gen_golden.py:fx_rle)."""
    g = golden("rle")
    masks, counts, offs = g["masks"], g["counts"], g["offsets"]
    for i in range(masks.shape[0]):
        want = counts[offs[i]:offs[i + 1]].tolist()
        got = IO.mask_to_rle(IO.force_binary_mask(masks[i].astype(np.float32)))
        assert got == {"counts": want, "size": list(masks.shape[1:])}
        if i in (0, 6, 7):
            assert IO.mask_to_rle_loop(IO.force_binary_mask(masks[i])) == got
        assert np.array_equal(IO.rle_to_mask(got), masks[i] > 0)
    # hand-worked: 3 x 4 mask with (0,0) and (2,3) set -> column-major 1,0*10,1 -> leading zero run of length 0
    m = np.zeros((3, 4)); m[0, 0] = 1; m[2, 3] = 1
    assert IO.mask_to_rle(m)["counts"] == [0, 1, 10, 1] == IO.mask_to_rle_loop(m)["counts"]
    assert IO.mask_to_rle(np.zeros((2, 5)))["counts"] == [10] and IO.mask_to_rle(np.ones((2, 5)))["counts"] == [0, 10]
    # soft masks go through force_binary_mask (> 0), column-major order: [[0.2, 0], [0, -1]] -> 1,0,0,0
    assert IO.mask_to_rle(IO.force_binary_mask(np.array([[0.2, 0.0], [0.0, -1.0]])))["counts"] == [0, 1, 3]


def test_detection_records_format():
    """save_to_file + convert_npz_to_json (ISM/model/utils.py:153-173, 199-216): key order, xywh without +1, category_id = id + 1
    (LM-O: table), python scalars."""
    import json
    masks = np.zeros((2, 4, 6), np.float32); masks[0, 1:3, 2:5] = 1; masks[1, 0, 0] = 0.5
    rec = IO.detections_to_records(np.array([0, 3]), np.array([0.75, 0.5], np.float32), np.array([[2, 1, 5, 3], [0, 0, 1, 1]]), masks,
                                   scene_id=0, frame_id=0, runtime=0)
    assert list(rec[0].keys()) == ["scene_id", "image_id", "category_id", "bbox", "score", "time", "segmentation"]
    assert rec[0]["category_id"] == 1 and rec[1]["category_id"] == 4 and rec[0]["bbox"] == [2, 1, 3, 2]
    assert rec[0]["segmentation"] == {"counts": [9, 2, 2, 2, 2, 2, 5], "size": [4, 6]} and rec[1]["segmentation"]["counts"] == [0, 1, 23]
    assert json.loads(json.dumps(rec)) == rec
    lmo = IO.detections_to_records(np.array([1, 7]), np.array([0.1, 0.2]), np.array([[0, 0, 1, 1], [0, 0, 1, 1]]), masks, dataset_name="lmo")
    assert [r["category_id"] for r in lmo] == [5, 12]


def test_compressed_rle_string_codec_known_answers():
    """pycocotools' compressed counts strings, restated (PARITY UNPINNED: the package is absent): hand-worked strings.  A value is
    written in 5-bit groups, least significant first, each + 48; bit 5 = another group follows; bit 4 of the last group = sign:
      0 -> '0'; 5 -> '5'; 15 -> '?' (0x0f + 48); 16: its first group 0x10 has bit 4 set although the value is positive, so a second
      group (0) must follow: chr(0x30 + 48) then '0'; 100 = 3 * 32 + 4 -> groups 4 (more), 3 -> 'T3';
      counts from the third on are stored as differences to the count two places before: [0, 5, 7, 5] -> '0', '5', '7', then 5 - 5 = 0 -> '0'."""
    enc, dec = IO.rle_counts_to_string, IO.rle_counts_from_string
    assert enc([0]) == "0" and enc([5]) == "5" and enc([15]) == "?"
    assert enc([100]) == "T3"                       # 4 | 0x20 -> 36 + 48 = 'T'; then 3 -> '3'
    assert enc([16]) == chr(0x30 + 48) + "0"        # group 0x10 has the sign bit set and x is not -1: continue (| 0x20), then 0
    assert enc([0, 5, 7, 5]) == "0570"
    assert enc([0, 5, 7, 3]) == "057" + chr((-2 & 0x1F) + 48)   # 3 - 5 = -2 -> one group 0b11110 (sign bit set, x becomes -1: stop)
    for counts in ([0, 5, 7, 5], [3, 1, 4, 1, 5, 9, 2, 6], [307200], [0, 307200], [1000, 2000, 1, 70000, 31, 32, 33]):
        assert dec(enc(counts)) == counts
    from sam6d_hip import ism
    for counts in ([0, 5, 7, 3], [12345, 1, 2, 3, 40000]):
        assert ism.rle_counts_to_string(counts) == enc(counts) and ism.rle_string_to_counts(enc(counts)) == counts
