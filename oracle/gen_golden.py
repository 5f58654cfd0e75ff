#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- captures golden vectors from the REFERENCE ITSELF (run in the build container,
where /root/reference exists; never on the GPU box) and writes them to tests/golden/*.npz.

What it does (SURVEY 8c oracle recipe):
  1. builds the reference's own pointnet2 `_ext` (CPU loops) from its sources in place -> oracle/_ref/ (build_ref.py)
  2. registers it as `pointnet2._ext`, puts the reference's PEM / ISM directories on sys.path exactly like
     PEM/run_inference_custom_pytorch.py:69-73 does, and imports the reference's hot-path modules
  3. drives them with seeded inputs (shapes/seeds follow the reference's own op tests) and stores
     inputs-or-seeds + outputs as small fixtures (large exact outputs as sha256 digests + slices)
  4. cross-checks this repo's oracle (oracle/pem_oracle.py, ism_oracle.py, pointops_oracle.c) on the same inputs.

No reference source text is stored: fixtures hold numbers only.
Usage:  python oracle/gen_golden.py [--only name,...]
"""
import argparse
import hashlib
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
sys.dont_write_bytecode = True

from oracle import build_ref, pem_oracle as O, ism_oracle as IO, pointops as P  # noqa: E402
from sam6d_hip import synth  # noqa: E402

PEM = "/root/reference/SAM-6D/Pose_Estimation_Model"
ISM = "/root/reference/SAM-6D/Instance_Segmentation_Model"


def sha(t):
    a = t.detach().contiguous().cpu().numpy()
    return hashlib.sha256(a.tobytes()).hexdigest()


def gen(seed):
    return torch.Generator().manual_seed(seed)


def save(name, **kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("  wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def same(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.equal(a, b), "%s: oracle != reference (max abs diff %g)" % (what, (a.double() - b.double()).abs().max())


def close(a, b, what, tol):
    d = (a.double() - b.double()).abs().max().item()
    assert d <= tol, "%s: oracle vs reference max abs diff %g > %g" % (what, d, tol)
    return d


# ------------------------------------------------------------------------------------------ reference import
def import_reference_pem():
    ext = build_ref.build()
    assert ext is not None, "reference sources not present"
    pkg = types.ModuleType("pointnet2")
    pkg.__path__ = []
    pkg._ext = ext
    sys.modules["pointnet2"] = pkg
    sys.modules["pointnet2._ext"] = ext
    for d in ("", "provider", "utils", "model", os.path.join("model", "pointnet2")):
        sys.path.append(os.path.join(PEM, d))
    mods = {n: importlib.import_module(n) for n in
            ("pointnet2_utils", "model_utils", "transformer", "coarse_point_matching", "fine_point_matching")}
    return ext, mods


class Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def ref_cfgs():
    geo = Cfg(sigma_d=0.2, sigma_a=15, angle_k=3, reduction_a="max", hidden_dim=256)
    coarse = Cfg(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, temp=0.1, sim_type="cosine",
                 normalize_feat=True, loss_dis_thres=0.15, nproposal1=6000, nproposal2=300)
    fine = Cfg(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, pe_radius1=0.1, pe_radius2=0.2,
               focusing_factor=3, temp=0.1, sim_type="cosine", normalize_feat=True, loss_dis_thres=0.15)
    return geo, coarse, fine


def sub_sd(sd, prefix):
    return {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


class RandPatch:
    """Replaces torch.rand so the reference's weighted sampling (model_utils.py:292) consumes OUR uniforms."""

    def __init__(self, u):
        self.u = u

    def __enter__(self):
        self.orig = torch.rand
        u = self.u

        def fake(*size, **kw):
            assert tuple(size) == tuple(u.shape), (size, u.shape)
            return u.clone()

        torch.rand = fake

    def __exit__(self, *a):
        torch.rand = self.orig


# ------------------------------------------------------------------------------------------------- fixtures
def fx_test_data(ext, mods):
    """the proposal geometry of get_test_data (PEM/run_inference_custom_pytorch.py:316-355).  The function itself
    reads files and needs cocomask / trimesh / cv2; its helpers get_bbox, get_resize_rgb_choose and get_point_cloud_from_depth
    (PEM/utils/data_utils.py) are called from the reference here and composed exactly as lines 316-355 compose them."""
    for name in ("imageio", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    du = importlib.import_module("data_utils")
    g = np.random.default_rng(11)
    H, Wd = 120, 160
    yy, xx = np.mgrid[0:H, 0:Wd]
    depth = (0.9 + 0.002 * xx + 0.001 * yy + 0.01 * g.standard_normal((H, Wd))).astype(np.float32)
    depth[g.random((H, Wd)) < 0.08] = 0
    depth[40:60, 60:64] = 3.5  # a strip far behind the object: removed by the radius filter
    K = np.array([[200.0, 0, 80.0], [0, 200.0, 60.0], [0, 0, 1]], np.float32)
    radius = np.float32(0.16)
    masks = np.zeros((7, H, Wd), np.uint8)
    masks[0] = ((xx - 70) ** 2 + (yy - 50) ** 2 < 24 ** 2) & (g.random((H, Wd)) > 0.15)  # blob in the middle
    masks[1, 0:30, 0:12] = 1            # tall box in the corner: the square crop is shifted back into the image
    masks[2, 100:120, 130:160] = 1      # bottom-right corner
    masks[3, 55:60, 10:150] = 1         # wide strip: side limited by min(H, W)
    masks[4, 10:15, 10:15] = 1          # 25 px: skipped (<= 32 valid pixels)
    masks[5] = (np.abs(xx - 100) + np.abs(yy - 70) < 21)  # diamond, odd tight box
    masks[6, 20:50, 20:50] = (g.random((30, 30)) > 0.5)
    whole = du.get_point_cloud_from_depth(depth, K)
    ns, img_size = 64, 224
    out = {}
    keep_ids, sels = [], []
    for i in range(masks.shape[0]):
        mask = np.logical_and(masks[i] > 0, depth > 0)
        ref = None
        if np.sum(mask) > 32:
            bbox = du.get_bbox(mask)
            y1, y2, x1, x2 = bbox
            m = mask[y1:y2, x1:x2]
            choose = m.astype(np.float32).flatten().nonzero()[0]
            cloud = whole.copy()[y1:y2, x1:x2, :].reshape(-1, 3)[choose, :]
            center = np.mean(cloud, axis=0)
            tmp = cloud - center[None, :]
            flag = np.linalg.norm(tmp, axis=1) < np.float32(np.float64(radius) * 1.2)  # `radius * 1.2` under numpy 1.26
            if np.sum(flag) >= 4:
                choose, cloud = choose[flag], cloud[flag]
                sel = g.integers(0, len(choose), ns)
                rc = du.get_resize_rgb_choose(choose[sel], [y1, y2, x1, x2], img_size)
                ref = dict(bbox=[int(v) for v in bbox], n_keep=len(choose), center=center, pts=cloud[sel], rgb_choose=rc, sel=sel,
                           choose_sha=sha(torch.from_numpy(choose.astype(np.int32))), cloud_sha=sha(torch.from_numpy(cloud.copy())))
        o = O.proposal_geometry(masks[i], depth, K, radius)
        assert (o is None) == (ref is None), "oracle / reference disagree on skipping proposal %d" % i
        if ref is None:
            continue
        assert o["bbox"] == ref["bbox"] and np.array_equal(o["center"], ref["center"]) and len(o["choose"]) == ref["n_keep"]
        assert np.array_equal(o["cloud"][ref["sel"]], ref["pts"])
        assert np.array_equal(O.get_resize_rgb_choose(o["choose"][ref["sel"]], o["bbox"], img_size), ref["rgb_choose"])
        keep_ids.append(i)
        for k, v in ref.items():
            out["p%d_%s" % (i, k)] = np.asarray(v) if not isinstance(v, str) else v
    save("test_data", depth=depth, masks=masks, K=K, radius=radius, ns=np.int32(ns), img_size=np.int32(img_size),
         kept=np.array(keep_ids, np.int32), **out)


def fx_depth_cloud(ext, mods):
    """get_point_cloud_from_depth (PEM/utils/data_utils.py:92-110) run from the reference itself (empty
    stub modules for the absent imageio / cv2, which this function does not touch).  K is handed over as float32 so that the
    result does not depend on the numpy major version (1.26 value-based casting == 2.x with a float32 scalar)."""
    for name in ("imageio", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    du = importlib.import_module("data_utils")
    g = np.random.default_rng(5)
    depth = (g.random((120, 160)) * 1.5 + 0.3).astype(np.float32)
    depth[g.random((120, 160)) < 0.1] = 0
    K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], np.float32)
    full = du.get_point_cloud_from_depth(depth, K)
    crop = du.get_point_cloud_from_depth(depth, K, [10, 90, 33, 150])
    assert full.dtype == np.float32 and crop.dtype == np.float32
    o_full = O.depth_to_cloud(depth, K)
    o_crop = O.depth_to_cloud(depth, K, [10, 90, 33, 150])
    assert np.array_equal(o_full, full) and np.array_equal(o_crop, crop), "oracle depth_to_cloud differs from the reference"
    save("depth_cloud", depth=depth, K=K, bbox=np.array([10, 90, 33, 150], np.int32), full_sha=sha(torch.from_numpy(full.copy())),
         crop=crop)


def fx_pointops(ext, mods):
    g = gen(11)
    xyz = torch.rand(2, 2048, 3, generator=g) - 0.5
    xyz[1, ::7] *= 0.02  # cloud 1: ~290 points inside the 0.0316 origin ball (skip branch, sampling.cpp:102-103)
    xyz[1, 0] = torch.tensor([0.3, -0.2, 0.1])
    idx = ext.furthest_point_sampling(xyz, 196)
    same(P.furthest_point_sampling(xyz, 196), idx, "fps")
    # the reference op test's shape (ov_test_furthest_point_sampling_1input.py:31-34): (1,21000,3) -> 2048
    rs = np.random.RandomState(324)
    big = torch.from_numpy(rs.randn(1, 21000, 3).astype(np.float32))
    idx_big = ext.furthest_point_sampling(big, 2048)
    same(P.furthest_point_sampling(big, 2048), idx_big, "fps big")
    # gather (ov_test_gather_operation.py:42-45: feats (16,128,256), idx (16,64))
    feats = torch.randn(16, 128, 256, generator=g)
    gidx = torch.randint(0, 256, (16, 64), generator=g, dtype=torch.int32)
    gidx[0, 0] = -1
    gidx[0, 1] = 256  # out-of-range -> 0 (sampling.cpp:34-40)
    gout = ext.gather_points(feats, gidx)
    same(P.gather_points(feats, gidx), gout, "gather")
    # ball query at the hot-path shape (fine_point_matching.py:117: new_xyz = pts + 1e-8)
    pts = torch.rand(2, 2048, 3, generator=g) - 0.5
    q = pts + 0.00000001
    bq1 = ext.ball_query(q.contiguous(), pts, 0.1, 32)
    bq2 = ext.ball_query(q.contiguous(), pts, 0.2, 64)
    same(P.ball_query(q.contiguous(), pts, 0.1, 32), bq1, "ball_query r1")
    same(P.ball_query(q.contiguous(), pts, 0.2, 64), bq2, "ball_query r2")
    # reference op test shape (ov_test_ball_query.py:24-27): new_xyz (1,1024,3), xyz (1,256,3) randn, r=.1, ns=64
    nx = torch.from_numpy(rs.randn(1, 1024, 3).astype(np.float32))
    xx = torch.from_numpy(rs.randn(1, 256, 3).astype(np.float32))
    bq3 = ext.ball_query(nx, xx, 0.1, 64)
    same(P.ball_query(nx, xx, 0.1, 64), bq3, "ball_query ref-test")
    # grouping (ov_test_grouping_operation.py:29-33: feats (7,3,2048), idx (7,2048,32)) -> digest
    gf = torch.randn(7, 3, 2048, generator=g)
    gi = torch.randint(0, 2048, (7, 2048, 32), generator=g, dtype=torch.int32)
    go = ext.group_points(gf, gi)
    same(P.group_points(gf, gi), go, "group")
    grp_hot = ext.group_points(pts.transpose(1, 2).contiguous(), bq1)
    save("pointops", fps_xyz=xyz, fps_idx=idx, fps_big_idx=idx_big.to(torch.int16) if False else idx_big,
         gather_seed=11, gather_idx=gidx, gather_out_sha=sha(gout), gather_out_b0=gout[0],
         bq_pts=pts, bq_r1=bq1.to(torch.int16), bq_r2=bq2.to(torch.int16), bq_ref_test=bq3.to(torch.int16),
         group_out_sha=sha(go), group_out_b0c0=go[0, 0, :64], group_hot_sha=sha(grp_hot))


def fx_pairwise(ext, mods):
    MU = mods["model_utils"]
    g = gen(12)
    pts = torch.rand(2, 196, 3, generator=g) - 0.5
    pts[1] += torch.tensor([0.0, 0.0, 8.0])
    pts = torch.cat([torch.ones(2, 1, 3) * 100, pts], 1)
    pd = MU.pairwise_distance(pts, pts)
    same(O.pairwise_distance(pts, pts), pd, "pairwise 197")
    a = torch.rand(3, 196, 3, generator=g) - 0.5
    b = torch.rand(3, 1024, 3, generator=g) - 0.5
    pd2 = MU.pairwise_distance(a, b)
    same(O.pairwise_distance(a, b), pd2, "pairwise 196x1024")
    save("pairwise", pts=pts, pd=pd, a=a, b=b, pd2_sha=sha(pd2), pd2_b0=pd2[0, :16])


def fx_geo(ext, mods):
    T = mods["transformer"]
    geo_cfg, _, _ = ref_cfgs()
    sd = synth.make_pem_weights(1)
    m = T.GeometricStructureEmbedding(geo_cfg).eval()
    m.load_state_dict(sub_sd(sd, "geo_embedding"), strict=True)
    g = gen(13)
    pts = torch.rand(2, 196, 3, generator=g) - 0.5
    pts[1] += torch.tensor([0.0, 0.0, 8.0])
    pts = torch.cat([torch.ones(2, 1, 3) * 100, pts], 1)
    with torch.no_grad():
        d_idx, a_idx = m.get_embedding_indices(pts)
        out = m(pts)
    od, oa, knn = O.geo_embedding_indices(pts)
    same(od, d_idx, "geo d_idx")
    same(oa, a_idx, "geo a_idx")
    oo = O.geo_embedding(pts, sd)
    close(oo, out, "geo out", 1e-5)
    rows = np.array([0, 1, 57, 196])
    save("geo_embedding", pts=pts, d_idx=d_idx, a_idx=a_idx, knn=knn.to(torch.int16), rows=rows, out_rows=out[:, rows],
         out_absmax=out.abs().max())


def _layer_inputs(seed, B=1, n=197):
    g = gen(seed)
    x = torch.randn(B, n, 256, generator=g)
    y = torch.randn(B, n, 256, generator=g)
    e0 = 0.5 * torch.randn(B, n, n, 256, generator=g)
    e1 = 0.5 * torch.randn(B, n, n, 256, generator=g)
    return x, y, e0, e1


def fx_transformer(ext, mods):
    T = mods["transformer"]
    sd = synth.make_pem_weights(1)
    p = "coarse_point_matching.transformers.0"
    m = T.GeometricTransformer(blocks=["self", "cross"], d_model=256, num_heads=4, dropout=None,
                               activation_fn="ReLU", return_attention_scores=False).eval()
    m.load_state_dict(sub_sd(sd, p), strict=True)
    x, y, e0, e1 = _layer_inputs(14)
    with torch.no_grad():
        rpe = m.layers[0](x, x, e0)[0]
        crs = m.layers[1](x, y)[0]
        f0, f1 = m(x, e0, y, e1)
    close(O.rpe_transformer_layer(x, x, e0, sd, p + ".layers.0"), rpe, "rpe layer", 2e-5)
    close(O.transformer_layer(x, y, sd, p + ".layers.1"), crs, "cross layer", 2e-5)
    o0, o1 = O.geometric_transformer(x, e0, y, e1, sd, p)
    close(o0, f0, "geo transformer f0", 5e-5)
    close(o1, f1, "geo transformer f1", 5e-5)
    save("transformer", seed=14, rpe=rpe, cross=crs, f0=f0, f1=f1)


def fx_submodules(ext, mods):
    """Every sub-module forward of the reference's transformer.py that a caller can reach directly (SinusoidalPositionalEmbedding,
    MultiHeadAttention, AttentionLayer, AttentionOutput, RPEMultiHeadAttention, RPEAttentionLayer, LinearAttention, LinearAttentionLayer):
    outputs on the seeded layer inputs of fx_transformer / fx_linear_attention, sub-sampled rows."""
    T = mods["transformer"]
    sd = synth.make_pem_weights(1)
    p = "coarse_point_matching.transformers.0"
    m = T.GeometricTransformer(blocks=["self", "cross"], d_model=256, num_heads=4, dropout=None,
                               activation_fn="ReLU", return_attention_scores=False).eval()
    m.load_state_dict(sub_sd(sd, p), strict=True)
    x, y, e0, e1 = _layer_inputs(14)
    emb = T.SinusoidalPositionalEmbedding(256)
    idx = torch.tensor([[0.0, 0.37, 1.0, 4.5], [11.25, 23.9, 57.0, 866.0254]])
    with torch.no_grad():
        sin = emb(idx)
        rpe_l, cross_l = m.layers[0], m.layers[1]
        r_hid, r_sc = rpe_l.attention.attention(x, x, x, e0)
        r_out, r_sc2 = rpe_l.attention(x, x, e0)
        ffn = rpe_l.output(x)
        c_hid, c_sc = cross_l.attention.attention(x, y, y)
        c_out, _ = cross_l.attention(x, y)
    assert torch.equal(r_sc, r_sc2)
    pf = "fine_point_matching.transformers.0"
    s2d = T.SparseToDenseTransformer(256, num_heads=4, sparse_blocks=["self", "cross"], dropout=None, activation_fn="ReLU",
                                     focusing_factor=3, with_bg_token=True, replace_bg_token=True).eval()
    s2d.load_state_dict(sub_sd(sd, pf), strict=True)
    g = gen(15)
    d0 = torch.randn(1, 2049, 256, generator=g)
    d1 = torch.randn(1, 2049, 256, generator=g)
    q_in, m_in = d0[:, 1:].contiguous(), d1[:, 1:197].contiguous()
    with torch.no_grad():
        la = s2d.dense_layer.attention.attention(q_in, m_in, m_in)
        lal = s2d.dense_layer.attention(q_in, m_in)
    # (div_term is a registered buffer -- it travels with the checkpoint; torch.exp on another CPU can differ in the last bit)
    save("submodules", seed=14, seed_dense=15, sin_idx=idx, sin=sin, sin_div_term=emb.div_term, rpe_hidden=r_hid[:, ::4], rpe_scores=r_sc[:, :, ::8],
         rpe_attn_out=r_out[:, ::4], ffn_out=ffn[:, ::4], mha_hidden=c_hid[:, ::4], mha_scores=c_sc[:, :, ::8], attn_out=c_out[:, ::4],
         linattn=la[:, ::16], linattn_layer=lal[:, ::16])


def fx_linear_attention(ext, mods):
    T = mods["transformer"]
    sd = synth.make_pem_weights(1)
    p = "fine_point_matching.transformers.0"
    m = T.SparseToDenseTransformer(256, num_heads=4, sparse_blocks=["self", "cross"], dropout=None,
                                   activation_fn="ReLU", focusing_factor=3, with_bg_token=True,
                                   replace_bg_token=True).eval()
    m.load_state_dict(sub_sd(sd, p), strict=True)
    g = gen(15)
    d0 = torch.randn(1, 2049, 256, generator=g)
    d1 = torch.randn(1, 2049, 256, generator=g)
    _, _, e0, e1 = _layer_inputs(16)
    i0 = torch.randperm(2048, generator=g)[:196].to(torch.int32).unsqueeze(0).contiguous()
    i1 = torch.randperm(2048, generator=g)[:196].to(torch.int32).unsqueeze(0).contiguous()
    i0[0, 0] = 0  # fps always starts at 0 -> selects the bg token (transformer.py:667-705 quirk)
    i1[0, 0] = 0
    with torch.no_grad():
        lin = m.dense_layer(d0[:, 1:].contiguous(), d1[:, 1:197].contiguous())
        o0, o1 = m(d0, e0, i0, d1, e1, i1)
    close(O.linear_transformer_layer(d0[:, 1:].contiguous(), d1[:, 1:197].contiguous(), sd, p + ".dense_layer"), lin,
          "linear layer", 5e-5)
    q0, q1 = O.sparse_to_dense_transformer(d0, e0, i0, d1, e1, i1, sd, p)
    close(q0, o0, "s2d 0", 1e-4)
    close(q1, o1, "s2d 1", 1e-4)
    save("sparse_to_dense", seed_dense=15, seed_emb=16, idx0=i0, idx1=i1, lin_rows=lin[:, ::8],
         out0_rows=o0[:, ::8], out1_rows=o1[:, ::8], out0_head=o0[:, :4], out1_head=o1[:, :4])


def fx_pos_encoding(ext, mods):
    F_ = mods["fine_point_matching"]
    sd = synth.make_pem_weights(1)
    m = F_.PositionalEncoding(256, r1=0.1, r2=0.2).eval()
    m.load_state_dict(sub_sd(sd, "fine_point_matching.PE"), strict=True)
    g = gen(17)
    pts = torch.rand(2, 2048, 3, generator=g) - 0.5
    pts[1] = pts[1] * 0.6 + 0.1
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = m(pts)
    close(O.positional_encoding(pts, sd, "fine_point_matching.PE"), out, "PE", 2e-5)
    save("pos_encoding", pts=pts, out_rows=out[:, ::8], out_sum=out.double().sum(), out_absmax=out.abs().max())


def fx_similarity(ext, mods):
    MU = mods["model_utils"]
    g = gen(18)
    a = torch.randn(2, 197, 256, generator=g)
    b = torch.randn(2, 197, 256, generator=g)
    out = MU.compute_feature_similarity(a, b, "cosine", 0.1, True)
    close(O.feature_similarity(a, b, 0.1), out, "similarity", 1e-5)
    save("similarity", seed=18, out=out)


def kat_atten(p1, p2, sharp=4.0, bg=-10.0):
    """SURVEY 8c known-answer attention: att[1:,1:] = clamp(1 - sharp*dist, -1)/0.1, bg row/col = bg."""
    B, n, _ = p1.shape
    d = torch.cdist(p1, p2)
    a = torch.full((B, n + 1, p2.shape[1] + 1), bg)
    a[:, 1:, 1:] = torch.clamp(1 - sharp * d, min=-1) / 0.1
    return a


def fx_coarse_rt(ext, mods):
    MU = mods["model_utils"]
    g = gen(19)
    B = 2
    p2 = torch.rand(B, 196, 3, generator=g) - 0.5
    Rg = torch.stack([synth.random_rotation(g) for _ in range(B)])
    tg = torch.tensor([[0.1, -0.2, 0.3], [-0.05, 0.02, 0.4]])
    perm = torch.stack([torch.randperm(196, generator=g) for _ in range(B)])
    p1 = torch.gather(p2, 1, perm.unsqueeze(2).expand(B, 196, 3)) @ Rg.transpose(1, 2) + tg.unsqueeze(1)
    model = torch.rand(B, 1024, 3, generator=g) - 0.5
    model[:, :196] = p2
    p1_in_frame2 = (p1 - tg.unsqueeze(1)) @ Rg
    att = kat_atten(p1_in_frame2, p2)
    u = torch.rand(B, 18000, generator=g)
    with RandPatch(u):
        R, t = MU.compute_coarse_Rt(att, p1, p2, model, 6000, 300)
    oR, ot, aux = O.compute_coarse_Rt(att, p1, p2, model, u, return_aux=True)
    close(oR, R, "coarse KAT R", 1e-6)
    close(ot, t, "coarse KAT t", 1e-6)
    # faithful (dense compare) == binary search
    same(O.weighted_sampling(aux["weights"][:1], u[:1, :512], faithful=True),
         O.weighted_sampling(aux["weights"][:1], u[:1, :512], faithful=False), "sampling faithful")
    # a flat (random-feature-like) attention: no structure, exercises near-uniform sampling
    att2 = torch.randn(B, 197, 197, generator=g) * 2.0
    with RandPatch(u):
        R2, t2 = MU.compute_coarse_Rt(att2, p1, p2, model, 6000, 300)
    oR2, ot2, aux2 = O.compute_coarse_Rt(att2, p1, p2, model, u, return_aux=True)
    close(oR2, R2, "coarse flat R", 1e-6)
    close(ot2, t2, "coarse flat t", 1e-6)
    save("coarse_rt", att=att, att2=att2, p1=p1, p2=p2, model=model, u=u, R_gt=Rg, t_gt=tg, R=R, t=t, R2=R2, t2=t2,
         idx=aux["idx"].to(torch.int32), idx2=aux2["idx"].to(torch.int32), w1=aux["w1"], w1_2=aux2["w1"],
         weights_sha=sha(aux["weights"]), weights2_sha=sha(aux2["weights"]),
         top=aux["top"].to(torch.int32), top2=aux2["top"].to(torch.int32),
         dis=aux["dis"], dis2=aux2["dis"], scores=aux["scores"], scores2=aux2["scores"])


def fx_fine_rt(ext, mods):
    MU = mods["model_utils"]
    g = gen(20)
    B = 1
    p2 = torch.rand(B, 2048, 3, generator=g) - 0.5
    Rg = torch.stack([synth.random_rotation(g) for _ in range(B)])
    tg = torch.tensor([[0.02, -0.01, 0.03]])
    perm = torch.stack([torch.randperm(2048, generator=g) for _ in range(B)])
    p1 = torch.gather(p2, 1, perm.unsqueeze(2).expand(B, 2048, 3)) @ Rg.transpose(1, 2) + tg.unsqueeze(1)
    p1 = p1 + 0.002 * torch.randn(p1.shape, generator=g)
    model = p2[:, :1024].contiguous()
    att = kat_atten((p1 - tg.unsqueeze(1)) @ Rg, p2, sharp=8.0, bg=0.0)
    R, t, s = MU.compute_fine_Rt(att, p1, p2, model)
    oR, ot, os_ = O.compute_fine_Rt(att, p1, p2, model)
    close(oR, R, "fine R", 1e-6); close(ot, t, "fine t", 1e-6); close(os_, s, "fine score", 1e-6)
    att2 = torch.randn(B, 2049, 2049, generator=gen(21)) * 3.0
    R2, t2, s2 = MU.compute_fine_Rt(att2, p1, p2, model)
    oR2, ot2, os2 = O.compute_fine_Rt(att2, p1, p2, model)
    close(oR2, R2, "fine flat R", 1e-5); close(ot2, t2, "fine flat t", 1e-5); close(os2, s2, "fine flat s", 1e-6)
    save("fine_rt", p1=p1, p2=p2, R_gt=Rg, t_gt=tg, sharp=8.0, bg=0.0, att2_seed=21, att2_scale=3.0,
         R=R, t=t, score=s, R2=R2, t2=t2, score2=s2)


def fx_procrustes(ext, mods):
    MU = mods["model_utils"]
    rs = np.random.RandomState(324)
    torch.manual_seed(32)
    n = 4200  # the reference's SVD op test uses (42000,3,3) randn (ov_test_custom_svd.py:31-33); 1/10 here
    src = torch.from_numpy(rs.randn(n, 3, 3).astype(np.float32))
    ref = torch.from_numpy(rs.randn(n, 3, 3).astype(np.float32))
    # exact rigid pairs (zero residual) and degenerate (duplicate-sample => rank-1 H) cases
    g = gen(22)
    for i in range(0, 400):
        Rr = synth.random_rotation(g)
        ref[i] = src[i] @ Rr.T + torch.randn(3, generator=g)
    for i in range(400, 500):
        src[i, 2] = src[i, 1]
        ref[i, 2] = ref[i, 1]
    R, t = MU.weighted_procrustes(src, ref, None, weight_thresh=0.5)
    oR, ot = O.weighted_procrustes(src, ref, None, weight_thresh=0.5)
    same(oR, R, "procrustes R"); same(ot, t, "procrustes t")
    w = torch.rand(8, 2048, generator=g)
    s2 = torch.randn(8, 2048, 3, generator=g)
    Rr = torch.stack([synth.random_rotation(g) for _ in range(8)])
    r2 = s2 @ Rr.transpose(1, 2) + 0.01 * torch.randn(8, 2048, 3, generator=g)
    Rw, tw = MU.weighted_procrustes(s2, r2, w, weight_thresh=0.0)
    save("procrustes", src=src, ref=ref, R=R, t=t, seed_w=22, Rw=Rw, tw=tw, w=w.half(), s2_sha=sha(s2))


def ref_seam(mods, sd):
    """The reference's GeometricStructureEmbedding / CoarsePointMatching / FinePointMatching modules loaded (strict) with `sd` and
    driven at the post-feature-extraction seam: Net.forward:29-55 restated (pose_estimation_model.Net itself needs timm)."""
    T, C_, F_, MU = mods["transformer"], mods["coarse_point_matching"], mods["fine_point_matching"], mods["model_utils"]
    geo_cfg, ccfg, fcfg = ref_cfgs()
    geo = T.GeometricStructureEmbedding(geo_cfg).eval()
    cpm = C_.CoarsePointMatching(ccfg).eval()
    fpm = F_.FinePointMatching(fcfg).eval()
    geo.load_state_dict(sub_sd(sd, "geo_embedding"), strict=True)
    cpm.load_state_dict(sub_sd(sd, "coarse_point_matching"), strict=True)
    fpm.load_state_dict(sub_sd(sd, "fine_point_matching"), strict=True)
    nref = len(geo.state_dict()) + len(cpm.state_dict()) + len(fpm.state_dict())
    assert nref == len(sd), (nref, len(sd))

    def ref_forward(inp):
        pm, fm, po, fo, radius, model, u = (inp[k] for k in
                                            ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model", "rand"))
        B = pm.shape[0]
        bg = torch.ones(B, 1, 3).float() * 100
        import contextlib, io
        with torch.no_grad(), RandPatch(u), contextlib.redirect_stdout(io.StringIO()):
            spm, sfm, im = MU.sample_pts_feats(pm, fm, 196, return_index=True)
            gm = geo(torch.cat([bg, spm], dim=1))
            spo, sfo, io_ = MU.sample_pts_feats(po, fo, 196, return_index=True)
            go = geo(torch.cat([bg, spo], dim=1))
            R0, t0 = cpm(spm, sfm, gm, spo, sfo, go, radius, model)
            R, t, s = fpm(pm, fm, gm, im, po, fo, go, io_, radius, model, R0, t0)
        return R, t, s, R0, t0, im, io_

    return ref_forward


def fx_pem_e2e(ext, mods):
    """Whole path at the post-feature-extraction seam (Net.forward:29-55 restated in ref_seam) with the
    reference's CoarsePointMatching / FinePointMatching / GeometricStructureEmbedding modules."""
    sd = synth.make_pem_weights(1)
    ref_forward = ref_seam(mods, sd)

    out = {}
    for tag, inp in (("kat", synth.kat_inputs(B=2, seed=3)), ("cfg2", synth.config2_inputs(B=2, seed=1))):
        R, t, s, R0, t0, im, io_ = ref_forward(inp)
        oR, ot, os_, aux = O.pem_match(inp["dense_pm"], inp["dense_fm"], inp["dense_po"], inp["dense_fo"],
                                       inp["radius"], inp["model"], sd, inp["rand"], return_aux=True)
        same(aux["fps_idx_m"], im, tag + " fps m"); same(aux["fps_idx_o"], io_, tag + " fps o")
        print("   %s: oracle-vs-reference dR0 %.2e dt0 %.2e dR %.2e dt %.2e ds %.2e" % (
            tag, (aux["init_R"] - R0).abs().max(), (aux["init_t"] - t0).abs().max(), (oR - R).abs().max(),
            (ot - t).abs().max(), (os_ - s).abs().max()))
        if tag == "kat":
            print("   kat: reference-vs-gt dR %.2e dt %.2e" % ((R - inp["R_gt"]).abs().max(), (t - inp["t_gt"]).abs().max()))
        out.update({tag + "_R": R, tag + "_t": t, tag + "_score": s, tag + "_R0": R0, tag + "_t0": t0,
                    tag + "_fps_m": im.to(torch.int16), tag + "_fps_o": io_.to(torch.int16)})
    save("pem_e2e", weights_seed=1, kat_seed=3, cfg2_seed=1, **out)


def read_ply_ascii(path):
    """Own reader for the ASCII PLY of the demo CAD model (vertex x y z ... / face `3 i j k`): returns (V,3) float64, (F,3) int64."""
    with open(path, "rb") as f:
        nv = nf = 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("format"):
                assert "ascii" in line, "binary PLY not handled"
            if line.startswith("element vertex"):
                nv = int(line.split()[-1])
            if line.startswith("element face"):
                nf = int(line.split()[-1])
            if line == "end_header":
                break
        body = f.read().decode("ascii").split("\n")
    V = np.array([[float(x) for x in body[i].split()[:3]] for i in range(nv)], np.float64)
    Fc = np.array([[int(x) for x in body[nv + i].split()[1:4]] for i in range(nf)], np.int64)
    return V, Fc


def sample_surface(V, Fc, n, rs):
    """Area-weighted uniform surface sample (what trimesh's mesh.sample does; trimesh is not installed, so own code + own seed)."""
    a, b, c = V[Fc[:, 0]], V[Fc[:, 1]], V[Fc[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    f = rs.choice(len(Fc), size=n, p=area / area.sum())
    u, v = rs.rand(n), rs.rand(n)
    flip = u + v > 1
    u[flip], v[flip] = 1 - u[flip], 1 - v[flip]
    return a[f] + u[:, None] * (b[f] - a[f]) + v[:, None] * (c[f] - a[f])


CONFIG1_SEED_PIXEL = (270, 405)  # (row, col) on the watering can (LM-O object 5) in SAM-6D/Data/Example/rgb.png


def fx_config1(ext, mods):
    """the demo Example (SAM-6D/Data/Example/{depth.png,camera.json,obj_000005.ply}), one proposal, 2048 points,
    through the reference modules at the post-feature-extraction seam.  No ISM weights exist offline, so the proposal mask is the set of
    pixels whose back-projected point lies within 1.2 x radius of the point at CONFIG1_SEED_PIXEL; the rest follows get_test_data
    (PEM/run_inference_custom_pytorch.py:292-355: depth -> cloud, mask & depth>0, get_bbox, radius filter, 2048 random points) with the
    reference's own helpers, and ViTEncoder.forward's radius normalisation (PEM/model/feature_extraction.py:133-137).  Features are
    N(0,1) (seed 1) and the weights random-init (seed 1): the ViT backbone and its checkpoint are outside the path."""
    import json
    from PIL import Image
    for name in ("imageio", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    du = importlib.import_module("data_utils")
    MU = mods["model_utils"]
    EX = "/root/reference/SAM-6D/Data/Example"
    cam = json.load(open(os.path.join(EX, "camera.json")))
    # K as float32: get_point_cloud_from_depth then computes in float32 under numpy 1.26 (the version the reference pins; value-based
    # casting) and numpy 2.x alike
    K = np.array(cam["cam_K"]).reshape(3, 3).astype(np.float32)
    depth_u16 = np.array(Image.open(os.path.join(EX, "depth.png")))
    assert depth_u16.dtype == np.uint16 and depth_u16.shape == (480, 640)
    whole_depth = (depth_u16.astype(np.float32) * np.float32(cam["depth_scale"]) / np.float32(1000.0)).astype(np.float32)
    whole_pts = du.get_point_cloud_from_depth(whole_depth, K)
    assert whole_pts.dtype == np.float32
    V, Fc = read_ply_ascii(os.path.join(EX, "obj_000005.ply"))
    rs = np.random.RandomState(0)
    model_points = (sample_surface(V, Fc, 1024, rs).astype(np.float32) / np.float32(1000.0)).astype(np.float32)
    tem_pts = (sample_surface(V, Fc, 5000, rs).astype(np.float32) / np.float32(1000.0)).astype(np.float32)
    radius = np.max(np.linalg.norm(model_points, axis=1))
    r0, c0 = CONFIG1_SEED_PIXEL
    assert whole_depth[r0, c0] > 0
    seed_pt = whole_pts[r0, c0]
    mask = np.linalg.norm(whole_pts - seed_pt[None, None, :], axis=2) < np.float32(np.float64(radius) * 1.2)
    mask = np.logical_and(mask, whole_depth > 0)
    assert np.sum(mask) > 32
    bbox = du.get_bbox(mask)
    y1, y2, x1, x2 = bbox
    m = mask[y1:y2, x1:x2]
    choose = m.astype(np.float32).flatten().nonzero()[0]
    cloud = whole_pts.copy()[y1:y2, x1:x2, :].reshape(-1, 3)[choose, :]
    center = np.mean(cloud, axis=0)
    flag = np.linalg.norm(cloud - center[None, :], axis=1) < np.float32(np.float64(radius) * 1.2)  # `radius * 1.2` under numpy 1.26
    assert np.sum(flag) >= 4
    choose, cloud = choose[flag], cloud[flag]
    rs1 = np.random.RandomState(1)
    assert len(choose) > 2048
    choose_idx = rs1.choice(np.arange(len(choose)), 2048, replace=False)
    pts = cloud[choose_idx]
    rgb_choose = du.get_resize_rgb_choose(choose[choose_idx], [y1, y2, x1, x2], 224)
    # template side: 2048 FPS points of the 5000-point surface sample with N(0,1) features (sample_pts_feats, as get_obj_feats does)
    g = gen(1)
    tem_feat = torch.randn(1, 5000, 256, generator=g)
    dense_fm = torch.randn(1, 2048, 256, generator=g)
    rand = torch.rand(1, 18000, generator=g)
    with torch.no_grad():
        dense_po_raw, dense_fo, tem_idx = MU.sample_pts_feats(torch.from_numpy(tem_pts)[None], tem_feat, 2048, return_index=True)
    # the caller `.repeat`s the template tensors per instance (run_inference_custom_pytorch.py:445-446), which makes them contiguous
    dense_po_raw, dense_fo = dense_po_raw.repeat(1, 1, 1).contiguous(), dense_fo.repeat(1, 1, 1).contiguous()
    # ViTEncoder.forward (feature_extraction.py:133-137)
    pts_t = torch.from_numpy(pts)[None]
    rad_t = torch.norm(dense_po_raw, dim=2).max(1)[0]
    dense_pm = pts_t / (rad_t.reshape(-1, 1, 1) + 1e-6)
    dense_po = dense_po_raw / (rad_t.reshape(-1, 1, 1) + 1e-6)
    model_t = torch.from_numpy(model_points)[None]
    sd = synth.make_pem_weights(1)
    inp = dict(dense_pm=dense_pm, dense_fm=dense_fm, dense_po=dense_po, dense_fo=dense_fo, radius=rad_t, model=model_t, rand=rand)
    R, t, sc, R0, t0, im, io_ = ref_seam(mods, sd)(inp)
    oR, ot, os_, aux = O.pem_match(dense_pm, dense_fm, dense_po, dense_fo, rad_t, model_t, sd, rand, return_aux=True)
    same(aux["fps_idx_m"], im, "config1 fps m"); same(aux["fps_idx_o"], io_, "config1 fps o")
    print("   config1: %d masked px, %d kept, radius %.4f; oracle-vs-reference dR0 %.2e dt0 %.2e dR %.2e dt %.2e ds %.2e" % (
        int(mask.sum()), len(choose), float(radius), (aux["init_R"] - R0).abs().max(), (aux["init_t"] - t0).abs().max(),
        (oR - R).abs().max(), (ot - t).abs().max(), (os_ - sc).abs().max()))
    save("config1", depth_u16=depth_u16, K=K, depth_scale=np.float32(cam["depth_scale"]), seed_pixel=np.array(CONFIG1_SEED_PIXEL, np.int32),
         mask_bits=np.packbits(mask), bbox=np.array(bbox, np.int32), n_keep=np.int32(len(choose)), center=center,
         choose_idx=choose_idx.astype(np.int32), pts=pts, rgb_choose=rgb_choose.astype(np.int64), model=model_points,
         model_radius=np.float32(radius), tem_pts=tem_pts, tem_idx=tem_idx.to(torch.int16), feat_seed=np.int32(1), weights_seed=np.int32(1),
         radius=rad_t, dense_pm=dense_pm, dense_po=dense_po, R=R, t=t, score=sc, R0=R0, t0=t0, fps_m=im.to(torch.int16),
         fps_o=io_.to(torch.int16))


# ----------------------------------------------------------------------------------------------------- ISM
def import_reference_ism():
    """model.loss / model.detector import after inserting EMPTY stub modules for the packages the
    container lacks (torchvision, pytorch_lightning, hydra, trimesh, ruamel.yaml ...)."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Any()

        def __getattr__(self, n):
            return _Any()

    for name in ("torchvision", "torchvision.ops", "torchvision.ops.boxes", "torchvision.utils",
                 "torchvision.transforms", "torchvision.transforms.functional", "ruamel", "ruamel.yaml", "hydra", "hydra.utils", "trimesh",
                 "pytorch_lightning", "omegaconf", "pycocotools", "pycocotools.mask", "imageio", "cv2",
                 "skimage", "skimage.feature", "skimage.transform", "distinctipy", "pandas_stub"):
        if name not in sys.modules:
            stub(name)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["hydra.utils"].instantiate = _Any()
    for n, attrs in (("torchvision.ops.boxes", ("batched_nms", "box_area")), ("torchvision.utils", ("make_grid", "save_image")),
                     ("torchvision.transforms", ("Compose", "Normalize", "Resize", "InterpolationMode")),
                     ("torchvision.ops", ("masks_to_boxes",)), ("torchvision.transforms.functional", ("resize", "to_pil_image")), ("omegaconf", ("DictConfig", "OmegaConf"))):
        for a in attrs:
            setattr(sys.modules[n], a, _Any())
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    # the PEM dirs on sys.path also hold top-level `utils` / `model` packages: make ISM's win
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k == "model" or k.startswith("model.")]:
        del sys.modules[k]
    sys.path[:] = [p for p in sys.path if not p.startswith(PEM)]
    sys.path.insert(0, ISM)
    loss = importlib.import_module("model.loss")
    det = importlib.import_module("model.detector")
    bbox = importlib.import_module("utils.bbox_utils")
    return loss, det, bbox


def fx_ism():
    loss, det, bbox = import_reference_ism()
    g = gen(0)
    Nq, Nt, D, Pn = 200, 42, 1024, 256
    q = torch.randn(Nq, D, generator=g)
    base = torch.randn(D, generator=g)
    ref = (base + 0.8 * torch.randn(1, Nt, D, generator=g))
    # realistic spread of cosine scores: 150 queries resemble the object (and one view more than the others),
    # 50 are clutter that must fall below the 0.2 confidence threshold (detector.py:285-287)
    q[:150] = base + 0.5 * ref[0, torch.randint(0, Nt, (150,), generator=g)] + 0.9 * q[:150]
    sim = loss.PairwiseSimilarity("cosine", 16)(q, ref)
    close(IO.pairwise_similarity(q, ref), sim, "ism pairwise", 1e-6)

    m = det.Instance_Segmentation_Model.__new__(det.Instance_Segmentation_Model)
    torch.nn.Module.__init__(m)
    m.matching_config = Cfg(metric=loss.PairwiseSimilarity("cosine", 16), aggregation_function="avg_5",
                            confidence_thresh=0.2)
    m.visible_thred = 0.5
    q_appe = torch.nn.functional.normalize(torch.randn(Nq, Pn, D, generator=g), dim=-1)
    r_appe = torch.nn.functional.normalize(torch.randn(1, Nt, Pn, D, generator=g), dim=-1)
    q_appe = q_appe * (torch.rand(Nq, Pn, 1, generator=g) > 0.3)
    r_appe = r_appe * (torch.rand(1, Nt, Pn, 1, generator=g) > 0.3)
    # plant visible correspondences so that the >0.5 threshold branch is exercised
    for i in range(0, Nq, 2):
        tsel = int(torch.randint(0, Nt, (1,), generator=g))
        q_appe[i, :128] = 0.9 * r_appe[0, tsel, :128] + 0.1 * q_appe[i, :128]
    poses = torch.eye(4).repeat(Nt, 1, 1)
    for i in range(Nt):
        poses[i, :3, :3] = synth.random_rotation(g)
    poses[:, :3, 3] = torch.randn(Nt, 3, generator=g) * 0.4
    pc = (torch.rand(1, 2048, 3, generator=g) - 0.5) * 0.2
    m.ref_data = {"descriptors": ref, "appe_descriptors": r_appe, "poses": poses, "pointcloud": pc}
    sel, obj, sem, best = m.compute_semantic_score(q)
    osel, oobj, osem, obest = IO.semantic_score(q, ref)
    same(osel, sel, "ism sel"); same(oobj, obj, "ism obj"); same(obest, best, "ism best")
    close(osem, sem, "ism sem", 1e-6)
    qa = q_appe[sel]
    appe, ref_sel = m.compute_appearance_score(best, obj, qa)
    oappe, oref = IO.appearance_score(best, obj, qa, r_appe)
    close(oappe, appe, "ism appe", 1e-6)
    # masks / depth / intrinsics (Example camera: SAM-6D/Data/Example/camera.json)
    H, W = 480, 640
    # dtypes exactly as the reference's caller builds them (ISM/run_inference_custom.py:86-96 batch_input_data):
    # cam_K = np.array(json list).reshape(3,3) -> float64; depth_scale = np.array(json float) -> float64 (1,); depth int32
    K = torch.from_numpy(np.array(IO.EXAMPLE_CAM_K).reshape((3, 3)))
    depth_scale = torch.from_numpy(np.array(1.0)).unsqueeze(0)
    assert K.dtype == torch.float64 and depth_scale.dtype == torch.float64
    depth = (800 + 200 * torch.rand(H, W, generator=g)).to(torch.int32)
    depth[torch.rand(H, W, generator=g) < 0.1] = 0
    Ns = len(sel)
    masks = torch.zeros(Ns, H, W)
    boxes = torch.zeros(Ns, 4, dtype=torch.long)
    for i in range(Ns):
        x0 = int(torch.randint(0, W - 120, (1,), generator=g)); y0 = int(torch.randint(0, H - 120, (1,), generator=g))
        w = int(torch.randint(40, 120, (1,), generator=g)); h = int(torch.randint(40, 120, (1,), generator=g))
        masks[i, y0:y0 + h, x0:x0 + w] = 1
        boxes[i] = torch.tensor([x0, y0, x0 + w, y0 + h])
    batch = {"depth": depth[None], "cam_intrinsic": K[None], "depth_scale": depth_scale}
    vu = m.project_template_to_image(best, obj, batch, masks.clone())
    tr = m.Calculate_the_query_translation(masks.clone(), depth, K, depth_scale)
    ovu = IO.project_template_to_image(best, obj, poses, pc, masks.clone(), depth, K, depth_scale)
    same(ovu, vu, "ism vu")
    same(IO.query_translation(masks.clone(), depth, K, depth_scale), tr, "ism translate")
    dets = Cfg(boxes=boxes)
    xyxy = torch.concatenate((torch.min(vu, dim=1).values, torch.max(vu, dim=1).values), dim=-1)
    iou, vis = m.compute_geometric_score(vu, dets, qa, ref_sel, visible_thred=0.5)
    oiou, ovis = IO.geometric_score(vu, boxes, qa, oref, 0.5)
    close(ovis, vis, "ism vis", 1e-6)
    assert torch.is_tensor(iou), "projected boxes overlap their own proposals -> tensor branch"
    close(oiou, iou, "ism iou", 1e-6)
    # the quirk branch (bbox_utils.py:214-220): ONE non-overlapping pair turns the whole result into scalar 0.0
    boxes_q = boxes.clone()
    boxes_q[3] = torch.tensor([0, 0, 2, 2])
    iou_q = bbox.compute_iou(xyxy, boxes_q)
    assert isinstance(iou_q, float) and iou_q == 0.0 and IO.compute_iou(xyxy, boxes_q) == 0.0
    fin = (sem + appe + iou * vis) / (1 + 1 + vis)
    close(IO.final_score(osem, oappe, oiou, ovis), fin, "ism final", 1e-6)
    save("ism", seed=0, sim_rows=sim[:8], sel=sel.to(torch.int32), obj=obj.to(torch.int32), sem=sem,
         best=best.to(torch.int32), appe=appe, vis=vis, vu_sha=sha(vu), vu=vu.to(torch.int16), translate=tr,
         xyxy=xyxy.to(torch.int32),
         boxes=boxes.to(torch.int32), iou=iou, iou_quirk=np.float32(iou_q), final=fin)


def fx_rle():
    """the `segmentation` field of detection_ism.json.  ISM/model/utils.py (mask_to_rle, convert_npz_to_json) imports
    torchvision, which is not installed, so it cannot be imported; the SAME uncompressed COCO RLE format is produced and read by the
    reference's segment_anything/utils/amg.py (mask_to_rle_pytorch :107-135, rle_to_mask :138-150), which IS importable: it is run
    here on seeded masks and edge cases and pins oracle/ism_oracle.py's restatement of mask_to_rle / rle_to_mask."""
    spec = importlib.util.spec_from_file_location("ref_amg", os.path.join(ISM, "segment_anything", "utils", "amg.py"))
    amg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(amg)
    g = torch.Generator().manual_seed(21)
    H, W = 60, 80
    masks = torch.zeros(8, H, W)
    for i in range(5):  # blobs: random rectangles with holes
        x0 = int(torch.randint(0, W - 30, (1,), generator=g)); y0 = int(torch.randint(0, H - 30, (1,), generator=g))
        w = int(torch.randint(8, 30, (1,), generator=g)); h = int(torch.randint(8, 30, (1,), generator=g))
        masks[i, y0:y0 + h, x0:x0 + w] = 1
        masks[i][torch.rand(H, W, generator=g) < 0.05] = 0
    masks[5] = (torch.rand(H, W, generator=g) < 0.5).float()  # noise: thousands of runs
    masks[6] = 1.0                                             # full: counts = [0, H*W]
    masks[7, 0, 0] = 1.0                                       # first pixel set, and (masks[7] otherwise empty)
    masks[4, H - 1, W - 1] = 1.0                               # last pixel set
    rles = amg.mask_to_rle_pytorch(masks > 0)
    flat, offs = [], [0]
    for i, r in enumerate(rles):
        assert r["size"] == [H, W]
        o = IO.mask_to_rle(IO.force_binary_mask(masks[i].numpy()))
        assert o == {"counts": r["counts"], "size": r["size"]}, "oracle mask_to_rle differs from the reference's RLE (mask %d)" % i
        assert o == IO.mask_to_rle_loop(IO.force_binary_mask(masks[i].numpy()))
        back = amg.rle_to_mask(r)
        assert np.array_equal(back, masks[i].numpy() > 0) and np.array_equal(IO.rle_to_mask(r), back)
        flat.extend(r["counts"]); offs.append(len(flat))
    empty = amg.mask_to_rle_pytorch(torch.zeros(1, H, W, dtype=torch.bool))[0]
    assert empty["counts"] == [H * W] == IO.mask_to_rle(np.zeros((H, W)))["counts"]
    save("rle", seed=21, masks=masks.to(torch.uint8), counts=np.asarray(flat, np.int32), offsets=np.asarray(offs, np.int64))


ALL = ["pointops", "pairwise", "geo", "transformer", "submodules", "linear_attention", "pos_encoding", "similarity", "coarse_rt",
       "fine_rt", "procrustes", "pem_e2e", "depth_cloud", "test_data", "config1", "ism", "rle"]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = [s for s in a.only.split(",") if s] or ALL
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    pem = [t for t in todo if t not in ("ism", "rle")]
    if pem:
        ext, mods = import_reference_pem()
        for t in pem:
            print("[gen_golden]", t)
            globals()["fx_" + t](ext, mods)
    if "ism" in todo:
        print("[gen_golden] ism")
        fx_ism()
    if "rle" in todo:
        print("[gen_golden] rle")
        fx_rle()
    print("done")
