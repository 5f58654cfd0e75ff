"""GPU behaviour of the drop-in PEM modules: the same call sites the reference's scripts use, checked against the
reference's outputs (tests/golden/) -- reads like the reference's own op tests (module in, tensors out)."""
import importlib

import numpy as np
import pytest
import torch

from tests._util import golden
from sam6d_hip import synth

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def net(dev):
    m = importlib.import_module("pose_estimation_model").Net(synth.default_model_cfg())
    m.load_state_dict(synth.make_pem_weights(1), strict=False)
    return m.to(dev).eval()


def _close(got, want, atol, what):
    d = float((got.detach().float().cpu() - torch.as_tensor(want).float()).abs().max())
    assert d <= atol, "%s: max abs diff %.3e > %.1e" % (what, d, atol)


def test_net_match_known_answer(net, dev):
    g = golden("pem_e2e")
    inp = synth.kat_inputs(B=2, seed=int(g["kat_seed"]))
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    net.coarse_point_matching.hypothesis_rand = d["rand"]
    try:
        for fused in (True, False):  # one fused pipeline / module by module through the reference's call graph
            net.fused = fused
            with torch.no_grad():
                R, t, s = net.match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"])
            tag = " (fused=%s)" % fused
            _close(R, g["kat_R"], 1e-4, "pred_R" + tag); _close(t, g["kat_t"], 1e-4, "pred_t" + tag)
            _close(s, g["kat_score"], 1e-4, "score" + tag)
    finally:
        net.coarse_point_matching.hypothesis_rand = None
        net.fused = True


def test_pointnet2_utils_call_sites(dev):
    import pointnet2_utils as pu
    g = golden("pointops")
    xyz = _t(g["fps_xyz"]).to(dev)
    idx = pu.furthest_point_sample(xyz, torch.ones(196, dtype=torch.int32))  # npoint passed as a tensor (model_utils.py:75)
    assert np.array_equal(idx.cpu().numpy(), g["fps_idx"])
    pts = _t(g["bq_pts"]).to(dev)
    grp = pu.QueryAndGroup(0.1, 32)(pts, pts + 0.00000001, pts.transpose(1, 2).contiguous())
    assert grp.shape == (2, 6, 2048, 32)
    from oracle import pem_oracle as O
    want, _ = O.query_and_group(_t(g["bq_pts"]), _t(g["bq_pts"]) + 0.00000001, 0.1, 32)
    assert torch.equal(grp.cpu(), want)


def test_modules_forward(net, dev):
    g = golden("transformer")
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = torch.randn(1, 197, 256, generator=gen); y = torch.randn(1, 197, 256, generator=gen)
    e0 = 0.5 * torch.randn(1, 197, 197, 256, generator=gen); e1 = 0.5 * torch.randn(1, 197, 197, 256, generator=gen)
    tr = net.coarse_point_matching.transformers[0]
    with torch.no_grad():
        f0, f1 = tr(x.to(dev), e0.to(dev), y.to(dev), e1.to(dev))
    _close(f0, g["f0"], 1e-4, "GeometricTransformer f0"); _close(f1, g["f1"], 1e-4, "GeometricTransformer f1")
    gp = golden("pos_encoding")
    with torch.no_grad():
        pe = net.fine_point_matching.PE(_t(gp["pts"]).to(dev))
    _close(pe[:, ::8], gp["out_rows"], 5e-5, "PositionalEncoding")
    gg = golden("geo_embedding")
    with torch.no_grad():
        ge = net.geo_embedding(_t(gg["pts"]).to(dev))
    _close(ge[:, gg["rows"]], gg["out_rows"], 6e-5, "GeometricStructureEmbedding")


def test_model_utils_functions(dev):
    import model_utils as MU
    g = golden("coarse_rt")
    R, t = MU.compute_coarse_Rt(_t(g["att"]).to(dev), _t(g["p1"]).to(dev), _t(g["p2"]).to(dev), _t(g["model"]).to(dev),
                                6000, 300, rand=_t(g["u"]).to(dev))
    _close(R, g["R"], 1e-4, "compute_coarse_Rt R"); _close(t, g["t"], 1e-4, "compute_coarse_Rt t")
    gp = golden("procrustes")
    Rw, tw = MU.WeightedProcrustes(weight_thresh=0.5)(_t(gp["src"][:400]).to(dev), _t(gp["ref"][:400]).to(dev))
    _close(Rw, gp["R"][:400], 1e-4, "WeightedProcrustes R (exact rigid triples)")
    gs = golden("similarity")
    gen = torch.Generator().manual_seed(int(gs["seed"]))
    a = torch.randn(2, 197, 256, generator=gen); b = torch.randn(2, 197, 256, generator=gen)
    _close(MU.compute_feature_similarity(a.to(dev), b.to(dev), "cosine", 0.1, True), gs["out"], 2e-5, "similarity")


def test_net_forward_six_tensor_signature(net, dev):
    """The reference's call site (PEM/run_inference_custom_pytorch.py:447-454): model(pts, rgb, rgb_choose, model, dense_po, dense_fo)
    -> (pred_R, pred_t, pred_pose_score).  The ViT runs as plain PyTorch (random weights, out of scope); everything after it on the
    HIP path.  Checked: shapes, proper rotations, and that forward == match on the features the encoder produced."""
    g = torch.Generator().manual_seed(4)
    B = 2
    pts = ((torch.rand(B, 2048, 3, generator=g) - 0.5) * 0.2 + torch.tensor([0.0, 0.0, 0.8])).to(dev)
    rgb = torch.rand(B, 3, 224, 224, generator=g).to(dev)
    rgb_choose = torch.randint(0, 224 * 224, (B, 2048), generator=g).to(dev)
    model = ((torch.rand(B, 1024, 3, generator=g) - 0.5) * 0.2).to(dev)
    dense_po = ((torch.rand(B, 2048, 3, generator=g) - 0.5) * 0.2).to(dev)
    dense_fo = torch.randn(B, 2048, 256, generator=g).to(dev)
    rand = torch.rand(B, 18000, generator=g).to(dev)
    net.coarse_point_matching.hypothesis_rand = rand
    try:
        with torch.no_grad():
            R, t, s = net(pts, rgb, rgb_choose, model, dense_po, dense_fo)
            pm, fm, po, fo, radius = net.feature_extraction(pts, rgb, rgb_choose, dense_po, dense_fo)
            R2, t2, s2 = net.match(pm, fm, po, fo, radius, model)
    finally:
        net.coarse_point_matching.hypothesis_rand = None
    assert R.shape == (B, 3, 3) and t.shape == (B, 3) and s.shape == (B,)
    assert torch.isfinite(R).all() and torch.isfinite(t).all() and torch.isfinite(s).all()
    eye = torch.eye(3, device=dev).expand(B, 3, 3)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-5 and float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    assert torch.equal(R, R2) and torch.equal(t, t2) and torch.equal(s, s2)
    want_r = torch.norm(dense_po.cpu(), dim=2).max(1)[0]  # the torch-CPU recipe the reference runs (bit-exact on the device)
    assert torch.equal(radius.cpu(), want_r)


def test_get_obj_feats_template_sampling(net, dev):
    """ViTEncoder.get_obj_feats (PEM/model/feature_extraction.py:152-172): template views -> 2048 FPS points + features."""
    g = torch.Generator().manual_seed(5)
    T = 3
    rgbs = [torch.rand(1, 3, 224, 224, generator=g).to(dev) for _ in range(T)]
    ptss = [((torch.rand(1, 5000, 3, generator=g) - 0.5) * 0.2).to(dev) for _ in range(T)]
    chs = [torch.randint(0, 224 * 224, (1, 5000), generator=g).to(dev) for _ in range(T)]
    with torch.no_grad():
        p, f = net.feature_extraction.get_obj_feats(rgbs, ptss, chs)[:2]
    assert p.shape == (1, 2048, 3) and f.shape == (1, 2048, 256)
    allp = torch.cat(ptss, 1)[0]
    # every sampled point is one of the template points, and FPS starts with the first one
    member = (p[0][:, None, :] == allp[None, :, :]).all(-1).any(1)
    assert bool(member.all()) and torch.equal(p[0, 0], allp[0])
