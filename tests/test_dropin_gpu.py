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
