"""GPU parity on BASELINE.json's configurations, through the C ABI via sam6d_hip:

* config 1 -- the demo Example (real depth map, camera and CAD model; tests/golden/config1.npz, written by oracle/gen_golden.py
  fx_config1 from the reference's own modules): get_test_data geometry, template FPS, radius normalisation, then the whole matching
  path, against the reference's outputs;
* config 2 at full size -- B = 32 proposals in one pem_match call (default kernels), all 32 against the CPU oracle, staged so that every
  discrete choice is either identical or an asserted within-rounding threshold case;
* config 4 on one GPU -- 200 proposals dealt round-robin to 8 shards, every shard through pem_match, the gathered rows put back in
  global order: bit-for-bit the unsharded result;
* the fine stage's label / weight work (index work) against the oracle.
"""
import numpy as np
import pytest
import torch

from tests._util import golden
from tests.test_oracle_golden import config1_inputs

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _close(got, want, atol, what=""):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what + ": non-finite values"
    d = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    assert d <= atol, "%s: max abs diff %.3e > %.1e" % (what, d, atol)
    return d


@pytest.fixture(scope="module")
def sd():
    from sam6d_hip import synth
    return synth.make_pem_weights(1)


@pytest.fixture(scope="module")
def W(sd, dev):
    from sam6d_hip import pem
    return pem.PemWeights(sd, dev)


# ------------------------------------------------------------------------------------------------------- config 1
def test_config1_test_data_geometry_bit_exact(dev):
    """Real depth map + camera: depth -> cloud, mask & depth > 0, crop box, radius filter, the caller's 2048 random picks and the
    resized-crop indices (PEM/run_inference_custom_pytorch.py:292-355), bit-exact against the reference's own helpers."""
    from sam6d_hip import pem
    g = golden("config1")
    depth = (g["depth_u16"].astype(np.float32) * np.float32(g["depth_scale"]) / np.float32(1000.0)).astype(np.float32)
    mask = np.unpackbits(g["mask_bits"])[: 480 * 640].reshape(1, 480, 640)
    geom = pem.proposal_geometry(torch.from_numpy(mask).to(dev), torch.from_numpy(depth).to(dev), g["K"], float(g["model_radius"]))
    assert geom["bbox"][0].cpu().tolist() == [int(v) for v in g["bbox"]]
    assert int(geom["n_keep"][0]) == int(g["n_keep"])
    assert np.array_equal(geom["center"][0].cpu().numpy(), g["center"])
    pts, rc = pem.proposal_choose(geom, torch.from_numpy(g["choose_idx"])[None].to(dev), 224)
    assert np.array_equal(pts[0].cpu().numpy(), g["pts"]), "observed points"
    assert np.array_equal(rc[0].cpu().numpy(), g["rgb_choose"]), "rgb_choose"


def test_config1_template_fps_and_radius_bit_exact(dev):
    """5000 CAD surface points -> 2048 by FPS (get_obj_feats' sample_pts_feats) and ViTEncoder.forward's radius normalisation
    (PEM/model/feature_extraction.py:133-137) on the real CAD geometry."""
    from sam6d_hip import ops, pem
    g = golden("config1")
    tem = torch.from_numpy(g["tem_pts"])[None].to(dev)
    idx = ops.furthest_point_sampling(tem, 2048)
    assert np.array_equal(idx.cpu().numpy().astype(np.int16), g["tem_idx"])
    po_raw = tem[0][idx[0].long()][None].contiguous()
    pm, po, radius = pem.radius_normalize(torch.from_numpy(g["pts"])[None].to(dev), po_raw)
    assert np.array_equal(radius.cpu().numpy(), g["radius"])
    assert np.array_equal(pm.cpu().numpy(), g["dense_pm"]) and np.array_equal(po.cpu().numpy(), g["dense_po"])


def test_config1_matching_path_vs_reference(dev, W):
    """The demo proposal through the whole matching path: FPS indices bit-exact, poses and score within 1e-4 of the reference."""
    from sam6d_hip import pem
    g = golden("config1")
    inp = config1_inputs(g)
    d = {k: v.to(dev) for k, v in inp.items()}
    R, t, s, aux = pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"],
                                 return_aux=True)
    assert np.array_equal(aux["fps_idx_m"].cpu().numpy().astype(np.int16), g["fps_m"])
    assert np.array_equal(aux["fps_idx_o"].cpu().numpy().astype(np.int16), g["fps_o"])
    _close(aux["init_R"], g["R0"], 1e-4, "coarse R"); _close(aux["init_t"], g["t0"], 1e-4, "coarse t")
    _close(R, g["R"], 1e-4, "R"); _close(t, g["t"], 1e-4, "t"); _close(s, g["score"], 1e-4, "score")
    # the default (fused, no aux) path must give the same poses
    R2, t2, s2 = pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"])
    _close(R2, g["R"], 1e-4, "R (fused)"); _close(t2, g["t"], 1e-4, "t (fused)"); _close(s2, g["score"], 1e-4, "score (fused)")


# ------------------------------------------------------------------------------------------------------- config 2
def _oracle_proposal(O, inp, b, sd, cfg):
    """The oracle on proposal b alone (O.pem_match's body, PEM/model/pose_estimation_model.py:29-55), keeping what the staged checks
    need: the sparse clouds, both geometric embeddings, the coarse intermediates and a closure that runs the fine stage from any pose."""
    sl = lambda k: inp[k][b:b + 1].contiguous()
    pm, fm, po, fo, radius, model, rand = (sl(k) for k in ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model", "rand"))
    n = cfg["coarse_npoint"]
    bgp = torch.ones(1, 1, 3) * 100
    spm, sfm, im = O.sample_pts_feats(pm, fm, n)
    gm = O.geo_embedding(torch.cat([bgp, spm], 1), sd, "geo_embedding", cfg["sigma_d"], cfg["sigma_a"], cfg["angle_k"])
    spo, sfo, io = O.sample_pts_feats(po, fo, n)
    go = O.geo_embedding(torch.cat([bgp, spo], 1), sd, "geo_embedding", cfg["sigma_d"], cfg["sigma_a"], cfg["angle_k"])
    R0, t0, caux = O.coarse_point_matching(spm, sfm, gm, spo, sfo, go, radius, model, sd, rand, cfg, False, True)
    fine = lambda Ri, ti: O.fine_point_matching(pm, fm, gm, im, po, fo, go, io, radius, model, Ri, ti, sd, cfg)
    R, t, s = fine(R0, t0)
    return dict(R=R, t=t, s=s, R0=R0, t0=t0, coarse=caux, fine=fine, spm=spm, spo=spo, im=im, io=io,
                mp=model / (radius.reshape(-1, 1, 1) + 1e-6), rand=rand)


def _d(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def test_config2_full_batch_vs_oracle(dev, W, sd):
    """B = 32 (the benchmark's step) in ONE pem_match call with the DEFAULT kernels, ALL 32 proposals against the CPU oracle, no waiver.

    The coarse pose is a discrete choice (18 000 threshold searches, a top-300, an arg-max: PEM/utils/model_utils.py:241-275), so on
    config 2's random features a 1e-7 difference can select another hypothesis and the end-to-end poses of such a proposal differ by
    O(1).  The test therefore proves the chain link by link, for every proposal:
      (1) FPS indices bit-exact; coarse attention within 1e-4;
      (2) every sampled index is a correct first-(cum >= u) index of the ORACLE's cumulative weights of that attention up to 1e-6
          (cum is normalised to 1), and the foreground masks are identical;
      (3) the oracle, continued from the GPU's sampled indices, selects the GPU's coarse pose (1e-4) -- or the GPU's pick scores within
          1e-6 relative of the oracle's best (an asserted tie) and is that hypothesis of the oracle (1e-4) -- or the GPU's pick is a
          hypothesis that samples a point twice (rank <= 1: the rotation is determined up to a twist, the reference's answer is LAPACK
          rounding noise), in which case the GPU's pose is validated as a minimiser and its score as the correctly taken maximum;
      (4) the oracle's fine stage continued from the GPU's coarse pose gives the GPU's final pose / score (1e-4);
      (5) the GPU's fine stage started from the ORACLE's coarse pose (pem_match(init_pose=...)) gives the oracle's final pose (1e-4).
    Where (2) and (3) hold with identical choices, (4) is the plain end-to-end comparison."""
    from oracle import pem_oracle as O
    from sam6d_hip import pem, synth
    B = 32
    cfg = O.DEFAULT_CFG
    inp = synth.config2_inputs(B=B, seed=1)
    d = {k: v.to(dev) for k, v in inp.items()}
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    R, t, s, aux = pem.pem_match(*[d[k] for k in keys], W, d["rand"], return_aux=True)
    R_plain, t_plain, s_plain = pem.pem_match(*[d[k] for k in keys], W, d["rand"])
    torch.cuda.synchronize()
    assert torch.equal(R, R_plain) and torch.equal(t, t_plain) and torch.equal(s, s_plain), "return_aux changed the result"
    assert torch.isfinite(R).all() and torch.isfinite(t).all() and torch.isfinite(s).all()
    _close(torch.linalg.det(R.cpu().double()).float(), torch.ones(B), 1e-5, "proper rotations")
    ca = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in aux["coarse"].items()}
    Rc, tc, sc = R.cpu(), t.cpu(), s.cpu()
    R0c, t0c = aux["init_R"].cpu(), aux["init_t"].cpu()
    fps_m, fps_o = aux["fps_idx_m"].cpu(), aux["fps_idx_o"].cpu()
    torch.set_num_threads(min(16, torch.get_num_threads() or 8))
    n1 = cfg["nproposal1"]
    oR0, ot0, oR, ot, os_ = [], [], [], [], []
    n_e2e_same, n_idx_diff, n_tie, n_degenerate, worst = 0, 0, 0, 0, dict(att=0.0, e2e=0.0, fine_given_pose=0.0)
    with torch.no_grad():
        for b in range(B):
            o = _oracle_proposal(O, inp, b, sd, cfg)
            oR0.append(o["R0"]); ot0.append(o["t0"]); oR.append(o["R"]); ot.append(o["t"]); os_.append(o["s"])
            tag = "proposal %d: " % b
            # (1)
            assert torch.equal(fps_m[b:b + 1], o["im"]) and torch.equal(fps_o[b:b + 1], o["io"]), tag + "FPS indices"
            att = ca["atten"][b:b + 1]
            worst["att"] = max(worst["att"], _d(att, o["coarse"]["atten"]))
            assert _d(att, o["coarse"]["atten"]) <= 1e-4, tag + "coarse attention off by %.2e" % _d(att, o["coarse"]["atten"])
            # (2) the oracle's sampling weights of the GPU's attention; thresholds up to 1e-6
            w_o, w1_o = O.coarse_sampling_weights(att)
            assert torch.equal(w1_o, ca["w1"][b:b + 1]), tag + "foreground mask differs"
            cum = torch.cumsum(w_o, dim=1)
            cum = (cum / (cum[:, -1].unsqueeze(1) + 1e-8))[0].double()
            u = o["rand"][0].double()
            gi = ca["idx"][b].long()
            eps = 1e-6
            ok_hi = cum[gi] >= u - eps
            ok_lo = (gi == 0) | (cum[(gi - 1).clamp(min=0)] < u + eps)
            none = u > cum[-1] - eps  # no cumulative weight reaches u: the reference's argmax of an all-false row is 0
            valid = (ok_hi & ok_lo) | (none & (gi == 0))
            assert bool(valid.all()), tag + "%d sampled indices are not first-(cum >= u) indices" % int((~valid).sum())
            n_idx_diff += int((gi != o["coarse"]["idx"][0]).sum())
            # (3) the oracle from the GPU's samples
            Rs_o, ts_o, dis_o = O.coarse_hypotheses(gi.unsqueeze(0), o["spm"], o["spo"], n1)
            R0o, t0o, top_o, sc_o = O.coarse_select(Rs_o, ts_o, dis_o, w1_o, o["spm"], o["mp"], cfg["nproposal2"])
            if max(_d(R0c[b:b + 1], R0o), _d(t0c[b:b + 1], t0o)) > 1e-4:
                h = int(ca["best"][b])
                tri = gi[3 * h:3 * h + 3]
                i1, i2 = (tri // 196).tolist(), (tri % 196).tolist()
                if len(set(i1)) < 3 or len(set(i2)) < 3:
                    # The winning hypothesis samples a scene point or a template point twice: its correlation matrix has rank <= 1 and
                    # the least-squares rotation is determined only up to a twist about one axis.  The reference fills that freedom with
                    # whatever LAPACK's sgesdd makes of rounding noise (SURVEY 7 'hard parts'), the GPU with a fixed rule -- both are
                    # minimisers.  Asserted instead: the GPU's pose IS a minimiser (a proper rotation whose residual on the triple equals
                    # the oracle's, which does not depend on the twist), the oracle's scoring of the GPU's pose reproduces the GPU's
                    # score, and that score is the maximum of the GPU's 300: the arg-max was taken correctly over valid poses.
                    Rg, tg = ca["Rs"][b, h].reshape(1, 3, 3), ca["ts"][b, h].reshape(1, 3)
                    assert abs(float(torch.linalg.det(Rg.double())) - 1.0) < 1e-5 and _d(Rg @ Rg.transpose(1, 2), torch.eye(3)[None]) < 1e-5
                    assert abs(float(ca["dis"][b, h]) - float(dis_o[0, h])) < 1e-4, tag + "rank-deficient pick: residual on its triple"
                    tp = ((o["spm"] - tg.unsqueeze(1)) @ Rg).contiguous()
                    dmin = torch.sqrt(O.pairwise_distance(tp, o["mp"].contiguous())).min(2)[0]
                    sc_g = float(w1_o.sum() / ((dmin * w1_o).sum() + 1e-8))
                    k_h = (ca["top"][b] == h).nonzero()
                    assert k_h.numel() == 1
                    got = float(ca["scores"][b, k_h[0, 0]])
                    assert abs(sc_g - got) <= 1e-4 * max(1.0, abs(sc_g)), tag + "rank-deficient pick: score %.6f vs the oracle's scoring of that pose %.6f" % (got, sc_g)
                    assert got >= float(ca["scores"][b].max()), tag + "rank-deficient pick is not the GPU's arg-max"
                    n_degenerate += 1
                else:
                    pos = (top_o[0] == h).nonzero()
                    assert pos.numel() == 1, tag + "the GPU's coarse pick (hypothesis %d) is not among the oracle's 300 candidates" % h
                    rel = float((sc_o[0].max() - sc_o[0, pos[0, 0]]) / sc_o[0].max())
                    assert rel < 1e-6, tag + "coarse pick differs and is no tie: the oracle scores it %.3e (relative) below its best" % rel
                    assert max(_d(R0c[b:b + 1], Rs_o[:, h]), _d(t0c[b:b + 1], ts_o[:, h, 0])) <= 1e-4, tag + "coarse pick: pose of hypothesis"
                    n_tie += 1
            # (4) fine stage given the GPU's coarse pose
            same = max(_d(R0c[b:b + 1], o["R0"]), _d(t0c[b:b + 1], o["t0"])) <= 1e-5
            if same:
                Rf, tf, sf = o["R"], o["t"], o["s"]
                n_e2e_same += 1
            else:
                Rf, tf, sf = o["fine"](R0c[b:b + 1], t0c[b:b + 1])
            dfin = max(_d(Rc[b:b + 1], Rf), _d(tc[b:b + 1], tf), _d(sc[b:b + 1], sf))
            worst["fine_given_pose"] = max(worst["fine_given_pose"], dfin)
            assert dfin <= 1e-4, tag + "final pose / score given the coarse pose off by %.2e (same coarse choice: %s)" % (dfin, same)
            if same:
                worst["e2e"] = max(worst["e2e"], dfin)
    # (5) the GPU's fine stage from the oracle's coarse pose, all 32 in one call
    init = (torch.cat(oR0).to(dev), torch.cat(ot0).to(dev))
    R5, t5, s5 = pem.pem_match(*[d[k] for k in keys], W, d["rand"], init_pose=init)
    torch.cuda.synchronize()
    d5 = max(_close(R5, torch.cat(oR), 1e-4, "R from the oracle's coarse pose"), _close(t5, torch.cat(ot), 1e-4, "t from the oracle's coarse pose"),
             _close(s5, torch.cat(os_), 1e-4, "score from the oracle's coarse pose"))
    print("\nconfig 2, B = 32: %d of 32 proposals make the oracle's coarse choice (end to end max diff %.2e); %d sampled indices of %d "
          "differ (all within 1e-6 of their threshold); %d asserted score ties, %d picks of a rank-deficient hypothesis (validated as "
          "minimisers); coarse attention max diff %.2e; fine stage given the GPU's pose %.2e, given the oracle's pose %.2e" % (
              n_e2e_same, worst["e2e"], n_idx_diff, B * 3 * n1, n_tie, n_degenerate, worst["att"],
                                                             worst["fine_given_pose"], d5))


# ------------------------------------------------------------------------------------------------------- config 4
def test_config4_sharded_equals_unsharded(dev, W):
    """200 proposals (8 objects x 25) dealt round-robin to 8 ranks (shard_indices), each shard through pem_match, rows gathered
    rank-major and put back with unshard: identical, bit for bit, to the 200 proposals in one call -- every kernel treats a proposal
    independently of its batch neighbours, so the sharded multi-GPU job returns exactly the single-GPU poses."""
    from sam6d_hip import parallel, pem, synth
    n_total, world = 200, 8
    inp = synth.config2_inputs(B=n_total, seed=4)
    d = {k: v.to(dev) for k, v in inp.items()}
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    R, t, s = pem.pem_match(*[d[k] for k in keys], W, d["rand"])
    whole = parallel.pack_poses(R, t, s)
    blocks = []
    for r in range(world):
        ids, n_valid = parallel.shard_indices(n_total, r, world)
        ids = ids.to(dev)
        assert n_valid == 25 and len(ids) == 25
        Rr, tr, sr = pem.pem_match(*[d[k][ids].contiguous() for k in keys], W, d["rand"][ids].contiguous())
        blocks.append(parallel.pack_poses(Rr, tr, sr))
    gathered = torch.cat(blocks, 0)  # what all_gather_into_tensor returns: rank-major
    back = parallel.unshard(gathered, n_total, world)
    torch.cuda.synchronize()
    assert torch.isfinite(back).all()
    nbad = int((back != whole).any(dim=1).sum())
    assert nbad == 0, "%d of %d proposals differ between the sharded and the unsharded run (max |d| %.3e)" % (
        nbad, n_total, float((back - whole).abs().max()))


def test_config4_ragged_shards(dev, W):
    """13 proposals over 4 ranks: the padded rows (last id repeated) are dropped by unshard."""
    from sam6d_hip import parallel, pem, synth
    n_total, world = 13, 4
    inp = synth.config2_inputs(B=n_total, seed=6)
    d = {k: v.to(dev) for k, v in inp.items()}
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    whole = parallel.pack_poses(*pem.pem_match(*[d[k] for k in keys], W, d["rand"]))
    blocks = []
    for r in range(world):
        ids, n_valid = parallel.shard_indices(n_total, r, world)
        ids = ids.to(dev)
        blocks.append(parallel.pack_poses(*pem.pem_match(*[d[k][ids].contiguous() for k in keys], W, d["rand"][ids].contiguous())))
    back = parallel.unshard(torch.cat(blocks, 0), n_total, world)
    assert torch.equal(back, whole)


# ------------------------------------------------------------------------------------ fine-stage labels (index work)
def _kat_atten(p1, p2, sharp=4.0, bg=-10.0):
    d = torch.cdist(p1, p2)
    att = torch.full((p1.shape[0], p1.shape[1] + 1, p2.shape[1] + 1), bg)
    att[:, 1:, 1:] = (1 - sharp * d).clamp(min=-1) / 0.1
    return att


@pytest.mark.parametrize("which", ["kat", "flat"])
def test_fine_stage_labels_and_weights_vs_oracle(dev, which):
    """soft assignment of the 2049 x 2049 fine attention (PEM/utils/model_utils.py:308-318): the argmax labels of both sides are index
    work (bit-exact against the oracle); assignment weights and weighted targets to 1e-5 (the label passes use the hardware exp / rcp
    instructions)."""
    from oracle import pem_oracle as O
    from sam6d_hip import _lib, pem
    g = golden("fine_rt")
    p1, p2, Rg, tg = _t(g["p1"]), _t(g["p2"]), _t(g["R_gt"]), _t(g["t_gt"])
    if which == "kat":
        att = _kat_atten((p1 - tg.unsqueeze(1)) @ Rg, p2, float(g["sharp"]), float(g["bg"]))
    else:
        att = torch.randn(1, 2049, 2049, generator=torch.Generator().manual_seed(int(g["att2_seed"]))) * float(g["att2_scale"])
    S, l1, l2 = O.soft_assignment(att)
    A = S[:, 1:, 1:] * (l1 > 0).float().unsqueeze(2) * (l2 > 0).float().unsqueeze(1)
    want_w = A.sum(2)
    want_pred = (A / (A.sum(2, keepdim=True) + 1e-6)) @ p2
    a = att.to(dev).contiguous()
    st = pem.soft_assign(a)
    n1 = int((st["l1"].cpu() != l1.to(torch.int32)).sum()); n2 = int((st["l2"].cpu() != l2.to(torch.int32)).sum())
    assert n1 == 0 and n2 == 0, "labels differ from the oracle: %d of %d rows, %d of %d columns" % (n1, l1.numel(), n2, l2.numel())
    B, R_, Cn = a.shape
    pred = torch.empty(B, R_ - 1, 3, device=dev)
    wgt = torch.empty(B, R_ - 1, device=dev)
    p2d = p2.to(dev).contiguous()  # (named: a temporary's memory could be reused before the launch)
    _lib.call("sam6d_fine_assign", a.data_ptr(), B, R_, Cn, st["rmax"].data_ptr(), st["rsum"].data_ptr(), st["cmax"].data_ptr(),
              st["csum"].data_ptr(), st["l1"].data_ptr(), st["l2"].data_ptr(), p2d.data_ptr(), pred.data_ptr(),
              wgt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _close(wgt, want_w, 1e-5, "assignment weights")
    _close(pred, want_pred, 1e-5, "weighted targets")


# ------------------------------------------------------------------------------------------- template-side de-duplication (8e)
def test_shared_template_equals_repeated_form(dev, W):
    """One object, 12 proposals: dense_po / dense_fo are one cloud `.repeat`ed per instance (PEM/run_inference_custom_pytorch.py:445-446).
    pem_match(shared_template=True) computes the template's dense tokens (in_proj, ball queries, PE MLPs) once and must return bit for
    bit what the repeated form returns; the detector recognises the repetition and rejects a single differing element."""
    from sam6d_hip import pem, synth
    B = 12
    inp = synth.config2_inputs(B=B, seed=9)
    d = {k: v.to(dev) for k, v in inp.items()}
    d["dense_po"] = d["dense_po"][:1].repeat(B, 1, 1).contiguous()
    d["dense_fo"] = d["dense_fo"][:1].repeat(B, 1, 1).contiguous()
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    assert pem.template_is_shared(d["dense_po"], d["dense_fo"])
    a = pem.pem_match(*[d[k] for k in keys], W, d["rand"])
    b = pem.pem_match(*[d[k] for k in keys], W, d["rand"], shared_template=True)
    for x, y, what in zip(a, b, ("R", "t", "score")):
        assert torch.equal(x, y), "%s differs between the repeated and the shared-template form: %.3e" % (what, float((x - y).abs().max()))
    other = d["dense_fo"].clone()
    other[B - 1, 2047, 255] += 1.0
    assert not pem.template_is_shared(d["dense_po"], other)
    assert not pem.template_is_shared(d["dense_po"][:1], d["dense_fo"][:1])


# ------------------------------------------------------------------------------------------- config 5: fp16 single-product mode
def test_config5_fp16_single_product_mode(dev, sd):
    """BASELINE.json config 5: the GeometricTransformer contractions with ONE fp16 MFMA product (matmul mode 2: no lo halves, fp32
    accumulate, fp32 geometry).  Not the 1e-4 contract: the deviation is MEASURED -- GEMM ~3e-4 of the result scale; poses on the
    known-answer scene of tests/golden/pem_e2e.npz against the REFERENCE's outputs -- and a 4096-point fine stage runs in that mode."""
    from sam6d_hip import _lib, pem, synth
    W = pem.PemWeights(sd, dev)
    g = torch.Generator().manual_seed(55)
    A = torch.randn(900, 256, generator=g).to(dev)
    Wt = (torch.randn(256, 256, generator=g) / 16).to(dev)
    want = A.double().cpu() @ Wt.double().cpu().t()
    gold = golden("pem_e2e")
    kat = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.kat_inputs(B=2, seed=int(gold["kat_seed"])).items()}
    big = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.kat_inputs(B=2, seed=5, n_dense=4096).items()}
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    prev = _lib.load().sam6d_get_matmul_mode()
    res = {}
    try:
        for mode in (1, 2):
            _lib.call("sam6d_set_thread_matmul_mode", mode)
            out = torch.empty(900, 256, device=dev)
            pem.gemm(A, Wt, None, out, 900, 256, 256, 256, 256, 256)
            gerr = float((out.cpu().double() - want).abs().max()) / float(want.abs().max())
            R, t, s = pem.pem_match(*[kat[k] for k in keys], W, kat["rand"])
            Rb, tb, sb = pem.pem_match(*[big[k] for k in keys], W, big["rand"])
            torch.cuda.synchronize()
            res[mode] = (gerr, R.cpu(), t.cpu(), s.cpu(), Rb.cpu(), tb.cpu(), sb.cpu())
    finally:
        _lib.call("sam6d_set_thread_matmul_mode", -1)
    g1 = res[1][0]
    g2, R2, t2, s2, Rb2, tb2, sb2 = res[2]
    dR = float((R2 - _t(gold["kat_R"])).abs().max()); dt = float((t2 - _t(gold["kat_t"])).abs().max())
    ds = float((s2 - _t(gold["kat_score"])).abs().max())
    dRb = float((Rb2 - res[1][4]).abs().max()); dtb = float((tb2 - res[1][5]).abs().max())
    print("\nconfig 5 (fp16 single product): GEMM rel. err %.2e (split: %.2e); known-answer scene vs the reference: dR %.2e dt %.2e "
          "dscore %.2e; 4096-point fine stage vs the split mode: dR %.2e dt %.2e" % (g2, g1, dR, dt, ds, dRb, dtb))
    assert g1 < 4e-6 and 1e-5 < g2 < 5e-3, "mode 2 must really drop the lo halves (GEMM error %.2e)" % g2
    # The pose deviations are REPORTED, not bounded.  Measured on MI355X (DESIGN "Mode 2"): with random-init weights the matching of
    # the known-answer scene rests on corresponding points producing identical features; a 3e-4 relative perturbation of the
    # transformer products moves the final rotation by up to 0.5 (entries of R) while translation and score move by < 1e-2
    # (SAM6D_HALF_MASK bisect: generic GEMMs alone 0.46, block kernels alone 0.52, cross attention alone 0.08, fine similarity alone
    # 2e-5).  What IS required: finite outputs and proper rotations.
    for Rm, tm, sm in ((R2, t2, s2), (Rb2, tb2, sb2)):
        assert torch.isfinite(Rm).all() and torch.isfinite(tm).all() and torch.isfinite(sm).all()
        assert float((torch.linalg.det(Rm.double()) - 1).abs().max()) < 1e-4
    assert dt < 5e-2 and ds < 1e-1


def test_template_ids_multi_object_batch_equals_repeated_form(dev, W):
    """BASELINE config 4's batch shape: proposals of 8 objects mixed in one batch (PEM/provider/bop_test_dataset.py:107,156 returns an
    `obj` index per instance).  pem_match(template_ids=...) takes the 8 UNIQUE templates, runs the template-side pose-independent work
    once per template and must return bit for bit what the per-instance repeated form returns -- 8 objects x 3 proposals here, with an
    uneven assignment (one object without a proposal, one with five) and the ids out of order; also through the drop-in Net.match."""
    from sam6d_hip import pem, synth
    B, T = 24, 8
    inp = synth.config4_inputs(B=B, n_obj=T, seed=14)
    ids = inp["template_ids"].clone()
    ids[ids == 7] = 2          # object 7 has no proposal, object 2 twice as many
    ids = ids[torch.randperm(B, generator=torch.Generator().manual_seed(3))]
    inp["template_ids"] = ids
    inp["model"] = (torch.rand(T, 1024, 3, generator=torch.Generator().manual_seed(5)) - 0.5)[ids].contiguous()
    d = {k: v.to(dev) for k, v in inp.items()}
    rep = synth.repeated(d)
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    a = pem.pem_match(*[rep[k] for k in keys], W, rep["rand"])
    b = pem.pem_match(*[d[k] for k in keys], W, d["rand"], template_ids=d["template_ids"])
    for x, y, what in zip(a, b, ("R", "t", "score")):
        assert torch.equal(x, y), "%s differs between the repeated form and template_ids: %.3e" % (what, float((x - y).abs().max()))
    # without the side stream (the static fine work inline) and with aux
    c = pem.pem_match(*[d[k] for k in keys], W, d["rand"], template_ids=d["template_ids"], cfg=dict(pem.DEFAULT_CFG, overlap=False))
    e = pem.pem_match(*[d[k] for k in keys], W, d["rand"], template_ids=d["template_ids"], return_aux=True)
    for x, y, z in zip(a, c, e):
        assert torch.equal(x, y) and torch.equal(x, z)
    with pytest.raises(ValueError):
        pem.pem_match(*[d[k] for k in keys], W, d["rand"], template_ids=d["template_ids"] + T)
    with pytest.raises(ValueError):
        pem.pem_match(*[d[k] for k in keys], W, d["rand"], template_ids=d["template_ids"][:-1])


def test_config4_200_proposals_8_objects_template_ids(dev, W):
    """Config 4 at full size on one GPU: 200 proposals of 8 objects in ONE call with template_ids; equal (bitwise) to the same scene run
    as 8 round-robin shards with their own template_ids (what 8 ranks would compute), and to the repeated form on a 25-proposal shard."""
    from sam6d_hip import pem, synth
    from sam6d_hip.parallel import shard_indices
    inp = synth.config4_inputs(B=200, n_obj=8, seed=4)
    d = {k: v.to(dev) for k, v in inp.items()}
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    whole = pem.pem_match(*[d[k] for k in keys], W, d["rand"], template_ids=d["template_ids"])
    assert all(torch.isfinite(x).all() for x in whole)
    R = torch.empty_like(whole[0]); t = torch.empty_like(whole[1]); s = torch.empty_like(whole[2])
    for r in range(8):
        sl = shard_indices(200, r, 8)[0].to(dev)
        part = pem.pem_match(d["dense_pm"][sl].contiguous(), d["dense_fm"][sl].contiguous(), d["dense_po"], d["dense_fo"],
                             d["radius"][sl].contiguous(), d["model"][sl].contiguous(), W, d["rand"][sl].contiguous(),
                             template_ids=d["template_ids"][sl])
        R[sl], t[sl], s[sl] = part
        if r == 3:
            rep = pem.pem_match(d["dense_pm"][sl].contiguous(), d["dense_fm"][sl].contiguous(),
                                d["dense_po"][d["template_ids"][sl]].contiguous(), d["dense_fo"][d["template_ids"][sl]].contiguous(),
                                d["radius"][sl].contiguous(), d["model"][sl].contiguous(), W, d["rand"][sl].contiguous())
            for x, y in zip(part, rep):
                assert torch.equal(x, y)
    assert torch.equal(R, whole[0]) and torch.equal(t, whole[1]) and torch.equal(s, whole[2])


# ------------------------------------------------------------------------------------------- weight-guarded fallback routes, end to end
@pytest.mark.parametrize("case", ["fused_guard_no", "image_guard_no", "three_product_stage1"])
def test_weight_guard_fallback_routes_end_to_end_vs_oracle(dev, sd, case):
    """The default fast paths are chosen per WEIGHT SET on the host (fused-RPE range guard, fp16-image guard, two-product stage 1): the
    released checkpoint -- not available offline -- may send any of them to its "no" branch.  Each branch is forced here and checked END
    TO END against the oracle with the staged contract of test_config2_full_batch_vs_oracle (test_rpe_fused_range_guard checks the
    embedding alone): (1) FPS indices bit-exact and the COARSE ATTENTION -- the output of three geometric-transformer blocks through the
    routed kernels -- within 1e-4; (2) the fine stage started from the ORACLE's coarse pose (pem_match(init_pose=...)) gives the
    oracle's final R, t, score within 1e-4 (the coarse pose itself is a discrete choice on random features, proved link by link there).
    fused / image guard: proj_d and proj_a (weights and biases) are scaled by s and every proj_p weight by 1 / s -- the same function in
    exact arithmetic, but the projected embedding leaves the fp16 range of the fused score kernel (s = 20: materialised embedding +
    attention_kernel<RPE>) or of the split-precision embedding images altogether (s = 4000: the exact fp32 embedding kernel).
    three_product_stage1: sigma_a = 7.5 doubles the angular index range, for which the host check refuses the two-product form."""
    from oracle import pem_oracle as O
    from sam6d_hip import _lib, pem, synth
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("the guards choose between routes of the default (fp16x3) arithmetic")
    sd2 = dict(sd)
    cfg, ocfg = dict(pem.DEFAULT_CFG), dict(O.DEFAULT_CFG)
    if case == "three_product_stage1":
        cfg["sigma_a"] = ocfg["sigma_a"] = 7.5
    else:
        s_ = 20.0 if case == "fused_guard_no" else 4000.0
        for k in sd:
            if k.startswith("geo_embedding.proj_"):
                sd2[k] = sd[k] * s_
            elif k.endswith("attention.proj_p.weight"):
                sd2[k] = sd[k] / s_
    W2 = pem.PemWeights(sd2, dev, options=pem.Options())  # the default routes, whatever A/B switches the environment carries
    paths = pem.describe_paths(W2, cfg)
    if case == "three_product_stage1":
        assert paths["rpe_stage1_products"] == 3, paths
    elif case == "fused_guard_no":
        assert paths["rpe_stage1_products"] is None and paths["fused_rpe_guard_passed"] is False and "geo_cheb" in paths["embedding_rows"], paths
    else:
        assert paths["fused_rpe_guard_passed"] is False and "exact fp32" in paths["embedding_rows"], paths
    B = 2
    # (seed 1: the benchmark's generator.  Seed 23 was tried first and has, in proposal 1, two scene points at bitwise the same distance
    #  from a third: torch.topk's tie order is unspecified and differs from the lowest-index rule here, which moves that proposal's
    #  attention by 1e-2 on EVERY route, the default one included -- scratch/dbg_fallback.py)
    inp = synth.config2_inputs(B=B, seed=1)
    d = {k: v.to(dev) for k, v in inp.items()}
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    oR0, ot0, oR, ot, os_ = [], [], [], [], []
    R, t, s, aux = pem.pem_match(*[d[k] for k in keys], W2, d["rand"], cfg=cfg, return_aux=True)
    assert torch.isfinite(R).all() and torch.isfinite(t).all() and torch.isfinite(s).all()
    worst = 0.0
    with torch.no_grad():
        for b in range(B):
            o = _oracle_proposal(O, inp, b, sd2, ocfg)
            assert torch.equal(aux["fps_idx_m"][b:b + 1].cpu(), o["im"]) and torch.equal(aux["fps_idx_o"][b:b + 1].cpu(), o["io"])
            da = _d(aux["coarse"]["atten"][b:b + 1], o["coarse"]["atten"])
            worst = max(worst, da)
            assert da <= 1e-4, "%s, proposal %d: coarse attention off by %.2e" % (case, b, da)
            oR0.append(o["R0"]); ot0.append(o["t0"]); oR.append(o["R"]); ot.append(o["t"]); os_.append(o["s"])
    R2, t2, s2 = pem.pem_match(*[d[k] for k in keys], W2, d["rand"], cfg=cfg,
                               init_pose=(torch.cat(oR0).to(dev), torch.cat(ot0).to(dev)))
    print("\n%s: coarse attention %.2e; fine stage from the oracle's coarse pose: dR %.2e dt %.2e ds %.2e"
          % (case, worst, _d(R2, torch.cat(oR)), _d(t2, torch.cat(ot)), _d(s2, torch.cat(os_))))
    _close(R2, torch.cat(oR), 1e-4, case + ": R"); _close(t2, torch.cat(ot), 1e-4, case + ": t"); _close(s2, torch.cat(os_), 1e-4, case + ": score")


# ------------------------------------------------------------------------------------------------------------ hipGraph replay
def test_pem_graph_replay_equals_eager(dev, W):
    """pem.PemGraph: one pem_match launch sequence captured into a hipGraph.  A replay on new inputs (copied into the graph's static
    buffers) is bit for bit the eager call -- plain batch, micro-batch slices on concurrent streams of the graph (flat forks: the only
    shape the capture produces), and a multi-object batch with template_ids fixed at capture."""
    from sam6d_hip import pem, synth
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    a = {k: v.to(dev) for k, v in synth.config2_inputs(B=16, seed=31).items()}
    b = {k: v.to(dev) for k, v in synth.config2_inputs(B=16, seed=32).items()}
    want_a = [o.clone() for o in pem.pem_match(*[a[k] for k in keys], W, a["rand"])]
    want_b = [o.clone() for o in pem.pem_match(*[b[k] for k in keys], W, b["rand"])]
    for mb in (1, 2):
        g = pem.PemGraph(W, *[a[k] for k in keys], a["rand"], microbatch=mb)
        out = g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(out, want_a)), "replay on the capture inputs, microbatch %d" % mb
        out = g(*[b[k] for k in keys], b["rand"])
        torch.cuda.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(out, want_b)), "replay on new inputs, microbatch %d" % mb
        with pytest.raises(ValueError):
            g(*[b[k][:8] for k in keys], b["rand"][:8])
        del g
    c = {k: v.to(dev) for k, v in synth.config4_inputs(B=12, n_obj=4, seed=33).items()}
    want_c = [o.clone() for o in pem.pem_match(*[c[k] for k in keys], W, c["rand"], template_ids=c["template_ids"])]
    g = pem.PemGraph(W, *[c[k] for k in keys], c["rand"], template_ids=c["template_ids"])
    out = g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(out, want_c))
    with pytest.raises(ValueError):
        pem.PemGraph(W, *[c[k] for k in keys], c["rand"], template_ids=c["template_ids"] + 4)
