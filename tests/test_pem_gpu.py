"""GPU parity of the PEM matching path: every HIP stage (through the C ABI, via sam6d_hip.pem) against golden vectors
captured from the reference (tests/golden/) and against the CPU oracle on seeded inputs.

Bars: index/integer outputs bit-exact; fp32 outputs within the tolerance written in each test (north_star: poses and
scores within 1e-4; intermediate activations are O(1) LayerNorm outputs compared at 1e-4 or tighter)."""
import math

import numpy as np
import pytest
import torch

from tests._util import golden

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def sd():
    from sam6d_hip import synth
    return synth.make_pem_weights(1)


@pytest.fixture(scope="module")
def W(sd, dev):
    from sam6d_hip import pem
    return pem.PemWeights(sd, dev)


def _close(got, want, atol, what=""):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what + ": non-finite values"
    d = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    assert d <= atol, "%s: max abs diff %.3e > %.1e" % (what, d, atol)
    return d


# ------------------------------------------------------------------------------------------------------ GEMM / LN
@pytest.mark.parametrize("M,N,K,act,res", [(197 * 3, 256, 256, 0, True), (1000, 512, 256, 1, False), (300, 256, 512, 0, True),
                                            (4096, 768, 256, 0, False), (129, 130, 64, 0, False), (77, 32, 6, 1, False)])
def test_gemm_nt_vs_fp64(dev, M, N, K, act, res):
    from sam6d_hip import pem
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    want = A.double() @ Wt.double().t() + b.double()
    if act:
        want = want.clamp(min=0)
    if res:
        want = want + r.double()
    out = torch.empty(M, N, device=dev)
    pem.gemm(A.to(dev), Wt.to(dev), b.to(dev), out, M, N, K, K, K, N, residual=r.to(dev) if res else None, ldr=N, act=act)
    _close(out, want.float(), 2e-5 * max(1.0, math.sqrt(K) / 4), "gemm")


def test_gemm_batched_divisor_colscale(dev):
    from sam6d_hip import pem
    g = torch.Generator().manual_seed(9)
    B, n, K = 3, 197, 256
    f = torch.randn(2 * B, n, K, generator=g)
    out = torch.empty(B, n, n, device=dev)
    fd = f.to(dev)
    pem.gemm(fd, fd, None, out, n, n, K, K, K, n, w_off=B * n * K, batch=B, sA=n * K, sW=n * K, sC=n * n, divisor=0.1)
    want = (f[:B].double() @ f[B:].double().transpose(1, 2)) / 0.1
    _close(out, want.float(), 2e-3, "batched similarity gemm")  # values ~ +-500
    A = torch.randn(500, 32, generator=g); Wt = torch.randn(64, 32, generator=g)
    sc = torch.rand(64, generator=g) + 0.5; sh = torch.randn(64, generator=g)
    out2 = torch.empty(500, 64, device=dev)
    pem.gemm(A.to(dev), Wt.to(dev), sh.to(dev), out2, 500, 64, 32, 32, 32, 64, colscale=sc.to(dev), act=1)
    want2 = ((A.double() @ Wt.double().t()) * sc.double() + sh.double()).clamp(min=0)
    _close(out2, want2.float(), 5e-5, "colscale gemm")


def test_layernorm(dev):
    from sam6d_hip import pem
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1001, 256, generator=g) * 3 + 1
    w = torch.randn(256, generator=g); b = torch.randn(256, generator=g)
    want = torch.nn.functional.layer_norm(x.double(), (256,), w.double(), b.double(), 1e-5)
    got = pem.layernorm(x.to(dev), (w.to(dev), b.to(dev)))
    _close(got, want.float(), 2e-5, "layernorm")


# ------------------------------------------------------------------------------------------------ geo embedding
def test_geo_embedding_golden(dev, W, sd):
    from sam6d_hip import pem
    from oracle import pem_oracle as O
    g = golden("geo_embedding")
    pts = _t(g["pts"])
    out = pem.geo_embedding(pts.to(dev), W)
    rows = g["rows"]
    # 6e-5: pairs with the bg point have d_idx ~ 866 where one fp32 ulp is 6.1e-5, and torch's vectorised CPU sqrt is
    # not correctly rounded on 0.6 % of the entries (see test_geo_indices_and_knn); everything else agrees to ~8e-6
    _close(out[:, rows], g["out_rows"], 6e-5, "geo embedding vs reference rows")
    _close(out[:, rows][:, 1:], g["out_rows"][:, 1:], 2e-5, "geo embedding vs reference rows (non-bg centres)")
    full = O.geo_embedding(pts, sd)
    _close(out, full, 1e-4, "geo embedding vs oracle (all pairs, bg pairs included)")
    _close(out[:, 1:, 1:], full[:, 1:, 1:], 2e-5, "geo embedding vs oracle (pairs without the bg point)")


def test_geo_indices_and_knn(dev, W):
    """kNN indices bit-exact; d_idx exact; angular indices to libm tolerance."""
    from sam6d_hip import _lib, pem
    g = golden("geo_embedding")
    pts = _t(g["pts"]).to(dev)
    B, n, _ = pts.shape
    knn = torch.empty(B * n * 3 + 1, dtype=torch.int32, device=dev)
    idx = torch.empty(B, n, n, 4, device=dev)
    out = torch.empty(B, n, n, 256, device=dev)
    _lib.call("sam6d_geo_embedding", pts.data_ptr(), B, n, W.div_term.data_ptr(), W.geo_d.w.data_ptr(), W.geo_d.b.data_ptr(),
              W.geo_a.w.data_ptr(), W.geo_a.b.data_ptr(), 0.2, 180.0 / (15 * math.pi), 3, 256, knn.data_ptr(), idx.data_ptr(),
              out.data_ptr(), pem._s())
    assert int(knn[-1]) == 0, "range flag must stay clear for normalised clouds"
    knn = knn[:-1].reshape(B, n, 3)
    assert np.array_equal(knn.cpu().numpy(), g["knn"].astype(np.int32))
    # d_idx = sqrt(pd) / 0.2: the device result is the correctly rounded one (checked here in fp64); the reference's
    # torch-CPU sqrt is 1 ulp low on ~0.6 % of the entries, so the golden tensor is matched to <= 1 ulp, >= 99 % exact
    from oracle import pem_oracle as O
    pd = O.pairwise_distance(_t(g["pts"]), _t(g["pts"])).numpy()
    cr = (np.sqrt(pd.astype(np.float64)).astype(np.float32).astype(np.float64) / np.float64(np.float32(0.2))).astype(np.float32)
    d = idx[..., 0].cpu().numpy()
    assert np.array_equal(d, cr), "d_idx must be the correctly rounded sqrt and quotient of the bit-exact squared distance"
    ulp = np.abs(d.view(np.int32).astype(np.int64) - g["d_idx"].view(np.int32).astype(np.int64))
    assert ulp.max() <= 2 and (ulp == 0).mean() > 0.99
    _close(idx[..., 1:], g["a_idx"], 5e-6, "a_idx")


# -------------------------------------------------------------------------------------------------- transformer
def _layer_inputs(seed, B=1, n=197):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, n, 256, generator=gen)
    y = torch.randn(B, n, 256, generator=gen)
    e0 = 0.5 * torch.randn(B, n, n, 256, generator=gen)
    e1 = 0.5 * torch.randn(B, n, n, 256, generator=gen)
    return x, y, e0, e1


def test_transformer_layers_golden(dev, W):
    from sam6d_hip import pem
    g = golden("transformer")
    x, y, e0, e1 = _layer_inputs(int(g["seed"]))
    T = W.coarse["blocks"][0]
    rpe = pem.rpe_self_layer(x.to(dev), e0.to(dev), T["self"])
    _close(rpe, g["rpe"], 1e-4, "RPE self layer")
    crs = pem.cross_layer(x.to(dev), y.to(dev), T["cross"])
    _close(crs, g["cross"], 1e-4, "cross layer")
    S = torch.cat([x, y], 0).to(dev)
    E = torch.cat([e0, e1], 0).to(dev)
    out = pem.geometric_transformer(S, E, T)
    _close(out[0:1], g["f0"], 1e-4, "geometric transformer f0")
    _close(out[1:2], g["f1"], 1e-4, "geometric transformer f1")


def test_sparse_to_dense_golden(dev, W, sd):
    from sam6d_hip import pem
    g = golden("sparse_to_dense")
    gen = torch.Generator().manual_seed(int(g["seed_dense"]))
    d0 = torch.randn(1, 2049, 256, generator=gen)
    d1 = torch.randn(1, 2049, 256, generator=gen)
    _, _, e0, e1 = _layer_inputs(int(g["seed_emb"]))
    T = W.fine["blocks"][0]
    D = torch.cat([d0, d1], 0).to(dev)
    E = torch.cat([e0, e1], 0).to(dev)
    idx = torch.cat([_t(g["idx0"]), _t(g["idx1"])], 0).to(dev)
    # dense layer alone: dense tokens d0[:,1:], memory d1[:,1:197] (row 0 of both buffers is the bg slot)
    Dd = d0.to(dev).contiguous()
    Sm = d1[:, :197].to(dev).contiguous()
    lin = pem.linear_transformer_layer(Dd, Sm, T["dense"])
    _close(lin[:, 1:][:, ::8], g["lin_rows"], 1e-4, "linear transformer layer")
    out = pem.sparse_to_dense_transformer(D, E, idx, T)
    _close(out[0:1, ::8], g["out0_rows"], 2e-4, "s2d out0")
    _close(out[1:2, ::8], g["out1_rows"], 2e-4, "s2d out1")
    _close(out[0:1, :4], g["out0_head"], 2e-4, "s2d out0 head rows (bg token + first points)")


def test_positional_encoding_golden(dev, W):
    from sam6d_hip import pem
    g = golden("pos_encoding")
    pts = _t(g["pts"]).to(dev)
    B, N, _ = pts.shape
    dst = torch.zeros(B, N + 1, 256, device=dev)
    pem.positional_encoding_add(pts, W, dst, 256, (N + 1) * 256)
    _close(dst[:, 1:][:, ::8], g["out_rows"], 5e-5, "positional encoding")
    assert float(dst[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("B,N,S", [(3, 701, 32), (2, 1500, 64), (1, 5, 32), (40, 2048, 64)])
def test_pe_mlp_max_register_chained_vs_exact_and_fp64(dev, W, B, N, S):
    """The default (fp16x3, register-chained, persistent + prefetching) kernel against the exact-fp32 kernel and a
    float64 recompute of QueryAndGroup -> SharedMLP -> max (fine_point_matching.py:126-139) on the same indices; ragged
    point counts exercise the persistent loop's tail, (40, 2048, 64) more than one pass of the 768 workgroups."""
    from sam6d_hip import _lib, pem
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("default (fp16x3) mode only")
    gen = torch.Generator().manual_seed(B * 1000 + N + S)
    pts = torch.rand(B, N, 3, generator=gen) - 0.5
    idx = torch.randint(0, N, (B, N, S), generator=gen, dtype=torch.int32)
    idx[:, :, 0] = torch.arange(N, dtype=torch.int32)[None]
    L = W.pe["mlp"][0 if S == 32 else 1]

    pts_d, idx_d = pts.to(dev), idx.to(dev)  # (named: a temporary's memory could be reused before the launch)

    def run():
        out = torch.full((B * N, 256), -7.0, device=dev)
        _lib.call("sam6d_pe_mlp_max", pts_d.data_ptr(), idx_d.data_ptr(), B, N, S, L[0]["w"].data_ptr(),
                  L[0]["scale"].data_ptr(), L[0]["shift"].data_ptr(), L[1]["w"].data_ptr(), L[1]["scale"].data_ptr(),
                  L[1]["shift"].data_ptr(), L[2]["w"].data_ptr(), L[2]["scale"].data_ptr(), L[2]["shift"].data_ptr(),
                  out.data_ptr(), 256, 128, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return out.cpu()

    got = run()
    try:
        _lib.call("sam6d_set_thread_matmul_mode", 0)
        exact = run()
    finally:
        _lib.call("sam6d_set_thread_matmul_mode", -1)
    assert float((got[:, :128] + 7.0).abs().max()) == 0.0, "columns outside [off, off+128) were touched"
    if B * N * S <= 4_000_000:
        nb = torch.gather(pts.double()[:, None].expand(B, N, N, 3), 2, idx.long()[..., None].expand(B, N, S, 3))
        h = torch.cat([nb - (pts.double()[:, :, None] + 1e-8), nb], -1)
        for l in L:
            h = (h @ l["w"].double().cpu().t() * l["scale"].double().cpu() + l["shift"].double().cpu()).clamp(min=0)
        want = h.max(2).values.reshape(B * N, 128).float()
        _close(exact[:, 128:], want, 2e-5, "exact pe_mlp_max vs float64")
        _close(got[:, 128:], want, 2e-5, "register-chained pe_mlp_max vs float64")
    d = float((got - exact).abs().max())
    print("\npe_mlp_max fp16x3 vs exact, B %d N %d S %d: max abs diff %.2e (scale %.2f)" % (B, N, S, d, float(exact.abs().max())))
    assert d < 3e-6 * max(1.0, float(exact.abs().max()))


@pytest.mark.parametrize("extent,offset", [(1e-2, 0.0), (3e-4, 0.0), (1.0, 8.0), (1500.0, 0.0), (8000.0, 2000.0)])
def test_pe_mlp_max_coordinate_range(dev, W, extent, offset):
    """layer 1 of the PE MLP splits the raw coordinates into fp16 hi / lo halves.  Unscaled, a cloud of extent ~1e-2 had
    subnormal lo halves (an absolute error floor), and millimetre-unit clouds beyond 65504 overflowed to inf.  With the per-neighbour
    power-of-two scale layer 1 is accurate relative to the features' magnitude for any extent: checked against a float64 recompute from
    3e-4 to 1e4 coordinate units.  (The hidden activations of layers 2 / 3 are split unscaled, in this kernel as in round 3's: they must
    stay below 65504, i.e. coordinates below ~1e4 / |W| -- PEM applies the encoding to radius-normalised clouds, |x| < 10:
    include/sam6d_hip.h states the range.)"""
    from sam6d_hip import _lib
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("default (fp16x3) mode only")
    B, N, S = 2, 600, 32
    gen = torch.Generator().manual_seed(int(extent * 1000) % 1000 + 17)
    pts = (torch.rand(B, N, 3, generator=gen) - 0.5) * extent + offset
    idx = torch.randint(0, N, (B, N, S), generator=gen, dtype=torch.int32)
    idx[:, :, 0] = torch.arange(N, dtype=torch.int32)[None]
    L = W.pe["mlp"][0]
    pts_d, idx_d = pts.to(dev), idx.to(dev)
    out = torch.zeros((B * N, 128), device=dev)
    _lib.call("sam6d_pe_mlp_max", pts_d.data_ptr(), idx_d.data_ptr(), B, N, S, L[0]["w"].data_ptr(), L[0]["scale"].data_ptr(),
              L[0]["shift"].data_ptr(), L[1]["w"].data_ptr(), L[1]["scale"].data_ptr(), L[1]["shift"].data_ptr(), L[2]["w"].data_ptr(),
              L[2]["scale"].data_ptr(), L[2]["shift"].data_ptr(), out.data_ptr(), 128, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    nb = torch.gather(pts.double()[:, None].expand(B, N, N, 3), 2, idx.long()[..., None].expand(B, N, S, 3))
    # the kernel (like the reference: fine_point_matching.py:117) forms new_xyz = pts + 1e-8 in fp32 first
    q = (pts + 0.00000001).double()
    h = torch.cat([nb - q[:, :, None], nb], -1)
    for l in L:
        h = (h @ l["w"].double().cpu().t() * l["scale"].double().cpu() + l["shift"].double().cpu()).clamp(min=0)
    want = h.max(2).values.reshape(B * N, 128)
    got = out.cpu().double()
    assert torch.isfinite(got).all(), "non-finite PE features at extent %g" % extent
    # relative to the output scale, never below 1: the hidden activations of layers 2 / 3 are split unscaled (their magnitude is set by
    # the BN shifts, O(0.1 .. 1) for any trained network), which leaves an absolute floor of ~1e-7 -- fp32 resolution at that scale
    scale = max(float(want.abs().max()), 1.0)
    err = float((got - want).abs().max())
    print("\npe_mlp_max extent %g offset %g: max abs err %.2e of scale %.3g" % (extent, offset, err, scale))
    assert err <= 3e-6 * scale, "extent %g: %.3e of %.3e" % (extent, err, scale)


def test_feature_similarity_golden(dev):
    from sam6d_hip import pem
    g = golden("similarity")
    gen = torch.Generator().manual_seed(int(g["seed"]))
    a = torch.randn(2, 197, 256, generator=gen)
    b = torch.randn(2, 197, 256, generator=gen)
    ident = pem.Linear(torch.eye(256, device=dev), torch.zeros(256, device=dev))
    att = pem.feature_similarity(torch.cat([a, b], 0).to(dev), 2, 197, ident, 0.1)
    _close(att, g["out"], 2e-5, "feature similarity")


# ------------------------------------------------------------------------------------------------------- poses
def test_coarse_rt_golden(dev):
    """Known-answer scene: everything, including every sampled index, matches the reference and the ground truth."""
    from sam6d_hip import pem
    g = golden("coarse_rt")
    p1, p2, model, u = (_t(g[k]).to(dev) for k in ("p1", "p2", "model", "u"))
    radius = torch.ones(2, device=dev)
    att = _t(g["att"]).to(dev)
    R, t, aux = pem.compute_coarse_Rt(att, p1, p2, model, radius, u, return_aux=True)
    assert np.array_equal(aux["w1"].cpu().numpy(), g["w1"]), "foreground mask"
    assert np.array_equal(aux["idx"].cpu().numpy(), g["idx"]), "sampled hypothesis indices (bit-exact)"
    # (the top-300 SET is not compared here: thousands of exact-match hypotheses have residual ~1e-7 = rounding noise)
    _close(aux["dis"], g["dis"], 1e-5, "hypothesis residuals")
    _close(R, g["R"], 1e-4, "coarse R"); _close(t, g["t"], 1e-4, "coarse t")
    _close(R, g["R_gt"], 1e-4, "coarse KAT R vs ground truth"); _close(t, g["t_gt"], 1e-4, "coarse KAT t vs ground truth")


def test_coarse_rt_flat_attention(dev):
    """Structure-less attention (what random-init weights produce).  Here ~10 % of the sampled triples repeat a point
    (rank-1 correlation matrix): the reference's rotation for those is whatever LAPACK's sgesdd makes of rounding noise
    (SURVEY 7 'hard parts'), so they are compared for finiteness only; every well-posed hypothesis must agree."""
    from sam6d_hip import pem
    from oracle import pem_oracle as O
    g = golden("coarse_rt")
    cpu = {k: _t(g[k]) for k in ("p1", "p2", "model", "u", "att2")}
    p1, p2, model, u, att = (cpu[k].to(dev) for k in ("p1", "p2", "model", "u", "att2"))
    radius = torch.ones(2, device=dev)
    R, t, aux = pem.compute_coarse_Rt(att, p1, p2, model, radius, u, return_aux=True)
    assert torch.isfinite(R).all() and torch.isfinite(t).all() and torch.isfinite(aux["Rs"]).all()
    assert np.array_equal(aux["w1"].cpu().numpy(), g["w1_2"])
    ref_idx = _t(g["idx2"]).long()
    gidx = aux["idx"].cpu().long()
    same_frac = (gidx == ref_idx).float().mean().item()
    assert same_frac >= 0.995, "sampled indices agree on %.5f (expf/powf ulp differences move a few thresholds)" % same_frac
    Rs_o, ts_o, dis_o = O.coarse_hypotheses(ref_idx, cpu["p1"], cpu["p2"], 6000)
    tri1 = (ref_idx // 196).reshape(2, 6000, 3); tri2 = (ref_idx % 196).reshape(2, 6000, 3)
    distinct = lambda x: (x[..., 0] != x[..., 1]) & (x[..., 0] != x[..., 2]) & (x[..., 1] != x[..., 2])
    same = (gidx == ref_idx).reshape(2, 6000, 3).all(-1)
    wellposed = distinct(tri1) & distinct(tri2) & same
    # conditioning of the remaining 3-point problems (fp64): drop near-collinear triples
    a = torch.gather(cpu["p2"], 1, (ref_idx % 196).unsqueeze(2).expand(2, 18000, 3)).reshape(2, 6000, 3, 3).double()
    bq = torch.gather(cpu["p1"], 1, (ref_idx // 196).unsqueeze(2).expand(2, 18000, 3)).reshape(2, 6000, 3, 3).double()
    Hm = (a - a.mean(2, keepdim=True)).transpose(2, 3) @ (bq - bq.mean(2, keepdim=True))
    sv = torch.linalg.svdvals(Hm)
    wellposed &= (sv[..., 1] / sv[..., 0]) > 0.05
    assert wellposed.float().mean() > 0.5
    Rs = aux["Rs"].cpu().reshape(2, 6000, 3, 3); ts = aux["ts"].cpu().reshape(2, 6000, 1, 3)
    _close(Rs[wellposed], Rs_o[wellposed], 1e-4, "well-posed hypothesis rotations")
    _close(ts[wellposed], ts_o[wellposed], 1e-4, "well-posed hypothesis translations")
    _close(aux["dis"].cpu()[same], dis_o[same], 2e-6, "hypothesis residuals (all hypotheses with identical samples)")
    # scores of well-posed hypotheses present in both top-300 sets
    for b in range(2):
        sg = dict(zip(g["top2"][b].tolist(), g["scores2"][b].tolist()))
        sd_ = dict(zip(aux["top"][b].tolist(), aux["scores"][b].tolist()))
        common = [h for h in sg if h in sd_ and bool(wellposed[b, h])]
        assert len(common) > 100
        rel = max(abs(sg[h] - sd_[h]) / sg[h] for h in common)
        assert rel < 1e-4, "hypothesis scores, rel diff %.2e" % rel
        assert len(set(sg) & set(sd_)) >= 295, "top-300 sets"


def test_weighted_sample_bit_exact(dev):
    """cumsum (double accumulate) + normalise + first-ge search: bit-exact vs the oracle on injected weights."""
    from sam6d_hip import _lib, pem
    from oracle import pem_oracle as O
    gen = torch.Generator().manual_seed(4)
    w = torch.rand(3, 38416, generator=gen) ** 6
    w[1] *= (torch.rand(38416, generator=gen) > 0.97)
    w[2] = 0  # all-false row -> index 0
    u = torch.rand(3, 18000, generator=gen)
    want = O.weighted_sampling(w, u)
    cum = torch.empty(3, 38416, device=dev)
    idx = torch.empty(3, 18000, dtype=torch.int32, device=dev)
    wd, ud = w.to(dev), u.to(dev)  # keep the device tensors alive across the launch
    _lib.call("sam6d_weighted_sample", wd.data_ptr(), ud.data_ptr(), 3, 38416, 18000, cum.data_ptr(), idx.data_ptr(), pem._s())
    torch.cuda.synchronize()
    assert torch.equal(idx.cpu().long(), want)
    assert (idx[2] == 0).all()


def test_select_smallest_matches_topk_set(dev):
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(5)
    d = torch.rand(4, 6000, generator=gen)
    d[0, 100:140] = 0.0  # exact ties
    sel = torch.empty(4, 300, dtype=torch.int32, device=dev)
    dd = d.to(dev)
    _lib.call("sam6d_select_smallest", dd.data_ptr(), 4, 6000, 300, sel.data_ptr(), pem._s())
    torch.cuda.synchronize()
    s = sel.cpu().long()
    vals = torch.gather(d, 1, s)
    assert (vals[:, 1:] >= vals[:, :-1]).all(), "ascending order"
    want = torch.topk(d, 300, dim=1, largest=False)[0]
    assert torch.equal(vals, want)
    assert all(len(set(r.tolist())) == 300 for r in s)


@pytest.mark.parametrize("B,n,k", [(3, 6000, 300), (2, 6001, 6001), (1, 37, 5), (2, 4099, 64), (1, 64, 64), (1, 1, 1)])
def test_select_smallest_is_the_stable_rank_order(dev, B, n, k):
    """sel[b, r] = the element with rank r under (value, index) order -- a stable ascending argsort, including exact ties, sizes
    that are not multiples of the wave / the 32-pair popcount group, zeros and negative zeros."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(n + k)
    d = torch.rand(B, n, generator=gen)
    d[:, ::7] = d[:, 0:1].clone()  # many exact ties
    if n > 40:
        d[0, 10:20] = 0.0
        d[0, 20:30] = -0.0        # -0 == +0: ties broken by index
    sel = torch.full((B, k), -1, dtype=torch.int32, device=dev)
    dd = d.to(dev)
    _lib.call("sam6d_select_smallest", dd.data_ptr(), B, n, k, sel.data_ptr(), pem._s())
    torch.cuda.synchronize()
    want = torch.sort(d, dim=1, stable=True)[1][:, :k].to(torch.int32)
    assert torch.equal(sel.cpu(), want)


@pytest.mark.parametrize("n,k", [(14000, 150), (14318, 1), (14319, 1), (14400, 300), (15000, 300)])
def test_select_smallest_long_rows(dev, n, k):
    """Rows around the LDS limit of the radix select ((n + 2k) * 4 bytes + its 8 KB histogram must fit 64 KB): below it the radix kernel,
    above it the all-pairs kernel, same stable rank order either way (the old guard let 60 000 dynamic bytes through)."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(n)
    d = torch.rand(2, n, generator=gen)
    d[:, ::5] = d[:, 1:2].clone()
    sel = torch.full((2, k), -1, dtype=torch.int32, device=dev)
    dd = d.to(dev)
    _lib.call("sam6d_select_smallest", dd.data_ptr(), 2, n, k, sel.data_ptr(), pem._s())
    torch.cuda.synchronize()
    assert torch.equal(sel.cpu(), torch.sort(d, dim=1, stable=True)[1][:, :k].to(torch.int32))


@pytest.mark.parametrize("P", [3277, 4096, 4097])
def test_coarse_rt_large_model_cloud(dev, P):
    """CAD clouds of 3277 .. 4096 points need more than the default 64 KB of dynamic LDS in the matrix-core scoring kernel (the
    attribute was never raised and the launch failed); 4097 takes the vector-ALU kernel.  The score of every selected hypothesis must
    equal the vector-ALU kernel's to summation order (same distance bits) and the pose the oracle's."""
    from sam6d_hip import _lib, pem
    from oracle import pem_oracle as O
    g = golden("coarse_rt")
    p1, p2, u = (_t(g[k]) for k in ("p1", "p2", "u"))
    gen = torch.Generator().manual_seed(P)
    extra = torch.rand(2, P - 1024, 3, generator=gen) - 0.5
    model = torch.cat([_t(g["model"]), extra], 1).contiguous()
    att = _t(g["att"])
    radius = torch.ones(2, device=dev)
    R, t, aux = pem.compute_coarse_Rt(att.to(dev), p1.to(dev), p2.to(dev), model.to(dev), radius, u.to(dev), return_aux=True)
    Ro, to = O.compute_coarse_Rt(att, p1, p2, model, u)[:2]
    _close(R, Ro, 1e-4, "coarse R, P = %d" % P); _close(t, to, 1e-4, "coarse t, P = %d" % P)
    # the plain kernel on the same selection
    sc = torch.empty_like(aux["scores"]); Rb = torch.empty(2, 3, 3, device=dev); tb = torch.empty(2, 3, device=dev)
    best = torch.empty(2, dtype=torch.int32, device=dev)
    md, p1d = model.to(dev), p1.to(dev)
    _lib.call("sam6d_score_select_hypotheses", aux["top"].data_ptr(), aux["Rs"].data_ptr(), aux["ts"].data_ptr(), p1d.data_ptr(),
              aux["w1"].data_ptr(), md.data_ptr(), radius.data_ptr(), 2, 196, P, 6000, 300, sc.data_ptr(), Rb.data_ptr(), tb.data_ptr(),
              best.data_ptr(), pem._s())
    torch.cuda.synchronize()
    # (the distances have the same bits; the two kernels sum them over the scene points in different fixed orders)
    rel = float(((sc - aux["scores"]).abs() / aux["scores"].abs().clamp(min=1e-20)).max())
    assert rel < 2e-6, "matrix-core and vector-ALU hypothesis scores differ by %.2e" % rel
    assert torch.equal(best.cpu(), aux["best"].cpu())


def test_cumsum_norm_and_sampling_vs_float64(dev):
    """sam6d_weighted_sample: cum = cumsum in double rounded to float, / (cum[-1] + 1e-8); idx = first cum >= u (model_utils.py
    :241-243, 277-305) at the path's row length (196 x 196) and at ragged lengths."""
    from sam6d_hip import _lib, pem
    for B, L, ns in ((3, 38416, 18000), (2, 1000, 50), (1, 70001, 300), (2, 64, 7)):
        gen = torch.Generator().manual_seed(L)
        w = torch.rand(B, L, generator=gen) ** 4
        w[:, ::3] = 0
        u = torch.rand(B, ns, generator=gen)
        cum = torch.empty(B, L, device=dev)
        idx = torch.empty(B, ns, dtype=torch.int32, device=dev)
        wd, ud = w.to(dev), u.to(dev)
        _lib.call("sam6d_weighted_sample", wd.data_ptr(), ud.data_ptr(), B, L, ns, cum.data_ptr(), idx.data_ptr(), pem._s())
        torch.cuda.synchronize()
        c64 = torch.cumsum(w.double(), 1).float()
        want = c64 / (c64[:, -1:] + 1e-8)
        got = cum.cpu()
        assert float((got - want).abs().max()) <= 1.2e-7, (L, float((got - want).abs().max()))
        assert (got[:, 1:] >= got[:, :-1]).all()
        # the index search is checked on the kernel's own cum (a 1-ulp difference to the float64 recipe may move a boundary)
        widx = torch.searchsorted(got, u, right=False)
        widx = torch.where(widx >= L, torch.zeros_like(widx), widx)
        assert torch.equal(idx.cpu().long(), widx)


def _kat_atten(p1, p2, sharp, bg):
    B, n, _ = p1.shape
    d = torch.cdist(p1, p2)
    a = torch.full((B, n + 1, p2.shape[1] + 1), float(bg))
    a[:, 1:, 1:] = torch.clamp(1 - sharp * d, min=-1) / 0.1
    return a


def test_fine_rt_golden(dev):
    from sam6d_hip import pem
    g = golden("fine_rt")
    p1, p2, Rg, tg = _t(g["p1"]), _t(g["p2"]), _t(g["R_gt"]), _t(g["t_gt"])
    model = p2[:, :1024].contiguous()
    radius = torch.ones(1, device=dev)
    att = _kat_atten((p1 - tg.unsqueeze(1)) @ Rg, p2, float(g["sharp"]), float(g["bg"]))
    R, t, s = pem.compute_fine_Rt(att.to(dev), p1.to(dev), p2.to(dev), model.to(dev), radius)
    _close(R, g["R"], 1e-5, "fine R"); _close(t, g["t"], 1e-5, "fine t"); _close(s, g["score"], 1e-6, "fine score")
    att2 = torch.randn(1, 2049, 2049, generator=torch.Generator().manual_seed(int(g["att2_seed"]))) * float(g["att2_scale"])
    R2, t2, s2 = pem.compute_fine_Rt(att2.to(dev), p1.to(dev), p2.to(dev), model.to(dev), radius)
    _close(R2, g["R2"], 1e-4, "fine flat R"); _close(t2, g["t2"], 1e-4, "fine flat t"); _close(s2, g["score2"], 1e-5, "fine flat s")


def test_fine_rt_radius_rescale(dev):
    from sam6d_hip import pem
    from oracle import pem_oracle as O
    gen = torch.Generator().manual_seed(8)
    B = 2
    att = torch.randn(B, 301, 301, generator=gen) * 3
    p1 = torch.rand(B, 300, 3, generator=gen) - 0.5
    p2 = torch.rand(B, 300, 3, generator=gen) - 0.5
    model = (torch.rand(B, 500, 3, generator=gen) - 0.5) * 0.3
    radius = torch.tensor([0.25, 0.4])
    R, t, s = pem.compute_fine_Rt(att.to(dev), p1.to(dev), p2.to(dev), model.to(dev), radius.to(dev))
    oR, ot, os_ = O.compute_fine_Rt(att, p1, p2, model / (radius.reshape(-1, 1, 1) + 1e-6))
    ot = ot * (radius.reshape(-1, 1) + 1e-6)
    _close(R, oR, 1e-4, "R"); _close(t, ot, 1e-4, "t"); _close(s, os_, 1e-5, "score")


@pytest.mark.parametrize("P", [500, 4096, 5000, 8192])
def test_fine_score_large_model_cloud(dev, P):
    """sam6d_fine_score (the nearest-CAD-point count of compute_fine_Rt, model_utils.py:331-339) with up to 8192 CAD points -- 128 KB of
    dynamic LDS -- and a point count that is not a multiple of the 64-point workgroups: against a float64 recompute; points whose nearest
    distance lies within 1e-6 of the threshold may fall on either side."""
    from sam6d_hip import _lib
    gen = torch.Generator().manual_seed(P)
    B, N = 3, 301
    p1 = (torch.rand(B, N, 3, generator=gen) - 0.5)
    model = (torch.rand(B, P, 3, generator=gen) - 0.5)
    radius = torch.tensor([1.0, 0.5, 2.0])
    R = torch.linalg.qr(torch.randn(B, 3, 3, generator=gen))[0].contiguous()
    t = (torch.rand(B, 3, generator=gen) - 0.5) * 0.1
    l1 = (torch.rand(B, N, generator=gen) > 0.3).to(torch.int32)
    thr = 0.03
    d = [x.to(dev).contiguous() for x in (p1, R, t, model, radius, l1)]
    cnt = torch.empty(2 * B, device=dev); score = torch.empty(B, device=dev)
    tt = d[2].clone()
    _lib.call("sam6d_fine_score", d[0].data_ptr(), d[1].data_ptr(), tt.data_ptr(), d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(), B, N, P,
              thr, cnt.data_ptr(), score.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    x = (p1.double() - t.double().unsqueeze(1)) @ R.double()
    m = model.double() / (radius.double().reshape(B, 1, 1) + 1e-6)
    dis = torch.cdist(x, m).min(dim=2).values
    mk = (l1 > 0).double()
    lo = (((dis < thr - 1e-6).double() * mk).sum(1) / (mk.sum(1) + 1e-8)) * mk.mean(1)
    hi = (((dis < thr + 1e-6).double() * mk).sum(1) / (mk.sum(1) + 1e-8)) * mk.mean(1)
    got = score.cpu().double()
    assert bool(((got >= lo - 1e-6) & (got <= hi + 1e-6)).all()), (got, lo, hi)
    assert float((hi - lo).max()) < 0.02  # the band is a few threshold cases wide: the check is not vacuous
    assert torch.allclose(tt.cpu(), t * (radius.reshape(B, 1) + 1e-6), rtol=1e-6, atol=0)


def test_fine_score_matrix_core_kernel_equals_vector_kernel(dev, monkeypatch):
    """The nearest-CAD-point count on the fp32 matrix cores (fine_near_mfma_kernel) and on the vector ALU (fine_near_kernel) evaluate the
    same pairwise-distance bit recipe: identical counts, hence identical scores, at the step's shape."""
    from sam6d_hip import _lib
    gen = torch.Generator().manual_seed(77)
    B, N, P = 4, 2048, 1024
    p1 = (torch.rand(B, N, 3, generator=gen) - 0.5)
    model = (torch.rand(B, P, 3, generator=gen) - 0.5)
    radius = torch.tensor([1.0, 0.5, 2.0, 1.3])
    R = torch.linalg.qr(torch.randn(B, 3, 3, generator=gen))[0].contiguous()
    t = (torch.rand(B, 3, generator=gen) - 0.5) * 0.1
    l1 = (torch.rand(B, N, generator=gen) > 0.3).to(torch.int32)
    d = [x.to(dev).contiguous() for x in (p1, R, t, model, radius, l1)]
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SAM6D_FINE_NEAR_MFMA", flag)
        cnt = torch.empty(2 * B, device=dev); score = torch.empty(B, device=dev); tt = d[2].clone()
        _lib.call("sam6d_fine_score", d[0].data_ptr(), d[1].data_ptr(), tt.data_ptr(), d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(), B, N,
                  P, 0.02, cnt.data_ptr(), score.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out[flag] = (cnt.cpu(), score.cpu())
    assert torch.equal(out["1"][0], out["0"][0]) and torch.equal(out["1"][1], out["0"][1])
    assert 0.0 < float(out["1"][1].min()) and float(out["1"][1].max()) < 1.0


def test_procrustes_golden(dev):
    from sam6d_hip import pem
    g = golden("procrustes")
    src, ref = _t(g["src"]).to(dev), _t(g["ref"]).to(dev)
    R, t = pem.weighted_procrustes(src, ref, None, 0.5)
    assert torch.isfinite(R).all() and torch.isfinite(t).all()
    # conditioning of each problem (fp64 on the host): R is well defined when sigma2 - sigma3 is not tiny.  Rank-1 H
    # (rows 400:500, duplicate samples) is LAPACK-dependent in the reference itself -> finite + proper only.
    s_, r_ = _t(g["src"]).double(), _t(g["ref"]).double()
    w = 1.0 / (3.0 + 1e-5)
    sc, rc = (s_ * w).sum(1, keepdim=True), (r_ * w).sum(1, keepdim=True)
    Hm = (s_ - sc).transpose(1, 2) @ (w * (r_ - rc))
    sv = torch.linalg.svdvals(Hm)
    ok = ((sv[:, 1] - sv[:, 2]) / sv[:, 0] > 0.05).numpy()
    ok[400:500] = False
    assert ok.sum() > 3000
    _close(R[torch.from_numpy(ok)], g["R"][ok], 1e-4, "3-point procrustes R (well-conditioned cases)")
    _close(t[torch.from_numpy(ok)], g["t"][ok], 5e-4, "3-point procrustes t (well-conditioned cases)")
    det = torch.linalg.det(R.cpu().double())
    assert (det - 1).abs().max() < 1e-5, "proper rotations, degenerate cases included"
    gen = torch.Generator().manual_seed(int(g["seed_w"]))
    from sam6d_hip import synth
    for _ in range(400):
        synth.random_rotation(gen); torch.randn(3, generator=gen)
    w = torch.rand(8, 2048, generator=gen)
    s2 = torch.randn(8, 2048, 3, generator=gen)
    Rr = torch.stack([synth.random_rotation(gen) for _ in range(8)])
    r2 = s2 @ Rr.transpose(1, 2) + 0.01 * torch.randn(8, 2048, 3, generator=gen)
    Rw, tw = pem.weighted_procrustes(s2.to(dev), r2.to(dev), w.to(dev), 0.0)
    _close(Rw, g["Rw"], 2e-5, "weighted procrustes R")
    _close(tw, g["tw"], 2e-5, "weighted procrustes t")


@pytest.mark.parametrize("sa,sw", [(1.0e6, 1.0), (1.0, 3.0e5), (1.0e-6, 1.0), (1.0e-6, 1.0e-5), (2.0e5, 1.0e-7), (1.0e15, 1.0e-12)])
def test_gemm_operand_range(dev, sa, sw):
    """sam6d_gemm_nt in the default split-precision mode must hold its ~1e-6 relative bound whatever the operands' magnitude: without
    the per-tile power-of-two scaling |x| >= 65520 turned into inf / NaN and uniformly small operands (1e-6) lost their fp16 lo halves
    (and most of their hi halves) to subnormals."""
    from sam6d_hip import _lib, pem
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("split-precision mode only")
    g = torch.Generator().manual_seed(int(abs(math.log10(sa)) * 10 + abs(math.log10(sw))))
    M, N, K = 700, 256, 256
    A = torch.randn(M, K, generator=g) * sa
    A[:, ::7] *= 1e-3  # mixed magnitudes inside a tile
    Wt = torch.randn(N, K, generator=g) * sw
    want = A.double() @ Wt.double().t()
    out = torch.empty(M, N, device=dev)
    pem.gemm(A.to(dev), Wt.to(dev), None, out, M, N, K, K, K, N)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err < 4e-6 * scale, "relative error %.2e of the result scale" % (err / scale)


# --------------------------------------------------------------------------------------------------- a6: pairwise_distance
def test_pairwise_distance_golden_bit_exact(dev):
    """SURVEY 8a row a6 directly: the device recipe of pairwise_distance (PEM/utils/model_utils.py:101-128) against the reference's
    own outputs in tests/golden/pairwise.npz -- the 197-point sparse cloud with the (100,100,100) bg point (the geometric embedding's
    use) and 196 x 1024 (the hypothesis scoring's use; full SHA-256 of all three batches + the first 16 rows) -- bit for bit, through
    the C ABI (sam6d_pairwise_distance) and through the drop-in model_utils.pairwise_distance."""
    import hashlib
    import model_utils as MU
    from sam6d_hip import pem
    g = golden("pairwise")
    pts = _t(g["pts"]).to(dev)
    pd = pem.pairwise_distance(pts, pts)
    assert np.array_equal(pd.cpu().numpy(), g["pd"]), "pairwise_distance 197 x 197 (incl. the bg point): %d entries differ" % int(
        (pd.cpu().numpy() != g["pd"]).sum())
    a, b = _t(g["a"]).to(dev), _t(g["b"]).to(dev)
    pd2 = MU.pairwise_distance(a, b)
    assert np.array_equal(pd2[0, :16].cpu().numpy(), g["pd2_b0"])
    assert hashlib.sha256(np.ascontiguousarray(pd2.cpu().numpy()).tobytes()).hexdigest() == str(g["pd2_sha"])
    # channel-first form of the same call
    pd3 = MU.pairwise_distance(a.transpose(1, 2), b.transpose(1, 2), channel_first=True)
    assert torch.equal(pd3, pd2)
    # the diagonal is NOT exactly zero in this recipe -- and must not be "fixed"
    assert float(pd.diagonal(dim1=1, dim2=2).abs().max()) > 0


# --------------------------------------------------------------------------------------------------- end to end
def _to(dev, inp):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}


def _rot_deg(Ra, Rb):
    M = Ra.double().cpu() @ Rb.double().cpu().transpose(1, 2)
    c = ((M.diagonal(dim1=1, dim2=2).sum(1) - 1) / 2).clamp(-1, 1)
    return torch.rad2deg(torch.acos(c))


@pytest.mark.parametrize("kernels", ["default", "materialised"])
@pytest.mark.parametrize("tag", ["kat", "cfg2"])
def test_pem_end_to_end_vs_reference(dev, W, tag, kernels):
    """Whole path at the post-feature-extraction seam vs the reference's outputs (tests/golden/pem_e2e.npz, B=2): with the DEFAULT
    kernels (fused RPE attention without the embedding tensor + the fine-match pipeline; return_aux does not change the kernel choice)
    and with the materialised launch-per-op forms (embedding tensor, (B,2049,2049) attention matrix)."""
    from sam6d_hip import pem, synth
    g = golden("pem_e2e")
    inp = synth.kat_inputs(B=2, seed=int(g["kat_seed"])) if tag == "kat" else synth.config2_inputs(B=2, seed=int(g["cfg2_seed"]))
    d = _to(dev, inp)
    # (explicit cfg keys: they win over the SAM6D_FUSED_* environment switches an A/B run of the suite may carry)
    cfg = dict(pem.DEFAULT_CFG, fused_rpe=(kernels == "default"), fused_fine=(kernels == "default"))
    R, t, s, aux = pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W,
                                 d["rand"], cfg=cfg, return_aux=True)
    torch.cuda.synchronize()
    from sam6d_hip import _lib
    if _lib.load().sam6d_get_matmul_mode() >= 1:
        assert isinstance(aux["geo"], pem.GeoContext) == (kernels == "default"), "the fused RPE path must be what `default` runs"
        assert ("atten" in aux["fine"]) == (kernels != "default"), "the fine-match pipeline must be what `default` runs"
    assert np.array_equal(aux["fps_idx_m"].cpu().numpy().astype(np.int16), g[tag + "_fps_m"])
    assert np.array_equal(aux["fps_idx_o"].cpu().numpy().astype(np.int16), g[tag + "_fps_o"])
    dR0 = _rot_deg(aux["init_R"], _t(g[tag + "_R0"]))
    dR = _rot_deg(R, _t(g[tag + "_R"]))
    print("\n[%s] coarse: rot diff %s deg, |dt| %s ; final: rot diff %s deg, |dt| %s, dscore %s" % (
        tag, dR0.tolist(), (aux["init_t"].cpu() - _t(g[tag + "_t0"])).norm(dim=1).tolist(), dR.tolist(),
        (t.cpu() - _t(g[tag + "_t"])).norm(dim=1).tolist(), (s.cpu() - _t(g[tag + "_score"])).abs().tolist()))
    _close(aux["init_R"], g[tag + "_R0"], 1e-4, "coarse R")
    _close(aux["init_t"], g[tag + "_t0"], 1e-4, "coarse t")
    _close(R, g[tag + "_R"], 1e-4, "final R")
    _close(t, g[tag + "_t"], 1e-4, "final t")
    _close(s, g[tag + "_score"], 1e-4, "final score")


def test_matmul_modes_error_vs_fp64(dev):
    """exact fp32 MFMA (mode 0) vs fp16x3 split (mode 1, default): both against an fp64 reference; the split mode must
    stay within a few 1e-6 of the result's scale (it carries 22 of fp32's 24 significand bits per operand)."""
    from sam6d_hip import _lib, pem
    g = torch.Generator().manual_seed(12)
    M, N, K = 4096, 256, 256
    A = torch.randn(M, K, generator=g) * 2
    A[:, :8] *= 1e-3  # small-magnitude columns: lo parts fall into fp16 subnormals
    Wt = torch.randn(N, K, generator=g) / 16
    want = A.double() @ Wt.double().t()
    Ad, Wd = A.to(dev), Wt.to(dev)
    errs = {}
    prev = _lib.load().sam6d_get_matmul_mode()
    try:
        for mode in (0, 1):
            _lib.call("sam6d_set_thread_matmul_mode", mode)
            out = torch.empty(M, N, device=dev)
            pem.gemm(Ad, Wd, None, out, M, N, K, K, K, N)
            errs[mode] = float((out.cpu().double() - want).abs().max())
    finally:
        _lib.call("sam6d_set_thread_matmul_mode", -1)
    scale = float(want.abs().max())
    print("\nmatmul max abs err vs fp64: exact-fp32 %.2e, fp16x3 %.2e (result scale %.1f)" % (errs[0], errs[1], scale))
    assert errs[0] < 2e-6 * scale and errs[1] < 4e-6 * scale


def test_geo_embedding_large_index_fallback(dev, W, sd):
    """Indices beyond the branch-free sincos range (un-normalised clouds) must take the exact sincosf kernel: the
    default (fp16x3) mode and the exact mode agree and match the oracle."""
    from sam6d_hip import _lib, pem
    from oracle import pem_oracle as O
    gen = torch.Generator().manual_seed(21)
    pts = (torch.rand(1, 40, 3, generator=gen) - 0.5) * 4.0e4  # d_idx up to ~3e5
    got = pem.geo_embedding(pts.to(dev), W)
    prev = _lib.load().sam6d_get_matmul_mode()
    try:
        _lib.call("sam6d_set_thread_matmul_mode", 0)
        exact = pem.geo_embedding(pts.to(dev), W)
    finally:
        _lib.call("sam6d_set_thread_matmul_mode", -1)
    assert torch.equal(got, exact), "flagged call must be produced by the exact kernel"
    want = O.geo_embedding(pts, sd)
    # arguments ~1e5: one fp32 ulp of the index (0.03) already moves sin/cos by O(1e-2); compare loosely
    assert torch.isfinite(got).all()
    assert float((got.cpu() - want).abs().median()) < 0.2


@pytest.mark.parametrize("spread", [0.5, 3.0, 12.0])
def test_geo_embedding_chebyshev_vs_sinusoid_kernels(dev, W, sd, spread):
    """Default mode: 32-term Chebyshev contraction for indices in [0, 24], the sinusoid kernel (list fix-up) for the rest.
    spread 0.5: only the bg pairs leave the range; 3.0: a large share of d indices do; 12.0: nearly all of them.  All three
    must agree with the exact fp32 kernel and the oracle to the split-precision error."""
    from sam6d_hip import _lib, pem
    from oracle import pem_oracle as O
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("Chebyshev path is the default (fp16x3) mode")
    gen = torch.Generator().manual_seed(int(spread * 10))
    B, n = 3, 197
    pts = (torch.rand(B, n, 3, generator=gen) - 0.5) * 2 * spread + torch.tensor([0.3, -0.2, 4.0])
    pts[:, 0] = 100.0  # bg token
    pts[1, 5] = pts[1, 6]  # duplicate point: d = 0, degenerate angles
    got = pem.geo_embedding(pts.to(dev), W).cpu()
    try:
        _lib.call("sam6d_set_thread_matmul_mode", 0)
        exact = pem.geo_embedding(pts.to(dev), W).cpu()
    finally:
        _lib.call("sam6d_set_thread_matmul_mode", -1)
    d = float((got - exact).abs().max())
    scale = float(exact.abs().max())
    print("\nchebyshev vs exact-fp32 kernel, spread %.1f: max abs diff %.2e (scale %.1f)" % (spread, d, scale))
    assert d < 3e-6 * max(scale, 1.0)
    want = O.geo_embedding(pts, sd)
    _close(got[:, 1:, 1:], want[:, 1:, 1:], 2e-5, "chebyshev geo embedding vs oracle (non-bg pairs)")


@pytest.mark.parametrize("spread,n", [(0.5, 197), (0.5, 64), (3.0, 197), (12.0, 150)])
def test_rpe_self_layer_fused_vs_materialised(dev, W, sd, spread, n):
    """rpe.hip (no embedding tensor: Chebyshev basis + MFMA inside the score kernel, q.k^T / P.v as batched GEMMs) against
    the same layer on the materialised embedding (attention.hip) and against the oracle layer."""
    from sam6d_hip import _lib, pem
    from oracle import pem_oracle as O
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("the fused RPE path is the default (fp16x3) mode")
    gen = torch.Generator().manual_seed(int(spread * 10) + n)
    B = 3
    pts = (torch.rand(B, n, 3, generator=gen) - 0.5) * 2 * spread + torch.tensor([0.3, -0.2, 4.0])
    pts[:, 0] = 100.0
    x = torch.randn(B, n, 256, generator=gen)
    L = W.coarse["blocks"][1]["self"]
    E = pem.geo_embedding(pts.to(dev), W)
    want = pem.rpe_self_layer(x.to(dev), E, L).cpu()
    G = pem.geo_context(pts.to(dev), W)
    got = pem.rpe_self_layer(x.to(dev), G, L).cpu()
    d = float((got - want).abs().max())
    print("\nfused vs materialised RPE layer (spread %.1f, n %d): max abs diff %.2e (scale %.1f)" % (spread, n, d, float(want.abs().max())))
    assert torch.isfinite(got).all() and d < 2e-5
    ora = O.rpe_transformer_layer(x, x, O.geo_embedding(pts, sd), sd, "coarse_point_matching.transformers.1.layers.0")
    _close(got, ora, 1e-4, "fused RPE layer vs oracle")


@pytest.mark.parametrize("xs", [1e-4, 1.0, 3e3])
def test_rpe_fused_query_magnitudes(dev, W, xs):
    """rpe_score_kernel scales the folded query by a power of two per query before its fp16 split: token features of any magnitude
    give the same probabilities as the materialised-embedding layer (compared on the attention output before the layer tail's
    LayerNorm would hide a scale: here on the whole layer with proportionally scaled inputs)."""
    from sam6d_hip import _lib, pem
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("the fused RPE path is the default (fp16x3) mode")
    gen = torch.Generator().manual_seed(77)
    B, n = 2, 197
    pts = (torch.rand(B, n, 3, generator=gen) - 0.5) + torch.tensor([0.3, -0.2, 4.0])
    pts[:, 0] = 100.0
    x = torch.randn(B, n, 256, generator=gen)
    x[0, 5] *= 50.0            # one token far above the others
    x[1, 7] *= 1e-3            # and one far below
    x = x * xs
    L = W.coarse["blocks"][0]["self"]
    E = pem.geo_embedding(pts.to(dev), W)
    want = pem.rpe_self_layer(x.to(dev), E, L).cpu()
    G = pem.geo_context(pts.to(dev), W)
    got = pem.rpe_self_layer(x.to(dev), G, L).cpu()
    rel = float((got - want).abs().max() / want.abs().max())
    print("\nfused RPE layer, feature scale %g: max rel diff %.2e" % (xs, rel))
    assert torch.isfinite(got).all() and rel < 2e-5


@pytest.mark.parametrize("sigma_a,forced", [(15, 0), (15, 3), (7.5, 0)])
def test_rpe_stage1_two_and_three_products(dev, W, sigma_a, forced):
    """The score kernel's stage 1 runs two MFMAs per (block, angular row) when the angular indices' own Chebyshev range ([0, 180 /
    sigma_a]: 12 for the reference's 15 degrees) makes the cross terms of the orders >= 16 negligible (host check of
    geo_cheb_a_packed), else three.  Both variants, and the forced three-product one on the short range, reproduce the materialised
    layer; sigma_a = 7.5 doubles the index range (24 = the distance range), for which the check must refuse the two-product form."""
    import os
    from sam6d_hip import _lib, pem
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("the fused RPE path is the default (fp16x3) mode")
    gen = torch.Generator().manual_seed(91)
    B, n = 17, 197  # 3349 queries: every one of the 256 persistent workgroups runs more queries than it has waves
    pts = (torch.rand(B, n, 3, generator=gen) - 0.5) * 1.6
    pts[:, 0] = 100.0
    x = torch.randn(B, n, 256, generator=gen)
    L = W.coarse["blocks"][2]["self"]
    E = pem.geo_embedding(pts.to(dev), W, sigma_a=sigma_a)
    want = pem.rpe_self_layer(x.to(dev), E, L).cpu()
    old = os.environ.get("SAM6D_RPE_PRODUCTS")
    try:
        if forced:
            os.environ["SAM6D_RPE_PRODUCTS"] = str(forced)
        G = pem.geo_context(pts.to(dev), W, sigma_a=sigma_a)
        got = pem.rpe_self_layer(x.to(dev), G, L).cpu()
    finally:
        if old is None:
            os.environ.pop("SAM6D_RPE_PRODUCTS", None)
        else:
            os.environ["SAM6D_RPE_PRODUCTS"] = old
    assert G.products == (3 if (forced == 3 or sigma_a != 15) else 2), (G.products, G.xmax_a)
    assert abs(G.xmax_a - min(24.0, 180.0 / sigma_a * (1 + 2.0 ** -6))) < 1e-6
    d = float((got - want).abs().max())
    print("\nfused RPE layer, sigma_a %g, products %d, xmax_a %.4f: max abs diff %.2e (scale %.1f)"
          % (sigma_a, G.products, G.xmax_a, d, float(want.abs().max())))
    assert torch.isfinite(got).all() and d < 2e-5


def test_rpe_self_layer_one_launch_attention_equals_gemm_path(dev, W):
    """The default RPE self layer (geometric scores, then q.k^T + softmax + P.v in one launch per (cloud, head)) against the same
    layer with the q.k^T / P.v GEMMs and the softmax inside the score kernel (SAM6D_SELF_ATTN=0)."""
    import os
    from sam6d_hip import _lib, pem
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("the fused RPE path is the default (fp16x3) mode")
    gen = torch.Generator().manual_seed(5)
    B, n = 4, 197
    pts = (torch.rand(B, n, 3, generator=gen) - 0.5) * 1.2
    pts[:, 0] = 100.0
    x = torch.randn(B, n, 256, generator=gen).to(dev)
    L = W.coarse["blocks"][0]["self"]
    G = pem.geo_context(pts.to(dev), W)
    outs = []
    old = os.environ.get("SAM6D_SELF_ATTN")
    try:
        for v in ("1", "0"):
            os.environ["SAM6D_SELF_ATTN"] = v
            outs.append(pem.rpe_self_layer(x, G, L).cpu())
    finally:
        if old is None:
            os.environ.pop("SAM6D_SELF_ATTN", None)
        else:
            os.environ["SAM6D_SELF_ATTN"] = old
    d = float((outs[0] - outs[1]).abs().max())
    print("\none-launch attention vs GEMM path: max abs diff %.2e (scale %.1f)" % (d, float(outs[1].abs().max())))
    assert torch.isfinite(outs[0]).all() and d < 1e-5


def test_rpe_fused_range_guard(dev, sd):
    """Weights whose projected angular embedding could leave the fp16 range of the score kernel's second contraction are detected on
    the host (sum of |Chebyshev coefficients| per channel) and keep the materialised-embedding path; weights whose x1024 images would
    overflow fp16 altogether take the exact fp32 embedding kernel.  Every routed result is checked against matmul mode 0."""
    from sam6d_hip import _lib, pem
    W1 = pem.PemWeights(sd, dev)
    assert pem.fused_rpe_in_range(W1) and pem.geo_images_in_range(W1)
    key = [k for k in sd if k.endswith("geometric_structure_embedding.proj_a.weight") or k.endswith("proj_a.weight")][0]
    keyd = key.replace("proj_a", "proj_d")
    pts = torch.rand(2, 40, 3, generator=torch.Generator().manual_seed(3))
    pts[:, 0] = 100.0  # the bg point: indices far outside the Chebyshev range
    pts = pts.to(dev)
    seen = set()
    prev = _lib.load().sam6d_get_matmul_mode()
    for which, scale in ((key, 20.0), (key, 150.0), (key, 4000.0), (keyd, 20.0), (keyd, 4000.0)):
        big = dict(sd)
        big[which] = sd[which] * scale
        W2 = pem.PemWeights(big, dev)
        fits_fused, fits_img = pem.fused_rpe_in_range(W2), pem.geo_images_in_range(W2)
        seen.add((fits_fused, fits_img))
        if not fits_fused:
            with pytest.raises(ValueError):
                pem.geo_context(pts, W2)
        got = pem.geo_embedding(pts, W2)
        try:
            _lib.call("sam6d_set_thread_matmul_mode", 0)
            want = pem.geo_embedding(pts, W2)
        finally:
            _lib.call("sam6d_set_thread_matmul_mode", -1)
        assert torch.isfinite(got).all(), "scale %g: non-finite embedding (fp16 image overflow)" % scale
        sc = float(want.abs().max())
        err = float((got - want).abs().max())
        assert err <= 2e-5 * sc, "scale %g (fused ok %s, images ok %s): %.3e of %.3e" % (scale, fits_fused, fits_img, err, sc)
    if prev >= 1:
        assert (False, True) in seen and (False, False) in seen, "the scales no longer exercise both guards: %s" % seen


def test_pem_match_fused_vs_materialised(dev, W):
    from sam6d_hip import _lib, pem, synth
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("the fused RPE path is the default (fp16x3) mode")
    inp = synth.config2_inputs(B=3, seed=11)
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    outs = []
    for fused in (True, False):
        cfg = dict(pem.DEFAULT_CFG, fused_rpe=fused)
        outs.append([o.cpu() for o in pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"],
                                                    d["model"], W, d["rand"], cfg=cfg)])
    # the exact-fp32 arithmetic as an explicit per-call option (no process-global switch is touched)
    exact = [o.cpu() for o in pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"],
                                            d["model"], W, d["rand"], options=pem.Options(matmul_mode=0))]
    assert _lib.load().sam6d_get_matmul_mode() == 1 and _lib.load().sam6d_get_thread_matmul_mode() == -1
    for a, b, e, what in zip(outs[0], outs[1], exact, ("R", "t", "score")):
        print("\n%s: fused-vs-exact %.2e  materialised-vs-exact %.2e" % (what, float((a - e).abs().max()), float((b - e).abs().max())))
    for a, b, e, what in zip(outs[0], outs[1], exact, ("R", "t", "score")):
        _close(a, e, 1e-4, "pem_match fused vs exact-fp32 mode: " + what)
        _close(b, e, 1e-4, "pem_match materialised (Chebyshev) vs exact-fp32 mode: " + what)


def test_two_weight_sets_with_different_arithmetic_in_one_process(dev, sd):
    """matmul mode and kernel routes travel with the weight set (PemWeights(..., options=...)), not with the process: a split-precision
    model and an exact-fp32 model interleaved call by call give what each gives alone (bitwise), and differ from each other."""
    from sam6d_hip import _lib, pem, synth
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("needs the default (fp16x3) process mode")
    inp = synth.config2_inputs(B=2, seed=21)
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    args = [d[k] for k in ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")]
    Wa = pem.PemWeights(sd, dev, options=pem.Options(matmul_mode=1))
    Wb = pem.PemWeights(sd, dev, options=pem.Options(matmul_mode=0))
    run = lambda Wx: [o.cpu() for o in pem.pem_match(*args, Wx, d["rand"])]
    a0, b0 = run(Wa), run(Wb)
    b1, a1 = run(Wb), run(Wa)
    for x, y in zip(a0 + b0, a1 + b1):
        assert torch.equal(x, y)
    assert any(not torch.equal(x, y) for x, y in zip(a0, b0)), "the two arithmetic modes cannot be bit-identical"
    for x, y, what in zip(a0, b0, ("R", "t", "score")):
        _close(x, y, 1e-4, "split vs exact arithmetic: " + what)
    assert _lib.load().sam6d_get_thread_matmul_mode() == -1


def test_two_threads_two_arithmetic_modes(dev, sd):
    """The entry points keep their state (Options in flight, nesting depth) per THREAD and the library its matmul mode per thread: two
    Python threads driving a split-precision and an exact-fp32 weight set at the same time each get what they get alone."""
    import threading
    from sam6d_hip import _lib, pem, synth
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("needs the default (fp16x3) process mode")
    inp = synth.config2_inputs(B=2, seed=22)
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    args = [d[k] for k in ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")]
    Ws = [pem.PemWeights(sd, dev, options=pem.Options(matmul_mode=1)), pem.PemWeights(sd, dev, options=pem.Options(matmul_mode=0))]
    alone = [[o.cpu() for o in pem.pem_match(*args, Wx, d["rand"])] for Wx in Ws]
    got, errs = [[], []], []

    def work(i):
        try:
            with torch.cuda.device(dev):
                for _ in range(4):
                    got[i].append([o.cpu() for o in pem.pem_match(*args, Ws[i], d["rand"])])
        except Exception as e:  # surfaced in the main thread
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for i in range(2):
        for res in got[i]:
            assert all(torch.equal(x, y) for x, y in zip(res, alone[i])), "thread %d: result differs from the single-threaded run" % i
    assert _lib.load().sam6d_get_thread_matmul_mode() == -1


def test_pem_match_repeatable_with_side_stream(dev, W):
    """The default pipeline (one batch, pose-independent fine work on a side stream) must reproduce the serial result bit for
    bit, run after run (the micro-batch mode has its own invariance test; both have been bit-stable since the library is built
    without packed-fp32 instructions, DESIGN "Concurrency caveat")."""
    from sam6d_hip import pem, synth
    inp = synth.config2_inputs(B=16, seed=5)
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    run = lambda ov: [o.cpu() for o in pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"],
                                                     d["model"], W, d["rand"], cfg=dict(pem.DEFAULT_CFG, overlap=ov, microbatch=1))]
    ref = run(False)
    for rep in range(12):
        for a, b, what in zip(run(True), ref, ("R", "t", "score")):
            assert torch.equal(a, b), "run %d with the side stream: %s differs by %.3e" % (rep, what, float((a - b).abs().max()))


def test_pem_match_microbatch_mode_is_bit_invariant(dev, W):
    """SAM6D_MICROBATCH=2 (two slices on two HIP streams, off by default) returns the serial result bit for bit: the slices are
    independent proposals and no kernel's arithmetic depends on the batch size.  (Before the library was built without packed-fp32
    instructions a third of such runs differed: DESIGN "Concurrency caveat"; scratch/dbg_ov.py is the long stress.)"""
    from sam6d_hip import pem, synth
    inp = synth.config2_inputs(B=16, seed=6)
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    run = lambda mb: [o.cpu() for o in pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"],
                                                     d["model"], W, d["rand"], cfg=dict(pem.DEFAULT_CFG, overlap=True, microbatch=mb))]
    ref = run(1)
    for rep in range(6):
        for a, b, what in zip(run(2), ref, ("R", "t", "score")):
            assert torch.equal(a, b), "run %d with two slices: %s differs by %.3e" % (rep, what, float((a - b).abs().max()))


@pytest.mark.parametrize("M,K", [(1, 256), (197, 256), (6304, 512), (65, 32), (4099, 256)])
def test_gemm_ln256_matches_gemm_then_layernorm(dev, M, K):
    """sam6d_gemm_ln256 (projection + residual + LayerNorm fused) against the two-launch form and torch fp64."""
    from sam6d_hip import _lib, pem
    if _lib.load().sam6d_get_matmul_mode() != 1:
        pytest.skip("fused projection + LayerNorm exists in the split-precision mode only")
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g); w = torch.randn(256, K, generator=g) / math.sqrt(K); b = torch.randn(256, generator=g)
    res = torch.randn(M, 256, generator=g); gam = torch.rand(256, generator=g) + 0.5; bet = torch.randn(256, generator=g)
    lin = pem.Linear(w.to(dev), b.to(dev))
    got = pem.gemm_ln(x.to(dev), lin, res.to(dev), (gam.to(dev), bet.to(dev))).cpu()
    two = pem.layernorm(pem.linear(x.to(dev), lin, residual=res.to(dev)), (gam.to(dev), bet.to(dev))).cpu()
    want = torch.nn.functional.layer_norm(x.double() @ w.double().t() + b.double() + res.double(), (256,), gam.double(), bet.double(), 1e-5)
    assert torch.isfinite(got).all()
    _close(got, two, 5e-6, "fused vs gemm + layernorm")
    _close(got, want.float(), 2e-5, "fused vs torch fp64")


def test_config5_shape_4096_points(dev, W, sd):
    """BASELINE config 5's geometry (fine_npoint = 4096): the whole path at N = 4096 dense points, B = 1, against the CPU
    oracle (fp32; the config's fp16 attention variant is a later round)."""
    from sam6d_hip import pem, synth
    from oracle import pem_oracle as O
    inp = synth.kat_inputs(B=1, seed=9, n_dense=4096, n_model=1024)
    d = _to(dev, inp)
    R, t, s, aux = pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W,
                                 d["rand"], return_aux=True)
    oR, ot, os_, oaux = O.pem_match(inp["dense_pm"], inp["dense_fm"], inp["dense_po"], inp["dense_fo"], inp["radius"],
                                    inp["model"], sd, inp["rand"], return_aux=True)
    assert torch.equal(aux["fps_idx_m"].cpu(), oaux["fps_idx_m"]) and torch.equal(aux["fps_idx_o"].cpu(), oaux["fps_idx_o"])
    _close(aux["init_R"], oaux["init_R"], 1e-4, "coarse R"); _close(aux["init_t"], oaux["init_t"], 1e-4, "coarse t")
    _close(R, oR, 1e-4, "R"); _close(t, ot, 1e-4, "t"); _close(s, os_, 1e-4, "score")


def test_template_cloud_fps_210k(dev):
    """FPS over the 42 x 5000 = 210 000-point template cloud of get_obj_feats -> 2048 samples, plus
    the feature gather (PEM/model/feature_extraction.py:152-158), bit-exact vs the oracle."""
    from sam6d_hip import pem
    from oracle import pointops as P
    gen = torch.Generator().manual_seed(31)
    pts = (torch.rand(1, 210000, 3, generator=gen) - 0.5) * 0.3
    feats = torch.randn(1, 210000, 8, generator=gen)
    want = P.furthest_point_sampling(pts, 2048)
    sp, sf, idx = pem.sample_pts_feats(pts.to(dev), feats.to(dev), 2048)
    assert torch.equal(idx.cpu(), want)
    assert torch.equal(sp.cpu(), torch.gather(pts, 1, want.long().unsqueeze(2).expand(1, 2048, 3)))
    assert torch.equal(sf.cpu(), torch.gather(feats, 1, want.long().unsqueeze(2).expand(1, 2048, 8)))


@pytest.mark.parametrize("B,R,C,peaky", [(3, 197, 197, False), (2, 197, 197, True), (2, 50, 120, False), (1, 2, 2, False)])
def test_coarse_soft_assign_one_launch_is_bit_identical(dev, B, R, C, peaky):
    """sam6d_coarse_soft_assign (matrix in LDS, one workgroup per proposal) against sam6d_soft_assign + sam6d_coarse_weights: every
    statistic, label and weight bit for bit (the sampled hypothesis indices downstream depend on the exact weights)."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(B * 1000 + R + C + int(peaky))
    att = torch.randn(B, R, C, generator=gen) * (0.2 if not peaky else 1.0)
    if peaky:
        for b in range(B):
            perm = torch.randperm(C - 1, generator=gen)[: R - 1] + 1
            att[b, torch.arange(1, R), perm[: R - 1]] += 12.0
            att[b, 1:40, 0] += 15.0  # some rows prefer the bg column
    att = att.to(dev).contiguous()
    st = pem.soft_assign(att)
    w = torch.zeros(B, (R - 1) * (C - 1), device=dev); w1 = torch.zeros(B, R - 1, device=dev)
    _lib.call("sam6d_coarse_weights", pem._p(att), B, R, C, pem._p(st["rmax"]), pem._p(st["rsum"]), pem._p(st["cmax"]), pem._p(st["csum"]),
              pem._p(st["l1"]), pem._p(st["l2"]), pem._p(w), pem._p(w1), pem._s())
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
    o = dict(rmax=z(B, R), rsum=z(B, R), cmax=z(B, C), csum=z(B, C), l1=z(B, R - 1, dt=torch.int32), l2=z(B, C - 1, dt=torch.int32))
    w2 = z(B, (R - 1) * (C - 1)); w12 = z(B, R - 1)
    _lib.call("sam6d_coarse_soft_assign", pem._p(att), B, R, C, pem._p(o["rmax"]), pem._p(o["rsum"]), pem._p(o["cmax"]), pem._p(o["csum"]),
              pem._p(o["l1"]), pem._p(o["l2"]), pem._p(w2), pem._p(w12), pem._s())
    torch.cuda.synchronize()
    for k in ("rmax", "rsum", "cmax", "csum", "l1", "l2"):
        assert torch.equal(st[k], o[k]), k
    assert torch.equal(w, w2), "weights: %d differ" % int((w != w2).sum())
    assert torch.equal(w1, w12)
    if peaky:
        assert int((o["l1"] > 0).sum()) > 0 and int((o["l1"] == 0).sum()) > 0
