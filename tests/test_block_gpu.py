"""GPU parity of the fused transformer-block kernels (csrc/block.hip) through the C ABI: the layer tail (linear + residual + LayerNorm +
FFN + residual + LayerNorm; PEM/model/transformer.py:152-160, 184-199) and the whole dense LinearTransformerLayer (:532-622), against a
float64 recompute and the CPU oracle; operand magnitudes far outside fp16's range check the power-of-two scaling."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(gen, wscale=1.0):
    from sam6d_hip import pem
    mk = lambda o, i: pem.Linear((torch.rand(o, i, generator=gen) * 2 - 1) / math.sqrt(i) * wscale, (torch.rand(o, generator=gen) * 2 - 1) / math.sqrt(i))
    return dict(lin=mk(256, 256), n1=(1 + 0.1 * torch.randn(256, generator=gen), 0.1 * torch.randn(256, generator=gen)), exp=mk(512, 256),
                sq=mk(256, 512), n2=(1 + 0.1 * torch.randn(256, generator=gen), 0.1 * torch.randn(256, generator=gen)))


def _to(L, dev):
    from sam6d_hip import pem
    out = {}
    for k, v in L.items():
        if isinstance(v, pem.Linear):
            out[k] = pem.Linear(v.w.to(dev), v.b.to(dev))
        elif isinstance(v, tuple):
            out[k] = tuple(x.to(dev).contiguous() for x in v)
        else:
            out[k] = v.to(dev)
    return out


def _tail64(hidden, x, L):
    d = lambda t: t.double()
    ln = lambda v, gb: torch.nn.functional.layer_norm(v, (256,), d(gb[0]), d(gb[1]), 1e-5)
    y = ln(d(hidden) @ d(L["lin"].w).t() + d(L["lin"].b) + d(x), L["n1"])
    h = torch.relu(y @ d(L["exp"].w).t() + d(L["exp"].b))
    return ln(h @ d(L["sq"].w).t() + d(L["sq"].b) + y, L["n2"])


@pytest.mark.parametrize("M,hs,ws", [(1, 1.0, 1.0), (127, 1.0, 1.0), (128, 1.0, 1.0), (1000, 1.0, 1.0), (12608, 1.0, 1.0),
                                     (300, 1.0e5, 1.0), (300, 1.0e-6, 1.0), (300, 3.0, 300.0), (300, 1.0, 1.0e-4)])
def test_token_block_vs_fp64(dev, M, hs, ws):
    """hs scales the attention output (1e5: beyond fp16's 65504; 1e-6: every lo half would be a subnormal without the row scale),
    ws the linear weight."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(M + int(math.log10(hs) * 7) + int(math.log10(ws) * 3))
    L = _layer(gen)
    L["lin"] = pem.Linear(L["lin"].w * ws, L["lin"].b)
    hidden = torch.randn(M, 256, generator=gen) * hs
    x = torch.randn(M, 256, generator=gen) * (hs * ws if hs * ws > 1 else 1.0)
    want = _tail64(hidden, x, L)
    Ld = _to(L, dev)
    tb = pem.pack_token_block(Ld)
    out = torch.full((M, 256), float("nan"), device=dev)
    hd, xd = hidden.to(dev), x.to(dev)  # (named: a temporary would be freed, and its memory reused, before the launch)
    _lib.call("sam6d_token_block", hd.data_ptr(), xd.data_ptr(), tb["img"].data_ptr(), tb["cst"].data_ptr(), out.data_ptr(), M, 1e-5,
              torch.cuda.current_stream().cuda_stream)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max())
    assert err < 2e-5, "token block vs fp64: %.3e" % err


def test_token_block_matches_unfused_path(dev, monkeypatch):
    """same layer through the launch-per-op path (GEMM, LayerNorm kernels) and the fused kernel"""
    from sam6d_hip import pem
    gen = torch.Generator().manual_seed(5)
    Ld = _to(_layer(gen), dev)
    Ld["tb"] = pem.pack_token_block(Ld)
    hidden = torch.randn(777, 256, generator=gen).to(dev)
    x = torch.randn(777, 256, generator=gen).to(dev)
    a = pem._post_attention(hidden, x, Ld)
    monkeypatch.setenv("SAM6D_FUSED_BLOCK", "0")
    b = pem._post_attention(hidden, x, Ld)
    assert float((a - b).abs().max()) < 2e-5


@pytest.mark.parametrize("Bp,J", [(1, 196), (5, 196), (3, 37)])
def test_linattn_kv_image_equals_three_launches(dev, Bp, J):
    """sam6d_linattn_kv_image (phi(k) + kv^T + key sums + packed image in one launch) gives the same bits as
    sam6d_linattn_focus_k + sam6d_linattn_kv + sam6d_linattn_kv_pack, and leaves the kv rows untouched."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(Bp * 100 + J)
    kv = (torch.randn(Bp, J, 512, generator=gen) * 1.7).to(dev).contiguous()
    scale = (0.3 * torch.randn(256, generator=gen)).to(dev)
    nb = int(_lib.load().sam6d_linattn_kv_image_bytes())
    img1 = torch.zeros(Bp * nb, dtype=torch.uint8, device=dev); inv1 = torch.zeros(Bp, 4, device=dev); ks1 = torch.zeros(Bp, 4, 64, device=dev)
    kv_in = kv.clone()
    _lib.call("sam6d_linattn_kv_image", pem._p(kv), pem._p(scale), Bp, J, 512, J * 512, img1.data_ptr(), pem._p(inv1), pem._p(ks1), pem._s())
    assert torch.equal(kv, kv_in)
    k2 = kv.clone()
    _lib.call("sam6d_linattn_focus_k", pem._p(k2), pem._p(scale), Bp * J, 512, pem._s())
    kvT = torch.zeros(Bp, 4, 64, 64, device=dev); ks2 = torch.zeros(Bp, 4, 64, device=dev)
    _lib.call("sam6d_linattn_kv", pem._p(k2), pem._p(k2, 256), Bp, J, 512, 512, J * 512, J * 512, pem._p(kvT), pem._p(ks2), pem._s())
    img2 = torch.zeros(Bp * nb, dtype=torch.uint8, device=dev); inv2 = torch.zeros(Bp, 4, device=dev)
    _lib.call("sam6d_linattn_kv_pack", pem._p(kvT), Bp, img2.data_ptr(), pem._p(inv2), pem._s())
    torch.cuda.synchronize()
    assert torch.equal(ks1, ks2), "key sums"
    assert torch.equal(inv1, inv2), "image scales (one per head)"
    assert inv1.shape == (Bp, 4) and bool((inv1 > 0).all())
    assert torch.equal(img1, img2), "packed kv^T image: %d bytes differ" % int((img1 != img2).sum())


@pytest.mark.parametrize("Bp,I,qs", [(2, 2049, 1.0), (3, 300, 1.0), (1, 130, 1.0), (2, 257, 1.0e4), (2, 257, 1.0e-5)])
def test_linattn_layer_vs_oracle_math(dev, Bp, I, qs):
    """Whole dense layer on rows 1 .. I-1 against a float64 recompute of LinearAttention + tail (PEM/model/transformer.py:532-622);
    qs scales the dense tokens (focused attention is scale-free in q up to the 1e-6 offsets, the residual / LayerNorm are not)."""
    from sam6d_hip import pem
    gen = torch.Generator().manual_seed(Bp * 1000 + I)
    L = _layer(gen)
    mk = lambda o, i: pem.Linear((torch.rand(o, i, generator=gen) * 2 - 1) / math.sqrt(i), (torch.rand(o, generator=gen) * 2 - 1) / math.sqrt(i))
    L["q"], L["kv"] = mk(256, 256), mk(512, 256)
    L["scale"] = 0.3 * torch.randn(256, generator=gen)
    J = 196
    D = torch.randn(Bp, I, 256, generator=gen) * qs
    S = torch.randn(Bp, J + 1, 256, generator=gen)
    # float64 reference
    d = lambda t: t.double()
    Dt = d(D[:, 1:])
    mem = d(S[:, 1:])
    q = Dt @ d(L["q"].w).t() + d(L["q"].b)
    k = mem @ d(L["kv"].w[:256]).t() + d(L["kv"].b[:256])
    v = mem @ d(L["kv"].w[256:]).t() + d(L["kv"].b[256:])
    sp = torch.nn.functional.softplus(d(L["scale"]))

    def phi(z):
        z = (torch.relu(z) + 1e-6) / sp
        n = z.norm(dim=-1, keepdim=True)
        z3 = z ** 3
        return z3 / z3.norm(dim=-1, keepdim=True) * n
    q, k = phi(q), phi(k)
    hid = torch.zeros_like(q)
    for h in range(4):
        sl = slice(64 * h, 64 * h + 64)
        z = 1.0 / (torch.einsum("bic,bc->bi", q[..., sl], k[..., sl].sum(1)) + 1e-6)
        kvm = torch.einsum("bjc,bjd->bcd", k[..., sl], v[..., sl])
        hid[..., sl] = torch.einsum("bic,bcd,bi->bid", q[..., sl], kvm, z)
    want = _tail64(hid, Dt, L)
    Ld = _to(L, dev)
    Ld["scale"] = L["scale"].to(dev)
    Ld["tb"] = pem.pack_token_block(Ld)
    Ld["tbd"] = pem.pack_token_block(Ld, Ld["q"], Ld["scale"])
    got = pem.linear_transformer_layer(D.to(dev).contiguous(), S.to(dev).contiguous(), Ld)[:, 1:].cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max())
    assert err < 5e-5, "dense linear-attention layer vs fp64: %.3e" % err


# ------------------------------------------------------------------------------------- fine similarity + soft assignment pipeline
def _fine_match_oracle(F, B, temp, pts2):
    """compute_feature_similarity + the head of compute_fine_Rt on the CPU (oracle/pem_oracle.py restates model_utils.py:131-153, 308-331)."""
    from oracle import pem_oracle as O
    n = F.shape[1]
    f1 = torch.nn.functional.normalize(F[:B], dim=2)
    f2 = torch.nn.functional.normalize(F[B:], dim=2)
    att = f1 @ f2.transpose(1, 2) / temp
    S, l1, l2 = O.soft_assignment(att)
    A = S[:, 1:, 1:] * (l1 > 0).float().unsqueeze(2) * (l2 > 0).float().unsqueeze(1)
    w = A.sum(2)
    pred = (A / (w.unsqueeze(2) + 1e-6)) @ pts2
    return att, l1, l2, w, pred


@pytest.mark.parametrize("kind", ["matched", "flat"])
def test_fine_match_pipeline_vs_oracle(dev, kind, n=2049):
    """sam6d_fine_match (finematch.hip) against the oracle.  `matched`: every scene feature is a noisy copy of one template feature (a
    third of them of the bg token), so the arg-max labels are well separated and must be bit-exact; `flat`: unrelated random features
    (a nearly uniform assignment matrix whose arg-max is decided in the last bits): >= 99.5 % of the labels, weights / targets to 1e-5."""
    from sam6d_hip import pem
    gen = torch.Generator().manual_seed(3 if kind == "matched" else 4)
    B = 2
    F = torch.randn(2 * B, n, 256, generator=gen)
    if kind == "matched":
        perm = torch.stack([torch.randperm(n, generator=gen) for _ in range(B)])
        for b in range(B):
            src = F[B + b][perm[b]]
            src[::3] = F[B + b][0]  # a third of the scene points look like the template's bg token
            F[b] = src + 0.25 * torch.randn(n, 256, generator=gen)
    pts2 = torch.rand(B, n - 1, 3, generator=gen) - 0.5
    att, l1, l2, w, pred = _fine_match_oracle(F, B, 0.1, pts2)
    gl1, gl2, gpred, gw = pem.fine_match(F.reshape(-1, 256).to(dev).contiguous(), B, n, 0.1, pts2.to(dev).contiguous())
    n1 = int((gl1.cpu() != l1.to(torch.int32)).sum()); n2 = int((gl2.cpu() != l2.to(torch.int32)).sum())
    print("\\n[%s] label mismatches: rows %d, columns %d of %d; fg rows %d" % (kind, n1, n2, l1.numel(), int((l1 > 0).sum())))
    if kind == "matched":
        assert n1 == 0 and n2 == 0
        assert 0 < int((l1 > 0).sum()) < l1.numel() and 0 < int((l2 > 0).sum())
        assert float((gw.cpu() - w).abs().max()) < 1e-5 and float((gpred.cpu() - pred).abs().max()) < 1e-5
    else:
        assert n1 <= 0.005 * l1.numel() and n2 <= 0.005 * l2.numel()
        same = ((gl1.cpu() == l1.to(torch.int32)) & (l1 > 0)) | ((gl1.cpu() == 0) & (l1 == 0))
        assert float((gw.cpu() - w).abs()[same].max()) < 1e-5


def test_fine_match_4097_tokens_vs_oracle(dev):
    """config 5's fine stage size: labels bit-exact against the oracle on well-separated (matched) features"""
    test_fine_match_pipeline_vs_oracle(dev, "matched", n=4097)


@pytest.mark.parametrize("B,n", [(3, 2049), (2, 4097)])
def test_fine_match_equals_unfused_path(dev, B, n):
    """same features through the launch-per-op path (l2norm, GEMM, sam6d_soft_assign, sam6d_fine_assign) and the pipeline; n = 4097 is
    BASELINE config 5's 4096-point fine stage (label / assignment passes per 2048-column chunk + merges)"""
    from sam6d_hip import pem
    gen = torch.Generator().manual_seed(8)
    F = torch.randn(2 * B, n, 256, generator=gen)
    perm = torch.randperm(n, generator=gen)
    F[:B] = F[B:, perm] + 0.3 * torch.randn(B, n, 256, generator=gen)
    pts1 = (torch.rand(B, n - 1, 3, generator=gen) - 0.5).to(dev)
    pts2 = (torch.rand(B, n - 1, 3, generator=gen) - 0.5).to(dev)
    model = (torch.rand(B, 512, 3, generator=gen) - 0.5).to(dev)
    radius = torch.ones(B, device=dev)
    Fd = F.to(dev)
    f = Fd.reshape(-1, 256).clone()
    pem._lib.call("sam6d_l2norm256", f.data_ptr(), f.data_ptr(), 2 * B * n, 256, 256, torch.cuda.current_stream().cuda_stream)
    att = torch.empty(B, n, n, device=dev)
    pem.gemm(f, f, None, att, n, n, 256, 256, 256, n, w_off=B * n * 256, batch=B, sA=n * 256, sW=n * 256, sC=n * n, divisor=0.1)
    Ra, ta, sa = pem.compute_fine_Rt(att, pts1, pts2, model, radius)
    Rb, tb, sb = pem.compute_fine_Rt_fused(Fd.reshape(-1, 256).contiguous(), B, n, 0.1, pts1, pts2, model, radius)
    assert float((Ra - Rb).abs().max()) < 1e-5 and float((ta - tb).abs().max()) < 1e-5 and float((sa - sb).abs().max()) < 1e-5


# --------------------------------------------------------------------------------------------------- fused cross attention
@pytest.mark.parametrize("B,n,m,xs", [(2, 197, 197, 1.0), (3, 50, 208, 1.0), (1, 16, 1, 1.0), (2, 64, 32, 1.0), (2, 197, 197, 3.0e4),
                                      (2, 130, 197, 1.0e-5)])
def test_cross_attention_vs_fp64(dev, B, n, m, xs):
    """sam6d_cross_attention (proj_q + 4-head softmax attention, xattn.hip) against a float64 recompute of MultiHeadAttention
    (PEM/model/transformer.py:95-150); xs scales the query-side tokens and 1 / xs the keys (operands far outside fp16's range, logits
    of ordinary size: with logits of 1e5 the softmax would be decided by fp32's own last bits).  (2, 64, 32) once had one probability
    off by 2^-11: clang rounded hi and lo of one fp16 split from differently rounded products (common.h sam6d_split_f16)."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(B * 100 + n + m)
    mk = lambda o, i: pem.Linear((torch.rand(o, i, generator=gen) * 2 - 1) / math.sqrt(i), (torch.rand(o, generator=gen) * 2 - 1) / math.sqrt(i))
    q = mk(256, 256)
    x = torch.randn(B, n, 256, generator=gen) * xs
    kv = torch.randn(B, m, 512, generator=gen)
    kv[..., :256] /= xs
    q = pem.Linear(q.w, q.b * xs)
    d = lambda t: t.double()
    qq = d(x) @ d(q.w).t() + d(q.b)
    want = torch.zeros(B, n, 256, dtype=torch.float64)
    for h in range(4):
        sl = slice(64 * h, 64 * h + 64)
        att = torch.softmax(qq[..., sl] @ d(kv[..., sl]).transpose(1, 2) / 8.0, dim=-1)
        want[..., sl] = att @ d(kv[..., 256:][..., sl])
    qd = pem.Linear(q.w.to(dev), q.b.to(dev))
    xq = pem.pack_cross_query(qd)
    xd, kvd = x.to(dev).contiguous(), kv.to(dev).contiguous()
    out = torch.full((B, n, 256), float("nan"), device=dev)
    _lib.call("sam6d_cross_attention", xd.data_ptr(), kvd.data_ptr(), xq["img"].data_ptr(), qd.b.data_ptr(), float(xq["inv"]),
              out.data_ptr(), B, n, m, torch.cuda.current_stream().cuda_stream)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max())
    assert err < 2e-6 * max(1.0, float(want.abs().max())), "cross attention vs fp64: %.3e" % err


@pytest.mark.parametrize("B,n,m,xs", [(2, 197, 197, 1.0), (3, 50, 208, 1.0), (1, 16, 1, 1.0), (2, 64, 32, 1.0), (2, 197, 197, 3.0e4),
                                      (2, 130, 197, 1.0e-5), (1, 197, 100, 1.0)])
def test_cross_attention_kv_vs_fp64(dev, B, n, m, xs):
    """sam6d_cross_attention_kv -- proj_q, proj_k, proj_v and the 4-head softmax attention of MultiHeadAttention in ONE launch
    (PEM/model/transformer.py:111-150), the memory tokens as input -- against a float64 recompute; xs scales the query-side tokens and
    1 / xs the key projection (see test_cross_attention_vs_fp64).  Also: equal to proj GEMM + sam6d_cross_attention within rounding."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(B * 100 + n + m + 7)
    mk = lambda o, i: ((torch.rand(o, i, generator=gen) * 2 - 1) / math.sqrt(i), (torch.rand(o, generator=gen) * 2 - 1) / math.sqrt(i))
    qw, qb = mk(256, 256)
    kw, kb = mk(256, 256)
    vw, vb = mk(256, 256)
    x = torch.randn(B, n, 256, generator=gen) * xs
    mem = torch.randn(B, m, 256, generator=gen)
    qb = qb * xs
    kw, kb = kw / xs, kb / xs
    d = lambda t: t.double()
    qq = d(x) @ d(qw).t() + d(qb)
    kk = d(mem) @ d(kw).t() + d(kb)
    vv = d(mem) @ d(vw).t() + d(vb)
    want = torch.zeros(B, n, 256, dtype=torch.float64)
    for h in range(4):
        sl = slice(64 * h, 64 * h + 64)
        att = torch.softmax(qq[..., sl] @ kk[..., sl].transpose(1, 2) / 8.0, dim=-1)
        want[..., sl] = att @ vv[..., sl]
    qd = pem.Linear(qw.to(dev), qb.to(dev))
    kvd = pem.Linear(torch.cat([kw, vw], 0).to(dev), torch.cat([kb, vb], 0).to(dev))
    xq, xkv = pem.pack_cross_query(qd), pem.pack_cross_kv(kvd)
    xd, md = x.to(dev).contiguous(), mem.to(dev).contiguous()
    out = torch.full((B, n, 256), float("nan"), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("sam6d_cross_attention_kv", xd.data_ptr(), md.data_ptr(), xq["img"].data_ptr(), qd.b.data_ptr(), float(xq["inv"]),
              xkv["img"].data_ptr(), kvd.b.data_ptr(), float(xkv["inv"]), out.data_ptr(), B, n, m, st)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max())
    assert err < 2e-6 * max(1.0, float(want.abs().max())), "cross attention (kv inside) vs fp64: %.3e" % err
    kvt = pem.linear(md.reshape(B * m, 256), kvd)
    out2 = torch.full((B, n, 256), float("nan"), device=dev)
    _lib.call("sam6d_cross_attention", xd.data_ptr(), kvt.data_ptr(), xq["img"].data_ptr(), qd.b.data_ptr(), float(xq["inv"]),
              out2.data_ptr(), B, n, m, st)
    assert float((out2.cpu().double() - got).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,n,qs,gs", [(2, 197, 1.0, 1.0), (3, 50, 1.0, 1.0), (1, 4, 1.0, 1.0), (2, 208, 1.0, 1.0), (2, 197, 300.0, 1.0),
                                       (2, 130, 1e-3, 40.0), (5, 16, 1.0, 1.0)])
def test_rpe_self_attention_vs_fp64(dev, B, n, qs, gs):
    """sam6d_rpe_self_attention: hidden_h = softmax((q_h k_h^T + G) / 8) v_h for the four heads of one RPE self layer
    (RPEMultiHeadAttention.forward, PEM/model/transformer.py:405-416, with the geometric term G given) against a float64 recompute;
    qs scales the queries (and 1 / qs the keys), gs the geometric term; rows of G are ldp = 4 ceil(n / 4) floats apart."""
    from sam6d_hip import _lib
    gen = torch.Generator().manual_seed(17 * B + n)
    ldp = (n + 3) // 4 * 4
    qkv = torch.randn(B * n, 768, generator=gen)
    qkv[:, :256] *= qs
    qkv[:, 256:512] /= qs
    qkv[3 % (B * n), :256] *= 30.0  # one query row far above the others
    G = torch.full((B * n, 4, ldp), float("nan"))
    G[:, :, :n] = torch.randn(B * n, 4, n, generator=gen) * gs
    G[:, :, n:] = 1e30  # padding of a row: never read as a key
    d = lambda t: t.double()
    q, k, v = (d(qkv[:, i * 256:(i + 1) * 256]).reshape(B, n, 4, 64).permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(-1, -2) + d(G[:, :, :n]).reshape(B, n, 4, n).permute(0, 2, 1, 3)) / 8.0
    want = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * n, 256)
    out = torch.full((B * n, 256), float("nan"), device=dev)
    qd, Gd = qkv.to(dev), G.to(dev)
    _lib.call("sam6d_rpe_self_attention", qd.data_ptr(), Gd.data_ptr(), out.data_ptr(), B, n, ldp, torch.cuda.current_stream().cuda_stream)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max())
    assert err < 3e-6 * max(1.0, float(want.abs().max())), "rpe self attention vs fp64: %.3e" % err


@pytest.mark.parametrize("M,xs", [(64 * 5, 1.0), (130, 1.0), (1, 1.0), (4098, 30.0), (257, 1e-4)])
def test_linear_norm_split_vs_fp64(dev, M, xs):
    """sam6d_linear_norm_split: fh + fl = normalize(x W^T + b) * 2^10 (out_proj + F.normalize + the operand split of the fine similarity,
    PEM/model/fine_point_matching.py:70-72, PEM/utils/model_utils.py:141-142) against float64."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(M)
    w = (torch.rand(256, 256, generator=gen) * 2 - 1) / 16
    b = (torch.rand(256, generator=gen) * 2 - 1) / 16 * xs
    x = torch.randn(M, 256, generator=gen) * xs
    x[M // 2] *= 40.0
    L = pem.Linear(w.to(dev), b.to(dev))
    fh, fl = pem.linear_norm_split(x.to(dev).contiguous(), L)
    got = (fh.float() + fl.float()).cpu().double().reshape(M, 256) / 1024.0
    y = x.double() @ w.double().t() + b.double()
    want = y / y.norm(dim=1, keepdim=True).clamp_min(1e-12)
    err = float((got - want).abs().max())
    assert torch.isfinite(got).all() and err < 3e-7, "linear + normalize + split vs fp64: %.3e" % err
    # the lo half is a genuine second half: at most 2^-10 of the hi half's magnitude scale, and the pair carries >= 21 bits
    assert float(fl.float().abs().max()) <= float(fh.float().abs().max()) * 2.0 ** -10 + 1e-3


@pytest.mark.parametrize("clouds,rpb,N,xs", [(4, 196, 512, 1.0), (3, 197, 256, 1.0), (1, 1, 256, 1.0), (2, 130, 512, 3e3), (64, 196, 256, 1e-3)])
def test_rows_linear_vs_fp64(dev, clouds, rpb, N, xs):
    """sam6d_rows_linear: y = x W^T + b for the token rows of a buffer with a skipped bg slot on the input side and on the output side
    (the sparse-token projections; PEM/model/coarse_point_matching.py:35-38, PEM/model/transformer.py:556-558) against float64."""
    from sam6d_hip import _lib, pem
    if _lib.load().sam6d_get_matmul_mode() == 0:
        pytest.skip("the panel kernel serves the split-precision modes (matmul mode 0 routes these projections to the exact GEMM)")
    gen = torch.Generator().manual_seed(clouds * 1000 + rpb + N)
    w = (torch.rand(N, 256, generator=gen) * 2 - 1) / 16
    b = (torch.rand(N, generator=gen) * 2 - 1) * xs
    x = torch.randn(clouds, rpb + 1, 256, generator=gen) * xs   # row 0 of every cloud: the bg slot, skipped
    x[:, 0] = float("nan")
    L = pem.Linear(w.to(dev), b.to(dev))
    out = torch.full((clouds, rpb + 2, N), 7.0, device=dev)     # rows 0 and 1 of every cloud stay untouched
    ok = pem.rows_linear(x.to(dev).contiguous(), L, out, clouds * rpb, rpb, rpb + 1, 1, rpb + 2, 2)
    assert ok
    got = out.cpu().double()
    want = x[:, 1:].double() @ w.double().t() + b.double()
    assert torch.all(got[:, :2] == 7.0)
    err = float((got[:, 2:] - want).abs().max())
    assert torch.isfinite(got).all() and err < 2e-6 * max(1.0, float(want.abs().max())), "rows_linear vs fp64: %.3e" % err


def test_geometric_transformer_writes_stacked_halves(dev):
    """The two sequential cross layers write their halves of the stacked (2B, n, 256) result in place: equal to the layers called one
    by one (PEM/model/transformer.py:517-524: feats1 attends to the already-updated feats0)."""
    from sam6d_hip import pem, synth
    W = pem.PemWeights(synth.make_pem_weights(1), dev)
    T = W.coarse["blocks"][1]
    gen = torch.Generator().manual_seed(8)
    B, n = 3, 197
    S = torch.randn(2 * B, n, 256, generator=gen).to(dev)
    pts = (torch.rand(2 * B, n - 1, 3, generator=gen) - 0.5).to(dev)
    pb = torch.cat([torch.full((2 * B, 1, 3), 100.0, device=dev), pts], 1).contiguous()
    E = pem.geo_embedding(pb, W)
    got = pem.geometric_transformer(S, E, T)
    s1 = pem.rpe_self_layer(S, E, T["self"])
    f0 = pem.cross_layer(s1[:B].contiguous(), s1[B:].contiguous(), T["cross"])
    f1 = pem.cross_layer(s1[B:].contiguous(), f0, T["cross"])
    assert torch.equal(got[:B], f0) and torch.equal(got[B:], f1)


def test_cross_layer_fused_matches_unfused(dev, monkeypatch):
    from sam6d_hip import pem, synth
    W = pem.PemWeights(synth.make_pem_weights(1), dev)
    L = W.coarse["blocks"][0]["cross"]
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(5, 197, 256, generator=gen).to(dev)
    mem = torch.randn(5, 197, 256, generator=gen).to(dev)
    a = pem.cross_layer(x, mem, L)
    monkeypatch.setenv("SAM6D_FUSED_BLOCK", "0")
    b = pem.cross_layer(x, mem, L)
    assert float((a - b).abs().max()) < 3e-5


# ------------------------------------------------------------------------------------ RPE front and pre-split weight GEMM
def test_rpe_front_vs_fp64(dev):
    """sam6d_rpe_front (qkv projection + proj_p fold + D_c fold of the query, block.hip) against float64:
    qkv = x Wqkv^T + b; qp[n,h,:] = Wp[64h:64h+64, :]^T q_h; qd[n,h,:] = D_c^T qp[n,h,:]   (PEM/model/transformer.py:395-405)."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(77)
    M = 1000
    mk = lambda o, i: pem.Linear((torch.rand(o, i, generator=gen) * 2 - 1) / math.sqrt(i), (torch.rand(o, generator=gen) * 2 - 1) / math.sqrt(i))
    qkv = mk(768, 256)
    Wp = (torch.rand(256, 256, generator=gen) * 2 - 1) / 16
    dcT = torch.randn(32, 256, generator=gen) * torch.logspace(0, -6, 32).reshape(32, 1)  # decaying coefficient rows, like D_c
    x = torch.randn(M, 256, generator=gen)
    d = lambda t: t.double()
    w_qkv = d(x) @ d(qkv.w).t() + d(qkv.b)
    q = w_qkv[:, :256]
    w_qp = torch.stack([q[:, 64 * h:64 * h + 64] @ d(Wp)[64 * h:64 * h + 64, :] for h in range(4)], 1)  # (M, 4, 256)
    w_qd = w_qp @ d(dcT).t()                                                                            # (M, 4, 32)
    L = dict(qkv=pem.Linear(qkv.w.to(dev), qkv.b.to(dev)), wpT=Wp.t().contiguous().to(dev))
    fr = pem.pack_rpe_front(L, dcT.to(dev).contiguous())
    xd = x.to(dev)
    o_qkv = torch.full((M, 768), float("nan"), device=dev); o_qp = torch.full((M, 1024), float("nan"), device=dev)
    o_qd = torch.full((M * 4, 32), float("nan"), device=dev)
    _lib.call("sam6d_rpe_front", xd.data_ptr(), fr["img"].data_ptr(), L["qkv"].b.data_ptr(), fr["inv"][0], fr["inv"][1], fr["inv"][2],
              o_qkv.data_ptr(), o_qp.data_ptr(), o_qd.data_ptr(), M, torch.cuda.current_stream().cuda_stream)
    for got, want, what in ((o_qkv, w_qkv, "qkv"), (o_qp.reshape(M, 4, 256), w_qp, "qp"), (o_qd.reshape(M, 4, 32), w_qd, "qd")):
        g = got.cpu().double()
        assert torch.isfinite(g).all(), what
        err = float((g - want).abs().max()) / float(want.abs().max())
        assert err < 2e-6, "%s: relative error %.2e" % (what, err)


@pytest.mark.parametrize("Bp,n", [(3, 197), (2, 50), (5, 64)])
def test_rpe_front_transposed_values(dev, Bp, n):
    """sam6d_rpe_front_vt writes the values transposed per cloud (the P.v operand) instead of the v third of qkv: the same bits as
    sam6d_rpe_front + sam6d_transpose, q / k / qp / qd unchanged."""
    from sam6d_hip import _lib, pem
    gen = torch.Generator().manual_seed(Bp * 10 + n)
    M = Bp * n
    mk = lambda o, i: pem.Linear((torch.rand(o, i, generator=gen) * 2 - 1) / math.sqrt(i), (torch.rand(o, generator=gen) * 2 - 1) / math.sqrt(i))
    qkv = mk(768, 256)
    L = dict(qkv=pem.Linear(qkv.w.to(dev), qkv.b.to(dev)), wpT=((torch.rand(256, 256, generator=gen) * 2 - 1) / 16).to(dev))
    fr = pem.pack_rpe_front(L, (torch.randn(32, 256, generator=gen) * 0.1).to(dev).contiguous())
    x = torch.randn(M, 256, generator=gen).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    ldp = (n + 3) // 4 * 4
    a = [torch.zeros(M, 768, device=dev), torch.zeros(M, 1024, device=dev), torch.zeros(M * 4, 32, device=dev)]
    b = [torch.zeros(M, 768, device=dev), torch.zeros(M, 1024, device=dev), torch.zeros(M * 4, 32, device=dev)]
    _lib.call("sam6d_rpe_front", x.data_ptr(), fr["img"].data_ptr(), L["qkv"].b.data_ptr(), fr["inv"][0], fr["inv"][1], fr["inv"][2],
              a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), M, st)
    vT_ref = torch.zeros(Bp, 256, ldp, device=dev)
    _lib.call("sam6d_transpose", pem._p(a[0], 512), 768, n * 768, Bp, n, 256, pem._p(vT_ref), ldp, 256 * ldp, st)
    vT = torch.zeros(Bp, 256, ldp, device=dev)
    _lib.call("sam6d_rpe_front_vt", x.data_ptr(), fr["img"].data_ptr(), L["qkv"].b.data_ptr(), fr["inv"][0], fr["inv"][1], fr["inv"][2],
              b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), M, vT.data_ptr(), n, ldp, st)
    torch.cuda.synchronize()
    assert torch.equal(vT, vT_ref), "transposed values"
    assert torch.equal(a[0][:, :512], b[0][:, :512]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert float(b[0][:, 512:].abs().max()) == 0.0  # the v third of qkv is left untouched


@pytest.mark.parametrize("M,N,K,ws", [(700, 256, 256, 1.0), (5000, 768, 256, 1.0e-3), (300, 32, 256, 40.0), (129, 130, 96, 1.0)])
def test_gemm_presplit_weights_vs_fp64(dev, M, N, K, ws):
    """sam6d_gemm_nt_w16 (weights cut into fp16 hi / lo once, with a power-of-two pack scale) against float64 and against the
    split-on-the-fly kernel."""
    from sam6d_hip import pem
    gen = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=gen)
    lin = pem.Linear(((torch.rand(N, K, generator=gen) * 2 - 1) * ws).to(dev), torch.randn(N, generator=gen).to(dev))
    want = A.double() @ lin.w.cpu().double().t() + lin.b.cpu().double()
    Ad = A.to(dev)
    a = torch.empty(M, N, device=dev); b = torch.empty(M, N, device=dev)
    pem.gemm(Ad, lin.w, lin.b, a, M, N, K, K, K, N, w16=lin.w16())
    pem.gemm(Ad, lin.w, lin.b, b, M, N, K, K, K, N)
    scale = float(want.abs().max())
    assert float((a.cpu().double() - want).abs().max()) < 4e-6 * scale
    assert float((a - b).abs().max()) < 4e-6 * scale
