"""Uninitialised-read probe: every buffer the host code allocates is pre-filled with NaN (floats) / a huge value (ints);
a kernel that consumes memory nobody wrote shows up as a changed or non-finite result."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1)
W = pem.PemWeights(sd, dev)
inp = synth.config2_inputs(B=3, seed=11)
d = {k: v.to(dev).contiguous() for k, v in inp.items()}
run = lambda cfg: pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"], cfg=cfg)
clean = {}
for fused in (True, False):
    clean[fused] = [o.clone() for o in run(dict(pem.DEFAULT_CFG, fused_rpe=fused, overlap=False))]
orig = pem._empty
def poisoned(shape, like, dtype=torch.float32):
    t = orig(shape, like, dtype)
    if dtype.is_floating_point:
        t.fill_(float("nan"))
    else:
        t.fill_(0x3fffffff if dtype != torch.uint8 else 255)
    return t
pem._empty = poisoned
for fused in (True, False):
    out = run(dict(pem.DEFAULT_CFG, fused_rpe=fused, overlap=False))
    print("fused", fused, "poisoned-vs-clean:", [float((a - b).abs().max()) for a, b in zip(out, clean[fused])], "finite:", [bool(torch.isfinite(a).all()) for a in out], flush=True)
# stage-wise for the materialised path
B = 3; cfg = pem.DEFAULT_CFG
def stages(empty):
    pem._empty = empty
    dp = pem._cat0(d["dense_pm"], d["dense_po"]); df = pem._cat0(d["dense_fm"], d["dense_fo"])
    sp, sf, idx = pem.sample_pts_feats(dp, df, 196)
    pb = pem._empty((2 * B, 197, 3), dp)
    _lib.call("sam6d_prepend_bg_point", pem._p(sp), 2 * B, 196, pem._p(pb), pem._s())
    E = pem.geo_embedding(pb, W)
    S = pem._tokens_with_bg(sf, W.coarse["in_proj"], W.coarse["bg"])
    res = {"E": E.clone(), "S0": S.clone()}
    for i, blk in enumerate(W.coarse["blocks"]):
        S = pem.geometric_transformer(S, E, blk); res["S%d" % (i + 1)] = S.clone()
    att = pem.feature_similarity(S, B, 197, W.coarse["out_proj"], cfg["temp"]); res["att"] = att.clone()
    c = pem.compute_coarse_Rt(att, sp[:B], sp[B:], d["model"], d["radius"], d["rand"], 6000, 300, False)
    res["R0"] = c[0].clone(); res["t0"] = c[1].clone()
    D = pem.fine_static(dp, df, W, cfg); res["D0"] = D.clone()
    f = pem.fine_point_matching(dp, df, E, idx, d["radius"], d["model"], c[0], c[1], W, cfg, D=D)
    res["R"] = f[0].clone(); res["t"] = f[1].clone(); res["score"] = f[2].clone()
    return res
a = stages(orig); b = stages(poisoned)
for k in a:
    x, y = a[k], b[k]
    same = torch.equal(torch.nan_to_num(x, nan=12345.0), torch.nan_to_num(y, nan=12345.0))
    print("%-6s identical=%s  nan(clean)=%d nan(poisoned)=%d" % (k, same, int(torch.isnan(x).sum()), int(torch.isnan(y).sum())), flush=True)
