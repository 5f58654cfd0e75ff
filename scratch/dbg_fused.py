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
cfg = pem.DEFAULT_CFG
B = 3
dp = pem._cat0(d["dense_pm"], d["dense_po"]); df = pem._cat0(d["dense_fm"], d["dense_fo"])
n = cfg["coarse_npoint"]
sp, sf, idx = pem.sample_pts_feats(dp, df, n)
pb = pem._empty((2 * B, n + 1, 3), dp)
_lib.call("sam6d_prepend_bg_point", pem._p(sp), 2 * B, n, pem._p(pb), pem._s())
E = pem.geo_embedding(pb, W); G = pem.geo_context(pb, W)
torch.cuda.synchronize()
print("listed pairs:", int(G.keep[1][0]), "of", G.pos.numel(), " pos>=0:", int((G.pos >= 0).sum()))
S0 = pem._tokens_with_bg(sf, W.coarse["in_proj"], W.coarse["bg"])
Sa, Sb = S0, S0
for i, blk in enumerate(W.coarse["blocks"]):
    Ta = pem.rpe_self_layer(Sa, E, blk["self"]); Tb = pem.rpe_self_layer(Sa, G, blk["self"])
    print("block", i, "self layer on same input: max diff %.3e" % float((Ta - Tb).abs().max()), "scale %.2f" % float(Ta.abs().max()))
    Sa = pem.geometric_transformer(Sa, E, blk); Sb = pem.geometric_transformer(Sb, G, blk)
    print("block", i, "chain diff %.3e" % float((Sa - Sb).abs().max()))
for name, EE in (("mat", E), ("fused", G)):
    c = pem.coarse_point_matching(sp, sf, EE, d["radius"], d["model"], W, d["rand"], cfg)
    print(name, "coarse R0[0]", c[0][0].flatten().tolist()[:4], "t0", c[1][0].tolist())
outs = {}
for name, EE in (("mat", E), ("fused", G)):
    c = pem.coarse_point_matching(sp, sf, EE, d["radius"], d["model"], W, d["rand"], cfg)
    f = pem.fine_point_matching(dp, df, EE, idx, d["radius"], d["model"], c[0], c[1], W, cfg)
    outs[name] = (c[0].cpu(), c[1].cpu(), f[0].cpu(), f[1].cpu(), f[2].cpu())
for i, what in enumerate(("R0", "t0", "R", "t", "score")):
    a, b = outs["mat"][i], outs["fused"][i]
    print(what, "per-proposal max diff", (a - b).abs().reshape(a.shape[0], -1).amax(1).tolist())
# fine stage with the SAME init pose for both
c = pem.coarse_point_matching(sp, sf, E, d["radius"], d["model"], W, d["rand"], cfg)
fa = pem.fine_point_matching(dp, df, E, idx, d["radius"], d["model"], c[0], c[1], W, cfg)
fb = pem.fine_point_matching(dp, df, G, idx, d["radius"], d["model"], c[0], c[1], W, cfg)
for i, what in enumerate(("R", "t", "score")):
    print("fine-only", what, (fa[i] - fb[i]).abs().reshape(3, -1).amax(1).tolist())
for ov in (True, False):
    res = []
    for fused in (True, False):
        cc = dict(cfg, fused_rpe=fused, overlap=ov)
        res.append(pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"], cfg=cc))
    print("pem_match overlap", ov, [float((a - b).abs().max()) for a, b in zip(*res)])
