"""Where does the multi-stream corruption first appear?  Run the stages of 3 slices on 3 streams, keep every intermediate,
compare with the same slice run alone."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
inp = synth.config2_inputs(B=24, seed=5); dall = {k: v.to(dev).contiguous() for k, v in inp.items()}
cfg = dict(pem.DEFAULT_CFG)
C = 256
def stages(d):
    out = collections.OrderedDict()
    B = d["dense_pm"].shape[0]
    dp = pem._cat0(d["dense_pm"], d["dense_po"]); df = pem._cat0(d["dense_fm"], d["dense_fo"])
    sp, sf, idx = pem.sample_pts_feats(dp, df, 196)
    out["fps_idx"] = idx.clone(); out["sf"] = sf.clone()
    pb = pem._empty((2 * B, 197, 3), dp)
    _lib.call("sam6d_prepend_bg_point", pem._p(sp), 2 * B, 196, pem._p(pb), pem._s())
    E = pem.geo_embedding(pb, W)
    out["E"] = E.clone()
    S = pem._tokens_with_bg(sf, W.coarse["in_proj"], W.coarse["bg"]); out["S0"] = S.clone()
    for i, blk in enumerate(W.coarse["blocks"]):
        S = pem.geometric_transformer(S, E, blk); out["S%d" % (i + 1)] = S.clone()
    att = pem.feature_similarity(S, B, 197, W.coarse["out_proj"], cfg["temp"]); out["att"] = att.clone()
    c = pem.compute_coarse_Rt(att, sp[:B], sp[B:], d["model"], d["radius"], d["rand"], 6000, 300, False)
    out["R0"] = c[0].clone()
    D = pem.fine_static(dp, df, W, cfg); out["D0"] = D.clone()
    N = dp.shape[1]
    p1 = pem._empty((B, N, 3), dp)
    _lib.call("sam6d_rigid_inverse", pem._p(dp), pem._p(c[0]), pem._p(c[1]), B, N, pem._p(p1), pem._s())
    pem.positional_encoding_add(p1, W, D, C, (N + 1) * C, 0.1, 0.2, 32, 64); out["D0pe"] = D.clone()
    for i, blk in enumerate(W.fine["blocks"]):
        D = pem.sparse_to_dense_transformer(D, E, idx, blk); out["D%d" % (i + 1)] = D.clone()
    att2 = pem.feature_similarity(D, B, N + 1, W.fine["out_proj"], cfg["temp"]); out["att2"] = att2[:, :64].clone()
    R, t, s = pem.compute_fine_Rt(att2, dp[:B], dp[B:], d["model"], d["radius"], cfg["dis_thres"])
    out["R"] = R.clone()
    return out
slices = [{k: v[i * 8:(i + 1) * 8].contiguous() for k, v in dall.items()} for i in range(3)]
refs = [stages(s) for s in slices]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
first = collections.Counter()
for rep in range(16):
    outs = []
    for st, s in zip(streams, slices):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            outs.append(stages(s))
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, refs)):
        for k in o:
            if not torch.equal(o[k], r[k]):
                dd = (o[k].float() - r[k].float()).abs()
                first[k] += 1
                print("rep %d slice %d: first mismatch at %-6s  max %.3e, %d of %d values" % (rep, i, k, float(dd.max()), int((dd > 0).sum()), dd.numel()), flush=True)
                if k == "E":
                    rows = (dd > 0).any(-1)  # (2B, n, n)
                    idx = rows.nonzero()
                    full = int(((dd > 0).sum(-1) == 256).sum())
                    bg = ((idx[:, 1] == 0) | (idx[:, 2] == 0)).float().mean()
                    print("    rows wrong %d (fully wrong %d), bg share %.2f, clouds %s" % (idx.shape[0], full, float(bg), sorted(set(idx[:, 0].tolist()))))
                    print("    first rows", idx[:5].tolist(), "last rows", idx[-3:].tolist())
                    b, i2, j2 = idx[0].tolist()
                    print("    got", o[k][b, i2, j2, :6].tolist()); print("    ref", r[k][b, i2, j2, :6].tolist())
                    # does the wrong row equal the right row of another pair (misplaced write)?
                    flat = r[k].reshape(-1, 256); w = o[k][b, i2, j2]
                    hit = (flat == w).all(-1).nonzero().flatten().tolist()[:5]
                    print("    wrong row equals reference row(s):", hit, " (own flat index %d)" % ((b * 197 + i2) * 197 + j2))
                break
print(dict(first))
