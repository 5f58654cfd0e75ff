"""Which stage interferes with a concurrent copy of itself?  Two streams run the same stage on different data; compare with solo."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
cfg = pem.DEFAULT_CFG; C = 256
def mk(seed):
    inp = synth.config2_inputs(B=16, seed=seed); d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    dp = pem._cat0(d["dense_pm"], d["dense_po"]); df = pem._cat0(d["dense_fm"], d["dense_fo"])
    sp, sf, idx = pem.sample_pts_feats(dp, df, 196)
    pb = pem._empty((32, 197, 3), dp)
    _lib.call("sam6d_prepend_bg_point", pem._p(sp), 32, 196, pem._p(pb), pem._s())
    G = pem.geo_context(pb, W)
    S = pem._tokens_with_bg(sf, W.coarse["in_proj"], W.coarse["bg"])
    D = pem.fine_static(dp, df, W, cfg)
    return dict(d=d, dp=dp, df=df, sp=sp, sf=sf, idx=idx, pb=pb, G=G, S=S, D=D)
X = [mk(3), mk(4)]
torch.cuda.synchronize()
stages = {
    "geo_context": lambda x: pem.geo_context(x["pb"], W).rows[:4000].clone(),
    "rpe_self_layer(fused)": lambda x: pem.rpe_self_layer(x["S"], x["G"], W.coarse["blocks"][0]["self"]),
    "cross_layer": lambda x: pem.cross_layer(x["S"][:16], x["S"][16:], W.coarse["blocks"][0]["cross"]),
    "fine_static(PE,in_proj)": lambda x: pem.fine_static(x["dp"], x["df"], W, cfg),
    "linear_transformer_layer": lambda x: pem.linear_transformer_layer(x["D"].clone(), x["S"], W.fine["blocks"][0]["dense"]),
    "feature_similarity+fine_Rt": lambda x: torch.cat([o.reshape(16, -1) for o in pem.compute_fine_Rt(
        pem.feature_similarity(x["D"], 16, 2049, W.fine["out_proj"], cfg["temp"]), x["dp"][:16], x["dp"][16:], x["d"]["model"], x["d"]["radius"])], 1),
    "sample_pts_feats(FPS)": lambda x: pem.sample_pts_feats(x["dp"], x["df"], 196)[1],
}
stages["geo_context"] = lambda x: pem.rpe_self_layer(x["S"], pem.geo_context(x["pb"], W), W.coarse["blocks"][0]["self"])
stages["geo_embedding(materialised)"] = lambda x: pem.geo_embedding(x["pb"], W)
stages["coarse_Rt"] = lambda x: torch.cat([o.reshape(16, -1) for o in pem.compute_coarse_Rt(
    pem.feature_similarity(x["S"], 16, 197, W.coarse["out_proj"], cfg["temp"]), x["sp"][:16], x["sp"][16:], x["d"]["model"], x["d"]["radius"], x["d"]["rand"], 6000, 300, False)[:2]], 1)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
A = torch.randn(131136, 256, device=dev); Wt = torch.randn(256, 256, device=dev); Cc = torch.empty(131136, 256, device=dev)
interf = lambda: pem.gemm(A, Wt, None, Cc, 131136, 256, 256, 256, 256, 256)
for name, fn in stages.items():
    ref = fn(X[0]).clone(); torch.cuda.synchronize()
    bad = 0
    for rep in range(15):
        for st in streams: st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[1]):
            for _ in range(30): interf()
        with torch.cuda.stream(streams[0]):
            o = fn(X[0])
        torch.cuda.synchronize()
        bad += 0 if torch.equal(o, ref) else 1
        del o
    print("%-30s beside a loop of dense GEMMs: %2d/15 wrong" % (name, bad), flush=True)
