import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "openvino-sam-6d_amd")):
    sys.path.insert(0, p)
import torch
from sam6d_hip import pem, synth
from oracle import pem_oracle as O
from tests.test_configs_gpu import _oracle_proposal, _d
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1)
keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
for seed in (23,):
    inp = synth.config2_inputs(B=2, seed=seed)
    d = {k: v.to(dev) for k, v in inp.items()}
    W = pem.PemWeights(sd, dev)
    outs = {}
    for name, kw in (("default", {}), ("materialised", dict(cfg=dict(pem.DEFAULT_CFG, fused_rpe=False))), ("exact", dict(options=pem.Options(matmul_mode=0)))):
        R, t, s, aux = pem.pem_match(*[d[k] for k in keys], W, d["rand"], return_aux=True, **kw)
        outs[name] = aux["coarse"]["atten"].cpu()
    with torch.no_grad():
        for b in range(2):
            o = _oracle_proposal(O, inp, b, sd, O.DEFAULT_CFG)
            oa = o["coarse"]["atten"]
            print("seed %d proposal %d: vs oracle: default %.2e materialised %.2e exact %.2e | default vs exact %.2e | |att| max %.2f"
                  % (seed, b, _d(outs["default"][b:b+1], oa), _d(outs["materialised"][b:b+1], oa), _d(outs["exact"][b:b+1], oa),
                     _d(outs["default"][b:b+1], outs["exact"][b:b+1]), float(oa.abs().max())), flush=True)
            # geometric indices: knn comparison
            pts = torch.cat([torch.ones(1, 1, 3) * 100, o["spm"]], 1)
            for nm, sp_ in (("scene", o["spm"]), ("template", o["spo"])):
                pts = torch.cat([torch.ones(1, 1, 3) * 100, sp_], 1)
                di, ai, knn = O.geo_embedding_indices(pts)
                G = pem.geo_context(pts.to(dev), W)
                gi = G.idx.cpu()[0]
                gk = G.keep[0].cpu()[:197 * 3].reshape(197, 3).long()
                nk = int((gk != knn[0]).sum())
                da = (gi[..., 1:] - ai[0]).abs()
                print("   %s cloud: kNN entries differing %d; d_idx max diff %.2e, a_idx max diff %.2e (pairs > 1e-3: %d)" % (nm, nk, float((gi[..., 0] - di[0]).abs().max()), float(da.max()), int((da > 1e-3).sum())), flush=True)
                if nk:
                    rows = (gk != knn[0]).any(1).nonzero().flatten().tolist()
                    dist = torch.sqrt(O.pairwise_distance(pts, pts))[0]
                    for r in rows[:4]:
                        print("      row %d: gpu knn %s oracle knn %s; dists gpu %s oracle %s" % (r, gk[r].tolist(), knn[0, r].tolist(), dist[r, gk[r]].tolist(), dist[r, knn[0, r]].tolist()), flush=True)
