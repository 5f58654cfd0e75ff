"""pem_match at odd batch sizes: default (fused) path against the materialised-embedding path and against itself run twice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
for B in (1, 5, 33, 64):
    d = {k: v.to(dev).contiguous() for k, v in synth.kat_inputs(B=B, seed=100 + B).items() if torch.is_tensor(v)}
    run = lambda cfg=pem.DEFAULT_CFG: [o.cpu() for o in pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"], cfg=cfg)]
    a = run(); b = run(); c = run(dict(pem.DEFAULT_CFG, fused_rpe=False))
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    dm = max(float((x - y).abs().max()) for x, y in zip(a, c))
    gt = d.get("R_gt")
    print("B %2d: repeat bit-identical %s; fused vs materialised max diff %.2e; finite %s" % (B, same, dm, all(torch.isfinite(x).all() for x in a)))
