"""Stress probe for the micro-batch pipeline: repeat mb = 1, 2, 3 and compare with the first mb = 1 result bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
inp = synth.config2_inputs(B=24, seed=5); d = {k: v.to(dev).contiguous() for k, v in inp.items()}
keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
def run(mb, fused=True, overlap=True):
    cfg = dict(pem.DEFAULT_CFG, microbatch=mb, fused_rpe=fused, overlap=overlap)
    return [o.clone() for o in pem.pem_match(*[d[k] for k in keys], W, d["rand"], cfg=cfg)]
import collections
refs = {}
bad = collections.Counter(); tot = collections.Counter()
mode_env = os.environ.get("SAM6D_MATMUL_MODE", "1")
for rep in range(30):
    for mb, fused, ov in ((1, True, False), (3, True, False), (1, False, False), (3, False, False)):
        o = run(mb, fused, ov)
        key = fused
        if key not in refs:
            refs[key] = o
            continue
        dm = [float((a - b).abs().max()) for a, b in zip(o, refs[key])]
        tot[(mb, fused)] += 1
        if max(dm) != 0.0:
            bad[(mb, fused)] += 1
print("mode", mode_env, {k: "%d/%d" % (bad[k], tot[k]) for k in tot})
