"""Stress: single pipeline with / without the side stream, many repetitions, bitwise vs the first result."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
inp = synth.config2_inputs(B=int(os.environ.get("NB", "32")), seed=5); d = {k: v.to(dev).contiguous() for k, v in inp.items()}
keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
def run(ov, mb=1):
    cfg = dict(pem.DEFAULT_CFG, microbatch=mb, overlap=ov)
    return [o.clone() for o in pem.pem_match(*[d[k] for k in keys], W, d["rand"], cfg=cfg)]
ref = run(False)
reps = int(os.environ.get("REPS", "150"))
for ov, mb in ((True, 1), (False, 1), (True, 2)):
    bad = 0
    for rep in range(reps):
        o = run(ov, mb)
        bad += any(float((a - b).abs().max()) != 0.0 for a, b in zip(o, ref))
    print("overlap %s microbatch %d: %d/%d runs differ from the serial reference" % (ov, mb, bad, reps), flush=True)
