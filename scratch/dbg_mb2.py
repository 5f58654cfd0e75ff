"""Bisect the multi-stream corruption: force individual op families to the exact-fp32 kernels (the mode is read at launch)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
inp = synth.config2_inputs(B=24, seed=5); d = {k: v.to(dev).contiguous() for k, v in inp.items()}
keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
lib = _lib.load()
orig_call = _lib.call
force = set()
def call(name, *a):
    fam = "gemm" if name.startswith("sam6d_gemm") else "pe" if name == "sam6d_pe_mlp_max" else "geo" if name.startswith("sam6d_geo_embed") else None
    if fam in force:
        lib.sam6d_set_matmul_mode(0)
        try:
            return orig_call(name, *a)
        finally:
            lib.sam6d_set_matmul_mode(1)
    return orig_call(name, *a)
_lib.call = call; pem._lib.call = call
def run(mb):
    cfg = dict(pem.DEFAULT_CFG, microbatch=mb, fused_rpe=False, overlap=False)
    return [o.clone() for o in pem.pem_match(*[d[k] for k in keys], W, d["rand"], cfg=cfg)]
for name, f in (("all split", set()), ("gemm exact", {"gemm"}), ("pe exact", {"pe"}), ("geo exact", {"geo"}), ("gemm+pe exact", {"gemm", "pe"})):
    force.clear(); force.update(f)
    ref = run(1); bad = 0
    for rep in range(20):
        o = run(3)
        bad += any(float((a - b).abs().max()) != 0.0 for a, b in zip(o, ref))
    print("%-14s mismatching mb=3 runs: %d/20" % (name, bad), flush=True)
