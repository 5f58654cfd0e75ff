#!/usr/bin/env python3
"""Wall time of the first 16 pem_match steps (B = 32) in a fresh process: shows the allocator / first-use transient."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
d = {k: v.to(dev).contiguous() for k, v in synth.config2_inputs(B=32, seed=1).items()}
torch.cuda.synchronize()
ts = []
for i in range(16):
    t0 = time.perf_counter()
    pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"])
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in ts))
print("reserved %.2f GB, allocated %.2f GB" % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9))
