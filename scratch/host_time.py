#!/usr/bin/env python3
"""Host (Python + launch) time of one pem_match call vs its GPU time, B = 32.  usage: python scratch/host_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
d = {k: v.to(dev).contiguous() for k, v in synth.config2_inputs(B=32, seed=1).items()}
f = lambda: pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"])
for _ in range(3): f()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host %.2f ms, until GPU done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
