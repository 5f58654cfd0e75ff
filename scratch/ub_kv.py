#!/usr/bin/env python3
"""Times sam6d_linattn_kv_image alone (64 clouds x 196 key rows of 256 k | 256 v channels).  usage: python scratch/ub_kv.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
Bp, J, C = 64, 196, 256
kv = torch.randn(Bp, J, 2 * C, generator=g).to(dev)
scale = torch.randn(C, generator=g).to(dev)
img = torch.empty(Bp * 8 * 8192, dtype=torch.uint8, device=dev); inv = torch.empty(Bp, 4, device=dev); ksum = torch.empty(Bp, 4, 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(): _lib.call("sam6d_linattn_kv_image", kv.data_ptr(), scale.data_ptr(), Bp, J, 2 * C, J * 2 * C, img.data_ptr(), inv.data_ptr(), ksum.data_ptr(), st)
run(); torch.cuda.synchronize()
print("sha", hashlib.sha256(img.cpu().numpy().tobytes() + ksum.cpu().numpy().tobytes() + inv.cpu().numpy().tobytes()).hexdigest()[:16])
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): run()
torch.cuda.synchronize(); a.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print("linattn_kv_image (64 x 196): %.1f us per launch" % (a.elapsed_time(e) / 20 * 1e3))
