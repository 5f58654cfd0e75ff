#!/usr/bin/env python3
"""Times the dense-token projections of the fine stage on sam6d_gemm_nt_w16: in_proj (32 clouds x 2048 rows, 256 -> 256, output rows
strided by the background slot) and mlp3 (the same with a residual), and checks them against float64.  usage: python scratch/ub_gemm.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
B, N, C = 32, 2048, 256
x = torch.randn(B, N, C, generator=g).to(dev)
lin = pem.Linear(((torch.rand(C, C, generator=g) * 2 - 1) / 16).to(dev), ((torch.rand(C, generator=g) * 2 - 1) / 16).to(dev))
D = torch.zeros(B, N + 1, C, device=dev)
def inproj(): pem.gemm(x, lin.w, lin.b, D, N, C, C, C, C, C, c_off=C, batch=B, sA=N * C, sC=(N + 1) * C, w16=lin.w16())
def mlp3(): pem.gemm(x, lin.w, lin.b, D, N, C, C, C, C, C, c_off=C, residual=D, r_off=C, ldr=C, batch=B, sA=N * C, sC=(N + 1) * C, sR=(N + 1) * C, w16=lin.w16())
inproj(); torch.cuda.synchronize()
want = (x[:2].double().cpu() @ lin.w.double().cpu().T + lin.b.double().cpu())
got = D[:2, 1:].double().cpu()
print("in_proj max err vs fp64 (2 clouds): %.2e   sha %s" % (float((got - want).abs().max()), hashlib.sha256(D.cpu().numpy().tobytes()).hexdigest()[:16]))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / n * 1e3
u = t(inproj); print("in_proj 65536 x 256 x 256: %.1f us  (%.2f TB/s of 134 MB)" % (u, 134.2e6 / u / 1e6))
D.zero_(); u = t(mlp3); print("mlp3 (+ residual)        : %.1f us  (%.2f TB/s of 201 MB)" % (u, 201.3e6 / u / 1e6))
