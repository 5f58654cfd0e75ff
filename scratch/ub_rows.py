import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for N in (512, 256):
    L = pem.Linear(((torch.rand(N, 256, generator=g) - 0.5) / 8).to(dev), torch.zeros(N).to(dev))
    L.w16(); L.pimg()
    S = torch.randn(64, 197, 256, generator=g).to(dev)
    out = torch.empty(64, 196, N, device=dev)
    def a(): pem.rows_linear(S, L, out, 64 * 196, 196, 197, 1, 196, 0)
    def b(): pem.gemm(S, L.w, L.b, out, 196, N, 256, 256, 256, N, a_off=256, batch=64, sA=197 * 256, sC=196 * N, w16=L.w16())
    for name, f in (("panel", a), ("gemm", b), ("panel", a), ("gemm", b)):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        print("N=%d %-6s %.1f us" % (N, name, e0.elapsed_time(e1) * 1000 / 50))
