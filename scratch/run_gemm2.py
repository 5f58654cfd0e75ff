"""Shape probe for the dense-token GEMM: is the time per tile or per byte?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'openvino-sam-6d_amd')]
import torch
from sam6d_hip import pem, _lib
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (M, N, K, res) in ((131136, 256, 256, True), (131136, 256, 256, False), (131136, 128, 256, True), (131136, 128, 256, False),
                       (65568, 256, 256, True), (131136, 256, 128, True), (131136, 256, 64, True), (131136, 512, 256, False)):
    A = torch.randn(M, K, generator=g).to(dev); Wt = (torch.randn(N, K, generator=g) / 16).to(dev); b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev) if res else None
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        pem.gemm(A, Wt, b, out, M, N, K, K, K, N, residual=R, ldr=N)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        pem.gemm(A, Wt, b, out, M, N, K, K, K, N, residual=R, ldr=N)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 20
    by = (M * K + (2 if res else 1) * M * N + N * K) * 4
    print("gemm %dx%dx%d res=%d: %.1f us  %.1f TFLOP/s  %.2f TB/s (%.0f MB)" % (M, N, K, res, ms * 1e3, 2.0 * M * N * K / ms / 1e9, by / ms / 1e9, by / 1e6))
