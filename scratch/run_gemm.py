import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'openvino-sam-6d_amd')]
import torch
from sam6d_hip import pem, _lib
dev = torch.device('cuda:0')
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_lib.call("sam6d_set_matmul_mode", mode)
g = torch.Generator().manual_seed(0)
for (M, N, K) in ((131136, 256, 256), (131136, 512, 256), (131136, 256, 512), (12608, 768, 256), (6304, 256, 256)):
    A = torch.randn(M, K, generator=g).to(dev); Wt = (torch.randn(N, K, generator=g) / 16).to(dev); b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        pem.gemm(A, Wt, b, out, M, N, K, K, K, N, residual=R, ldr=N)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        pem.gemm(A, Wt, b, out, M, N, K, K, K, N, residual=R, ldr=N)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 10
    print("mode %d gemm %dx%dx%d: %.1f us  %.1f TFLOP/s  %.2f TB/s" % (mode, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, (M * K + 2 * M * N + N * K) * 4 / ms / 1e9))
