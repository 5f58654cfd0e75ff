"""Batched GEMM probe (fine similarity 32 x 2049 x 2049 x 256; RPE q.k^T-like shapes): SAM6D_AB_LIB selects the library."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'openvino-sam-6d_amd')]
from sam6d_hip import _lib
if os.environ.get("SAM6D_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SAM6D_AB_LIB"])
import torch
from sam6d_hip import pem
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (B, n, K) in ((32, 2049, 256), (64, 197, 256), (256, 197, 64)):
    f = torch.randn(2 * B, n, K, generator=g).to(dev)
    att = torch.empty(B, n, n, device=dev)
    def run():
        pem.gemm(f, f, None, att, n, n, K, K, K, n, w_off=B * n * K, batch=B, sA=n * K, sW=n * K, sC=n * n, divisor=0.1)
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): run()
    t1.record(); torch.cuda.synchronize()
    print("batched gemm B=%d n=%d K=%d: %.1f us" % (B, n, K, t0.elapsed_time(t1) / 20 * 1e3))
