"""Fine-stage similarity + soft assignment at BASELINE config 5's size (4096 dense points): pipeline (finematch.hip) vs launch-per-op."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for B, n in ((16, 4097), (32, 2049)):
    F = torch.randn(2 * B, n, 256, generator=g).to(dev)
    pts1 = (torch.rand(B, n - 1, 3, generator=g) - 0.5).to(dev); pts2 = (torch.rand(B, n - 1, 3, generator=g) - 0.5).to(dev)
    model = (torch.rand(B, 1024, 3, generator=g) - 0.5).to(dev); radius = torch.ones(B, device=dev)
    f = F.reshape(-1, 256).contiguous()
    def fused(): return pem.compute_fine_Rt_fused(f, B, n, 0.1, pts1, pts2, model, radius)
    def unfused():
        x = f.clone()
        pem._lib.call("sam6d_l2norm256", x.data_ptr(), x.data_ptr(), 2 * B * n, 256, 256, torch.cuda.current_stream().cuda_stream)
        att = torch.empty(B, n, n, device=dev)
        pem.gemm(x, x, None, att, n, n, 256, 256, 256, n, w_off=B * n * 256, batch=B, sA=n * 256, sW=n * 256, sC=n * n, divisor=0.1)
        return pem.compute_fine_Rt(att, pts1, pts2, model, radius)
    for name, fn in (("pipeline", fused), ("launch-per-op", unfused)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): fn()
        b.record(); torch.cuda.synchronize()
        print("B %d n %d %-14s %.3f ms" % (B, n, name, a.elapsed_time(b) / 5))
