"""linear_norm_split (fine out_proj + normalize + split) at M = 2 x 32 x 2049 rows: time per launch and a checksum (A/B over
SAM6D_STREAM_LINEAR in separate processes: the switch is read once)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
M = 2 * 32 * 2049
x = torch.randn(M, 256, generator=g).to(dev)
w = (torch.randn(256, 256, generator=g) / 16).to(dev); b = (torch.randn(256, generator=g) * 0.1).to(dev)
L = pem.Linear(w, b)
img = pem.pack_cross_query(L)
def run():
    return pem.linear_norm_split(x, L, img)
fh, fl = run(); torch.cuda.synchronize()
y = x.double().cpu()[:4096] @ w.double().cpu().t() + b.double().cpu()
want = y / y.norm(dim=1, keepdim=True).clamp_min(1e-12)
got = (fh.float() + fl.float()).cpu().double().reshape(M, 256)[:4096] / 1024.0
print("max err vs fp64 (first 4096 rows): %.2e" % float((got - want).abs().max()))
print("sha", hashlib.sha256(fh.cpu().numpy().tobytes() + fl.cpu().numpy().tobytes()).hexdigest()[:16])
for _ in range(3): run()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print("linear_norm_split M=%d: %.1f us per launch (%.2f TB/s of 2 x M x 1 KiB)" % (M, a.elapsed_time(e) / 20 * 1e3, 2 * M * 1024 / (a.elapsed_time(e) / 20 * 1e-3) / 1e12))
