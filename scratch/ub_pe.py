import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib
if os.environ.get("LIBP"): _lib.LIB_PATH = os.environ["LIBP"]
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
g = torch.Generator().manual_seed(0)
pts = (torch.rand(32, 2048, 3, generator=g) - 0.5).to(dev)
grp = pem.pe_group(pts)
D = torch.zeros(32, 2049, 256, device=dev)
def f(): pem.pe_apply(pts, grp, W, D, 256, 2049 * 256)
for _ in range(3): f()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f()
b.record(); torch.cuda.synchronize()
print("pe_apply (2 MLP kernels + mlp3 GEMM): %.1f us" % (a.elapsed_time(b) * 1000 / 20))
