import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'openvino-sam-6d_amd')]
import torch
from sam6d_hip import pem, synth, _lib
dev = torch.device('cuda:0')
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_lib.call("sam6d_set_matmul_mode", mode)
W = pem.PemWeights(synth.make_pem_weights(1), dev)
g = torch.Generator().manual_seed(0)
pts = (torch.rand(64, 196, 3, generator=g) - 0.5)
pts = torch.cat([torch.ones(64, 1, 3) * 100, pts], 1).to(dev)
for _ in range(3):
    E = pem.geo_embedding(pts, W)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5):
    E = pem.geo_embedding(pts, W)
t1.record(); torch.cuda.synchronize()
print("geo_embedding mode %d: %.3f ms per call" % (mode, t0.elapsed_time(t1) / 5))
