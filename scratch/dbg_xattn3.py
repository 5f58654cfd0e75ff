import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib, pem
dev = torch.device("cuda:0")
B, n, m = 2, 64, 32
gen = torch.Generator().manual_seed(B * 100 + n + m)
mk = lambda o, i: pem.Linear((torch.rand(o, i, generator=gen) * 2 - 1) / math.sqrt(i), (torch.rand(o, generator=gen) * 2 - 1) / math.sqrt(i))
q = mk(256, 256)
x = torch.randn(B, n, 256, generator=gen)
kv0 = torch.randn(B, m, 512, generator=gen)
d = lambda t: t.double()
qq = d(x) @ d(q.w).t() + d(q.b)
h = 1
sl = slice(64 * h, 64 * h + 64)
Pref = torch.softmax(qq[..., sl] @ d(kv0[..., sl]).transpose(1, 2) / 8.0, dim=-1)  # (B, n, m)
qd = pem.Linear(q.w.to(dev), q.b.to(dev)); xq = pem.pack_cross_query(qd)
xd = x.to(dev).contiguous()
P = torch.zeros(B, n, m, dtype=torch.float64)
for j in range(m):
    kv = kv0.clone(); kv[..., 256:] = 0.0; kv[:, j, 256:] = 1.0
    kvd = kv.to(dev).contiguous()
    out = torch.zeros(B, n, 256, device=dev)
    _lib.call("sam6d_cross_attention", xd.data_ptr(), kvd.data_ptr(), xq["img"].data_ptr(), qd.b.data_ptr(), float(xq["inv"]), out.data_ptr(), B, n, m, torch.cuda.current_stream().cuda_stream)
    P[:, :, j] = out[..., 64 * h].cpu().double()
E = (P - Pref).abs()
bad = torch.nonzero(E > 2e-6)
print("entries with error > 2e-6:", len(bad))
for b_, t_, j_ in bad[:40].tolist():
    print(" b %d token %2d key %2d: got %.8f want %.8f  (P*2^14 = %.3f)" % (b_, t_, j_, P[b_, t_, j_], Pref[b_, t_, j_], Pref[b_, t_, j_] * 16384))
