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
x0 = torch.randn(B, n, 256, generator=gen)
kv0 = torch.randn(B, m, 512, generator=gen)
def run(x, kv, tag):
    d = lambda t: t.double()
    qq = d(x) @ d(q.w).t() + d(q.b)
    w64 = torch.zeros(B, n, 256, dtype=torch.float64)
    for h in range(4):
        sl = slice(64 * h, 64 * h + 64)
        att = torch.softmax(qq[..., sl] @ d(kv[..., sl]).transpose(1, 2) / 8.0, dim=-1)
        w64[..., sl] = att @ d(kv[..., 256:][..., sl])
    qd = pem.Linear(q.w.to(dev), q.b.to(dev)); xq = pem.pack_cross_query(qd)
    xd, kvd = x.to(dev).contiguous(), kv.to(dev).contiguous()
    out = torch.zeros(B, n, 256, device=dev)
    _lib.call("sam6d_cross_attention", xd.data_ptr(), kvd.data_ptr(), xq["img"].data_ptr(), qd.b.data_ptr(), float(xq["inv"]), out.data_ptr(), B, n, m, torch.cuda.current_stream().cuda_stream)
    e = (out.cpu().double() - w64).abs()
    print("%-28s max err %.3e  per head" % (tag, float(e.max())), ["%.1e" % float(e[..., 64 * h:64 * h + 64].max()) for h in range(4)],
          "worst (b,tok,ch)", [int(v) for v in torch.nonzero(e == e.max())[0]])
run(x0, kv0, "random")
kv = kv0.clone(); kv[..., 256:] = 1.0; run(x0, kv, "v = 1")
kv = kv0.clone(); kv[..., :256] = 0.0; run(x0, kv, "k = 0 (uniform P)")
kv = kv0.clone(); kv[..., 256:] = torch.arange(m).float().reshape(1, m, 1).expand(B, m, 256); run(x0, kv, "v = key index")
run(torch.zeros_like(x0), kv0, "x = 0 (q = bias)")
