import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib, pem
dev = torch.device("cuda:0")
for (B, n, m) in [(2, 197, 197), (3, 50, 208), (2, 64, 32), (2, 64, 16)]:
    gen = torch.Generator().manual_seed(B * 100 + n + m)
    mk = lambda o, i: pem.Linear((torch.rand(o, i, generator=gen) * 2 - 1) / math.sqrt(i), (torch.rand(o, generator=gen) * 2 - 1) / math.sqrt(i))
    q = mk(256, 256)
    x = torch.randn(B, n, 256, generator=gen)
    kv = torch.randn(B, m, 512, generator=gen)
    def ref(dt):
        d = lambda t: t.to(dt)
        qq = d(x) @ d(q.w).t() + d(q.b)
        out = torch.zeros(B, n, 256, dtype=dt); P = []
        for h in range(4):
            sl = slice(64 * h, 64 * h + 64)
            att = torch.softmax(qq[..., sl] @ d(kv[..., sl]).transpose(1, 2) / 8.0, dim=-1)
            out[..., sl] = att @ d(kv[..., 256:][..., sl])
        return out
    w64, w32 = ref(torch.float64), ref(torch.float32)
    qd = pem.Linear(q.w.to(dev), q.b.to(dev)); xq = pem.pack_cross_query(qd)
    xd, kvd = x.to(dev).contiguous(), kv.to(dev).contiguous()
    out = torch.zeros(B, n, 256, device=dev)
    _lib.call("sam6d_cross_attention", xd.data_ptr(), kvd.data_ptr(), xq["img"].data_ptr(), qd.b.data_ptr(), float(xq["inv"]), out.data_ptr(), B, n, m, torch.cuda.current_stream().cuda_stream)
    got = out.cpu().double()
    e = (got - w64).abs()
    print((B, n, m), "ours vs fp64 %.3e  torch-fp32 vs fp64 %.3e   per-head max err" % (float(e.max()), float((w32.double() - w64).abs().max())),
          [float(e[..., 64 * h:64 * h + 64].max()) for h in range(4)], "worst token", int(e.amax(dim=(0, 2)).argmax()))
