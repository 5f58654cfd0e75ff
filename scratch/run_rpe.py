"""Micro-benchmark of the fused RPE score kernel (Bp = 64 clouds, n = 197) + correctness vs the materialised layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib
if os.environ.get('LIBP'): _lib.LIB_PATH = os.environ['LIBP']
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1)
W = pem.PemWeights(sd, dev)
g = torch.Generator().manual_seed(5)
Bp, n = 64, 197
pts = (torch.rand(Bp, n, 3, generator=g) - 0.5) + torch.tensor([0.3, -0.2, 8.0])
pts[:, 0] = 100.0
pts = pts.to(dev)
x = torch.randn(Bp, n, 256, generator=g).to(dev)
L = W.coarse["blocks"][0]["self"]
G = pem.geo_context(pts, W)
got = pem.rpe_self_layer(x, G, L)
if os.environ.get("CHECK", "1") == "1":
    E = pem.geo_embedding(pts, W)
    want = pem.rpe_self_layer(x, E, L)
    print("fused vs materialised: %.2e" % float((got - want).abs().max()))
    del E, want
M = Bp * n; C = 256; H = 4; ldp = 200
x2 = x.reshape(M, C)
qkv = pem.linear(x2, L["qkv"])
qp = pem._empty((M, H * C), x)
pem.gemm(qkv, L["wpT"], None, qp, M, C, 64, 3 * C, C, H * C, batch=H, sA=64, sW=64, sC=C)
qd = pem._empty((M * H, 32), x)
pem.gemm(qp, G.dcT, None, qd, M * H, 32, C, C, C, 32)
qk = pem._empty((M, H, ldp), x)
pem.gemm_b2(qkv, qkv, qk, n, n, 64, 3 * C, 3 * C, H * ldp, Bp, n * 3 * C, n * 3 * C, n * H * ldp, H, 64, 64, ldp, w_off=C)
P = pem._empty((M, H, ldp), x)
def run(products):
    _lib.call("sam6d_rpe_scores2", pem._p(G.idx), pem._p(G.pos), pem._p(G.keep[1]), pem._p(G.rows), G.wa_cheb, float(pem.GEO_XMAX),
              float(G.xmax_a), products, pem._p(qp), pem._p(qd), pem._p(qk), pem._p(P), M, n, ldp, pem._s())
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
for products in (3, 2, 3, 2):
    for _ in range(3): run(products)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10): run(products)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("rpe_scores, %d products: %.1f us per launch (incl. the listed-pair kernel)" % (products, ms * 1e3))
a.record()
for _ in range(10): pem.rpe_self_layer(x, G, L)
b.record(); torch.cuda.synchronize()
print("whole fused layer: %.1f us" % (a.elapsed_time(b) * 100))
