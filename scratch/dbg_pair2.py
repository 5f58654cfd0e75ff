"""geo_embedding (victim, stream 0) beside single C-ABI calls looping on stream 1."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
from sam6d_hip.pem import _p, _s, _empty
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
g = torch.Generator().manual_seed(7)
B, n, C = 32, 197, 256
def cloud():
    p = (torch.rand(B, n, 3, generator=g) - 0.5) + torch.tensor([0.3, -0.2, 8.0]); p[:, 0] = 100.0
    return p.to(dev)
pv, po = cloud(), cloud()
vref = pem.geo_embedding(pv, W).clone()
# persistent buffers of the interferer
pairs = B * n * n
knn = _empty((B * n * 3 + 1,), po, torch.int32); idx = _empty((B, n, n, 4), po); out = _empty((B, n, n, C), po)
lst = _empty((pairs + 1,), po, torch.int32); pos = _empty((pairs,), po, torch.int32)
flag = knn.data_ptr() + 4 * B * n * 3
fa = 180.0 / (15 * math.pi)
_lib.call("sam6d_geo_indices", _p(po), B, n, 0.2, fa, 3, _p(knn), _p(idx), _s())
torch.cuda.synchronize()
A = torch.randn(131136, 256, device=dev); Wt = torch.randn(256, 256, device=dev); Cc = torch.empty(131136, 256, device=dev)
calls = {
    "nothing": lambda: None,
    "geo_indices": lambda: _lib.call("sam6d_geo_indices", _p(po), B, n, 0.2, fa, 3, _p(knn), _p(idx), _s()),
    "geo_embed_cheb (classify+cheb+list)": lambda: _lib.call("sam6d_geo_embed_cheb", _p(idx), pairs, pem.geo_cheb_packed(W).data_ptr(), 24.0, _p(W.div_term), pem.geo_packed(W).data_ptr(), _p(W.geo_d.b), _p(W.geo_a.b), C, flag, _p(pos), _p(lst), _p(out), _s()),
    "geo_embed exact (exits at once)": lambda: _lib.call("sam6d_geo_embed", _p(idx), pairs, _p(W.div_term), _p(W.geo_d.w), _p(W.geo_d.b), _p(W.geo_a.w), _p(W.geo_a.b), C, flag, 1, _p(out), _s()),
    "geo_embed_h3 (full sinusoid kernel)": lambda: _lib.call("sam6d_geo_embed_h3", _p(idx), pairs, _p(W.div_term), pem.geo_packed(W).data_ptr(), _p(W.geo_d.b), _p(W.geo_a.b), C, flag, _p(out), _s()),
    "dense GEMM 131136x256x256": lambda: pem.gemm(A, Wt, None, Cc, 131136, 256, 256, 256, 256, 256),
    "torch matmul": lambda: torch.matmul(A, Wt),
}
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
gemm_fn = calls["dense GEMM 131136x256x256"]
def indices(points):
    k2 = _empty((B * n * 3 + 1,), points, torch.int32); i2 = _empty((B, n, n, 4), points)
    _lib.call("sam6d_geo_indices", _p(points), B, n, 0.2, fa, 3, _p(k2), _p(i2), _s())
    return k2, i2
kref, iref = indices(pv); kref = kref.clone(); iref = iref.clone(); torch.cuda.synchronize()
# persistent victim buffers: knn already holds the final values before every concurrent run
k2 = _empty((B * n * 3 + 1,), pv, torch.int32); i2 = _empty((B, n, n, 4), pv)
_lib.call("sam6d_geo_indices", _p(pv), B, n, 0.2, fa, 3, _p(k2), _p(i2), _s()); torch.cuda.synchronize()
lib = _lib.load()
def gemm_exact():
    lib.sam6d_set_matmul_mode(0)
    try: gemm_fn()
    finally: lib.sam6d_set_matmul_mode(1)
X2 = torch.randn(131136, 256, device=dev); Y2 = torch.empty_like(X2); gam = torch.ones(256, device=dev); bet = torch.zeros(256, device=dev)
def ln(): _lib.call("sam6d_layernorm256", _p(X2), _p(gam), _p(bet), _p(Y2), 131136, 256, 256, 1e-5, _s())
bq_pts = torch.rand(32, 2048, 3, device=dev); bq_idx = torch.empty(32, 2048, 64, dtype=torch.int32, device=dev)
def bq(): _lib.call("sam6d_ball_query", _p(bq_pts), _p(bq_pts), 32, 2048, 2048, 0.2, 64, _p(bq_idx), _s())
def cp(): _lib.call("sam6d_copy_f32", _p(X2), _p(Y2), X2.numel(), _s())
for label, interf in (("my GEMM, split fp16", gemm_fn), ("my GEMM, exact fp32", gemm_exact), ("layernorm256 (no LDS)", ln), ("ball_query (48 KB LDS)", bq), ("copy_f32", cp), ("torch matmul", calls["torch matmul"])):
    bad = 0; cnt = []
    for rep in range(20):
        i2.fill_(-1.0); torch.cuda.synchronize()
        for st in streams: st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[1]):
            for _ in range(4): interf()
        with torch.cuda.stream(streams[0]):
            _lib.call("sam6d_geo_indices", _p(pv), B, n, 0.2, fa, 3, _p(k2), _p(i2), _s())
        torch.cuda.synchronize()
        di = (i2 != iref).any(-1)
        if di.any():
            bad += 1; cnt.append(int(di.sum()))
    print("geo_indices (knn pre-computed, same buffers) beside %-14s: %d/20 wrong %s" % (label, bad, cnt[:5]), flush=True)
sys.exit(0)
