#!/usr/bin/env python3
"""Micro-benchmarks of the sparse (197-token) stage at the benchmark's shapes (B = 32 -> 64 clouds): wall time per call of a
back-to-back loop (kernel time incl. launch gaps on one stream).  usage: ub.py [names...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib
if os.environ.get("LIBP"): _lib.LIB_PATH = os.environ["LIBP"]
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
B = 32
g = torch.Generator().manual_seed(0)
S = torch.randn(2 * B, 197, 256, generator=g).to(dev)
pts = (torch.rand(2 * B, 196, 3, generator=g) - 0.5).to(dev)
pb = torch.empty(2 * B, 197, 3, device=dev)
_lib.call("sam6d_prepend_bg_point", pem._p(pts), 2 * B, 196, pem._p(pb), pem._s())
pem.geo_packed(W), pem.geo_cheb_packed(W), pem.geo_dcT(W), pem.geo_dcT16(W); pem._ensure_w16(W)
G = pem.geo_context(pb, W)
T = W.coarse["blocks"][0]
x2 = S.reshape(-1, 256)
hid = torch.randn_like(x2)


def _xattn():
    L = T["cross"]
    out = torch.empty(B, 197, 256, device=dev)
    _lib.call("sam6d_cross_attention_kv", pem._p(S[:B]), pem._p(S[B:]), L["xq"]["img"].data_ptr(), pem._p(L["q"].b), float(L["xq"]["inv"]),
              L["xkv"]["img"].data_ptr(), pem._p(L["kv"].b), float(L["xkv"]["inv"]), pem._p(out), B, 197, 197, pem._s())


_QKV = torch.randn(2 * B * 197, 768, device=dev)
_G = torch.randn(2 * B * 197, 4, 200, device=dev)


def _sattn():
    out = torch.empty(2 * B * 197, 256, device=dev)
    _lib.call("sam6d_rpe_self_attention", pem._p(_QKV), pem._p(_G), pem._p(out), 2 * B, 197, 200, pem._s())


def _front():
    L = T["self"]
    fr = L["front"]
    M = x2.shape[0]
    qkv = torch.empty(M, 768, device=dev); qp = torch.empty(M, 1024, device=dev); qd = torch.empty(M * 4, 32, device=dev)
    vT = torch.empty(2 * B, 256, 200, device=dev)
    _lib.call("sam6d_rpe_front_vt", pem._p(x2), fr["img"].data_ptr(), pem._p(L["qkv"].b), fr["inv"][0], fr["inv"][1], fr["inv"][2], pem._p(qkv),
              pem._p(qp), pem._p(qd), M, pem._p(vT), 197, 200, pem._s())


def timeit(name, fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print("%-40s %8.1f us" % (name, (time.perf_counter() - t0) / n * 1e6), flush=True)


# hypothesis scoring at the benchmark's shape: 32 proposals x 300 hypotheses x 196 points x 1024 CAD points
gs = torch.Generator().manual_seed(1)
nh, k, N1, P = 6000, 300, 196, 1024
Q, _ = torch.linalg.qr(torch.randn(B * nh, 3, 3, generator=gs))
Rs = Q.reshape(B, nh, 9).contiguous().to(dev)
ts = (torch.randn(B, nh, 3, generator=gs) * 0.1).to(dev)
sel = torch.stack([torch.randperm(nh, generator=gs)[:k] for _ in range(B)]).to(torch.int32).to(dev)
p1 = (torch.rand(B, N1, 3, generator=gs) - 0.5).to(dev)
w1 = (torch.rand(B, N1, generator=gs) > 0.2).float().to(dev)
model = (torch.rand(B, P, 3, generator=gs) - 0.5).to(dev)
radius = torch.ones(B, device=dev)
scores = torch.empty(B, k, device=dev); Rb = torch.empty(B, 9, device=dev); tb = torch.empty(B, 3, device=dev)
best = torch.empty(B, dtype=torch.int32, device=dev)
wsb = torch.empty(B * N1 * k, device=dev)
P_ = pem._p


def score_mfma():
    _lib.call("sam6d_score_select_hypotheses_ws", P_(sel), P_(Rs), P_(ts), P_(p1), P_(w1), P_(model), P_(radius), B, N1, P, nh, k,
              P_(scores), P_(Rb), P_(tb), P_(best), P_(wsb), wsb.numel() * 4, pem._s())


def score_valu():
    _lib.call("sam6d_score_select_hypotheses", P_(sel), P_(Rs), P_(ts), P_(p1), P_(w1), P_(model), P_(radius), B, N1, P, nh, k,
              P_(scores), P_(Rb), P_(tb), P_(best), pem._s())


dis_t = torch.rand(B, nh, generator=gs).to(dev)
sel_t = torch.empty(B, k, dtype=torch.int32, device=dev)
idx_t = torch.randint(0, 196 * 196, (B, 3 * nh), generator=gs).to(torch.int32).to(dev)
p2 = (torch.rand(B, N1, 3, generator=gs) - 0.5).to(dev)
Rs_o = torch.empty(B, nh, 9, device=dev); ts_o = torch.empty(B, nh, 3, device=dev); dis_o = torch.empty(B, nh, device=dev)

ops = {
    "select_smallest": lambda: _lib.call("sam6d_select_smallest", P_(dis_t), B, nh, k, P_(sel_t), pem._s()),
    "coarse_hyp": lambda: _lib.call("sam6d_coarse_hypotheses", P_(idx_t), P_(p1), P_(p2), B, N1, N1, nh, P_(Rs_o), P_(ts_o), P_(dis_o), pem._s()),
    "rpe_front": lambda: pem.rpe_self_layer(S, G, T["self"]) if False else _front(),
    "score_mfma": score_mfma,
    "score_valu": score_valu,
    "token_block_12608": lambda: pem._post_attention(hid, x2, T["self"]),
    "token_block_6304": lambda: pem._post_attention(hid[:6304], x2[:6304], T["cross"]),
    "self_layer": lambda: pem.rpe_self_layer(S, G, T["self"]),
    "xattn_only": lambda: _xattn(),
    "sattn_only": lambda: _sattn(),
    "cross_layer": lambda: pem.cross_layer(S[:B], S[B:], T["cross"]),
    "geo_transformer_block": lambda: pem.geometric_transformer(S, G, T),
    "dense_layer": lambda: pem.linear_transformer_layer(DD, S, FT),
    "out_split": lambda: pem.linear_norm_split(DD.reshape(-1, 256), W.fine["out_proj"], W.fine["out_img"]),
    "kv_linear_6304": lambda: pem.linear(x2[:6304], T["cross"]["kv"]),
}
DD = torch.randn(2 * B, 2049, 256, generator=g).to(dev) if ("dense_layer" in sys.argv or "out_split" in sys.argv) else None
FT = W.fine["blocks"][0]["dense"] if "dense" in W.fine["blocks"][0] else None
want = sys.argv[1:] or [k for k in ops if k not in ("dense_layer", "out_split")]
for name in want:
    timeit(name, ops[name])
