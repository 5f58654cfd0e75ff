"""Diagnostic: per-wave s_memtime stamps of rpe_score_kernel (scratch/libsam6d_stamp.so, built with -DRP_STAMP)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch, numpy as np
from sam6d_hip import _lib
_lib.LIB_PATH = os.path.join(ROOT, "scratch", "libsam6d_stamp.so")
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
g = torch.Generator().manual_seed(5)
Bp, n = 64, 197
pts = (torch.rand(Bp, n, 3, generator=g) - 0.5) + torch.tensor([0.3, -0.2, 8.0]); pts[:, 0] = 100.0
pts = pts.to(dev)
x = torch.randn(Bp, n, 256, generator=g).to(dev)
L = W.coarse["blocks"][0]["self"]
G = pem.geo_context(pts, W)
for _ in range(30):
    out = pem.rpe_self_layer(x, G, L)
torch.cuda.synchronize()
buf = np.zeros(4096 * 12 * 12, dtype=np.uint64)
lib = _lib.load()
lib.sam6d_rpe_debug_stamps.argtypes = [ctypes.c_void_p]
rc = lib.sam6d_rpe_debug_stamps(buf.ctypes.data)
assert rc == 0
s = buf.reshape(4096, 12, 12)[:256].astype(np.int64)   # [wg][wave][slot]
t0 = s[..., 0]; base = t0.min()
nq = s[..., 11]
print("queries per wave: ", np.bincount(nq.ravel()))
print("wave start spread (cycles after first): p50 %d  max %d" % (np.median(t0 - base), (t0 - base).max()))
setup = s[..., 1] - s[..., 0]
print("image copy + barrier: p50 %d max %d" % (np.median(setup), setup.max()))
end = np.zeros_like(t0)
for w in range(256):
    for v in range(12):
        k = min(nq[w, v], 6)
        end[w, v] = s[w, v, 1 + k] if k > 0 else s[w, v, 1]
hw = s[..., 10]
xcc = (hw >> 32) & 15; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
real = (s[..., 9] - s[..., 8]).astype(np.float64)
clk = (end - t0) / real * 100e6
print("in-kernel clock (GHz): p50 %.3f  min %.3f max %.3f" % (np.median(clk) / 1e9, clk.min() / 1e9, clk.max() / 1e9))
for k in range(5):
    a = s[..., 1 + k]; b = s[..., 2 + k]
    m = nq > k
    d = (b - a)[m]
    print("query #%d: n %4d  p50 %7d  min %7d  max %7d cycles (%.0f per tile)" % (k, m.sum(), np.median(d), d.min(), d.max(), np.median(d) / 13))
print("SIMD of waves 0..11 (WG 0):", simd[0].tolist())
for X in range(8):
    m = xcc[:, 0] == X
    wgs = np.nonzero(m)[0]
    if len(wgs) == 0: continue
    b0 = t0[wgs].min()
    places = sorted(set((int(se[w, 0]), int(sh[w, 0]), int(cu[w, 0])) for w in wgs))
    starts = sorted(int(t0[w].min() - b0) for w in wgs)
    ends = sorted(int(end[w].max() - b0) for w in wgs)
    print("XCC %d: %d WGs on %d distinct CUs; WG starts p50 %d max %d; WG ends min %d p50 %d max %d" % (X, len(wgs), len(places), starts[len(starts)//2], starts[-1], ends[0], ends[len(ends)//2], ends[-1]))
for w in (0, 100, 200):
    print("WG %d" % w)
    for v in range(12):
        d = [int(s[w, v, 1] - s[w, v, 0])] + [int(s[w, v, 2 + k] - s[w, v, 1 + k]) for k in range(int(nq[w, v]))]
        print("  wave %2d simd %d start+%6d: %s  total %d" % (v, simd[w, v], int(s[w, v, 0] - s[w, :, 0].min()), d, sum(d)))
pb = np.zeros(4096 * 8, dtype=np.uint64)
lib.sam6d_rpe_debug_phases.argtypes = [ctypes.c_void_p]
assert lib.sam6d_rpe_debug_phases(pb.ctypes.data) == 0
ph = pb.reshape(4096, 8)[:256 * 12].reshape(256, 12, 8).astype(np.float64)
names = ["staging", "d+split", "main loop", "epilogue", "softmax", "recurrence", "swaps"]
tiles = nq * 13.0
for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8)), ("waves 8-11", slice(8, 12))):
    print(grp, "cycles per tile (staging/softmax per query):")
    for i, nm in enumerate(names):
        per = ph[:, sl, i] / (nq[:, sl] if i in (0, 4) else tiles[:, sl])
        print("   %-10s p50 %7.0f" % (nm, np.median(per)))
