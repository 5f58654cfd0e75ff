#!/usr/bin/env python3
"""Times sam6d furthest_point_sampling at the step's shape (64 clouds x 2048 points -> 196 samples) and checks it against a numpy scan
with the reference's rule (largest running min-distance, first index on ties, origin-ball points skipped).  usage: python scratch/ub_fps.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import numpy as np, torch
from sam6d_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
B, N, m = 64, 2048, 196
x = (torch.rand(B, N, 3, generator=g) - 0.5)
x[3, 7:40] = 0.0          # origin-ball points
x[5, 100] = x[5, 900]      # duplicate points (distance ties)
xd = x.to(dev).contiguous()
idx = ops.furthest_point_sampling(xd, m)
torch.cuda.synchronize()
def ref(p, m):
    n = p.shape[0]; td = np.full(n, np.float32(3.4028234663852886e38), np.float32); out = [0]; last = 0
    mag = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1] + p[:, 2] * p[:, 2]).astype(np.float32)
    livem = ~(mag.astype(np.float64) <= 1e-3)
    for _ in range(1, m):
        d = p - p[last]; d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32)
        td = np.where(livem, np.minimum(td, d2), td)
        cand = np.where(livem, td, np.float32(-1))
        last = int(np.argmax(cand)) if cand.max() >= 0 else 0
        out.append(last)
    return np.array(out)
bad = 0
for b in (0, 3, 5, 63):
    want = ref(x[b].numpy(), m); got = idx[b].cpu().numpy()
    bad += int((want != got).sum())
print("mismatches vs numpy scan (4 clouds):", bad, " sha", hashlib.sha256(idx.cpu().numpy().tobytes()).hexdigest()[:16])
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): ops.furthest_point_sampling(xd, m)
torch.cuda.synchronize(); a.record()
for _ in range(20): ops.furthest_point_sampling(xd, m)
e.record(); torch.cuda.synchronize()
print("fps (64, 2048) -> 196: %.1f us per launch" % (a.elapsed_time(e) / 20 * 1e3))
