#!/usr/bin/env python3
"""Times sam6d_ball_query2_grid at the step's shape (32 clouds x 2048 queries over 2048 points, radii / nsample of the fine PE) and
compares it with the all-pairs scan sam6d_ball_query2.  usage: python scratch/ub_bq.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib, synth
dev = torch.device("cuda:0")
d = synth.config2_inputs(B=32, seed=2)
pts = d["dense_pm"].to(dev).contiguous()
B, N, _ = pts.shape
r1, ns1, r2, ns2 = 0.1, 32, 0.2, 64
st = torch.cuda.current_stream().cuda_stream
g1 = torch.empty(B, N, ns1, dtype=torch.int32, device=dev); g2 = torch.empty(B, N, ns2, dtype=torch.int32, device=dev)
a1 = torch.empty_like(g1); a2 = torch.empty_like(g2)
nbytes = int(_lib.load().sam6d_ball_query2_grid_workspace_bytes(B, N))
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
def grid(): _lib.call("sam6d_ball_query2_grid", pts.data_ptr(), pts.data_ptr(), B, N, N, r1, ns1, g1.data_ptr(), r2, ns2, g2.data_ptr(), ws.data_ptr(), nbytes, st)
grid()
_lib.call("sam6d_ball_query2", pts.data_ptr(), pts.data_ptr(), B, N, N, r1, ns1, a1.data_ptr(), r2, ns2, a2.data_ptr(), st)
torch.cuda.synchronize()
print("equal to the all-pairs scan:", bool(torch.equal(g1, a1)), bool(torch.equal(g2, a2)), " mean hits", float((a2 != a2[..., :1]).float().sum(-1).mean()) + 1)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): grid()
torch.cuda.synchronize(); a.record()
for _ in range(20): grid()
e.record(); torch.cuda.synchronize()
print("ball_query2_grid (32, 2048 x 2048): %.1f us per call (build + query)" % (a.elapsed_time(e) / 20 * 1e3))
