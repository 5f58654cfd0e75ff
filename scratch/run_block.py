#!/usr/bin/env python3
"""Times the fused block kernels alone (HIP events, 20 launches each): token_block at the sparse size (M = 64 x 197) and at a dense
size (M = 131072), linattn_layer at (64, 2049).  usage: python scratch/run_block.py"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib, pem, synth
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1)
W = pem.PemWeights(sd, dev)
L = W.fine["blocks"][0]["dense"]
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M in (12608, 6304, 131072):
    h = torch.randn(M, 256, device=dev); x = torch.randn(M, 256, device=dev); o = torch.empty_like(h)
    tb = L["tb"]
    us = timeit(lambda: _lib.call("sam6d_token_block", h.data_ptr(), x.data_ptr(), tb["img"].data_ptr(), tb["cst"].data_ptr(), o.data_ptr(), M, 1e-5, st))
    fl = M * 2.0 * (256 * 256 + 2 * 256 * 512)
    print("token_block M=%6d: %8.1f us  %6.1f TFLOP/s (fp32-equivalent)  %d tiles" % (M, us, fl / us / 1e6, (M + 127) // 128))
Bp, I = 64, 2049
D = torch.randn(Bp, I, 256, device=dev); S = torch.randn(Bp, 197, 256, device=dev)
us = timeit(lambda: pem.linear_transformer_layer(D, S, L))
fl = Bp * 2048 * 2.0 * (256 * 256 * 2 + 256 * 64 + 2 * 256 * 512)
print("linear_transformer_layer (64,2049): %8.1f us  %6.1f TFLOP/s fp32-equivalent (whole layer incl. kv side)" % (us, fl / us / 1e6))
