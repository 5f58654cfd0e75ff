#!/usr/bin/env python3
"""token_block at M = 131072, five launches (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import _lib, pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
L = W.fine["blocks"][0]["dense"]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
h = torch.randn(M, 256, device=dev); x = torch.randn(M, 256, device=dev); o = torch.empty_like(h)
tb = L["tb"]
for _ in range(5):
    _lib.call("sam6d_token_block", h.data_ptr(), x.data_ptr(), tb["img"].data_ptr(), tb["cst"].data_ptr(), o.data_ptr(), M, 1e-5, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
