#!/usr/bin/env python3
"""Where a token_block_kernel<0> workgroup spends its time: s_memtime stamps of the diagnostic build (scratch/stamp_build.sh),
run with SAM6D_LIB=scratch/stamp/libsam6d_hip.so.  Per panel: cycles parked at s_waitcnt vmcnt (DMA), at the barrier, and computing."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import numpy as np, torch
from sam6d_hip import pem, synth, _lib
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12608
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 256, generator=g).to(dev); hid = torch.randn(M, 256, generator=g).to(dev)
L = W.coarse["blocks"][0]["self"]
for _ in range(3):
    out = pem._post_attention(hid, x, L)
torch.cuda.synchronize()
lib = _lib.load()
lib.sam6d_tb_debug_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros(512 * 8 * 256, dtype=np.uint64)
rc = lib.sam6d_tb_debug_stamps(buf.ctypes.data)
assert rc == 0
nwg = (M + 63) // 64
st = buf.reshape(512, 8, 256)[:min(nwg, 512), :4].astype(np.int64)
tot = st[:, :, 3] - st[:, :, 0]
npan = 60
wb = st[:, :, 4:4 + 3 * npan:3]; wa = st[:, :, 5:5 + 3 * npan:3]; ba = st[:, :, 6:6 + 3 * npan:3]
dma_wait = (wa - wb); bar_wait = (ba - wa)
comp = np.concatenate([wb[:, :, 1:] - ba[:, :, :-1], (st[:, :, 2] - ba[:, :, -1])[:, :, None]], axis=2)
print("M = %d, %d workgroups; s_memtime ticks (100 MHz constant clock? see ratio) per wave, averages over waves" % (M, nwg))
print("total %.0f | prologue (start -> X split) %.0f | first wait begins at %.0f" % (tot.mean(), (st[:, :, 1] - st[:, :, 0]).mean(), (wb[:, :, 0] - st[:, :, 0]).mean()))
print("sum over 60 panels: vmcnt wait %.0f  barrier wait %.0f  compute %.0f  | store tail %.0f" % (dma_wait.sum(2).mean(), bar_wait.sum(2).mean(), comp.sum(2).mean(), (st[:, :, 3] - st[:, :, 2]).mean()))
print("per panel (mean over waves) vmcnt:", np.round(dma_wait.mean((0, 1))).astype(int).tolist())
print("per panel barrier:", np.round(bar_wait.mean((0, 1))).astype(int).tolist())
print("per panel compute:", np.round(comp.mean((0, 1))).astype(int).tolist())
