#!/usr/bin/env python3
"""token_block_kernel<1> (dense linear-attention layer): s_memtime stamps of the diagnostic build (scratch/stamp_build.sh).  Per panel
step: cycles at s_waitcnt vmcnt, at the barrier, and from the barrier to the next step's wait (MMA + whatever epilogue follows)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import numpy as np, torch
from sam6d_hip import _lib
_lib.LIB_PATH = os.path.join(ROOT, "scratch/stamp/libsam6d_hip.so")
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
g = torch.Generator().manual_seed(0)
B = 32
D = torch.randn(2 * B, 2049, 256, generator=g).to(dev); S = torch.randn(2 * B, 197, 256, generator=g).to(dev)
L = W.fine["blocks"][0]["dense"]
for _ in range(2):
    pem.linear_transformer_layer(D, S, L)
torch.cuda.synchronize()
lib = _lib.load(); lib.sam6d_tb_debug_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros(512 * 8 * 256, dtype=np.uint64)
assert lib.sam6d_tb_debug_stamps(buf.ctypes.data) == 0
st = buf.reshape(512, 8, 256)[:512, :4].astype(np.int64)
npan = 72
tot = st[:, :, 3] - st[:, :, 0]
wb = st[:, :, 4:4 + 3 * npan:3]; wa = st[:, :, 5:5 + 3 * npan:3]; ba = st[:, :, 6:6 + 3 * npan:3]
dma_wait = wa - wb; bar_wait = ba - wa
comp = np.concatenate([wb[:, :, 1:] - ba[:, :, :-1], (st[:, :, 2] - ba[:, :, -1])[:, :, None]], axis=2)
md = lambda a: np.median(a)
print("512 workgroups sampled (two per CU resident), cycles per wave (medians)")
print("total %.0f | prologue (start -> D rows split) %.0f | store tail %.0f" % (md(tot), md(st[:, :, 1] - st[:, :, 0]), md(st[:, :, 3] - st[:, :, 2])))
print("sums over 72 panels: vmcnt wait %.0f  barrier wait %.0f  barrier->next wait %.0f" % (md(dma_wait.sum(2)), md(bar_wait.sum(2)), md(comp.sum(2))))
names = ["q proj"] * 8 + ["kv"] * 8 + ["lin"] * 8 + sum([["exp%d" % c] * 4 + ["sq%d" % c] * 8 for c in range(4)], [])
cm = np.median(comp, axis=(0, 1)); dw = np.median(dma_wait, axis=(0, 1)); bw = np.median(bar_wait, axis=(0, 1))
for grp in ("q proj", "kv", "lin", "exp0", "sq0", "exp1", "sq1", "exp3", "sq3"):
    ix = [i for i, n in enumerate(names) if n == grp]
    print("%-7s compute per panel %s | vmcnt %s | barrier %s" % (grp, np.round(cm[ix]).astype(int).tolist(), np.round(dw[ix]).astype(int).tolist(), np.round(bw[ix]).astype(int).tolist()))
