import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch, numpy as np
from sam6d_hip import _lib
_lib.LIB_PATH = os.path.join(ROOT, "scratch/stamp/libsam6d_hip.so")
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
g = torch.Generator().manual_seed(0)
pts = (torch.rand(32, 2048, 3, generator=g) - 0.5).to(dev)
grp = pem.pe_group(pts)
lib = _lib.load(); lib.sam6d_pe_debug_stamps.argtypes = [ctypes.c_void_p]
names = ["prefetch issue", "layer 1 MFMA issue", "layer-1 epilogue", "layer 2", "layer 3", "max epilogue", "rotation (coord wait)"]
for k in (0, 1):
    feat = torch.empty(32 * 2048, 256, device=dev)
    L = W.pe["mlp"][k]; idx = grp[k]
    for _ in range(2):
        _lib.call("sam6d_pe_mlp_max_wg", pem._p(pts), pem._p(idx), 32, 2048, idx.shape[2], pem._p(L[0]["w"]), pem._p(L[0]["scale"]), pem._p(L[0]["shift"]),
                  pem._p(L[1]["w"]), pem._p(L[1]["scale"]), pem._p(L[1]["shift"]), pem._p(L[2]["w"]), pem._p(L[2]["scale"]), pem._p(L[2]["shift"]),
                  pem._p(feat), 256, k * 128, 0, pem._s())
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.sam6d_pe_debug_stamps(buf.ctypes.data)
    st = buf.reshape(4096, 8)[:3072].astype(np.float64)
    tiles = st[:, 7]
    print("nsample %d: tiles per wave median %d; cycles per tile (median over 3072 waves)" % (idx.shape[2], np.median(tiles)))
    tot = 0
    for i, nm in enumerate(names):
        v = np.median(st[:, i] / tiles); tot += v
        print("   %-26s %7.0f" % (nm, v))
    print("   %-26s %7.0f   (MFMA issue alone: 1984)" % ("sum", tot))
