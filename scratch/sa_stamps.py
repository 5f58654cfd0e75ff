import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch, numpy as np
from sam6d_hip import _lib
_lib.LIB_PATH = os.path.join(ROOT, "scratch/stamp/libsam6d_hip.so")
from sam6d_hip import pem
dev = torch.device("cuda:0")
B, n = 64, 197
qkv = torch.randn(B * n, 768, device=dev); G = torch.randn(B * n, 4, 200, device=dev); out = torch.empty(B * n, 256, device=dev)
for _ in range(3):
    _lib.call("sam6d_rpe_self_attention", pem._p(qkv), pem._p(G), pem._p(out), B, n, 200, pem._s())
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
lib = _lib.load()
lib.sam6d_sattn_debug_stamps.argtypes = [ctypes.c_void_p]
lib.sam6d_sattn_debug_stamps(buf.ctypes.data)
st = buf.reshape(256, 8, 16).astype(np.int64)
names = ["start", "loads+max", "barrier1", "images", "barrier2", "g0 ready", "g0 S", "g0 softmax+split", "g0 PV", "g0 store", "g1 ready", "g1 S", "g1 softmax+split", "g1 PV", "g1 store", "end"]
print("per-wave cycles since the wave's own start (s_memtime is per XCD); median / min / max over the waves that passed the stamp")
rel = st - st[:, :, :1]
for i, nm in enumerate(names):
    v = rel[:, :, i][st[:, :, i] > 0]
    print("%-18s median %7.0f   min %7.0f   max %7.0f   (waves %d)" % (nm, np.median(v), v.min(), v.max(), v.size))
# spread of wave starts inside a workgroup and of workgroup ends is not comparable across XCDs; per workgroup: last end - first start
wg = (st[:, :, 15].max(axis=1) - st[:, :, 0].min(axis=1))
print("workgroup lifetime (last wave end - first wave start): median %d, min %d, max %d cycles" % (np.median(wg), wg.min(), wg.max()))
