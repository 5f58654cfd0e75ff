import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liborder.so"))
lib.order_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
dev = torch.device("cuda:0")
n = 6304 * 64
buf = torch.zeros(n, dtype=torch.int32, device=dev); zeros = torch.zeros(1, dtype=torch.int32, device=dev)
A = torch.randn(131136, 256, device=dev); Wt = torch.randn(256, 256, device=dev); Cc = torch.empty(131136, 256, device=dev)
from sam6d_hip import pem
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for label, interf in (("alone", None), ("beside my dense GEMM", lambda: pem.gemm(A, Wt, None, Cc, 131136, 256, 256, 256, 256, 256)), ("beside torch matmul", lambda: torch.matmul(A, Wt))):
    bad = 0; mx = 0
    for rep in range(30):
        s0.wait_stream(torch.cuda.current_stream()); s1.wait_stream(torch.cuda.current_stream())
        if interf is not None:
            with torch.cuda.stream(s1):
                for _ in range(4): interf()
        with torch.cuda.stream(s0):
            lib.order_probe(buf.data_ptr(), n, zeros.data_ptr(), 20000, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        z = int(zeros.item()); bad += z > 0; mx = max(mx, z)
    print("%-24s reader saw unwritten entries in %d/30 runs (max %d of %d)" % (label, bad, mx, n), flush=True)
