"""Do kernels of two HIP streams overlap on this stack?  L = a latency-bound launch (the 197-token layer tail on 6304 rows: 99 workgroups),
T = a throughput-bound one (a 131072 x 256 x 256 GEMM).  Times: each alone, both on one stream, on two streams, and as two branches of
a captured graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "openvino-sam-6d_amd")):
    sys.path.insert(0, p)
import torch
from sam6d_hip import pem, synth, _lib

dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
L = W.coarse["blocks"][0]["cross"]
M = 6304
hid = torch.randn(M, 256, device=dev); x = torch.randn(M, 256, device=dev); out = torch.empty(M, 256, device=dev)
A = torch.randn(131072, 256, device=dev); Wt = torch.randn(256, 256, device=dev) / 16; Cq = torch.empty(131072, 256, device=dev)
lin = pem.Linear(Wt, None)
K = 20

def run_L():
    for _ in range(K):
        pem._post_attention(hid, x, L, out=out)
def run_T():
    for _ in range(K):
        pem.gemm(A, Wt, None, Cq, 131072, 256, 256, 256, 256, 256, w16=lin.w16())

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both_two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): run_L()
    with torch.cuda.stream(s2): run_T()
    cur.wait_stream(s1); cur.wait_stream(s2)
def both_one_stream():
    run_L(); run_T()
def two_L():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): run_L()
    with torch.cuda.stream(s2): run_L()
    cur.wait_stream(s1); cur.wait_stream(s2)

tL, tT = timeit(run_L), timeit(run_T)
print("L alone: %.3f ms (%d launches, %.1f us each)   T alone: %.3f ms (%.1f us each)" % (tL, K, 1e3 * tL / K, tT, 1e3 * tT / K))
print("one stream  L then T : %.3f ms" % timeit(both_one_stream))
print("two streams L || T   : %.3f ms" % timeit(both_two_streams))
print("two streams L || L   : %.3f ms" % timeit(two_L))
for name, fn in (("L || T", both_two_streams), ("L || L", two_L)):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    print("graph, two branches %s: %.3f ms" % (name, timeit(g.replay)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    both_one_stream()
print("graph, one branch L then T: %.3f ms" % timeit(g.replay))
