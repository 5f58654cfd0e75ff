"""Log every C-ABI call of one pem_match step (B=32) with its shape arguments; aggregate the GEMMs."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
inp = synth.config2_inputs(B=32, seed=1); d = {k: v.to(dev).contiguous() for k, v in inp.items()}
run = lambda: pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"], cfg=dict(pem.DEFAULT_CFG, overlap=False))
run(); torch.cuda.synchronize()
log = []
orig = _lib.call
def call(name, *a):
    log.append((name, a)); return orig(name, *a)
_lib.call = call; pem._lib.call = call
run(); torch.cuda.synchronize()
print("calls per step:", len(log))
cnt = collections.Counter(n for n, _ in log)
print(cnt.most_common())
g = collections.Counter()
for n, a in log:
    if n == "sam6d_gemm_nt":
        M, N, K = a[6], a[7], a[8]; batch = a[13]
        g[(M, N, K, batch, "res" if a[4] else "", "bias" if a[2] else "", "relu" if a[19] else "")] += 1
    if n == "sam6d_gemm_nt_b2":
        g[("b2", a[3], a[4], a[5], a[9], a[13])] += 1
for k, v in sorted(g.items(), key=lambda kv: -kv[1]):
    print(v, k)
