"""Host-side cost of one pem_match call with the C ABI mocked out (runs without a GPU)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
ncalls = [0]
def fake(name, *a):
    ncalls[0] += 1
_lib.call = fake; pem._lib.call = fake
pem._s = lambda: 0
sd = synth.make_pem_weights(1)
W = pem.PemWeights(sd, torch.device("cpu"))
inp = synth.config2_inputs(B=4, seed=1)
cfg = dict(pem.DEFAULT_CFG, overlap=False, microbatch=1)
run = lambda: pem.pem_match(inp["dense_pm"], inp["dense_fm"], inp["dense_po"], inp["dense_fo"], inp["radius"], inp["model"], W, inp["rand"], cfg=cfg)
run(); ncalls[0] = 0
t0 = time.perf_counter()
for _ in range(20): run()
dt = (time.perf_counter() - t0) / 20
print("host time per pem_match call: %.2f ms, %d C calls" % (dt * 1e3, ncalls[0] // 20))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
