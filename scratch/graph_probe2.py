"""Which part of the micro-batch launch sequence breaks hipGraph capture?  Each variant in its own child process (a native abort must not
take the others down), python -X faulthandler for the Python stack at the crash."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth
variant = sys.argv[1]
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
B = 32
inp = synth.config2_inputs(B=B, seed=1)
d = {k: v.to(dev).contiguous() for k, v in inp.items()}
args = [d[k] for k in ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")]
eager = [o.clone() for o in pem.pem_match(*args, W, d["rand"])]
cfg = dict(pem.DEFAULT_CFG)
# variant = mb<k>[_nopipe][_serialprep][_nested]
parts = variant.split("_")
mb = int(parts[0][2:])
cfg["mb_pipeline"] = "nopipe" not in parts
cfg["mb_serial_prepare"] = "serialprep" in parts
cfg["mb_overlap"] = "nested" in parts
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
ecfg = dict(cfg, microbatch=mb)
out = pem.pem_match(*args, W, d["rand"], cfg=ecfg); torch.cuda.synchronize()
print("eager identical:", all(torch.equal(a, b) for a, b in zip(eager, out)), "ms/step %%.3f" %% timeit(lambda: pem.pem_match(*args, W, d["rand"], cfg=ecfg)), flush=True)
print("variant", variant, "capturing", flush=True)
g = pem.PemGraph(W, *args, d["rand"], cfg=cfg, microbatch=mb)
print("captured", flush=True)
out = g.replay(); torch.cuda.synchronize()
print("replayed; identical:", all(torch.equal(a, b) for a, b in zip(eager, out)), flush=True)
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); print("ms/replay %%.3f" %% (1e3 * (time.perf_counter() - t0) / 20), flush=True)
''' % (ROOT, ROOT)
for v in sys.argv[1:] or ["mb1", "mb2_nopipe", "mb2", "mb4_nopipe", "mb4", "mb2_serialprep"]:
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD, v], capture_output=True, text=True, timeout=280)
    print("=== %s rc=%d\n%s\n--- stderr tail:\n%s" % (v, p.returncode, p.stdout[-1500:], p.stderr[-3000:]), flush=True)
