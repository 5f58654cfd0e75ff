"""Probe: pem_match captured into a hipGraph (PemGraph), eager vs replay, micro-batch slices 1 / 2 / 4; B = 32, 25, 1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "openvino-sam-6d_amd")):
    sys.path.insert(0, p)
import torch
from sam6d_hip import pem, synth

dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1)
W = pem.PemWeights(sd, dev)
KEYS = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model", "rand")


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for B in [int(x) for x in (sys.argv[1:] or ["32", "25", "1"])]:
    inp = synth.config2_inputs(B=B, seed=1)
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    args = [d[k] for k in KEYS[:6]]
    eager = [o.clone() for o in pem.pem_match(*args, W, d["rand"])]
    ms_e = timeit(lambda: pem.pem_match(*args, W, d["rand"]))
    print("B=%d eager %.3f ms" % (B, ms_e), flush=True)
    for mb in (1, 2, 4):
        if mb > 1 and B < 8 * mb:
            continue
        try:
            g = pem.PemGraph(W, *args, d["rand"], microbatch=mb)
            out = g.replay()
            torch.cuda.synchronize()
            same = all(torch.equal(a, b) for a, b in zip(eager, out))
            ms = timeit(g.replay)
            ms_c = timeit(lambda: g(*args, d["rand"]))
            print("B=%d graph mb=%d: %.3f ms/replay (%.3f with input copies), bit-identical to eager: %s" % (B, mb, ms, ms_c, same), flush=True)
        except Exception as e:
            print("B=%d graph mb=%d FAILED: %s: %s" % (B, mb, type(e).__name__, e), flush=True)
    if B >= 16:
        os.environ["SAM6D_MICROBATCH"] = "2"
        ms2 = timeit(lambda: pem.pem_match(*args, W, d["rand"]))
        os.environ.pop("SAM6D_MICROBATCH")
        print("B=%d eager mb=2 %.3f ms" % (B, ms2), flush=True)
