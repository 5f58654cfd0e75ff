"""Does running the 32 proposals as k micro-batches on k streams hide the latency-bound small-kernel chains?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
inp = synth.config2_inputs(B=32, seed=1); d = {k: v.to(dev).contiguous() for k, v in inp.items()}
keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
def whole():
    return pem.pem_match(*[d[k] for k in keys], W, d["rand"])
def make_split(k, overlap):
    streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
    chunks = [{kk: v[i::1][i * (32 // k):(i + 1) * (32 // k)].contiguous() if False else v[i * (32 // k):(i + 1) * (32 // k)].contiguous() for kk, v in d.items()} for i in range(k)]
    cfg = dict(pem.DEFAULT_CFG, overlap=overlap)
    def run():
        main = torch.cuda.current_stream()
        outs = []
        for s, c in zip(streams, chunks):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(pem.pem_match(*[c[kk] for kk in keys], W, c["rand"], cfg=cfg))
        for s in streams:
            main.wait_stream(s)
        return outs
    return run
def bench(f, name, n=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-34s %.2f ms/step  %.0f proposals/s" % (name, dt * 1e3, 32 / dt), flush=True)
bench(whole, "one batch of 32")
for k in (2, 4):
    for ov in (True, False):
        bench(make_split(k, ov), "%d micro-batches, side-stream %s" % (k, ov))
R = whole(); S = make_split(2, True)()
print("max dR", float((torch.cat([o[0] for o in S]) - R[0]).abs().max()))
