"""hipGraph capture of the whole matching step (with k micro-batches on k streams inside the graph)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
inp = synth.config2_inputs(B=32, seed=1); d = {k: v.to(dev).contiguous() for k, v in inp.items()}
keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
def eager():
    return pem.pem_match(*[d[k] for k in keys], W, d["rand"])
def split_fn(k):
    streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
    n = 32 // k
    chunks = [{kk: v[i * n:(i + 1) * n].contiguous() for kk, v in d.items()} for i in range(k)]
    def run():
        main = torch.cuda.current_stream()
        outs = []
        for s, c in zip(streams, chunks):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(pem.pem_match(*[c[kk] for kk in keys], W, c["rand"]))
        for s in streams:
            main.wait_stream(s)
        return [torch.cat([o[j] for o in outs]) for j in range(3)]
    return run
def bench(f, name, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-40s %.2f ms/step  %.0f proposals/s" % (name, dt * 1e3, 32 / dt), flush=True)
ref = eager()
bench(eager, "eager, one batch")
for k in (1, 2, 4):
    fn = eager if k == 1 else split_fn(k)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = fn()
    torch.cuda.synchronize()
    bench(g.replay, "graph, %d micro-batch(es)" % k)
    print("   max dR vs eager %.2e  dt %.2e  dscore %.2e" % tuple(float((a - b).abs().max()) for a, b in zip(out, ref)), flush=True)
