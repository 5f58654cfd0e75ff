"""geo_embedding alone on 3 concurrent streams vs solo references; xmax variants isolate the Chebyshev / sinusoid kernels."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from sam6d_hip import pem, synth, _lib
dev = torch.device("cuda:0")
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
g = torch.Generator().manual_seed(7)
B, n = 16, 197
pts = [((torch.rand(B, n, 3, generator=g) - 0.5) + torch.tensor([0.3, -0.2, 8.0])) for _ in range(3)]
for p in pts: p[:, 0] = 100.0
pts = [p.to(dev) for p in pts]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
# a heavy unrelated kernel stream to perturb timing
noise_a = torch.randn(8192, 8192, device=dev); noise_s = torch.cuda.Stream(device=dev)
for xmax, label in ((24.0, "mixed (default)"), (1e9, "Chebyshev kernel only"), (1e-9, "sinusoid list kernel only")):
    pem.GEO_XMAX = xmax
    W._geo_cheb = None; W._geo_dcT = None
    refs = [pem.geo_embedding(p, W).clone() for p in pts]
    torch.cuda.synchronize()
    bad = 0; info = []
    for rep in range(15):
        outs = []
        with torch.cuda.stream(noise_s):
            tmp = noise_a @ noise_a
        for st, p in zip(streams, pts):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(pem.geo_embedding(p, W))
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(outs, refs)):
            if not torch.equal(o, r):
                bad += 1
                rows = (o != r).any(-1)  # (B,n,n)
                idx = rows.nonzero()
                bgshare = float(((idx[:, 1] == 0) | (idx[:, 2] == 0)).float().mean())
                info.append("rep %d stream %d: %d rows wrong, bg share %.2f, clouds %s" % (rep, i, idx.shape[0], bgshare, sorted(set(idx[:, 0].tolist()))))
    print("%-28s mismatches %d/45" % (label, bad), flush=True)
    for s in info[:4]: print("   ", s)
