"""Which torch (aten) ops run inside one pem_match call: the step should be library launches only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from sam6d_hip import pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
d = {k: v.to(dev).contiguous() for k, v in synth.config2_inputs(B=32, seed=1).items()}
step = lambda: pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"])
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::")]
agg = {}
for e in rows:
    st = [s for s in e.stack if "sam6d_hip" in s or "pem" in s][:1]
    k = (e.key, st[0] if st else "?")
    agg[k] = agg.get(k, 0) + e.count
for (k, st), c in sorted(agg.items(), key=lambda x: -x[1])[:40]:
    print("%4d  %-28s %s" % (c, k, st[-90:]))
