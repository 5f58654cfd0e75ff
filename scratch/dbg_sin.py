import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd")); sys.path.insert(0, os.path.join(ROOT, "openvino-sam-6d_amd", "pem"))
import numpy as np, torch
import transformer as T
g = np.load(os.path.join(ROOT, "tests/golden/submodules.npz"))
dev = torch.device("cuda:0")
emb = T.SinusoidalPositionalEmbedding(256).to(dev)
out = emb(torch.from_numpy(g["sin_idx"]).to(dev)).cpu().numpy()
d = np.abs(out - g["sin"])
i = np.unravel_index(d.argmax(), d.shape)
print("max diff", d.max(), "at", i, "got", out[i], "want", g["sin"][i], "idx", g["sin_idx"][i[0], i[1]], "div", float(emb.div_term[i[2] // 2]))
dt = emb.div_term.cpu().numpy()
cpu_dt = torch.exp(torch.arange(0, 256, 2).float() * (-np.log(10000.0) / 256)).numpy()
print("div_term device vs cpu equal:", np.array_equal(dt, cpu_dt))
print("rows max diff:", d.reshape(8, 256).max(axis=1))
