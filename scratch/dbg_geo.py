import sys, math
sys.path.insert(0, '.'); 
from tests._util import golden
import numpy as np, torch
from sam6d_hip import pem, synth, _lib
dev = torch.device('cuda:0')
sd = synth.make_pem_weights(1); W = pem.PemWeights(sd, dev)
g = golden('geo_embedding')
pts = torch.from_numpy(g['pts']).to(dev)
B, n, _ = pts.shape
knn = torch.empty(B, n, 3, dtype=torch.int32, device=dev)
idx = torch.empty(B, n, n, 4, device=dev)
out = torch.empty(B, n, n, 256, device=dev)
_lib.call("sam6d_geo_embedding", pts.data_ptr(), B, n, W.div_term.data_ptr(), W.geo_d.w.data_ptr(), W.geo_d.b.data_ptr(),
          W.geo_a.w.data_ptr(), W.geo_a.b.data_ptr(), 0.2, 180.0 / (15 * math.pi), 3, 256, knn.data_ptr(), idx.data_ptr(), out.data_ptr(), pem._s())
torch.cuda.synchronize()
k = knn.cpu().numpy(); gk = g['knn'].astype(np.int32)
bad = np.argwhere((k != gk).any(-1))
print("knn mismatching rows:", bad[:20].tolist(), "count", len(bad))
for b, i in bad[:5]:
    print(b, i, k[b, i], gk[b, i])
d = idx[..., 0].cpu().numpy()
print("d_idx equal:", np.array_equal(d, g['d_idx']), np.abs(d - g['d_idx']).max())
a = idx[..., 1:].cpu().numpy()
da = np.abs(a - g['a_idx'])
print("a_idx max diff", da.max(), "at", np.unravel_index(da.argmax(), da.shape))
rows = g['rows']
o = out[:, rows].cpu().numpy()
df = np.abs(o - g['out_rows'])
print("out diff per (b,row):", df.max(axis=(2, 3)))
print("out diff by column m (max over rest) top:", np.sort(df.max(axis=(0, 1, 3)))[-5:], np.argsort(df.max(axis=(0,1,3)))[-5:])
np.savez_compressed('gpurun_out/dbg_geo.npz', d=d, a=a, knn=k)
