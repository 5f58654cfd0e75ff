import sys; sys.path.insert(0, '.')
from tests._util import golden
import numpy as np, torch
from sam6d_hip import pem, _lib
from oracle import pem_oracle as O
dev = torch.device('cuda:0')
g = golden('coarse_rt')
t = lambda k: torch.from_numpy(g[k]).to(dev)
p1, p2, model, u = t('p1'), t('p2'), t('model'), t('u')
radius = torch.ones(2, device=dev)
R, tt, aux = pem.compute_coarse_Rt(t('att2'), p1, p2, model, radius, u, return_aux=True)
torch.cuda.synchronize()
np.savez_compressed('gpurun_out/dbg_coarse.npz', **{k: v.cpu().numpy() for k, v in aux.items()}, R=R.cpu().numpy(), t=tt.cpu().numpy())
# weighted sample test data
gen = torch.Generator().manual_seed(4)
w = torch.rand(3, 38416, generator=gen) ** 6
w[1] *= (torch.rand(38416, generator=gen) > 0.97)
w[2] = 0
uu = torch.rand(3, 18000, generator=gen)
cum = torch.empty(3, 38416, device=dev); idx = torch.empty(3, 18000, dtype=torch.int32, device=dev)
_lib.call("sam6d_weighted_sample", w.to(dev).data_ptr(), uu.to(dev).data_ptr(), 3, 38416, 18000, cum.data_ptr(), idx.data_ptr(), pem._s())
torch.cuda.synchronize()
want = O.weighted_sampling(w, uu)
c = torch.cumsum(w, 1); cn = c / (c[:, -1:] + 1e-8)
print("cum equal:", torch.equal(cum.cpu(), cn), (cum.cpu() != cn).sum(1).tolist(), (cum.cpu() - cn).abs().max().item())
print("idx mismatch per row:", (idx.cpu().long() != want).sum(1).tolist())
bad = (cum.cpu() != cn).nonzero()[:5]
for b, i in bad.tolist():
    print(b, i, cum[b, i].item(), cn[b, i].item(), c[b, i].item(), c[b, -1].item())
