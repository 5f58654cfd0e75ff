"""RCCL sanity of exactly the collectives bench.py issues for N > 1, with one rank (a 1-GPU box cannot host two)."""
import os, socket, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'openvino-sam-6d_amd')]
import torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from sam6d_hip.parallel import pack_poses, unpack_poses
R = torch.randn(32, 3, 3, device=dev); t = torch.randn(32, 3, device=dev); sc = torch.rand(32, device=dev)
mine = pack_poses(R, t, sc)
out = torch.empty_like(mine)
dist.all_gather_into_tensor(out, mine)
dist.barrier()
tt = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
R2, t2, s2 = unpack_poses(out)
assert torch.equal(R2, R) and torch.equal(t2, t) and torch.equal(s2, sc) and float(tt.item()) == 1.25
dist.destroy_process_group()
print("rccl world-1: all_gather_into_tensor / barrier / all_reduce(MAX, f64) ok")
