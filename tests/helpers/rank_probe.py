"""Child program for tests/test_bench_launcher.py: what bench.py's ranks do around the timed region, on CPU with the gloo backend.
Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (set by bench.launch_ranks), all-gathers one fake pose block per
rank with sam6d_hip.parallel.gather_poses and lets rank 0 print one JSON line.  --fail-rank R makes rank R exit with code 3 before the
collective (the launcher must then stop the others and return 3)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "openvino-sam-6d_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from sam6d_hip import parallel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--fail-rank", type=int, default=-1)
ap.add_argument("--out", default="")
a = ap.parse_args()
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
assert world == a.gpus and local == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
if rank == a.fail_rank:
    sys.exit(3)
dist.init_process_group("gloo", rank=rank, world_size=world)
B = 4
ids = torch.arange(B).float() + 100 * rank
R = ids.reshape(B, 1, 1).repeat(1, 3, 3)
t = ids.reshape(B, 1).repeat(1, 3)
s = ids.clone()
dist.barrier()
gR, gt, gs = parallel.gather_poses(R, t, s, dist)
dist.barrier()
ok = gs.tolist() == [float(i + 100 * r) for r in range(world) for i in range(B)]
if rank == 0:
    line = json.dumps({"n_gpus": world, "world_size": dist.get_world_size(), "rows": int(gs.shape[0]), "ok": bool(ok)})
    print(line)
    if a.out:
        open(a.out, "w").write(line)
dist.destroy_process_group()
