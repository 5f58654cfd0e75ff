"""N>1 path on CPU: world_size-2 gloo processes shard proposals round-robin, all-gather the 13-float pose blocks and
reassemble them in global order.  The per-rank 'compute' here is a deterministic stand-in; the GPU kernels
are covered by the -m gpu tests."""
import os
import sys

import torch
import torch.multiprocessing as mp

from tests._util import ROOT, PKG  # noqa: F401


def _fake_pose(ids):
    g = ids.float().unsqueeze(1)
    R = (torch.arange(9).float().unsqueeze(0) + 100 * g).reshape(-1, 3, 3)
    t = torch.arange(3).float().unsqueeze(0) - g
    s = g.squeeze(1) * 0.5
    return R, t, s


def _worker(rank, world, port, n_total, ret):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from sam6d_hip import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, n_valid = parallel.shard_indices(n_total, rank, world)
    R, t, s = _fake_pose(ids)
    gR, gt, gs = parallel.gather_poses(R, t, s, dist)
    allR = parallel.unshard(gR.reshape(-1, 9), n_total, world).reshape(-1, 3, 3)
    allt = parallel.unshard(gt, n_total, world)
    alls = parallel.unshard(gs.reshape(-1, 1), n_total, world).reshape(-1)
    wR, wt, ws = _fake_pose(torch.arange(n_total))
    ok = torch.equal(allR, wR) and torch.equal(allt, wt) and torch.equal(alls, ws)
    ret[rank] = bool(ok) and n_valid == len(range(rank, n_total, world))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n_total, port):
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_total, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_gather_poses_world2_even():
    _run(2, 8, 29611)


def test_gather_poses_world2_ragged():
    _run(2, 7, 29612)  # 7 proposals over 2 ranks: padding on rank 1


def test_shard_indices_cover():
    from sam6d_hip import parallel
    for n, w in ((200, 8), (7, 2), (1, 4), (32, 1)):
        seen = []
        for r in range(w):
            ids, nv = parallel.shard_indices(n, r, w)
            assert len(ids) == (n + w - 1) // w
            seen += ids[:nv].tolist()
        assert sorted(seen) == list(range(n))
