"""Body of __graft_entry__.smoke(): small hot-path invocation on cuda:0 vs the CPU oracle."""
import torch


def run():
    import pointnet2._ext as ext
    from oracle import pointops as P
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(2, 2048, 3, generator=g) - 0.5
    idx = ext.furthest_point_sampling(xyz.to(dev), 196)
    assert torch.equal(idx.cpu(), P.furthest_point_sampling(xyz, 196)), "FPS mismatch vs oracle"
    q = (xyz + 0.00000001).contiguous()
    bq = ext.ball_query(q.to(dev), xyz.to(dev), 0.1, 32)
    assert torch.equal(bq.cpu(), P.ball_query(q, xyz, 0.1, 32)), "ball_query mismatch vs oracle"
    try:
        from tests import smoke_pem
    except ImportError:
        return
    smoke_pem.run()
