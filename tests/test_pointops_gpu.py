"""GPU parity of the `pointnet2._ext` seam: HIP kernels vs golden vectors captured from the reference's own
compiled CPU loops (tests/golden/pointops.npz) and vs the C oracle on fresh seeded inputs.  Bit-exact."""
import hashlib

import numpy as np
import pytest
import torch

from tests._util import golden

pytestmark = pytest.mark.gpu


def _sha(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ext():
    import pointnet2._ext as e
    return e


def test_fps_golden(dev, ext):
    g = golden("pointops")
    xyz = torch.from_numpy(g["fps_xyz"]).to(dev)
    idx = ext.furthest_point_sampling(xyz, 196)
    assert idx.dtype == torch.int32 and idx.shape == (2, 196)
    assert np.array_equal(idx.cpu().numpy(), g["fps_idx"])


def test_fps_reference_test_shape(dev, ext):
    g = golden("pointops")
    rs = np.random.RandomState(324)  # ov_test_furthest_point_sampling_1input.py seeds/shapes
    big = torch.from_numpy(rs.randn(1, 21000, 3).astype(np.float32)).to(dev)
    idx = ext.furthest_point_sampling(big, 2048)
    assert np.array_equal(idx.cpu().numpy(), g["fps_big_idx"])


@pytest.mark.parametrize("B,N,m", [(3, 2048, 196), (2, 1000, 64), (1, 4096, 300), (2, 5000, 128), (1, 64, 64), (2, 7, 3)])
def test_fps_vs_oracle(dev, ext, B, N, m):
    from oracle import pointops as P
    g = torch.Generator().manual_seed(B * 1000 + N)
    xyz = torch.rand(B, N, 3, generator=g) - 0.5
    xyz[0, ::5] *= 0.03  # origin-ball skip branch
    if B > 1:
        xyz[1] += torch.tensor([0.0, 0.0, 8.0])
    want = P.furthest_point_sampling(xyz, m)
    got = ext.furthest_point_sampling(xyz.to(dev), m)
    assert torch.equal(got.cpu(), want)


def test_fps_template_cloud_multi_workgroup(dev, ext):
    """get_obj_feats' shape (feature_extraction.py:152-158): 42 x 5000 = 210 000 template points -> 2048, on the
    multi-workgroup kernel (206 workgroups, grid-wide arg-max per round); identical to the C oracle and to the
    one-workgroup global-memory kernel (reached here through a scratch pointer that is not 8-byte aligned)."""
    import time
    from oracle import pointops as P
    from sam6d_hip import _lib
    g = torch.Generator().manual_seed(42)
    xyz = (torch.rand(1, 210000, 3, generator=g) - 0.5) * 0.3
    xyz[0, ::7] *= 0.05  # origin-ball skip branch
    want = P.furthest_point_sampling(xyz, 2048)
    d = xyz.to(dev)
    ext.furthest_point_sampling(d, 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = ext.furthest_point_sampling(d, 2048)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    assert torch.equal(got.cpu(), want)
    out = torch.empty(1, 2048, dtype=torch.int32, device=dev)
    temp = torch.empty(210000 + 1, dtype=torch.float32, device=dev)
    _lib.call("sam6d_furthest_point_sampling", d.data_ptr(), 1, 210000, 2048, temp.data_ptr() + 4, out.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert torch.equal(out.cpu(), want)
    print("\nFPS 210000 -> 2048: multi-workgroup %.1f ms, one workgroup %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))


def test_fps_many_large_clouds_fall_back_to_one_workgroup_each(dev, ext):
    """More workgroups than the co-residency bound of the grid-synchronised kernel (60 clouds x 5 workgroups > 256)."""
    from oracle import pointops as P
    g = torch.Generator().manual_seed(7)
    xyz = torch.rand(60, 5000, 3, generator=g) - 0.5
    assert torch.equal(ext.furthest_point_sampling(xyz.to(dev), 64).cpu(), P.furthest_point_sampling(xyz, 64))


def test_fps_all_points_in_origin_ball(dev, ext):
    from oracle import pointops as P
    xyz = torch.full((1, 256, 3), 0.001)
    assert torch.equal(ext.furthest_point_sampling(xyz.to(dev), 8).cpu(), P.furthest_point_sampling(xyz, 8))


def test_fps_duplicate_points_tie_break(dev, ext):
    from oracle import pointops as P
    g = torch.Generator().manual_seed(5)
    base = torch.rand(1, 64, 3, generator=g) + 1.0
    xyz = base.repeat(1, 8, 1).contiguous()  # every point 8x: exact ties, lowest index must win
    assert torch.equal(ext.furthest_point_sampling(xyz.to(dev), 40).cpu(), P.furthest_point_sampling(xyz, 40))


def test_gather_rows_lead(dev):
    """sam6d_gather_rows_lead: row 0 of every feats[b] is supplied by `lead` -- out row 0 and every gathered index 0 read it (feats row 0
    holds garbage here), out-of-range indices give zeros, the other rows are plain gathers."""
    from sam6d_hip import _lib
    gen = torch.Generator().manual_seed(12)
    B, N, M, Cc = 3, 50, 9, 256
    feats = torch.randn(B, N, Cc, generator=gen)
    lead = torch.randn(B, 4, Cc, generator=gen)  # row 0 of each block is the lead row
    idx = torch.randint(1, N, (B, M), generator=gen, dtype=torch.int32)
    idx[:, 0] = 0
    idx[1, 3] = N + 5
    idx[2, 4] = -2
    want = torch.zeros(B, M + 1, Cc)
    for b in range(B):
        want[b, 0] = lead[b, 0]
        for j in range(M):
            a = int(idx[b, j])
            want[b, 1 + j] = lead[b, 0] if a == 0 else (feats[b, a] if 0 < a < N else 0.0)
    f = feats.clone(); f[:, 0] = float("nan")
    fd, ld, idd = f.to(dev), lead.to(dev), idx.to(dev)
    out = torch.empty(B, M + 1, Cc, device=dev)
    _lib.call("sam6d_gather_rows_lead", fd.data_ptr(), idd.data_ptr(), B, N, M, Cc, N * Cc, (M + 1) * Cc, 0, ld.data_ptr(), 4 * Cc,
              out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)


def test_gather_golden(dev, ext):
    g = golden("pointops")
    gen = torch.Generator().manual_seed(11)
    _ = torch.rand(2, 2048, 3, generator=gen)  # same stream position as oracle/gen_golden.py:fx_pointops
    feats = torch.randn(16, 128, 256, generator=gen)
    idx = torch.from_numpy(g["gather_idx"])
    out = ext.gather_points(feats.to(dev), idx.to(dev))
    assert _sha(out) == str(g["gather_out_sha"])
    assert np.array_equal(out[0].cpu().numpy(), g["gather_out_b0"])
    assert out[0, :, 0].abs().max() == 0 and out[0, :, 1].abs().max() == 0  # idx -1 and 256 -> 0


def test_ball_query_golden(dev, ext):
    g = golden("pointops")
    pts = torch.from_numpy(g["bq_pts"]).to(dev)
    q = (pts + 0.00000001).contiguous()
    i1 = ext.ball_query(q, pts, 0.1, 32)
    i2 = ext.ball_query(q, pts, 0.2, 64)
    assert np.array_equal(i1.cpu().numpy(), g["bq_r1"].astype(np.int32))
    assert np.array_equal(i2.cpu().numpy(), g["bq_r2"].astype(np.int32))
    grp = ext.group_points(pts.transpose(1, 2).contiguous(), i1)
    assert _sha(grp) == str(g["group_hot_sha"])


def test_ball_query_reference_test_shape(dev, ext):
    g = golden("pointops")
    rs = np.random.RandomState(324)
    _ = rs.randn(1, 21000, 3)
    nx = torch.from_numpy(rs.randn(1, 1024, 3).astype(np.float32)).to(dev)
    xx = torch.from_numpy(rs.randn(1, 256, 3).astype(np.float32)).to(dev)
    got = ext.ball_query(nx, xx, 0.1, 64)
    assert np.array_equal(got.cpu().numpy(), g["bq_ref_test"].astype(np.int32))


@pytest.mark.parametrize("B,N,M,r,ns", [(2, 2048, 2048, 0.2, 64), (1, 5000, 300, 0.15, 16), (3, 100, 37, 0.5, 8),
                                         (1, 9000, 65, 0.05, 32)])
def test_ball_query_group_vs_oracle(dev, ext, B, N, M, r, ns):
    from oracle import pointops as P
    g = torch.Generator().manual_seed(N + M)
    xyz = torch.rand(B, N, 3, generator=g) - 0.5
    new = (torch.rand(B, M, 3, generator=g) - 0.5) * 1.3  # some queries have empty balls
    want = P.ball_query(new, xyz, r, ns)
    got = ext.ball_query(new.to(dev), xyz.to(dev), r, ns)
    assert torch.equal(got.cpu(), want)
    feats = torch.randn(B, 5, N, generator=g)
    assert torch.equal(ext.group_points(feats.to(dev), got).cpu(), P.group_points(feats, want))


def test_group_golden(dev, ext):
    g = golden("pointops")
    gen = torch.Generator().manual_seed(11)
    _ = torch.rand(2, 2048, 3, generator=gen)
    _ = torch.randn(16, 128, 256, generator=gen)
    _ = torch.randint(0, 256, (16, 64), generator=gen, dtype=torch.int32)
    _ = torch.rand(2, 2048, 3, generator=gen)
    gf = torch.randn(7, 3, 2048, generator=gen)
    gi = torch.randint(0, 2048, (7, 2048, 32), generator=gen, dtype=torch.int32)
    out = ext.group_points(gf.to(dev), gi.to(dev))
    assert _sha(out) == str(g["group_out_sha"])


def test_gather_rows(dev):
    from sam6d_hip import ops
    g = torch.Generator().manual_seed(3)
    f = torch.randn(3, 300, 256, generator=g)
    idx = torch.randint(0, 300, (3, 50), generator=g, dtype=torch.int32)
    got = ops.gather_rows(f.to(dev), idx.to(dev))
    want = torch.gather(f, 1, idx.long().unsqueeze(2).expand(3, 50, 256))
    assert torch.equal(got.cpu(), want)


def test_error_contract(dev, ext):
    x = torch.zeros(1, 16, 3, device=dev)
    with pytest.raises(RuntimeError):
        ext.ball_query(x.transpose(1, 2), x, 0.1, 4)  # non-contiguous
    with pytest.raises(RuntimeError):
        ext.gather_points(x, torch.zeros(1, 4, dtype=torch.int64, device=dev))  # idx must be int32
    with pytest.raises(NotImplementedError):
        ext.three_nn(x, x)


@pytest.mark.parametrize("B,N,M,r1,ns1,r2,ns2", [(2, 2048, 2048, 0.1, 32, 0.2, 64), (1, 5000, 300, 0.05, 16, 0.4, 64), (3, 100, 100, 0.3, 8, 0.01, 4),
                                                   (3, 3000, 1500, 0.3, 8, 0.02, 5), (5, 1111, 1111, 0.12, 32, 0.5, 64), (32, 2048, 2048, 0.1, 32, 0.2, 64)])
def test_ball_query2_equals_two_single_queries(dev, B, N, M, r1, ns1, r2, ns2):
    """The two-radius pass used by the positional encoding returns exactly what two single-radius calls return (large query sets go
    through the one-lane-per-query kernel, small ones through the wave-per-query kernel)."""
    from sam6d_hip import _lib
    g = torch.Generator().manual_seed(N + M)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    new = (xyz[:, :M] + 1e-8).contiguous() if M <= N else torch.rand(B, M, 3, generator=g).to(dev)
    new[:, 0] = 5.0  # an empty ball
    single = []
    for r, ns in ((r1, ns1), (r2, ns2)):
        idx = torch.full((B, M, ns), -7, dtype=torch.int32, device=dev)
        _lib.call("sam6d_ball_query", new.data_ptr(), xyz.data_ptr(), B, N, M, float(r), ns, idx.data_ptr(), None)
        single.append(idx)
    i1 = torch.full((B, M, ns1), -7, dtype=torch.int32, device=dev)
    i2 = torch.full((B, M, ns2), -7, dtype=torch.int32, device=dev)
    _lib.call("sam6d_ball_query2", new.data_ptr(), xyz.data_ptr(), B, N, M, float(r1), ns1, i1.data_ptr(), float(r2), ns2,
              i2.data_ptr(), None)
    torch.cuda.synchronize()
    assert torch.equal(i1, single[0]) and torch.equal(i2, single[1])


@pytest.mark.parametrize("B,N,M,r1,ns1,r2,ns2,kind", [
    (2, 2048, 2048, 0.1, 32, 0.2, 64, "cube"), (32, 2048, 2048, 0.1, 32, 0.2, 64, "cube"), (1, 5000, 300, 0.05, 16, 0.4, 64, "cube"),
    (3, 100, 100, 0.3, 8, 0.01, 4, "cube"), (3, 3000, 1500, 0.3, 8, 0.02, 5, "cube"), (2, 2048, 2048, 0.1, 32, 0.2, 64, "dense"),
    (2, 2048, 2048, 0.1, 32, 0.2, 64, "outliers"), (2, 7000, 900, 0.2, 32, 0.1, 64, "surface"), (1, 5, 9, 0.5, 4, 1.0, 8, "cube"),
    (2, 2048, 2048, 0.1, 32, 0.2, 64, "lattice"), (1, 9000, 64, 0.1, 32, 0.2, 64, "cube")])
def test_ball_query2_grid_equals_all_pairs_scan(dev, B, N, M, r1, ns1, r2, ns2, kind):
    """sam6d_ball_query2_grid (cell grid + per-query hit bit mask) returns, bit for bit, what the all-pairs scan returns -- which the
    tests above pin to EXT/src/ball_query.cpp:16-62 through the oracle / the compiled reference: uniform cubes, a cloud denser than the
    ball (every point a hit: the first-nsample-in-index-order rule decides), far outliers (clamped cells), a thin surface, queries far
    outside the cloud, points exactly on cell boundaries (lattice), N beyond the pruned kernel's range (falls back)."""
    from oracle import pointops as P
    from sam6d_hip import _lib
    g = torch.Generator().manual_seed(N * 3 + M + len(kind))
    xyz = torch.rand(B, N, 3, generator=g) - 0.5
    if kind == "dense":
        xyz = xyz * 0.08
    elif kind == "outliers":
        xyz[:, ::97] *= 400.0
        xyz[:, 5] = -1.0e4
    elif kind == "surface":
        xyz[..., 2] = 0.02 * torch.sin(6 * xyz[..., 0])
    elif kind == "lattice":
        xyz = torch.round(xyz * 10) * 0.1001 * 2  # coordinates on multiples of the cell edge 0.2002
    new = (xyz[:, :M] + 1e-8).contiguous() if M <= N else (torch.rand(B, M, 3, generator=g) - 0.5)
    new[:, 0] = 5.0  # an empty ball, outside the cloud's box
    if M > 2:
        new[:, 1] = -7.0
    xyz, new = xyz.to(dev).contiguous(), new.to(dev).contiguous()
    a1 = torch.full((B, M, ns1), -7, dtype=torch.int32, device=dev)
    a2 = torch.full((B, M, ns2), -7, dtype=torch.int32, device=dev)
    _lib.call("sam6d_ball_query2", new.data_ptr(), xyz.data_ptr(), B, N, M, float(r1), ns1, a1.data_ptr(), float(r2), ns2, a2.data_ptr(), None)
    g1 = torch.full((B, M, ns1), -9, dtype=torch.int32, device=dev)
    g2 = torch.full((B, M, ns2), -9, dtype=torch.int32, device=dev)
    nbytes = int(_lib.load().sam6d_ball_query2_grid_workspace_bytes(B, N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.call("sam6d_ball_query2_grid", new.data_ptr(), xyz.data_ptr(), B, N, M, float(r1), ns1, g1.data_ptr(), float(r2), ns2, g2.data_ptr(),
              ws.data_ptr(), nbytes, None)
    torch.cuda.synchronize()
    assert torch.equal(g1, a1), "radius %g: %d of %d query rows differ" % (r1, int((g1 != a1).any(dim=2).sum()), B * M)
    assert torch.equal(g2, a2), "radius %g: %d of %d query rows differ" % (r2, int((g2 != a2).any(dim=2).sum()), B * M)
    if B * N * M <= 3 * 3000 * 1500:  # and against the oracle itself where that takes seconds
        assert torch.equal(g1.cpu(), P.ball_query(new.cpu(), xyz.cpu(), r1, ns1))
        assert torch.equal(g2.cpu(), P.ball_query(new.cpu(), xyz.cpu(), r2, ns2))


def test_fps_grid_barrier_abort_falls_back(dev):
    """The multi-workgroup FPS (N > 4096) closes every round with a hand-rolled grid-wide barrier that needs all its workgroups on the
    chip.  When a workgroup gives up waiting (forced here: spin cap 0) the cloud is recomputed by the one-workgroup kernel queued behind
    it -- the call must still return the exact indices, never a truncated list."""
    from oracle import pointops as P
    from sam6d_hip import _lib, ops
    g = torch.Generator().manual_seed(41)
    xyz = torch.rand(2, 9000, 3, generator=g) - 0.5
    want = P.furthest_point_sampling(xyz, 300)
    try:
        _lib.call("sam6d_fps_debug_spin_cap", 0)
        got = ops.furthest_point_sampling(xyz.to(dev), 300)
        torch.cuda.synchronize()
    finally:
        _lib.call("sam6d_fps_debug_spin_cap", -1)
    assert torch.equal(got.cpu(), want.to(torch.int32)), "indices after a forced abort of the grid kernel"
    again = ops.furthest_point_sampling(xyz.to(dev), 300)  # default cap: the grid kernel completes
    assert torch.equal(again.cpu(), want.to(torch.int32))
