"""Tensor-level wrappers over the C ABI: argument checks (the reference's TORCH_CHECK contract,
EXT/include/utils.h:20-45), output allocation on the input's device, launch on torch's current HIP stream.

Device memory, streams and allocation are torch's (plumbing); all arithmetic happens in libsam6d_hip.so.
"""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s must be a tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a %s tensor" % (name, "float" if dtype == torch.float32 else "int"))
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError("%s must have %d dimensions (got %d)" % (name, ndim, t.dim()))
    if not t.is_cuda:
        raise RuntimeError("%s must be a HIP device tensor (this build has no CPU path; the CPU oracle lives in "
                           "oracle/ and is test-only)" % name)


def _same_dev(a, b, na, nb):
    if a.device != b.device:
        raise RuntimeError("%s and %s must be on the same device" % (na, nb))


def p(t):
    return t.data_ptr()


def furthest_point_sampling(xyz, m):
    """(B,N,3) f32 -> (B,m) i32.  Bit-exact with EXT/src/sampling.cpp:76-118."""
    _chk(xyz, "points", torch.float32, 3)
    if xyz.shape[2] != 3:
        raise RuntimeError("points must be (B,N,3)")
    B, N, _ = xyz.shape
    with torch.cuda.device(xyz.device):
        out = torch.empty(B, m, dtype=torch.int32, device=xyz.device)
        temp = torch.empty(B, N, dtype=torch.float32, device=xyz.device) if N > 4096 else None
        _lib.call("sam6d_furthest_point_sampling", p(xyz), B, N, m, p(temp) if temp is not None else None, p(out),
                  _stream())
    return out


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M).  EXT/src/sampling.cpp:23-44."""
    _chk(points, "points", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 2)
    _same_dev(points, idx, "points", "idx")
    B, C, N = points.shape
    M = idx.shape[1]
    with torch.cuda.device(points.device):
        out = torch.empty(B, C, M, dtype=torch.float32, device=points.device)
        _lib.call("sam6d_gather_points", p(points), p(idx), B, C, N, M, p(out), _stream())
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (B,M,3), xyz (B,N,3) -> (B,M,nsample) i32.  EXT/src/ball_query.cpp:16-62."""
    _chk(new_xyz, "new_xyz", torch.float32, 3)
    _chk(xyz, "xyz", torch.float32, 3)
    _same_dev(new_xyz, xyz, "new_xyz", "xyz")
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    with torch.cuda.device(xyz.device):
        out = torch.empty(B, M, nsample, dtype=torch.int32, device=xyz.device)
        _lib.call("sam6d_ball_query", p(new_xyz), p(xyz), B, N, M, float(radius), int(nsample), p(out), _stream())
    return out


def group_points(points, idx):
    """(B,C,N) f32, (B,M,S) i32 -> (B,C,M,S).  EXT/src/group_points.cpp:20-45."""
    _chk(points, "points", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 3)
    _same_dev(points, idx, "points", "idx")
    B, C, N = points.shape
    _, M, S = idx.shape
    with torch.cuda.device(points.device):
        out = torch.empty(B, C, M, S, dtype=torch.float32, device=points.device)
        _lib.call("sam6d_group_points", p(points), p(idx), B, C, N, M, S, p(out), _stream())
    return out


def gather_rows(feats, idx, idx_off=0, out=None):
    """feats (B,N,C) f32, idx (B,M) i32 -> (B,M,C): out[b,j] = feats[b, idx[b,j]+idx_off]."""
    _chk(feats, "feats", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 2)
    B, N, C = feats.shape
    M = idx.shape[1]
    with torch.cuda.device(feats.device):
        if out is None:
            out = torch.empty(B, M, C, dtype=torch.float32, device=feats.device)
        _lib.call("sam6d_gather_rows", p(feats), p(idx), B, N, M, C, N * C, out.stride(0), idx_off, p(out), _stream())
    return out
