"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/pointops_oracle.c (CPU restatement of the
reference's point-cloud primitives; see that file's header for the reference file:line of each loop).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_pointops.so")
_lib = None


def build():
    src = os.path.join(_HERE, "pointops_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fp(t):
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def _ip(t):
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int))


def _chk(t, dtype):
    assert t.device.type == "cpu" and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())


def furthest_point_sampling(xyz, m):
    """(B,N,3) f32 -> (B,m) i32   [EXT/src/sampling.cpp:76-118,184-212]"""
    _chk(xyz, torch.float32)
    B, N, _ = xyz.shape
    out = torch.zeros(B, m, dtype=torch.int32)
    lib().orc_furthest_point_sampling(B, N, m, _fp(xyz), _ip(out))
    return out


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M)   [EXT/src/sampling.cpp:23-44,120-150]"""
    _chk(points, torch.float32); _chk(idx, torch.int32)
    B, C, N = points.shape
    M = idx.shape[1]
    out = torch.zeros(B, C, M, dtype=torch.float32)
    lib().orc_gather_points(B, C, N, M, _fp(points), _ip(idx), _fp(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (B,M,3), xyz (B,N,3) -> (B,M,nsample) i32   [EXT/src/ball_query.cpp:16-93]"""
    _chk(new_xyz, torch.float32); _chk(xyz, torch.float32)
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    out = torch.zeros(B, M, nsample, dtype=torch.int32)
    lib().orc_ball_query(B, N, M, ctypes.c_float(radius), nsample, _fp(new_xyz), _fp(xyz), _ip(out))
    return out


def group_points(points, idx):
    """(B,C,N) f32, (B,M,S) i32 -> (B,C,M,S)   [EXT/src/group_points.cpp:20-45,79-108]"""
    _chk(points, torch.float32); _chk(idx, torch.int32)
    B, C, N = points.shape
    _, M, S = idx.shape
    out = torch.zeros(B, C, M, S, dtype=torch.float32)
    lib().orc_group_points(B, C, N, M, S, _fp(points), _ip(idx), _fp(out))
    return out


def pairwise_distance(x, y):
    """(B,N,3),(B,M,3) -> (B,N,M) squared distances, torch-CPU bit recipe   [PEM/utils/model_utils.py:101-128]"""
    x = x.contiguous(); y = y.contiguous()
    _chk(x, torch.float32); _chk(y, torch.float32)
    assert x.shape[-1] == 3 and y.shape[-1] == 3
    lead = x.shape[:-2]
    xb = x.reshape(-1, x.shape[-2], 3); yb = y.reshape(-1, y.shape[-2], 3)
    B, N, _ = xb.shape
    M = yb.shape[1]
    out = torch.empty(B, N, M, dtype=torch.float32)
    lib().orc_pairwise_distance(B, N, M, _fp(xb), _fp(yb), _fp(out))
    return out.reshape(*lead, N, M)


def cumsum_f32(x):
    """torch.cumsum(float32, dim=1) CPU semantics (double accumulator)."""
    _chk(x, torch.float32)
    R, N = x.shape
    out = torch.empty_like(x)
    lib().orc_cumsum_f32_via_f64(R, N, _fp(x), _fp(out))
    return out


def first_ge(cum, u):
    """first index i with cum[r,i] >= u[r,s]; 0 if none   [PEM/utils/model_utils.py:277-305]"""
    _chk(cum, torch.float32); _chk(u, torch.float32)
    R, N = cum.shape
    ns = u.shape[1]
    out = torch.empty(R, ns, dtype=torch.int64)
    lib().orc_first_ge(R, N, ns, _fp(cum), _fp(u), ctypes.cast(out.data_ptr(), ctypes.POINTER(ctypes.c_int64)))
    return out
