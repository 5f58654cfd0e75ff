"""The C-ABI library loads and exports every symbol include/sam6d_hip.h declares (no compute, CPU-only)."""
import ctypes
import os
import re

from tests._util import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "sam6d_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sam6d_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    from sam6d_hip import _lib
    names = _declared()
    assert len(names) >= 5
    assert os.path.exists(_lib.LIB_PATH), "build libsam6d_hip.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in sam6d_hip.h but not exported" % n
    bound = set(_lib.SIGNATURES) | {"sam6d_last_error", "sam6d_abi_version", "sam6d_get_matmul_mode"}
    assert set(names) == bound, (set(names) ^ bound)
    _lib.load()
    assert _lib.load().sam6d_abi_version() >= 1


def test_every_declaration_cites_the_reference():
    txt = open(os.path.join(ROOT, "include", "sam6d_hip.h")).read()
    # each entry point's comment names the reference file:line it replaces
    blocks = re.findall(r"/\*((?:(?!\*/).)*?)\*/\s*int\s+(sam6d_[a-z0-9_]+)\s*\(", txt, flags=re.S)
    assert blocks
    for comment, name in blocks:
        if name in ("sam6d_abi_version", "sam6d_get_matmul_mode"):
            continue
        assert re.search(r"\.(cpp|py|h):\d+", comment), "%s: no reference file:line in its comment" % name


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from sam6d_hip import ops
    with pytest.raises(RuntimeError):
        ops.furthest_point_sampling(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError):
        ops.ball_query(torch.zeros(1, 8, 3), torch.zeros(1, 8, 3).transpose(1, 2), 0.1, 4)
