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
    bound = set(_lib.SIGNATURES) | {"sam6d_last_error", "sam6d_abi_version", "sam6d_get_matmul_mode", "sam6d_get_thread_matmul_mode"}
    assert set(names) == bound, (set(names) ^ bound)
    _lib.load()
    hdr = open(os.path.join(ROOT, "include", "sam6d_hip.h")).read()
    ver = int(re.search(r"#define\s+SAM6D_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert _lib.load().sam6d_abi_version() == ver == _lib.ABI_VERSION, "header, library and Python binding disagree on the ABI version"


def test_every_declaration_cites_the_reference():
    txt = open(os.path.join(ROOT, "include", "sam6d_hip.h")).read()
    # each entry point's comment names the reference file:line it replaces
    blocks = re.findall(r"/\*((?:(?!\*/).)*?)\*/\s*int\s+(sam6d_[a-z0-9_]+)\s*\(", txt, flags=re.S)
    assert blocks
    for comment, name in blocks:
        if name in ("sam6d_abi_version", "sam6d_get_matmul_mode", "sam6d_get_thread_matmul_mode", "sam6d_set_thread_matmul_mode"):
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


def _gfx950_code_objects(path):
    """Every gfx950 code object in the library's clang offload bundles (one bundle per translation unit)."""
    import struct
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = data.find(magic)
    while at >= 0:
        count = struct.unpack_from("<Q", data, at + 24)[0]
        off = at + 32
        for _ in range(count):
            o, s, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and s:
                yield data[at + o:at + o + s]
        at = data.find(magic, at + 1)


def test_no_packed_fp32_math_in_device_code(tmp_path):
    """DESIGN.md (concurrency caveat): compiler-formed v_pk_{mul,add,fma}_f32 returned wrong lanes on MI355X while
    f16-MFMA kernels ran on another stream, so the library is built with -fno-slp-vectorize -fno-vectorize and the
    kernels spell their fp32 math as scalar fmaf.  This guards the flags: no packed-fp32 arithmetic in any kernel."""
    import subprocess
    from sam6d_hip import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        import pytest
        pytest.skip("llvm-objdump not available")
    n_obj = n_mfma = 0
    for k, blob in enumerate(_gfx950_code_objects(_lib.LIB_PATH)):
        p = tmp_path / ("co%d.elf" % k)
        p.write_bytes(blob)
        asm = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(p)], capture_output=True, text=True, check=True).stdout
        hits = re.findall(r"^\s*(v_pk_(?:fma|mul|add)_f32\b.*)$", asm, flags=re.M)
        assert not hits, "packed fp32 math in code object %d: %s" % (k, hits[:3])
        n_mfma += len(re.findall(r"\bv_mfma_", asm))
        n_obj += 1
    assert n_obj >= 5 and n_mfma > 0, "disassembly looks empty (%d code objects, %d MFMA)" % (n_obj, n_mfma)


def test_header_is_plain_c_and_the_c_consumer_builds():
    """include/sam6d_hip.h compiles as C99 with gcc and links against the library: tests/cabi/cabi_check.c is a consumer with no
    Python or torch in it (it runs on the GPU box: tests/test_cabi_c_gpu.py)."""
    import subprocess
    from sam6d_hip import _lib
    assert os.path.exists(_lib.LIB_PATH)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cabi"), "-s", "-B"])
    assert os.path.exists(os.path.join(ROOT, "tests", "cabi", "cabi_check"))
    src = open(os.path.join(ROOT, "tests", "cabi", "cabi_check.c")).read()
    assert "Python.h" not in src and "torch" not in src.replace("no torch", "")
