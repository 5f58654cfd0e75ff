#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds the reference's own pointnet2 `_ext` (CPU loops) from the
sources where they lie under /root/reference into oracle/_ref/ (git-ignored, never committed).

Mirrors the reference's build recipe PEM/model/pointnet2/setup.py:51-58 (CppExtension, -O3,
-DCUDA_AVAILABLE=0) without running the reference's own build system.  Nothing from the reference
is copied into the repo; only the compiled `_ext.so` lands in oracle/_ref/.

Usage: python oracle/build_ref.py          (no-op when /root/reference is absent, e.g. on the GPU box)
"""
import glob
import os
import sys

REF = "/root/reference/SAM-6D/Pose_Estimation_Model/model/pointnet2/_ext_src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def build(verbose=False):
    if not os.path.isdir(REF):
        print("[oracle/build_ref] /root/reference absent - skipping (prebuilt oracle/_ref is used if present)")
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    ext = load(
        name="_ext",
        sources=sorted(glob.glob(REF + "/src/*.cpp")),
        extra_include_paths=[REF + "/include"],
        extra_cflags=["-O3", "-DCUDA_AVAILABLE=0"],
        build_directory=OUT,
        verbose=verbose,
    )
    return ext


if __name__ == "__main__":
    e = build(verbose=True)
    print("built:", e)
