"""The C ABI driven from plain C (tests/cabi/cabi_check.c: gcc, hipMalloc'd buffers, no Python / torch in the process), with the C
oracle as the checker.  The binary is built by __graft_entry__.build() and travels to the GPU box with the snapshot."""
import os
import subprocess

import pytest

from tests._util import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_consumer_of_the_abi(dev):
    exe = os.path.join(ROOT, "tests", "cabi", "cabi_check")
    assert os.path.exists(exe), "tests/cabi/cabi_check is missing: run __graft_entry__.build() (make -C tests/cabi)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)  # a child process; this one keeps its GPU context
    print(r.stdout)
    assert r.returncode == 0, "cabi_check failed (rc=%d):\n%s\n%s" % (r.returncode, r.stdout, r.stderr)
    assert "cabi_check: ok" in r.stdout and r.stdout.count("bit-exact") == 3
