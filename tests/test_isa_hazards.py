"""Static guard (no GPU): the gfx950 assembly of every kernel file holds no wide vector store with a REGISTER soffset whose data
registers the VALU overwrites within two instructions -- the pattern behind the store-data hazard of DESIGN.md section 3 (LLVM
inserts no wait state for it; with two workgroups per CU the store then read part of its data after the overwrite)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not installed")
def test_no_register_soffset_store_is_overwritten_right_away():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_store_hazard_scan.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 wide store(s)" in r.stdout
