"""ISA-level check of the built kernels (CPU only: hipcc cross-compiles): the gfx950 store-data hazard round 5 ran into -- a 16-byte
vector-memory store whose data registers are overwritten by a VALU instruction less than two issue cycles later (hipcc pads it
itself except for buffer stores with a register in the soffset field; an s_waitcnt in between does not count). See
jxlatte_amd/csrc/k_modular_vh.hip (vh_store) and tools/scan_store_hazard.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_store_data_hazard_in_any_kernel():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_store_hazard.py"), "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "total 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
