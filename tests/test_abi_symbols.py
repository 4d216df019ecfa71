"""CPU: the C-ABI library loads and exports every symbol include/daala_hip.h
declares (no compute calls: there is no GPU here), and it does NOT link or load
anything from oracle/."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from daala_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built):
    import daala_amd
    L = daala_amd.lib()
    hdr = open(os.path.join(ROOT, "include", "daala_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(od_[a-z0-9_]+_hip|odhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert b"gfx950" in L.odhip_version()


def test_product_does_not_depend_on_oracle(built):
    out = subprocess.run(["ldd", built], capture_output=True, text=True).stdout
    assert "oracle" not in out and "daalaref" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "daala_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "liboracle" not in src and "od_oracle" not in src, f


def test_argument_validation_without_gpu(built):
    """Pure host-side validation paths return OD_EINVAL before touching HIP."""
    import ctypes
    import daala_amd
    L = daala_amd.lib()
    assert L.odhip_fdct2d_batch(7, None, None, ctypes.c_long(1), 0, None) == -10
    assert L.odhip_fdct2d_batch(0, None, None, ctypes.c_long(0), 0, None) == 0
    assert L.odhip_pvq_search_batch(None, 16, None, None, None, ctypes.c_double(0.1), None,
                                    None, ctypes.c_long(1), None) == -10
    nb = ctypes.c_int()
    offs = (ctypes.c_int * 13)()
    ln = ctypes.c_int()
    assert L.odhip_pvq_band_layout(2, ctypes.byref(nb), offs, ctypes.byref(ln)) == 0
    assert nb.value == 7 and ln.value == 256
    assert [offs[i] for i in range(8)] == [1, 16, 24, 32, 64, 96, 128, 256]
