"""DESIGN.md describes what ships: every kernel it names exists in the DEFAULT build of the library, and
the file stays a design document (the round-by-round narrative lives in HISTORY.md)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "daala_amd", "lib", "libdaalahip.so")


def test_every_kernel_named_in_design_is_in_the_default_library():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(text.splitlines()) <= 500
    out = subprocess.run(["nm", "-C", LIB], capture_output=True, text=True, check=True).stdout
    have = set(re.findall(r"\bk_[a-z0-9_]+", out))
    named = set()
    for m in re.finditer(r"`(k_[a-z0-9_]+)([^`]*)`", text):
        base, rest = m.group(1), m.group(2)
        named.add(base)
        # `k_hist/prefix/scatter`, `k_prep_corner/lane/wide`: alternatives share the prefix up to the last underscore
        if rest.startswith("/") and "<" not in rest.split("/")[1]:
            stem = base[:base.rfind("_") + 1]
            for alt in rest.split("/")[1:]:
                alt = re.match(r"[a-z0-9_]+", alt)
                if alt:
                    named.add(stem + alt.group(0))
    named = {n.rstrip("_") for n in named}
    missing = sorted(n for n in named if n not in have and not any(h.startswith(n) for h in have))
    assert not missing, "DESIGN.md names kernels the default build does not have: %s" % missing
    assert len(named) > 40
