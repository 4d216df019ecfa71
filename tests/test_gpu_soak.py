"""Short slices of the randomised soaks (tests/soak/parity_soak.py, tests/soak/decode_soak.py, tests/soak/encode_soak.py; their long runs
are recorded in profiles/r6_*_soak.txt): a few dozen random cases each, so that every run of the GPU suite also covers
picture sizes, quantisers and switches nobody wrote down."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
HAVE_REF = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so"))


def _tool(name, *args, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak", name)] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=1200, env=e)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    return p.stdout.strip().splitlines()[-1]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
def test_parity_soak_slice():
    """40 random cases of the priced step against the compiled reference (pixels of every level, gain / theta / K /
    pulses of every band): sizes 32..998 x 32..598, quantisers 8..6574, keyframes / inter frames, batches, FPR modes."""
    last = _tool("parity_soak.py", 40, 500000)
    assert last.startswith("parity soak: 40 cases equal"), last


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
def test_parity_soak_slice_large_pictures():
    last = _tool("parity_soak.py", 4, 600000, SOAK_BIG=1)
    assert last.startswith("parity soak: 4 cases equal"), last


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
def test_decode_soak_slice():
    """Random clips through the real reference encoder + decoder with odhip_inverse_partition, the synthesis and the
    deringing cache bound in turn (about 15 s)."""
    last = _tool("decode_soak.py", 15, 700000)
    assert last.startswith("decode soak:") and " clips equal" in last, last
    assert int(last.split()[2]) >= 3, last


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
def test_encode_soak_slice():
    """Random clips through the real reference encoder with the batched / per-call bindings: packets byte-identical
    (about 20 s)."""
    last = _tool("encode_soak.py", 20, 800000)
    assert last.startswith("encode soak:") and "byte-identical" in last, last
    assert int(last.split()[2]) >= 3, last
