"""ctypes handles on the checker libraries used by the tests.

oracle()  -> oracle/liboracle.so         (CPU restatement; always available)
ref()     -> oracle/_ref/libdaalaref.so  (the real reference, when built; None
                                          otherwise - tests then rely on the
                                          committed golden vectors)
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

_cache = {}


def _load(path):
    if path not in _cache:
        _cache[path] = ctypes.CDLL(path) if os.path.exists(path) else None
    return _cache[path]


def oracle():
    lib = _load(os.path.join(ROOT, "oracle", "liboracle.so"))
    assert lib is not None, "oracle/liboracle.so missing: make -C oracle"
    lib.odo_pvq_search_rdo_double.restype = ctypes.c_double
    lib.odo_now.restype = ctypes.c_double
    return lib


def ref():
    lib = _load(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so"))
    if lib is not None:
        lib.ref_pvq_search_rdo_double.restype = ctypes.c_double
        lib.ref_now.restype = ctypes.c_double
    return lib


def ref_simd():
    """The same reference library built with its x86 intrinsics (oracle/Makefile); None when
    not built."""
    return _load(os.path.join(ROOT, "oracle", "_ref", "libdaalaref_simd.so"))


def P(a):
    """numpy array -> void* (array must stay alive and be C-contiguous)."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def lcg_stream(seed, n):
    """The survey's generator: x = x*1103515245 + 12345 (mod 2^32), bits 16.."""
    out = np.empty(n, dtype=np.int64)
    x = seed & 0xFFFFFFFF
    for i in range(n):
        x = (x * 1103515245 + 12345) & 0xFFFFFFFF
        out[i] = x >> 16
    return out


def synth_frame(w, h, seed=12345, phase=0):
    """Synthetic 8-bit 4:2:0 frame: ramp + 32-px checker (+-20) + noise (+-8)
    (SURVEY.md section 8(d), C1/C2 generator; vectorised LCG-free variant with a
    fixed numpy seed so that it is cheap at 1080p)."""
    rng = np.random.RandomState(seed + phase)
    planes = []
    for (pw, ph, base) in ((w, h, 0), (w // 2, h // 2, 40), (w // 2, h // 2, 80)):
        yy, xx = np.mgrid[0:ph, 0:pw]
        ramp = ((xx + phase) * 96 // max(pw, 1) + yy * 64 // max(ph, 1)) - 80
        checker = ((((xx + phase) // 32) + (yy // 32)) & 1) * 40 - 20
        noise = rng.randint(-8, 9, size=(ph, pw))
        p = np.clip(128 + ramp + checker + noise + base // 8, 0, 255)
        planes.append(p.astype(np.uint8))
    return planes
