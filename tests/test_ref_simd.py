"""The x86-intrinsics build of the reference (oracle/_ref/libdaalaref_simd.so, bench.py's
cpu_baseline_simd) runs the stage with its SSE4.1 / AVX2 4x4 and 8x8 transforms and
reconstructs exactly what the plain C build reconstructs."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _libs import ref, ref_simd  # noqa: E402


@pytest.mark.skipif(ref() is None or ref_simd() is None, reason="oracle/_ref not built")
def test_simd_build_equals_plain_c_on_a_whole_frame():
    import _pipeline_check as C
    import bench
    from daala_amd.quant import QuantTables
    qt = QuantTables.load()
    full = bench.synth_frame_np(2, 31)
    pw, ph = 200, 136
    pics = [full[0][:ph, :pw], full[1][:ph // 2, :pw // 2], full[2][:ph // 2, :pw // 2]]
    simd = ref_simd()
    assert simd.ref_stage_simd() >= 1, "no SSE2 on this host?"
    assert ref().ref_stage_simd() == 0
    a, blocks_a, _ = C.cpu_frame(qt, pics, pw, ph)
    b, blocks_b, _ = C.cpu_frame(qt, pics, pw, ph, lib=simd)
    assert blocks_a == blocks_b
    for pli in range(3):
        for bs in range(len(a[pli])):
            assert np.array_equal(a[pli][bs], b[pli][bs]), (pli, bs)
