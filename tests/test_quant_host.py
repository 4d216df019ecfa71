"""Row a17 in the library (daala_amd/csrc/quant.hip): od_init_qm, od_interp_qm, the
per-band steps and OD_PVQ_BETA against the reference encoder's own state - committed
fixtures (tests/golden/quant_multi.npz, quant_v20.npz, made from oracle/_ref by
tools/make_golden_quant.py / make_golden.py) and, when the compiled reference is
present, live.  Host code only: runs without a GPU."""
import ctypes

import numpy as np
import pytest

from _libs import GOLDEN, P, ref

import daala_amd.quant as Q

NBANDS = [1, 4, 7, 9, 9]


def load(name):
    import os
    return np.load(os.path.join(GOLDEN, name))


def _filled(qt_arr, offs):
    """The entries od_init_qm writes: min(n*n, 512) per (block size, decimation)."""
    out = []
    for bs in range(5):
        n = 4 << bs
        for d in range(2):
            o = int(offs[bs][d])
            out.append(qt_arr[o:o + min(n * n, 512)])
    return np.concatenate(out)


def test_quant_setup_matches_reference_fixture():
    g = load("quant_multi.npz")
    assert len(g["rows"]) >= 5 * 4 and 0 in g["rows"][:, 0]
    for row, pq in zip(g["rows"], g["pvq_qm_q4"]):
        v, masking, hvs, quantizer, base_q = [int(x) for x in row]
        if v in Q.QUALITY_QUANTIZERS:
            assert Q.QUALITY_QUANTIZERS[v] == (base_q, quantizer)
        qt = Q.QuantTables(base_q, quantizer, masking, hvs)
        assert np.array_equal(qt.pvq_qm_q4, pq), (v, masking, hvs)
        name = "hvs" if hvs else "flat"
        assert np.array_equal(_filled(qt.qm, qt.qm_offset), _filled(g["qm_" + name], qt.qm_offset))
        assert np.array_equal(_filled(qt.qm_inv, qt.qm_offset),
                              _filled(g["qm_inv_" + name], qt.qm_offset))
        for pli in range(3):
            for bs in range(5):
                q0 = max(1, quantizer)
                idx = [bs * (bs + 1) + b - b // 3 for b in range(1, NBANDS[bs] + 1)]
                assert qt.q_band(pli, bs) == [max(1, q0 * int(pq[pli][i]) >> 4) for i in idx]


def test_v20_tables_equal_the_round1_fixture():
    """The dumped -v 20 table set the round-1 kernels were validated with."""
    g = load("quant_v20.npz")
    qt = Q.QuantTables.load()
    assert qt.quantizer == int(g["quantizer"]) == 243
    assert np.array_equal(qt.pvq_qm_q4, g["pvq_qm_q4"])
    assert np.array_equal(qt.qm_offset, g["qm_offset"])
    assert np.array_equal(_filled(qt.qm, qt.qm_offset), _filled(g["qm"], qt.qm_offset))
    assert np.array_equal(_filled(qt.qm_inv, qt.qm_offset), _filled(g["qm_inv"], qt.qm_offset))
    flat = Q.QuantTables(320, 243, 1, 0)
    assert np.array_equal(_filled(flat.qm, qt.qm_offset), _filled(g["qm_flat"], qt.qm_offset))
    for m in range(2):
        for pli in range(3):
            for bs in range(5):
                assert qt.beta_band(pli, bs, masking=m) == g["beta"][m, pli, bs, :NBANDS[bs]].tolist()
    for bs in range(5):
        for b in range(13):
            assert Q.lib().odhip_qm_get_index(bs, b) == int(g["qm_index"][bs, b])


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not built here")
@pytest.mark.parametrize("quality", [0, 3, 17, 63, 200, 300, 400])
def test_quant_setup_vs_reference_live(quality):
    r = ref()
    for masking in (0, 1):
        q, bq = ctypes.c_int(), ctypes.c_int()
        pq = np.zeros(90, np.uint8)
        qm = np.zeros(Q.QM_BUFFER_SIZE, np.int16)
        qmi = np.zeros(Q.QM_BUFFER_SIZE, np.int16)
        assert r.ref_dump_quant_tables2(quality, masking, 1, ctypes.byref(q), ctypes.byref(bq),
                                        P(pq), P(qm), P(qmi)) == 30
        qt = Q.QuantTables(bq.value, q.value, masking, 1)
        assert np.array_equal(qt.pvq_qm_q4, pq.reshape(3, 30))
        assert np.array_equal(_filled(qt.qm, qt.qm_offset), _filled(qm, qt.qm_offset))
        assert np.array_equal(_filled(qt.qm_inv, qt.qm_offset), _filled(qmi, qt.qm_offset))
