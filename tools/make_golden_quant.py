#!/usr/bin/env python3
"""tests/golden/quant_multi.npz: the quantiser set-up of the REAL reference encoder
(oracle/_ref, ref_dump_quant_tables2: one 64x64 keyframe encoded, then state.quantizer,
rc.base_quantizer, state.pvq_qm_q4, state.qm, state.qm_inv read back) at several
qualities incl. -v 0 (lossless), with activity masking on / off and the flat / HVS
matrices.  Pins the library's a17 code (daala_amd/csrc/quant.hip) on boxes where the
reference is absent.  Dev-container only."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import GOLDEN, P, ref  # noqa: E402

QUALITIES = (0, 1, 5, 10, 20, 40, 100, 511)


def main():
    r = ref()
    assert r is not None
    rows = []
    pq_all = []
    qms = {}
    for v in QUALITIES:
        for masking in (0, 1):
            for hvs in (0, 1):
                q = ctypes.c_int()
                bq = ctypes.c_int()
                pq = np.zeros(90, np.uint8)
                qm = np.zeros(10912, np.int16)
                qmi = np.zeros(10912, np.int16)
                n = r.ref_dump_quant_tables2(v, masking, hvs, ctypes.byref(q), ctypes.byref(bq),
                                             P(pq), P(qm), P(qmi))
                assert n == 30
                rows.append((v, masking, hvs, q.value, bq.value))
                pq_all.append(pq.reshape(3, 30).copy())
                if hvs in qms:
                    assert np.array_equal(qms[hvs][0], qm) and np.array_equal(qms[hvs][1], qmi)
                qms[hvs] = (qm, qmi)
    np.savez_compressed(os.path.join(GOLDEN, "quant_multi.npz"), rows=np.array(rows, np.int32),
                        pvq_qm_q4=np.stack(pq_all), qm_flat=qms[0][0], qm_inv_flat=qms[0][1],
                        qm_hvs=qms[1][0], qm_inv_hvs=qms[1][1])
    print(len(rows), "table sets")


if __name__ == "__main__":
    main()
