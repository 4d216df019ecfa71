"""Host-side quantiser set-up for the PVQ band stage.

In the reference this is per-frame host initialisation:
od_init_qm (src/pvq.c:322-381) fills state.qm / qm_inv, od_interp_qm
(src/encode.c:2903-2940) fills state.pvq_qm_q4, and od_pvq_encode derives the
per-band step (src/pvq_encoder.c:874).  The kernels take the resulting tables
as plain data.  `QuantTables.load()` reads a table set that was dumped from the
reference at encoder_example's `-v 20` (tests/golden/quant_v20.npz,
tools/make_golden.py); an integrating encoder passes its own.
"""
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OD_PVQ_LAMBDA = 0.147  # src/pvq.h:49; enc->pvq_norm_lambda, src/rate.c:1077
NBANDS = [1, 4, 7, 9, 9]


class QuantTables:
    def __init__(self, npz):
        self.quantizer = int(npz["quantizer"])
        self.pvq_qm_q4 = npz["pvq_qm_q4"]
        self.qm = npz["qm"]
        self.qm_inv = npz["qm_inv"]
        self.qm_offset = npz["qm_offset"]
        self.qm_index = npz["qm_index"]
        self.beta = npz["beta"]

    @classmethod
    def load(cls, path=None):
        path = path or os.path.join(_ROOT, "tests", "golden", "quant_v20.npz")
        return cls(np.load(path))

    def q_band(self, pli, bs):
        """max(1, q0*pvq_qm_q4[od_qm_get_index(bs, i + 1)] >> 4), pvq_encoder.c:874."""
        q0 = max(1, self.quantizer)
        return [max(1, (q0 * int(self.pvq_qm_q4[pli][self.qm_index[bs][i + 1]])) >> 4)
                for i in range(NBANDS[bs])]

    def beta_band(self, pli, bs, masking=1):
        """OD_PVQ_BETA[use_masking][pli][bs], src/pvq.c:243-268."""
        return [int(self.beta[masking][pli][bs][i]) for i in range(NBANDS[bs])]

    def qm_slices(self, pli, bs):
        """(qm, qm_inv) in coding order for this block size / decimation
        (state.qm + od_qm_offset(bs, xdec), src/encode.c:1355-1358)."""
        off = int(self.qm_offset[bs][1 if pli else 0])
        n = 4 << bs
        ln = min(n * n, 512)
        return (np.ascontiguousarray(self.qm[off:off + ln]),
                np.ascontiguousarray(self.qm_inv[off:off + ln]))
