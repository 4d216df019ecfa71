"""Host-side quantiser set-up for the PVQ band stage.

The set-up itself lives in the C library (daala_amd/csrc/quant.hip: odhip_init_qm =
od_init_qm src/pvq.c:322-381, odhip_interp_qm = od_interp_qm src/encode.c:2903-2940
as selected at :3052-3072, the per-band step of src/pvq_encoder.c:874, OD_PVQ_BETA);
this module is its ctypes mirror.  The kernels take the resulting tables as plain
data.

The two quantisers an encoder derives from `-v quality` come from the reference's
rate-control code (src/rate.c:717-725, :790-840), which is outside this path: the
integrating encoder passes enc->rc.base_quantizer and enc->state.quantizer;
QUALITY_QUANTIZERS lists the pairs the reference produces for a few qualities
(keyframes, rate control off) for tests and bench.py.
"""
import ctypes

import numpy as np

from .api import lib, _check

OD_PVQ_LAMBDA = 0.147  # src/pvq.h:49; enc->pvq_norm_lambda, src/rate.c:1077
NBANDS = [1, 4, 7, 9, 9]
QM_SIZE = 30
QM_BUFFER_SIZE = 10912
# quality (encoder_example -v) -> (rc.base_quantizer, state.quantizer) of a keyframe
QUALITY_QUANTIZERS = {0: (0, 0), 1: (16, 12), 5: (80, 65), 10: (160, 125), 20: (320, 243),
                      40: (640, 525), 100: (1600, 1264), 511: (8176, 6574)}


class _Quant(ctypes.Structure):
    _fields_ = [("quantizer", ctypes.c_int), ("base_quantizer", ctypes.c_int),
                ("use_masking", ctypes.c_int), ("hvs_qm", ctypes.c_int),
                ("pvq_qm_q4", (ctypes.c_uint8 * QM_SIZE) * 3),
                ("qm", ctypes.c_int16 * QM_BUFFER_SIZE),
                ("qm_inv", ctypes.c_int16 * QM_BUFFER_SIZE)]


class QuantTables:
    """odhip_quant + the per-band helpers, as numpy views."""

    def __init__(self, base_quantizer, quantizer, use_masking=1, hvs_qm=1):
        self.c = _Quant()
        _check(lib().odhip_quant_setup(ctypes.byref(self.c), int(base_quantizer), int(quantizer),
                                       int(use_masking), int(hvs_qm)), "odhip_quant_setup")
        self.quantizer = int(quantizer)
        self.base_quantizer = int(base_quantizer)
        self.use_masking = int(bool(use_masking))
        self.pvq_qm_q4 = np.ctypeslib.as_array(self.c.pvq_qm_q4).reshape(3, QM_SIZE)
        self.qm = np.ctypeslib.as_array(self.c.qm)
        self.qm_inv = np.ctypeslib.as_array(self.c.qm_inv)
        self.qm_offset = np.array([[lib().odhip_qm_offset(bs, d) for d in range(2)]
                                   for bs in range(5)], np.int32)

    @classmethod
    def for_quality(cls, quality, use_masking=1, hvs_qm=1):
        bq, q = QUALITY_QUANTIZERS[quality]
        return cls(bq, q, use_masking, hvs_qm)

    @classmethod
    def load(cls):
        """encoder_example's -v 20, the quality of BASELINE configs[1]."""
        return cls.for_quality(20)

    def _bands(self, pli, bs):
        q = (ctypes.c_int32 * 12)()
        b = (ctypes.c_int32 * 12)()
        nb = lib().odhip_quant_bands(ctypes.byref(self.c), int(pli), int(bs), q, b)
        if nb < 0:
            _check(nb, "odhip_quant_bands")
        return [q[i] for i in range(nb)], [b[i] for i in range(nb)]

    def q_band(self, pli, bs):
        """max(1, q0*pvq_qm_q4[od_qm_get_index(bs, i + 1)] >> 4), pvq_encoder.c:874."""
        return self._bands(pli, bs)[0]

    def beta_band(self, pli, bs, masking=None):
        """OD_PVQ_BETA[use_masking][pli][bs], src/pvq.c:243-268 (masking: the table
        set's own setting unless given)."""
        if masking is None or int(bool(masking)) == self.use_masking:
            return self._bands(pli, bs)[1]
        other = QuantTables(self.base_quantizer, self.quantizer, masking, self.c.hvs_qm)
        return other._bands(pli, bs)[1]

    def qm_slices(self, pli, bs):
        """(qm, qm_inv) in coding order for this block size / decimation
        (state.qm + od_qm_offset(bs, xdec), src/encode.c:1355-1358)."""
        off = int(self.qm_offset[bs][1 if pli else 0])
        n = 4 << bs
        ln = min(n * n, 512)
        return (np.ascontiguousarray(self.qm[off:off + ln]),
                np.ascontiguousarray(self.qm_inv[off:off + ln]))
