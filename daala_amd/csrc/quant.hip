/* quant.hip - quantiser set-up of the PVQ band stages (host code; SURVEY.md 8(a) row
   a17).  In the reference this is per-encoder / per-frame host initialisation:

     od_init_qm          src/pvq.c:322-381     state.qm / state.qm_inv
     od_qm_offset        src/pvq.c:306-309
     od_qm_get_index     src/pvq.c:408-413
     od_interp_qm        src/encode.c:2903-2940, selected per plane at :3052-3072
                                               state.pvq_qm_q4[pli][OD_QM_SIZE]
     per-band step       src/pvq_encoder.c:874  q = max(1, q0*pvq_qm_q4[idx] >> 4)
     OD_PVQ_BETA         src/pvq.c:243-268

   The kernels take the results as plain data.  The data tables come from
   gen/od_quant_tables.h (tools/make_quant_tables.py). */
#include <math.h>
#include <string.h>
#include "../../include/daala_hip.h"
#include "gen/od_scan_tables.h"
#include "gen/od_quant_tables.h"

namespace {

constexpr int kQmShift = 11;        /* OD_QM_SHIFT, src/pvq.h:60 (integer build)  */
constexpr int kQmScale = 1 << kQmShift;
constexpr int kQmScaleMax = 32767;  /* OD_QM_SCALE_MAX, src/pvq.h:66              */
constexpr int kQmInvScale = 1 << 12; /* OD_QM_INV_SCALE, src/pvq.h:67-68          */
constexpr int kCoeffShift = 4;      /* OD_COEFF_SHIFT, src/internal.h:124         */

inline int qm_block_offset(int bs) {
  /* OD_QM_OFFSET(bs), src/pvq.h:72: blocks of 16, 64, 256, ... entries in a row */
  return (((1 << 2*bs) - 1) << 4)/3;
}

}  // namespace

extern "C" int odhip_qm_offset(int bs, int xydec) {
  if (bs < 0 || bs > ODHIP_NBSIZES || (xydec != 0 && xydec != 1)) return ODHIP_EINVAL;
  return xydec*qm_block_offset(ODHIP_NBSIZES) + qm_block_offset(bs);
}

extern "C" int odhip_qm_get_index(int bs, int band) {
  if (bs < 0 || bs >= ODHIP_NBSIZES || band < 0) return ODHIP_EINVAL;
  /* horizontal and vertical bands of a level share one entry */
  return bs*(bs + 1) + band - band/3;
}

extern "C" void odhip_init_qm(int16_t *x, int16_t *x_inv, const int *qm) {
  /* raster work arrays of 8 KiB each: off the stack, one pair per thread */
  static thread_local int16_t ty[64*64];
  static thread_local int16_t ty_inv[64*64];
  memset(x, 0, sizeof(*x)*ODHIP_QM_BUFFER_SIZE);
  memset(x_inv, 0, sizeof(*x_inv)*ODHIP_QM_BUFFER_SIZE);
  for (int bs = 0; bs < ODHIP_NBSIZES; bs++) {
    const int n = 4 << bs;
    for (int xydec = 0; xydec < 2; xydec++) {
      const double *mags = OD_QT_BASIS_MAG[xydec][bs];
      for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
          int32_t mag = (int32_t)floor(.5 + kQmScale*mags[i]*mags[j]);
          if (i == 0 && j == 0) mag = kQmScale;
          else {
            const int qmv = qm[(i << 1 >> bs)*8 + (j << 1 >> bs)];
            mag *= 16;
            mag = (mag + (qmv >> 1))/qmv;
          }
          const int16_t v = (int16_t)(mag < kQmScaleMax ? mag : kQmScaleMax);
          ty[i*n + j] = v;
          ty_inv[i*n + j] = (int16_t)((kQmScale*kQmInvScale + (v >> 1))/v);
        }
      }
      /* od_raster_to_coding_order_16: the composite scan; blocks above 16x16
         have coding positions for their lowest 512 coefficients only
         (src/partition.c:35-49) */
      const int off = odhip_qm_offset(bs, xydec);
      const int len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
      for (int c = 0; c < len; c++) {
        const int pos = OD_SCAN_XY[c][1]*n + OD_SCAN_XY[c][0];
        x[off + c] = ty[pos];
        x_inv[off + c] = ty_inv[pos];
      }
    }
  }
}

extern "C" int odhip_interp_qm(uint8_t out[ODHIP_QM_SIZE], int base_quantizer, int use_masking,
 int pli) {
  if (!out || pli < 0 || pli > 2 || base_quantizer < 0) return ODHIP_EINVAL;
  const int m = use_masking != 0;
  const unsigned char *qm_q4 = pli == 0 ? OD_QT_LUMA_QM_Q4[m] : OD_QT_CHROMA_QM_Q4[m];
  const int q = base_quantizer;
  const int q1 = OD_QT_DEFAULT_QMS[0][pli][0] << kCoeffShift;
  const int s1 = OD_QT_DEFAULT_QMS[0][pli][1];
  const int q2 = OD_QT_DEFAULT_QMS[1][pli][0] << kCoeffShift;
  const int s2 = OD_QT_DEFAULT_QMS[1][pli][1];
  /* two interpolation points per plane (the third table entry is the terminator):
     below the first and above the second the matrix of that point is used */
  if (q <= q1 || q > q2) {
    const int s = q <= q1 ? s1 : s2;
    for (int i = 0; i < ODHIP_QM_SIZE; i++) {
      const int v = qm_q4[i]*s >> 8;
      out[i] = (uint8_t)(v < 255 ? v : 255);
    }
    return ODHIP_SUCCESS;
  }
  /* linear in log(q) against log(matrix*scale); libm log / exp as in the
     reference - the result is rounded to 8 bits */
  const double x = (log(q) - log(q1))/(log(q2) - log(q1));
  for (int i = 0; i < ODHIP_QM_SIZE; i++) {
    const int v = (int)floor(.5 + (1./256)*exp(x*log(qm_q4[i]*s2) + (1 - x)*log(qm_q4[i]*s1)));
    out[i] = (uint8_t)(v < 255 ? v : 255);
  }
  return ODHIP_SUCCESS;
}

extern "C" int odhip_quant_setup(odhip_quant *qt, int base_quantizer, int quantizer, int use_masking,
 int hvs_qm) {
  if (!qt || base_quantizer < 0 || quantizer < 0) return ODHIP_EINVAL;
  memset(qt, 0, sizeof(*qt));
  qt->quantizer = quantizer;
  qt->base_quantizer = base_quantizer;
  qt->use_masking = use_masking != 0;
  qt->hvs_qm = hvs_qm != 0;
  odhip_init_qm(qt->qm, qt->qm_inv, hvs_qm ? OD_QT_QM8_Q4_HVS : OD_QT_QM8_Q4_FLAT);
  for (int pli = 0; pli < 3; pli++) {
    const int rc = odhip_interp_qm(qt->pvq_qm_q4[pli], base_quantizer, use_masking, pli);
    if (rc) return rc;
  }
  return ODHIP_SUCCESS;
}

extern "C" int odhip_quant_bands(const odhip_quant *qt, int pli, int bs, int32_t *q_band,
 int32_t *beta_band) {
  if (!qt || pli < 0 || pli > 2 || bs < 0 || bs >= ODHIP_NBSIZES) return ODHIP_EINVAL;
  const int nb = OD_NBANDS[bs];
  /* od_block_encode passes quant = max(1, state.quantizer) as q0 (src/encode.c:1336) */
  const int q0 = qt->quantizer > 1 ? qt->quantizer : 1;
  for (int i = 0; i < nb; i++) {
    if (q_band) {
      const int q = q0*qt->pvq_qm_q4[pli][odhip_qm_get_index(bs, i + 1)] >> 4;
      q_band[i] = q > 1 ? q : 1;
    }
    if (beta_band) beta_band[i] = OD_QT_PVQ_BETA[qt->use_masking][pli][bs][i];
  }
  return nb;
}
