/* oracle/ref_dist_shim.c - TEST INFRASTRUCTURE.  od_compute_dist is file-static in the
   reference's src/encode.c (:1202); it is reached here by textual inclusion of that
   translation unit (the pattern of src/tests/test_coef_coder.c:25-34) into a SEPARATE
   shared object, oracle/_ref/libdaalaref_dist.so, linked -Bsymbolic against
   libdaalaref.so so that its private copy of encode.c's functions never interposes the
   real library.  Only the wrapper below is called. */
#include <stdlib.h>
#include <string.h>
#define static
#include "encode.c"
#undef static

__attribute__((visibility("default"))) double ref_compute_dist(const od_coeff *x, const od_coeff *y,
 int n, int flat_qm, int use_masking, int coded_quantizer) {
  static daala_enc_ctx *enc;
  od_coeff xb[OD_BSIZE_MAX*OD_BSIZE_MAX];
  od_coeff yb[OD_BSIZE_MAX*OD_BSIZE_MAX];
  if (!enc) enc = (daala_enc_ctx *)calloc(1, sizeof(*enc));
  enc->qm = flat_qm ? OD_FLAT_QM : OD_HVS_QM;
  enc->use_activity_masking = use_masking;
  enc->state.coded_quantizer = coded_quantizer;
  memcpy(xb, x, sizeof(*x)*n*n);
  memcpy(yb, y, sizeof(*y)*n*n);
  return od_compute_dist(enc, xb, yb, n);
}
