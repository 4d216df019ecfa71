/* tests/interpose/interpose.c - TEST INFRASTRUCTURE.

   Load-time replacement of the reference's lapping-filter drivers and of its
   PVQ search by libdaalahip, WITHOUT touching the reference: this library
   defines the reference's own symbol names (src/filter.h:80-87,
   src/pvq_encoder.c:93) and forwards them to the *_hip entry points.  Loaded
   with RTLD_GLOBAL before oracle/_ref/libdaalaref.so, the dynamic linker binds
   every call inside the unmodified reference encoder (src/encode.c:1489,1760,
   1789,2571,2675; src/pvq_encoder.c:542,589) to these definitions - the link-time
   override INTEGRATION.md sections 2 and 3 describe.  Counters prove the calls really went
   through. */
#include <stdint.h>
#include "../../include/daala_hip.h"

long odhip_interposed_calls[5];

void od_prefilter_split(od_coeff *c0, int stride, int bs, int f, int hfilter, int vfilter) {
  odhip_interposed_calls[0]++;
  od_prefilter_split_hip(c0, stride, bs, f, hfilter, vfilter);
}

void od_postfilter_split(od_coeff *c0, int stride, int bs, int f, int q, unsigned char *skip,
 int skip_stride, int hfilter, int vfilter) {
  odhip_interposed_calls[1]++;
  od_postfilter_split_hip(c0, stride, bs, f, q, skip, skip_stride, hfilter, vfilter);
}

void od_apply_prefilter_frame_sbs(od_coeff *c, int stride, int nhsb, int nvsb, int xdec,
 int ydec) {
  odhip_interposed_calls[2]++;
  od_apply_prefilter_frame_sbs_hip(c, stride, nhsb, nvsb, xdec, ydec);
}

void od_apply_postfilter_frame_sbs(od_coeff *c, int stride, int nhsb, int nvsb, int xdec,
 int ydec, int q, unsigned char *skip, int skip_stride) {
  odhip_interposed_calls[3]++;
  od_apply_postfilter_frame_sbs_hip(c, stride, nhsb, nvsb, xdec, ydec, q, skip, skip_stride);
}

double pvq_search_rdo_double(const int16_t *xcoeff, int n, int k, od_coeff *ypulse, double g2,
 double pvq_norm_lambda, int prev_k) {
  odhip_interposed_calls[4]++;
  return od_pvq_search_rdo_double_hip(xcoeff, n, k, ypulse, g2, pvq_norm_lambda, prev_k);
}
