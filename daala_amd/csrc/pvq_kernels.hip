/* pvq_kernels.hip - PVQ K-pulse search on gfx950 (fp64, bit-exact).

   Restates pvq_search_rdo_double (reference src/pvq_encoder.c:93-224).

   Mapping: ONE BAND PER LANE, 64 bands per wavefront.  The greedy argmax at
   src/pvq_encoder.c:172-183 compares cross-multiplied ROUNDED doubles
   (tmp_xy^2*best_yy > best_xy*tmp_yy, first index wins ties), which is not
   guaranteed transitive, so a tree reduction across lanes could pick a
   different pulse position than the reference's left-to-right scan.  A lane
   that owns its whole band simply runs the same scan in the same order:
   identical results by construction, no cross-lane traffic, and the wave's 64
   lanes are 64 independent bands.  |x| and y live in LDS as 16-bit values,
   laid out [j][lane] so every access of a wave is one contiguous 128-byte
   row (bank-conflict free).

   Every floating-point operation is a single IEEE-754 binary64 operation in
   the reference's order; this file MUST be compiled with -ffp-contract=off
   (hipcc contracts a*b+c into FMA by default, gcc -O2 on x86-64 does not).
   sqrt and division use the correctly rounded forms. */
#include "../../include/daala_hip.h"
#include "od_common.cuh"
#include "pvq_search.cuh"

namespace {

__global__ __launch_bounds__(kWave) void k_pvq_search(const int16_t *x_in, int n,
 const int32_t *k_in, od_coeff *y_io, const double *g2_in, double pvq_norm_lambda,
 const int32_t *prev_k_in, double *cos_out, long nbands) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  short *xs = (short *)lds;              /* x  [n][64] */
  unsigned short *ys = lds + n*kWave;    /* |y| [n][64] */
  od_rsqrt_init(threadIdx.x);
  const int lane = threadIdx.x;
  const long band = (long)blockIdx.x*kWave + lane;
  const bool live = band < nbands;
  const long b = live ? band : nbands - 1;
  const int16_t *xc = x_in + b*n;
  od_coeff *yp = y_io + b*n;
  /* pulse magnitudes live in 16 bits: a K outside 0..65535 is reported (y = 0,
     cos = NaN), never searched */
  const bool k_ok = k_in[b] >= 0 && k_in[b] <= 65535;
  const int k = k_ok ? k_in[b] : 0;
  const int prev_k = prev_k_in && k_ok ? prev_k_in[b] : 0;
  for (int j = 0; j < n; j++) xs[j*kWave + lane] = xc[j];
  if (prev_k > 0 && prev_k <= k) {
    for (int j = 0; j < n; j++) {
      const int yj = yp[j];
      ys[j*kWave + lane] = (unsigned short)(yj < 0 ? -yj : yj);
    }
  }
  double yy;
  const double c = od_pvq_search_lane(xs, ys, lane, n, k, prev_k, g2_in[b], pvq_norm_lambda,
   &yy);
  if (live) {
    for (int j = 0; j < n; j++) {
      const int yj = ys[j*kWave + lane];
      yp[j] = xc[j] < 0 ? -yj : yj;
    }
    cos_out[band] = k_ok ? c : __longlong_as_double(0x7ff8000000000000LL);
  }
}

}  // namespace


extern "C" int odhip_pvq_search_batch(const int16_t *d_x, int n, const int32_t *d_k,
 od_coeff *d_y, const double *d_g2, double pvq_norm_lambda,
 const int32_t *d_prev_k, double *d_cos, long nbands, odhip_stream stream) {
  if (!d_x || !d_k || !d_y || !d_g2 || !d_cos || n < 1 || n > 128) return ODHIP_EINVAL;
  if (nbands <= 0) return ODHIP_SUCCESS;
  const long grid = (nbands + kWave - 1)/kWave;
  if (grid > 0x7fffffffL) return ODHIP_EINVAL;
  const size_t lds = (size_t)2*n*kWave*sizeof(unsigned short);
  k_pvq_search<<<(unsigned)grid, kWave, lds, (hipStream_t)stream>>>(d_x, n, d_k, d_y,
   d_g2, pvq_norm_lambda, d_prev_k, d_cos, nbands);
  return odhip_check_launch();
}

