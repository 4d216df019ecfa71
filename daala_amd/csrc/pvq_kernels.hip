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

namespace {

constexpr int kWave = 64;

/* od_rsqrt_table, src/pvq_encoder.c:52-60: 6-digit decimal literals for
   i <= 16, 1/sqrt(i) beyond. */
__device__ const double kRsqrtTable[16] = {
  1.000000, 0.707107, 0.577350, 0.500000,
  0.447214, 0.408248, 0.377964, 0.353553,
  0.333333, 0.316228, 0.301511, 0.288675,
  0.277350, 0.267261, 0.258199, 0.250000};

__device__ __forceinline__ double od_rsqrt_table(int i) {
  if (i <= 16) return kRsqrtTable[i - 1];
  return __ddiv_rn(1., __dsqrt_rn((double)i));
}

__global__ __launch_bounds__(kWave) void k_pvq_search(const int16_t *x_in, int n,
 const int32_t *k_in, od_coeff *y_io, const double *g2_in, double pvq_norm_lambda,
 const int32_t *prev_k_in, double *cos_out, long nbands) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short *xs = lds;              /* |x|  [n][64] */
  unsigned short *ys = lds + n*kWave;    /* y    [n][64] */
  const int lane = threadIdx.x;
  const long band = (long)blockIdx.x*kWave + lane;
  const bool live = band < nbands;
  const long b = live ? band : nbands - 1;
  const int16_t *xc = x_in + b*n;
  od_coeff *yp = y_io + b*n;
  const int k = k_in[b];
  const int prev_k = prev_k_in ? prev_k_in[b] : 0;
  const double g2 = g2_in[b];
  double xx = 0;
  double xy = 0;
  double yy = 0;
  /* x[j] = fabs((float)xcoeff[j]): exact for int16. */
  for (int j = 0; j < n; j++) {
    const int v = xc[j];
    const int a = v < 0 ? -v : v;
    xs[j*kWave + lane] = (unsigned short)a;
    const double xj = (double)a;
    xx += xj*xj;
  }
  const double norm_1 = __ddiv_rn(1., __dsqrt_rn(1e-30 + xx));
  const double lambda = __ddiv_rn(pvq_norm_lambda, 1e-30 + g2);
  int i = 0;
  if (prev_k > 0 && prev_k <= k) {
    for (int j = 0; j < n; j++) {
      int yj = yp[j];
      yj = yj < 0 ? -yj : yj;
      ys[j*kWave + lane] = (unsigned short)yj;
      const double xj = (double)xs[j*kWave + lane];
      xy += xj*yj;
      yy += (double)(yj*yj);
      i += yj;
    }
  }
  else if (k > 2) {
    double l1_norm = 0;
    for (int j = 0; j < n; j++) l1_norm += (double)xs[j*kWave + lane];
    const double l1_inv = __ddiv_rn(1., l1_norm > 1e-100 ? l1_norm : 1e-100);
    for (int j = 0; j < n; j++) {
      const double xj = (double)xs[j*kWave + lane];
      const double tmp = (k*xj)*l1_inv;
      int yj = (int)floor(tmp);
      yj = yj > 0 ? yj : 0;
      ys[j*kWave + lane] = (unsigned short)yj;
      xy += xj*yj;
      yy += (double)(yj*yj);
      i += yj;
    }
  }
  else {
    for (int j = 0; j < n; j++) ys[j*kWave + lane] = 0;
  }
  const int rdo_pulses = 1 + k/4;
  double delta_rate = __ddiv_rn(3., (double)n);
  double accel_rate = 0.;
  if (k == 1) {
    if (n == 15) {
      accel_rate = __ddiv_rn(-8., (double)n);
      delta_rate = __ddiv_rn(4.5, (double)n) - accel_rate;
    }
    else if (n == 8) {
      accel_rate = __ddiv_rn(5.7, (double)n);
      delta_rate = __ddiv_rn(9.3, (double)n) - accel_rate;
    }
  }
  /* Greedy pulses, src/pvq_encoder.c:165-187. */
  for (; i < k - rdo_pulses; i++) {
    int pos = 0;
    double best_xy = -10;
    double best_yy = 1;
    for (int j = 0; j < n; j++) {
      double tmp_xy = xy + (double)xs[j*kWave + lane];
      const double tmp_yy = yy + (double)(2*ys[j*kWave + lane]) + 1;
      tmp_xy = tmp_xy*tmp_xy;
      if (j == 0 || tmp_xy*best_yy > best_xy*tmp_yy) {
        best_xy = tmp_xy;
        best_yy = tmp_yy;
        pos = j;
      }
    }
    const int yp_ = ys[pos*kWave + lane];
    xy = xy + (double)xs[pos*kWave + lane];
    yy = yy + (double)(2*yp_) + 1;
    ys[pos*kWave + lane] = (unsigned short)(yp_ + 1);
  }
  /* Last pulses with the rate term, src/pvq_encoder.c:192-219. */
  for (; i < k; i++) {
    double rsqrt_tab[4];
    for (int j = 0; j < 4; j++) rsqrt_tab[j] = od_rsqrt_table((int)(yy + 2*j + 1));
    int pos = 0;
    double best_cost = -1e5;
    for (int j = 0; j < n; j++) {
      double tmp_xy = xy + (double)xs[j*kWave + lane];
      const int yj = ys[j*kWave + lane];
      double tmp_yy;
      if (yj < 4) {
        tmp_yy = yj == 0 ? rsqrt_tab[0] : yj == 1 ? rsqrt_tab[1]
         : yj == 2 ? rsqrt_tab[2] : rsqrt_tab[3];
      }
      else tmp_yy = od_rsqrt_table((int)(yy + (double)(2*yj) + 1));
      tmp_xy = ((2*tmp_xy)*norm_1)*tmp_yy
       - (lambda*j)*(delta_rate + j*accel_rate);
      if (j == 0 || tmp_xy > best_cost) {
        best_cost = tmp_xy;
        pos = j;
      }
    }
    const int yp_ = ys[pos*kWave + lane];
    xy = xy + (double)xs[pos*kWave + lane];
    yy = yy + (double)(2*yp_) + 1;
    ys[pos*kWave + lane] = (unsigned short)(yp_ + 1);
  }
  if (live) {
    for (int j = 0; j < n; j++) {
      const int yj = ys[j*kWave + lane];
      yp[j] = xc[j] < 0 ? -yj : yj;
    }
    cos_out[band] = __ddiv_rn(xy, 1e-100 + __dsqrt_rn(xx*yy));
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
