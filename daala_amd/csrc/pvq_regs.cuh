/* pvq_regs.cuh - pvq_search_rdo_double (reference src/pvq_encoder.c:93-224), one
   band per lane with the band in REGISTERS: the form for the short bands
   (N = 8, 15) of the with-reference stage, where a lane runs a chain of up to 14
   searches on the same vector.  Every loop over the band is fully unrolled
   (static register indices); each lane scans its candidates j = 0..n-1 in order
   with the reference's comparator, so the result is the reference's by
   construction.

   n == N or N - 1: the n - 1 reflected coefficients of a with-reference search
   occupy a vector of N with a PAD in the last position (|x| = 0, y = 0) whose
   candidate key can never win; n enters the rate term and selects the k == 1
   special cases (:154-163) as in the reference.

   Requires pvq_search.cuh (od_rsqrt_table) before it and -ffp-contract=off. */
#pragma once
#include <type_traits>
#include "od_sel.cuh"

namespace {

/* xx = sum x^2 and 1/sqrt(1e-30 + xx) (:107-109, :147): properties of the
   vector, computed once per chain of searches. */
template <int N>
__device__ __forceinline__ void od_regs_norm(const int (&ax)[N], double *xx_out, double *norm_1_out) {
  double xx = 0;
#pragma unroll
  for (int j = 0; j < N; j++) xx += (double)ax[j]*(double)ax[j];
  *xx_out = xx;
  *norm_1_out = __ddiv_rn(1., __dsqrt_rn(1e-30 + xx));
}

template <int N>
__device__ __forceinline__ double od_pvq_search_regs(const int (&ax)[N], int (&y)[N], int n, int k,
 int prev_k, double g2, double pvq_norm_lambda, double xx, double norm_1, double *cxy, double *cyy) {
  const bool padded = n != N;
  /* |x_j| is converted where it is used: a second, double copy of the band costs 2N
     VGPRs and an occupancy step */
#define OD_XD(j) od_cvt_u(ax[j])
  const double lambda = __ddiv_rn(pvq_norm_lambda, 1e-30 + g2);
  double xy = 0;
  double yy = 0;
  int i = 0;
  if (prev_k > 0 && prev_k <= k) {
    /* The chain's previous search (same vector, prev_k pulses) ended on exactly these sums - integers
       below 2^53, exact in any order - and left them in *cxy / *cyy: round 4 stopped recomputing
       them (:116-121 of the reference does, 8 instructions per position per search). */
    xy = *cxy;
    yy = *cyy;
    i = prev_k;
  }
  else if (k > 2) {
    double l1_norm = 0;
#pragma unroll
    for (int j = 0; j < N; j++) l1_norm += OD_XD(j);
    const double l1_inv = __ddiv_rn(1., l1_norm > 1e-100 ? l1_norm : 1e-100);
#pragma unroll
    for (int j = 0; j < N; j++) {
      const double tmp = (k*OD_XD(j))*l1_inv;
      int yj = (int)floor(tmp);
      yj = yj > 0 ? yj : 0;
      y[j] = yj;
      xy += OD_XD(j)*yj;
      yy += (double)(yj*yj);
      i += yj;
    }
  }
  else {
#pragma unroll
    for (int j = 0; j < N; j++) y[j] = 0;
  }
  const int rdo_pulses = 1 + k/4;
  /* n is N or N - 1: the quotients of :154-163 are compile-time constants (correctly rounded, as
     the divisions are), not a division sequence per search */
  double delta_rate = padded ? 3./(N - 1) : 3./N;
  double accel_rate = 0.;
  if (k == 1) {
    if (n == 15) {
      accel_rate = -8./15;
      delta_rate = 4.5/15 - accel_rate;
    }
    else if (n == 8) {
      accel_rate = 5.7/8;
      delta_rate = 9.3/8 - accel_rate;
    }
  }
  /* Greedy pulses, :165-187. */
  for (; i < k - rdo_pulses; i++) {
    int pos = 0;
    double best_xy = -10;
    double best_yy = 1;
    /* yy + 2*y_j + 1 (:175) is an integer below 2^31: formed in integers, converted once */
    const int yyp1 = (int)yy + 1;
#pragma unroll
    for (int j = 0; j < N; j++) {
      double tmp_xy = xy + OD_XD(j);
      const double tmp_yy = (double)(yyp1 + 2*y[j]);
      tmp_xy = tmp_xy*tmp_xy;
      if (j == N - 1 && padded) tmp_xy = -1;   /* PAD: loses every comparison */
      if (j == 0) {
        best_xy = tmp_xy;
        best_yy = tmp_yy;
      }
      else {
        const unsigned long long take = od_cmp_gt(tmp_xy*best_yy, best_xy*tmp_yy);
        best_xy = od_sel(take, tmp_xy, best_xy);
        best_yy = od_sel(take, tmp_yy, best_yy);
        pos = od_sel(take, j, pos);
      }
    }
    int xp = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      if (j == pos) {
        xp = ax[j];
        y[j]++;
      }
    }
    xy = xy + (double)xp;
    /* yy + 2*y[pos] + 1 is the winner's denominator, an exact integer either way: no need to read
       y[pos] back */
    yy = best_yy;
  }
  /* Last pulses with the rate term, :192-219. */
  /* rate penalty of each candidate, (lambda*j)*(delta_rate + j*accel_rate):
     constant over the pulses of a search */
  double pen[N];
  if (k == 1) {
#pragma unroll
    for (int j = 0; j < N; j++) pen[j] = (lambda*j)*(delta_rate + j*accel_rate);
  }
  else {
    /* accel_rate == 0: delta_rate + j*0. is delta_rate */
#pragma unroll
    for (int j = 0; j < N; j++) pen[j] = (lambda*j)*delta_rate;
  }
  /* (2*t)*norm_1 == t*(2*norm_1): scaling by two is exact */
  const double norm2 = 2*norm_1;
  for (; i < k; i++) {
    /* od_rsqrt_table(yy + 2*y_j + 1) for every candidate straight from the LDS
       table (the reference's four-entry cache :199-200 holds the same values).  No y_j
       exceeds the i pulses placed so far: when yy + 2*i + 1 is inside the table - every
       pulse of every band of the frames measured - the lookups are unconditional; round 2
       tested each of them against the table size, a saveexec / branch pair and an
       if-converted sqrt + division per candidate position. */
    const int yyi = (int)yy;
    int pos = 0;
    double best_cost = -1e5;
    auto scan = [&](auto fast) {
#pragma unroll
      for (int j = 0; j < N; j++) {
        double tmp_xy = xy + OD_XD(j);
        const int idx = yyi + 2*y[j] + 1;
        const double tmp_yy = decltype(fast)::value ? od_rsq_lds[idx - 1] : od_rsqrt_table(idx);
        tmp_xy = (tmp_xy*norm2)*tmp_yy - pen[j];
        if (j == N - 1 && padded) tmp_xy = -1.7976931348623157e308;   /* PAD */
        if (j == 0) best_cost = tmp_xy;
        else {
          const unsigned long long take = od_cmp_gt(tmp_xy, best_cost);
          best_cost = od_sel(take, tmp_xy, best_cost);
          pos = od_sel(take, j, pos);
        }
      }
    };
    if (yyi + 2*i + 1 <= OD_RSQ_TABLE_N) scan(std::true_type());
    else scan(std::false_type());
    int xp = 0;
    int yp = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      if (j == pos) {
        xp = ax[j];
        yp = y[j];
        y[j] = yp + 1;
      }
    }
    xy = xy + (double)xp;
    yy = yy + (double)(2*yp) + 1;
  }
#undef OD_XD
  *cxy = xy;
  *cyy = yy;
  return __ddiv_rn(xy, 1e-100 + __dsqrt_rn(xx*yy));
}

}  // namespace
