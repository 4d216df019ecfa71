/* pvq_search.cuh - pvq_search_rdo_double (reference src/pvq_encoder.c:93-224)
   with one band per lane, |x| and y in LDS, for an arbitrary band size n: the
   raw batched search (pvq_kernels.hip: odhip_pvq_search_batch and the per-call
   od_pvq_search_rdo_double_hip).  The PVQ band stage (pvq_bands.hip) uses the
   engineered form of the same mapping in pvq_lane.cuh.

   Every floating-point operation is a single IEEE-754 binary64 operation in
   the reference's order; translation units including this header MUST be
   compiled with -ffp-contract=off.  sqrt and division use the correctly
   rounded forms. */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int kWave = 64;

/* od_rsqrt_table, src/pvq_encoder.c:52-60: 6-digit decimal literals for
   i <= 16, 1/sqrt(i) beyond. */
__device__ const double kRsqrtTable[16] = {
  1.000000, 0.707107, 0.577350, 0.500000,
  0.447214, 0.408248, 0.377964, 0.353553,
  0.333333, 0.316228, 0.301511, 0.288675,
  0.277350, 0.267261, 0.258199, 0.250000};

/* The table is read with a different index in every lane: it is staged in LDS
   (a global or constant load with 64 addresses would serialise or miss).
   Every kernel that searches calls od_rsqrt_init() once.  OD_RSQ_TABLE_N
   (default 16) may be raised by the including translation unit: entries beyond
   16 are 1/sqrt(i) with the correctly rounded sqrt and division - the values
   od_rsqrt_table computes on the fly - read from a table filled once per
   process (od_rsqrt_fill_launch).  The last pulses of a search evaluate four
   entries per pulse plus one per candidate holding four or more pulses; without
   the table each is an fp64 sqrt and an fp64 division. */
#ifndef OD_RSQ_TABLE_N
# define OD_RSQ_TABLE_N 16
#endif
__shared__ double od_rsq_lds[OD_RSQ_TABLE_N];

#if OD_RSQ_TABLE_N > 16
__device__ double gRsqBig[OD_RSQ_TABLE_N];

__global__ void k_rsq_big_fill(void) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < OD_RSQ_TABLE_N) {
    gRsqBig[i] = i < 16 ? kRsqrtTable[i] : __ddiv_rn(1., __dsqrt_rn((double)(i + 1)));
  }
}

static inline void od_rsqrt_fill_launch(void) {
  k_rsq_big_fill<<<(OD_RSQ_TABLE_N + 255)/256, 256, 0, 0>>>();
}

/* (every thread of the workgroup calls it, before any of them leaves: one table per workgroup) */
__device__ __forceinline__ void od_rsqrt_init(int tid) {
  for (int i = tid; i < OD_RSQ_TABLE_N; i += (int)blockDim.x) od_rsq_lds[i] = gRsqBig[i];
  __syncthreads();
}
#else
__device__ __forceinline__ void od_rsqrt_init(int tid) {
  if (tid < 16) od_rsq_lds[tid] = kRsqrtTable[tid];
  __syncthreads();
}
#endif

/* Beyond the LDS table (bands that place tens of pulses: fine quantisers, band 0 of the large blocks): the same
   correctly rounded values from a table in global memory (512 KB, cache resident; filled once per device by the
   expression it replaces), an fp64 square root and division - ~40 instructions - only beyond THAT (round 5:
   `quality_sweep` at -v 5). */
#ifdef OD_RSQ_HUGE      /* defined by the translation units that fill the table in their per-device set-up */
constexpr int kRsqHugeN = 65536;
__device__ double gRsqHuge[kRsqHugeN];

__global__ void k_rsq_huge_fill(void) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < kRsqHugeN) gRsqHuge[i] = i < 16 ? kRsqrtTable[i] : __ddiv_rn(1., __dsqrt_rn((double)(i + 1)));
}

static inline void od_rsqrt_huge_fill_launch(void) {
  k_rsq_huge_fill<<<(kRsqHugeN + 255)/256, 256, 0, 0>>>();
}

__device__ __forceinline__ double od_rsqrt_beyond(int i) {
  if (i <= kRsqHugeN) return gRsqHuge[i - 1];
  return __ddiv_rn(1., __dsqrt_rn((double)i));
}
#else
__device__ __forceinline__ double od_rsqrt_beyond(int i) {
  return __ddiv_rn(1., __dsqrt_rn((double)i));
}
#endif

__device__ __forceinline__ double od_rsqrt_table(int i) {
  if (i <= OD_RSQ_TABLE_N) return od_rsq_lds[i - 1];
  return od_rsqrt_beyond(i);
}

#include "od_dpp.cuh"

/* One band per lane: the K-pulse search proper.  xs holds the band's SIGNED
   x (int16) and ys the pulse magnitudes, both laid out [j][64 lanes].  On
   entry with prev_k > 0 ys holds the magnitudes of the previous search.
   Returns the cosine distance; ys holds the new magnitudes (the caller
   restores signs, src/pvq_encoder.c:220-222). */
__device__ __forceinline__ double od_pvq_search_lane(const short *xs, unsigned short *ys,
 int lane, int n, int k, int prev_k, double g2, double pvq_norm_lambda, double *yy_out) {
  double xx = 0;
  double xy = 0;
  double yy = 0;
  /* x[j] = fabs((float)xcoeff[j]): exact for int16. */
  for (int j = 0; j < n; j++) {
    const double xj = (double)abs((int)xs[j*kWave + lane]);
    xx += xj*xj;
  }
  const double norm_1 = __ddiv_rn(1., __dsqrt_rn(1e-30 + xx));
  const double lambda = __ddiv_rn(pvq_norm_lambda, 1e-30 + g2);
  int i = 0;
  if (prev_k > 0 && prev_k <= k) {
    for (int j = 0; j < n; j++) {
      const int yj = ys[j*kWave + lane];
      const double xj = (double)abs((int)xs[j*kWave + lane]);
      xy += xj*yj;
      yy += (double)(yj*yj);
      i += yj;
    }
  }
  else if (k > 2) {
    double l1_norm = 0;
    for (int j = 0; j < n; j++) l1_norm += (double)abs((int)xs[j*kWave + lane]);
    const double l1_inv = __ddiv_rn(1., l1_norm > 1e-100 ? l1_norm : 1e-100);
    for (int j = 0; j < n; j++) {
      const double xj = (double)abs((int)xs[j*kWave + lane]);
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
      double tmp_xy = xy + (double)abs((int)xs[j*kWave + lane]);
      const double tmp_yy = yy + (double)(2*ys[j*kWave + lane]) + 1;
      tmp_xy = tmp_xy*tmp_xy;
      if (j == 0 || tmp_xy*best_yy > best_xy*tmp_yy) {
        best_xy = tmp_xy;
        best_yy = tmp_yy;
        pos = j;
      }
    }
    const int yp_ = ys[pos*kWave + lane];
    xy = xy + (double)abs((int)xs[pos*kWave + lane]);
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
      double tmp_xy = xy + (double)abs((int)xs[j*kWave + lane]);
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
    xy = xy + (double)abs((int)xs[pos*kWave + lane]);
    yy = yy + (double)(2*yp_) + 1;
    ys[pos*kWave + lane] = (unsigned short)(yp_ + 1);
  }
  *yy_out = yy;
  return __ddiv_rn(xy, 1e-100 + __dsqrt_rn(xx*yy));
}

}  // namespace
