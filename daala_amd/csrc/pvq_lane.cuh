/* pvq_lane.cuh - pvq_search_rdo_double (reference src/pvq_encoder.c:93-224) with
   ONE BAND PER LANE, the form the sorted band stage (pvq_bands.hip) uses for
   every band size.

   This is the literal algorithm - each lane scans its own candidates j = 0..n-1
   in order with the reference's comparator - so there is nothing to verify or
   replay; what is engineered is the cost per candidate:

     * |x| (16 bits) and twice the pulse count, 2*y (16 bits), of a coefficient
       share one LDS word, laid out [j][64]: lane l only ever touches column l
       (bank l % 32: conflict-free).  Signs are not kept: they only matter
       when the pulses are written out, and x16 is re-read there.
     * sums of integers (xx, L1, and xy/yy/i of the projection) are accumulated
       in integer registers and converted once: every partial sum is below 2^53,
       so the double sums of the reference are exact and order-independent.
     * od_rsqrt_table(i) (src/pvq_encoder.c:52-60) is a 512-entry LDS table
       (6-digit literals for i <= 16, correctly rounded 1/sqrt(i) above); a
       pulse whose index could exceed the table takes the computed path.
     * b = yy + 2*y + 1 is formed as (yy + 1) + 2*y and (2*t)*norm_1 as
       t*(2*norm_1): integers below 2^53 and exact scalings by two, so the
       rounded results are the reference's.

   Translation units including this header are compiled with -ffp-contract=off;
   sqrt and division are the correctly rounded forms. */
#pragma once

namespace {

constexpr int kPitch = 64;   /* LDS row pitch in 32-bit words: column `lane`, bank lane % 32 */
constexpr int kRsqN = 512;   /* entries of the LDS 1/sqrt table */

/* od_rsqrt_table(i) for i = 1..kRsqN, filled once per process by k_rsq_fill. */
__device__ double gRsqTable[kRsqN];

__global__ void k_rsq_fill(void) {
  const int i = threadIdx.x;
  if (i < kRsqN) {
    gRsqTable[i] = i < 16 ? kRsqrtTable[i] : __ddiv_rn(1., __dsqrt_rn((double)(i + 1)));
  }
}

/* Selects through an SGPR-pair mask (VOP3 encodings).  Measured on gfx950
   (tools/ubench/fp64_rate.hip): back-to-back VOP2 `v_cndmask_b32 ..., vcc` -
   what the compiler emits for `c ? a : b` on doubles, two per select - issue at
   ~16-19 cycles each instead of ~4.4; the e64 forms do not.  The search loops
   are five selects per candidate, so they are spelled out. */
__device__ __forceinline__ unsigned long long od_cmp_gt(double a, double b) {
  unsigned long long m;
  asm("v_cmp_gt_f64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  return m;
}

__device__ __forceinline__ int od_sel(unsigned long long m, int t, int f) {
  int d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(f), "v"(t), "s"(m));
  return d;
}

__device__ __forceinline__ double od_sel(unsigned long long m, double t, double f) {
  const int lo = od_sel(m, __double2loint(t), __double2loint(f));
  const int hi = od_sel(m, __double2hiint(t), __double2hiint(f));
  return __hiloint2double(hi, lo);
}

struct LaneSearch {
  double xx;
  double norm2;   /* 2*norm_1 */
  double l1_inv;
  unsigned xy;    /* sum |x|*y   (< 2^30 for K <= kMaxK) */
  unsigned yy;    /* sum y*y     (< 2^30) */
  int i;          /* pulses placed */
};

/* Largest pulse count the packed layout holds (2*y in 16 bits).  Far above
   anything 8..12-bit video reaches; the band stage reports a candidate with a
   larger K as flags = 2 instead of evaluating it. */
constexpr int kMaxK = 32767;

/* Per-band constants: xx, 2/sqrt(1e-30 + xx), 1/max(L1, 1e-100)
   (src/pvq_encoder.c:118-125,:139-141). */
template <int N>
__device__ __forceinline__ void od_lane_prepare(LaneSearch &s, const uint32_t *pk, int lane) {
  unsigned long long xx = 0;
  unsigned l1 = 0;
#pragma unroll 8
  for (int j = 0; j < N; j++) {
    const unsigned ax = pk[j*kPitch + lane] >> 16;
    xx += (unsigned long long)ax*ax;
    l1 += ax;
  }
  s.xx = (double)xx;
  s.norm2 = 2*__ddiv_rn(1., __dsqrt_rn(1e-30 + s.xx));
  const double l1d = (double)l1;
  s.l1_inv = __ddiv_rn(1., l1d > 1e-100 ? l1d : 1e-100);
  s.xy = 0;
  s.yy = 0;
  s.i = 0;
}

/* LDS reads of the search loops are software-pipelined by hand in groups of
   kGrp candidates: the words of group g+1 (and, for the rate pass, the table
   entries of group g+1 and the words of group g+2) are requested before group g
   is evaluated, and __builtin_amdgcn_sched_barrier keeps the compiler from
   sinking the requests back to their first use.  With 32 KiB of LDS per
   wavefront (n = 128) only one wavefront fits a SIMD, so nothing else hides
   the LDS latency (42% of wave cycles were s_waitcnt). */
constexpr int kGrp = 8;

template <int N>
__device__ __forceinline__ void od_lane_load_group(uint32_t (&w)[kGrp], const uint32_t *pk, int lane,
 int g) {
#pragma unroll
  for (int t = 0; t < kGrp; t++) {
    const int j = g*kGrp + t;
    w[t] = j < N ? pk[j*kPitch + lane] : 0;
  }
}

/* One pass of the rate-penalised argmax (src/pvq_encoder.c:196-213) over the
   lane's band: cost_j = 2*(xy + x_j)*norm_1*rsqrt(yy + 2*y_j + 1)
   - lambda*j*(delta_rate + j*accel_rate).  ACCEL = false drops the j*accel
   term (accel_rate is zero unless k == 1 and n is 8 or 15, where
   delta + j*0 == delta exactly). */
template <int N, bool ACCEL>
__device__ __forceinline__ int od_lane_rdo_scan(const uint32_t *pk, const double *rsq, int lane,
 unsigned xy, unsigned base, double norm2, double lambda, double delta_rate, double accel_rate) {
  if (N <= 16) {
    /* short bands: fully unrolled, every load is requested up front anyway */
    double best = 0;
    int pos = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const uint32_t w = pk[j*kPitch + lane];
      const double tt = (double)(xy + (w >> 16));
      const double r = rsq[base + (w & 0xffffu)];
      const double val = (tt*norm2)*r - (lambda*j)*(ACCEL ? delta_rate + j*accel_rate : delta_rate);
      if (j == 0) best = val;
      else {
        const unsigned long long m = od_cmp_gt(val, best);
        best = od_sel(m, val, best);
        pos = od_sel(m, j, pos);
      }
    }
    return pos;
  }
  constexpr int NG = (N + kGrp - 1)/kGrp;
  uint32_t w0[kGrp];
  uint32_t w1[kGrp];
  uint32_t w2[kGrp];
  double r0[kGrp];
  double r1[kGrp];
  od_lane_load_group<N>(w0, pk, lane, 0);
  od_lane_load_group<N>(w1, pk, lane, 1);
#pragma unroll
  for (int t = 0; t < kGrp; t++) r0[t] = rsq[base + (w0[t] & 0xffffu)];
  double best = 0;
  int pos = 0;
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (g + 2 < NG) od_lane_load_group<N>(w2, pk, lane, g + 2);
    if (g + 1 < NG) {
#pragma unroll
      for (int t = 0; t < kGrp; t++) r1[t] = rsq[base + (w1[t] & 0xffffu)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kGrp; t++) {
      const int j = g*kGrp + t;
      if (j < N) {
        const double tt = (double)(xy + (w0[t] >> 16));
        const double val = (tt*norm2)*r0[t] - (lambda*j)*(ACCEL ? delta_rate + j*accel_rate : delta_rate);
        if (j == 0) best = val;
        else {
          const unsigned long long m = od_cmp_gt(val, best);
          best = od_sel(m, val, best);
          pos = od_sel(m, j, pos);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kGrp; t++) {
      w0[t] = w1[t];
      w1[t] = w2[t];
      r0[t] = r1[t];
    }
  }
  return pos;
}

/* One pass of the greedy argmax (src/pvq_encoder.c:172-183): the candidate
   maximising (xy + x_j)^2/(yy + 2*y_j + 1), scanned upward with the
   reference's cross-multiplied comparison. */
template <int N>
__device__ __forceinline__ int od_lane_greedy_scan(const uint32_t *pk, int lane, unsigned xy,
 unsigned yyp1) {
  if (N <= 16) {
    double ba = 0;
    double bb = 1;
    int pos = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const uint32_t w = pk[j*kPitch + lane];
      const double tt = (double)(xy + (w >> 16));
      const double a = tt*tt;
      const double b = (double)(yyp1 + (w & 0xffffu));
      if (j == 0) {
        ba = a;
        bb = b;
      }
      else {
        const unsigned long long m = od_cmp_gt(a*bb, ba*b);
        ba = od_sel(m, a, ba);
        bb = od_sel(m, b, bb);
        pos = od_sel(m, j, pos);
      }
    }
    return pos;
  }
  constexpr int NG = (N + kGrp - 1)/kGrp;
  uint32_t w0[kGrp];
  uint32_t w1[kGrp];
  od_lane_load_group<N>(w0, pk, lane, 0);
  double ba = 0;
  double bb = 1;
  int pos = 0;
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (g + 1 < NG) od_lane_load_group<N>(w1, pk, lane, g + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kGrp; t++) {
      const int j = g*kGrp + t;
      if (j < N) {
        const double tt = (double)(xy + (w0[t] >> 16));
        const double a = tt*tt;
        const double b = (double)(yyp1 + (w0[t] & 0xffffu));
        if (j == 0) {
          ba = a;
          bb = b;
        }
        else {
          const unsigned long long m = od_cmp_gt(a*bb, ba*b);
          ba = od_sel(m, a, ba);
          bb = od_sel(m, b, bb);
          pos = od_sel(m, j, pos);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kGrp; t++) w0[t] = w1[t];
  }
  return pos;
}

/* One candidate: K-pulse search for the lanes with `on` set (k <= kMaxK).
   `fresh` lanes start from the L1 projection (k > 2) or from zero; the others
   continue from the pulses of the previous candidate (prev_k <= k,
   src/pvq_encoder.c:128-137), whose xy / yy / i are still in `s`.  LDS word of
   coefficient j: |x| << 16 | 2*y.  xy, yy and the per-candidate t = xy + x_j,
   b = yy + 2*y_j + 1 are formed in 32-bit integers and converted: they are
   below 2^31, so the doubles are the reference's.  Returns the cosine
   distance. */
template <int N>
__device__ __forceinline__ double od_lane_search(LaneSearch &s, uint32_t *pk, const double *rsq,
 int lane, bool on, bool fresh, int k, double g2, double pvq_norm_lambda) {
  fresh = fresh && on;
  if (__any(fresh)) {
    /* src/pvq_encoder.c:139-153; k <= 2 starts from zero, which is the same
       loop with a projection factor of 0. */
    const double kd = fresh && k > 2 ? (double)k : 0.;
    unsigned xy = 0;
    unsigned yy = 0;
    int i = 0;
#pragma unroll 4
    for (int j = 0; j < N; j++) {
      const unsigned ax = pk[j*kPitch + lane] >> 16;
      const double tmp = (kd*(double)ax)*s.l1_inv;
      int yj = (int)floor(tmp);
      yj = yj > 0 ? yj : 0;
      if (fresh) pk[j*kPitch + lane] = ax << 16 | (unsigned)yj << 1;
      xy += ax*(unsigned)yj;
      yy += (unsigned)yj*(unsigned)yj;
      i += yj;
    }
    if (fresh) {
      s.xy = xy;
      s.yy = yy;
      s.i = i;
    }
  }
  const int rdo_pulses = 1 + k/4;
  const int n_greedy = k - rdo_pulses;
  /* Greedy pulses, src/pvq_encoder.c:165-187. */
  while (__any(on && s.i < n_greedy)) {
    const bool step = on && s.i < n_greedy;
    const int pos = od_lane_greedy_scan<N>(pk, lane, s.xy, s.yy + 1);
    if (step) {
      const uint32_t w = pk[pos*kPitch + lane];
      s.xy += w >> 16;
      s.yy += (w & 0xffffu) + 1;
      pk[pos*kPitch + lane] = w + 2;
      s.i++;
    }
  }
  /* Last pulses with the rate term, src/pvq_encoder.c:192-219. */
  const double lambda = __ddiv_rn(pvq_norm_lambda, 1e-30 + g2);
  double delta_rate = 3./N;
  double accel_rate = 0.;
  if (k == 1) {
    if (N == 15) {
      accel_rate = -8./N;
      delta_rate = 4.5/N - accel_rate;
    }
    else if (N == 8) {
      accel_rate = 5.7/N;
      delta_rate = 9.3/N - accel_rate;
    }
  }
  while (__any(on && s.i < k)) {
    const bool step = on && s.i < k;
    int pos = 0;
    /* Largest table index any candidate of this pulse can use: yy + 2*y + 1
       with y <= i < k. */
    if (!__any(step && s.yy + 2*(unsigned)k + 1 > (unsigned)kRsqN)) {
      const unsigned base = step ? s.yy : 0;   /* table slot of index yy + 1 */
      const unsigned xy = step ? s.xy : 0;
      if ((N == 8 || N == 15) && __any(step && k == 1)) {
        pos = od_lane_rdo_scan<N, true>(pk, rsq, lane, xy, base, s.norm2, lambda, delta_rate,
         accel_rate);
      }
      else {
        pos = od_lane_rdo_scan<N, false>(pk, rsq, lane, xy, base, s.norm2, lambda, delta_rate,
         accel_rate);
      }
    }
    else {
      double best = 0;
      for (int j = 0; j < N; j++) {
        const uint32_t w = pk[j*kPitch + lane];
        const double t = (double)(s.xy + (w >> 16));
        const unsigned idx = s.yy + (w & 0xffffu) + 1;
        double r;
        if (step && idx <= (unsigned)kRsqN) r = rsq[idx - 1];
        else r = __ddiv_rn(1., __dsqrt_rn((double)idx));
        const double val = (t*s.norm2)*r - (lambda*j)*(delta_rate + j*accel_rate);
        if (j == 0 || val > best) {
          best = val;
          pos = j;
        }
      }
    }
    if (step) {
      const uint32_t w = pk[pos*kPitch + lane];
      s.xy += w >> 16;
      s.yy += (w & 0xffffu) + 1;
      pk[pos*kPitch + lane] = w + 2;
      s.i++;
    }
  }
  return __ddiv_rn((double)s.xy, 1e-100 + __dsqrt_rn(s.xx*(double)s.yy));
}

}  // namespace
