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
#include "od_sel.cuh"

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

struct LaneSearch {
  double xx;
  double norm2;   /* 2*norm_1 */
  double l1_inv;
  unsigned xmax;  /* largest |x| of the band (pair mode: exactness bound) */
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
/* PAIR MODE (S = 2): a band of 2*N coefficients is shared by two adjacent
   lanes, the even lane holding coding positions 0..N-1 and the odd lane
   N..2N-1 (`jbase` = first position of the lane).  It halves the LDS of a
   wavefront (n = 128: 16 KiB instead of 32, so two wavefronts fit a SIMD and
   other kernels fit beside them) and the latency of a pulse.  Sums are
   combined with one DPP swap; the two halves of an argmax are combined as
   described at od_lane_search. */
__device__ __forceinline__ int od_pair_swap(int v) {
  return row_mov<OD_DPP_XOR1>(v);
}

__device__ __forceinline__ unsigned od_pair_swap(unsigned v) {
  return (unsigned)row_mov<OD_DPP_XOR1>((int)v);
}

__device__ __forceinline__ double od_pair_swap(double v) {
  return row_mov<OD_DPP_XOR1>(v);
}

__device__ __forceinline__ unsigned long long od_pair_swap(unsigned long long v) {
  const unsigned lo = od_pair_swap((unsigned)v);
  const unsigned hi = od_pair_swap((unsigned)(v >> 32));
  return (unsigned long long)hi << 32 | lo;
}

/* GROUP MODE in general (S = 2 or 4 lanes per band; round 5 added the quad): lane `sub` = lane % S of the
   group holds coding positions sub*N .. sub*N + N - 1.  Four lanes per 128-coefficient band halve the
   LDS of a wavefront again (8 KiB of columns): twice the wavefronts per SIMD for the same bands in
   flight, each lane's argmax chain half as long - the pair form ran at 2 wavefronts per SIMD and 0.67 of
   VALU issue, a dependency-latency bound. */
__device__ __forceinline__ int od_quad_swap2(int v) {
  return row_mov<OD_DPP_XOR2>(v);
}
__device__ __forceinline__ unsigned od_quad_swap2(unsigned v) {
  return (unsigned)row_mov<OD_DPP_XOR2>((int)v);
}
__device__ __forceinline__ double od_quad_swap2(double v) {
  return row_mov<OD_DPP_XOR2>(v);
}
__device__ __forceinline__ unsigned long long od_quad_swap2(unsigned long long v) {
  const unsigned lo = od_quad_swap2((unsigned)v);
  const unsigned hi = od_quad_swap2((unsigned)(v >> 32));
  return (unsigned long long)hi << 32 | lo;
}
/* sum / maximum / bitwise or over the S lanes of a group, in every lane of it */
template <int S, typename V>
__device__ __forceinline__ V od_grp_add(V v) {
  if (S >= 2) v += od_pair_swap(v);
  if (S >= 4) v += od_quad_swap2(v);
  return v;
}
template <int S>
__device__ __forceinline__ unsigned od_grp_umax(unsigned v) {
  if (S >= 2) {
    const unsigned o = od_pair_swap(v);
    v = o > v ? o : v;
  }
  if (S >= 4) {
    const unsigned o = od_quad_swap2(v);
    v = o > v ? o : v;
  }
  return v;
}
template <int S>
__device__ __forceinline__ uint32_t od_grp_or(uint32_t v) {
  if (S >= 2) v |= od_pair_swap(v);
  if (S >= 4) v |= od_quad_swap2(v);
  return v;
}
/* the value of lane Q of every group of S lanes, in all of its lanes: quad_perm [Q,Q,Q,Q] for a quad,
   [0,0,2,2] / [1,1,3,3] for the two pairs of a quad */
template <int S, int Q>
__device__ __forceinline__ int od_grp_bcast(int v) {
  static_assert((S == 4 && Q < 4) || (S == 2 && Q < 2), "a lane of the group");
  return row_mov<(S == 4 ? Q*0x55 : Q == 0 ? 0xA0 : 0xF5)>(v);
}
template <int S, int Q>
__device__ __forceinline__ double od_grp_bcast(double v) {
  return row_mov<(S == 4 ? Q*0x55 : Q == 0 ? 0xA0 : 0xF5)>(v);
}

/* Per-band constants: xx, 2/sqrt(1e-30 + xx), 1/max(L1, 1e-100)
   (src/pvq_encoder.c:118-125,:139-141). */
template <int N, int S>
__device__ __forceinline__ void od_lane_prepare(LaneSearch &s, const uint32_t *pk, int lane) {
  unsigned long long xx = 0;
  unsigned l1 = 0;
  unsigned xmax = 0;
#pragma unroll 8
  for (int j = 0; j < N; j++) {
    const unsigned ax = pk[j*kPitch + lane] >> 16;
    xx += (unsigned long long)ax*ax;
    l1 += ax;
    xmax = ax > xmax ? ax : xmax;
  }
  xx = od_grp_add<S>(xx);
  l1 = od_grp_add<S>(l1);
  xmax = od_grp_umax<S>(xmax);
  s.xx = (double)xx;
  s.norm2 = 2*__ddiv_rn(1., __dsqrt_rn(1e-30 + s.xx));
  const double l1d = (double)l1;
  s.l1_inv = __ddiv_rn(1., l1d > 1e-100 ? l1d : 1e-100);
  s.xmax = xmax;
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
 unsigned xy, unsigned base, double norm2, double lambda, double delta_rate, double accel_rate,
 int jbase, double &best_out) {
  if (N <= 16) {
    /* short bands: fully unrolled, every load is requested up front anyway */
    double best = 0;
    int pos = jbase;
    int jv = jbase;
#pragma unroll 4
    for (int j = 0; j < N; j++) {
      const uint32_t w = pk[j*kPitch + lane];
      const double tt = (double)(xy + (w >> 16));
      const double r = rsq[base + (w & 0xffffu)];
      if (j > 0) od_inc(jv);
      /* the position as a double from the running index register: exact, and
         (being opaque to the compiler) it keeps the per-candidate penalties
         from being hoisted out of the pulse loop into 2 registers each */
      const double jd = (double)jv;
      const double val = (tt*norm2)*r - (lambda*jd)*(ACCEL ? delta_rate + jd*accel_rate : delta_rate);
      if (j == 0) best = val;
      else {
        const unsigned long long m = od_cmp_gt(val, best);
        best = od_sel(m, val, best);
        pos = od_sel(m, jv, pos);
      }
    }
    best_out = best;
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
  int pos = jbase;
  int jv = jbase;
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
        if (j > 0) od_inc(jv);
        const double jd = (double)jv;
        const double val = (tt*norm2)*r0[t]
         - (lambda*jd)*(ACCEL ? delta_rate + jd*accel_rate : delta_rate);
        if (j == 0) best = val;
        else {
          const unsigned long long m = od_cmp_gt(val, best);
          best = od_sel(m, val, best);
          pos = od_sel(m, jv, pos);
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
  best_out = best;
  return pos;
}

/* One pass of the greedy argmax (src/pvq_encoder.c:172-183): the candidate
   maximising (xy + x_j)^2/(yy + 2*y_j + 1), scanned upward with the
   reference's cross-multiplied comparison. */
template <int N>
__device__ __forceinline__ int od_lane_greedy_scan(const uint32_t *pk, int lane, unsigned xy,
 unsigned yyp1, int jbase, double &ba_out, double &bb_out) {
  if (N <= 16) {
    double ba = 0;
    double bb = 1;
    int pos = jbase;
    int jv = jbase;
#pragma unroll 4
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
        od_inc(jv);
        const unsigned long long m = od_cmp_gt(a*bb, ba*b);
        ba = od_sel(m, a, ba);
        bb = od_sel(m, b, bb);
        pos = od_sel(m, jv, pos);
      }
    }
    ba_out = ba;
    bb_out = bb;
    return pos;
  }
  constexpr int NG = (N + kGrp - 1)/kGrp;
  uint32_t w0[kGrp];
  uint32_t w1[kGrp];
  od_lane_load_group<N>(w0, pk, lane, 0);
  double ba = 0;
  double bb = 1;
  int pos = jbase;
  int jv = jbase;
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
          od_inc(jv);
          const unsigned long long m = od_cmp_gt(a*bb, ba*b);
          ba = od_sel(m, a, ba);
          bb = od_sel(m, b, bb);
          pos = od_sel(m, jv, pos);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kGrp; t++) w0[t] = w1[t];
  }
  ba_out = ba;
  bb_out = bb;
  return pos;
}

/* ---- greedy pulses by CLASS FRONTS (round 6; pair mode) --------------------------------------------------
   The greedy argmax (src/pvq_encoder.c:172-183) maximises (xy + |x_j|)^2 / (yy + 2*y_j + 1).  Among the
   candidates of one CLASS - equal pulse count y_j, hence equal denominator - the quotient is strictly
   increasing in |x_j|, so the maximum over the band is attained by the FRONT of some class: its member with the
   largest |x|, lowest position among equals.  In the exact regime (every product of the pulse an exact integer
   below 2^52, the condition the pair mode checks anyway) the reference's left-to-right scan ends on the lowest
   position among the candidates of maximal quotient, which is therefore the front that wins the exact comparison
   of the fronts, ties to the lower position.  A pulse moves its winner from class c to c + 1: only class c needs a
   new front (one scan of the lane's column: 5 instructions per word against the 15 of an evaluation) and class
   c + 1 takes the winner if it beats its front.  Fronts are tracked for pulse counts 0 .. kFrontClasses - 1; a
   band that holds a larger count, or leaves the exact regime, goes on in the literal loop below - so does
   ODHIP_PVQ_FORCE_SEQ=1, which is how the tests cross-check the two.
   Key of a member: (|x| + 1) << 16 | (N - 1 - j): larger |x| first, lower position on ties; 0 = empty class. */
#ifndef ODHIP_GREEDY_FRONTS
# define ODHIP_GREEDY_FRONTS 1     /* 0: the literal scan for every pulse (the A/B baseline, profiles/r6_fronts.txt) */
#endif
#ifndef ODHIP_FRONT_CLASSES
# define ODHIP_FRONT_CLASSES 8
#endif
#ifndef ODHIP_FRONT_MIN
# define ODHIP_FRONT_MIN 4
#endif
constexpr int kFrontClasses = ODHIP_FRONT_CLASSES;
constexpr int kFrontMinPulses = ODHIP_FRONT_MIN;     /* fewer greedy pulses left in every band of the wavefront: not worth the set-up */

template <int N>
__device__ __forceinline__ uint32_t od_lane_front_scan(const uint32_t *pk, int lane, uint32_t c2) {
  static_assert(N % kGrp == 0, "whole groups");
  constexpr int NG = N/kGrp;
  uint32_t best = 0;
  uint32_t w0[kGrp];
  uint32_t w1[kGrp];
  od_lane_load_group<N>(w0, pk, lane, 0);
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (g + 1 < NG) od_lane_load_group<N>(w1, pk, lane, g + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kGrp; t++) {
      const int j = g*kGrp + t;
      const uint32_t key = (w0[t] & 0xffff0000u) + (0x10000u + (uint32_t)(N - 1 - j));
      const uint32_t m = (w0[t] & 0xffffu) == c2 ? key : 0u;
      best = m > best ? m : best;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kGrp; t++) w0[t] = w1[t];
  }
  return best;
}

/* Greedy pulses of a lane pair's band until n_greedy pulses are placed or the conditions above end; the caller's
   literal loop places whatever is left. */
template <int N>
__device__ __forceinline__ void od_lane_greedy_fronts(LaneSearch &s, uint32_t *pk, int lane, int half, bool on,
 int k, int n_greedy) {
  const int jbase = half*N;
  unsigned present = 0;
#pragma unroll 8
  for (int j = 0; j < N; j++) {
    const unsigned yc = (pk[j*kPitch + lane] & 0xffffu) >> 1;
    present |= 1u << (yc < (unsigned)kFrontClasses ? yc : (unsigned)kFrontClasses);
  }
  const bool active = on && s.i < n_greedy;
  const unsigned both = present | od_pair_swap(present);
  if (__any(active && (both >> kFrontClasses) != 0)) return;
  uint32_t fr[kFrontClasses];
#pragma unroll
  for (int c = 0; c < kFrontClasses; c++) {
    fr[c] = 0;
    if (__any(active && (present >> c & 1u))) fr[c] = od_lane_front_scan<N>(pk, lane, 2u*c);
  }
  while (__any(on && s.i < n_greedy)) {
    const bool step = on && s.i < n_greedy;
    const double tmax = (double)(s.xy + s.xmax);
    const bool exact = (tmax*tmax)*(double)(s.yy + 2*(unsigned)k + 1) < 4503599627370496.;   /* 2^52, as in the literal loop */
    if (__any(step && !exact)) return;
    /* the best front of this lane's half ... */
    double ba = 0;
    double bb = 1;
    int bpos = 0;
    int bc = 0;
    bool have = false;
#pragma unroll
    for (int c = 0; c < kFrontClasses; c++) {
      if (__any(step && fr[c] != 0)) {
        const uint32_t key = fr[c];
        const bool v = key != 0;
        const double tt = (double)(s.xy + ((key >> 16) - 1u));
        const double a = tt*tt;
        const double b = (double)(s.yy + 1u + 2u*c);
        const int p = jbase + (N - 1 - (int)(key & 0xffffu));
        const double l = a*bb;
        const double r = ba*b;
        const bool better = v && (!have || l > r || (l == r && p < bpos));
        ba = better ? a : ba;
        bb = better ? b : bb;
        bpos = better ? p : bpos;
        bc = better ? c : bc;
        have = have || v;
      }
    }
    /* ... against the other half's */
    const double oa = od_pair_swap(ba);
    const double ob = od_pair_swap(bb);
    const int op = od_pair_swap(bpos);
    const int oc = od_pair_swap(bc);
    const bool ohave = od_pair_swap((int)have) != 0;
    const double l = oa*bb;
    const double r = ba*ob;
    const bool other = ohave && (!have || l > r || (l == r && op < bpos));
    const int pos = other ? op : bpos;
    const int cw = other ? oc : bc;
    const bool owner = pos/N == half;
    const int local = pos - jbase;
    uint32_t w = 0;
    if (step && owner) w = pk[local*kPitch + lane];
    w = od_grp_or<2>(w);
    const bool mine = step && owner;
    if (step) {
      s.xy += w >> 16;
      s.yy += (w & 0xffffu) + 1;
      if (owner) pk[local*kPitch + lane] = w + 2;
      s.i++;
    }
    /* the winner's old class needs a new front (the winner now counts one pulse more: the scan passes it
       over), its new class takes it if it beats that front */
    const uint32_t nf = od_lane_front_scan<N>(pk, lane, mine ? 2u*(uint32_t)cw : 0xffffffffu);
    const uint32_t wkey = ((w + 2) & 0xffff0000u) + (0x10000u + (uint32_t)(N - 1 - local));
#pragma unroll
    for (int c = 0; c < kFrontClasses; c++) {
      if (mine && c == cw) fr[c] = nf;
      if (mine && c == cw + 1) fr[c] = wkey > fr[c] ? wkey : fr[c];
    }
    if (__any(mine && cw + 1 >= kFrontClasses)) return;     /* a pulse count without a tracked class */
  }
}

/* One candidate: K-pulse search for the lanes with `on` set (k <= kMaxK).
   `fresh` lanes start from the L1 projection (k > 2) or from zero; the others
   continue from the pulses of the previous candidate (prev_k <= k,
   src/pvq_encoder.c:128-137), whose xy / yy / i are still in `s`.  LDS word of
   coefficient j: |x| << 16 | 2*y.  xy, yy and the per-candidate t = xy + x_j,
   b = yy + 2*y_j + 1 are formed in 32-bit integers and converted: they are
   below 2^31, so the doubles are the reference's.  Returns the cosine
   distance.

   N = coefficients per lane, S = lanes per band (band size n = N*S), `half` =
   lane % S.  In pair mode each lane scans its half and the halves are
   combined:
     - rate pass: the costs are doubles compared with `>`, (max cost, lowest
       position) is a total order, so max-of-halves with the lower half winning
       ties IS the sequential scan;
     - greedy pass: the reference's `a_j*b_best > a_best*b_j` compares ROUNDED
       products and need not be transitive.  When (xy + max|x|)^2 *
       (yy + 2k + 1) < 2^53 every product of the pulse is an exact integer, the
       comparison is the exact order of the rationals a/b, and "upper half's
       best beats lower half's best, else lower" is what the sequential scan
       ends on.  Otherwise (never seen with 8-bit video; the bound is checked
       every pulse) the upper lane rescans its half starting from the lower
       lane's best, which is the sequential scan literally
       (ODHIP_PVQ_FORCE_SEQ=1 forces that path: tests run both). */
template <int N, int S>
__device__ __forceinline__ double od_lane_search(LaneSearch &s, uint32_t *pk, const double *rsq,
 int lane, int half, bool on, bool fresh, int k, double g2, double pvq_norm_lambda,
 bool force_seq) {
  constexpr int NBAND = N*S;
  static_assert(S == 1 || S == 2 || S == 4, "one lane, a pair or a quad per band");
  const int jbase = S > 1 ? half*N : 0;       /* `half` = the lane's index in its group */
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
    xy = od_grp_add<S>(xy);
    yy = od_grp_add<S>(yy);
    i = od_grp_add<S>(i);
    if (fresh) {
      s.xy = xy;
      s.yy = yy;
      s.i = i;
    }
  }
  const int rdo_pulses = 1 + k/4;
  const int n_greedy = k - rdo_pulses;
#if ODHIP_GREEDY_FRONTS
  if constexpr (S == 2 && N >= 32 && N % kGrp == 0) {   /* (the 32-coefficient bands, 16 per lane: +26 VGPRs cost their kernel a wavefront per SIMD) */
    if (!force_seq && __any(on && n_greedy - s.i >= kFrontMinPulses)) {
      od_lane_greedy_fronts<N>(s, pk, lane, half & 1, on, k, n_greedy);
    }
  }
#endif
  /* Greedy pulses, src/pvq_encoder.c:165-187. */
  while (__any(on && s.i < n_greedy)) {
    const bool step = on && s.i < n_greedy;
    double ba;
    double bb;
    int pos = od_lane_greedy_scan<N>(pk, lane, s.xy, s.yy + 1, jbase, ba, bb);
    if constexpr (S > 1) {
      const double tmax = (double)(s.xy + s.xmax);
      const bool exact = !force_seq
       && (tmax*tmax)*(double)(s.yy + 2*(unsigned)k + 1) < 4503599627370496.;   /* 2^52: a factor 2 of margin for the bound's own rounding */
      if (__any(step && !exact)) {
        /* sequential: lane q of the group rescans its part starting from the result of lanes 0..q-1
           (lane 0's own scan IS the start of the sequential scan) */
        double ra = od_grp_bcast<S, 0>(ba);
        double rb = od_grp_bcast<S, 0>(bb);
        int rp = od_grp_bcast<S, 0>(pos);
        auto rescan = [&](double &xa, double &xb, int &xp) {
#pragma unroll 1
          for (int j = 0; j < N; j++) {
            const uint32_t w = pk[j*kPitch + lane];
            const double tt = (double)(s.xy + (w >> 16));
            const double a = tt*tt;
            const double b = (double)(s.yy + 1 + (w & 0xffffu));
            if (a*xb > xa*b) {
              xa = a;
              xb = b;
              xp = jbase + j;
            }
          }
        };
        {
          double ta = ra;
          double tb = rb;
          int tp = rp;
          rescan(ta, tb, tp);                    /* valid in lane 1 */
          ra = od_grp_bcast<S, 1>(ta);
          rb = od_grp_bcast<S, 1>(tb);
          rp = od_grp_bcast<S, 1>(tp);
        }
        if constexpr (S == 4) {
          double ta = ra;
          double tb = rb;
          int tp = rp;
          rescan(ta, tb, tp);                    /* valid in lane 2 */
          ra = od_grp_bcast<4, 2>(ta);
          rb = od_grp_bcast<4, 2>(tb);
          rp = od_grp_bcast<4, 2>(tp);
          ta = ra;
          tb = rb;
          tp = rp;
          rescan(ta, tb, tp);                    /* valid in lane 3 */
          rp = od_grp_bcast<4, 3>(tp);
        }
        pos = rp;
      }
      else {
        /* exact regime: the comparison is the exact order of the rationals, so a tournament in which
           the part with the HIGHER positions wins only when strictly greater is the sequential scan */
        {
          const double oa = od_pair_swap(ba);
          const double ob = od_pair_swap(bb);
          const int op = od_pair_swap(pos);
          const bool up = (half & 1) != 0;
          const double la = up ? oa : ba;      /* lower positions */
          const double lb = up ? ob : bb;
          const int lp = up ? op : pos;
          const double xa = up ? ba : oa;      /* upper positions */
          const double xb = up ? bb : ob;
          const int xp = up ? pos : op;
          const bool win = xa*lb > la*xb;
          ba = win ? xa : la;
          bb = win ? xb : lb;
          pos = win ? xp : lp;
        }
        if constexpr (S == 4) {
          const double oa = od_quad_swap2(ba);
          const double ob = od_quad_swap2(bb);
          const int op = od_quad_swap2(pos);
          const bool up = (half & 2) != 0;
          const double la = up ? oa : ba;
          const double lb = up ? ob : bb;
          const int lp = up ? op : pos;
          const double xa = up ? ba : oa;
          const double xb = up ? bb : ob;
          const int xp = up ? pos : op;
          pos = xa*lb > la*xb ? xp : lp;
        }
      }
    }
    int owner = 1;
    int local = pos;
    if constexpr (S > 1) {
      owner = pos/N == half;
      local = pos - jbase;
    }
    uint32_t w = 0;
    if (step && owner) w = pk[local*kPitch + lane];
    w = od_grp_or<S>(w);
    if (step) {
      s.xy += w >> 16;
      s.yy += (w & 0xffffu) + 1;
      if (owner) pk[local*kPitch + lane] = w + 2;
      s.i++;
    }
  }
  /* Last pulses with the rate term, src/pvq_encoder.c:192-219. */
  const double lambda = __ddiv_rn(pvq_norm_lambda, 1e-30 + g2);
  double delta_rate = 3./NBAND;
  double accel_rate = 0.;
  if (k == 1) {
    if (NBAND == 15) {
      accel_rate = -8./NBAND;
      delta_rate = 4.5/NBAND - accel_rate;
    }
    else if (NBAND == 8) {
      accel_rate = 5.7/NBAND;
      delta_rate = 9.3/NBAND - accel_rate;
    }
  }
  while (__any(on && s.i < k)) {
    const bool step = on && s.i < k;
    int pos = 0;
    double best = 0;
    /* Largest table index any candidate of this pulse can use: yy + 2*y + 1
       with y <= i < k. */
    if (!__any(step && s.yy + 2*(unsigned)k + 1 > (unsigned)kRsqN)) {
      const unsigned base = step ? s.yy : 0;   /* table slot of index yy + 1 */
      const unsigned xy = step ? s.xy : 0;
      if ((NBAND == 8 || NBAND == 15) && __any(step && k == 1)) {
        pos = od_lane_rdo_scan<N, true>(pk, rsq, lane, xy, base, s.norm2, lambda, delta_rate,
         accel_rate, jbase, best);
      }
      else {
        pos = od_lane_rdo_scan<N, false>(pk, rsq, lane, xy, base, s.norm2, lambda, delta_rate,
         accel_rate, jbase, best);
      }
    }
    else {
      /* rare (an index beyond the table): not unrolled, it must not set the
         register budget of the kernel */
#pragma unroll 1
      for (int j = 0; j < N; j++) {
        const uint32_t w = pk[j*kPitch + lane];
        const double t = (double)(s.xy + (w >> 16));
        const unsigned idx = s.yy + (w & 0xffffu) + 1;
        double r;
        if (step && idx <= (unsigned)kRsqN) r = rsq[idx - 1];
        else r = od_rsqrt_beyond((int)(idx ? idx : 1));      /* (idx >= 1; a lane that does not step computes nothing used) */
        const int jg = jbase + j;
        const double val = (t*s.norm2)*r - (lambda*jg)*(delta_rate + jg*accel_rate);
        if (j == 0 || val > best) {
          best = val;
          pos = jg;
        }
      }
    }
    if constexpr (S > 1) {
      /* (max cost, lowest position): a total order - a tournament in which the higher positions win
         only when strictly greater */
      {
        const double ob = od_pair_swap(best);
        const int op = od_pair_swap(pos);
        const bool up = (half & 1) != 0;
        const double va = up ? ob : best;   /* lower positions */
        const double vb = up ? best : ob;   /* upper positions */
        const int pa = up ? op : pos;
        const int pb = up ? pos : op;
        const bool win = vb > va;
        best = win ? vb : va;
        pos = win ? pb : pa;
      }
      if constexpr (S == 4) {
        const double ob = od_quad_swap2(best);
        const int op = od_quad_swap2(pos);
        const bool up = (half & 2) != 0;
        const double va = up ? ob : best;
        const double vb = up ? best : ob;
        const int pa = up ? op : pos;
        const int pb = up ? pos : op;
        pos = vb > va ? pb : pa;
      }
    }
    int owner = 1;
    int local = pos;
    if constexpr (S > 1) {
      owner = pos/N == half;
      local = pos - jbase;
    }
    uint32_t w = 0;
    if (step && owner) w = pk[local*kPitch + lane];
    w = od_grp_or<S>(w);
    if (step) {
      s.xy += w >> 16;
      s.yy += (w & 0xffffu) + 1;
      if (owner) pk[local*kPitch + lane] = w + 2;
      s.i++;
    }
  }
  return __ddiv_rn((double)s.xy, 1e-100 + __dsqrt_rn(s.xx*(double)s.yy));
}

}  // namespace
