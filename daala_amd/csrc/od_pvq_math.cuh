/* od_pvq_math.cuh - device-side fixed-point PVQ helpers (gain, companding,
   K selection, synthesis scale), bit-exact with the reference's integer
   arithmetic in src/pvq.c and src/odintrin.h:164-199.

   Integer widths, arithmetic shifts, truncating divisions and the int16
   truncation on store are part of the specification and are spelled out. */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ODQ_CGAIN_SHIFT 8     /* OD_CGAIN_SHIFT, src/pvq.h:88 */
#define ODQ_COMPAND_SHIFT 12  /* OD_COMPAND_SHIFT, src/pvq.h:104 */
#define ODQ_BETA_SHIFT 12     /* OD_BETA_SHIFT, src/pvq.h:81 */
#define ODQ_QM_SHIFT 11       /* OD_QM_SHIFT, src/pvq.h:60 */
#define ODQ_QM_INV_SHIFT 12   /* OD_QM_INV_SHIFT, src/pvq.h:67 */
#define ODQ_BETA_1_0 4096
#define ODQ_BETA_1_5 6144

__device__ __forceinline__ int odq_ilog(uint32_t v) { /* OD_ILOG */
  return v ? 32 - __clz((int)v) : 0;
}

__device__ __forceinline__ int32_t odq_shl32(int64_t a, int s) { /* OD_SHL */
  return (int32_t)((uint32_t)a << s);
}

__device__ __forceinline__ int32_t odq_shr_round(int64_t x, int s) { /* OD_SHR_ROUND */
  return (int32_t)((x + ((1 << s) >> 1)) >> s);
}

__device__ __forceinline__ int32_t odq_vshr(int64_t x, int s) { /* OD_VSHR */
  return s > 0 ? (int32_t)(x >> s) : odq_shl32(x, -s);
}

__device__ __forceinline__ int32_t odq_vshr_round(int64_t x, int s) { /* OD_VSHR_ROUND */
  return s > 0 ? odq_shr_round(x, s) : odq_shl32(x, -s);
}

__device__ __forceinline__ int32_t odq_mult16_16_q15(int32_t a, int32_t b) {
  return ((int16_t)a*(int32_t)(int16_t)b) >> 15;
}

__device__ __forceinline__ int32_t odq_mult16_16_qbeta(int32_t a, int32_t b) {
  return ((int16_t)a*(int32_t)(int16_t)b) >> ODQ_BETA_SHIFT;
}

/* od_rsqrt_norm, src/pvq.c:968-997. */
__device__ __forceinline__ int16_t odq_rsqrt_norm(int16_t t) {
  const int16_t n = (int16_t)(t - 32768);
  const int32_t r = 23565 + odq_mult16_16_q15(n, -13481 + odq_mult16_16_q15(n, 6711));
  const int32_t r2 = r*r;
  const int32_t y = (((r2 >> 15)*n + r2) >> 12) - 131077;
  const int32_t ry = r*y;
  return (int16_t)(r + ((((ry >> 16)*(3*y) >> 3) - ry) >> 18));
}

/* od_rsqrt, src/pvq.c:999-1015. */
__device__ __forceinline__ int16_t odq_rsqrt(int32_t x, int *rsqrt_shift) {
  const int k = (odq_ilog(x) - 1) >> 1;
  const int s = 2*k - 14;
  *rsqrt_shift = 14 + ((s + 16) >> 1);
  return odq_rsqrt_norm((int16_t)odq_vshr(x, s));
}

/* od_sqrt_norm + od_sqrt, src/pvq.c:729-756. */
__device__ __forceinline__ int16_t odq_sqrt(int32_t x, int *sqrt_shift) {
  if (x == 0) {
    *sqrt_shift = 0;
    return 0;
  }
  const int k = (odq_ilog(x) - 1) >> 1;
  const int s = 2*k - 14;
  const int32_t t = odq_vshr(x, s);
  *sqrt_shift = 15 - ((s + 16) >> 1);
  const int32_t v = odq_shr_round(t*odq_rsqrt_norm((int16_t)t), 15);
  return (int16_t)(v < 32767 ? v : 32767);
}

/* od_rcp, src/pvq.c:526-549. */
__device__ __forceinline__ int16_t odq_rcp(int16_t x) {
  const int i = odq_ilog(x) - 1;
  const int16_t n = (int16_t)(odq_vshr_round(x, i - 15) - 32768);
  int16_t r = (int16_t)(30840 + odq_mult16_16_q15(-15420, n));
  r = (int16_t)(r - odq_mult16_16_q15(r, odq_mult16_16_q15(r, n) + r - 32768));
  r = (int16_t)(r - (1 + odq_mult16_16_q15(r, odq_mult16_16_q15(r, n) + r - 32768)));
  return (int16_t)odq_vshr_round(r, i - 14);
}

/* od_beta_rcp, src/pvq.c:626-637. */
__device__ __forceinline__ int16_t odq_beta_rcp(int16_t beta) {
  if (beta == ODQ_BETA_1_0) return ODQ_BETA_1_0;
  if (beta == ODQ_BETA_1_5) return 2731;
  return (int16_t)odq_shr_round(odq_rcp((int16_t)(beta << (15 - 1 - ODQ_BETA_SHIFT))),
   14 + 1 - ODQ_BETA_SHIFT);
}

/* od_exp2, src/pvq.c:642-665. */
__device__ __forceinline__ int32_t odq_exp2(int32_t x) {
  const int integer = x >> 15;
  if (integer > 14) return 0x7f000000;
  if (integer < -15) return 0;
  const int32_t f = x - odq_shl32(integer, 15);
  const int32_t frac = odq_mult16_16_q15(f, 22709 + odq_mult16_16_q15(f, 7913
   + odq_mult16_16_q15(f, 1704 + odq_mult16_16_q15(f, 443))));
  return odq_vshr_round(32768 + frac, -integer) + 1;
}

/* od_log2, src/pvq.c:671-676. */
__device__ __forceinline__ int16_t odq_log2(int16_t x) {
  return (int16_t)(x + odq_mult16_16_q15(x, 14482 + odq_mult16_16_q15(x, -23234
   + odq_mult16_16_q15(x, 13643 + odq_mult16_16_q15(x, -6403
   + odq_mult16_16_q15(x, 1515))))));
}

/* od_pow, src/pvq.c:678-696. */
__device__ __forceinline__ int32_t odq_pow(int32_t x, int16_t beta) {
  if (x == 0) return 0;
  const int log2_x = odq_ilog(x) - 1;
  const int16_t t = (int16_t)(odq_vshr(x, log2_x - 15) - 32768);
  int32_t logr = odq_log2(t) + (log2_x - ODQ_COMPAND_SHIFT)*32768;
  logr = (int32_t)(beta*(int64_t)logr >> ODQ_BETA_SHIFT);
  return odq_exp2(logr);
}

/* od_gain_compand, src/pvq.c:706-722. */
__device__ __forceinline__ int32_t odq_gain_compand(int32_t g, int q0, int16_t beta) {
  if (beta == ODQ_BETA_1_0) return (256*g + (q0 >> 1))/q0;
  int32_t expr = odq_pow(g, odq_beta_rcp(beta));
  expr <<= ODQ_CGAIN_SHIFT + ODQ_COMPAND_SHIFT - 15;
  return (expr + (q0 >> 1))/q0;
}

/* od_gain_expand, src/pvq.c:766-811. */
__device__ __forceinline__ int32_t odq_gain_expand(int32_t cg0, int q0, int beta) {
  if (beta == ODQ_BETA_1_0) return odq_shr_round(cg0*q0, ODQ_CGAIN_SHIFT);
  if (beta == ODQ_BETA_1_5) {
    int sqrt_outshift;
    const int32_t irt = odq_sqrt(cg0*q0, &sqrt_outshift);
    const int64_t tmp = cg0*q0*(int64_t)irt;
    return odq_vshr_round(tmp, ODQ_CGAIN_SHIFT + sqrt_outshift
     + ((ODQ_CGAIN_SHIFT + ODQ_COMPAND_SHIFT) >> 1));
  }
  return odq_shr_round(odq_pow(odq_shr_round(cg0*q0, ODQ_CGAIN_SHIFT), (int16_t)beta),
   15 - ODQ_COMPAND_SHIFT);
}

/* Tail of od_pvq_compute_gain, src/pvq.c:841-846, given acc = sum x^2. */
__device__ __forceinline__ int32_t odq_gain_from_acc(int32_t acc, int q0, int beta,
 int bshift, int32_t *g) {
  int sqrt_shift;
  const int32_t irt = odq_sqrt(acc, &sqrt_shift);
  *g = odq_vshr_round(irt, sqrt_shift - bshift);
  return odq_gain_compand(*g, q0, (int16_t)beta);
}

/* od_pvq_compute_k, noref branch, src/pvq.c:912-931. */
__device__ __forceinline__ int odq_compute_k_noref(int32_t qcg, int n, int beta) {
  if (qcg == 0) return 0;
  if (n == 15 && qcg == 256 && beta > 5120) return 1;
  /* od_sqrt_table[1][OD_ILOG(n + 1)], src/pvq.c:908-910. */
  const int il = odq_ilog(n + 1);
  const int rt = il == 4 ? 2401 : il == 5 ? 3072 : il == 6 ? 4284 : il == 8 ? 8287
   : il == 10 ? 16432 : il == 12 ? 32767 : 0;
  const int32_t v = odq_shr_round((qcg - (int64_t)51)
   *odq_mult16_16_qbeta(odq_beta_rcp((int16_t)beta), rt), ODQ_CGAIN_SHIFT + 10);
  return v > 1 ? v : 1;
}

/* ---- with-reference (theta / Householder) arithmetic --------------------------- */

__device__ __forceinline__ int32_t odq_mult16_16(int32_t a, int32_t b) { /* OD_MULT16_16 */
  return (int32_t)(int16_t)a*(int32_t)(int16_t)b;
}

__device__ __forceinline__ int32_t odq_mult16_16_q16(int32_t a, int32_t b) {
  return ((int16_t)a*(int32_t)(int16_t)b) >> 16;
}

__device__ __forceinline__ int64_t odq_mult16_32_q16(int32_t a, int32_t b) {
  return (int16_t)a*(int64_t)b >> 16;
}

/* od_pvq_cos_pi_2, src/pvq.c:417-423. */
__device__ __forceinline__ int16_t odq_cos_pi_2(int16_t x) {
  const int16_t x2 = (int16_t)odq_mult16_16_q15(x, x);
  const int32_t v = (1073758164 - x*x + x2*(-7654 + odq_mult16_16_q16(x2, 16573
   + odq_mult16_16_q16(-2529, x2)))) >> 15;
  return (int16_t)(v < 32767 ? v : 32767);
}

/* od_pvq_cos, src/pvq.c:428-457 (angle in units of pi/2 / 2^15). */
__device__ __forceinline__ int odq_pvq_cos(int32_t x) {
  x &= 0x1ffff;
  if (x > (1 << 16)) x = (1 << 17) - x;
  if (x & 0x7fff) {
    if (x < (1 << 15)) return odq_cos_pi_2((int16_t)x);
    return (int16_t)-odq_cos_pi_2((int16_t)(65536 - x));
  }
  if (x & 0xffff) return 0;
  if (x & 0x1ffff) return -32767;
  return 32767;
}

/* od_pvq_sin, src/pvq.c:461-467. */
__device__ __forceinline__ int odq_pvq_sin(int32_t x) {
  return odq_pvq_cos(32768 - x);
}

/* od_pvq_compute_max_theta, src/pvq.c:855-865. */
__device__ __forceinline__ int odq_pvq_compute_max_theta(int32_t qcg, int beta) {
  int ts = odq_shr_round(qcg*odq_mult16_16_qbeta(402, odq_beta_rcp((int16_t)beta)),
   2*ODQ_CGAIN_SHIFT);
  if (qcg < 358) ts = 1;
  return ts;
}

/* od_pvq_compute_theta, src/pvq.c:874-884 (C truncating division). */
__device__ __forceinline__ int32_t odq_pvq_compute_theta(int t, int max_theta) {
  if (max_theta == 0) return 0;
  return (32768*(t < max_theta - 1 ? t : max_theta - 1) + (max_theta >> 1))/max_theta;
}

/* od_pvq_compute_k with a reference, nodesync (all-integer) branch,
   src/pvq.c:941-953; od_sqrt_table[0][OD_ILOG(n + 1)]. */
__device__ __forceinline__ int odq_compute_k_ref(int itheta, int n) {
  if (itheta == 0) return 0;
  const int il = odq_ilog(n + 1);
  const int rt = il == 4 ? 2290 : il == 5 ? 2985 : il == 6 ? 4222 : il == 8 ? 8256
   : il == 10 ? 16416 : il == 12 ? 32767 : 0;
  const int32_t v = odq_vshr_round((odq_shl32(itheta, 15) - 6554)*(int64_t)rt, 10 + 15);
  return v > 1 ? v : 1;
}

/* od_pvq_rate with speed > 0 (src/pvq_encoder.c:247-287: the closed form the reference's
   RDO pass prices with below complexity 5, src/encode.c:1359) from the candidate's
   "centre of mass" sum = SUM i*|y_i| over its n - (theta != -1) coded positions.
   Written once for both sides of the library: ODQ_RATE_LOG is the device's log in the
   choice kernels and the HOST libm's log where flagged bands are re-decided - the two
   may differ in the last place, which is why a choice whose costs come within
   ODQ_RATE_TOL of each other is never taken from the device (see k_choose). */
/* Two halves: the part that depends on the pulses alone (one value per SEARCH: candidates
   that share a pulse vector share it) and the theta terms of the candidate; the sum is
   formed in the reference's order, so pricing a candidate from a cached pulse part gives
   the very double odq_pvq_rate_fast returns. */
#define ODQ_RATE_PULSES_BODY(LOGFN, DIV) \
  if (k == 0) return 0; \
  const double f = DIV((double)sum, (double)(k*n)); \
  const double a = DIV(LOGFN(((double)(n*2))*(f + .025))*k, (double)n); \
  return ((1 + .4*f)*n)*(1.4426950408889634073599246810019*LOGFN(1 + (0 > a ? 0 : a))) + 3;

/* .9*log2(ts), :278 */
#define ODQ_RATE_TS_BODY(LOGFN) \
  return .9*(1.4426950408889634073599246810019*LOGFN((double)ts));

__device__ __forceinline__ double odq_pvq_rate_pulses(int sum, int k, int n) {
  ODQ_RATE_PULSES_BODY(log, __ddiv_rn)
}

__device__ __forceinline__ double odq_pvq_rate_ts(int ts) {
  ODQ_RATE_TS_BODY(log)
}

/* rate of a candidate from its two halves (ts_term is read only for a theta candidate
   with qg > 0) */
__host__ __device__ static inline double odq_pvq_rate_join(double pulses, double ts_term, int qg, int icgr,
 int theta, int is_keyframe, int pli) {
  double rate = pulses;
  if (qg > 0 && theta >= 0) {
    rate += ts_term;
    if (is_keyframe && pli == 0) rate += 6;
    if (qg == icgr) rate -= .5;
  }
  return rate;
}

__device__ __forceinline__ double odq_pvq_rate_fast(int sum, int k, int n, int qg, int icgr, int theta,
 int ts, int is_keyframe, int pli) {
  return odq_pvq_rate_join(odq_pvq_rate_pulses(sum, k, n), qg > 0 && theta >= 0 ? odq_pvq_rate_ts(ts) : 0.,
   qg, icgr, theta, is_keyframe, pli);
}

static inline double odq_host_div(double a, double b) { return a/b; }
static inline double odq_pvq_rate_pulses_host(int sum, int k, int n) {
  ODQ_RATE_PULSES_BODY(log, odq_host_div)
}
static inline double odq_pvq_rate_ts_host(int ts) {
  ODQ_RATE_TS_BODY(log)
}
static inline double odq_pvq_rate_fast_host(int sum, int k, int n, int qg, int icgr, int theta, int ts,
 int is_keyframe, int pli) {
  return odq_pvq_rate_join(odq_pvq_rate_pulses_host(sum, k, n),
   qg > 0 && theta >= 0 ? odq_pvq_rate_ts_host(ts) : 0., qg, icgr, theta, is_keyframe, pli);
}

/* |cost_a - cost_b| at or below this is "too close to call on the device". */
__host__ __device__ static inline double odq_rate_tol(double a, double b) {
  const double m = (a < 0 ? -a : a) + (b < 0 ? -b : b);
  return 1e-10*m + 1e-10;
}
