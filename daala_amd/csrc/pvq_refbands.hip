/* pvq_refbands.hip - pvq_theta WITH a reference (reference
   src/pvq_encoder.c:333-641: keyframe chroma predicted from luma, inter frames)
   for every block and band of a batch of coefficient planes, up to the
   rate-dependent choice, and the choice + synthesis that follows it.

     k_refb_prep_*      per band: chroma-from-luma sign flip of the block
                        (od_pvq_encode, :846-872), gather of x and r in coding
                        order (od_raster_to_coding_order, src/partition.c:144),
                        QM scaling, gains, correlation, initial distortion
                        (:381-455), od_compute_householder / od_apply_householder
                        (src/pvq.c:498-623), theta = floor(.5 + OD_THETA_SCALE *
                        acos(corr)) (:478) with the DEVICE acos and the
                        uncertainty test described in include/daala_hip.h
     k_refb_cands       the (gain, theta) candidates in the reference's stable
                        (k, gain) order (:466-504) followed by the no-reference
                        candidates (:571-581)
     k_refb_search      one band per lane: the candidate loop (:506-565) - pruning,
                        K-pulse searches on the reflected vector chained through
                        prev_k, distortions - then the no-reference loop
                        (:578-595)
     k_refb_choose      `cost < best_cost` / `cost <= best_cost` (:553, :600),
                        skip rules (:611-622), od_gain_expand and the band-wide
                        part of od_pvq_synthesis_partial (:623-633,
                        src/pvq.c:1037-1115)
     k_refb_synth       its per-coefficient part and od_coding_order_to_raster

   The libm call of the path: the reference's theta comes from glibc's acos.
   Everything downstream depends only on the INTEGER theta, so the device value
   is the reference's whenever OD_THETA_SCALE*acos(corr) + .5 is not within the
   margin (1e-9, 30x the worst disagreement of two <= 2-ulp implementations at
   this magnitude) of an integer; the bands inside the margin are listed and
   odhip_pvq_ref_resolve recomputes their theta on the host with the very libm
   the reference calls, re-running a band when it differs.

   Mappings: 15- and 8-coefficient bands one band per lane, 32- and
   128-coefficient bands one band per 16-lane DPP row (pvq_row.cuh).  The per-lane
   searches are not yet sorted by pulse count.  All double arithmetic
   is one IEEE operation per reference operation (-ffp-contract=off). */
#include "../../include/daala_hip.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "od_common.cuh"
#include "od_pvq_math.cuh"
#include "gen/od_scan_tables.h"
#include "pvq_search.cuh"
#include "pvq_row.cuh"

namespace {

constexpr int kMaxJobs = 16;
constexpr int kMaxItems = kMaxJobs*ODHIP_MAX_BANDS;
constexpr int kSlots = ODHIP_PVQ_REF_SLOTS;
constexpr int kUncCap = 1 << 16;
constexpr double kDefaultMargin = 1e-9;

struct RJob {
  const od_coeff *coef;
  const od_coeff *ref;
  const int16_t *qm;
  const int16_t *qm_inv;
  odhip_pvq_refband *rec;
  odhip_pvq_refitem *items;
  int16_t *y;
  int16_t *r16;
  int16_t *x16;
  int16_t *xr;
  const double *rate;
  int32_t *choice;
  od_coeff *dq;
  long nblocks;
  int nplanes;
  int w;
  int h;
  int bs;
  int nb_bands;
  int len;
  int bw;
  int bh;
  int is_keyframe;
  int pli;
  int q[ODHIP_MAX_BANDS];
  int beta[ODHIP_MAX_BANDS];
  int off[ODHIP_MAX_BANDS + 1];
};

struct RItems {
  int nitems;
  int perturb;
  double lambda;
  double margin;
  int wg_start[kMaxItems + 1];
  unsigned char job[kMaxItems];
  unsigned char band[kMaxItems];
};

/* One band whose theta lies inside the margin (written by k_refb_prep) or has
   to be re-run with the host's theta (read by the *_list kernels). */
struct Unc {
  int job;
  int band;
  unsigned blk;
  int theta;
  double corr;
};

__device__ RJob g_rjobs[kMaxJobs];
__constant__ unsigned char kRScanXY[OD_SCAN_LEN][2];
__device__ unsigned g_unc_count;
__device__ Unc g_unc[kUncCap];

__device__ __forceinline__ int find_item(const RItems &it, int wg) {
  int lo = 0;
  int hi = it.nitems - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (it.wg_start[mid] <= wg) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

/* Offset of block blk inside its plane set. */
__device__ __forceinline__ long block_base(const RJob &j, long blk) {
  const int N = 4 << j.bs;
  const long per = (long)j.bw*j.bh;
  const int p = (int)(blk/per);
  const int rem = (int)(blk - p*per);
  const int by = rem/j.bw;
  const int bx = rem - by*j.bw;
  return (long)p*j.w*j.h + (long)by*N*j.w + bx*N;
}

__device__ __forceinline__ long coef_pos(const RJob &j, int c) {
  return (long)kRScanXY[c][1]*j.w + kRScanXY[c][0];
}

constexpr double kPi = 3.14159265358979323846;       /* M_PI */
constexpr double kThetaScale = 32768*2./kPi;         /* OD_THETA_SCALE, src/pvq.h:78 */

/* .5 + OD_THETA_SCALE*acos(corr): the argument of the floor at
   src/pvq_encoder.c:478 (OD_ROUND32, src/odintrin.h:169). */
__device__ __forceinline__ double theta_arg(double corr) {
  return .5 + kThetaScale*acos(corr);
}

__global__ __launch_bounds__(kWave) void k_theta_probe(const double *corr, double *t, long n) {
  const long i = (long)blockIdx.x*kWave + threadIdx.x;
  if (i < n) t[i] = theta_arg(corr[i]);
}

/* ---- preparation -------------------------------------------------------------------
   Two mappings: the 15- and 8-coefficient bands one band per lane with the band
   in registers (k_refb_prep_lane), the 32- and 128-coefficient bands one band per
   16-lane row (k_refb_prep_row; a frame batch has too few of them to fill the
   chip one per lane, and a lane walking 128 coefficients three times is a chain
   of exposed latencies).  Band 0 of every block goes first and decides the
   chroma-from-luma flip of its block; the other bands read it from band 0's
   record.  Sums of integers are accumulated in any order (exact); the
   Householder pivot is the first largest |r| (src/pvq.c:505-512; |r16| < 2^15
   by construction of rshift, so the int16 running maximum of the reference
   cannot wrap). */
__device__ unsigned short gRScanPk[OD_SCAN_LEN];   /* y << 8 | x */

struct PrepScalars {
  int32_t g, gr, cg, cgr, gain_offset;
  int icgr;
  double corr, dist0;
  bool ran;
};

/* :404-455 from the band sums. */
__device__ __forceinline__ PrepScalars prep_scalars(int accx, int accr, double corr_sum, int xshift,
 int rshift, int q0, int beta, int cfl_enabled, int is_keyframe, int r_null) {
  PrepScalars o;
  o.cg = odq_gain_from_acc(accx, q0, beta, xshift, &o.g);
  o.cgr = odq_gain_from_acc(accr, q0, beta, rshift, &o.gr);
  if (cfl_enabled) o.cgr = 256;
  o.icgr = odq_shr_round(o.cgr, ODQ_CGAIN_SHIFT);
  o.gain_offset = o.cgr - odq_shl32(o.icgr, ODQ_CGAIN_SHIFT);
  double corr = __ddiv_rn(corr_sum,
   1e-100 + __ddiv_rn(o.g*(double)o.gr, (double)odq_shl32(1, xshift + rshift)));
  corr = corr < 1. ? corr : 1.;
  corr = corr > -1. ? corr : -1.;
  o.corr = corr;
  const double s2 = (1./256)*(1./256);
  double dist0 = ((1.4*o.cg)*o.cg)*s2;
  if (!is_keyframe && o.icgr == 0) {
    const int32_t scgr = o.gain_offset > 0 ? o.gain_offset : 0;
    dist0 = (1.4*(o.cg - scgr))*(o.cg - scgr) + (scgr*(double)o.cg)*(2 - 2*corr);
    dist0 *= s2;
  }
  o.dist0 = dist0;
  o.ran = !r_null && corr > 0;
  return o;
}

/* Householder projection constants from l2r = <r, r> and proj = <r, x>,
   src/pvq.c:573-590. */
__device__ __forceinline__ void householder_consts(int32_t l2r, int32_t proj, int16_t *proj_1,
 int *outshift) {
  const int l2r_shift = (odq_ilog(l2r) - 1) - 14;
  const int16_t l2r_norm = (int16_t)odq_vshr_round(l2r, l2r_shift);
  const int16_t rcp = odq_rcp(l2r_norm);
  const int proj_shift = (odq_ilog(abs(proj)) - 1) - 14;
  const int16_t proj_norm = (int16_t)odq_vshr_round(proj, proj_shift);
  *proj_1 = (int16_t)odq_mult16_16_q15(proj_norm, rcp);
  int os = 14 - proj_shift - 1 + l2r_shift;
  *outshift = os > 30 ? 30 : os;
}

/* Record of the band, theta with the device acos and the uncertainty list
   (one lane per band). */
__device__ __forceinline__ void prep_write(const RItems &it, odhip_pvq_refband *rec, int job, int band,
 long blk, int xshift, int rshift, const PrepScalars &p, int r_null, int flip, int m, int s) {
  int flags = (r_null ? ODHIP_REFBAND_R_NULL : 0) | (flip ? ODHIP_REFBAND_FLIP : 0);
  int32_t theta = 0;
  if (p.ran) {
    flags |= ODHIP_REFBAND_THETA;
    const double u = theta_arg(p.corr);
    theta = (int32_t)floor(u);
    if (fabs(u - rint(u)) < it.margin) {
      flags |= ODHIP_REFBAND_UNCERTAIN;
      if (it.perturb & 1) theta += 1;
      const unsigned slot = atomicAdd(&g_unc_count, 1u);
      if (slot < (unsigned)kUncCap) {
        Unc e;
        e.job = job;
        e.band = band;
        e.blk = (unsigned)blk;
        e.theta = theta;
        e.corr = p.corr;
        g_unc[slot] = e;
      }
    }
  }
  odhip_pvq_refband o;
  o.xshift = xshift;
  o.rshift = rshift;
  o.g = p.g;
  o.gr = p.gr;
  o.cg = p.cg;
  o.cgr = p.cgr;
  o.icgr = p.icgr;
  o.gain_offset = p.gain_offset;
  o.m = (int16_t)m;
  o.s = (int8_t)s;
  o.flags = (uint8_t)flags;
  o.theta = theta;
  o.nitems = 0;
  o.ntheta = 0;
  o.corr = p.corr;
  o.dist0 = p.dist0;
  *rec = o;
}

__device__ __forceinline__ uint32_t pack16(int lo, int hi) {
  return (uint32_t)(lo & 0xffff) | (uint32_t)hi << 16;
}

/* N = 15 (band 0: the flip is decided here) or N = 8. */
template <int N>
__global__ __launch_bounds__(kWave) void k_refb_prep_lane(RItems it) {
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const RJob &jb = g_rjobs[job];
  const int band = it.band[item];
  const long blk = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  if (blk >= jb.nblocks) return;
  const int off = jb.off[band];
  const int w = jb.w;
  const int len = jb.len;
  const int nb_bands = jb.nb_bands;
  const int q0 = jb.q[band];
  const int beta = jb.beta[band];
  const int is_keyframe = jb.is_keyframe;
  const int cfl_enabled = is_keyframe && jb.pli != 0;
  const long base = block_base(jb, blk);
  const od_coeff *x0 = jb.coef + base;
  const od_coeff *r0 = jb.ref + base;
  const int16_t *qmp = jb.qm + off;
  odhip_pvq_refband *rec = jb.rec + blk*nb_bands;
  int16_t *x16o = jb.x16 + blk*len;
  int16_t *r16o = jb.r16 + blk*len;
  int16_t *xro = jb.xr + blk*len;
  int xv[N];
  int rv[N];
  int qm[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    const long p = (long)kRScanXY[off + i][1]*w + kRScanXY[off + i][0];
    xv[i] = x0[p];
    rv[i] = r0[p];
    qm[i] = qmp[i];
  }
  int flip = 0;
  if (cfl_enabled) {
    if (N == 15 && band == 0) {
      /* src/pvq_encoder.c:846-872: OD_QM_SHIFT + OD_CFL_FLIP_SHIFT = 11 + 4, doubled */
      uint32_t xy = 0;
#pragma unroll
      for (int i = 0; i < N; i++) {
        const int32_t rq = (int32_t)((uint32_t)rv[i]*(uint32_t)qm[i]);
        const int32_t inq = (int32_t)((uint32_t)xv[i]*(uint32_t)qm[i]);
        xy += (uint32_t)((rq*(int64_t)inq) >> 30);
      }
      flip = (int32_t)xy < 0;
    }
    else flip = (rec[0].flags & ODHIP_REFBAND_FLIP) != 0;
  }
  /* od_vector_log_mag, src/pvq.c:472-484; src/pvq_encoder.c:381-385 */
  int sx = 0;
  int sr = 0;
  int r_null = 1;
#pragma unroll
  for (int i = 0; i < N; i++) {
    if (flip) rv[i] = -rv[i];
    const int tx = (int16_t)(xv[i] >> 8);
    const int tr = (int16_t)(rv[i] >> 8);
    sx += tx*tx;
    sr += tr*tr;
    if (rv[i]) r_null = 0;
  }
  int xshift = 8 + 1 + odq_ilog(N + sx)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int rshift = 8 + 1 + odq_ilog(N + sr)/2 - 14;
  rshift = rshift > 0 ? rshift : 0;
  int x16[N];
  int r16[N];
  double corr = 0;
  int accx = 0;
  int accr = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    x16[i] = (int16_t)odq_shr_round((int32_t)((uint32_t)xv[i]*(uint32_t)qm[i]), ODQ_QM_SHIFT + xshift);
    r16[i] = (int16_t)odq_shr_round((int32_t)((uint32_t)rv[i]*(uint32_t)qm[i]), ODQ_QM_SHIFT + rshift);
    corr += odq_mult16_16(x16[i], r16[i]);
    accx += x16[i]*x16[i];
    accr += r16[i]*r16[i];
  }
  const PrepScalars p = prep_scalars(accx, accr, corr, xshift, rshift, q0, beta, cfl_enabled,
   is_keyframe, r_null);
  int m = 0;
  int s = 1;
  int xr[N];
#pragma unroll
  for (int i = 0; i < N; i++) xr[i] = 0;
  if (p.ran) {
    /* od_compute_householder, src/pvq.c:498-521 */
    int maxr = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int a = abs(r16[i]);
      if (a > maxr) {
        maxr = a;
        m = i;
      }
    }
    int rm = 0;
#pragma unroll
    for (int i = 0; i < N; i++) if (i == m) rm = r16[i];
    s = rm > 0 ? 1 : -1;
    const int upd = (int16_t)(rm + odq_shr_round(p.gr*s, rshift));
    int32_t l2r = 0;
    int32_t proj = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (i == m) r16[i] = upd;
      l2r += odq_mult16_16(r16[i], r16[i]);
      proj += odq_mult16_16(r16[i], x16[i]);
    }
    int16_t proj_1;
    int outshift;
    householder_consts(l2r, proj, &proj_1, &outshift);
    int v[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      int32_t tmp = odq_mult16_16(r16[i], proj_1);
      tmp = outshift >= 0 ? odq_shr_round(tmp, outshift) : odq_shl32(tmp, -outshift);
      v[i] = (int16_t)(x16[i] - tmp);
    }
    /* the reflected vector without element m (src/pvq_encoder.c:481) */
#pragma unroll
    for (int i = 0; i < N - 1; i++) xr[i] = i < m ? v[i] : v[i + 1];
  }
  /* whole 16-byte vectors: the 15-coefficient band shares its first vector with
     the (unused) DC slot of the block */
  if (N == 15) {
    uint4 *xo = reinterpret_cast<uint4 *>(x16o);
    uint4 *ro = reinterpret_cast<uint4 *>(r16o);
    uint4 *xro4 = reinterpret_cast<uint4 *>(xro);
    xo[0] = make_uint4(pack16(0, x16[0]), pack16(x16[1], x16[2]), pack16(x16[3], x16[4]),
     pack16(x16[5], x16[6]));
    xo[1] = make_uint4(pack16(x16[7], x16[8]), pack16(x16[9], x16[10]), pack16(x16[11], x16[12]),
     pack16(x16[13], x16[N - 1]));
    ro[0] = make_uint4(pack16(0, r16[0]), pack16(r16[1], r16[2]), pack16(r16[3], r16[4]),
     pack16(r16[5], r16[6]));
    ro[1] = make_uint4(pack16(r16[7], r16[8]), pack16(r16[9], r16[10]), pack16(r16[11], r16[12]),
     pack16(r16[13], r16[N - 1]));
    xro4[0] = make_uint4(pack16(0, xr[0]), pack16(xr[1], xr[2]), pack16(xr[3], xr[4]),
     pack16(xr[5], xr[6]));
    xro4[1] = make_uint4(pack16(xr[7], xr[8]), pack16(xr[9], xr[10]), pack16(xr[11], xr[12]),
     pack16(xr[13], 0));
  }
  else {
    *reinterpret_cast<uint4 *>(x16o + off) = make_uint4(pack16(x16[0], x16[1]), pack16(x16[2], x16[3]),
     pack16(x16[4], x16[5]), pack16(x16[6], x16[7]));
    *reinterpret_cast<uint4 *>(r16o + off) = make_uint4(pack16(r16[0], r16[1]), pack16(r16[2], r16[3]),
     pack16(r16[4], r16[5]), pack16(r16[6], r16[7]));
    *reinterpret_cast<uint4 *>(xro + off) = make_uint4(pack16(xr[0], xr[1]), pack16(xr[2], xr[3]),
     pack16(xr[4], xr[5]), pack16(xr[6], 0));
  }
  prep_write(it, rec + band, job, band, blk, xshift, rshift, p, r_null, flip, m, s);
}

/* n = 16*E coefficients per 16-lane row, four bands per wavefront; lane l of the
   row owns coding positions l*E .. l*E+E-1 of the band. */
template <int E>
__global__ __launch_bounds__(kWave) void k_refb_prep_row(RItems it) {
  constexpr int n = 16*E;
  __shared__ unsigned short s_scan[n];
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const RJob &jb = g_rjobs[job];
  const int band = it.band[item];
  const int off = jb.off[band];
  const int lane = threadIdx.x;
  for (int j = lane; j < n; j += kWave) s_scan[j] = gRScanPk[off + j];
  __syncthreads();
  const int row = lane >> 4;
  const int l = lane & 15;
  const long nblocks = jb.nblocks;
  const long blk0 = (long)(blockIdx.x - it.wg_start[item])*4 + row;
  const bool live = blk0 < nblocks;
  const long blk = live ? blk0 : nblocks - 1;
  const int w = jb.w;
  const int len = jb.len;
  const int nb_bands = jb.nb_bands;
  const int q0 = jb.q[band];
  const int beta = jb.beta[band];
  const int is_keyframe = jb.is_keyframe;
  const int cfl_enabled = is_keyframe && jb.pli != 0;
  const long base = block_base(jb, blk);
  const od_coeff *x0 = jb.coef + base;
  const od_coeff *r0 = jb.ref + base;
  const int16_t *qmp = jb.qm + off + l*E;
  odhip_pvq_refband *rec = jb.rec + blk*nb_bands;
  int16_t *x16o = jb.x16 + blk*len + off;
  int16_t *r16o = jb.r16 + blk*len + off;
  int16_t *xro = jb.xr + blk*len + off;
  const int flip = cfl_enabled ? (rec[0].flags & ODHIP_REFBAND_FLIP) != 0 : 0;
  int xv[E];
  int rv[E];
  int qm[E];
  int sx = 0;
  int sr = 0;
  int nz = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int pk = s_scan[l*E + e];
    const long p = (long)(pk >> 8)*w + (pk & 255);
    xv[e] = x0[p];
    const int r = r0[p];
    rv[e] = flip ? -r : r;
    qm[e] = qmp[e];
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int tx = (int16_t)(xv[e] >> 8);
    const int tr = (int16_t)(rv[e] >> 8);
    sx += tx*tx;
    sr += tr*tr;
    nz |= rv[e] != 0;
  }
  sx = row_sum(sx);
  sr = row_sum(sr);
  const int r_null = row_max(nz) == 0;
  int xshift = 8 + 1 + odq_ilog(n + sx)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int rshift = 8 + 1 + odq_ilog(n + sr)/2 - 14;
  rshift = rshift > 0 ? rshift : 0;
  int x16[E];
  int r16[E];
  double corr = 0;
  int accx = 0;
  int accr = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    x16[e] = (int16_t)odq_shr_round((int32_t)((uint32_t)xv[e]*(uint32_t)qm[e]), ODQ_QM_SHIFT + xshift);
    r16[e] = (int16_t)odq_shr_round((int32_t)((uint32_t)rv[e]*(uint32_t)qm[e]), ODQ_QM_SHIFT + rshift);
    corr += odq_mult16_16(x16[e], r16[e]);
    accx += x16[e]*x16[e];
    accr += r16[e]*r16[e];
  }
  corr = row_sum(corr);
  accx = row_sum(accx);
  accr = row_sum(accr);
  const PrepScalars p = prep_scalars(accx, accr, corr, xshift, rshift, q0, beta, cfl_enabled,
   is_keyframe, r_null);
  int m = 0;
  int s = 1;
  if (p.ran) {
    /* od_compute_householder: (largest |r|, lowest index) over the row */
    int ba = -1;
    int bi = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int a = abs(r16[e]);
      if (a > ba) {
        ba = a;
        bi = l*E + e;
      }
    }
#define OD_ARGMAX_STEP(CTRL) \
    { \
      const int oa = row_mov<CTRL>(ba); \
      const int oi = row_mov<CTRL>(bi); \
      const bool take = oa > ba || (oa == ba && oi < bi); \
      ba = take ? oa : ba; \
      bi = take ? oi : bi; \
    }
    OD_ARGMAX_STEP(OD_DPP_XOR1)
    OD_ARGMAX_STEP(OD_DPP_XOR2)
    OD_ARGMAX_STEP(OD_DPP_HALF_MIRROR)
    OD_ARGMAX_STEP(OD_DPP_MIRROR)
#undef OD_ARGMAX_STEP
    m = bi;
    int rm = 0;
#pragma unroll
    for (int e = 0; e < E; e++) if (l*E + e == m) rm = r16[e];
    rm = row_sum(rm);
    s = rm > 0 ? 1 : -1;
    const int upd = (int16_t)(rm + odq_shr_round(p.gr*s, rshift));
    int32_t l2r = 0;
    int32_t proj = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (l*E + e == m) r16[e] = upd;
      l2r += odq_mult16_16(r16[e], r16[e]);
      proj += odq_mult16_16(r16[e], x16[e]);
    }
    l2r = row_sum(l2r);
    proj = row_sum(proj);
    int16_t proj_1;
    int outshift;
    householder_consts(l2r, proj, &proj_1, &outshift);
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = l*E + e;
      int32_t tmp = odq_mult16_16(r16[e], proj_1);
      tmp = outshift >= 0 ? odq_shr_round(tmp, outshift) : odq_shl32(tmp, -outshift);
      const int16_t v = (int16_t)(x16[e] - tmp);
      /* the reflected vector without element m (src/pvq_encoder.c:481) */
      if (live && i != m) xro[i - (i > m)] = v;
    }
  }
  if (live) {
    if constexpr (E == 8) {
      *reinterpret_cast<uint4 *>(x16o + l*E) = make_uint4(pack16(x16[0], x16[1]),
       pack16(x16[2], x16[3]), pack16(x16[4], x16[5]), pack16(x16[6], x16[7]));
      *reinterpret_cast<uint4 *>(r16o + l*E) = make_uint4(pack16(r16[0], r16[1]),
       pack16(r16[2], r16[3]), pack16(r16[4], r16[5]), pack16(r16[6], r16[7]));
    }
    else {
#pragma unroll
      for (int e = 0; e < E; e += 2) {
        *reinterpret_cast<uint32_t *>(x16o + l*E + e) = pack16(x16[e], x16[e + 1]);
        *reinterpret_cast<uint32_t *>(r16o + l*E + e) = pack16(r16[e], r16[e + 1]);
      }
    }
    if (l == 0) prep_write(it, rec + band, job, band, blk, xshift, rshift, p, r_null, flip, m, s);
  }
}


/* ---- candidate lists -------------------------------------------------------------- */
__device__ __forceinline__ void refb_candidates(const RJob &jb, int band, long blk,
 int theta_override) {
  odhip_pvq_refband *rp = jb.rec + blk*jb.nb_bands + band;
  odhip_pvq_refband r = *rp;
  if (theta_override >= 0) r.theta = theta_override;
  const int n = jb.off[band + 1] - jb.off[band];
  const int beta = jb.beta[band];
  odhip_pvq_refitem *items = jb.items + (blk*jb.nb_bands + band)*kSlots;
  int nitems = 0;
  if (r.flags & ODHIP_REFBAND_THETA) {
    const int gain_bound = (r.cg - r.gain_offset) >> ODQ_CGAIN_SHIFT;
    const double scale_1 = __ddiv_rn(1., kThetaScale);   /* OD_THETA_SCALE_1 */
    for (int i = gain_bound - 1 > 1 ? gain_bound - 1 : 1; i <= gain_bound + 1; i++) {
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT) + r.gain_offset;
      const int ts = odq_pvq_compute_max_theta(qcg, beta);
      /* same left-to-right products as src/pvq_encoder.c:482-484 */
      const double t = __ddiv_rn(((r.theta*scale_1)*2), kPi)*ts;
      int lower = (int)floor(.5 + t) - 2;
      if (lower < 0) lower = 0;
      int upper = (int)ceil(t);
      if (upper > ts - 1) upper = ts - 1;
      for (int j = lower; j <= upper && nitems < kSlots - 2; j++) {
        odhip_pvq_refitem c;
        c.gain = i;
        c.theta = j;
        c.ts = ts;
        c.k = odq_compute_k_ref(j, n);
        c.qcg = qcg;
        c.qtheta = odq_pvq_compute_theta(j, ts);
        c.flags = ODHIP_REFITEM_WITH_REF;
        c.yslot = -1;
        c.cos_dist = 0;
        c.dist = 0;
        /* stable insertion by (k, gain): items_compare, src/pvq_encoder.c:301-305
           (glibc's qsort is a stable merge sort at this size) */
        int pos = nitems;
        while (pos > 0) {
          const odhip_pvq_refitem q = items[pos - 1];
          const int cmp = q.k == c.k ? q.gain - c.gain : q.k - c.k;
          if (cmp <= 0) break;
          items[pos] = q;
          pos--;
        }
        items[pos] = c;
        nitems++;
      }
    }
  }
  const int ntheta = nitems;
  int flags = r.flags & ~ODHIP_REFBAND_NOREF;
  /* src/pvq_encoder.c:571-581 */
  if ((jb.is_keyframe && jb.pli == 0) || r.corr < .5 || r.cg < odq_shl32(2, ODQ_CGAIN_SHIFT)) {
    flags |= ODHIP_REFBAND_NOREF;
    const int gain_bound = r.cg >> ODQ_CGAIN_SHIFT;
    for (int i = gain_bound > 1 ? gain_bound : 1; i <= gain_bound + 1; i++) {
      odhip_pvq_refitem c;
      c.gain = i;
      c.theta = -1;
      c.ts = 0;
      c.qcg = odq_shl32(i, ODQ_CGAIN_SHIFT);
      c.k = odq_compute_k_noref(c.qcg, n, beta);
      c.qtheta = 0;
      c.flags = 0;
      c.yslot = -1;
      c.cos_dist = 0;
      c.dist = 0;
      items[nitems++] = c;
    }
  }
  rp->theta = r.theta;
  rp->flags = (uint8_t)flags;
  rp->nitems = nitems;
  rp->ntheta = ntheta;
}

__global__ __launch_bounds__(kWave) void k_refb_cands(RItems it) {
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = g_rjobs[it.job[item]];
  const long blk = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  if (blk >= jb.nblocks) return;
  refb_candidates(jb, it.band[item], blk, -1);
}

__global__ __launch_bounds__(kWave) void k_refb_cands_list(const Unc *list, int count) {
  const int i = blockIdx.x*kWave + threadIdx.x;
  if (i >= count) return;
  const Unc e = list[i];
  refb_candidates(g_rjobs[e.job], e.band, e.blk, e.theta);
}

/* ---- the candidate loops ---------------------------------------------------------- */
__device__ __forceinline__ void store_pulses(int16_t *dst, const short *xs, const unsigned short *ys,
 int lane, int n) {
  for (int j = 0; j < n; j++) {
    const int yj = ys[j*kWave + lane];
    dst[j] = (int16_t)(xs[j*kWave + lane] < 0 ? -yj : yj);
  }
}

__device__ __forceinline__ void refb_search(const RJob &jb, int band, long blk, int lane, short *xs,
 unsigned short *ys, double lambda) {
  const odhip_pvq_refband r = jb.rec[blk*jb.nb_bands + band];
  const int off = jb.off[band];
  const int n = jb.off[band + 1] - off;
  odhip_pvq_refitem *items = jb.items + (blk*jb.nb_bands + band)*kSlots;
  const double s2 = (1./256)*(1./256);   /* OD_CGAIN_SCALE_2 */
  const double t1 = 1./32768;            /* OD_TRIG_SCALE_1 */
  const int32_t cg = r.cg;
  const double dist0 = r.dist0;
  if (r.ntheta > 0) {
    const int16_t *xr = jb.xr + blk*jb.len + off;
    for (int j = 0; j < n - 1; j++) xs[j*kWave + lane] = xr[j];
    int prev_k = 0;
    int cur_slot = -1;
    double cos_dist = 0;
    const int32_t theta = r.theta;
    for (int idx = 0; idx < r.ntheta; idx++) {
      odhip_pvq_refitem *ip = items + idx;
      const int32_t qcg = ip->qcg;
      const int32_t qtheta = ip->qtheta;
      const int k = ip->k;
      /* src/pvq_encoder.c:526-531 */
      double dist_theta = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1;
      double dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      if (dist > dist0 + 1.0*lambda && k != 0) {
        ip->flags = ODHIP_REFITEM_WITH_REF;
        ip->yslot = -1;
        continue;
      }
      const double sin_prod = ((odq_pvq_sin(theta)*t1)*odq_pvq_sin(qtheta))*t1;
      if (k == 0) {
        cos_dist = 0;
        cur_slot = -1;
      }
      else if (k != prev_k) {
        double yy;
        cos_dist = od_pvq_search_lane(xs, ys, lane, n - 1, k, prev_k,
         ((qcg*(double)cg)*sin_prod)*s2, lambda, &yy);
        cur_slot = idx;
        store_pulses(jb.y + ((long)idx*jb.nblocks + blk)*jb.len + off, xs, ys, lane, n - 1);
      }
      prev_k = k;
      /* :548-552 */
      dist_theta = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1 + sin_prod*(2 - 2*cos_dist);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      ip->flags = ODHIP_REFITEM_WITH_REF | ODHIP_REFITEM_SEARCHED;
      ip->yslot = cur_slot;
      ip->cos_dist = cos_dist;
      ip->dist = dist;
    }
  }
  if (r.nitems > r.ntheta) {
    const int16_t *x16 = jb.x16 + blk*jb.len + off;
    for (int j = 0; j < n; j++) xs[j*kWave + lane] = x16[j];
    int prev_k = 0;
    for (int idx = r.ntheta; idx < r.nitems; idx++) {
      odhip_pvq_refitem *ip = items + idx;
      const int32_t qcg = ip->qcg;
      const int k = ip->k;
      /* :585-595 */
      double dist = (1.4*(qcg - cg))*(qcg - cg);
      dist *= s2;
      if (dist > dist0 && k != 0) {
        ip->flags = 0;
        ip->yslot = -1;
        continue;
      }
      double yy;
      const double cos_dist = od_pvq_search_lane(xs, ys, lane, n, k, prev_k, (qcg*(double)cg)*s2,
       lambda, &yy);
      prev_k = k;
      store_pulses(jb.y + ((long)idx*jb.nblocks + blk)*jb.len + off, xs, ys, lane, n);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*cos_dist);
      dist *= s2;
      ip->flags = ODHIP_REFITEM_SEARCHED;
      ip->yslot = idx;
      ip->cos_dist = cos_dist;
      ip->dist = dist;
    }
  }
}

__global__ __launch_bounds__(kWave) void k_refb_search(RItems it) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  od_rsqrt_init(threadIdx.x);
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = g_rjobs[it.job[item]];
  const int band = it.band[item];
  const int n = jb.off[band + 1] - jb.off[band];
  short *xs = (short *)lds;
  unsigned short *ys = lds + n*kWave;
  const long blk = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  if (blk >= jb.nblocks) return;
  refb_search(jb, band, blk, threadIdx.x, xs, ys, it.lambda);
}

/* The same candidate loops with one band per 16-lane row (pvq_row.cuh): bands of
   16*E coefficients (E = 2: 32, E = 8: 128), all state in registers.  The
   decisions of a row are uniform over its lanes (every lane evaluates them on
   the same record and items); lane 0 of the row writes the item fields, every
   lane its E pulses as one vector store. */
template <int E>
__device__ __forceinline__ void store_row_pulses(int16_t *dst, const int (&sg)[E], const int (&y)[E]) {
  uint32_t w[E/2];
#pragma unroll
  for (int e = 0; e < E; e += 2) {
    const int lo = sg[e] ? -y[e] : y[e];
    const int hi = sg[e + 1] ? -y[e + 1] : y[e + 1];
    w[e/2] = (uint32_t)(lo & 0xffff) | (uint32_t)hi << 16;
  }
  if constexpr (E == 8) *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
  else {
#pragma unroll
    for (int e = 0; e < E/2; e++) reinterpret_cast<uint32_t *>(dst)[e] = w[e];
  }
}

template <int E>
__device__ __forceinline__ void load_row_vector(const int16_t *src, int (&ax)[E], int (&sg)[E]) {
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int v = src[e];
    ax[e] = abs(v);
    sg[e] = v < 0;
  }
}

template <int E>
__global__ __launch_bounds__(kWave) void k_refb_search_row(RItems it) {
  constexpr int n = 16*E;
  od_rsqrt_init(threadIdx.x);
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = g_rjobs[it.job[item]];
  const int band = it.band[item];
  const int off = jb.off[band];
  const int lane = threadIdx.x;
  const int row = lane >> 4;
  const int l = lane & 15;
  const long nblocks = jb.nblocks;
  const int len = jb.len;
  const int nb_bands = jb.nb_bands;
  int16_t *const yout = jb.y;
  const long blk0 = (long)(blockIdx.x - it.wg_start[item])*4 + row;
  const bool live = blk0 < nblocks;
  const long blk = live ? blk0 : nblocks - 1;
  const bool writer = live && l == 0;
  const odhip_pvq_refband r = jb.rec[blk*nb_bands + band];
  odhip_pvq_refitem *items = jb.items + (blk*nb_bands + band)*kSlots;
  const double lambda = it.lambda;
  const int force = it.perturb >> 1;
  const double s2 = (1./256)*(1./256);   /* OD_CGAIN_SCALE_2 */
  const double t1 = 1./32768;            /* OD_TRIG_SCALE_1 */
  const int32_t cg = r.cg;
  const double dist0 = r.dist0;
  int ax[E];
  int sg[E];
  int y[E];
  if (r.ntheta > 0) {
    load_row_vector<E>(jb.xr + blk*len + off + l*E, ax, sg);
    if (l == 15) {        /* the pad: xr holds n - 1 values */
      ax[E - 1] = 0;
      sg[E - 1] = 0;
    }
#pragma unroll
    for (int e = 0; e < E; e++) y[e] = 0;
    int prev_k = 0;
    int cur_slot = -1;
    double cos_dist = 0;
    const int32_t theta = r.theta;
    for (int idx = 0; idx < r.ntheta; idx++) {
      odhip_pvq_refitem *ip = items + idx;
      const int32_t qcg = ip->qcg;
      const int32_t qtheta = ip->qtheta;
      const int k = ip->k;
      double dist_theta = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1;
      double dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      if (dist > dist0 + 1.0*lambda && k != 0) {
        if (writer) {
          ip->flags = ODHIP_REFITEM_WITH_REF;
          ip->yslot = -1;
        }
        continue;
      }
      const double sin_prod = ((odq_pvq_sin(theta)*t1)*odq_pvq_sin(qtheta))*t1;
      if (k == 0) {
        cos_dist = 0;
        cur_slot = -1;
      }
      else if (k != prev_k) {
        double yy;
        cos_dist = od_pvq_search_row<E>(ax, y, row, l, n - 1, k, prev_k,
         ((qcg*(double)cg)*sin_prod)*s2, lambda, force, &yy);
        cur_slot = idx;
        if (live) store_row_pulses<E>(yout + ((long)idx*nblocks + blk)*len + off + l*E, sg, y);
      }
      prev_k = k;
      dist_theta = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1 + sin_prod*(2 - 2*cos_dist);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      if (writer) {
        ip->flags = ODHIP_REFITEM_WITH_REF | ODHIP_REFITEM_SEARCHED;
        ip->yslot = cur_slot;
        ip->cos_dist = cos_dist;
        ip->dist = dist;
      }
    }
  }
  if (r.nitems > r.ntheta) {
    load_row_vector<E>(jb.x16 + blk*len + off + l*E, ax, sg);
    int prev_k = 0;
    for (int idx = r.ntheta; idx < r.nitems; idx++) {
      odhip_pvq_refitem *ip = items + idx;
      const int32_t qcg = ip->qcg;
      const int k = ip->k;
      double dist = (1.4*(qcg - cg))*(qcg - cg);
      dist *= s2;
      if (dist > dist0 && k != 0) {
        if (writer) {
          ip->flags = 0;
          ip->yslot = -1;
        }
        continue;
      }
      double yy;
      const double cos_dist = od_pvq_search_row<E>(ax, y, row, l, n, k, prev_k, (qcg*(double)cg)*s2,
       lambda, force, &yy);
      prev_k = k;
      if (live) store_row_pulses<E>(yout + ((long)idx*nblocks + blk)*len + off + l*E, sg, y);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*cos_dist);
      dist *= s2;
      if (writer) {
        ip->flags = ODHIP_REFITEM_SEARCHED;
        ip->yslot = idx;
        ip->cos_dist = cos_dist;
        ip->dist = dist;
      }
    }
  }
}

/* One listed band per wavefront (lane 0): the list is a handful of bands. */
__global__ __launch_bounds__(kWave) void k_refb_search_list(const Unc *list, int count, double lambda) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  od_rsqrt_init(threadIdx.x);
  if (threadIdx.x != 0 || (int)blockIdx.x >= count) return;
  const Unc e = list[blockIdx.x];
  short *xs = (short *)lds;
  unsigned short *ys = lds + 128*kWave;
  refb_search(g_rjobs[e.job], e.band, e.blk, 0, xs, ys, lambda);
}

/* ---- choice + synthesis -----------------------------------------------------------
   k_refb_choose<N>  one band per lane: the reference's selection among the
                     candidates, the skip rules, and everything of
                     od_pvq_synthesis_partial that is a property of the whole
                     band (sum of squared pulses -> scale, the Householder
                     projection of the synthesised vector), with the band's
                     pulses and reference held as packed 16-byte vectors
   k_refb_synth      one coefficient per thread in coding order: the
                     per-coefficient part (scale, reflection, inverse QM) and
                     od_coding_order_to_raster

   choice[(blk*nb + band)*16 ..]: [0..7] as documented in include/daala_hip.h;
   [8] mode (0 zero, 1 copy the reference, 4 copy the negated reference, 2
   no-reference synthesis, 3 with reference), [9] pulse slot, [10] scale, [11]
   qshift, [12] xm, [13] m, [14] proj_1, [15] outshift. */
__device__ __forceinline__ int neg_interleave(int x, int ref) { /* src/pvq_encoder.c:235-239 */
  if (x < ref) return -2*(x - ref) - 1;
  if (x < 2*ref) return 2*(x - ref);
  return x - 1;
}

__device__ unsigned char gRBandOf[OD_SCAN_LEN];

template <int N>
__global__ __launch_bounds__(kWave) void k_refb_choose(RItems it) {
  constexpr int SH = N == 15 ? 1 : 0;     /* the 15-coefficient band is read from the DC slot on */
  constexpr int NW = (N + SH)/2;
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = g_rjobs[it.job[item]];
  const int band = it.band[item];
  const long blk = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  if (blk >= jb.nblocks) return;
  const long bi = blk*jb.nb_bands + band;
  const odhip_pvq_refband r = jb.rec[bi];
  const odhip_pvq_refitem *items = jb.items + bi*kSlots;
  const double *rate = jb.rate ? jb.rate + bi*(kSlots + 1) : nullptr;
  const double lambda = it.lambda;
  const int off = jb.off[band];
  const int cfl_enabled = jb.is_keyframe && jb.pli != 0;
  /* :417-455 */
  double best_cost = r.dist0 + lambda*(rate ? rate[0] : 0.);
  int qg = 0;
  int noref = jb.is_keyframe ? 1 : 0;
  int itheta = jb.is_keyframe ? -1 : 0;
  int max_theta = 0;
  int best_k = 0;
  int32_t best_qtheta = 0;
  int chosen = -1;
  int yslot = -1;
  for (int idx = 0; idx < r.nitems; idx++) {
    const int4 head = *reinterpret_cast<const int4 *>(&items[idx].gain);   /* gain, theta, ts, k */
    const int4 tail = *reinterpret_cast<const int4 *>(&items[idx].qcg);    /* qcg, qtheta, flags, yslot */
    if (!(tail.z & ODHIP_REFITEM_SEARCHED)) continue;
    const double cost = items[idx].dist + lambda*(rate ? rate[1 + idx] : 0.);
    if (idx < r.ntheta ? cost < best_cost : cost <= best_cost) {
      best_cost = cost;
      qg = head.x;
      best_k = head.w;
      chosen = idx;
      yslot = tail.w;
      if (idx < r.ntheta) {
        best_qtheta = tail.y;
        itheta = head.y;
        max_theta = head.z;
        noref = 0;
      }
      else {
        noref = 1;
        itheta = -1;
        max_theta = 0;
      }
    }
  }
  /* :611-622 */
  int skip = 0;
  if (noref) {
    if (qg == 0) skip = 1;
  }
  else {
    if (!jb.is_keyframe && qg == 0) skip = r.icgr ? 1 : 2;
    if (qg == r.icgr && itheta == 0 && !cfl_enabled) skip = 2;
  }
  int4 *ch = reinterpret_cast<int4 *>(jb.choice + bi*16);
  ch[0] = make_int4(chosen, qg, noref, itheta);
  ch[1] = make_int4(max_theta, best_k, skip, jb.is_keyframe ? (noref ? qg : neg_interleave(qg, r.icgr))
   : (noref ? qg - 1 : neg_interleave(qg + 1, r.icgr + 1)));
  if (skip) {
    const int flip = (r.flags & ODHIP_REFBAND_FLIP) != 0;
    ch[2] = make_int4(skip == 2 ? (flip ? 4 : 1) : 0, -1, 0, 0);
    ch[3] = make_int4(0, 0, 0, 0);
    return;
  }
  /* od_gain_expand + od_pvq_synthesis_partial (band-wide part), :623-633,
     src/pvq.c:1037-1115 */
  const int32_t g = odq_gain_expand(odq_shl32(qg, ODQ_CGAIN_SHIFT) + (noref ? 0 : r.gain_offset),
   jb.q[band], jb.beta[band]);
  uint32_t yw[NW];
  if (yslot >= 0) {
    const uint4 *yp = reinterpret_cast<const uint4 *>(jb.y + ((long)yslot*jb.nblocks + blk)*jb.len
     + off - SH);
    if constexpr (NW >= 4) {
#pragma unroll
      for (int v = 0; v < NW/4; v++) {
        const uint4 q = yp[v];
        yw[4*v] = q.x;
        yw[4*v + 1] = q.y;
        yw[4*v + 2] = q.z;
        yw[4*v + 3] = q.w;
      }
    }
  }
  else {
#pragma unroll
    for (int v = 0; v < NW; v++) yw[v] = 0;
  }
  auto yget = [&](int i) -> int { return (int16_t)(yw[(i + SH) >> 1] >> (16*((i + SH) & 1))); };
  const int nn = N - (!noref);
  int yy = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int v = i < nn ? yget(i) : 0;
    yy += v*v;
  }
  int gshift = odq_ilog(g) - 14;
  gshift = gshift > 0 ? gshift : 0;
  int32_t scale = 0;
  if (yy != 0) {
    int rsqrt_shift;
    const int16_t rsqrt = odq_rsqrt(yy, &rsqrt_shift);
    scale = odq_vshr_round(rsqrt*(int64_t)g, rsqrt_shift + gshift - 16);
  }
  const int qshift = ODQ_QM_INV_SHIFT - gshift;
  if (noref) {
    ch[2] = make_int4(2, yslot, scale, qshift);
    ch[3] = make_int4(0, 0, 0, 0);
    return;
  }
  const int m = r.m;
  const int s = r.s;
  /* src/pvq.c:1094-1114: the two double products by 2^-15 are exact */
  scale = (int32_t)floor(.5 + (scale*(1./32768))*odq_pvq_sin(best_qtheta));
  const int16_t xm = (int16_t)floor(.5 + ((-s*odq_shr_round(g, gshift))*(1./32768))*odq_pvq_cos(best_qtheta));
  uint32_t rw[NW];
  {
    const uint4 *rp = reinterpret_cast<const uint4 *>(jb.r16 + blk*jb.len + off - SH);
#pragma unroll
    for (int v = 0; v < NW/4; v++) {
      const uint4 q = rp[v];
      rw[4*v] = q.x;
      rw[4*v + 1] = q.y;
      rw[4*v + 2] = q.z;
      rw[4*v + 3] = q.w;
    }
  }
  int32_t l2r = 0;
  int32_t proj = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int ri = (int16_t)(rw[(i + SH) >> 1] >> (16*((i + SH) & 1)));
    const int ysrc = i == 0 ? yget(0) : (i < m ? yget(i < N - 1 ? i : N - 2) : yget(i - 1));
    const int16_t xi = i == m ? xm : (int16_t)odq_mult16_32_q16(ysrc, scale);
    l2r += odq_mult16_16(ri, ri);
    proj += odq_mult16_16(ri, xi);
  }
  int16_t proj_1;
  int outshift;
  householder_consts(l2r, proj, &proj_1, &outshift);
  ch[2] = make_int4(3, yslot, scale, qshift);
  ch[3] = make_int4(xm, m, proj_1, outshift);
}

__global__ __launch_bounds__(256) void k_refb_synth(RItems it) {
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = g_rjobs[it.job[item]];
  const int len = jb.len;
  const long t = (long)(blockIdx.x - it.wg_start[item])*256 + threadIdx.x;
  const long blk = t/len;
  if (blk >= jb.nblocks) return;
  const int c = (int)(t - blk*len);
  const long base = block_base(jb, blk);
  od_coeff *out = jb.dq + base;
  if (c == 0) {
    out[0] = jb.coef[base];
    return;
  }
  const int band = gRBandOf[c];
  const int off = jb.off[band];
  const int i = c - off;
  const int pk = gRScanPk[c];
  const long p = (long)(pk >> 8)*jb.w + (pk & 255);
  const int4 *ch = reinterpret_cast<const int4 *>(jb.choice + (blk*jb.nb_bands + band)*16);
  const int4 a = ch[2];
  const int mode = a.x;
  if (mode == 0) {
    out[p] = 0;
    return;
  }
  if (mode == 1 || mode == 4) {
    const od_coeff rv = jb.ref[base + p];
    out[p] = mode == 4 ? -rv : rv;
    return;
  }
  const int yslot = a.y;
  const int32_t scale = a.z;
  const int qshift = a.w;
  const int qmi = jb.qm_inv[c];
  const int16_t *yp = jb.y + ((long)yslot*jb.nblocks + blk)*len + off;
  if (mode == 2) {
    const int32_t x = (int32_t)odq_mult16_32_q16(yslot >= 0 ? yp[i] : 0, scale);
    out[p] = odq_shr_round(x*qmi, qshift);
    return;
  }
  const int4 b = ch[3];
  const int m = b.y;
  const int16_t xi = i == m ? (int16_t)b.x
   : (int16_t)odq_mult16_32_q16(yslot >= 0 ? yp[i < m ? i : i - 1] : 0, scale);
  int32_t tmp = odq_mult16_16(jb.r16[blk*len + c], b.z);
  tmp = b.w >= 0 ? odq_shr_round(tmp, b.w) : odq_shl32(tmp, -b.w);
  const int16_t v = (int16_t)(xi - tmp);
  out[p] = odq_shr_round(v*qmi, qshift);
}


/* ---- host side --------------------------------------------------------------------- */
bool g_tables_uploaded = false;
double g_margin = kDefaultMargin;
int g_perturb = 0;

int upload_tables(void) {
  if (g_tables_uploaded) return ODHIP_SUCCESS;
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kRScanXY), OD_SCAN_XY, sizeof(OD_SCAN_XY)));
  unsigned short packed[OD_SCAN_LEN];
  for (int j = 0; j < OD_SCAN_LEN; j++) packed[j] = (unsigned short)(OD_SCAN_XY[j][1] << 8 | OD_SCAN_XY[j][0]);
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gRScanPk), packed, sizeof(packed)));
  unsigned char band_of[OD_SCAN_LEN];
  for (int j = 0; j < OD_SCAN_LEN; j++) {
    int b = 0;
    while (b + 1 < OD_NBANDS[4] && j >= OD_BAND_OFFS[4][b + 1]) b++;
    band_of[j] = (unsigned char)b;
  }
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gRBandOf), band_of, sizeof(band_of)));
  g_tables_uploaded = true;
  return ODHIP_SUCCESS;
}

/* mode 0: band stage; 1: choice + synthesis */
int fill_job(RJob &d, const odhip_pvq_refjob &j, int mode) {
  if (!j.d_coef || !j.d_ref || !j.q_band || !j.beta_band || j.bs < 0 || j.bs >= ODHIP_NBSIZES
   || j.nplanes <= 0 || !j.band || !j.items || !j.y || !j.r16) {
    return ODHIP_EINVAL;
  }
  if (((uintptr_t)j.band & 63) || ((uintptr_t)j.items & 15) || ((uintptr_t)j.y & 15)
   || ((uintptr_t)j.r16 & 15) || ((uintptr_t)j.x16 & 15) || ((uintptr_t)j.xr & 15)
   || ((uintptr_t)j.choice & 15)) {
    return ODHIP_EINVAL;
  }
  if (mode == 0 ? (!j.d_qm || !j.x16 || !j.xr) : (!j.d_qm_inv || !j.choice || !j.d_dq)) {
    return ODHIP_EINVAL;
  }
  const int n = 4 << j.bs;
  if (j.w <= 0 || j.h <= 0 || j.w % n || j.h % n) return ODHIP_EINVAL;
  memset(&d, 0, sizeof(d));
  d.coef = j.d_coef;
  d.ref = j.d_ref;
  d.qm = j.d_qm;
  d.qm_inv = j.d_qm_inv;
  d.rec = j.band;
  d.items = j.items;
  d.y = j.y;
  d.r16 = j.r16;
  d.x16 = j.x16;
  d.xr = j.xr;
  d.rate = j.d_rate;
  d.choice = j.choice;
  d.dq = j.d_dq;
  d.nplanes = j.nplanes;
  d.w = j.w;
  d.h = j.h;
  d.bs = j.bs;
  d.bw = j.w/n;
  d.bh = j.h/n;
  d.nblocks = (long)j.nplanes*d.bw*d.bh;
  if (d.nblocks > 0xffffffffL) return ODHIP_EINVAL;
  d.nb_bands = OD_NBANDS[j.bs];
  d.len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  d.is_keyframe = j.is_keyframe != 0;
  d.pli = j.pli;
  for (int i = 0; i <= d.nb_bands; i++) d.off[i] = OD_BAND_OFFS[j.bs][i];
  for (int i = 0; i < d.nb_bands; i++) {
    if (j.q_band[i] < 1) return ODHIP_EINVAL;
    d.q[i] = j.q_band[i];
    d.beta[i] = j.beta_band[i];
  }
  return ODHIP_SUCCESS;
}

int stage_jobs(const odhip_pvq_refjob *jobs, int njobs, int mode, RJob *host, hipStream_t s) {
  if (!jobs || njobs <= 0 || njobs > kMaxJobs) return ODHIP_EINVAL;
  int rc = upload_tables();
  if (rc) return rc;
  for (int i = 0; i < njobs; i++) {
    rc = fill_job(host[i], jobs[i], mode);
    if (rc) return rc;
  }
  ODHIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rjobs), host, sizeof(RJob)*njobs, 0,
   hipMemcpyHostToDevice, s));
  return ODHIP_SUCCESS;
}

void items_begin(RItems &it, double lambda) {
  memset(&it, 0, sizeof(it));
  it.lambda = lambda;
  it.margin = g_margin;
  it.perturb = g_perturb;
}

void items_add(RItems &it, int job, int band, long wgs) {
  if (wgs <= 0) return;
  it.job[it.nitems] = (unsigned char)job;
  it.band[it.nitems] = (unsigned char)band;
  it.wg_start[it.nitems + 1] = it.wg_start[it.nitems] + (int)wgs;
  it.nitems++;
}

/* All (job, band) items; n_only > 0 keeps the bands of that size. */
void items_all(RItems &it, const RJob *host, int njobs, double lambda, int n_only) {
  items_begin(it, lambda);
  for (int j = 0; j < njobs; j++) {
    for (int b = 0; b < host[j].nb_bands; b++) {
      if (n_only > 0 && host[j].off[b + 1] - host[j].off[b] != n_only) continue;
      items_add(it, j, b, (host[j].nblocks + kWave - 1)/kWave);
    }
  }
}

}  // namespace

extern "C" void odhip_pvq_ref_set_theta_margin(double margin, int perturb) {
  g_margin = margin > 0 ? margin : kDefaultMargin;
  g_perturb = perturb != 0;
}

extern "C" int odhip_pvq_ref_theta_probe(const double *d_corr, double *d_t, long n,
 odhip_stream stream) {
  if (!d_corr || !d_t || n < 0) return ODHIP_EINVAL;
  if (n == 0) return ODHIP_SUCCESS;
  k_theta_probe<<<(unsigned)((n + kWave - 1)/kWave), kWave, 0, (hipStream_t)stream>>>(d_corr, d_t, n);
  return odhip_check_launch();
}

extern "C" int odhip_pvq_ref_bands_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  RJob host[kMaxJobs];
  int rc = stage_jobs(jobs, njobs, 0, host, s);
  if (rc) return rc;
  void *cnt = nullptr;
  ODHIP_TRY(hipGetSymbolAddress(&cnt, HIP_SYMBOL(g_unc_count)));
  ODHIP_TRY(hipMemsetAsync(cnt, 0, sizeof(unsigned), s));
  RItems it;
  items_all(it, host, njobs, pvq_norm_lambda, 0);
  if (!it.nitems) return ODHIP_SUCCESS;
  /* band 0 of every block first (it decides the chroma-from-luma flip of the
     block), then the 8-coefficient bands per lane and the 32- / 128-coefficient
     bands per row */
  {
    RItems pi;
    items_begin(pi, pvq_norm_lambda);
    for (int j = 0; j < njobs; j++) items_add(pi, j, 0, (host[j].nblocks + kWave - 1)/kWave);
    k_refb_prep_lane<15><<<pi.wg_start[pi.nitems], kWave, 0, s>>>(pi);
    items_begin(pi, pvq_norm_lambda);
    for (int j = 0; j < njobs; j++) {
      for (int b = 1; b < host[j].nb_bands; b++) {
        if (host[j].off[b + 1] - host[j].off[b] == 8) items_add(pi, j, b, (host[j].nblocks + kWave - 1)/kWave);
      }
    }
    if (pi.nitems) k_refb_prep_lane<8><<<pi.wg_start[pi.nitems], kWave, 0, s>>>(pi);
    for (int sz = 32; sz <= 128; sz *= 4) {
      items_begin(pi, pvq_norm_lambda);
      for (int j = 0; j < njobs; j++) {
        for (int b = 1; b < host[j].nb_bands; b++) {
          if (host[j].off[b + 1] - host[j].off[b] == sz) items_add(pi, j, b, (host[j].nblocks + 3)/4);
        }
      }
      if (!pi.nitems) continue;
      if (sz == 32) k_refb_prep_row<2><<<pi.wg_start[pi.nitems], kWave, 0, s>>>(pi);
      else k_refb_prep_row<8><<<pi.wg_start[pi.nitems], kWave, 0, s>>>(pi);
    }
  }
  k_refb_cands<<<it.wg_start[it.nitems], kWave, 0, s>>>(it);
  /* 128- and 32-coefficient bands: one band per 16-lane row; 15 and 8: per lane */
  const bool lane_only = getenv("ODHIP_PVQ_REF_LANE") != nullptr;
  static const int sizes[4] = {128, 32, 15, 8};
  for (int i = 0; i < 4; i++) {
    if (sizes[i] >= 32 && !lane_only) {
      items_begin(it, pvq_norm_lambda);
      const char *e = getenv("ODHIP_PVQ_FORCE_SEQ");
      if (e && e[0] == '1') it.perturb |= 2;   /* every greedy pulse by the literal scan */
      for (int j = 0; j < njobs; j++) {
        for (int b = 0; b < host[j].nb_bands; b++) {
          if (host[j].off[b + 1] - host[j].off[b] == sizes[i]) {
            items_add(it, j, b, (host[j].nblocks + 3)/4);
          }
        }
      }
      if (!it.nitems) continue;
      if (sizes[i] == 128) k_refb_search_row<8><<<it.wg_start[it.nitems], kWave, 0, s>>>(it);
      else k_refb_search_row<2><<<it.wg_start[it.nitems], kWave, 0, s>>>(it);
      continue;
    }
    items_all(it, host, njobs, pvq_norm_lambda, sizes[i]);
    if (!it.nitems) continue;
    const size_t lds = (size_t)2*sizes[i]*kWave*sizeof(unsigned short);
    k_refb_search<<<it.wg_start[it.nitems], kWave, lds, s>>>(it);
  }
  return odhip_check_launch();
}

extern "C" int odhip_pvq_ref_resolve(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  ODHIP_TRY(hipStreamSynchronize(s));
  unsigned count = 0;
  ODHIP_TRY(hipMemcpyFromSymbol(&count, HIP_SYMBOL(g_unc_count), sizeof(count)));
  if (count == 0) return 0;
  if (count > (unsigned)kUncCap) {
    fprintf(stderr, "libdaalahip: %u bands inside the theta margin exceed the list (%d)\n", count,
     kUncCap);
    return ODHIP_EFAULT;
  }
  Unc *list = (Unc *)malloc(sizeof(Unc)*count);
  if (!list) return ODHIP_EFAULT;
  if (hipMemcpyFromSymbol(list, HIP_SYMBOL(g_unc), sizeof(Unc)*count) != hipSuccess) {
    free(list);
    return ODHIP_EFAULT;
  }
  /* the reference's own expression with the host libm, src/pvq_encoder.c:478 */
  unsigned nfix = 0;
  for (unsigned i = 0; i < count; i++) {
    const int32_t theta = (int32_t)floor(.5 + (32768*2./M_PI)*acos(list[i].corr));
    if (theta != list[i].theta) {
      list[nfix] = list[i];
      list[nfix].theta = theta;
      nfix++;
    }
  }
  if (nfix == 0) {
    free(list);
    return 0;
  }
  RJob host[kMaxJobs];
  int rc = stage_jobs(jobs, njobs, 0, host, s);
  if (rc) {
    free(list);
    return rc;
  }
  for (unsigned i = 0; i < nfix; i++) {
    if (list[i].job >= njobs) {
      free(list);
      return ODHIP_EINVAL;
    }
  }
  Unc *d_list = nullptr;
  if (hipMalloc((void **)&d_list, sizeof(Unc)*nfix) != hipSuccess
   || hipMemcpyAsync(d_list, list, sizeof(Unc)*nfix, hipMemcpyHostToDevice, s) != hipSuccess) {
    free(list);
    if (d_list) (void)hipFree(d_list);
    return ODHIP_EFAULT;
  }
  k_refb_cands_list<<<(nfix + kWave - 1)/kWave, kWave, 0, s>>>(d_list, (int)nfix);
  k_refb_search_list<<<nfix, kWave, (size_t)2*128*kWave*sizeof(unsigned short), s>>>(d_list,
   (int)nfix, pvq_norm_lambda);
  rc = odhip_check_launch();
  hipError_t e = hipStreamSynchronize(s);
  free(list);
  (void)hipFree(d_list);
  if (rc) return rc;
  if (e != hipSuccess) return ODHIP_EFAULT;
  return (int)nfix;
}

extern "C" int odhip_pvq_ref_select_synth_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  RJob host[kMaxJobs];
  int rc = stage_jobs(jobs, njobs, 1, host, s);
  if (rc) return rc;
  for (int j = 0; j < njobs; j++) {
    /* 32x32 and 64x64 blocks code their lowest 512 coefficients only */
    if (host[j].bs >= 3) {
      ODHIP_TRY(hipMemsetAsync(host[j].dq, 0, sizeof(od_coeff)*(size_t)host[j].nplanes*host[j].w
       *host[j].h, s));
    }
  }
  RItems it;
  static const int sizes[4] = {128, 32, 15, 8};
  for (int i = 0; i < 4; i++) {
    items_all(it, host, njobs, pvq_norm_lambda, sizes[i]);
    if (!it.nitems) continue;
    const unsigned grid = it.wg_start[it.nitems];
    if (sizes[i] == 128) k_refb_choose<128><<<grid, kWave, 0, s>>>(it);
    else if (sizes[i] == 32) k_refb_choose<32><<<grid, kWave, 0, s>>>(it);
    else if (sizes[i] == 15) k_refb_choose<15><<<grid, kWave, 0, s>>>(it);
    else k_refb_choose<8><<<grid, kWave, 0, s>>>(it);
  }
  items_begin(it, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) items_add(it, j, 0, (host[j].nblocks*host[j].len + 255)/256);
  k_refb_synth<<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  return odhip_check_launch();
}
