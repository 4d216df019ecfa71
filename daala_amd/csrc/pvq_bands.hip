/* pvq_bands.hip - the PVQ band stage on gfx950: everything in pvq_theta's
   no-reference keyframe path (reference src/pvq_encoder.c:333-641) that does
   not depend on the adaptive entropy coder, for every block of every level of
   a batch of coefficient planes, plus the choice/dequantisation that follows.

   MULTI-JOB launches.  One (plane set, block size) pair is a "job"; a frame
   batch has nine (5 luma + 4 chroma levels).  Small jobs (510 64x64 blocks per
   frame) cannot fill 256 CUs and a launch cannot end before its slowest
   wavefront, so launching jobs back to back serialises their tails.  Instead
   every kernel here takes a table of (job, band) work items with a prefix sum
   of workgroup counts and covers ALL jobs in one launch:

     k_bands_narrow   bands of 8 / 15 coefficients, one band per lane
     k_bands_wide<E>  bands of 32 / 128 coefficients, one band per 16-lane row
     k_choose         per band: `cost <= best_cost` choice, od_gain_expand,
                      synthesis scale (src/pvq.c:766, :1057-1078)
     k_synth          per coefficient: y*scale, inverse QM, scan -> raster
                      (src/pvq.c:1081-1092, src/partition.c:176-194), written as
                      coalesced rows of the dequantised plane

   Layouts: see odhip_pvq_cands in include/daala_hip.h. */
#include "../../include/daala_hip.h"
#include <stdlib.h>
#include <string.h>
#include "od_common.cuh"
#include "od_pvq_math.cuh"
#include "gen/od_scan_tables.h"
#include "pvq_search.cuh"

namespace {

constexpr int kMaxJobs = 16;
constexpr int kMaxItems = kMaxJobs*ODHIP_MAX_BANDS;

struct DJob {
  const od_coeff *coef;
  const int16_t *qm;
  const int16_t *qm_inv;
  odhip_pvq_cands c;
  od_coeff *dq;
  const double *rate;
  int32_t *qg_out;
  long nblocks;
  int nplanes;
  int w;
  int h;
  int bs;
  int nb_bands;
  int len;
  int bw;
  int bh;
  int q[ODHIP_MAX_BANDS];
  int beta[ODHIP_MAX_BANDS];
  int off[ODHIP_MAX_BANDS + 1];
};

struct Items {
  int nitems;
  int force_scan;
  double lambda;
  int wg_start[kMaxItems + 1];
  unsigned char job[kMaxItems];
  unsigned char band[kMaxItems];
};

__device__ DJob g_jobs[kMaxJobs];
/* Scan tables.  kScanXY is indexed wave-uniformly by the one-band-per-lane
   kernel (scalar loads); the kernels that index per lane copy their table into
   LDS first - a constant-memory access with 64 different addresses
   serialises. */
__constant__ unsigned char kScanXY[OD_SCAN_LEN][2];
__device__ unsigned short gScanXY[OD_SCAN_LEN];  /* y << 8 | x */
__device__ short gInvScan[32*32];                /* raster (y*32 + x) -> coding index, -1 */
__device__ unsigned char gBandOf[OD_SCAN_LEN];

__device__ __forceinline__ int find_item(const Items &it, int wg) {
  int lo = 0;
  int hi = it.nitems - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (it.wg_start[mid] <= wg) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

struct BlockPos {
  const od_coeff *src;
  long blk;
  bool live;
};

__device__ __forceinline__ BlockPos locate(const DJob &j, long blk) {
  BlockPos r;
  r.live = blk < j.nblocks;
  r.blk = blk;
  const long b = r.live ? blk : j.nblocks - 1;
  const int N = 4 << j.bs;
  const long per = (long)j.bw*j.bh;
  const int p = (int)(b/per);
  const int rem = (int)(b - p*per);
  const int by = rem/j.bw;
  const int bx = rem - by*j.bw;
  r.src = j.coef + (long)p*j.w*j.h + (long)by*N*j.w + bx*N;
  return r;
}

/* Per-candidate bookkeeping shared by both band kernels: src/pvq_encoder.c:
   :575-595. */
struct CandOut {
  int gain;
  int k;
  int flag;
  int yy;
  double cos_dist;
  double dist;
};

/* ---- short bands: one band per lane ------------------------------------------ */
__global__ __launch_bounds__(kWave) void k_bands_narrow(Items it) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  od_rsqrt_init(threadIdx.x);
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = g_jobs[it.job[item]];
  const int band = it.band[item];
  const int off = jb.off[band];
  const int n = jb.off[band + 1] - off;
  const int q = jb.q[band];
  const int beta = jb.beta[band];
  const int lane = threadIdx.x;
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*kWave + lane);
  int *x0s = (int *)lds;                 /* [n][64] int32 stash, then ...   */
  short *xs = (short *)lds;              /* ... x16 [n][64] in its low half */
  unsigned short *ys = lds + n*kWave;    /* |y| [n][64]                     */
  /* od_vector_log_mag, src/pvq.c:472-484, while gathering the band in coding
     order (od_raster_to_coding_order, src/partition.c:144-170). */
  int sum = 0;
  for (int j = 0; j < n; j++) {
    const int v = bp.src[kScanXY[off + j][1]*jb.w + kScanXY[off + j][0]];
    x0s[j*kWave + lane] = v;
    const int t = (int16_t)(v >> 8);
    sum += t*t;
  }
  int xshift = 8 + 1 + odq_ilog(n + sum)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int acc = 0;
  for (int j = 0; j < n; j++) {
    const int v = x0s[j*kWave + lane];
    const int16_t x16 = (int16_t)odq_shr_round(v*jb.qm[off + j], ODQ_QM_SHIFT + xshift);
    /* In-place narrowing: row j of the int16 view lies inside int32 row j/2,
       which every lane of this (single-wave) workgroup has already read. */
    xs[j*kWave + lane] = x16;
    acc += x16*(int)x16;
  }
  int32_t g;
  const int32_t cg = odq_gain_from_acc(acc, q, beta, xshift, &g);
  const double s2 = (1./256)*(1./256);  /* OD_CGAIN_SCALE_2 */
  const double dist0 = ((1.4*cg)*cg)*s2;
  const long sb = bp.blk*jb.nb_bands + band;
  if (bp.live) {
    jb.c.cg[sb] = cg;
    jb.c.dist0[sb] = dist0;
  }
  const int gain_bound = cg >> ODQ_CGAIN_SHIFT;
  const int first = gain_bound > 1 ? gain_bound : 1;
  int prev_k = 0;
  for (int c = 0; c < 2; c++) {
    const int i = first + c;
    CandOut o = {0, 0, 0, 0, 0., 0.};
    if (i <= gain_bound + 1) {
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT);
      o.gain = i;
      o.k = odq_compute_k_noref(qcg, n, beta);
      o.dist = ((1.4*(qcg - cg))*(qcg - cg))*s2;
      if (!(o.dist > dist0 && o.k != 0)) {
        double yy;
        o.flag = 1;
        o.cos_dist = od_pvq_search_lane(xs, ys, lane, n, o.k, prev_k, (qcg*(double)cg)*s2,
         it.lambda, &yy);
        o.yy = (int)yy;
        prev_k = o.k;
        o.dist = ((1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*o.cos_dist))*s2;
      }
    }
    if (bp.live) {
      jb.c.gain[2*sb + c] = o.gain;
      jb.c.k[2*sb + c] = o.k;
      jb.c.flags[2*sb + c] = o.flag;
      jb.c.yy[2*sb + c] = o.yy;
      jb.c.cos_dist[2*sb + c] = o.cos_dist;
      jb.c.dist[2*sb + c] = o.dist;
      od_coeff *yo = jb.c.y + ((long)c*jb.nblocks + bp.blk)*jb.len + off;
      if (o.flag) {
        for (int j = 0; j < n; j++) {
          const int yj = ys[j*kWave + lane];
          yo[j] = xs[j*kWave + lane] < 0 ? -yj : yj;
        }
      }
      else {
        for (int j = 0; j < n; j++) yo[j] = 0;
      }
    }
  }
}

/* ---- long bands: n = 16*E coefficients per 16-lane DPP row, 4 bands per wave */
template <int E>
__global__ __launch_bounds__(kWave) void k_bands_wide(Items it) {
  constexpr int n = 16*E;
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = g_jobs[it.job[item]];
  const int band = it.band[item];
  const int off = jb.off[band];
  const int q = jb.q[band];
  const int beta = jb.beta[band];
  const int lane = threadIdx.x;
  const int row = lane >> 4;
  const int l = lane & 15;
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*4 + row);
  __shared__ unsigned short s_scan[n];
  for (int j = lane; j < n; j += kWave) s_scan[j] = gScanXY[off + j];
  od_rsqrt_init(lane);
  int v[E];
  int sum = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int xy = s_scan[l*E + e];
    v[e] = bp.src[(xy >> 8)*jb.w + (xy & 255)];
    const int t = (int16_t)(v[e] >> 8);
    sum += t*t;
  }
  sum = row_sum(sum);
  int xshift = 8 + 1 + odq_ilog(n + sum)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int x16[E];
  int ax[E];
  int y[E];
  int acc = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    x16[e] = (int16_t)odq_shr_round(v[e]*jb.qm[off + l*E + e], ODQ_QM_SHIFT + xshift);
    ax[e] = abs(x16[e]);
    y[e] = 0;
    acc += x16[e]*x16[e];
  }
  acc = row_sum(acc);
  int32_t g;
  const int32_t cg = odq_gain_from_acc(acc, q, beta, xshift, &g);
  const double s2 = (1./256)*(1./256);
  const double dist0 = ((1.4*cg)*cg)*s2;
  const long sb = bp.blk*jb.nb_bands + band;
  if (bp.live && l == 0) {
    jb.c.cg[sb] = cg;
    jb.c.dist0[sb] = dist0;
  }
  const int gain_bound = cg >> ODQ_CGAIN_SHIFT;
  const int first = gain_bound > 1 ? gain_bound : 1;
  int prev_k = 0;
  for (int c = 0; c < 2; c++) {
    const int i = first + c;
    CandOut o = {0, 0, 0, 0, 0., 0.};
    if (i <= gain_bound + 1) {
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT);
      o.gain = i;
      o.k = odq_compute_k_noref(qcg, n, beta);
      o.dist = ((1.4*(qcg - cg))*(qcg - cg))*s2;
      if (!(o.dist > dist0 && o.k != 0)) {
        double yy;
        o.flag = 1;
        o.cos_dist = od_pvq_search_row<E>(ax, y, row, l, o.k, prev_k, (qcg*(double)cg)*s2,
         it.lambda, it.force_scan, &yy);
        o.yy = (int)yy;
        prev_k = o.k;
        o.dist = ((1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*o.cos_dist))*s2;
      }
    }
    if (bp.live) {
      if (l == 0) {
        jb.c.gain[2*sb + c] = o.gain;
        jb.c.k[2*sb + c] = o.k;
        jb.c.flags[2*sb + c] = o.flag;
        jb.c.yy[2*sb + c] = o.yy;
        jb.c.cos_dist[2*sb + c] = o.cos_dist;
        jb.c.dist[2*sb + c] = o.dist;
      }
      od_coeff *yo = jb.c.y + ((long)c*jb.nblocks + bp.blk)*jb.len + off + l*E;
#pragma unroll
      for (int e = 0; e < E; e++) yo[e] = o.flag ? (x16[e] < 0 ? -y[e] : y[e]) : 0;
    }
  }
}

/* ---- choice: one (block, band) per lane --------------------------------------
   The comparison `cost <= best_cost` of src/pvq_encoder.c:597-609 with
   cost = dist + lambda*rate (rate from the host entropy model, or absent),
   then od_gain_expand (src/pvq.c:766-811) and the synthesis scale of
   od_pvq_synthesis_partial (src/pvq.c:1057-1078).  choice = {sel, qg, scale,
   qshift}. */
__global__ __launch_bounds__(256) void k_choose(Items it) {
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = g_jobs[it.job[item]];
  const long sb = (long)(blockIdx.x - it.wg_start[item])*256 + threadIdx.x;
  if (sb >= jb.nblocks*jb.nb_bands) return;
  const int band = (int)(sb % jb.nb_bands);
  double best_cost = jb.c.dist0[sb];
  int qg = 0;
  int sel = 0;
  for (int c = 0; c < 2; c++) {
    if (!jb.c.flags[2*sb + c]) continue;
    double cost = jb.c.dist[2*sb + c];
    if (jb.rate) cost = cost + it.lambda*jb.rate[2*sb + c];
    if (cost <= best_cost) {
      best_cost = cost;
      qg = jb.c.gain[2*sb + c];
      sel = c;
    }
  }
  int32_t scale = 0;
  int qshift = ODQ_QM_INV_SHIFT;
  if (qg != 0) {
    const int32_t g = odq_gain_expand(odq_shl32(qg, ODQ_CGAIN_SHIFT), jb.q[band], jb.beta[band]);
    const int yy = jb.c.yy[2*sb + sel];
    int gshift = odq_ilog(g) - 14;
    gshift = gshift > 0 ? gshift : 0;
    if (yy != 0) {
      int rsqrt_shift;
      const int16_t rsqrt = odq_rsqrt(yy, &rsqrt_shift);
      scale = odq_vshr_round(rsqrt*(int64_t)g, rsqrt_shift + gshift - 16);
    }
    qshift = ODQ_QM_INV_SHIFT - gshift;
  }
  reinterpret_cast<int4 *>(jb.c.choice)[sb] = make_int4(sel, qg, scale, qshift);
  if (jb.qg_out) jb.qg_out[sb] = qg;
}

/* ---- synthesis: four horizontally adjacent coefficients per lane -------------
   x = y*scale (Q16, no rounding), out = SHR_ROUND(x*qm_inv, qshift)
   (src/pvq.c:1081-1092) scattered to raster by the inverse scan; DC passed
   through (keyframes quantise it in the Haar DC tree, src/encode.c:1377);
   positions PVQ never codes are zero (od_init_skipped_coeffs,
   src/state.c:1347-1358).  Every store is a coalesced 16-byte row segment. */
__global__ __launch_bounds__(256) void k_synth(Items it) {
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = g_jobs[it.job[item]];
  __shared__ short s_inv[32*32];
  __shared__ unsigned char s_band[OD_SCAN_LEN];
  for (int i = threadIdx.x; i < 32*32; i += 256) s_inv[i] = gInvScan[i];
  for (int i = threadIdx.x; i < OD_SCAN_LEN; i += 256) s_band[i] = gBandOf[i];
  /* One workgroup = one 1024-coefficient segment of one plane row: the
     decomposition of the workgroup index is scalar (once per wave), the
     per-lane one is shifts only. */
  __shared__ int4 s_choice[256*ODHIP_MAX_BANDS/4];  /* <= 1024/N blocks x nb bands */
  const int segs = (jb.w + 1023) >> 10;
  const int wg = blockIdx.x - it.wg_start[item];
  const int seg = wg % segs;
  const int prow = wg / segs;                 /* plane * h + y */
  const int y = prow % jb.h;
  const int p = prow / jb.h;
  const int sh = jb.bs + 2;
  const int N = 1 << sh;
  const int by = y >> sh;
  const int ly = y & (N - 1);
  const int x0 = seg << 10;
  const int bx0 = x0 >> sh;
  const int nbx = min(1024 >> sh, jb.bw - bx0);    /* blocks this segment touches */
  const long blk0 = ((long)p*jb.bh + by)*jb.bw + bx0;
  for (int i = threadIdx.x; i < nbx*jb.nb_bands; i += 256) {
    s_choice[i] = reinterpret_cast<const int4 *>(jb.c.choice)[blk0*jb.nb_bands + i];
  }
  __syncthreads();
  const int x = x0 + threadIdx.x*4;
  if (x >= jb.w) return;
  const int bxl = (x >> sh) - bx0;
  const int lx = x & (N - 1);
  const long blk = blk0 + bxl;
  const long idx = (long)p*jb.w*jb.h + (long)y*jb.w + x;
  int out[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int v = 0;
    if (ly < 32 && lx + i < 32) {
      const int j = s_inv[ly*32 + lx + i];
      if (j == 0) v = jb.coef[idx];
      else if (j > 0 && j < jb.len) {
        const int band = s_band[j];
        const int4 ch = s_choice[bxl*jb.nb_bands + band];
        if (ch.y != 0) {
          const int yv = jb.c.y[((long)ch.x*jb.nblocks + blk)*jb.len + j];
          const int32_t xq = (int32_t)((int16_t)yv*(int64_t)ch.z >> 16);
          v = odq_shr_round(xq*jb.qm_inv[j], ch.w);
        }
      }
    }
    out[i] = v;
  }
  *reinterpret_cast<int4 *>(jb.dq + idx) = make_int4(out[0], out[1], out[2], out[3]);
}

/* ---- host side ----------------------------------------------------------------- */
bool g_tables_uploaded = false;

int upload_tables(void) {
  if (g_tables_uploaded) return ODHIP_SUCCESS;
  short inv[32*32];
  unsigned char band_of[OD_SCAN_LEN];
  for (int i = 0; i < 32*32; i++) inv[i] = -1;
  for (int j = 0; j < OD_SCAN_LEN; j++) inv[OD_SCAN_XY[j][1]*32 + OD_SCAN_XY[j][0]] = (short)j;
  for (int j = 0; j < OD_SCAN_LEN; j++) {
    int b = 0;
    while (b + 1 < OD_NBANDS[4] && j >= OD_BAND_OFFS[4][b + 1]) b++;
    band_of[j] = (unsigned char)b;
  }
  unsigned short packed[OD_SCAN_LEN];
  for (int j = 0; j < OD_SCAN_LEN; j++) packed[j] = (unsigned short)(OD_SCAN_XY[j][1] << 8 | OD_SCAN_XY[j][0]);
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kScanXY), OD_SCAN_XY, sizeof(OD_SCAN_XY)));
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gScanXY), packed, sizeof(packed)));
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gInvScan), inv, sizeof(inv)));
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gBandOf), band_of, sizeof(band_of)));
  g_tables_uploaded = true;
  return ODHIP_SUCCESS;
}

int fill_job(DJob &d, const odhip_pvq_job &j, int mode) {
  if (!j.d_coef || !j.q_band || !j.beta_band || j.bs < 0 || j.bs >= ODHIP_NBSIZES
   || j.nplanes <= 0) {
    return ODHIP_EINVAL;
  }
  const odhip_pvq_cands &c = j.cands;
  if (!c.cg || !c.dist0 || !c.gain || !c.k || !c.flags || !c.yy || !c.cos_dist || !c.dist
   || !c.y || !c.choice) {
    return ODHIP_EINVAL;
  }
  /* mode 0: band stage (needs qm); 1: choice + synthesis (qm_inv, dq); 2: choice only */
  if (mode == 0 ? !j.d_qm : mode == 1 ? (!j.d_qm_inv || !j.d_dq) : false) return ODHIP_EINVAL;
  const int n = 4 << j.bs;
  if (j.w <= 0 || j.h <= 0 || j.w % n || j.h % n || (j.w & 3)) return ODHIP_EINVAL;
  memset(&d, 0, sizeof(d));
  d.coef = j.d_coef;
  d.qm = j.d_qm;
  d.qm_inv = j.d_qm_inv;
  d.c = c;
  d.dq = j.d_dq;
  d.rate = j.d_rate;
  d.qg_out = j.d_qg;
  d.nplanes = j.nplanes;
  d.w = j.w;
  d.h = j.h;
  d.bs = j.bs;
  d.bw = j.w/n;
  d.bh = j.h/n;
  d.nblocks = (long)j.nplanes*d.bw*d.bh;
  d.nb_bands = OD_NBANDS[j.bs];
  d.len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  for (int i = 0; i <= d.nb_bands; i++) d.off[i] = OD_BAND_OFFS[j.bs][i];
  for (int i = 0; i < d.nb_bands; i++) {
    if (j.q_band[i] < 1) return ODHIP_EINVAL;
    d.q[i] = j.q_band[i];
    d.beta[i] = j.beta_band[i];
  }
  return ODHIP_SUCCESS;
}

int stage_jobs(const odhip_pvq_job *jobs, int njobs, int mode, DJob *host,
 hipStream_t s) {
  if (!jobs || njobs <= 0 || njobs > kMaxJobs) return ODHIP_EINVAL;
  int rc = upload_tables();
  if (rc) return rc;
  for (int i = 0; i < njobs; i++) {
    rc = fill_job(host[i], jobs[i], mode);
    if (rc) return rc;
  }
  /* Pageable source: the runtime stages the bytes before returning, so `host`
     may live on the caller's stack; stream order protects g_jobs itself. */
  ODHIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_jobs), host, sizeof(DJob)*njobs, 0,
   hipMemcpyHostToDevice, s));
  return ODHIP_SUCCESS;
}

/* Side streams for kernels that may overlap (created once per process). */
hipStream_t g_side[2] = {nullptr, nullptr};
hipEvent_t g_fork = nullptr;
hipEvent_t g_join[2] = {nullptr, nullptr};

int fork_streams(hipStream_t s, hipStream_t side[2]) {
  if (getenv("ODHIP_PVQ_SERIAL")) return ODHIP_SUCCESS;
  if (!g_fork) {
    ODHIP_TRY(hipEventCreateWithFlags(&g_fork, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
      ODHIP_TRY(hipStreamCreateWithFlags(&g_side[i], hipStreamNonBlocking));
      ODHIP_TRY(hipEventCreateWithFlags(&g_join[i], hipEventDisableTiming));
    }
  }
  ODHIP_TRY(hipEventRecord(g_fork, s));
  for (int i = 0; i < 2; i++) {
    ODHIP_TRY(hipStreamWaitEvent(g_side[i], g_fork, 0));
    side[i] = g_side[i];
  }
  return ODHIP_SUCCESS;
}

int join_streams(hipStream_t s, hipStream_t side[2]) {
  for (int i = 0; i < 2; i++) {
    if (side[i] == s) continue;
    ODHIP_TRY(hipEventRecord(g_join[i], side[i]));
    ODHIP_TRY(hipStreamWaitEvent(s, g_join[i], 0));
  }
  return ODHIP_SUCCESS;
}

void items_begin(Items &it, double lambda) {
  memset(&it, 0, sizeof(it));
  it.lambda = lambda;
  const char *e = getenv("ODHIP_PVQ_FORCE_SCAN");
  it.force_scan = e && e[0] == '1';
}

void items_add(Items &it, int job, int band, long wgs) {
  if (wgs <= 0) return;
  it.job[it.nitems] = (unsigned char)job;
  it.band[it.nitems] = (unsigned char)band;
  it.wg_start[it.nitems + 1] = it.wg_start[it.nitems] + (int)wgs;
  it.nitems++;
}

}  // namespace

extern "C" int odhip_pvq_band_layout(int bs, int *nb_bands, int *offsets, int *len) {
  if (bs < 0 || bs >= ODHIP_NBSIZES) return ODHIP_EINVAL;
  const int n = 4 << bs;
  if (nb_bands) *nb_bands = OD_NBANDS[bs];
  if (offsets) for (int i = 0; i <= OD_NBANDS[bs]; i++) offsets[i] = OD_BAND_OFFS[bs][i];
  if (len) *len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pvq_noref_bands_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DJob host[kMaxJobs];
  int rc = stage_jobs(jobs, njobs, 0, host, s);
  if (rc) return rc;
  Items it;
  /* short bands: 64 bands per workgroup */
  items_begin(it, pvq_norm_lambda);
  int nmax = 0;
  for (int j = 0; j < njobs; j++) {
    for (int b = 0; b < host[j].nb_bands; b++) {
      const int n = host[j].off[b + 1] - host[j].off[b];
      if (n < 32) {
        items_add(it, j, b, (host[j].nblocks + kWave - 1)/kWave);
        nmax = n > nmax ? n : nmax;
      }
    }
  }
  /* The three band kernels are independent: the two long-band kernels run on
     side streams forked from / joined to the caller's stream, so their tails
     and their different bottlenecks (LDS vs DPP/VALU) overlap. */
  hipStream_t side[2] = {s, s};
  if (fork_streams(s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
  if (it.nitems) {
    k_bands_narrow<<<it.wg_start[it.nitems], kWave, (size_t)nmax*kWave*4, s>>>(it);
  }
  /* long bands: 4 bands per workgroup */
  for (int width = 32; width <= 128; width *= 4) {
    hipStream_t ws = side[width == 32 ? 0 : 1];
    items_begin(it, pvq_norm_lambda);
    for (int j = 0; j < njobs; j++) {
      for (int b = 0; b < host[j].nb_bands; b++) {
        if (host[j].off[b + 1] - host[j].off[b] == width) {
          items_add(it, j, b, (host[j].nblocks + 3)/4);
        }
      }
    }
    if (!it.nitems) continue;
    if (width == 32) k_bands_wide<2><<<it.wg_start[it.nitems], kWave, 0, ws>>>(it);
    else k_bands_wide<8><<<it.wg_start[it.nitems], kWave, 0, ws>>>(it);
  }
  if (join_streams(s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
  return odhip_check_launch();
}

extern "C" int odhip_pvq_select_synth_noref_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DJob host[kMaxJobs];
  int rc = stage_jobs(jobs, njobs, 1, host, s);
  if (rc) return rc;
  Items it;
  items_begin(it, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) {
    items_add(it, j, 0, (host[j].nblocks*host[j].nb_bands + 255)/256);
  }
  k_choose<<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  items_begin(it, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) {
    items_add(it, j, 0, (long)host[j].nplanes*host[j].h*((host[j].w + 1023) >> 10));
  }
  k_synth<<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  return odhip_check_launch();
}

extern "C" int odhip_pvq_choose_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DJob host[kMaxJobs];
  int rc = stage_jobs(jobs, njobs, 2, host, s);
  if (rc) return rc;
  Items it;
  items_begin(it, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) {
    items_add(it, j, 0, (host[j].nblocks*host[j].nb_bands + 255)/256);
  }
  k_choose<<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  return odhip_check_launch();
}

extern "C" int odhip_pvq_noref_bands(const od_coeff *d_coef, int nplanes, int w, int h,
 int bs, const int16_t *d_qm, const int32_t *q_band, const int32_t *beta_band,
 double pvq_norm_lambda, const odhip_pvq_cands *out, odhip_stream stream) {
  if (!out) return ODHIP_EINVAL;
  odhip_pvq_job j;
  memset(&j, 0, sizeof(j));
  j.d_coef = d_coef;
  j.nplanes = nplanes;
  j.w = w;
  j.h = h;
  j.bs = bs;
  j.d_qm = d_qm;
  j.q_band = q_band;
  j.beta_band = beta_band;
  j.cands = *out;
  return odhip_pvq_noref_bands_multi(&j, 1, pvq_norm_lambda, stream);
}

extern "C" int odhip_pvq_select_synth_noref(od_coeff *d_dq, const od_coeff *d_coef,
 int nplanes, int w, int h, int bs, const int16_t *d_qm_inv, const int32_t *q_band,
 const int32_t *beta_band, double pvq_norm_lambda, const odhip_pvq_cands *in,
 const double *d_rate, int32_t *d_qg_out, odhip_stream stream) {
  if (!in) return ODHIP_EINVAL;
  odhip_pvq_job j;
  memset(&j, 0, sizeof(j));
  j.d_coef = d_coef;
  j.nplanes = nplanes;
  j.w = w;
  j.h = h;
  j.bs = bs;
  j.d_qm_inv = d_qm_inv;
  j.q_band = q_band;
  j.beta_band = beta_band;
  j.cands = *in;
  j.d_dq = d_dq;
  j.d_rate = d_rate;
  j.d_qg = d_qg_out;
  return odhip_pvq_select_synth_noref_multi(&j, 1, pvq_norm_lambda, stream);
}
