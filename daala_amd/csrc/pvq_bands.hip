/* pvq_bands.hip - the PVQ band stage on gfx950: everything in pvq_theta's
   no-reference keyframe path (reference src/pvq_encoder.c:333-641) that does
   not depend on the adaptive entropy coder, for every block of every level of
   a batch of coefficient planes, plus the choice/dequantisation that follows.

   MULTI-JOB launches.  One (plane set, block size) pair is a "job"; a frame
   batch has nine (5 luma + 4 chroma levels).  Every kernel takes a table of
   (job, band) work items with a prefix sum of workgroup counts and covers ALL
   jobs in one launch, so small levels overlap with large ones.

   SORTED SEARCH.  The cost of a band is its pulse count, which spans 1..350
   within one level, and a wavefront runs as long as its slowest lane.  So:

     k_prep_lane / k_prep_wide     QM scaling to x16 (library scratch), gain, the
                                   two candidates' K and pruning decision
                                   (first half of the band record), and a SORT
                                   KEY = binned (pulses of candidate 0, extra
                                   pulses of candidate 1)
     k_hist / k_prefix / k_scatter counting sort of the block indices of each
                                   (job, band) item by key, heavy first
     k_search<N>                   one band per lane over the sorted order: the
                                   64 bands of a wavefront need (almost) the same
                                   number of pulses (pvq_lane.cuh)
     k_choose                      per band: `cost <= best_cost` choice,
                                   od_gain_expand, synthesis scale
                                   (src/pvq.c:766, :1057-1078)
     k_synth                       per coefficient: y*scale, inverse QM,
                                   scan -> raster (src/pvq.c:1081-1092,
                                   src/partition.c:176-194)

   Memory: every global access of the hot kernels is a whole 16-byte vector per
   lane and every record half a whole 32-byte sector (odhip_pvq_band in
   include/daala_hip.h): the earlier structure-of-arrays layout with 2..8-byte
   scattered stores wrote 3x the bytes (rocprofv3 WRITE_SIZE). */
#include "../../include/daala_hip.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "od_ctx.cuh"
#include "od_pvq_math.cuh"
#include "od_occupancy.cuh"
#include "od_krange.cuh"
#include "gen/od_scan_tables.h"
#define OD_RSQ_HUGE
#include "pvq_search.cuh"
#include "pvq_lane.cuh"

namespace {

constexpr int kMaxJobs = 16;
constexpr int kMaxItems = kMaxJobs*ODHIP_MAX_BANDS;

struct DJob {
  const od_coeff *coef;
  const int16_t *qm;
  const int16_t *qm_inv;
  odhip_pvq_band *rec;
  int16_t *y;
  int32_t *choice;
  double *cos_dist;
  od_coeff *dq;
  const double *rate;
  int32_t *qg_out;
  /* library scratch of the sorted band stage */
  int16_t *x16;            /* [B][len]  QM-scaled band vectors, coding order      */
  unsigned short *keys;    /* [nb][B]   sort key (pulse-count class)              */
  unsigned *ids;           /* [nb][B]   block indices sorted by key               */
  unsigned *krange;        /* od_krange.cuh: the calling context's counter        */
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
  /* planes from plane_split on (blocks from split_blk on) take q2: the Cr half of a
     chroma plane set - pvq_qm_q4[pli] is per plane, src/encode.c:3052-3072 */
  long split_blk;
  int q2[ODHIP_MAX_BANDS];
};

/* A band whose priced choice the host (re)decides: candidate rates from the host libm. */
struct PUnc {
  int job;
  unsigned sb;             /* blk*nb_bands + band */
  double rate[2];
};
constexpr int kPUncCap = 1 << 16;

struct Items {
  int nitems;
  int reserved;
  double lambda;
  const DJob *jobs;        /* the calling context's device job table [kMaxJobs]   */
  unsigned *sort;          /* its counting-sort arrays: histogram, bin starts and
                              cursors, kMaxItems*kKeyBins words each              */
  unsigned *pcount;        /* priced choice: bands too close to call on the device; pcount[1]: bands with a
                              candidate above ODHIP_PVQ_MAX_K (od_krange.cuh), cleared only when taken */
  struct PUnc *plist;      /* ... and their list [kPUncCap]                       */
  double tol_scale;        /* test hook: multiplies the decision margin           */
  int fuse;                /* the search kernels also make the priced choice      */
  int reserved1;
  int wg_start[kMaxItems + 1];
  unsigned char job[kMaxItems];
  unsigned char band[kMaxItems];
};

/* Scan tables.  kScanXY is indexed wave-uniformly by the one-band-per-lane
   preparation kernel (scalar loads); the kernels that index per lane copy
   their table into LDS first - a constant-memory access with 64 different
   addresses serialises. */
__constant__ unsigned char kScanXY[OD_SCAN_LEN][2];
/* Round 5, k_decide_pair128's staged load.  A 128-coefficient band (coding positions 128..255, 256..383 or
   384..511) covers a region of its block that consists of 32 aligned 4-coefficient row segments: kSeg128[g][q] =
   y << 8 | x of segment q of geometry g = off/128 - 1, in raster order; kSlot128[g][p] = where coding position
   off + p sits in that staged order (segment*4 + element). */
__constant__ unsigned short kSeg128[3][32];
__constant__ unsigned char kSlot128[3][128];
/* ... and the 32-coefficient bands (coding positions 32..63, 64..95, 96..127; k_decide_lane32): 16 aligned
   2-coefficient segments (band 3 of an 8x8 block starts at x = 2 in its rows 4..7), geometry off/32 - 1. */
__constant__ unsigned short kSeg32[3][16];
__constant__ unsigned char kSlot32[3][32];
__device__ __attribute__((aligned(16))) unsigned short gScanXY[OD_SCAN_LEN];  /* y << 8 | x */
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

constexpr int kKeyBins = 1024;
constexpr int kSortChunk = 4096;

__device__ __forceinline__ int od_pulse_bin(int p) {   /* 0..63, monotone */
  if (p < 24) return p;
  if (p < 32) return 24 + ((p - 24) >> 1);
  if (p < 64) return 28 + ((p - 32) >> 2);
  if (p < 128) return 36 + ((p - 64) >> 3);
  if (p < 256) return 44 + ((p - 128) >> 4);
  if (p < 512) return 52 + ((p - 256) >> 5);
  const int t = (p - 512) >> 7;
  return 60 + (t < 3 ? t : 3);
}

__device__ __forceinline__ int od_extra_bin(int p) {   /* 0..15, monotone */
  if (p < 12) return p;
  if (p < 16) return 12;
  if (p < 24) return 13;
  if (p < 48) return 14;
  return 15;
}

/* First half of a band record as the two 16-byte vectors it is stored in. */
struct RecHead {
  int32_t cg;
  int32_t gain[2];
  int k[2];
  int flags[2];
};

__device__ __forceinline__ RecHead rec_head_load(const odhip_pvq_band *r) {
  const int4 a = reinterpret_cast<const int4 *>(r)[0];
  const int4 b = reinterpret_cast<const int4 *>(r)[1];
  RecHead h;
  h.cg = a.x;
  h.gain[0] = a.y;
  h.gain[1] = a.z;
  h.k[0] = (int16_t)(a.w & 0xffff);
  h.k[1] = a.w >> 16;
  h.flags[0] = b.x & 0xff;
  h.flags[1] = b.x >> 8 & 0xff;
  return h;
}

/* Everything the preparation kernels need of a job, read ONCE at the top of the
   kernel: g_jobs is global memory that the kernels' own stores might alias as
   far as the compiler knows, so a field read after a store is reloaded with an
   exposed memory latency (five serialised reloads per band before this). */
struct PrepCtx {
  const int16_t *qm;       /* + off */
  int16_t *x16;            /* + off - pad */
  odhip_pvq_band *rec;     /* + band */
  unsigned short *keys;    /* + band*nblocks */
  int w;
  int len;
  int nb_bands;
  int q;
  int q2;
  long split_blk;
  int beta;
  unsigned *krange;        /* od_krange.cuh: the context's counter (Items::pcount + 1) */
};

/* The band's quantiser step for block blk. */
__device__ __forceinline__ int prep_q(const PrepCtx &cx, long blk) {
  return blk >= cx.split_blk ? cx.q2 : cx.q;
}

__device__ __forceinline__ PrepCtx prep_ctx(const DJob &jb, int band, int off, int pad) {
  PrepCtx c;
  c.krange = jb.krange;
  c.qm = jb.qm + off;
  c.x16 = jb.x16 + off - pad;
  c.rec = jb.rec + band;
  c.keys = jb.keys + (long)band*jb.nblocks;
  c.w = jb.w;
  c.len = jb.len;
  c.nb_bands = jb.nb_bands;
  c.q = jb.q[band];
  c.q2 = jb.q2[band];
  c.split_blk = jb.split_blk;
  c.beta = jb.beta[band];
  return c;
}

/* Everything of a band that precedes the search, from cg on: the two gain
   candidates (src/pvq_encoder.c:578-592), the first half of the record and the
   sort key.  Heavy bands get SMALL keys so that they are dispatched first. */
struct BandHead {
  int4 h0;                 /* cg, gain[2], k[0] | k[1] << 16                  */
  int4 h1;                 /* flags[0] | flags[1] << 8, 0, dist0 (lo, hi)     */
  unsigned short key;
};

__device__ __forceinline__ BandHead od_band_head(int beta, int n, int32_t cg) {
  const double s2 = (1./256)*(1./256);  /* OD_CGAIN_SCALE_2 */
  const double dist0 = ((1.4*cg)*cg)*s2;
  const int gain_bound = cg >> ODQ_CGAIN_SHIFT;
  const int first = gain_bound > 1 ? gain_bound : 1;
  int kk[2] = {0, 0};
  int fl[2] = {0, 0};
  int gain[2] = {0, 0};
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int i = first + c;
    if (i <= gain_bound + 1) {
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT);
      gain[c] = i;
      kk[c] = odq_compute_k_noref(qcg, n, beta);
      const double dist = ((1.4*(qcg - cg))*(qcg - cg))*s2;
      fl[c] = !(dist > dist0 && kk[c] != 0);
      if (fl[c] && kk[c] > kMaxK) fl[c] = 2;   /* not representable: reported, not searched */
      if (kk[c] > kMaxK) kk[c] = kMaxK;
    }
  }
  BandHead bh;
  bh.h0 = make_int4(cg, gain[0], gain[1], (kk[0] & 0xffff) | kk[1] << 16);
  bh.h1 = make_int4(fl[0] | fl[1] << 8, 0, __double2loint(dist0), __double2hiint(dist0));
  const bool s0 = fl[0] == 1;
  const bool s1 = fl[1] == 1;
  const int p0 = s0 ? kk[0] : 0;
  int p1 = 0;
  if (s1) p1 = s0 && kk[0] > 0 && kk[0] <= kk[1] ? kk[1] - kk[0] : kk[1];
  bh.key = (unsigned short)(kKeyBins - 1 - (od_pulse_bin(p0) << 4 | od_extra_bin(p1)));
  return bh;
}

__device__ __forceinline__ void od_band_candidates(const PrepCtx &cx, int n, long blk, int32_t cg) {
  const BandHead bh = od_band_head(cx.beta, n, cg);
  if (bh.h1.x & 0x0202) atomicAdd(cx.krange, 1u);
  int4 *out = reinterpret_cast<int4 *>(cx.rec + blk*cx.nb_bands);
  out[0] = bh.h0;
  out[1] = bh.h1;
  cx.keys[blk] = bh.key;
}

/* ---- prep, bands of up to 32 coefficients: one band per lane ------------------
   The band is gathered in coding order (od_raster_to_coding_order,
   src/partition.c:144-170) with ALL its loads in flight at once (one exposed
   memory latency per band, not one per four coefficients),
   od_vector_log_mag (src/pvq.c:472-484) gives the scaling shift, and the QM
   scaling (src/pvq_encoder.c:381,:398) is written as whole 16-byte groups of
   x16 (the 15-coefficient band includes the unused DC slot of its block). */
template <int N>
__device__ __forceinline__ void od_prep_lane(const DJob &jb, int band, int off, const BlockPos &bp) {
  constexpr int PAD = N == 15 ? 1 : 0;
  constexpr int NV = (N + PAD)/8;
  const PrepCtx cx = prep_ctx(jb, band, off, PAD);
  int qm[N];
#pragma unroll
  for (int j = 0; j < N; j++) qm[j] = cx.qm[j];
  int v[N];
#pragma unroll
  for (int j = 0; j < N; j++) v[j] = bp.src[kScanXY[off + j][1]*cx.w + kScanXY[off + j][0]];
  int sum = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int t = (int16_t)(v[j] >> 8);
    sum += t*t;
  }
  int xshift = 8 + 1 + odq_ilog(N + sum)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int4 *xo = reinterpret_cast<int4 *>(cx.x16 + bp.blk*cx.len);
  int acc = 0;
#pragma unroll
  for (int g = 0; g < NV; g++) {
    int d[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      int x[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int j = g*8 + 2*t + u - PAD;
        x[u] = 0;
        if (j >= 0) x[u] = (int16_t)odq_shr_round(v[j]*qm[j], ODQ_QM_SHIFT + xshift);
        acc += x[u]*x[u];
      }
      d[t] = (x[0] & 0xffff) | x[1] << 16;
    }
    if (bp.live) xo[g] = make_int4(d[0], d[1], d[2], d[3]);
  }
  if (!bp.live) return;
  int32_t g;
  const int32_t cg = odq_gain_from_acc(acc, prep_q(cx, bp.blk), cx.beta, xshift, &g);
  od_band_candidates(cx, N, bp.blk, cg);
}

__global__ __launch_bounds__(kWave) void k_prep_lane(Items it) {
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = it.jobs[it.job[item]];
  const int band = it.band[item];
  const int off = jb.off[band];
  const int n = jb.off[band + 1] - off;
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x);
  if (n == 8) od_prep_lane<8>(jb, band, off, bp);
  else if (n == 15) od_prep_lane<15>(jb, band, off, bp);
  else od_prep_lane<32>(jb, band, off, bp);
}

/* ---- prep, the CxC low-frequency corner of every block: one block per lane -----
   Coding positions 0..15 of a 4x4 block and 0..63 of every larger block lie in
   its top-left 4x4 / 8x8 corner (od_raster_to_coding_order,
   src/partition.c:144-170) and hold band 0 (15 coefficients) resp. bands 0..3
   (15, 8, 8, 32).  A lane loads its block's corner as whole 16-byte row
   segments - adjacent lanes read adjacent blocks, so a wavefront's loads are
   (nearly) contiguous, where one-band-per-lane gathers fetched every 32-byte
   sector once per coefficient - permutes it to coding order in registers (the
   scan is a compile-time constant), and prepares all of the corner's bands. */
/* Compile-time loop: f(std::integral_constant<int, J>) for J in [J0, J1).  The
   corner kernel indexes its register arrays with scan-table entries; the
   indices must be constant expressions for the arrays to stay in registers
   (a plain unrolled loop left them in scratch memory: 784 bytes per lane). */
template <int J0, int J1, class F>
__device__ __forceinline__ void od_static_for(F &&f) {
  if constexpr (J0 < J1) {
    f(std::integral_constant<int, J0>{});
    od_static_for<J0 + 1, J1>(f);
  }
}

template <int C>
__global__ __launch_bounds__(kWave) void k_prep_corner(Items it) {
  constexpr int NC = C*C;
  constexpr int NBANDS = C == 4 ? 1 : 4;
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = it.jobs[it.job[item]];
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x);
  PrepCtx cx[NBANDS];
#pragma unroll
  for (int b = 0; b < NBANDS; b++) cx[b] = prep_ctx(jb, b, 0, 0);
  const int16_t *const qmp = cx[0].qm;
  const int w = cx[0].w;
  int r[NC];        /* raster, row-major CxC */
#pragma unroll
  for (int y = 0; y < C; y++) {
#pragma unroll
    for (int x = 0; x < C; x += 4) {
      const int4 q = *reinterpret_cast<const int4 *>(bp.src + y*w + x);
      r[y*C + x] = q.x;
      r[y*C + x + 1] = q.y;
      r[y*C + x + 2] = q.z;
      r[y*C + x + 3] = q.w;
    }
  }
  int x16[NC];
  x16[0] = 0;
  od_static_for<0, NBANDS>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int kOff[5] = {1, 16, 24, 32, 64};
    constexpr int off = kOff[b];
    constexpr int n = kOff[b + 1] - off;
    /* od_vector_log_mag, src/pvq.c:472-484 */
    int sum = 0;
    od_static_for<off, off + n>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int t = (int16_t)(r[OD_SCAN_XY[j][1]*C + OD_SCAN_XY[j][0]] >> 8);
      sum += t*t;
    });
    int xshift = 8 + 1 + odq_ilog(n + sum)/2 - 15;
    xshift = xshift > 0 ? xshift : 0;
    int acc = 0;
    od_static_for<off, off + n>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      x16[j] = (int16_t)odq_shr_round(r[OD_SCAN_XY[j][1]*C + OD_SCAN_XY[j][0]]*qmp[j],
       ODQ_QM_SHIFT + xshift);
      acc += x16[j]*x16[j];
    });
    if (bp.live) {
      int32_t g;
      const int32_t cg = odq_gain_from_acc(acc, prep_q(cx[b], bp.blk), cx[b].beta, xshift, &g);
      od_band_candidates(cx[b], n, bp.blk, cg);
    }
  });
  if (bp.live) {
    int4 *xo = reinterpret_cast<int4 *>(cx[0].x16 + bp.blk*cx[0].len);
    od_static_for<0, NC/8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      xo[g] = make_int4((x16[g*8] & 0xffff) | x16[g*8 + 1] << 16,
       (x16[g*8 + 2] & 0xffff) | x16[g*8 + 3] << 16,
       (x16[g*8 + 4] & 0xffff) | x16[g*8 + 5] << 16,
       (x16[g*8 + 6] & 0xffff) | x16[g*8 + 7] << 16);
    });
  }
}

/* ---- prep, 128-coefficient bands: one band per 16-lane DPP row ---------------- */
__global__ __launch_bounds__(kWave) void k_prep_wide(Items it) {
  constexpr int E = 8;
  constexpr int n = 16*E;
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = it.jobs[it.job[item]];
  const int band = it.band[item];
  const int off = jb.off[band];
  const int lane = threadIdx.x;
  const int row = lane >> 4;
  const int l = lane & 15;
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*4 + row);
  const PrepCtx cx = prep_ctx(jb, band, off, 0);
  __shared__ unsigned short s_scan[n];
  for (int j = lane; j < n; j += kWave) s_scan[j] = gScanXY[off + j];
  const int4 qm4 = *reinterpret_cast<const int4 *>(cx.qm + l*E);
  __syncthreads();
  int v[E];
  int sum = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int xy = s_scan[l*E + e];
    v[e] = bp.src[(xy >> 8)*cx.w + (xy & 255)];
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int t = (int16_t)(v[e] >> 8);
    sum += t*t;
  }
  sum = row_sum(sum);
  int xshift = 8 + 1 + odq_ilog(n + sum)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int acc = 0;
  int d[E/2];
  const int qd[4] = {qm4.x, qm4.y, qm4.z, qm4.w};
#pragma unroll
  for (int e = 0; e < E; e += 2) {
    const int x0 = (int16_t)odq_shr_round(v[e]*(int)(short)qd[e/2], ODQ_QM_SHIFT + xshift);
    const int x1 = (int16_t)odq_shr_round(v[e + 1]*(qd[e/2] >> 16), ODQ_QM_SHIFT + xshift);
    acc += x0*x0 + x1*x1;
    d[e/2] = (x0 & 0xffff) | x1 << 16;
  }
  if (bp.live) {
    *reinterpret_cast<int4 *>(cx.x16 + bp.blk*cx.len + l*E) = make_int4(d[0], d[1], d[2], d[3]);
  }
  acc = row_sum(acc);
  if (!bp.live || l != 0) return;
  int32_t g;
  const int32_t cg = odq_gain_from_acc(acc, prep_q(cx, bp.blk), cx.beta, xshift, &g);
  od_band_candidates(cx, n, bp.blk, cg);
}

/* ---- counting sort of each item's block indices by key (heavy first) ------------
   The order inside a key bin stays close to block order (workgroups reserve
   contiguous ranges per bin), so the 64 bands of a search wavefront are near
   one another in memory as well as in pulse count.  Heavy-first matters: one
   wavefront of 128-coefficient bands with K = 90 runs for ~250 us, as long as
   the rest of its launch. */
/* it.sort: histogram (zero between calls), bin starts, cursors. */
__device__ __forceinline__ unsigned *sort_hist(const Items &it) { return it.sort; }
__device__ __forceinline__ unsigned *sort_binstart(const Items &it) { return it.sort + kMaxItems*kKeyBins; }
__device__ __forceinline__ unsigned *sort_cursor(const Items &it) { return it.sort + 2*kMaxItems*kKeyBins; }

__device__ __forceinline__ int item_id(const Items &it, int item) {
  return it.job[item]*ODHIP_MAX_BANDS + it.band[item];
}

__global__ __launch_bounds__(256) void k_hist(Items it) {
  __shared__ unsigned h[kKeyBins];
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = it.jobs[it.job[item]];
  const unsigned short *keys = jb.keys + (long)it.band[item]*jb.nblocks;
  for (int b = threadIdx.x; b < kKeyBins; b += 256) h[b] = 0;
  __syncthreads();
  const long start = (long)(blockIdx.x - it.wg_start[item])*kSortChunk;
  const long end = start + kSortChunk < jb.nblocks ? start + kSortChunk : jb.nblocks;
  for (long i = start + threadIdx.x; i < end; i += 256) atomicAdd(&h[keys[i]], 1u);
  __syncthreads();
  unsigned *gh = sort_hist(it) + item_id(it, item)*kKeyBins;
  for (int b = threadIdx.x; b < kKeyBins; b += 256) {
    if (h[b]) atomicAdd(&gh[b], h[b]);
  }
}

/* One workgroup per item: exclusive prefix sum of the histogram; clears the
   histogram and the scatter cursors for the next use. */
__global__ __launch_bounds__(256) void k_prefix(Items it) {
  __shared__ unsigned part[256];
  const int id = item_id(it, blockIdx.x);
  unsigned *gh = sort_hist(it) + id*kKeyBins;
  unsigned c[kKeyBins/256];
  unsigned sum = 0;
  for (int i = 0; i < kKeyBins/256; i++) {
    c[i] = gh[threadIdx.x*(kKeyBins/256) + i];
    gh[threadIdx.x*(kKeyBins/256) + i] = 0;
    sum += c[i];
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const unsigned t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned run = part[threadIdx.x] - sum;
  for (int i = 0; i < kKeyBins/256; i++) {
    sort_binstart(it)[id*kKeyBins + threadIdx.x*(kKeyBins/256) + i] = run;
    sort_cursor(it)[id*kKeyBins + threadIdx.x*(kKeyBins/256) + i] = 0;
    run += c[i];
  }
}

__global__ __launch_bounds__(256) void k_scatter(Items it) {
  __shared__ unsigned h[kKeyBins];
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = it.jobs[it.job[item]];
  const long ibase = (long)it.band[item]*jb.nblocks;
  const unsigned short *keys = jb.keys + ibase;
  for (int b = threadIdx.x; b < kKeyBins; b += 256) h[b] = 0;
  __syncthreads();
  const long start = (long)(blockIdx.x - it.wg_start[item])*kSortChunk;
  int key[kSortChunk/256];
  unsigned rank[kSortChunk/256];
#pragma unroll
  for (int t = 0; t < kSortChunk/256; t++) {
    const long i = start + t*256 + threadIdx.x;
    key[t] = -1;
    if (i < jb.nblocks) {
      key[t] = keys[i];
      rank[t] = atomicAdd(&h[key[t]], 1u);
    }
  }
  __syncthreads();
  const int id = item_id(it, item);
  for (int b = threadIdx.x; b < kKeyBins; b += 256) {
    if (h[b]) h[b] = sort_binstart(it)[id*kKeyBins + b] + atomicAdd(&sort_cursor(it)[id*kKeyBins + b], h[b]);
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < kSortChunk/256; t++) {
    if (key[t] >= 0) jb.ids[ibase + h[key[t]] + rank[t]] = (unsigned)(start + t*256 + threadIdx.x);
  }
}

/* ---- search: one band per lane over the sorted order --------------------------
   Every lane reads its band's x16 with 16-byte loads, unpacks |x| << 16 into
   its LDS column, searches both candidates, and writes the signed pulses
   (int16, src/pvq_encoder.c:220-222) with 16-byte stores and the second half
   of the band record as one 32-byte sector.  No cross-lane traffic.

   A wavefront can handle NB groups of 64 sorted positions, strided by the
   item's wavefront count (group g = b*nwaves + w: one group from each of NB
   weight classes), with the dependent loads of a group (sorted index -> record
   head and x16) issued one / two groups ahead.  Measured on MI355X, NB = 1 is
   the fastest for every band size (NB = 4 / 2: +30% on the short bands - the
   extra live registers cost more occupancy than the prefetch buys), so that
   is what the launches use; the exposed latency that motivated it turned out
   to be eight serialised loads of the 1/sqrt table (fixed above). */
template <int NL, int PAD>
struct BandFetch {
  static constexpr int NV = (NL + PAD)/8;          /* 16-byte groups of int16 */
  int4 head[2];
  int4 x[NV];
};

template <int PRICE>
__device__ __forceinline__ int choose_core(const Items &it, int job, const DJob &jb, long sb, int band,
 const RecHead &hd, double best_cost, int yy0, int yy1, int mom0, int mom1, double dist0, double dist1,
 const double *given);

/* N = band size, S = lanes per band (pvq_lane.cuh: S = 2 for the 128-coefficient
   band), NB = groups per wavefront. */
template <int N, int S, int NB>
__global__ __launch_bounds__(kWave, (S == 2 ? 2 : 1)) void k_search(Items it) {
  constexpr int NL = N/S;                   /* coefficients per lane */
  constexpr int PAD = N == 15 ? 1 : 0;      /* leading DC slot */
  constexpr int NV = BandFetch<NL, PAD>::NV;
  constexpr int SLOTS = kWave/S;            /* bands per wavefront */
  static_assert(N % S == 0 && (NL + PAD)%8 == 0, "a lane holds whole 16-byte groups");
  static_assert(S == 1 || PAD == 0, "pair mode has no padded band");
  extern __shared__ __attribute__((aligned(16))) double lds_d[];
  double *rsq = lds_d;                                   /* [kRsqN]  */
  uint32_t *pk = (uint32_t *)(rsq + kRsqN);              /* [NL][64] */
  const int lane = threadIdx.x;
  {
    /* all eight loads in flight before the first LDS store (the compiler keeps
       a load -> wait -> store order otherwise: eight exposed latencies) */
    double r[kRsqN/kWave];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) r[i] = gRsqTable[i*kWave + lane];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) rsq[i*kWave + lane] = r[i];
  }
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = it.jobs[it.job[item]];
  const int band = it.band[item];
  const int half = S == 2 ? lane & 1 : 0;
  const int off = jb.off[band] - PAD + half*NL;
  const long nblocks = jb.nblocks;
  const int len = jb.len;
  const int nb_bands = jb.nb_bands;
  const unsigned *const ids = jb.ids + (long)band*nblocks;
  odhip_pvq_band *const recs = jb.rec + band;
  const int16_t *const x16 = jb.x16 + off;
  int16_t *const yout = jb.y + off;
  double *const cosd = jb.cos_dist;
  const long stride = (long)(it.wg_start[item + 1] - it.wg_start[item])*SLOTS;
  const long spos0 = (long)(blockIdx.x - it.wg_start[item])*SLOTS + lane/S;
  /* prologue: indices of groups 0 and 1, data of group 0 */
  unsigned blk_next = spos0 < nblocks ? ids[spos0] : 0;
  unsigned blk_next2 = NB > 1 && spos0 + stride < nblocks ? ids[spos0 + stride] : 0;
  BandFetch<NL, PAD> nx;
  {
    const int4 *hp = reinterpret_cast<const int4 *>(recs + (long)blk_next*nb_bands);
    const int4 *xp = reinterpret_cast<const int4 *>(x16 + (long)blk_next*len);
    nx.head[0] = hp[0];
    nx.head[1] = hp[1];
#pragma unroll
    for (int v = 0; v < NV; v++) nx.x[v] = xp[v];
  }
  __syncthreads();   /* the 1/sqrt table */
#pragma unroll 1
  for (int b = 0; b < NB; b++) {
    const long spos = spos0 + b*stride;
    const bool live = spos < nblocks;
    const long blk = blk_next;
    const BandFetch<NL, PAD> cur = nx;
    if (b + 1 < NB) {
      /* data of group b+1 (its index arrived during the previous iteration),
         index of group b+2 */
      blk_next = blk_next2;
      const int4 *hp = reinterpret_cast<const int4 *>(recs + (long)blk_next*nb_bands);
      const int4 *xp = reinterpret_cast<const int4 *>(x16 + (long)blk_next*len);
      nx.head[0] = hp[0];
      nx.head[1] = hp[1];
#pragma unroll
      for (int v = 0; v < NV; v++) nx.x[v] = xp[v];
      blk_next2 = b + 2 < NB && spos + 2*stride < nblocks ? ids[spos + 2*stride] : 0;
    }
    RecHead hd;
    hd.cg = cur.head[0].x;
    hd.gain[0] = cur.head[0].y;
    hd.gain[1] = cur.head[0].z;
    hd.k[0] = (int16_t)(cur.head[0].w & 0xffff);
    hd.k[1] = cur.head[0].w >> 16;
    hd.flags[0] = live ? cur.head[1].x & 0xff : 0;
    hd.flags[1] = live ? cur.head[1].x >> 8 & 0xff : 0;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int d[4] = {cur.x[v].x, cur.x[v].y, cur.x[v].z, cur.x[v].w};
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int j0 = v*8 + 2*t - PAD;
        if (j0 >= 0) pk[j0*kWave + lane] = (uint32_t)abs((int)(short)d[t]) << 16;
        pk[(j0 + 1)*kWave + lane] = (uint32_t)abs(d[t] >> 16) << 16;
      }
    }
    LaneSearch st;
    od_lane_prepare<NL, S>(st, pk, lane);
    const int32_t cg = hd.cg;
    const double s2 = (1./256)*(1./256);
    int prev_k = 0;
    int yy0 = 0;
    int yy1 = 0;
    int mom0 = 0;
    int mom1 = 0;
    double dist0 = 0;
    double dist1 = 0;
    /* the candidate in LDS -> signed int16 pulses, 16 bytes at a time into `sink`; returns the moment */
    auto pack = [&](bool on, auto &&sink) {
      int mom = 0;
      const int4 *xp = reinterpret_cast<const int4 *>(x16 + blk*len);
#pragma unroll
      for (int v = 0; v < NV; v++) {
        /* signs: from the registers for the short bands, re-read (L2) for the
           128-coefficient band, whose registers are better spent on the
           pipelined search loops */
        const int4 q = N > 32 ? xp[v] : cur.x[v];
        const int d[4] = {q.x, q.y, q.z, q.w};
        int o[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int j0 = v*8 + 2*t - PAD;
          const int y0 = j0 >= 0 && on ? (int)(pk[j0*kWave + lane] >> 1 & 0x7fffu) : 0;
          const int y1 = on ? (int)(pk[(j0 + 1)*kWave + lane] >> 1 & 0x7fffu) : 0;
          const int s0 = (int)(short)d[t] >> 31;
          const int s1 = d[t] >> 31;
          o[t] = (((y0 ^ s0) - s0) & 0xffff) | ((y1 ^ s1) - s1) << 16;
          mom += (half*NL + j0)*y0 + (half*NL + j0 + 1)*y1;     /* y0 = 0 where j0 < 0 */
        }
        sink(v, make_int4(o[0], o[1], o[2], o[3]));
      }
      return mom;
    };
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
      /* selects, not hd.x[c]: a dynamically indexed local array lives in scratch */
      const bool on = (c ? hd.flags[1] : hd.flags[0]) == 1;
      const int k = c ? hd.k[1] : hd.k[0];
      const int gain = c ? hd.gain[1] : hd.gain[0];
      const int32_t qcg = odq_shl32(gain, ODQ_CGAIN_SHIFT);
      const double g2 = (qcg*(double)cg)*s2;
      const bool fresh = !(prev_k > 0 && prev_k <= k);
      const double cos_dist = od_lane_search<NL, S>(st, pk, rsq, lane, half, on, fresh, k, g2,
       it.lambda, it.reserved != 0);
      /* src/pvq_encoder.c:586,:593-595; a slot that is not in use has distortion 0 */
      int yyc = 0;
      double distc = gain ? ((1.4*(qcg - cg))*(qcg - cg))*s2 : 0.;
      if (on) {
        prev_k = k;
        yyc = (int)st.yy;
        distc = ((1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*cos_dist))*s2;
      }
      if (c) {
        yy1 = yyc;
        dist1 = distc;
      }
      else {
        yy0 = yyc;
        dist0 = distc;
      }
      /* od_pvq_rate's centre-of-mass sum SUM i*|y_i| (src/pvq_encoder.c:258-259), taken
         while the pulses pass through on their way out: the priced choice then never
         reads the vectors again.  With the fused choice the second candidate's pulses stay
         in LDS until the decision (below). */
      int mom = 0;
      if (live) {
        if (cosd && half == 0) cosd[2*(blk*nb_bands + band) + c] = on ? cos_dist : 0.;
        if (!(c && it.fuse)) {
          int4 *yo = reinterpret_cast<int4 *>(yout + ((long)c*nblocks + blk)*len);
          mom = pack(on, [&](int v, int4 q) { yo[v] = q; });
        }
      }
      if (S == 2) mom += row_mov<OD_DPP_XOR1>(mom);    /* the two halves of the band */
      if (c) mom1 = mom;
      else mom0 = mom;
    }
    int4 ysec[NV];
    if (it.fuse) {
      int mom = 0;
      if (live) mom = pack(hd.flags[1] == 1, [&](int v, int4 q) { ysec[v] = q; });
      if (S == 2) mom += row_mov<OD_DPP_XOR1>(mom);
      mom1 = mom;
    }
    /* The priced choice right here, from the registers (odhip_pvq_noref_bands_priced_multi).  What nobody
       reads is not written: the second candidate's pulses go out only when it was chosen, and - like the
       second half of the record (sums, distortions, moments) - when the decision was a close call, which
       the host-libm resolve decides again from the record and may turn over.  (The first candidate's
       pulses left before the second search overwrote them.) */
    int res = 3;
    if (it.fuse) {
      if (live && half == 0) {
        res = choose_core<1>(it, it.job[item], jb, blk*nb_bands + band, band, hd,
         __hiloint2double(cur.head[1].w, cur.head[1].z), yy0, yy1, mom0, mom1, dist0, dist1, nullptr);
      }
      if (S == 2) {
        const int other = row_mov<OD_DPP_XOR1>(res);
        if (half) res = other;
      }
      if (live && res) {
        int4 *yo = reinterpret_cast<int4 *>(yout + (nblocks + blk)*len);
#pragma unroll
        for (int v = 0; v < NV; v++) yo[v] = ysec[v];
      }
    }
    if (live && half == 0 && (res & 2)) {
      int4 *out = reinterpret_cast<int4 *>(recs + blk*nb_bands) + 2;
      out[0] = make_int4(yy0, yy1, __double2loint(dist0), __double2hiint(dist0));
      out[1] = make_int4(__double2loint(dist1), __double2hiint(dist1), mom0, mom1);
    }
  }
}

/* ---- corner bands decided where they are prepared ---------------------------------
   odhip_pvq_noref_bands_priced_multi: band 0 of every block (15 coefficients) and bands 1, 2
   of every block of 8x8 and up (8 each) - 4.2 of the 5.5 M bands of a 16-frame step - never
   leave the lane that prepared them.  The block's corner is in registers (k_prep_corner);
   the lane scales it, forms the band's record head in registers, runs both K-pulse searches
   (od_lane_search, |x| and pulses in its LDS column as in k_search), makes the priced choice
   and writes the choice record and the pulses that are read: no x16, no band record, no sort
   key, no sorted index for these bands - the two-pass stage wrote and re-read ~300 bytes per
   band for them, 16 bytes at a time in sorted order.  Natural block order instead of the
   sort by pulse class costs these short bands 3-12 % (measured with the sort keys forced
   equal); the records of a band listed as a close call are written whole, for the resolve.
   Band 3 of the larger blocks (32 coefficients) stays with the sorted two-pass stage. */
/* N = band size, S = lanes per band (2: the lane pair of pvq_lane.cuh, 4: the quad; `half` = the lane's
   index in its group, each lane holding NL = N/S coefficients in xs). */
/* MASK: the lane's LDS column already holds |x| << 16 and the signs come as one bit per coefficient in
   `sg` (a band too long to keep signed in registers across the searches); xs is not read. */
template <int N, int S = 1, bool MASK = false>
__device__ __forceinline__ void od_decide_band(const Items &it, int job, const DJob &jb, int band, long blk,
 bool live, const BandHead &bh, const int *xs, uint32_t *pk, const double *rsq, int lane, int half = 0,
 unsigned long long sg = 0) {
  constexpr int NL = N/S;
  constexpr int PAD = N == 15 ? 1 : 0;
  constexpr int NV = (NL + PAD)/8;
  /* a long band's first candidate leaves as it is packed (32 registers not held across the second
     search); a short one waits for the decision */
  constexpr bool EARLY0 = NL >= 64 || (S == 4 && NL >= 32);
  static_assert(S == 1 || PAD == 0, "group mode has no padded band");
  RecHead hd;
  hd.cg = bh.h0.x;
  hd.gain[0] = bh.h0.y;
  hd.gain[1] = bh.h0.z;
  hd.k[0] = (int16_t)(bh.h0.w & 0xffff);
  hd.k[1] = bh.h0.w >> 16;
  hd.flags[0] = live ? bh.h1.x & 0xff : 0;
  hd.flags[1] = live ? bh.h1.x >> 8 & 0xff : 0;
  /* flags == 2: the reference would search this candidate; its K does not fit (od_krange.cuh).  In group mode
     every lane of the group holds the same head: the first one counts */
  if (live && half == 0 && (bh.h1.x & 0x0202)) atomicAdd(jb.krange, 1u);
  if constexpr (!MASK) {
#pragma unroll
    for (int j = 0; j < NL; j++) pk[j*kWave + lane] = (uint32_t)abs(xs[j]) << 16;
  }
  LaneSearch st;
  od_lane_prepare<NL, S>(st, pk, lane);
  const int32_t cg = hd.cg;
  const double s2 = (1./256)*(1./256);
  const int off = jb.off[band] - PAD + half*NL;
  const long nblocks = jb.nblocks;
  const int len = jb.len;
  int prev_k = 0;
  int yyv[2] = {0, 0};
  int momv[2] = {0, 0};
  double distv[2] = {0, 0};
  int4 yq[2][EARLY0 ? 1 : NV];
  int4 ylate[EARLY0 ? NV : 1];
#pragma unroll 1
  for (int c = 0; c < 2; c++) {
    const bool on = (c ? hd.flags[1] : hd.flags[0]) == 1;
    const int k = c ? hd.k[1] : hd.k[0];
    const int gain = c ? hd.gain[1] : hd.gain[0];
    const int32_t qcg = odq_shl32(gain, ODQ_CGAIN_SHIFT);
    const double g2 = (qcg*(double)cg)*s2;
    const bool fresh = !(prev_k > 0 && prev_k <= k);
    const double cos_dist = od_lane_search<NL, S>(st, pk, rsq, lane, half, on, fresh, k, g2, it.lambda,
     it.reserved != 0);
    /* src/pvq_encoder.c:586,:593-595; a slot that is not in use has distortion 0 */
    int yyc = 0;
    double distc = gain ? ((1.4*(qcg - cg))*(qcg - cg))*s2 : 0.;
    if (on) {
      prev_k = k;
      yyc = (int)st.yy;
      distc = ((1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*cos_dist))*s2;
    }
    if (live && half == 0 && jb.cos_dist) jb.cos_dist[2*(blk*jb.nb_bands + band) + c] = on ? cos_dist : 0.;
    int mom = 0;
    int4 q[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
      int o[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int j0 = v*8 + 2*t - PAD;
        const int y0 = j0 >= 0 && on ? (int)(pk[(j0 < 0 ? 0 : j0)*kWave + lane] >> 1 & 0x7fffu) : 0;
        const int y1 = on ? (int)(pk[(j0 + 1)*kWave + lane] >> 1 & 0x7fffu) : 0;
        int s0;
        int s1;
        if constexpr (MASK) {
          s0 = -(int)(sg >> (j0 < 0 ? 0 : j0) & 1u);
          s1 = -(int)(sg >> (j0 + 1) & 1u);
        }
        else {
          s0 = j0 >= 0 ? xs[j0 < 0 ? 0 : j0] >> 31 : 0;
          s1 = xs[j0 + 1] >> 31;
        }
        o[t] = (((y0 ^ s0) - s0) & 0xffff) | ((y1 ^ s1) - s1) << 16;
        mom += (half*NL + j0)*y0 + (half*NL + j0 + 1)*y1;          /* y0 = 0 where j0 < 0 */
      }
      q[v] = make_int4(o[0], o[1], o[2], o[3]);
    }
    mom = od_grp_add<S>(mom);                /* the parts of the band */
    /* selects, not [c]: dynamically indexed locals live in scratch */
    if (c) {
      yyv[1] = yyc;
      distv[1] = distc;
      momv[1] = mom;
#pragma unroll
      for (int v = 0; v < NV; v++) {
        if constexpr (EARLY0) ylate[v] = q[v];
        else yq[1][v] = q[v];
      }
    }
    else {
      yyv[0] = yyc;
      distv[0] = distc;
      momv[0] = mom;
      if constexpr (EARLY0) {
        if (live) {
          int4 *yo = reinterpret_cast<int4 *>(jb.y + off + blk*len);
#pragma unroll
          for (int v = 0; v < NV; v++) yo[v] = q[v];
        }
      }
      else {
#pragma unroll
        for (int v = 0; v < NV; v++) yq[0][v] = q[v];
      }
    }
  }
  int res = 0;
  if (live && half == 0) {
    const double dist0 = __hiloint2double(bh.h1.w, bh.h1.z);
    res = choose_core<1>(it, job, jb, blk*jb.nb_bands + band, band, hd, dist0, yyv[0], yyv[1], momv[0],
     momv[1], distv[0], distv[1], nullptr);
  }
  if constexpr (S > 1) res = od_grp_bcast<S, 0>(res);      /* decided by lane 0 of the group */
  if (!live) return;
  /* sel | close << 1: the chosen candidate's pulses; both, and the whole record, for a close call */
  if constexpr (!EARLY0) {
    if (!(res & 1) || (res & 2)) {
      int4 *yo = reinterpret_cast<int4 *>(jb.y + off + blk*len);
#pragma unroll
      for (int v = 0; v < NV; v++) yo[v] = yq[0][v];
    }
  }
  if (res) {
    int4 *yo = reinterpret_cast<int4 *>(jb.y + off + (nblocks + blk)*len);
#pragma unroll
    for (int v = 0; v < NV; v++) yo[v] = EARLY0 ? ylate[v] : yq[1][v];
  }
  if ((res & 2) && half == 0) {
    int4 *out = reinterpret_cast<int4 *>(jb.rec + blk*jb.nb_bands + band);
    out[0] = bh.h0;
    out[1] = bh.h1;
    out[2] = make_int4(yyv[0], yyv[1], __double2loint(distv[0]), __double2hiint(distv[0]));
    out[3] = make_int4(__double2loint(distv[1]), __double2hiint(distv[1]), momv[0], momv[1]);
  }
}

/* PART 0: band 0 of every block of every size (the 4x4 corner); PART 1: bands 1 and 2 of the blocks
   of 8x8 and up (x 4..7 of rows 0..1 and x 0..1 of rows 4..7, gen/od_scan_tables.h).  Two kernels
   rather than one per block: searched in one kernel the three bands hold 150 VGPRs (three waves
   per SIMD), apart 121 and ~100 (four and five). */
template <int PART>
__global__ __launch_bounds__(kWave) OD_DECIDE_OCC_ATTR void k_decide_corner(Items it) {
  extern __shared__ __attribute__((aligned(16))) double lds_d[];
  OD_SEARCH_VGPR_FLOOR();
  double *rsq = lds_d;                                   /* [kRsqN]  */
  uint32_t *pk = (uint32_t *)(rsq + kRsqN);              /* [15][64] */
  const int lane = threadIdx.x;
  {
    double r[kRsqN/kWave];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) r[i] = gRsqTable[i*kWave + lane];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) rsq[i*kWave + lane] = r[i];
  }
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const DJob &jb = it.jobs[job];
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x);
  const int16_t *const qmp = jb.qm;
  const int w = jb.w;
  constexpr int C = 8;
  int r[C*C];       /* raster, row-major; only what the part's bands cover is loaded */
  if constexpr (PART == 0) {
#pragma unroll
    for (int y = 0; y < 4; y++) {
      const int4 q = *reinterpret_cast<const int4 *>(bp.src + y*w);
      r[y*C] = q.x;
      r[y*C + 1] = q.y;
      r[y*C + 2] = q.z;
      r[y*C + 3] = q.w;
    }
  }
  else {
#pragma unroll
    for (int y = 0; y < 2; y++) {
      const int4 q = *reinterpret_cast<const int4 *>(bp.src + y*w + 4);
      r[y*C + 4] = q.x;
      r[y*C + 5] = q.y;
      r[y*C + 6] = q.z;
      r[y*C + 7] = q.w;
    }
#pragma unroll
    for (int y = 4; y < 8; y++) {
      const int2 q = *reinterpret_cast<const int2 *>(bp.src + y*w);
      r[y*C] = q.x;
      r[y*C + 1] = q.y;
    }
  }
  __syncthreads();   /* the 1/sqrt table */
  int x16[32];
  int32_t cgs[3] = {0, 0, 0};
  auto scale_band = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int kOff[4] = {1, 16, 24, 32};
    constexpr int off = kOff[b];
    constexpr int n = kOff[b + 1] - off;
    /* od_vector_log_mag, src/pvq.c:472-484 */
    int sum = 0;
    od_static_for<off, off + n>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int t = (int16_t)(r[OD_SCAN_XY[j][1]*C + OD_SCAN_XY[j][0]] >> 8);
      sum += t*t;
    });
    int xshift = 8 + 1 + odq_ilog(n + sum)/2 - 15;
    xshift = xshift > 0 ? xshift : 0;
    int acc = 0;
    od_static_for<off, off + n>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      x16[j] = (int16_t)odq_shr_round(r[OD_SCAN_XY[j][1]*C + OD_SCAN_XY[j][0]]*qmp[j],
       ODQ_QM_SHIFT + xshift);
      acc += x16[j]*x16[j];
    });
    const int qb = bp.blk >= jb.split_blk ? jb.q2[b] : jb.q[b];
    int32_t g;
    cgs[b] = odq_gain_from_acc(acc, qb, jb.beta[b], xshift, &g);
  };
  if constexpr (PART == 0) {
    scale_band(std::integral_constant<int, 0>{});
    od_decide_band<15>(it, job, jb, 0, bp.blk, bp.live, od_band_head(jb.beta[0], 15, cgs[0]), x16 + 1, pk, rsq,
     lane);
  }
  else {
    /* band 2 waits as packed int16 pairs and its gain while band 1 is searched */
    scale_band(std::integral_constant<int, 2>{});
    int packed[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      packed[i] = (x16[24 + 2*i] & 0xffff) | x16[25 + 2*i] << 16;
      /* materialise here: sunk below the search, the raster values it is made of would stay live */
      asm volatile("" : "+v"(packed[i]));
    }
    asm volatile("" : "+v"(cgs[2]));
    scale_band(std::integral_constant<int, 1>{});
    od_decide_band<8>(it, job, jb, 1, bp.blk, bp.live, od_band_head(jb.beta[1], 8, cgs[1]), x16 + 16, pk, rsq, lane);
    int xs[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      xs[2*i] = (int)(short)packed[i];
      xs[2*i + 1] = packed[i] >> 16;
    }
    od_decide_band<8>(it, job, jb, 2, bp.blk, bp.live, od_band_head(jb.beta[2], 8, cgs[2]), xs, pk, rsq, lane);
  }
}

/* The 32-coefficient bands the same way (band 3 of every block of 8x8 and up, bands 4 and 5 of 16x16 and
   up), two lanes per band as in k_search<32, 2, 1>: each lane gathers its 16 coding positions, the
   pair shares the sums by DPP.  Natural order costs these bands 1 % (measured). */
__global__ __launch_bounds__(kWave) OD_DECIDE_OCC_ATTR void k_decide_lane32(Items it) {
  constexpr int NL = 16;
  extern __shared__ __attribute__((aligned(16))) double lds_d[];
  OD_SEARCH_VGPR_FLOOR();
  double *rsq = lds_d;                                   /* [kRsqN]  */
  uint32_t *pk = (uint32_t *)(rsq + kRsqN);              /* [16][64] */
  const int lane = threadIdx.x;
  {
    double r[kRsqN/kWave];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) r[i] = gRsqTable[i*kWave + lane];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) rsq[i*kWave + lane] = r[i];
  }
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const DJob &jb = it.jobs[job];
  const int band = it.band[item];
  const int half = lane & 1;
  const int off = jb.off[band];
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*(kWave/2) + (lane >> 1));
  const int w = jb.w;
  const int16_t *const qmp = jb.qm + off + half*NL;
  int v[NL];
  int qm[NL];
  /* Round 5 (as k_decide_pair128): the band's 32 coefficients come in as 16 aligned 8-byte row segments, 8 per
     lane of the pair, are parked in the pair's two LDS columns and picked out in coding order (kSlot32). */
  {
    const int geo = (off >> 5) - 1;
    int2 raw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int s0 = kSeg32[geo][i];
      const int s1 = kSeg32[geo][8 + i];
      const int sg_ = half ? s1 : s0;
      raw[i] = *reinterpret_cast<const int2 *>(bp.src + (sg_ >> 8)*w + (sg_ & 255));
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      pk[(2*i)*kWave + lane] = (uint32_t)raw[i].x;
      pk[(2*i + 1)*kWave + lane] = (uint32_t)raw[i].y;
    }
    const int pair0 = lane & ~1;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const int t0 = kSlot32[geo][j];
      const int t1 = kSlot32[geo][NL + j];
      const int sl = half ? t1 : t0;
      v[j] = (int)pk[(sl & 15)*kWave + pair0 + (sl >> 4)];
      qm[j] = qmp[j];
    }
  }
  __syncthreads();   /* the 1/sqrt table */
  int sum = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int t = (int16_t)(v[j] >> 8);
    sum += t*t;
  }
  sum += od_pair_swap(sum);
  int xshift = 8 + 1 + odq_ilog(2*NL + sum)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int xs[NL];
  int acc = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    xs[j] = (int16_t)odq_shr_round(v[j]*qm[j], ODQ_QM_SHIFT + xshift);
    acc += xs[j]*xs[j];
  }
  acc += od_pair_swap(acc);
  const int qb = bp.blk >= jb.split_blk ? jb.q2[band] : jb.q[band];
  int32_t g;
  const int32_t cg = odq_gain_from_acc(acc, qb, jb.beta[band], xshift, &g);
  od_decide_band<32, 2>(it, job, jb, band, bp.blk, bp.live, od_band_head(jb.beta[band], 32, cg), xs, pk, rsq, lane,
   half);
}

/* ... and the 128-coefficient bands: a lane pair per band as k_search<128, 2, 1>, each lane gathering its
   64 coding positions straight into its LDS column (the raw coefficients wait in registers only until the
   band's scaling shift is known; the signs as 64 bits).  The sort by pulse class buys these bands -2 % on
   the default content and +8 % on the natural one (measured with the keys forced equal) against 130 us of
   preparation and sort and a 600 MB round trip of scaled vectors. */
__global__ __launch_bounds__(kWave, 2) void k_decide_pair128(Items it) {
  constexpr int NL = 64;
  extern __shared__ __attribute__((aligned(16))) double lds_d[];
  double *rsq = lds_d;                                   /* [kRsqN]  */
  uint32_t *pk = (uint32_t *)(rsq + kRsqN);              /* [64][64] */
  const int lane = threadIdx.x;
  {
    double r[kRsqN/kWave];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) r[i] = gRsqTable[i*kWave + lane];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) rsq[i*kWave + lane] = r[i];
  }
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const DJob &jb = it.jobs[job];
  const int band = it.band[item];
  const int half = lane & 1;
  const int off = jb.off[band];
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*(kWave/2) + (lane >> 1));
  const int w = jb.w;
  const int16_t *const qmp = jb.qm + off + half*NL;
  int v[NL];
  int sum = 0;
  /* Round 5: the band's 128 coefficients come in as 32 aligned 16-byte row segments, 16 per lane of the pair
     (whole sectors, every byte fetched is used; the 64 four-byte gathers per lane this replaces touched a
     different 64-byte line per lane and instruction: 951 MB of HBM traffic per launch for 198 MB of
     coefficients, profiles/r4_pmc_traffic.json), are parked in the pair's two LDS columns in that staged order
     (the column is free until the scaled vector is written below), and each lane picks its 64 coding positions
     out of them (kSlot128; both lanes of a pair are in one wavefront: no barrier). */
  {
    const int geo = (off >> 7) - 1;
    int4 raw[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int s0 = kSeg128[geo][i];
      const int s1 = kSeg128[geo][16 + i];
      const int sg_ = half ? s1 : s0;
      raw[i] = *reinterpret_cast<const int4 *>(bp.src + (sg_ >> 8)*w + (sg_ & 255));
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      pk[(4*i)*kWave + lane] = (uint32_t)raw[i].x;
      pk[(4*i + 1)*kWave + lane] = (uint32_t)raw[i].y;
      pk[(4*i + 2)*kWave + lane] = (uint32_t)raw[i].z;
      pk[(4*i + 3)*kWave + lane] = (uint32_t)raw[i].w;
    }
    const int pair0 = lane & ~1;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const int t0 = kSlot128[geo][j];
      const int t1 = kSlot128[geo][NL + j];
      const int sl = half ? t1 : t0;
      v[j] = (int)pk[(sl & 63)*kWave + pair0 + (sl >> 6)];
    }
  }
  __syncthreads();   /* the 1/sqrt table */
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int t = (int16_t)(v[j] >> 8);
    sum += t*t;
  }
  sum += od_pair_swap(sum);
  int xshift = 8 + 1 + odq_ilog(2*NL + sum)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  unsigned long long sg = 0;
  int acc = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int x = (int16_t)odq_shr_round(v[j]*qmp[j], ODQ_QM_SHIFT + xshift);
    acc += x*x;
    sg |= (unsigned long long)(x < 0) << j;
    pk[j*kWave + lane] = (uint32_t)abs(x) << 16;
  }
  acc += od_pair_swap(acc);
  const int qb = bp.blk >= jb.split_blk ? jb.q2[band] : jb.q[band];
  int32_t g;
  const int32_t cg = odq_gain_from_acc(acc, qb, jb.beta[band], xshift, &g);
  od_decide_band<128, 2, true>(it, job, jb, band, bp.blk, bp.live, od_band_head(jb.beta[band], 128, cg), nullptr, pk,
   rsq, lane, half, sg);
}

#ifdef ODHIP_EXPERIMENTS
/* The 128-coefficient bands on a QUAD of lanes per band (round 5; k_decide_pair128 above is the pair form it
   was measured against, experiments build only): 32 coding positions per lane, 8 KiB of LDS columns per wavefront instead
   of 16.  The pair form held two wavefronts per SIMD (20 KiB each) and ran at 0.67 of VALU issue: its
   argmax scans are chains of dependent selects, and two wavefronts do not cover a dependency chain
   (the same effect the 64x64 transforms showed at 2.25 wavefronts per SIMD).  Same bands
   in flight per CU, twice the wavefronts, every scan half as long; the 64 gathers per lane become 32 and
   the raw coefficients held until the band's shift is known 32 registers instead of 64. */
__global__ __launch_bounds__(kWave, 3) void k_decide_quad128(Items it) {
  constexpr int NL = 32;
  extern __shared__ __attribute__((aligned(16))) double lds_d[];
  double *rsq = lds_d;                                   /* [kRsqN]  */
  uint32_t *pk = (uint32_t *)(rsq + kRsqN);              /* [32][64] */
  const int lane = threadIdx.x;
  {
    double r[kRsqN/kWave];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) r[i] = gRsqTable[i*kWave + lane];
#pragma unroll
    for (int i = 0; i < kRsqN/kWave; i++) rsq[i*kWave + lane] = r[i];
  }
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const DJob &jb = it.jobs[job];
  const int band = it.band[item];
  const int sub = lane & 3;
  const int off = jb.off[band];
  const BlockPos bp = locate(jb, (long)(blockIdx.x - it.wg_start[item])*(kWave/4) + (lane >> 2));
  const int w = jb.w;
  const int16_t *const qmp = jb.qm + off + sub*NL;
  int v[NL];
  int sum = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    /* the scan positions of the four parts are wave-uniform (scalar loads), the lane takes its own */
    const int xa = kScanXY[off + j][0];
    const int ya = kScanXY[off + j][1];
    const int xb = kScanXY[off + NL + j][0];
    const int yb = kScanXY[off + NL + j][1];
    const int xc = kScanXY[off + 2*NL + j][0];
    const int yc = kScanXY[off + 2*NL + j][1];
    const int xd = kScanXY[off + 3*NL + j][0];
    const int yd = kScanXY[off + 3*NL + j][1];
    const int oa = ya*w + xa;
    const int ob = yb*w + xb;
    const int oc = yc*w + xc;
    const int od = yd*w + xd;
    v[j] = bp.src[sub == 0 ? oa : sub == 1 ? ob : sub == 2 ? oc : od];
  }
  __syncthreads();   /* the 1/sqrt table */
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int t = (int16_t)(v[j] >> 8);
    sum += t*t;
  }
  sum = od_grp_add<4>(sum);
  int xshift = 8 + 1 + odq_ilog(4*NL + sum)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  unsigned long long sg = 0;
  int acc = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int x = (int16_t)odq_shr_round(v[j]*qmp[j], ODQ_QM_SHIFT + xshift);
    acc += x*x;
    sg |= (unsigned long long)(x < 0) << j;
    pk[j*kWave + lane] = (uint32_t)abs(x) << 16;
  }
  acc = od_grp_add<4>(acc);
  const int qb = bp.blk >= jb.split_blk ? jb.q2[band] : jb.q[band];
  int32_t g;
  const int32_t cg = odq_gain_from_acc(acc, qb, jb.beta[band], xshift, &g);
  od_decide_band<128, 4, true>(it, job, jb, band, bp.blk, bp.live, od_band_head(jb.beta[band], 128, cg), nullptr, pk,
   rsq, lane, sub, sg);
}

#endif

/* ---- choice: one (block, band) per lane --------------------------------------
   The comparison `cost <= best_cost` of src/pvq_encoder.c:597-609 with
   cost = dist + lambda*rate (rate from the host entropy model, or absent),
   then od_gain_expand (src/pvq.c:766-811) and the synthesis scale of
   od_pvq_synthesis_partial (src/pvq.c:1057-1078).  choice = {sel, qg, scale,
   qshift}. */
/* PRICE = 0: cost = dist + lambda*rate with the host's rate table (or no rate at all);
   PRICE = 1: the rate is od_pvq_rate's closed form (speed > 0) evaluated HERE from the
   candidate's pulses with the device's log.  The device's and the host libm's log may
   differ in the last place, so a comparison whose two costs lie within odq_rate_tol of each
   other is not trusted: the band is listed, odhip_pvq_choose_priced_resolve recomputes its
   rates with the host's libm (the function the reference calls) and k_choose_list decides
   it again; PRICE = 2 is that second decision (rates given per band). */
/* The decision itself, on values: from the record (k_choose) or straight from the
   registers of the search that produced them (k_search with Items::fuse). */
/* od_pvq_rate's pulse part (the whole rate of a no-reference candidate: theta == -1) is a function of
   (sum, k, n) alone: read from tables filled ONCE per device by the function they replace (k_nrate_fill:
   same code, same device log - identical values; pvq_refbands.hip does the same for the with-reference
   stage), two logs and two divisions per candidate otherwise.  Entries [k][sum], k <= NRateTab<N>::K,
   sum <= (N - 1)*K; the depths cover the pulse counts the luma bands of a 1080p keyframe reach at the
   operating points measured except band 0 of the 32x32 / 64x64 blocks (K 100-360: computed inline). */
template <int N> struct NRateTab {
  static constexpr int K = N == 15 ? 128 : N == 8 ? 64 : 80;
  static constexpr int W = (N - 1)*K + 1;
  static constexpr int SIZE = (K + 1)*W;
};
__device__ double gNRate8[NRateTab<8>::SIZE];
__device__ double gNRate15[NRateTab<15>::SIZE];
__device__ double gNRate32[NRateTab<32>::SIZE];
__device__ double gNRate128[NRateTab<128>::SIZE];

template <int N>
__device__ __forceinline__ double *nrate_tab(void) {
  return N == 8 ? gNRate8 : N == 15 ? gNRate15 : N == 32 ? gNRate32 : gNRate128;
}

template <int N>
__global__ void k_nrate_fill(void) {
  constexpr int W = NRateTab<N>::W;
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= NRateTab<N>::SIZE) return;
  const int k = i/W;
  const int sum = i - k*W;
  nrate_tab<N>()[i] = k == 0 ? 0. : odq_pvq_rate_pulses(sum, k, N);
}

/* odq_pvq_rate_fast(sum, k, n, qg, 0, -1, 0, 1, 0): the join adds nothing when theta < 0 */
__device__ __forceinline__ double noref_rate(int sum, int k, int n) {
  if (n == 15) {
    if (k <= NRateTab<15>::K) return gNRate15[k*NRateTab<15>::W + sum];
  }
  else if (n == 8) {
    if (k <= NRateTab<8>::K) return gNRate8[k*NRateTab<8>::W + sum];
  }
  else if (n == 32) {
    if (k <= NRateTab<32>::K) return gNRate32[k*NRateTab<32>::W + sum];
  }
  else if (n == 128) {
    if (k <= NRateTab<128>::K) return gNRate128[k*NRateTab<128>::W + sum];
  }
  return odq_pvq_rate_pulses(sum, k, n);
}

/* Returns sel | close << 1. */
template <int PRICE>
__device__ __forceinline__ int choose_core(const Items &it, int job, const DJob &jb, long sb, int band,
 const RecHead &hd, double best_cost, int yy0, int yy1, int mom0, int mom1, double dist0, double dist1,
 const double *given) {
  const int qb = sb/jb.nb_bands >= jb.split_blk ? jb.q2[band] : jb.q[band];
  const int betab = jb.beta[band];
  const double *const rate = jb.rate;
  int4 *const choice = reinterpret_cast<int4 *>(jb.choice);
  int32_t *const qg_out = jb.qg_out;
  const int yys[2] = {yy0, yy1};
  const int moms[2] = {mom0, mom1};
  const double dists[2] = {dist0, dist1};
  int qg = 0;
  int sel = 0;
  bool close = false;
  for (int c = 0; c < 2; c++) {
    if (hd.flags[c] != 1) continue;
    double cost = dists[c];
    if (PRICE == 0) {
      if (rate) cost = cost + it.lambda*rate[2*sb + c];
    }
    else if (PRICE == 1) {
      const int n = jb.off[band + 1] - jb.off[band];
      cost = cost + it.lambda*noref_rate(moms[c], hd.k[c], n);
      const double d = cost - best_cost;
      if ((d < 0 ? -d : d) <= it.tol_scale*odq_rate_tol(cost, best_cost)) close = true;
    }
    else cost = cost + it.lambda*given[c];
    if (cost <= best_cost) {
      best_cost = cost;
      qg = hd.gain[c];
      sel = c;
    }
  }
  if (PRICE == 1 && close) {
    const unsigned slot = atomicAdd(it.pcount, 1u);
    if (slot < (unsigned)kPUncCap) {
      PUnc e;
      e.job = job;
      e.sb = (unsigned)sb;
      e.rate[0] = 0;
      e.rate[1] = 0;
      it.plist[slot] = e;
    }
  }
  int32_t scale = 0;
  int qshift = ODQ_QM_INV_SHIFT;
  if (qg != 0) {
    const int32_t g = odq_gain_expand(odq_shl32(qg, ODQ_CGAIN_SHIFT), qb, betab);
    const int yy = sel ? yys[1] : yys[0];
    int gshift = odq_ilog(g) - 14;
    gshift = gshift > 0 ? gshift : 0;
    if (yy != 0) {
      int rsqrt_shift;
      const int16_t rsqrt = odq_rsqrt(yy, &rsqrt_shift);
      scale = odq_vshr_round(rsqrt*(int64_t)g, rsqrt_shift + gshift - 16);
    }
    qshift = ODQ_QM_INV_SHIFT - gshift;
  }
  choice[sb] = make_int4(sel, qg, scale, qshift);
  if (qg_out) qg_out[sb] = qg;
  return sel | (close ? 2 : 0);
}

template <int PRICE>
__device__ __forceinline__ void choose_band(const Items &it, int job, const DJob &jb, long sb,
 const double *given) {
  const int band = (int)(sb % jb.nb_bands);
  const int4 *r = reinterpret_cast<const int4 *>(jb.rec + sb);
  const RecHead hd = rec_head_load(jb.rec + sb);
  const int4 r1 = r[1];
  const int4 r2 = r[2];
  const int4 r3 = r[3];
  choose_core<PRICE>(it, job, jb, sb, band, hd, __hiloint2double(r1.w, r1.z), r2.x, r2.y, r3.z, r3.w,
   __hiloint2double(r2.w, r2.z), __hiloint2double(r3.y, r3.x), given);
}

template <int PRICE>
__global__ __launch_bounds__(256) void k_choose(Items it) {
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const DJob &jb = it.jobs[job];
  const long sb = (long)(blockIdx.x - it.wg_start[item])*256 + threadIdx.x;
  if (sb >= jb.nblocks*jb.nb_bands) return;
  choose_band<PRICE>(it, job, jb, sb, nullptr);
}

__global__ __launch_bounds__(kWave) void k_choose_list(Items it, const PUnc *list, int count) {
  const int i = blockIdx.x*kWave + threadIdx.x;
  if (i >= count) return;
  const PUnc e = list[i];
  choose_band<2>(it, e.job, it.jobs[e.job], e.sb, list[i].rate);
}

/* ---- synthesis: four horizontally adjacent coefficients per lane -------------
   x = y*scale (Q16, no rounding), out = SHR_ROUND(x*qm_inv, qshift)
   (src/pvq.c:1081-1092) scattered to raster by the inverse scan; DC passed
   through (keyframes quantise it in the Haar DC tree, src/encode.c:1377);
   positions PVQ never codes are zero (od_init_skipped_coeffs,
   src/state.c:1347-1358).  Every store is a coalesced 16-byte row segment. */
__global__ __launch_bounds__(256) void k_synth(Items it) {
  const int item = find_item(it, blockIdx.x);
  const DJob &jb = it.jobs[it.job[item]];
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
    s_choice[i] = reinterpret_cast<const int4 *>(jb.choice)[blk0*jb.nb_bands + i];
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
          const int yv = jb.y[((long)ch.x*jb.nblocks + blk)*jb.len + j];
          const int32_t xq = (int32_t)((int16_t)yv*(int64_t)ch.z >> 16);
          v = odq_shr_round(xq*jb.qm_inv[j], ch.w);
        }
      }
    }
    out[i] = v;
  }
  *reinterpret_cast<int4 *>(jb.dq + idx) = make_int4(out[0], out[1], out[2], out[3]);
}

/* ---- chroma-from-luma predictions ------------------------------------------------
   od_resample_luma_coeffs (reference src/intra.c:72-109) for 4:2:0 when the luma
   block is at least 8x8 (:97-108): the prediction of the chroma block is the
   upper-left quarter of the decoded luma block's coefficients.  Here the decoded
   luma coefficients are not read from a plane: they are dequantised on the fly
   from the chosen pulses of the luma band stage (od_pvq_synthesis_partial noref,
   src/pvq.c:1081-1092, the arithmetic of k_synth), the DC passed through as
   k_synth passes it.  One output coefficient per thread; the result is written
   `copies` times (Cb and Cr share the prediction). */
struct CflOut {
  od_coeff *ref[kMaxJobs];
  int copies;
};

/* Eight consecutive coding positions of one luma block per thread.  The scan is
   nested: the first n*n coding positions of a 2n x 2n block are its n x n corner
   (for 64x64 blocks only 512 of the corner's 1024 positions are coded; the host
   clears those planes first), and a chunk of eight never straddles a band, so
   pulses, inverse QM and scan positions are 16-byte loads and one choice record
   serves the chunk. */
__global__ __launch_bounds__(256) void k_cfl_ref(Items it, CflOut out) {
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const DJob &jb = it.jobs[job];
  const int N = 4 << jb.bs;
  const int n = N >> 1;
  const int cw = jb.w >> 1;
  const int ncode = n*n < jb.len ? n*n : jb.len;       /* coded positions inside the corner */
  const int cpb = ncode >> 3;
  const long t = (long)(blockIdx.x - it.wg_start[item])*256 + threadIdx.x;
  const long blk = t/cpb;
  if (blk >= jb.nblocks) return;
  const int c0 = (int)(t - blk*cpb) << 3;
  const long per_blocks = (long)jb.bw*jb.bh;
  const int p = (int)(blk/per_blocks);
  const int rem = (int)(blk - p*per_blocks);
  const int by = rem/jb.bw;
  const int bx = rem - by*jb.bw;
  const long per = (long)cw*(jb.h >> 1);
  od_coeff *dst = out.ref[job] + p*per + (long)by*n*cw + bx*n;
  const long copy_stride = per*jb.nplanes;
  const uint4 sc4 = *reinterpret_cast<const uint4 *>(gScanXY + c0);
  const unsigned scw[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
  const int4 ch = reinterpret_cast<const int4 *>(jb.choice)[blk*jb.nb_bands + gBandOf[c0 ? c0 : 1]];
  unsigned yw[4] = {0, 0, 0, 0};
  if (ch.y != 0) {
    const uint4 y4 = *reinterpret_cast<const uint4 *>(jb.y + ((long)ch.x*jb.nblocks + blk)*jb.len + c0);
    yw[0] = y4.x;
    yw[1] = y4.y;
    yw[2] = y4.z;
    yw[3] = y4.w;
  }
  const uint4 q4 = *reinterpret_cast<const uint4 *>(jb.qm_inv + c0);
  const unsigned qw[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const unsigned pk = (scw[e >> 1] >> (16*(e & 1))) & 0xffffu;
    int val;
    if (c0 == 0 && e == 0) val = jb.coef[(long)p*jb.w*jb.h + (long)by*N*jb.w + bx*N];
    else {
      const int yv = (int16_t)(yw[e >> 1] >> (16*(e & 1)));
      const int qmi = (int16_t)(qw[e >> 1] >> (16*(e & 1)));
      val = odq_shr_round(odq_mult16_32_q16(yv, ch.z)*qmi, ch.w);
    }
    od_coeff *d = dst + (long)(pk >> 8)*cw + (pk & 255);
    for (int k = 0; k < out.copies; k++) d[k*copy_stride] = val;
  }
}

/* Decoded coefficient (v, u), v, u < 2, of the 4x4 luma block blk of a level-0
   job: the DC passed through, the others dequantised from the chosen pulses. */
__device__ __forceinline__ int cfl_luma4(const DJob &jb, long blk, int p, int by, int bx, int v, int u) {
  const int c = gInvScan[v*32 + u];
  if (c == 0) return jb.coef[(long)p*jb.w*jb.h + (long)by*4*jb.w + bx*4];
  const int4 ch = reinterpret_cast<const int4 *>(jb.choice)[blk*jb.nb_bands];
  if (ch.y == 0) return 0;
  const int yv = jb.y[((long)ch.x*jb.nblocks + blk)*jb.len + c];
  return odq_shr_round(odq_mult16_32_q16(yv, ch.z)*jb.qm_inv[c], ch.w);
}

/* The 4x4 luma case of od_resample_luma_coeffs (src/intra.c:77-89): the four 4x4
   luma blocks over a 4x4 chroma block are merged by od_tf_up_hv_lp
   (src/tf.c:82-108, OD_HAAR_KERNEL src/tf.h:34-45) and scaled by
   OD_CFL_SCALING4 (src/intra.c:65-70).  One output coefficient per thread: it
   evaluates the Haar kernel of its 2x2 group and keeps its own output. */
__device__ const short kCflScaling4[4][4] = {
  {128, 128, 100, 36}, {128, 80, 71, 35}, {100, 71, 35, 31}, {36, 35, 31, 18}};

__global__ __launch_bounds__(256) void k_cfl_ref_tf(Items it, CflOut out) {
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const DJob &jb = it.jobs[job];
  const int cw = jb.w >> 1;
  const int chh = jb.h >> 1;
  const long t = (long)(blockIdx.x - it.wg_start[item])*256 + threadIdx.x;
  const long per = (long)cw*chh;
  if (t >= per*jb.nplanes) return;
  const int p = (int)(t/per);
  const int rem = (int)(t - p*per);
  const int y = rem/cw;
  const int x = rem - y*cw;
  const int cby = y >> 2;
  const int i = y & 3;
  const int cbx = x >> 2;
  const int j = x & 3;
  const int gy = i >> 1;          /* the loop indices y, x of od_tf_up_hv_lp */
  const int gx = j >> 1;
  int v[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {   /* ll, lh (right block), hl (lower block), hh */
    const int by = 2*cby + (q >> 1);
    const int bx = 2*cbx + (q & 1);
    const long blk = ((long)p*jb.bh + by)*jb.bw + bx;
    v[q] = cfl_luma4(jb, blk, p, by, bx, gy, gx);
  }
  int ll = v[0];
  int lh = v[1];
  int hl = v[2];
  int hh = v[3];
  /* OD_HAAR_KERNEL(ll, hl, lh, hh) */
  ll += lh;
  hh -= hl;
  const int tt = (ll - hh) >> 1;
  hl = tt - hl;
  lh = tt - lh;
  ll -= hl;
  hh += lh;
  const int vswap = gy & 1;
  const int hswap = gx & 1;
  const int row_first = (i & 1) == vswap;       /* rows 2y + vswap hold ll / lh */
  const int col_first = (j & 1) == hswap;       /* columns 2x + hswap hold ll / hl */
  const int val0 = row_first ? (col_first ? ll : lh) : (col_first ? hl : hh);
  const int val = (kCflScaling4[j][i]*val0 + 64) >> 7;
  od_coeff *dst = out.ref[job] + t;
  for (int k = 0; k < out.copies; k++) dst[k*per*jb.nplanes] = val;
}

/* ---- host side ----------------------------------------------------------------- */
odhip_device_once g_tables_once;

int upload_tables_now(void) {
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
  {
    unsigned short seg[3][32];
    unsigned char slot[3][128];
    for (int g = 0; g < 3; g++) {
      const int off = 128*(g + 1);
      /* the distinct (y, x / 4) groups of the band's positions, in raster order */
      int n = 0;
      for (int y = 0; y < 32; y++) {
        for (int x4 = 0; x4 < 8; x4++) {
          int cnt = 0;
          for (int p = 0; p < 128; p++) cnt += OD_SCAN_XY[off + p][1] == y && OD_SCAN_XY[off + p][0]/4 == x4;
          if (cnt == 0) continue;
          if (cnt != 4 || n >= 32) return ODHIP_EFAULT;     /* the region is made of whole aligned segments */
          seg[g][n++] = (unsigned short)(y << 8 | 4*x4);
        }
      }
      if (n != 32) return ODHIP_EFAULT;
      for (int p = 0; p < 128; p++) {
        const int key = OD_SCAN_XY[off + p][1] << 8 | (OD_SCAN_XY[off + p][0] & ~3);
        int q = 0;
        while (seg[g][q] != key) q++;
        slot[g][p] = (unsigned char)(4*q + (OD_SCAN_XY[off + p][0] & 3));
      }
    }
    ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kSeg128), seg, sizeof(seg)));
    ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kSlot128), slot, sizeof(slot)));
  }
  {
    unsigned short seg[3][16];
    unsigned char slot[3][32];
    for (int g = 0; g < 3; g++) {
      const int off = 32*(g + 1);
      int n = 0;
      for (int y = 0; y < 16; y++) {
        for (int x2 = 0; x2 < 8; x2++) {
          int cnt = 0;
          for (int p = 0; p < 32; p++) cnt += OD_SCAN_XY[off + p][1] == y && OD_SCAN_XY[off + p][0]/2 == x2;
          if (cnt == 0) continue;
          if (cnt != 2 || n >= 16) return ODHIP_EFAULT;
          seg[g][n++] = (unsigned short)(y << 8 | 2*x2);
        }
      }
      if (n != 16) return ODHIP_EFAULT;
      for (int p = 0; p < 32; p++) {
        const int key = OD_SCAN_XY[off + p][1] << 8 | (OD_SCAN_XY[off + p][0] & ~1);
        int q = 0;
        while (seg[g][q] != key) q++;
        slot[g][p] = (unsigned char)(2*q + (OD_SCAN_XY[off + p][0] & 1));
      }
    }
    ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kSeg32), seg, sizeof(seg)));
    ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kSlot32), slot, sizeof(slot)));
  }
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gScanXY), packed, sizeof(packed)));
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gInvScan), inv, sizeof(inv)));
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gBandOf), band_of, sizeof(band_of)));
  k_rsq_fill<<<1, kRsqN, 0, 0>>>();
  od_rsqrt_huge_fill_launch();
  k_nrate_fill<8><<<(NRateTab<8>::SIZE + 255)/256, 256, 0, 0>>>();
  k_nrate_fill<15><<<(NRateTab<15>::SIZE + 255)/256, 256, 0, 0>>>();
  k_nrate_fill<32><<<(NRateTab<32>::SIZE + 255)/256, 256, 0, 0>>>();
  k_nrate_fill<128><<<(NRateTab<128>::SIZE + 255)/256, 256, 0, 0>>>();
  ODHIP_TRY(hipDeviceSynchronize());
  return ODHIP_SUCCESS;
}

int upload_tables(void) {
  return odhip_once_per_device(g_tables_once, upload_tables_now);
}

/* Everything the band stage keeps between calls, owned by the calling thread's
   current context (od_ctx.cuh): ONE call sequence may be in flight per context. */
constexpr int kProfSlots = 256;
struct Scratch {
  int16_t *x16 = nullptr;
  size_t x16_cap = 0;     /* elements */
  unsigned short *keys = nullptr;
  unsigned *ids = nullptr;
  size_t band_cap = 0;    /* (band, block) pairs */
};
/* Device job tables are CACHED by content: a caller that repeats its calls (a frame
   pipeline: the same jobs step after step) finds every table already resident and
   nothing is copied - a hipMemcpyAsync from pageable host memory stalls the host
   until the stream has drained, which is what kept round 1's driver from running
   ahead of the GPU. */
constexpr int kTableSlots = 8;
struct BandState {
  DJob *d_jobs = nullptr;          /* kTableSlots device job tables of kMaxJobs    */
  unsigned *d_pcount = nullptr;    /* priced choice: bands too close to call ...   */
  PUnc *d_plist = nullptr;         /* ... and their list                           */
  unsigned *pcount_host = nullptr; /* pinned mirror of the counter                 */
  hipEvent_t pcount_event = nullptr;
  DJob host_tab[kTableSlots][kMaxJobs];
  int tab_n[kTableSlots] = {};
  unsigned long tab_stamp[kTableSlots] = {};
  unsigned long tab_clock = 0;
  const DJob *cur = nullptr;       /* the table of the call in progress            */
  unsigned *d_sort = nullptr;      /* histogram / bin starts / cursors             */
  Scratch scr;                     /* x16, sort keys, sorted ids; grown on demand  */
  hipStream_t side[2] = {nullptr, nullptr};   /* kernels that may overlap          */
  hipEvent_t fork = nullptr;
  hipEvent_t join[2] = {nullptr, nullptr};
  bool serial = false;             /* the context's setting, refreshed per call    */
  bool sort_dirty = false;         /* the sort histograms may hold counts of a failed call */
  double tol_scale = 1.;           /* odhip_ctx_set_test_hooks */
  bool prof_on = false;            /* odhip_pvq_profile                            */
  bool prof_made = false;
  int prof_n = 0;
  hipEvent_t prof_ev[kProfSlots][2];
  ~BandState() {
    if (d_jobs) (void)hipFree(d_jobs);
    if (d_pcount) (void)hipFree(d_pcount);
    if (d_plist) (void)hipFree(d_plist);
    if (pcount_host) (void)hipHostFree(pcount_host);
    if (pcount_event) (void)hipEventDestroy(pcount_event);
    if (d_sort) (void)hipFree(d_sort);
    if (scr.x16) (void)hipFree(scr.x16);
    if (scr.keys) (void)hipFree(scr.keys);
    if (scr.ids) (void)hipFree(scr.ids);
    for (int i = 0; i < 2; i++) {
      if (side[i]) (void)hipStreamDestroy(side[i]);
      if (join[i]) (void)hipEventDestroy(join[i]);
    }
    if (fork) (void)hipEventDestroy(fork);
    if (prof_made) {
      for (int i = 0; i < kProfSlots; i++) {
        (void)hipEventDestroy(prof_ev[i][0]);
        (void)hipEventDestroy(prof_ev[i][1]);
      }
    }
  }
};

int band_state(BandState **out) {
  ODHIP_CTX_OR_RETURN(ctx);
  BandState *st = odhip_ctx_state<BandState>(ctx, ODHIP_SLOT_BANDS);
  if (!st->d_jobs) {
    ODHIP_TRY(hipMalloc((void **)&st->d_jobs, sizeof(DJob)*kMaxJobs*kTableSlots));
    ODHIP_TRY(hipMalloc((void **)&st->d_sort, sizeof(unsigned)*3*kMaxItems*kKeyBins));
    ODHIP_TRY(hipMemset(st->d_sort, 0, sizeof(unsigned)*3*kMaxItems*kKeyBins));
    ODHIP_TRY(hipMalloc((void **)&st->d_pcount, 2*sizeof(unsigned)));
    ODHIP_TRY(hipMalloc((void **)&st->d_plist, sizeof(PUnc)*kPUncCap));
    ODHIP_TRY(hipMemset(st->d_pcount, 0, 2*sizeof(unsigned)));
  }
  st->serial = ctx->serial != 0;
  st->tol_scale = ctx->price_tol_scale > 0 ? ctx->price_tol_scale : 1.;   /* test hook of the context */
  *out = st;
  return ODHIP_SUCCESS;
}

int fill_job(DJob &d, const odhip_pvq_job &j, int mode) {
  if (!j.d_coef || !j.q_band || !j.beta_band || j.bs < 0 || j.bs >= ODHIP_NBSIZES
   || j.nplanes <= 0) {
    return ODHIP_EINVAL;
  }
  const odhip_pvq_cands &c = j.cands;
  if (!c.band || !c.y || !c.choice) return ODHIP_EINVAL;
  if (((uintptr_t)c.band & 63) || ((uintptr_t)c.y & 15) || ((uintptr_t)c.choice & 15)) {
    return ODHIP_EINVAL;
  }
  /* mode 0: band stage (needs qm); 1: choice + synthesis (qm_inv, dq); 2: choice only */
  if (mode == 0 ? !j.d_qm : mode == 1 ? (!j.d_qm_inv || !j.d_dq) : false) return ODHIP_EINVAL;
  /* 16-byte loads of QM rows and of coefficient row segments */
  if (mode == 0 && (((uintptr_t)j.d_qm & 15) || ((uintptr_t)j.d_coef & 15))) return ODHIP_EINVAL;
  const int n = 4 << j.bs;
  if (j.w <= 0 || j.h <= 0 || j.w % n || j.h % n || (j.w & 3)) return ODHIP_EINVAL;
  memset(&d, 0, sizeof(d));
  d.coef = j.d_coef;
  d.qm = j.d_qm;
  d.qm_inv = j.d_qm_inv;
  d.rec = c.band;
  d.y = c.y;
  d.choice = c.choice;
  d.cos_dist = c.cos_dist;
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
    d.q2[i] = j.q_band2 ? j.q_band2[i] : j.q_band[i];
    if (d.q2[i] < 1) return ODHIP_EINVAL;
    d.beta[i] = j.beta_band[i];
  }
  d.split_blk = d.nblocks;
  if (j.q_band2) {
    if (j.plane_split <= 0 || j.plane_split >= j.nplanes) return ODHIP_EINVAL;
    d.split_blk = (long)j.plane_split*d.bw*d.bh;
  }
  return ODHIP_SUCCESS;
}

int fill_jobs(const odhip_pvq_job *jobs, int njobs, int mode, DJob *host) {
  if (!jobs || njobs <= 0 || njobs > kMaxJobs) return ODHIP_EINVAL;
  int rc = upload_tables();
  if (rc) return rc;
  for (int i = 0; i < njobs; i++) {
    rc = fill_job(host[i], jobs[i], mode);
    if (rc) return rc;
  }
  return ODHIP_SUCCESS;
}

int upload_jobs(BandState &st, DJob *host, int njobs, hipStream_t s) {
  for (int i = 0; i < njobs; i++) host[i].krange = st.d_pcount + 1;
  int lru = 0;
  for (int i = 0; i < kTableSlots; i++) {
    if (st.tab_n[i] == njobs && memcmp(st.host_tab[i], host, sizeof(DJob)*njobs) == 0) {
      st.tab_stamp[i] = ++st.tab_clock;
      st.cur = st.d_jobs + (size_t)i*kMaxJobs;
      return ODHIP_SUCCESS;
    }
    if (st.tab_stamp[i] < st.tab_stamp[lru]) lru = i;
  }
  /* miss: the least recently used slot is rewritten once nothing in flight on the
     caller's stream (side streams are joined into it at the end of every call) can
     still read it */
  if (st.tab_n[lru]) ODHIP_TRY(hipStreamSynchronize(s));
  memcpy(st.host_tab[lru], host, sizeof(DJob)*njobs);
  st.tab_n[lru] = njobs;
  st.tab_stamp[lru] = ++st.tab_clock;
  DJob *dst = st.d_jobs + (size_t)lru*kMaxJobs;
  ODHIP_TRY(hipMemcpy(dst, host, sizeof(DJob)*njobs, hipMemcpyHostToDevice));
  st.cur = dst;
  return ODHIP_SUCCESS;
}

int stage_jobs(BandState &st, const odhip_pvq_job *jobs, int njobs, int mode, DJob *host,
 hipStream_t s) {
  const int rc = fill_jobs(jobs, njobs, mode, host);
  if (rc) return rc;
  return upload_jobs(st, host, njobs, s);
}

int scratch_reserve(BandState &st, size_t x16_elems, size_t band_elems, hipStream_t s) {
  Scratch &g_scr = st.scr;
  if (x16_elems > g_scr.x16_cap) {
    ODHIP_TRY(hipStreamSynchronize(s));
    if (g_scr.x16) ODHIP_TRY(hipFree(g_scr.x16));
    g_scr.x16 = nullptr;
    g_scr.x16_cap = 0;
    ODHIP_TRY(hipMalloc((void **)&g_scr.x16, x16_elems*sizeof(int16_t)));
    g_scr.x16_cap = x16_elems;
  }
  if (band_elems > g_scr.band_cap) {
    ODHIP_TRY(hipStreamSynchronize(s));
    if (g_scr.keys) ODHIP_TRY(hipFree(g_scr.keys));
    if (g_scr.ids) ODHIP_TRY(hipFree(g_scr.ids));
    g_scr.keys = nullptr;
    g_scr.ids = nullptr;
    g_scr.band_cap = 0;
    ODHIP_TRY(hipMalloc((void **)&g_scr.keys, band_elems*sizeof(unsigned short)));
    ODHIP_TRY(hipMalloc((void **)&g_scr.ids, band_elems*sizeof(unsigned)));
    g_scr.band_cap = band_elems;
  }
  return ODHIP_SUCCESS;
}

/* Side streams for kernels that may overlap (created once per context). */
int fork_streams(BandState &st, hipStream_t s, hipStream_t side[2]) {
  if (st.serial || odhip_env_serial()) return ODHIP_SUCCESS;
  if (!st.fork) {
    ODHIP_TRY(hipEventCreateWithFlags(&st.fork, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
      ODHIP_TRY(hipStreamCreateWithFlags(&st.side[i], hipStreamNonBlocking));
      ODHIP_TRY(hipEventCreateWithFlags(&st.join[i], hipEventDisableTiming));
    }
  }
  ODHIP_TRY(hipEventRecord(st.fork, s));
  for (int i = 0; i < 2; i++) {
    ODHIP_TRY(hipStreamWaitEvent(st.side[i], st.fork, 0));
    side[i] = st.side[i];
  }
  return ODHIP_SUCCESS;
}

int join_streams(BandState &st, hipStream_t s, hipStream_t side[2]) {
  for (int i = 0; i < 2; i++) {
    if (side[i] == s) continue;
    ODHIP_TRY(hipEventRecord(st.join[i], side[i]));
    ODHIP_TRY(hipStreamWaitEvent(s, st.join[i], 0));
  }
  return ODHIP_SUCCESS;
}

void items_begin(Items &it, const BandState &st, double lambda) {
  memset(&it, 0, sizeof(it));
  it.lambda = lambda;
  it.jobs = st.cur;
  it.sort = st.d_sort;
  it.pcount = st.d_pcount;
  it.plist = st.d_plist;
  it.tol_scale = st.tol_scale;
  it.reserved = odhip_env_force_seq();   /* pair-mode search: always take the sequential combine */
}

/* Heaviest items first: the jobs arrive by ascending block size, and the bands of the largest
   blocks place the most pulses (K ~ 70 against 0-25 for the 128-coefficient luma bands) in the fewest
   wavefronts - launched last they were the tail of their kernel.  ODHIP_ITEMS_FWD=1 keeps the
   order of the jobs (experiments). */
void items_heavy_first(Items &it) {
  static const bool fwd = ODHIP_EXP_ENV("ODHIP_ITEMS_FWD") != nullptr;
  if (fwd) return;
  const int n = it.nitems;
  int size[kMaxItems];
  for (int i = 0; i < n; i++) size[i] = it.wg_start[i + 1] - it.wg_start[i];
  for (int i = 0; i < n/2; i++) {
    const unsigned char j = it.job[i];
    const unsigned char b = it.band[i];
    const int z = size[i];
    it.job[i] = it.job[n - 1 - i];
    it.band[i] = it.band[n - 1 - i];
    size[i] = size[n - 1 - i];
    it.job[n - 1 - i] = j;
    it.band[n - 1 - i] = b;
    size[n - 1 - i] = z;
  }
  for (int i = 0; i < n; i++) it.wg_start[i + 1] = it.wg_start[i] + size[i];
}

void items_add(Items &it, int job, int band, long wgs) {
  if (wgs <= 0) return;
  it.job[it.nitems] = (unsigned char)job;
  it.band[it.nitems] = (unsigned char)band;
  it.wg_start[it.nitems + 1] = it.wg_start[it.nitems] + (int)wgs;
  it.nitems++;
}

template <int N, int S, int NB>
void launch_search(BandState &st, const DJob *host, int njobs, double lambda, hipStream_t s, bool fuse) {
  constexpr int per_wave = kWave/S*NB;
  Items it;
  items_begin(it, st, lambda);
  it.fuse = fuse;
  for (int j = 0; j < njobs; j++) {
    for (int b = 0; b < host[j].nb_bands; b++) {
      if (host[j].off[b + 1] - host[j].off[b] == N) {
        items_add(it, j, b, (host[j].nblocks + per_wave - 1)/per_wave);
      }
    }
  }
  if (!it.nitems) return;
  constexpr size_t lds = kRsqN*sizeof(double) + (size_t)(N/S)*kPitch*4;
  /* odhip_pvq_profile: HIP events around the dominant kernel of the band stage, on
     the stream it is launched on */
  const bool prof = N == 128 && st.prof_on && st.prof_n < kProfSlots;
  if (prof) (void)hipEventRecord(st.prof_ev[st.prof_n][0], s);
  k_search<N, S, NB><<<it.wg_start[it.nitems], kWave, lds, s>>>(it);
  if (prof) (void)hipEventRecord(st.prof_ev[st.prof_n++][1], s);
}

}  // namespace

extern "C" int odhip_pvq_profile(int enable) {
  BandState *stp;
  int rc = band_state(&stp);
  if (rc) return rc;
  BandState &st = *stp;
  if (enable && !st.prof_made) {
    for (int i = 0; i < kProfSlots; i++) {
      ODHIP_TRY(hipEventCreate(&st.prof_ev[i][0]));
      ODHIP_TRY(hipEventCreate(&st.prof_ev[i][1]));
    }
    st.prof_made = true;
  }
  st.prof_on = enable != 0;
  st.prof_n = 0;
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pvq_profile_read(float *ms, int max_n) {
  BandState *stp;
  int rc = band_state(&stp);
  if (rc) return rc;
  BandState &st = *stp;
  int n = 0;
  for (; n < st.prof_n && n < max_n; n++) {
    ODHIP_TRY(hipEventSynchronize(st.prof_ev[n][1]));
    ODHIP_TRY(hipEventElapsedTime(&ms[n], st.prof_ev[n][0], st.prof_ev[n][1]));
  }
  st.prof_n = 0;
  return n;
}

extern "C" int odhip_pvq_band_layout(int bs, int *nb_bands, int *offsets, int *len) {
  if (bs < 0 || bs >= ODHIP_NBSIZES) return ODHIP_EINVAL;
  const int n = 4 << bs;
  if (nb_bands) *nb_bands = OD_NBANDS[bs];
  if (offsets) for (int i = 0; i <= OD_NBANDS[bs]; i++) offsets[i] = OD_BAND_OFFS[bs][i];
  if (len) *len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  return ODHIP_SUCCESS;
}

namespace {

/* The count of bands the priced choice listed travels to pinned host memory behind it. */
int price_count_begin(BandState &st, hipStream_t s) {
  if (!st.pcount_host) {
    ODHIP_TRY(hipHostMalloc((void **)&st.pcount_host, sizeof(unsigned), hipHostMallocDefault));
    ODHIP_TRY(hipEventCreateWithFlags(&st.pcount_event, hipEventDisableTiming));
  }
  *st.pcount_host = 0xffffffffu;
  ODHIP_TRY(hipMemcpyAsync(st.pcount_host, st.d_pcount, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  ODHIP_TRY(hipEventRecord(st.pcount_event, s));
  return ODHIP_SUCCESS;
}

int noref_bands(const odhip_pvq_job *jobs, int njobs, double pvq_norm_lambda, odhip_stream stream,
 bool fuse) {
  BandState *stp;
  {
    const int rc0 = band_state(&stp);
    if (rc0) return rc0;
  }
  BandState &st = *stp;
  hipStream_t s = (hipStream_t)stream;
  const double lambda = pvq_norm_lambda;
  DJob host[kMaxJobs];
  int rc = fill_jobs(jobs, njobs, 0, host);
  if (rc) return rc;
  if (fuse) ODHIP_TRY(hipMemsetAsync(st.d_pcount, 0, sizeof(unsigned), s));
  size_t x16_elems = 0;
  size_t band_elems = 0;
  for (int j = 0; j < njobs; j++) {
    x16_elems += (size_t)host[j].nblocks*host[j].len;
    band_elems += (size_t)host[j].nblocks*host[j].nb_bands;
    for (int b = 0; b < host[j].nb_bands; b++) {
      const int n = host[j].off[b + 1] - host[j].off[b];
      if (n != 8 && n != 15 && n != 32 && n != 128) return ODHIP_EINVAL;
    }
  }
  if (!fuse) {
    /* scaled vectors, sort keys and sorted indices of the two-pass stage (the stage that decides in place
       keeps none of them) */
    rc = scratch_reserve(st, x16_elems, band_elems, s);
    if (rc) return rc;
    x16_elems = 0;
    band_elems = 0;
    for (int j = 0; j < njobs; j++) {
      host[j].x16 = st.scr.x16 + x16_elems;
      host[j].keys = st.scr.keys + band_elems;
      host[j].ids = st.scr.ids + band_elems;
      x16_elems += (size_t)host[j].nblocks*host[j].len;
      band_elems += (size_t)host[j].nblocks*host[j].nb_bands;
    }
  }
  rc = upload_jobs(st, host, njobs, s);
  if (rc) return rc;
  hipStream_t side[2] = {s, s};
  if (fork_streams(st, s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
  Items it;
  if (fuse) {
    /* With the priced choice every band is prepared, searched and decided by the lanes that load it
       (k_decide_corner, k_decide_lane32, k_decide_pair128): no preparation pass, no sort, no scaled
       vectors or records in memory.  Four independent launches. */
    constexpr size_t lane_lds = kRsqN*sizeof(double) + (size_t)16*kPitch*4;
    constexpr size_t pair_lds = kRsqN*sizeof(double) + (size_t)64*kPitch*4;
    /* the 128-coefficient bands: a lane pair per band.  Round 5 measured a QUAD per band
       (k_decide_quad128: half the LDS per wavefront, three wavefronts per SIMD instead of two, every scan
       half as long) at 586 us against 550-568 for the pair on the same content: what the pair form lacks
       is not occupancy, and the quad pays a two-level combine per pulse.  The quad stays in the experiments
       build (ODHIP_PVQ_QUAD128=1), bit-identical. */
    const bool pair128 = ODHIP_EXP_ENV("ODHIP_PVQ_QUAD128") == nullptr;
    const int per_wg = pair128 ? kWave/2 : kWave/4;
    items_begin(it, st, lambda);
    it.fuse = 1;
    for (int j = 0; j < njobs; j++) {
      for (int b = 0; b < host[j].nb_bands; b++) {
        if (host[j].off[b + 1] - host[j].off[b] == 128) items_add(it, j, b, (host[j].nblocks + per_wg - 1)/per_wg);
      }
    }
    const bool prof = st.prof_on && st.prof_n < kProfSlots;
    if (prof) (void)hipEventRecord(st.prof_ev[st.prof_n][0], s);
    items_heavy_first(it);
    if (it.nitems) {
      if (pair128) k_decide_pair128<<<it.wg_start[it.nitems], kWave, pair_lds, s>>>(it);
#ifdef ODHIP_EXPERIMENTS
      else {
        constexpr size_t quad_lds = kRsqN*sizeof(double) + (size_t)32*kPitch*4;
        k_decide_quad128<<<it.wg_start[it.nitems], kWave, quad_lds, s>>>(it);
      }
#endif
    }
    if (prof) (void)hipEventRecord(st.prof_ev[st.prof_n++][1], s);
    items_begin(it, st, lambda);
    it.fuse = 1;
    for (int j = 0; j < njobs; j++) {
      for (int b = 0; b < host[j].nb_bands; b++) {
        if (host[j].off[b + 1] - host[j].off[b] == 32) items_add(it, j, b, (host[j].nblocks + kWave/2 - 1)/(kWave/2));
      }
    }
    items_heavy_first(it);
    if (it.nitems) k_decide_lane32<<<it.wg_start[it.nitems], kWave, lane_lds, side[0]>>>(it);
    items_begin(it, st, lambda);
    it.fuse = 1;
    for (int j = 0; j < njobs; j++) items_add(it, j, 0, (host[j].nblocks + kWave - 1)/kWave);
    items_heavy_first(it);
    if (it.nitems) k_decide_corner<0><<<it.wg_start[it.nitems], kWave, lane_lds, side[1]>>>(it);
    items_begin(it, st, lambda);
    it.fuse = 1;
    for (int j = 0; j < njobs; j++) {
      if (host[j].bs > 0) items_add(it, j, 1, (host[j].nblocks + kWave - 1)/kWave);
    }
    items_heavy_first(it);
    if (it.nitems) k_decide_corner<1><<<it.wg_start[it.nitems], kWave, lane_lds, side[1]>>>(it);
    if (join_streams(st, s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
    const int rc2 = price_count_begin(st, s);
    if (rc2) return rc2;
    return odhip_check_launch();
  }
  /* prep: the low-frequency corner of every block one block per lane (bands
     0..3), the remaining 32-coefficient bands one band per lane, the
     128-coefficient bands one per 16-lane row */
  items_begin(it, st, lambda);
  for (int j = 0; j < njobs; j++) {
    if (host[j].bs == 0) items_add(it, j, 0, (host[j].nblocks + kWave - 1)/kWave);
  }
  if (it.nitems) k_prep_corner<4><<<it.wg_start[it.nitems], kWave, 0, s>>>(it);
  items_begin(it, st, lambda);
  for (int j = 0; j < njobs; j++) {
    if (host[j].bs > 0) items_add(it, j, 0, (host[j].nblocks + kWave - 1)/kWave);
  }
  if (it.nitems) k_prep_corner<8><<<it.wg_start[it.nitems], kWave, 0, s>>>(it);
  items_begin(it, st, lambda);
  for (int j = 0; j < njobs; j++) {
    for (int b = 4; b < host[j].nb_bands; b++) {
      const int n = host[j].off[b + 1] - host[j].off[b];
      if (n <= 32) items_add(it, j, b, (host[j].nblocks + kWave - 1)/kWave);
    }
  }
  if (it.nitems) k_prep_lane<<<it.wg_start[it.nitems], kWave, 0, side[1]>>>(it);
  items_begin(it, st, lambda);
  for (int j = 0; j < njobs; j++) {
    for (int b = 0; b < host[j].nb_bands; b++) {
      if (host[j].off[b + 1] - host[j].off[b] == 128) items_add(it, j, b, (host[j].nblocks + 3)/4);
    }
  }
  if (it.nitems) k_prep_wide<<<it.wg_start[it.nitems], kWave, 0, side[0]>>>(it);
  if (join_streams(st, s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
  /* counting sort of every item's blocks by pulse class */
  Items all;
  items_begin(all, st, lambda);
  items_begin(it, st, lambda);
  for (int j = 0; j < njobs; j++) {
    for (int b = 0; b < host[j].nb_bands; b++) {
      items_add(it, j, b, (host[j].nblocks + kSortChunk - 1)/kSortChunk);
      items_add(all, j, b, 1);
    }
  }
  /* the histograms are consumed and cleared by k_prefix; only a call that failed between
     the two leaves them dirty */
  if (st.sort_dirty) {
    ODHIP_TRY(hipMemsetAsync(st.d_sort, 0, sizeof(unsigned)*kMaxItems*kKeyBins, s));
    st.sort_dirty = false;
  }
  st.sort_dirty = true;
  k_hist<<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  k_prefix<<<all.nitems, 256, 0, s>>>(all);
  st.sort_dirty = odhip_check_launch() != ODHIP_SUCCESS;
  k_scatter<<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  /* search: the band sizes are independent launches on forked streams */
  if (fork_streams(st, s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
  launch_search<128, 2, 1>(st, host, njobs, lambda, s, false);
  launch_search<32, 2, 1>(st, host, njobs, lambda, side[0], false);
  launch_search<15, 1, 1>(st, host, njobs, lambda, side[1], false);
  launch_search<8, 1, 1>(st, host, njobs, lambda, side[1], false);
  if (join_streams(st, s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
  return odhip_check_launch();
}

}  // namespace

extern "C" int odhip_pvq_noref_bands_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  return noref_bands(jobs, njobs, pvq_norm_lambda, stream, false);
}

/* The band stage AND the priced choice of odhip_pvq_choose_priced_multi in one pass: the
   search kernels decide each band from the values they hold in registers (no separate
   choice kernel reading the records back).  Follow with odhip_pvq_choose_priced_resolve. */
extern "C" int odhip_pvq_noref_bands_priced_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  return noref_bands(jobs, njobs, pvq_norm_lambda, stream, true);
}

extern "C" int odhip_pvq_select_synth_noref_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  BandState *stp;
  {
    const int rc0 = band_state(&stp);
    if (rc0) return rc0;
  }
  BandState &st = *stp;
  hipStream_t s = (hipStream_t)stream;
  DJob host[kMaxJobs];
  int rc = stage_jobs(st, jobs, njobs, 1, host, s);
  if (rc) return rc;
  Items it;
  items_begin(it, st, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) {
    items_add(it, j, 0, (host[j].nblocks*host[j].nb_bands + 255)/256);
  }
  k_choose<0><<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  items_begin(it, st, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) {
    items_add(it, j, 0, (long)host[j].nplanes*host[j].h*((host[j].w + 1023) >> 10));
  }
  k_synth<<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  return odhip_check_launch();
}

extern "C" int odhip_pvq_choose_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  BandState *stp;
  {
    const int rc0 = band_state(&stp);
    if (rc0) return rc0;
  }
  BandState &st = *stp;
  hipStream_t s = (hipStream_t)stream;
  DJob host[kMaxJobs];
  int rc = stage_jobs(st, jobs, njobs, 2, host, s);
  if (rc) return rc;
  Items it;
  items_begin(it, st, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) {
    items_add(it, j, 0, (host[j].nblocks*host[j].nb_bands + 255)/256);
  }
  k_choose<0><<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  return odhip_check_launch();
}

/* The choice with od_pvq_rate's closed form evaluated on the device (see choose_band). */
extern "C" int odhip_pvq_choose_priced_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  BandState *stp;
  {
    const int rc0 = band_state(&stp);
    if (rc0) return rc0;
  }
  BandState &st = *stp;
  hipStream_t s = (hipStream_t)stream;
  DJob host[kMaxJobs];
  int rc = stage_jobs(st, jobs, njobs, 2, host, s);
  if (rc) return rc;
  ODHIP_TRY(hipMemsetAsync(st.d_pcount, 0, sizeof(unsigned), s));
  Items it;
  items_begin(it, st, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) {
    items_add(it, j, 0, (host[j].nblocks*host[j].nb_bands + 255)/256);
  }
  k_choose<1><<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  rc = price_count_begin(st, s);
  if (rc) return rc;
  return odhip_check_launch();
}

/* Waits for the stream, then settles the bands odhip_pvq_choose_priced_multi listed:
   their candidates' rates are recomputed with the HOST libm's log (src/pvq_encoder.c:263,
   the call the reference makes) and the bands decided again.  Returns how many (normally
   0), or a negative code.  Pass the same jobs. */
extern "C" int odhip_pvq_choose_priced_resolve(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  BandState *stp;
  {
    const int rc0 = band_state(&stp);
    if (rc0) return rc0;
  }
  BandState &st = *stp;
  hipStream_t s = (hipStream_t)stream;
  /* normal case: only the count is waited for (it was sent right behind the choice) */
  if (!st.pcount_event) return ODHIP_EINVAL;
  ODHIP_TRY(hipEventSynchronize(st.pcount_event));
  if (*st.pcount_host == 0) return 0;
  ODHIP_TRY(hipStreamSynchronize(s));
  unsigned count = 0;
  ODHIP_TRY(hipMemcpy(&count, st.d_pcount, sizeof(count), hipMemcpyDeviceToHost));
  if (count == 0) return 0;
  if (count > (unsigned)kPUncCap) {
    fprintf(stderr, "libdaalahip: %u priced bands inside the decision margin exceed the list (%d)\n", count,
     kPUncCap);
    return ODHIP_EFAULT;
  }
  DJob host[kMaxJobs];
  int rc = stage_jobs(st, jobs, njobs, 2, host, s);
  if (rc) return rc;
  PUnc *list = (PUnc *)malloc(sizeof(PUnc)*count);
  if (!list) return ODHIP_EFAULT;
  rc = ODHIP_SUCCESS;
  if (hipMemcpy(list, st.d_plist, sizeof(PUnc)*count, hipMemcpyDeviceToHost) != hipSuccess) rc = ODHIP_EFAULT;
  for (unsigned i = 0; i < count && !rc; i++) {
    if (list[i].job < 0 || list[i].job >= njobs) {
      rc = ODHIP_EINVAL;
      break;
    }
    const DJob &jb = host[list[i].job];
    const long sb = list[i].sb;
    const int band = (int)(sb % jb.nb_bands);
    const int n = jb.off[band + 1] - jb.off[band];
    odhip_pvq_band rec;
    if (hipMemcpy(&rec, jb.rec + sb, sizeof(rec), hipMemcpyDeviceToHost) != hipSuccess) {
      rc = ODHIP_EFAULT;
      break;
    }
    for (int c = 0; c < 2; c++) {
      list[i].rate[c] = 0;
      if (rec.flags[c] != 1) continue;
      list[i].rate[c] = odq_pvq_rate_fast_host(rec.moment[c], rec.k[c], n, rec.gain[c], 0, -1, 0, 1, 0);
    }
  }
  PUnc *d_list = nullptr;
  if (!rc && (hipMalloc((void **)&d_list, sizeof(PUnc)*count) != hipSuccess
   || hipMemcpy(d_list, list, sizeof(PUnc)*count, hipMemcpyHostToDevice) != hipSuccess)) {
    rc = ODHIP_EFAULT;
  }
  free(list);
  if (!rc) {
    Items it;
    items_begin(it, st, pvq_norm_lambda);
    k_choose_list<<<(count + kWave - 1)/kWave, kWave, 0, s>>>(it, d_list, (int)count);
    rc = odhip_check_launch();
    if (hipStreamSynchronize(s) != hipSuccess) rc = ODHIP_EFAULT;
  }
  if (d_list) (void)hipFree(d_list);
  return rc ? rc : (int)count;
}

/* Test hook: scales the decision margin of the priced choices (1 restores it). */

extern "C" int odhip_cfl_refs_from_luma(const odhip_pvq_job *luma_jobs, int njobs,
 od_coeff *const *d_ref, int copies, odhip_stream stream) {
  return odhip_cfl_refs_from_luma_ex(luma_jobs, njobs, d_ref, copies, 0, stream);
}

extern "C" int odhip_cfl_refs_from_luma_ex(const odhip_pvq_job *luma_jobs, int njobs,
 od_coeff *const *d_ref, int copies, int prezeroed, odhip_stream stream) {
  BandState *stp;
  {
    const int rc0 = band_state(&stp);
    if (rc0) return rc0;
  }
  BandState &st = *stp;
  hipStream_t s = (hipStream_t)stream;
  if (!d_ref || copies < 1 || copies > 4) return ODHIP_EINVAL;
  DJob host[kMaxJobs];
  int rc = fill_jobs(luma_jobs, njobs, 2, host);
  if (rc) return rc;
  rc = upload_jobs(st, host, njobs, s);     /* before items_begin: it selects the table */
  if (rc) return rc;
  CflOut out;
  memset(&out, 0, sizeof(out));
  out.copies = copies;
  Items it;
  items_begin(it, st, 0.);
  for (int j = 0; j < njobs; j++) {
    if (!d_ref[j] || !luma_jobs[j].d_qm_inv || (host[j].w & 7) || (host[j].h & 7)) return ODHIP_EINVAL;
    out.ref[j] = d_ref[j];
    if (host[j].bs >= 1) {
      const int n = 2 << host[j].bs;
      const int ncode = n*n < host[j].len ? n*n : host[j].len;
      items_add(it, j, 0, (host[j].nblocks*(ncode >> 3) + 255)/256);
      if (n*n > ncode && !prezeroed) {
        /* 64x64 luma blocks: half of the 32x32 corner is not coded */
        ODHIP_TRY(hipMemsetAsync(d_ref[j], 0, sizeof(od_coeff)*(size_t)copies*host[j].nplanes
         *(host[j].w >> 1)*(host[j].h >> 1), s));
      }
    }
  }
  if (it.nitems) k_cfl_ref<<<it.wg_start[it.nitems], 256, 0, s>>>(it, out);
  /* level-0 jobs: the TF branch */
  items_begin(it, st, 0.);
  for (int j = 0; j < njobs; j++) {
    if (host[j].bs == 0) {
      items_add(it, j, 0, ((long)host[j].nplanes*(host[j].w >> 1)*(host[j].h >> 1) + 255)/256);
    }
  }
  if (it.nitems) k_cfl_ref_tf<<<it.wg_start[it.nitems], 256, 0, s>>>(it, out);
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

/* od_krange.cuh: the current context's count since the last call, then cleared (blocking copies: the stream of the band
   stage must have been synchronised) */
int od_k_range_take_noref(unsigned *count) {
  BandState *st = nullptr;
  const int rc = band_state(&st);
  if (rc) return rc;
  unsigned v = 0;
  ODHIP_TRY(hipMemcpy(&v, st->d_pcount + 1, sizeof(v), hipMemcpyDeviceToHost));
  if (v) ODHIP_TRY(hipMemset(st->d_pcount + 1, 0, sizeof(v)));
  *count = v;
  return ODHIP_SUCCESS;
}
