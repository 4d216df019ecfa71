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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "od_ctx.cuh"
#include "od_pvq_math.cuh"
#include "od_occupancy.cuh"
#include "od_krange.cuh"
#include "gen/od_scan_tables.h"
#define OD_RSQ_TABLE_N 512
#define OD_RSQ_HUGE
#include "pvq_search.cuh"
#include "pvq_row.cuh"
#include "pvq_regs.cuh"

namespace {

/* All state between calls (device job table, uncertainty list, sort arrays, scratch,
   side streams, pinned counter) belongs to the calling thread's current context
   (od_ctx.cuh): independent call sequences (e.g. the Cb and the Cr planes of a
   batch on two streams) run in two contexts. */
constexpr int kMaxJobs = 8;
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
  void *items;
  int16_t *y;
  int16_t *r16;
  int16_t *x16;
  int16_t *xr;
  const double *rate;
  int32_t *choice;
  od_coeff *dq;
  /* library scratch of the sorted searches */
  unsigned short *keys;   /* [nb][B] work class of the band (heavy = small)       */
  unsigned *ids;          /* [nb][B] block indices sorted by key                  */
  unsigned *krange;       /* od_krange.cuh: the calling context's counter          */
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
  /* planes from plane_split on (blocks from split_blk on) take q2: the Cr half of a
     chroma plane set - pvq_qm_q4[pli] is per plane, src/encode.c:3052-3072 */
  long split_blk;
  int q2[ODHIP_MAX_BANDS];
  /* chroma-from-luma reference taken straight from the luma band stage (odhip_pvq_refjob.luma):
     the chosen pulses / choice records / inverse QM of the luma level one size up; the
     reference plane itself never exists */
  const int16_t *ly;
  const int32_t *lchoice;
  const int16_t *lqmi;
  long lnblocks;
  long lsplit;      /* blocks of one luma plane set: chroma plane p uses luma plane p mod planes */
  int llen;
  int lnb;
};

/* The band's quantiser step for block blk. */
__device__ __forceinline__ int job_q(const RJob &jb, int band, long blk) {
  return blk >= jb.split_blk ? jb.q2[band] : jb.q[band];
}

/* od_resample_luma_coeffs for luma blocks of 8x8 and larger (src/intra.c:97-108: the
   upper-left quarter of the decoded luma block) for the eight coding positions c0 .. c0 + 7
   of chroma block blk, band `band`: the scan is nested (the first n*n coding positions of a
   2n x 2n block are its n x n corner, and the band offsets coincide), so they are coding
   positions c0 .. c0 + 7 of the same band of the co-located luma block, dequantised from its
   chosen pulses exactly as od_pvq_synthesis_partial without reference writes them
   (src/pvq.c:1081-1092).  Position 0 (the DC slot, never used by PVQ) comes out as 0. */
__device__ __forceinline__ void lref_piece(const RJob &jb, int band, long blk, int c0, int (&rv)[8]) {
  const long lblk = blk >= jb.lsplit ? blk - jb.lsplit : blk;
  const int4 ch = reinterpret_cast<const int4 *>(jb.lchoice)[lblk*jb.lnb + band];   /* sel, qg, scale, qshift */
  uint4 y4 = make_uint4(0, 0, 0, 0);
  if (ch.y != 0) y4 = *reinterpret_cast<const uint4 *>(jb.ly + ((long)ch.x*jb.lnblocks + lblk)*jb.llen + c0);
  const uint4 q4 = *reinterpret_cast<const uint4 *>(jb.lqmi + c0);
  const unsigned yw[4] = {y4.x, y4.y, y4.z, y4.w};
  const unsigned qw[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int yv = (int16_t)(yw[e >> 1] >> (16*(e & 1)));
    const int qmi = (int16_t)(qw[e >> 1] >> (16*(e & 1)));
    rv[e] = odq_shr_round(odq_mult16_32_q16(yv, ch.z)*qmi, ch.w);
  }
}

struct Unc;
/* A band whose priced choice the host (re)decides: rate[0] = the initial candidate,
   rate[1 + i] = candidate i, from the host libm. */
struct PUncR {
  int job;
  int band;
  unsigned blk;
  int reserved;
  double rate[ODHIP_PVQ_REF_SLOTS + 1];
};
constexpr int kPUncCap = 1 << 14;

struct RItems {
  int nitems;
  int perturb;
  double lambda;
  double margin;
  const RJob *jobs;        /* the context's device job table [kMaxJobs]            */
  unsigned *unc_count;     /* its uncertainty list: counter ...                    */
  Unc *unc;                /* ... and entries [kUncCap]                            */
  unsigned *rhist;         /* counting sort: histogram (zero between calls) ...    */
  unsigned *rcursor;       /* ... and cursors, kMaxItems*kSortBins words each      */
  unsigned *pcount;        /* priced choice: bands too close to call on the device; pcount[1]: bands with a
                              candidate above ODHIP_PVQ_MAX_K (od_krange.cuh), cleared only when taken */
  struct PUncR *plist;     /* ... and their list [kPUncCap]                        */
  double tol_scale;        /* test hook: multiplies the decision margin            */
  int fuse;                /* the per-lane searches also make the priced choice    */
  int reserved1;
#ifdef ODHIP_EXPERIMENTS
  int abl;                 /* ODHIP_REFB_ABL: ablation bits (tools/gpu_r6_ablate.sh)  */
#endif
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

__constant__ unsigned char kRScanXY[OD_SCAN_LEN][2];

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
__device__ __attribute__((aligned(16))) unsigned short gRScanPk[OD_SCAN_LEN];   /* y << 8 | x */

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


/* ---- sorting the bands of an item by work -----------------------------------------
   The cost of a band is the number of pulses its chains place, which spans
   0..hundreds within one level, and a wavefront (64 bands, or 4 rows) runs as
   long as its slowest member.  Every band is classified (kSortBins classes, heavy
   first); a counting sort per (job, band) item - LDS histogram per chunk of blocks,
   one global atomic per non-empty class per chunk - yields the block order the
   searches walk. */
constexpr int kSortBins = 256;
constexpr int kSortChunk = 2048;

__device__ __forceinline__ int od_work_bin(int pulses) {   /* 0..kSortBins-1, monotone */
  if (pulses < 96) return pulses;
  const int b = 96 + ((pulses - 96) >> 3);
  return b < kSortBins ? b : kSortBins - 1;
}

/* The candidates of a band in the reference's ENUMERATION order (src/pvq_encoder.c:466-504,
   then :571-581), handed to a sink:
     sink.gain(gi, i, ts, lower)   gain index i = gain_bound - 1 + gi (gi = 0..2) is about to
                                   be enumerated: its angular resolution and first angle
     sink.theta(gi, i, j, ts, k)   one (gain, theta) candidate with its pulse count
     sink.noref(c, i, k)           no-reference candidate c = 0 / 1
   At most kSlots - 2 theta candidates, as refb_candidates keeps. */
template <class Sink>
__device__ __forceinline__ void refb_enumerate(const RJob &jb, int band, int32_t cg, int32_t gain_offset,
 int32_t theta, int flags, double corr, Sink &sink) {
  const int n = jb.off[band + 1] - jb.off[band];
  const int beta = jb.beta[band];
  if (flags & ODHIP_REFBAND_THETA) {
    int nth = 0;
    const int gain_bound = (cg - gain_offset) >> ODQ_CGAIN_SHIFT;
    const double scale_1 = __ddiv_rn(1., kThetaScale);   /* OD_THETA_SCALE_1 */
    for (int i = gain_bound - 1 > 1 ? gain_bound - 1 : 1; i <= gain_bound + 1; i++) {
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT) + gain_offset;
      const int ts = odq_pvq_compute_max_theta(qcg, beta);
      /* same left-to-right products as src/pvq_encoder.c:482-484 */
      const double t = __ddiv_rn(((theta*scale_1)*2), kPi)*ts;
      int lower = (int)floor(.5 + t) - 2;
      if (lower < 0) lower = 0;
      int upper = (int)ceil(t);
      if (upper > ts - 1) upper = ts - 1;
      const int gi = i - (gain_bound - 1);
      sink.gain(gi, i, ts, lower);
      for (int j = lower; j <= upper && nth < ODHIP_PVQ_REF_SLOTS - 2; j++) {
        sink.theta(gi, i, j, ts, odq_compute_k_ref(j, n));
        nth++;
      }
    }
  }
  /* src/pvq_encoder.c:571-581 */
  if ((jb.is_keyframe && jb.pli == 0) || corr < .5 || cg < odq_shl32(2, ODQ_CGAIN_SHIFT)) {
    const int gain_bound = cg >> ODQ_CGAIN_SHIFT;
    int c = 0;
    for (int i = gain_bound > 1 ? gain_bound : 1; i <= gain_bound + 1; i++) {
      sink.noref(c++, i, odq_compute_k_noref(odq_shl32(i, ODQ_CGAIN_SHIFT), n, beta));
    }
  }
}

/* The work class of a band.  Its chains place about kmax pulses (~240 instructions each for a
   15-coefficient band), but most bands place 0-3 pulses and their cost is the candidates themselves
   (~300 instructions each) and the searches (~400 each beside their pulses): round 4 measured 0.56 of
   the lanes active in k_refb_lean_lane<15> with the pulses alone as the key
   (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, tools/gpu_r4_lanes.sh), so the class is
   pulses + candidates + 2 * distinct pulse counts (one search each). */
struct KeySink {
  int kt = 0;
  int kn = 0;
  int c = 0;
  unsigned long long mt = 0;
  unsigned long long mn = 0;
  __device__ __forceinline__ void gain(int, int, int, int) {}
  __device__ __forceinline__ void theta(int, int, int, int, int k) {
    kt = k > kt ? k : kt;
    c++;
    if (k) mt |= 1ull << (k & 63);
  }
  __device__ __forceinline__ void noref(int, int, int k) {
    kn = k > kn ? k : kn;
    c++;
    if (k) mn |= 1ull << (k & 63);
  }
  /* w: weights in quarters, pulses | candidates << 8 | searches << 16 */
  __device__ __forceinline__ int work(int w) const {
    return ((w & 255)*(kt + kn) + (w >> 8 & 255)*c + (w >> 16 & 255)*(__popcll(mt) + __popcll(mn))) >> 2;
  }
};

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
      const unsigned slot = atomicAdd(it.unc_count, 1u);
      if (slot < (unsigned)kUncCap) {
        Unc e;
        e.job = job;
        e.band = band;
        e.blk = (unsigned)blk;
        e.theta = theta;
        e.corr = p.corr;
        it.unc[slot] = e;
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
  if (it.fuse == 2) {
    /* the decided stage has no candidate kernel: the work class comes from here */
    const RJob &jb = it.jobs[job];
    KeySink ks;
    refb_enumerate(jb, band, o.cg, o.gain_offset, o.theta, flags, o.corr, ks);
    jb.keys[(long)band*jb.nblocks + blk] = (unsigned short)(kSortBins - 1 - od_work_bin(ks.work(it.reserved1)));
    /* a candidate of this band's list has more pulses than the pulse vectors hold: the searches skip it
       (conservative: counted even when the reference's own pruning test would have dropped it) */
    if (ks.kt > ODHIP_PVQ_MAX_K || ks.kn > ODHIP_PVQ_MAX_K) atomicAdd(jb.krange, 1u);
  }
}

__device__ __forceinline__ uint32_t pack16(int lo, int hi) {
  return (uint32_t)(lo & 0xffff) | (uint32_t)hi << 16;
}

/* N = 15 (band 0: the flip is decided here) or N = 8. */
template <int N>
__global__ __launch_bounds__(kWave) void k_refb_prep_lane(RItems it) {
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const RJob &jb = it.jobs[job];
  const int band = it.band[item];
  const long blk = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  if (blk >= jb.nblocks) return;
  const int off = jb.off[band];
  const int w = jb.w;
  const int len = jb.len;
  const int nb_bands = jb.nb_bands;
  const int q0 = job_q(jb, band, blk);
  const int beta = jb.beta[band];
  const int is_keyframe = jb.is_keyframe;
  const int cfl_enabled = is_keyframe && jb.pli != 0;
  const long base = block_base(jb, blk);
  const od_coeff *x0 = jb.coef + base;
  /* the reference: a plane of the job's layout, or (keyframe chroma) taken straight from
     the luma band stage - lref_piece */
  const bool lref = jb.ly != nullptr;
  const od_coeff *r0 = lref ? jb.coef + base : jb.ref + base;
  const int16_t *qmp = jb.qm + off;
  odhip_pvq_refband *rec = jb.rec + blk*nb_bands;
  int16_t *x16o = jb.x16 + blk*len;
  int16_t *r16o = jb.r16 + blk*len;
  int16_t *xro = jb.xr + blk*len;
  int xv[N];
  int rv[N];
  int qm[N];
  if constexpr (N == 15) {
    /* band 0 is the 4x4 low-frequency corner without the DC: four 16-byte row segments
       of x and of r (adjacent lanes = adjacent blocks: contiguous for 4x4 blocks),
       permuted to coding order through the compile-time scan */
    int4 xr4[4];
    int4 rr4[4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
      xr4[y] = *reinterpret_cast<const int4 *>(x0 + (long)y*w);
      rr4[y] = lref ? make_int4(0, 0, 0, 0) : *reinterpret_cast<const int4 *>(r0 + (long)y*w);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int sx = OD_SCAN_XY[1 + i][0];
      const int sy = OD_SCAN_XY[1 + i][1];
      const int4 a = xr4[sy];
      const int4 b = rr4[sy];
      xv[i] = sx == 0 ? a.x : sx == 1 ? a.y : sx == 2 ? a.z : a.w;
      rv[i] = sx == 0 ? b.x : sx == 1 ? b.y : sx == 2 ? b.z : b.w;
      qm[i] = qmp[i];
    }
  }
  else if (off == 16) {
    /* band 1: columns 4..7 of rows 0 and 1 - two 16-byte row segments */
    int4 xr4[2];
    int4 rr4[2];
#pragma unroll
    for (int y = 0; y < 2; y++) {
      xr4[y] = *reinterpret_cast<const int4 *>(x0 + (long)y*w + 4);
      rr4[y] = lref ? make_int4(0, 0, 0, 0) : *reinterpret_cast<const int4 *>(r0 + (long)y*w + 4);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int sx = OD_SCAN_XY[16 + (i < 8 ? i : 0)][0] - 4;
      const int sy = OD_SCAN_XY[16 + (i < 8 ? i : 0)][1];
      const int4 a = xr4[sy & 1];
      const int4 b = rr4[sy & 1];
      xv[i] = sx == 0 ? a.x : sx == 1 ? a.y : sx == 2 ? a.z : a.w;
      rv[i] = sx == 0 ? b.x : sx == 1 ? b.y : sx == 2 ? b.z : b.w;
      qm[i] = qmp[i];
    }
  }
  else if (off == 24) {
    /* band 2: columns 0..1 of rows 4..7 - four 8-byte row segments */
    int2 xr2[4];
    int2 rr2[4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
      xr2[y] = *reinterpret_cast<const int2 *>(x0 + (long)(4 + y)*w);
      rr2[y] = lref ? make_int2(0, 0) : *reinterpret_cast<const int2 *>(r0 + (long)(4 + y)*w);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int sx = OD_SCAN_XY[24 + (i < 8 ? i : 0)][0];
      const int sy = OD_SCAN_XY[24 + (i < 8 ? i : 0)][1] - 4;
      const int2 a = xr2[sy & 3];
      const int2 b = rr2[sy & 3];
      xv[i] = sx == 0 ? a.x : a.y;
      rv[i] = sx == 0 ? b.x : b.y;
      qm[i] = qmp[i];
    }
  }
  else {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const long p = (long)kRScanXY[off + i][1]*w + kRScanXY[off + i][0];
      xv[i] = x0[p];
      rv[i] = lref ? 0 : r0[p];
      qm[i] = qmp[i];
    }
  }
#ifdef ODHIP_EXPERIMENTS
  const bool lref_read = lref && !(it.abl & 2);
#else
  const bool lref_read = lref;
#endif
  if (lref_read) {
    /* coding order is what the luma stage keeps: whole 16-byte pieces */
    if constexpr (N == 15) {
      int lo[8];
      int hi[8];
      lref_piece(jb, band, blk, 0, lo);
      lref_piece(jb, band, blk, 8, hi);
#pragma unroll
      for (int i = 0; i < N; i++) rv[i] = i + 1 < 8 ? lo[(i + 1) & 7] : hi[(i + 1) & 7];
    }
    else {
      int pc[8];
      lref_piece(jb, band, blk, off, pc);
#pragma unroll
      for (int i = 0; i < N; i++) rv[i] = pc[i & 7];
    }
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
#ifdef ODHIP_EXPERIMENTS
  const bool abl_nostore = (it.abl & 1) != 0;
  const bool abl_norec = (it.abl & 4) != 0;
#else
  constexpr bool abl_nostore = false;
  constexpr bool abl_norec = false;
#endif
  /* whole 16-byte vectors: the 15-coefficient band shares its first vector with
     the (unused) DC slot of the block */
  if (abl_nostore) {
  }
  else if (N == 15) {
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
  if (!abl_norec) prep_write(it, rec + band, job, band, blk, xshift, rshift, p, r_null, flip, m, s);
}

/* n = G*E coefficients per group of G lanes (G = 16: a DPP row, G = 4: a quad),
   64/G bands per wavefront; lane l of the group owns coding positions l*E ..
   l*E+E-1 of the band. */
template <int E, int G>
__global__ __launch_bounds__(kWave) void k_refb_prep_row(RItems it) {
  constexpr int n = G*E;
  __shared__ unsigned short s_scan[n];
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const RJob &jb = it.jobs[job];
  const int band = it.band[item];
  const int off = jb.off[band];
  const int lane = threadIdx.x;
  for (int j = lane; j < n; j += kWave) s_scan[j] = gRScanPk[off + j];
  __syncthreads();
  const int row = lane/G;
  const int l = lane%G;
  const long nblocks = jb.nblocks;
  const long blk0 = (long)(blockIdx.x - it.wg_start[item])*(kWave/G) + row;
  const bool live = blk0 < nblocks;
  const long blk = live ? blk0 : nblocks - 1;
  const int w = jb.w;
  const int len = jb.len;
  const int nb_bands = jb.nb_bands;
  const int q0 = job_q(jb, band, blk);
  const int beta = jb.beta[band];
  const int is_keyframe = jb.is_keyframe;
  const int cfl_enabled = is_keyframe && jb.pli != 0;
  const long base = block_base(jb, blk);
  const od_coeff *x0 = jb.coef + base;
  const bool lref = jb.ly != nullptr;
  const od_coeff *r0 = lref ? jb.coef + base : jb.ref + base;
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
    const int r = lref ? 0 : r0[p];
    rv[e] = flip ? -r : r;
    qm[e] = qmp[e];
  }
  if (lref) {
    static_assert(E == 8, "one 16-byte piece of the luma stage's pulses per lane");
    int pc[8];
    lref_piece(jb, band, blk, off + l*E, pc);
#pragma unroll
    for (int e = 0; e < E; e++) rv[e] = flip ? -pc[e] : pc[e];
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int tx = (int16_t)(xv[e] >> 8);
    const int tr = (int16_t)(rv[e] >> 8);
    sx += tx*tx;
    sr += tr*tr;
    nz |= rv[e] != 0;
  }
  sx = grp_sum<G>(sx);
  sr = grp_sum<G>(sr);
  const int r_null = grp_max<G>(nz) == 0;
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
  corr = grp_sum<G>(corr);
  accx = grp_sum<G>(accx);
  accr = grp_sum<G>(accr);
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
    if (G == 16) {
      OD_ARGMAX_STEP(OD_DPP_HALF_MIRROR)
      OD_ARGMAX_STEP(OD_DPP_MIRROR)
    }
#undef OD_ARGMAX_STEP
    m = bi;
    int rm = 0;
#pragma unroll
    for (int e = 0; e < E; e++) if (l*E + e == m) rm = r16[e];
    rm = grp_sum<G>(rm);
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
    l2r = grp_sum<G>(l2r);
    proj = grp_sum<G>(proj);
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


/* ---- candidate records -------------------------------------------------------------
   odhip_pvq_refitem in memory: three planes of 16-byte vectors, each
   [nb][ODHIP_PVQ_REF_SLOTS][B] (block index fastest, so the lanes of a
   wavefront - adjacent blocks - touch one contiguous kilobyte per access):
     head  {gain, theta, ts, k}          written by the candidate kernel
     tail  {qcg, qtheta, flags, yslot}   written by the search
     res   {cos_dist, dist}              written by the search
   The first layout - 48-byte records per (block, band, slot), fields read and
   written one by one - made every access of a wavefront 64 separate sectors. */
struct ItemPtr {
  int4 *head;
  int4 *tail;
  int4 *res;
  long stride;   /* vectors between consecutive slots */
};

__device__ __forceinline__ ItemPtr item_ptr(const RJob &jb, int band, long blk) {
  ItemPtr p;
  const long plane = (long)jb.nb_bands*kSlots*jb.nblocks;
  p.head = reinterpret_cast<int4 *>(jb.items) + (long)band*kSlots*jb.nblocks + blk;
  p.tail = p.head + plane;
  p.res = p.tail + plane;
  p.stride = jb.nblocks;
  return p;
}


__device__ __forceinline__ int item_slot(const RItems &it, int item) {
  return it.job[item]*ODHIP_MAX_BANDS + it.band[item];
}

__global__ __launch_bounds__(256) void k_refb_hist(RItems it) {
  __shared__ unsigned h[kSortBins];
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = it.jobs[it.job[item]];
  const long nblocks = jb.nblocks;
  const unsigned short *keys = jb.keys + (long)it.band[item]*nblocks;
  h[threadIdx.x] = 0;
  __syncthreads();
  const long first = (long)(blockIdx.x - it.wg_start[item])*kSortChunk;
  for (int i = threadIdx.x; i < kSortChunk; i += 256) {
    const long blk = first + i;
    if (blk < nblocks) atomicAdd(&h[keys[blk]], 1u);
  }
  __syncthreads();
  const unsigned c = h[threadIdx.x];
  if (c) atomicAdd(&it.rhist[item_slot(it, item)*kSortBins + threadIdx.x], c);
}

/* One workgroup per item: exclusive prefix over the classes -> start cursors;
   clears the histogram for the next call. */
__global__ __launch_bounds__(256) void k_refb_prefix(RItems it) {
  __shared__ unsigned h[kSortBins];
  const int slot = item_slot(it, blockIdx.x);
  h[threadIdx.x] = it.rhist[slot*kSortBins + threadIdx.x];
  it.rhist[slot*kSortBins + threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned acc = 0;
    for (int b = 0; b < kSortBins; b++) {
      const unsigned c = h[b];
      h[b] = acc;
      acc += c;
    }
  }
  __syncthreads();
  it.rcursor[slot*kSortBins + threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_refb_scatter(RItems it) {
  __shared__ unsigned h[kSortBins];
  __shared__ unsigned base[kSortBins];
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = it.jobs[it.job[item]];
  const long nblocks = jb.nblocks;
  const unsigned short *keys = jb.keys + (long)it.band[item]*nblocks;
  unsigned *ids = jb.ids + (long)it.band[item]*nblocks;
  h[threadIdx.x] = 0;
  __syncthreads();
  const long first = (long)(blockIdx.x - it.wg_start[item])*kSortChunk;
  unsigned rank[kSortChunk/256];
  int key[kSortChunk/256];
#pragma unroll
  for (int t = 0; t < kSortChunk/256; t++) {
    const long blk = first + t*256 + threadIdx.x;
    key[t] = -1;
    if (blk < nblocks) {
      key[t] = keys[blk];
      rank[t] = atomicAdd(&h[key[t]], 1u);
    }
  }
  __syncthreads();
  const unsigned c = h[threadIdx.x];
  base[threadIdx.x] = c ? atomicAdd(&it.rcursor[item_slot(it, item)*kSortBins + threadIdx.x], c) : 0;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < kSortChunk/256; t++) {
    if (key[t] >= 0) ids[base[key[t]] + rank[t]] = (unsigned)(first + t*256 + threadIdx.x);
  }
}

/* ---- candidate lists -------------------------------------------------------------- */
/* The list is built in LDS (stage[slot][lane]: the stable insertion walks it
   backwards) and written to the head plane once, slot by slot - adjacent lanes are
   adjacent blocks, so every slot is one coalesced 1 KiB store per wavefront.  Round 1
   inserted directly in the head plane: a read-modify-write chain through HBM per
   candidate (242 us per 16-frame launch). */
__device__ __forceinline__ void refb_candidates(const RJob &jb, int band, long blk,
 int theta_override, int4 (*stage)[kWave]) {
  const int lane = threadIdx.x;
  odhip_pvq_refband *rp = jb.rec + blk*jb.nb_bands + band;
  odhip_pvq_refband r = *rp;
  if (theta_override >= 0) r.theta = theta_override;
  const int n = jb.off[band + 1] - jb.off[band];
  const int beta = jb.beta[band];
  const ItemPtr ip = item_ptr(jb, band, blk);
  int nitems = 0;
  int kmax_theta = 0;
  int kmax_noref = 0;
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
        const int4 c = make_int4(i, j, ts, odq_compute_k_ref(j, n));
        kmax_theta = c.w > kmax_theta ? c.w : kmax_theta;
        /* stable insertion by (k, gain): items_compare, src/pvq_encoder.c:301-305
           (glibc's qsort is a stable merge sort at this size) */
        int pos = nitems;
        while (pos > 0) {
          const int4 q = stage[pos - 1][lane];
          const int cmp = q.w == c.w ? q.x - c.x : q.w - c.w;
          if (cmp <= 0) break;
          stage[pos][lane] = q;
          pos--;
        }
        stage[pos][lane] = c;
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
      const int kk = odq_compute_k_noref(odq_shl32(i, ODQ_CGAIN_SHIFT), n, beta);
      stage[nitems][lane] = make_int4(i, -1, 0, kk);
      kmax_noref = kk > kmax_noref ? kk : kmax_noref;
      nitems++;
    }
  }
  for (int i = 0; i < nitems; i++) ip.head[i*ip.stride] = stage[i][lane];
  /* third vector of the record: {m, s, flags | theta | nitems | ntheta} */
  int4 v;
  v.x = (int)((uint32_t)(uint16_t)r.m | (uint32_t)(uint8_t)r.s << 16 | (uint32_t)flags << 24);
  v.y = r.theta;
  v.z = nitems;
  v.w = ntheta;
  reinterpret_cast<int4 *>(rp)[2] = v;
  /* work class for the sorted searches: the chains place about kmax pulses */
  jb.keys[(long)band*jb.nblocks + blk] = (unsigned short)(kSortBins - 1 - od_work_bin(kmax_theta + kmax_noref));
}

__global__ __launch_bounds__(kWave) void k_refb_cands(RItems it) {
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = it.jobs[it.job[item]];
  const long blk = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  __shared__ int4 stage[kSlots][kWave];
  if (blk >= jb.nblocks) return;
  refb_candidates(jb, it.band[item], blk, -1, stage);
}

__global__ __launch_bounds__(kWave) void k_refb_cands_list(const RJob *jobs, const Unc *list, int count) {
  __shared__ int4 stage[kSlots][kWave];
  const int i = blockIdx.x*kWave + threadIdx.x;
  if (i >= count) return;
  const Unc e = list[i];
  refb_candidates(jobs[e.job], e.band, e.blk, e.theta, stage);
}

/* ---- the candidate loops -----------------------------------------------------------
   src/pvq_encoder.c:506-565 (theta candidates) and :578-595 (no-reference
   candidates) for one band, over a vector holder V that owns the band's |x|,
   signs and pulses in whatever form its mapping uses:
     V::load(src, pad)      the band vector (pad: the last position is not part
                            of it - the n - 1 reflected coefficients)
     V::search(n, k, prev_k, g2, lambda)   pvq_search_rdo_double on it
     V::store(dst)          the signed pulses, coding order
     V::moment()            SUM i*|y_i| of the pulses it holds (od_pvq_rate's centre-of-
                            mass sum, src/pvq_encoder.c:258-259), kept in the item's
                            flags word above ODHIP_REFITEM_MOMENT_SHIFT so that the priced
                            choice never reads the vectors
   `writer`: this lane records the band's results (one lane per band). */
/* What the choice among a band's candidates keeps (src/pvq_encoder.c:417-609). */
struct RefBest {
  double best_cost;
  int chosen;              /* item, -1 = the initial candidate */
  int yslot;
  int noref;
  int32_t qtheta;
  int gain;                /* the winner's head {gain, theta, ts, k} */
  int theta;
  int ts;
  int k;
  bool close;              /* a comparison too close for the device's log */
};

/* refb_loops offers every searched candidate, in the reference's order, to a decider:
   nothing (the choice is a later kernel's) or FuseDecide below. */
struct NoDecide {
  template <class V>
  __device__ __forceinline__ void offer(int, bool, const int4 &, int32_t, int, double, int, const V &) {}
};

template <class V, class D>
__device__ __forceinline__ void refb_loops(const RJob &jb, int band, long blk, bool writer,
 bool may_store, double lambda, V &v, D &dec) {
  const odhip_pvq_refband r = jb.rec[blk*jb.nb_bands + band];
  const int off = jb.off[band];
  const int n = jb.off[band + 1] - off;
  const long nblocks = jb.nblocks;
  const int len = jb.len;
  int16_t *const yout = jb.y;
  const ItemPtr ip = item_ptr(jb, band, blk);
  const double s2 = (1./256)*(1./256);   /* OD_CGAIN_SCALE_2 */
  const double t1 = 1./32768;            /* OD_TRIG_SCALE_1 */
  const int32_t cg = r.cg;
  const double dist0 = r.dist0;
  if (r.ntheta > 0) {
    v.load(jb.xr + blk*len + off, true);
    int prev_k = 0;
    int cur_slot = -1;
    int cur_mom = 0;
    double cos_dist = 0;
    const int32_t theta = r.theta;
    int4 hnext = ip.head[0];
    for (int idx = 0; idx < r.ntheta; idx++) {
      const int4 h = hnext;                       /* gain, theta, ts, k */
      /* the next candidate's head is requested before this one is worked on */
      hnext = ip.head[(idx + 1 < r.nitems ? idx + 1 : idx)*ip.stride];
      const int32_t qcg = odq_shl32(h.x, ODQ_CGAIN_SHIFT) + r.gain_offset;
      const int32_t qtheta = odq_pvq_compute_theta(h.y, h.z);
      const int k = h.w;
      /* :526-531 */
      double dist_theta = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1;
      double dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      if (dist > dist0 + 1.0*lambda && k != 0) {
        if (writer) ip.tail[idx*ip.stride] = make_int4(qcg, qtheta, ODHIP_REFITEM_WITH_REF, -1);
        continue;
      }
      if (k > ODHIP_PVQ_MAX_K) {
        /* pulses are stored as int16: reported, never searched or chosen (the
           candidates are sorted by K, so every later one is skipped as well) */
        if (writer) {
          atomicAdd(jb.krange, 1u);
          ip.tail[idx*ip.stride] = make_int4(qcg, qtheta, ODHIP_REFITEM_WITH_REF | ODHIP_REFITEM_K_RANGE,
           -1);
        }
        continue;
      }
      const double sin_prod = ((odq_pvq_sin(theta)*t1)*odq_pvq_sin(qtheta))*t1;
      if (k == 0) {
        cos_dist = 0;
        cur_slot = -1;
        cur_mom = 0;
      }
      else if (k != prev_k) {
        cos_dist = v.search(n - 1, k, prev_k, ((qcg*(double)cg)*sin_prod)*s2, lambda);
        cur_slot = idx;
        cur_mom = v.moment();
        if (may_store) v.store(yout + ((long)idx*nblocks + blk)*len + off);
      }
      prev_k = k;
      /* :548-552 */
      dist_theta = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1 + sin_prod*(2 - 2*cos_dist);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      if (writer) {
        ip.tail[idx*ip.stride] = make_int4(qcg, qtheta,
         ODHIP_REFITEM_WITH_REF | ODHIP_REFITEM_SEARCHED | cur_mom << ODHIP_REFITEM_MOMENT_SHIFT, cur_slot);
        ip.res[idx*ip.stride] = make_int4(__double2loint(cos_dist), __double2hiint(cos_dist),
         __double2loint(dist), __double2hiint(dist));
      }
      dec.offer(idx, true, h, qtheta, cur_slot, dist, cur_mom, v);
    }
  }
  if (r.nitems > r.ntheta) {
    v.load(jb.x16 + blk*len + off, false);
    int prev_k = 0;
    for (int idx = r.ntheta; idx < r.nitems; idx++) {
      const int4 h = ip.head[idx*ip.stride];
      const int32_t qcg = odq_shl32(h.x, ODQ_CGAIN_SHIFT);
      const int k = h.w;
      /* :585-595 */
      double dist = (1.4*(qcg - cg))*(qcg - cg);
      dist *= s2;
      if (dist > dist0 && k != 0) {
        if (writer) ip.tail[idx*ip.stride] = make_int4(qcg, 0, 0, -1);
        continue;
      }
      if (k > ODHIP_PVQ_MAX_K) {
        if (writer) atomicAdd(jb.krange, 1u);
        if (writer) ip.tail[idx*ip.stride] = make_int4(qcg, 0, ODHIP_REFITEM_K_RANGE, -1);
        continue;
      }
      const double cos_dist = v.search(n, k, prev_k, (qcg*(double)cg)*s2, lambda);
      prev_k = k;
      const int mom = v.moment();
      if (may_store) v.store(yout + ((long)idx*nblocks + blk)*len + off);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*cos_dist);
      dist *= s2;
      if (writer) {
        ip.tail[idx*ip.stride] = make_int4(qcg, 0, ODHIP_REFITEM_SEARCHED | mom << ODHIP_REFITEM_MOMENT_SHIFT, idx);
        ip.res[idx*ip.stride] = make_int4(__double2loint(cos_dist), __double2hiint(cos_dist),
         __double2loint(dist), __double2hiint(dist));
      }
      dec.offer(idx, false, h, 0, idx, dist, mom, v);
    }
  }
}

/* One band per lane, |x| and pulses in LDS columns, any n (pvq_search.cuh): the
   resolve path and, with ODHIP_PVQ_REF_LANE=1, every band size. */
struct LdsVector {
  short *xs;
  unsigned short *ys;
  int lane;
  int n;
  __device__ __forceinline__ void load(const int16_t *src, bool pad) {
    const int cnt = n - (pad ? 1 : 0);
    for (int j = 0; j < cnt; j++) xs[j*kWave + lane] = src[j];
  }
  __device__ __forceinline__ double search(int n_true, int k, int prev_k, double g2, double lambda) {
    double yy;
    cur = n_true;
    return od_pvq_search_lane(xs, ys, lane, n_true, k, prev_k, g2, lambda, &yy);
  }
  __device__ __forceinline__ void store(int16_t *dst) {
    for (int j = 0; j < cur; j++) {
      const int yj = ys[j*kWave + lane];
      dst[j] = (int16_t)(xs[j*kWave + lane] < 0 ? -yj : yj);
    }
  }
  __device__ __forceinline__ int moment() const {
    int m = 0;
    for (int j = 1; j < cur; j++) m += j*ys[j*kWave + lane];
    return m;
  }
  int cur;
};

#ifdef ODHIP_EXPERIMENTS
/* ODHIP_PVQ_REF_LANE (experiments build): every band size one band per lane, vectors in LDS. */
__global__ __launch_bounds__(kWave) void k_refb_search(RItems it) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  od_rsqrt_init(threadIdx.x);
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = it.jobs[it.job[item]];
  const int band = it.band[item];
  const int n = jb.off[band + 1] - jb.off[band];
  const long pos = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  if (pos >= jb.nblocks) return;
  const long blk = jb.ids[(long)band*jb.nblocks + pos];
  LdsVector v = {(short *)lds, lds + n*kWave, (int)threadIdx.x, n, 0};
  NoDecide nd;
  refb_loops(jb, band, blk, true, true, it.lambda, v, nd);
}
#endif

/* One listed band per wavefront (lane 0): the list is a handful of bands. */
__global__ __launch_bounds__(kWave) void k_refb_search_list(const RJob *jobs, const Unc *list, int count,
 double lambda) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  od_rsqrt_init(threadIdx.x);
  if (threadIdx.x != 0 || (int)blockIdx.x >= count) return;
  const Unc e = list[blockIdx.x];
  const RJob &jb = jobs[e.job];
  LdsVector v = {(short *)lds, lds + 128*kWave, 0, jb.off[e.band + 1] - jb.off[e.band], 0};
  NoDecide nd;
  refb_loops(jb, e.band, e.blk, true, true, lambda, v, nd);
}

__device__ __forceinline__ uint32_t pack_pulses(int s0, int y0, int s1, int y1) {
  return pack16(s0 ? -y0 : y0, s1 ? -y1 : y1);
}

/* One band per group of G lanes (pvq_row.cuh): bands of G*E coefficients (G = 16,
   E = 8: 128; G = 4, E = 8: 32), all state in registers; lane l of the group owns
   positions l*E .. l*E+E-1.  The decisions of a group are uniform over its lanes
   (every lane evaluates them on the same record and items). */
template <int E, int G>
struct RowVector {
  int ax[E];
  unsigned sg;      /* bit e: coefficient e is negative */
  int y[E];
  int row;
  int l;
  int force;
  double xx;
  double norm_1;
  double cxy;       /* sum |x_j|*y_j and sum y_j^2 as the last search of the chain left them */
  double cyy;
  __device__ __forceinline__ void load(const int16_t *src, bool pad) {
    const int16_t *p = src + l*E;
    sg = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int v = p[e];
      ax[e] = abs(v);
      sg |= (unsigned)(v < 0) << e;
      y[e] = 0;
    }
    if (pad && l == G - 1) {
      ax[E - 1] = 0;
      sg &= ~(1u << (E - 1));
    }
    od_row_norm<E, G>(ax, &xx, &norm_1);
  }
  __device__ __forceinline__ double search(int n_true, int k, int prev_k, double g2, double lambda) {
#ifdef ODHIP_EXPERIMENTS
    /* how often the double-precision replay of a greedy pulse fires on real content
       (odhip_exp_row_replay_stats; tools/replay_rate.py) */
    const int placed = prev_k > 0 && prev_k <= k ? prev_k : 0;
    const int greedy = k - (1 + k/4) - placed;
    pulses += greedy > 0 ? greedy : 0;
    /* (an upper bound when k > 2 starts from the projection, which places some pulses itself) */
    rate_pulses += k - placed - (greedy > 0 ? greedy : 0);
    return od_pvq_search_row<E, G>(ax, y, row, l, n_true, k, prev_k, g2, lambda, force, xx, norm_1,
     &cxy, &cyy, &replays);
#else
    return od_pvq_search_row<E, G>(ax, y, row, l, n_true, k, prev_k, g2, lambda, force, xx, norm_1,
     &cxy, &cyy);
#endif
  }
#ifdef ODHIP_EXPERIMENTS
  int replays = 0;
  int pulses = 0;
  int rate_pulses = 0;
#endif
  __device__ __forceinline__ int moment() const {
    int m = 0;
#pragma unroll
    for (int e = 0; e < E; e++) m += (l*E + e)*y[e];
    return grp_sum<G>(m);
  }
  __device__ __forceinline__ void store(int16_t *dst) {
    int16_t *p = dst + l*E;
    if constexpr (E == 8) {
      *reinterpret_cast<uint4 *>(p) = make_uint4(pack_pulses(sg & 1, y[0], sg & 2, y[1]),
       pack_pulses(sg & 4, y[2], sg & 8, y[3]), pack_pulses(sg & 16, y[4], sg & 32, y[5]),
       pack_pulses(sg & 64, y[6], sg & 128, y[7]));
    }
    else {
#pragma unroll
      for (int e = 0; e < E; e += 2) {
        *reinterpret_cast<uint32_t *>(p + e) = pack_pulses(sg & (1u << e), y[e], sg & (2u << e), y[e + 1]);
      }
    }
  }
};

#ifdef ODHIP_EXPERIMENTS
/* [0] greedy pulses placed by the row searches of the with-reference stage, [1] those that took the
   double-precision replay (pvq_row.cuh, point 3), [2] pulses placed by the rate-penalised pass (:192-219),
   [3] bands - counted once per band */
__device__ unsigned long long gRowReplayStats[4];
#endif

/* The row search on plain band vectors (odhip_pvq_search_row_batch): one band per group of G lanes,
   no chain (prev_k = 0).  replays_out[b] = the greedy pulses of band b that the single-precision
   screen could not vouch for and replayed with the literal double-precision scan. */
template <int E, int G>
__global__ __launch_bounds__(kWave) void k_pvq_search_row(const int16_t *x_in, int n_true, const int32_t *k_in,
 od_coeff *y_out, const double *g2_in, double lambda, int force_scan, double *cos_out, int32_t *replays_out,
 long nbands) {
  constexpr int n = E*G;
  constexpr int C = kWave/G;
  od_rsqrt_init(threadIdx.x);
  const int lane = threadIdx.x;
  const int row = lane/G;
  const int l = lane%G;
  const long band = (long)blockIdx.x*C + row;
  const bool live = band < nbands;
  const long b = live ? band : nbands - 1;
  int ax[E];
  int y[E];
  unsigned sg = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int j = l*E + e;
    const int v = j < n_true ? x_in[b*n_true + j] : 0;
    ax[e] = abs(v);
    sg |= (unsigned)(v < 0) << e;
    y[e] = 0;
  }
  double xx;
  double norm_1;
  od_row_norm<E, G>(ax, &xx, &norm_1);
  double cxy = 0;
  double cyy = 0;
  int replays = 0;
  int k = k_in[b];
  k = k < 0 ? 0 : k > 32767 ? 32767 : k;
  const double c = od_pvq_search_row<E, G>(ax, y, row, l, n_true, k, 0, g2_in[b], lambda, force_scan, xx, norm_1,
   &cxy, &cyy, &replays);
  if (live) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int j = l*E + e;
      if (j < n_true) y_out[band*n_true + j] = sg >> e & 1 ? -y[e] : y[e];
    }
    if (l == 0) {
      cos_out[band] = c;
      if (replays_out) replays_out[band] = replays;
    }
  }
  (void)n;
}

template <int E, int G>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_refb_search_row(RItems it) {
  od_rsqrt_init(threadIdx.x);
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = it.jobs[it.job[item]];
  const int lane = threadIdx.x;
  RowVector<E, G> v;
  v.row = lane/G;
  v.l = lane%G;
  v.force = it.perturb >> 1;
  const long nblocks = jb.nblocks;
  const long pos = (long)(blockIdx.x - it.wg_start[item])*(kWave/G) + v.row;
  const bool live = pos < nblocks;
  /* rows beyond the end redo the last band without storing anything */
  const long blk = jb.ids[(long)it.band[item]*nblocks + (live ? pos : nblocks - 1)];
  NoDecide nd;
  refb_loops(jb, it.band[item], blk, live && v.l == 0, live, it.lambda, v, nd);
}

/* The short bands (N = 15, 8): one band per lane, the band in registers
   (pvq_regs.cuh).  Vectors are read and written as whole 16-byte pieces; the
   15-coefficient band shares its first piece with the unused DC slot. */
template <int N>
struct RegVector {
  static constexpr int SH = N == 15 ? 1 : 0;
  int ax[N];
  unsigned sg;      /* bit i: coefficient i is negative */
  int y[N];
  double xx;
  double norm_1;
  double cxy;       /* sum |x_j|*y_j and sum y_j^2 as the last search of the chain left them */
  double cyy;
  __device__ __forceinline__ void load(const int16_t *src, bool pad) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src - SH);
    sg = 0;
    uint32_t w[(N + SH)/2];
#pragma unroll
    for (int q = 0; q < (N + SH)/8; q++) {
      const uint4 t = p[q];
      w[4*q] = t.x;
      w[4*q + 1] = t.y;
      w[4*q + 2] = t.z;
      w[4*q + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int t = (int16_t)(w[(i + SH) >> 1] >> (16*((i + SH) & 1)));
      ax[i] = abs(t);
      sg |= (unsigned)(t < 0) << i;
      y[i] = 0;
    }
    if (pad) {
      ax[N - 1] = 0;
      sg &= ~(1u << (N - 1));
    }
    od_regs_norm<N>(ax, &xx, &norm_1);
  }
  __device__ __forceinline__ double search(int n_true, int k, int prev_k, double g2, double lambda) {
    return od_pvq_search_regs<N>(ax, y, n_true, k, prev_k, g2, lambda, xx, norm_1, &cxy, &cyy);
  }
  __device__ __forceinline__ int moment() const {
    int m = 0;
#pragma unroll
    for (int i = 1; i < N; i++) m += i*y[i];
    return m;
  }
  __device__ __forceinline__ void store(int16_t *dst) {
    uint4 *p = reinterpret_cast<uint4 *>(dst - SH);
    int t[N + SH];
    if (SH) t[0] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t[i + SH] = (sg >> i) & 1 ? -y[i] : y[i];
#pragma unroll
    for (int q = 0; q < (N + SH)/8; q++) {
      p[q] = make_uint4(pack16(t[8*q], t[8*q + 1]), pack16(t[8*q + 2], t[8*q + 3]),
       pack16(t[8*q + 4], t[8*q + 5]), pack16(t[8*q + 6], t[8*q + 7]));
    }
  }
};



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

template <int N, class YGet>
__device__ __forceinline__ void refb_finish(const RItems &it, int job, const RJob &jb, int band, long blk,
 const odhip_pvq_refband &r, const RefBest &best, YGet yget);

/* PRICE = 0: the host's rate table (or none); 1: od_pvq_rate's closed form (speed > 0,
   src/pvq_encoder.c:247-287) evaluated here from every searched candidate's pulses with the
   device's log - a comparison whose two costs lie within odq_rate_tol of each other is not
   trusted and the band is listed for odhip_pvq_ref_choose_priced_resolve, which prices it
   with the host libm; 2: that second decision (rates given per band). */
template <int N, int PRICE>
__device__ __forceinline__ void refb_choose_band(const RItems &it, int job, const RJob &jb, int band,
 long blk, const double *given) {
  constexpr int SH = N == 15 ? 1 : 0;     /* the 15-coefficient band is read from the DC slot on */
  constexpr int NW = (N + SH)/2;
  const long bi = blk*jb.nb_bands + band;
  const odhip_pvq_refband r = jb.rec[bi];
  const ItemPtr ip = item_ptr(jb, band, blk);
  const double *rate = PRICE == 2 ? given : PRICE == 0 && jb.rate ? jb.rate + bi*(kSlots + 1) : nullptr;
  const double lambda = it.lambda;
  const int off = jb.off[band];
  /* :417-455 (the initial candidate places no pulse and has qg = 0: its rate is 0) */
  double best_cost = r.dist0 + lambda*(rate ? rate[0] : 0.);
  bool close = false;
  int noref = jb.is_keyframe ? 1 : 0;
  int32_t best_qtheta = 0;
  int chosen = -1;
  int yslot = -1;
  /* Every candidate's result and tail vector is requested before the first one is
     looked at (static slots, predicated on the band's candidate count): one exposed
     memory latency per band instead of one per candidate - the loop was a serial chain
     of up to fourteen dependent round trips.  The head {gain, theta, ts, k} is read for
     the winner only. */
  int4 tails[kSlots];
  int dlo[kSlots];
  int dhi[kSlots];
#pragma unroll
  for (int idx = 0; idx < kSlots; idx++) {
    tails[idx] = make_int4(0, 0, 0, -1);
    dlo[idx] = 0;
    dhi[idx] = 0;
    if (idx < r.nitems) {
      tails[idx] = ip.tail[idx*ip.stride];      /* qcg, qtheta, flags, yslot */
      const int4 res = ip.res[idx*ip.stride];   /* cos_dist, dist */
      dlo[idx] = res.z;
      dhi[idx] = res.w;
    }
  }
#pragma unroll
  for (int idx = 0; idx < kSlots; idx++) {
    if (idx >= r.nitems) continue;
    const int4 tail = tails[idx];
    if (!(tail.z & ODHIP_REFITEM_SEARCHED)) continue;
    double cost = __hiloint2double(dhi[idx], dlo[idx]);
    if (PRICE == 1) {
      const int4 hd = ip.head[idx*ip.stride];     /* gain, theta, ts, k */
      const bool with_ref = idx < r.ntheta;
      const int sum = (int)((unsigned)tail.z >> ODHIP_REFITEM_MOMENT_SHIFT);
      cost = cost + lambda*odq_pvq_rate_fast(sum, hd.w, N, hd.x, with_ref ? r.icgr : 0, with_ref ? hd.y : -1,
       with_ref ? hd.z : 0, jb.is_keyframe, jb.pli);
      const double d = cost - best_cost;
      if ((d < 0 ? -d : d) <= it.tol_scale*odq_rate_tol(cost, best_cost)) close = true;
    }
    else cost = cost + lambda*(rate ? rate[1 + idx] : 0.);
    if (idx < r.ntheta ? cost < best_cost : cost <= best_cost) {
      best_cost = cost;
      chosen = idx;
      yslot = tail.w;
      if (idx < r.ntheta) {
        best_qtheta = tail.y;
        noref = 0;
      }
      else noref = 1;
    }
  }
  RefBest best;
  best.best_cost = best_cost;
  best.chosen = chosen;
  best.yslot = yslot;
  best.noref = noref;
  best.qtheta = best_qtheta;
  best.gain = best.theta = best.ts = best.k = 0;
  best.close = PRICE == 1 && close;
  if (chosen >= 0) {
    const int4 head = ip.head[chosen*ip.stride];    /* gain, theta, ts, k */
    best.gain = head.x;
    best.theta = head.y;
    best.ts = head.z;
    best.k = head.w;
  }
  /* the winner's pulses, as packed 16-byte pieces */
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
  refb_finish<N>(it, job, jb, band, blk, r, best,
   [&](int i) -> int { return (int16_t)(yw[(i + SH) >> 1] >> (16*((i + SH) & 1))); });
}

/* Everything after the decision: the band is listed when a comparison was too close, the
   skip rules (:611-622), and what od_pvq_synthesis_partial needs of the whole band
   (:623-633, src/pvq.c:1037-1115); yget(i) = the winner's signed pulse i. */
template <int N, class YGet>
__device__ __forceinline__ void refb_finish(const RItems &it, int job, const RJob &jb, int band, long blk,
 const odhip_pvq_refband &r, const RefBest &best, YGet yget) {
  constexpr int SH = N == 15 ? 1 : 0;
  constexpr int NW = (N + SH)/2;
  const long bi = blk*jb.nb_bands + band;
  const int off = jb.off[band];
  const int cfl_enabled = jb.is_keyframe && jb.pli != 0;
  const int chosen = best.chosen;
  const int yslot = best.yslot;
  const int noref = best.noref;
  const int32_t best_qtheta = best.qtheta;
  int qg = 0;
  int itheta = jb.is_keyframe ? -1 : 0;
  int max_theta = 0;
  int best_k = 0;
  if (best.close) {
    const unsigned slot = atomicAdd(it.pcount, 1u);
    if (slot < (unsigned)kPUncCap) {
      PUncR *e = it.plist + slot;
      e->job = job;
      e->band = band;
      e->blk = (unsigned)blk;
    }
  }
  if (chosen >= 0) {
    qg = best.gain;
    best_k = best.k;
    if (noref) {
      itheta = -1;
      max_theta = 0;
    }
    else {
      itheta = best.theta;
      max_theta = best.ts;
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
   job_q(jb, band, blk), jb.beta[band]);
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

/* The running choice of a band searched one per lane (RegVector): every searched candidate
   is priced with od_pvq_rate's closed form as it comes out of the search, in the
   reference's order and with its comparisons (`<` for theta candidates, :559; `<=` for the
   no-reference ones, :601), and the winner's signed pulses are kept in registers - what
   k_refb_choose<N, 1> does from the candidate records, without reading them back. */
template <int N>
struct FuseDecide {
  RefBest b;
  int y[N];
  double lambda;
  double tol_scale;
  int icgr;
  int is_keyframe;
  int pli;
  __device__ __forceinline__ void init(const odhip_pvq_refband &r, const RJob &jb, double lam, double tol) {
    b.best_cost = r.dist0;       /* the initial candidate places no pulse: its rate is 0 */
    b.chosen = -1;
    b.yslot = -1;
    b.noref = jb.is_keyframe ? 1 : 0;
    b.qtheta = 0;
    b.gain = b.theta = b.ts = b.k = 0;
    b.close = false;
    lambda = lam;
    tol_scale = tol;
    icgr = r.icgr;
    is_keyframe = jb.is_keyframe;
    pli = jb.pli;
#pragma unroll
    for (int i = 0; i < N; i++) y[i] = 0;
  }
  __device__ __forceinline__ void offer(int idx, bool with_ref, const int4 &h, int32_t qtheta, int slot,
   double dist, int mom, const RegVector<N> &v) {
    const double cost = dist + lambda*odq_pvq_rate_fast(mom, h.w, N, h.x, with_ref ? icgr : 0,
     with_ref ? h.y : -1, with_ref ? h.z : 0, is_keyframe, pli);
    const double d = cost - b.best_cost;
    if ((d < 0 ? -d : d) <= tol_scale*odq_rate_tol(cost, b.best_cost)) b.close = true;
    if (with_ref ? cost < b.best_cost : cost <= b.best_cost) {
      b.best_cost = cost;
      b.chosen = idx;
      b.yslot = slot;
      b.noref = !with_ref;
      if (with_ref) b.qtheta = qtheta;
      b.gain = h.x;
      b.theta = h.y;
      b.ts = h.z;
      b.k = h.w;
#pragma unroll
      for (int i = 0; i < N; i++) y[i] = slot < 0 ? 0 : ((v.sg >> i) & 1 ? -v.y[i] : v.y[i]);
    }
  }
};

/* FUSE: odhip_pvq_ref_bands_priced_multi - the lane also decides its band and writes the
   choice record (the candidate records are still written: a listed band's resolve and the
   theta-margin re-run read them). */
template <int N, bool FUSE>
__global__ __launch_bounds__(kWave) void k_refb_search_regs(RItems it) {
  od_rsqrt_init(threadIdx.x);
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const RJob &jb = it.jobs[job];
  const long pos = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  if (pos >= jb.nblocks) return;
  const int band = it.band[item];
  const long blk = jb.ids[(long)band*jb.nblocks + pos];
  RegVector<N> v;
  if constexpr (FUSE) {
    const odhip_pvq_refband r = jb.rec[blk*jb.nb_bands + band];
    FuseDecide<N> dec;
    dec.init(r, jb, it.lambda, it.tol_scale);
    refb_loops(jb, band, blk, true, true, it.lambda, v, dec);
    refb_finish<N>(it, job, jb, band, blk, r, dec.b, [&](int i) -> int { return dec.y[i]; });
  }
  else {
    NoDecide nd;
    refb_loops(jb, band, blk, true, true, it.lambda, v, nd);
  }
}


/* ==== the DECIDED band stage (odhip_pvq_ref_bands_decided_multi) =========================
   With od_pvq_rate's closed form the choice of a band needs nothing but the band itself, so
   the searches make it and NOTHING per candidate ever reaches HBM: no candidate heads (the
   lists are rebuilt in LDS by the kernel that walks them - round 2's candidate kernel wrote
   0.33 GB of them per 16-frame step and the searches read them back one exposed latency per
   candidate), no tail / result vectors, no pulse vector per searched candidate (1.76 GB per
   launch of the 15-coefficient bands alone).  A band costs one 64-byte record and two
   16/32-byte vectors in, one 64-byte choice record and the winner's pulses (slot 0 of the
   job's pulse buffer) out.  The resolve paths (theta inside the acos margin, a priced
   comparison inside the log margin: a handful of bands per 10^9) re-run their bands through
   the exporting kernels above.

   The pulse part of a candidate's rate is computed once per SEARCH (candidates that reuse a
   pulse vector, :539-544, share it) and .9*log2(ts) once per gain index; both are the very
   doubles odq_pvq_rate_fast forms, joined in its order. */
constexpr int kLeanAux = 12;            /* per band: ts[3], lower[3], .9*log2(ts)[3] (two words each) */

/* od_pvq_rate's pulse part is a function of (sum, k, n) alone and .9*log2(ts) of ts: for the
   small pulse counts nearly every search ends with they are read from tables filled ONCE per
   device by the very functions they replace (k_rate_fill: same code, same device log - the
   values are identical, two logs and two divisions per search become one load).  NR = band
   size; entries [k][sum], k = 1..K, sum = 0..(NR - 1)*K, K = RateTab<NR>::K: the pulse counts the
   chroma bands of a 1080p keyframe reach at the operating points measured (band 0 of the 16x16 and
   32x32 blocks: K up to 95 / 280; the 128-coefficient bands of the 32x32 blocks: up to 47). */
constexpr int kRateTs = 64;
template <int NR> struct RateTab {
  static constexpr int K = NR == 15 ? 128 : NR == 128 ? 64 : NR == 32 ? 48 : 32;
  static constexpr int W = (NR - 1)*K + 1;
  static constexpr int SIZE = (K + 1)*W;
};
__device__ double gRate8[RateTab<8>::SIZE];
__device__ double gRate15[RateTab<15>::SIZE];
__device__ double gRate32[RateTab<32>::SIZE];
/* round 4: the 128-coefficient bands too (4.2 MB): their searches run four bands to a wavefront,
   so the two logs and two divisions were paid once per candidate per FOUR bands */
__device__ double gRate128[RateTab<128>::SIZE];
__device__ double gRateTs[kRateTs];

template <int NR>
__device__ __forceinline__ double *rate_tab(void) {
  return NR == 8 ? gRate8 : NR == 15 ? gRate15 : NR == 32 ? gRate32 : gRate128;
}

template <int NR>
__global__ void k_rate_fill(void) {
  constexpr int W = RateTab<NR>::W;
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= RateTab<NR>::SIZE) return;
  const int k = i/W;
  const int sum = i - k*W;
  rate_tab<NR>()[i] = k == 0 ? 0. : odq_pvq_rate_pulses(sum, k, NR);
}

__global__ void k_rate_ts_fill(void) {
  const int ts = threadIdx.x;
  if (ts < kRateTs) gRateTs[ts] = ts > 0 ? odq_pvq_rate_ts(ts) : 0.;
}

template <int NR>
__device__ __forceinline__ double lean_rate_pulses(int sum, int k) {
  if constexpr (NR == 8 || NR == 15 || NR == 32 || NR == 128) {
    if (k <= RateTab<NR>::K) return rate_tab<NR>()[k*RateTab<NR>::W + sum];
  }
  return odq_pvq_rate_pulses(sum, k, NR);
}

__device__ __forceinline__ double lean_rate_ts(int ts) {
  if (ts > 0 && ts < kRateTs) return gRateTs[ts];
  return odq_pvq_rate_ts(ts);
}
constexpr int kLeanWords = kSlots + kLeanAux;

/* The lean kernels run kSearchWaves INDEPENDENT wavefronts per workgroup (wavefront w of workgroup g is
   work unit g*kSearchWaves + w: what a 64-thread workgroup was in rounds 3-5) that share one 1/sqrt
   table; how many of them a CU holds: od_occupancy.cuh. */
constexpr int kSearchWaves = ODHIP_SEARCH_WAVES;
constexpr int kSearchThreads = kSearchWaves*kWave;

/* A band's candidate list in LDS: word s of the band at col[s*stride].  Theta candidate:
   gi | min(k, 65535) << 2 | (j - lower[gi]) << 18 - the low 18 bits order by (k, gain) as
   items_compare does (src/pvq_encoder.c:301-305); no-reference candidate: c | k << 2. */
struct CandList {
  uint32_t *col;
  int stride;
  int ntheta;
  int nitems;
  int gb1;       /* gain index of gi = 0 */
  int gbn;       /* gain index of no-reference candidate 0 */
};

struct ListSink {
  uint32_t *col;
  int stride;
  bool writer;
  int n = 0;
  int lower_cur = 0;
  __device__ __forceinline__ void gain(int gi, int, int ts, int lower) {
    lower_cur = lower;
    if (!writer) return;
    const double l = lean_rate_ts(ts);
    col[(kSlots + gi)*stride] = (uint32_t)ts;
    col[(kSlots + 3 + gi)*stride] = (uint32_t)lower;
    col[(kSlots + 6 + 2*gi)*stride] = (uint32_t)__double2loint(l);
    col[(kSlots + 7 + 2*gi)*stride] = (uint32_t)__double2hiint(l);
  }
  __device__ __forceinline__ void theta(int gi, int, int j, int, int k) {
    const uint32_t c = (uint32_t)gi | (uint32_t)(k < 65535 ? k : 65535) << 2 | (uint32_t)(j - lower_cur) << 18;
    /* stable insertion by (k, gain): glibc's qsort is a stable merge sort at this size */
    int pos = n;
    while (pos > 0) {
      const uint32_t q = col[(pos - 1)*stride];
      if ((q & 0x3ffffu) <= (c & 0x3ffffu)) break;
      if (writer) col[pos*stride] = q;
      pos--;
    }
    if (writer) col[pos*stride] = c;
    n++;
  }
  __device__ __forceinline__ void noref(int c, int, int k) {
    if (writer) col[n*stride] = (uint32_t)c | (uint32_t)(k < 65535 ? k : 65535) << 2;
    n++;
  }
};

/* Every lane that calls this computes the list (uniformly over a group that shares the
   band); `writer` lanes store it. */
__device__ __forceinline__ CandList refb_build_list(const RJob &jb, int band, const odhip_pvq_refband &r,
 uint32_t *col, int stride, bool writer) {
  ListSink sink;
  sink.col = col;
  sink.stride = stride;
  sink.writer = writer;
  CandList cl;
  cl.col = col;
  cl.stride = stride;
  cl.gb1 = ((r.cg - r.gain_offset) >> ODQ_CGAIN_SHIFT) - 1;
  const int gbn = r.cg >> ODQ_CGAIN_SHIFT;
  cl.gbn = gbn > 1 ? gbn : 1;
  /* the theta candidates first: ListSink::n is their count when the no-reference ones start */
  struct Split {
    ListSink &s;
    int ntheta = -1;
    __device__ __forceinline__ void gain(int gi, int i, int ts, int lower) { s.gain(gi, i, ts, lower); }
    __device__ __forceinline__ void theta(int gi, int i, int j, int ts, int k) { s.theta(gi, i, j, ts, k); }
    __device__ __forceinline__ void noref(int c, int i, int k) {
      if (ntheta < 0) ntheta = s.n;
      s.noref(c, i, k);
    }
  } split{sink};
  refb_enumerate(jb, band, r.cg, r.gain_offset, r.theta, r.flags, r.corr, split);
  cl.nitems = sink.n;
  cl.ntheta = split.ntheta < 0 ? sink.n : split.ntheta;
  return cl;
}

/* The decided stage's walk over a band's candidate list (refb_loops without a single store:
   candidates from the LDS list, every searched candidate offered to the decider with its rate halves)
   for one band per LANE, with the searches of a wavefront gathered.  Every lane walks its own
   candidate list in list order (the skips, searches and offers a lock-step walk by list index would
   make - round 3's form, removed in round 5 - in the same order: the result per band is identical),
   but a lane whose next candidate needs
   a SEARCH waits while any other lane can still advance without one, so a search runs when every
   lane still working is at one.  Lists are sorted by K, not aligned between bands: walking them in
   lock step by index ran a search (~700 instructions beside its pulses) in almost every iteration
   with about half of the lanes idle (0.56-0.62 of the lanes active, SQ_THREAD_CYCLES_VALU /
   SQ_ACTIVE_INST_VALU, tools/gpu_r4_lanes.sh). */
template <int NR, class V, class D>
__device__ __forceinline__ void refb_loops_lean_gathered(const RJob &jb, int band, long blk,
 const odhip_pvq_refband &r, const CandList &cl, double lambda, V &v, D &dec) {
  const int off = jb.off[band];
  const int n = jb.off[band + 1] - off;
  const int len = jb.len;
  const double s2 = (1./256)*(1./256);   /* OD_CGAIN_SCALE_2 */
  const double t1 = 1./32768;            /* OD_TRIG_SCALE_1 */
  const int32_t cg = r.cg;
  const double dist0 = r.dist0;
  if (cl.ntheta > 0) {
    v.load(jb.xr + blk*len + off, true);
    int prev_k = 0;
    bool has = false;
    double cos_dist = 0;
    double prate = 0;
    const int32_t theta = r.theta;
    int idx = 0;
    while (idx < cl.ntheta) {
      const uint32_t w = cl.col[idx*cl.stride];
      const int gi = (int)(w & 3u);
#ifdef ODHIP_EXPERIMENTS
      const int k = dec.abl & 8 ? 0 : (int)(w >> 2 & 0xffffu);
#else
      const int k = (int)(w >> 2 & 0xffffu);
#endif
      const int i = cl.gb1 + gi;
      const int ts = (int)cl.col[(kSlots + gi)*cl.stride];
      const int j = (int)cl.col[(kSlots + 3 + gi)*cl.stride] + (int)(w >> 18);
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT) + r.gain_offset;
      const int32_t qtheta = odq_pvq_compute_theta(j, ts);
      /* :526-531 */
      double dist_theta = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1;
      double dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      const bool skip = (dist > dist0 + 1.0*lambda && k != 0) || k > ODHIP_PVQ_MAX_K;
      const bool want = !skip && k != 0 && k != prev_k;
      /* over the lanes still inside the loop; evaluated by all of them */
      const bool others = __any(!want);
      if (want && others) continue;
      idx++;
      if (skip) continue;
      const double sin_prod = ((odq_pvq_sin(theta)*t1)*odq_pvq_sin(qtheta))*t1;
      if (k == 0) {
        cos_dist = 0;
        has = false;
        prate = 0;
      }
      else if (want) {
        cos_dist = v.search(n - 1, k, prev_k, ((qcg*(double)cg)*sin_prod)*s2, lambda);
        has = true;
        prate = lean_rate_pulses<NR>(v.moment(), k);
      }
      prev_k = k;
      /* :548-552 */
      dist_theta = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1 + sin_prod*(2 - 2*cos_dist);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      const double lts = __hiloint2double((int)cl.col[(kSlots + 7 + 2*gi)*cl.stride],
       (int)cl.col[(kSlots + 6 + 2*gi)*cl.stride]);
      dec.offer(idx - 1, true, i, j, ts, k, qtheta, dist, prate, lts, has, v);
    }
  }
  if (cl.nitems > cl.ntheta) {
    v.load(jb.x16 + blk*len + off, false);
    int prev_k = 0;
    int idx = cl.ntheta;
    while (idx < cl.nitems) {
      const uint32_t w = cl.col[idx*cl.stride];
      const int i = cl.gbn + (int)(w & 3u);
#ifdef ODHIP_EXPERIMENTS
      const int k = dec.abl & 8 ? 0 : (int)(w >> 2 & 0xffffu);
#else
      const int k = (int)(w >> 2 & 0xffffu);
#endif
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT);
      /* :585-595 */
      double dist = (1.4*(qcg - cg))*(qcg - cg);
      dist *= s2;
      const bool skip = (dist > dist0 && k != 0) || k > ODHIP_PVQ_MAX_K;
      const bool others = __any(skip);
      if (!skip && others) continue;
      idx++;
      if (skip) continue;
      const double cos_dist = v.search(n, k, prev_k, (qcg*(double)cg)*s2, lambda);
      prev_k = k;
      const double prate = lean_rate_pulses<NR>(v.moment(), k);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*cos_dist);
      dist *= s2;
      dec.offer(idx - 1, false, i, -1, 0, k, 0, dist, prate, 0., true, v);
    }
  }
}

/* refb_loops_lean_gathered for a band spread over a group of G lanes.  Everything a theta candidate
   needs before its search is a function of the band record and the list entry alone (quantised gain,
   quantised theta, the two cosine terms, whether it is skipped): in lock step every lane of the group
   computed it again for every candidate - ~140 instructions each, the same values in all G lanes.
   Here lane c % G of the group computes candidate c once, before the chain, into `pre` (LDS, kPreWords
   words per candidate, the group's own slice), and the chain reads five words back. */
constexpr int kPreWords = 6;       /* qcg, skip, (2 - 2cos(theta - qtheta)/32768) lo/hi, sin(theta)sin(qtheta)/2^30 lo/hi */

template <int NR, int G, class V, class D>
__device__ __forceinline__ void refb_loops_lean_rows(const RJob &jb, int band, long blk,
 const odhip_pvq_refband &r, const CandList &cl, double lambda, V &v, D &dec, uint32_t *pre, int pre_stride) {
  const int off = jb.off[band];
  const int n = jb.off[band + 1] - off;
  const int len = jb.len;
  const double s2 = (1./256)*(1./256);   /* OD_CGAIN_SCALE_2 */
  const double t1 = 1./32768;            /* OD_TRIG_SCALE_1 */
  const int32_t cg = r.cg;
  const double dist0 = r.dist0;
  const int32_t theta = r.theta;
  /* every lane takes part (a group past the end of the item replays its last band) */
  for (int c = v.l; c < kSlots; c += G) {
    if (c < cl.ntheta) {
      const uint32_t w = cl.col[c*cl.stride];
      const int gi = (int)(w & 3u);
      const int k = (int)(w >> 2 & 0xffffu);
      const int i = cl.gb1 + gi;
      const int ts = (int)cl.col[(kSlots + gi)*cl.stride];
      const int j = (int)cl.col[(kSlots + 3 + gi)*cl.stride] + (int)(w >> 18);
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT) + r.gain_offset;
      const int32_t qtheta = odq_pvq_compute_theta(j, ts);
      /* :526-531 */
      const double cosfix = 2 - (2.*odq_pvq_cos(theta - qtheta))*t1;
      double dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*cosfix;
      dist *= s2;
      const bool skip = (dist > dist0 + 1.0*lambda && k != 0) || k > ODHIP_PVQ_MAX_K;
      const double sin_prod = ((odq_pvq_sin(theta)*t1)*odq_pvq_sin(qtheta))*t1;
      uint32_t *p = pre + c*kPreWords*pre_stride;
      p[0] = (uint32_t)qcg;
      p[pre_stride] = skip;
      p[2*pre_stride] = (uint32_t)__double2loint(cosfix);
      p[3*pre_stride] = (uint32_t)__double2hiint(cosfix);
      p[4*pre_stride] = (uint32_t)__double2loint(sin_prod);
      p[5*pre_stride] = (uint32_t)__double2hiint(sin_prod);
    }
  }
  /* `pre` is the wavefront's own slice: with several independent wavefronts per workgroup (ODHIP_SEARCH_WAVES > 1,
     some of which may have left already) only this wavefront's LDS writes have to be ordered */
  if constexpr (kSearchWaves == 1) __syncthreads();
  else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (cl.ntheta > 0) {
    v.load(jb.xr + blk*len + off, true);
    int prev_k = 0;
    bool has = false;
    double cos_dist = 0;
    double prate = 0;
    int idx = 0;
    while (idx < cl.ntheta) {
      const uint32_t w = cl.col[idx*cl.stride];
      const int k = (int)(w >> 2 & 0xffffu);
      const uint32_t *p = pre + idx*kPreWords*pre_stride;
      const bool skip = p[pre_stride] != 0;
      const bool want = !skip && k != 0 && k != prev_k;
      const bool others = __any(!want);
      if (want && others) continue;
      idx++;
      if (skip) continue;
      const int gi = (int)(w & 3u);
      const int i = cl.gb1 + gi;
      const int ts = (int)cl.col[(kSlots + gi)*cl.stride];
      const int j = (int)cl.col[(kSlots + 3 + gi)*cl.stride] + (int)(w >> 18);
      const int32_t qcg = (int32_t)p[0];
      const double cosfix = __hiloint2double((int)p[3*pre_stride], (int)p[2*pre_stride]);
      const double sin_prod = __hiloint2double((int)p[5*pre_stride], (int)p[4*pre_stride]);
      if (k == 0) {
        cos_dist = 0;
        has = false;
        prate = 0;
      }
      else if (want) {
        cos_dist = v.search(n - 1, k, prev_k, ((qcg*(double)cg)*sin_prod)*s2, lambda);
        has = true;
        prate = lean_rate_pulses<NR>(v.moment(), k);
      }
      prev_k = k;
      /* :548-552 */
      const double dist_theta = cosfix + sin_prod*(2 - 2*cos_dist);
      double dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*dist_theta;
      dist *= s2;
      const double lts = __hiloint2double((int)cl.col[(kSlots + 7 + 2*gi)*cl.stride],
       (int)cl.col[(kSlots + 6 + 2*gi)*cl.stride]);
      dec.offer(idx - 1, true, i, j, ts, k, 0, dist, prate, lts, has, v);
    }
  }
  if (cl.nitems > cl.ntheta) {
    v.load(jb.x16 + blk*len + off, false);
    int prev_k = 0;
    int idx = cl.ntheta;
    while (idx < cl.nitems) {
      const uint32_t w = cl.col[idx*cl.stride];
      const int i = cl.gbn + (int)(w & 3u);
      const int k = (int)(w >> 2 & 0xffffu);
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT);
      /* :585-595 */
      double dist = (1.4*(qcg - cg))*(qcg - cg);
      dist *= s2;
      const bool skip = (dist > dist0 && k != 0) || k > ODHIP_PVQ_MAX_K;
      const bool others = __any(skip);
      if (!skip && others) continue;
      idx++;
      if (skip) continue;
      const double cos_dist = v.search(n, k, prev_k, (qcg*(double)cg)*s2, lambda);
      prev_k = k;
      const double prate = lean_rate_pulses<NR>(v.moment(), k);
      dist = (1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*cos_dist);
      dist *= s2;
      dec.offer(idx - 1, false, i, -1, 0, k, 0, dist, prate, 0., true, v);
    }
  }
}

/* The running choice over NY pulses held by this lane (the whole band, or its E positions
   of a band spread over a group): the reference's comparisons (`<` theta candidates, :559;
   `<=` no-reference ones, :601) on cost = dist + lambda*rate.  Only the cost, the winner's
   list index and its pulses are carried through the chain; everything else about the winner
   is read back from the list at the end (lean_best). */
template <int NY>
struct LeanDecide {
  double best_cost;
  int chosen;              /* list index, -1 = the initial candidate */
  bool close;              /* a comparison too close for the device's log */
  int y[NY];
  unsigned sg;
  double lambda;
  double tol_scale;
  int icgr;
  int is_keyframe;
  int pli;
#ifdef ODHIP_EXPERIMENTS
  int abl = 0;
#endif
  __device__ __forceinline__ void init(const odhip_pvq_refband &r, const RJob &jb, double lam, double tol) {
    best_cost = r.dist0;       /* the initial candidate places no pulse: its rate is 0 */
    chosen = -1;
    close = false;
    lambda = lam;
    tol_scale = tol;
    icgr = r.icgr;
    is_keyframe = jb.is_keyframe;
    pli = jb.pli;
    sg = 0;
#pragma unroll
    for (int i = 0; i < NY; i++) y[i] = 0;
  }
  template <class V>
  __device__ __forceinline__ void offer(int idx, bool with_ref, int i, int j, int ts, int k, int32_t qtheta,
   double dist, double prate, double lts, bool has, const V &v) {
    const double rate = odq_pvq_rate_join(prate, lts, i, with_ref ? icgr : 0, with_ref ? j : -1, is_keyframe,
     pli);
    const double cost = dist + lambda*rate;
    const double d = cost - best_cost;
    if ((d < 0 ? -d : d) <= tol_scale*odq_rate_tol(cost, best_cost)) close = true;
    if (with_ref ? cost < best_cost : cost <= best_cost) {
      best_cost = cost;
      chosen = idx;
      sg = v.sg;
#pragma unroll
      for (int e = 0; e < NY; e++) y[e] = has ? v.y[e] : 0;
    }
    (void)ts;
    (void)k;
    (void)qtheta;
  }
  __device__ __forceinline__ int signed_y(int e) const {
    return (sg >> e) & 1 ? -y[e] : y[e];
  }
};

/* RefBest of the winner, decoded from the list. */
template <int NY>
__device__ __forceinline__ RefBest lean_best(const LeanDecide<NY> &dec, const CandList &cl, int is_keyframe) {
  RefBest b;
  b.best_cost = dec.best_cost;
  b.chosen = dec.chosen;
  b.yslot = -1;
  b.noref = is_keyframe ? 1 : 0;
  b.qtheta = 0;
  b.gain = b.theta = b.ts = b.k = 0;
  b.close = dec.close;
  if (dec.chosen >= 0) {
    const uint32_t w = cl.col[dec.chosen*cl.stride];
    b.k = (int)(w >> 2 & 0xffffu);
    b.yslot = b.k > 0 ? 0 : -1;     /* the winner's pulses go to slot 0 */
    if (dec.chosen < cl.ntheta) {
      const int gi = (int)(w & 3u);
      b.noref = 0;
      b.gain = cl.gb1 + gi;
      b.ts = (int)cl.col[(kSlots + gi)*cl.stride];
      b.theta = (int)cl.col[(kSlots + 3 + gi)*cl.stride] + (int)(w >> 18);
      b.qtheta = odq_pvq_compute_theta(b.theta, b.ts);
    }
    else {
      b.noref = 1;
      b.gain = cl.gbn + (int)(w & 3u);
    }
  }
  return b;
}

template <int N>
__global__ __launch_bounds__(kSearchThreads) OD_SEARCH_OCC_ATTR void k_refb_lean_lane(RItems it) {
  constexpr int SH = N == 15 ? 1 : 0;
  __shared__ uint32_t s_list_w[kSearchWaves][kLeanWords*kWave];
  OD_SEARCH_VGPR_FLOOR();
  od_rsqrt_init(threadIdx.x);
  const int lane = threadIdx.x%kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x/kWave);     /* (uniform: keeps the item search scalar) */
  const int unit = blockIdx.x*kSearchWaves + wave;
  if (unit >= it.wg_start[it.nitems]) return;
  uint32_t *s_list = s_list_w[wave];
  const int item = find_item(it, unit);
  const int job = it.job[item];
  const RJob &jb = it.jobs[job];
  const long pos = (long)(unit - it.wg_start[item])*kWave + lane;
  if (pos >= jb.nblocks) return;
  const int band = it.band[item];
  const long blk = jb.ids[(long)band*jb.nblocks + pos];
  const odhip_pvq_refband r = jb.rec[blk*jb.nb_bands + band];
  const CandList cl = refb_build_list(jb, band, r, s_list + lane, kWave, true);
  RegVector<N> v;
  LeanDecide<N> dec;
  dec.init(r, jb, it.lambda, it.tol_scale);
#ifdef ODHIP_EXPERIMENTS
  dec.abl = it.abl;
#endif
  refb_loops_lean_gathered<N>(jb, band, blk, r, cl, it.lambda, v, dec);
  const RefBest best = lean_best(dec, cl, jb.is_keyframe);
  if (best.yslot >= 0) {
    uint4 *p = reinterpret_cast<uint4 *>(jb.y + blk*jb.len + jb.off[band] - SH);
    int t[N + SH];
    if (SH) t[0] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t[i + SH] = dec.signed_y(i);
#pragma unroll
    for (int q = 0; q < (N + SH)/8; q++) {
      p[q] = make_uint4(pack16(t[8*q], t[8*q + 1]), pack16(t[8*q + 2], t[8*q + 3]),
       pack16(t[8*q + 4], t[8*q + 5]), pack16(t[8*q + 6], t[8*q + 7]));
    }
  }
  refb_finish<N>(it, job, jb, band, blk, r, best, [&](int i) -> int { return dec.signed_y(i); });
}

/* One lane to the left within the group (0 into the group's first lane). */
template <int G>
__device__ __forceinline__ int grp_from_prev(int v, int l) {
  if (G == 16) return __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
  const int t = row_mov<0x90>(v);     /* quad_perm [0,0,1,2] */
  return l == 0 ? 0 : t;
}

/* refb_finish for a band spread over a group of G lanes (lane l holds the winner's pulses
   l*E .. l*E+E-1): the decisions are uniform over the group, the band-wide sums are group
   reductions, lane 0 writes the record. */
template <int E, int G>
__device__ __forceinline__ void refb_finish_row(const RItems &it, int job, const RJob &jb, int band, long blk,
 const odhip_pvq_refband &r, const LeanDecide<E> &dec, const RefBest &best, int l, bool live) {
  constexpr int n = G*E;
  const long bi = blk*jb.nb_bands + band;
  const int off = jb.off[band];
  const int cfl_enabled = jb.is_keyframe && jb.pli != 0;
  const bool writer = live && l == 0;
  const int chosen = best.chosen;
  const int yslot = best.yslot;
  const int noref = best.noref;
  int qg = 0;
  int itheta = jb.is_keyframe ? -1 : 0;
  int max_theta = 0;
  int best_k = 0;
  if (best.close && writer) {
    const unsigned slot = atomicAdd(it.pcount, 1u);
    if (slot < (unsigned)kPUncCap) {
      PUncR *e = it.plist + slot;
      e->job = job;
      e->band = band;
      e->blk = (unsigned)blk;
    }
  }
  if (chosen >= 0) {
    qg = best.gain;
    best_k = best.k;
    if (!noref) {
      itheta = best.theta;
      max_theta = best.ts;
    }
    else {
      itheta = -1;
      max_theta = 0;
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
  if (writer) {
    ch[0] = make_int4(chosen, qg, noref, itheta);
    ch[1] = make_int4(max_theta, best_k, skip, jb.is_keyframe ? (noref ? qg : neg_interleave(qg, r.icgr))
     : (noref ? qg - 1 : neg_interleave(qg + 1, r.icgr + 1)));
  }
  if (skip) {
    if (writer) {
      const int flip = (r.flags & ODHIP_REFBAND_FLIP) != 0;
      ch[2] = make_int4(skip == 2 ? (flip ? 4 : 1) : 0, -1, 0, 0);
      ch[3] = make_int4(0, 0, 0, 0);
    }
    return;
  }
  int wy[E];
#pragma unroll
  for (int e = 0; e < E; e++) wy[e] = dec.signed_y(e);
  /* the winner's pulses: slot 0 of the job's pulse buffer */
  if (live && yslot >= 0) {
    int16_t *p = jb.y + blk*jb.len + off + l*E;
    static_assert(E == 8, "one 16-byte piece per lane");
    *reinterpret_cast<uint4 *>(p) = make_uint4(pack16(wy[0], wy[1]), pack16(wy[2], wy[3]), pack16(wy[4], wy[5]),
     pack16(wy[6], wy[7]));
  }
  const int32_t g = odq_gain_expand(odq_shl32(qg, ODQ_CGAIN_SHIFT) + (noref ? 0 : r.gain_offset),
   job_q(jb, band, blk), jb.beta[band]);
  int yy = 0;
#pragma unroll
  for (int e = 0; e < E; e++) yy += wy[e]*wy[e];    /* the pad of a theta winner holds 0 */
  yy = grp_sum<G>(yy);
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
    if (writer) {
      ch[2] = make_int4(2, yslot, scale, qshift);
      ch[3] = make_int4(0, 0, 0, 0);
    }
    return;
  }
  const int m = r.m;
  const int s = r.s;
  /* src/pvq.c:1094-1114: the two double products by 2^-15 are exact */
  scale = (int32_t)floor(.5 + (scale*(1./32768))*odq_pvq_sin(best.qtheta));
  const int16_t xm = (int16_t)floor(.5 + ((-s*odq_shr_round(g, gshift))*(1./32768))*odq_pvq_cos(best.qtheta));
  const uint4 r4 = *reinterpret_cast<const uint4 *>(jb.r16 + blk*jb.len + off + l*E);
  const uint32_t rw[4] = {r4.x, r4.y, r4.z, r4.w};
  /* position i of the band takes pulse i below the Householder pivot m and pulse i - 1 above */
  const int yprev = grp_from_prev<G>(wy[E - 1], l);
  int32_t l2r = 0;
  int32_t proj = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = l*E + e;
    const int ri = (int16_t)(rw[e >> 1] >> (16*(e & 1)));
    const int ysrc = i < m ? wy[e] : (e == 0 ? yprev : wy[e > 0 ? e - 1 : 0]);
    const int16_t xi = i == m ? xm : (int16_t)odq_mult16_32_q16(ysrc, scale);
    l2r += odq_mult16_16(ri, ri);
    proj += odq_mult16_16(ri, xi);
  }
  l2r = grp_sum<G>(l2r);
  proj = grp_sum<G>(proj);
  int16_t proj_1;
  int outshift;
  householder_consts(l2r, proj, &proj_1, &outshift);
  if (writer) {
    ch[2] = make_int4(3, yslot, scale, qshift);
    ch[3] = make_int4(xm, m, proj_1, outshift);
  }
  (void)n;
}

template <int E, int G>
__global__ __launch_bounds__(kSearchThreads) OD_SEARCH_OCC_ATTR void k_refb_lean_row(RItems it) {
  constexpr int C = kWave/G;            /* bands per wavefront */
  __shared__ uint32_t s_list_w[kSearchWaves][kLeanWords*C];
  __shared__ uint32_t s_pre_w[kSearchWaves][kSlots*kPreWords*C];
  OD_SEARCH_VGPR_FLOOR();
  od_rsqrt_init(threadIdx.x);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x/kWave);     /* (uniform: keeps the item search scalar) */
  const int unit = blockIdx.x*kSearchWaves + wave;
  if (unit >= it.wg_start[it.nitems]) return;
  uint32_t *s_list = s_list_w[wave];
  uint32_t *s_pre = s_pre_w[wave];
  const int item = find_item(it, unit);
  const int job = it.job[item];
  const RJob &jb = it.jobs[job];
  const int band = it.band[item];
  const int lane = threadIdx.x%kWave;
  RowVector<E, G> v;
  v.row = lane/G;
  v.l = lane%G;
  v.force = it.perturb >> 1;
  const long nblocks = jb.nblocks;
  const long pos = (long)(unit - it.wg_start[item])*C + v.row;
  const bool live = pos < nblocks;
  /* rows beyond the end redo the last band without storing anything */
  const long blk = jb.ids[(long)band*nblocks + (live ? pos : nblocks - 1)];
  const odhip_pvq_refband r = jb.rec[blk*jb.nb_bands + band];
  const CandList cl = refb_build_list(jb, band, r, s_list + v.row, C, v.l == 0);
  LeanDecide<E> dec;
  dec.init(r, jb, it.lambda, it.tol_scale);
  refb_loops_lean_rows<G*E, G>(jb, band, blk, r, cl, it.lambda, v, dec, s_pre + v.row, C);
  refb_finish_row<E, G>(it, job, jb, band, blk, r, dec, lean_best(dec, cl, jb.is_keyframe), v.l, live);
#ifdef ODHIP_EXPERIMENTS
  if (live && v.l == 0) {
    if (v.pulses) atomicAdd(&gRowReplayStats[0], (unsigned long long)v.pulses);
    if (v.replays) atomicAdd(&gRowReplayStats[1], (unsigned long long)v.replays);
    if (v.rate_pulses) atomicAdd(&gRowReplayStats[2], (unsigned long long)v.rate_pulses);
    atomicAdd(&gRowReplayStats[3], 1ull);
  }
#endif
}

/* The bands a theta-margin re-run rebuilt (records, candidates and pulses of every slot, by
   the exporting kernels) decided from those records, as k_refb_choose<N, 1> does. */
template <int N>
__global__ __launch_bounds__(kWave) void k_refb_choose_unc_list(RItems it, const Unc *list, int count) {
  const int i = blockIdx.x*kWave + threadIdx.x;
  if (i >= count) return;
  const Unc e = list[i];
  const RJob &jb = it.jobs[e.job];
  if (jb.off[e.band + 1] - jb.off[e.band] != N) return;
  refb_choose_band<N, 1>(it, e.job, jb, e.band, e.blk, nullptr);
}

template <int N, int PRICE>
__global__ __launch_bounds__(kWave) void k_refb_choose(RItems it) {
  const int item = find_item(it, blockIdx.x);
  const int job = it.job[item];
  const RJob &jb = it.jobs[job];
  const long blk = (long)(blockIdx.x - it.wg_start[item])*kWave + threadIdx.x;
  if (blk >= jb.nblocks) return;
  refb_choose_band<N, PRICE>(it, job, jb, it.band[item], blk, nullptr);
}

/* The listed bands of size N decided again with the host's rates. */
template <int N>
__global__ __launch_bounds__(kWave) void k_refb_choose_list(RItems it, const PUncR *list, int count) {
  const int i = blockIdx.x*kWave + threadIdx.x;
  if (i >= count) return;
  const PUncR *e = list + i;
  const RJob &jb = it.jobs[e->job];
  if (jb.off[e->band + 1] - jb.off[e->band] != N) return;
  refb_choose_band<N, 2>(it, e->job, jb, e->band, e->blk, e->rate);
}

/* Eight consecutive coding positions of one block per thread: a chunk never
   straddles a band (bands start at 1, 16, 24, 32, 64, ...; the chunk at 0 holds
   the DC, which is passed through, and the first seven coefficients of band 0).
   Pulses, reference, inverse QM and scan positions are 16-byte loads. */
__global__ __launch_bounds__(256) void k_refb_synth(RItems it) {
  const int item = find_item(it, blockIdx.x);
  const RJob &jb = it.jobs[it.job[item]];
  const int len = jb.len;
  const int cpb = len >> 3;                       /* chunks per block */
  const long t = (long)(blockIdx.x - it.wg_start[item])*256 + threadIdx.x;
  const long blk = t/cpb;
  if (blk >= jb.nblocks) return;
  const int c0 = (int)(t - blk*cpb) << 3;
  const long base = block_base(jb, blk);
  od_coeff *out = jb.dq + base;
  const int w = jb.w;
  const int band = gRBandOf[c0 ? c0 : 1];
  const int off = jb.off[band];
  const uint4 sc4 = *reinterpret_cast<const uint4 *>(gRScanPk + c0);
  const unsigned scw[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
  long pos[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const unsigned pk = (scw[e >> 1] >> (16*(e & 1))) & 0xffffu;
    pos[e] = (long)(pk >> 8)*w + (pk & 255);
  }
  const int4 *ch = reinterpret_cast<const int4 *>(jb.choice + (blk*jb.nb_bands + band)*16);
  const int4 a = ch[2];
  const int mode = a.x;
  const int first = c0 == 0;                       /* element 0 is the DC */
  if (first) out[0] = jb.coef[base];
  if (mode == 0) {
#pragma unroll
    for (int e = 0; e < 8; e++) if (!(first && e == 0)) out[pos[e]] = 0;
    return;
  }
  if (mode == 1 || mode == 4) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (first && e == 0) continue;
      const od_coeff rv = jb.ref[base + pos[e]];
      out[pos[e]] = mode == 4 ? -rv : rv;
    }
    return;
  }
  const int yslot = a.y;
  const int32_t scale = a.z;
  const int qshift = a.w;
  const uint4 q4 = *reinterpret_cast<const uint4 *>(jb.qm_inv + c0);
  const unsigned qw[4] = {q4.x, q4.y, q4.z, q4.w};
  unsigned yw[4] = {0, 0, 0, 0};
  int yprev = 0;                                   /* pulse just before the chunk */
  if (yslot >= 0) {
    const int16_t *yp = jb.y + ((long)yslot*jb.nblocks + blk)*len + c0;
    const uint4 y4 = *reinterpret_cast<const uint4 *>(yp);
    yw[0] = y4.x;
    yw[1] = y4.y;
    yw[2] = y4.z;
    yw[3] = y4.w;
    if (mode == 3 && c0 > off) yprev = yp[-1];
  }
  if (mode == 2) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (first && e == 0) continue;
      const int yv = (int16_t)(yw[e >> 1] >> (16*(e & 1)));
      const int qmi = (int16_t)(qw[e >> 1] >> (16*(e & 1)));
      const int32_t x = (int32_t)odq_mult16_32_q16(yv, scale);
      out[pos[e]] = odq_shr_round(x*qmi, qshift);
    }
    return;
  }
  const int4 b = ch[3];
  const int m = b.y;
  const uint4 r4 = *reinterpret_cast<const uint4 *>(jb.r16 + blk*len + c0);
  const unsigned rw[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
  for (int e = 0; e < 8; e++) {
    if (first && e == 0) continue;
    const int i = c0 + e - off;                    /* position inside the band */
    const int yv = (int16_t)(yw[e >> 1] >> (16*(e & 1)));
    const int ym1 = e == 0 ? yprev : (int)(int16_t)(yw[(e - 1) >> 1] >> (16*((e - 1) & 1)));
    const int qmi = (int16_t)(qw[e >> 1] >> (16*(e & 1)));
    const int ri = (int16_t)(rw[e >> 1] >> (16*(e & 1)));
    const int16_t xi = i == m ? (int16_t)b.x : (int16_t)odq_mult16_32_q16(i < m ? yv : ym1, scale);
    int32_t tmp = odq_mult16_16(ri, b.z);
    tmp = b.w >= 0 ? odq_shr_round(tmp, b.w) : odq_shl32(tmp, -b.w);
    const int16_t v = (int16_t)(xi - tmp);
    out[pos[e]] = odq_shr_round(v*qmi, qshift);
  }
}

/* ---- host side --------------------------------------------------------------------- */
odhip_device_once g_tables_once;

int upload_tables_now(void) {
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
  od_rsqrt_fill_launch();
  od_rsqrt_huge_fill_launch();
  k_rate_fill<8><<<(RateTab<8>::SIZE + 255)/256, 256, 0, 0>>>();
  k_rate_fill<15><<<(RateTab<15>::SIZE + 255)/256, 256, 0, 0>>>();
  k_rate_fill<32><<<(RateTab<32>::SIZE + 255)/256, 256, 0, 0>>>();
  k_rate_fill<128><<<(RateTab<128>::SIZE + 255)/256, 256, 0, 0>>>();
  k_rate_ts_fill<<<1, kRateTs, 0, 0>>>();
  ODHIP_TRY(hipDeviceSynchronize());
  return ODHIP_SUCCESS;
}

int upload_tables(void) {
  return odhip_once_per_device(g_tables_once, upload_tables_now);
}

/* mode 0: band stage; 1: choice + synthesis */
int fill_job(RJob &d, const odhip_pvq_refjob &j, int mode) {
  if (!j.d_coef || (!j.d_ref && !j.luma) || !j.q_band || !j.beta_band || j.bs < 0 || j.bs >= ODHIP_NBSIZES
   || j.nplanes <= 0 || !j.band || !j.items || !j.y || !j.r16) {
    return ODHIP_EINVAL;
  }
  if (((uintptr_t)j.band & 63) || ((uintptr_t)j.items & 15) || ((uintptr_t)j.y & 15)
   || ((uintptr_t)j.r16 & 15) || ((uintptr_t)j.x16 & 15) || ((uintptr_t)j.xr & 15)
   || ((uintptr_t)j.choice & 15)) {
    return ODHIP_EINVAL;
  }
  /* mode 0: band stage; 1: choice + synthesis; 2: choice only */
  if (mode == 0 ? (!j.d_qm || !j.x16 || !j.xr) : mode == 1 ? (!j.d_qm_inv || !j.choice || !j.d_dq)
   : !j.choice) {
    return ODHIP_EINVAL;
  }
  const int n = 4 << j.bs;
  if (j.w <= 0 || j.h <= 0 || j.w % n || j.h % n) return ODHIP_EINVAL;
  /* 16-byte row loads of the coefficient and reference planes */
  if (mode == 0 && (((uintptr_t)j.d_coef & 15) || ((uintptr_t)j.d_ref & 15))) return ODHIP_EINVAL;
  if (j.luma && !j.d_ref && mode == 1) return ODHIP_EINVAL;   /* the synthesis into planes may copy the reference */
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
    d.q2[i] = j.q_band[i];
    d.beta[i] = j.beta_band[i];
  }
  d.split_blk = d.nblocks;
  if (j.q_band2) {
    if (j.plane_split <= 0 || j.plane_split >= j.nplanes) return ODHIP_EINVAL;
    d.split_blk = (long)j.plane_split*d.bw*d.bh;
    for (int i = 0; i < d.nb_bands; i++) {
      if (j.q_band2[i] < 1) return ODHIP_EINVAL;
      d.q2[i] = j.q_band2[i];
    }
  }
  if (j.luma) {
    /* the luma level one size up over the same grid of blocks, nplanes or nplanes / 2 planes
       (Cb and Cr share the prediction) */
    const odhip_pvq_job &l = *j.luma;
    if (l.bs != j.bs + 1 || l.w != 2*j.w || l.h != 2*j.h || !l.cands.y || !l.cands.choice || !l.d_qm_inv
     || (l.nplanes != j.nplanes && 2*l.nplanes != j.nplanes) || !j.is_keyframe || j.pli == 0
     || ((uintptr_t)l.cands.y & 15) || ((uintptr_t)l.cands.choice & 15) || ((uintptr_t)l.d_qm_inv & 15)) {
      return ODHIP_EINVAL;
    }
    const int ln = 4 << l.bs;
    d.ly = l.cands.y;
    d.lchoice = l.cands.choice;
    d.lqmi = l.d_qm_inv;
    d.lsplit = (long)l.nplanes*d.bw*d.bh;
    d.lnblocks = d.lsplit;
    d.llen = ln*ln < OD_SCAN_LEN ? ln*ln : OD_SCAN_LEN;
    d.lnb = OD_NBANDS[l.bs];
  }
  return ODHIP_SUCCESS;
}

/* Everything the stage keeps between calls, owned by the calling thread's current
   context: ONE call sequence (bands -> resolve -> choice / synthesis) may be in
   flight per context. */
constexpr int kProfSlots = 256;
/* Device job tables are cached by content (see pvq_bands.hip): a repeating caller
   copies nothing, and the host is not stalled by pageable-memory copies. */
constexpr int kTableSlots = 8;
struct RefState {
  long theta_listed = 0;             /* bands found inside the acos margin so far (recomputed on the host) */
  RJob *d_jobs = nullptr;            /* kTableSlots device job tables of kMaxJobs   */
  unsigned *d_pcount = nullptr;      /* priced choice: bands too close to call ...  */
  PUncR *d_plist = nullptr;          /* ... and their list                          */
  unsigned *pcount_host = nullptr;   /* pinned mirror of the counter                */
  hipEvent_t pcount_event = nullptr;
  RJob host_tab[kTableSlots][kMaxJobs];
  int tab_n[kTableSlots] = {};
  unsigned long tab_stamp[kTableSlots] = {};
  unsigned long tab_clock = 0;
  const RJob *cur = nullptr;         /* the table of the call in progress           */
  unsigned *d_unc_count = nullptr;   /* bands inside the theta margin: counter ...  */
  Unc *d_unc = nullptr;              /* ... and list [kUncCap]                      */
  unsigned *d_sort = nullptr;        /* histogram + cursors of the counting sort    */
  unsigned short *keys = nullptr;    /* sort keys / sorted block indices of every   */
  unsigned *ids = nullptr;           /* (band, block) pair; grown on demand         */
  size_t cap = 0;
  hipStream_t side[2] = {nullptr, nullptr};   /* searches of the four band sizes    */
  hipEvent_t fork = nullptr;
  hipEvent_t join[2] = {nullptr, nullptr};
  unsigned *unc_host = nullptr;      /* pinned mirror of the counter                */
  hipEvent_t unc_event = nullptr;
  bool serial = false;               /* the context's setting, refreshed per call   */
  bool sort_dirty = false;           /* the sort histogram may hold counts of a failed call */
  double margin = kDefaultMargin;    /* test hooks of the context */
  int perturb = 0;
  double tol_scale = 1.;
  bool lean = false;                 /* the last band stage was the decided one: no
                                        candidate records exist unless a resolve re-ran
                                        the band                                    */
  bool prof_on = false;              /* odhip_pvq_ref_profile                       */
  bool prof_made = false;
  int prof_n = 0;
  hipEvent_t prof_ev[kProfSlots][2];
  ~RefState() {
    if (d_jobs) (void)hipFree(d_jobs);
    if (d_plist) (void)hipFree(d_plist);
    if (pcount_host) (void)hipHostFree(pcount_host);
    if (pcount_event) (void)hipEventDestroy(pcount_event);
    if (d_unc_count) (void)hipFree(d_unc_count);
    if (d_unc) (void)hipFree(d_unc);
    if (d_sort) (void)hipFree(d_sort);
    if (keys) (void)hipFree(keys);
    if (ids) (void)hipFree(ids);
    for (int i = 0; i < 2; i++) {
      if (side[i]) (void)hipStreamDestroy(side[i]);
      if (join[i]) (void)hipEventDestroy(join[i]);
    }
    if (fork) (void)hipEventDestroy(fork);
    if (unc_host) (void)hipHostFree(unc_host);
    if (unc_event) (void)hipEventDestroy(unc_event);
    if (prof_made) {
      for (int i = 0; i < kProfSlots; i++) {
        (void)hipEventDestroy(prof_ev[i][0]);
        (void)hipEventDestroy(prof_ev[i][1]);
      }
    }
  }
};

int ref_state(RefState **out) {
  ODHIP_CTX_OR_RETURN(ctx);
  RefState *st = odhip_ctx_state<RefState>(ctx, ODHIP_SLOT_REFBANDS);
  if (!st->d_jobs) {
    ODHIP_TRY(hipMalloc((void **)&st->d_jobs, sizeof(RJob)*kMaxJobs*kTableSlots));
    /* the two counters are adjacent (unc_count, pcount): one clear per band stage; the third word counts the
       bands above ODHIP_PVQ_MAX_K and is cleared only by od_k_range_take_ref */
    ODHIP_TRY(hipMalloc((void **)&st->d_unc_count, 3*sizeof(unsigned)));
    ODHIP_TRY(hipMalloc((void **)&st->d_unc, sizeof(Unc)*kUncCap));
    ODHIP_TRY(hipMalloc((void **)&st->d_sort, sizeof(unsigned)*2*kMaxItems*kSortBins));
    ODHIP_TRY(hipMemset(st->d_unc_count, 0, 3*sizeof(unsigned)));
    st->d_pcount = st->d_unc_count + 1;
    ODHIP_TRY(hipMalloc((void **)&st->d_plist, sizeof(PUncR)*kPUncCap));
    ODHIP_TRY(hipMemset(st->d_pcount, 0, sizeof(unsigned)));
    ODHIP_TRY(hipMemset(st->d_sort, 0, sizeof(unsigned)*2*kMaxItems*kSortBins));
  }
  st->serial = ctx->serial != 0;
  /* the context's test hooks (odhip_ctx_set_test_hooks), refreshed per call like `serial` */
  st->margin = ctx->theta_margin > 0 ? ctx->theta_margin : kDefaultMargin;
  st->perturb = ctx->theta_perturb != 0;
  st->tol_scale = ctx->price_tol_scale > 0 ? ctx->price_tol_scale : 1.;
  *out = st;
  return ODHIP_SUCCESS;
}
#define REF_STATE_OR_RETURN(st) \
  RefState *st##_p; \
  { \
    const int rc0_ = ref_state(&st##_p); \
    if (rc0_) return rc0_; \
  } \
  RefState &st = *st##_p

int stage_jobs(RefState &st, const odhip_pvq_refjob *jobs, int njobs, int mode, RJob *host,
 hipStream_t s) {
  if (!jobs || njobs <= 0 || njobs > kMaxJobs) return ODHIP_EINVAL;
  int rc = upload_tables();
  if (rc) return rc;
  size_t pairs = 0;
  for (int i = 0; i < njobs; i++) {
    rc = fill_job(host[i], jobs[i], mode);
    if (rc) return rc;
    host[i].krange = st.d_pcount + 1;
    pairs += (size_t)host[i].nblocks*host[i].nb_bands;
  }
  if (mode == 0) {
    if (pairs > st.cap) {
      ODHIP_TRY(hipStreamSynchronize(s));
      if (st.keys) ODHIP_TRY(hipFree(st.keys));
      if (st.ids) ODHIP_TRY(hipFree(st.ids));
      st.keys = nullptr;
      st.ids = nullptr;
      st.cap = 0;
      ODHIP_TRY(hipMalloc((void **)&st.keys, pairs*sizeof(unsigned short)));
      ODHIP_TRY(hipMalloc((void **)&st.ids, pairs*sizeof(unsigned)));
      st.cap = pairs;
    }
    pairs = 0;
    for (int i = 0; i < njobs; i++) {
      host[i].keys = st.keys + pairs;
      host[i].ids = st.ids + pairs;
      pairs += (size_t)host[i].nblocks*host[i].nb_bands;
    }
  }
  int lru = 0;
  for (int i = 0; i < kTableSlots; i++) {
    if (st.tab_n[i] == njobs && memcmp(st.host_tab[i], host, sizeof(RJob)*njobs) == 0) {
      st.tab_stamp[i] = ++st.tab_clock;
      st.cur = st.d_jobs + (size_t)i*kMaxJobs;
      return ODHIP_SUCCESS;
    }
    if (st.tab_stamp[i] < st.tab_stamp[lru]) lru = i;
  }
  if (st.tab_n[lru]) ODHIP_TRY(hipStreamSynchronize(s));
  memcpy(st.host_tab[lru], host, sizeof(RJob)*njobs);
  st.tab_n[lru] = njobs;
  st.tab_stamp[lru] = ++st.tab_clock;
  RJob *dst = st.d_jobs + (size_t)lru*kMaxJobs;
  ODHIP_TRY(hipMemcpy(dst, host, sizeof(RJob)*njobs, hipMemcpyHostToDevice));
  st.cur = dst;
  return ODHIP_SUCCESS;
}

/* Weights of the work class (KeySink::work), quarters: pulses, candidates, searches.
   ODHIP_SORT_W="p,c,s" overrides them (experiments). */
int sort_weights(void) {
  static const int w = [] {
    int p = 4, c = 4, s = 8;
    const char *e = ODHIP_EXP_ENV("ODHIP_SORT_W");
    if (e) sscanf(e, "%d,%d,%d", &p, &c, &s);
    return (p & 255) | (c & 255) << 8 | (s & 255) << 16;
  }();
  return w;
}

void items_begin(RItems &it, const RefState &st, double lambda) {
  memset(&it, 0, sizeof(it));
  it.lambda = lambda;
  it.margin = st.margin;
  it.perturb = st.perturb;
  it.jobs = st.cur;
  it.unc_count = st.d_unc_count;
  it.unc = st.d_unc;
  it.rhist = st.d_sort;
  it.rcursor = st.d_sort + kMaxItems*kSortBins;
  it.pcount = st.d_pcount;
  it.plist = st.d_plist;
  it.tol_scale = st.tol_scale;
  it.reserved1 = sort_weights();
#ifdef ODHIP_EXPERIMENTS
  /* bit 0: the preparation kernels do not store x16 / r16 / xr; 1: they take a zero reference instead of reading the
     luma choices (lref_piece); 2: they do not write the band record; 3: k_refb_lean_lane places no pulse (K = 0 for
     every candidate); 4: k_refb_lean_lane does not load its vectors.  Timing only: results are wrong. */
  static const int abl = [] {
    const char *e = getenv("ODHIP_REFB_ABL");
    return e ? atoi(e) : 0;
  }();
  it.abl = abl;
#endif
}

/* Heaviest items first: the jobs arrive by ascending block size, and the bands of the largest
   blocks place the most pulses (K ~ 70 against 0-25 for the 128-coefficient luma bands) in the fewest
   wavefronts - launched last they were the tail of their kernel.  ODHIP_ITEMS_FWD=1 keeps the
   order of the jobs (experiments). */
void items_heavy_first(RItems &it) {
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

void items_add(RItems &it, int job, int band, long wgs) {
  if (wgs <= 0) return;
  it.job[it.nitems] = (unsigned char)job;
  it.band[it.nitems] = (unsigned char)band;
  it.wg_start[it.nitems + 1] = it.wg_start[it.nitems] + (int)wgs;
  it.nitems++;
}

/* All (job, band) items; n_only > 0 keeps the bands of that size. */
void items_all(RItems &it, const RefState &st, const RJob *host, int njobs, double lambda, int n_only) {
  items_begin(it, st, lambda);
  for (int j = 0; j < njobs; j++) {
    for (int b = 0; b < host[j].nb_bands; b++) {
      if (n_only > 0 && host[j].off[b + 1] - host[j].off[b] != n_only) continue;
      items_add(it, j, b, (host[j].nblocks + kWave - 1)/kWave);
    }
  }
}

/* Side streams for the searches of the four band sizes (independent launches;
   the no-reference stage measured the same fork at 1.40 -> 1.19 ms). */
int rfork(RefState &st, hipStream_t s, hipStream_t side[2]) {
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

int rjoin(RefState &st, hipStream_t s, hipStream_t side[2]) {
  for (int i = 0; i < 2; i++) {
    if (side[i] == s) continue;
    ODHIP_TRY(hipEventRecord(st.join[i], side[i]));
    ODHIP_TRY(hipStreamWaitEvent(s, st.join[i], 0));
  }
  return ODHIP_SUCCESS;
}

}  // namespace

/* Profiling aid: HIP events around the dominant kernel of the stage - the
   row-parallel search of the 128-coefficient bands - on the stream it is launched
   on, for the calls of the current context. */
extern "C" int odhip_pvq_ref_profile(int enable) {
  REF_STATE_OR_RETURN(st);
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

extern "C" int odhip_pvq_ref_profile_read(float *ms, int max_n) {
  REF_STATE_OR_RETURN(st);
  int n = 0;
  for (; n < st.prof_n && n < max_n; n++) {
    ODHIP_TRY(hipEventSynchronize(st.prof_ev[n][1]));
    ODHIP_TRY(hipEventElapsedTime(&ms[n], st.prof_ev[n][0], st.prof_ev[n][1]));
  }
  st.prof_n = 0;
  return n;
}

extern "C" int odhip_pvq_ref_theta_probe(const double *d_corr, double *d_t, long n,
 odhip_stream stream) {
  if (!d_corr || !d_t || n < 0) return ODHIP_EINVAL;
  if (n == 0) return ODHIP_SUCCESS;
  k_theta_probe<<<(unsigned)((n + kWave - 1)/kWave), kWave, 0, (hipStream_t)stream>>>(d_corr, d_t, n);
  return odhip_check_launch();
}

namespace {
int upload_tables(void);
}

/* pvq_search_rdo_double (src/pvq_encoder.c:93-224) in its ROW form - what the 32- and 128-coefficient
   bands of the with-reference stage run (pvq_row.cuh) - on plain band vectors. */
extern "C" int odhip_pvq_search_row_batch(const int16_t *d_x, int n, const int32_t *d_k, od_coeff *d_y,
 const double *d_g2, double pvq_norm_lambda, int force_scan, double *d_cos, int32_t *d_replays, long nbands,
 odhip_stream stream) {
  if (!d_x || !d_k || !d_y || !d_g2 || !d_cos || nbands < 0) return ODHIP_EINVAL;
  if (n != 31 && n != 32 && n != 127 && n != 128) return ODHIP_EINVAL;
  if (nbands == 0) return ODHIP_SUCCESS;
  const int rc = upload_tables();
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int per_wg = n >= 127 ? 4 : 16;
  const long grid = (nbands + per_wg - 1)/per_wg;
  if (grid > 0x7fffffffL) return ODHIP_EINVAL;
  if (n >= 127) {
    k_pvq_search_row<8, 16><<<(unsigned)grid, kWave, 0, s>>>(d_x, n, d_k, d_y, d_g2, pvq_norm_lambda, force_scan, d_cos,
     d_replays, nbands);
  }
  else {
    k_pvq_search_row<8, 4><<<(unsigned)grid, kWave, 0, s>>>(d_x, n, d_k, d_y, d_g2, pvq_norm_lambda, force_scan, d_cos,
     d_replays, nbands);
  }
  return odhip_check_launch();
}

#ifdef ODHIP_EXPERIMENTS
/* out[0] = greedy pulses placed by the row searches of the with-reference band stage since the last
   reset, out[1] = those that took the double-precision replay, out[2] = pulses of the rate-penalised pass,
   out[3] = bands.  Synchronises the device. */
extern "C" int odhip_exp_row_replay_stats(unsigned long long *out, int reset) {
  if (!out) return ODHIP_EINVAL;
  ODHIP_TRY(hipDeviceSynchronize());
  ODHIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(gRowReplayStats), 4*sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long zero[4] = {0, 0, 0, 0};
    ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gRowReplayStats), zero, sizeof(zero)));
  }
  return ODHIP_SUCCESS;
}
#endif

namespace {
int ref_bands(const odhip_pvq_refjob *jobs, int njobs, double pvq_norm_lambda, odhip_stream stream, int fuse);
}

extern "C" int odhip_pvq_ref_bands_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  return ref_bands(jobs, njobs, pvq_norm_lambda, stream, 0);
}

/* The band stage with the priced choice of the bands searched one per lane (the 15- and
   8-coefficient bands, 77 % of all bands) made inside their search kernels; follow with
   odhip_pvq_ref_choose_priced_rest_multi for the 32- and 128-coefficient bands and with
   odhip_pvq_ref_choose_priced_resolve.  A job must carry its choice buffer. */
extern "C" int odhip_pvq_ref_bands_priced_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  for (int j = 0; jobs && j < njobs; j++) {
    if (!jobs[j].choice) return ODHIP_EINVAL;
  }
  return ref_bands(jobs, njobs, pvq_norm_lambda, stream, 1);
}

/* The whole band stage with the priced choice of EVERY band made inside its search (see
   "the DECIDED band stage" above): choice records and the winners' pulse vectors (slot 0 of
   each job's y) are the only outputs; the candidate arrays of the jobs (items, the other
   slots of y) are scratch of the resolve paths.  The counts of bands inside the theta margin
   and inside the price margin are on their way to the host when this returns; follow with
   odhip_pvq_ref_resolve_finish and odhip_pvq_ref_choose_priced_resolve (both normally find
   nothing and return 0; a band they re-run is decided again by them), then consume the
   choices.  A job must carry its choice buffer. */
extern "C" int odhip_pvq_ref_bands_decided_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  for (int j = 0; jobs && j < njobs; j++) {
    if (!jobs[j].choice) return ODHIP_EINVAL;
  }
  return ref_bands(jobs, njobs, pvq_norm_lambda, stream, 2);
}

namespace {
int send_pcount(RefState &st, hipStream_t s) {
  if (!st.pcount_host) {
    ODHIP_TRY(hipHostMalloc((void **)&st.pcount_host, sizeof(unsigned), hipHostMallocDefault));
    ODHIP_TRY(hipEventCreateWithFlags(&st.pcount_event, hipEventDisableTiming));
  }
  *st.pcount_host = 0xffffffffu;
  ODHIP_TRY(hipMemcpyAsync(st.pcount_host, st.d_pcount, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  ODHIP_TRY(hipEventRecord(st.pcount_event, s));
  return ODHIP_SUCCESS;
}

int ref_bands(const odhip_pvq_refjob *jobs, int njobs, double pvq_norm_lambda, odhip_stream stream, int fuse) {
  hipStream_t s = (hipStream_t)stream;
  REF_STATE_OR_RETURN(st);
  RJob host[kMaxJobs];
  int rc = stage_jobs(st, jobs, njobs, 0, host, s);
  if (rc) return rc;
  ODHIP_TRY(hipMemsetAsync(st.d_unc_count, 0, (fuse ? 2 : 1)*sizeof(unsigned), s));
  /* the histogram is consumed and cleared by k_refb_prefix; only a call that failed
     between the two leaves it dirty */
  if (st.sort_dirty) {
    ODHIP_TRY(hipMemsetAsync(st.d_sort, 0, sizeof(unsigned)*kMaxItems*kSortBins, s));
    st.sort_dirty = false;
  }
  st.lean = fuse == 2;
  RItems it;
  items_all(it, st, host, njobs, pvq_norm_lambda, 0);
  if (!it.nitems) return ODHIP_SUCCESS;
  /* band 0 of every block first (it decides the chroma-from-luma flip of the
     block), then the 8-coefficient bands per lane and the 32- / 128-coefficient
     bands per row */
  {
    RItems pi;
    items_begin(pi, st, pvq_norm_lambda);
    pi.fuse = fuse;
    for (int j = 0; j < njobs; j++) items_add(pi, j, 0, (host[j].nblocks + kWave - 1)/kWave);
    k_refb_prep_lane<15><<<pi.wg_start[pi.nitems], kWave, 0, s>>>(pi);
    items_begin(pi, st, pvq_norm_lambda);
    pi.fuse = fuse;
    for (int j = 0; j < njobs; j++) {
      for (int b = 1; b < host[j].nb_bands; b++) {
        if (host[j].off[b + 1] - host[j].off[b] == 8) items_add(pi, j, b, (host[j].nblocks + kWave - 1)/kWave);
      }
    }
    if (pi.nitems) k_refb_prep_lane<8><<<pi.wg_start[pi.nitems], kWave, 0, s>>>(pi);
    for (int sz = 32; sz <= 128; sz *= 4) {
      items_begin(pi, st, pvq_norm_lambda);
      pi.fuse = fuse;
      for (int j = 0; j < njobs; j++) {
        for (int b = 1; b < host[j].nb_bands; b++) {
          if (host[j].off[b + 1] - host[j].off[b] == sz) {
            items_add(pi, j, b, sz == 32 ? (host[j].nblocks + 15)/16 : (host[j].nblocks + 3)/4);
          }
        }
      }
      if (!pi.nitems) continue;
      if (sz == 32) k_refb_prep_row<8, 4><<<pi.wg_start[pi.nitems], kWave, 0, s>>>(pi);
      else k_refb_prep_row<8, 16><<<pi.wg_start[pi.nitems], kWave, 0, s>>>(pi);
    }
  }
  /* (the decided stage has no candidate kernel: the work classes came from the preparation) */
  if (fuse != 2) k_refb_cands<<<it.wg_start[it.nitems], kWave, 0, s>>>(it);
  /* counting sort of every item's blocks by work class */
  {
    RItems chunks;
    RItems all;
    items_begin(chunks, st, pvq_norm_lambda);
    items_begin(all, st, pvq_norm_lambda);
    for (int j = 0; j < njobs; j++) {
      for (int b = 0; b < host[j].nb_bands; b++) {
        items_add(chunks, j, b, (host[j].nblocks + kSortChunk - 1)/kSortChunk);
        items_add(all, j, b, 1);
      }
    }
    st.sort_dirty = true;
    k_refb_hist<<<chunks.wg_start[chunks.nitems], 256, 0, s>>>(chunks);
    k_refb_prefix<<<all.nitems, 256, 0, s>>>(all);
    st.sort_dirty = odhip_check_launch() != ODHIP_SUCCESS;
    k_refb_scatter<<<chunks.wg_start[chunks.nitems], 256, 0, s>>>(chunks);
  }
  /* 128- and 32-coefficient bands: one band per 16-lane row; 15 and 8: per lane */
  const bool lane_only = fuse != 2 && ODHIP_EXP_ENV("ODHIP_PVQ_REF_LANE") != nullptr;
  static const int sizes[4] = {128, 32, 15, 8};
  hipStream_t side[2] = {s, s};
  if (rfork(st, s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
  const hipStream_t main_stream = s;
  for (int i = 0; i < 4; i++) {
    /* 128 on the caller's stream, 15 and 8 on one side stream, 32 on the other */
    s = sizes[i] == 128 ? main_stream : sizes[i] == 32 ? side[0] : side[1];
    if (sizes[i] >= 32 && !lane_only) {
      items_begin(it, st, pvq_norm_lambda);
      if (odhip_env_force_seq()) it.perturb |= 2;   /* every greedy pulse by the literal scan */
      for (int j = 0; j < njobs; j++) {
        for (int b = 0; b < host[j].nb_bands; b++) {
          if (host[j].off[b + 1] - host[j].off[b] == sizes[i]) {
            items_add(it, j, b, sizes[i] == 32 ? (host[j].nblocks + 15)/16 : (host[j].nblocks + 3)/4);
          }
        }
      }
      if (!it.nitems) continue;
      if (fuse == 2) items_heavy_first(it);
      if (sizes[i] == 128) {
        const bool prof = st.prof_on && st.prof_n < kProfSlots;
        if (prof) (void)hipEventRecord(st.prof_ev[st.prof_n][0], s);
        if (fuse == 2) k_refb_lean_row<8, 16><<<(it.wg_start[it.nitems] + kSearchWaves - 1)/kSearchWaves, kSearchThreads, 0, s>>>(it);
        else k_refb_search_row<8, 16><<<it.wg_start[it.nitems], kWave, 0, s>>>(it);
        if (prof) (void)hipEventRecord(st.prof_ev[st.prof_n++][1], s);
      }
      else if (fuse == 2) k_refb_lean_row<8, 4><<<(it.wg_start[it.nitems] + kSearchWaves - 1)/kSearchWaves, kSearchThreads, 0, s>>>(it);
      else k_refb_search_row<8, 4><<<it.wg_start[it.nitems], kWave, 0, s>>>(it);
      continue;
    }
    items_all(it, st, host, njobs, pvq_norm_lambda, sizes[i]);
    if (!it.nitems) continue;
    if (sizes[i] < 32 && !lane_only) {
      if (fuse == 2) items_heavy_first(it);
      const int wgs = it.wg_start[it.nitems];
      if (fuse == 2) {
        const int groups = (wgs + kSearchWaves - 1)/kSearchWaves;
        if (sizes[i] == 15) k_refb_lean_lane<15><<<groups, kSearchThreads, 0, s>>>(it);
        else k_refb_lean_lane<8><<<groups, kSearchThreads, 0, s>>>(it);
      }
      else if (fuse) {
        if (sizes[i] == 15) k_refb_search_regs<15, true><<<wgs, kWave, 0, s>>>(it);
        else k_refb_search_regs<8, true><<<wgs, kWave, 0, s>>>(it);
      }
      else if (sizes[i] == 15) k_refb_search_regs<15, false><<<wgs, kWave, 0, s>>>(it);
      else k_refb_search_regs<8, false><<<wgs, kWave, 0, s>>>(it);
      continue;
    }
#ifdef ODHIP_EXPERIMENTS
    /* ODHIP_PVQ_REF_LANE: every size one band per lane with its vectors in LDS (rounds 1-2) */
    const size_t lds = (size_t)2*sizes[i]*kWave*sizeof(unsigned short);
    k_refb_search<<<it.wg_start[it.nitems], kWave, lds, s>>>(it);
#endif
  }
  s = main_stream;
  if (rjoin(st, s, side) != ODHIP_SUCCESS) return ODHIP_EFAULT;
  if (fuse == 2) {
    /* both counts of listed bands on their way to the host */
    const int rc2 = odhip_pvq_ref_resolve_begin(s);
    if (rc2) return rc2;
    const int rc3 = send_pcount(st, s);
    if (rc3) return rc3;
  }
  return odhip_check_launch();
}
}  // namespace

/* The count of listed bands travels to pinned host memory behind the band stage;
   nothing waits for it here. */
extern "C" int odhip_pvq_ref_resolve_begin(odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  REF_STATE_OR_RETURN(st);
  if (!st.unc_host) {
    ODHIP_TRY(hipHostMalloc((void **)&st.unc_host, sizeof(unsigned), hipHostMallocDefault));
    ODHIP_TRY(hipEventCreateWithFlags(&st.unc_event, hipEventDisableTiming));
  }
  *st.unc_host = 0xffffffffu;
  ODHIP_TRY(hipMemcpyAsync(st.unc_host, st.d_unc_count, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  ODHIP_TRY(hipEventRecord(st.unc_event, s));
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pvq_ref_resolve(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);

/* Bands of the current context found inside the margin of the device acos so far: each one had its theta
   recomputed by the host's libm (odhip_pvq_ref_resolve); the ones whose theta CHANGED are the resolve's return
   value. */
extern "C" long odhip_pvq_ref_theta_listed(void) {
  RefState *st = nullptr;
  if (ref_state(&st) != ODHIP_SUCCESS) return -1;
  return st->theta_listed;
}

extern "C" int odhip_pvq_ref_resolve_finish(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  REF_STATE_OR_RETURN(st);
  if (!st.unc_event) return ODHIP_EINVAL;
  ODHIP_TRY(hipEventSynchronize(st.unc_event));
  if (*st.unc_host == 0) return 0;
  return odhip_pvq_ref_resolve(jobs, njobs, pvq_norm_lambda, stream);
}

extern "C" int odhip_pvq_ref_resolve(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  REF_STATE_OR_RETURN(st);
  ODHIP_TRY(hipStreamSynchronize(s));
  unsigned count = 0;
  ODHIP_TRY(hipMemcpy(&count, st.d_unc_count, sizeof(count), hipMemcpyDeviceToHost));
  if (count == 0) return 0;
  st.theta_listed += count;
  if (count > (unsigned)kUncCap) {
    fprintf(stderr, "libdaalahip: %u bands inside the theta margin exceed the list (%d)\n", count,
     kUncCap);
    return ODHIP_EFAULT;
  }
  Unc *list = (Unc *)malloc(sizeof(Unc)*count);
  if (!list) return ODHIP_EFAULT;
  if (hipMemcpy(list, st.d_unc, sizeof(Unc)*count, hipMemcpyDeviceToHost) != hipSuccess) {
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
  int rc = stage_jobs(st, jobs, njobs, 0, host, s);
  if (rc) {
    free(list);
    return rc;
  }
  for (unsigned i = 0; i < nfix; i++) {
    if (list[i].job < 0 || list[i].job >= njobs) {
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
  k_refb_cands_list<<<(nfix + kWave - 1)/kWave, kWave, 0, s>>>(st.cur, d_list, (int)nfix);
  k_refb_search_list<<<nfix, kWave, (size_t)2*128*kWave*sizeof(unsigned short), s>>>(st.cur,
   d_list, (int)nfix, pvq_norm_lambda);
  if (st.lean) {
    /* the decided stage: nobody else will choose for these bands - decided here from the
       records the re-run just wrote (a close call joins the price list, whose count is
       sent again) */
    RItems it;
    items_begin(it, st, pvq_norm_lambda);
    const unsigned grid = (nfix + kWave - 1)/kWave;
    k_refb_choose_unc_list<128><<<grid, kWave, 0, s>>>(it, d_list, (int)nfix);
    k_refb_choose_unc_list<32><<<grid, kWave, 0, s>>>(it, d_list, (int)nfix);
    k_refb_choose_unc_list<15><<<grid, kWave, 0, s>>>(it, d_list, (int)nfix);
    k_refb_choose_unc_list<8><<<grid, kWave, 0, s>>>(it, d_list, (int)nfix);
    (void)send_pcount(st, s);
  }
  rc = odhip_check_launch();
  hipError_t e = hipStreamSynchronize(s);
  free(list);
  (void)hipFree(d_list);
  if (rc) return rc;
  if (e != hipSuccess) return ODHIP_EFAULT;
  return (int)nfix;
}

namespace {
int ref_select(const odhip_pvq_refjob *jobs, int njobs, double pvq_norm_lambda, hipStream_t s,
 bool synth, bool price = false, bool rest_only = false);
}  // namespace

extern "C" int odhip_pvq_ref_select_synth_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  return ref_select(jobs, njobs, pvq_norm_lambda, (hipStream_t)stream, true);
}

extern "C" int odhip_pvq_ref_choose_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  return ref_select(jobs, njobs, pvq_norm_lambda, (hipStream_t)stream, false);
}

namespace {
int ref_select(const odhip_pvq_refjob *jobs, int njobs, double pvq_norm_lambda, hipStream_t s,
 bool synth, bool price, bool rest_only) {
  REF_STATE_OR_RETURN(st);
  RJob host[kMaxJobs];
  int rc = stage_jobs(st, jobs, njobs, synth ? 1 : 2, host, s);
  if (rc) return rc;
  for (int j = 0; synth && j < njobs; j++) {
    /* 32x32 and 64x64 blocks code their lowest 512 coefficients only; the rest is what
       od_init_skipped_coeffs leaves (src/state.c:1347-1366): zero on a keyframe, the
       prediction's coefficients on an inter frame */
    if (host[j].bs >= 3) {
      const size_t bytes = sizeof(od_coeff)*(size_t)host[j].nplanes*host[j].w*host[j].h;
      if (host[j].is_keyframe) ODHIP_TRY(hipMemsetAsync(host[j].dq, 0, bytes, s));
      else ODHIP_TRY(hipMemcpyAsync(host[j].dq, host[j].ref, bytes, hipMemcpyDeviceToDevice, s));
    }
  }
  /* rest_only: the per-lane bands were decided (and their close calls listed) inside
     odhip_pvq_ref_bands_priced_multi, which also cleared the counter */
  if (price && !rest_only) ODHIP_TRY(hipMemsetAsync(st.d_pcount, 0, sizeof(unsigned), s));
  RItems it;
  static const int sizes[4] = {128, 32, 15, 8};
  for (int i = 0; i < (rest_only ? 2 : 4); i++) {
    items_all(it, st, host, njobs, pvq_norm_lambda, sizes[i]);
    if (!it.nitems) continue;
    const unsigned grid = it.wg_start[it.nitems];
    if (price) {
      if (sizes[i] == 128) k_refb_choose<128, 1><<<grid, kWave, 0, s>>>(it);
      else if (sizes[i] == 32) k_refb_choose<32, 1><<<grid, kWave, 0, s>>>(it);
      else if (sizes[i] == 15) k_refb_choose<15, 1><<<grid, kWave, 0, s>>>(it);
      else k_refb_choose<8, 1><<<grid, kWave, 0, s>>>(it);
    }
    else if (sizes[i] == 128) k_refb_choose<128, 0><<<grid, kWave, 0, s>>>(it);
    else if (sizes[i] == 32) k_refb_choose<32, 0><<<grid, kWave, 0, s>>>(it);
    else if (sizes[i] == 15) k_refb_choose<15, 0><<<grid, kWave, 0, s>>>(it);
    else k_refb_choose<8, 0><<<grid, kWave, 0, s>>>(it);
  }
  if (price) {
    const int rcp = send_pcount(st, s);
    if (rcp) return rcp;
  }
  if (!synth) return odhip_check_launch();
  items_begin(it, st, pvq_norm_lambda);
  for (int j = 0; j < njobs; j++) items_add(it, j, 0, (host[j].nblocks*(host[j].len >> 3) + 255)/256);
  k_refb_synth<<<it.wg_start[it.nitems], 256, 0, s>>>(it);
  return odhip_check_launch();
}
}  // namespace

/* The choice alone with od_pvq_rate's closed form (speed > 0) evaluated on the device from
   every searched candidate's pulses (see refb_choose_band); odhip_pvq_ref_choose_priced_resolve
   waits for the stream and settles the bands whose decision was too close to take from the
   device's log with the host libm.  Returns how many (normally 0) or a negative code. */
extern "C" int odhip_pvq_ref_choose_priced_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  return ref_select(jobs, njobs, pvq_norm_lambda, (hipStream_t)stream, false, true);
}

/* After odhip_pvq_ref_bands_priced_multi: the priced choice of the bands it did not decide
   (32 and 128 coefficients, searched one per group of lanes), and the count of listed
   bands of both on its way to the host. */
extern "C" int odhip_pvq_ref_choose_priced_rest_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  return ref_select(jobs, njobs, pvq_norm_lambda, (hipStream_t)stream, false, true, true);
}

extern "C" int odhip_pvq_ref_choose_priced_resolve(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  REF_STATE_OR_RETURN(st);
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
  RJob host[kMaxJobs];
  int rc = stage_jobs(st, jobs, njobs, 2, host, s);
  if (rc) return rc;
  PUncR *list = (PUncR *)malloc(sizeof(PUncR)*count);
  if (!list) return ODHIP_EFAULT;
  if (hipMemcpy(list, st.d_plist, sizeof(PUncR)*count, hipMemcpyDeviceToHost) != hipSuccess) rc = ODHIP_EFAULT;
  for (unsigned i = 0; i < count && !rc; i++) {
    if (list[i].job < 0 || list[i].job >= njobs) rc = ODHIP_EINVAL;
  }
  if (!rc && st.lean) {
    /* the decided stage kept no candidate records: the listed bands are searched again by
       the exporting kernels (their theta as the record holds it) */
    Unc *ul = (Unc *)malloc(sizeof(Unc)*count);
    Unc *d_ul = nullptr;
    if (!ul) rc = ODHIP_EFAULT;
    for (unsigned i = 0; i < count && !rc; i++) {
      ul[i].job = list[i].job;
      ul[i].band = list[i].band;
      ul[i].blk = list[i].blk;
      ul[i].theta = -1;
      ul[i].corr = 0;
    }
    if (!rc && (hipMalloc((void **)&d_ul, sizeof(Unc)*count) != hipSuccess
     || hipMemcpy(d_ul, ul, sizeof(Unc)*count, hipMemcpyHostToDevice) != hipSuccess)) {
      rc = ODHIP_EFAULT;
    }
    /* the band-stage form of the job table (sort scratch, work vectors) for the re-run, then
       the choice form again */
    RJob host0[kMaxJobs];
    if (!rc) rc = stage_jobs(st, jobs, njobs, 0, host0, s);
    if (!rc) {
      k_refb_cands_list<<<(count + kWave - 1)/kWave, kWave, 0, s>>>(st.cur, d_ul, (int)count);
      k_refb_search_list<<<count, kWave, (size_t)2*128*kWave*sizeof(unsigned short), s>>>(st.cur, d_ul,
       (int)count, pvq_norm_lambda);
      rc = odhip_check_launch();
      if (hipStreamSynchronize(s) != hipSuccess) rc = ODHIP_EFAULT;
    }
    if (!rc) rc = stage_jobs(st, jobs, njobs, 2, host, s);
    free(ul);
    if (d_ul) (void)hipFree(d_ul);
  }
  for (unsigned i = 0; i < count && !rc; i++) {
    PUncR &e = list[i];
    const RJob &jb = host[e.job];
    const int n = jb.off[e.band + 1] - jb.off[e.band];
    const long B = jb.nblocks;
    odhip_pvq_refband rec;
    if (hipMemcpy(&rec, jb.rec + (long)e.blk*jb.nb_bands + e.band, sizeof(rec), hipMemcpyDeviceToHost)
     != hipSuccess) {
      rc = ODHIP_EFAULT;
      break;
    }
    const int4 *head = reinterpret_cast<const int4 *>(jb.items) + (long)e.band*kSlots*B + e.blk;
    const int4 *tail = head + (long)jb.nb_bands*kSlots*B;
    for (int c = 0; c <= kSlots; c++) e.rate[c] = 0;
    for (int idx = 0; idx < rec.nitems && idx < kSlots && !rc; idx++) {
      int4 hd;
      int4 tl;
      if (hipMemcpy(&hd, head + (long)idx*B, sizeof(hd), hipMemcpyDeviceToHost) != hipSuccess
       || hipMemcpy(&tl, tail + (long)idx*B, sizeof(tl), hipMemcpyDeviceToHost) != hipSuccess) {
        rc = ODHIP_EFAULT;
        break;
      }
      if (!(tl.z & ODHIP_REFITEM_SEARCHED)) continue;
      const bool with_ref = idx < rec.ntheta;
      const int sum = (int)((unsigned)tl.z >> ODHIP_REFITEM_MOMENT_SHIFT);
      e.rate[1 + idx] = odq_pvq_rate_fast_host(sum, hd.w, n, hd.x, with_ref ? rec.icgr : 0,
       with_ref ? hd.y : -1, with_ref ? hd.z : 0, jb.is_keyframe, jb.pli);
    }
  }
  PUncR *d_list = nullptr;
  if (!rc && (hipMalloc((void **)&d_list, sizeof(PUncR)*count) != hipSuccess
   || hipMemcpy(d_list, list, sizeof(PUncR)*count, hipMemcpyHostToDevice) != hipSuccess)) {
    rc = ODHIP_EFAULT;
  }
  free(list);
  if (!rc) {
    RItems it;
    items_begin(it, st, pvq_norm_lambda);
    const unsigned grid = (count + kWave - 1)/kWave;
    k_refb_choose_list<128><<<grid, kWave, 0, s>>>(it, d_list, (int)count);
    k_refb_choose_list<32><<<grid, kWave, 0, s>>>(it, d_list, (int)count);
    k_refb_choose_list<15><<<grid, kWave, 0, s>>>(it, d_list, (int)count);
    k_refb_choose_list<8><<<grid, kWave, 0, s>>>(it, d_list, (int)count);
    rc = odhip_check_launch();
    if (hipStreamSynchronize(s) != hipSuccess) rc = ODHIP_EFAULT;
  }
  if (d_list) (void)hipFree(d_list);
  return rc ? rc : (int)count;
}

/* od_krange.cuh */
int od_k_range_take_ref(unsigned *count) {
  RefState *st = nullptr;
  const int rc = ref_state(&st);
  if (rc) return rc;
  unsigned v = 0;
  ODHIP_TRY(hipMemcpy(&v, st->d_pcount + 1, sizeof(v), hipMemcpyDeviceToHost));
  if (v) ODHIP_TRY(hipMemset(st->d_pcount + 1, 0, sizeof(v)));
  *count = v;
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pvq_k_range_take(unsigned *noref_bands, unsigned *ref_bands) {
  unsigned a = 0;
  unsigned b = 0;
  const int rc = od_k_range_take_noref(&a);
  if (rc) return rc;
  const int rc2 = od_k_range_take_ref(&b);
  if (rc2) return rc2;
  if (noref_bands) *noref_bands = a;
  if (ref_bands) *ref_bands = b;
  return a || b ? ODHIP_ERANGE : ODHIP_SUCCESS;
}
