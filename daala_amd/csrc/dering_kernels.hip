/* dering_kernels.hip - the deringing filter of the reference (src/dering.c),
   SURVEY.md 8(f) rank 1, one superblock (64x64 luma / 32x32 chroma) per workgroup.

   A superblock reads only the UNFILTERED plane x (its own samples plus a 3-sample
   border, src/dering.c:270-279), so every superblock of every plane - and every
   candidate threshold the encoder's level search tries (src/encode.c:2785-2810)
   - is independent: the x tile, the direction search and everything of the
   directional pass that does not depend on the threshold are done once per
   workgroup, the thresholded part of the two passes once per candidate.

   Mapping (round 2; round 1 ran one thread per pixel on scalar int16 and one thread
   per block for the direction search, 7 % of HBM):
     * one thread per ROW SEGMENT of a filter block (8 samples of an 8x8 luma block, 4 of
       a 4x4 chroma block): direction and threshold are uniform over a thread, and its
       samples are 4 / 2 registers of PACKED int16 pairs.  The reference's arithmetic is
       int16 with wrap-around casts - exactly what v_pk_{sub,add,mad}_i16 compute - so
       every filter step is one packed instruction for two samples;
     * tile A = x with border in LDS (30000 where the border is outside the frame, the
       reference's OD_DERING_VERY_LARGE), interior 16-byte aligned; a tap window at any
       (dy, dx) is read as whole dwords from the aligned position below it and shifted
       into place with v_alignbit;
     * per thread ONCE: the six directional differences p = tap - x, |p| and taps[k]*p
       (they do not depend on the threshold).  Per candidate the directional pass is
       three packed instructions per tap pair: `|p| < th` as min(1, th -sat |p|), then a
       multiply-add;
     * the directional output goes through LDS (two buffers alternate: one barrier per
       candidate) for the orthogonal pass, whose taps are whole rows above / below or a
       funnel shift of the thread's own registers with one dword from each neighbour;
     * direction search (luma): the eight directions are eight WAVEFRONTS, lane = block:
       a wave computes one direction's cost for all 64 blocks from 16-byte row reads -
       line sums with compile-time line indices, weighted by 840 / (samples on the line)
       - and the 64 block lanes then take the arg-max.
   Everything is the reference's integer arithmetic, casts included, for every int16
   input; a threshold outside [0, 32767] (never produced by the encoder) takes a
   per-sample path that spells the reference's int arithmetic out. */
#include "../../include/daala_hip.h"
#include <stdlib.h>
#include <string.h>
#include "od_common.cuh"

namespace {

constexpr int kBorder = 3;          /* OD_FILT_BORDER */
constexpr int kVeryLarge = 30000;   /* OD_DERING_VERY_LARGE */
constexpr int kPadL = 8;            /* tile column of the superblock's first sample */

/* OD_DIRECTION_OFFSETS_TABLE, src/dering.c:39-48, as (dy, dx) for k = 1..3. */
__constant__ signed char kDirStep[8][3][2] = {
  {{-1, 1}, {-2, 2}, {-3, 3}},
  {{0, 1}, {-1, 2}, {-1, 3}},
  {{0, 1}, {0, 2}, {0, 3}},
  {{0, 1}, {1, 2}, {1, 3}},
  {{1, 1}, {2, 2}, {3, 3}},
  {{1, 0}, {2, 1}, {3, 1}},
  {{1, 0}, {2, 0}, {3, 0}},
  {{1, 0}, {2, -1}, {3, -1}}
};

/* OD_THRESH_TABLE_Q8, src/dering.c:225-229. */
__constant__ short kThreshQ8[18] = {
  128, 134, 150, 168, 188, 210, 234, 262, 292, 327, 365, 408, 455, 509, 569, 635, 710, 768};

struct DeringArgs {
  const int16_t *x;
  int16_t *y;             /* [plane][cand][h][w] */
  int32_t *dirs;          /* [plane][nvsb*8][nhsb*8] */
  const uint8_t *bskip;   /* [plane][...][skip_stride] */
  const int32_t *thr;     /* [plane][cand][nvsb*nhsb] */
  long x_plane_stride;    /* samples */
  long bskip_plane_stride;
  int stride;             /* samples per row of x and y */
  int nhsb;
  int nvsb;
  int ncand;
  int pli;
  int skip_stride;
  int overlap;
  int coeff_shift;
  int vec;                /* x, y and the stride allow 16-byte row pieces */
  /* per-call mode: x is an (n + 6) x (n + 6) window with its border already in
     place (sentinels included), the superblock's frame position is given here */
  int window;
  int sbx;
  int sby;
  int fnhsb;
  int fnvsb;
};

/* ---- packed int16 pairs ------------------------------------------------------------ */
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 pk_u(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ s16x2 pk_s(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t pk_r(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_r(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_splat(int v) { return (uint32_t)(v & 0xffff)*0x10001u; }
/* wrap-around add / sub / multiply-add: the reference's (int16_t) casts */
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_r(pk_u(a) + pk_u(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_r(pk_u(a) - pk_u(b)); }
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
  return pk_r(pk_u(a)*pk_u(b) + pk_u(c));
}
__device__ __forceinline__ uint32_t pk_smax(uint32_t a, uint32_t b) {
  return pk_r(__builtin_elementwise_max(pk_s(a), pk_s(b)));
}
__device__ __forceinline__ uint32_t pk_umin(uint32_t a, uint32_t b) {
  return pk_r(__builtin_elementwise_min(pk_u(a), pk_u(b)));
}
__device__ __forceinline__ uint32_t pk_usub_sat(uint32_t a, uint32_t b) {
  return pk_r(__builtin_elementwise_sub_sat(pk_u(a), pk_u(b)));
}
__device__ __forceinline__ uint32_t pk_uadd_sat(uint32_t a, uint32_t b) {
  return pk_r(__builtin_elementwise_add_sat(pk_u(a), pk_u(b)));
}
__device__ __forceinline__ uint32_t pk_ssub_sat(uint32_t a, uint32_t b) {
  return pk_r(__builtin_elementwise_sub_sat(pk_s(a), pk_s(b)));
}
__device__ __forceinline__ uint32_t pk_sar(uint32_t a, int n) {
  const s16x2 sh = {(short)n, (short)n};
  return pk_r(pk_s(a) >> sh);
}
__device__ __forceinline__ uint32_t pk_shr(uint32_t a, int n) {
  const u16x2 sh = {(unsigned short)n, (unsigned short)n};
  return pk_r(pk_u(a) >> sh);
}
/* |p| of the int16 halves as UNSIGNED 16-bit values: abs((int)p), 32768 included */
__device__ __forceinline__ uint32_t pk_abs(uint32_t p) { return pk_smax(p, pk_sub(0, p)); }
/* 1 where the unsigned half of a is below the half of bound, else 0: `abs(p) < threshold`
   as min(1, bound -sat a).  `one` is 0x00010001 in a register the optimiser cannot see
   into (pk_opaque_one): knowing the constant it rewrites flag*value as a compare and a
   select PER HALF - two v_cmp, two v_cndmask and a v_perm for what is v_pk_sub_u16 clamp,
   v_pk_min_u16 and v_pk_mad_u16. */
__device__ __forceinline__ uint32_t pk_opaque_one() {
  uint32_t one = 0x00010001u;
  asm("" : "+v"(one));
  return one;
}
__device__ __forceinline__ uint32_t pk_below(uint32_t a, uint32_t bound, uint32_t one) {
  return pk_umin(pk_usub_sat(bound, a), one);
}

/* R packed registers holding the SEG = 2R samples that start at short index e of an LDS
   tile: whole dwords from the even position at or below e, funnel-shifted into place. */
template <int R>
__device__ __forceinline__ void window(uint32_t (&out)[R], const short *tile, int e) {
  const uint32_t *src = reinterpret_cast<const uint32_t *>(tile) + (e >> 1);
  uint32_t w[R + 1];
#pragma unroll
  for (int r = 0; r <= R; r++) w[r] = src[r];
  const int sh = (e & 1)*16;
#pragma unroll
  for (int r = 0; r < R; r++) out[r] = __builtin_amdgcn_alignbit(w[r + 1], w[r], sh);
}

/* ---- direction search ---------------------------------------------------------------
   od_dir_find8 (src/dering.c:61-124) scores direction D by SUM over its lines of
   (sum of the line's samples)^2 * 840/(samples on the line) - the variance of the
   block projected along D, up to a constant.  dline<D>(i, j) is the line sample (i, j)
   lies on; all indices are compile-time, so the line sums are registers. */
template <int D>
__host__ __device__ constexpr int dline(int i, int j) {
  return D == 0 ? i + j : D == 1 ? i + j/2 : D == 2 ? i : D == 3 ? 3 + i - j/2 : D == 4 ? 7 + i - j
   : D == 5 ? 3 - i/2 + j : D == 6 ? j : i/2 + j;
}

template <int D>
__host__ __device__ constexpr int dline_count(int l) {
  int n = 0;
  for (int i = 0; i < 8; i++) {
    for (int j = 0; j < 8; j++) n += dline<D>(i, j) == l;
  }
  return n;
}

template <int D>
__device__ __forceinline__ int dir_cost(const int (&v)[64]) {
  constexpr int NL = (D == 2 || D == 6) ? 8 : (D & 1) ? 11 : 15;
  int line[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) line[l] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) line[dline<D>(i, j)] += v[i*8 + j];
  }
  int cost = 0;
#pragma unroll
  for (int l = 0; l < NL; l++) cost += line[l]*line[l]*(840/dline_count<D>(l));
  return cost;
}

/* ---- the reference's arithmetic spelled out per sample: thresholds outside the packed
   path's range (src/dering.c:132-159, :172-208) --------------------------------------- */
template <int SEG>
__device__ __forceinline__ void slow_direction(uint32_t (&y)[SEG/2], const short *a, int e, int4 offs,
 int th) {
  const int off3[3] = {offs.x, offs.y, offs.z};
  short out[SEG];
#pragma unroll
  for (int t = 0; t < SEG; t++) {
    const short xx = a[e + t];
    short sum = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const short p0 = (short)(a[e + t + off3[k]] - xx);
      const short p1 = (short)(a[e + t - off3[k]] - xx);
      if (abs((int)p0) < th) sum = (short)(sum + (3 - k)*p0);
      if (abs((int)p1) < th) sum = (short)(sum + (3 - k)*p1);
    }
    out[t] = (short)(xx + ((sum + 8) >> 4));
  }
#pragma unroll
  for (int r = 0; r < SEG/2; r++) y[r] = (uint32_t)(unsigned short)out[2*r] | (uint32_t)out[2*r + 1] << 16;
}

template <int SEG>
__device__ __forceinline__ void slow_orthogonal(uint32_t (&y)[SEG/2], const short *a, const short *b, int e,
 int offset, int th) {
  short out[SEG];
#pragma unroll
  for (int t = 0; t < SEG; t++) {
    const short yy = b[e + t];
    const int tt = th/3 + abs((int)yy - (int)a[e + t]);
    const short athresh = (short)(th < tt ? th : tt);
    short sum = 0;
    short p = (short)(b[e + t + offset] - yy);
    if (abs((int)p) < athresh) sum = (short)(sum + p);
    p = (short)(b[e + t - offset] - yy);
    if (abs((int)p) < athresh) sum = (short)(sum + p);
    p = (short)(b[e + t + 2*offset] - yy);
    if (abs((int)p) < athresh) sum = (short)(sum + p);
    p = (short)(b[e + t - 2*offset] - yy);
    if (abs((int)p) < athresh) sum = (short)(sum + p);
    out[t] = (short)(yy + ((3*sum + 8) >> 4));
  }
#pragma unroll
  for (int r = 0; r < SEG/2; r++) y[r] = (uint32_t)(unsigned short)out[2*r] | (uint32_t)out[2*r + 1] << 16;
}

template <int XDEC>
__global__ __launch_bounds__(512 >> XDEC) void k_dering(DeringArgs a) {
  constexpr int N = 64 >> XDEC;           /* superblock side */
  constexpr int SEG = 8 >> XDEC;          /* samples per thread: one row of a filter block */
  constexpr int R = SEG/2;                /* packed registers per thread */
  constexpr int NT = 512 >> XDEC;         /* one thread per segment */
  constexpr int P = N + 24;               /* LDS pitch in shorts: 16-byte rows; 8 rows shift the banks by 32 */
  constexpr int ROWS = N + 2*kBorder;
  constexpr int NCH = (kPadL + N + kBorder + 7)/8;   /* 16-byte pieces of a tile row that hold samples */
  __shared__ __attribute__((aligned(16))) short A[ROWS*P];
  __shared__ __attribute__((aligned(16))) short B[2][ROWS*P];
  __shared__ int s_cost[8][64];
  __shared__ int s_dir[64];
  __shared__ int s_var[64];
  __shared__ unsigned char s_skip[64];
  __shared__ __attribute__((aligned(16))) int s_off[64][4];   /* tile offsets of the three directional taps; [3] = orthogonal step */
  const int tid = threadIdx.x;
  const int plane = blockIdx.z;
  const int sbx = a.window ? a.sbx : blockIdx.x;
  const int sby = a.window ? a.sby : blockIdx.y;
  const int nhsb = a.window ? a.fnhsb : a.nhsb;
  const int nvsb = a.window ? a.fnvsb : a.nvsb;
  const int16_t *xp = a.x + plane*a.x_plane_stride;
  /* ---- tile A: the superblock and its border (src/dering.c:270-279), copied to both B
     buffers (the directional pass overwrites their interior; the border stays x) */
  if (a.vec) {
    const int c_lo = sbx != 0 ? 0 : 1;                       /* piece 0 = the 8 samples left of the superblock */
    const int c_hi = sbx != nhsb - 1 ? NCH : NCH - 1;        /* the last piece = the samples right of it */
    for (int q = tid; q < ROWS*NCH; q += NT) {
      const int row = q/NCH;
      const int c = q - row*NCH;
      const int i = row - kBorder;
      const bool in_y = i >= -kBorder*(sby != 0) && i < N + kBorder*(sby != nvsb - 1);
      uint4 v = make_uint4(pk_splat(kVeryLarge), pk_splat(kVeryLarge), pk_splat(kVeryLarge),
       pk_splat(kVeryLarge));
      if (in_y && c >= c_lo && c < c_hi) {
        v = *reinterpret_cast<const uint4 *>(xp + (long)(sby*N + i)*a.stride + sbx*N + c*8 - kPadL);
      }
      *reinterpret_cast<uint4 *>(A + row*P + c*8) = v;
      *reinterpret_cast<uint4 *>(B[0] + row*P + c*8) = v;
      *reinterpret_cast<uint4 *>(B[1] + row*P + c*8) = v;
    }
  }
  else {
    for (int t = tid; t < ROWS*ROWS; t += NT) {
      const int i = t/ROWS - kBorder;
      const int j = t%ROWS - kBorder;
      short v = kVeryLarge;
      if (a.window) v = xp[(long)(i + kBorder)*a.stride + j + kBorder];
      else {
        const bool in_y = i >= -kBorder*(sby != 0) && i < N + kBorder*(sby != nvsb - 1);
        const bool in_x = j >= -kBorder*(sbx != 0) && j < N + kBorder*(sbx != nhsb - 1);
        if (in_y && in_x) v = xp[(long)(sby*N + i)*a.stride + sbx*N + j];
      }
      const int e = (i + kBorder)*P + j + kPadL;
      A[e] = v;
      B[0][e] = v;
      B[1][e] = v;
    }
  }
  __syncthreads();
  int32_t *dirs = a.dirs + (long)plane*a.nvsb*8*a.nhsb*8;
  const long dir_row = (long)a.nhsb*8;
  /* ---- direction search: wavefront = direction, lane = block */
  if (XDEC == 0 && a.pli == 0) {
    const int lane = tid & 63;
    const int by = lane >> 3;
    const int bx = lane & 7;
    int v[64];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint4 q = *reinterpret_cast<const uint4 *>(A + (by*8 + i + kBorder)*P + kPadL + bx*8);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int t = 0; t < 4; t++) {
        v[i*8 + 2*t] = (int)(short)(w[t] & 0xffff) >> a.coeff_shift;
        v[i*8 + 2*t + 1] = (int)w[t] >> 16 >> a.coeff_shift;
      }
    }
    int cost = 0;
    const int d = __builtin_amdgcn_readfirstlane(tid >> 6);
    switch (d) {
      case 0: cost = dir_cost<0>(v); break;
      case 1: cost = dir_cost<1>(v); break;
      case 2: cost = dir_cost<2>(v); break;
      case 3: cost = dir_cost<3>(v); break;
      case 4: cost = dir_cost<4>(v); break;
      case 5: cost = dir_cost<5>(v); break;
      case 6: cost = dir_cost<6>(v); break;
      default: cost = dir_cost<7>(v); break;
    }
    s_cost[d][lane] = cost;
    __syncthreads();
  }
  /* one thread per block: arg-max of the costs (luma) / stored direction (chroma), skip test */
  if (tid < 64) {
    const int by = tid >> 3;
    const int bx = tid & 7;
    int dir;
    int var = 0;
    const long dpos = a.window ? tid : ((long)sby*8 + by)*dir_row + sbx*8 + bx;
    if (a.pli == 0) {
      /* src/dering.c:110-123: the first strict maximum, against the cost of the
         direction at right angles to it */
      int best_cost = 0;
      dir = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int c = s_cost[i][tid];
        if (c > best_cost) {
          best_cost = c;
          dir = i;
        }
      }
      var = (best_cost - s_cost[(dir + 4) & 7][tid]) >> 10;
      dirs[dpos] = dir;
    }
    else dir = dirs[dpos];
    s_dir[tid] = dir;
    s_var[tid] = var;
    /* a __constant__ table indexed by a per-lane direction would be read with
       up to eight different addresses per wavefront: resolve it once per block */
#pragma unroll
    for (int k = 0; k < 3; k++) s_off[tid][k] = kDirStep[dir][k][0]*P + kDirStep[dir][k][1];
    s_off[tid][3] = dir > 0 && dir < 4 ? P : 1;
    /* src/dering.c:306-325 */
    int xstart = 0;
    int ystart = 0;
    int xend = 2 >> XDEC;
    int yend = 2 >> XDEC;
    if (a.overlap) {
      xstart -= sbx != 0;
      ystart -= sby != 0;
      xend += sbx != nhsb - 1;
      yend += sby != nvsb - 1;
    }
    const uint8_t *bs = a.bskip + plane*a.bskip_plane_stride;
    if (!a.window) bs += (long)(sby << (4 - XDEC))*a.skip_stride + (sbx << (4 - XDEC));
    int skip = 1;
    for (int i = ystart; i < yend; i++) {
      for (int j = xstart; j < xend; j++) {
        skip = skip && bs[(long)((by << 1 >> XDEC) + i)*a.skip_stride + (bx << 1 >> XDEC) + j];
      }
    }
    s_skip[tid] = (unsigned char)skip;
  }
  __syncthreads();
  /* ---- this thread's segment: row i, samples j0 .. j0 + SEG - 1, all in block blk */
  const int i = tid >> 3;
  const int seg = tid & 7;
  const int j0 = seg*SEG;
  const int blk = (i >> (3 - XDEC))*8 + seg;
  const int e0 = (i + kBorder)*P + kPadL + j0;       /* tile index of its first sample */
  const int4 offs = *reinterpret_cast<const int4 *>(s_off[blk]);
  const bool skip = s_skip[blk] != 0;
  int q8 = 256;
  if (a.pli == 0) {
    /* od_compute_thresh, src/dering.c:237-250 */
    int v1 = s_var[blk] >> 6;
    v1 = v1 < 32767 ? v1 : 32767;
    q8 = kThreshQ8[v1 ? 32 - __clz(v1) : 0];
  }
  uint32_t x[R];
  window<R>(x, A, e0);
  /* the threshold-independent half of the directional pass (src/dering.c:145-152):
     for the six taps the difference to x, its magnitude and taps[k] times it */
  uint32_t mag[6][R];
  uint32_t wp[6][R];
  {
    const int off3[3] = {offs.x, offs.y, offs.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
      for (int sgn = 0; sgn < 2; sgn++) {
        uint32_t nb[R];
        window<R>(nb, A, e0 + (sgn ? -off3[k] : off3[k]));
#pragma unroll
        for (int r = 0; r < R; r++) {
          const uint32_t p = pk_sub(nb[r], x[r]);
          mag[2*k + sgn][r] = pk_abs(p);
          wp[2*k + sgn][r] = pk_mad(p, pk_splat(3 - k), 0);
        }
      }
    }
  }
  const long nsb = (long)a.nhsb*a.nvsb;
  const uint32_t one = pk_opaque_one();
  for (int c = 0; c < a.ncand; c++) {
    const int threshold = a.window ? a.thr[c] : a.thr[((long)plane*a.ncand + c)*nsb + (long)sby*a.nhsb + sbx];
    int th = a.pli == 0 ? (threshold*q8 + 128) >> 8 : threshold;
    if (skip) th = 0;
    const bool packed = (unsigned)th <= 32767u;
    short *Bc = B[c & 1];
    /* directional pass */
    uint32_t y[R];
    if (packed) {
      const uint32_t thp = pk_splat(th);
#pragma unroll
      for (int r = 0; r < R; r++) {
        uint32_t sum = 0;
#pragma unroll
        for (int t = 0; t < 6; t++) sum = pk_mad(wp[t][r], pk_below(mag[t][r], thp, one), sum);
        /* (sum + 8) >> 4 without leaving 16 bits: ((sum >> 1) + 4) >> 3 */
        y[r] = pk_add(x[r], pk_sar(pk_add(pk_sar(sum, 1), pk_splat(4)), 3));
      }
    }
    else slow_direction<SEG>(y, A, e0, offs, th);
    if (R == 4) *reinterpret_cast<uint4 *>(Bc + e0) = make_uint4(y[0], y[1], y[2], y[R - 1]);
    else *reinterpret_cast<uint2 *>(Bc + e0) = make_uint2(y[0], y[1]);
    __syncthreads();
    /* orthogonal pass: the four taps at +-1, +-2 steps across the direction */
    uint32_t out[R];
    if (packed) {
      uint32_t tap[4][R];     /* +1, -1, +2, -2 */
      if (offs.w == 1) {
        const uint32_t left = *reinterpret_cast<const uint32_t *>(Bc + e0 - 2);
        const uint32_t right = *reinterpret_cast<const uint32_t *>(Bc + e0 + SEG);
#pragma unroll
        for (int r = 0; r < R; r++) {
          const uint32_t lo = r == 0 ? left : y[r - 1];
          const uint32_t hi = r == R - 1 ? right : y[r + 1];
          tap[0][r] = __builtin_amdgcn_alignbit(hi, y[r], 16);
          tap[1][r] = __builtin_amdgcn_alignbit(y[r], lo, 16);
          tap[2][r] = hi;
          tap[3][r] = lo;
        }
      }
      else {
        window<R>(tap[0], Bc, e0 + P);
        window<R>(tap[1], Bc, e0 - P);
        window<R>(tap[2], Bc, e0 + 2*P);
        window<R>(tap[3], Bc, e0 - 2*P);
      }
      const uint32_t thp = pk_splat(th);
      const uint32_t th3 = pk_splat(th/3);
#pragma unroll
      for (int r = 0; r < R; r++) {
        /* athresh = min(th, th/3 + |y - x|), src/dering.c:193-194: the saturated
           difference is exact wherever it can decide the minimum (th <= 32767) */
        const uint32_t d = pk_ssub_sat(y[r], x[r]);
        const uint32_t ad = pk_smax(d, pk_ssub_sat(0, d));
        const uint32_t ath = pk_umin(thp, pk_uadd_sat(th3, ad));
        uint32_t sum = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const uint32_t p = pk_sub(tap[t][r], y[r]);
          sum = pk_mad(p, pk_below(pk_abs(p), ath, one), sum);
        }
        /* (3*sum + 8) >> 4 = 3*(sum >> 4) + ((3*(sum & 15) + 8) >> 4) */
        const uint32_t lo = pk_shr(pk_mad(sum & 0x000f000fu, pk_splat(3), pk_splat(8)), 4);
        out[r] = pk_add(y[r], pk_mad(pk_sar(sum, 4), pk_splat(3), lo));
      }
    }
    else slow_orthogonal<SEG>(out, A, Bc, e0, offs.w, th);
    int16_t *yp = a.y + ((long)plane*a.ncand + c)*(a.window ? (long)N*N : a.x_plane_stride);
    int16_t *dst = a.window ? yp + i*N + j0 : yp + (long)(sby*N + i)*a.stride + sbx*N + j0;
    if (a.vec) {
      if (R == 4) *reinterpret_cast<uint4 *>(dst) = make_uint4(out[0], out[1], out[2], out[R - 1]);
      else *reinterpret_cast<uint2 *>(dst) = make_uint2(out[0], out[1]);
    }
    else {
#pragma unroll
      for (int r = 0; r < R; r++) {
        dst[2*r] = (int16_t)(out[r] & 0xffff);
        dst[2*r + 1] = (int16_t)(out[r] >> 16);
      }
    }
  }
}

}  // namespace

extern "C" int odhip_dering_planes(int16_t *d_y, const int16_t *d_x, int stride, int nhsb, int nvsb,
 int xdec, int nplanes, int32_t *d_dirs, int pli, const uint8_t *d_bskip, int skip_stride,
 long bskip_plane_stride, const int32_t *d_thresholds, int ncand, int overlap, int coeff_shift,
 odhip_stream stream) {
  if (nplanes == 0 || ncand == 0) return ODHIP_SUCCESS;
  if (!d_y || !d_x || !d_dirs || !d_bskip || !d_thresholds || nhsb <= 0 || nvsb <= 0 || nplanes < 0
   || ncand < 0 || (xdec != 0 && xdec != 1) || stride < (nhsb*64 >> xdec) || coeff_shift < 0
   || coeff_shift > 15 || (pli == 0 && xdec != 0)) {   /* the direction search is 8x8 on luma */
    return ODHIP_EINVAL;
  }
  DeringArgs a;
  memset(&a, 0, sizeof(a));
  a.x = d_x;
  a.y = d_y;
  a.dirs = d_dirs;
  a.bskip = d_bskip;
  a.thr = d_thresholds;
  a.x_plane_stride = (long)stride*(nvsb*64 >> xdec);
  a.bskip_plane_stride = bskip_plane_stride;
  a.stride = stride;
  a.nhsb = nhsb;
  a.nvsb = nvsb;
  a.ncand = ncand;
  a.pli = pli;
  a.skip_stride = skip_stride;
  a.overlap = overlap;
  a.coeff_shift = coeff_shift;
  /* 16-byte row pieces need aligned planes and an 8-sample stride */
  a.vec = stride%8 == 0 && ((uintptr_t)d_x | (uintptr_t)d_y)%16 == 0;
  const dim3 grid(nhsb, nvsb, nplanes);
  if (xdec) k_dering<1><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else k_dering<0><<<grid, 512, 0, (hipStream_t)stream>>>(a);
  return odhip_check_launch();
}

/* Per-call surface with od_dering's argument list (src/dering.h:64-69) minus
   the function table: host pointers, synchronous. */
extern "C" void od_dering_hip(int16_t *y, int ystride, const int16_t *x, int xstride, int nhb, int nvb,
 int sbx, int sby, int nhsb, int nvsb, int xdec, int dir[8][8], int pli, unsigned char *bskip,
 int skip_stride, int threshold, int overlap, int coeff_shift) {
  const int n = 64 >> xdec;
  const int wn = n + 2*kBorder;
  if (nhb != 8 || nvb != 8 || (xdec != 0 && xdec != 1) || (pli == 0 && xdec != 0)) {
    fprintf(stderr, "libdaalahip: od_dering_hip supports full superblocks only (nhb = nvb = 8)\n");
    abort();
  }
  /* marshal the window the reference builds in `inbuf` (data movement only) */
  int16_t win[(64 + 6)*(64 + 6)];
  for (int i = -kBorder; i < n + kBorder; i++) {
    for (int j = -kBorder; j < n + kBorder; j++) {
      const bool in_y = i >= -kBorder*(sby != 0) && i < n + kBorder*(sby != nvsb - 1);
      const bool in_x = j >= -kBorder*(sbx != 0) && j < n + kBorder*(sbx != nhsb - 1);
      win[(i + kBorder)*wn + j + kBorder] = in_y && in_x ? x[(long)i*xstride + j] : (int16_t)kVeryLarge;
    }
  }
  /* the skip flags the test can touch: rows/cols -1 .. 16 >> xdec of the map */
  const int sn = (16 >> xdec) + 2;
  unsigned char skipwin[18*18];
  memset(skipwin, 1, sizeof(skipwin));
  for (int i = -(sby != 0); i < (16 >> xdec) + (sby != nvsb - 1); i++) {
    for (int j = -(sbx != 0); j < (16 >> xdec) + (sbx != nhsb - 1); j++) {
      skipwin[(i + 1)*sn + j + 1] = bskip[(long)i*skip_stride + j];
    }
  }
  int16_t *d_win;
  int16_t *d_y;
  int32_t *d_dir;
  uint8_t *d_skip;
  int32_t *d_thr;
  const size_t wbytes = (size_t)wn*wn*sizeof(int16_t);
  const size_t ybytes = (size_t)n*n*sizeof(int16_t);
  char *buf;
  if (hipMalloc((void **)&buf, wbytes + ybytes + 64*4 + sizeof(skipwin) + 16 + 64) != hipSuccess) {
    fprintf(stderr, "libdaalahip: od_dering_hip: hipMalloc failed (no CPU fallback)\n");
    abort();
  }
  size_t o = 0;
  d_win = (int16_t *)(buf + o);
  o += (wbytes + 15) & ~(size_t)15;
  d_y = (int16_t *)(buf + o);
  o += (ybytes + 15) & ~(size_t)15;
  d_dir = (int32_t *)(buf + o);
  o += 64*4;
  d_thr = (int32_t *)(buf + o);
  o += 16;
  d_skip = (uint8_t *)(buf + o);
  int dflat[64];
  for (int i = 0; i < 64; i++) dflat[i] = dir[i >> 3][i & 7];
  bool ok = hipMemcpy(d_win, win, wbytes, hipMemcpyHostToDevice) == hipSuccess
   && hipMemcpy(d_dir, dflat, sizeof(dflat), hipMemcpyHostToDevice) == hipSuccess
   && hipMemcpy(d_thr, &threshold, sizeof(int), hipMemcpyHostToDevice) == hipSuccess
   && hipMemcpy(d_skip, skipwin, sizeof(skipwin), hipMemcpyHostToDevice) == hipSuccess;
  DeringArgs a;
  memset(&a, 0, sizeof(a));
  a.x = d_win;
  a.y = d_y;
  a.dirs = d_dir;
  a.bskip = d_skip + sn + 1;   /* entry (0, 0) of the superblock's skip map */
  a.thr = d_thr;
  a.stride = wn;
  a.nhsb = 1;
  a.nvsb = 1;
  a.ncand = 1;
  a.pli = pli;
  a.skip_stride = sn;
  a.overlap = overlap;
  a.coeff_shift = coeff_shift;
  a.window = 1;
  a.sbx = sbx;
  a.sby = sby;
  a.fnhsb = nhsb;
  a.fnvsb = nvsb;
  if (ok) {
    if (xdec) k_dering<1><<<dim3(1, 1, 1), 256>>>(a);
    else k_dering<0><<<dim3(1, 1, 1), 512>>>(a);
    int16_t ytmp[64*64];
    ok = hipGetLastError() == hipSuccess
     && hipMemcpy(ytmp, d_y, ybytes, hipMemcpyDeviceToHost) == hipSuccess
     && hipMemcpy(dflat, d_dir, sizeof(dflat), hipMemcpyDeviceToHost) == hipSuccess;
    if (ok) {
      for (int i = 0; i < n; i++) memcpy(y + (long)i*ystride, ytmp + i*n, n*sizeof(int16_t));
      for (int i = 0; i < 64; i++) dir[i >> 3][i & 7] = dflat[i];
    }
  }
  (void)hipFree(buf);
  if (!ok) {
    fprintf(stderr, "libdaalahip: od_dering_hip failed on the device (no CPU fallback)\n");
    abort();
  }
}
