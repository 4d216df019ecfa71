/* lapped_kernels.hip - the fused "filter + DCT" stage for gfx950.

   FORWARD (k_forward_pyramid): one workgroup per superblock (64x64 luma /
   32x32 4:2:0 chroma).  The superblock's pixels are read once from HBM
   (coalesced 4-pixel loads plus a 2-sample halo), converted to coefficients,
   lapped across the superblock edges, and then kept in LDS while EVERY level
   of the block-size pyramid is produced from it:

       level bs = top..0:   2-D fDCT of every (4<<bs)-block  -> HBM level plane
                            4-point split pre-filter of every block (in LDS)

   so the algorithmic HBM traffic is 1 B read + 4 B written per pixel and level
   (21 B / luma pixel for five levels) with no
   re-reads.  Restates, level by level, the recursion of od_compute_dcts
   (reference src/encode.c:1455-1512) over od_ref_plane_to_coeff
   (src/state.c:1216-1277), od_apply_prefilter_frame_sbs (src/filter.c:
   1529-1559), od_prefilter_split (:1459-1483) and fdct_2d[bs].

   INVERSE: k_inverse_sb (iDCT + split post-filters, src/encode.c:1780-1789,
   src/filter.c:1485-1527) and k_edge_rows/k_edge_cols (superblock-edge post-filter
   src/filter.c:1589-1618 + od_coeff_to_ref_plane src/state.c:1281-1345).

   Pixels are bounded, so every lifting multiply uses the full-rate 24-bit
   multiplier (OdMul24, see od_lift.cuh). */
#include "../../include/daala_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "od_ctx.cuh"
#include "od_tile.cuh"
#include "od_pvq_math.cuh"
#include "gen/od_scan_tables.h"

namespace {

/* Workgroup barrier for LDS-only hand-offs.  __syncthreads() is a full
   workgroup-scope release/acquire: it also drains vmcnt, i.e. every wave would
   wait at each level boundary until its coefficient stores have reached memory.
   All cross-wave communication in these kernels goes through LDS, so waiting
   for the LDS queue (lgkmcnt) alone is sufficient and lets the 16-byte stores
   of one level drain underneath the arithmetic of the next. */
__device__ __forceinline__ void od_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* 16-byte store of coefficients to a level plane.  OD_PYR_STORE_NT (experiment): non-temporal. */
typedef int od_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void od_store_coef4(od_coeff *p, int4 v) {
#ifdef OD_PYR_STORE_NT
  od_v4i w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<od_v4i *>(p));
#else
  *reinterpret_cast<int4 *>(p) = v;
#endif
}

template <int TILE>
struct Geo {
#ifndef OD_PYR_NT64
#define OD_PYR_NT64 256
#endif
  static constexpr int kNT = TILE == 64 ? OD_PYR_NT64 : TILE*TILE/16;
  static constexpr int kPitch = TILE + 4;
  /* Tile with a 2-sample halo: rows/cols -2,-1 live at TILE, TILE+1 and rows/
     cols TILE, TILE+1 at TILE+2, TILE+3 (the halo columns occupy the pitch
     padding, the halo rows four extra rows), so the interior keeps its 16-byte
     row alignment. */
  static constexpr int kHaloWords = (TILE + 4)*kPitch;
  __device__ static __forceinline__ int map(int i) {
    return i < 0 ? TILE + 2 + i : (i >= TILE ? i + 2 : i);
  }
};

struct PyramidArgs {
  od_coeff *levels[ODHIP_NBSIZES];
  const uint8_t *px;
  int px_stride;
  long px_plane_stride;
  int w;
  int h;
  int pic_w;
  int pic_h;
  /* full-precision references (odhip_ctx_set_fpr): px holds int16 samples at 12 bits,
     strides in samples (the reference's xstride 2, src/state.c:1238-1254) */
  int px16;
};

/* The plane's first sample. */
__device__ __forceinline__ const uint8_t *pyr_plane(const PyramidArgs &a, int plane) {
  return a.px + ((long)plane*a.px_plane_stride << a.px16);
}

/* 4-point filter across 4 LDS words a, a+step, a+2*step, a+3*step. */
template <bool INV, typename E>
__device__ __forceinline__ void lds_filter4(E *p, int step) {
  int t0 = p[0];
  int t1 = p[step];
  int t2 = p[2*step];
  int t3 = p[3*step];
  if constexpr (INV) od_post_filter4_dev(t0, t1, t2, t3);
  else od_pre_filter4_dev(t0, t1, t2, t3);
  p[0] = (E)t0;
  p[step] = (E)t1;
  p[2*step] = (E)t2;
  p[3*step] = (E)t3;
}

/* Column-direction half of od_prefilter_split / od_postfilter_split for every
   level-LN block of the tile: taps across the horizontal mid-line, gated by
   `hfilter` (derived from the block's x index, src/encode.c:1487). */
template <int TILE, int LN, bool INV, int NT = Geo<TILE>::kNT, typename E>
__device__ __forceinline__ void split_filter_cols(E *t, int tid, int x0, int pic_w) {
  constexpr int N = 4 << LN;
  constexpr int P = Geo<TILE>::kPitch;
  for (int k = tid; k < TILE*(TILE/N); k += NT) {
    const int x = k % TILE;
    const int by = k / TILE;
    const int gbx = (x0 + x)/N;
    if ((gbx + 1)*N <= pic_w) lds_filter4<INV>(t + (by*N + N/2 - 2)*P + x, P);
  }
}

/* Row-direction half: taps across the vertical mid-line, gated by `vfilter`
   (from the block's y index, src/encode.c:1488). */
template <int TILE, int LN, bool INV, int NT = Geo<TILE>::kNT, typename E>
__device__ __forceinline__ void split_filter_rows(E *t, int tid, int y0, int pic_h) {
  constexpr int N = 4 << LN;
  constexpr int P = Geo<TILE>::kPitch;
  for (int k = tid; k < TILE*(TILE/N); k += NT) {
    const int y = k % TILE;
    const int bx = k / TILE;
    const int gby = (y0 + y)/N;
    if ((gby + 1)*N <= pic_h) lds_filter4<INV>(t + y*P + bx*N + N/2 - 2, 1);
  }
}

/* Tile -> raster plane.  Every lane's LDS reads are issued before its first store: one exposed
   LDS latency per tile instead of one per 16 bytes (the rolled loop was a chain of
   ds_read -> s_waitcnt -> global_store, ~150 cycles each, 160 times per wave and superblock pair). */
template <int TILE, int NT = Geo<TILE>::kNT>
__device__ __forceinline__ void store_tile(od_coeff *plane, int w, int x0, int y0,
 const int *z, int tid) {
  constexpr int P = Geo<TILE>::kPitch;
  if constexpr ((TILE*TILE/4) % NT == 0) {
    constexpr int K = TILE*TILE/4/NT;
    int4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = tid + k*NT;
      v[k] = *reinterpret_cast<const int4 *>(z + (i/(TILE/4))*P + (i % (TILE/4))*4);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = tid + k*NT;
      od_store_coef4(plane + (long)(y0 + i/(TILE/4))*w + x0 + (i % (TILE/4))*4, v[k]);
    }
  }
  else {
    for (int i = tid; i < TILE*TILE/4; i += NT) {
      const int y = i/(TILE/4);
      const int x = (i % (TILE/4))*4;
      od_store_coef4(plane + (long)(y0 + y)*w + x0 + x, *reinterpret_cast<const int4 *>(z + y*P + x));
    }
  }
}

/* 64- and 32-point levels of a luma superblock: only 64 / 128 columns exist for
   256 threads, so each column (and then each row) is shared by TWO lanes
   running the even and the odd half network (od_fdct_lift_half), the halves
   assigned wave-uniformly.  The intermediate is kept TRANSPOSED and in
   parity-split order (position p' = (p & 1)*N/2 + (p >> 1)) at an odd pitch of
   65 words, which makes every access of both passes and of the copy-out a
   conflict-free ds_read/write_b32:
     pass 1  lane (x, h):       reads t[.., x] down the column, writes
                                zt[x][by*N + v'], v' in half h
     pass 2  lane (c, bx, h):   reads zt[bx*N + i][c], i = 0..N-1, then (after a
                                barrier: its partner reads the same words) writes
                                zt[bx*N + u'][c], u' in half h
     store   raster (y, x)  <-  zt[bx*N + u'(x)][by*N + v'(y)]                  */
template <int LN, typename T = OdMul24, int NT = Geo<64>::kNT>
__device__ __forceinline__ void pyramid_level_split64(short *t, int *z, const PyramidArgs &a,
 long plane_off, int x0, int y0, int tid) {
  constexpr int TILE = 64;
  constexpr int N = 4 << LN;
  constexpr int H = N/2;
  constexpr int PT = Geo<TILE>::kPitch;
  constexpr int PZ = 65;
  constexpr int kTasks = TILE*(TILE/N);
  static_assert(kTasks % 64 == 0, "wave-uniform halves");
  const bool act = tid < 2*kTasks;
  const int half = tid/kTasks;
  const int tt = tid - half*kTasks;
  const int c = tt % TILE;
  const int bq = tt/TILE;
  if (act) {
    T in[N];
    T out[H];
#pragma unroll
    for (int r = 0; r < N; r++) in[r] = T(t[(bq*N + r)*PT + c]);
    if (half == 0) od_fdct_lift_half<LN, 0>(out, in);
    else od_fdct_lift_half<LN, 1>(out, in);
#pragma unroll
    for (int k = 0; k < H; k++) z[c*PZ + bq*N + half*H + k] = out[k];
  }
  od_lds_barrier();
  /* The tile is free now: lapping of the next level overlaps the row pass. */
  split_filter_cols<TILE, LN, false, NT>(t, tid, x0, a.pic_w);
  {
    T in[N];
    T out[H];
    if (act) {
#pragma unroll
      for (int i = 0; i < N; i++) in[i] = T(z[(bq*N + i)*PZ + c]);
    }
    od_lds_barrier();
    if (act) {
      if (half == 0) od_fdct_lift_half<LN, 0>(out, in);
      else od_fdct_lift_half<LN, 1>(out, in);
#pragma unroll
      for (int k = 0; k < H; k++) z[(bq*N + half*H + k)*PZ + c] = out[k];
    }
  }
  od_lds_barrier();
  if (a.levels[LN]) {
    od_coeff *plane = a.levels[LN] + plane_off;
    static_assert((TILE*TILE/4) % NT == 0, "whole trips");
    constexpr int K = TILE*TILE/4/NT;
    int o[K][4];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = tid + k*NT;
      const int y = i/(TILE/4);
      const int x = (i % (TILE/4))*4;
      const int v = y & (N - 1);
      const int col = (y - v) + (v & 1)*H + (v >> 1);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int u = (x + j) & (N - 1);
        o[k][j] = z[((x + j - u) + (u & 1)*H + (u >> 1))*PZ + col];
      }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = tid + k*NT;
      od_store_coef4(plane + (long)(y0 + i/(TILE/4))*a.w + x0 + (i % (TILE/4))*4, make_int4(o[k][0], o[k][1], o[k][2], o[k][3]));
    }
  }
  split_filter_rows<TILE, LN, false, NT>(t, tid, y0, a.pic_h);
  od_lds_barrier();
}

/* 4x4 level: one block per lane, both passes in registers, rows read from the
   tile and written to HBM as 16-byte vectors (16 consecutive lanes = one
   256-byte row segment). */
template <int TILE, typename T = OdMul24, int NT = Geo<TILE>::kNT>
__device__ __forceinline__ void pyramid_level4(const short *t, const PyramidArgs &a,
 long plane_off, int x0, int y0, int tid) {
  constexpr int P = Geo<TILE>::kPitch;
  constexpr int NB = TILE/4;
  for (int blk = tid; blk < NB*NB; blk += NT) {
  const int bx = blk % NB;
  const int by = blk/NB;
  T m[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const short4 v = *reinterpret_cast<const short4 *>(t + (by*4 + r)*P + bx*4);
    m[r][0] = T(v.x);
    m[r][1] = T(v.y);
    m[r][2] = T(v.z);
    m[r][3] = T(v.w);
  }
  /* columns, then rows (od_bin_fdct4x4, src/dct.c:151-156) */
#pragma unroll
  for (int c = 0; c < 4; c++) {
    T in[4] = {m[0][c], m[1][c], m[2][c], m[3][c]};
    T out[4];
    od_fdct4_lift(out, in);
    m[0][c] = out[0];
    m[1][c] = out[1];
    m[2][c] = out[2];
    m[3][c] = out[3];
  }
  if (!a.levels[0]) continue;
  od_coeff *plane = a.levels[0] + plane_off;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    T out[4];
    od_fdct4_lift(out, m[r]);
    od_store_coef4(plane + (long)(y0 + by*4 + r)*a.w + x0 + bx*4, make_int4(out[0], out[1], out[2], out[3]));
  }
  }
}

template <int TILE, int LN, typename T = OdMul24, int NT = Geo<TILE>::kNT>
__device__ __forceinline__ void pyramid_level(short *t, int *z, const PyramidArgs &a,
 long plane_off, int x0, int y0, int tid) {
  if constexpr (LN == 0) {
    pyramid_level4<TILE, T, NT>(t, a, plane_off, x0, y0, tid);
  }
  else if constexpr (TILE == 64 && ((LN >= 3 && NT == 256) || (LN == 4 && NT == 128))) {
    /* half networks: 2*TILE*(TILE/N) lanes = 256 for the 32-point level, 128 for the
       64-point level */
    pyramid_level_split64<LN, T, NT>(t, z, a, plane_off, x0, y0, tid);
    pyramid_level<TILE, LN - 1, T, NT>(t, z, a, plane_off, x0, y0, tid);
  }
  else {
    od_tile_cols<TILE, LN, false, T, NT>(z, t, tid, OdAllBlocks());
    od_lds_barrier();
    od_tile_rows<TILE, LN, false, T, NT>(z, z, tid, OdAllBlocks());
    split_filter_cols<TILE, LN, false, NT>(t, tid, x0, a.pic_w);
    od_lds_barrier();
    if (a.levels[LN]) store_tile<TILE, NT>(a.levels[LN] + plane_off, a.w, x0, y0, z, tid);
    split_filter_rows<TILE, LN, false, NT>(t, tid, y0, a.pic_h);
    od_lds_barrier();
    pyramid_level<TILE, LN - 1, T, NT>(t, z, a, plane_off, x0, y0, tid);
  }
}

/* The three steps that bring a superblock tile (with its 2-sample halo) into
   LDS and lap it across the superblock edges; a barrier separates them. */
template <int TILE, int NT>
__device__ __forceinline__ void sb_halo_pos(int i, int &r, int &c) {
  if (i < 4*(TILE + 4)) {
    const int k = i/(TILE + 4);
    r = k < 2 ? k - 2 : TILE + k - 2;
    c = i % (TILE + 4) - 2;
  }
  else {
    const int j = i - 4*(TILE + 4);
    const int k = j & 3;
    r = j >> 2;
    c = k < 2 ? k - 2 : TILE + k - 2;
  }
}

/* One lane's share of an 8-bit superblock tile and its halo ring (2 samples of each
   neighbouring superblock, where it exists).  issue() puts every global load in flight,
   commit() converts (od_ref_buf_to_coeff, src/state.c:1231-1237: (p - 128) << OD_COEFF_SHIFT)
   and writes the tile: a workgroup pays ONE memory latency for its pictures, not one per
   256 samples (the rolled loops were load -> s_waitcnt vmcnt(0) -> ds_write, 7 times per tile). */
template <int TILE, int NT>
struct SbFetch {
  using G = Geo<TILE>;
  static_assert((TILE*TILE/4) % NT == 0, "whole trips");
  static constexpr int kMain = TILE*TILE/4/NT;
  static constexpr int kHalo = (8*TILE + 16 + NT - 1)/NT;
  uint32_t m[kMain];
  int hv[kHalo];
  __device__ __forceinline__ void issue(const PyramidArgs &a, const uint8_t *px, int x0, int y0, int tid) {
#pragma unroll
    for (int k = 0; k < kMain; k++) {
      const int i = tid + k*NT;
      m[k] = *reinterpret_cast<const uint32_t *>(px + (long)(y0 + i/(TILE/4))*a.px_stride + x0
       + (i % (TILE/4))*4);
    }
#pragma unroll
    for (int k = 0; k < kHalo; k++) {
      const int i = tid + k*NT;
      int r;
      int c;
      sb_halo_pos<TILE, NT>(i, r, c);
      const int gx = x0 + c;
      const int gy = y0 + r;
      hv[k] = -1;
      if (i < 8*TILE + 16 && gx >= 0 && gx < a.w && gy >= 0 && gy < a.h) hv[k] = px[(long)gy*a.px_stride + gx];
    }
  }
  __device__ __forceinline__ void commit(short *t, int tid) const {
    constexpr int P = G::kPitch;
#pragma unroll
    for (int k = 0; k < kMain; k++) {
      const int i = tid + k*NT;
      const uint32_t v = m[k];
      *reinterpret_cast<short4 *>(t + (i/(TILE/4))*P + (i % (TILE/4))*4) =
       make_short4(((int)(v & 255) - 128)*16, ((int)((v >> 8) & 255) - 128)*16,
       ((int)((v >> 16) & 255) - 128)*16, ((int)(v >> 24) - 128)*16);
    }
#pragma unroll
    for (int k = 0; k < kHalo; k++) {
      int r;
      int c;
      sb_halo_pos<TILE, NT>(tid + k*NT, r, c);
      if (hv[k] >= 0) t[G::map(r)*P + G::map(c)] = (short)((hv[k] - 128)*16);
    }
  }
};

/* Full-precision references (src/state.c:1245-1250: p - 2048): int16 samples, the plain loops. */
template <int TILE, int NT>
__device__ __forceinline__ void sb_load16(short *t, const PyramidArgs &a, const uint8_t *px,
 int x0, int y0, int tid) {
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  const short *px_w = reinterpret_cast<const short *>(px);
  for (int i = tid; i < TILE*TILE/4; i += NT) {
    const int y = i/(TILE/4);
    const int x = (i % (TILE/4))*4;
    const short4 v = *reinterpret_cast<const short4 *>(px_w + (long)(y0 + y)*a.px_stride + x0 + x);
    *reinterpret_cast<short4 *>(t + y*P + x) = make_short4(v.x - 2048, v.y - 2048, v.z - 2048, v.w - 2048);
  }
  for (int i = tid; i < 8*TILE + 16; i += NT) {
    int r;
    int c;
    sb_halo_pos<TILE, NT>(i, r, c);
    const int gx = x0 + c;
    const int gy = y0 + r;
    if (gx >= 0 && gx < a.w && gy >= 0 && gy < a.h) {
      t[G::map(r)*P + G::map(c)] = (short)(px_w[(long)gy*a.px_stride + gx] - 2048);
    }
  }
}

/* NS horizontally adjacent tiles starting at x0. */
template <int TILE, int NT, int NS = 1>
__device__ __forceinline__ void sb_load(short (*t)[Geo<TILE>::kHaloWords], const PyramidArgs &a,
 const uint8_t *px, int x0, int y0, int tid) {
  if (a.px16) {
    for (int s = 0; s < NS; s++) sb_load16<TILE, NT>(t[s], a, px, x0 + s*TILE, y0, tid);
  }
  else {
    SbFetch<TILE, NT> f[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) f[s].issue(a, px, x0 + s*TILE, y0, tid);
#pragma unroll
    for (int s = 0; s < NS; s++) f[s].commit(t[s], tid);
  }
}

template <int TILE, int NT>
__device__ __forceinline__ void sb_edge_cols(short *t, const PyramidArgs &a, int x0, int y0,
 int tid) {
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  const int w = a.w;
  const int h = a.h;
  /* od_apply_prefilter_frame_sbs, src/filter.c:1540-1550: column taps across
     every interior horizontal superblock edge, for every column (halo columns
     included: the row taps below read them at the edge crossings). */
  for (int i = tid; i < 2*(TILE + 4); i += NT) {
    const int bottom = i/(TILE + 4);
    const int c = i % (TILE + 4) - 2;
    const int gx = x0 + c;
    const bool edge = bottom ? y0 + TILE < h : y0 > 0;
    if (edge && gx >= 0 && gx < w) {
      const int r = bottom ? TILE - 2 : -2;
      const int col = G::map(c);
      int t0 = t[G::map(r)*P + col];
      int t1 = t[G::map(r + 1)*P + col];
      int t2 = t[G::map(r + 2)*P + col];
      int t3 = t[G::map(r + 3)*P + col];
      od_pre_filter4_dev(t0, t1, t2, t3);
      t[G::map(r)*P + col] = (short)t0;
      t[G::map(r + 1)*P + col] = (short)t1;
      t[G::map(r + 2)*P + col] = (short)t2;
      t[G::map(r + 3)*P + col] = (short)t3;
    }
  }
}

template <int TILE, int NT>
__device__ __forceinline__ void sb_edge_rows(short *t, const PyramidArgs &a, int x0, int tid) {
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  const int w = a.w;
  /* ... then row taps across every interior vertical edge, src/filter.c:
     1551-1557. */
  for (int i = tid; i < 2*TILE; i += NT) {
    const int right = i/TILE;
    const int r = i % TILE;
    const bool edge = right ? x0 + TILE < w : x0 > 0;
    if (edge) {
      const int c = right ? TILE - 2 : -2;
      short *row = t + r*P;
      int t0 = row[G::map(c)];
      int t1 = row[G::map(c + 1)];
      int t2 = row[G::map(c + 2)];
      int t3 = row[G::map(c + 3)];
      od_pre_filter4_dev(t0, t1, t2, t3);
      row[G::map(c)] = (short)t0;
      row[G::map(c + 1)] = (short)t1;
      row[G::map(c + 2)] = (short)t2;
      row[G::map(c + 3)] = (short)t3;
    }
  }
}

template <int TILE, typename T = OdMul24, int NT = Geo<TILE>::kNT>
__global__ __launch_bounds__(NT) void k_forward_pyramid(PyramidArgs a) {
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  constexpr int TOP = TILE == 64 ? 4 : 3;
  /* Source tile as int16: (p - 128) << 4 lapped at most once per direction
     stays below 2^13 * 1.78^2 < 2^15 (every sample lies in the support of
     exactly one vertical and one horizontal 4-tap filter of the whole pyramid),
     so the narrow type is exact and halves the tile's LDS footprint. */
  __shared__ __attribute__((aligned(16))) short t[G::kHaloWords];
  __shared__ __attribute__((aligned(16))) int z[TILE*P];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x*TILE;
  const int y0 = blockIdx.y*TILE;
  const uint8_t *px = pyr_plane(a, blockIdx.z);
  const long plane_off = (long)blockIdx.z*a.w*a.h;
  sb_load<TILE, NT>(&t, a, px, x0, y0, tid);
  od_lds_barrier();
  sb_edge_cols<TILE, NT>(t, a, x0, y0, tid);
  od_lds_barrier();
  sb_edge_rows<TILE, NT>(t, a, x0, tid);
  od_lds_barrier();
  pyramid_level<TILE, TOP, T, NT>(t, z, a, plane_off, x0, y0, tid);
}

/* TWO horizontally adjacent luma superblocks per 256-thread workgroup.  With
   one superblock only 128 (64-point level, split) of the 256 lanes have work
   in the most expensive phases; with two, every phase of every level keeps all
   four waves busy and one barrier covers twice the arithmetic:
     64-point level: 2 tiles x 64 columns x 2 half networks = 256 lanes
     32-point level: 2 tiles x 128 columns (full network)    = 256 lanes
     16/8-point:     generic passes, tile after tile
     4x4:            one block per lane, tile after tile                       */
template <typename T>
__global__ __launch_bounds__(256) void k_forward_pyramid64x2(PyramidArgs a) {
  constexpr int TILE = 64;
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  constexpr int NT = 256;
  constexpr int PZ = 65;
  static_assert(G::kNT == NT, "kernel is written for 256-thread workgroups");
  __shared__ __attribute__((aligned(16))) short t[2][G::kHaloWords];
  __shared__ __attribute__((aligned(16))) int z[2][TILE*P];
  const int tid = threadIdx.x;
  const int xb = blockIdx.x*2*TILE;
  const int y0 = blockIdx.y*TILE;
  const uint8_t *px = pyr_plane(a, blockIdx.z);
  const long plane_off = (long)blockIdx.z*a.w*a.h;
  sb_load<TILE, NT, 2>(t, a, px, xb, y0, tid);
  od_lds_barrier();
  for (int s = 0; s < 2; s++) sb_edge_cols<TILE, NT>(t[s], a, xb + s*TILE, y0, tid);
  od_lds_barrier();
  for (int s = 0; s < 2; s++) sb_edge_rows<TILE, NT>(t[s], a, xb + s*TILE, tid);
  od_lds_barrier();
  const int s2 = tid >> 7;        /* tile owned in the 64- and 32-point phases */
  const int lt = tid & 127;
  /* ---- 64-point level (see pyramid_level_split64 for the layout) ---------- */
  {
    constexpr int N = 64;
    constexpr int H = 32;
    const int half = lt >> 6;
    const int c = lt & 63;
    short *ts = t[s2];
    int *zs = z[s2];
    {
      T in[N];
      T out[H];
#pragma unroll
      for (int r = 0; r < N; r++) in[r] = T(ts[r*P + c]);
      if (half == 0) od_fdct_lift_half<4, 0>(out, in);
      else od_fdct_lift_half<4, 1>(out, in);
#pragma unroll
      for (int k = 0; k < H; k++) zs[c*PZ + half*H + k] = out[k];
    }
    od_lds_barrier();
    for (int s = 0; s < 2; s++) split_filter_cols<TILE, 4, false>(t[s], tid, xb + s*TILE, a.pic_w);
    {
      T in[N];
      T out[H];
#pragma unroll
      for (int i = 0; i < N; i++) in[i] = T(zs[i*PZ + c]);
      od_lds_barrier();
      if (half == 0) od_fdct_lift_half<4, 0>(out, in);
      else od_fdct_lift_half<4, 1>(out, in);
#pragma unroll
      for (int k = 0; k < H; k++) zs[(half*H + k)*PZ + c] = out[k];
    }
    od_lds_barrier();
    if (a.levels[4]) {
      /* copy-out: every lane reads its 8 x 16 bytes of both tiles, then stores them */
      od_coeff *plane = a.levels[4] + plane_off;
      constexpr int K = 2*TILE*TILE/4/NT;
      int o[K][4];
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int i = tid + k*NT;
        const int s = i/(TILE*TILE/4);
        const int j = i - s*(TILE*TILE/4);
        const int y = j/(TILE/4);
        const int x = (j % (TILE/4))*4;
        const int col = (y & 1)*H + (y >> 1);
#pragma unroll
        for (int q = 0; q < 4; q++) o[k][q] = z[s][(((x + q) & 1)*H + ((x + q) >> 1))*PZ + col];
      }
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int i = tid + k*NT;
        const int s = i/(TILE*TILE/4);
        const int j = i - s*(TILE*TILE/4);
        od_store_coef4(plane + (long)(y0 + j/(TILE/4))*a.w + xb + s*TILE + (j % (TILE/4))*4,
         make_int4(o[k][0], o[k][1], o[k][2], o[k][3]));
      }
    }
    for (int s = 0; s < 2; s++) split_filter_rows<TILE, 4, false>(t[s], tid, y0, a.pic_h);
    od_lds_barrier();
  }
  /* ---- 32-point level: full network, 128 lanes per tile -------------------- */
  od_tile_cols<TILE, 3, false, T, 128>(z[s2], t[s2], lt, OdAllBlocks());
  od_lds_barrier();
  od_tile_rows<TILE, 3, false, T, 128>(z[s2], z[s2], lt, OdAllBlocks());
  for (int s = 0; s < 2; s++) split_filter_cols<TILE, 3, false>(t[s], tid, xb + s*TILE, a.pic_w);
  od_lds_barrier();
  for (int s = 0; s < 2; s++) {
    if (a.levels[3]) store_tile<TILE>(a.levels[3] + plane_off, a.w, xb + s*TILE, y0, z[s], tid);
    split_filter_rows<TILE, 3, false>(t[s], tid, y0, a.pic_h);
  }
  od_lds_barrier();
  /* ---- 16-point level ------------------------------------------------------ */
  for (int s = 0; s < 2; s++) od_tile_cols<TILE, 2, false, T, NT>(z[s], t[s], tid, OdAllBlocks());
  od_lds_barrier();
  for (int s = 0; s < 2; s++) {
    od_tile_rows<TILE, 2, false, T, NT>(z[s], z[s], tid, OdAllBlocks());
    split_filter_cols<TILE, 2, false>(t[s], tid, xb + s*TILE, a.pic_w);
  }
  od_lds_barrier();
  for (int s = 0; s < 2; s++) {
    if (a.levels[2]) store_tile<TILE>(a.levels[2] + plane_off, a.w, xb + s*TILE, y0, z[s], tid);
    split_filter_rows<TILE, 2, false>(t[s], tid, y0, a.pic_h);
  }
  od_lds_barrier();
  /* ---- 8-point level ------------------------------------------------------- */
  for (int s = 0; s < 2; s++) od_tile_cols<TILE, 1, false, T, NT>(z[s], t[s], tid, OdAllBlocks());
  od_lds_barrier();
  for (int s = 0; s < 2; s++) {
    od_tile_rows<TILE, 1, false, T, NT>(z[s], z[s], tid, OdAllBlocks());
    split_filter_cols<TILE, 1, false>(t[s], tid, xb + s*TILE, a.pic_w);
  }
  od_lds_barrier();
  for (int s = 0; s < 2; s++) {
    if (a.levels[1]) store_tile<TILE>(a.levels[1] + plane_off, a.w, xb + s*TILE, y0, z[s], tid);
    split_filter_rows<TILE, 1, false>(t[s], tid, y0, a.pic_h);
  }
  od_lds_barrier();
  /* ---- 4x4 level ----------------------------------------------------------- */
  for (int s = 0; s < 2; s++) pyramid_level4<TILE, T>(t[s], a, plane_off, xb + s*TILE, y0, tid);
}

/* TWO horizontally adjacent 4:2:0 chroma superblocks per 128-thread workgroup (round 4).  A 32-point
   pass has 32 columns (rows) per 32x32 tile and a wavefront 64 lanes: with one tile per wavefront
   (k_forward_pyramid<32>) half of the lanes idle through the most expensive level - 55 % of the
   kernel's network instructions.  Here wavefront 0 runs the 32-point passes of BOTH tiles (lane =
   (tile, column), then (tile, row)), and below that level each wavefront takes its own tile through
   the generic 64-lane code (the barriers are workgroup barriers that both wavefronts meet in the same
   order). */
template <typename T>
__global__ __launch_bounds__(128) void k_forward_pyramid32x2(PyramidArgs a) {
  constexpr int TILE = 32;
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  constexpr int NT = 128;
  __shared__ __attribute__((aligned(16))) short t[2][G::kHaloWords];
  __shared__ __attribute__((aligned(16))) int z[2][TILE*P];
  const int tid = threadIdx.x;
  const int wv = tid >> 6;
  const int lane = tid & 63;
  const int xb = blockIdx.x*2*TILE;
  const int y0 = blockIdx.y*TILE;
  const uint8_t *px = pyr_plane(a, blockIdx.z);
  const long plane_off = (long)blockIdx.z*a.w*a.h;
  sb_load<TILE, NT, 2>(t, a, px, xb, y0, tid);
  od_lds_barrier();
  for (int s = 0; s < 2; s++) sb_edge_cols<TILE, NT>(t[s], a, xb + s*TILE, y0, tid);
  od_lds_barrier();
  for (int s = 0; s < 2; s++) sb_edge_rows<TILE, NT>(t[s], a, xb + s*TILE, tid);
  od_lds_barrier();
  /* ---- 32-point level ------------------------------------------------------------------- */
  {
    const int s = lane >> 5;
    const int c = lane & 31;
    if (wv == 0) {
      T in[TILE];
      T out[TILE];
#pragma unroll
      for (int r = 0; r < TILE; r++) in[r] = T(t[s][r*P + c]);
      od_fdct_lift<3>(out, in);
#pragma unroll
      for (int r = 0; r < TILE; r++) z[s][r*P + c] = out[r];
    }
    od_lds_barrier();
    if (wv == 0) {
      T in[TILE];
      T out[TILE];
      int *row = z[s] + c*P;
#pragma unroll
      for (int q = 0; q < TILE; q += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(row + q);
        in[q] = T(v.x);
        in[q + 1] = T(v.y);
        in[q + 2] = T(v.z);
        in[q + 3] = T(v.w);
      }
      od_fdct_lift<3>(out, in);
#pragma unroll
      for (int q = 0; q < TILE; q += 4) {
        *reinterpret_cast<int4 *>(row + q) = make_int4(out[q], out[q + 1], out[q + 2], out[q + 3]);
      }
    }
    /* the tiles are free: lapping of the next level overlaps the row pass */
    split_filter_cols<TILE, 3, false, 64>(t[wv], lane, xb + wv*TILE, a.pic_w);
    od_lds_barrier();
    if (a.levels[3]) store_tile<TILE, 64>(a.levels[3] + plane_off, a.w, xb + wv*TILE, y0, z[wv], lane);
    split_filter_rows<TILE, 3, false, 64>(t[wv], lane, y0, a.pic_h);
    od_lds_barrier();
  }
  /* ---- 16-, 8- and 4-point levels: each wavefront its own tile -------------------------------- */
  pyramid_level<TILE, 2, T, 64>(t[wv], z[wv], a, plane_off, xb + wv*TILE, y0, lane);
}

#ifdef ODHIP_EXPERIMENTS
#include "lapped_pyramid_exp.cuh"
#endif

/* ---- inverse ------------------------------------------------------------ */

struct InverseArgs {
  const od_coeff *coef;  /* quantised coefficients, plane layout */
  uint8_t *px;           /* out: pixels (edge strips are finished by k_edge_*) */
  od_coeff *vs;          /* out: [plane][nhsb-1][h][4] samples around vertical SB edges */
  od_coeff *hs;          /* out: [plane][nvsb-1][4][w] samples around horizontal SB edges */
  int px_stride;
  long px_plane_stride;
  int w;
  int h;
  int pic_w;
  int pic_h;
  int leaf_bs;
  /* Optional PVQ source (odhip_inverse_level_pvq): when y != NULL the tile is
     not read from a dequantised plane but synthesised on load from the chosen
     pulse vectors; `coef` then only supplies the DCs. */
  const int16_t *y;         /* [2][nblocks][len] */
  const int4 *choice;       /* [nblocks][nb_bands] {slot, qg, scale, qshift} */
  const int16_t *qm_inv;    /* coding order */
  long nblocks;
  int len;
  int nb_bands;
  /* With-reference source (odhip_inverse_levels_pvq_ref): y / choice are those of an
     odhip_pvq_refjob (choice: [nblocks][nb_bands][16] ints, the synthesis parameters
     in words 8..15), r16 the QM-scaled reference after od_compute_householder, ref
     the reference plane itself (skip-copy bands). */
  const int16_t *r16;
  const od_coeff *ref;
  int px16;                 /* full-precision references: px holds int16 samples, see PyramidArgs */
  int inter;                /* with-reference source of an inter frame (is_keyframe == 0) */
  int dbg;                  /* ODHIP_INVERSE_DBG (k_inverse_walk, timing experiments only): 1 = no pixel
                               stores, 2 = no strip stores, 4 = no source loads (WRONG results) */
  int wide;                 /* k_inverse_walk: 8-bit samples in a 16-byte aligned plane - rows go out as
                               whole lines, one group late (walk_store_prev) */
};

/* Several partition levels of one plane set in ONE launch (blockIdx.z = level *
   nplanes + plane): five back-to-back launches of 4080 workgroups each leave a
   partially filled last round per launch, and the two edge kernels per level
   (a few microseconds of work) cost a launch each. */
constexpr int kMaxInvLevels = 5;
struct InverseArgsMulti {
  InverseArgs a[kMaxInvLevels];
  int nplanes;
};

__device__ __attribute__((aligned(16))) unsigned short gInvScanXY[OD_SCAN_LEN];  /* y << 8 | x of coding index j */
__device__ unsigned char gInvBandOf[OD_SCAN_LEN];

/* od_coeff_to_ref_buf, src/state.c:1296-1304. */
__device__ __forceinline__ unsigned char od_to_px(int c) {
  return (unsigned char)min(max(((c + 8) >> 4) + 128, 0), 255);
}
/* ... with full-precision references, :1313-1318: OD_CLAMPFPR(c + (128 << OD_COEFF_SHIFT)). */
__device__ __forceinline__ short od_to_px16(int c) {
  return (short)min(max(c + 2048, 0), 4095);
}
__device__ __forceinline__ void od_store_px(uint8_t *plane, long at, int c, bool px16) {
  if (px16) reinterpret_cast<short *>(plane)[at] = od_to_px16(c);
  else plane[at] = od_to_px(c);
}

/* OD_MULT16_32_Q16(y, scale) = (int16)y * (int32)scale >> 16 (src/internal.h) without a 32-bit
   multiply: v_mul_hi_i32 and v_mul_lo_u32 issue at a quarter of the rate of the 24-bit multiplier.
   scale = sh*65536 + sl with sh = scale >> 16 (16 bits, signed) and sl = scale & 0xffff, so
   y*scale >> 16 = y*sh + (y*sl >> 16) EXACTLY (y*sh*65536 is a multiple of 65536; |y*sl| < 2^31):
   two 24-bit multiplies of 16-bit operands and a shift.  The value it returns is a QM-scaled
   coefficient of the synthesis, |x| < 2^16 by construction (od_pvq_synthesis_partial scales g so
   that the vector's NORM fits 16 bits, src/pvq.c:1055-1075, and |y_i| <= sqrt(yy)), so the inverse
   quantisation-matrix product x*qm_inv[i] that follows (src/pvq.c:1092) goes through the 24-bit
   multiplier too: the low 32 bits of its 48-bit product are the reference's int product. */
struct OdQ16 {
  int sh;
  int sl;
  __device__ __forceinline__ explicit OdQ16(int scale) : sh(scale >> 16), sl(scale & 0xffff) {}
  __device__ __forceinline__ int mul(int y) const { return __mul24(y, sh) + (__mul24(y, sl) >> 16); }
};
/* pulse j of four packed 16-byte words (eight int16 per pair of words), sign-extended */
__device__ __forceinline__ int od_pulse16(unsigned w, int odd) {
  return odd ? (int)w >> 16 : (int)(w << 16) >> 16;
}

/* Raster plane -> tile, every lane's global loads in flight before its first LDS write. */
template <int TILE>
__device__ __forceinline__ void load_plane_tile(int *t, const od_coeff *plane, int w, int x0, int y0, int tid) {
  constexpr int P = Geo<TILE>::kPitch;
  constexpr int NT = Geo<TILE>::kNT;
  static_assert((TILE*TILE/4) % NT == 0, "whole trips");
  constexpr int K = TILE*TILE/4/NT;
  int4 v[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = tid + k*NT;
    v[k] = *reinterpret_cast<const int4 *>(plane + (long)(y0 + i/(TILE/4))*w + x0 + (i % (TILE/4))*4);
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = tid + k*NT;
    *reinterpret_cast<int4 *>(t + (i/(TILE/4))*P + (i % (TILE/4))*4) = v[k];
  }
}

template <int TILE, int LN>
__device__ __forceinline__ void inverse_split_levels(int *t, const InverseArgs &a,
 int x0, int y0, int tid) {
  constexpr int TOP = TILE == 64 ? 4 : 3;
  if (LN > a.leaf_bs) {
    /* od_postfilter_split, src/filter.c:1510-1525: rows first, then columns. */
    split_filter_rows<TILE, LN, true>(t, tid, y0, a.pic_h);
    __syncthreads();
    split_filter_cols<TILE, LN, true>(t, tid, x0, a.pic_w);
    __syncthreads();
  }
  if constexpr (LN < TOP) inverse_split_levels<TILE, LN + 1>(t, a, x0, y0, tid);
}

template <int TILE, int LN>
__device__ __forceinline__ void inverse_leaf(int *t, int tid) {
  using T = OdMul24;
  constexpr int NT = Geo<TILE>::kNT;
  od_tile_rows<TILE, LN, true, T, NT>(t, t, tid, OdAllBlocks());
  __syncthreads();
  od_tile_cols<TILE, LN, true, T, NT>(t, t, tid, OdAllBlocks());
  __syncthreads();
}

/* Dequantise-on-load of a with-reference band stage's choice (the per-coefficient
   part of od_pvq_synthesis_partial, src/pvq.c:1081-1114, as k_refb_synth computes
   it, but into the LDS tile instead of a plane): eight consecutive coding
   positions of one block per thread; a chunk never straddles a band. */
template <int TILE>
__device__ __forceinline__ void inverse_load_ref(int *t, const InverseArgs &a, int plane, long plane_off,
 int x0, int y0, int tid) {
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  constexpr int NT = G::kNT;
  const int sh = a.leaf_bs + 2;
  const int nbw = TILE >> sh;
  const int nbsb = nbw*nbw;
  const int bw = a.w >> sh;
  const int bh = a.h >> sh;
  if ((a.len >> sh) < (1 << sh)) {
    /* 32x32 / 64x64: the positions PVQ never codes are what od_init_skipped_coeffs leaves
       (src/state.c:1347-1366): zero on a keyframe, the prediction's coefficients otherwise */
    if (a.inter) {
      load_plane_tile<TILE>(t, a.ref + plane_off, a.w, x0, y0, tid);
    }
    else {
      for (int i = tid; i < TILE*P/4; i += NT) reinterpret_cast<int4 *>(t)[i] = make_int4(0, 0, 0, 0);
    }
    __syncthreads();
  }
  const int cpb = a.len >> 3;
  const int4 *choice4 = a.choice;
  for (int c = tid; c < nbsb*cpb; c += NT) {
    const int b = c/cpb;
    const int c0 = (c - b*cpb) << 3;
    const int lby = b/nbw;
    const int lbx = b - lby*nbw;
    const long blk = ((long)plane*bh + (y0 >> sh) + lby)*bw + (x0 >> sh) + lbx;
    const int band = gInvBandOf[c0 ? c0 : 1];
    const int4 ca = choice4[(blk*a.nb_bands + band)*4 + 2];   /* mode, slot, scale, qshift */
    const int mode = ca.x;
    const int base = (lby << sh)*P + (lbx << sh);
    const uint4 sc4 = *reinterpret_cast<const uint4 *>(gInvScanXY + c0);
    const unsigned scw[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
    int pos[8];
    long gpos[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const unsigned pk = (scw[e >> 1] >> (16*(e & 1))) & 0xffffu;
      pos[e] = base + (int)(pk >> 8)*P + (int)(pk & 255);
      gpos[e] = (long)(y0 + (lby << sh) + (int)(pk >> 8))*a.w + x0 + (lbx << sh) + (int)(pk & 255);
    }
    const int first = c0 == 0;
    if (first) t[pos[0]] = a.coef[plane_off + gpos[0]];
    if (mode == 0) {
#pragma unroll
      for (int e = 0; e < 8; e++) if (!(first && e == 0)) t[pos[e]] = 0;
      continue;
    }
    if (mode == 1 || mode == 4) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (first && e == 0) continue;
        const od_coeff rv = a.ref[plane_off + gpos[e]];
        t[pos[e]] = mode == 4 ? -rv : rv;
      }
      continue;
    }
    /* first coding position of the band (OD_BAND_OFFSETS, src/partition.c:77-91) */
    const int off = c0 < 16 ? 1 : c0 < 24 ? 16 : c0 < 32 ? 24 : c0 < 64 ? 32 : c0 < 96 ? 64
     : c0 < 128 ? 96 : c0 < 256 ? 128 : c0 < 384 ? 256 : 384;
    const int yslot = ca.y;
    const int32_t scale = ca.z;
    const int qshift = ca.w;
    const uint4 q4 = *reinterpret_cast<const uint4 *>(a.qm_inv + c0);
    const unsigned qw[4] = {q4.x, q4.y, q4.z, q4.w};
    unsigned yw[4] = {0, 0, 0, 0};
    int yprev = 0;
    if (yslot >= 0) {
      const int16_t *yp = a.y + ((long)yslot*a.nblocks + blk)*a.len + c0;
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
        t[pos[e]] = odq_shr_round(x*qmi, qshift);
      }
      continue;
    }
    const int4 cb = choice4[(blk*a.nb_bands + band)*4 + 3];   /* xm, m, proj_1, outshift */
    const int m = cb.y;
    const uint4 r4 = *reinterpret_cast<const uint4 *>(a.r16 + blk*a.len + c0);
    const unsigned rw[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (first && e == 0) continue;
      const int i = c0 + e - off;
      const int yv = (int16_t)(yw[e >> 1] >> (16*(e & 1)));
      const int ym1 = e == 0 ? yprev : (int)(int16_t)(yw[(e - 1) >> 1] >> (16*((e - 1) & 1)));
      const int qmi = (int16_t)(qw[e >> 1] >> (16*(e & 1)));
      const int ri = (int16_t)(rw[e >> 1] >> (16*(e & 1)));
      const int16_t xi = i == m ? (int16_t)cb.x : (int16_t)odq_mult16_32_q16(i < m ? yv : ym1, scale);
      int32_t tmp = odq_mult16_16(ri, cb.z);
      tmp = cb.w >= 0 ? odq_shr_round(tmp, cb.w) : odq_shl32(tmp, -cb.w);
      const int16_t v = (int16_t)(xi - tmp);
      t[pos[e]] = odq_shr_round(v*qmi, qshift);
    }
  }
}

/* Everything but the 2-sample strips along interior superblock edges is final after
   the tile's own post-filters: convert and store it (the strips are stored too and
   overwritten by k_edge_rows / k_edge_cols).  The strips go out as od_coeff so that the
   edge post-filter, which couples neighbouring superblocks, can run on them: 4 B read
   + 1 B written per pixel plus ~12 % for the strips, instead of an int32 round trip of
   the whole plane. */
template <int TILE>
__device__ __forceinline__ void inverse_store(const int *t, const InverseArgs &a, int plane, int x0,
 int y0, int tid, int sbx, int sby, int done_edges = 0) {
  constexpr int P = Geo<TILE>::kPitch;
  constexpr int NT = Geo<TILE>::kNT;
  const int w = a.w;
  const int h = a.h;
  const bool px16 = a.px16 != 0;
  uint8_t *px = a.px + ((long)plane*a.px_plane_stride << a.px16);
  static_assert((TILE*TILE/4) % NT == 0, "whole trips");
  constexpr int K = TILE*TILE/4/NT;
  int4 tv[K];
#pragma unroll
  for (int k = 0; k < K; k++) {               /* all LDS reads in flight before the first store */
    const int i = tid + k*NT;
    tv[k] = *reinterpret_cast<const int4 *>(t + (i/(TILE/4))*P + (i % (TILE/4))*4);
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = tid + k*NT;
    const int y = i/(TILE/4);
    const int x = (i % (TILE/4))*4;
    const int4 v = tv[k];
    const long at = (long)(y0 + y)*a.px_stride + x0 + x;
    if (px16) {
      *reinterpret_cast<short4 *>(reinterpret_cast<short *>(px) + at) =
       make_short4(od_to_px16(v.x), od_to_px16(v.y), od_to_px16(v.z), od_to_px16(v.w));
    }
    else {
      uchar4 o;
      o.x = od_to_px(v.x);
      o.y = od_to_px(v.y);
      o.z = od_to_px(v.z);
      o.w = od_to_px(v.w);
      *reinterpret_cast<uchar4 *>(px + at) = o;
    }
  }
  const int nv = w/TILE - 1;
  const int nh = h/TILE - 1;
  od_coeff *vs = a.vs + (long)plane*nv*h*4;
  od_coeff *hs = a.hs + (long)plane*nh*4*w;
  for (int i = tid; i < 2*TILE; i += NT) {
    const int right = i/TILE;
    const int r = i % TILE;
    if (done_edges & (right ? 1 : 2)) continue;     /* finished in LDS by the caller */
    if (right ? sbx < nv : sbx > 0) {
      const int e = right ? sbx : sbx - 1;
      const int c = right ? TILE - 2 : 0;
      int2 v;
      v.x = t[r*P + c];
      v.y = t[r*P + c + 1];
      *reinterpret_cast<int2 *>(vs + ((long)e*h + y0 + r)*4 + (right ? 0 : 2)) = v;
    }
  }
  for (int i = tid; i < 4*TILE; i += NT) {
    const int k = i/TILE;          /* 0,1: top rows 0,1; 2,3: bottom rows TILE-2, TILE-1 */
    const int c = i % TILE;
    const bool bottom = k >= 2;
    if (bottom ? sby < nh : sby > 0) {
      const int e = bottom ? sby : sby - 1;
      const int r = bottom ? TILE - 4 + k : k;
      hs[((long)e*4 + (bottom ? k - 2 : k + 2))*w + x0 + c] = t[r*P + c];
    }
  }
}

/* ---- inverse at an ARBITRARY partition (the decoder's reconstruction) ----------------
   The block-size map of od_state (bsize, one entry per 8x8 luma area: 0 = four 4x4
   blocks, 1 = 8x8, ... 4 = 64x64; src/state.h:250-259, OD_BLOCK_SIZE4x4
   src/block_size.h:32-35) says where od_decode_recursive / od_encode_recursive
   (src/decode.c:603-660, src/encode.c:1657-1810) stopped splitting.  The recursion is
   replayed level by level inside the tile: at level L the blocks that are LEAVES of
   size L are inverse-transformed, then the nodes of size L that were SPLIT get their
   od_postfilter_split (their subtrees are complete: every level below ran first;
   regions of different nodes are disjoint, so the order between them is free).  A
   chroma plane of a 4:2:0 frame codes the block of a luma block one size down, 4x4 for
   both 8x8 and 4x4 luma (bs = OD_MAXI(obs, xdec) - xdec, src/encode.c:1674-1680). */
struct PartLeaf {
  const unsigned char *map;   /* LDS: leaf level per 8x8-luma unit of the tile, 8 per row */
  int level;
  int dec;
  /* block (bx, by) of size 4 << level, in block units of the tile */
  __device__ __forceinline__ int at(int bx, int by) const {
    const int u = (4 << level) >> (3 - dec);     /* map units per block side (0 for 4x4 luma) */
    return u ? map[(by*u)*8 + bx*u] : map[(by >> 1)*8 + (bx >> 1)];
  }
};
struct PartIsLeaf : PartLeaf {
  __device__ __forceinline__ bool operator()(int bx, int by) const { return at(bx, by) == level; }
};

template <int TILE, int LN, bool INV, typename Pred>
__device__ __forceinline__ void split_filter_cols_if(int *t, int tid, int x0, int pic_w, Pred split) {
  constexpr int N = 4 << LN;
  constexpr int P = Geo<TILE>::kPitch;
  constexpr int NT = Geo<TILE>::kNT;
  for (int k = tid; k < TILE*(TILE/N); k += NT) {
    const int x = k % TILE;
    const int by = k / TILE;
    const int gbx = (x0 + x)/N;
    if (split(x/N, by) && (gbx + 1)*N <= pic_w) lds_filter4<INV>(t + (by*N + N/2 - 2)*P + x, P);
  }
}

template <int TILE, int LN, bool INV, typename Pred>
__device__ __forceinline__ void split_filter_rows_if(int *t, int tid, int y0, int pic_h, Pred split) {
  constexpr int N = 4 << LN;
  constexpr int P = Geo<TILE>::kPitch;
  constexpr int NT = Geo<TILE>::kNT;
  for (int k = tid; k < TILE*(TILE/N); k += NT) {
    const int y = k % TILE;
    const int bx = k / TILE;
    const int gby = (y0 + y)/N;
    if (split(bx, y/N) && (gby + 1)*N <= pic_h) lds_filter4<INV>(t + y*P + bx*N + N/2 - 2, 1);
  }
}

template <int TILE, int LN>
__device__ __forceinline__ void inverse_part_levels(int *t, const unsigned char *map, const InverseArgs &a,
 int x0, int y0, int tid) {
  using T = OdMul24;
  constexpr int NT = Geo<TILE>::kNT;
  constexpr int TOP = TILE == 64 ? 4 : 3;
  constexpr int DEC = TILE == 64 ? 0 : 1;
  PartIsLeaf leaf;
  leaf.map = map;
  leaf.level = LN;
  leaf.dec = DEC;
  od_tile_rows<TILE, LN, true, T, NT>(t, t, tid, leaf);
  __syncthreads();
  od_tile_cols<TILE, LN, true, T, NT>(t, t, tid, leaf);
  __syncthreads();
  if constexpr (LN >= 1) {
    /* od_postfilter_split of the nodes of this size that were split (src/filter.c:
       1510-1525: rows first, then columns) */
    const PartLeaf node = leaf;
    auto split = [&](int bx, int by) { return node.at(bx, by) < LN; };
    split_filter_rows_if<TILE, LN, true>(t, tid, y0, a.pic_h, split);
    __syncthreads();
    split_filter_cols_if<TILE, LN, true>(t, tid, x0, a.pic_w, split);
    __syncthreads();
  }
  if constexpr (LN < TOP) inverse_part_levels<TILE, LN + 1>(t, map, a, x0, y0, tid);
}

struct InversePartArgs {
  InverseArgs a;
  const unsigned char *bsize;    /* frame f's map at bsize + f*bsize_frame_stride */
  int bstride;
  long bsize_frame_stride;
  int planes_per_frame;          /* consecutive planes that share one map */
  int nplanes;
};

template <int TILE>
__global__ __launch_bounds__(Geo<TILE>::kNT) void k_inverse_part(InversePartArgs pa) {
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  constexpr int NT = G::kNT;
  constexpr int DEC = TILE == 64 ? 0 : 1;
  __shared__ __attribute__((aligned(16))) int t[TILE*P];
  __shared__ unsigned char map[64];
  const InverseArgs &a = pa.a;
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x*TILE;
  const int y0 = blockIdx.y*TILE;
  const int plane = blockIdx.z;
  const long plane_off = (long)plane*a.w*a.h;
  if (tid < 64) {
    const unsigned char *bs = pa.bsize + (plane/pa.planes_per_frame)*pa.bsize_frame_stride
     + (long)(blockIdx.y*8 + (tid >> 3))*pa.bstride + blockIdx.x*8 + (tid & 7);
    int v = *bs;
    if (DEC) v = (v > 1 ? v : 1) - 1;      /* the chroma block of that luma block */
    map[tid] = (unsigned char)v;
  }
  load_plane_tile<TILE>(t, a.coef + plane_off, a.w, x0, y0, tid);
  __syncthreads();
  inverse_part_levels<TILE, 0>(t, map, a, x0, y0, tid);
  inverse_store<TILE>(t, a, plane, x0, y0, tid, blockIdx.x, blockIdx.y);
}

/* MINLEAF..MAXLEAF: the leaf levels this instance is launched for.  The 64- and 32-point networks
   need ~120 VGPRs (4 waves per SIMD), the 16-point and smaller ones about half of that: the luma
   levels go out as two launches so that three of the five run at twice the occupancy. */
template <int TILE, bool REF = false, int MINLEAF = 0, int MAXLEAF = (TILE == 64 ? 4 : 3)>
__global__ __launch_bounds__(Geo<TILE>::kNT) void k_inverse_sb(InverseArgsMulti mm) {
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  constexpr int NT = G::kNT;
  __shared__ __attribute__((aligned(16))) int t[TILE*P];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x*TILE;
  const int y0 = blockIdx.y*TILE;
  const int plane = blockIdx.z % mm.nplanes;
  const InverseArgs &a = mm.a[blockIdx.z / mm.nplanes];
  const long plane_off = (long)plane*a.w*a.h;
  if constexpr (REF) {
    inverse_load_ref<TILE>(t, a, plane, plane_off, x0, y0, tid);
  }
  else if (a.y) {
    /* Dequantise on load: x = y*scale (Q16, no rounding), out = SHR_ROUND(x *
       qm_inv, qshift) (od_pvq_synthesis_partial noref, src/pvq.c:1081-1092),
       scattered to raster inside LDS (od_coding_order_to_raster,
       src/partition.c:176-194).  The pulse vectors are read as contiguous
       len-word rows; the dequantised plane never exists in HBM. */
    const int sh = a.leaf_bs + 2;
    const int nbw = TILE >> sh;                 /* blocks per tile row */
    const int nbsb = nbw*nbw;
    const int bw = a.w >> sh;
    const int bh = a.h >> sh;
    /* One CHUNK of 16 consecutive coding indices of one block per thread (len/16 chunks per
       block; nbsb*len/16 <= NT for every level, so a thread has at most one chunk).  Everything
       the chunk needs comes straight from global memory into registers - band of its two halves
       (gInvBandOf: 1 KB, cache-resident), their choice records, two 16-byte loads of pulses, of
       inverse-QM entries and of scan positions - in one dependent chain band -> choice -> pulses
       with the table loads beside it: the workgroup pays that chain ONCE.  (Staging the tables
       and the choices in LDS first cost a load -> wait -> LDS store per 256 entries, up to 12
       serial memory latencies per workgroup, and 9 KB of LDS: 6 instead of 9 workgroups per CU.)
       A chunk touches at most two bands (coding indices 16..31 hold two bands of 8). */
    const int lsh = 31 - __clz(a.len);          /* len is a power of two */
    const int csh = lsh - 4;                    /* chunks per block = len/16 */
    const int lnb = 31 - __clz(nbw);            /* so is the block count per tile row */
    const long blk0 = ((long)plane*bh + (y0 >> sh))*bw + (x0 >> sh);
    const int c = tid;
    const bool act = c < (nbsb << csh);
    const int b = c >> csh;
    const int j0 = (c & ((1 << csh) - 1)) << 4;
    const int lby = b >> lnb;
    const int lbx = b & (nbw - 1);
    const unsigned blk = (unsigned)blk0 + lby*bw + lbx;   /* < 2^31/len, checked by the host */
    int4 chs[2] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
    int4 yq[2] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
    int4 qm4[2];
    int4 sc4[2];
    int dc = 0;
    if (act) {
      const int bnd0 = gInvBandOf[j0 ? j0 : 1];
      const int bnd1 = gInvBandOf[j0 + 8];
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        qm4[hf] = *reinterpret_cast<const int4 *>(a.qm_inv + j0 + 8*hf);
        sc4[hf] = *reinterpret_cast<const int4 *>(gInvScanXY + j0 + 8*hf);
      }
      if (j0 == 0) dc = a.coef[plane_off + (long)(y0 + (lby << sh))*a.w + x0 + (lbx << sh)];
      chs[0] = a.choice[(long)blk*a.nb_bands + bnd0];
      chs[1] = a.choice[(long)blk*a.nb_bands + bnd1];
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        if (chs[hf].y != 0) {
          yq[hf] = *reinterpret_cast<const int4 *>(a.y
           + (((unsigned)chs[hf].x*(unsigned)a.nblocks + blk) << lsh) + j0 + 8*hf);
        }
      }
    }
    if ((a.len >> sh) < (1 << sh)) {            /* 32x32 / 64x64: uncoded positions are zero */
      for (int i = tid; i < TILE*P/4; i += NT) reinterpret_cast<int4 *>(t)[i] = make_int4(0, 0, 0, 0);
      __syncthreads();
    }
    if (act) {
      const int base = (lby << sh)*P + (lbx << sh);
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const int yd[4] = {yq[hf].x, yq[hf].y, yq[hf].z, yq[hf].w};
        const int qd[4] = {qm4[hf].x, qm4[hf].y, qm4[hf].z, qm4[hf].w};
        const int sd[4] = {sc4[hf].x, sc4[hf].y, sc4[hf].z, sc4[hf].w};
        const int4 ch = chs[hf];
        const int rnd = (1 << ch.w) >> 1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
          for (int u = 0; u < 2; u++) {
            /* OD_MULT16_32_Q16: (int16)y * (int32)scale >> 16 == mulhi(y << 16, scale) */
            const int yhi = u ? (yd[e] & (int)0xffff0000) : yd[e] << 16;
            const int qmi = u ? qd[e] >> 16 : (int)(short)qd[e];
            const int xy = u ? (unsigned)sd[e] >> 16 : sd[e] & 0xffff;
            int v = (__mulhi(yhi, ch.z)*qmi + rnd) >> ch.w;
            if (ch.y == 0) v = 0;
            if (hf == 0 && e == 0 && u == 0 && j0 == 0) v = dc;
            t[base + (xy >> 8)*P + (xy & 255)] = v;
          }
        }
      }
    }
  }
  else {
    load_plane_tile<TILE>(t, a.coef + plane_off, a.w, x0, y0, tid);
  }
  __syncthreads();
  switch (a.leaf_bs) {
    case 0: if constexpr (MINLEAF <= 0 && MAXLEAF >= 0) inverse_leaf<TILE, 0>(t, tid); break;
    case 1: if constexpr (MINLEAF <= 1 && MAXLEAF >= 1) inverse_leaf<TILE, 1>(t, tid); break;
    case 2: if constexpr (MINLEAF <= 2 && MAXLEAF >= 2) inverse_leaf<TILE, 2>(t, tid); break;
    case 3: if constexpr (MINLEAF <= 3 && MAXLEAF >= 3) inverse_leaf<TILE, 3>(t, tid); break;
    default:
      if constexpr (TILE == 64 && MAXLEAF >= 4) inverse_leaf<TILE, 4>(t, tid);
      break;
  }
  inverse_split_levels<TILE, 1>(t, a, x0, y0, tid);
  inverse_store<TILE>(t, a, plane, x0, y0, tid, blockIdx.x, blockIdx.y);
}

/* The 32x32 and 64x64 leaf levels of luma, TWO horizontally adjacent superblocks per 256-thread
   workgroup: a 64-point pass has 64 columns (rows) per tile and a 32-point pass 128, so one tile keeps one /
   two of a workgroup's four waves busy while the others wait at the barriers; with two tiles the 32-point
   passes use every lane and the 64-point ones half.  Pulse-fed source only (odhip_inverse_levels_pvq). */
__global__ __launch_bounds__(256) void k_inverse_sb_top2(InverseArgsMulti mm) {
  constexpr int TILE = 64;
  using G = Geo<TILE>;
  using T = OdMul24;
  constexpr int P = G::kPitch;
  constexpr int NT = 256;
  __shared__ __attribute__((aligned(16))) int t[2][TILE*P];
  const int tid = threadIdx.x;
  const int xb = blockIdx.x*2*TILE;
  const int y0 = blockIdx.y*TILE;
  const int plane = blockIdx.z % mm.nplanes;
  const InverseArgs &a = mm.a[blockIdx.z / mm.nplanes];
  const long plane_off = (long)plane*a.w*a.h;
  const int sh = a.leaf_bs + 2;               /* 5 or 6 */
  const int nbw = TILE >> sh;                 /* 2 or 1 blocks per tile row */
  const int nbsb = nbw*nbw;
  const int bw = a.w >> sh;
  const int bh = a.h >> sh;
  const int lsh = 31 - __clz(a.len);
  const int csh = lsh - 4;
  const int lnb = 31 - __clz(nbw);
  const int nchunks = nbsb << csh;            /* per tile: 128 (32x32 leaves: 4 blocks x 512 coded) or 32 (64x64) */
  /* thread tid takes chunk tid % nchunks of tile tid / nchunks: both tiles' chunks in one trip */
  const int st = tid/nchunks;
  const int c = tid - st*nchunks;
  const bool act = st < 2;
  const int x0 = xb + st*TILE;
  const int b = c >> csh;
  const int j0 = (c & ((1 << csh) - 1)) << 4;
  const int lby = b >> lnb;
  const int lbx = b & (nbw - 1);
  int4 chs[2] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
  int4 yq[2] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
  int4 qm4[2];
  int4 sc4[2];
  int dc = 0;
  if (act) {
    const unsigned blk = (unsigned)(((long)plane*bh + (y0 >> sh))*bw + (x0 >> sh)) + lby*bw + lbx;
    const int bnd0 = gInvBandOf[j0 ? j0 : 1];
    const int bnd1 = gInvBandOf[j0 + 8];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      qm4[hf] = *reinterpret_cast<const int4 *>(a.qm_inv + j0 + 8*hf);
      sc4[hf] = *reinterpret_cast<const int4 *>(gInvScanXY + j0 + 8*hf);
    }
    if (j0 == 0) dc = a.coef[plane_off + (long)(y0 + (lby << sh))*a.w + x0 + (lbx << sh)];
    chs[0] = a.choice[(long)blk*a.nb_bands + bnd0];
    chs[1] = a.choice[(long)blk*a.nb_bands + bnd1];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      if (chs[hf].y != 0) {
        yq[hf] = *reinterpret_cast<const int4 *>(a.y
         + (((unsigned)chs[hf].x*(unsigned)a.nblocks + blk) << lsh) + j0 + 8*hf);
      }
    }
  }
  /* uncoded positions of these levels are zero */
  for (int i = tid; i < 2*TILE*P/4; i += NT) reinterpret_cast<int4 *>(&t[0][0])[i] = make_int4(0, 0, 0, 0);
  __syncthreads();
  if (act) {
    const int base = (lby << sh)*P + (lbx << sh);
    int *tt = t[st];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      const int yd[4] = {yq[hf].x, yq[hf].y, yq[hf].z, yq[hf].w};
      const int qd[4] = {qm4[hf].x, qm4[hf].y, qm4[hf].z, qm4[hf].w};
      const int sd[4] = {sc4[hf].x, sc4[hf].y, sc4[hf].z, sc4[hf].w};
      const int4 ch = chs[hf];
      const int rnd = (1 << ch.w) >> 1;
      const OdQ16 sc(ch.z);
#pragma unroll
      for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int yv = od_pulse16((unsigned)yd[e], u);
          const int qmi = u ? qd[e] >> 16 : (int)(short)qd[e];
          const int xy = u ? (unsigned)sd[e] >> 16 : sd[e] & 0xffff;
          /* (pulses of a band that codes nothing were not read: zero in, zero out) */
          int v = (__mul24(sc.mul(yv), qmi) + rnd) >> ch.w;
          if (hf == 0 && e == 0 && u == 0 && j0 == 0) v = dc;
          tt[base + (xy >> 8)*P + (xy & 255)] = v;
        }
      }
    }
  }
  __syncthreads();
  if (a.leaf_bs == 3) {
    /* 32-point: 128 rows (columns) per tile, tile = tid >> 7 */
    const int s2 = tid >> 7;
    const int lt = tid & 127;
    od_tile_rows<TILE, 3, true, T, 128>(t[s2], t[s2], lt, OdAllBlocks());
    __syncthreads();
    od_tile_cols<TILE, 3, true, T, 128>(t[s2], t[s2], lt, OdAllBlocks());
    __syncthreads();
  }
  else {
    /* 64-point: 64 per tile, waves 0 and 1.  Rows: only rows 0..31 of a 64x64 block hold coded
       coefficients (the first 512 coding positions lie in its top-left 32x32 corner,
       src/partition.c:144-194), and the network maps a zero row to a zero row: one wavefront
       transforms the 32 live rows of both tiles */
    const int g = tid >> 6;
    if (tid < 64) {
      int *tt = t[tid >> 5];
      const int base = (tid & 31)*P;
      T in[64];
      T out[64];
      /* ... and only columns 0..31 of a live row: the upper 32 inputs are literal zeros, which prunes
         the 64-point network at compile time (round 5) */
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(tt + base + c);
        in[c] = T(v.x);
        in[c + 1] = T(v.y);
        in[c + 2] = T(v.z);
        in[c + 3] = T(v.w);
      }
#pragma unroll
      for (int c = 32; c < 64; c++) in[c] = T(0);
      od_idct_lift<4>(out, in);
#pragma unroll
      for (int c = 0; c < 64; c += 4) {
        *reinterpret_cast<int4 *>(tt + base + c) = make_int4(out[c], out[c + 1], out[c + 2], out[c + 3]);
      }
    }
    __syncthreads();
    if (g < 2) {
      /* columns: rows 32..63 of the tile were not transformed and are zero - the same pruned network */
      int *tt = t[g] + (tid & 63);
      T in[64];
      T out[64];
#pragma unroll
      for (int r = 0; r < 32; r++) in[r] = T(tt[r*P]);
#pragma unroll
      for (int r = 32; r < 64; r++) in[r] = T(0);
      od_idct_lift<4>(out, in);
#pragma unroll
      for (int r = 0; r < 64; r++) tt[r*P] = out[r];
    }
    __syncthreads();
  }
  if (a.leaf_bs == 3) {
    /* od_postfilter_split of the 64x64 node above (src/filter.c:1510-1525: rows first, then columns) */
    for (int s = 0; s < 2; s++) split_filter_rows<TILE, 4, true>(t[s], tid, y0, a.pic_h);
    __syncthreads();
    for (int s = 0; s < 2; s++) split_filter_cols<TILE, 4, true>(t[s], tid, xb + s*TILE, a.pic_w);
    __syncthreads();
  }
  /* the vertical superblock edge between the two tiles (od_apply_postfilter_frame_sbs,
     src/filter.c:1600-1606) is finished here; k_edge_rows takes the odd edges only */
  if (tid < TILE) {
    int *l = t[0] + tid*P + TILE - 2;
    int *r = t[1] + tid*P;
    int t0 = l[0];
    int t1 = l[1];
    int t2 = r[0];
    int t3 = r[1];
    od_post_filter4_dev24(t0, t1, t2, t3);
    l[0] = t0;
    l[1] = t1;
    r[0] = t2;
    r[1] = t3;
  }
  __syncthreads();
  for (int s = 0; s < 2; s++) {
    inverse_store<TILE>(t[s], a, plane, xb + s*TILE, y0, tid, 2*blockIdx.x + s, blockIdx.y, s == 0 ? 1 : 2);
  }
}

/* ---- inverse, workgroups that WALK along a superblock row --------------------------------
   One workgroup reconstructs a SEGMENT of horizontally adjacent superblocks of one (level, plane,
   superblock row), group after group (a group = G adjacent tiles that are in LDS together).  What
   that buys over one workgroup per superblock (k_inverse_sb above):

   * the post-filter across the VERTICAL superblock edges inside a segment
     (od_apply_postfilter_frame_sbs, first half, src/filter.c:1600-1606) runs in LDS: the last four
     columns of a group stay behind in the pad columns of the tile ("keep": pitch = TILE + 4, the
     four spare words of every row), the next group filters keep[2..3] | its own columns 0..1, and
     the pixel rows go out in a window shifted four samples to the left.  No 2-sample strips, no
     second kernel touching 4 bytes of every 64-byte line of the plane (k_edge_rows wrote 170 MB
     for 12 MB of pixels and took 127 us per step); only the JOINTS between segments still go
     through the strips and k_edge_rows;
   * the leaf level is a compile-time constant per body (one kernel, a wave-uniform switch):
     block / chunk indices are shifts, no runtime divisions in the dequantise-on-load;
   * 4:2:0 chroma takes its 32x32 superblocks in PAIRS: a 32-point pass has 32 rows per tile, a
     wavefront 64 lanes;
   * LDS-only barriers between the phases (the stores of a group drain under the next group).

   The horizontal edges still go through the hs strips (written after the vertical-edge filter,
   which is the reference's order) and k_edge_cols. */
#ifndef OD_WALK_WAVES
#define OD_WALK_WAVES 1
#endif
/* InverseArgs::dbg (ODHIP_INVERSE_DBG) exists in the experiments build only. */
#ifdef ODHIP_EXPERIMENTS
#define OD_INV_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define OD_INV_DBG(a, bit) false
#endif
struct InverseWalkArgs {
  InverseArgs a[kMaxInvLevels];
  int nplanes;
  int seg_len;                   /* groups per workgroup */
};

/* One pass (rows, then columns: od_bin_idctNxN, src/dct.c:158-163 ...) of the leaf transforms of
   the G tiles at t. */
template <int TILE, int G, int NT, int LN, bool ROWS>
__device__ __forceinline__ void walk_idct_pass(int *t, unsigned tid) {
  using T = OdMul24;
  constexpr int N = 4 << LN;
  constexpr int P = TILE + 4;
  constexpr int kTasks = TILE*(TILE/N);
  for (unsigned k = tid; k < G*kTasks; k += NT) {
    int *ts = t + (k/kTasks)*(TILE*P);
    const unsigned kk = k % kTasks;
    T in[N];
    T out[N];
    if constexpr (ROWS) {
      const unsigned base = (kk % TILE)*P + (kk/TILE)*N;
#pragma unroll
      for (int c = 0; c < N; c += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(ts + base + c);
        in[c] = T(v.x);
        in[c + 1] = T(v.y);
        in[c + 2] = T(v.z);
        in[c + 3] = T(v.w);
      }
      od_idct_lift<LN>(out, in);
#pragma unroll
      for (int c = 0; c < N; c += 4) {
        *reinterpret_cast<int4 *>(ts + base + c) = make_int4(out[c], out[c + 1], out[c + 2], out[c + 3]);
      }
    }
    else {
      const unsigned base = (kk/TILE)*N*P + kk % TILE;
#pragma unroll
      for (int r = 0; r < N; r++) in[r] = T(ts[base + r*P]);
      od_idct_lift<LN>(out, in);
#pragma unroll
      for (int r = 0; r < N; r++) ts[base + r*P] = out[r];
    }
  }
}

/* 4-point post-filter across p[0], p[step], p[2 step], p[3 step] (bounded values, see od_lift.cuh). */
__device__ __forceinline__ void walk_post4(int *p, int step) {
  int t0 = p[0];
  int t1 = p[step];
  int t2 = p[2*step];
  int t3 = p[3*step];
  od_post_filter4_dev24(t0, t1, t2, t3);
  p[0] = t0;
  p[step] = t1;
  p[2*step] = t2;
  p[3*step] = t3;
}

/* One half of od_postfilter_split (src/filter.c:1510-1525: rows first, then columns) of every
   level-LN node of the G tiles. */
template <int TILE, int G, int NT, int LN, bool ROWS>
__device__ __forceinline__ void walk_split_pass(int *t, unsigned tid, int xg, int y0, int pic_w, int pic_h) {
  constexpr int N = 4 << LN;
  constexpr int P = TILE + 4;
  constexpr int kTasks = TILE*(TILE/N);
  for (unsigned k = tid; k < G*kTasks; k += NT) {
    const unsigned s = k/kTasks;
    const unsigned kk = k % kTasks;
    int *ts = t + s*(TILE*P);
    if constexpr (ROWS) {
      const unsigned y = kk % TILE;
      const unsigned bx = kk/TILE;
      if (((y0 + y)/N + 1)*N <= pic_h) walk_post4(ts + y*P + bx*N + N/2 - 2, 1);
    }
    else {
      const unsigned x = kk % TILE;
      const unsigned by = kk/TILE;
      if (((xg + s*TILE + x)/N + 1)*N <= pic_w) walk_post4(ts + (by*N + N/2 - 2)*P + x, P);
    }
  }
}

template <int TILE, int G, int NT, int LEAF, int LN>
__device__ __forceinline__ void walk_split_levels(int *t, unsigned tid, int xg, int y0, int pic_w, int pic_h) {
  constexpr int TOP = TILE == 64 ? 4 : 3;
  if constexpr (LN > LEAF) {
    walk_split_pass<TILE, G, NT, LN, true>(t, tid, xg, y0, pic_w, pic_h);
    od_lds_barrier();
    walk_split_pass<TILE, G, NT, LN, false>(t, tid, xg, y0, pic_w, pic_h);
    od_lds_barrier();
  }
  if constexpr (LN < TOP) walk_split_levels<TILE, G, NT, LEAF, LN + 1>(t, tid, xg, y0, pic_w, pic_h);
}

/* Band of coding index j (OD_BAND_OFFSETS, src/partition.c:77-91) and its first index, for a block
   of level LEAF. */
template <int LEAF>
__device__ __forceinline__ int walk_band_of(int j, int &off) {
  int b = 0;
  off = 1;
  if constexpr (LEAF >= 1) {
    if (j >= 16) { b = 1; off = 16; }
    if (j >= 24) { b = 2; off = 24; }
    if (j >= 32) { b = 3; off = 32; }
  }
  if constexpr (LEAF >= 2) {
    if (j >= 64) { b = 4; off = 64; }
    if (j >= 96) { b = 5; off = 96; }
    if (j >= 128) { b = 6; off = 128; }
  }
  if constexpr (LEAF >= 3) {
    if (j >= 256) { b = 7; off = 256; }
    if (j >= 384) { b = 8; off = 384; }
  }
  return b;
}

/* Columns 0..TILE-1 of the G tiles <- 0 (the pad columns hold the keep). */
template <int TILE, int G, int NT>
__device__ __forceinline__ void walk_zero(int *t, unsigned tid) {
  constexpr int P = TILE + 4;
  for (unsigned i = tid; i < G*TILE*(TILE/4); i += NT) {
    const unsigned s = i/(TILE*(TILE/4));
    const unsigned j = i % (TILE*(TILE/4));
    *reinterpret_cast<int4 *>(t + s*(TILE*P) + (j/(TILE/4))*P + (j % (TILE/4))*4) = make_int4(0, 0, 0, 0);
  }
}

/* Source 0: dequantised coefficient planes. */
template <int TILE, int G, int NT>
__device__ __forceinline__ void walk_load_plane(int *t, const od_coeff *plane, int w, int xg, int y0, unsigned tid) {
  constexpr int P = TILE + 4;
  constexpr int Q = TILE/4;
  static_assert((G*TILE*Q) % NT == 0, "whole trips");
  constexpr int K = G*TILE*Q/NT;
  int4 v[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const unsigned i = tid + k*NT;
    const unsigned s = i/(TILE*Q);
    const unsigned j = i % (TILE*Q);
    v[k] = *reinterpret_cast<const int4 *>(plane + (long)(y0 + j/Q)*w + xg + s*TILE + (j % Q)*4);
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const unsigned i = tid + k*NT;
    const unsigned s = i/(TILE*Q);
    const unsigned j = i % (TILE*Q);
    *reinterpret_cast<int4 *>(t + s*(TILE*P) + (j/Q)*P + (j % Q)*4) = v[k];
  }
}

/* Compile-time loop: f(std::integral_constant<int, J>) for J in [J0, J1) (register arrays indexed with
   scan-table entries stay in registers only when the indices are constant expressions). */
template <int J0, int J1, class F>
__device__ __forceinline__ void inv_static_for(F &&f) {
  if constexpr (J0 < J1) {
    f(std::integral_constant<int, J0>{});
    inv_static_for<J0 + 1, J1>(f);
  }
}

/* Source 1: the no-reference band stage's choices, dequantised on load (od_pvq_synthesis_partial
   noref, src/pvq.c:1081-1092; od_coding_order_to_raster, src/partition.c:176-194): one chunk of 16
   consecutive coding indices of one block per thread and trip, as in k_inverse_sb.

   Round 5: the loads of a group are issued one group AHEAD.  The ablations of
   profiles/r5_inverse_phases.txt put half of the walking kernels' time into this load - not into its
   arithmetic (removing a quarter of the instructions of a level, or both quarter-rate multiplies of
   every coefficient, moved nothing) but into its two DEPENDENT memory latencies per group (choice
   record -> pulse vector), which a workgroup spent idle before its first barrier.  heads() (choice
   records, the DC) of group g + 1 goes out right after group g's tile has been filled, pulses() (which
   needs the records) after the leaf transforms, commit() dequantises from registers into the tile at
   the top of the next iteration: 17 VGPRs per trip. */
template <int TILE, int G, int NT, int LEAF>
struct WalkPvq {
  static constexpr int P = TILE + 4;
  static constexpr int sh = LEAF + 2;
  static constexpr int n = 4 << LEAF;
  static constexpr int len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  static constexpr int lsh = LEAF >= 3 ? 9 : 2*LEAF + 4;           /* log2(len) */
  static_assert((1 << lsh) == len, "len is a power of two");
  static constexpr int nbw = TILE >> sh;
  static constexpr int cpb = len/16;                                /* chunks per block */
  static constexpr int nch = nbw*nbw*cpb;                           /* chunks per tile */
  static constexpr int K = (G*nch + NT - 1)/NT;                     /* trips per thread */
  static constexpr int NBANDS = OD_NBANDS[LEAF];
  int4 chs[K][2];
  int4 yq[K][2];
  int dc[K];
  struct Where {
    bool on;
    unsigned s;
    unsigned j0;
    unsigned lby;
    unsigned lbx;
  };
  __device__ __forceinline__ static Where where(unsigned tid, int k) {
    const unsigned c = tid + k*NT;
    Where q;
    q.on = c < (unsigned)(G*nch);
    q.s = c/nch;
    const unsigned cc = c % nch;
    const unsigned b = cc/cpb;
    q.j0 = (cc % cpb) << 4;
    q.lby = b/nbw;
    q.lbx = b % nbw;
    return q;
  }
  __device__ __forceinline__ static unsigned block(const InverseArgs &a, int plane, int xg, int y0, const Where &q) {
    const int bw = a.w >> sh;
    const int bh = a.h >> sh;
    return (unsigned)(((long)plane*bh + (y0 >> sh))*bw + (xg >> sh)) + q.lby*(unsigned)bw + q.s*nbw + q.lbx;
  }
  /* choice records and the DC of the group at xg */
  __device__ __forceinline__ void heads(const InverseArgs &a, int plane, long plane_off, int xg, int y0, unsigned tid) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      const Where q = where(tid, k);
      chs[k][0] = chs[k][1] = make_int4(0, 0, 0, 0);
      dc[k] = 0;
      if (!q.on) continue;
      const unsigned blk = block(a, plane, xg, y0, q);
      int o0;
      int o1;
      const int bnd0 = walk_band_of<LEAF>(q.j0 ? q.j0 : 1, o0);
      const int bnd1 = walk_band_of<LEAF>(q.j0 + 8, o1);
      if (q.j0 == 0) dc[k] = a.coef[plane_off + (long)(y0 + (q.lby << sh))*a.w + xg + q.s*TILE + (q.lbx << sh)];
      chs[k][0] = a.choice[(long)blk*NBANDS + bnd0];
      if constexpr (LEAF > 0) chs[k][1] = a.choice[(long)blk*NBANDS + bnd1];
      else chs[k][1] = chs[k][0];
    }
  }
  /* ... and, once those records are here, the chosen pulse vectors */
  __device__ __forceinline__ void pulses(const InverseArgs &a, int plane, int xg, int y0, unsigned tid) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      const Where q = where(tid, k);
      yq[k][0] = yq[k][1] = make_int4(0, 0, 0, 0);
      if (!q.on) continue;
      const unsigned blk = block(a, plane, xg, y0, q);
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        if (chs[k][hf].y != 0) {
          yq[k][hf] = *reinterpret_cast<const int4 *>(a.y
           + (((unsigned)chs[k][hf].x*(unsigned)a.nblocks + blk) << lsh) + q.j0 + 8*hf);
        }
      }
    }
  }
  /* registers -> tile: the scatter to raster through the scan table (any leaf level) */
  __device__ __forceinline__ void commit(int *t, const InverseArgs &a, unsigned tid) const {
#pragma unroll
    for (int k = 0; k < K; k++) {
      const Where q = where(tid, k);
      if (!q.on) continue;
      int4 qm4[2];
      int4 sc4[2];
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        qm4[hf] = *reinterpret_cast<const int4 *>(a.qm_inv + q.j0 + 8*hf);
        sc4[hf] = *reinterpret_cast<const int4 *>(gInvScanXY + q.j0 + 8*hf);
      }
      int *ts = t + q.s*(TILE*P) + (q.lby << sh)*P + (q.lbx << sh);
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const int yd[4] = {yq[k][hf].x, yq[k][hf].y, yq[k][hf].z, yq[k][hf].w};
        const int qd[4] = {qm4[hf].x, qm4[hf].y, qm4[hf].z, qm4[hf].w};
        const int sd[4] = {sc4[hf].x, sc4[hf].y, sc4[hf].z, sc4[hf].w};
        const int4 ch = chs[k][hf];
        const int rnd = (1 << ch.w) >> 1;
        const OdQ16 sc(ch.z);
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int yv = od_pulse16((unsigned)yd[e], u);
            const int qmi = u ? qd[e] >> 16 : (int)(short)qd[e];
            const int xy = u ? (unsigned)sd[e] >> 16 : sd[e] & 0xffff;
            /* a band that codes nothing (qg == 0) left its pulses unread: zero, and (0 + rnd) >> qshift == 0 */
            int v = (__mul24(sc.mul(yv), qmi) + rnd) >> ch.w;
            if (hf == 0 && e == 0 && u == 0 && q.j0 == 0) v = dc[k];
            ts[(xy >> 8)*P + (xy & 255)] = v;
          }
        }
      }
    }
  }
  /* registers -> tile for 4x4 leaves: a chunk IS a block (one band of 15 coefficients and the DC) -
     the sixteen coding positions dequantised straight into a 4x4 register array (the scan is a
     compile-time constant: no scan table, no scatter through LDS, the inverse quantisation matrix in
     scalar registers), od_bin_idct4x4 (rows, then columns, src/dct.c:158-163) in registers, the four
     rows of the block written into the tile as 16-byte pieces.  Replaces commit(), two barriers and
     both 4-point LDS passes. */
  __device__ __forceinline__ void commit_leaf4(int *t, const InverseArgs &a, unsigned tid) const {
    static_assert(LEAF == 0 || sizeof(int) == 4, "");
    using T = OdMul24;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const Where q = where(tid, k);
      if (!q.on) continue;
      const int4 ch = chs[k][0];
      const unsigned yw[8] = {(unsigned)yq[k][0].x, (unsigned)yq[k][0].y, (unsigned)yq[k][0].z, (unsigned)yq[k][0].w,
       (unsigned)yq[k][1].x, (unsigned)yq[k][1].y, (unsigned)yq[k][1].z, (unsigned)yq[k][1].w};
      const int rnd = (1 << ch.w) >> 1;
      const OdQ16 sc(ch.z);
      int m[4][4];
      m[0][0] = dc[k];
      inv_static_for<1, 16>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        m[OD_SCAN_XY[j][1]][OD_SCAN_XY[j][0]] =
         (__mul24(sc.mul(od_pulse16(yw[j >> 1], j & 1)), (int)a.qm_inv[j]) + rnd) >> ch.w;
      });
      T qr[4][4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const T in[4] = {T(m[r][0]), T(m[r][1]), T(m[r][2]), T(m[r][3])};
        od_idct_lift<0>(qr[r], in);
      }
      T o[4][4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const T in[4] = {qr[0][c], qr[1][c], qr[2][c], qr[3][c]};
        T out[4];
        od_idct_lift<0>(out, in);
        o[0][c] = out[0];
        o[1][c] = out[1];
        o[2][c] = out[2];
        o[3][c] = out[3];
      }
      int *ts = t + q.s*(TILE*P) + (4*q.lby)*P + 4*q.lbx;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        *reinterpret_cast<int4 *>(ts + r*P) = make_int4(o[r][0], o[r][1], o[r][2], o[r][3]);
      }
    }
  }
};

/* Source 2: the with-reference band stage's choices (the per-coefficient part of
   od_pvq_synthesis_partial, src/pvq.c:1081-1114, with or without reference, skip-copy and skip-zero
   bands; inverse_load_ref above): eight consecutive coding positions of one block per thread and
   trip; a chunk never straddles a band. */
template <int TILE, int G, int NT, int LEAF>
__device__ __forceinline__ void walk_load_ref(int *t, const InverseArgs &a, int plane, long plane_off, int xg,
 int y0, unsigned tid) {
  constexpr int P = TILE + 4;
  constexpr int sh = LEAF + 2;
  constexpr int n = 4 << LEAF;
  constexpr int len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  constexpr int nbw = TILE >> sh;
  constexpr int cpb = len/8;
  constexpr int nch = nbw*nbw*cpb;
  const int bw = a.w >> sh;
  const int bh = a.h >> sh;
  if constexpr (len < n*n) {
    /* the positions PVQ never codes are what od_init_skipped_coeffs leaves (src/state.c:1347-1366):
       zero on a keyframe, the prediction's coefficients otherwise */
    if (a.inter) walk_load_plane<TILE, G, NT>(t, a.ref + plane_off, a.w, xg, y0, tid);
    else walk_zero<TILE, G, NT>(t, tid);
    od_lds_barrier();
  }
  const int4 *choice4 = a.choice;
  for (unsigned c = tid; c < G*nch; c += NT) {
    const unsigned s = c/nch;
    const unsigned cc = c % nch;
    const unsigned b = cc/cpb;
    const unsigned c0 = (cc % cpb) << 3;
    const unsigned lby = b/nbw;
    const unsigned lbx = b % nbw;
    const int x0 = xg + s*TILE;
    const long blk = ((long)plane*bh + (y0 >> sh) + lby)*bw + (x0 >> sh) + lbx;
    int off;
    const int band = walk_band_of<LEAF>(c0 ? c0 : 1, off);
    const int4 ca = choice4[(blk*a.nb_bands + band)*4 + 2];   /* mode, slot, scale, qshift */
    const int mode = ca.x;
    int *ts = t + s*(TILE*P) + (lby << sh)*P + (lbx << sh);
    const long gbase = plane_off + (long)(y0 + (lby << sh))*a.w + x0 + (lbx << sh);
    const uint4 sc4 = *reinterpret_cast<const uint4 *>(gInvScanXY + c0);
    const unsigned scw[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
    int py[8];
    int px_[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const unsigned pk = (scw[e >> 1] >> (16*(e & 1))) & 0xffffu;
      py[e] = (int)(pk >> 8);
      px_[e] = (int)(pk & 255);
    }
    const bool first = c0 == 0;
    if (first) ts[0] = a.coef[gbase];       /* the DC sits at (0, 0) */
    if (mode == 0) {
#pragma unroll
      for (int e = 0; e < 8; e++) if (!(first && e == 0)) ts[py[e]*P + px_[e]] = 0;
      continue;
    }
    if (mode == 1 || mode == 4) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (first && e == 0) continue;
        const od_coeff rv = a.ref[gbase + (long)py[e]*a.w + px_[e]];
        ts[py[e]*P + px_[e]] = mode == 4 ? -rv : rv;
      }
      continue;
    }
    const int yslot = ca.y;
    const int32_t scale = ca.z;
    const int qshift = ca.w;
    const int rnd = (1 << qshift) >> 1;
    const uint4 q4 = *reinterpret_cast<const uint4 *>(a.qm_inv + c0);
    const unsigned qw[4] = {q4.x, q4.y, q4.z, q4.w};
    unsigned yw[4] = {0, 0, 0, 0};
    int yprev = 0;
    if (yslot >= 0) {
      const int16_t *yp = a.y + ((long)yslot*a.nblocks + blk)*len + c0;
      const uint4 y4 = *reinterpret_cast<const uint4 *>(yp);
      yw[0] = y4.x;
      yw[1] = y4.y;
      yw[2] = y4.z;
      yw[3] = y4.w;
      if (mode == 3 && c0 > off) yprev = yp[-1];
    }
    const OdQ16 sc(scale);
    if (mode == 2) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (first && e == 0) continue;
        /* OD_MULT16_32_Q16 (OdQ16); OD_SHR_ROUND in the reference's 32 bits */
        const int yv = od_pulse16(yw[e >> 1], e & 1);
        const int qmi = (int16_t)(qw[e >> 1] >> (16*(e & 1)));
        ts[py[e]*P + px_[e]] = (__mul24(sc.mul(yv), qmi) + rnd) >> qshift;
      }
      continue;
    }
    const int4 cb = choice4[(blk*a.nb_bands + band)*4 + 3];   /* xm, m, proj_1, outshift */
    const int m = cb.y;
    const uint4 r4 = *reinterpret_cast<const uint4 *>(a.r16 + blk*len + c0);
    const unsigned rw[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (first && e == 0) continue;
      const int i = c0 + e - off;
      const int yv = od_pulse16(yw[e >> 1], e & 1);
      const int ym1 = e == 0 ? yprev : od_pulse16(yw[(e - 1) >> 1], (e - 1) & 1);
      const int qmi = (int16_t)(qw[e >> 1] >> (16*(e & 1)));
      const int ri = (int16_t)(rw[e >> 1] >> (16*(e & 1)));
      const int16_t xi = i == m ? (int16_t)cb.x : (int16_t)sc.mul(i < m ? yv : ym1);
      int32_t tmp = ri*(int)(int16_t)cb.z;                    /* OD_MULT16_16(r[i], proj_1) */
      tmp = cb.w >= 0 ? (tmp + ((1 << cb.w) >> 1)) >> cb.w : (int32_t)((uint32_t)tmp << -cb.w);
      const int16_t v = (int16_t)(xi - tmp);
      ts[py[e]*P + px_[e]] = (v*qmi + rnd) >> qshift;
    }
  }
}

/* ---- leaf level 0 in registers, with-reference source (round 5) ------------------------------
   (The no-reference source: WalkPvq::commit_leaf4 above.)
   A 4x4 leaf block is ONE band of 15 coefficients plus its DC: one lane takes the whole block -
   its choice record(s), its 32 bytes of pulses (and of the reference, with-reference source), the DC
   - dequantises the sixteen coding positions straight into a 4x4 register array (the scan is a
   compile-time constant: no scan table, no coding-order -> raster scatter through LDS, the inverse
   quantisation matrix in scalar registers), runs od_bin_idct4x4 (rows, then columns,
   src/dct.c:158-163) on it and writes the four rows of the block into the tile as 16-byte pieces.
   That replaces the dequantise-on-load of walk_load_pvq / walk_load_ref, two barriers and both
   4-point LDS passes (12 of whose 20 instructions per network were LDS addressing and loop control,
   profiles/r4_inverse_budget.txt).  Same arithmetic, same results. */
template <int TILE, int G, int NT, int SRC>
__device__ __forceinline__ void walk_leaf4(int *t, const InverseArgs &a, int plane, long plane_off, int xg,
 int y0, unsigned tid) {
  using T = OdMul24;
  constexpr int P = TILE + 4;
  constexpr int NB = TILE/4;
  static_assert(SRC == 2, "the with-reference source (the no-reference one: WalkPvq::commit_leaf4)");
  const int bw = a.w >> 2;
  const int bh = a.h >> 2;
  for (unsigned k = tid; k < G*NB*NB; k += NT) {
    const unsigned s = k/(NB*NB);
    const unsigned kk = k % (NB*NB);
    const unsigned lby = kk/NB;
    const unsigned lbx = kk % NB;
    const int x0 = xg + s*TILE;
    const unsigned blk = (unsigned)(((long)plane*bh + (y0 >> 2) + lby)*bw + (x0 >> 2) + lbx);
    const long gbase = plane_off + (long)(y0 + 4*lby)*a.w + x0 + 4*lbx;
    int m[4][4];
    const int dc = a.coef[gbase];
    {
      const int4 *choice4 = a.choice;
      const int4 ca = choice4[(long)blk*4 + 2];          /* mode, slot, scale, qshift (nb_bands == 1) */
      const int mode = ca.x;
      m[0][0] = dc;
      if (mode == 0) {
        inv_static_for<1, 16>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          m[OD_SCAN_XY[j][1]][OD_SCAN_XY[j][0]] = 0;
        });
      }
      else if (mode == 1 || mode == 4) {
        inv_static_for<1, 16>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const od_coeff rv = a.ref[gbase + (long)OD_SCAN_XY[j][1]*a.w + OD_SCAN_XY[j][0]];
          m[OD_SCAN_XY[j][1]][OD_SCAN_XY[j][0]] = mode == 4 ? -rv : rv;
        });
      }
      else {
        const int yslot = ca.y;
        const int32_t scale = ca.z;
        const int qshift = ca.w;
        const int rnd = (1 << qshift) >> 1;
        uint4 yq[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (yslot >= 0) {
          const int16_t *yp = a.y + (((long)yslot*a.nblocks + blk) << 4);
          yq[0] = *reinterpret_cast<const uint4 *>(yp);
          yq[1] = *reinterpret_cast<const uint4 *>(yp + 8);
        }
        const unsigned yw[8] = {yq[0].x, yq[0].y, yq[0].z, yq[0].w, yq[1].x, yq[1].y, yq[1].z, yq[1].w};
        const OdQ16 sc(scale);
        if (mode == 2) {
          inv_static_for<1, 16>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            m[OD_SCAN_XY[j][1]][OD_SCAN_XY[j][0]] =
             (__mul24(sc.mul(od_pulse16(yw[j >> 1], j & 1)), (int)a.qm_inv[j]) + rnd) >> qshift;
          });
        }
        else {
          const int4 cb = choice4[(long)blk*4 + 3];      /* xm, m, proj_1, outshift */
          const int mm = cb.y;
          const int16_t *rp = a.r16 + ((long)blk << 4);
          const uint4 r0 = *reinterpret_cast<const uint4 *>(rp);
          const uint4 r1 = *reinterpret_cast<const uint4 *>(rp + 8);
          const unsigned rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
          const int proj = (int)(int16_t)cb.z;
          inv_static_for<1, 16>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int i = j - 1;                       /* the band starts at coding index 1 */
            const int yv = od_pulse16(yw[j >> 1], j & 1);
            /* the pulse before: y[j - 1] (position 0 of the vector, the DC's unused slot, for j = 1:
               only selected when i > m, which i = 0 never is) */
            const int ym1 = od_pulse16(yw[(j - 1) >> 1], (j - 1) & 1);
            const int ri = (int16_t)(rw[j >> 1] >> (16*(j & 1)));
            const int16_t xi = i == mm ? (int16_t)cb.x : (int16_t)sc.mul(i < mm ? yv : ym1);
            int32_t tmp = ri*proj;                         /* OD_MULT16_16(r[i], proj_1) */
            tmp = cb.w >= 0 ? (tmp + ((1 << cb.w) >> 1)) >> cb.w : (int32_t)((uint32_t)tmp << -cb.w);
            const int16_t v = (int16_t)(xi - tmp);
            m[OD_SCAN_XY[j][1]][OD_SCAN_XY[j][0]] = (v*(int)a.qm_inv[j] + rnd) >> qshift;
          });
        }
      }
    }
    /* od_bin_idct4x4: rows, then columns */
    T q[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const T in[4] = {T(m[r][0]), T(m[r][1]), T(m[r][2]), T(m[r][3])};
      od_idct_lift<0>(q[r], in);
    }
    int *ts = t + s*(TILE*P) + (4*lby)*P + 4*lbx;
    T o[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const T in[4] = {q[0][c], q[1][c], q[2][c], q[3][c]};
      T out[4];
      od_idct_lift<0>(out, in);
      o[0][c] = out[0];
      o[1][c] = out[1];
      o[2][c] = out[2];
      o[3][c] = out[3];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      *reinterpret_cast<int4 *>(ts + r*P) = make_int4(o[r][0], o[r][1], o[r][2], o[r][3]);
    }
  }
}

/* The group's pixel rows and horizontal-edge strips, in a window shifted four samples to the
   left: [xg - 4, xg + G*TILE - 4) - the keep columns first (not for the first group of a segment),
   the last four columns only when the segment ends with this group - and the keep renewed. */
template <int TILE, int G, int NT>
__device__ __forceinline__ void walk_store(int *t, const InverseArgs &a, int plane, int xg, int y0, int sby,
 bool first, bool last, unsigned tid) {
  constexpr int P = TILE + 4;
  constexpr int VG = G*TILE/4;
  static_assert((TILE*VG) % NT == 0, "whole trips");
  constexpr int K = TILE*VG/NT;
  const bool px16 = a.px16 != 0;
  uint8_t *px = a.px + ((long)plane*a.px_plane_stride << a.px16);
  const int nh = a.h/TILE - 1;
  od_coeff *hs = a.hs + (long)plane*nh*4*a.w;
  auto put = [&](int r, int x, int4 v) {
    const long at = (long)(y0 + r)*a.px_stride + x;
    if (OD_INV_DBG(a, 1)) {
    }
    else if (px16) {
      *reinterpret_cast<short4 *>(reinterpret_cast<short *>(px) + at) =
       make_short4(od_to_px16(v.x), od_to_px16(v.y), od_to_px16(v.z), od_to_px16(v.w));
    }
    else {
      uchar4 o;
      o.x = od_to_px(v.x);
      o.y = od_to_px(v.y);
      o.z = od_to_px(v.z);
      o.w = od_to_px(v.w);
      *reinterpret_cast<uchar4 *>(px + at) = o;
    }
    /* rows 0, 1 / TILE-2, TILE-1 also feed the post-filter across the horizontal superblock edge
       above / below (k_edge_cols) */
    if (OD_INV_DBG(a, 2)) {
    }
    else if (r < 2) {
      if (sby > 0) *reinterpret_cast<int4 *>(hs + ((long)(sby - 1)*4 + r + 2)*a.w + x) = v;
    }
    else if (r >= TILE - 2) {
      if (sby < nh) *reinterpret_cast<int4 *>(hs + ((long)sby*4 + r - (TILE - 2))*a.w + x) = v;
    }
  };
  /* in batches of four 16-byte pieces per lane: every LDS read of a batch in flight before its
     first store */
  constexpr int KB = K < 4 ? K : 4;
  static_assert(K % KB == 0, "whole batches");
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += KB) {
    int4 tv[KB];
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const unsigned i = tid + (k0 + k)*NT;
      const unsigned r = i/VG;
      const unsigned vg = i % VG;
      const unsigned c = 4*vg - 4;
      const int *src = vg == 0 ? t + r*P + TILE : t + (c/TILE)*(TILE*P) + r*P + c % TILE;
      tv[k] = *reinterpret_cast<const int4 *>(src);
    }
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const unsigned i = tid + (k0 + k)*NT;
      const unsigned r = i/VG;
      if (i % VG == 0) {                      /* this lane read the old keep of row r: renew it */
        *reinterpret_cast<int4 *>(t + r*P + TILE) =
         *reinterpret_cast<const int4 *>(t + (G - 1)*(TILE*P) + r*P + TILE - 4);
      }
    }
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const unsigned i = tid + (k0 + k)*NT;
      const unsigned r = i/VG;
      const unsigned vg = i % VG;
      if (vg == 0 && first) continue;
      put(r, xg - 4 + 4*vg, tv[k]);
    }
  }
  if (last) {
    for (unsigned r = tid; r < TILE; r += NT) {
      put(r, xg + G*TILE - 4, *reinterpret_cast<const int4 *>(t + (G - 1)*(TILE*P) + r*P + TILE - 4));
    }
  }
}

/* ---- the same window, one group LATE and line by line ------------------------------------
   With 8-bit samples a row of a 64-wide tile is exactly one 64-byte line of the plane: the window
   shifted by four samples (walk_store) writes every line with TWO store instructions of
   consecutive groups, which the memory system does not merge (26 us of the luma stage,
   profiles/r4_inverse_segments.txt).  Instead the finished group is packed to 8-bit samples in
   LDS (walk_pack), and its rows go out when the NEXT group has filtered the shared edge
   (walk_store_prev): whole 16-byte pieces, every line written once, the last four samples of a
   row converted from the keep.  Used when the pixel plane is 16-byte aligned and holds 8-bit
   samples (InverseArgs::wide); full-precision references keep walk_store. */
__device__ __forceinline__ uint32_t walk_pack4(int4 v) {
  return (uint32_t)od_to_px(v.x) | (uint32_t)od_to_px(v.y) << 8 | (uint32_t)od_to_px(v.z) << 16
   | (uint32_t)od_to_px(v.w) << 24;
}

/* The group at t -> packed samples in pxb ([tile][row][TILE/4] dwords), its strip rows (all but the
   last four columns, which wait for the edge filter) -> hs, and the keep renewed. */
template <int TILE, int G, int NT>
__device__ __forceinline__ void walk_pack(int *t, uint32_t *pxb, const InverseArgs &a, int plane, int xg, int y0,
 int sby, unsigned tid) {
  constexpr int P = TILE + 4;
  constexpr int Q = TILE/4;
  const int nh = a.h/TILE - 1;
  od_coeff *hs = a.hs + (long)plane*nh*4*a.w;
  for (unsigned i = tid; i < G*TILE*Q; i += NT) {
    const unsigned s = i/(TILE*Q);
    const unsigned j = i % (TILE*Q);
    const unsigned r = j/Q;
    const unsigned q = j % Q;
    const int4 v = *reinterpret_cast<const int4 *>(t + s*(TILE*P) + r*P + 4*q);
    pxb[i] = walk_pack4(v);
    if (!OD_INV_DBG(a, 2) && !(s == G - 1 && q == Q - 1)) {
      const int x = xg + s*TILE + 4*q;
      if (r < 2) {
        if (sby > 0) *reinterpret_cast<int4 *>(hs + ((long)(sby - 1)*4 + r + 2)*a.w + x) = v;
      }
      else if (r >= TILE - 2) {
        if (sby < nh) *reinterpret_cast<int4 *>(hs + ((long)sby*4 + r - (TILE - 2))*a.w + x) = v;
      }
    }
  }
  for (unsigned r = tid; r < TILE; r += NT) {
    *reinterpret_cast<int4 *>(t + r*P + TILE) = *reinterpret_cast<const int4 *>(t + (G - 1)*(TILE*P) + r*P + TILE - 4);
  }
}

/* The rows of the group whose packed samples are in pxb and whose last four columns are in the
   keep, at plane position (xp, y0): 16-byte pieces, whole lines. */
template <int TILE, int G, int NT>
__device__ __forceinline__ void walk_store_prev(const int *t, const uint32_t *pxb, const InverseArgs &a, int plane,
 int xp, int y0, int sby, unsigned tid) {
  constexpr int P = TILE + 4;
  constexpr int Q = TILE/4;
  constexpr int PR = G*TILE/16;            /* 16-byte pieces per row of the group */
  const int nh = a.h/TILE - 1;
  od_coeff *hs = a.hs + (long)plane*nh*4*a.w;
  uint8_t *px = a.px + (long)plane*a.px_plane_stride;
  for (unsigned i = tid; i < TILE*PR; i += NT) {
    const unsigned r = i/PR;
    const unsigned q = i % PR;
    const unsigned d = 4*q;                /* first dword of the piece in the group's row */
    const unsigned s = d/Q;
    uint4 v = *reinterpret_cast<const uint4 *>(pxb + s*(TILE*Q) + r*Q + d % Q);
    if (q == PR - 1) {
      const int4 k = *reinterpret_cast<const int4 *>(t + r*P + TILE);
      v.w = walk_pack4(k);
      if (!OD_INV_DBG(a, 2)) {
        const int x = xp + G*TILE - 4;
        if (r < 2) {
          if (sby > 0) *reinterpret_cast<int4 *>(hs + ((long)(sby - 1)*4 + r + 2)*a.w + x) = k;
        }
        else if (r >= TILE - 2) {
          if (sby < nh) *reinterpret_cast<int4 *>(hs + ((long)sby*4 + r - (TILE - 2))*a.w + x) = k;
        }
      }
    }
    if (!OD_INV_DBG(a, 1)) *reinterpret_cast<uint4 *>(px + (long)(y0 + r)*a.px_stride + xp + 16*q) = v;
  }
}

template <int TILE, int G, int NT, int SRC, int LEAF>
__device__ __forceinline__ void inverse_walk(int *t, uint32_t *pxb, const InverseArgs &a, int plane, int seg_len,
 unsigned tid_in) {
  unsigned tid = tid_in;
  constexpr int P = TILE + 4;
  const int ng = a.w/(TILE*G);
  const int g0 = blockIdx.x*seg_len;
  const int g1 = min(g0 + seg_len, ng);
  const int sby = blockIdx.y;
  const int y0 = sby*TILE;
  const long plane_off = (long)plane*a.w*a.h;
  const int nv = a.w/TILE - 1;
  od_coeff *vs = a.vs + (long)plane*nv*a.h*4;
  /* no-reference source: the loads of a group go out one group ahead (WalkPvq) */
  using Fetch = WalkPvq<TILE, G, NT, LEAF>;
  [[maybe_unused]] Fetch f;
  const bool ahead = SRC == 1 && !OD_INV_DBG(a, 4) && !OD_INV_DBG(a, 128);
  if constexpr (SRC == 1) {
    if (ahead && g0 < g1) {
      f.heads(a, plane, plane_off, g0*G*TILE, y0, tid);
      f.pulses(a, plane, g0*G*TILE, y0, tid);
    }
  }
  for (int g = g0; g < g1; g++) {
    const int xg = g*G*TILE;
    /* per-lane index arithmetic is redone for every group (as one workgroup per superblock did):
       hoisted out of this loop it occupies ~50 VGPRs and halves the occupancy */
    asm volatile("" : "+v"(tid));
    /* 4x4 leaves of a pulse-fed source: dequantisation and both 4-point passes in registers, one
       block per lane (WalkPvq::commit_leaf4 / walk_leaf4; OD_INV_DBG bit 8, experiments build: the
       LDS passes of round 4) */
    const bool leaf4 = LEAF == 0 && SRC != 0 && !OD_INV_DBG(a, 4) && !OD_INV_DBG(a, 8);
    if constexpr (SRC == 1) {
      if (!OD_INV_DBG(a, 4)) {
        if (!ahead) {                      /* (experiments build, bit 128: the loads where round 4 had them) */
          f.heads(a, plane, plane_off, xg, y0, tid);
          f.pulses(a, plane, xg, y0, tid);
        }
        if constexpr (Fetch::len < Fetch::n*Fetch::n) {   /* 32x32 / 64x64: uncoded positions are zero */
          walk_zero<TILE, G, NT>(t, tid);
          od_lds_barrier();
        }
        if (leaf4) f.commit_leaf4(t, a, tid);
        else f.commit(t, a, tid);
      }
      else walk_zero<TILE, G, NT>(t, tid);
      od_lds_barrier();
      if (ahead && g + 1 < g1) f.heads(a, plane, plane_off, xg + G*TILE, y0, tid);
    }
    else if (leaf4) {
      if constexpr (LEAF == 0 && SRC == 2) walk_leaf4<TILE, G, NT, SRC>(t, a, plane, plane_off, xg, y0, tid);
      od_lds_barrier();
    }
    else {
      if (OD_INV_DBG(a, 4)) walk_zero<TILE, G, NT>(t, tid);
      else if constexpr (SRC == 0) walk_load_plane<TILE, G, NT>(t, a.coef + plane_off, a.w, xg, y0, tid);
      else if constexpr (SRC == 2) walk_load_ref<TILE, G, NT, LEAF>(t, a, plane, plane_off, xg, y0, tid);
      od_lds_barrier();
    }
    if (!leaf4 && !OD_INV_DBG(a, 16)) {           /* (bit 16, experiments build: timing ablation, WRONG results) */
      walk_idct_pass<TILE, G, NT, LEAF, true>(t, tid);
      od_lds_barrier();
      walk_idct_pass<TILE, G, NT, LEAF, false>(t, tid);
      od_lds_barrier();
    }
    if constexpr (SRC == 1) {
      /* the choice records of the next group have had the leaf transforms to arrive */
      if (ahead && g + 1 < g1) f.pulses(a, plane, xg + G*TILE, y0, tid);
    }
    if (!OD_INV_DBG(a, 32)) walk_split_levels<TILE, G, NT, LEAF, 1>(t, tid, xg, y0, a.pic_w, a.pic_h);
    /* the vertical superblock edges of the group: left of every tile (against the keep for the
       first tile; a segment's first group hands its columns 0..1 to the strips instead) and, at a
       segment's end inside the plane, columns TILE-2..TILE-1 of the last tile to the strips */
    for (unsigned k = tid; k < G*TILE; k += NT) {
      const unsigned s = k/TILE;
      const unsigned r = k % TILE;
      int *row = t + s*(TILE*P) + r*P;
      if (s == 0 && g == g0) {
        if (xg > 0) {
          *reinterpret_cast<int2 *>(vs + ((long)(g*G - 1)*a.h + y0 + r)*4 + 2) = make_int2(row[0], row[1]);
        }
        continue;
      }
      int *left = s == 0 ? row + TILE + 2 : row - TILE*P + TILE - 2;
      int t0 = left[0];
      int t1 = left[1];
      int t2 = row[0];
      int t3 = row[1];
      od_post_filter4_dev24(t0, t1, t2, t3);
      left[0] = t0;
      left[1] = t1;
      row[0] = t2;
      row[1] = t3;
    }
    if (g == g1 - 1 && g1 < ng) {
      for (unsigned r = tid; r < TILE; r += NT) {
        const int *row = t + (G - 1)*(TILE*P) + r*P;
        *reinterpret_cast<int2 *>(vs + ((long)(g1*G - 1)*a.h + y0 + r)*4) = make_int2(row[TILE - 2], row[TILE - 1]);
      }
    }
    od_lds_barrier();
    if (OD_INV_DBG(a, 64)) {
    }
    else if (a.wide) {
      if (g != g0) walk_store_prev<TILE, G, NT>(t, pxb, a, plane, xg - G*TILE, y0, sby, tid);
      od_lds_barrier();
      walk_pack<TILE, G, NT>(t, pxb, a, plane, xg, y0, sby, tid);
      od_lds_barrier();
      if (g == g1 - 1) walk_store_prev<TILE, G, NT>(t, pxb, a, plane, xg, y0, sby, tid);
    }
    else {
      walk_store<TILE, G, NT>(t, a, plane, xg, y0, sby, g == g0, g == g1 - 1, tid);
      od_lds_barrier();
    }
  }
}

/* SRC: 0 = coefficient planes, 1 = no-reference band stage (odhip_inverse_levels_pvq), 2 =
   with-reference band stage (odhip_inverse_levels_pvq_ref).  MINLEAF..MAXLEAF: the leaf levels this
   instance is launched for (the register budget is that of its largest network). */
template <int TILE, int G, int NT, int SRC, int MINLEAF, int MAXLEAF>
__global__ __launch_bounds__(NT, OD_WALK_WAVES) void k_inverse_walk(InverseWalkArgs mm) {
  constexpr int P = TILE + 4;
  __shared__ __attribute__((aligned(16))) int t[G*TILE*P];
  __shared__ __attribute__((aligned(16))) uint32_t pxb[G*TILE*TILE/4];
  const unsigned tid = threadIdx.x;
  const int plane = blockIdx.z % mm.nplanes;
  const InverseArgs &a = mm.a[blockIdx.z / mm.nplanes];
  switch (a.leaf_bs) {
    case 0: if constexpr (MINLEAF <= 0 && MAXLEAF >= 0) inverse_walk<TILE, G, NT, SRC, 0>(t, pxb, a, plane, mm.seg_len, tid); break;
    case 1: if constexpr (MINLEAF <= 1 && MAXLEAF >= 1) inverse_walk<TILE, G, NT, SRC, 1>(t, pxb, a, plane, mm.seg_len, tid); break;
    case 2: if constexpr (MINLEAF <= 2 && MAXLEAF >= 2) inverse_walk<TILE, G, NT, SRC, 2>(t, pxb, a, plane, mm.seg_len, tid); break;
    case 3: if constexpr (MINLEAF <= 3 && MAXLEAF >= 3) inverse_walk<TILE, G, NT, SRC, 3>(t, pxb, a, plane, mm.seg_len, tid); break;
    default:
      if constexpr (TILE == 64 && MAXLEAF >= 4) inverse_walk<TILE, G, NT, SRC, 4>(t, pxb, a, plane, mm.seg_len, tid);
      break;
  }
}

struct EdgeArgs {
  od_coeff *vs;
  od_coeff *hs;
  uint8_t *px;
  int px_stride;
  long px_plane_stride;
  int w;
  int h;
  int tile;
  int px16;
  /* the vertical edges k_edge_rows finishes: edge0 + i*edge_step, i < nedges (all of them behind
     k_inverse_sb / k_inverse_part, the joints between segments behind k_inverse_walk) */
  int edge0;
  int edge_step;
  int nedges;
};

struct EdgeArgsMulti {
  EdgeArgs a[kMaxInvLevels];
  int nplanes;
};

/* od_apply_postfilter_frame_sbs, first half (src/filter.c:1600-1606): row taps
   across every interior vertical superblock edge, every row.  Rows that also lie
   in a horizontal strip hand their result on to k_edge_cols through hs. */
__global__ __launch_bounds__(256) void k_edge_rows(EdgeArgsMulti mm) {
  const EdgeArgs &a = mm.a[blockIdx.z / mm.nplanes];
  const int plane = blockIdx.z % mm.nplanes;
  const int y = blockIdx.x*256 + threadIdx.x;
  if (y >= a.h || (int)blockIdx.y >= a.nedges) return;
  const int e = a.edge0 + blockIdx.y*a.edge_step;
  const int nv = a.w/a.tile - 1;
  const int nh = a.h/a.tile - 1;
  const int4 v = *reinterpret_cast<const int4 *>(a.vs + (((long)plane*nv + e)*a.h + y)*4);
  int t0 = v.x;
  int t1 = v.y;
  int t2 = v.z;
  int t3 = v.w;
  od_post_filter4_dev(t0, t1, t2, t3);
  const int x = (e + 1)*a.tile - 2;
  const int m = (y + 2) % a.tile;           /* < 4 inside a horizontal strip */
  const int he = (y + 2)/a.tile - 1;
  if (m < 4 && he >= 0 && he < nh) {
    od_coeff *row = a.hs + (((long)plane*nh + he)*4 + m)*a.w + x;
    row[0] = t0;
    row[1] = t1;
    row[2] = t2;
    row[3] = t3;
  }
  else {
    uint8_t *pl = a.px + ((long)plane*a.px_plane_stride << a.px16);
    const long at = (long)y*a.px_stride + x;
    const bool px16 = a.px16 != 0;
    if (!px16 && !((a.px_stride | a.px_plane_stride | (long)(uintptr_t)a.px) & 1)) {
      /* x = (e + 1)*tile - 2 is even: the four samples as two aligned 16-bit stores */
      unsigned short *p2 = reinterpret_cast<unsigned short *>(pl + at);
      p2[0] = (unsigned short)(od_to_px(t0) | od_to_px(t1) << 8);
      p2[1] = (unsigned short)(od_to_px(t2) | od_to_px(t3) << 8);
    }
    else {
      od_store_px(pl, at, t0, px16);
      od_store_px(pl, at + 1, t1, px16);
      od_store_px(pl, at + 2, t2, px16);
      od_store_px(pl, at + 3, t3, px16);
    }
  }
}

/* Second half (src/filter.c:1607-1617): column taps across every interior
   horizontal edge, every column, then od_coeff_to_ref_buf. */
__global__ __launch_bounds__(256) void k_edge_cols(EdgeArgsMulti mm) {
  const EdgeArgs &a = mm.a[blockIdx.z / mm.nplanes];
  const int plane = blockIdx.z % mm.nplanes;
  const int x = blockIdx.x*256 + threadIdx.x;
  if (x >= a.w) return;
  const int e = blockIdx.y;
  const int nh = a.h/a.tile - 1;
  const od_coeff *col = a.hs + ((long)plane*nh + e)*4*a.w + x;
  int t0 = col[0];
  int t1 = col[a.w];
  int t2 = col[2*a.w];
  int t3 = col[3*a.w];
  od_post_filter4_dev(t0, t1, t2, t3);
  uint8_t *pl = a.px + ((long)plane*a.px_plane_stride << a.px16);
  const long at = (long)((e + 1)*a.tile - 2)*a.px_stride + x;
  const bool px16 = a.px16 != 0;
  od_store_px(pl, at, t0, px16);
  od_store_px(pl, at + a.px_stride, t1, px16);
  od_store_px(pl, at + 2*a.px_stride, t2, px16);
  od_store_px(pl, at + 3*a.px_stride, t3, px16);
}

/* Scratch for the edge strips between k_inverse_sb and k_edge_rows / k_edge_cols:
   owned by the calling thread's current context (od_ctx.cuh), grown on demand.
   Two inverse calls in flight at once (e.g. the luma and the chroma chain of a
   step on two streams) must use two contexts. */
struct LappedState {
  od_coeff *strips = nullptr;
  size_t bytes = 0;
  ~LappedState() {
    if (strips) (void)hipFree(strips);
  }
};

}  // namespace

extern "C" int odhip_forward_pyramid(od_coeff *const d_levels[ODHIP_NBSIZES],
 const uint8_t *d_px, int px_stride, long px_plane_stride, int nplanes, int w,
 int h, int dec, int pic_w, int pic_h, odhip_stream stream) {
  if (!d_levels || !d_px || nplanes <= 0 || (dec != 0 && dec != 1)) return ODHIP_EINVAL;
  const int tile = 64 >> dec;
  if (w <= 0 || h <= 0 || w % tile || h % tile || (px_stride & 3)
   || (px_plane_stride & 3)) {
    return ODHIP_EINVAL;
  }
  PyramidArgs a;
  for (int i = 0; i < ODHIP_NBSIZES; i++) a.levels[i] = i <= 4 - dec ? d_levels[i] : nullptr;
  a.px = d_px;
  a.px_stride = px_stride;
  a.px_plane_stride = px_plane_stride;
  a.w = w;
  a.h = h;
  a.pic_w = pic_w;
  a.pic_h = pic_h;
  {
    ODHIP_CTX_OR_RETURN(ctx);
    a.px16 = ctx->fpr != 0;
    if (a.px16 && ((uintptr_t)d_px & 7)) return ODHIP_EINVAL;     /* 8-byte sample groups */
  }
  const dim3 grid(w/tile, h/tile, nplanes);
  hipStream_t s = (hipStream_t)stream;
#ifdef ODHIP_EXPERIMENTS
  /* ODHIP_PYR_VARIANT selects the kernel shapes round 2 measured against each other
     (tools/pyr_variants.py; all bit-identical): bit 1 = OD_DCT_RSHIFT in two
     instructions (OdMul24S; default), bit 0 = one luma superblock per 128-thread
     workgroup instead of two per 256, bit 2 = the same with wave-local levels below
     64 points (k_forward_pyramid_halves).  16 frames of 1080p: 2 -> 239 us,
     0 -> 243, 6 -> 245, 3 -> 253, 4 -> 258, 1 -> 259. */
  static const int variant = getenv("ODHIP_PYR_VARIANT") ? atoi(getenv("ODHIP_PYR_VARIANT")) : 2;
  /* ODHIP_PYR_LDS_PAD: extra dynamic LDS bytes per workgroup of the luma kernel, to measure how the
     time follows occupancy (tools/pyr_stalls.py); 0 outside that experiment. */
  static const unsigned lds_pad = getenv("ODHIP_PYR_LDS_PAD") ? (unsigned)atoi(getenv("ODHIP_PYR_LDS_PAD")) : 0;
  static const bool x1 = getenv("ODHIP_PYRAMID_X1") != nullptr;      /* one superblock per workgroup (A/B baseline) */
  if (variant != 2 || lds_pad || x1) {
    if (dec) {
      if ((w/tile) % 2 == 0 && !x1) {
        if (variant & 2) k_forward_pyramid32x2<OdMul24S><<<dim3(w/(2*tile), h/tile, nplanes), 128, 0, s>>>(a);
        else k_forward_pyramid32x2<OdMul24><<<dim3(w/(2*tile), h/tile, nplanes), 128, 0, s>>>(a);
      }
      else if (variant & 2) k_forward_pyramid<32, OdMul24S><<<grid, Geo<32>::kNT, 0, s>>>(a);
      else k_forward_pyramid<32><<<grid, Geo<32>::kNT, 0, s>>>(a);
    }
    else if (variant & 4) {
      if (variant & 2) k_forward_pyramid_halves<OdMul24S><<<grid, 128, 0, s>>>(a);
      else k_forward_pyramid_halves<OdMul24><<<grid, 128, 0, s>>>(a);
    }
    else if (variant & 1) {
      if (variant & 2) k_forward_pyramid<64, OdMul24S, 128><<<grid, 128, 0, s>>>(a);
      else k_forward_pyramid<64, OdMul24, 128><<<grid, 128, 0, s>>>(a);
    }
    else if ((w/tile) % 2 == 0 && !x1) {
      if (variant & 2) k_forward_pyramid64x2<OdMul24S><<<dim3(w/(2*tile), h/tile, nplanes), 256, lds_pad, s>>>(a);
      else k_forward_pyramid64x2<OdMul24><<<dim3(w/(2*tile), h/tile, nplanes), 256, 0, s>>>(a);
    }
    else if (variant & 2) k_forward_pyramid<64, OdMul24S><<<grid, Geo<64>::kNT, 0, s>>>(a);
    else k_forward_pyramid<64><<<grid, Geo<64>::kNT, 0, s>>>(a);
    return odhip_check_launch();
  }
#endif
  /* Pairs of superblocks per workgroup (k_forward_pyramid64x2: 256 threads per two 64x64 luma tiles;
     k_forward_pyramid32x2: 128 threads per two 32x32 chroma tiles) whenever the plane is an even
     number of them wide, one superblock per workgroup otherwise; OD_DCT_RSHIFT in two instructions
     (OdMul24S). */
  if (dec) {
    if ((w/tile) % 2 == 0) k_forward_pyramid32x2<OdMul24S><<<dim3(w/(2*tile), h/tile, nplanes), 128, 0, s>>>(a);
    else k_forward_pyramid<32, OdMul24S><<<grid, Geo<32>::kNT, 0, s>>>(a);
  }
  else if ((w/tile) % 2 == 0) k_forward_pyramid64x2<OdMul24S><<<dim3(w/(2*tile), h/tile, nplanes), 256, 0, s>>>(a);
  else k_forward_pyramid<64, OdMul24S><<<grid, Geo<64>::kNT, 0, s>>>(a);
  return odhip_check_launch();
}

namespace {

/* All levels share w, h, nplanes and dec (one plane set). */
int inverse_launch(const InverseArgs *levels, int nlevels, int nplanes, int dec, hipStream_t s,
 bool ref = false) {
  if (nlevels <= 0 || nlevels > kMaxInvLevels) return ODHIP_EINVAL;
  const int tile = 64 >> dec;
  const int w = levels[0].w;
  const int h = levels[0].h;
  const int nv = w/tile - 1;
  const int nh = h/tile - 1;
  const size_t vs_words = ((size_t)nplanes*nv*h*4 + 3) & ~(size_t)3;
  const size_t hs_words = ((size_t)nplanes*nh*4*w + 3) & ~(size_t)3;
  const size_t need = ((vs_words + hs_words)*nlevels + 4)*sizeof(od_coeff);
  ODHIP_CTX_OR_RETURN(ctx);
  LappedState &st = *odhip_ctx_state<LappedState>(ctx, ODHIP_SLOT_LAPPED);
  if (need > st.bytes) {
    ODHIP_TRY(hipStreamSynchronize(s));
    if (st.strips) ODHIP_TRY(hipFree(st.strips));
    st.strips = nullptr;
    st.bytes = 0;
    ODHIP_TRY(hipMalloc((void **)&st.strips, need));
    st.bytes = need;
  }
  od_coeff *const g_strips = st.strips;
  InverseArgsMulti im;
  EdgeArgsMulti em;
  memset(&im, 0, sizeof(im));
  memset(&em, 0, sizeof(em));
  im.nplanes = nplanes;
  em.nplanes = nplanes;
  for (int l = 0; l < nlevels; l++) {
    if (levels[l].w != w || levels[l].h != h) return ODHIP_EINVAL;
    im.a[l] = levels[l];
#ifdef ODHIP_EXPERIMENTS
    /* timing ablations of k_inverse_walk (WRONG results by design): experiments build only - a stray
       environment variable cannot corrupt a reconstruction of the default build */
    static const int dbg = getenv("ODHIP_INVERSE_DBG") ? atoi(getenv("ODHIP_INVERSE_DBG")) : 0;
    im.a[l].dbg = dbg;
#endif
    static const bool narrow = ODHIP_EXP_ENV("ODHIP_INVERSE_NARROW") != nullptr;     /* A/B: the shifted window */
    im.a[l].wide = !narrow && !ctx->fpr && !(levels[l].px_stride & 15) && !(levels[l].px_plane_stride & 15)
     && !((uintptr_t)levels[l].px & 15);
    im.a[l].px16 = ctx->fpr != 0;
    em.a[l].px16 = ctx->fpr != 0;
    if (ctx->fpr && ((uintptr_t)levels[l].px & 7)) return ODHIP_EINVAL;
    im.a[l].vs = g_strips + (vs_words + hs_words)*l;
    im.a[l].hs = im.a[l].vs + vs_words;
    em.a[l].vs = im.a[l].vs;
    em.a[l].hs = im.a[l].hs;
    em.a[l].px = levels[l].px;
    em.a[l].px_stride = levels[l].px_stride;
    em.a[l].px_plane_stride = levels[l].px_plane_stride;
    em.a[l].w = w;
    em.a[l].h = h;
    em.a[l].tile = tile;
  }
  for (int l = 0; l < nlevels; l++) {
    em.a[l].edge0 = 0;
    em.a[l].edge_step = 1;
    em.a[l].nedges = nv;
  }
  /* Walking workgroups (k_inverse_walk) wherever the plane width is a whole number of groups:
     every level of 4:2:0 chroma (pairs of 32x32 superblocks, one wavefront per pair), the leaf
     levels up to 16x16 of luma.  The 32x32 / 64x64 leaf levels of luma keep one workgroup per
     superblock pair (k_inverse_sb_top2): a 64x64 tile per wavefront bounds them at four
     workgroups per CU, where segments long enough to matter leave a partially filled last round.
     ODHIP_INVERSE_OLD=1: one workgroup per superblock for everything (the A/B baseline);
     ODHIP_INVERSE_SEG=n: groups per workgroup. */
  static const bool old_kernels = ODHIP_EXP_ENV("ODHIP_INVERSE_OLD") != nullptr;
  const int src = ref ? 2 : (levels[0].y ? 1 : 0);
  bool same_src = true;
  for (int l = 0; l < nlevels; l++) same_src = same_src && ((levels[l].y != nullptr) == (levels[0].y != nullptr));
  const bool top2_ok = !dec && (w/tile) % 2 == 0 && !ODHIP_EXP_ENV("ODHIP_INVERSE_X1");
  const int G = dec ? 2 : 1;
  const int ng = w/(tile*G);
  const bool walk = !old_kernels && same_src && (w/tile) % G == 0;
  /* groups per workgroup (measured, profiles/r4_inverse_segments.txt: 16 frames of 1080p): short
     segments win - many more workgroups than slots keep the phases of co-resident workgroups
     staggered (walkers that start together load, compute and store together), and a long
     segment's last, partially filled round costs a whole segment.  Luma 6 superblocks, chroma 3
     pairs: 25 / 29 and 25 / 29 of the vertical edges never leave LDS.
     ODHIP_INVERSE_SEG="luma,chroma_lo,chroma_hi" overrides. */
  static int seg_cfg[3] = {6, 3, 3};
  static const bool seg_parsed = [] {
    const char *e = ODHIP_EXP_ENV("ODHIP_INVERSE_SEG");
    if (e) (void)sscanf(e, "%d,%d,%d", &seg_cfg[0], &seg_cfg[1], &seg_cfg[2]);
    for (int i = 0; i < 3; i++) if (seg_cfg[i] < 1) seg_cfg[i] = 1;
    return true;
  }();
  (void)seg_parsed;
  const int seg_lo = dec ? seg_cfg[1] : seg_cfg[0];
  const int seg_hi = dec ? seg_cfg[2] : seg_cfg[0];
  InverseWalkArgs lo;
  InverseWalkArgs hi;
  memset(&lo, 0, sizeof(lo));
  memset(&hi, 0, sizeof(hi));
  lo.nplanes = hi.nplanes = nplanes;
  lo.seg_len = seg_lo;
  hi.seg_len = seg_hi;
  InverseArgsMulti top;
  memset(&top, 0, sizeof(top));
  top.nplanes = nplanes;
  int nlo = 0;
  int nhi = 0;
  int ntop = 0;
  if (walk) {
    for (int l = 0; l < nlevels; l++) {
      const int leaf = im.a[l].leaf_bs;
      const bool walks = dec || leaf <= 2;
      if (walks) {
        (leaf <= 2 ? lo.a[nlo++] : hi.a[nhi++]) = im.a[l];
        const int seg_len = leaf <= 2 ? seg_lo : seg_hi;
        em.a[l].edge0 = seg_len*G - 1;
        em.a[l].edge_step = seg_len*G;
        em.a[l].nedges = (ng + seg_len - 1)/seg_len - 1;
      }
      else {
        top.a[ntop++] = im.a[l];
        if (top2_ok && src == 1) {       /* k_inverse_sb_top2 finishes the edge inside every pair */
          em.a[l].edge0 = 1;
          em.a[l].edge_step = 2;
          em.a[l].nedges = nv/2;
        }
      }
    }
  }
  /* ODHIP_INV_LDS_PAD=bytes (experiments build): extra dynamic LDS per workgroup of the LUMA walker, i.e. fewer of
     them per CU and registers left for the other chain's preparation kernels (profiles/r6_overlap.txt, section 7) */
  static const size_t walk_pad = [] {
    const char *e = ODHIP_EXP_ENV("ODHIP_INV_LDS_PAD");
    return e ? (size_t)atoi(e) : (size_t)0;
  }();
  if (walk) {
    const dim3 glo((ng + seg_lo - 1)/seg_lo, h/tile, nplanes*nlo);
    const dim3 ghi((ng + seg_hi - 1)/seg_hi, h/tile, nplanes*nhi);
    if (dec) {
      if (nhi) {
        if (src == 2) k_inverse_walk<32, 2, 128, 2, 3, 3><<<ghi, 128, 0, s>>>(hi);
        else if (src == 1) k_inverse_walk<32, 2, 128, 1, 3, 3><<<ghi, 128, 0, s>>>(hi);
        else k_inverse_walk<32, 2, 128, 0, 3, 3><<<ghi, 128, 0, s>>>(hi);
      }
      if (nlo) {
        if (src == 2) k_inverse_walk<32, 2, 128, 2, 0, 2><<<glo, 128, 0, s>>>(lo);
        else if (src == 1) k_inverse_walk<32, 2, 128, 1, 0, 2><<<glo, 128, 0, s>>>(lo);
        else k_inverse_walk<32, 2, 128, 0, 0, 2><<<glo, 128, 0, s>>>(lo);
      }
    }
    else {
      if (ntop) {
        if (src == 2) k_inverse_sb<64, true><<<dim3(w/tile, h/tile, nplanes*ntop), Geo<64>::kNT, 0, s>>>(top);
        else if (src == 1 && top2_ok) {
          k_inverse_sb_top2<<<dim3(w/(2*tile), h/tile, nplanes*ntop), 256, 0, s>>>(top);
        }
        else k_inverse_sb<64, false, 3, 4><<<dim3(w/tile, h/tile, nplanes*ntop), Geo<64>::kNT, 0, s>>>(top);
      }
      if (nlo) {
        if (src == 2) k_inverse_walk<64, 1, 256, 2, 0, 2><<<glo, 256, 0, s>>>(lo);
        else if (src == 1) k_inverse_walk<64, 1, 256, 1, 0, 2><<<glo, 256, walk_pad, s>>>(lo);
        else k_inverse_walk<64, 1, 256, 0, 0, 2><<<glo, 256, 0, s>>>(lo);
      }
    }
  }
  else {
  const dim3 grid(w/tile, h/tile, nplanes*nlevels);
  if (ref) {
    if (dec) k_inverse_sb<32, true><<<grid, Geo<32>::kNT, 0, s>>>(im);
    else k_inverse_sb<64, true><<<grid, Geo<64>::kNT, 0, s>>>(im);
  }
  else if (dec) k_inverse_sb<32><<<grid, Geo<32>::kNT, 0, s>>>(im);
  else {
    /* leaf levels up to 16x16 and the 32x32 / 64x64 ones as separate launches (see k_inverse_sb) */
    InverseArgsMulti lo;
    InverseArgsMulti hi;
    memset(&lo, 0, sizeof(lo));
    memset(&hi, 0, sizeof(hi));
    lo.nplanes = nplanes;
    hi.nplanes = nplanes;
    int nlo = 0;
    int nhi = 0;
    for (int l = 0; l < nlevels; l++) {
      if (im.a[l].leaf_bs <= 2) lo.a[nlo++] = im.a[l];
      else hi.a[nhi++] = im.a[l];
    }
    if (nhi) {
      bool pulse_fed = true;
      for (int l = 0; l < nhi; l++) pulse_fed = pulse_fed && hi.a[l].y != nullptr;
      if (pulse_fed && top2_ok) {
        k_inverse_sb_top2<<<dim3(w/(2*tile), h/tile, nplanes*nhi), 256, 0, s>>>(hi);
        for (int l = 0; l < nlevels; l++) {
          if (im.a[l].leaf_bs > 2) {
            em.a[l].edge0 = 1;
            em.a[l].edge_step = 2;
            em.a[l].nedges = nv/2;
          }
        }
      }
      else k_inverse_sb<64, false, 3, 4><<<dim3(w/tile, h/tile, nplanes*nhi), Geo<64>::kNT, 0, s>>>(hi);
    }
    if (nlo) k_inverse_sb<64, false, 0, 2><<<dim3(w/tile, h/tile, nplanes*nlo), Geo<64>::kNT, 0, s>>>(lo);
  }
  }
  int max_edges = 0;
  for (int l = 0; l < nlevels; l++) max_edges = em.a[l].nedges > max_edges ? em.a[l].nedges : max_edges;
  if (max_edges > 0) k_edge_rows<<<dim3((h + 255)/256, max_edges, nplanes*nlevels), 256, 0, s>>>(em);
  if (nh > 0) k_edge_cols<<<dim3((w + 255)/256, nh, nplanes*nlevels), 256, 0, s>>>(em);
  return odhip_check_launch();
}

int inverse_launch(InverseArgs ia, int nplanes, int dec, hipStream_t s) {
  return inverse_launch(&ia, 1, nplanes, dec, s);
}

odhip_device_once g_inv_tables;

int upload_inv_tables_now(void) {
  unsigned short packed[OD_SCAN_LEN];
  unsigned char band_of[OD_SCAN_LEN];
  for (int j = 0; j < OD_SCAN_LEN; j++) {
    packed[j] = (unsigned short)(OD_SCAN_XY[j][1] << 8 | OD_SCAN_XY[j][0]);
    int b = 0;
    while (b + 1 < OD_NBANDS[4] && j >= OD_BAND_OFFS[4][b + 1]) b++;
    band_of[j] = (unsigned char)b;
  }
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gInvScanXY), packed, sizeof(packed)));
  ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gInvBandOf), band_of, sizeof(band_of)));
  return ODHIP_SUCCESS;
}

int upload_inv_tables(void) {
  return odhip_once_per_device(g_inv_tables, upload_inv_tables_now);
}

}  // namespace

/* The decoder's reconstruction of a batch of frames at their own partitions:
   od_decode_recursive's idct_2d + od_postfilter_split (src/decode.c:482-660), then
   od_apply_postfilter_frame_sbs and od_coeff_to_ref_plane (src/decode.c:988-996,
   src/state.c:1281-1345), from the dequantised coefficient planes and the block-size
   map. */
extern "C" int odhip_inverse_partition(uint8_t *d_px, int px_stride, long px_plane_stride,
 const od_coeff *d_coef, int nplanes, int w, int h, int dec, const uint8_t *d_bsize, int bstride,
 long bsize_frame_stride, int planes_per_frame, int pic_w, int pic_h, odhip_stream stream) {
  if (!d_px || !d_coef || !d_bsize || nplanes <= 0 || planes_per_frame <= 0 || (dec != 0 && dec != 1)) {
    return ODHIP_EINVAL;
  }
  const int tile = 64 >> dec;
  if (w <= 0 || h <= 0 || w % tile || h % tile || (px_stride & 3) || (px_plane_stride & 3)
   || bstride < (w/tile)*8 || ((uintptr_t)d_coef & 15)) {
    return ODHIP_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nv = w/tile - 1;
  const int nh = h/tile - 1;
  const size_t vs_words = ((size_t)nplanes*nv*h*4 + 3) & ~(size_t)3;
  const size_t hs_words = ((size_t)nplanes*nh*4*w + 3) & ~(size_t)3;
  const size_t need = (vs_words + hs_words + 4)*sizeof(od_coeff);
  ODHIP_CTX_OR_RETURN(ctx);
  LappedState &st = *odhip_ctx_state<LappedState>(ctx, ODHIP_SLOT_LAPPED);
  if (need > st.bytes) {
    ODHIP_TRY(hipStreamSynchronize(s));
    if (st.strips) ODHIP_TRY(hipFree(st.strips));
    st.strips = nullptr;
    st.bytes = 0;
    ODHIP_TRY(hipMalloc((void **)&st.strips, need));
    st.bytes = need;
  }
  InversePartArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.a.coef = d_coef;
  pa.a.px = d_px;
  pa.a.px_stride = px_stride;
  pa.a.px_plane_stride = px_plane_stride;
  pa.a.w = w;
  pa.a.h = h;
  pa.a.pic_w = pic_w;
  pa.a.pic_h = pic_h;
  pa.a.vs = st.strips;
  pa.a.hs = st.strips + vs_words;
  pa.a.px16 = ctx->fpr != 0;
  if (ctx->fpr && ((uintptr_t)d_px & 7)) return ODHIP_EINVAL;
  pa.bsize = d_bsize;
  pa.bstride = bstride;
  pa.bsize_frame_stride = bsize_frame_stride;
  pa.planes_per_frame = planes_per_frame;
  pa.nplanes = nplanes;
  EdgeArgsMulti em;
  memset(&em, 0, sizeof(em));
  em.nplanes = nplanes;
  em.a[0].vs = pa.a.vs;
  em.a[0].hs = pa.a.hs;
  em.a[0].px = d_px;
  em.a[0].px_stride = px_stride;
  em.a[0].px_plane_stride = px_plane_stride;
  em.a[0].w = w;
  em.a[0].h = h;
  em.a[0].tile = tile;
  em.a[0].px16 = ctx->fpr != 0;
  em.a[0].edge0 = 0;
  em.a[0].edge_step = 1;
  em.a[0].nedges = nv;
  const dim3 grid(w/tile, h/tile, nplanes);
  if (dec) k_inverse_part<32><<<grid, Geo<32>::kNT, 0, s>>>(pa);
  else k_inverse_part<64><<<grid, Geo<64>::kNT, 0, s>>>(pa);
  if (nv > 0) k_edge_rows<<<dim3((h + 255)/256, nv, nplanes), 256, 0, s>>>(em);
  if (nh > 0) k_edge_cols<<<dim3((w + 255)/256, nh, nplanes), 256, 0, s>>>(em);
  return odhip_check_launch();
}

extern "C" int odhip_inverse_level(uint8_t *d_px, int px_stride, long px_plane_stride,
 const od_coeff *d_coef, int nplanes, int w, int h, int dec, int leaf_bs,
 int pic_w, int pic_h, odhip_stream stream) {
  if (!d_px || !d_coef || nplanes <= 0 || (dec != 0 && dec != 1)) return ODHIP_EINVAL;
  const int tile = 64 >> dec;
  if (w <= 0 || h <= 0 || w % tile || h % tile || (px_stride & 3)
   || (px_plane_stride & 3) || leaf_bs < 0 || leaf_bs > 4 - dec) {
    return ODHIP_EINVAL;
  }
  InverseArgs ia;
  memset(&ia, 0, sizeof(ia));
  ia.coef = d_coef;
  ia.px = d_px;
  ia.px_stride = px_stride;
  ia.px_plane_stride = px_plane_stride;
  ia.w = w;
  ia.h = h;
  ia.pic_w = pic_w;
  ia.pic_h = pic_h;
  ia.leaf_bs = leaf_bs;
  return inverse_launch(ia, nplanes, dec, (hipStream_t)stream);
}

/* odhip_inverse_level for several partition levels of ONE plane set in a single
   set of launches: level leaf_bs[i] is reconstructed from d_coef[i] into d_px[i]. */
extern "C" int odhip_inverse_levels(uint8_t *const *d_px, int px_stride, long px_plane_stride,
 const od_coeff *const *d_coef, const int *leaf_bs, int nlevels, int nplanes, int w, int h, int dec,
 int pic_w, int pic_h, odhip_stream stream) {
  if (!d_px || !d_coef || !leaf_bs || nlevels <= 0 || nlevels > kMaxInvLevels || nplanes <= 0
   || (dec != 0 && dec != 1)) {
    return ODHIP_EINVAL;
  }
  const int tile = 64 >> dec;
  if (w <= 0 || h <= 0 || w % tile || h % tile || (px_stride & 3) || (px_plane_stride & 3)) {
    return ODHIP_EINVAL;
  }
  InverseArgs ia[kMaxInvLevels];
  memset(ia, 0, sizeof(ia));
  for (int i = 0; i < nlevels; i++) {
    if (!d_px[i] || !d_coef[i] || leaf_bs[i] < 0 || leaf_bs[i] > 4 - dec) return ODHIP_EINVAL;
    ia[i].coef = d_coef[i];
    ia[i].px = d_px[i];
    ia[i].px_stride = px_stride;
    ia[i].px_plane_stride = px_plane_stride;
    ia[i].w = w;
    ia[i].h = h;
    ia[i].pic_w = pic_w;
    ia[i].pic_h = pic_h;
    ia[i].leaf_bs = leaf_bs[i];
  }
  return inverse_launch(ia, nlevels, nplanes, dec, (hipStream_t)stream);
}

namespace {

int pvq_inverse_args(InverseArgs &ia, uint8_t *d_px, int px_stride, long px_plane_stride,
 const odhip_pvq_job *job, int dec, int pic_w, int pic_h) {
  if (!d_px || !job || !job->d_coef || !job->cands.y || !job->cands.choice || !job->d_qm_inv
   || job->nplanes <= 0 || (dec != 0 && dec != 1)) {
    return ODHIP_EINVAL;
  }
  const int tile = 64 >> dec;
  const int w = job->w;
  const int h = job->h;
  const int bs = job->bs;
  if (w <= 0 || h <= 0 || w % tile || h % tile || (px_stride & 3) || (px_plane_stride & 3)
   || bs < 0 || bs > 4 - dec || ((uintptr_t)job->cands.y & 15) || ((uintptr_t)job->cands.choice & 15)
   || ((uintptr_t)job->d_qm_inv & 15)) {        /* 16-byte vector loads of all three */
    return ODHIP_EINVAL;
  }
  const int n = 4 << bs;
  memset(&ia, 0, sizeof(ia));
  ia.coef = job->d_coef;
  ia.px = d_px;
  ia.px_stride = px_stride;
  ia.px_plane_stride = px_plane_stride;
  ia.w = w;
  ia.h = h;
  ia.pic_w = pic_w;
  ia.pic_h = pic_h;
  ia.leaf_bs = bs;
  ia.y = job->cands.y;
  ia.choice = reinterpret_cast<const int4 *>(job->cands.choice);
  ia.qm_inv = job->d_qm_inv;
  ia.nblocks = (long)job->nplanes*(w/n)*(h/n);
  ia.len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  if (2*ia.nblocks*ia.len >= 0x7fffffffL) return ODHIP_EINVAL;  /* 32-bit element indices */
  ia.nb_bands = OD_NBANDS[bs];
  return ODHIP_SUCCESS;
}

}  // namespace

extern "C" int odhip_inverse_level_pvq(uint8_t *d_px, int px_stride, long px_plane_stride,
 const odhip_pvq_job *job, int dec, int pic_w, int pic_h, odhip_stream stream) {
  InverseArgs ia;
  int rc = pvq_inverse_args(ia, d_px, px_stride, px_plane_stride, job, dec, pic_w, pic_h);
  if (rc) return rc;
  rc = upload_inv_tables();
  if (rc) return rc;
  return inverse_launch(ia, job->nplanes, dec, (hipStream_t)stream);
}

extern "C" int odhip_inverse_levels_pvq(uint8_t *const *d_px, int px_stride, long px_plane_stride,
 const odhip_pvq_job *jobs, int njobs, int dec, int pic_w, int pic_h, odhip_stream stream) {
  if (!d_px || !jobs || njobs <= 0 || njobs > kMaxInvLevels) return ODHIP_EINVAL;
  InverseArgs ia[kMaxInvLevels];
  for (int i = 0; i < njobs; i++) {
    const int rc = pvq_inverse_args(ia[i], d_px[i], px_stride, px_plane_stride, &jobs[i], dec, pic_w,
     pic_h);
    if (rc) return rc;
    if (jobs[i].nplanes != jobs[0].nplanes || jobs[i].w != jobs[0].w || jobs[i].h != jobs[0].h) {
      return ODHIP_EINVAL;
    }
  }
  const int rc = upload_inv_tables();
  if (rc) return rc;
  return inverse_launch(ia, njobs, jobs[0].nplanes, dec, (hipStream_t)stream);
}

/* odhip_inverse_levels_pvq for the WITH-reference band stage: the chosen candidates
   of jobs[i] (choice records of odhip_pvq_ref_select_synth_multi or
   odhip_pvq_ref_choose_multi) are dequantised while the superblock tile is loaded -
   od_pvq_synthesis_partial with or without reference, the skip-copy and skip-zero
   bands included - so the dequantised plane never exists in HBM. */
extern "C" int odhip_inverse_levels_pvq_ref(uint8_t *const *d_px, int px_stride, long px_plane_stride,
 const odhip_pvq_refjob *jobs, int njobs, int dec, int pic_w, int pic_h, odhip_stream stream) {
  if (!d_px || !jobs || njobs <= 0 || njobs > kMaxInvLevels || (dec != 0 && dec != 1)) return ODHIP_EINVAL;
  const int tile = 64 >> dec;
  InverseArgs ia[kMaxInvLevels];
  memset(ia, 0, sizeof(ia));
  for (int i = 0; i < njobs; i++) {
    const odhip_pvq_refjob &j = jobs[i];
    /* a job whose reference is read in place from the luma stage (keyframe chroma from luma)
       has no reference plane, and none of its bands can be a skip-copy
       (src/pvq_encoder.c:620: not with CfL) */
    const bool lref = j.luma != nullptr && j.is_keyframe && j.pli != 0;
    if (!d_px[i] || !j.d_coef || (!j.d_ref && !lref) || !j.y || !j.r16 || !j.choice || !j.d_qm_inv || j.nplanes <= 0
     || j.nplanes != jobs[0].nplanes || j.w != jobs[0].w || j.h != jobs[0].h) {
      return ODHIP_EINVAL;
    }
    const int w = j.w;
    const int h = j.h;
    const int bs = j.bs;
    if (w <= 0 || h <= 0 || w % tile || h % tile || (px_stride & 3) || (px_plane_stride & 3) || bs < 0
     || bs > 4 - dec || ((uintptr_t)j.y & 15) || ((uintptr_t)j.choice & 15) || ((uintptr_t)j.r16 & 15)
     || ((uintptr_t)j.d_qm_inv & 15)) {
      return ODHIP_EINVAL;
    }
    const int n = 4 << bs;
    ia[i].coef = j.d_coef;
    ia[i].px = d_px[i];
    ia[i].px_stride = px_stride;
    ia[i].px_plane_stride = px_plane_stride;
    ia[i].w = w;
    ia[i].h = h;
    ia[i].pic_w = pic_w;
    ia[i].pic_h = pic_h;
    ia[i].leaf_bs = bs;
    ia[i].y = j.y;
    ia[i].choice = reinterpret_cast<const int4 *>(j.choice);
    ia[i].qm_inv = j.d_qm_inv;
    ia[i].nblocks = (long)j.nplanes*(w/n)*(h/n);
    ia[i].len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
    ia[i].nb_bands = OD_NBANDS[bs];
    ia[i].r16 = j.r16;
    ia[i].ref = j.d_ref;
    ia[i].inter = j.is_keyframe == 0;
  }
  const int rc = upload_inv_tables();
  if (rc) return rc;
  return inverse_launch(ia, njobs, jobs[0].nplanes, dec, (hipStream_t)stream, true);
}
