/* dct_kernels.hip - standalone batched N x N fDCT / iDCT for gfx950.

   One 256-thread workgroup stages a 64 x 64 coefficient tile (one 64x64
   block, four 32x32, ... 256 4x4) in LDS with coalesced 16-byte global
   accesses, runs the separable transform on it (od_tile.cuh) and streams the
   result back.  HBM-bound by design: 4 B read + 4 B written per coefficient
   (SURVEY.md section 8(d)), no re-reads.

   Reference: od_bin_fdctNxN / od_bin_idctNxN, src/dct.c:151-163, 351-363,
   792-806, 4890-4920. */
#include "../../include/daala_hip.h"
#include <type_traits>
#include "od_common.cuh"
#include "od_tile.cuh"
#include "od_filters.cuh"

namespace {

constexpr int kTile = 64;

/* Threads per 64x64 tile: a pass has 64*(64/N) column (row) tasks.  The 64-point passes
   have 64: one wave per tile (its barriers become wave-local) measured 44.6 % of HBM
   against 42.2 % with 256 threads; the 32-point level (128 tasks) measured 63 % with 128
   threads against 65.5 % with 256 and keeps 256.  Both are bound by the per-lane network
   (1300 dependent-ish operations per 64-point transform), not by idle waves. */
template <int LN>
struct DctGeo {
  static constexpr int kNT = LN == 4 ? 64 : 256;
};

/* Addressing of a workgroup's 64x64 tile in global memory. */
struct BatchMap {
  /* nblocks contiguous N x N tiles; block g of the workgroup sits at grid
     position (g % (64/N), g / (64/N)) of the LDS tile. */
  long first_block;
  long nblocks;
  int n;
  __device__ __forceinline__ bool block_active(int bx, int by) const {
    return first_block + by*(kTile/n) + bx < nblocks;
  }
  /* Offset of 4 consecutive coefficients at tile position (x, y); -1 = absent. */
  __device__ __forceinline__ long offset(int x, int y) const {
    const int bx = x/n;
    const int by = y/n;
    const long b = first_block + by*(kTile/n) + bx;
    if (b >= nblocks) return -1;
    return (b*n + (y - by*n))*n + (x - bx*n);
  }
};

struct PlaneMap {
  int x0;
  int y0;
  int w;
  int h;
  int stride;
  int n;
  __device__ __forceinline__ bool block_active(int bx, int by) const {
    return x0 + bx*n < w && y0 + by*n < h;
  }
  __device__ __forceinline__ long offset(int x, int y) const {
    if (x0 + x >= w || y0 + y >= h) return -1;
    return (long)(y0 + y)*stride + x0 + x;
  }
};

template <int LN, bool INV, typename T, typename InMap, typename OutMap>
__device__ __forceinline__ void dct2d_tile(od_coeff *out, const od_coeff *in,
 const InMap &imap, const OutMap &omap) {
  constexpr int P = OdTile<kTile>::kPitch;
  constexpr int kNT = DctGeo<LN>::kNT;
  __shared__ __attribute__((aligned(16))) int tile[OdTile<kTile>::kWords];
  const int tid = threadIdx.x;
  /* Global -> LDS, 16 B per lane, consecutive lanes consecutive addresses
     inside a block row. */
  for (int i = tid; i < kTile*kTile/4; i += kNT) {
    const int y = i/(kTile/4);
    const int x = (i % (kTile/4))*4;
    const long off = imap.offset(x, y);
    int4 v = make_int4(0, 0, 0, 0);
    if (off >= 0) v = *reinterpret_cast<const int4 *>(in + off);
    *reinterpret_cast<int4 *>(tile + y*P + x) = v;
  }
  __syncthreads();
  auto active = [&](int bx, int by) { return imap.block_active(bx, by); };
  if constexpr (!INV) {
    od_tile_cols<kTile, LN, false, T, kNT>(tile, tile, tid, active);
    __syncthreads();
    od_tile_rows<kTile, LN, false, T, kNT>(tile, tile, tid, active);
  }
  else {
    od_tile_rows<kTile, LN, true, T, kNT>(tile, tile, tid, active);
    __syncthreads();
    od_tile_cols<kTile, LN, true, T, kNT>(tile, tile, tid, active);
  }
  __syncthreads();
  for (int i = tid; i < kTile*kTile/4; i += kNT) {
    const int y = i/(kTile/4);
    const int x = (i % (kTile/4))*4;
    const long off = omap.offset(x, y);
    if (off >= 0) {
      *reinterpret_cast<int4 *>(out + off) =
       *reinterpret_cast<const int4 *>(tile + y*P + x);
    }
  }
}

template <int LN, bool INV, typename T>
__global__ __launch_bounds__(DctGeo<LN>::kNT) void k_dct2d_batch(od_coeff *out,
 const od_coeff *in, long nblocks) {
  constexpr int N = 4 << LN;
  constexpr int kPerWg = (kTile/N)*(kTile/N);
  BatchMap m;
  m.first_block = (long)blockIdx.x*kPerWg;
  m.nblocks = nblocks;
  m.n = N;
  dct2d_tile<LN, INV, T>(out, in, m, m);
}

template <int LN, bool INV, typename T>
__global__ __launch_bounds__(DctGeo<LN>::kNT) void k_dct2d_plane(od_coeff *out,
 int out_stride, const od_coeff *in, int in_stride, int w, int h) {
  constexpr int N = 4 << LN;
  PlaneMap im;
  im.x0 = blockIdx.x*kTile;
  im.y0 = blockIdx.y*kTile;
  im.w = w;
  im.h = h;
  im.stride = in_stride;
  im.n = N;
  PlaneMap om = im;
  om.stride = out_stride;
  dct2d_tile<LN, INV, T>(out, in, im, om);
}

/* ---- 64x64 forward, two wavefronts per block (round 5) -------------------------------------
   One wavefront per 64x64 block (above) runs a 1191-instruction dependent network twice with
   17 KB of LDS per wavefront: 2.25 wavefronts per SIMD, each a long dependency chain - measured at
   8.5 cycles per VALU instruction, twice what a filled SIMD sustains (0.44 of the HBM spec,
   profiles/r4_microbench_configs2_3.txt).  Here every column, then every row, is shared by TWO
   lanes running the even / odd half network (od_fdct_lift_half: after its first butterfly stage a
   forward transform separates into two independent halves), the halves wave-uniform: 128 threads
   per block, half the registers per lane, and the columns are read straight from global memory
   (coalesced over the 64 lanes of a wavefront; the second wavefront's reads hit the cache), so the
   only LDS is the intermediate - kept TRANSPOSED and in parity-split order (position p' =
   (p & 1)*32 + (p >> 1)) at an odd pitch of 65 words, every access of both passes and of the
   copy-out a conflict-free ds_read / write_b32 (the layout of pyramid_level_split64,
   lapped_kernels.hip).  16.6 KB per block of two wavefronts: 4.5 wavefronts per SIMD. */
template <typename T>
__global__ __launch_bounds__(128) void k_fdct64_split(od_coeff *out, long out_stride, long out_block,
 const od_coeff *in, long in_stride, long in_block, int blocks_x) {
  constexpr int N = 64;
  constexpr int H = 32;
  constexpr int PZ = 65;
  __shared__ int z[N*PZ];
  const int tid = threadIdx.x;
  const int half = tid >> 6;
  const int c = tid & 63;
  /* batch: block b at b*4096, rows 64 apart; plane: block (bx, by) at (by*64)*stride + bx*64 */
  const long bx = blocks_x ? blockIdx.x % blocks_x : 0;
  const long by = blocks_x ? blockIdx.x / blocks_x : blockIdx.x;
  const od_coeff *src = in + by*in_block + bx*N;
  od_coeff *dst = out + by*out_block + bx*N;
  {
    T a[N];
    T o[H];
#pragma unroll
    for (int r = 0; r < N; r++) a[r] = T(src[r*in_stride + c]);
    if (half == 0) od_fdct_lift_half<4, 0>(o, a);
    else od_fdct_lift_half<4, 1>(o, a);
#pragma unroll
    for (int k = 0; k < H; k++) z[c*PZ + half*H + k] = o[k];
  }
  __syncthreads();
  {
    T a[N];
    T o[H];
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = T(z[i*PZ + c]);
    __syncthreads();          /* the partner lane reads the same words */
    if (half == 0) od_fdct_lift_half<4, 0>(o, a);
    else od_fdct_lift_half<4, 1>(o, a);
#pragma unroll
    for (int k = 0; k < H; k++) z[(half*H + k)*PZ + c] = o[k];
  }
  __syncthreads();
  /* raster (y, x) <- z[x'][y']: 16 bytes per lane and store */
  constexpr int K = N*N/4/128;
  int q[K][4];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = tid + k*128;
    const int y = i/(N/4);
    const int x = (i % (N/4))*4;
    const int col = (y & 1)*H + (y >> 1);
#pragma unroll
    for (int j = 0; j < 4; j++) q[k][j] = z[(((x + j) & 1)*H + ((x + j) >> 1))*PZ + col];
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = tid + k*128;
    *reinterpret_cast<int4 *>(dst + (long)(i/(N/4))*out_stride + (i % (N/4))*4) =
     make_int4(q[k][0], q[k][1], q[k][2], q[k][3]);
  }
}

/* ---- 64x64 inverse, two wavefronts per block (round 5) -------------------------------------
   The same occupancy argument as k_fdct64_split.  An inverse network does not split by output
   parity; read backwards it STARTS with two independent sub-networks - one fed by the even-indexed
   inputs, one by the odd-indexed ones - and ends in a joining stage (od_idct64_lift_part / _join,
   od_lift.cuh): the lane pair of a line runs one part each, exchanges through LDS the 32 values the
   other lane's join needs, and each produces one half of the outputs.  The block is staged in one
   64 x 68 tile (16-byte global loads); the exchange area aliases the tile once every lane holds its
   inputs in registers; the rows go back into the tile as 16-byte pieces, the columns straight to
   global memory (coalesced over the 64 lanes of a wavefront).  17 KB per block of two wavefronts. */
template <int J0, int J1, class F>
__device__ __forceinline__ void dct_static_for(F &&f) {
  if constexpr (J0 < J1) {
    f(std::integral_constant<int, J0>{});
    dct_static_for<J0 + 1, J1>(f);
  }
}

/* One lane's share of a 64-point inverse line in three steps with a workgroup barrier between them
   (the barriers stay OUTSIDE the wave-uniform branches on the half): part HALF on its 32 inputs;
   export of what the other lane's join needs into x[slot][lane]; import + join HALF. */
template <int HALF, typename T>
__device__ __forceinline__ void idct64_export(const T (&own)[48], int *x, int c) {
  dct_static_for<0, 32>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int idx = HALF == 0 ? kIdct64NeedIdx1[j] : kIdct64NeedIdx0[j];
    x[(HALF*32 + j)*64 + c] = own[idx];
  });
}

template <int HALF, typename T>
__device__ __forceinline__ void idct64_join(T (&o)[32], const T (&own)[48], const int *x, int c) {
  T other[48];
  dct_static_for<0, 32>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int idx = HALF == 0 ? kIdct64NeedIdx0[j] : kIdct64NeedIdx1[j];
    other[idx] = T(x[((1 - HALF)*32 + j)*64 + c]);
  });
  if constexpr (HALF == 0) od_idct64_lift_join<0>(o, own, other);
  else od_idct64_lift_join<1>(o, other, own);
}

/* a[]: the lane's 32 inputs (parity `half`) -> o[]: outputs half*32 .. half*32 + 31 of the line. */
template <typename T>
__device__ __forceinline__ void idct64_pair(T (&o)[32], const T (&a)[32], int *x, int c, int half) {
  T own[48];
  if (half == 0) od_idct64_lift_part<0>(own, a);
  else od_idct64_lift_part<1>(own, a);
  __syncthreads();                       /* every lane of the block holds its inputs: the tile is free */
  if (half == 0) idct64_export<0>(own, x, c);
  else idct64_export<1>(own, x, c);
  __syncthreads();
  if (half == 0) idct64_join<0>(o, own, x, c);
  else idct64_join<1>(o, own, x, c);
}

template <typename T>
__global__ __launch_bounds__(128) void k_idct64_split(od_coeff *out, long out_stride, long out_block,
 const od_coeff *in, long in_stride, long in_block, int blocks_x) {
  constexpr int N = 64;
  constexpr int H = 32;
  constexpr int P = OdTile<kTile>::kPitch;
  static_assert(kIdct64Need0 == 32 && kIdct64Need1 == 32, "exchange area: 64 slots of 64 lanes");
  __shared__ __attribute__((aligned(16))) int t[N*P];
  const int tid = threadIdx.x;
  const int half = tid >> 6;
  const int c = tid & 63;
  const long bx = blocks_x ? blockIdx.x % blocks_x : 0;
  const long by = blocks_x ? blockIdx.x / blocks_x : blockIdx.x;
  const od_coeff *src = in + by*in_block + bx*N;
  od_coeff *dst = out + by*out_block + bx*N;
  {
    constexpr int K = N*N/4/128;
    int4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = tid + k*128;
      v[k] = *reinterpret_cast<const int4 *>(src + (long)(i/(N/4))*in_stride + (i % (N/4))*4);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = tid + k*128;
      *reinterpret_cast<int4 *>(t + (i/(N/4))*P + (i % (N/4))*4) = v[k];
    }
  }
  __syncthreads();
  /* rows (od_bin_idct64x64: rows, then columns): lane c of wavefront `half` takes row c */
  {
    T a[H];
    T o[H];
#pragma unroll
    for (int q = 0; q < N/4; q++) {
      const int4 v = *reinterpret_cast<const int4 *>(t + c*P + 4*q);
      a[2*q] = T(half ? v.y : v.x);
      a[2*q + 1] = T(half ? v.w : v.z);
    }
    idct64_pair(o, a, t, c, half);
    __syncthreads();                     /* the exchange area is read: the tile takes the rows back */
#pragma unroll
    for (int q = 0; q < H/4; q++) {
      *reinterpret_cast<int4 *>(t + c*P + half*H + 4*q) = make_int4(o[4*q], o[4*q + 1], o[4*q + 2], o[4*q + 3]);
    }
  }
  __syncthreads();
  /* columns: lane c takes column c, its outputs go straight to global memory */
  {
    T a[H];
    T o[H];
#pragma unroll
    for (int k = 0; k < H; k++) a[k] = T(t[(2*k + half)*P + c]);
    idct64_pair(o, a, t, c, half);
#pragma unroll
    for (int k = 0; k < H; k++) dst[(long)(half*H + k)*out_stride + c] = o[k];
  }
}

template <bool INV, typename T>
int launch_batch(int ln, od_coeff *out, const od_coeff *in, long nblocks,
 hipStream_t s) {
  if (nblocks <= 0) return ODHIP_SUCCESS;
  const int n = 4 << ln;
  const long per = (long)(kTile/n)*(kTile/n);
  const long grid = (nblocks + per - 1)/per;
  if (grid > 0x7fffffffL) return ODHIP_EINVAL;
  switch (ln) {
    case 0: k_dct2d_batch<0, INV, T><<<(unsigned)grid, DctGeo<0>::kNT, 0, s>>>(out, in, nblocks); break;
    case 1: k_dct2d_batch<1, INV, T><<<(unsigned)grid, DctGeo<1>::kNT, 0, s>>>(out, in, nblocks); break;
    case 2: k_dct2d_batch<2, INV, T><<<(unsigned)grid, DctGeo<2>::kNT, 0, s>>>(out, in, nblocks); break;
    case 3: k_dct2d_batch<3, INV, T><<<(unsigned)grid, DctGeo<3>::kNT, 0, s>>>(out, in, nblocks); break;
    case 4:
      if constexpr (!INV) k_fdct64_split<T><<<(unsigned)grid, 128, 0, s>>>(out, 64, 4096, in, 64, 4096, 0);
      else k_idct64_split<T><<<(unsigned)grid, 128, 0, s>>>(out, 64, 4096, in, 64, 4096, 0);
      break;
    default: return ODHIP_EINVAL;
  }
  return odhip_check_launch();
}

template <bool INV, typename T>
int launch_plane(int ln, od_coeff *out, int out_stride, const od_coeff *in,
 int in_stride, int w, int h, hipStream_t s) {
  const int n = 4 << ln;
  if (w <= 0 || h <= 0 || w % n || h % n || (in_stride & 3) || (out_stride & 3)) {
    return ODHIP_EINVAL;
  }
  const dim3 grid((w + kTile - 1)/kTile, (h + kTile - 1)/kTile);
  switch (ln) {
    case 0: k_dct2d_plane<0, INV, T><<<grid, DctGeo<0>::kNT, 0, s>>>(out, out_stride, in, in_stride, w, h); break;
    case 1: k_dct2d_plane<1, INV, T><<<grid, DctGeo<1>::kNT, 0, s>>>(out, out_stride, in, in_stride, w, h); break;
    case 2: k_dct2d_plane<2, INV, T><<<grid, DctGeo<2>::kNT, 0, s>>>(out, out_stride, in, in_stride, w, h); break;
    case 3: k_dct2d_plane<3, INV, T><<<grid, DctGeo<3>::kNT, 0, s>>>(out, out_stride, in, in_stride, w, h); break;
    case 4:
      if constexpr (!INV) {
        /* whole 64x64 blocks only (w, h are multiples of n) */
        k_fdct64_split<T><<<(unsigned)(grid.x*grid.y), 128, 0, s>>>(out, out_stride, 64L*out_stride, in, in_stride,
         64L*in_stride, (int)grid.x);
      }
      else {
        k_idct64_split<T><<<(unsigned)(grid.x*grid.y), 128, 0, s>>>(out, out_stride, 64L*out_stride, in, in_stride,
         64L*in_stride, (int)grid.x);
      }
      break;
    default: return ODHIP_EINVAL;
  }
  return odhip_check_launch();
}

}  // namespace

namespace {

/* One n-tap lapping filter per lane on contiguous vectors [count][n]. */
template <int N, bool INV>
__global__ __launch_bounds__(256) void k_filter_batch(od_coeff *out, const od_coeff *in, long count) {
  const long i = (long)blockIdx.x*256 + threadIdx.x;
  if (i >= count) return;
  int t[N];
  const int4 *src = reinterpret_cast<const int4 *>(in + i*N);
#pragma unroll
  for (int v = 0; v < N/4; v++) {
    const int4 q = src[v];
    t[4*v] = q.x;
    t[4*v + 1] = q.y;
    t[4*v + 2] = q.z;
    t[4*v + 3] = q.w;
  }
  if (INV) od_post_filter_dev<N>(t);
  else od_pre_filter_dev<N>(t);
  int4 *dst = reinterpret_cast<int4 *>(out + i*N);
#pragma unroll
  for (int v = 0; v < N/4; v++) dst[v] = make_int4(t[4*v], t[4*v + 1], t[4*v + 2], t[4*v + 3]);
}

template <int N>
int launch_filter(int inverse, od_coeff *d_out, const od_coeff *d_in, long count, hipStream_t s) {
  const unsigned grid = (unsigned)((count + 255)/256);
  if (inverse) k_filter_batch<N, true><<<grid, 256, 0, s>>>(d_out, d_in, count);
  else k_filter_batch<N, false><<<grid, 256, 0, s>>>(d_out, d_in, count);
  return odhip_check_launch();
}

}  // namespace

extern "C" int odhip_filter_batch(int f, int inverse, od_coeff *d_out, const od_coeff *d_in, long count,
 odhip_stream stream) {
  if (f < 0 || f > 3) return ODHIP_EINVAL;
  if (count <= 0) return count < 0 ? ODHIP_EINVAL : ODHIP_SUCCESS;
  if (!d_out || !d_in || ((uintptr_t)d_out & 15) || ((uintptr_t)d_in & 15)) return ODHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (f) {
    case 0: return launch_filter<4>(inverse, d_out, d_in, count, s);
    case 1: return launch_filter<8>(inverse, d_out, d_in, count, s);
    case 2: return launch_filter<16>(inverse, d_out, d_in, count, s);
    default: return launch_filter<32>(inverse, d_out, d_in, count, s);
  }
}

extern "C" int odhip_fdct2d_batch(int ln, od_coeff *d_out, const od_coeff *d_in,
 long nblocks, int exact32, odhip_stream stream) {
  if (ln < 0 || ln >= ODHIP_NBSIZES) return ODHIP_EINVAL;
  if (nblocks <= 0) return nblocks < 0 ? ODHIP_EINVAL : ODHIP_SUCCESS;
  if (!d_out || !d_in) return ODHIP_EINVAL;
  /* 16-byte vector loads and stores */
  if (((uintptr_t)d_out | (uintptr_t)d_in) & 15) return ODHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  return exact32 ? launch_batch<false, OdMul32>(ln, d_out, d_in, nblocks, s)
                 : launch_batch<false, OdMul24>(ln, d_out, d_in, nblocks, s);
}

extern "C" int odhip_idct2d_batch(int ln, od_coeff *d_out, const od_coeff *d_in,
 long nblocks, int exact32, odhip_stream stream) {
  if (ln < 0 || ln >= ODHIP_NBSIZES) return ODHIP_EINVAL;
  if (nblocks <= 0) return nblocks < 0 ? ODHIP_EINVAL : ODHIP_SUCCESS;
  if (!d_out || !d_in) return ODHIP_EINVAL;
  /* 16-byte vector loads and stores */
  if (((uintptr_t)d_out | (uintptr_t)d_in) & 15) return ODHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  return exact32 ? launch_batch<true, OdMul32>(ln, d_out, d_in, nblocks, s)
                 : launch_batch<true, OdMul24>(ln, d_out, d_in, nblocks, s);
}

extern "C" int odhip_fdct2d_plane(int ln, od_coeff *d_out, int out_stride,
 const od_coeff *d_in, int in_stride, int w, int h, int exact32,
 odhip_stream stream) {
  if (ln < 0 || ln >= ODHIP_NBSIZES || !d_out || !d_in) return ODHIP_EINVAL;
  if (((uintptr_t)d_out | (uintptr_t)d_in) & 15) return ODHIP_EINVAL;   /* 16-byte vectors */
  hipStream_t s = (hipStream_t)stream;
  return exact32
   ? launch_plane<false, OdMul32>(ln, d_out, out_stride, d_in, in_stride, w, h, s)
   : launch_plane<false, OdMul24>(ln, d_out, out_stride, d_in, in_stride, w, h, s);
}

extern "C" int odhip_idct2d_plane(int ln, od_coeff *d_out, int out_stride,
 const od_coeff *d_in, int in_stride, int w, int h, int exact32,
 odhip_stream stream) {
  if (ln < 0 || ln >= ODHIP_NBSIZES || !d_out || !d_in) return ODHIP_EINVAL;
  if (((uintptr_t)d_out | (uintptr_t)d_in) & 15) return ODHIP_EINVAL;   /* 16-byte vectors */
  hipStream_t s = (hipStream_t)stream;
  return exact32
   ? launch_plane<true, OdMul32>(ln, d_out, out_stride, d_in, in_stride, w, h, s)
   : launch_plane<true, OdMul24>(ln, d_out, out_stride, d_in, in_stride, w, h, s);
}
