/* dct_kernels.hip - standalone batched N x N fDCT / iDCT for gfx950.

   One 256-thread workgroup stages a 64 x 64 coefficient tile (one 64x64
   block, four 32x32, ... 256 4x4) in LDS with coalesced 16-byte global
   accesses, runs the separable transform on it (od_tile.cuh) and streams the
   result back.  HBM-bound by design: 4 B read + 4 B written per coefficient
   (SURVEY.md section 8(d)), no re-reads.

   Reference: od_bin_fdctNxN / od_bin_idctNxN, src/dct.c:151-163, 351-363,
   792-806, 4890-4920. */
#include "../../include/daala_hip.h"
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
    case 4: k_dct2d_batch<4, INV, T><<<(unsigned)grid, DctGeo<4>::kNT, 0, s>>>(out, in, nblocks); break;
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
    case 4: k_dct2d_plane<4, INV, T><<<grid, DctGeo<4>::kNT, 0, s>>>(out, out_stride, in, in_stride, w, h); break;
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
