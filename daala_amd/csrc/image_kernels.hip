/* image_kernels.hip - the input side of the path:
   od_img_plane_copy_pad (reference src/encode.c:752-837; called for every plane
   by daala_image_copy_pad :1896-1909 when a frame enters the input queue,
   od_input_queue_add :272-287): copy the picture into the padded plane the
   encoder codes, then extend it into the padding by a [1 2 1]/4 low-pass of the
   previous column (right side, picture rows only) and of the previous row
   (bottom, the whole padded width).  8-bit planes (xstride 1), and - for an encoder with
   full-precision references - 16-bit planes at 12 bits (xstride 2, :791-803, :821-832),
   whose copy step is od_img_plane_copy's bit-depth conversion (src/state.c:93-213).

   k_img_copy   the picture region of every plane, 16 bytes per thread where the
                alignment allows
   k_img_pad    one workgroup per plane: the extension is a recurrence over the
                padded columns, then over the padded rows (at most 63 of each);
                the previous column / row lives in LDS, every step is parallel
                over the other dimension.  The last picture row's extension is
                kept in LDS for the bottom pass. */
#include "../../include/daala_hip.h"
#include "od_common.cuh"

namespace {

constexpr int kMaxDim = 8192;
constexpr int kPadThreads = 1024;

__global__ __launch_bounds__(256) void k_img_copy(uint8_t *dst, int dst_stride, long dst_plane_stride,
 const uint8_t *src, int src_stride, long src_plane_stride, int pic_w, int pic_h) {
  const int p = blockIdx.z;
  const int y = blockIdx.y;
  const int x = (blockIdx.x*256 + threadIdx.x)*16;
  if (x >= pic_w) return;
  const uint8_t *s = src + p*src_plane_stride + (long)y*src_stride + x;
  uint8_t *d = dst + p*dst_plane_stride + (long)y*dst_stride + x;
  if (x + 16 <= pic_w && (((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
    *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(s);
  }
  else {
    const int n = pic_w - x < 16 ? pic_w - x : 16;
    for (int i = 0; i < n; i++) d[i] = s[i];
  }
}

/* Full-precision references: an 8-bit source (uint8_t samples) or a 10 / 12-bit one
   (int16_t samples) shifted up into 12 bits and clamped, src/state.c:141-160 / :186-197. */
__global__ __launch_bounds__(256) void k_img_copy16(uint16_t *dst, int dst_stride, long dst_plane_stride,
 const void *src, int src_bitdepth, int src_stride, long src_plane_stride, int pic_w, int pic_h) {
  const int p = blockIdx.z;
  const int y = blockIdx.y;
  const int x = blockIdx.x*256 + threadIdx.x;
  if (x >= pic_w) return;
  const long at = p*src_plane_stride + (long)y*src_stride + x;
  const int v = src_bitdepth > 8 ? static_cast<const int16_t *>(src)[at] << (12 - src_bitdepth)
   : static_cast<const uint8_t *>(src)[at] << 4;
  dst[p*dst_plane_stride + (long)y*dst_stride + x] = (uint16_t)min(max(v, 0), 4095);
}

template <typename S>
__global__ __launch_bounds__(kPadThreads) void k_img_pad(S *dst, int dst_stride,
 long dst_plane_stride, int plane_w, int plane_h, int pic_w, int pic_h) {
  __shared__ S buf[2][kMaxDim];
  __shared__ S last_row[64];     /* extension of picture row pic_h - 1 */
  S *d = dst + blockIdx.x*dst_plane_stride;
  const int tid = threadIdx.x;
  if (pic_w == 0 || pic_h == 0) {
    for (long i = tid; i < (long)plane_w*plane_h; i += kPadThreads) {
      d[(i/plane_w)*dst_stride + i%plane_w] = 0;
    }
    return;
  }
  /* right side, src/encode.c:778-806 */
  int cur = 0;
  if (pic_w < plane_w) {
    for (int y = tid; y < pic_h; y += kPadThreads) buf[0][y] = d[(long)y*dst_stride + pic_w - 1];
    __syncthreads();
    for (int x = pic_w; x < plane_w; x++) {
      for (int y = tid; y < pic_h; y += kPadThreads) {
        const int c = buf[cur][y];
        const int u = buf[cur][y > 0 ? y - 1 : y];
        const int dn = buf[cur][y + 1 < pic_h ? y + 1 : y];
        const S v = (S)((2*c + u + dn + 2) >> 2);
        buf[cur ^ 1][y] = v;
        d[(long)y*dst_stride + x] = v;
        if (y == pic_h - 1) last_row[x - pic_w] = v;
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  /* bottom, :808-834 */
  if (pic_h < plane_h) {
    cur = 0;
    for (int x = tid; x < plane_w; x += kPadThreads) {
      buf[0][x] = x < pic_w ? d[(long)(pic_h - 1)*dst_stride + x] : last_row[x - pic_w];
    }
    __syncthreads();
    for (int y = pic_h; y < plane_h; y++) {
      for (int x = tid; x < plane_w; x += kPadThreads) {
        const int c = buf[cur][x];
        const int l = buf[cur][x - (x > 0)];
        const int r = buf[cur][x + (x + 1 < plane_w)];
        const S v = (S)((2*c + l + r + 2) >> 2);
        buf[cur ^ 1][x] = v;
        d[(long)y*dst_stride + x] = v;
      }
      __syncthreads();
      cur ^= 1;
    }
  }
}

}  // namespace

extern "C" int odhip_image_planes_copy_pad(uint8_t *d_dst, int dst_stride, long dst_plane_stride,
 int plane_w, int plane_h, const uint8_t *d_src, int src_stride, long src_plane_stride, int pic_w,
 int pic_h, int nplanes, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  if (nplanes == 0) return ODHIP_SUCCESS;
  if (!d_dst || nplanes < 0 || plane_w <= 0 || plane_h <= 0 || plane_w > kMaxDim || plane_h > kMaxDim
   || pic_w < 0 || pic_h < 0 || pic_w > plane_w || pic_h > plane_h || dst_stride < plane_w) {
    return ODHIP_EINVAL;
  }
  /* last_row holds one superblock of padded columns */
  if (pic_w > 0 && pic_h > 0 && plane_w - pic_w > 64) return ODHIP_EINVAL;
  if (pic_w > 0 && pic_h > 0) {
    if (!d_src || src_stride < pic_w) return ODHIP_EINVAL;
    const dim3 grid((unsigned)((pic_w + 16*256 - 1)/(16*256)), (unsigned)pic_h, (unsigned)nplanes);
    if (grid.y > 65535u || grid.z > 65535u) return ODHIP_EINVAL;
    k_img_copy<<<grid, 256, 0, s>>>(d_dst, dst_stride, dst_plane_stride, d_src, src_stride,
     src_plane_stride, pic_w, pic_h);
  }
  if (pic_w == 0 || pic_h == 0 || pic_w < plane_w || pic_h < plane_h) {
    k_img_pad<uint8_t><<<(unsigned)nplanes, kPadThreads, 0, s>>>(d_dst, dst_stride, dst_plane_stride, plane_w,
     plane_h, pic_w, pic_h);
  }
  return odhip_check_launch();
}

/* The same for full-precision references: d_dst holds uint16_t samples at 12 bits; the
   source is `src_bitdepth` 8 (uint8_t samples) or 10 / 12 (int16_t samples).  Strides in
   samples. */
extern "C" int odhip_image_planes_copy_pad16(uint16_t *d_dst, int dst_stride, long dst_plane_stride,
 int plane_w, int plane_h, const void *d_src, int src_bitdepth, int src_stride, long src_plane_stride,
 int pic_w, int pic_h, int nplanes, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  if (nplanes == 0) return ODHIP_SUCCESS;
  if (!d_dst || nplanes < 0 || plane_w <= 0 || plane_h <= 0 || plane_w > kMaxDim || plane_h > kMaxDim
   || pic_w < 0 || pic_h < 0 || pic_w > plane_w || pic_h > plane_h || dst_stride < plane_w
   || (src_bitdepth != 8 && src_bitdepth != 10 && src_bitdepth != 12)) {
    return ODHIP_EINVAL;
  }
  if (pic_w > 0 && pic_h > 0 && plane_w - pic_w > 64) return ODHIP_EINVAL;
  if (pic_w > 0 && pic_h > 0) {
    if (!d_src || src_stride < pic_w) return ODHIP_EINVAL;
    const dim3 grid((unsigned)((pic_w + 255)/256), (unsigned)pic_h, (unsigned)nplanes);
    if (grid.y > 65535u || grid.z > 65535u) return ODHIP_EINVAL;
    k_img_copy16<<<grid, 256, 0, s>>>(d_dst, dst_stride, dst_plane_stride, d_src, src_bitdepth, src_stride,
     src_plane_stride, pic_w, pic_h);
  }
  if (pic_w == 0 || pic_h == 0 || pic_w < plane_w || pic_h < plane_h) {
    k_img_pad<uint16_t><<<(unsigned)nplanes, kPadThreads, 0, s>>>(d_dst, dst_stride, dst_plane_stride, plane_w,
     plane_h, pic_w, pic_h);
  }
  return odhip_check_launch();
}

/* ---- the practical HBM ceiling of this GPU, measured by the library itself ----
   A device-to-device copy in 16-byte vectors: U vectors per lane a whole workgroup stride apart (every
   instruction of a wavefront touches 1 KB of consecutive bytes), all U loads in flight before the first
   store.  What the filter + DCT rooflines are read against beside the 8 TB/s specification
   (bench.py `copy_ceiling`): round 5 used a torch copy_ (4.8 TB/s), which this repository's own 4x4
   transform kernel exceeds. */
namespace {
template <int U>
__global__ __launch_bounds__(256) void k_copy16(const int4 *__restrict__ src, int4 *__restrict__ dst, size_t nvec) {
  const size_t base = (size_t)blockIdx.x*(256*U) + threadIdx.x;
  if (base - threadIdx.x + (size_t)256*U <= nvec) {
    /* whole block inside the buffer (workgroup-uniform): the U vectors stay in registers */
    int4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = src[base + (size_t)u*256];
#pragma unroll
    for (int u = 0; u < U; u++) dst[base + (size_t)u*256] = v[u];
  }
  else {
    for (int u = 0; u < U; u++) {
      const size_t i = base + (size_t)u*256;
      if (i < nvec) dst[i] = src[i];
    }
  }
}
}  // namespace

/* Copies `bytes` (a multiple of 16, buffers owned by this call) n times after one warm-up and reports
   (read + written bytes) / time in GB/s; variant 0..2 = 2 / 4 / 8 vectors per lane. */
extern "C" int odhip_copy_ceiling(size_t bytes, int n, int variant, double *gbs, odhip_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!gbs || n <= 0 || bytes < 16 || (bytes & 15) || variant < 0 || variant > 2) return ODHIP_EINVAL;
  int4 *a = nullptr;
  int4 *b = nullptr;
  if (hipMalloc((void **)&a, bytes) != hipSuccess) return ODHIP_EFAULT;
  if (hipMalloc((void **)&b, bytes) != hipSuccess) {
    (void)hipFree(a);
    return ODHIP_EFAULT;
  }
  hipEvent_t e0 = nullptr;
  hipEvent_t e1 = nullptr;
  int rc = ODHIP_SUCCESS;
  const size_t nvec = bytes/16;
  auto launch = [&]() {
    const int u = 2 << variant;
    const unsigned grid = (unsigned)((nvec + (size_t)256*u - 1)/((size_t)256*u));
    if (variant == 0) k_copy16<2><<<grid, 256, 0, s>>>(a, b, nvec);
    else if (variant == 1) k_copy16<4><<<grid, 256, 0, s>>>(a, b, nvec);
    else k_copy16<8><<<grid, 256, 0, s>>>(a, b, nvec);
  };
  if (hipMemsetAsync(a, 1, bytes, s) != hipSuccess || hipEventCreate(&e0) != hipSuccess
   || hipEventCreate(&e1) != hipSuccess) {
    rc = ODHIP_EFAULT;
  }
  if (!rc) {
    launch();
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < n; i++) launch();
    (void)hipEventRecord(e1, s);
    float ms = 0;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0) rc = ODHIP_EFAULT;
    else *gbs = 2.*(double)bytes*n/(ms*1e-3)/1e9;
    if (!rc) rc = odhip_check_launch();
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  return rc;
}
