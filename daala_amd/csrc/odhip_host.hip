/* odhip_host.hip - the per-call, host-pointer half of the C ABI.

   These entry points have exactly the reference's signatures and calling
   conventions (synchronous, caller-owned host memory, void return) so they can
   be bound into the reference's function-pointer tables and call sites
   unchanged; each one stages its operands into device scratch, runs the SAME
   kernels as the batched API, and copies the result back.  They are the
   drop-in / parity surface - one block per call cannot be fast.

   There is no CPU fallback: if HIP fails the process aborts with a message,
   as the reference's void surfaces leave no error channel. */
#include <stdlib.h>
#include <string.h>
#include "../../include/daala_hip.h"
#include "od_common.cuh"
#include "od_lift.cuh"

namespace {

void die(const char *what, hipError_t e) {
  fprintf(stderr, "libdaalahip: fatal: %s: %s (no CPU fallback)\n", what,
   hipGetErrorString(e));
  abort();
}

#define HIP_OR_DIE(expr) \
  do { \
    hipError_t e_ = (expr); \
    if (e_ != hipSuccess) die(#expr, e_); \
  } while (0)

void ok_or_die(int rc, const char *what) {
  if (rc != ODHIP_SUCCESS) {
    fprintf(stderr, "libdaalahip: fatal: %s returned %d (no CPU fallback)\n", what, rc);
    abort();
  }
}

/* Per-thread scratch: the reference allows one encoder per host thread. */
struct Scratch {
  hipStream_t stream = nullptr;
  char *buf[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  void *get(int i, size_t bytes) {
    if (!stream) HIP_OR_DIE(hipStreamCreate(&stream));
    if (bytes > cap[i]) {
      if (buf[i]) HIP_OR_DIE(hipFree(buf[i]));
      size_t want = bytes < 65536 ? 65536 : bytes;
      HIP_OR_DIE(hipMalloc((void **)&buf[i], want));
      cap[i] = want;
    }
    return buf[i];
  }
};
thread_local Scratch g_scratch;

void dct_call(bool inverse, int ln, od_coeff *out, int out_stride,
 const od_coeff *in, int in_stride) {
  const int n = 4 << ln;
  Scratch &s = g_scratch;
  od_coeff *d_in = (od_coeff *)s.get(0, (size_t)n*n*sizeof(od_coeff));
  od_coeff *d_out = (od_coeff *)s.get(1, (size_t)n*n*sizeof(od_coeff));
  HIP_OR_DIE(hipMemcpy2DAsync(d_in, n*sizeof(od_coeff), in, in_stride*sizeof(od_coeff),
   n*sizeof(od_coeff), n, hipMemcpyHostToDevice, s.stream));
  ok_or_die(inverse ? odhip_idct2d_batch(ln, d_out, d_in, 1, 1, s.stream)
                    : odhip_fdct2d_batch(ln, d_out, d_in, 1, 1, s.stream), "dct2d");
  HIP_OR_DIE(hipMemcpy2DAsync(out, out_stride*sizeof(od_coeff), d_out, n*sizeof(od_coeff),
   n*sizeof(od_coeff), n, hipMemcpyDeviceToHost, s.stream));
  HIP_OR_DIE(hipStreamSynchronize(s.stream));
}

/* One 4-tap lapping filter per thread: tap k of task i of edge e lives at
   first + e*edge_step + i*task_step + k*tap_step. */
template <bool INV>
__global__ void k_filter_edges(od_coeff *c, long first, long edge_step, int ntasks,
 long task_step, long tap_step) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= ntasks) return;
  od_coeff *p = c + first + blockIdx.y*edge_step + i*task_step;
  int t0 = p[0];
  int t1 = p[tap_step];
  int t2 = p[2*tap_step];
  int t3 = p[3*tap_step];
  if (INV) od_post_filter4_dev(t0, t1, t2, t3);
  else od_pre_filter4_dev(t0, t1, t2, t3);
  p[0] = t0;
  p[tap_step] = t1;
  p[2*tap_step] = t2;
  p[3*tap_step] = t3;
}

template <bool INV>
void filter_edges(od_coeff *d, long first, long edge_step, int nedges, int ntasks,
 long task_step, long tap_step, hipStream_t s) {
  if (nedges <= 0 || ntasks <= 0) return;
  const dim3 grid((ntasks + 255)/256, nedges);
  k_filter_edges<INV><<<grid, 256, 0, s>>>(d, first, edge_step, ntasks, task_step, tap_step);
  ok_or_die(odhip_check_launch(), "k_filter_edges");
}

/* Round trip of a host sub-plane (rows x cols, host stride) through device
   scratch around `body`, which sees a compact device plane of stride cols. */
template <typename F>
void with_device_plane(od_coeff *c, int stride, int rows, int cols, F body) {
  Scratch &s = g_scratch;
  od_coeff *d = (od_coeff *)s.get(0, (size_t)rows*cols*sizeof(od_coeff));
  HIP_OR_DIE(hipMemcpy2DAsync(d, cols*sizeof(od_coeff), c, stride*sizeof(od_coeff),
   cols*sizeof(od_coeff), rows, hipMemcpyHostToDevice, s.stream));
  body(d, s.stream);
  HIP_OR_DIE(hipMemcpy2DAsync(c, stride*sizeof(od_coeff), d, cols*sizeof(od_coeff),
   cols*sizeof(od_coeff), rows, hipMemcpyDeviceToHost, s.stream));
  HIP_OR_DIE(hipStreamSynchronize(s.stream));
}

void require_f0(int f) {
  if (f != 0) {
    fprintf(stderr, "libdaalahip: fatal: lapping filter size f=%d requested; the codec "
     "only uses the 4-point filter (OD_FILT_SIZE == 0)\n", f);
    abort();
  }
}

}  // namespace

extern "C" {

int odhip_init(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    fprintf(stderr, "libdaalahip: no HIP device available (no CPU fallback)\n");
    return ODHIP_EFAULT;
  }
  if (device < 0 || device >= count) return ODHIP_EINVAL;
  ODHIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  ODHIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fprintf(stderr, "libdaalahip: device %d is %s; this library is built for gfx950 only\n",
     device, prop.gcnArchName);
    return ODHIP_EFAULT;
  }
  return ODHIP_SUCCESS;
}

const char *odhip_version(void) {
  return "libdaalahip 0.1 (gfx950; lapped DCT pyramid + PVQ search)";
}

#define ODHIP_DCT_PAIR(n, ln) \
  void od_bin_fdct##n##x##n##_hip(od_coeff *y, int ystride, const od_coeff *x, int xstride) { \
    dct_call(false, ln, y, ystride, x, xstride); \
  } \
  void od_bin_idct##n##x##n##_hip(od_coeff *x, int xstride, const od_coeff *y, int ystride) { \
    dct_call(true, ln, x, xstride, y, ystride); \
  }
ODHIP_DCT_PAIR(4, 0)
ODHIP_DCT_PAIR(8, 1)
ODHIP_DCT_PAIR(16, 2)
ODHIP_DCT_PAIR(32, 3)
ODHIP_DCT_PAIR(64, 4)

void odhip_install_dct_vtbl(odhip_dct_func_2d fdct_2d[ODHIP_NBSIZES],
 odhip_dct_func_2d idct_2d[ODHIP_NBSIZES]) {
  fdct_2d[0] = od_bin_fdct4x4_hip;
  fdct_2d[1] = od_bin_fdct8x8_hip;
  fdct_2d[2] = od_bin_fdct16x16_hip;
  fdct_2d[3] = od_bin_fdct32x32_hip;
  fdct_2d[4] = od_bin_fdct64x64_hip;
  idct_2d[0] = od_bin_idct4x4_hip;
  idct_2d[1] = od_bin_idct8x8_hip;
  idct_2d[2] = od_bin_idct16x16_hip;
  idct_2d[3] = od_bin_idct32x32_hip;
  idct_2d[4] = od_bin_idct64x64_hip;
}

void od_pre_filter4_hip(od_coeff y[4], const od_coeff x[4]) {
  od_coeff t[4];
  memcpy(t, x, sizeof(t));
  with_device_plane(t, 4, 1, 4, [](od_coeff *d, hipStream_t s) {
    filter_edges<false>(d, 0, 0, 1, 1, 0, 1, s);
  });
  memcpy(y, t, sizeof(t));
}

void od_post_filter4_hip(od_coeff x[4], const od_coeff y[4]) {
  od_coeff t[4];
  memcpy(t, y, sizeof(t));
  with_device_plane(t, 4, 1, 4, [](od_coeff *d, hipStream_t s) {
    filter_edges<true>(d, 0, 0, 1, 1, 0, 1, s);
  });
  memcpy(x, t, sizeof(t));
}

namespace {

void filter_call(int f, int inverse, od_coeff *out, const od_coeff *in) {
  const int n = 4 << f;
  Scratch &s = g_scratch;
  od_coeff *d_in = (od_coeff *)s.get(0, n*sizeof(od_coeff));
  od_coeff *d_out = (od_coeff *)s.get(1, n*sizeof(od_coeff));
  HIP_OR_DIE(hipMemcpyAsync(d_in, in, n*sizeof(od_coeff), hipMemcpyHostToDevice, s.stream));
  ok_or_die(odhip_filter_batch(f, inverse, d_out, d_in, 1, s.stream), "filter_batch");
  HIP_OR_DIE(hipMemcpyAsync(out, d_out, n*sizeof(od_coeff), hipMemcpyDeviceToHost, s.stream));
  HIP_OR_DIE(hipStreamSynchronize(s.stream));
}

}  // namespace

/* od_pre_filter8/16/32, od_post_filter8/16/32 (src/filter.c:279-1321): unused by
   the codec (OD_FILT_SIZE == 0) but part of OD_PRE_FILTER[] / OD_POST_FILTER[]. */
void od_pre_filter8_hip(od_coeff y[8], const od_coeff x[8]) { filter_call(1, 0, y, x); }
void od_post_filter8_hip(od_coeff x[8], const od_coeff y[8]) { filter_call(1, 1, x, y); }
void od_pre_filter16_hip(od_coeff y[16], const od_coeff x[16]) { filter_call(2, 0, y, x); }
void od_post_filter16_hip(od_coeff x[16], const od_coeff y[16]) { filter_call(2, 1, x, y); }
void od_pre_filter32_hip(od_coeff y[32], const od_coeff x[32]) { filter_call(3, 0, y, x); }
void od_post_filter32_hip(od_coeff x[32], const od_coeff y[32]) { filter_call(3, 1, x, y); }

void odhip_install_filter_tables(odhip_filter_func pre[4], odhip_filter_func post[4]) {
  pre[0] = od_pre_filter4_hip;
  pre[1] = od_pre_filter8_hip;
  pre[2] = od_pre_filter16_hip;
  pre[3] = od_pre_filter32_hip;
  post[0] = od_post_filter4_hip;
  post[1] = od_post_filter8_hip;
  post[2] = od_post_filter16_hip;
  post[3] = od_post_filter32_hip;
}

/* od_prefilter_split, src/filter.c:1459-1483: columns (hfilter) then rows. */
void od_prefilter_split_hip(od_coeff *c0, int stride, int bs, int f, int hfilter,
 int vfilter) {
  require_f0(f);
  const int n = 4 << bs;
  with_device_plane(c0, stride, n, n, [=](od_coeff *d, hipStream_t s) {
    if (hfilter) filter_edges<false>(d, (long)(n/2 - 2)*n, 0, 1, n, 1, n, s);
    if (vfilter) filter_edges<false>(d, n/2 - 2, 0, 1, n, n, 1, s);
  });
}

/* od_postfilter_split, src/filter.c:1485-1527: rows (vfilter) then columns. */
void od_postfilter_split_hip(od_coeff *c0, int stride, int bs, int f, int q,
 unsigned char *skip, int skip_stride, int hfilter, int vfilter) {
  (void)q;
  (void)skip;
  (void)skip_stride;
  require_f0(f);
  const int n = 4 << bs;
  with_device_plane(c0, stride, n, n, [=](od_coeff *d, hipStream_t s) {
    if (vfilter) filter_edges<true>(d, n/2 - 2, 0, 1, n, n, 1, s);
    if (hfilter) filter_edges<true>(d, (long)(n/2 - 2)*n, 0, 1, n, 1, n, s);
  });
}

/* od_apply_prefilter_frame_sbs, src/filter.c:1529-1559. */
void od_apply_prefilter_frame_sbs_hip(od_coeff *c, int stride, int nhsb, int nvsb,
 int xdec, int ydec) {
  const int sbw = 64 >> xdec;
  const int sbh = 64 >> ydec;
  const int w = nhsb*sbw;
  const int h = nvsb*sbh;
  with_device_plane(c, stride, h, w, [=](od_coeff *d, hipStream_t s) {
    filter_edges<false>(d, (long)(sbh - 2)*w, (long)sbh*w, nvsb - 1, w, 1, w, s);
    filter_edges<false>(d, sbw - 2, sbw, nhsb - 1, h, w, 1, s);
  });
}

/* od_apply_postfilter_frame_sbs, src/filter.c:1589-1618. */
void od_apply_postfilter_frame_sbs_hip(od_coeff *c, int stride, int nhsb, int nvsb,
 int xdec, int ydec, int q, unsigned char *skip, int skip_stride) {
  (void)q;
  (void)skip;
  (void)skip_stride;
  const int sbw = 64 >> xdec;
  const int sbh = 64 >> ydec;
  const int w = nhsb*sbw;
  const int h = nvsb*sbh;
  with_device_plane(c, stride, h, w, [=](od_coeff *d, hipStream_t s) {
    filter_edges<true>(d, sbw - 2, sbw, nhsb - 1, h, w, 1, s);
    filter_edges<true>(d, (long)(sbh - 2)*w, (long)sbh*w, nvsb - 1, w, 1, w, s);
  });
}

/* od_compute_dist (src/encode.c:1170-1226; file-static there, call sites :1418-1421,
   :1797-1798) for ONE block, host pointers, synchronous: x (source) and y (reconstruction)
   are compact n x n blocks (stride n) as the encoder's c_orig / split / nosplit buffers
   hold them; the device computes every 8x8 block's three doubles up to the pow
   (odhip_dist_parts), the host finishes with its libm (odhip_dist_finish).  The arguments
   the reference reads from enc are passed: enc->use_activity_masking, enc->qm ==
   OD_FLAT_QM, enc->state.coded_quantizer. */
double od_compute_dist_hip(const od_coeff *x, const od_coeff *y, int n, int use_masking, int flat_qm,
 int coded_quantizer) {
  int bs = 0;
  while (bs < ODHIP_NBSIZES && (4 << bs) != n) bs++;
  if (bs < 1 || bs >= ODHIP_NBSIZES) {
    fprintf(stderr, "libdaalahip: fatal: od_compute_dist_hip n=%d (8, 16, 32 or 64)\n", n);
    abort();
  }
  Scratch &s = g_scratch;
  const size_t blk = (size_t)n*n*sizeof(od_coeff);
  const size_t nparts = (size_t)(n/8)*(n/8)*3;
  char *d = (char *)s.get(0, 2*blk + nparts*sizeof(double));
  HIP_OR_DIE(hipMemcpyAsync(d, x, blk, hipMemcpyHostToDevice, s.stream));
  HIP_OR_DIE(hipMemcpyAsync(d + blk, y, blk, hipMemcpyHostToDevice, s.stream));
  double *d_parts = (double *)(d + 2*blk);
  ok_or_die(odhip_dist_parts(d_parts, (const od_coeff *)d, (const od_coeff *)(d + blk), 1, n, n, bs,
   use_masking, flat_qm, s.stream), "odhip_dist_parts");
  double parts[8*8*3];
  HIP_OR_DIE(hipMemcpyAsync(parts, d_parts, nparts*sizeof(double), hipMemcpyDeviceToHost, s.stream));
  HIP_OR_DIE(hipStreamSynchronize(s.stream));
  double dist = 0;
  ok_or_die(odhip_dist_finish(&dist, parts, 1, n, n, bs, use_masking, flat_qm, coded_quantizer),
   "odhip_dist_finish");
  return dist;
}

double od_pvq_search_rdo_double_hip(const int16_t *xcoeff, int n, int k,
 od_coeff *ypulse, double g2, double pvq_norm_lambda, int prev_k) {
  if (n < 1 || n > 128) {
    fprintf(stderr, "libdaalahip: fatal: pvq search n=%d out of range\n", n);
    abort();
  }
  Scratch &s = g_scratch;
  /* layout: x[n] int16 | pad | y[n] int32 | k | prev_k | g2 | cos */
  const size_t off_y = 256;
  const size_t off_k = off_y + 128*sizeof(od_coeff);
  const size_t off_g2 = off_k + 16;
  const size_t off_cos = off_g2 + 8;
  char host[1024];
  memset(host, 0, sizeof(host));
  memcpy(host, xcoeff, n*sizeof(int16_t));
  if (prev_k > 0) memcpy(host + off_y, ypulse, n*sizeof(od_coeff));
  ((int32_t *)(host + off_k))[0] = k;
  ((int32_t *)(host + off_k))[1] = prev_k;
  *(double *)(host + off_g2) = g2;
  char *d = (char *)s.get(0, sizeof(host));
  HIP_OR_DIE(hipMemcpyAsync(d, host, sizeof(host), hipMemcpyHostToDevice, s.stream));
  ok_or_die(odhip_pvq_search_batch((const int16_t *)d, n, (const int32_t *)(d + off_k),
   (od_coeff *)(d + off_y), (const double *)(d + off_g2), pvq_norm_lambda,
   (const int32_t *)(d + off_k + 4), (double *)(d + off_cos), 1, s.stream),
   "odhip_pvq_search_batch");
  HIP_OR_DIE(hipMemcpyAsync(host, d, sizeof(host), hipMemcpyDeviceToHost, s.stream));
  HIP_OR_DIE(hipStreamSynchronize(s.stream));
  memcpy(ypulse, host + off_y, n*sizeof(od_coeff));
  return *(double *)(host + off_cos);
}

/* od_pvq_synthesis_partial (src/pvq.c:1037-1115, declared src/pvq.h:164) with the reference's own signature: what
   pvq_decode_partition's synthesis (src/pvq_decoder.c:77-89, od_pvq_decode :300-376 per band) and pvq_theta's
   (src/pvq_encoder.c:631) call once per coded band.  One band per call through odhip_pvq_synthesis: a parity surface
   (the decoder of a whole frame at once is odhip_pvq_decode_bands). */
void od_pvq_synthesis_partial_hip(od_coeff *xcoeff, const od_coeff *ypulse, const int16_t *r16, int n, int noref,
 int32_t g, int32_t theta, int m, int s_, const int16_t *qm_inv) {
  if (n < 2 || n > 128) {
    fprintf(stderr, "libdaalahip: fatal: pvq synthesis n=%d out of range\n", n);
    abort();
  }
  Scratch &s = g_scratch;
  /* layout: y[128] int32 | r16[128] int16 | qm_inv[128] int16 | params[5] int32 | pad | out[128] int32 */
  const size_t off_r = 128*sizeof(od_coeff);
  const size_t off_q = off_r + 128*sizeof(int16_t);
  const size_t off_p = off_q + 128*sizeof(int16_t);
  const size_t off_o = off_p + 32;
  char host[128*4 + 128*2 + 128*2 + 32 + 128*4];
  memset(host, 0, sizeof(host));
  memcpy(host, ypulse, (size_t)(n - !noref)*sizeof(od_coeff));
  if (!noref) memcpy(host + off_r, r16, n*sizeof(int16_t));
  memcpy(host + off_q, qm_inv, n*sizeof(int16_t));
  int32_t *pr = (int32_t *)(host + off_p);
  pr[0] = noref;
  pr[1] = g;
  pr[2] = theta;
  pr[3] = m;
  pr[4] = s_;
  char *d = (char *)s.get(0, sizeof(host));
  HIP_OR_DIE(hipMemcpyAsync(d, host, off_o, hipMemcpyHostToDevice, s.stream));
  ok_or_die(odhip_pvq_synthesis((od_coeff *)(d + off_o), (const od_coeff *)d, (const int16_t *)(d + off_r), n, 1,
   (const int32_t *)(d + off_p), (const int16_t *)(d + off_q), s.stream), "odhip_pvq_synthesis");
  HIP_OR_DIE(hipMemcpyAsync(host + off_o, d + off_o, n*sizeof(od_coeff), hipMemcpyDeviceToHost, s.stream));
  HIP_OR_DIE(hipStreamSynchronize(s.stream));
  memcpy(xcoeff, host + off_o, n*sizeof(od_coeff));
}

/* Host-pointer form of odhip_inverse_partition for one plane (the decoder-side check
   of tests/interpose: the reference decoder's dtmp plane and bsize map in, pixels
   out): stages through device scratch, synchronous. */
int odhip_inverse_partition_host(uint8_t *px, int px_stride, const od_coeff *coef, int w, int h, int dec,
 const uint8_t *bsize, int bstride, int pic_w, int pic_h) {
  if (!px || !coef || !bsize || w <= 0 || h <= 0 || (dec != 0 && dec != 1) || px_stride < w) {
    return ODHIP_EINVAL;
  }
  const int tile = 64 >> dec;
  if (w % tile || h % tile) return ODHIP_EINVAL;
  Scratch &s = g_scratch;
  const size_t cbytes = (size_t)w*h*sizeof(od_coeff);
  const int mrows = (h/tile)*8;
  const int mcols = (w/tile)*8;
  const size_t mbytes = (size_t)mrows*mcols;
  /* buffer 0: coefficients, then the compact map; buffer 1: pixels */
  char *d0 = (char *)s.get(0, cbytes + mbytes + 16);
  uint8_t *d_px = (uint8_t *)s.get(1, (size_t)w*h);
  HIP_OR_DIE(hipMemcpyAsync(d0, coef, cbytes, hipMemcpyHostToDevice, s.stream));
  HIP_OR_DIE(hipMemcpy2DAsync(d0 + cbytes, mcols, bsize, bstride, mcols, mrows, hipMemcpyHostToDevice,
   s.stream));
  const int rc = odhip_inverse_partition(d_px, w, (long)w*h, (const od_coeff *)d0, 1, w, h, dec,
   (const uint8_t *)(d0 + cbytes), mcols, 0, 1, pic_w, pic_h, s.stream);
  if (rc) return rc;
  HIP_OR_DIE(hipMemcpy2DAsync(px, px_stride, d_px, w, w, h, hipMemcpyDeviceToHost, s.stream));
  HIP_OR_DIE(hipStreamSynchronize(s.stream));
  return ODHIP_SUCCESS;
}

}  /* extern "C" */
