/* dist_kernels.hip - od_compute_dist, the block-size RDO's distortion (reference
   src/encode.c:1082-1226), for every n x n block of a batch of
   plane pairs, split where the reference's arithmetic stops being reproducible:

     GPU (odhip_dist_parts)   everything up to the libm call: x - y, the [1 5 1] low-pass in
                              both directions with its block-edge taps (:1185-1203), and per
                              8x8 block (od_compute_dist_8x8 :1113-1156) the sum of squared
                              filtered error, the 3 x 3 window variances of x and y
                              (od_compute_var_4x4 :1082-1103), mean_var / min_var and
                              vardist (sqrt and division are correctly rounded; sums in the
                              reference's order, one lane per 8x8 block)
     host (odhip_dist_finish) activity = calibration*pow(arg, -1/6) with the HOST libm - the
                              function the reference itself calls, not reproducible bit for
                              bit on the device, and its result feeds sequential RDO
                              comparisons, so no rounding argument applies (unlike acos) -
                              then activity^2*(sum + vardist), the sum over the block's 8x8
                              blocks in raster order and the quantiser-dependent factor
                              (:1158-1169, :1205-1220)

   Flat matrices (:1173-1179): the plain squared error, exact in any order. */
#include <math.h>
#include "../../include/daala_hip.h"
#include "od_common.cuh"

namespace {

constexpr int kTile = 64;
constexpr int kP = kTile + 1;

struct DistArgs {
  const od_coeff *x;
  const od_coeff *y;
  /* odhip_dist_parts_px16: x as 8-bit source samples ((p - 128) << 4, od_ref_buf_to_coeff,
     src/state.c:1231-1237) and y as the int16 plane od_dering writes, each with its own row stride */
  const uint8_t *x8;
  const int16_t *y16;
  int x8_stride;
  int y16_stride;
  double *parts;
  int w;
  int h;
  int n;
  int use_masking;
  int flat;
};

__device__ __forceinline__ int var_4x4(const int *p) {
  int sum = 0;
  int s2 = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int t = p[i*kP + j] >> 2;
      sum += t;
      s2 += t*t;
    }
  }
  return s2 - (sum*sum >> 4);
}

__global__ __launch_bounds__(256) void k_dist_parts(DistArgs a) {
  __shared__ int X[kTile*kP];
  __shared__ int Y[kTile*kP];
  __shared__ int E[kTile*kP];
  __shared__ int T[kTile*kP];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x*kTile;
  const int y0 = blockIdx.y*kTile;
  const long plane = (long)blockIdx.z*a.w*a.h;
  const int tw = a.w - x0 < kTile ? a.w - x0 : kTile;     /* planes are multiples of n, not of 64 */
  const int th = a.h - y0 < kTile ? a.h - y0 : kTile;
  for (int i = tid; i < kTile*kTile; i += 256) {
    const int r = i >> 6;
    const int c = i & 63;
    int xv = 0;
    int yv = 0;
    if (r < th && c < tw) {
      if (a.x8) {
        xv = ((int)a.x8[(long)blockIdx.z*a.x8_stride*a.h + (long)(y0 + r)*a.x8_stride + x0 + c] - 128) << 4;
        yv = a.y16[(long)blockIdx.z*a.y16_stride*a.h + (long)(y0 + r)*a.y16_stride + x0 + c];
      }
      else {
        const long g = plane + (long)(y0 + r)*a.w + x0 + c;
        xv = a.x[g];
        yv = a.y[g];
      }
    }
    X[r*kP + c] = xv;
    Y[r*kP + c] = yv;
    E[r*kP + c] = xv - yv;
  }
  __syncthreads();
  const int n = a.n;
  if (!a.flat) {
    /* horizontal taps, :1185-1191 */
    for (int i = tid; i < kTile*kTile; i += 256) {
      const int r = i >> 6;
      const int c = i & 63;
      const int j = c & (n - 1);
      const int e = E[r*kP + c];
      int t;
      if (j == 0) t = 5*e + 2*E[r*kP + c + 1];
      else if (j == n - 1) t = 5*e + 2*E[r*kP + c - 1];
      else t = 5*e + E[r*kP + c - 1] + E[r*kP + c + 1];
      T[r*kP + c] = t;
    }
    __syncthreads();
    /* vertical taps, :1192-1203 */
    for (int i = tid; i < kTile*kTile; i += 256) {
      const int r = i >> 6;
      const int c = i & 63;
      const int j = r & (n - 1);
      const int t = T[r*kP + c];
      int e;
      if (j == 0) e = 5*t + 2*T[(r + 1)*kP + c];
      else if (j == n - 1) e = 5*t + 2*T[(r - 1)*kP + c];
      else e = 5*t + T[(r - 1)*kP + c] + T[(r + 1)*kP + c];
      E[r*kP + c] = e;
    }
    __syncthreads();
  }
  /* one lane per 8x8 block of the tile */
  if (tid < 64) {
    const int br = tid >> 3;
    const int bc = tid & 7;
    if (br*8 < th && bc*8 < tw) {
      const int *e = E + br*8*kP + bc*8;
      double sum = 0;
      for (int i = 0; i < 8; i++) {
        for (int j = 0; j < 8; j++) {
          const double v = (double)e[i*kP + j];
          sum += v*v;
        }
      }
      double vardist = 0;
      double arg = 0;
      if (!a.flat) {
        int min_var = 0x7fffffff;
        double mean_var = 0;
        for (int i = 0; i < 3; i++) {
          for (int j = 0; j < 3; j++) {
            const int varx = var_4x4(X + (br*8 + 2*i)*kP + bc*8 + 2*j);
            const int vary = var_4x4(Y + (br*8 + 2*i)*kP + bc*8 + 2*j);
            min_var = varx < min_var ? varx : min_var;
            mean_var += __ddiv_rn(1., (double)(1 + varx));
            vardist += (varx - 2*__dsqrt_rn(varx*(double)vary)) + vary;
          }
        }
        arg = .25 + __ddiv_rn(a.use_masking ? __ddiv_rn(9., mean_var) : (double)min_var, 256.);
      }
      double *out = a.parts + (((long)blockIdx.z*(a.h >> 3) + (y0 >> 3) + br)*(a.w >> 3) + (x0 >> 3) + bc)*3;
      out[0] = sum;
      out[1] = vardist;
      out[2] = arg;
    }
  }
}

}  // namespace

extern "C" int odhip_dist_parts(double *d_parts, const od_coeff *d_x, const od_coeff *d_y, int nplanes,
 int w, int h, int bs, int use_masking, int flat_qm, odhip_stream stream) {
  if (!d_parts || !d_x || !d_y || nplanes <= 0 || bs < 1 || bs >= ODHIP_NBSIZES) return ODHIP_EINVAL;
  const int n = 4 << bs;
  if (w <= 0 || h <= 0 || w % n || h % n) return ODHIP_EINVAL;
  DistArgs a;
  a.x = d_x;
  a.y = d_y;
  a.x8 = nullptr;
  a.y16 = nullptr;
  a.x8_stride = a.y16_stride = 0;
  a.parts = d_parts;
  a.w = w;
  a.h = h;
  a.n = n;
  a.use_masking = use_masking != 0;
  a.flat = flat_qm != 0;
  const dim3 grid((w + kTile - 1)/kTile, (h + kTile - 1)/kTile, nplanes);
  if (grid.y > 65535u || grid.z > 65535u) return ODHIP_EINVAL;
  k_dist_parts<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return odhip_check_launch();
}

/* odhip_dist_parts with x given as 8-bit source samples and y as an int16 plane (what the deringing
   level search compares, src/encode.c:2776-2801: the source picture against od_dering's output). */
extern "C" int odhip_dist_parts_px16(double *d_parts, const uint8_t *d_x8, int x8_stride, const int16_t *d_y16,
 int y16_stride, int nplanes, int w, int h, int bs, int use_masking, int flat_qm, odhip_stream stream) {
  if (!d_parts || !d_x8 || !d_y16 || nplanes <= 0 || bs < 1 || bs >= ODHIP_NBSIZES || x8_stride < w
   || y16_stride < w) {
    return ODHIP_EINVAL;
  }
  const int n = 4 << bs;
  if (w <= 0 || h <= 0 || w % n || h % n) return ODHIP_EINVAL;
  DistArgs a;
  a.x = nullptr;
  a.y = nullptr;
  a.x8 = d_x8;
  a.y16 = d_y16;
  a.x8_stride = x8_stride;
  a.y16_stride = y16_stride;
  a.parts = d_parts;
  a.w = w;
  a.h = h;
  a.n = n;
  a.use_masking = use_masking != 0;
  a.flat = flat_qm != 0;
  const dim3 grid((w + kTile - 1)/kTile, (h + kTile - 1)/kTile, nplanes);
  if (grid.y > 65535u || grid.z > 65535u) return ODHIP_EINVAL;
  k_dist_parts<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return odhip_check_launch();
}

extern "C" int odhip_dist_finish(double *dist, const double *parts, int nplanes, int w, int h, int bs,
 int use_masking, int flat_qm, int coded_quantizer) {
  if (!dist || !parts || nplanes <= 0 || bs < 1 || bs >= ODHIP_NBSIZES) return ODHIP_EINVAL;
  const int n = 4 << bs;
  if (w <= 0 || h <= 0 || w % n || h % n) return ODHIP_EINVAL;
  const int w8 = w >> 3;
  const int h8 = h >> 3;
  const int m = n >> 3;
  const double calibration = use_masking ? 1.95 : 1.62;
  const double qfactor = coded_quantizer >= 47 ? 1.2 : coded_quantizer <= 36 ? 1.7
   : 1.7 + (1.2 - 1.7)*(coded_quantizer - 36)/(47 - 36);
  long blk = 0;
  for (int p = 0; p < nplanes; p++) {
    for (int by = 0; by < h/n; by++) {
      for (int bx = 0; bx < w/n; bx++) {
        double sum = 0;
        for (int i = 0; i < m; i++) {
          for (int j = 0; j < m; j++) {
            const double *q = parts + (((long)p*h8 + by*m + i)*w8 + bx*m + j)*3;
            if (flat_qm) sum += q[0];
            else {
              const double activity = calibration*pow(q[2], -1./6);
              const double s = q[0]*(0.92/(7*7*7*7));
              sum += activity*activity*(s + q[1]);
            }
          }
        }
        dist[blk++] = flat_qm ? sum : sum*qfactor;
      }
    }
  }
  return ODHIP_SUCCESS;
}
