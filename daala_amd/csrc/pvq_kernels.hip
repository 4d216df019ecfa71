/* pvq_kernels.hip - PVQ K-pulse search on gfx950 (fp64, bit-exact).

   Restates pvq_search_rdo_double (reference src/pvq_encoder.c:93-224).

   Mapping: ONE BAND PER LANE, 64 bands per wavefront.  The greedy argmax at
   src/pvq_encoder.c:172-183 compares cross-multiplied ROUNDED doubles
   (tmp_xy^2*best_yy > best_xy*tmp_yy, first index wins ties), which is not
   guaranteed transitive, so a tree reduction across lanes could pick a
   different pulse position than the reference's left-to-right scan.  A lane
   that owns its whole band simply runs the same scan in the same order:
   identical results by construction, no cross-lane traffic, and the wave's 64
   lanes are 64 independent bands.  |x| and y live in LDS as 16-bit values,
   laid out [j][lane] so every access of a wave is one contiguous 128-byte
   row (bank-conflict free).

   Every floating-point operation is a single IEEE-754 binary64 operation in
   the reference's order; this file MUST be compiled with -ffp-contract=off
   (hipcc contracts a*b+c into FMA by default, gcc -O2 on x86-64 does not).
   sqrt and division use the correctly rounded forms. */
#include "../../include/daala_hip.h"
#include "od_common.cuh"
#include "od_pvq_math.cuh"
#include "gen/od_scan_tables.h"

namespace {

constexpr int kWave = 64;

/* od_rsqrt_table, src/pvq_encoder.c:52-60: 6-digit decimal literals for
   i <= 16, 1/sqrt(i) beyond. */
__device__ const double kRsqrtTable[16] = {
  1.000000, 0.707107, 0.577350, 0.500000,
  0.447214, 0.408248, 0.377964, 0.353553,
  0.333333, 0.316228, 0.301511, 0.288675,
  0.277350, 0.267261, 0.258199, 0.250000};

__device__ __forceinline__ double od_rsqrt_table(int i) {
  if (i <= 16) return kRsqrtTable[i - 1];
  return __ddiv_rn(1., __dsqrt_rn((double)i));
}

/* One band per lane: the K-pulse search proper.  xs holds the band's SIGNED
   x (int16) and ys the pulse magnitudes, both laid out [j][64 lanes].  On
   entry with prev_k > 0 ys holds the magnitudes of the previous search.
   Returns the cosine distance; ys holds the new magnitudes (the caller
   restores signs, src/pvq_encoder.c:220-222). */
__device__ __forceinline__ double od_pvq_search_lane(const short *xs, unsigned short *ys,
 int lane, int n, int k, int prev_k, double g2, double pvq_norm_lambda) {
  double xx = 0;
  double xy = 0;
  double yy = 0;
  /* x[j] = fabs((float)xcoeff[j]): exact for int16. */
  for (int j = 0; j < n; j++) {
    const double xj = (double)abs((int)xs[j*kWave + lane]);
    xx += xj*xj;
  }
  const double norm_1 = __ddiv_rn(1., __dsqrt_rn(1e-30 + xx));
  const double lambda = __ddiv_rn(pvq_norm_lambda, 1e-30 + g2);
  int i = 0;
  if (prev_k > 0 && prev_k <= k) {
    for (int j = 0; j < n; j++) {
      const int yj = ys[j*kWave + lane];
      const double xj = (double)abs((int)xs[j*kWave + lane]);
      xy += xj*yj;
      yy += (double)(yj*yj);
      i += yj;
    }
  }
  else if (k > 2) {
    double l1_norm = 0;
    for (int j = 0; j < n; j++) l1_norm += (double)abs((int)xs[j*kWave + lane]);
    const double l1_inv = __ddiv_rn(1., l1_norm > 1e-100 ? l1_norm : 1e-100);
    for (int j = 0; j < n; j++) {
      const double xj = (double)abs((int)xs[j*kWave + lane]);
      const double tmp = (k*xj)*l1_inv;
      int yj = (int)floor(tmp);
      yj = yj > 0 ? yj : 0;
      ys[j*kWave + lane] = (unsigned short)yj;
      xy += xj*yj;
      yy += (double)(yj*yj);
      i += yj;
    }
  }
  else {
    for (int j = 0; j < n; j++) ys[j*kWave + lane] = 0;
  }
  const int rdo_pulses = 1 + k/4;
  double delta_rate = __ddiv_rn(3., (double)n);
  double accel_rate = 0.;
  if (k == 1) {
    if (n == 15) {
      accel_rate = __ddiv_rn(-8., (double)n);
      delta_rate = __ddiv_rn(4.5, (double)n) - accel_rate;
    }
    else if (n == 8) {
      accel_rate = __ddiv_rn(5.7, (double)n);
      delta_rate = __ddiv_rn(9.3, (double)n) - accel_rate;
    }
  }
  /* Greedy pulses, src/pvq_encoder.c:165-187. */
  for (; i < k - rdo_pulses; i++) {
    int pos = 0;
    double best_xy = -10;
    double best_yy = 1;
    for (int j = 0; j < n; j++) {
      double tmp_xy = xy + (double)abs((int)xs[j*kWave + lane]);
      const double tmp_yy = yy + (double)(2*ys[j*kWave + lane]) + 1;
      tmp_xy = tmp_xy*tmp_xy;
      if (j == 0 || tmp_xy*best_yy > best_xy*tmp_yy) {
        best_xy = tmp_xy;
        best_yy = tmp_yy;
        pos = j;
      }
    }
    const int yp_ = ys[pos*kWave + lane];
    xy = xy + (double)abs((int)xs[pos*kWave + lane]);
    yy = yy + (double)(2*yp_) + 1;
    ys[pos*kWave + lane] = (unsigned short)(yp_ + 1);
  }
  /* Last pulses with the rate term, src/pvq_encoder.c:192-219. */
  for (; i < k; i++) {
    double rsqrt_tab[4];
    for (int j = 0; j < 4; j++) rsqrt_tab[j] = od_rsqrt_table((int)(yy + 2*j + 1));
    int pos = 0;
    double best_cost = -1e5;
    for (int j = 0; j < n; j++) {
      double tmp_xy = xy + (double)abs((int)xs[j*kWave + lane]);
      const int yj = ys[j*kWave + lane];
      double tmp_yy;
      if (yj < 4) {
        tmp_yy = yj == 0 ? rsqrt_tab[0] : yj == 1 ? rsqrt_tab[1]
         : yj == 2 ? rsqrt_tab[2] : rsqrt_tab[3];
      }
      else tmp_yy = od_rsqrt_table((int)(yy + (double)(2*yj) + 1));
      tmp_xy = ((2*tmp_xy)*norm_1)*tmp_yy
       - (lambda*j)*(delta_rate + j*accel_rate);
      if (j == 0 || tmp_xy > best_cost) {
        best_cost = tmp_xy;
        pos = j;
      }
    }
    const int yp_ = ys[pos*kWave + lane];
    xy = xy + (double)abs((int)xs[pos*kWave + lane]);
    yy = yy + (double)(2*yp_) + 1;
    ys[pos*kWave + lane] = (unsigned short)(yp_ + 1);
  }
  return __ddiv_rn(xy, 1e-100 + __dsqrt_rn(xx*yy));
}

__global__ __launch_bounds__(kWave) void k_pvq_search(const int16_t *x_in, int n,
 const int32_t *k_in, od_coeff *y_io, const double *g2_in, double pvq_norm_lambda,
 const int32_t *prev_k_in, double *cos_out, long nbands) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  short *xs = (short *)lds;              /* x  [n][64] */
  unsigned short *ys = lds + n*kWave;    /* |y| [n][64] */
  const int lane = threadIdx.x;
  const long band = (long)blockIdx.x*kWave + lane;
  const bool live = band < nbands;
  const long b = live ? band : nbands - 1;
  const int16_t *xc = x_in + b*n;
  od_coeff *yp = y_io + b*n;
  const int k = k_in[b];
  const int prev_k = prev_k_in ? prev_k_in[b] : 0;
  for (int j = 0; j < n; j++) xs[j*kWave + lane] = xc[j];
  if (prev_k > 0 && prev_k <= k) {
    for (int j = 0; j < n; j++) {
      const int yj = yp[j];
      ys[j*kWave + lane] = (unsigned short)(yj < 0 ? -yj : yj);
    }
  }
  const double c = od_pvq_search_lane(xs, ys, lane, n, k, prev_k, g2_in[b], pvq_norm_lambda);
  if (live) {
    for (int j = 0; j < n; j++) {
      const int yj = ys[j*kWave + lane];
      yp[j] = xc[j] < 0 ? -yj : yj;
    }
    cos_out[band] = c;
  }
}

/* ---- per-band front end: pvq_theta's no-reference path ------------------- */

struct BandsArgs {
  const od_coeff *coef;   /* level plane(s), raster layout */
  const int16_t *qm;      /* QM in coding order for this (bs, decimation) */
  odhip_pvq_cands out;
  int nplanes;
  int w;
  int h;
  int bs;
  int nb_bands;
  int len;                /* coded coefficients per block: min(N*N, 512) */
  int q[ODHIP_MAX_BANDS];
  int beta[ODHIP_MAX_BANDS];
  int off[ODHIP_MAX_BANDS + 1];
  double lambda;
};

__constant__ unsigned char kScanXY[OD_SCAN_LEN][2];

/* pvq_theta, src/pvq_encoder.c:333-641, restricted to what is independent of
   the adaptive entropy coder on the no-reference path of a keyframe:
   :381,:398 (x16), :404 (gain), :417 (null distortion), :575-595 (both gain
   candidates: K, pruning test, search, distortion). */
__global__ __launch_bounds__(kWave) void k_pvq_bands_noref(BandsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int band = blockIdx.y;
  const int off = a.off[band];
  const int n = a.off[band + 1] - off;
  const int q = a.q[band];
  const int beta = a.beta[band];
  const int N = 4 << a.bs;
  const int bw = a.w/N;
  const int bh = a.h/N;
  const long nblocks = (long)a.nplanes*bw*bh;
  const int lane = threadIdx.x;
  const long blk = (long)blockIdx.x*kWave + lane;
  const bool live = blk < nblocks;
  const long b = live ? blk : nblocks - 1;
  const int p = (int)(b/((long)bw*bh));
  const int rem = (int)(b - (long)p*bw*bh);
  const int by = rem/bw;
  const int bx = rem - by*bw;
  const od_coeff *src = a.coef + (long)p*a.w*a.h + (long)by*N*a.w + bx*N;
  int *x0s = (int *)lds;                 /* [n][64] int32 stash, then ...   */
  short *xs = (short *)lds;              /* ... x16 [n][64] in its low half */
  unsigned short *ys = lds + n*kWave;    /* |y| [n][64]                     */
  /* od_vector_log_mag, src/pvq.c:472-484, while gathering the band in coding
     order (od_raster_to_coding_order, src/partition.c:144-170). */
  int sum = 0;
  for (int j = 0; j < n; j++) {
    const int v = src[kScanXY[off + j][1]*a.w + kScanXY[off + j][0]];
    x0s[j*kWave + lane] = v;
    const int t = (int16_t)(v >> 8);
    sum += t*t;
  }
  int xshift = 8 + 1 + odq_ilog(n + sum)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int acc = 0;
  for (int j = 0; j < n; j++) {
    const int v = x0s[j*kWave + lane];
    const int16_t x16 = (int16_t)odq_shr_round(v*a.qm[off + j], ODQ_QM_SHIFT + xshift);
    /* In-place narrowing: row j of the int16 view lies inside int32 row j/2,
       which every lane of this (single-wave) workgroup has already read. */
    xs[j*kWave + lane] = x16;
    acc += x16*(int)x16;
  }
  int32_t g;
  const int32_t cg = odq_gain_from_acc(acc, q, beta, xshift, &g);
  const double s2 = (1./256)*(1./256);  /* OD_CGAIN_SCALE_2 */
  const double dist0 = ((1.4*cg)*cg)*s2;
  const long sb = blk*a.nb_bands + band;
  if (live) {
    a.out.cg[sb] = cg;
    a.out.dist0[sb] = dist0;
  }
  const int gain_bound = cg >> ODQ_CGAIN_SHIFT;
  const int first = gain_bound > 1 ? gain_bound : 1;
  int prev_k = 0;
  for (int c = 0; c < 2; c++) {
    const int i = first + c;
    int flag = 0;
    int k = 0;
    double cos_dist = 0;
    double dist = 0;
    if (i <= gain_bound + 1) {
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT);
      k = odq_compute_k_noref(qcg, n, beta);
      dist = ((1.4*(qcg - cg))*(qcg - cg))*s2;
      if (!(dist > dist0 && k != 0)) {
        flag = 1;
        cos_dist = od_pvq_search_lane(xs, ys, lane, n, k, prev_k,
         (qcg*(double)cg)*s2, a.lambda);
        prev_k = k;
        dist = ((1.4*(qcg - cg))*(qcg - cg) + (qcg*(double)cg)*(2 - 2*cos_dist))*s2;
      }
    }
    if (live) {
      a.out.gain[2*sb + c] = i <= gain_bound + 1 ? i : 0;
      a.out.k[2*sb + c] = k;
      a.out.flags[2*sb + c] = flag;
      a.out.cos_dist[2*sb + c] = cos_dist;
      a.out.dist[2*sb + c] = dist;
      od_coeff *yo = a.out.y + ((long)c*nblocks + blk)*a.len + off;
      if (flag) {
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

/* Selection + decoder-identical synthesis of the no-reference candidates:
   the comparison `cost <= best_cost` of src/pvq_encoder.c:597-609 with
   cost = dist + lambda*rate (rate supplied by the host-side entropy model, or
   0), then od_gain_expand + od_pvq_synthesis_partial(noref) (src/pvq.c:766,
   :1037-1093) and od_coding_order_to_raster (src/partition.c:176-194) into the
   dequantised plane.  DC is passed through (keyframes quantise it in the Haar
   DC tree, src/encode.c:1377); uncoded positions are zero
   (od_init_skipped_coeffs, src/state.c:1347-1358). */
struct SynthArgs {
  const od_coeff *coef;
  od_coeff *dq;           /* out: dequantised plane(s) */
  const int16_t *qm_inv;
  odhip_pvq_cands in;
  const double *rate;     /* [B][nb][2] bits, or NULL */
  int32_t *qg_out;        /* [B][nb] chosen gain index, or NULL */
  int nplanes;
  int w;
  int h;
  int bs;
  int nb_bands;
  int len;
  int q[ODHIP_MAX_BANDS];
  int beta[ODHIP_MAX_BANDS];
  int off[ODHIP_MAX_BANDS + 1];
  double lambda;
};

__global__ __launch_bounds__(kWave) void k_pvq_select_synth_noref(SynthArgs a) {
  const int band = blockIdx.y;
  const int off = a.off[band];
  const int n = a.off[band + 1] - off;
  const int N = 4 << a.bs;
  const int bw = a.w/N;
  const int bh = a.h/N;
  const long nblocks = (long)a.nplanes*bw*bh;
  const long blk = (long)blockIdx.x*kWave + threadIdx.x;
  if (blk >= nblocks) return;
  const int p = (int)(blk/((long)bw*bh));
  const int rem = (int)(blk - (long)p*bw*bh);
  const int by = rem/bw;
  const int bx = rem - by*bw;
  const long base = (long)p*a.w*a.h + (long)by*N*a.w + bx*N;
  od_coeff *dst = a.dq + base;
  const long sb = blk*a.nb_bands + band;
  double best_cost = a.in.dist0[sb];
  int qg = 0;
  int sel = -1;
  for (int c = 0; c < 2; c++) {
    if (!a.in.flags[2*sb + c]) continue;
    double cost = a.in.dist[2*sb + c];
    if (a.rate) cost = cost + a.lambda*a.rate[2*sb + c];
    if (cost <= best_cost) {
      best_cost = cost;
      qg = a.in.gain[2*sb + c];
      sel = c;
    }
  }
  if (a.qg_out) a.qg_out[sb] = qg;
  if (band == 0) dst[0] = a.coef[base];
  if (qg == 0) {
    for (int j = 0; j < n; j++) dst[kScanXY[off + j][1]*a.w + kScanXY[off + j][0]] = 0;
    return;
  }
  const od_coeff *y = a.in.y + ((long)sel*nblocks + blk)*a.len + off;
  const int32_t g = odq_gain_expand(odq_shl32(qg, ODQ_CGAIN_SHIFT), a.q[band], a.beta[band]);
  int yy = 0;
  for (int j = 0; j < n; j++) yy += y[j]*y[j];
  int gshift = odq_ilog(g) - 14;
  gshift = gshift > 0 ? gshift : 0;
  int32_t scale = 0;
  if (yy != 0) {
    int rsqrt_shift;
    const int16_t rsqrt = odq_rsqrt(yy, &rsqrt_shift);
    scale = odq_vshr_round(rsqrt*(int64_t)g, rsqrt_shift + gshift - 16);
  }
  const int qshift = ODQ_QM_INV_SHIFT - gshift;
  for (int j = 0; j < n; j++) {
    const int32_t x = (int32_t)((int16_t)y[j]*(int64_t)scale >> 16);
    dst[kScanXY[off + j][1]*a.w + kScanXY[off + j][0]] =
     odq_shr_round(x*a.qm_inv[off + j], qshift);
  }
}

}  // namespace

extern "C" int odhip_pvq_search_batch(const int16_t *d_x, int n, const int32_t *d_k,
 od_coeff *d_y, const double *d_g2, double pvq_norm_lambda,
 const int32_t *d_prev_k, double *d_cos, long nbands, odhip_stream stream) {
  if (!d_x || !d_k || !d_y || !d_g2 || !d_cos || n < 1 || n > 128) return ODHIP_EINVAL;
  if (nbands <= 0) return ODHIP_SUCCESS;
  const long grid = (nbands + kWave - 1)/kWave;
  if (grid > 0x7fffffffL) return ODHIP_EINVAL;
  const size_t lds = (size_t)2*n*kWave*sizeof(unsigned short);
  k_pvq_search<<<(unsigned)grid, kWave, lds, (hipStream_t)stream>>>(d_x, n, d_k, d_y,
   d_g2, pvq_norm_lambda, d_prev_k, d_cos, nbands);
  return odhip_check_launch();
}

namespace {

bool g_scan_uploaded = false;

int upload_scan(void) {
  if (!g_scan_uploaded) {
    ODHIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kScanXY), OD_SCAN_XY, sizeof(OD_SCAN_XY)));
    g_scan_uploaded = true;
  }
  return ODHIP_SUCCESS;
}

template <typename A>
int fill_band_geometry(A &a, int nplanes, int w, int h, int bs, const int32_t *q_band,
 const int32_t *beta_band) {
  if (bs < 0 || bs >= ODHIP_NBSIZES || nplanes <= 0 || !q_band || !beta_band) {
    return ODHIP_EINVAL;
  }
  const int n = 4 << bs;
  if (w <= 0 || h <= 0 || w % n || h % n) return ODHIP_EINVAL;
  a.nplanes = nplanes;
  a.w = w;
  a.h = h;
  a.bs = bs;
  a.nb_bands = OD_NBANDS[bs];
  a.len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  for (int i = 0; i <= a.nb_bands; i++) a.off[i] = OD_BAND_OFFS[bs][i];
  for (int i = 0; i < a.nb_bands; i++) {
    if (q_band[i] < 1) return ODHIP_EINVAL;
    a.q[i] = q_band[i];
    a.beta[i] = beta_band[i];
  }
  return ODHIP_SUCCESS;
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

extern "C" int odhip_pvq_noref_bands(const od_coeff *d_coef, int nplanes, int w, int h,
 int bs, const int16_t *d_qm, const int32_t *q_band, const int32_t *beta_band,
 double pvq_norm_lambda, const odhip_pvq_cands *out, odhip_stream stream) {
  if (!d_coef || !d_qm || !out || !out->cg || !out->gain || !out->k || !out->flags
   || !out->cos_dist || !out->dist || !out->dist0 || !out->y) {
    return ODHIP_EINVAL;
  }
  BandsArgs a;
  int rc = fill_band_geometry(a, nplanes, w, h, bs, q_band, beta_band);
  if (rc) return rc;
  rc = upload_scan();
  if (rc) return rc;
  a.coef = d_coef;
  a.qm = d_qm;
  a.out = *out;
  a.lambda = pvq_norm_lambda;
  const int n = 4 << bs;
  const long nblocks = (long)nplanes*(w/n)*(h/n);
  const dim3 grid((unsigned)((nblocks + kWave - 1)/kWave), a.nb_bands);
  /* LDS: the largest band of this block size, 4 bytes per element and lane. */
  int nmax = 0;
  for (int i = 0; i < a.nb_bands; i++) nmax = max(nmax, a.off[i + 1] - a.off[i]);
  const size_t lds = (size_t)nmax*kWave*4;
  k_pvq_bands_noref<<<grid, kWave, lds, (hipStream_t)stream>>>(a);
  return odhip_check_launch();
}

extern "C" int odhip_pvq_select_synth_noref(od_coeff *d_dq, const od_coeff *d_coef,
 int nplanes, int w, int h, int bs, const int16_t *d_qm_inv, const int32_t *q_band,
 const int32_t *beta_band, double pvq_norm_lambda, const odhip_pvq_cands *in,
 const double *d_rate, int32_t *d_qg_out, odhip_stream stream) {
  if (!d_dq || !d_coef || !d_qm_inv || !in) return ODHIP_EINVAL;
  SynthArgs a;
  int rc = fill_band_geometry(a, nplanes, w, h, bs, q_band, beta_band);
  if (rc) return rc;
  rc = upload_scan();
  if (rc) return rc;
  a.coef = d_coef;
  a.dq = d_dq;
  a.qm_inv = d_qm_inv;
  a.in = *in;
  a.rate = d_rate;
  a.qg_out = d_qg_out;
  a.lambda = pvq_norm_lambda;
  const int n = 4 << bs;
  const long nblocks = (long)nplanes*(w/n)*(h/n);
  const dim3 grid((unsigned)((nblocks + kWave - 1)/kWave), a.nb_bands);
  /* 32x32 and 64x64 blocks code only their lowest 512 coefficients
     (src/partition.c:80-83); everything else is zero on a keyframe. */
  if (n*n > a.len) {
    ODHIP_TRY(hipMemsetAsync(d_dq, 0, (size_t)nplanes*w*h*sizeof(od_coeff),
     (hipStream_t)stream));
  }
  k_pvq_select_synth_noref<<<grid, kWave, 0, (hipStream_t)stream>>>(a);
  return odhip_check_launch();
}
