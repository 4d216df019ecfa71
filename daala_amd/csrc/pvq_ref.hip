/* pvq_ref.hip - the data-parallel pieces of pvq_theta's WITH-REFERENCE path
   (reference src/pvq_encoder.c:381-565: inter frames and chroma-from-luma),
   one band per lane, on plain band vectors [band][n] in coding order:

     odhip_pvq_ref_prepare      QM scaling of x and r, gains, correlation,
                                Householder reflection (src/pvq.c:498-623) and
                                the n-1 reflected coefficients the K-pulse search
                                runs on - everything before the one libm call of
                                the path, acos(corr) (:478)
     odhip_pvq_ref_candidates   given theta = floor(.5 + OD_THETA_SCALE*acos(corr))
                                from the host: the (gain, theta) candidates with
                                their K, in the reference's sorted order
                                (:466-504)
     odhip_pvq_synthesis        od_pvq_synthesis_partial (src/pvq.c:1037-1115),
                                with or without reference: decoder-identical
                                dequantisation of a chosen candidate

   The searches themselves are odhip_pvq_search_batch on the reflected vectors
   (any n; prev_k continuation), the rate of a candidate needs the adaptive
   entropy coder and stays on the host (od_pvq_rate).  acos is not
   evaluated on the device: glibc's result is not reproducible there bit for bit. */
#include "../../include/daala_hip.h"
#include "od_common.cuh"
#include "od_pvq_math.cuh"

namespace {

constexpr int kMaxN = 128;   /* OD_MAX_PVQ_SIZE */

__global__ __launch_bounds__(64) void k_ref_prepare(const od_coeff *x0, const od_coeff *r0, int n,
 long nbands, const int16_t *qm, int q0, int beta, int cfl_enabled, int16_t *x16o, int16_t *r16o,
 int16_t *xro, odhip_pvq_refprep *out) {
  const long b = (long)blockIdx.x*64 + threadIdx.x;
  if (b >= nbands) return;
  const od_coeff *x = x0 + b*n;
  const od_coeff *r = r0 + b*n;
  int16_t *x16 = x16o + b*n;
  int16_t *r16 = r16o + b*n;
  /* od_vector_log_mag, src/pvq.c:472-484; src/pvq_encoder.c:381-385 */
  int sx = 0;
  int sr = 0;
  for (int i = 0; i < n; i++) {
    const int tx = (int16_t)(x[i] >> 8);
    const int tr = (int16_t)(r[i] >> 8);
    sx += tx*tx;
    sr += tr*tr;
  }
  int xshift = 8 + 1 + odq_ilog(n + sx)/2 - 15;
  xshift = xshift > 0 ? xshift : 0;
  int rshift = 8 + 1 + odq_ilog(n + sr)/2 - 14;
  rshift = rshift > 0 ? rshift : 0;
  double corr = 0;
  int r_null = 1;
  int accx = 0;
  int accr = 0;
  for (int i = 0; i < n; i++) {
    const int16_t xv = (int16_t)odq_shr_round(x[i]*qm[i], ODQ_QM_SHIFT + xshift);
    const int16_t rv = (int16_t)odq_shr_round(r[i]*qm[i], ODQ_QM_SHIFT + rshift);
    x16[i] = xv;
    r16[i] = rv;
    corr += odq_mult16_16(xv, rv);
    if (r[i]) r_null = 0;
    accx += xv*(int)xv;
    accr += rv*(int)rv;
  }
  int32_t g;
  int32_t gr;
  const int32_t cg = odq_gain_from_acc(accx, q0, beta, xshift, &g);
  int32_t cgr = odq_gain_from_acc(accr, q0, beta, rshift, &gr);
  if (cfl_enabled) cgr = 256;
  const int icgr = odq_shr_round(cgr, ODQ_CGAIN_SHIFT);
  const int32_t gain_offset = cgr - odq_shl32(icgr, ODQ_CGAIN_SHIFT);
  /* src/pvq_encoder.c:436-438 */
  corr = __ddiv_rn(corr, 1e-100 + __ddiv_rn(g*(double)gr, (double)odq_shl32(1, xshift + rshift)));
  corr = corr < 1. ? corr : 1.;
  corr = corr > -1. ? corr : -1.;
  int m = 0;
  int s = 1;
  int16_t *xr = xro + b*(n - 1);
  if (n <= kMaxN && !r_null && corr > 0) {
    /* od_compute_householder, src/pvq.c:498-521: first largest |r_i| wins */
    int maxr = 0;
    for (int i = 0; i < n; i++) {
      const int a = abs((int)r16[i]);
      if (a > maxr) {
        maxr = (int16_t)a;
        m = i;
      }
    }
    s = r16[m] > 0 ? 1 : -1;
    r16[m] = (int16_t)(r16[m] + odq_shr_round(gr*s, rshift));
    /* od_apply_householder, src/pvq.c:560-623 */
    int32_t l2r = 0;
    int32_t proj = 0;
    for (int i = 0; i < n; i++) {
      l2r += odq_mult16_16(r16[i], r16[i]);
      proj += odq_mult16_16(r16[i], x16[i]);
    }
    const int l2r_shift = (odq_ilog(l2r) - 1) - 14;
    const int16_t l2r_norm = (int16_t)odq_vshr_round(l2r, l2r_shift);
    const int16_t rcp = odq_rcp(l2r_norm);
    const int proj_shift = (odq_ilog(abs(proj)) - 1) - 14;
    const int16_t proj_norm = (int16_t)odq_vshr_round(proj, proj_shift);
    const int16_t proj_1 = (int16_t)odq_mult16_16_q15(proj_norm, rcp);
    int outshift = 14 - proj_shift - 1 + l2r_shift;
    if (outshift > 30) outshift = 30;
    /* the reflected vector without element m (src/pvq_encoder.c:481) */
    for (int i = 0; i < n; i++) {
      int32_t tmp = odq_mult16_16(r16[i], proj_1);
      tmp = outshift >= 0 ? odq_shr_round(tmp, outshift) : odq_shl32(tmp, -outshift);
      const int16_t v = (int16_t)(x16[i] - tmp);
      if (i < m) xr[i] = v;
      else if (i > m) xr[i - 1] = v;
    }
  }
  else {
    for (int i = 0; i < n - 1; i++) xr[i] = 0;
  }
  odhip_pvq_refprep o;
  o.xshift = xshift;
  o.rshift = rshift;
  o.g = g;
  o.gr = gr;
  o.cg = cg;
  o.cgr = cgr;
  o.icgr = icgr;
  o.gain_offset = gain_offset;
  o.m = m;
  o.s = s;
  o.r_null = r_null;
  o.reserved = 0;
  o.corr = corr;
  o.reserved2 = 0;
  out[b] = o;
}

__global__ __launch_bounds__(64) void k_ref_candidates(const odhip_pvq_refprep *prep,
 const int32_t *theta_in, int n, long nbands, int beta, odhip_pvq_refcand *items_out,
 int32_t *nitems_out) {
  const long b = (long)blockIdx.x*64 + threadIdx.x;
  if (b >= nbands) return;
  const odhip_pvq_refprep p = prep[b];
  odhip_pvq_refcand *items = items_out + b*ODHIP_PVQ_MAX_REFCANDS;
  int nitems = 0;
  if (n <= kMaxN && !p.r_null && p.corr > 0) {
    const int32_t theta = theta_in[b];
    const int gain_bound = (p.cg - p.gain_offset) >> ODQ_CGAIN_SHIFT;
    const double pi = 3.14159265358979323846;   /* M_PI */
    const double scale = 32768*2./pi;           /* OD_THETA_SCALE, src/pvq.h:78 */
    const double scale_1 = __ddiv_rn(1., scale);
    for (int i = gain_bound - 1 > 1 ? gain_bound - 1 : 1; i <= gain_bound + 1; i++) {
      const int32_t qcg = odq_shl32(i, ODQ_CGAIN_SHIFT) + p.gain_offset;
      const int ts = odq_pvq_compute_max_theta(qcg, beta);
      /* same left-to-right products as src/pvq_encoder.c:482-484 */
      const double t = __ddiv_rn(((theta*scale_1)*2), pi)*ts;
      int lower = (int)floor(.5 + t) - 2;
      if (lower < 0) lower = 0;
      int upper = (int)ceil(t);
      if (upper > ts - 1) upper = ts - 1;
      for (int j = lower; j <= upper && nitems < ODHIP_PVQ_MAX_REFCANDS; j++) {
        odhip_pvq_refcand c;
        c.gain = i;
        c.theta = j;
        c.ts = ts;
        c.k = odq_compute_k_ref(j, n);
        c.qcg = qcg;
        c.qtheta = odq_pvq_compute_theta(j, ts);
        /* stable insertion by (k, gain): items_compare, src/pvq_encoder.c:301-305
           (glibc's qsort is a stable merge sort at this size) */
        int pos = nitems;
        while (pos > 0) {
          const odhip_pvq_refcand q = items[pos - 1];
          const int cmp = q.k == c.k ? q.gain - c.gain : q.k - c.k;
          if (cmp <= 0) break;
          items[pos] = q;
          pos--;
        }
        items[pos] = c;
        nitems++;
      }
    }
  }
  nitems_out[b] = nitems;
}

/* params per band: {noref, g, theta, m, s}. */
__global__ __launch_bounds__(64) void k_synthesis(od_coeff *out, const od_coeff *y,
 const int16_t *r16a, int n, long nbands, const int32_t *params, const int16_t *qm_inv) {
  const long b = (long)blockIdx.x*64 + threadIdx.x;
  if (b >= nbands) return;
  const int noref = params[5*b];
  const int32_t g = params[5*b + 1];
  const int32_t theta = params[5*b + 2];
  const int m = params[5*b + 3];
  const int s = params[5*b + 4];
  const od_coeff *yp = y + b*n;
  const int16_t *r16 = r16a + b*n;
  od_coeff *xo = out + b*n;
  const int nn = n - (!noref);
  int yy = 0;
  for (int i = 0; i < nn; i++) yy += yp[i]*(int32_t)yp[i];
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
    for (int i = 0; i < n; i++) {
      const int32_t x = (int32_t)odq_mult16_32_q16(yp[i], scale);
      xo[i] = odq_shr_round(x*qm_inv[i], qshift);
    }
    return;
  }
  /* src/pvq.c:1094-1114: the two double products by 2^-15 are exact */
  scale = (int32_t)floor(.5 + (scale*(1./32768))*odq_pvq_sin(theta));
  const int16_t xm = (int16_t)floor(.5 + ((-s*odq_shr_round(g, gshift))*(1./32768))*odq_pvq_cos(theta));
  /* x = [y*scale with xm inserted at m]; Householder back; inverse QM.  Two
     passes over the band recompute x[i] instead of holding 128 values. */
  int32_t l2r = 0;
  int32_t proj = 0;
  for (int i = 0; i < n; i++) {
    const int16_t xi = i == m ? xm : (int16_t)odq_mult16_32_q16(yp[i < m ? i : i - 1], scale);
    l2r += odq_mult16_16(r16[i], r16[i]);
    proj += odq_mult16_16(r16[i], xi);
  }
  const int l2r_shift = (odq_ilog(l2r) - 1) - 14;
  const int16_t l2r_norm = (int16_t)odq_vshr_round(l2r, l2r_shift);
  const int16_t rcp = odq_rcp(l2r_norm);
  const int proj_shift = (odq_ilog(abs(proj)) - 1) - 14;
  const int16_t proj_norm = (int16_t)odq_vshr_round(proj, proj_shift);
  const int16_t proj_1 = (int16_t)odq_mult16_16_q15(proj_norm, rcp);
  int outshift = 14 - proj_shift - 1 + l2r_shift;
  if (outshift > 30) outshift = 30;
  for (int i = 0; i < n; i++) {
    const int16_t xi = i == m ? xm : (int16_t)odq_mult16_32_q16(yp[i < m ? i : i - 1], scale);
    int32_t tmp = odq_mult16_16(r16[i], proj_1);
    tmp = outshift >= 0 ? odq_shr_round(tmp, outshift) : odq_shl32(tmp, -outshift);
    const int16_t v = (int16_t)(xi - tmp);
    xo[i] = odq_shr_round(v*qm_inv[i], qshift);
  }
}


/* neg_deinterleave, src/pvq_decoder.c:53-59. */
__device__ __forceinline__ int odq_neg_deinterleave(int x, int ref) {
  if (x < 2*ref - 1) return x & 1 ? ref - 1 - (x >> 1) : ref + (x >> 1);
  return x + 1;
}

/* od_pvq_compute_k, nodesync (src/pvq.c:902-953): what the decoder derives the pulse count
   from under OD_ROBUST_STREAM - nothing that depends on the reference. */
__device__ __forceinline__ int odq_compute_k_dec(int32_t qcg, int itheta, int noref, int n, int beta) {
  return noref ? odq_compute_k_noref(qcg, n, beta) : odq_compute_k_ref(itheta, n);
}

/* pvq_decode_partition (src/pvq_decoder.c:122-298) after its entropy-decoder reads, one band
   per lane: sym[b] = {the gain symbol as read (before deinterleaving), itheta, noref, -}.
   The QM-scaled reference is recomputed where it is used (a lane cannot hold 128 values):
   ref16[i] = SHR_ROUND(ref[i]*qm[i], OD_QM_SHIFT + rshift), and after od_compute_householder
   element m carries + SHR_ROUND(gr*s, rshift) (src/pvq.c:498-521). */
/* The band's arithmetic, one band per lane, through accessors: REF(i) / Y(i) read the band's
   reference and pulses, OUT(i, v) writes coefficient i (OUT(i, .) may overwrite REF(i): every loop
   reads position i before it writes it). */
template <class RefAt, class YAt, class OutAt>
__device__ __forceinline__ void pvq_decode_band(RefAt REF, YAt Y, OutAt OUT, int n, int4 sy, const int16_t *qm,
 const int16_t *qm_inv, int q0, int beta, int is_keyframe, int pli, int2 *info_b) {
  int qg = sy.x;
  int itheta = sy.y;
  const int noref = sy.z;
  /* :213-224 */
  int sr = 0;
  for (int i = 0; i < n; i++) {
    const int tr = (int16_t)(REF(i) >> 8);
    sr += tr*tr;
  }
  int rshift = 8 + 1 + odq_ilog(n + sr)/2 - 14;
  rshift = rshift > 0 ? rshift : 0;
#define OD_REF16(i) ((int16_t)odq_shr_round((int32_t)((uint32_t)REF(i)*(uint32_t)(int32_t)qm[i]), ODQ_QM_SHIFT + rshift))
  int32_t theta = 0;
  int32_t gr = 0;
  int32_t qcg;
  int skip = 0;
  int m = 0;
  int sgn = 0;
  int rm = 0;          /* ref16[m] after od_compute_householder */
  if (!noref) {
    /* :225-255 */
    int accr = 0;
    int maxr = 0;
    for (int i = 0; i < n; i++) {
      const int rv = OD_REF16(i);
      accr += rv*rv;
      const int a = abs(rv);
      if (a > maxr) {
        maxr = a;
        m = i;
        rm = rv;
      }
    }
    if (maxr == 0) rm = OD_REF16(0);
    int32_t cgr = odq_gain_from_acc(accr, q0, beta, rshift, &gr);
    const int cfl_enabled = pli != 0 && is_keyframe;
    if (cfl_enabled) cgr = 256;
    const int icgr = odq_shr_round(cgr, ODQ_CGAIN_SHIFT);
    if (is_keyframe) qg = odq_neg_deinterleave(qg, icgr);
    else {
      qg = odq_neg_deinterleave(qg, icgr + 1) - 1;
      if (qg == 0) skip = icgr ? 1 : 2;
    }
    if (qg == icgr && itheta == 0 && !cfl_enabled) skip = 2;
    const int32_t gain_offset = cgr - odq_shl32(icgr, ODQ_CGAIN_SHIFT);
    qcg = odq_shl32(qg, ODQ_CGAIN_SHIFT) + gain_offset;
    theta = odq_pvq_compute_theta(itheta, odq_pvq_compute_max_theta(qcg, beta));
    /* od_compute_householder, src/pvq.c:498-521 */
    sgn = rm > 0 ? 1 : -1;
    rm = (int16_t)(rm + odq_shr_round(gr*sgn, rshift));
  }
  else {
    itheta = 0;
    if (!is_keyframe) qg++;
    qcg = odq_shl32(qg, ODQ_CGAIN_SHIFT);
    if (qg == 0) skip = 1;
  }
  if (info_b) *info_b = make_int2(odq_compute_k_dec(qcg, itheta, noref, n, beta), skip);
  if (skip) {
    for (int i = 0; i < n; i++) OUT(i, skip == 2 ? REF(i) : 0);
    return;
  }
  /* od_gain_expand + od_pvq_synthesis_partial, src/pvq.c:766-811, :1037-1115 */
  const int32_t g = odq_gain_expand(qcg, q0, beta);
  const int nn = n - (!noref);
  int yy = 0;
  for (int i = 0; i < nn; i++) yy += Y(i)*(int32_t)Y(i);
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
    for (int i = 0; i < n; i++) {
      const int32_t x = (int32_t)odq_mult16_32_q16(Y(i), scale);
      OUT(i, odq_shr_round(x*qm_inv[i], qshift));
    }
    return;
  }
  scale = (int32_t)floor(.5 + (scale*(1./32768))*odq_pvq_sin(theta));
  const int16_t xm = (int16_t)floor(.5 + ((-sgn*odq_shr_round(g, gshift))*(1./32768))*odq_pvq_cos(theta));
  int32_t l2r = 0;
  int32_t proj = 0;
  for (int i = 0; i < n; i++) {
    const int ri = i == m ? rm : OD_REF16(i);
    const int16_t xi = i == m ? xm : (int16_t)odq_mult16_32_q16(Y(i < m ? i : i - 1), scale);
    l2r += odq_mult16_16(ri, ri);
    proj += odq_mult16_16(ri, xi);
  }
  const int l2r_shift = (odq_ilog(l2r) - 1) - 14;
  const int16_t l2r_norm = (int16_t)odq_vshr_round(l2r, l2r_shift);
  const int16_t rcp = odq_rcp(l2r_norm);
  const int proj_shift = (odq_ilog(abs(proj)) - 1) - 14;
  const int16_t proj_norm = (int16_t)odq_vshr_round(proj, proj_shift);
  const int16_t proj_1 = (int16_t)odq_mult16_16_q15(proj_norm, rcp);
  int outshift = 14 - proj_shift - 1 + l2r_shift;
  if (outshift > 30) outshift = 30;
  for (int i = 0; i < n; i++) {
    const int ri = i == m ? rm : OD_REF16(i);
    const int16_t xi = i == m ? xm : (int16_t)odq_mult16_32_q16(Y(i < m ? i : i - 1), scale);
    int32_t tmp = odq_mult16_16(ri, proj_1);
    tmp = outshift >= 0 ? odq_shr_round(tmp, outshift) : odq_shl32(tmp, -outshift);
    const int16_t v = (int16_t)(xi - tmp);
    OUT(i, odq_shr_round(v*qm_inv[i], qshift));
  }
#undef OD_REF16
}

#ifdef ODHIP_EXPERIMENTS
/* One band per lane straight from memory (stride-n accesses): the first form, kept in the
   experiments build as the cross-check of k_pvq_decode (ODHIP_DECODE_LANE=1 selects it). */
__global__ __launch_bounds__(64) void k_pvq_decode_lane(od_coeff *out, const od_coeff *refa, const od_coeff *ya,
 int n, long nbands, const int4 *sym, const int16_t *qm, const int16_t *qm_inv, int q0, int beta,
 int is_keyframe, int pli, int2 *info) {
  const long b = (long)blockIdx.x*64 + threadIdx.x;
  if (b >= nbands) return;
  const od_coeff *ref = refa + b*n;
  const od_coeff *yp = ya + b*n;
  od_coeff *xo = out + b*n;
  pvq_decode_band([&](int i) { return ref[i]; }, [&](int i) { return yp[i]; },
   [&](int i, od_coeff v) { xo[i] = v; }, n, sym[b], qm, qm_inv, q0, beta, is_keyframe, pli, info ? info + b : nullptr);
}
#endif

/* The same with the bands of a workgroup staged through LDS: B bands (64, or 32 above 64
   coefficients: 2*B*(n|1) words of LDS) are contiguous in memory, so the 64 lanes read B*n
   references and B*n pulses with unit stride, each lane then works on its band in its LDS row
   (pitch n|1: odd, the rows of the 64 lanes start in different banks) writing the coefficients over
   the references, and the rows leave with unit stride again.  Round 3's form read and wrote HBM
   with a stride of n words per lane in three passes. */
__global__ __launch_bounds__(64) void k_pvq_decode(od_coeff *out, const od_coeff *refa, const od_coeff *ya,
 int n, int B, long nbands, const int4 *sym, const int16_t *qm, const int16_t *qm_inv, int q0, int beta,
 int is_keyframe, int pli, int2 *info) {
  extern __shared__ od_coeff s_dec[];
  const int S = n | 1;
  od_coeff *sref = s_dec;
  od_coeff *sy = s_dec + B*S;
  const int lane = threadIdx.x;
  const long b0 = (long)blockIdx.x*B;
  const long left = nbands - b0;
  const int nb = left < B ? (int)left : B;
  const int total = nb*n;
  const od_coeff *gref = refa + b0*n;
  const od_coeff *gy = ya + b0*n;
  {
    int band = lane/n;
    int i = lane - band*n;
    const int sb = 64/n;
    const int si = 64 - sb*n;
    for (int idx = lane; idx < total; idx += 64) {
      sref[band*S + i] = gref[idx];
      sy[band*S + i] = gy[idx];
      band += sb;
      i += si;
      if (i >= n) {
        i -= n;
        band++;
      }
    }
  }
  __syncthreads();
  if (lane < nb) {
    od_coeff *row = sref + lane*S;
    const od_coeff *yrow = sy + lane*S;
    pvq_decode_band([&](int i) { return row[i]; }, [&](int i) { return yrow[i]; },
     [&](int i, od_coeff v) { row[i] = v; }, n, sym[b0 + lane], qm, qm_inv, q0, beta, is_keyframe, pli,
     info ? info + b0 + lane : nullptr);
  }
  __syncthreads();
  {
    od_coeff *gout = out + b0*n;
    int band = lane/n;
    int i = lane - band*n;
    const int sb = 64/n;
    const int si = 64 - sb*n;
    for (int idx = lane; idx < total; idx += 64) {
      gout[idx] = sref[band*S + i];
      band += sb;
      i += si;
      if (i >= n) {
        i -= n;
        band++;
      }
    }
  }
}

}  // namespace

extern "C" int odhip_pvq_decode_bands(od_coeff *d_out, const od_coeff *d_ref, const od_coeff *d_y, int n,
 long nbands, const int32_t *d_sym, const int16_t *d_qm, const int16_t *d_qm_inv, int q0, int beta,
 int is_keyframe, int pli, int32_t *d_info, odhip_stream stream) {
  if (nbands == 0) return ODHIP_SUCCESS;
  if (!d_out || !d_ref || !d_y || !d_sym || !d_qm || !d_qm_inv || n < 2 || n > kMaxN || nbands < 0 || q0 < 1
   || ((uintptr_t)d_sym & 15) || ((uintptr_t)d_info & 7)) {
    return ODHIP_EINVAL;
  }
#ifdef ODHIP_EXPERIMENTS
  /* the one-band-per-lane form of rounds 1-3 (the cross-check of the LDS-staged kernel) */
  static const bool lane_form = getenv("ODHIP_DECODE_LANE") != nullptr;
  if (lane_form) {
    k_pvq_decode_lane<<<(unsigned)((nbands + 63)/64), 64, 0, (hipStream_t)stream>>>(d_out, d_ref, d_y, n, nbands,
     reinterpret_cast<const int4 *>(d_sym), d_qm, d_qm_inv, q0, beta, is_keyframe != 0, pli,
     reinterpret_cast<int2 *>(d_info));
    return odhip_check_launch();
  }
#endif
  const int B = n > 64 ? 32 : 64;
  const size_t lds = (size_t)2*B*(n | 1)*sizeof(od_coeff);
  k_pvq_decode<<<(unsigned)((nbands + B - 1)/B), 64, lds, (hipStream_t)stream>>>(d_out, d_ref, d_y, n, B, nbands,
   reinterpret_cast<const int4 *>(d_sym), d_qm, d_qm_inv, q0, beta, is_keyframe != 0, pli,
   reinterpret_cast<int2 *>(d_info));
  return odhip_check_launch();
}

extern "C" int odhip_pvq_ref_prepare(const od_coeff *d_x0, const od_coeff *d_r0, int n, long nbands,
 const int16_t *d_qm, int q0, int beta, int cfl_enabled, int16_t *d_x16, int16_t *d_r16,
 int16_t *d_xr, odhip_pvq_refprep *d_out, odhip_stream stream) {
  if (nbands == 0) return ODHIP_SUCCESS;
  if (!d_x0 || !d_r0 || !d_qm || !d_x16 || !d_r16 || !d_xr || !d_out || n < 2 || n > kMaxN
   || nbands < 0 || q0 < 1) {
    return ODHIP_EINVAL;
  }
  k_ref_prepare<<<(unsigned)((nbands + 63)/64), 64, 0, (hipStream_t)stream>>>(d_x0, d_r0, n, nbands,
   d_qm, q0, beta, cfl_enabled, d_x16, d_r16, d_xr, d_out);
  return odhip_check_launch();
}

extern "C" int odhip_pvq_ref_candidates(const odhip_pvq_refprep *d_prep, const int32_t *d_theta,
 int n, long nbands, int beta, odhip_pvq_refcand *d_items, int32_t *d_nitems,
 odhip_stream stream) {
  if (nbands == 0) return ODHIP_SUCCESS;
  if (!d_prep || !d_theta || !d_items || !d_nitems || n < 2 || n > kMaxN || nbands < 0) {
    return ODHIP_EINVAL;
  }
  k_ref_candidates<<<(unsigned)((nbands + 63)/64), 64, 0, (hipStream_t)stream>>>(d_prep, d_theta, n,
   nbands, beta, d_items, d_nitems);
  return odhip_check_launch();
}

extern "C" int odhip_pvq_synthesis(od_coeff *d_out, const od_coeff *d_y, const int16_t *d_r16, int n,
 long nbands, const int32_t *d_params, const int16_t *d_qm_inv, odhip_stream stream) {
  if (nbands == 0) return ODHIP_SUCCESS;
  if (!d_out || !d_y || !d_r16 || !d_params || !d_qm_inv || n < 2 || n > kMaxN || nbands < 0) {
    return ODHIP_EINVAL;
  }
  k_synthesis<<<(unsigned)((nbands + 63)/64), 64, 0, (hipStream_t)stream>>>(d_out, d_y, d_r16, n,
   nbands, d_params, d_qm_inv);
  return odhip_check_launch();
}
