/* rate_host.hip - od_pvq_rate at the DEFAULT complexity (speed == 0,
   src/pvq_encoder.c:247-287), as a batched host routine.

   At speed == 0 the reference prices a candidate by RUNNING its codeword coder:
   od_ec_enc_init (two mallocs), a copy of the whole live od_pvq_codeword_ctx (2 120 bytes),
   od_encode_pvq_codeword into the scratch range coder (src/pvq_encoder.c:44-50 ->
   od_encode_band_pvq_splits, src/laplace_encoder.c:38-84), od_ec_enc_tell_frac,
   od_ec_enc_clear (two frees) - 2.0 M times per 1080p frame.  What the price depends on is
   much less than what that moves:
     - the bit count is (shifts of the range so far + raw bits) and the fractional part a
       function of the 16-bit range alone (od_ec_tell_frac, src/entcode.c:65-91): `low`,
       the carry buffer and the output bytes never enter it;
     - a codeword touches a handful of the context's 110 adaptive CDF rows, and only the
       rows it touches (it adapts them as it goes, src/generic_encoder.c:74-85) need a
       private copy.
   odhip_pvq_rate_batch prices all candidates of a band against ONE snapshot of the live
   context: a rate-only range coder (range and bit count, the reference's own update
   od_ec_encode with OD_EC_REDUCED_OVERHEAD, src/entenc.c:173-212) and copy-on-touch CDF
   rows.  No allocation, no context copy, no bytes.  The result is the reference's double,
   bit for bit (tests/test_rate_host.py: against od_pvq_rate on live, adapted contexts).

   Host code only. */
#include <math.h>
#include <string.h>
#include "../../include/daala_hip.h"

namespace {

inline int ilog_nz(unsigned v) {   /* OD_ILOG_NZ */
  return 32 - __builtin_clz(v);
}

inline int ilog(unsigned v) {      /* OD_ILOG */
  return v ? ilog_nz(v) : 0;
}

/* The part of od_ec_enc that od_ec_enc_tell_frac reads. */
struct RateCoder {
  unsigned rng = 0x8000;   /* od_ec_enc_reset, src/entenc.c:136-146 */
  long nbits = 0;          /* bits shifted out of the range + raw bits */
  /* od_ec_encode, src/entenc.c:173-212 (OD_EC_REDUCED_OVERHEAD = 1, src/entcode.h:84) +
     the range part of od_ec_enc_normalize (:63-130) */
  void encode(unsigned fl, unsigned fh, unsigned ft) {
    unsigned r = rng;
    const int s = r - ft >= ft;
    ft <<= s;
    fl <<= s;
    fh <<= s;
    const unsigned d = r - ft;
    const unsigned e = 2*d > ft ? 2*d - ft : 0;                       /* OD_SUBSATU(2*d, ft) */
    const unsigned fle = fl > e ? fl - e : 0;
    const unsigned fhe = fh > e ? fh - e : 0;
    const unsigned u = fl + (fl < e ? fl : e) + ((fle >> 1) < d ? (fle >> 1) : d);
    const unsigned v = fh + (fh < e ? fh : e) + ((fhe >> 1) < d ? (fhe >> 1) : d);
    r = v - u;
    const int sh = 16 - ilog_nz(r);
    nbits += sh;
    rng = r << sh;
  }
  /* od_ec_encode_unscaled, src/entenc.c:254-263 */
  void encode_unscaled(unsigned fl, unsigned fh, unsigned ft) {
    const int s = 15 - ilog_nz(ft - 1);
    encode(fl << s, fh << s, ft << s);
  }
  void bits(int n) {       /* od_ec_enc_bits */
    nbits += n;
  }
  /* od_ec_enc_tell_frac (src/entenc.c:651-668, src/entcode.c:65-91), OD_BITRES = 3 */
  unsigned tell_frac() const {
    unsigned nb = (unsigned)(nbits + 1) << 3;
    unsigned r = rng;
    int l = 0;
    for (int i = 3; i-- > 0;) {
      r = r*r >> 15;
      const int b = (int)(r >> 16);
      l = l << 1 | b;
      r >>= b;
    }
    return nb - l;
  }
};

constexpr int kSplitRows = 14*7;
constexpr int kK1Rows = 12;

/* The context of one candidate: the snapshot, with private copies of the rows it touches. */
struct CowCtx {
  const odhip_pvq_codeword_ctx *snap;
  uint16_t split[kSplitRows][8];
  uint16_t k1[kK1Rows][16];
  unsigned split_epoch[kSplitRows];
  unsigned k1_epoch[kK1Rows];
  unsigned epoch;
  uint16_t *split_row(int r) {
    if (split_epoch[r] != epoch) {
      memcpy(split[r], snap->pvq_split_cdf[r], sizeof(split[r]));
      split_epoch[r] = epoch;
    }
    return split[r];
  }
  uint16_t *k1_row(int r) {
    if (k1_epoch[r] != epoch) {
      memcpy(k1[r], snap->pvq_k1_cdf[r], sizeof(k1[r]));
      k1_epoch[r] = epoch;
    }
    return k1[r];
  }
};

/* od_encode_cdf_adapt, src/generic_encoder.c:74-85 */
inline void encode_cdf_adapt(RateCoder &ec, int val, uint16_t *cdf, int n, int increment) {
  ec.encode_unscaled(val > 0 ? cdf[val - 1] : 0, cdf[val], cdf[n - 1]);
  if (cdf[n - 1] + increment > 32767) {
    for (int i = 0; i < n; i++) cdf[i] = (uint16_t)((cdf[i] >> 1) + i + 1);
  }
  for (int i = val; i < n; i++) cdf[i] = (uint16_t)(cdf[i] + increment);
}

inline int pvq_size_ctx(int n) {   /* od_pvq_size_ctx, src/pvq.c:389-395 */
  return 2*ilog((unsigned)(n - 1)) - 1 - (n & 1) - 7*(n == 14);
}

inline int pvq_k1_ctx(int n, int orig_length) {   /* od_pvq_k1_ctx, src/pvq.c:402-405 */
  return orig_length ? 8 + 2*(n > 8) + (n & 1) : pvq_size_ctx(n);
}

/* od_encode_pvq_split, src/laplace_encoder.c:38-54 */
void encode_pvq_split(RateCoder &ec, CowCtx &cx, int count, int sum, int ctx) {
  if (sum == 0) return;
  const int shift = ilog((unsigned)sum) - 3 > 0 ? ilog((unsigned)sum) - 3 : 0;
  if (shift) {
    count >>= shift;
    sum >>= shift;
  }
  const int fctx = 7*ctx + sum - 1;
  encode_cdf_adapt(ec, count, cx.split_row(fctx), sum + 1, cx.snap->pvq_split_increment);
  if (shift) ec.bits(shift);
}

/* od_encode_band_pvq_splits, src/laplace_encoder.c:56-80 */
template <class T>
void encode_band_pvq_splits(RateCoder &ec, CowCtx &cx, const T *y, int n, int k, int level) {
  if (n <= 1 || k == 0) return;
  if (k == 1 && n <= 16) {
    const int cdf_id = pvq_k1_ctx(n, level == 0);
    int pos = 0;
    while (!y[pos]) pos++;
    encode_cdf_adapt(ec, pos, cx.k1_row(cdf_id), n, cx.snap->pvq_k1_increment);
    return;
  }
  const int mid = n >> 1;
  int count_right = k;
  for (int i = 0; i < mid; i++) count_right -= y[i] < 0 ? -y[i] : y[i];
  encode_pvq_split(ec, cx, count_right, k, pvq_size_ctx(n));
  encode_band_pvq_splits(ec, cx, y, mid, k - count_right, level + 1);
  encode_band_pvq_splits(ec, cx, y + mid, n - mid, count_right, level + 1);
}

/* The codeword part of od_pvq_rate (src/pvq_encoder.c:265-275): bits of
   od_encode_pvq_codeword(y, n, k) on a fresh coder against the snapshot. */
template <class T>
double codeword_rate(CowCtx &cx, const T *y, int n, int k) {
  RateCoder ec;
  const unsigned tell = ec.tell_frac();
  cx.epoch++;
  encode_band_pvq_splits(ec, cx, y, n, k, 0);
  int nz = 0;
  for (int i = 0; i < n; i++) nz += y[i] != 0;
  ec.bits(nz);            /* one sign bit per non-zero pulse, :48-49 */
  return (ec.tell_frac() - tell)/8.;
}

thread_local CowCtx t_cx;   /* the epochs make a reused object as good as a fresh one */

template <class T>
int rate_batch(double *rate, const odhip_pvq_codeword_ctx *ctx, int ncand, const T *const *y, const int *k,
 const int *qg, const int *theta, const int *ts, int n, int icgr, int is_keyframe, int pli) {
  if (!rate || !ctx || ncand < 0 || (ncand && (!y || !k || !qg || !theta || !ts)) || n < 1 || n > 1024) {
    return ODHIP_EINVAL;
  }
  CowCtx &cx = t_cx;
  if (cx.epoch > 0xfffffff0u) {
    memset(cx.split_epoch, 0, sizeof(cx.split_epoch));
    memset(cx.k1_epoch, 0, sizeof(cx.k1_epoch));
    cx.epoch = 0;
  }
  cx.snap = ctx;
  for (int c = 0; c < ncand; c++) {
    double r;
    if (k[c] == 0) r = 0;
    else {
      if (!y[c]) return ODHIP_EINVAL;
      /* n - (theta != -1) coded positions, :272 */
      r = codeword_rate(cx, y[c], n - (theta[c] != -1), k[c]);
    }
    if (qg[c] > 0 && theta[c] >= 0) {
      /* :276-285; OD_LOG2(x) = M_LOG2E*log(x), src/odintrin.h */
      r += .9*(1.4426950408889634073599246810019*log((double)ts[c]));
      if (is_keyframe && pli == 0) r += 6;
      if (qg[c] == icgr) r -= .5;
    }
    rate[c] = r;
  }
  return ODHIP_SUCCESS;
}

}  // namespace

extern "C" int odhip_pvq_rate_batch(double *rate, const odhip_pvq_codeword_ctx *ctx, int ncand,
 const od_coeff *const *y, const int *k, const int *qg, const int *theta, const int *ts, int n, int icgr,
 int is_keyframe, int pli) {
  return rate_batch(rate, ctx, ncand, y, k, qg, theta, ts, n, icgr, is_keyframe, pli);
}

extern "C" int odhip_pvq_rate_batch16(double *rate, const odhip_pvq_codeword_ctx *ctx, int ncand,
 const int16_t *const *y, const int *k, const int *qg, const int *theta, const int *ts, int n, int icgr,
 int is_keyframe, int pli) {
  return rate_batch(rate, ctx, ncand, y, k, qg, theta, ts, n, icgr, is_keyframe, pli);
}
