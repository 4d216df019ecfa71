/* pvq_row.cuh - pvq_search_rdo_double (reference src/pvq_encoder.c:93-224) with
   one band per 16-lane DPP ROW, four bands per wavefront: the low-latency form
   for the long bands of the with-reference stage (pvq_refbands.hip), where a
   frame batch holds too few 128-coefficient bands to fill the chip one band per
   lane and each band runs a chain of up to 14 searches.

   A band of up to G*E coefficients is held by a GROUP of G lanes (G = 16, a DPP
   row, E = 8: n = 127/128; G = 4, a quad, E = 8: n = 31/32 - sixteen bands per
   wavefront, per-candidate bookkeeping replicated 4x instead of 16x) in VGPRs in
   BLOCKED index order (lane l of the group owns j = l*E .. l*E+E-1), so lane
   order is index order.  The cross-lane steps are DPP row
   operations (VALU, no LDS), two wave ballots and a few ds_bpermute broadcasts.
   n_true = 16*E - 1 (the reflected vector of a with-reference search) is
   handled by a PAD in the last position: |x| = 0, y = 0, and its candidate is
   given a key that can never win (see the two places marked PAD), so every sum,
   the projection and every argmax are those of the n_true-dimensional search;
   n_true enters the rate term (3./n).

   Sums (xx, xy, yy, L1) are sums of integers below 2^53: exact in any order.

   The greedy argmax (:172-183) compares ROUNDED cross products, `a_j*b_best >
   a_best*b_j` with a = (xy + x_j)^2, b = yy + 2*y_j + 1, scanning j upward from
   0; that relation need not be transitive, so a reduction is not equivalent a
   priori.  It is made exact:
     1. each lane folds its own E candidates left to right in SINGLE precision
        (round 5; rounds 2-4 folded with the reference's double-precision
        comparator); a float key a/b picks the row's proposal W;
     2. every lane checks every one of its candidates against W: it must be an
        exact duplicate of W (same |x| and same y as integers, hence same a and
        b) or lose to W by a relative margin of 2^-18 in single precision (>> the
        2^-52 rounding of the reference's products; the error of the
        single-precision ratios is below 2^-20).  Then the sequential scan
        provably ends on the lowest-indexed duplicate: before it the running
        best is a clear loser, which it beats; after it nothing beats it;
     3. otherwise (a near tie that is not exact, or a key that picked a
        non-maximal proposal) the row replays that pulse with the literal
        left-to-right scan in double precision.
   The last 1 + k/4 pulses (:192-219) maximise one double per candidate with
   `>`: (max value, lowest index) is a total order, so the reduction is exact.

   Requires pvq_search.cuh (od_rsqrt_table, od_dpp.cuh) before it and
   -ffp-contract=off.  Not valid for n_true in {8, 15} (the k == 1 special
   cases of :154-163 are not implemented here; those bands are searched one
   per lane). */
#pragma once
#include <type_traits>
#include "od_sel.cuh"

namespace {

/* Group reductions: G = 16 (a DPP row) or G = 4 (a quad: the first two steps). */
template <int G>
__device__ __forceinline__ int grp_sum(int v) {
  v += row_mov<OD_DPP_XOR1>(v);
  v += row_mov<OD_DPP_XOR2>(v);
  if (G == 16) {
    v += row_mov<OD_DPP_HALF_MIRROR>(v);
    v += row_mov<OD_DPP_MIRROR>(v);
  }
  return v;
}

template <int G>
__device__ __forceinline__ double grp_sum(double v) {
  v += row_mov<OD_DPP_XOR1>(v);
  v += row_mov<OD_DPP_XOR2>(v);
  if (G == 16) {
    v += row_mov<OD_DPP_HALF_MIRROR>(v);
    v += row_mov<OD_DPP_MIRROR>(v);
  }
  return v;
}

template <int G>
__device__ __forceinline__ int grp_min(int v) {
  v = min(v, row_mov<OD_DPP_XOR1>(v));
  v = min(v, row_mov<OD_DPP_XOR2>(v));
  if (G == 16) {
    v = min(v, row_mov<OD_DPP_HALF_MIRROR>(v));
    v = min(v, row_mov<OD_DPP_MIRROR>(v));
  }
  return v;
}

template <int G>
__device__ __forceinline__ int grp_max(int v) {
  v = max(v, row_mov<OD_DPP_XOR1>(v));
  v = max(v, row_mov<OD_DPP_XOR2>(v));
  if (G == 16) {
    v = max(v, row_mov<OD_DPP_HALF_MIRROR>(v));
    v = max(v, row_mov<OD_DPP_MIRROR>(v));
  }
  return v;
}

template <int G>
__device__ __forceinline__ float grp_max(float v) {
  v = fmaxf(v, __int_as_float(row_mov<OD_DPP_XOR1>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(row_mov<OD_DPP_XOR2>(__float_as_int(v))));
  if (G == 16) {
    v = fmaxf(v, __int_as_float(row_mov<OD_DPP_HALF_MIRROR>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(row_mov<OD_DPP_MIRROR>(__float_as_int(v))));
  }
  return v;
}

/* G-bit slice of a wave ballot belonging to this lane's group. */
template <int G>
__device__ __forceinline__ unsigned grp_ballot(bool p, int grp) {
  const unsigned long long m = __ballot(p);
  return (unsigned)(m >> (G*grp)) & ((1u << G) - 1u);
}

/* Value of `v` in lane `src` (0..G-1) of this lane's group. */
template <int G>
__device__ __forceinline__ int grp_bcast(int v, int grp, int src) {
  return __shfl(v, G*grp + src, 64);
}

template <int G>
__device__ __forceinline__ double grp_bcast(double v, int grp, int src) {
  return __shfl(v, G*grp + src, 64);
}

/* The 16-lane forms used by the preparation kernels. */
__device__ __forceinline__ double row_sum(double v) { return grp_sum<16>(v); }
__device__ __forceinline__ int row_max(int v) { return grp_max<16>(v); }

/* xx = sum x^2 of the band (every lane of the row) and 1/sqrt(1e-30 + xx)
   (:107-109, :147): properties of the vector, computed once per chain. */
template <int E, int G>
__device__ __forceinline__ void od_row_norm(const int (&ax)[E], double *xx_out, double *norm_1_out) {
  double xx = 0;
#pragma unroll
  for (int e = 0; e < E; e++) xx += (double)ax[e]*(double)ax[e];
  xx = grp_sum<G>(xx);
  *xx_out = xx;
  *norm_1_out = __ddiv_rn(1., __dsqrt_rn(1e-30 + xx));
}

template <int E, int G>
__device__ __forceinline__ double od_pvq_search_row(const int (&ax)[E], int (&y)[E], int row,
 int l, int n_true, int k, int prev_k, double g2, double pvq_norm_lambda, int force_scan,
 double xx, double norm_1, double *cxy, double *cyy, int *replays = nullptr) {
  constexpr int n = G*E;
  /* the screen's invariants: a lane's fold starts from a real candidate, never the PAD
     (E > 1: the PAD is the LAST of the last lane's E positions), and every operand of a key is
     non-negative - ax[] holds magnitudes, y[] pulse counts, xy = sum ax*y >= 0 - so the relative
     error bound of the single-precision ratios holds */
  static_assert(E > 1, "the single-precision screen folds from candidate 0 of each lane, which must not be the PAD");
  static_assert(G == 4 || G == 16, "a quad or a DPP row");
  const bool pad_lane = n_true != n && l == G - 1;
  const double lambda = __ddiv_rn(pvq_norm_lambda, 1e-30 + g2);
  double xy = 0;
  double yy = 0;
  int i = 0;
  const bool chained = prev_k > 0 && prev_k <= k;
  if (chained) {
    /* the chain's previous search (same vector, prev_k pulses) ended on exactly these band-wide sums
       and left them in *cxy / *cyy: not recomputed (round 4; pvq_regs.cuh) */
  }
  else if (k > 2) {
    double l1 = 0;
#pragma unroll
    for (int e = 0; e < E; e++) l1 += (double)ax[e];
    l1 = grp_sum<G>(l1);
    const double l1_inv = __ddiv_rn(1., l1 > 1e-100 ? l1 : 1e-100);
#pragma unroll
    for (int e = 0; e < E; e++) {
      const double tmp = (k*(double)ax[e])*l1_inv;
      const int v = (int)floor(tmp);
      y[e] = v > 0 ? v : 0;
      xy += (double)ax[e]*y[e];
      yy += (double)(y[e]*y[e]);
      i += y[e];
    }
  }
  else {
#pragma unroll
    for (int e = 0; e < E; e++) y[e] = 0;
  }
  if (chained) {
    xy = *cxy;
    yy = *cyy;
    i = prev_k;
  }
  else {
    xy = grp_sum<G>(xy);
    yy = grp_sum<G>(yy);
    i = grp_sum<G>(i);
  }
  const int rdo_pulses = 1 + k/4;
  const int n_greedy = k - rdo_pulses;
  /* n_true is n or n - 1: compile-time quotients, correctly rounded as the division is */
  const double delta_rate = n_true == n ? 3./n : 3./(n - 1);
  /* Rows of one wavefront run different pulse counts; the ballots inside the
     loops are wave-wide, so iterate to the maximum of the calling rows with
     per-row predication. */
  /* ---- greedy pulses ---------------------------------------------------- */
  /* Round 5: the candidates are SCREENED in single precision and the reference's double-precision
     comparison runs only when the screen cannot vouch for the pulse (header, points 1-3):
       key_j = fl32(a_j)/fl32(b_j) is within 6*2^-24 of the rational a_j/b_j (xy < 2^53 and b < 2^31
       converted once, one add, one multiply: every operand positive);
       the row's proposal W = best key (each lane folds its eight candidates by cross-multiplied
       single-precision products, one v_rcp_f32 per lane makes its key, an integer DPP max picks the row's);
       a candidate with a_j < b_j*thr, thr = key_W*(1 - 2^-17), is a CLEAR LOSER: its rational lies
       below W's by a relative 2^-18, 2^34 times the rounding of the reference's two products;
       every other candidate must be an exact DUPLICATE of W - the same |x_j| and the same y_j, hence the
       same a and b in the reference's arithmetic (compared as integers: equal single-precision values
       would not imply it).
     Then the sequential scan provably ends on the lowest-indexed duplicate (before it the running best
     is a clear loser, which it beats; after it nothing beats it); otherwise the row replays the pulse
     with the literal left-to-right scan in double precision.  Per candidate: 6 single-precision
     operations, 3 integer ones and 7 selects / compares against 12 double-precision operations and 11
     selects before (profiles/r5_search_budget.txt). */
  while (__any(i < n_greedy)) {
    const bool on = i < n_greedy;
    /* yy + 2*y_j + 1 (:175) is an integer below 2^31: formed in integers */
    const int yyp1 = (int)yy + 1;
    const float xyf = (float)xy;
    float af[E];
    float bf[E];
    int bi[E];
    float baf = 0;
    float bbf = 1;
    int bax = 0;
    int bbi = 1;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const float tf = xyf + (float)ax[e];
      af[e] = tf*tf;
      bi[e] = yyp1 + 2*y[e];
      bf[e] = (float)bi[e];
      if (e == E - 1 && pad_lane) af[e] = -1.f;   /* PAD: a clear loser against any proposal */
      if (e == 0 || af[e]*bbf > baf*bf[e]) {
        baf = af[e];
        bbf = bf[e];
        bax = ax[e];
        bbi = bi[e];
      }
    }
    /* proposal: the row's best key, lowest lane on equal keys.  The key is a non-negative finite
       float (a < 2^75; a lane's fold starts from its first candidate, never the PAD), so its bit
       pattern orders like its value and the row maximum is an integer DPP max */
    const int key = __float_as_int(baf*__builtin_amdgcn_rcpf(bbf));
    const int kmax = grp_max<G>(key);
    const unsigned wmask = grp_ballot<G>(key == kmax, row);
    const int wl = wmask ? __ffs(wmask) - 1 : 0;
    const int wx = grp_bcast<G>(bax, row, wl);
    const int wbi = grp_bcast<G>(bbi, row, wl);
    const float thr = __int_as_float(kmax)*(1.f - 7.62939453125e-06f);      /* 1 - 2^-17 */
    bool bad = force_scan != 0 || wmask == 0;
    int first_dup = n;
#pragma unroll
    for (int e = E - 1; e >= 0; e--) {
      bool dup = (ax[e] == wx) & (bi[e] == wbi);
      if (e == E - 1 && pad_lane) dup = false;
      const bool loses = af[e] < bf[e]*thr;     /* false for W and its duplicates */
      first_dup = dup ? l*E + e : first_dup;
      bad |= !(dup | loses);
    }
    int pos;
    int px = wx;                 /* every duplicate of W has W's |x| ... */
    double nb = (double)wbi;     /* ... and the winner's denominator yy + 2*y_pos + 1 IS the next yy */
    if (grp_ballot<G>(bad && on, row) != 0) {
      /* literal scan, src/pvq_encoder.c:172-183, in the reference's double precision */
      if (replays) ++*replays;     /* (odhip_pvq_search_row_batch reports it; a null pointer folds away) */
      double sa = 0;
      double sb = 1;
      int sx = 0;
      pos = 0;
#pragma unroll 1
      for (int j = 0; j < n; j++) {
        const int e = j % E;
        int cx = ax[0];
        int cy = y[0];
#pragma unroll
        for (int t = 1; t < E; t++) {
          if (e == t) {
            cx = ax[t];
            cy = y[t];
          }
        }
        cx = grp_bcast<G>(cx, row, j/E);
        cy = grp_bcast<G>(cy, row, j/E);
        const double tt = xy + od_cvt_u(cx);
        double ca = tt*tt;
        const double cb = (double)(yyp1 + 2*cy);
        if (n_true != n && j == n - 1) ca = -1;     /* PAD: loses every comparison */
        if (j == 0 || ca*sb > sa*cb) {
          sa = ca;
          sb = cb;
          sx = cx;
          pos = j;
        }
      }
      nb = sb;
      px = sx;
    }
    else pos = grp_min<G>(first_dup);
    /* xy += x[pos]; yy += 2*y[pos] + 1 (= the winner's b, an integer below 2^31 held exactly); y[pos]++ */
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (on && l*E + e == pos) y[e]++;
    }
    if (on) {
      xy = xy + (double)px;
      yy = nb;
      i++;
    }
  }
  /* ---- last pulses with the rate term ------------------------------------- */
  /* rate penalty of each of the lane's candidates, (lambda*j)*delta_rate: constant
     over the pulses of a search */
  double pen[E];
#pragma unroll
  for (int e = 0; e < E; e++) pen[e] = (lambda*(l*E + e))*delta_rate;
  /* (2*t)*norm_1 == t*(2*norm_1): scaling by two is exact */
  const double norm2 = 2*norm_1;
  while (__any(i < k)) {
    const bool on = i < k;
    /* od_rsqrt_table(yy + 2*y_j + 1) for every candidate straight from the LDS
       table (the reference's four-entry cache :199-200 holds the same values) */
    const int yyi = (int)yy;
    double bc = 0;
    int bi = 0;
    /* no y_j exceeds the i pulses placed so far: when yy + 2*i + 1 is inside the table in
       every row still searching, the lookups are unconditional (an index of a finished row
       is clamped, its result unused); otherwise each is tested (od_rsqrt_table) */
    auto scan = [&](auto fast) {
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int j = l*E + e;
        double tmp_xy = xy + od_cvt_u(ax[e]);
        const int idx = yyi + 2*y[e] + 1;
        const double r = decltype(fast)::value ? od_rsq_lds[(idx < OD_RSQ_TABLE_N ? idx : OD_RSQ_TABLE_N) - 1]
         : od_rsqrt_table(idx);
        tmp_xy = (tmp_xy*norm2)*r - pen[e];
        if (e == E - 1 && pad_lane) tmp_xy = -1.7976931348623157e308;   /* PAD */
        if (e == 0 || tmp_xy > bc) {
          bc = tmp_xy;
          bi = j;
        }
      }
    };
    if (!__any(on && yyi + 2*i + 1 > OD_RSQ_TABLE_N)) scan(std::true_type());
    else scan(std::false_type());
    /* (max cost, lowest index) over the row.  Round 4: the maximum alone is reduced by DPP (v_max_f64
       of finite doubles is exact), then the LOWEST lane whose own best equals it is found with one
       ballot and its index broadcast - lanes hold consecutive index ranges and every lane's own scan
       keeps its first maximum, so that is the lowest index attaining the maximum: the same total
       order the (cost, index) butterfly of rounds 2-3 implemented with three moves, three compares
       and three selects per step. */
    double bm = bc;
    bm = fmax(bm, row_mov<OD_DPP_XOR1>(bm));
    bm = fmax(bm, row_mov<OD_DPP_XOR2>(bm));
    if (G == 16) {
      bm = fmax(bm, row_mov<OD_DPP_HALF_MIRROR>(bm));
      bm = fmax(bm, row_mov<OD_DPP_MIRROR>(bm));
    }
    const unsigned tmask = grp_ballot<G>(bc == bm, row);
    bi = grp_bcast<G>(bi, row, tmask ? __ffs(tmask) - 1 : 0);
    const int pos = bi;
    int px = 0;
    int py = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (on && l*E + e == pos) {
        px = ax[e];
        py = y[e];
        y[e]++;
      }
    }
    px = grp_bcast<G>(px, row, pos/E);
    py = grp_bcast<G>(py, row, pos/E);
    if (on) {
      xy = xy + (double)px;
      yy = yy + (double)(2*py) + 1;
      i++;
    }
  }
  *cxy = xy;
  *cyy = yy;
  return __ddiv_rn(xy, 1e-100 + __dsqrt_rn(xx*yy));
}

}  // namespace
