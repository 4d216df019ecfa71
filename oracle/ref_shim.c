/* oracle/ref_shim.c - TEST INFRASTRUCTURE, not product code.

   Thin exported wrappers around the *real* reference implementation, compiled
   from the reference sources where they lie (REF=/root/reference) into
   oracle/_ref/libdaalaref.so by oracle/Makefile.  It exists so that the CPU
   restatement in oracle/od_oracle.c (and, through it, the HIP kernels) can be
   pinned against the reference itself, and so that bench.py can time the
   reference C path as cpu_baseline.kind == "reference".

   File-local (static) reference functions are reached by textual inclusion of
   the reference translation units -- the pattern the reference's own test uses
   (src/tests/test_coef_coder.c:25-34).  `static` is defined away for those two
   units so that pvq_search_rdo_double & co. become ordinary (interposable)
   symbols of the library; no reference source is copied into this repo.

   Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
   the resulting library. */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define static
#include "pvq.c"
#include "pvq_encoder.c"
#undef static

#include "dct.h"
#include "filter.h"
#include "partition.h"
#include "state.h"
#include "tf.h"

#define REF_EXPORT __attribute__((visibility("default")))

/* ---- the 2-D transform tables the stage wrappers below dispatch through -----------
   Plain build (libdaalaref.so): the C tables OD_FDCT_2D_C / OD_IDCT_2D_C.  The
   -DOD_X86ASM build (libdaalaref_simd.so, the reference's x86 sources compiled with their
   own flags, Makefile.am:48-60,136-141) applies what od_state_opt_vtbl_init_x86 applies
   (src/x86/x86state.c:66-90): SSE4.1 / AVX2 versions of the 4x4 and 8x8 transforms, the
   only parts of this stage the reference has intrinsics for.  ref_stage_simd() reports
   which. */
static od_dct_func_2d ref_fdct_tab[OD_NBSIZES + 1];
static od_dct_func_2d ref_idct_tab[OD_NBSIZES + 1];
static int ref_tab_state = -1;
#if defined(OD_X86ASM)
# include "x86/cpu.h"
# include "x86/x86int.h"
#endif

static void ref_tabs_init(void) {
  int i;
  if (ref_tab_state >= 0) return;
  for (i = 0; i <= OD_NBSIZES; i++) {
    ref_fdct_tab[i] = OD_FDCT_2D_C[i];
    ref_idct_tab[i] = OD_IDCT_2D_C[i];
  }
  ref_tab_state = 0;
#if defined(OD_X86ASM)
  {
    uint32_t flags;
    flags = od_cpu_flags_get();
    if (flags & OD_CPU_X86_SSE2) {
      ref_fdct_tab[0] = od_bin_fdct4x4_sse2;
      ref_idct_tab[0] = od_bin_idct4x4_sse2;
      ref_fdct_tab[1] = od_bin_fdct8x8_sse2;
      ref_idct_tab[1] = od_bin_idct8x8_sse2;
      ref_tab_state = 1;
      if (flags & OD_CPU_X86_SSE4_1) {
        ref_fdct_tab[0] = od_bin_fdct4x4_sse41;
        ref_idct_tab[0] = od_bin_idct4x4_sse41;
        ref_fdct_tab[1] = od_bin_fdct8x8_sse41;
        ref_idct_tab[1] = od_bin_idct8x8_sse41;
        ref_tab_state = 2;
      }
      if (flags & OD_CPU_X86_AVX2) {
        ref_fdct_tab[1] = od_bin_fdct8x8_avx2;
        ref_idct_tab[1] = od_bin_idct8x8_avx2;
        ref_tab_state = 3;
      }
    }
  }
#endif
}

/* 0: C tables; 1 / 2 / 3: SSE2 / SSE4.1 / SSE4.1 + AVX2 transforms for 4x4 and 8x8. */
REF_EXPORT int ref_stage_simd(void) {
  ref_tabs_init();
  return ref_tab_state;
}

/* ---- 1-D / 2-D transforms (src/dct.c:54-84 tables) ---------------------- */
REF_EXPORT void ref_fdct_1d(int ln, od_coeff *y, const od_coeff *x,
 int xstride) {
  (*OD_FDCT_1D[ln])(y, x, xstride);
}

REF_EXPORT void ref_idct_1d(int ln, od_coeff *x, int xstride,
 const od_coeff *y) {
  (*OD_IDCT_1D[ln])(x, xstride, y);
}

REF_EXPORT void ref_fdct_2d(int ln, od_coeff *y, int ystride,
 const od_coeff *x, int xstride) {
  (*OD_FDCT_2D_C[ln])(y, ystride, x, xstride);
}

REF_EXPORT void ref_idct_2d(int ln, od_coeff *x, int xstride,
 const od_coeff *y, int ystride) {
  (*OD_IDCT_2D_C[ln])(x, xstride, y, ystride);
}

/* Batched helpers so python-side timing/looping overhead stays out. */
REF_EXPORT void ref_fdct_2d_batch(int ln, od_coeff *y, const od_coeff *x,
 long nblocks) {
  long b;
  int n;
  n = 4 << ln;
  for (b = 0; b < nblocks; b++) {
    (*OD_FDCT_2D_C[ln])(y + b*n*n, n, x + b*n*n, n);
  }
}

REF_EXPORT void ref_idct_2d_batch(int ln, od_coeff *x, const od_coeff *y,
 long nblocks) {
  long b;
  int n;
  n = 4 << ln;
  for (b = 0; b < nblocks; b++) {
    (*OD_IDCT_2D_C[ln])(x + b*n*n, n, y + b*n*n, n);
  }
}

/* ---- deringing (src/dering.c) --------------------------------------------- */
#include "dering.h"
REF_EXPORT void ref_dering(int16_t *y, int ystride, const int16_t *x, int xstride, int nhb, int nvb,
 int sbx, int sby, int nhsb, int nvsb, int xdec, int dir[OD_DERING_NBLOCKS][OD_DERING_NBLOCKS],
 int pli, unsigned char *bskip, int skip_stride, int threshold, int overlap, int coeff_shift) {
  od_dering(&OD_DERING_VTBL_C, y, ystride, x, xstride, nhb, nvb, sbx, sby, nhsb, nvsb, xdec, dir, pli,
   bskip, skip_stride, threshold, overlap, coeff_shift);
}

/* ---- lapping filters (src/filter.c) ------------------------------------- */
REF_EXPORT void ref_pre_filter(int f, od_coeff *y, const od_coeff *x) {
  (*OD_PRE_FILTER[f])(y, x);
}

REF_EXPORT void ref_post_filter(int f, od_coeff *x, const od_coeff *y) {
  (*OD_POST_FILTER[f])(x, y);
}

REF_EXPORT void ref_prefilter_split(od_coeff *c0, int stride, int bs,
 int hfilter, int vfilter) {
  od_prefilter_split(c0, stride, bs, OD_FILT_SIZE(bs - 1, 0), hfilter,
   vfilter);
}

REF_EXPORT void ref_postfilter_split(od_coeff *c0, int stride, int bs,
 int hfilter, int vfilter) {
  od_postfilter_split(c0, stride, bs, OD_FILT_SIZE(bs - 1, 0), 0, NULL, 0,
   hfilter, vfilter);
}

REF_EXPORT void ref_apply_prefilter_frame_sbs(od_coeff *c, int stride,
 int nhsb, int nvsb, int xdec, int ydec) {
  od_apply_prefilter_frame_sbs(c, stride, nhsb, nvsb, xdec, ydec);
}

REF_EXPORT void ref_apply_postfilter_frame_sbs(od_coeff *c, int stride,
 int nhsb, int nvsb, int xdec, int ydec) {
  od_apply_postfilter_frame_sbs(c, stride, nhsb, nvsb, xdec, ydec, 0, NULL, 0);
}

/* ---- pixel <-> coefficient (src/state.c:1216-1323, lossy path) ----------------
   ref_set_fpr(1): full-precision references - every px pointer of this file's plane
   functions is an array of 16-bit samples at 12 bits (the reference's xstride 2), strides
   in SAMPLES; the reference's own conversion branches for xstride 2 do the work. */
static int ref_fpr;
REF_EXPORT void ref_set_fpr(int on) {
  ref_fpr = on != 0;
}

REF_EXPORT void ref_px_to_coeff(od_coeff *dst, int dst_stride,
 unsigned char *src, int src_stride, int w, int h) {
  od_ref_buf_to_coeff(NULL, dst, dst_stride, 0, src, ref_fpr ? 2 : 1, src_stride << ref_fpr, w, h);
}

REF_EXPORT void ref_coeff_to_px(unsigned char *dst, int dst_stride,
 od_coeff *src, int src_stride, int w, int h) {
  od_coeff_to_ref_buf(NULL, dst, ref_fpr ? 2 : 1, dst_stride << ref_fpr, src, src_stride, 0, w, h);
}

/* ---- forward lapped-transform pyramid of one plane ----------------------
   Drives the reference functions in the reference's own depth-first order
   (od_compute_dcts, src/encode.c:1455-1512, with every block forced to split
   down to 4x4), recording the fDCT of every block at every level.
   levels[bs] (bs = 0..top) receives a full w x h plane holding the 2-D DCT
   of every (4<<bs)-square block of that level; c is left holding the fully
   pre-filtered samples. */
static void ref_pyramid_rec(od_coeff **levels, od_coeff *c, int w, int bx,
 int by, int bs, int pic_w, int pic_h) {
  int n;
  int bo;
  n = 4 << bs;
  bo = by*n*w + bx*n;
  ref_tabs_init();
  (*ref_fdct_tab[bs])(levels[bs] + bo, w, c + bo, w);
  if (bs > 0) {
    int hfilter;
    int vfilter;
    /* Same gating expressions as src/encode.c:1487-1488: unpadded *luma*
       picture size, also for chroma planes. */
    hfilter = (bx + 1) << (OD_LOG_BSIZE0 + bs) <= pic_w;
    vfilter = (by + 1) << (OD_LOG_BSIZE0 + bs) <= pic_h;
    od_prefilter_split(c + bo, w, bs, OD_FILT_SIZE(bs - 1, 0), hfilter,
     vfilter);
    ref_pyramid_rec(levels, c, w, 2*bx + 0, 2*by + 0, bs - 1, pic_w, pic_h);
    ref_pyramid_rec(levels, c, w, 2*bx + 1, 2*by + 0, bs - 1, pic_w, pic_h);
    ref_pyramid_rec(levels, c, w, 2*bx + 0, 2*by + 1, bs - 1, pic_w, pic_h);
    ref_pyramid_rec(levels, c, w, 2*bx + 1, 2*by + 1, bs - 1, pic_w, pic_h);
  }
}

REF_EXPORT void ref_forward_pyramid_plane(od_coeff **levels, od_coeff *c,
 unsigned char *px, int px_stride, int w, int h, int dec, int pic_w,
 int pic_h) {
  int nhsb;
  int nvsb;
  int sbx;
  int sby;
  int top;
  top = OD_NBSIZES - 1 - dec;
  nhsb = w >> (OD_LOG_BSIZE_MAX - dec);
  nvsb = h >> (OD_LOG_BSIZE_MAX - dec);
  od_ref_buf_to_coeff(NULL, c, w, 0, px, ref_fpr ? 2 : 1, px_stride << ref_fpr, w, h);
  od_apply_prefilter_frame_sbs(c, w, nhsb, nvsb, dec, dec);
  for (sby = 0; sby < nvsb; sby++) {
    for (sbx = 0; sbx < nhsb; sbx++) {
      ref_pyramid_rec(levels, c, w, sbx, sby, top, pic_w, pic_h);
    }
  }
}

/* Inverse at one uniform partition level `bs`: iDCT of every block, undo the
   split filters of levels bs+1..top (od_postfilter_split, rows then columns,
   children before parents like od_encode_recursive src/encode.c:1780-1789),
   undo the superblock-edge filter, convert to pixels. */
static void ref_inverse_rec(od_coeff *c, const od_coeff *d, int w, int bx,
 int by, int bs, int leaf_bs, int pic_w, int pic_h) {
  int n;
  int bo;
  n = 4 << bs;
  bo = by*n*w + bx*n;
  if (bs == leaf_bs) {
    ref_tabs_init();
    (*ref_idct_tab[bs])(c + bo, w, d + bo, w);
  }
  else {
    int hfilter;
    int vfilter;
    hfilter = (bx + 1) << (OD_LOG_BSIZE0 + bs) <= pic_w;
    vfilter = (by + 1) << (OD_LOG_BSIZE0 + bs) <= pic_h;
    ref_inverse_rec(c, d, w, 2*bx + 0, 2*by + 0, bs - 1, leaf_bs, pic_w,
     pic_h);
    ref_inverse_rec(c, d, w, 2*bx + 1, 2*by + 0, bs - 1, leaf_bs, pic_w,
     pic_h);
    ref_inverse_rec(c, d, w, 2*bx + 0, 2*by + 1, bs - 1, leaf_bs, pic_w,
     pic_h);
    ref_inverse_rec(c, d, w, 2*bx + 1, 2*by + 1, bs - 1, leaf_bs, pic_w,
     pic_h);
    od_postfilter_split(c + bo, w, bs, OD_FILT_SIZE(bs - 1, 0), 0, NULL, 0,
     hfilter, vfilter);
  }
}

REF_EXPORT void ref_inverse_level_plane(unsigned char *px, int px_stride,
 od_coeff *c, const od_coeff *d, int w, int h, int dec, int leaf_bs,
 int pic_w, int pic_h) {
  int nhsb;
  int nvsb;
  int sbx;
  int sby;
  int top;
  top = OD_NBSIZES - 1 - dec;
  nhsb = w >> (OD_LOG_BSIZE_MAX - dec);
  nvsb = h >> (OD_LOG_BSIZE_MAX - dec);
  for (sby = 0; sby < nvsb; sby++) {
    for (sbx = 0; sbx < nhsb; sbx++) {
      ref_inverse_rec(c, d, w, sbx, sby, top, leaf_bs, pic_w, pic_h);
    }
  }
  od_apply_postfilter_frame_sbs(c, w, nhsb, nvsb, dec, dec, 0, NULL, 0);
  od_coeff_to_ref_buf(NULL, px, ref_fpr ? 2 : 1, px_stride << ref_fpr, c, w, 0, w, h);
}

/* ---- coefficient scan (src/partition.c:144-194) ------------------------- */
REF_EXPORT void ref_raster_to_coding_order(od_coeff *dst, int n,
 const od_coeff *src, int stride) {
  od_raster_to_coding_order(dst, n, src, stride);
}

REF_EXPORT void ref_coding_order_to_raster(od_coeff *dst, int stride,
 const od_coeff *src, int n) {
  od_coding_order_to_raster(dst, stride, src, n);
}

REF_EXPORT int ref_band_offsets(int bs, int *out) {
  int i;
  int nb;
  nb = OD_BAND_OFFSETS[bs][0];
  for (i = 0; i <= nb; i++) out[i] = OD_BAND_OFFSETS[bs][1 + i];
  return nb;
}

/* ---- quantisation matrices (src/pvq.c:322-381) -------------------------- */
REF_EXPORT int ref_qm_buffer_size(void) { return OD_QM_BUFFER_SIZE; }

REF_EXPORT void ref_init_qm(int16_t *qm, int16_t *qm_inv, int flat) {
  od_init_qm(qm, qm_inv, flat ? OD_QM8_Q4_FLAT : OD_QM8_Q4_HVS);
}

REF_EXPORT int ref_qm_offset(int bs, int xydec) {
  return od_qm_offset(bs, xydec);
}

REF_EXPORT int ref_qm_get_index(int bs, int band) {
  return od_qm_get_index(bs, band);
}

REF_EXPORT int ref_pvq_beta(int use_masking, int pli, int bs, int band) {
  return OD_PVQ_BETA[use_masking][pli][bs][band];
}

/* ---- PVQ search (src/pvq_encoder.c:93-224) ------------------------------ */
REF_EXPORT double ref_pvq_search_rdo_double(const int16_t *xcoeff, int n,
 int k, od_coeff *ypulse, double g2, double pvq_norm_lambda, int prev_k) {
  return pvq_search_rdo_double(xcoeff, n, k, ypulse, g2, pvq_norm_lambda,
   prev_k);
}

/* x is [nbands][n] int16, y is [nbands][n] int32 (in/out when prev_k > 0). */
REF_EXPORT void ref_pvq_search_batch(const int16_t *x, int n, const int *k,
 od_coeff *y, const double *g2, double pvq_norm_lambda, const int *prev_k,
 double *cos_out, long nbands) {
  long b;
  for (b = 0; b < nbands; b++) {
    cos_out[b] = pvq_search_rdo_double(x + b*n, n, k[b], y + b*n, g2[b],
     pvq_norm_lambda, prev_k ? prev_k[b] : 0);
  }
}

/* ---- PVQ fixed-point helpers (src/pvq.c) -------------------------------- */
REF_EXPORT int ref_vector_log_mag(const od_coeff *x, int n) {
  return od_vector_log_mag(x, n);
}

REF_EXPORT int32_t ref_pvq_compute_gain(const int16_t *x, int n, int q0,
 int32_t *g, int beta, int bshift) {
  return od_pvq_compute_gain(x, n, q0, g, (od_val16)beta, bshift);
}

REF_EXPORT int32_t ref_gain_expand(int32_t cg, int q0, int beta) {
  return od_gain_expand(cg, q0, (od_val16)beta);
}

REF_EXPORT int ref_pvq_compute_max_theta(int32_t qcg, int beta) {
  return od_pvq_compute_max_theta(qcg, (od_val16)beta);
}

REF_EXPORT int32_t ref_pvq_compute_theta(int t, int max_theta) {
  return od_pvq_compute_theta(t, max_theta);
}

REF_EXPORT int ref_pvq_compute_k(int32_t qcg, int itheta, int32_t theta,
 int noref, int n, int beta, int nodesync) {
  return od_pvq_compute_k(qcg, itheta, theta, noref, n, (od_val16)beta,
   nodesync);
}

REF_EXPORT int ref_pvq_cos(int32_t x) { return od_pvq_cos(x); }
REF_EXPORT int ref_pvq_sin(int32_t x) { return od_pvq_sin(x); }

REF_EXPORT int ref_compute_householder(int16_t *r, int n, int32_t gr,
 int *sign, int shift) {
  return od_compute_householder(r, n, gr, sign, shift);
}

REF_EXPORT void ref_apply_householder(int16_t *out, const int16_t *x,
 const int16_t *r, int n) {
  od_apply_householder(out, x, r, n);
}

REF_EXPORT void ref_pvq_synthesis_partial(od_coeff *xcoeff,
 const od_coeff *ypulse, const int16_t *r16, int n, int noref, int32_t g,
 int32_t theta, int m, int s, const int16_t *qm_inv) {
  od_pvq_synthesis_partial(xcoeff, ypulse, r16, n, noref, g, theta, m, s,
   qm_inv);
}

/* ---- one band through pvq_theta (src/pvq_encoder.c:333-641) -------------
   speed > 0 selects the closed-form rate model (:252-264) so the result does
   not depend on the adaptive entropy-coder state; speed == 0 prices with a
   freshly reset keyframe/inter adaptation context. */
REF_EXPORT int ref_pvq_theta(od_coeff *out, const od_coeff *x0,
 const od_coeff *r0, int n, int q0, od_coeff *y, int *itheta, int *max_theta,
 int *vk, int beta, double *skip_diff, int nodesync, int is_keyframe, int pli,
 const int16_t *qm, const int16_t *qm_inv, double pvq_norm_lambda, int speed) {
  od_adapt_ctx *adapt;
  int ret;
  adapt = (od_adapt_ctx *)malloc(sizeof(*adapt));
  memset(adapt, 0, sizeof(*adapt));
  od_adapt_pvq_ctx_reset(&adapt->pvq, is_keyframe);
  ret = pvq_theta(out, x0, r0, n, q0, y, itheta, max_theta, vk,
   (od_val16)beta, skip_diff, nodesync, is_keyframe, pli, adapt, qm, qm_inv,
   pvq_norm_lambda, speed);
  free(adapt);
  return ret;
}

/* Required by pvq_encoder.c's od_pvq_encode when encode.c is not linked
   (same two stubs the reference's own coefficient-coder test needs). */
#if defined(REF_SHIM_STANDALONE)
void od_encode_checkpoint(const daala_enc_ctx *enc, od_rollback_buffer *rbuf) {
  (void)enc;
  (void)rbuf;
}

void od_encode_rollback(daala_enc_ctx *enc, const od_rollback_buffer *rbuf) {
  (void)enc;
  (void)rbuf;
}
#endif

/* ---- od_pvq_rate at speed == 0 on a LIVE, adapted context (tests/test_rate_host.py) -----
   ref_adapt_new: a freshly reset adaptation context; ref_adapt_code: codes one codeword into a
   scratch range coder with the LIVE codeword context, i.e. adapts it as the encoder's real
   coding does; ref_pvq_rate0: the reference's own pricing of a candidate against it
   (src/pvq_encoder.c:247-287, speed = 0: the context is copied, never changed). */
REF_EXPORT void *ref_adapt_new(int is_keyframe) {
  od_adapt_ctx *a;
  a = (od_adapt_ctx *)calloc(1, sizeof(*a));
  od_adapt_pvq_ctx_reset(&a->pvq, is_keyframe);
  return a;
}

REF_EXPORT void ref_adapt_free(void *a) {
  free(a);
}

REF_EXPORT void ref_adapt_code(void *a, const od_coeff *y, int n, int k) {
  od_ec_enc ec;
  od_ec_enc_init(&ec, 1000);
  od_encode_pvq_codeword(&ec, &((od_adapt_ctx *)a)->pvq.pvq_codeword_ctx, y, n, k);
  od_ec_enc_clear(&ec);
}

REF_EXPORT double ref_pvq_rate0(void *a, int qg, int icgr, int theta, int ts, const od_coeff *y, int k,
 int n, int is_keyframe, int pli) {
  return od_pvq_rate(qg, icgr, theta, ts, (od_adapt_ctx *)a, y, k, n, is_keyframe, pli, 0);
}

REF_EXPORT int ref_codeword_ctx_size(void) {
  return (int)sizeof(od_pvq_codeword_ctx);
}

REF_EXPORT double ref_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9*ts.tv_nsec;
}

/* ---- the whole hot-path stage on one plane, with the reference's functions --
   Same work per block as bench.py's GPU step, so the two can be timed side by
   side (cpu_baseline.kind == "reference"):
     forward pyramid (above), then for every level bs and every block:
       od_raster_to_coding_order, pvq_theta per band on the no-reference keyframe
       path (speed = 1: closed-form rate), od_coding_order_to_raster,
     then iDCT + post-filters + pixels at that level (ref_inverse_level_plane).
   q_band/beta_band: [5][12] per-band quantiser and beta; qm/qm_inv: coding-order
   tables for this plane's decimation, one slice per bs at qm_off[bs].
   Returns the number of transform blocks processed. */
static unsigned char *const *g_recon_levels;   /* ref_stage_plane_levels: one recon per level */
/* ref_stage_set_inter(1): the stage functions below run pvq_theta as an INTER frame does
   (is_keyframe = 0, src/encode.c:1326-1360): ref_levels then holds the transformed
   motion-compensated prediction of every block at every level (mctmp / mdtmp in the
   encoder, od_encode_compute_pred :880-886), used as it is - no chroma-from-luma flip. */
static int ref_stage_inter;
REF_EXPORT void ref_stage_set_inter(int on) {
  ref_stage_inter = on != 0;
}

/* ref_stage_set_dump: the stage functions below also hand back what pvq_theta decided for
   every band of every block - what north_star calls "coefficients and PVQ pulse vectors":
   y_levels[bs] int32 [blocks][len] = the pulse vector y[] of each block in coding order
   (len = min(N*N, 512); zero where pvq_theta wrote nothing, i.e. DC and bands whose gain 0 /
   skip won), band_levels[bs] int32 [blocks][nb][4] = {return value (the coded gain
   index), itheta, max_theta, k}.  Blocks in raster order of the plane.  NULL, NULL = off. */
static int32_t *const *g_dump_y;
static int32_t *const *g_dump_band;
REF_EXPORT void ref_stage_set_dump(int32_t *const *y_levels, int32_t *const *band_levels) {
  g_dump_y = y_levels;
  g_dump_band = band_levels;
}

static long stage_plane_core(unsigned char *px, int px_stride, int w, int h,
 int dec, int pic_w, int pic_h, int pli, const int16_t *qm, const int16_t *qm_inv,
 const int *qm_off, const int *q_band, const int *beta_band,
 double pvq_norm_lambda, unsigned char *recon_px, od_coeff **dq_out,
 od_coeff *const *ref_levels) {
  od_coeff *levels[OD_NBSIZES];
  od_coeff *c;
  od_coeff *dq;
  od_adapt_ctx *adapt;
  long nblocks;
  int top;
  int bs;
  top = OD_NBSIZES - 1 - dec;
  nblocks = 0;
  c = (od_coeff *)malloc(sizeof(*c)*w*h);
  dq = (od_coeff *)malloc(sizeof(*dq)*w*h);
  adapt = (od_adapt_ctx *)calloc(1, sizeof(*adapt));
  od_adapt_pvq_ctx_reset(&adapt->pvq, 1);
  for (bs = 0; bs <= top; bs++) levels[bs] = (od_coeff *)malloc(sizeof(*c)*w*h);
  ref_forward_pyramid_plane(levels, c, px, px_stride, w, h, dec, pic_w, pic_h);
  for (bs = 0; bs <= top; bs++) {
    int n;
    int bx;
    int by;
    int nb;
    const int *off;
    long nblocks_level;
    n = 4 << bs;
    nb = OD_BAND_OFFSETS[bs][0];
    off = &OD_BAND_OFFSETS[bs][1];
    nblocks_level = 0;
    for (by = 0; by < h/n; by++) {
      for (bx = 0; bx < w/n; bx++) {
        od_coeff in[OD_BSIZE_MAX*OD_BSIZE_MAX];
        od_coeff out[OD_BSIZE_MAX*OD_BSIZE_MAX];
        od_coeff ref0[OD_BSIZE_MAX*OD_BSIZE_MAX];
        od_coeff y[OD_BSIZE_MAX*OD_BSIZE_MAX];
        double skip_diff;
        int i;
        int bo;
        bo = by*n*w + bx*n;
        od_raster_to_coding_order(in, n, levels[bs] + bo, w);
        if (ref_levels != NULL && ref_stage_inter) {
          od_raster_to_coding_order(ref0, n, ref_levels[bs] + bo, w);
        }
        else if (ref_levels != NULL) {
          /* keyframe chroma: the chroma-from-luma prediction and its sign, as
             od_pvq_encode applies it before the band loop
             (src/pvq_encoder.c:846-872; that block is not callable on its own,
             so its dot product and negation are spelled out here) */
          int32_t xy;
          const int16_t *bqm;
          od_raster_to_coding_order(ref0, n, ref_levels[bs] + bo, w);
          bqm = qm + qm_off[bs];
          xy = 0;
          for (i = off[0]; i < off[1]; i++) {
            int32_t rq;
            int32_t inq;
            rq = (int32_t)((uint32_t)ref0[i]*(uint32_t)(int32_t)bqm[i]);
            inq = (int32_t)((uint32_t)in[i]*(uint32_t)(int32_t)bqm[i]);
            xy = (int32_t)((uint32_t)xy + (uint32_t)((rq*(int64_t)inq)
             >> ((OD_QM_SHIFT + OD_CFL_FLIP_SHIFT) << 1)));
          }
          if (xy < 0) for (i = off[0]; i < off[nb]; i++) ref0[i] = -ref0[i];
        }
        else memset(ref0, 0, sizeof(*ref0)*n*n);
        skip_diff = 0;
        if (g_dump_y != NULL) memset(y, 0, sizeof(*y)*off[nb]);
        for (i = 0; i < nb; i++) {
          int itheta;
          int max_theta;
          int k;
          int qg;
          qg = pvq_theta(out + off[i], in + off[i], ref0 + off[i], off[i + 1] - off[i],
           q_band[bs*12 + i], y + off[i], &itheta, &max_theta, &k,
           (od_val16)beta_band[bs*12 + i], &skip_diff, 1, !ref_stage_inter, pli, adapt,
           qm + qm_off[bs] + off[i], qm_inv + qm_off[bs] + off[i],
           pvq_norm_lambda, 1);
          if (g_dump_band != NULL && g_dump_band[bs] != NULL) {
            int32_t *rec;
            rec = g_dump_band[bs] + ((size_t)nblocks_level*nb + i)*4;
            rec[0] = qg;
            rec[1] = itheta;
            rec[2] = max_theta;
            rec[3] = k;
          }
        }
        if (g_dump_y != NULL && g_dump_y[bs] != NULL) {
          int32_t *dst;
          dst = g_dump_y[bs] + (size_t)nblocks_level*off[nb];
          for (i = 0; i < off[nb]; i++) dst[i] = y[i];
        }
        nblocks_level++;
        out[0] = in[0];
        /* src/encode.c:1388: what PVQ never codes is zero on a keyframe, the prediction's
           own coefficients on an inter frame (src/state.c:1347-1366) */
        if (ref_levels != NULL && ref_stage_inter) {
          od_coeff predblk[OD_BSIZE_MAX*OD_BSIZE_MAX];
          int r;
          for (r = 0; r < n; r++) memcpy(predblk + r*n, ref_levels[bs] + bo + r*w, sizeof(od_coeff)*n);
          od_init_skipped_coeffs(dq, predblk, 0, bo, n, w);
        }
        else od_init_skipped_coeffs(dq, NULL, 1, bo, n, w);
        od_coding_order_to_raster(dq + bo, w, out, n);
        nblocks++;
      }
    }
    if (dq_out != NULL && dq_out[bs] != NULL) memcpy(dq_out[bs], dq, sizeof(*dq)*w*h);
    ref_inverse_level_plane(g_recon_levels != NULL && g_recon_levels[bs] != NULL
     ? g_recon_levels[bs] : recon_px, w, c, dq, w, h, dec, bs, pic_w, pic_h);
  }
  for (bs = 0; bs <= top; bs++) free(levels[bs]);
  free(adapt);
  free(dq);
  free(c);
  return nblocks;
}

/* Batch pricing with the reference's own od_pvq_rate (speed = 1: closed form; the
   adaptation context is not read on that path). */
#define PRICE_NAME(x) ref_##x
#define PRICE_COEFF od_coeff
#define PRICE_RATE(qg, icgr, theta, ts, y, k, n, kf, pli) \
  od_pvq_rate(qg, icgr, theta, ts, NULL, y, k, n, kf, pli, 1)
REF_EXPORT void ref_price_noref(double *rate, const unsigned char *rec, const int16_t *y, long B,
 int nb, const int *off, int len, int is_keyframe, int pli);
REF_EXPORT void ref_price_ref(double *rate, const unsigned char *rec, const unsigned char *items,
 const int16_t *y, long B, int nb, const int *off, int len, int is_keyframe, int pli);
#include "price_batch.inc"
#undef PRICE_NAME
#undef PRICE_COEFF
#undef PRICE_RATE

REF_EXPORT long ref_stage_plane(unsigned char *px, int px_stride, int w, int h,
 int dec, int pic_w, int pic_h, int pli, const int16_t *qm, const int16_t *qm_inv,
 const int *qm_off, const int *q_band, const int *beta_band,
 double pvq_norm_lambda, unsigned char *recon_px) {
  return stage_plane_core(px, px_stride, w, h, dec, pic_w, pic_h, pli, qm, qm_inv,
   qm_off, q_band, beta_band, pvq_norm_lambda, recon_px, NULL, NULL);
}

/* The same stage with, optionally, the dequantised coefficient plane of every
   level handed back (dq_out[bs], w x h each, may be NULL) and, optionally, a
   reference plane per level (ref_levels[bs], same layout as the coefficient
   planes): every block is then coded by pvq_theta WITH that reference, after
   the chroma-from-luma sign flip - the path keyframe chroma takes in the
   reference encoder (src/encode.c:1680-1687).  bench.py times luma with dq_out
   and chroma with ref_levels built from it (the non-TF branch of
   od_resample_luma_coeffs, src/intra.c:97-108). */
REF_EXPORT long ref_stage_plane_cfl(unsigned char *px, int px_stride, int w, int h,
 int dec, int pic_w, int pic_h, int pli, const int16_t *qm, const int16_t *qm_inv,
 const int *qm_off, const int *q_band, const int *beta_band,
 double pvq_norm_lambda, unsigned char *recon_px, od_coeff **dq_out,
 od_coeff *const *ref_levels) {
  return stage_plane_core(px, px_stride, w, h, dec, pic_w, pic_h, pli, qm, qm_inv,
   qm_off, q_band, beta_band, pvq_norm_lambda, recon_px, dq_out, ref_levels);
}

/* ref_stage_plane_cfl keeping the reconstruction of EVERY partition level
   (recon_levels[bs], w x h each) instead of the last one only: what the full-frame
   parity checks of the GPU pipeline compare with.  Not thread-safe (test use). */
REF_EXPORT long ref_stage_plane_levels(unsigned char *px, int px_stride, int w, int h,
 int dec, int pic_w, int pic_h, int pli, const int16_t *qm, const int16_t *qm_inv,
 const int *qm_off, const int *q_band, const int *beta_band,
 double pvq_norm_lambda, unsigned char *const *recon_levels, od_coeff **dq_out,
 od_coeff *const *ref_levels) {
  long n;
  g_recon_levels = recon_levels;
  n = stage_plane_core(px, px_stride, w, h, dec, pic_w, pic_h, pli, qm, qm_inv,
   qm_off, q_band, beta_band, pvq_norm_lambda, recon_levels[0], dq_out, ref_levels);
  g_recon_levels = NULL;
  return n;
}
