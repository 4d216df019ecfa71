/* oracle/od_oracle.h - CPU restatement of the Daala block-transform hot path.

   TEST INFRASTRUCTURE.  This is the checker for the HIP kernels in
   daala_amd/csrc; it is never the thing measured or shipped.  Only tests/,
   __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.

   Every function cites the reference file:line (relative to the xiph/daala
   tree) whose arithmetic it restates.  Parity is PINNED: tests/test_oracle_*.py
   check each function against the reference itself (oracle/_ref, built from the
   reference sources by oracle/Makefile) when present, and against committed
   golden vectors under tests/golden/ that were generated from it
   (tools/make_golden.py). */
#ifndef OD_ORACLE_H
#define OD_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t odo_coeff; /* od_coeff, src/filter.h:29 */

#define ODO_NBSIZES 5        /* 4,8,16,32,64: src/internal.h:53-59 */
#define ODO_MAX_PVQ_SIZE 128 /* src/pvq.h:55 */
#define ODO_MAX_CANDS 24     /* 20 with-reference items (src/pvq_encoder.c:289) + 2 noref + slack */

/* ---- transforms: src/dct.c ------------------------------------------------ */
void odo_fdct_1d(int ln, odo_coeff *y, const odo_coeff *x, int xstride);
void odo_idct_1d(int ln, odo_coeff *x, int xstride, const odo_coeff *y);
void odo_fdct_2d(int ln, odo_coeff *y, int ystride, const odo_coeff *x, int xstride);
void odo_idct_2d(int ln, odo_coeff *x, int xstride, const odo_coeff *y, int ystride);
void odo_fdct_2d_batch(int ln, odo_coeff *y, const odo_coeff *x, long nblocks);
void odo_idct_2d_batch(int ln, odo_coeff *x, const odo_coeff *y, long nblocks);

/* ---- 4-point lapping filter and its drivers: src/filter.c ----------------- */
void odo_pre_filter4(odo_coeff y[4], const odo_coeff x[4]);
void odo_post_filter4(odo_coeff x[4], const odo_coeff y[4]);
/* n = 4, 8, 16, 32 (src/filter.c:147-1321); return -1 for any other n.  In place allowed. */
int odo_pre_filter(int n, odo_coeff *y, const odo_coeff *x);
int odo_post_filter(int n, odo_coeff *x, const odo_coeff *y);
void odo_prefilter_split(odo_coeff *c0, int stride, int bs, int hfilter, int vfilter);
void odo_postfilter_split(odo_coeff *c0, int stride, int bs, int hfilter, int vfilter);
void odo_apply_prefilter_frame_sbs(odo_coeff *c0, int stride, int nhsb, int nvsb, int xdec, int ydec);
void odo_apply_postfilter_frame_sbs(odo_coeff *c0, int stride, int nhsb, int nvsb, int xdec, int ydec);

/* ---- pixel <-> coefficient: src/state.c:1216-1323 (8-bit, lossy) ---------- */
void odo_set_fpr(int on);
void odo_px16_to_coeff(odo_coeff *dst, int dst_stride, const uint16_t *src, int src_stride, int w, int h);
void odo_coeff_to_px16(uint16_t *dst, int dst_stride, const odo_coeff *src, int src_stride, int w, int h);
void odo_img_plane_copy_pad16(uint16_t *dst, int dstride, int plane_w, int plane_h, const void *src,
 int src_bitdepth, int sstride, int pic_w, int pic_h);
void odo_px_to_coeff(odo_coeff *dst, int dst_stride, const uint8_t *src, int src_stride, int w, int h);
void odo_coeff_to_px(uint8_t *dst, int dst_stride, const odo_coeff *src, int src_stride, int w, int h);

/* ---- whole-plane stages (level-by-level restatement of the recursion in
        src/encode.c:1455-1512 / :1660-1789) -------------------------------- */
void odo_forward_pyramid_plane(odo_coeff **levels, odo_coeff *c, const uint8_t *px, int px_stride,
 int w, int h, int dec, int pic_w, int pic_h);
void odo_inverse_level_plane(uint8_t *px, int px_stride, odo_coeff *c, const odo_coeff *d,
 int w, int h, int dec, int leaf_bs, int pic_w, int pic_h);

/* ---- scan order: src/partition.c:77-194 ----------------------------------- */
void odo_raster_to_coding_order(odo_coeff *dst, int n, const odo_coeff *src, int stride);
void odo_coding_order_to_raster(odo_coeff *dst, int stride, const odo_coeff *src, int n);
int odo_band_offsets(int bs, int *out);
void odo_resample_luma_coeffs(odo_coeff *chroma_pred, int cpstride,
 const odo_coeff *decoded_luma, int dlstride, int bs, int luma_bs);
void odo_img_plane_copy_pad(uint8_t *dst, int dstride, int plane_w, int plane_h,
 const uint8_t *src, int sstride, int pic_w, int pic_h);

/* ---- PVQ fixed-point helpers: src/pvq.c ----------------------------------- */
int odo_ilog(uint32_t v);
int odo_vector_log_mag(const odo_coeff *x, int n);
int32_t odo_pvq_compute_gain(const int16_t *x, int n, int q0, int32_t *g, int beta, int bshift);
int32_t odo_gain_expand(int32_t cg, int q0, int beta);
int odo_pvq_compute_max_theta(int32_t qcg, int beta);
int32_t odo_pvq_compute_theta(int t, int max_theta);
int odo_pvq_compute_k(int32_t qcg, int itheta, int32_t theta, int noref, int n, int beta, int nodesync);
int odo_pvq_cos(int32_t x);
int odo_pvq_sin(int32_t x);
int odo_compute_householder(int16_t *r, int n, int32_t gr, int *sign, int shift);
void odo_apply_householder(int16_t *out, const int16_t *x, const int16_t *r, int n);
void odo_pvq_synthesis_partial(odo_coeff *xcoeff, const odo_coeff *ypulse, const int16_t *r16, int n,
 int noref, int32_t g, int32_t theta, int m, int s, const int16_t *qm_inv);

/* ---- PVQ search: src/pvq_encoder.c:93-224 --------------------------------- */
double odo_pvq_search_rdo_double(const int16_t *xcoeff, int n, int k, odo_coeff *ypulse, double g2,
 double pvq_norm_lambda, int prev_k);
void odo_pvq_search_batch(const int16_t *x, int n, const int *k, odo_coeff *y, const double *g2,
 double pvq_norm_lambda, const int *prev_k, double *cos_out, long nbands);

/* ---- one band: src/pvq_encoder.c:333-641 ----------------------------------
   Everything in pvq_theta that does NOT depend on the adaptive entropy-coder
   state is captured per candidate; the final choice needs a rate, which is the
   closed form of od_pvq_rate (src/pvq_encoder.c:252-264) when speed > 0. */
typedef struct {
  int32_t with_ref; /* 1: theta search candidate (:507-565); 0: noref (:578-609) */
  int32_t gain;     /* i */
  int32_t theta;    /* j, or -1 */
  int32_t ts;
  int32_t k;
  int32_t qcg;
  int32_t qtheta;
  int32_t searched; /* 0 when pruned by the dist0 test (:531,:588) */
  double cos_dist;
  double dist;
  int32_t y[ODO_MAX_PVQ_SIZE];
} odo_pvq_cand;

typedef struct {
  int32_t xshift, rshift;
  int32_t g, gr, cg, cgr, icgr, gain_offset;
  int32_t m, s, theta;
  double corr, dist0, skip_dist;
  int16_t x16[ODO_MAX_PVQ_SIZE];
  int16_t r16[ODO_MAX_PVQ_SIZE]; /* after od_compute_householder when the theta search ran */
  int32_t ncands;
  odo_pvq_cand cands[ODO_MAX_CANDS];
} odo_pvq_band_trace;

double odo_pvq_rate_speed1(int qg, int icgr, int theta, int ts, const odo_coeff *y0, int k,
 int n, int is_keyframe, int pli);
int odo_cfl_flip(odo_coeff *ref, const odo_coeff *in, const int16_t *qm, int bs);
int odo_pvq_theta(odo_coeff *out, const odo_coeff *x0, const odo_coeff *r0, int n, int q0,
 odo_coeff *y, int *itheta, int *max_theta, int *vk, int beta, double *skip_diff, int nodesync,
 int is_keyframe, int pli, const int16_t *qm, const int16_t *qm_inv, double pvq_norm_lambda,
 int speed, odo_pvq_band_trace *trace);

/* pvq_decode_partition (src/pvq_decoder.c:122-298) after its entropy-decoder reads: the
   reconstructed band from the decoded gain symbol, itheta, noref, the pulses and the
   reference band; returns the skip code, *k_out = the pulse count the decoder derives. */
int odo_pvq_decode_band(odo_coeff *out, const odo_coeff *ref, const odo_coeff *y, int n, int q0,
 int beta, int qg_coded, int itheta, int noref, int is_keyframe, int pli, const int16_t *qm,
 const int16_t *qm_inv, int *k_out);

/* The whole stage on one plane (forward pyramid, PVQ noref bands of every block
   at every level, dequantisation, inverse at every level): the CPU "port" that
   bench.py times when oracle/_ref is absent.  rate_mode 0 = distortion-only
   choice (what the GPU bench step does), 1 = od_pvq_rate closed form. */
/* ---- deringing filter, src/dering.c ------------------- */
int odo_dir_find8(const int16_t *img, int stride, int32_t *var, int coeff_shift);
void odo_dering(int16_t *y, int ystride, const int16_t *x, int xstride, int nhb, int nvb, int sbx,
 int sby, int nhsb, int nvsb, int xdec, int dir[8][8], int pli, const unsigned char *bskip,
 int skip_stride, int threshold, int overlap, int coeff_shift);
void odo_dering_plane(int16_t *y, const int16_t *x, int stride, int nhsb, int nvsb, int xdec,
 int32_t *dirs, int pli, const unsigned char *bskip, int skip_stride, const int32_t *thresholds,
 int overlap, int coeff_shift);

long odo_stage_plane(const uint8_t *px, int px_stride, int w, int h, int dec, int pic_w, int pic_h,
 int pli, const int16_t *qm, const int16_t *qm_inv, const int *qm_off, const int *q_band,
 const int *beta_band, double pvq_norm_lambda, int rate_mode, uint8_t *recon_px);

double odo_now(void);

#ifdef __cplusplus
}
#endif
/* od_compute_dist, src/encode.c:1082-1226 */
void odo_dist_8x8_parts(double parts[3], const odo_coeff *x, const odo_coeff *y,
 const odo_coeff *e_lp, int stride, int use_masking);
double odo_compute_dist(const odo_coeff *x, const odo_coeff *y, int n, int flat_qm, int use_masking,
 int coded_quantizer);
#endif
