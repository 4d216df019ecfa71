/* daala_hip.h - C ABI of libdaalahip.so: the MI355X (gfx950) implementation of
   Daala's block-transform encode hot path (lapped pre/post filter, 4..64 point
   integer DCT/iDCT, PVQ band search), as a drop-in for the reference's own
   surfaces for that path.

   Plain C: pointers, sizes and ints only.  Two families of entry points:

   (1) PER-CALL, HOST-POINTER functions with exactly the reference's
       signatures and conventions (synchronous, caller-owned host buffers, void
       return).  These are what the reference's function-pointer tables / call
       sites bind.  One small block per call is
       dominated by launch + PCIe latency; they exist for parity and for
       drop-in completeness, not for throughput.

   (2) BATCHED, DEVICE-POINTER functions (prefix odhip_) that the per-call
       ones are built on and that a batched caller (bench.py, the frame-sharded
       driver) uses directly: inputs already resident in HBM, asynchronous on a
       caller-supplied HIP stream, int return (0 or a negative OD_E* value as
       in the reference's include/daala/codec.h:89-103).

   All coefficient data is od_coeff = int32_t (reference src/filter.h:29) at
   scale 2^4 (OD_COEFF_SHIFT, src/internal.h:124).  Results are bit-exact with
   the reference C path; the CPU checker lives in oracle/ (tests only). */
#ifndef DAALA_HIP_H
#define DAALA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t od_coeff;

#define ODHIP_SUCCESS (0)
#define ODHIP_EFAULT (-1)   /* OD_EFAULT: HIP runtime failure / bad pointer */
#define ODHIP_EINVAL (-10)  /* OD_EINVAL: bad argument */
#define ODHIP_EIMPL (-23)   /* OD_EIMPL: not implemented */
#define ODHIP_ERANGE (-24)  /* a band needs more pulses than ODHIP_PVQ_MAX_K: odhip_pvq_k_range_take */

#define ODHIP_NBSIZES 5     /* OD_NBSIZES: 4,8,16,32,64 (src/internal.h:53-59) */

/* A HIP stream as an opaque pointer (hipStream_t); NULL = the default stream. */
typedef void *odhip_stream;

/* ------------------------------------------------------------------------ */
/* (1) Per-call surfaces                                                     */
/* ------------------------------------------------------------------------ */

/* od_dct_func_2d, reference src/dct.h:61-62:
     void f(od_coeff *out, int out_stride, const od_coeff *in, int in_stride)
   Strides in elements.  Replace OD_FDCT_2D_C / OD_IDCT_2D_C
   (src/dct.c:54-68), i.e. od_bin_fdctNxN / od_bin_idctNxN (src/dct.c:151-163,
   351-363, 792-806, 4890-4920), in od_state_opt_vtbl.fdct_2d[] / idct_2d[]
   (src/state.h:128-129).  Called from src/encode.c:1304,1308,1397,1410,1449,
   1477 and src/decode.c:521,596.  Arbitrary od_coeff input: exact 32-bit
   products like the C. */
typedef void (*odhip_dct_func_2d)(od_coeff *out, int out_stride,
 const od_coeff *in, int in_stride);

void od_bin_fdct4x4_hip(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_bin_idct4x4_hip(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_bin_fdct8x8_hip(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_bin_idct8x8_hip(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_bin_fdct16x16_hip(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_bin_idct16x16_hip(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_bin_fdct32x32_hip(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_bin_idct32x32_hip(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_bin_fdct64x64_hip(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_bin_idct64x64_hip(od_coeff *x, int xstride, const od_coeff *y, int ystride);

/* Installer with the shape of od_state_opt_vtbl_init_x86
   (src/x86/x86state.c:39-97): overwrites the five fdct_2d and five idct_2d
   slots of a table pair.  The one line of reference glue is
     odhip_install_dct_vtbl(state->opt_vtbl.fdct_2d, state->opt_vtbl.idct_2d);
   after od_state_opt_vtbl_init_c (src/state.c:346-352). */
void odhip_install_dct_vtbl(odhip_dct_func_2d fdct_2d[ODHIP_NBSIZES],
 odhip_dct_func_2d idct_2d[ODHIP_NBSIZES]);

/* od_filter_func, reference src/filter.h:43 (OD_PRE_FILTER[0] /
   OD_POST_FILTER[0] = od_pre_filter4 / od_post_filter4, src/filter.c:147-222):
   four samples, in place allowed. */
void od_pre_filter4_hip(od_coeff y[4], const od_coeff x[4]);
void od_post_filter4_hip(od_coeff x[4], const od_coeff y[4]);
/* The larger lapping filters of the same tables, od_pre_filter8/16/32 and
   od_post_filter8/16/32 (src/filter.c:279-1321, the TYPE3 parameter sets its
   `#elif 1` chains select).  The codec does not use them (OD_FILT_SIZE == 0,
   src/filter.h:77); the reference's own tests do (dcttest dynamic_range,
   src/dct.c:8606; the filter test).  odhip_install_filter_tables fills
   OD_PRE_FILTER[0..3] / OD_POST_FILTER[0..3]-shaped arrays (src/filter.c:115-127)
   with the host-pointer functions; odhip_filter_batch runs `count` independent
   (4 << f)-tap filters on contiguous device vectors [count][4 << f] (16-byte
   aligned; in place allowed), f = 0..3, inverse = 0 pre / 1 post. */
typedef void (*odhip_filter_func)(od_coeff out[], const od_coeff in[]);   /* od_filter_func, src/filter.h:43 */
void od_pre_filter8_hip(od_coeff y[8], const od_coeff x[8]);
void od_post_filter8_hip(od_coeff x[8], const od_coeff y[8]);
void od_pre_filter16_hip(od_coeff y[16], const od_coeff x[16]);
void od_post_filter16_hip(od_coeff x[16], const od_coeff y[16]);
void od_pre_filter32_hip(od_coeff y[32], const od_coeff x[32]);
void od_post_filter32_hip(od_coeff x[32], const od_coeff y[32]);
void odhip_install_filter_tables(odhip_filter_func pre[4], odhip_filter_func post[4]);
int odhip_filter_batch(int f, int inverse, od_coeff *d_out, const od_coeff *d_in, long count,
 odhip_stream stream);

/* Block- and frame-level lapping drivers, same signatures as the reference
   (src/filter.h:80-87; definitions src/filter.c:1459-1619), in place on a host
   plane.  `f` must be 0 (OD_FILT_SIZE is identically 0, src/filter.h:77); q,
   skip, skip_stride are unused exactly as in the reference's non-deblocking
   build.  Link-time replacements for the calls at src/encode.c:1489,1760,1789,
   2571,2675 and src/decode.c:807,823,962,994. */
void od_prefilter_split_hip(od_coeff *c0, int stride, int bs, int f,
 int hfilter, int vfilter);
void od_postfilter_split_hip(od_coeff *c0, int stride, int bs, int f, int q,
 unsigned char *skip, int skip_stride, int hfilter, int vfilter);
void od_apply_prefilter_frame_sbs_hip(od_coeff *c, int stride, int nhsb,
 int nvsb, int xdec, int ydec);
void od_apply_postfilter_frame_sbs_hip(od_coeff *c, int stride, int nhsb,
 int nvsb, int xdec, int ydec, int q, unsigned char *skip, int skip_stride);

/* pvq_search_rdo_double, reference src/pvq_encoder.c:93-224 (file-static
   there; BASELINE.json calls it od_pvq_search_rdo_double).  Same arguments and
   return value: xcoeff[n] int16 (od_val16), ypulse[n] in/out (input only when
   prev_k > 0), returns the cosine distance.  Called from pvq_theta,
   src/pvq_encoder.c:542,589. */
double od_pvq_search_rdo_double_hip(const int16_t *xcoeff, int n, int k,
 od_coeff *ypulse, double g2, double pvq_norm_lambda, int prev_k);
/* od_pvq_synthesis_partial, src/pvq.h:164 (src/pvq.c:1037-1115): the synthesis of one decoded / chosen band -
   what od_pvq_decode's pvq_decode_partition (src/pvq_decoder.c:77-89) and pvq_theta (src/pvq_encoder.c:631) call. */
void od_pvq_synthesis_partial_hip(od_coeff *xcoeff, const od_coeff *ypulse, const int16_t *r16, int n, int noref,
 int32_t g, int32_t theta, int m, int s, const int16_t *qm_inv);

/* ------------------------------------------------------------------------ */
/* (2) Batched device-pointer API                                            */
/* ------------------------------------------------------------------------ */

/* Library / device bring-up.  odhip_init selects the HIP device for the
   calling thread and returns 0, or ODHIP_EFAULT when no gfx950 device is
   usable (there is NO CPU fallback in this library). */
int odhip_init(int device);
const char *odhip_version(void);

/* Contexts (shaped after the reference's per-od_state backend
   installation, src/x86/x86state.c:39-97: nothing is process-global).  Every
   piece of state the batched entry points keep between calls - edge strips of
   the inverse stage, scratch / job tables / side streams / sort arrays /
   profiling events of the two PVQ band stages, the list of bands inside the
   device-acos margin - is owned by an odhip_ctx.  The batched odhip_* functions
   below use the CURRENT context of the calling thread: the one passed to
   odhip_make_current, or (NULL / never called) a default context the thread gets
   for the HIP device that is current when it calls in.  Rules:
     - one call sequence in flight per context (calls of a context are ordered on
       the caller's stream; scratch is reused from call to call);
     - sequences that overlap - two streams, e.g. the luma and the chroma chain of
       a frame batch, or two host threads, or two devices - use one context each;
     - a context belongs to the device it was created for; calling with another
       current device returns ODHIP_EINVAL.
   odhip_create returns NULL when the device does not exist; odhip_destroy waits
   for the device to go idle, then frees everything the context owns. */
typedef struct odhip_ctx odhip_ctx;
odhip_ctx *odhip_create(int device);
void odhip_destroy(odhip_ctx *ctx);
int odhip_make_current(odhip_ctx *ctx);    /* also makes ctx's device current */
odhip_ctx *odhip_get_current(void);        /* NULL = the thread's default context */
int odhip_ctx_device(const odhip_ctx *ctx);
/* serial != 0: the band stages of this context launch everything on the caller's
   stream instead of forking independent kernels onto the context's side streams
   (exclusive kernel durations for profiling; ODHIP_PVQ_SERIAL=1 does the same for
   every context). */
int odhip_ctx_set_serial(odhip_ctx *ctx, int serial);

/* ---- quantiser set-up (host) ----------------------------

   What the reference derives on the host per encoder / per frame and the band
   stages take as plain data:
     odhip_init_qm        od_init_qm, src/pvq.c:322-381 (same arguments: x, x_inv
                          of ODHIP_QM_BUFFER_SIZE int16, qm = an 8x8 Q4 matrix):
                          per-coefficient QM with magnitude compensation (Q11) and
                          its inverse (Q12) in coding order, for every block size
                          and both decimations; block (bs, xydec) starts at
                          odhip_qm_offset(bs, xydec) (od_qm_offset, src/pvq.c:306).
                          Entries the reference leaves unwritten (32x32 / 64x64
                          blocks have coding positions for 512 coefficients only)
                          are zero here.
     odhip_interp_qm      od_interp_qm as the frame header code selects its
                          entries per plane (src/encode.c:2903-2940, :3052-3072)
                          from the encoder's default matrices: pvq_qm_q4 of plane
                          pli for rc.base_quantizer.
     odhip_qm_get_index   od_qm_get_index, src/pvq.c:408-413.
     odhip_quant_setup    both of the above into one odhip_quant: quantizer =
                          od_state.quantizer (0 = lossless), base_quantizer =
                          rc.base_quantizer, hvs_qm = OD_SET_QM (1 = HVS, the
                          encoder's default), use_masking = OD_SET_ACTIVITY_MASKING.
     odhip_quant_bands    per band of block size bs in plane pli: the step
                          max(1, q0*pvq_qm_q4[index(bs, i + 1)] >> 4) with q0 =
                          max(1, quantizer) (src/pvq_encoder.c:874,
                          src/encode.c:1336) and OD_PVQ_BETA[masking][pli][bs][i]
                          (Q12, src/pvq.c:243-268); returns the number of bands. */
#define ODHIP_QM_SIZE 30              /* OD_QM_SIZE, src/pvq.h:108 */
#define ODHIP_QM_BUFFER_SIZE 10912    /* OD_QM_BUFFER_SIZE, src/pvq.h:72-74 */
typedef struct {
  int quantizer;
  int base_quantizer;
  int use_masking;
  int hvs_qm;
  uint8_t pvq_qm_q4[3][ODHIP_QM_SIZE];
  int16_t qm[ODHIP_QM_BUFFER_SIZE];
  int16_t qm_inv[ODHIP_QM_BUFFER_SIZE];
} odhip_quant;
void odhip_init_qm(int16_t *x, int16_t *x_inv, const int *qm);
int odhip_qm_offset(int bs, int xydec);
int odhip_qm_get_index(int bs, int band);
int odhip_interp_qm(uint8_t out[ODHIP_QM_SIZE], int base_quantizer, int use_masking, int pli);
int odhip_quant_setup(odhip_quant *qt, int base_quantizer, int quantizer, int use_masking,
 int hvs_qm);
int odhip_quant_bands(const odhip_quant *qt, int pli, int bs, int32_t *q_band, int32_t *beta_band);

/* nblocks contiguous N x N tiles, N = 4 << ln; d_out may equal d_in; both 16-byte
   aligned (ODHIP_EINVAL otherwise: the kernels move 16-byte vectors).
   exact32 != 0 forces exact 32-bit products (arbitrary input); 0 uses the
   24-bit multiplier, valid while |intermediate| < 2^23 (always true for data
   that came from pixels). */
int odhip_fdct2d_batch(int ln, od_coeff *d_out, const od_coeff *d_in,
 long nblocks, int exact32, odhip_stream stream);
int odhip_idct2d_batch(int ln, od_coeff *d_out, const od_coeff *d_in,
 long nblocks, int exact32, odhip_stream stream);

/* Every N x N block of a w x h plane (w, h multiples of N; pointers 16-byte
   aligned, strides multiples of 4), strides in
   elements: the `d` plane layout of the reference encoder (block (bx,by)'s
   coefficient (v,u) at [(by*N+v)*stride + bx*N+u]). */
int odhip_fdct2d_plane(int ln, od_coeff *d_out, int out_stride,
 const od_coeff *d_in, int in_stride, int w, int h, int exact32,
 odhip_stream stream);
int odhip_idct2d_plane(int ln, od_coeff *d_out, int out_stride,
 const od_coeff *d_in, int in_stride, int w, int h, int exact32,
 odhip_stream stream);

/* Forward lapped-transform pyramid of a batch of planes: for each plane
     od_ref_plane_to_coeff            (src/state.c:1216-1277)
     od_apply_prefilter_frame_sbs     (src/filter.c:1529-1559)
     for bs = top .. 0:  fdct_2d[bs] of every block, then od_prefilter_split
                         of every block   (src/encode.c:1455-1512 forced to
                                           split everywhere)
   d_px:  nplanes planes of w x h 8-bit pixels, plane p at d_px + p*px_plane_stride,
          row stride px_stride.
   d_levels[bs] (bs = 0..4-dec): nplanes coefficient planes of w x h, plane p at
          + p*(long)w*h, row stride w.  NULL skips the store of that level.
   dec:   0 for luma (64x64 superblocks), 1 for 4:2:0 chroma (32x32).
   pic_w, pic_h: UNPADDED LUMA picture size; gates the split filters exactly
          like src/encode.c:1487-1488.  w, h: multiples of 64 >> dec. */
int odhip_forward_pyramid(od_coeff *const d_levels[ODHIP_NBSIZES],
 const uint8_t *d_px, int px_stride, long px_plane_stride, int nplanes, int w,
 int h, int dec, int pic_w, int pic_h, odhip_stream stream);

/* Inverse at a uniform partition level leaf_bs for a batch of planes:
   idct_2d[leaf_bs] of every block, od_postfilter_split for levels
   leaf_bs+1 .. top (src/encode.c:1780-1789), od_apply_postfilter_frame_sbs
   (src/filter.c:1589-1618), od_coeff_to_ref_plane (src/state.c:1281-1345).
   d_coef: nplanes planes w x h, row stride w.  d_px as above (output). */
int odhip_inverse_level(uint8_t *d_px, int px_stride, long px_plane_stride,
 const od_coeff *d_coef, int nplanes, int w, int h, int dec, int leaf_bs,
 int pic_w, int pic_h, odhip_stream stream);

/* odhip_inverse_level for several partition levels of ONE plane set (same
   nplanes, w, h, dec; at most 5) in a single set of launches: level leaf_bs[i]
   is reconstructed from d_coef[i] into d_px[i] (distinct buffers, same strides). */
int odhip_inverse_levels(uint8_t *const *d_px, int px_stride, long px_plane_stride,
 const od_coeff *const *d_coef, const int *leaf_bs, int nlevels, int nplanes, int w, int h,
 int dec, int pic_w, int pic_h, odhip_stream stream);

/* Inverse at an ARBITRARY partition: the decoder's reconstruction of nplanes planes
   from their dequantised coefficient planes (d_coef, layout as above) and the
   block-size map of their frames: idct_2d of every leaf block, od_postfilter_split of
   every split node (od_decode_recursive, src/decode.c:603-660; the encoder's final pass
   src/encode.c:1657-1810 is the same), od_apply_postfilter_frame_sbs, od_coeff_to_ref_plane.
   d_bsize: od_state.bsize (src/state.h:250-259) - one byte per 8x8 LUMA area, 0 = four
   4x4 blocks ... 4 = one 64x64 block, rows of bstride bytes, origin at the frame's
   first superblock (the reference's pointer already skips its one-superblock border);
   plane p uses the map at d_bsize + (p / planes_per_frame)*bsize_frame_stride (luma:
   planes_per_frame = 1; 4:2:0 chroma with Cb and Cr of a frame adjacent: 2).  dec = 1
   derives the chroma blocks (one size down, 4x4 for 8x8 and 4x4 luma). */
int odhip_inverse_partition(uint8_t *d_px, int px_stride, long px_plane_stride,
 const od_coeff *d_coef, int nplanes, int w, int h, int dec, const uint8_t *d_bsize, int bstride,
 long bsize_frame_stride, int planes_per_frame, int pic_w, int pic_h, odhip_stream stream);
/* The same for ONE plane with host pointers (synchronous; staging through device
   scratch): coef = the decoder's state.dtmp[pli] (stride w), bsize = state.bsize. */
int odhip_inverse_partition_host(uint8_t *px, int px_stride, const od_coeff *coef, int w, int h, int dec,
 const uint8_t *bsize, int bstride, int pic_w, int pic_h);

/* Batched pvq_search_rdo_double: band b has d_x[b*n .. b*n+n) int16,
   d_k[b], d_g2[b], optional d_prev_k[b] (NULL = 0; when > 0 d_y holds the
   previous pulses), writes d_y[b*n ..) and d_cos[b].  n <= 128; 0 <= k <= 65535
   (pulse magnitudes are kept in 16 bits): a band with K outside that range is not
   searched, its d_y is zero and its d_cos NaN. */
int odhip_pvq_search_batch(const int16_t *d_x, int n, const int32_t *d_k,
 od_coeff *d_y, const double *d_g2, double pvq_norm_lambda,
 const int32_t *d_prev_k, double *d_cos, long nbands, odhip_stream stream);

/* The same search (src/pvq_encoder.c:93-224, no chain: prev_k = 0) in its ROW form - one band per
   quad (n = 31, 32) or per 16-lane row (n = 127, 128), what the 32- and 128-coefficient bands of
   the with-reference stage run.  Its greedy pulses are screened in single precision and replayed
   with the reference's left-to-right double-precision scan whenever the screen cannot vouch for
   the argmax (a candidate within a relative 2^-17 of the best key that is not its exact
   duplicate); d_replays[b], when given, receives the number of pulses of band b that were
   replayed, force_scan != 0 replays every one (the cross-check).  Other n: ODHIP_EINVAL;
   K is clamped to 0..32767. */
int odhip_pvq_search_row_batch(const int16_t *d_x, int n, const int32_t *d_k, od_coeff *d_y,
 const double *d_g2, double pvq_norm_lambda, int force_scan, double *d_cos, int32_t *d_replays,
 long nbands, odhip_stream stream);

/* ---- PVQ band stage (the data-parallel part of pvq_theta) -------------------

   For every block of side N = 4 << bs of a batch of coefficient planes and
   every PVQ band of that block size (OD_BAND_OFFSETS, src/partition.c:77-91),
   the no-reference path of pvq_theta (src/pvq_encoder.c:333-641) up to, but
   not including, the rate-dependent choice: QM scaling to x16 (:381,:398),
   companded gain (:404), null-candidate distortion (:417), and for both gain
   candidates (:578-609) the pulse count K, the pruning test (:588), the
   K-pulse search and the distortion (:593-595).  The choice needs the rate of
   each candidate from the ADAPTIVE entropy coder (od_pvq_rate, :247-287), which
   is sequential host state in the reference and stays there; the host (or
   odhip_pvq_select_synth_noref with a rate table) applies `cost <= best_cost`.

   Block index: blk = (plane*(h/N) + by)*(w/N) + bx; B = number of blocks. */
#define ODHIP_MAX_BANDS 12
#define ODHIP_PVQ_MAX_K 32767
/* The reference's pulse count K is an int (od_pvq_compute_k, src/pvq.c:436-470); the pulse vectors here are int16.  A
   candidate the reference SEARCHES with K above ODHIP_PVQ_MAX_K - seen only below encoder_example's quantiser range, on
   saturated content in 64x64 blocks - is never searched or chosen by the band stages, so the band would differ from the
   reference's.  Every such band is counted on the device, per context (with-reference bands conservatively: before the
   reference's pruning test).  odhip_pvq_k_range_take: the current context's counts since the last call, cleared; returns
   ODHIP_ERANGE when either is non-zero (blocking copies; call it after the stream of the band stage has been synchronised).
   odhip_pipe_sync calls it and returns ODHIP_ERANGE: the results of the steps since the last sync are NOT the reference's. */
int odhip_pvq_k_range_take(unsigned *noref_bands, unsigned *ref_bands);
/* One record per (block, band), 64 bytes, 64-byte aligned: both halves are
   whole 32-byte HBM sectors and are written by different kernels (the first by
   the preparation pass, the second by the search). */
typedef struct {
  int32_t cg;          /* companded gain of x, Q8 (od_pvq_compute_gain)            */
  int32_t gain[2];     /* candidate gain index i; 0 = slot unused                  */
  int16_t k[2];        /* pulses (od_pvq_compute_k), saturated at ODHIP_PVQ_MAX_K  */
  uint8_t flags[2];    /* 1 = searched, 0 = pruned / unused, 2 = K above
                          ODHIP_PVQ_MAX_K (not representable: never chosen)        */
  uint8_t reserved0[6];
  double dist0;        /* distortion of the null (gain 0) candidate                */
  int32_t yy[2];       /* sum of squared pulses of the candidate                   */
  double dist[2];      /* distortion of the candidate                              */
  int32_t moment[2];   /* SUM i*|y_i| of the candidate's pulses: od_pvq_rate's
                          centre-of-mass sum (src/pvq_encoder.c:258-259)           */
} odhip_pvq_band;

typedef struct {
  odhip_pvq_band *band; /* [B][nb]                                                  */
  int16_t *y;           /* [2][B][len] signed pulse vectors in coding order (index
                           0 = DC slot, unused, may be overwritten with 0),
                           len = min(N*N, 512); 16-byte aligned                    */
  int32_t *choice;      /* [B][nb][4] written by odhip_pvq_select_synth_noref* /
                           odhip_pvq_choose_multi: {chosen slot, chosen gain index
                           qg (0 = null), synthesis scale, qshift}; 16-byte aligned */
  double *cos_dist;     /* optional [B][nb][2]: return value of
                           pvq_search_rdo_double per candidate (parity checks);
                           NULL = not stored                                       */
} odhip_pvq_cands;

/* One (plane set, block size) unit of work for the multi-job entry points.
   q_band / beta_band are HOST arrays [nb_bands]; everything prefixed d_ and the
   arrays inside `cands` are device memory.  d_qm is needed by the band stage,
   d_qm_inv and d_dq by select_synth; d_rate and d_qg are optional (NULL). */
typedef struct odhip_pvq_job_s {
  const od_coeff *d_coef;
  int nplanes;
  int w;
  int h;
  int bs;
  const int16_t *d_qm;
  const int16_t *d_qm_inv;
  const int32_t *q_band;
  const int32_t *beta_band;
  odhip_pvq_cands cands;
  od_coeff *d_dq;
  const double *d_rate;
  int32_t *d_qg;
  /* optional: planes plane_split .. nplanes-1 use q_band2 instead of q_band (HOST
     [nb_bands]): a chroma plane set holding the Cb planes first and the Cr planes
     after them - the per-band step comes from pvq_qm_q4[pli], which differs between
     Cb and Cr (OD_DEFAULT_QMS, src/encode.c:118-131, :3052-3072).  NULL = one table. */
  const int32_t *q_band2;
  int plane_split;
} odhip_pvq_job;

/* All jobs (at most 16) in one set of launches, so that small levels (510
   64x64 blocks per frame) overlap with large ones instead of serialising their
   tails.  Jobs of one call must use one stream. */
int odhip_pvq_noref_bands_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);
int odhip_pvq_select_synth_noref_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);

/* The choice alone (fills cands.choice and the optional d_qg of every job):
   for callers that dequantise inside the inverse stage, below. */
int odhip_pvq_choose_multi(const odhip_pvq_job *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);

/* odhip_inverse_level fed directly by the band stage: the chosen pulse vectors
   of `job` (cands.y, cands.choice after odhip_pvq_choose_multi or
   odhip_pvq_select_synth_noref*) are dequantised while the superblock tile is
   loaded (od_pvq_synthesis_partial noref src/pvq.c:1081-1092,
   od_coding_order_to_raster src/partition.c:176-194), DCs come from
   job->d_coef; the dequantised plane is never written to HBM.  Partition level
   = job->bs; needs job->d_qm_inv.  Same output as odhip_pvq_select_synth_noref
   followed by odhip_inverse_level. */
int odhip_inverse_level_pvq(uint8_t *d_px, int px_stride, long px_plane_stride,
 const odhip_pvq_job *job, int dec, int pic_w, int pic_h, odhip_stream stream);

/* Chroma-from-luma predictions of a 4:2:0 keyframe: od_resample_luma_coeffs
   (src/intra.c:72-109) for luma blocks of 8x8 and larger (:97-108: the upper-left
   quarter of the decoded luma block's coefficients), fed directly by the luma band
   stage - the decoded coefficients are dequantised on the fly from the chosen
   pulses (cands.y, cands.choice after odhip_pvq_choose_multi) exactly as
   odhip_pvq_select_synth_noref would write them; no dequantised luma plane is
   needed.  luma_jobs[j] (level bs >= 1, blocks of N = 4 << bs; needs d_qm_inv)
   yields the reference planes of the chroma level bs - 1 (blocks of N/2) in
   d_ref[j]: `copies` consecutive plane sets of [nplanes][h/2][w/2] (Cb and Cr
   share the prediction: copies = 2), ready to be odhip_pvq_refjob.d_ref.  A
   level-0 job (4x4 luma blocks) yields the 4x4 chroma predictions of the 4:2:0
   TF branch (:77-89: od_tf_up_hv_lp of the four luma blocks over the chroma block,
   src/tf.c:82-108, then OD_CFL_SCALING4), the alternative prediction of chroma
   level 0 when the luma partition there is 4x4. */
int odhip_cfl_refs_from_luma(const odhip_pvq_job *luma_jobs, int njobs, od_coeff *const *d_ref,
 int copies, odhip_stream stream);
/* prezeroed != 0: the caller guarantees that the positions a 64x64 luma level never codes
   (half of each 32x32 chroma reference block) are already zero in d_ref - buffers that were
   cleared once and are only ever written by this function - and the per-call clear of that
   plane set is skipped. */
int odhip_cfl_refs_from_luma_ex(const odhip_pvq_job *luma_jobs, int njobs, od_coeff *const *d_ref,
 int copies, int prezeroed, odhip_stream stream);

/* Profiling aid (bench.py): while enabled, odhip_pvq_noref_bands_multi brackets
   its dominant kernel - the search of the 128-coefficient bands,
   k_search<128,2,1> - with HIP events on the stream the kernel is launched on
   (up to 256 calls are kept).  odhip_pvq_profile_read waits for the recorded
   events, writes one duration in milliseconds per call to ms[] and returns how
   many (or a negative ODHIP_E* code), then starts over. */
int odhip_pvq_profile(int enable);
int odhip_pvq_profile_read(float *ms, int max_n);

/* odhip_inverse_level_pvq for several partition levels of ONE plane set (same
   nplanes, w, h, dec; at most 5 jobs) in a single set of launches: level i of
   jobs[] is reconstructed into d_px[i] (distinct buffers, same strides). */
int odhip_inverse_levels_pvq(uint8_t *const *d_px, int px_stride, long px_plane_stride,
 const odhip_pvq_job *jobs, int njobs, int dec, int pic_w, int pic_h, odhip_stream stream);

/* nb_bands, offsets[nb_bands+1] and len for block size bs. */
int odhip_pvq_band_layout(int bs, int *nb_bands, int *offsets, int *len);

/* d_qm: QM in coding order for this (bs, decimation): od_state.qm +
   od_qm_offset(bs, xdec) (src/pvq.c:306, src/encode.c:1355-1358), int16,
   device.  q_band / beta_band: HOST arrays [nb_bands]: the per-band quantiser
   max(1, q0*pvq_qm_q4[od_qm_get_index(bs, i+1)] >> 4) (src/pvq_encoder.c:874)
   and OD_PVQ_BETA[masking][pli][bs][i] (src/pvq.c:243-268). */
int odhip_pvq_noref_bands(const od_coeff *d_coef, int nplanes, int w, int h,
 int bs, const int16_t *d_qm, const int32_t *q_band, const int32_t *beta_band,
 double pvq_norm_lambda, const odhip_pvq_cands *out, odhip_stream stream);

/* Choice + decoder-identical dequantisation of the chosen candidate into d_dq
   (same plane layout as d_coef; DC passed through; uncoded positions zero):
   od_gain_expand (src/pvq.c:766), od_pvq_synthesis_partial noref
   (src/pvq.c:1037-1093), od_coding_order_to_raster (src/partition.c:176).
   d_rate: [B][nb][2] bits per candidate from the host entropy model, or NULL
   (distortion-only choice).  There is no slot for the null (gain 0) candidate:
   its rate is identically zero in the reference (od_pvq_rate returns 0 for k == 0
   and adds the theta terms only when qg > 0, src/pvq_encoder.c:250-251,:276), so
   best_cost starts at dist0 exactly as at :417-421.  d_qg_out: optional [B][nb] chosen gain index. */
int odhip_pvq_select_synth_noref(od_coeff *d_dq, const od_coeff *d_coef,
 int nplanes, int w, int h, int bs, const int16_t *d_qm_inv,
 const int32_t *q_band, const int32_t *beta_band, double pvq_norm_lambda,
 const odhip_pvq_cands *in, const double *d_rate, int32_t *d_qg_out,
 odhip_stream stream);

/* ---- with-reference (theta / Householder) path: data-parallel pieces -----------

   pvq_theta with a reference vector r (inter frames, chroma-from-luma;
   src/pvq_encoder.c:381-565) around its one libm call, acos(corr) (:478), which
   is not reproducible bit for bit on the device and stays on the host, as does
   od_pvq_rate (adaptive entropy coder).  All vectors are plain band vectors
   [band][n] in coding order, device memory; n <= 128 (OD_MAX_PVQ_SIZE).

   odhip_pvq_ref_prepare    :381-438 and the Householder reflection
                            (od_compute_householder / od_apply_householder,
                            src/pvq.c:498-623): x16, r16 (r16[m] updated as the
                            reference does when the reflection is computed, i.e.
                            when r != 0 and corr > 0), the reflected x without
                            element m in d_xr [band][n-1], and one record per
                            band.  q0 = per-band quantiser, beta = OD_PVQ_BETA
                            entry (Q12), cfl_enabled as at :410.
   odhip_pvq_ref_candidates :466-504 given d_theta[band] =
                            floor(.5 + OD_THETA_SCALE*acos(corr)) from the host:
                            the (gain, theta) candidates with their K
                            (od_pvq_compute_max_theta, od_pvq_compute_theta,
                            od_pvq_compute_k nodesync branch, src/pvq.c:855-953)
                            in the reference's stable (k, gain) order; up to
                            ODHIP_PVQ_MAX_REFCANDS per band.
   odhip_pvq_synthesis      od_pvq_synthesis_partial (src/pvq.c:1037-1115) of a
                            chosen candidate; d_params[band] = {noref, g
                            (od_gain_expand result), theta (Q15 angle), m, s};
                            d_y holds n pulses (n-1 used when noref == 0).

   The K-pulse searches of the candidates are odhip_pvq_search_batch on d_xr. */
typedef struct {
  int32_t xshift, rshift;
  int32_t g, gr;          /* raw gains (od_pvq_compute_gain *g)                 */
  int32_t cg, cgr;        /* companded gains, Q8 (cgr = 256 when cfl_enabled)   */
  int32_t icgr, gain_offset;
  int32_t m, s;           /* Householder pivot and sign (0, 1 when not computed) */
  int32_t r_null;         /* 1 = the reference vector is all zero                */
  int32_t reserved;
  double corr;            /* clamped correlation, :436-438                       */
  double reserved2;
} odhip_pvq_refprep;      /* 64 bytes */

#define ODHIP_PVQ_MAX_REFCANDS 24
typedef struct {
  int32_t gain, theta, ts, k, qcg, qtheta;
} odhip_pvq_refcand;

int odhip_pvq_ref_prepare(const od_coeff *d_x0, const od_coeff *d_r0, int n, long nbands,
 const int16_t *d_qm, int q0, int beta, int cfl_enabled, int16_t *d_x16, int16_t *d_r16,
 int16_t *d_xr, odhip_pvq_refprep *d_out, odhip_stream stream);
int odhip_pvq_ref_candidates(const odhip_pvq_refprep *d_prep, const int32_t *d_theta, int n,
 long nbands, int beta, odhip_pvq_refcand *d_items, int32_t *d_nitems, odhip_stream stream);
int odhip_pvq_synthesis(od_coeff *d_out, const od_coeff *d_y, const int16_t *d_r16, int n,
 long nbands, const int32_t *d_params, const int16_t *d_qm_inv, odhip_stream stream);

/* ---- decoder side: a PVQ band from its decoded symbols ------------------------------

   pvq_decode_partition (src/pvq_decoder.c:122-298; od_pvq_decode :300-376 calls it per band)
   AFTER its entropy-decoder reads, for a batch of band vectors [band][n] in coding order
   (device memory): d_sym[band] = {the gain symbol as read - id & 1, or 1 + the
   generic_decode value, before deinterleaving (:190-208) -, itheta after the theta decode
   (:248-254), noref, unused}; d_y the decoded pulses (n - !noref of them, :262-266), d_ref
   the reference band (the prediction, after the chroma-from-luma flip where one was read).
   Computes what the function computes from them: the reference's scaling and gain, the
   deinterleaved gain, gain_offset, theta (:213-255), the skip rules, od_gain_expand and
   pvq_synthesis = od_compute_householder + od_pvq_synthesis_partial (:78-89, :268-279) into
   d_out; d_info (optional) [band][2] = {K as the decoder derives it under OD_ROBUST_STREAM
   (od_pvq_compute_k, nodesync), skip code 0 / OD_PVQ_SKIP_ZERO 1 / OD_PVQ_SKIP_COPY 2}.
   The entropy decoding itself (od_decode_cdf_adapt, generic_decode,
   od_decode_pvq_codeword) is sequential host state and stays the reference's; under
   OD_ROBUST_STREAM (nodesync = 1, src/encode.c:1354) none of its reads depends on a value
   computed here, so a decoder can parse a whole frame first and reconstruct its bands in
   batches.  q0 / beta: the band's quantiser step and OD_PVQ_BETA entry. */
int odhip_pvq_decode_bands(od_coeff *d_out, const od_coeff *d_ref, const od_coeff *d_y, int n,
 long nbands, const int32_t *d_sym, const int16_t *d_qm, const int16_t *d_qm_inv, int q0, int beta,
 int is_keyframe, int pli, int32_t *d_info, odhip_stream stream);

/* ---- with-reference band stage on whole planes ---------------------------------

   pvq_theta (src/pvq_encoder.c:333-641) for every block of side N = 4 << bs of a
   batch of coefficient planes and every PVQ band, WITH a reference plane of the
   same layout (keyframe chroma: the chroma-from-luma prediction; inter frames:
   the transformed motion-compensated prediction), up to but not including the
   rate-dependent choice:

     - the chroma-from-luma sign flip of od_pvq_encode (:846-872) per block when
       is_keyframe && pli != 0 (the reference plane itself is not modified: the
       flip is recorded and applied on the fly);
     - QM scaling of x and r, gains, correlation, null-candidate distortion
       (:381-455), od_compute_householder / od_apply_householder
       (src/pvq.c:498-623);
     - theta = floor(.5 + OD_THETA_SCALE*acos(corr)) (:478) with the device's
       acos.  The reference's value comes from the host libm, which is not
       reproducible bit for bit on the device; the integer theta is, except when
       OD_THETA_SCALE*acos(corr) + .5 lies within 1e-9 of an integer (the two
       libraries agree to < 1e-10 there, tests/test_gpu_pvq_refbands.py).  Such bands
       are listed and odhip_pvq_ref_resolve settles them with the host's libm;
     - the (gain, theta) candidates in the reference's order (:466-504), the
       pruning test (:531), the K-pulse searches on the reflected vector with the
       prev_k chain (:536-545) and the distortion of every candidate (:548-552);
     - the no-reference candidates (:571-609) when the reference's condition
       (:571-573) holds.

   od_pvq_rate (adaptive entropy coder) stays on the host:
   odhip_pvq_ref_select_synth_multi takes its rate table, applies `cost <
   best_cost` to the theta candidates and `cost <= best_cost` to the
   no-reference ones in the reference's order, then the skip rules (:611-622)
   and the decoder-identical synthesis (:623-633).

   Block index blk = (plane*(h/N) + by)*(w/N) + bx, B = number of blocks, nb =
   bands of the block size, len = min(N*N, 512).  Band i of block blk: record
   band[blk*nb + i]; its candidates occupy slots 0 .. nitems-1 (theta candidates
   first, in search order, then the no-reference ones); pulse vector of slot s at
   y[(s*B + blk)*len + off[i] ..] (signed int16, coding order; n-1 values for a
   theta candidate, n for a no-reference one).

   `items` holds the candidates of all bands as THREE planes of 16-byte vectors,
   each [nb][ODHIP_PVQ_REF_SLOTS][B] (block index fastest: adjacent lanes =
   adjacent blocks = adjacent vectors): vector (i*ODHIP_PVQ_REF_SLOTS + s)*B + blk
   of plane 0 is {gain, theta, ts, k}, of plane 1 (at + nb*SLOTS*B vectors)
   {qcg, qtheta, flags, yslot}, of plane 2 {cos_dist, dist}; odhip_pvq_refitem
   below names the fields (it is the gathered form, 48 bytes). */
#define ODHIP_PVQ_REF_SLOTS 16
#define ODHIP_REFBAND_R_NULL 1      /* the reference band is all zero                   */
#define ODHIP_REFBAND_THETA 2       /* the theta search ran (:452)                      */
#define ODHIP_REFBAND_NOREF 4       /* the no-reference candidates ran (:571)           */
#define ODHIP_REFBAND_FLIP 8        /* chroma-from-luma: reference of the block negated */
#define ODHIP_REFBAND_UNCERTAIN 16  /* theta within the margin: see odhip_pvq_ref_resolve */
typedef struct {
  int32_t xshift, rshift;
  int32_t g, gr;          /* raw gains                                          */
  int32_t cg, cgr;        /* companded gains, Q8 (cgr = 256 with CfL)           */
  int32_t icgr, gain_offset;
  int16_t m;              /* Householder pivot (0 when not computed)            */
  int8_t s;               /* Householder sign (1 when not computed)             */
  uint8_t flags;          /* ODHIP_REFBAND_*                                    */
  int32_t theta;          /* Q15 angle index, 0 when the theta search did not run */
  int32_t nitems;         /* candidates in items[]                              */
  int32_t ntheta;         /* how many of them are theta candidates (the first)  */
  double corr;            /* clamped correlation (:436-438)                     */
  double dist0;           /* distortion of the initial candidate (:455)         */
} odhip_pvq_refband;      /* 64 bytes */

#define ODHIP_REFITEM_SEARCHED 1    /* not pruned (:531, :588)                          */
#define ODHIP_REFITEM_WITH_REF 2    /* theta candidate                                  */
#define ODHIP_REFITEM_K_RANGE 4     /* K above ODHIP_PVQ_MAX_K: pulses would not fit the
                                       int16 vectors; reported, never searched or chosen */
#define ODHIP_REFITEM_MOMENT_SHIFT 8 /* flags >> 8: SUM i*|y_i| of the candidate's pulses
                                       (od_pvq_rate's centre-of-mass sum, :258-259; below
                                       2^24 for n <= 128, K <= 32767), 0 when not searched */
typedef struct {
  int32_t gain;           /* i                                                  */
  int32_t theta;          /* j, -1 for a no-reference candidate                 */
  int32_t ts;             /* max_theta                                          */
  int32_t k;
  int32_t qcg;
  int32_t qtheta;
  int32_t flags;          /* ODHIP_REFITEM_* in bits 0-7, the pulse moment above  */
  int32_t yslot;          /* slot holding this candidate's pulses (a candidate
                             with the K of its predecessor shares its slot,
                             :539-544); -1 = all zero                           */
  double cos_dist;        /* return value of pvq_search_rdo_double              */
  double dist;
} odhip_pvq_refitem;      /* 48 bytes */

/* choice[(blk*nb + i)*16 ..]: {item (-1 = the initial candidate), qg, noref,
   itheta, max_theta, k, skip (0, OD_PVQ_SKIP_ZERO 1, OD_PVQ_SKIP_COPY 2), the
   return value of pvq_theta (:636-637)}, then 8 words of synthesis parameters
   private to the library. */
typedef struct {
  const od_coeff *d_coef;    /* nplanes planes w x h of level bs               */
  const od_coeff *d_ref;     /* reference planes, same layout                  */
  int nplanes;
  int w;
  int h;
  int bs;
  int is_keyframe;
  int pli;
  const int16_t *d_qm;       /* as odhip_pvq_job                               */
  const int16_t *d_qm_inv;   /* select_synth only                              */
  const int32_t *q_band;     /* HOST [nb]                                      */
  const int32_t *beta_band;  /* HOST [nb]                                      */
  odhip_pvq_refband *band;   /* out [B][nb], 64-byte aligned                   */
  void *items;               /* out: 3 planes [nb][ODHIP_PVQ_REF_SLOTS][B] of 16-byte
                                vectors (see above), 16-byte aligned           */
  int16_t *y;                /* out [ODHIP_PVQ_REF_SLOTS][B][len]; y, r16, x16,
                                xr 16-byte aligned                             */
  int16_t *r16;              /* out [B][len]: QM-scaled reference after
                                od_compute_householder (input of the synthesis) */
  int16_t *x16;              /* work [B][len]                                  */
  int16_t *xr;               /* work [B][len]: reflected x without element m   */
  const double *d_rate;      /* select_synth: optional [B][nb][ODHIP_PVQ_REF_SLOTS + 1]
                                bits; entry 0 = the initial candidate (:420 /
                                :452), entry 1 + i = items[i]; NULL = choose on
                                distortion alone                               */
  int32_t *choice;           /* select_synth out [B][nb][16], 16-byte aligned  */
  od_coeff *d_dq;            /* select_synth out: dequantised planes, layout of
                                d_coef; DC passed through; uncoded positions 0 */
  const int32_t *q_band2;    /* optional, HOST [nb]: as odhip_pvq_job.q_band2 - */
  int plane_split;           /* planes plane_split.. (the Cr half) use q_band2  */
  const struct odhip_pvq_job_s *luma; /* optional (keyframe chroma): the chroma-from-luma reference
                                taken STRAIGHT from the luma band stage instead of a
                                reference plane - od_resample_luma_coeffs for luma blocks
                                of 8x8 and larger, src/intra.c:97-108.  `luma` is the
                                job of the luma level bs + 1 over the same picture(s)
                                (planes 2w x 2h, nplanes or nplanes / 2 of them: Cb and
                                Cr share the prediction) after its choice
                                (odhip_pvq_choose_multi / the priced band stage): its
                                cands.y, cands.choice and d_qm_inv are read by the
                                preparation kernels, which dequantise the co-located
                                coefficients on the fly.  d_ref may then be NULL (the
                                synthesis into planes, odhip_pvq_ref_select_synth_multi,
                                still needs it).  What odhip_cfl_refs_from_luma would
                                have written, without the planes: 0.8 GB of traffic per
                                16-frame step and a kernel less.                     */
} odhip_pvq_refjob;

/* At most 8 jobs per call, all on one stream.  The stage keeps per-call state
   (job table, scratch, uncertainty list) in the current odhip_ctx: ONE call
   sequence (bands -> resolve -> select_synth) may be in flight per context.
   odhip_pvq_ref_set_context(0 | 1) is round 1's form of that: it selects which of
   the calling thread's two DEFAULT contexts the next calls use while no explicit
   context is current (so that e.g. the Cb and the Cr planes of a batch can run on
   two streams at once); new code creates contexts with odhip_create. */
int odhip_pvq_ref_set_context(int ctx);
int odhip_pvq_ref_bands_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);
/* Settles the bands odhip_pvq_ref_bands_multi flagged ODHIP_REFBAND_UNCERTAIN:
   waits for the stream, evaluates floor(.5 + OD_THETA_SCALE*acos(corr)) with the
   HOST libm (the function the reference itself calls) for the listed bands and
   re-runs those whose theta differs.  Pass the same jobs.  Returns the number
   of bands re-run (normally 0; about 2 in 10^9 bands are listed at all), or a
   negative ODHIP_E* code.  Results are exact only after this call. */
int odhip_pvq_ref_resolve(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);
/* The same without a host wait in the middle of a pipeline: _begin (right after
   odhip_pvq_ref_bands_multi, same stream) sends the count of listed bands to
   pinned host memory; the caller enqueues whatever follows (choice, synthesis,
   inverse) speculatively; _finish waits for the count only, returns 0 when no band
   is listed (the normal case) and otherwise does what odhip_pvq_ref_resolve does
   and returns how many bands were re-run - the caller then repeats the steps that
   consumed the candidates. */
int odhip_pvq_ref_resolve_begin(odhip_stream stream);
int odhip_pvq_ref_resolve_finish(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);
int odhip_pvq_ref_select_synth_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);
/* Profiling aid (bench.py), as odhip_pvq_profile: while enabled,
   odhip_pvq_ref_bands_multi brackets the dominant kernel of this stage - the
   row-parallel search of the 128-coefficient bands, k_refb_search_row<8> - with
   HIP events on the stream the kernel is launched on (calls of the current context). */
int odhip_pvq_ref_profile(int enable);
int odhip_pvq_ref_profile_read(float *ms, int max_n);
/* The choice alone (fills `choice` incl. the synthesis parameters), for callers that
   dequantise inside the inverse stage: odhip_inverse_levels_pvq_ref reconstructs the
   partition levels jobs[i].bs of ONE plane set (same nplanes, w, h, dec; at most 5
   jobs) into d_px[i], dequantising the chosen candidates of the with-reference
   stage while the superblock tile is loaded (od_pvq_synthesis_partial with or
   without reference, skip-copy and skip-zero bands included; DCs from d_coef); the
   dequantised plane never exists in HBM.  Same pixels as
   odhip_pvq_ref_select_synth_multi followed by odhip_inverse_levels. */
int odhip_pvq_ref_choose_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);
int odhip_inverse_levels_pvq_ref(uint8_t *const *d_px, int px_stride, long px_plane_stride,
 const odhip_pvq_refjob *jobs, int njobs, int dec, int pic_w, int pic_h, odhip_stream stream);
/* The whole band stage with the priced choice (od_pvq_rate's closed form, speed > 0,
   src/pvq_encoder.c:250-264) of EVERY band made inside its search: nothing per candidate
   reaches memory.  Outputs: choice[] of every job and the winner's pulse vector of every
   coded band in slot 0 of the job's y (choice word 9 names the slot: 0, or -1 when the
   winner places no pulse); items and the other slots of y are scratch of the resolve
   paths.  The counts of bands inside the theta margin and inside the price margin are on
   their way to the host when this returns: follow with odhip_pvq_ref_resolve_finish and
   odhip_pvq_ref_choose_priced_resolve (same jobs and stream; both normally return 0; a band
   they re-run through the exporting kernels is decided again by them), then consume the
   choices (odhip_inverse_levels_pvq_ref).  A job must carry its choice buffer. */
int odhip_pvq_ref_bands_decided_multi(const odhip_pvq_refjob *jobs, int njobs,
 double pvq_norm_lambda, odhip_stream stream);
/* (Test hooks - the uncertainty margin, a deliberately wrong device theta for listed bands,
   the scale of the priced choice's margin - are per context: odhip_ctx_set_test_hooks.) */
/* S*acos(corr) + .5 as the device evaluates it (parity check of the margin). */
int odhip_pvq_ref_theta_probe(const double *d_corr, double *d_t, long n, odhip_stream stream);

/* ---- deringing filter (src/dering.c) ----------------------

   od_dering_hip has od_dering's arguments (src/dering.h:64-69, definition
   src/dering.c:252-257: ..., nhb, nvb, ...) minus the function table it
   dispatches through (od_dering_opt_vtbl): host pointers, synchronous, full
   superblocks (nhb == nvb == 8).  A reference build binds it with
     #define od_dering(vtbl, ...) od_dering_hip(__VA_ARGS__)
   at its call sites (src/encode.c:2787,2826; src/decode.c).

   odhip_dering_planes filters EVERY superblock of nplanes planes for ncand
   candidate thresholds in one launch (what the encoder's per-superblock level
   search, src/encode.c:2785-2810, evaluates): d_x int16 [plane][h][stride]
   (h = nvsb*64 >> xdec), d_y int16 [plane][cand][h][stride], d_dirs int32
   [plane][nvsb*8][nhsb*8] (written when pli == 0, read otherwise), d_bskip the
   skip map of each plane (rows of skip_stride bytes, planes
   bskip_plane_stride bytes apart), d_thresholds int32 [plane][cand][nvsb*nhsb]
   (0 leaves a superblock unchanged). */
void od_dering_hip(int16_t *y, int ystride, const int16_t *x, int xstride, int nhb, int nvb,
 int sbx, int sby, int nhsb, int nvsb, int xdec, int dir[8][8], int pli, unsigned char *bskip,
 int skip_stride, int threshold, int overlap, int coeff_shift);
int odhip_dering_planes(int16_t *d_y, const int16_t *d_x, int stride, int nhsb, int nvsb, int xdec,
 int nplanes, int32_t *d_dirs, int pli, const uint8_t *d_bskip, int skip_stride,
 long bskip_plane_stride, const int32_t *d_thresholds, int ncand, int overlap, int coeff_shift,
 odhip_stream stream);

/* ---- the deringing level search of a frame behind od_dering -----------------------

   After coding a frame the reference searches the deringing level superblock by
   superblock (src/encode.c:2697-2832): od_dering on the luma superblock for each of
   the five non-zero levels, then on the three planes with the level chosen - up to
   4 080 calls per 1080p frame, every one reading the same unfiltered copy of the frame
   (state->etmp) and skip map.  An odhip_dering_cache serves them from batched passes:
   the first call with a (plane, threshold) pair filters every superblock of the plane
   in one launch (odhip_dering_planes) and keeps the filtered plane on the host; that
   call and all later ones with the same pair copy their superblock out (and, for luma,
   the directions od_dering returns in dir[][]).  The level decision, its cost and the
   adaptation stay in the encoder.

   odhip_dering_cache_begin marks a new frame (the planes and the skip map have been
   rewritten: where the encoder copies ctmp to etmp, src/encode.c:2700-2707).
   odhip_dering_cache_call has od_dering's arguments minus the function table; x must
   point into the plane at the superblock (as the encoder's call sites do, :2787,:2826)
   and bskip into the skip map likewise.  Glue in a reference build:
     odhip_dering_cache_begin(cache);                                    at :2697
     #define od_dering(vtbl, ...) odhip_dering_cache_call(cache, __VA_ARGS__)
   Partial superblocks and chroma calls that precede every luma call of the frame go
   to the per-call path od_dering_hip.  One cache per encoder (it binds the thread's
   current odhip_ctx at creation). */
typedef struct odhip_dering_cache odhip_dering_cache;
odhip_dering_cache *odhip_dering_cache_create(void);
void odhip_dering_cache_destroy(odhip_dering_cache *c);
void odhip_dering_cache_begin(odhip_dering_cache *c);
int odhip_dering_cache_call(odhip_dering_cache *c, int16_t *y, int ystride, const int16_t *x, int xstride,
 int nhb, int nvb, int sbx, int sby, int nhsb, int nvsb, int xdec, int dir[8][8], int pli,
 unsigned char *bskip, int skip_stride, int threshold, int overlap, int coeff_shift);
void odhip_dering_cache_stats(const odhip_dering_cache *c, long *launches, long *served);
/* The level search's distortions (od_compute_dist, src/encode.c:1170-1226, called six times per
   superblock at :2776-2801) from the same batched passes: after odhip_dering_cache_begin, hand the
   frame's luma source picture (8-bit samples; host and device copies, e.g. odhip_cache_plane_pixels
   of the frame cache) to odhip_dering_cache_set_source - every luma pass then also computes the
   distortion parts of its whole output against it (odhip_dist_parts_px16) - and ask
   odhip_dering_cache_dist in front of od_compute_dist: it answers (1, *dist) only when x IS the
   source superblock and y IS the cached filtered superblock of that threshold (both compared sample
   by sample), with the host-libm finish of odhip_dist_finish; otherwise 0 and the C function runs. */
int odhip_dering_cache_set_source(odhip_dering_cache *c, const uint8_t *h_px, const uint8_t *d_px, int stride,
 int use_masking, int flat_qm);
int odhip_dering_cache_dist(odhip_dering_cache *c, const od_coeff *x, const od_coeff *y, int n, int sbx, int sby,
 int threshold, int use_masking, int flat_qm, int coded_quantizer, double *dist);
long odhip_dering_cache_dist_served(const odhip_dering_cache *c);

/* ---- od_compute_dist: the block-size RDO's distortion ----------

   od_compute_dist(enc, x, y, n) (src/encode.c:1202-1226, od_compute_dist_8x8
   :1113-1169, od_compute_var_4x4 :1082-1103) for EVERY n x n block (n = 4 << bs, bs =
   1..4) of a batch of plane pairs x (source) / y (reconstruction), both od_coeff planes
   of w x h (multiples of n) in the lapped domain, as the encoder compares them at
   src/encode.c:1418-1421, :1797-1798.  Split at the libm call:
     odhip_dist_parts   (device) per 8x8 block three doubles: the sum of the squared
                        [1 5 1]^2 low-passed error, vardist, and the ARGUMENT of pow
                        (.25 + var_stat/256) - d_parts [plane][h/8][w/8][3];
     odhip_dist_finish  (host)   activity = calibration*pow(arg, -1/6) with the host
                        libm the reference calls, activity^2*(0.92/7^4*sum + vardist), the
                        sum over the block's 8x8 blocks in raster order and the
                        coded-quantiser factor: dist [plane][h/n][w/n].
   use_masking = enc->use_activity_masking, flat_qm = (enc->qm == OD_FLAT_QM: plain
   squared error), coded_quantizer = state.coded_quantizer. */
/* Per-call form with od_compute_dist's arguments (host pointers, compact n x n blocks,
   synchronous; enc's three fields passed): a reference build binds it at the top of its
   file-static od_compute_dist (oracle/Makefile builds that variant,
   _ref/libdaalaref_disthip.so, for tests/test_gpu_dropin_encoder.py). */
double od_compute_dist_hip(const od_coeff *x, const od_coeff *y, int n, int use_masking, int flat_qm,
 int coded_quantizer);
int odhip_dist_parts(double *d_parts, const od_coeff *d_x, const od_coeff *d_y, int nplanes, int w,
 int h, int bs, int use_masking, int flat_qm, odhip_stream stream);
int odhip_dist_parts_px16(double *d_parts, const uint8_t *d_x8, int x8_stride, const int16_t *d_y16,
 int y16_stride, int nplanes, int w, int h, int bs, int use_masking, int flat_qm, odhip_stream stream);
int odhip_dist_finish(double *dist, const double *parts, int nplanes, int w, int h, int bs,
 int use_masking, int flat_qm, int coded_quantizer);

/* ---- input side: padding a picture to the coded frame size ----

   od_img_plane_copy_pad (src/encode.c:752-837; daala_image_copy_pad :1896-1909,
   called when a frame enters the input queue) for a batch of 8-bit planes of one
   size: the pic_w x pic_h picture at d_src (row stride src_stride, planes
   src_plane_stride bytes apart) is copied into the plane_w x plane_h plane at
   d_dst and extended into the padding by the reference's [1 2 1]/4 low-pass
   recurrences (right side over the picture rows, then the bottom over the whole
   width).  plane_w = frame_width >> xdec, pic_w = (pic_width + xdec) >> xdec, and
   likewise for the heights; at most 64 padded columns / rows, sides up to 8192.
   pic_w == 0 or pic_h == 0 clears the plane (:764-770). */
int odhip_image_planes_copy_pad(uint8_t *d_dst, int dst_stride, long dst_plane_stride,
 int plane_w, int plane_h, const uint8_t *d_src, int src_stride, long src_plane_stride,
 int pic_w, int pic_h, int nplanes, odhip_stream stream);

/* ---- full-precision references ------------------------------------------------------
   An encoder created with daala_info.full_precision_references keeps its picture buffers
   as 16-bit samples at 8 + OD_COEFF_SHIFT = 12 bits (src/encode.c:212-213,
   src/state.c:256-258: xstride 2), which is how 10- and 12-bit video is coded and how an
   8-bit source keeps four more bits through prediction.  odhip_ctx_set_fpr(ctx, 1) puts a
   context in that mode: every PICTURE plane its calls take or produce - the `px` arguments
   of odhip_forward_pyramid, odhip_inverse_level(s), odhip_inverse_level(s)_pvq,
   odhip_inverse_levels_pvq_ref and odhip_inverse_partition, typed uint8_t * like the
   reference's daala_image_plane.data - then holds int16 samples, 8-byte aligned, with
   strides in SAMPLES: coefficient = p - 2048 on the way in (od_ref_buf_to_coeff,
   src/state.c:1238-1254), OD_CLAMPFPR(c + 2048) on the way out (od_coeff_to_ref_buf,
   :1306-1321).  Everything between (filters, transforms, PVQ) is the same arithmetic.
   odhip_image_planes_copy_pad16 is od_img_plane_copy_pad for such buffers: the copy step is
   od_img_plane_copy's bit-depth conversion (src/state.c:93-213; src_bitdepth 8: uint8_t
   samples, 10 / 12: int16_t samples, shifted up to 12 bits and clamped), the extension runs
   on the 16-bit samples (src/encode.c:791-803, :821-832). */
int odhip_ctx_set_fpr(odhip_ctx *ctx, int on);
int odhip_ctx_get_fpr(const odhip_ctx *ctx);
/* TEST HOOKS, per context (ctx == NULL: the calling thread's current context; nothing here
   is process-wide).  theta_margin: the distance from an integer inside which
   OD_THETA_SCALE*acos(corr) + .5 is not trusted to the device acos (default 1e-9; <= 0
   restores it); theta_perturb != 0: the device theta of a listed band is deliberately wrong
   (+1), so that tests exercise the host-libm re-run on real data; price_tol_scale multiplies
   the margin inside which a priced choice is left to the host libm (<= 0: 1).  A production
   caller never touches these. */
int odhip_ctx_set_test_hooks(odhip_ctx *ctx, double theta_margin, int theta_perturb,
 double price_tol_scale);
int odhip_image_planes_copy_pad16(uint16_t *d_dst, int dst_stride, long dst_plane_stride, int plane_w,
 int plane_h, const void *d_src, int src_bitdepth, int src_stride, long src_plane_stride, int pic_w,
 int pic_h, int nplanes, odhip_stream stream);

/* ---- frame cache: one batched pyramid serving every per-block fdct_2d call ---

   The samples a block of level bs sees in the reference encoder depend only on
   the source plane and on "every ancestor was split" (SURVEY.md section 7), so
   one odhip_forward_pyramid per plane yields the output of every fdct_2d call
   the encoder will make on that plane, in both RDO passes.

   odhip_cache_load_plane is called when the encoder has just converted plane
   `pli` to coefficients and is about to lap it across superblock edges
   (od_apply_prefilter_frame_sbs, src/encode.c:2571): `coef` must still hold
   (p - 128) << 4 (src/state.c:1233), stride == w.  odhip_cache_lookup serves an
   fdct_2d(out, out_stride, in, in_stride) whose `in` lies inside a loaded plane
   (returns 1) or reports a miss (returns 0).  odhip_install_cached_dct_vtbl
   installs fdct_2d entries that try the calling thread's current cache first
   and fall back to the per-call GPU transform (idct_2d entries are the per-call
   ones).  ODHIP_CACHE_CHECK=1 verifies every hit against the per-call path. */
typedef struct odhip_frame_cache odhip_frame_cache;
odhip_frame_cache *odhip_cache_create(void);
void odhip_cache_destroy(odhip_frame_cache *c);
void odhip_cache_set_picture(odhip_frame_cache *c, int pic_w, int pic_h);
void odhip_cache_make_current(odhip_frame_cache *c);
/* The 8-bit source samples of a loaded plane (host and device copies; valid until the slot is
   loaded again): the input of odhip_dering_cache_set_source. */
int odhip_cache_plane_pixels(const odhip_frame_cache *c, int pli, const uint8_t **h_px, const uint8_t **d_px,
 int *w, int *h);
int odhip_cache_load_plane(odhip_frame_cache *c, int pli, const od_coeff *coef,
 int stride, int w, int h, int dec);
int odhip_cache_lookup(odhip_frame_cache *c, const od_coeff *in, int in_stride,
 int bs, od_coeff *out, int out_stride);
void odhip_cache_stats(const odhip_frame_cache *c, long *hits, long *misses);
void odhip_install_cached_dct_vtbl(odhip_dct_func_2d fdct_2d[ODHIP_NBSIZES],
 odhip_dct_func_2d idct_2d[ODHIP_NBSIZES]);

/* ---- pricing on the device: od_pvq_rate's closed form (the
   speed > 0 half) --------------------------------------------------------------------

   od_pvq_rate (src/pvq_encoder.c:247-287) has two modes.  speed == 0 (the default
   complexity) runs the real codeword coder on a copy of the live adaptive context:
   sequential host state, not batchable.  speed > 0 - what the
   block-size RDO pass prices with below complexity 5, src/encode.c:1359 - is a closed
   form in (K, n, the centre of mass of the pulse vector, theta terms): a pure function of
   the candidate.  odhip_pvq_choose_priced_multi / odhip_pvq_ref_choose_priced_multi are
   odhip_pvq_choose_multi / odhip_pvq_ref_choose_multi with that rate evaluated ON THE
   DEVICE from the pulses the searches stored - no rate table, no candidate export.

   Its two libm calls (log) are not reproducible bit for bit on the device; the costs
   they enter are compared with each other, so a decision is taken from the device only
   when the two costs differ by more than 1e-10 of their magnitude (the two logs agree to
   ~1e-16).  A band with a closer decision is listed; *_priced_resolve (same jobs, same
   stream, after the *_priced_multi call) waits for the count of listed bands only -
   normally 0 - and otherwise recomputes those bands' rates with the HOST libm, the
   function the reference calls, and decides them again.  It returns the number of
   bands re-decided; consumers of the choice records enqueued before it must then be
   repeated.  (odhip_ctx_set_test_hooks can widen the margin to force this path.) */
int odhip_pvq_choose_priced_multi(const odhip_pvq_job *jobs, int njobs, double pvq_norm_lambda,
 odhip_stream stream);
int odhip_pvq_choose_priced_resolve(const odhip_pvq_job *jobs, int njobs, double pvq_norm_lambda,
 odhip_stream stream);
/* odhip_pvq_noref_bands_multi AND odhip_pvq_choose_priced_multi in one pass: every band is
   decided where its search ends, from the values held in registers; no choice kernel reads the
   records back.  What leaves the stage is what its consumers read: the choice record and the
   CHOSEN candidate's pulses (slot = choice[0]).  Every band is prepared, searched and
   decided by the lane (or lane pair) that loads its coefficients, in natural block order: no
   scaled vector, band record, sort key or sorted index exists in memory for it.  A losing
   candidate's pulses and the band record (gains, pulse counts, sums, distortions, moments) are
   written only for a band listed as a close call - the resolve decides it again from exactly
   those.  A host that wants both candidates of every band uses
   odhip_pvq_noref_bands_multi.  Follow with odhip_pvq_choose_priced_resolve. */
int odhip_pvq_noref_bands_priced_multi(const odhip_pvq_job *jobs, int njobs, double pvq_norm_lambda,
 odhip_stream stream);
int odhip_pvq_ref_choose_priced_multi(const odhip_pvq_refjob *jobs, int njobs, double pvq_norm_lambda,
 odhip_stream stream);
int odhip_pvq_ref_choose_priced_resolve(const odhip_pvq_refjob *jobs, int njobs, double pvq_norm_lambda,
 odhip_stream stream);
/* The same split as odhip_pvq_noref_bands_priced_multi: the band stage with the priced choice
   of the bands searched one per lane (15 and 8 coefficients: 77 % of the bands of a 4:2:0
   chroma plane) made INSIDE their search kernels, from the registers; then the choice
   kernels for the rest (32 and 128 coefficients) and the count of listed bands.  The
   candidate records are written as always (the theta-margin re-run and the price-margin
   resolve read them; after a theta-margin re-run use odhip_pvq_ref_choose_priced_multi, which
   decides every band from the records). */
int odhip_pvq_ref_bands_priced_multi(const odhip_pvq_refjob *jobs, int njobs, double pvq_norm_lambda,
 odhip_stream stream);
int odhip_pvq_ref_choose_priced_rest_multi(const odhip_pvq_refjob *jobs, int njobs, double pvq_norm_lambda,
 odhip_stream stream);

/* ---- od_pvq_rate at the default complexity (speed == 0), batched on the host ----------

   At speed == 0 od_pvq_rate (src/pvq_encoder.c:247-287) prices a candidate by running
   od_encode_pvq_codeword on a scratch range coder against a COPY of the live adaptive
   od_pvq_codeword_ctx (:265-275): the price depends on every symbol coded before, which is
   sequential host state (SURVEY hard part 1) - but not on most of what that call moves.
   odhip_pvq_rate_batch prices all `ncand` candidates of one band against one snapshot of
   the live context with a rate-only range coder (range + bit count: all od_ec_enc_tell_frac
   reads) and copy-on-touch CDF rows: no allocation, no context copy, no output bytes; the
   doubles are the reference's bit for bit.  ctx: the live context itself (the layout below
   is the reference's struct, src/pvq.h:125-132; &adapt->pvq.pvq_codeword_ctx sits at offset
   0 of od_adapt_ctx); it is only read.  Candidate c: pulses y[c][0 .. n - (theta[c] != -1))
   (NULL allowed when k[c] == 0), k[c], gain index qg[c], theta[c] (-1: no reference),
   ts[c] = max_theta; n = band size, icgr, is_keyframe, pli as od_pvq_rate takes them.
   _batch16: the pulses as int16 (what the band stages export). */
typedef struct {
  int32_t pvq_adapt[2*ODHIP_NBSIZES*4];
  int32_t pvq_k1_increment;
  uint16_t pvq_k1_cdf[12][16];
  uint16_t pvq_split_cdf[14*7][8];
  int32_t pvq_split_increment;
} odhip_pvq_codeword_ctx;
int odhip_pvq_rate_batch(double *rate, const odhip_pvq_codeword_ctx *ctx, int ncand,
 const od_coeff *const *y, const int *k, const int *qg, const int *theta, const int *ts, int n, int icgr,
 int is_keyframe, int pli);
int odhip_pvq_rate_batch16(double *rate, const odhip_pvq_codeword_ctx *ctx, int ncand,
 const int16_t *const *y, const int *k, const int *qg, const int *theta, const int *ts, int n, int icgr,
 int is_keyframe, int pli);

/* ---- frame cache, second half: the batched band stage behind pvq_theta ------------

   odhip_cache_load_bands (after odhip_cache_load_plane(pli), same frame): the
   no-reference PVQ band stage of every block of every level of that plane in one
   set of launches on the pyramid still resident in HBM, with the quantiser set-up
   `qt` (the encoder's state.qm / pvq_qm_q4 / quantizer: odhip_quant_setup, or the
   caller copies its own tables into an odhip_quant); records and pulse vectors go
   to pinned host memory.

   odhip_cache_band serves ONE pvq_theta call of the encoder's block loop
   (src/pvq_encoder.c:333-641) whose reference vector r0 is null - the case in which
   pvq_theta runs exactly the no-reference candidates (:452 fails, :571-609): block
   (bx, by) of level bs (block units), band index `band`.  Returns 1 and fills *out
   (views into the cache, valid until the next load), or 0 (not loaded / out of
   range: the caller falls back to the reference's own pvq_theta).  x0 is optional:
   with ODHIP_CACHE_CHECK=1 the band the encoder presents is compared with the band
   the batch coded and a mismatch aborts.  The caller then does what stays on the
   host in the reference: per candidate s with flags[s] == 1, cost = dist[s] +
   lambda*od_pvq_rate(gain[s], 0, -1, 0, adapt, y[s], k[s], n, ...), `cost <=
   best_cost` starting from best_cost = dist0 (:417-421, :597-609), then skip rule,
   od_gain_expand and od_pvq_synthesis_partial (:611-633).  INTEGRATION.md section 7
   shows that glue; tests/interpose/interpose.c is its load-time form. */
typedef struct {
  int n;                 /* coefficients of the band                               */
  int32_t q;             /* the band's quantiser step and beta (Q12) the batch used:   */
  int32_t beta;          /* the caller checks them against its own pvq_theta arguments */
  int32_t cg;            /* companded gain of x, Q8                                */
  int32_t gain[2];       /* candidate gain index i (0 = slot unused)               */
  int32_t k[2];
  int32_t flags[2];      /* 1 = searched, 0 = pruned / unused, 2 = K out of range  */
  double dist0;          /* distortion of the null candidate                       */
  double dist[2];
  const int16_t *y[2];   /* signed pulses of the candidate, coding order, n values */
} odhip_band_cands;
int odhip_cache_load_bands(odhip_frame_cache *c, int pli, const odhip_quant *qt,
 double pvq_norm_lambda);
int odhip_cache_band(odhip_frame_cache *c, int pli, int bs, int bx, int by, int band,
 const od_coeff *x0, odhip_band_cands *out);
void odhip_cache_band_stats(const odhip_frame_cache *c, long *hits, long *misses);

/* ---- YUV4MPEG2 input (host; SURVEY.md 8(f) rank 4, input side) ---------------------

   The reference reads its input in examples/encoder_example.c:89-160, :190-400,
   :449-508.  odhip_y4m_open parses the stream header (W, H, F, I, C tags) of a
   progressive 8-bit 4:2:0 file (C420, C420jpeg, C420mpeg2, C420paldv or no C tag);
   anything else - 4:4:4, 4:2:2, 4:1:1, mono, 10..16-bit, interlaced - returns NULL with
   *err = ODHIP_EIMPL (ODHIP_EINVAL: unreadable / not a YUV4MPEG2 stream).
   odhip_y4m_read reads one FRAME into tightly packed planes (luma w x h, chroma
   ((w + 1) >> 1) x ((h + 1) >> 1): what odhip_pipe_set_pictures takes, one picture
   at a time); returns 1, 0 at the end of the stream, a negative code on loss of
   framing or a short read. */
typedef struct odhip_y4m odhip_y4m;
odhip_y4m *odhip_y4m_open(const char *path, int *pic_w, int *pic_h, int *fps_n, int *fps_d, int *err);
int odhip_y4m_read(odhip_y4m *y, uint8_t *luma, uint8_t *cb, uint8_t *cr);
/* Steps over one FRAME without reading it (frame-sharded input: a rank reads only the
   frames it owns); 1, 0 at the end of the stream, negative on loss of framing. */
int odhip_y4m_skip(odhip_y4m *y);
void odhip_y4m_close(odhip_y4m *y);

/* ---- odhip_pipe: the frame-batch step as one C call ------------------------------

   One step = one pass of the hot path over `frames` resident 4:2:0 pictures of
   pic_w x pic_h (coded size = both rounded up to 64, src/state.c:376-379): input
   padding, forward pyramid (every level), PVQ band stage of every block of every
   level (luma: no reference; chroma: WITH the chroma-from-luma reference produced
   inside the step from the luma choices when chroma_cfl != 0, as the reference codes
   keyframe chroma, src/encode.c:1680-1687; without reference otherwise), choice,
   dequantisation + inverse of every level.  The luma and the chroma chain run on the
   pipe's two streams, each in its own odhip_ctx, software-pipelined over steps (the
   luma chain of step i+1 overlaps the chroma chain of step i); `serial` (or
   ODHIP_PVQ_SERIAL=1) puts everything on one stream.  The pipe owns all device
   memory and both streams.  The quantiser tables are copied at creation.

   odhip_pipe_step enqueues one step and returns (asynchronous); odhip_pipe_flush
   settles the last step's device-acos margin check (see odhip_pvq_ref_resolve_*);
   odhip_pipe_sync waits for both streams.  odhip_pipe_stage runs ONE stage in order
   on the first stream (tests; the flow in which a host prices the candidates between
   the band stage and the choice); odhip_pipe_buffer hands out the device buffers
   (ODHIP_PIPE_BUF_RATE allocates the rate table of a (plane set, level) on first
   use, zero-filled, and from then on the choice of that level is
   cost = dist + lambda*rate: [B][nb][2] doubles without reference,
   [B][nb][ODHIP_PVQ_REF_SLOTS + 1] with - the d_rate layouts above);
   odhip_pipe_read / _write copy after odhip_pipe_sync.  odhip_pipe_record brackets
   every stage (and the dominant kernel of each band stage) with HIP events on the
   stream it is launched on; odhip_pipe_timings returns the average milliseconds and
   the count per stage, odhip_pipe_search_timings the bracketed search kernels
   (chroma = 0: k_search<128,2,1>; 1: k_refb_search_row<8,16>). */
typedef struct odhip_pipe odhip_pipe;
typedef struct {
  int device;
  int frames;
  int pic_w;
  int pic_h;
  int chroma_cfl;
  int serial;
  int price;                    /* 1: the choice prices every candidate with od_pvq_rate's
                                   closed form on the device (odhip_pvq_*choose_priced_*);
                                   0: on distortion alone, or with rate tables             */
  int fpr_bits;                 /* 0: 8-bit picture buffers.  8 / 10 / 12: full-precision
                                   references - the coded planes and the reconstructions are
                                   int16 samples at 12 bits (odhip_ctx_set_fpr on both
                                   contexts), the resident pictures are uint8_t (8) or int16
                                   (10, 12) samples of that depth */
  int inter;                    /* 1: an INTER frame - every plane goes through the with-reference
                                   stage (pvq_theta with is_keyframe = 0) against the pyramid of
                                   its prediction picture (odhip_pipe_set_reference_pictures: what
                                   motion compensation produced; the encoder's mctmp / mdtmp,
                                   src/encode.c:880-886, :1326-1360); chroma_cfl is ignored */
  int reserved;
  double pvq_norm_lambda;       /* OD_PVQ_LAMBDA, src/pvq.h:49 */
  const odhip_quant *quant;
} odhip_pipe_config;
enum {
  ODHIP_PIPE_PAD_LUMA = 0, ODHIP_PIPE_PYRAMID_LUMA, ODHIP_PIPE_BANDS_LUMA, ODHIP_PIPE_CHOOSE_LUMA,
  ODHIP_PIPE_CFL_REFS, ODHIP_PIPE_INVERSE_LUMA, ODHIP_PIPE_PAD_CHROMA, ODHIP_PIPE_PYRAMID_CHROMA,
  ODHIP_PIPE_BANDS_CHROMA, ODHIP_PIPE_CHOOSE_CHROMA, ODHIP_PIPE_INVERSE_CHROMA, ODHIP_PIPE_NSTAGES
};
enum {
  ODHIP_PIPE_BUF_PIC = 0,   /* pictures: luma [F][pic_h][pic_w], chroma [2F][pic_h/2][pic_w/2]
                               (all Cb, then all Cr)                                   */
  ODHIP_PIPE_BUF_PX,        /* padded planes                                           */
  ODHIP_PIPE_BUF_LEVEL,     /* coefficient planes of a pyramid level                   */
  ODHIP_PIPE_BUF_RECON,     /* reconstructed pixels of a partition level               */
  ODHIP_PIPE_BUF_BAND,      /* odhip_pvq_band / odhip_pvq_refband records              */
  ODHIP_PIPE_BUF_Y,         /* pulse vectors                                           */
  ODHIP_PIPE_BUF_CHOICE,    /* choice records                                          */
  ODHIP_PIPE_BUF_ITEMS,     /* with-reference candidates (3 planes of 16-byte vectors) */
  ODHIP_PIPE_BUF_REF,       /* inter mode: the prediction pyramid.  (Keyframe chroma takes
                               its chroma-from-luma reference in place from the luma
                               choices, odhip_pvq_refjob.luma: no plane exists)        */
  ODHIP_PIPE_BUF_RATE       /* rate table (allocated on first request)                 */
};
odhip_pipe *odhip_pipe_create(const odhip_pipe_config *cfg);
void odhip_pipe_destroy(odhip_pipe *p);
int odhip_pipe_set_pictures(odhip_pipe *p, const uint8_t *luma, const uint8_t *chroma, int on_device);
/* A stream of pictures instead of resident ones: odhip_pipe_feed copies the pictures of
   the NEXT step from host memory (same layouts as odhip_pipe_set_pictures) into the pipe's
   back buffers on its own copy stream - behind the padding kernels that may still read
   them, beside the steps already enqueued - and the next odhip_pipe_step codes them.
   Asynchronous for pinned host memory; the host buffers must stay untouched until
   odhip_pipe_sync (or any later odhip_pipe_feed of the same pipe followed by a sync).
     for (;;) { odhip_pipe_feed(p, next_luma, next_chroma); odhip_pipe_step(p); ... } */
int odhip_pipe_feed(odhip_pipe *p, const uint8_t *luma, const uint8_t *chroma);
/* ---- the decisions of a step in the form a host entropy coder reads them (export_kernels.hip) ----
   What od_pvq_encode hands to its entropy coder per band (src/pvq_encoder.c:874-979, the arguments of
   pvq_encode_partition): the coded gain index, theta and its range, K, the skip / no-reference flags and
   the pulse vector - K pulses over n positions, nearly all zero.  One SECTION per (plane set, level):
     records     odhip_export_record4 / odhip_export_record8 [blocks][bands] (record_bytes of the section)
     stream      uint16 words.  The words of a band are consecutive; the bands of one GROUP (the records
                 of blocks_per_group consecutive blocks: 128 / 32 / 8 / 4 / 4 for 4x4 ... 64x64) follow each
                 other in record order; group g starts at word group_base[g] of the section's stream (groups
                 are placed in the order their workgroups finish: placement varies from run to run, content
                 does not).
     word        bits 0-6 the position inside the band, bits 7-15 the signed pulse count (-255 .. 255); a
                 count of -256 is an escape - the next word holds the count as an int16.
   A band without words holds no pulse (skipped, null gain, or K = 0).  Header first: words written per
   section, and a flag per section that is set when a stream outgrew its capacity (as many words as the
   level has coefficients; never seen - the dense vectors remain readable through odhip_pipe_read). */
#define ODHIP_EXPORT_MAX_SECTIONS 16
/* `fn` of a record: bits 0-8 the words of this band in the stream (0 .. 256), bit 9 coded without reference
   (pvq_theta's noref), bits 10-11 skip (0, OD_PVQ_SKIP_ZERO 1, OD_PVQ_SKIP_COPY 2).  K is not exported: for a
   coded band it is the sum of the magnitudes of its words' counts. */
#define ODHIP_EXPORT_NWORDS(fn) ((fn) & 0x1ff)
#define ODHIP_EXPORT_NOREF(fn) ((fn) >> 9 & 1)
#define ODHIP_EXPORT_SKIP(fn) ((fn) >> 10 & 3)
typedef struct {
  int16_t qg;          /* the coded gain index; 0 = the null vector */
  uint16_t fn;
} odhip_export_record4;   /* sections coded without a reference (keyframe luma): 4 bytes */
typedef struct {
  int16_t qg;          /* the coded gain index: pvq_theta's return value (:636-637) */
  int16_t itheta;
  int16_t max_theta;
  uint16_t fn;
} odhip_export_record8;   /* sections coded against a reference: 8 bytes */
typedef struct {
  uint32_t total_words[ODHIP_EXPORT_MAX_SECTIONS];
  uint32_t overflow[ODHIP_EXPORT_MAX_SECTIONS];
} odhip_export_header;
typedef struct {
  int32_t bs;
  uint32_t ngroups;
  uint32_t cap_words;
  uint32_t blocks_per_group;
  uint32_t record_bytes;     /* 4: odhip_export_record4, 8: odhip_export_record8 */
  uint32_t pad;
  uint64_t nrecords;         /* blocks x bands */
  uint64_t records_off;      /* byte offsets from the start of the buffer, multiples of 16 */
  uint64_t group_base_off;   /* uint32 [ngroups] */
  uint64_t stream_off;
} odhip_export_section;
typedef struct {
  int32_t nsections;
  int32_t pad;
  uint64_t fixed_bytes;      /* header + records + group bases of every section (always shipped) */
  uint64_t total_bytes;      /* the whole buffer: fixed part + the streams at full capacity */
  odhip_export_section section[ODHIP_EXPORT_MAX_SECTIONS];
} odhip_export_layout;
int odhip_export_layout_make(odhip_export_layout *lay, int nsections, const long *nblocks, const int *bs,
 const int *with_ref);
int odhip_export_begin(void *d_buf, const odhip_export_layout *lay, odhip_stream stream);
int odhip_export_pack(void *d_buf, const odhip_export_layout *lay, int section, const int32_t *d_choice,
 const int16_t *d_y, long nblocks, int bs, int with_ref, odhip_stream stream);
/* sections first_section .. first_section + n - 1 in one launch per five (arrays of n device pointers / sizes) */
int odhip_export_pack_multi(void *d_buf, const odhip_export_layout *lay, int first_section, int n,
 const int32_t *const *d_choice, const int16_t *const *d_y, const long *nblocks, const int *bs, int with_ref,
 odhip_stream stream);
int odhip_export_ship(void *pinned_host, const void *d_buf, const odhip_export_layout *lay, odhip_stream stream);

/* The OUTPUT side of a streaming host (SURVEY hard part 5: "timed GPU-side incl. transfers"): with a
   host buffer set, every following step leaves what a host entropy coder consumes - the record and the
   pulses of every band of every level, compacted as above (sections: luma levels 0..4, then chroma
   levels 0..3) - in pinned host memory, packed and shipped on a third stream behind the stage that
   produced it, overlapped with the rest of the step.  odhip_pipe_export_layout: the offsets;
   odhip_pipe_export_bytes: the size the host buffer must have (= total_bytes; only the used part of
   every stream crosses the bus; 0: this pipe's mode does not export - keyframe steps with chroma from
   luma and device pricing only).  The buffer holds step i after the odhip_pipe_sync that follows it.
   A band that the host-libm resolve re-decides one step late (none on any content measured) is shipped
   again when that happens inside odhip_pipe_flush - step, flush, sync, read is exact - and counted by
   odhip_pipe_export_stale when it happens inside the next step, i.e. after the host read the buffer. */
size_t odhip_pipe_export_bytes(const odhip_pipe *p);
int odhip_pipe_export_layout(const odhip_pipe *p, odhip_export_layout *out);
long odhip_pipe_export_stale(const odhip_pipe *p);
int odhip_pipe_set_export(odhip_pipe *p, void *pinned_host);
/* Inter mode (odhip_pipe_config.inter): the prediction pictures of the batch, same layouts
   and depth as odhip_pipe_set_pictures. */
int odhip_pipe_set_reference_pictures(odhip_pipe *p, const uint8_t *luma, const uint8_t *chroma,
 int on_device);
int odhip_pipe_step(odhip_pipe *p);
int odhip_pipe_flush(odhip_pipe *p);
int odhip_pipe_sync(odhip_pipe *p);       /* ODHIP_ERANGE: see ODHIP_PVQ_MAX_K - the steps since the last sync coded
                                             a band otherwise than the reference would */
long odhip_pipe_k_range(const odhip_pipe *p);      /* such bands over this pipe's syncs so far */
int odhip_pipe_stage(odhip_pipe *p, int stage, int parity);
/* parity: with chroma from luma the luma pulse vectors / choice records alternate between
   two sets from step to step (the chroma chain of step i reads them while the luma chain of
   step i + 1 writes the next); 0 / 1 selects one, -1 the set of the last odhip_pipe_step. */
int odhip_pipe_buffer(odhip_pipe *p, int what, int set, int level, int parity, void **d_ptr,
 size_t *bytes);
int odhip_pipe_read(odhip_pipe *p, void *host, const void *d_ptr, size_t bytes);
int odhip_pipe_write(odhip_pipe *p, void *d_ptr, const void *host, size_t bytes);
int odhip_pipe_record(odhip_pipe *p, int enable);
int odhip_pipe_timings(odhip_pipe *p, double avg_ms[ODHIP_PIPE_NSTAGES], int count[ODHIP_PIPE_NSTAGES]);
int odhip_pipe_search_timings(odhip_pipe *p, int chroma, float *ms, int max_n);
int odhip_pipe_time_pyramid(odhip_pipe *p, int n, double *avg_ms);
/* The practical HBM ceiling beside the 8 TB/s specification: a device-to-device copy of
   `bytes` (a multiple of 16; scratch owned by the call) in 16-byte vectors, n timed launches after a
   warm-up, (read + written) bytes per second in GB/s.  variant 0 / 1 / 2: 2 / 4 / 8 vectors in flight
   per lane. */
int odhip_copy_ceiling(size_t bytes, int n, int variant, double *gbs, odhip_stream stream);
/* One stage of the filter + DCT path (ODHIP_PIPE_PAD_*, _PYRAMID_*, _INVERSE_*) launched n times on
   the idle GPU over the buffers the last step left: average milliseconds per launch group (the
   roofline_* entries of bench.py).  parity as for odhip_pipe_stage, -1 = the last step's. */
int odhip_pipe_time_stage(odhip_pipe *p, int stage, int parity, int n, double *avg_ms);
long odhip_pipe_theta_reruns(const odhip_pipe *p);
/* ... and the bands that were listed (inside the acos margin, theta recomputed by the host), changed or not */
long odhip_pipe_theta_listed(odhip_pipe *p);
long odhip_pvq_ref_theta_listed(void);
long odhip_pipe_price_reruns(const odhip_pipe *p);
double odhip_pipe_host_wait_ms(const odhip_pipe *p);
/* odhip_ctx_set_test_hooks on both contexts of the pipe. */
int odhip_pipe_set_test_hooks(odhip_pipe *p, double theta_margin, int theta_perturb, double price_tol_scale);

#ifdef __cplusplus
}
#endif
#endif
