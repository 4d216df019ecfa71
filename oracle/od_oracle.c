/* oracle/od_oracle.c - CPU restatement of the Daala block-transform hot path.

   TEST INFRASTRUCTURE - see od_oracle.h.  Plain C, no dependency on the
   reference tree at build or run time.  Integer semantics (arithmetic >>,
   truncating /, 16/32/64-bit intermediate widths, int16 truncation on store)
   are spelled out because they are the specification.  Compiled with
   -ffp-contract=off: the reference's x86-64 -O2 build emits no FMA. */
#include "od_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "od_lifting_tables.h"
#include "od_scan_tables.h"

/* ======================================================================== */
/* Transforms                                                                */
/* ======================================================================== */

static const od_lift_op *const ODO_FWD[ODO_NBSIZES] = {
  OD_LIFT_FDCT4, OD_LIFT_FDCT8, OD_LIFT_FDCT16, OD_LIFT_FDCT32, OD_LIFT_FDCT64
};
static const int ODO_FWD_NOPS[ODO_NBSIZES] = {
  OD_LIFT_FDCT4_NOPS, OD_LIFT_FDCT8_NOPS, OD_LIFT_FDCT16_NOPS,
  OD_LIFT_FDCT32_NOPS, OD_LIFT_FDCT64_NOPS
};
static const od_lift_op *const ODO_INV[ODO_NBSIZES] = {
  OD_LIFT_IDCT4, OD_LIFT_IDCT8, OD_LIFT_IDCT16, OD_LIFT_IDCT32, OD_LIFT_IDCT64
};
static const int ODO_INV_NOPS[ODO_NBSIZES] = {
  OD_LIFT_IDCT4_NOPS, OD_LIFT_IDCT8_NOPS, OD_LIFT_IDCT16_NOPS,
  OD_LIFT_IDCT32_NOPS, OD_LIFT_IDCT64_NOPS
};

/* OD_DCT_RSHIFT(a, 1), src/filter.h:38-41: a/2 rounded toward zero. */
static int32_t odo_rs1(int32_t a) {
  return (int32_t)(((uint32_t)a >> 31) + (uint32_t)a) >> 1;
}

/* Table-driven execution of one lifting network (the networks restate
   od_bin_fdct4..64 / od_bin_idct4..64, src/dct.c:87-790,4219-4820; see
   tools/extract_lifting.py).  in/out are accessed with their own strides so
   the same interpreter serves the forward (strided in) and inverse (strided
   out) conventions of src/dct.h:61-68. */
static void odo_lift_run(const od_lift_op *ops, int nops, odo_coeff *out,
 int out_stride, const odo_coeff *in, int in_stride) {
  int32_t r[640];
  int i;
  for (i = 0; i < nops; i++) {
    const od_lift_op *o;
    o = ops + i;
    switch (o->op) {
      case 0: r[o->d] = in[o->c*in_stride]; break;
      case 1: out[o->c*out_stride] = r[o->a]; break;
      case 2: r[o->d] = r[o->a] + r[o->b]; break;
      case 3: r[o->d] = r[o->a] - r[o->b]; break;
      case 4: r[o->d] = odo_rs1(r[o->a]); break;
      case 5: r[o->d] = (r[o->a]*o->c + o->r) >> o->s; break;
      case 6: r[o->d] = -r[o->a]; break;
      case 7: r[o->d] = r[o->a] >> o->s; break;
      case 8: r[o->d] = r[o->a]; break;
      default: abort();
    }
  }
}

void odo_fdct_1d(int ln, odo_coeff *y, const odo_coeff *x, int xstride) {
  odo_lift_run(ODO_FWD[ln], ODO_FWD_NOPS[ln], y, 1, x, xstride);
}

void odo_idct_1d(int ln, odo_coeff *x, int xstride, const odo_coeff *y) {
  odo_lift_run(ODO_INV[ln], ODO_INV_NOPS[ln], x, xstride, y, 1);
}

/* od_bin_fdctNxN, src/dct.c:151-156,351-356,792-798,4890-4904: pass 1
   transforms column i of x into row i of z, pass 2 column i of z into row i
   of y. */
void odo_fdct_2d(int ln, odo_coeff *y, int ystride, const odo_coeff *x,
 int xstride) {
  odo_coeff z[64*64];
  int n;
  int i;
  n = 4 << ln;
  for (i = 0; i < n; i++) odo_fdct_1d(ln, z + n*i, x + i, xstride);
  for (i = 0; i < n; i++) odo_fdct_1d(ln, y + ystride*i, z + i, n);
}

/* od_bin_idctNxN, src/dct.c:158-163,358-363,800-806,4906-4920: rows of y
   into columns of z, rows of z into columns of x. */
void odo_idct_2d(int ln, odo_coeff *x, int xstride, const odo_coeff *y,
 int ystride) {
  odo_coeff z[64*64];
  int n;
  int i;
  n = 4 << ln;
  for (i = 0; i < n; i++) odo_idct_1d(ln, z + i, n, y + ystride*i);
  for (i = 0; i < n; i++) odo_idct_1d(ln, x + i, xstride, z + n*i);
}

void odo_fdct_2d_batch(int ln, odo_coeff *y, const odo_coeff *x,
 long nblocks) {
  long b;
  int n;
  n = 4 << ln;
  for (b = 0; b < nblocks; b++) odo_fdct_2d(ln, y + b*n*n, n, x + b*n*n, n);
}

void odo_idct_2d_batch(int ln, odo_coeff *x, const odo_coeff *y,
 long nblocks) {
  long b;
  int n;
  n = 4 << ln;
  for (b = 0; b < nblocks; b++) odo_idct_2d(ln, x + b*n*n, n, y + b*n*n, n);
}

/* ======================================================================== */
/* 4-point lapping filter                                                    */
/* ======================================================================== */

/* od_pre_filter4, src/filter.c:147-193 with OD_FILTER_PARAMS4 = {85, 75, -15,
   33} (:137-140).  "+1 if positive" after each scaling makes the scaling
   exactly invertible. */
void odo_pre_filter4(odo_coeff y[4], const odo_coeff x[4]) {
  int32_t d30;
  int32_t d21;
  int32_t s1;
  int32_t s0;
  d30 = x[0] - x[3];
  d21 = x[1] - x[2];
  s1 = x[1] - (d21 >> 1);
  s0 = x[0] - (d30 >> 1);
  d21 = d21*85 >> 6;
  d21 += d21 > 0;
  d30 = d30*75 >> 6;
  d30 += d30 > 0;
  d30 += (d21*-15 + 32) >> 6;
  d21 += (d30*33 + 32) >> 6;
  s0 += d30 >> 1;
  s1 += d21 >> 1;
  y[0] = s0;
  y[1] = s1;
  y[2] = s1 - d21;
  y[3] = s0 - d30;
}

/* od_post_filter4, src/filter.c:195-222: exact inverse; the two divisions are
   C truncating divisions (:210,:213). */
void odo_post_filter4(odo_coeff x[4], const odo_coeff y[4]) {
  int32_t d30;
  int32_t d21;
  int32_t s1;
  int32_t s0;
  d30 = y[0] - y[3];
  d21 = y[1] - y[2];
  s1 = y[1] - (d21 >> 1);
  s0 = y[0] - (d30 >> 1);
  d21 -= (d30*33 + 32) >> 6;
  d30 -= (d21*-15 + 32) >> 6;
  d30 = d30*64/75;
  d21 = d21*64/85;
  s0 += d30 >> 1;
  s1 += d21 >> 1;
  x[0] = s0;
  x[1] = s1;
  x[2] = s1 - d21;
  x[3] = s0 - d30;
}

/* od_pre_filter8/16/32, src/filter.c:279-364, :519-676, :852-1144 (the TYPE3
   variants the `#elif 1` chains select), stated once for n = 2h taps: +1/-1
   butterflies, Q6 scaling of the high half with the "+1 if positive" that makes
   it invertible (skipped for a factor of 64), the rotation ladder from the top
   pair down, butterflies back.  n = 4 gives od_pre_filter4. */
#include "od_filter_params.h"

static const int *odo_filter_params(int n) {
  return n == 4 ? OD_FPARAMS4 : n == 8 ? OD_FPARAMS8 : n == 16 ? OD_FPARAMS16
   : n == 32 ? OD_FPARAMS32 : NULL;
}

int odo_pre_filter(int n, odo_coeff *y, const odo_coeff *x) {
  const int *p;
  int32_t t[32];
  int h;
  int i;
  int k;
  p = odo_filter_params(n);
  if (!p) return -1;
  h = n >> 1;
  for (i = 0; i < h; i++) t[n - 1 - i] = x[i] - x[n - 1 - i];
  for (i = 0; i < h; i++) t[i] = x[i] - (t[n - 1 - i] >> 1);
  for (i = 0; i < h; i++) {
    if (p[i] != 64) {
      t[h + i] = t[h + i]*p[i] >> 6;
      t[h + i] += t[h + i] > 0;
    }
  }
  for (k = h - 2; k >= 0; k--) {
    t[h + k + 1] += (t[h + k]*p[h + k] + 32) >> 6;
    t[h + k] += (t[h + k + 1]*p[2*h - 1 + k] + 32) >> 6;
  }
  for (i = 0; i < h; i++) {
    t[i] += t[n - 1 - i] >> 1;
    y[i] = t[i];
  }
  for (i = 0; i < h; i++) y[n - 1 - i] = t[i] - t[n - 1 - i];
  return 0;
}

/* od_post_filter8/16/32, src/filter.c:366-421, :678-783, :1146-1321: the exact
   inverse; the scalings are undone with C truncating divisions. */
int odo_post_filter(int n, odo_coeff *x, const odo_coeff *y) {
  const int *p;
  int32_t t[32];
  int h;
  int i;
  int k;
  p = odo_filter_params(n);
  if (!p) return -1;
  h = n >> 1;
  for (i = 0; i < h; i++) t[n - 1 - i] = y[i] - y[n - 1 - i];
  for (i = 0; i < h; i++) t[i] = y[i] - (t[n - 1 - i] >> 1);
  for (k = 0; k <= h - 2; k++) {
    t[h + k] -= (t[h + k + 1]*p[2*h - 1 + k] + 32) >> 6;
    t[h + k + 1] -= (t[h + k]*p[h + k] + 32) >> 6;
  }
  for (i = 0; i < h; i++) {
    if (p[i] != 64) t[h + i] = t[h + i]*64/p[i];
  }
  for (i = 0; i < h; i++) {
    t[i] += t[n - 1 - i] >> 1;
    x[i] = t[i];
  }
  for (i = 0; i < h; i++) x[n - 1 - i] = t[i] - t[n - 1 - i];
  return 0;
}

static void odo_filter4_col(odo_coeff *c, int stride, int inverse) {
  odo_coeff t[4];
  int k;
  for (k = 0; k < 4; k++) t[k] = c[stride*k];
  if (inverse) odo_post_filter4(t, t);
  else odo_pre_filter4(t, t);
  for (k = 0; k < 4; k++) c[stride*k] = t[k];
}

static void odo_filter4_row(odo_coeff *c, int inverse) {
  if (inverse) odo_post_filter4(c, c);
  else odo_pre_filter4(c, c);
}

/* od_prefilter_split, src/filter.c:1459-1483 with f = OD_FILT_SIZE == 0
   (src/filter.h:77): `hfilter` gates the column-direction taps across the
   horizontal mid-line, `vfilter` the row-direction taps across the vertical
   mid-line; columns first. */
void odo_prefilter_split(odo_coeff *c0, int stride, int bs, int hfilter,
 int vfilter) {
  int n;
  int i;
  n = 4 << bs;
  if (hfilter) {
    for (i = 0; i < n; i++) odo_filter4_col(c0 + (n/2 - 2)*stride + i, stride, 0);
  }
  if (vfilter) {
    for (i = 0; i < n; i++) odo_filter4_row(c0 + i*stride + n/2 - 2, 0);
  }
}

/* od_postfilter_split, src/filter.c:1485-1527 (non-deblocking branch): rows
   first, then columns. */
void odo_postfilter_split(odo_coeff *c0, int stride, int bs, int hfilter,
 int vfilter) {
  int n;
  int i;
  n = 4 << bs;
  if (vfilter) {
    for (i = 0; i < n; i++) odo_filter4_row(c0 + i*stride + n/2 - 2, 1);
  }
  if (hfilter) {
    for (i = 0; i < n; i++) odo_filter4_col(c0 + (n/2 - 2)*stride + i, stride, 1);
  }
}

/* od_apply_prefilter_frame_sbs, src/filter.c:1529-1559: every interior
   horizontal superblock edge (column taps, whole plane width) first, then every
   interior vertical edge (row taps, whole plane height). */
void odo_apply_prefilter_frame_sbs(odo_coeff *c0, int stride, int nhsb,
 int nvsb, int xdec, int ydec) {
  int sbw;
  int sbh;
  int sbx;
  int sby;
  int i;
  sbw = 64 >> xdec;
  sbh = 64 >> ydec;
  for (sby = 1; sby < nvsb; sby++) {
    for (i = 0; i < nhsb*sbw; i++) {
      odo_filter4_col(c0 + (sby*sbh - 2)*stride + i, stride, 0);
    }
  }
  for (sbx = 1; sbx < nhsb; sbx++) {
    for (i = 0; i < nvsb*sbh; i++) {
      odo_filter4_row(c0 + i*stride + sbx*sbw - 2, 0);
    }
  }
}

/* od_apply_postfilter_frame_sbs, src/filter.c:1589-1618: vertical edges
   (rows) first, then horizontal edges (columns). */
void odo_apply_postfilter_frame_sbs(odo_coeff *c0, int stride, int nhsb,
 int nvsb, int xdec, int ydec) {
  int sbw;
  int sbh;
  int sbx;
  int sby;
  int i;
  sbw = 64 >> xdec;
  sbh = 64 >> ydec;
  for (sbx = 1; sbx < nhsb; sbx++) {
    for (i = 0; i < nvsb*sbh; i++) {
      odo_filter4_row(c0 + i*stride + sbx*sbw - 2, 1);
    }
  }
  for (sby = 1; sby < nvsb; sby++) {
    for (i = 0; i < nhsb*sbw; i++) {
      odo_filter4_col(c0 + (sby*sbh - 2)*stride + i, stride, 1);
    }
  }
}

/* ======================================================================== */
/* Pixel <-> coefficient                                                     */
/* ======================================================================== */

/* Full-precision references (info.full_precision_references, src/encode.c:212-213,
   src/state.c:256-258): the picture buffers hold 16-bit samples at 8 + OD_COEFF_SHIFT =
   12 bits (xstride 2 in the reference).  odo_set_fpr(1) makes every px pointer of this
   file's plane functions (typed uint8_t * as daala_image_plane.data is) an array of
   uint16_t samples, strides in SAMPLES. */
static int odo_fpr;
void odo_set_fpr(int on) {
  odo_fpr = on != 0;
}

/* od_ref_buf_to_coeff, src/state.c:1238-1254 (xstride 2, lossy: coeff_shift 0):
   p - (1 << (8 + OD_COEFF_SHIFT) >> 1). */
void odo_px16_to_coeff(odo_coeff *dst, int dst_stride, const uint16_t *src, int src_stride, int w,
 int h) {
  int x;
  int y;
  for (y = 0; y < h; y++) {
    for (x = 0; x < w; x++) dst[y*dst_stride + x] = (int16_t)src[y*src_stride + x] - 2048;
  }
}

/* od_coeff_to_ref_buf, src/state.c:1306-1321 (xstride 2, lossy): OD_CLAMPFPR(c + (128 <<
   OD_COEFF_SHIFT)), src/odintrin.h:117-120. */
void odo_coeff_to_px16(uint16_t *dst, int dst_stride, const odo_coeff *src, int src_stride, int w,
 int h) {
  int x;
  int y;
  for (y = 0; y < h; y++) {
    for (x = 0; x < w; x++) {
      int v;
      v = src[y*src_stride + x] + 2048;
      dst[y*dst_stride + x] = (uint16_t)(v < 0 ? 0 : v > 4095 ? 4095 : v);
    }
  }
}

/* od_ref_buf_to_coeff, src/state.c:1231-1237 with coeff_shift =
   OD_COEFF_SHIFT = 4 (src/internal.h:124). */
void odo_px_to_coeff(odo_coeff *dst, int dst_stride, const uint8_t *src,
 int src_stride, int w, int h) {
  int x;
  int y;
  if (odo_fpr) {
    odo_px16_to_coeff(dst, dst_stride, (const uint16_t *)src, src_stride, w, h);
    return;
  }
  for (y = 0; y < h; y++) {
    for (x = 0; x < w; x++) dst[y*dst_stride + x] = (src[y*src_stride + x] - 128)*16;
  }
}

/* od_coeff_to_ref_buf, src/state.c:1296-1304: OD_CLAMP255(((c + 8) >> 4) +
   128). */
void odo_coeff_to_px(uint8_t *dst, int dst_stride, const odo_coeff *src,
 int src_stride, int w, int h) {
  int x;
  int y;
  if (odo_fpr) {
    odo_coeff_to_px16((uint16_t *)dst, dst_stride, src, src_stride, w, h);
    return;
  }
  for (y = 0; y < h; y++) {
    for (x = 0; x < w; x++) {
      int v;
      v = ((src[y*src_stride + x] + 8) >> 4) + 128;
      dst[y*dst_stride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
}

/* ======================================================================== */
/* Whole-plane stages                                                        */
/* ======================================================================== */

/* The pre-DCT samples of a block at level bs depend only on the source pixels
   and on "every ancestor was split": od_prefilter_split touches nothing
   outside its own block and is applied before recursing (src/encode.c:1489,
   :1760).  So the depth-first recursion of od_compute_dcts can be restated
   level by level: transform every level-bs block, split-filter every level-bs
   block, descend.  The filter gating reproduces src/encode.c:1487-1488: it
   uses the UNPADDED LUMA picture size, also for chroma planes. */
void odo_forward_pyramid_plane(odo_coeff **levels, odo_coeff *c,
 const uint8_t *px, int px_stride, int w, int h, int dec, int pic_w,
 int pic_h) {
  int top;
  int bs;
  top = ODO_NBSIZES - 1 - dec;
  odo_px_to_coeff(c, w, px, px_stride, w, h);
  odo_apply_prefilter_frame_sbs(c, w, w >> (6 - dec), h >> (6 - dec), dec, dec);
  for (bs = top; bs >= 0; bs--) {
    int n;
    int bx;
    int by;
    n = 4 << bs;
    for (by = 0; by < h/n; by++) {
      for (bx = 0; bx < w/n; bx++) {
        odo_fdct_2d(bs, levels[bs] + by*n*w + bx*n, w, c + by*n*w + bx*n, w);
      }
    }
    if (bs > 0) {
      for (by = 0; by < h/n; by++) {
        for (bx = 0; bx < w/n; bx++) {
          odo_prefilter_split(c + by*n*w + bx*n, w, bs,
           (bx + 1)*n <= pic_w, (by + 1)*n <= pic_h);
        }
      }
    }
  }
}

/* Inverse at a uniform partition: iDCT of every level-leaf_bs block, then the
   split post-filters of levels leaf_bs+1..top bottom-up (children before
   parents, src/encode.c:1780-1789), the superblock-edge post-filter
   (src/encode.c:2675) and the pixel conversion (:2845). */
void odo_inverse_level_plane(uint8_t *px, int px_stride, odo_coeff *c,
 const odo_coeff *d, int w, int h, int dec, int leaf_bs, int pic_w,
 int pic_h) {
  int top;
  int bs;
  int n;
  int bx;
  int by;
  top = ODO_NBSIZES - 1 - dec;
  n = 4 << leaf_bs;
  for (by = 0; by < h/n; by++) {
    for (bx = 0; bx < w/n; bx++) {
      odo_idct_2d(leaf_bs, c + by*n*w + bx*n, w, d + by*n*w + bx*n, w);
    }
  }
  for (bs = leaf_bs + 1; bs <= top; bs++) {
    n = 4 << bs;
    for (by = 0; by < h/n; by++) {
      for (bx = 0; bx < w/n; bx++) {
        odo_postfilter_split(c + by*n*w + bx*n, w, bs, (bx + 1)*n <= pic_w,
         (by + 1)*n <= pic_h);
      }
    }
  }
  odo_apply_postfilter_frame_sbs(c, w, w >> (6 - dec), h >> (6 - dec), dec, dec);
  odo_coeff_to_px(px, px_stride, c, w, w, h);
}

/* ======================================================================== */
/* Scan order                                                                */
/* ======================================================================== */

/* od_raster_to_coding_order, src/partition.c:144-170, through the composite
   scan table (tools/make_tables.py). */
void odo_raster_to_coding_order(odo_coeff *dst, int n, const odo_coeff *src,
 int stride) {
  int len;
  int i;
  len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  for (i = 0; i < len; i++) {
    dst[i] = src[OD_SCAN_XY[i][1]*stride + OD_SCAN_XY[i][0]];
  }
}

/* od_coding_order_to_raster, src/partition.c:176-194: only the coded
   positions are written. */
void odo_coding_order_to_raster(odo_coeff *dst, int stride,
 const odo_coeff *src, int n) {
  int len;
  int i;
  len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
  for (i = 0; i < len; i++) {
    dst[OD_SCAN_XY[i][1]*stride + OD_SCAN_XY[i][0]] = src[i];
  }
}

int odo_band_offsets(int bs, int *out) {
  int i;
  for (i = 0; i <= OD_NBANDS[bs]; i++) out[i] = OD_BAND_OFFS[bs][i];
  return OD_NBANDS[bs];
}

/* od_img_plane_copy_pad, src/encode.c:752-837, 8-bit planes (xstride 1): copy
   the picture, then extend it into the padding by a [1 2 1]/4 low-pass of the
   previous column (right side, rows of the picture only) and of the previous
   row (bottom, the whole padded width).  SURVEY.md 8(f) rank 4, input side. */
void odo_img_plane_copy_pad(uint8_t *dst, int dstride, int plane_w, int plane_h,
 const uint8_t *src, int sstride, int pic_w, int pic_h) {
  int x;
  int y;
  if (pic_w == 0 || pic_h == 0) {
    for (y = 0; y < plane_h; y++) memset(dst + y*dstride, 0, plane_w);
    return;
  }
  for (y = 0; y < pic_h; y++) memcpy(dst + y*dstride, src + y*sstride, pic_w);
  for (x = pic_w; x < plane_w; x++) {
    for (y = 0; y < pic_h; y++) {
      int c;
      int u;
      int d;
      c = dst[y*dstride + x - 1];
      u = dst[(y > 0 ? y - 1 : y)*dstride + x - 1];
      d = dst[(y + 1 < pic_h ? y + 1 : y)*dstride + x - 1];
      dst[y*dstride + x] = (uint8_t)((2*c + u + d + 2) >> 2);
    }
  }
  for (y = pic_h; y < plane_h; y++) {
    const uint8_t *up;
    up = dst + (y - 1)*dstride;
    for (x = 0; x < plane_w; x++) {
      int c;
      int l;
      int r;
      c = up[x];
      l = up[x - (x > 0)];
      r = up[x + (x + 1 < plane_w)];
      dst[y*dstride + x] = (uint8_t)((2*c + l + r + 2) >> 2);
    }
  }
}

/* The same for a full-precision-references encoder: the padded plane holds 16-bit samples
   at 12 bits.  Step 1 is od_img_plane_copy's bit-depth conversion (src/state.c:93-213): an
   8-bit source (src_bitdepth 8, uint8_t samples) or a 10 / 12-bit one (uint16_t samples) is
   shifted up into 12 bits and clamped to [0, 4095]; step 2 the [1 2 1]/4 extension on
   uint16_t (src/encode.c:791-803, :821-832). */
void odo_img_plane_copy_pad16(uint16_t *dst, int dstride, int plane_w, int plane_h, const void *src,
 int src_bitdepth, int sstride, int pic_w, int pic_h) {
  int x;
  int y;
  if (pic_w == 0 || pic_h == 0) {
    for (y = 0; y < plane_h; y++) memset(dst + y*dstride, 0, plane_w*sizeof(*dst));
    return;
  }
  for (y = 0; y < pic_h; y++) {
    for (x = 0; x < pic_w; x++) {
      int v;
      if (src_bitdepth > 8) v = ((const int16_t *)src)[y*sstride + x] << (12 - src_bitdepth);
      else v = ((const uint8_t *)src)[y*sstride + x] << 4;
      /* the reference stores through an int16_t pointer: OD_CLAMPI(0, v, 4095) */
      dst[y*dstride + x] = (uint16_t)(v < 0 ? 0 : v > 4095 ? 4095 : v);
    }
  }
  for (x = pic_w; x < plane_w; x++) {
    for (y = 0; y < pic_h; y++) {
      int c;
      int u;
      int d;
      c = dst[y*dstride + x - 1];
      u = dst[(y > 0 ? y - 1 : y)*dstride + x - 1];
      d = dst[(y + 1 < pic_h ? y + 1 : y)*dstride + x - 1];
      dst[y*dstride + x] = (uint16_t)((2*c + u + d + 2) >> 2);
    }
  }
  for (y = pic_h; y < plane_h; y++) {
    const uint16_t *up;
    up = dst + (y - 1)*dstride;
    for (x = 0; x < plane_w; x++) {
      int c;
      int l;
      int r;
      c = up[x];
      l = up[x - (x > 0)];
      r = up[x + (x + 1 < plane_w)];
      dst[y*dstride + x] = (uint16_t)((2*c + l + r + 2) >> 2);
    }
  }
}

/* od_resample_luma_coeffs, src/intra.c:72-109, for 4:2:0 (xdec = ydec = 1): the
   chroma-from-luma prediction of a chroma block of side n = 4 << bs from the
   decoded luma coefficients of the co-located 2n x 2n area (decoded_luma, row
   stride dlstride).  luma_bs == 0 (the luma area is four 4x4 blocks, so the chroma
   block is 4x4): od_tf_up_hv_lp (src/tf.c:82-108) merges the 2x2 low-frequency
   corners of the four blocks with the Haar kernel OD_HAAR_KERNEL (src/tf.h:34-45),
   then OD_CFL_SCALING4 (src/intra.c:65-70).  Otherwise the luma area is one block
   and the prediction is its upper-left quarter (:97-108). */
void odo_resample_luma_coeffs(odo_coeff *chroma_pred, int cpstride,
 const odo_coeff *decoded_luma, int dlstride, int bs, int luma_bs) {
  static const int16_t scaling4[4][4] = {
    {128, 128, 100, 36}, {128, 80, 71, 35}, {100, 71, 35, 31}, {36, 35, 31, 18}};
  int n;
  int x;
  int y;
  n = 4 << bs;
  if (luma_bs == 0) {
    for (y = 0; y < n >> 1; y++) {
      int vswap;
      vswap = y & 1;
      for (x = 0; x < n >> 1; x++) {
        odo_coeff ll;
        odo_coeff lh;
        odo_coeff hl;
        odo_coeff hh;
        odo_coeff t;
        int hswap;
        ll = decoded_luma[y*dlstride + x];
        lh = decoded_luma[y*dlstride + x + n];
        hl = decoded_luma[(y + n)*dlstride + x];
        hh = decoded_luma[(y + n)*dlstride + x + n];
        /* OD_HAAR_KERNEL(ll, hl, lh, hh): lh and hl swapped, src/tf.c:99-100 */
        ll += lh;
        hh -= hl;
        t = (ll - hh) >> 1;
        hl = t - hl;
        lh = t - lh;
        ll -= hl;
        hh += lh;
        hswap = x & 1;
        chroma_pred[(2*y + vswap)*cpstride + 2*x + hswap] = ll;
        chroma_pred[(2*y + vswap)*cpstride + 2*x + 1 - hswap] = lh;
        chroma_pred[(2*y + 1 - vswap)*cpstride + 2*x + hswap] = hl;
        chroma_pred[(2*y + 1 - vswap)*cpstride + 2*x + 1 - hswap] = hh;
      }
    }
    for (y = 0; y < 4; y++) {
      for (x = 0; x < 4; x++) {
        chroma_pred[y*cpstride + x] = (scaling4[x][y]*chroma_pred[y*cpstride + x] + 64) >> 7;
      }
    }
  }
  else {
    for (y = 0; y < n; y++) {
      for (x = 0; x < n; x++) chroma_pred[y*cpstride + x] = decoded_luma[y*dlstride + x];
    }
  }
}

/* ======================================================================== */
/* PVQ fixed-point helpers (src/pvq.c, src/odintrin.h:164-199)               */
/* ======================================================================== */

#define Q_CGAIN_SHIFT 8     /* OD_CGAIN_SHIFT, src/pvq.h:88 */
#define Q_COMPAND_SHIFT 12  /* OD_COMPAND_SHIFT = 8 + OD_COEFF_SHIFT, :104 */
#define Q_BETA_SHIFT 12     /* OD_BETA_SHIFT, :81 */
#define Q_QM_SHIFT 11       /* OD_QM_SHIFT, :60 */
#define Q_QM_INV_SHIFT 12   /* OD_QM_INV_SHIFT, :67 */
#define BETA_1_0 4096
#define BETA_1_5 6144

int odo_ilog(uint32_t v) { /* od_ilog, src/internal.c:401 */
  int n;
  n = 0;
  while (v) {
    n++;
    v >>= 1;
  }
  return n;
}

static int32_t shl32(int64_t a, int s) { /* OD_SHL: wraps to 32 bits */
  return (int32_t)((uint32_t)a << s);
}

static int32_t shr_round(int64_t x, int s) { /* OD_SHR_ROUND */
  return (int32_t)((x + ((1 << s) >> 1)) >> s);
}

static int32_t vshr(int64_t x, int s) { /* OD_VSHR */
  return s > 0 ? (int32_t)(x >> s) : shl32(x, -s);
}

static int32_t vshr_round(int64_t x, int s) { /* OD_VSHR_ROUND */
  return s > 0 ? shr_round(x, s) : shl32(x, -s);
}

static int32_t mult16_16(int32_t a, int32_t b) { /* OD_MULT16_16 */
  return (int32_t)(int16_t)a*(int32_t)(int16_t)b;
}

static int32_t mult16_16_q15(int32_t a, int32_t b) {
  return ((int16_t)a*(int32_t)(int16_t)b) >> 15;
}

static int32_t mult16_16_q16(int32_t a, int32_t b) {
  return ((int16_t)a*(int32_t)(int16_t)b) >> 16;
}

static int32_t mult16_16_qbeta(int32_t a, int32_t b) {
  return ((int16_t)a*(int32_t)(int16_t)b) >> Q_BETA_SHIFT;
}

static int64_t mult16_32_q16(int32_t a, int32_t b) {
  return (int16_t)a*(int64_t)b >> 16;
}

/* od_vector_log_mag, src/pvq.c:472-484. */
int odo_vector_log_mag(const odo_coeff *x, int n) {
  int32_t sum;
  int i;
  sum = 0;
  for (i = 0; i < n; i++) {
    int16_t t;
    t = (int16_t)(x[i] >> 8);
    sum += t*(int32_t)t;
  }
  return 8 + 1 + odo_ilog(n + sum)/2;
}

/* od_rsqrt_norm, src/pvq.c:968-997. */
static int16_t odo_rsqrt_norm(int16_t t) {
  int16_t n;
  int32_t r;
  int32_t r2;
  int32_t ry;
  int32_t y;
  n = (int16_t)(t - 32768);
  r = 23565 + mult16_16_q15(n, -13481 + mult16_16_q15(n, 6711));
  r2 = r*r;
  y = (((r2 >> 15)*n + r2) >> 12) - 131077;
  ry = r*y;
  return (int16_t)(r + ((((ry >> 16)*(3*y) >> 3) - ry) >> 18));
}

/* od_rsqrt, src/pvq.c:999-1015. */
static int16_t odo_rsqrt(int32_t x, int *rsqrt_shift) {
  int k;
  int s;
  k = (odo_ilog(x) - 1) >> 1;
  s = 2*k - 14;
  *rsqrt_shift = 14 + ((s + 16) >> 1);
  return odo_rsqrt_norm((int16_t)vshr(x, s));
}

/* od_sqrt_norm + od_sqrt, src/pvq.c:729-756. */
static int16_t odo_sqrt(int32_t x, int *sqrt_shift) {
  int k;
  int s;
  int32_t t;
  int32_t v;
  if (x == 0) {
    *sqrt_shift = 0;
    return 0;
  }
  k = (odo_ilog(x) - 1) >> 1;
  s = 2*k - 14;
  t = vshr(x, s);
  *sqrt_shift = 15 - ((s + 16) >> 1);
  v = shr_round(t*odo_rsqrt_norm((int16_t)t), 15);
  return (int16_t)(v < 32767 ? v : 32767);
}

/* od_rcp, src/pvq.c:526-549. */
static int16_t odo_rcp(int16_t x) {
  int i;
  int16_t n;
  int16_t r;
  i = odo_ilog(x) - 1;
  n = (int16_t)(vshr_round(x, i - 15) - 32768);
  r = (int16_t)(30840 + mult16_16_q15(-15420, n));
  r = (int16_t)(r - mult16_16_q15(r, mult16_16_q15(r, n) + r - 32768));
  r = (int16_t)(r - (1 + mult16_16_q15(r, mult16_16_q15(r, n) + r - 32768)));
  return (int16_t)vshr_round(r, i - 14);
}

/* od_beta_rcp, src/pvq.c:626-637. */
static int16_t odo_beta_rcp(int16_t beta) {
  if (beta == BETA_1_0) return BETA_1_0;
  if (beta == BETA_1_5) return 2731; /* OD_BETA(1./1.5) */
  return (int16_t)shr_round(odo_rcp((int16_t)(beta << (15 - 1 - Q_BETA_SHIFT))),
   14 + 1 - Q_BETA_SHIFT);
}

/* od_exp2_frac / od_exp2, src/pvq.c:642-665. */
static int32_t odo_exp2(int32_t x) {
  int integer;
  int32_t f;
  int32_t frac;
  integer = x >> 15;
  if (integer > 14) return 0x7f000000;
  if (integer < -15) return 0;
  f = x - shl32(integer, 15);
  frac = mult16_16_q15(f, 22709 + mult16_16_q15(f, 7913
   + mult16_16_q15(f, 1704 + mult16_16_q15(f, 443))));
  return vshr_round(32768 + frac, -integer) + 1;
}

/* od_log2, src/pvq.c:671-676. */
static int16_t odo_log2(int16_t x) {
  return (int16_t)(x + mult16_16_q15(x, 14482 + mult16_16_q15(x, -23234
   + mult16_16_q15(x, 13643 + mult16_16_q15(x, -6403
   + mult16_16_q15(x, 1515))))));
}

/* od_pow, src/pvq.c:678-696. */
static int32_t odo_pow(int32_t x, int16_t beta) {
  int16_t t;
  int log2_x;
  int32_t logr;
  if (x == 0) return 0;
  log2_x = odo_ilog(x) - 1;
  t = (int16_t)(vshr(x, log2_x - 15) - 32768);
  logr = odo_log2(t) + (log2_x - Q_COMPAND_SHIFT)*32768;
  logr = (int32_t)(beta*(int64_t)logr >> Q_BETA_SHIFT);
  return odo_exp2(logr);
}

/* od_gain_compand, src/pvq.c:706-722. */
static int32_t odo_gain_compand(int32_t g, int q0, int16_t beta) {
  int32_t expr;
  if (beta == BETA_1_0) return (256*g + (q0 >> 1))/q0;
  expr = odo_pow(g, odo_beta_rcp(beta));
  expr <<= Q_CGAIN_SHIFT + Q_COMPAND_SHIFT - 15;
  return (expr + (q0 >> 1))/q0;
}

/* od_gain_expand, src/pvq.c:766-811. */
int32_t odo_gain_expand(int32_t cg0, int q0, int beta) {
  if (beta == BETA_1_0) return shr_round(cg0*q0, Q_CGAIN_SHIFT);
  if (beta == BETA_1_5) {
    int32_t irt;
    int64_t tmp;
    int sqrt_outshift;
    irt = odo_sqrt(cg0*q0, &sqrt_outshift);
    tmp = cg0*q0*(int64_t)irt;
    return vshr_round(tmp, Q_CGAIN_SHIFT + sqrt_outshift
     + ((Q_CGAIN_SHIFT + Q_COMPAND_SHIFT) >> 1));
  }
  return shr_round(odo_pow(shr_round(cg0*q0, Q_CGAIN_SHIFT), (int16_t)beta),
   15 - Q_COMPAND_SHIFT);
}

/* od_pvq_compute_gain, src/pvq.c:824-847. */
int32_t odo_pvq_compute_gain(const int16_t *x, int n, int q0, int32_t *g,
 int beta, int bshift) {
  int32_t acc;
  int32_t irt;
  int sqrt_shift;
  int i;
  acc = 0;
  for (i = 0; i < n; i++) acc += x[i]*(int32_t)x[i];
  irt = odo_sqrt(acc, &sqrt_shift);
  *g = vshr_round(irt, sqrt_shift - bshift);
  return odo_gain_compand(*g, q0, (int16_t)beta);
}

/* od_pvq_compute_max_theta, src/pvq.c:855-866.
   OD_QCONST32(M_PI/2, 8) = 402, OD_QCONST32(1.4, 8) = 358. */
int odo_pvq_compute_max_theta(int32_t qcg, int beta) {
  int ts;
  ts = shr_round(qcg*mult16_16_qbeta(402, odo_beta_rcp((int16_t)beta)),
   2*Q_CGAIN_SHIFT);
  if (qcg < 358) ts = 1;
  return ts;
}

/* od_pvq_compute_theta, src/pvq.c:874-884. */
int32_t odo_pvq_compute_theta(int t, int max_theta) {
  if (max_theta == 0) return 0;
  return (32768*(t < max_theta - 1 ? t : max_theta - 1) + (max_theta >> 1))
   /max_theta;
}

/* od_pvq_compute_k, src/pvq.c:902-960 (fixed point; the non-nodesync branch
   :955-957 is float and unreachable with OD_ROBUST_STREAM, src/encode.c:1354).
   OD_QCONST32(.2, 8) = 51, OD_QCONST32(.2, 15) = 6554, OD_BETA(1.25) = 5120. */
int odo_pvq_compute_k(int32_t qcg, int itheta, int32_t theta, int noref,
 int n, int beta, int nodesync) {
  static const int16_t SQRT_TABLE[2][13] = {
    {0, 0, 0, 0, 2290, 2985, 4222, 0, 8256, 0, 16416, 0, 32767},
    {0, 0, 0, 0, 2401, 3072, 4284, 0, 8287, 0, 16432, 0, 32767}
  };
  int32_t v;
  (void)theta;
  if (noref) {
    if (qcg == 0) return 0;
    if (n == 15 && qcg == 256 && beta > 5120) return 1;
    v = shr_round((qcg - (int64_t)51)
     *mult16_16_qbeta(odo_beta_rcp((int16_t)beta), SQRT_TABLE[1][odo_ilog(n + 1)]),
     Q_CGAIN_SHIFT + 10);
    return v > 1 ? v : 1;
  }
  if (itheta == 0) return 0;
  if (!nodesync) abort();
  v = vshr_round((shl32(itheta, 15) - 6554)
   *(int64_t)SQRT_TABLE[0][odo_ilog(n + 1)], 10 + 15);
  return v > 1 ? v : 1;
}

/* od_pvq_cos_pi_2, src/pvq.c:417-423. */
static int16_t odo_cos_pi_2(int16_t x) {
  int16_t x2;
  int32_t v;
  x2 = (int16_t)mult16_16_q15(x, x);
  v = (1073758164 - x*x + x2*(-7654 + mult16_16_q16(x2, 16573
   + mult16_16_q16(-2529, x2)))) >> 15;
  return (int16_t)(v < 32767 ? v : 32767);
}

/* od_pvq_cos, src/pvq.c:428-457. */
int odo_pvq_cos(int32_t x) {
  x &= 0x1ffff;
  if (x > (1 << 16)) x = (1 << 17) - x;
  if (x & 0x7fff) {
    if (x < (1 << 15)) return odo_cos_pi_2((int16_t)x);
    return (int16_t)-odo_cos_pi_2((int16_t)(65536 - x));
  }
  if (x & 0xffff) return 0;
  if (x & 0x1ffff) return -32767;
  return 32767;
}

/* od_pvq_sin, src/pvq.c:461-467. */
int odo_pvq_sin(int32_t x) {
  return odo_pvq_cos(32768 - x);
}

/* od_compute_householder, src/pvq.c:498-521: first largest |r_i| wins. */
int odo_compute_householder(int16_t *r, int n, int32_t gr, int *sign,
 int shift) {
  int m;
  int i;
  int s;
  int16_t maxr;
  m = 0;
  maxr = 0;
  for (i = 0; i < n; i++) {
    if (abs(r[i]) > maxr) {
      maxr = (int16_t)abs(r[i]);
      m = i;
    }
  }
  s = r[m] > 0 ? 1 : -1;
  r[m] = (int16_t)(r[m] + shr_round(gr*s, shift));
  *sign = s;
  return m;
}

/* od_apply_householder, src/pvq.c:560-623 (fixed point). */
void odo_apply_householder(int16_t *out, const int16_t *x, const int16_t *r,
 int n) {
  int32_t l2r;
  int32_t proj;
  int16_t proj_1;
  int16_t proj_norm;
  int16_t l2r_norm;
  int16_t rcp;
  int proj_shift;
  int l2r_shift;
  int outshift;
  int i;
  l2r = 0;
  for (i = 0; i < n; i++) l2r += mult16_16(r[i], r[i]);
  proj = 0;
  for (i = 0; i < n; i++) proj += mult16_16(r[i], x[i]);
  l2r_shift = (odo_ilog(l2r) - 1) - 14;
  l2r_norm = (int16_t)vshr_round(l2r, l2r_shift);
  rcp = odo_rcp(l2r_norm);
  proj_shift = (odo_ilog(abs(proj)) - 1) - 14;
  proj_norm = (int16_t)vshr_round(proj, proj_shift);
  proj_1 = (int16_t)mult16_16_q15(proj_norm, rcp);
  outshift = 14 - proj_shift - 1 + l2r_shift;
  if (outshift > 30) outshift = 30;
  for (i = 0; i < n; i++) {
    int32_t tmp;
    tmp = mult16_16(r[i], proj_1);
    tmp = outshift >= 0 ? shr_round(tmp, outshift) : shl32(tmp, -outshift);
    out[i] = (int16_t)(x[i] - tmp);
  }
}

/* od_pvq_synthesis_partial, src/pvq.c:1037-1115 (fixed point; the two double
   multiplies by 2^-15 at :1096,:1102 are exact). */
void odo_pvq_synthesis_partial(odo_coeff *xcoeff, const odo_coeff *ypulse,
 const int16_t *r16, int n, int noref, int32_t g, int32_t theta, int m, int s,
 const int16_t *qm_inv) {
  int i;
  int yy;
  int32_t scale;
  int nn;
  int gshift;
  int qshift;
  nn = n - (!noref);
  yy = 0;
  for (i = 0; i < nn; i++) yy += ypulse[i]*(int32_t)ypulse[i];
  gshift = odo_ilog(g) - 14;
  if (gshift < 0) gshift = 0;
  if (yy == 0) scale = 0;
  else {
    int rsqrt_shift;
    int16_t rsqrt;
    rsqrt = odo_rsqrt(yy, &rsqrt_shift);
    scale = vshr_round(rsqrt*(int64_t)g, rsqrt_shift + gshift - 16);
  }
  qshift = Q_QM_INV_SHIFT - gshift;
  if (noref) {
    for (i = 0; i < n; i++) {
      int32_t x;
      x = (int32_t)mult16_32_q16(ypulse[i], scale);
      xcoeff[i] = shr_round(x*qm_inv[i], qshift);
    }
  }
  else {
    int16_t x[ODO_MAX_PVQ_SIZE];
    scale = (int32_t)floor(.5 + scale*(1./32768)*odo_pvq_sin(theta));
    for (i = 0; i < m; i++) x[i] = (int16_t)mult16_32_q16(ypulse[i], scale);
    x[m] = (int16_t)floor(.5 + -s*shr_round(g, gshift)*(1./32768)
     *odo_pvq_cos(theta));
    for (i = m; i < nn; i++) x[i + 1] = (int16_t)mult16_32_q16(ypulse[i], scale);
    odo_apply_householder(x, x, r16, n);
    for (i = 0; i < n; i++) xcoeff[i] = shr_round(x[i]*qm_inv[i], qshift);
  }
}

/* neg_deinterleave, src/pvq_decoder.c:53-59. */
static int odo_neg_deinterleave(int x, int ref) {
  if (x < 2*ref - 1) {
    if (x & 1) return ref - 1 - (x >> 1);
    return ref + (x >> 1);
  }
  return x + 1;
}

/* pvq_decode_partition (src/pvq_decoder.c:122-298) WITHOUT its entropy-decoder reads:
   given what it read - qg_coded = the gain symbol before deinterleaving (id & 1, or 1 + the
   generic_decode value, :190-208), itheta after the theta decode (:248-254), noref, and the
   pulses y (n - !noref of them, :262-266) - and the reference band: rshift / ref16 (:213-224),
   the reference's gain and the deinterleaved gain (:225-241), gain_offset, qcg, max_theta,
   theta (:242-255), the skip rules, K (:261; nodesync, OD_ROBUST_STREAM), od_gain_expand and
   pvq_synthesis = od_compute_householder + od_pvq_synthesis_partial (:78-89, :275-279).
   Returns the skip code (0, 1 = OD_PVQ_SKIP_ZERO, 2 = OD_PVQ_SKIP_COPY); *k_out = K. */
int odo_pvq_decode_band(odo_coeff *out, const odo_coeff *ref, const odo_coeff *y, int n, int q0,
 int beta, int qg_coded, int itheta, int noref, int is_keyframe, int pli, const int16_t *qm,
 const int16_t *qm_inv, int *k_out) {
  int16_t ref16[ODO_MAX_PVQ_SIZE];
  int32_t theta;
  int32_t gr;
  int32_t gain_offset;
  int32_t qcg;
  int qg;
  int skip;
  int rshift;
  int k;
  int i;
  theta = 0;
  gr = 0;
  gain_offset = 0;
  qg = qg_coded;
  skip = 0;
  rshift = odo_vector_log_mag(ref, n) - 14;
  if (rshift < 0) rshift = 0;
  for (i = 0; i < n; i++) {
    ref16[i] = (int16_t)shr_round((int32_t)((uint32_t)ref[i]*(uint32_t)(int32_t)qm[i]), Q_QM_SHIFT + rshift);
  }
  if (!noref) {
    int32_t cgr;
    int icgr;
    int cfl_enabled;
    int max_theta;
    cfl_enabled = pli != 0 && is_keyframe;
    cgr = odo_pvq_compute_gain(ref16, n, q0, &gr, beta, rshift);
    if (cfl_enabled) cgr = 256;
    icgr = shr_round(cgr, Q_CGAIN_SHIFT);
    if (is_keyframe) qg = odo_neg_deinterleave(qg, icgr);
    else {
      qg = odo_neg_deinterleave(qg, icgr + 1) - 1;
      if (qg == 0) skip = icgr ? 1 : 2;
    }
    if (qg == icgr && itheta == 0 && !cfl_enabled) skip = 2;
    gain_offset = cgr - shl32(icgr, Q_CGAIN_SHIFT);
    qcg = shl32(qg, Q_CGAIN_SHIFT) + gain_offset;
    max_theta = odo_pvq_compute_max_theta(qcg, beta);
    theta = odo_pvq_compute_theta(itheta, max_theta);
  }
  else {
    itheta = 0;
    if (!is_keyframe) qg++;
    qcg = shl32(qg, Q_CGAIN_SHIFT);
    if (qg == 0) skip = 1;
  }
  k = odo_pvq_compute_k(qcg, itheta, theta, noref, n, beta, 1);
  if (k_out) *k_out = k;
  if (skip) {
    if (skip == 2) for (i = 0; i < n; i++) out[i] = ref[i];
    else for (i = 0; i < n; i++) out[i] = 0;
  }
  else {
    int32_t g;
    int s;
    int m;
    g = odo_gain_expand(qcg, q0, beta);
    s = 0;
    m = noref ? 0 : odo_compute_householder(ref16, n, gr, &s, rshift);
    odo_pvq_synthesis_partial(out, y, ref16, n, noref, g, theta, m, s, qm_inv);
  }
  return skip;
}

/* ======================================================================== */
/* PVQ search                                                                */
/* ======================================================================== */

/* od_rsqrt_table, src/pvq_encoder.c:52-60: the first 16 entries are 6-digit
   DECIMAL literals, not 1/sqrt(i). */
static double odo_rsqrt_table(int i) {
  static const double table[16] = {
    1.000000, 0.707107, 0.577350, 0.500000,
    0.447214, 0.408248, 0.377964, 0.353553,
    0.333333, 0.316228, 0.301511, 0.288675,
    0.277350, 0.267261, 0.258199, 0.250000};
  if (i <= 16) return table[i - 1];
  return 1./sqrt(i);
}

/* pvq_search_rdo_double, src/pvq_encoder.c:93-224.  Evaluation order of every
   floating-point expression follows the reference statement by statement. */
double odo_pvq_search_rdo_double(const int16_t *xcoeff, int n, int k,
 odo_coeff *ypulse, double g2, double pvq_norm_lambda, int prev_k) {
  double x[1024];
  double xx;
  double xy;
  double yy;
  double lambda;
  double norm_1;
  double delta_rate;
  double accel_rate;
  int rdo_pulses;
  int i;
  int j;
  xx = xy = yy = 0;
  for (j = 0; j < n; j++) {
    x[j] = fabs((float)xcoeff[j]);
    xx += x[j]*x[j];
  }
  norm_1 = 1./sqrt(1e-30 + xx);
  lambda = pvq_norm_lambda/(1e-30 + g2);
  i = 0;
  if (prev_k > 0 && prev_k <= k) {
    for (j = 0; j < n; j++) {
      ypulse[j] = abs(ypulse[j]);
      xy += x[j]*ypulse[j];
      yy += ypulse[j]*ypulse[j];
      i += ypulse[j];
    }
  }
  else if (k > 2) {
    double l1_norm;
    double l1_inv;
    l1_norm = 0;
    for (j = 0; j < n; j++) l1_norm += x[j];
    l1_inv = 1./(l1_norm > 1e-100 ? l1_norm : 1e-100);
    for (j = 0; j < n; j++) {
      double tmp;
      int v;
      tmp = k*x[j]*l1_inv;
      v = (int)floor(tmp);
      ypulse[j] = v > 0 ? v : 0;
      xy += x[j]*ypulse[j];
      yy += ypulse[j]*ypulse[j];
      i += ypulse[j];
    }
  }
  else memset(ypulse, 0, n*sizeof(*ypulse));
  rdo_pulses = 1 + k/4;
  delta_rate = 3./n;
  accel_rate = 0.;
  if (k == 1) {
    if (n == 15) {
      accel_rate = -8./n;
      delta_rate = 4.5/n - accel_rate;
    }
    else if (n == 8) {
      accel_rate = 5.7/n;
      delta_rate = 9.3/n - accel_rate;
    }
  }
  for (; i < k - rdo_pulses; i++) {
    int pos;
    double best_xy;
    double best_yy;
    pos = 0;
    best_xy = -10;
    best_yy = 1;
    for (j = 0; j < n; j++) {
      double tmp_xy;
      double tmp_yy;
      tmp_xy = xy + x[j];
      tmp_yy = yy + 2*ypulse[j] + 1;
      tmp_xy *= tmp_xy;
      if (j == 0 || tmp_xy*best_yy > best_xy*tmp_yy) {
        best_xy = tmp_xy;
        best_yy = tmp_yy;
        pos = j;
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*ypulse[pos] + 1;
    ypulse[pos]++;
  }
  for (; i < k; i++) {
    double rsqrt_tab[4];
    int pos;
    double best_cost;
    pos = 0;
    best_cost = -1e5;
    for (j = 0; j < 4; j++) rsqrt_tab[j] = odo_rsqrt_table((int)(yy + 2*j + 1));
    for (j = 0; j < n; j++) {
      double tmp_xy;
      double tmp_yy;
      tmp_xy = xy + x[j];
      tmp_yy = ypulse[j] < 4 ? rsqrt_tab[ypulse[j]]
       : odo_rsqrt_table((int)(yy + 2*ypulse[j] + 1));
      tmp_xy = 2*tmp_xy*norm_1*tmp_yy - lambda*j*(delta_rate + j*accel_rate);
      if (j == 0 || tmp_xy > best_cost) {
        best_cost = tmp_xy;
        pos = j;
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*ypulse[pos] + 1;
    ypulse[pos]++;
  }
  for (i = 0; i < n; i++) {
    if (xcoeff[i] < 0) ypulse[i] = -ypulse[i];
  }
  return xy/(1e-100 + sqrt(xx*yy));
}

void odo_pvq_search_batch(const int16_t *x, int n, const int *k, odo_coeff *y,
 const double *g2, double pvq_norm_lambda, const int *prev_k, double *cos_out,
 long nbands) {
  long b;
  for (b = 0; b < nbands; b++) {
    cos_out[b] = odo_pvq_search_rdo_double(x + b*n, n, k[b], y + b*n, g2[b],
     pvq_norm_lambda, prev_k ? prev_k[b] : 0);
  }
}

/* ======================================================================== */
/* One band: pvq_theta                                                       */
/* ======================================================================== */

/* The chroma-from-luma sign decision of od_pvq_encode, src/pvq_encoder.c:846-872
   (keyframe chroma blocks only): the QM-weighted dot product of the first band
   of the block with its luma-derived reference; when negative, the reference of
   the whole block is negated.  ref/in: one block in coding order, qm: the
   block's QM in coding order.  OD_CFL_FLIP_SHIFT = OD_LIMIT_BSIZE_MAX = 4
   (src/pvq_encoder.c:42, src/internal.h:101).  Returns flip. */
int odo_cfl_flip(odo_coeff *ref, const odo_coeff *in, const int16_t *qm, int bs) {
  int32_t xy;
  int i;
  const int *off;
  int nb;
  off = OD_BAND_OFFS[bs];
  nb = OD_NBANDS[bs];
  xy = 0;
  for (i = off[0]; i < off[1]; i++) {
    int32_t rq;
    int32_t inq;
    rq = (int32_t)((uint32_t)ref[i]*(uint32_t)(int32_t)qm[i]);
    inq = (int32_t)((uint32_t)in[i]*(uint32_t)(int32_t)qm[i]);
    xy = (int32_t)((uint32_t)xy + (uint32_t)((rq*(int64_t)inq) >> ((Q_QM_SHIFT + 4) << 1)));
  }
  if (xy < 0) {
    for (i = off[0]; i < off[nb]; i++) ref[i] = -ref[i];
    return 1;
  }
  return 0;
}

/* od_pvq_rate, src/pvq_encoder.c:247-287, speed > 0 branch only (closed form;
   the speed == 0 branch prices with the live adaptive entropy coder and stays
   in the reference's host code). */
static double odo_pvq_rate_fast(int qg, int icgr, int theta, int ts,
 const odo_coeff *y0, int k, int n, int is_keyframe, int pli) {
  double rate;
  if (k == 0) rate = 0;
  else {
    int i;
    int sum;
    double f;
    double a;
    sum = 0;
    for (i = 0; i < n - (theta != -1); i++) sum += i*abs(y0[i]);
    f = sum/(double)(k*n);
    a = log(n*2*(1*f + .025))*k/n;
    rate = (1 + .4*f)*n*(1.4426950408889634073599246810019
     *log(1 + (a > 0 ? a : 0))) + 3;
  }
  if (qg > 0 && theta >= 0) {
    rate += .9*(1.4426950408889634073599246810019*log(ts));
    if (is_keyframe && pli == 0) rate += 6;
    if (qg == icgr) rate -= .5;
  }
  return rate;
}

/* The same, callable from the tests that play the host's part of the split
   band stage (price every candidate, let the GPU choose). */
double odo_pvq_rate_speed1(int qg, int icgr, int theta, int ts, const odo_coeff *y0, int k,
 int n, int is_keyframe, int pli) {
  return odo_pvq_rate_fast(qg, icgr, theta, ts, y0, k, n, is_keyframe, pli);
}

#define PRICE_NAME(x) odo_##x
#define PRICE_COEFF odo_coeff
#define PRICE_RATE(qg, icgr, theta, ts, y, k, n, kf, pli) \
  odo_pvq_rate_fast(qg, icgr, theta, ts, y, k, n, kf, pli)
#include "price_batch.inc"
#undef PRICE_NAME
#undef PRICE_COEFF
#undef PRICE_RATE

static int odo_neg_interleave(int x, int ref) { /* src/pvq_encoder.c:235-239 */
  if (x < ref) return -2*(x - ref) - 1;
  if (x < 2*ref) return 2*(x - ref);
  return x - 1;
}

typedef struct {
  int gain, k, theta, ts;
  int32_t qcg, qtheta;
} odo_item;

/* pvq_theta, src/pvq_encoder.c:333-641.  Requires speed > 0 for the final
   selection; the per-candidate quantities recorded in *trace are independent
   of speed and of the adaptation state. */
int odo_pvq_theta(odo_coeff *out, const odo_coeff *x0, const odo_coeff *r0,
 int n, int q0, odo_coeff *y, int *itheta, int *max_theta, int *vk, int beta,
 double *skip_diff, int nodesync, int is_keyframe, int pli, const int16_t *qm,
 const int16_t *qm_inv, double pvq_norm_lambda, int speed,
 odo_pvq_band_trace *trace) {
  const double gain_weight = 1.4;
  const double cgain_scale_2 = (1./256)*(1./256);
  int32_t g;
  int32_t gr;
  int32_t cg;
  int32_t cgr;
  int32_t gain_offset;
  int32_t theta;
  int32_t best_qtheta;
  odo_coeff y_tmp[ODO_MAX_PVQ_SIZE];
  int16_t x16[ODO_MAX_PVQ_SIZE];
  int16_t r16[ODO_MAX_PVQ_SIZE];
  double best_cost;
  double dist0;
  double best_dist;
  double dist;
  double corr;
  double skip_dist;
  int icgr;
  int qg;
  int s;
  int m;
  int k;
  int best_k;
  int noref;
  int cfl_enabled;
  int skip;
  int xshift;
  int rshift;
  int r_null;
  int i;
  (void)speed;
  corr = 0;
  xshift = odo_vector_log_mag(x0, n) - 15;
  if (xshift < 0) xshift = 0;
  rshift = odo_vector_log_mag(r0, n) - 14;
  if (rshift < 0) rshift = 0;
  r_null = 1;
  for (i = 0; i < n; i++) {
    x16[i] = (int16_t)shr_round(x0[i]*qm[i], Q_QM_SHIFT + xshift);
    r16[i] = (int16_t)shr_round(r0[i]*qm[i], Q_QM_SHIFT + rshift);
    corr += mult16_16(x16[i], r16[i]);
    if (r0[i]) r_null = 0;
  }
  cfl_enabled = is_keyframe && pli != 0;
  cg = odo_pvq_compute_gain(x16, n, q0, &g, beta, xshift);
  cgr = odo_pvq_compute_gain(r16, n, q0, &gr, beta, rshift);
  if (cfl_enabled) cgr = 256;
  icgr = shr_round(cgr, Q_CGAIN_SHIFT);
  gain_offset = cgr - shl32(icgr, Q_CGAIN_SHIFT);
  qg = 0;
  dist = gain_weight*cg*cg*cgain_scale_2;
  best_dist = dist;
  best_cost = dist + pvq_norm_lambda*odo_pvq_rate_fast(0, 0, -1, 0, NULL, 0,
   n, is_keyframe, pli);
  noref = 1;
  best_k = 0;
  *itheta = -1;
  *max_theta = 0;
  memset(y, 0, n*sizeof(*y));
  best_qtheta = 0;
  m = 0;
  s = 1;
  theta = 0;
  corr = corr/(1e-100 + g*(double)gr/shl32(1, xshift + rshift));
  corr = corr < 1. ? corr : 1.;
  corr = corr > -1. ? corr : -1.;
  if (is_keyframe) skip_dist = gain_weight*cg*cg*cgain_scale_2;
  else {
    skip_dist = gain_weight*(cg - cgr)*(cg - cgr) + cgr*(double)cg*(2 - 2*corr);
    skip_dist *= cgain_scale_2;
  }
  if (!is_keyframe) {
    int32_t scgr;
    scgr = gain_offset > 0 ? gain_offset : 0;
    if (icgr == 0) {
      best_dist = gain_weight*(cg - scgr)*(cg - scgr)
       + scgr*(double)cg*(2 - 2*corr);
      best_dist *= cgain_scale_2;
    }
    best_cost = best_dist + pvq_norm_lambda*odo_pvq_rate_fast(0, icgr, 0, 0,
     NULL, 0, n, is_keyframe, pli);
    best_qtheta = 0;
    *itheta = 0;
    *max_theta = 0;
    noref = 0;
  }
  dist0 = best_dist;
  if (trace) {
    memset(trace, 0, sizeof(*trace));
    trace->xshift = xshift;
    trace->rshift = rshift;
    trace->g = g;
    trace->gr = gr;
    trace->cg = cg;
    trace->cgr = cgr;
    trace->icgr = icgr;
    trace->gain_offset = gain_offset;
    trace->corr = corr;
    trace->dist0 = dist0;
    trace->skip_dist = skip_dist;
    memcpy(trace->x16, x16, n*sizeof(*x16));
  }
  if (n <= ODO_MAX_PVQ_SIZE && !r_null && corr > 0) {
    int16_t xr[ODO_MAX_PVQ_SIZE];
    odo_item items[24];
    int gain_bound;
    int prev_k;
    int nitems;
    int idx;
    double cos_dist;
    nitems = 0;
    gain_bound = (cg - gain_offset) >> Q_CGAIN_SHIFT;
    /* OD_THETA_SCALE = 2^15*2/pi, src/pvq.h:78. */
    theta = (int32_t)floor(.5 + (32768*2./M_PI)*acos(corr));
    m = odo_compute_householder(r16, n, gr, &s, rshift);
    odo_apply_householder(xr, x16, r16, n);
    prev_k = 0;
    for (i = m; i < n - 1; i++) xr[i] = xr[i + 1];
    for (i = gain_bound - 1 > 1 ? gain_bound - 1 : 1; i <= gain_bound + 1; i++) {
      int j;
      int32_t qcg;
      int ts;
      int theta_lower;
      int theta_upper;
      qcg = shl32(i, Q_CGAIN_SHIFT) + gain_offset;
      ts = odo_pvq_compute_max_theta(qcg, beta);
      /* OD_THETA_SCALE_1 = 1./OD_THETA_SCALE; same left-to-right products as
         src/pvq_encoder.c:482-484. */
      theta_lower = (int)floor(.5 + theta*(1./(32768*2./M_PI))*2/M_PI*ts) - 2;
      if (theta_lower < 0) theta_lower = 0;
      theta_upper = (int)ceil(theta*(1./(32768*2./M_PI))*2/M_PI*ts);
      if (theta_upper > ts - 1) theta_upper = ts - 1;
      for (j = theta_lower; j <= theta_upper; j++) {
        int32_t qtheta;
        qtheta = odo_pvq_compute_theta(j, ts);
        items[nitems].gain = i;
        items[nitems].theta = j;
        items[nitems].k = odo_pvq_compute_k(qcg, j, qtheta, 0, n, beta, nodesync);
        items[nitems].qcg = qcg;
        items[nitems].qtheta = qtheta;
        items[nitems].ts = ts;
        nitems++;
      }
    }
    cos_dist = 0;
    /* qsort with items_compare (src/pvq_encoder.c:301-305,504): glibc's
       qsort is a stable merge sort at this size, so ties in (k, gain) keep
       enumeration order.  Stable insertion sort here. */
    for (i = 1; i < nitems; i++) {
      odo_item t;
      int j;
      t = items[i];
      for (j = i; j > 0; j--) {
        int c;
        c = items[j - 1].k == t.k ? items[j - 1].gain - t.gain
         : items[j - 1].k - t.k;
        if (c <= 0) break;
        items[j] = items[j - 1];
      }
      items[j] = t;
    }
    for (idx = 0; idx < nitems; idx++) {
      int j;
      int ts;
      int32_t qcg;
      int32_t qtheta;
      double cost;
      double dist_theta;
      double sin_prod;
      odo_pvq_cand *cand;
      qcg = items[idx].qcg;
      i = items[idx].gain;
      j = items[idx].theta;
      ts = items[idx].ts;
      qtheta = items[idx].qtheta;
      k = items[idx].k;
      cand = NULL;
      if (trace && trace->ncands < ODO_MAX_CANDS) {
        cand = &trace->cands[trace->ncands++];
        cand->with_ref = 1;
        cand->gain = i;
        cand->theta = j;
        cand->ts = ts;
        cand->k = k;
        cand->qcg = qcg;
        cand->qtheta = qtheta;
      }
      dist_theta = 2 - 2.*odo_pvq_cos(theta - qtheta)*(1./32768);
      dist = gain_weight*(qcg - cg)*(qcg - cg) + qcg*(double)cg*dist_theta;
      dist *= cgain_scale_2;
      if (dist > dist0 + 1.0*pvq_norm_lambda && k != 0) continue;
      sin_prod = odo_pvq_sin(theta)*(1./32768)*odo_pvq_sin(qtheta)*(1./32768);
      if (k == 0) {
        cos_dist = 0;
        memset(y_tmp, 0, (n - 1)*sizeof(*y_tmp));
      }
      else if (k != prev_k) {
        cos_dist = odo_pvq_search_rdo_double(xr, n - 1, k, y_tmp,
         qcg*(double)cg*sin_prod*cgain_scale_2, pvq_norm_lambda, prev_k);
      }
      prev_k = k;
      dist_theta = 2 - 2.*odo_pvq_cos(theta - qtheta)*(1./32768)
       + sin_prod*(2 - 2*cos_dist);
      dist = gain_weight*(qcg - cg)*(qcg - cg) + qcg*(double)cg*dist_theta;
      dist *= cgain_scale_2;
      if (cand) {
        cand->searched = 1;
        cand->cos_dist = cos_dist;
        cand->dist = dist;
        memcpy(cand->y, y_tmp, (n - 1)*sizeof(*y_tmp));
      }
      cost = dist + pvq_norm_lambda*odo_pvq_rate_fast(i, icgr, j, ts, y_tmp,
       k, n, is_keyframe, pli);
      if (cost < best_cost) {
        best_cost = cost;
        best_dist = dist;
        qg = i;
        best_k = k;
        best_qtheta = qtheta;
        *itheta = j;
        *max_theta = ts;
        noref = 0;
        memcpy(y, y_tmp, (n - 1)*sizeof(*y));
      }
    }
  }
  if (n <= ODO_MAX_PVQ_SIZE && ((is_keyframe && pli == 0) || corr < .5
   || cg < shl32(2, Q_CGAIN_SHIFT))) {
    int gain_bound;
    int prev_k;
    gain_bound = cg >> Q_CGAIN_SHIFT;
    prev_k = 0;
    for (i = gain_bound > 1 ? gain_bound : 1; i <= gain_bound + 1; i++) {
      double cos_dist;
      double cost;
      int32_t qcg;
      odo_pvq_cand *cand;
      qcg = shl32(i, Q_CGAIN_SHIFT);
      k = odo_pvq_compute_k(qcg, -1, -1, 1, n, beta, nodesync);
      cand = NULL;
      if (trace && trace->ncands < ODO_MAX_CANDS) {
        cand = &trace->cands[trace->ncands++];
        cand->with_ref = 0;
        cand->gain = i;
        cand->theta = -1;
        cand->k = k;
        cand->qcg = qcg;
      }
      dist = gain_weight*(qcg - cg)*(qcg - cg);
      dist *= cgain_scale_2;
      if (dist > dist0 && k != 0) continue;
      cos_dist = odo_pvq_search_rdo_double(x16, n, k, y_tmp,
       qcg*(double)cg*cgain_scale_2, pvq_norm_lambda, prev_k);
      prev_k = k;
      dist = gain_weight*(qcg - cg)*(qcg - cg) + qcg*(double)cg*(2 - 2*cos_dist);
      dist *= cgain_scale_2;
      if (cand) {
        cand->searched = 1;
        cand->cos_dist = cos_dist;
        cand->dist = dist;
        memcpy(cand->y, y_tmp, n*sizeof(*y_tmp));
      }
      cost = dist + pvq_norm_lambda*odo_pvq_rate_fast(i, 0, -1, 0, y_tmp, k, n,
       is_keyframe, pli);
      if (cost <= best_cost) {
        best_cost = cost;
        best_dist = dist;
        qg = i;
        noref = 1;
        best_k = k;
        *itheta = -1;
        *max_theta = 0;
        memcpy(y, y_tmp, n*sizeof(*y));
      }
    }
  }
  k = best_k;
  theta = best_qtheta;
  skip = 0;
  if (noref) {
    if (qg == 0) skip = 1; /* OD_PVQ_SKIP_ZERO */
  }
  else {
    if (!is_keyframe && qg == 0) skip = icgr ? 1 : 2;
    if (qg == icgr && *itheta == 0 && !cfl_enabled) skip = 2; /* _SKIP_COPY */
  }
  if (skip) {
    if (skip == 2) memcpy(out, r0, n*sizeof(*out));
    else memset(out, 0, n*sizeof(*out));
  }
  else {
    if (noref) gain_offset = 0;
    g = odo_gain_expand(shl32(qg, Q_CGAIN_SHIFT) + gain_offset, q0, beta);
    odo_pvq_synthesis_partial(out, y, r16, n, noref, g, theta, m, s, qm_inv);
  }
  if (trace) {
    trace->m = m;
    trace->s = s;
    trace->theta = theta;
    memcpy(trace->r16, r16, n*sizeof(*r16));
  }
  *vk = k;
  *skip_diff += skip_dist - best_dist;
  if (is_keyframe) return noref ? qg : odo_neg_interleave(qg, icgr);
  return noref ? qg - 1 : odo_neg_interleave(qg + 1, icgr + 1);
}

/* ======================================================================== */
/* Deringing filter, src/dering.c                    */
/* ======================================================================== */

#define ODO_FILT_BORDER 3                       /* OD_FILT_BORDER, src/dering.h:45 */
#define ODO_FILT_BSTRIDE (64 + 2*ODO_FILT_BORDER)  /* OD_FILT_BSTRIDE, :46 */
#define ODO_DERING_VERY_LARGE 30000             /* src/dering.c:127 */

/* OD_DIRECTION_OFFSETS_TABLE, src/dering.c:39-48, as (dy, dx) steps k = 1..3
   along direction d: 0 = 45 degrees up-right, 2 = horizontal, 6 = vertical. */
static const signed char ODO_DIR_STEP[8][3][2] = {
  {{-1, 1}, {-2, 2}, {-3, 3}},
  {{0, 1}, {-1, 2}, {-1, 3}},
  {{0, 1}, {0, 2}, {0, 3}},
  {{0, 1}, {1, 2}, {1, 3}},
  {{1, 1}, {2, 2}, {3, 3}},
  {{1, 0}, {2, 1}, {3, 1}},
  {{1, 0}, {2, 0}, {3, 0}},
  {{1, 0}, {2, -1}, {3, -1}}
};

/* od_dir_find8, src/dering.c:61-124: the direction whose lines best predict
   the 8x8 block (largest sum of squared line sums, each weighted by 840/n for
   a line of n pixels); *var = (best - orthogonal) >> 10. */
int odo_dir_find8(const int16_t *img, int stride, int32_t *var, int coeff_shift) {
  static const int DIV[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
  int32_t cost[8];
  int partial[8][15];
  int32_t best_cost;
  int best_dir;
  int i;
  int j;
  memset(cost, 0, sizeof(cost));
  memset(partial, 0, sizeof(partial));
  for (i = 0; i < 8; i++) {
    for (j = 0; j < 8; j++) {
      int x;
      x = img[i*stride + j] >> coeff_shift;
      partial[0][i + j] += x;
      partial[1][i + j/2] += x;
      partial[2][i] += x;
      partial[3][3 + i - j/2] += x;
      partial[4][7 + i - j] += x;
      partial[5][3 - i/2 + j] += x;
      partial[6][j] += x;
      partial[7][i/2 + j] += x;
    }
  }
  for (i = 0; i < 8; i++) {
    cost[2] += partial[2][i]*partial[2][i];
    cost[6] += partial[6][i]*partial[6][i];
  }
  cost[2] *= DIV[8];
  cost[6] *= DIV[8];
  for (i = 0; i < 7; i++) {
    cost[0] += (partial[0][i]*partial[0][i] + partial[0][14 - i]*partial[0][14 - i])*DIV[i + 1];
    cost[4] += (partial[4][i]*partial[4][i] + partial[4][14 - i]*partial[4][14 - i])*DIV[i + 1];
  }
  cost[0] += partial[0][7]*partial[0][7]*DIV[8];
  cost[4] += partial[4][7]*partial[4][7]*DIV[8];
  for (i = 1; i < 8; i += 2) {
    for (j = 0; j < 5; j++) cost[i] += partial[i][3 + j]*partial[i][3 + j];
    cost[i] *= DIV[8];
    for (j = 0; j < 3; j++) {
      cost[i] += (partial[i][j]*partial[i][j] + partial[i][10 - j]*partial[i][10 - j])*DIV[2*j + 2];
    }
  }
  best_cost = 0;
  best_dir = 0;
  for (i = 0; i < 8; i++) {
    if (cost[i] > best_cost) {
      best_cost = cost[i];
      best_dir = i;
    }
  }
  *var = (best_cost - cost[(best_dir + 4) & 7]) >> 10;
  return best_dir;
}

/* od_filter_dering_direction_c, src/dering.c:132-159: taps {3,2,1} on both
   sides along the direction, a neighbour counts only if it differs from the
   centre by less than the threshold.  int16 arithmetic as in the reference. */
static void odo_dering_direction(int16_t *y, int ystride, const int16_t *in, int ln, int threshold,
 int dir) {
  static const int TAPS[3] = {3, 2, 1};
  int i;
  int j;
  int k;
  for (i = 0; i < 1 << ln; i++) {
    for (j = 0; j < 1 << ln; j++) {
      int16_t sum;
      int16_t xx;
      xx = in[i*ODO_FILT_BSTRIDE + j];
      sum = 0;
      for (k = 0; k < 3; k++) {
        int o;
        int16_t p0;
        int16_t p1;
        o = ODO_DIR_STEP[dir][k][0]*ODO_FILT_BSTRIDE + ODO_DIR_STEP[dir][k][1];
        p0 = (int16_t)(in[i*ODO_FILT_BSTRIDE + j + o] - xx);
        p1 = (int16_t)(in[i*ODO_FILT_BSTRIDE + j - o] - xx);
        if (abs(p0) < threshold) sum = (int16_t)(sum + TAPS[k]*p0);
        if (abs(p1) < threshold) sum = (int16_t)(sum + TAPS[k]*p1);
      }
      y[i*ystride + j] = (int16_t)(xx + ((sum + 8) >> 4));
    }
  }
}

/* od_filter_dering_orthogonal_c, src/dering.c:172-208. */
static void odo_dering_orthogonal(int16_t *y, int ystride, const int16_t *in, const int16_t *x,
 int xstride, int ln, int threshold, int dir) {
  int i;
  int j;
  int offset;
  offset = dir > 0 && dir < 4 ? ODO_FILT_BSTRIDE : 1;
  for (i = 0; i < 1 << ln; i++) {
    for (j = 0; j < 1 << ln; j++) {
      int16_t athresh;
      int16_t yy;
      int16_t sum;
      int16_t p;
      int t;
      t = threshold/3 + abs(in[i*ODO_FILT_BSTRIDE + j] - x[i*xstride + j]);
      athresh = (int16_t)(threshold < t ? threshold : t);
      yy = in[i*ODO_FILT_BSTRIDE + j];
      sum = 0;
      p = (int16_t)(in[i*ODO_FILT_BSTRIDE + j + offset] - yy);
      if (abs(p) < athresh) sum = (int16_t)(sum + p);
      p = (int16_t)(in[i*ODO_FILT_BSTRIDE + j - offset] - yy);
      if (abs(p) < athresh) sum = (int16_t)(sum + p);
      p = (int16_t)(in[i*ODO_FILT_BSTRIDE + j + 2*offset] - yy);
      if (abs(p) < athresh) sum = (int16_t)(sum + p);
      p = (int16_t)(in[i*ODO_FILT_BSTRIDE + j - 2*offset] - yy);
      if (abs(p) < athresh) sum = (int16_t)(sum + p);
      y[i*ystride + j] = (int16_t)(yy + ((3*sum + 8) >> 4));
    }
  }
}

/* OD_THRESH_TABLE_Q8, src/dering.c:225-229 (x^0.16, index = ilog2). */
static const int16_t ODO_THRESH_TABLE_Q8[18] = {
  128, 134, 150, 168, 188, 210, 234, 262, 292, 327, 365, 408, 455, 509, 569, 635, 710, 768};

/* od_dering, src/dering.c:252-349 (DAALA_ODINTRIN build): one superblock of
   nhb x nvb blocks of side 8 >> xdec.  y: [nvb << bsize][ystride]; x points at
   the superblock inside its plane; dir is written for pli == 0 and read
   otherwise. */
void odo_dering(int16_t *y, int ystride, const int16_t *x, int xstride, int nhb, int nvb, int sbx,
 int sby, int nhsb, int nvsb, int xdec, int dir[8][8], int pli, const unsigned char *bskip,
 int skip_stride, int threshold, int overlap, int coeff_shift) {
  int16_t inbuf[ODO_FILT_BSTRIDE*ODO_FILT_BSTRIDE];
  int16_t *in;
  int32_t var[8][8];
  int thresh[8][8];
  int bsize;
  int i;
  int j;
  int bx;
  int by;
  bsize = 3 - xdec;
  in = inbuf + ODO_FILT_BORDER*ODO_FILT_BSTRIDE + ODO_FILT_BORDER;
  for (i = 0; i < ODO_FILT_BSTRIDE*ODO_FILT_BSTRIDE; i++) inbuf[i] = ODO_DERING_VERY_LARGE;
  for (i = -ODO_FILT_BORDER*(sby != 0); i < (nvb << bsize) + ODO_FILT_BORDER*(sby != nvsb - 1); i++) {
    for (j = -ODO_FILT_BORDER*(sbx != 0); j < (nhb << bsize) + ODO_FILT_BORDER*(sbx != nhsb - 1); j++) {
      in[i*ODO_FILT_BSTRIDE + j] = x[i*xstride + j];
    }
  }
  if (pli == 0) {
    for (by = 0; by < nvb; by++) {
      for (bx = 0; bx < nhb; bx++) {
        int v1;
        dir[by][bx] = odo_dir_find8(&x[8*by*xstride + 8*bx], xstride, &var[by][bx], coeff_shift);
        /* od_compute_thresh, src/dering.c:237-250 */
        v1 = var[by][bx] >> 6;
        if (v1 > 32767) v1 = 32767;
        thresh[by][bx] = (threshold*ODO_THRESH_TABLE_Q8[odo_ilog(v1)] + 128) >> 8;
      }
    }
  }
  else {
    for (by = 0; by < nvb; by++) for (bx = 0; bx < nhb; bx++) thresh[by][bx] = threshold;
  }
  for (by = 0; by < nvb; by++) {
    for (bx = 0; bx < nhb; bx++) {
      int skip;
      int xstart;
      int ystart;
      int xend;
      int yend;
      xstart = ystart = 0;
      xend = yend = 2 >> xdec;
      if (overlap) {
        xstart -= sbx != 0;
        ystart -= sby != 0;
        xend += sbx != nhsb - 1;
        yend += sby != nvsb - 1;
      }
      skip = 1;
      for (i = ystart; i < yend; i++) {
        for (j = xstart; j < xend; j++) {
          skip = skip && bskip[((by << 1 >> xdec) + i)*skip_stride + (bx << 1 >> xdec) + j];
        }
      }
      if (skip) thresh[by][bx] = 0;
    }
  }
  for (by = 0; by < nvb; by++) {
    for (bx = 0; bx < nhb; bx++) {
      odo_dering_direction(&y[(by*ystride << bsize) + (bx << bsize)], ystride,
       &in[(by*ODO_FILT_BSTRIDE << bsize) + (bx << bsize)], bsize, thresh[by][bx], dir[by][bx]);
    }
  }
  for (i = 0; i < nvb << bsize; i++) {
    for (j = 0; j < nhb << bsize; j++) in[i*ODO_FILT_BSTRIDE + j] = y[i*ystride + j];
  }
  for (by = 0; by < nvb; by++) {
    for (bx = 0; bx < nhb; bx++) {
      odo_dering_orthogonal(&y[(by*ystride << bsize) + (bx << bsize)], ystride,
       &in[(by*ODO_FILT_BSTRIDE << bsize) + (bx << bsize)],
       &x[(by*xstride << bsize) + (bx << bsize)], xstride, bsize, thresh[by][bx], dir[by][bx]);
    }
  }
}

/* Every superblock of a plane (the loop of src/encode.c:2709-2843 without the
   level decision): plane of (nhsb*64 >> xdec) x (nvsb*64 >> xdec) samples,
   thresholds[nvsb*nhsb] per superblock, dirs[nvsb*8][nhsb*8] written for
   pli == 0 and read otherwise; y is a whole plane with the same stride. */
void odo_dering_plane(int16_t *y, const int16_t *x, int stride, int nhsb, int nvsb, int xdec,
 int32_t *dirs, int pli, const unsigned char *bskip, int skip_stride, const int32_t *thresholds,
 int overlap, int coeff_shift) {
  int sbx;
  int sby;
  int ln;
  ln = 6 - xdec;
  for (sby = 0; sby < nvsb; sby++) {
    for (sbx = 0; sbx < nhsb; sbx++) {
      int dir[8][8];
      int bx;
      int by;
      for (by = 0; by < 8; by++) for (bx = 0; bx < 8; bx++) dir[by][bx] = dirs[(sby*8 + by)*nhsb*8 + sbx*8 + bx];
      odo_dering(y + ((long)(sby << ln)*stride + (sbx << ln)), stride,
       x + ((long)(sby << ln)*stride + (sbx << ln)), stride, 8, 8, sbx, sby, nhsb, nvsb, xdec, dir, pli,
       bskip + (sby << (4 - xdec))*skip_stride + (sbx << (4 - xdec)), skip_stride,
       thresholds[sby*nhsb + sbx], overlap, coeff_shift);
      for (by = 0; by < 8; by++) for (bx = 0; bx < 8; bx++) dirs[(sby*8 + by)*nhsb*8 + sbx*8 + bx] = dir[by][bx];
    }
  }
}

double odo_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9*ts.tv_nsec;
}

/* The whole hot-path stage on one plane with the oracle's functions: the CPU
   "port" timed by bench.py when oracle/_ref is not available, and the
   end-to-end checker of the GPU step.  Mirrors ref_stage_plane in ref_shim.c.
   The choice between candidates uses `cost <= best_cost` with cost = dist when
   rate_mode == 0 (what the GPU bench step does without a host entropy model)
   or the closed-form rate of od_pvq_rate (speed > 0) when rate_mode == 1. */
long odo_stage_plane(const uint8_t *px, int px_stride, int w, int h, int dec,
 int pic_w, int pic_h, int pli, const int16_t *qm, const int16_t *qm_inv,
 const int *qm_off, const int *q_band, const int *beta_band,
 double pvq_norm_lambda, int rate_mode, uint8_t *recon_px) {
  odo_coeff *levels[ODO_NBSIZES];
  odo_coeff *c;
  odo_coeff *dq;
  long nblocks;
  int top;
  int bs;
  top = ODO_NBSIZES - 1 - dec;
  nblocks = 0;
  c = (odo_coeff *)malloc(sizeof(*c)*w*h);
  dq = (odo_coeff *)malloc(sizeof(*dq)*w*h);
  for (bs = 0; bs <= top; bs++) levels[bs] = (odo_coeff *)malloc(sizeof(*c)*w*h);
  odo_forward_pyramid_plane(levels, c, px, px_stride, w, h, dec, pic_w, pic_h);
  for (bs = 0; bs <= top; bs++) {
    int n;
    int bx;
    int by;
    int nb;
    const int *off;
    n = 4 << bs;
    nb = OD_NBANDS[bs];
    off = OD_BAND_OFFS[bs];
    memset(dq, 0, sizeof(*dq)*w*h);
    for (by = 0; by < h/n; by++) {
      for (bx = 0; bx < w/n; bx++) {
        odo_coeff in[512];
        odo_coeff out[512];
        odo_coeff ref0[128];
        odo_coeff y[128];
        double skip_diff;
        int i;
        int bo;
        bo = by*n*w + bx*n;
        odo_raster_to_coding_order(in, n, levels[bs] + bo, w);
        memset(ref0, 0, sizeof(ref0));
        skip_diff = 0;
        for (i = 0; i < nb; i++) {
          int m;
          int itheta;
          int max_theta;
          int k;
          m = off[i + 1] - off[i];
          if (rate_mode) {
            odo_pvq_theta(out + off[i], in + off[i], ref0, m, q_band[bs*12 + i], y,
             &itheta, &max_theta, &k, beta_band[bs*12 + i], &skip_diff, 1, 1, pli,
             qm + qm_off[bs] + off[i], qm_inv + qm_off[bs] + off[i],
             pvq_norm_lambda, 1, NULL);
          }
          else {
            /* distortion-only choice from the candidate trace */
            static odo_pvq_band_trace tr;
            double best;
            int qg;
            int sel;
            int j;
            odo_pvq_theta(out + off[i], in + off[i], ref0, m, q_band[bs*12 + i], y,
             &itheta, &max_theta, &k, beta_band[bs*12 + i], &skip_diff, 1, 1, 0,
             qm + qm_off[bs] + off[i], qm_inv + qm_off[bs] + off[i],
             pvq_norm_lambda, 1, &tr);
            best = tr.dist0;
            qg = 0;
            sel = -1;
            for (j = 0; j < tr.ncands; j++) {
              if (tr.cands[j].with_ref || !tr.cands[j].searched) continue;
              if (tr.cands[j].dist <= best) {
                best = tr.cands[j].dist;
                qg = tr.cands[j].gain;
                sel = j;
              }
            }
            if (qg == 0) memset(out + off[i], 0, m*sizeof(*out));
            else {
              odo_pvq_synthesis_partial(out + off[i], tr.cands[sel].y, NULL, m, 1,
               odo_gain_expand(qg << 8, q_band[bs*12 + i], beta_band[bs*12 + i]),
               0, 0, 1, qm_inv + qm_off[bs] + off[i]);
            }
          }
        }
        out[0] = in[0];
        odo_coding_order_to_raster(dq + bo, w, out, n);
        nblocks++;
      }
    }
    odo_inverse_level_plane(recon_px, w, c, dq, w, h, dec, bs, pic_w, pic_h);
  }
  for (bs = 0; bs <= top; bs++) free(levels[bs]);
  free(dq);
  free(c);
  return nblocks;
}

/* ---- od_compute_dist, src/encode.c:1082-1226 -------------
   The block-size RDO's distortion between a source block x and a reconstruction y,
   both n x n in the lapped domain, n = 8..64.  HVS matrices (the default): error
   low-passed by [1 5 1]/7 in both directions, then per 8x8 block an activity-masked
   sum of the filtered error energy and a variance-difference term; flat matrices:
   plain squared error.  libm: sqrt (exact) and pow (host libm, as in the reference). */
static int odo_var_4x4(const odo_coeff *x, int stride) {   /* :1082-1103 */
  int sum;
  int s2;
  int i;
  int j;
  sum = 0;
  s2 = 0;
  for (i = 0; i < 4; i++) {
    for (j = 0; j < 4; j++) {
      int t;
      t = x[i*stride + j] >> 2;
      sum += t;
      s2 += t*t;
    }
  }
  return s2 - (sum*sum >> 4);
}

/* The libm-free part of od_compute_dist_8x8 (:1113-1169): parts[0] = sum of squared
   low-passed error, parts[1] = vardist, parts[2] = the argument of pow. */
void odo_dist_8x8_parts(double parts[3], const odo_coeff *x, const odo_coeff *y,
 const odo_coeff *e_lp, int stride, int use_masking) {
  double sum;
  double mean_var;
  double vardist;
  int min_var;
  int i;
  int j;
  vardist = 0;
  min_var = 0x7fffffff;
  mean_var = 0;
  for (i = 0; i < 3; i++) {
    for (j = 0; j < 3; j++) {
      int varx;
      int vary;
      varx = odo_var_4x4(x + 2*i*stride + 2*j, stride);
      vary = odo_var_4x4(y + 2*i*stride + 2*j, stride);
      min_var = varx < min_var ? varx : min_var;
      mean_var += 1./(1 + varx);
      vardist += varx - 2*sqrt(varx*(double)vary) + vary;
    }
  }
  sum = 0;
  for (i = 0; i < 8; i++) {
    for (j = 0; j < 8; j++) sum += e_lp[i*stride + j]*(double)e_lp[i*stride + j];
  }
  parts[0] = sum;
  parts[1] = vardist;
  parts[2] = .25 + (use_masking ? 9./mean_var : (double)min_var)/(1 << 2*4);
}

static double odo_dist_8x8_finish(const double parts[3], int use_masking) {   /* :1158-1169 */
  double activity;
  double sum;
  activity = (use_masking ? 1.95 : 1.62)*pow(parts[2], -1./6);
  sum = parts[0]*(0.92/(7*7*7*7));
  return activity*activity*(sum + parts[1]);
}

double odo_compute_dist(const odo_coeff *x, const odo_coeff *y, int n, int flat_qm, int use_masking,
 int coded_quantizer) {
  static odo_coeff e[64*64];
  static odo_coeff tmp[64*64];
  static odo_coeff e_lp[64*64];
  double sum;
  int i;
  int j;
  sum = 0;
  if (flat_qm) {
    for (i = 0; i < n*n; i++) {
      double t;
      t = x[i] - y[i];
      sum += t*t;
    }
    return sum;
  }
  for (i = 0; i < n*n; i++) e[i] = x[i] - y[i];
  for (i = 0; i < n; i++) {
    tmp[i*n] = 5*e[i*n] + 2*e[i*n + 1];
    tmp[i*n + n - 1] = 5*e[i*n + n - 1] + 2*e[i*n + n - 2];
    for (j = 1; j < n - 1; j++) tmp[i*n + j] = 5*e[i*n + j] + e[i*n + j - 1] + e[i*n + j + 1];
  }
  for (j = 0; j < n; j++) {
    e_lp[j] = 5*tmp[j] + 2*tmp[n + j];
    e_lp[(n - 1)*n + j] = 5*tmp[(n - 1)*n + j] + 2*tmp[(n - 2)*n + j];
  }
  for (i = 1; i < n - 1; i++) {
    for (j = 0; j < n; j++) e_lp[i*n + j] = 5*tmp[i*n + j] + tmp[(i - 1)*n + j] + tmp[(i + 1)*n + j];
  }
  for (i = 0; i < n; i += 8) {
    for (j = 0; j < n; j += 8) {
      double parts[3];
      odo_dist_8x8_parts(parts, x + i*n + j, y + i*n + j, e_lp + i*n + j, n, use_masking);
      sum += odo_dist_8x8_finish(parts, use_masking);
    }
  }
  sum *= coded_quantizer >= 47 ? 1.2 : coded_quantizer <= 36 ? 1.7
   : 1.7 + (1.2 - 1.7)*(coded_quantizer - 36)/(47 - 36);
  return sum;
}
