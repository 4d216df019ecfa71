/* oracle/ref_encoder_driver.c - TEST INFRASTRUCTURE, not product code.

   Headless driver of the REAL reference encoder through its public API only
   (include/daala/daalaenc.h:75-139), the same call sequence as
   examples/encoder_example.c:974-1099 minus the Ogg container (libogg is not
   part of the codec arithmetic and is not installed here).  Used for the
   end-to-end CPU baseline (seconds per 1080p all-intra frame, packets) and for
   counting the transform blocks the reference evaluates per frame.

   Linked into oracle/_ref/libdaalaref.so together with the unmodified
   reference sources; see oracle/Makefile. */
#include <stdlib.h>
#include <string.h>
#include "daala/codec.h"
#include "daala/daalaenc.h"
#include "state.h"
#include "encint.h"
#include "block_size.h"

#define REF_EXPORT __attribute__((visibility("default")))

/* Call counters wrapped around the reference's C function-pointer tables so
   that "evaluated transform blocks" is measured, not assumed.  Installed by
   ref_encode_yuv420 after daala_encode_create (the vtbl lives in od_state,
   src/state.h:112-131; daala_enc_ctx starts with its od_state,
   src/encint.h). */
static long ref_last_luma_blocks;
static long ref_last_chroma_blocks;
static long ref_fdct_calls[OD_NBSIZES];
static long ref_idct_calls[OD_NBSIZES];
static od_dct_func_2d ref_fdct_real[OD_NBSIZES];
static od_dct_func_2d ref_idct_real[OD_NBSIZES];

#define REF_COUNTED(dir, bs) \
  static void ref_##dir##_counted_##bs(od_coeff *out, int out_stride, \
   const od_coeff *in, int in_stride) { \
    ref_##dir##_calls[bs]++; \
    (*ref_##dir##_real[bs])(out, out_stride, in, in_stride); \
  }
REF_COUNTED(fdct, 0)
REF_COUNTED(fdct, 1)
REF_COUNTED(fdct, 2)
REF_COUNTED(fdct, 3)
REF_COUNTED(fdct, 4)
REF_COUNTED(idct, 0)
REF_COUNTED(idct, 1)
REF_COUNTED(idct, 2)
REF_COUNTED(idct, 3)
REF_COUNTED(idct, 4)

static const od_dct_func_2d REF_FDCT_COUNTED[OD_NBSIZES] = {
  ref_fdct_counted_0, ref_fdct_counted_1, ref_fdct_counted_2,
  ref_fdct_counted_3, ref_fdct_counted_4
};
static const od_dct_func_2d REF_IDCT_COUNTED[OD_NBSIZES] = {
  ref_idct_counted_0, ref_idct_counted_1, ref_idct_counted_2,
  ref_idct_counted_3, ref_idct_counted_4
};

/* Optional external transform tables installed in place of the C ones: this is
   how the drop-in test binds libdaalahip's od_bin_*_hip entry points into the
   UNMODIFIED reference encoder - the same ten slots od_state_opt_vtbl_init_x86
   overwrites (src/x86/x86state.c:66-91).  NULL = keep the reference's. */
static od_dct_func_2d ref_ext_fdct[OD_NBSIZES];
static od_dct_func_2d ref_ext_idct[OD_NBSIZES];

REF_EXPORT void ref_set_external_dct_vtbl(void **fdct, void **idct) {
  int i;
  for (i = 0; i < OD_NBSIZES; i++) {
    ref_ext_fdct[i] = fdct ? (od_dct_func_2d)fdct[i] : NULL;
    ref_ext_idct[i] = idct ? (od_dct_func_2d)idct[i] : NULL;
  }
}

/* Writes external 2-D transform tables into an od_state's opt_vtbl (the slots
   od_state_opt_vtbl_init_x86 overwrites, src/x86/x86state.c:66-91).  Used by
   tests/interpose's od_state_opt_vtbl_init, which does not see the reference's
   headers. */
REF_EXPORT void ref_state_set_dct_vtbl(void *state, void **fdct, void **idct) {
  od_state *st;
  int i;
  st = (od_state *)state;
  for (i = 0; i < OD_NBSIZES; i++) {
    if (fdct && fdct[i]) st->opt_vtbl.fdct_2d[i] = (od_dct_func_2d)fdct[i];
    if (idct && idct[i]) st->opt_vtbl.idct_2d[i] = (od_dct_func_2d)idct[i];
  }
}

REF_EXPORT void ref_get_dct_call_counts(long *fdct, long *idct) {
  int i;
  for (i = 0; i < OD_NBSIZES; i++) {
    fdct[i] = ref_fdct_calls[i];
    idct[i] = ref_idct_calls[i];
  }
}

/* Encodes nframes planar 4:2:0 8-bit frames (Y, Cb, Cr contiguous per frame,
   tightly packed) as all-intra (keyframe_rate = 1).  Packets are appended to
   out[]; pkt_bytes[i] receives the size of data packet i.  Returns the number
   of data packets, or a negative OD_E* code. */
static int ref_encode_core(const unsigned char *frames, int w, int h,
 int nframes, int quality, int complexity, int count_calls,
 unsigned char *out, long out_cap, long *pkt_bytes, const int *global_index);

REF_EXPORT int ref_encode_yuv420(const unsigned char *frames, int w, int h,
 int nframes, int quality, int complexity, int count_calls,
 unsigned char *out, long out_cap, long *pkt_bytes) {
  return ref_encode_core(frames, w, h, nframes, quality, complexity, count_calls, out, out_cap,
   pkt_bytes, NULL);
}

/* A SHARD of an all-intra encode: frame f of this call is frame
   global_index[f] of the whole sequence.  The only per-frame state of an
   all-intra encode that reaches the packet bytes is the display frame number
   coded in the frame header (OD_REORDER_INDEX(curr_display_order),
   src/encode.c:3043), which the input queue takes from its frame counter
   (src/encode.c:309-324); the shard seeds that counter before it queues each of
   its frames, so its packets are the sequential encoder's byte for byte. */
REF_EXPORT int ref_encode_yuv420_shard(const unsigned char *frames, int w, int h,
 int nframes, const int *global_index, int quality, int complexity,
 unsigned char *out, long out_cap, long *pkt_bytes) {
  return ref_encode_core(frames, w, h, nframes, quality, complexity, 0, out, out_cap, pkt_bytes,
   global_index);
}

static int ref_encode_core(const unsigned char *frames, int w, int h,
 int nframes, int quality, int complexity, int count_calls,
 unsigned char *out, long out_cap, long *pkt_bytes, const int *global_index) {
  daala_info di;
  daala_comment dc;
  daala_enc_ctx *enc;
  daala_packet dp;
  daala_image img;
  long out_pos;
  long frame_bytes;
  int npackets;
  int f;
  int i;
  int ret;
  daala_info_init(&di);
  di.pic_width = w;
  di.pic_height = h;
  di.bitdepth_mode = OD_BITDEPTH_MODE_8;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = 1;
  di.pixel_aspect_denominator = 1;
  di.full_precision_references = 0;
  di.nplanes = 3;
  di.plane_info[0].xdec = 0;
  di.plane_info[0].ydec = 0;
  di.plane_info[1].xdec = 1;
  di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = 1;
  di.plane_info[2].ydec = 1;
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  daala_comment_init(&dc);
  daala_encode_ctl(enc, OD_SET_QUANT, &quality, sizeof(quality));
  daala_encode_ctl(enc, OD_SET_COMPLEXITY, &complexity, sizeof(complexity));
  {
    od_state *st;
    st = (od_state *)enc;
    for (i = 0; i < OD_NBSIZES; i++) {
      if (ref_ext_fdct[i]) st->opt_vtbl.fdct_2d[i] = ref_ext_fdct[i];
      if (ref_ext_idct[i]) st->opt_vtbl.idct_2d[i] = ref_ext_idct[i];
    }
  }
  if (count_calls) {
    od_state *state;
    state = (od_state *)enc;
    for (i = 0; i < OD_NBSIZES; i++) {
      ref_fdct_calls[i] = ref_idct_calls[i] = 0;
      ref_fdct_real[i] = state->opt_vtbl.fdct_2d[i];
      ref_idct_real[i] = state->opt_vtbl.idct_2d[i];
      state->opt_vtbl.fdct_2d[i] = REF_FDCT_COUNTED[i];
      state->opt_vtbl.idct_2d[i] = REF_IDCT_COUNTED[i];
    }
  }
  for (;;) {
    ret = daala_encode_flush_header(enc, &dc, &dp);
    if (ret < 0) return ret;
    if (ret == 0) break;
  }
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = w;
  img.height = h;
  frame_bytes = (long)w*h + 2L*((w + 1) >> 1)*((h + 1) >> 1);
  out_pos = 0;
  npackets = 0;
  for (f = 0; f <= nframes; f++) {
    int last;
    last = f == nframes;
    while (daala_encode_packet_out(enc, last, &dp)) {
      if (out_pos + dp.bytes > out_cap) return -2;
      memcpy(out + out_pos, dp.packet, dp.bytes);
      out_pos += dp.bytes;
      pkt_bytes[npackets++] = dp.bytes;
    }
    if (!last) {
      unsigned char *base;
      base = (unsigned char *)frames + f*frame_bytes;
      img.planes[0].data = base;
      img.planes[0].xdec = 0;
      img.planes[0].ydec = 0;
      img.planes[0].xstride = 1;
      img.planes[0].ystride = w;
      img.planes[0].bitdepth = 8;
      img.planes[1].data = base + (long)w*h;
      img.planes[1].xdec = 1;
      img.planes[1].ydec = 1;
      img.planes[1].xstride = 1;
      img.planes[1].ystride = (w + 1) >> 1;
      img.planes[1].bitdepth = 8;
      img.planes[2].data = img.planes[1].data
       + (long)((w + 1) >> 1)*((h + 1) >> 1);
      img.planes[2].xdec = 1;
      img.planes[2].ydec = 1;
      img.planes[2].xstride = 1;
      img.planes[2].ystride = (w + 1) >> 1;
      img.planes[2].bitdepth = 8;
      if (global_index != NULL) {
        /* the queue is empty here (every packet was drained above) */
        ((struct daala_enc_ctx *)enc)->input_queue.frame_number = global_index[f];
      }
      ret = daala_encode_img_in(enc, &img, 0);
      if (ret < 0) return ret;
    }
  }
  {
    /* the final partition of the last frame coded (state.bsize, src/state.h:250-259: one
       entry per 8x8, the size of the block that covers it; 0 = four 4x4 luma blocks over
       ONE 4x4 chroma block): how many transform blocks the bitstream really carries */
    od_state *st;
    int bx;
    int by;
    st = (od_state *)enc;
    ref_last_luma_blocks = ref_last_chroma_blocks = 0;
    for (by = 0; by < st->nvsb*OD_BSIZE_GRID; by++) {
      for (bx = 0; bx < st->nhsb*OD_BSIZE_GRID; bx++) {
        int v;
        v = OD_BLOCK_SIZE8x8(st->bsize, st->bstride, bx, by);
        if (v == 0) {
          ref_last_luma_blocks += 4;
          ref_last_chroma_blocks++;
        }
        else if ((bx & ((1 << (v - 1)) - 1)) == 0 && (by & ((1 << (v - 1)) - 1)) == 0) {
          ref_last_luma_blocks++;
          ref_last_chroma_blocks++;
        }
      }
    }
  }
  daala_comment_clear(&dc);
  daala_encode_free(enc);
  return npackets;
}

/* Transform blocks of the final partition of the last frame the drivers above coded:
   out[0] luma, out[1] one chroma plane of a 4:2:0 picture. */
REF_EXPORT void ref_last_coded_blocks(long out[2]) {
  out[0] = ref_last_luma_blocks;
  out[1] = ref_last_chroma_blocks;
}

/* What the batched band stage needs to know about a live encoder when a frame's
   luma plane is loaded (tests/interpose's od_apply_prefilter_frame_sbs): exactly the
   values od_block_encode / od_pvq_encode hand to pvq_theta for that frame
   (src/encode.c:1336-1359, src/pvq_encoder.c:874).  The tables are copied out;
   returns 1 when the frame being coded is a keyframe coded with PVQ (the only case
   the batch serves), else 0. */
REF_EXPORT int ref_enc_band_setup(const void *encp, int *quantizer, int *use_masking,
 double *pvq_norm_lambda, unsigned char *pvq_qm_q4, int16_t *qm, int16_t *qm_inv) {
  const daala_enc_ctx *enc;
  int pli;
  enc = (const daala_enc_ctx *)encp;
  *quantizer = enc->state.quantizer;
  *use_masking = enc->use_activity_masking;
  *pvq_norm_lambda = enc->pvq_norm_lambda;
  for (pli = 0; pli < 3; pli++) {
    memcpy(pvq_qm_q4 + pli*OD_QM_SIZE, enc->state.pvq_qm_q4[pli], OD_QM_SIZE);
  }
  memcpy(qm, enc->state.qm, OD_QM_BUFFER_SIZE*sizeof(*qm));
  memcpy(qm_inv, enc->state.qm_inv, OD_QM_BUFFER_SIZE*sizeof(*qm_inv));
  return enc->state.frame_type == OD_I_FRAME && !enc->use_haar_wavelet && !OD_LOSSLESS(enc);
}

/* What od_compute_dist reads from the encoder besides its arguments (src/encode.c:1113-1226):
   activity masking on / off and whether the flat quantisation matrices are in use; for
   odhip_dering_cache_set_source (the level search's distortions from the batched passes). */
REF_EXPORT void ref_enc_dist_setup(const void *encp, int *use_masking, int *flat_qm) {
  const daala_enc_ctx *enc;
  enc = (const daala_enc_ctx *)encp;
  *use_masking = enc->use_activity_masking;
  *flat_qm = enc->qm == OD_FLAT_QM;
}

/* Quantiser set-up the reference derives per frame on the host (SURVEY.md
   8(a) row a17): after encoding one frame at `quality`, copies out
   state.quantizer, state.pvq_qm_q4[pli][OD_QM_SIZE] (od_interp_qm,
   src/encode.c:2903-2940,:3052-3072) and state.qm / qm_inv (od_init_qm,
   src/pvq.c:322-381).  Used by tools/make_golden.py to create the fixture the
   GPU tests and bench.py feed to the kernels. */
#include "encint.h"
REF_EXPORT int ref_dump_quant_tables(int quality, int *quantizer,
 unsigned char *pvq_qm_q4, int16_t *qm, int16_t *qm_inv) {
  daala_info di;
  daala_comment dc;
  daala_enc_ctx *enc;
  daala_packet dp;
  daala_image img;
  unsigned char *buf;
  int complexity;
  int pli;
  int ret;
  daala_info_init(&di);
  di.pic_width = 64;
  di.pic_height = 64;
  di.bitdepth_mode = OD_BITDEPTH_MODE_8;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = 1;
  di.pixel_aspect_denominator = 1;
  di.nplanes = 3;
  di.plane_info[0].xdec = di.plane_info[0].ydec = 0;
  di.plane_info[1].xdec = di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = di.plane_info[2].ydec = 1;
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  daala_comment_init(&dc);
  complexity = 2;
  daala_encode_ctl(enc, OD_SET_QUANT, &quality, sizeof(quality));
  daala_encode_ctl(enc, OD_SET_COMPLEXITY, &complexity, sizeof(complexity));
  while ((ret = daala_encode_flush_header(enc, &dc, &dp)) > 0);
  buf = (unsigned char *)malloc(64*64*3/2);
  memset(buf, 128, 64*64*3/2);
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = img.height = 64;
  for (pli = 0; pli < 3; pli++) {
    img.planes[pli].data = buf + (pli == 0 ? 0 : pli == 1 ? 4096 : 5120);
    img.planes[pli].xdec = img.planes[pli].ydec = pli > 0;
    img.planes[pli].xstride = 1;
    img.planes[pli].ystride = pli ? 32 : 64;
    img.planes[pli].bitdepth = 8;
  }
  daala_encode_img_in(enc, &img, 0);
  while (daala_encode_packet_out(enc, 1, &dp));
  *quantizer = enc->state.quantizer;
  for (pli = 0; pli < 3; pli++) {
    memcpy(pvq_qm_q4 + pli*OD_QM_SIZE, enc->state.pvq_qm_q4[pli], OD_QM_SIZE);
  }
  memcpy(qm, enc->state.qm, OD_QM_BUFFER_SIZE*sizeof(*qm));
  memcpy(qm_inv, enc->state.qm_inv, OD_QM_BUFFER_SIZE*sizeof(*qm_inv));
  free(buf);
  daala_comment_clear(&dc);
  daala_encode_free(enc);
  return OD_QM_SIZE;
}

/* ref_dump_quant_tables with the knobs a17 depends on: activity masking on/off
   (OD_SET_MASKING) and flat / HVS matrices (OD_SET_QM); also returns
   rc.base_quantizer, the argument of od_interp_qm (src/encode.c:3056).  qm /
   qm_inv may be NULL. */
REF_EXPORT int ref_dump_quant_tables2(int quality, int use_masking, int hvs_qm, int *quantizer,
 int *base_quantizer, unsigned char *pvq_qm_q4, int16_t *qm, int16_t *qm_inv) {
  daala_info di;
  daala_comment dc;
  daala_enc_ctx *enc;
  daala_packet dp;
  daala_image img;
  unsigned char *buf;
  int complexity;
  int pli;
  int ret;
  daala_info_init(&di);
  di.pic_width = 64;
  di.pic_height = 64;
  di.bitdepth_mode = OD_BITDEPTH_MODE_8;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = 1;
  di.pixel_aspect_denominator = 1;
  di.nplanes = 3;
  di.plane_info[0].xdec = di.plane_info[0].ydec = 0;
  di.plane_info[1].xdec = di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = di.plane_info[2].ydec = 1;
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  daala_comment_init(&dc);
  complexity = 2;
  daala_encode_ctl(enc, OD_SET_QUANT, &quality, sizeof(quality));
  daala_encode_ctl(enc, OD_SET_COMPLEXITY, &complexity, sizeof(complexity));
  daala_encode_ctl(enc, OD_SET_ACTIVITY_MASKING, &use_masking, sizeof(use_masking));
  daala_encode_ctl(enc, OD_SET_QM, &hvs_qm, sizeof(hvs_qm));
  while ((ret = daala_encode_flush_header(enc, &dc, &dp)) > 0);
  buf = (unsigned char *)malloc(64*64*3/2);
  memset(buf, 128, 64*64*3/2);
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = img.height = 64;
  for (pli = 0; pli < 3; pli++) {
    img.planes[pli].data = buf + (pli == 0 ? 0 : pli == 1 ? 4096 : 5120);
    img.planes[pli].xdec = img.planes[pli].ydec = pli > 0;
    img.planes[pli].xstride = 1;
    img.planes[pli].ystride = pli ? 32 : 64;
    img.planes[pli].bitdepth = 8;
  }
  daala_encode_img_in(enc, &img, 0);
  while (daala_encode_packet_out(enc, 1, &dp));
  *quantizer = enc->state.quantizer;
  *base_quantizer = enc->rc.base_quantizer;
  for (pli = 0; pli < 3; pli++) {
    memcpy(pvq_qm_q4 + pli*OD_QM_SIZE, enc->state.pvq_qm_q4[pli], OD_QM_SIZE);
  }
  if (qm) memcpy(qm, enc->state.qm, OD_QM_BUFFER_SIZE*sizeof(*qm));
  if (qm_inv) memcpy(qm_inv, enc->state.qm_inv, OD_QM_BUFFER_SIZE*sizeof(*qm_inv));
  free(buf);
  daala_comment_clear(&dc);
  daala_encode_free(enc);
  return OD_QM_SIZE;
}

/* One block through the REAL od_pvq_encode (src/pvq_encoder.c:789-979) on a
   fresh encoder context: range coder and adaptation state reset as at the
   start of a frame (src/encode.c:3029,3080).  `ref` is mutated exactly as the
   reference mutates it (the chroma-from-luma sign flip, :846-872), which is
   what tests/test_oracle_golden.py pins odo_cfl_flip against; `out` receives the
   dequantised block, *flip_out whether ref was negated.  Returns the skip flag.
   ref/in/out: one block in coding order; qm/qm_inv: that block size's tables. */
#include "pvq_encoder.h"
REF_EXPORT int ref_pvq_encode_block(od_coeff *ref, const od_coeff *in, od_coeff *out,
 int q0, int pli, int bs, const int *beta_band, int is_keyframe, const int16_t *qm,
 const int16_t *qm_inv, int speed) {
  daala_info di;
  daala_enc_ctx *enc;
  od_val16 beta[16];
  int i;
  int ret;
  daala_info_init(&di);
  di.pic_width = 64;
  di.pic_height = 64;
  di.bitdepth_mode = OD_BITDEPTH_MODE_8;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = 1;
  di.pixel_aspect_denominator = 1;
  di.nplanes = 3;
  di.plane_info[1].xdec = di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = di.plane_info[2].ydec = 1;
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  od_ec_enc_reset(&enc->ec);
  od_adapt_ctx_reset(&enc->state.adapt, is_keyframe);
  for (i = 0; i < OD_QM_SIZE; i++) enc->state.pvq_qm_q4[pli][i] = 16;
  for (i = 0; i < 16; i++) beta[i] = (od_val16)beta_band[i < 12 ? i : 11];
  ret = od_pvq_encode(enc, ref, in, out, q0, pli, bs, beta, 1, is_keyframe, 0, 0, 0,
   qm, qm_inv, speed);
  daala_encode_free(enc);
  return ret;
}

/* The padded input image the REAL encoder keeps for a picture of w x h
   (daala_image_copy_pad -> od_img_plane_copy_pad, src/encode.c:752-837,:1896-1909):
   daala_encode_img_in queues the frame (src/encode.c:3216-3219,
   od_input_queue_add :272-287), nothing is encoded.  planes[pli] receives the
   plane_w x plane_h region (frame_width >> xdec by frame_height >> ydec, tightly
   packed); dims[2*pli], dims[2*pli + 1] its size.  Returns 0 or a negative
   OD_E* code. */
REF_EXPORT int ref_image_copy_pad(const unsigned char *frame, int w, int h,
 unsigned char *const planes[3], int *dims) {
  daala_info di;
  daala_enc_ctx *enc;
  daala_image img;
  daala_image *pad;
  int pli;
  int ret;
  daala_info_init(&di);
  di.pic_width = w;
  di.pic_height = h;
  di.bitdepth_mode = OD_BITDEPTH_MODE_8;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = 1;
  di.pixel_aspect_denominator = 1;
  di.nplanes = 3;
  di.plane_info[1].xdec = di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = di.plane_info[2].ydec = 1;
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = w;
  img.height = h;
  img.planes[0].data = (unsigned char *)frame;
  img.planes[0].xstride = 1;
  img.planes[0].ystride = w;
  img.planes[0].bitdepth = 8;
  for (pli = 1; pli < 3; pli++) {
    img.planes[pli].data = (unsigned char *)frame + (long)w*h
     + (pli - 1)*(long)((w + 1) >> 1)*((h + 1) >> 1);
    img.planes[pli].xdec = 1;
    img.planes[pli].ydec = 1;
    img.planes[pli].xstride = 1;
    img.planes[pli].ystride = (w + 1) >> 1;
    img.planes[pli].bitdepth = 8;
  }
  ret = daala_encode_img_in(enc, &img, 0);
  if (ret < 0) {
    daala_encode_free(enc);
    return ret;
  }
  pad = &enc->input_queue.images[enc->input_queue.input_head];
  for (pli = 0; pli < 3; pli++) {
    int pw;
    int ph;
    int y;
    pw = pad->width >> pad->planes[pli].xdec;
    ph = pad->height >> pad->planes[pli].ydec;
    dims[2*pli] = pw;
    dims[2*pli + 1] = ph;
    for (y = 0; y < ph; y++) {
      memcpy(planes[pli] + (long)y*pw, pad->planes[pli].data + (long)y*pad->planes[pli].ystride, pw);
    }
  }
  daala_encode_free(enc);
  return 0;
}

/* The same for an encoder with full_precision_references = 1 (16-bit picture buffers at
   12 bits, src/encode.c:212-213): the source is 4:2:0 at `bitdepth` 8 (uint8_t samples) or
   10 / 12 (uint16_t samples, little endian in memory); planes[pli] receives uint16_t
   samples, tightly packed. */
REF_EXPORT int ref_image_copy_pad_fpr(const unsigned char *frame, int w, int h, int bitdepth,
 uint16_t *const planes[3], int *dims) {
  daala_info di;
  daala_enc_ctx *enc;
  daala_image img;
  daala_image *pad;
  int pli;
  int ret;
  int bytes;
  bytes = bitdepth > 8 ? 2 : 1;
  daala_info_init(&di);
  di.pic_width = w;
  di.pic_height = h;
  di.bitdepth_mode = bitdepth == 12 ? OD_BITDEPTH_MODE_12 : bitdepth == 10 ? OD_BITDEPTH_MODE_10
   : OD_BITDEPTH_MODE_8;
  di.full_precision_references = 1;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = 1;
  di.pixel_aspect_denominator = 1;
  di.nplanes = 3;
  di.plane_info[1].xdec = di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = di.plane_info[2].ydec = 1;
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = w;
  img.height = h;
  img.planes[0].data = (unsigned char *)frame;
  img.planes[0].xstride = bytes;
  img.planes[0].ystride = w*bytes;
  img.planes[0].bitdepth = bitdepth;
  for (pli = 1; pli < 3; pli++) {
    img.planes[pli].data = (unsigned char *)frame + ((long)w*h
     + (pli - 1)*(long)((w + 1) >> 1)*((h + 1) >> 1))*bytes;
    img.planes[pli].xdec = 1;
    img.planes[pli].ydec = 1;
    img.planes[pli].xstride = bytes;
    img.planes[pli].ystride = ((w + 1) >> 1)*bytes;
    img.planes[pli].bitdepth = bitdepth;
  }
  ret = daala_encode_img_in(enc, &img, 0);
  if (ret < 0) {
    daala_encode_free(enc);
    return ret;
  }
  pad = &enc->input_queue.images[enc->input_queue.input_head];
  for (pli = 0; pli < 3; pli++) {
    int pw;
    int ph;
    int y;
    if (pad->planes[pli].xstride != 2) {
      daala_encode_free(enc);
      return -2;
    }
    pw = pad->width >> pad->planes[pli].xdec;
    ph = pad->height >> pad->planes[pli].ydec;
    dims[2*pli] = pw;
    dims[2*pli + 1] = ph;
    for (y = 0; y < ph; y++) {
      memcpy(planes[pli] + (long)y*pw, pad->planes[pli].data + (long)y*pad->planes[pli].ystride, pw*2);
    }
  }
  daala_encode_free(enc);
  return 0;
}

/* ---- decoder side ------------------------------------------------------------------
   ref_roundtrip_yuv420: encodes like ref_encode_yuv420 but keeps the three header
   packets, then feeds headers and data packets to the REAL reference decoder
   (daala_decode_header_in / _create / _packet_in / _img_out, include/daala/daaladec.h)
   and copies every decoded picture (planar 4:2:0, tightly packed) to decoded[].
   Returns the number of decoded frames or a negative code.  The GPU decode check
   (tests/interpose's od_apply_postfilter_frame_sbs inside the decoder) rides on it. */
#include "daala/daaladec.h"
REF_EXPORT int ref_roundtrip_yuv420(const unsigned char *frames, int w, int h, int nframes,
 int quality, int complexity, unsigned char *decoded) {
  daala_info di;
  daala_info di2;
  daala_comment dc;
  daala_comment dc2;
  daala_setup_info *ds;
  daala_enc_ctx *enc;
  daala_dec_ctx *dec;
  daala_packet dp;
  daala_image img;
  daala_image out;
  long frame_bytes;
  int ndecoded;
  int f;
  int ret;
  int pli;
  daala_info_init(&di);
  di.pic_width = w;
  di.pic_height = h;
  di.bitdepth_mode = OD_BITDEPTH_MODE_8;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = 1;
  di.pixel_aspect_denominator = 1;
  di.nplanes = 3;
  di.plane_info[0].xdec = di.plane_info[0].ydec = 0;
  di.plane_info[1].xdec = di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = di.plane_info[2].ydec = 1;
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  daala_comment_init(&dc);
  daala_encode_ctl(enc, OD_SET_QUANT, &quality, sizeof(quality));
  daala_encode_ctl(enc, OD_SET_COMPLEXITY, &complexity, sizeof(complexity));
  daala_info_init(&di2);
  daala_comment_init(&dc2);
  ds = NULL;
  dec = NULL;
  for (;;) {
    ret = daala_encode_flush_header(enc, &dc, &dp);
    if (ret < 0) return ret;
    if (ret == 0) break;
    ret = daala_decode_header_in(&di2, &dc2, &ds, &dp);
    if (ret < 0) return ret;
  }
  dec = daala_decode_create(&di2, ds);
  if (dec == NULL) return -3;
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = w;
  img.height = h;
  frame_bytes = (long)w*h + 2L*((w + 1) >> 1)*((h + 1) >> 1);
  ndecoded = 0;
  for (f = 0; f <= nframes; f++) {
    int last;
    last = f == nframes;
    while (daala_encode_packet_out(enc, last, &dp)) {
      ret = daala_decode_packet_in(dec, &dp);
      if (ret < 0) return ret;
      while (daala_decode_img_out(dec, &out) > 0) {
        unsigned char *dst;
        dst = decoded + ndecoded*frame_bytes;
        for (pli = 0; pli < 3; pli++) {
          int pw;
          int ph;
          int y;
          pw = (w + out.planes[pli].xdec) >> out.planes[pli].xdec;
          ph = (h + out.planes[pli].ydec) >> out.planes[pli].ydec;
          for (y = 0; y < ph; y++) {
            memcpy(dst + y*pw, out.planes[pli].data + (long)y*out.planes[pli].ystride, pw);
          }
          dst += (long)pw*ph;
        }
        ndecoded++;
      }
    }
    if (!last) {
      unsigned char *base;
      base = (unsigned char *)frames + f*frame_bytes;
      for (pli = 0; pli < 3; pli++) {
        img.planes[pli].xdec = img.planes[pli].ydec = pli > 0;
        img.planes[pli].xstride = 1;
        img.planes[pli].ystride = pli ? (w + 1) >> 1 : w;
        img.planes[pli].bitdepth = 8;
      }
      img.planes[0].data = base;
      img.planes[1].data = base + (long)w*h;
      img.planes[2].data = img.planes[1].data + (long)((w + 1) >> 1)*((h + 1) >> 1);
      ret = daala_encode_img_in(enc, &img, 0);
      if (ret < 0) return ret;
    }
  }
  daala_setup_free(ds);
  daala_decode_free(dec);
  daala_comment_clear(&dc);
  daala_comment_clear(&dc2);
  daala_encode_free(enc);
  return ndecoded;
}

/* What a reconstruction check needs from an od_state (the first member of both
   daala_enc_ctx and daala_dec_ctx): which plane a ctmp pointer is, that plane's
   dequantised coefficients, the block-size map and the geometry. */
REF_EXPORT int ref_state_recon_view(void *statep, const od_coeff *c, const od_coeff **d,
 const unsigned char **bsize, int *bstride, int *pic_w, int *pic_h) {
  od_state *state;
  int pli;
  state = (od_state *)statep;
  for (pli = 0; pli < state->info.nplanes; pli++) {
    if (state->ctmp[pli] == c) {
      *d = state->dtmp[pli];
      *bsize = state->bsize;
      *bstride = state->bstride;
      *pic_w = state->info.pic_width;
      *pic_h = state->info.pic_height;
      return pli;
    }
  }
  return -1;
}
