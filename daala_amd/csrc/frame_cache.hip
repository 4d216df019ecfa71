/* frame_cache.hip - serving the reference encoder's per-block fdct_2d calls
   from ONE batched GPU pyramid per plane.

   The reference calls fdct_2d[bs](d + bo, w, c + bo, w) once per evaluated
   block (src/encode.c:1304,1477): 366 264 calls per 1080p keyframe, one small
   block each - a per-call offload pays launch + PCIe latency every time.  But
   the samples a block of level bs sees depend only on the source plane and on
   "every ancestor was split" (verified there on a whole
   frame: 161 268 repeated calls, 0 differing inputs), i.e. they are exactly
   what odhip_forward_pyramid computes for that (level, bx, by).  So:

     odhip_cache_load_plane   at the moment the encoder has converted a plane to
                              coefficients and is about to lap it across
                              superblock edges (od_apply_prefilter_frame_sbs,
                              src/encode.c:2571): pixels are recovered from the
                              coefficients ((p - 128) << 4 is invertible), the
                              whole pyramid is computed in one launch and copied
                              to pinned host memory
     odhip_cache_lookup       an fdct_2d call whose input pointer lies in a
                              cached plane becomes a strided host copy

   Anything that does not hit (other buffers, inter-frame prediction planes,
   unaligned positions) falls through to the per-call GPU transform - never to a
   CPU implementation.  ODHIP_CACHE_CHECK=1 recomputes every hit with the
   per-call path and aborts on a mismatch (the OD_CHECKASM idea,
   src/dct.c:4922-4948). */
#include <stdlib.h>
#include <string.h>
#include "../../include/daala_hip.h"
#include "od_ctx.cuh"
#include "gen/od_scan_tables.h"

/* ---- the batched no-reference band stage behind the encoder's pvq_theta ----------
   odhip_cache_load_bands (after odhip_cache_load_plane of the same plane): the PVQ
   band stage of EVERY block of EVERY level of the plane in one set of launches
   (odhip_pvq_noref_bands_multi on the pyramid still resident in HBM), records and
   pulse vectors copied to pinned host memory.  odhip_cache_band then serves one
   pvq_theta call of the encoder's block loop - the gains, K, pruning decisions,
   pulse vectors and distortions of its candidates - whenever that call takes the
   no-reference path, i.e. its reference vector is null (keyframe luma whose
   neighbours predict nothing, src/encode.c:874-878, src/pvq_encoder.c:452,:571).
   The encoder prices the candidates with od_pvq_rate on its live adaptive state
   (src/pvq_encoder.c:597-599), chooses and synthesises: that part is sequential
   host state in the reference and stays there. */
struct BandLevel {
  odhip_pvq_band *d_band = nullptr;
  int16_t *d_y = nullptr;
  int32_t *d_choice = nullptr;
  int16_t *d_qm = nullptr;
  odhip_pvq_band *h_band = nullptr;
  int16_t *h_y = nullptr;
  long nblocks = 0;
  int nb = 0;
  int len = 0;
  int32_t q[ODHIP_MAX_BANDS];
  int32_t beta[ODHIP_MAX_BANDS];
  int off[ODHIP_MAX_BANDS + 1];
};

struct odhip_frame_cache {
  struct Plane {
    const od_coeff *base;
    int w;
    int h;
    int dec;
    int valid;
    uint8_t *h_px;
    uint8_t *d_px;
    od_coeff *h_levels[ODHIP_NBSIZES];
    od_coeff *d_levels[ODHIP_NBSIZES];
    size_t cap;
    int pic_w, pic_h;        /* the picture size the cached pyramid was made for */
    BandLevel *bands;        /* [ODHIP_NBSIZES], allocated by odhip_cache_load_bands */
    int bands_valid;
    int bands_quantizer;     /* the set-up the cached band stage was run with */
    int bands_masking;
    double bands_lambda;
    uint8_t bands_qm4[ODHIP_QM_SIZE];
    unsigned long long bands_qm_hash;
  } planes[4];
  hipStream_t stream;
  odhip_ctx *ctx;
  long band_hits;
  long band_misses;
  long reloads_skipped;
  int pic_w;
  int pic_h;
  int check;
  long hits;
  long misses;
};

namespace {

thread_local odhip_frame_cache *g_current = nullptr;

int plane_reserve(odhip_frame_cache::Plane &p, int w, int h, int dec) {
  const size_t n = (size_t)w*h;
  if (n <= p.cap && p.w == w && p.h == h && p.dec == dec) return ODHIP_SUCCESS;
  if (p.h_px) ODHIP_TRY(hipHostFree(p.h_px));
  if (p.d_px) ODHIP_TRY(hipFree(p.d_px));
  p.h_px = nullptr;
  p.d_px = nullptr;
  for (int i = 0; i < ODHIP_NBSIZES; i++) {
    if (p.h_levels[i]) ODHIP_TRY(hipHostFree(p.h_levels[i]));
    if (p.d_levels[i]) ODHIP_TRY(hipFree(p.d_levels[i]));
    p.h_levels[i] = nullptr;
    p.d_levels[i] = nullptr;
  }
  p.cap = 0;
  ODHIP_TRY(hipHostMalloc((void **)&p.h_px, n, hipHostMallocDefault));
  ODHIP_TRY(hipMalloc((void **)&p.d_px, n));
  for (int i = 0; i <= 4 - dec; i++) {
    ODHIP_TRY(hipHostMalloc((void **)&p.h_levels[i], n*sizeof(od_coeff), hipHostMallocDefault));
    ODHIP_TRY(hipMalloc((void **)&p.d_levels[i], n*sizeof(od_coeff)));
  }
  p.cap = n;
  p.w = w;
  p.h = h;
  p.dec = dec;
  return ODHIP_SUCCESS;
}

void band_level_free(BandLevel &b) {
  if (b.d_band) (void)hipFree(b.d_band);
  if (b.d_y) (void)hipFree(b.d_y);
  if (b.d_choice) (void)hipFree(b.d_choice);
  if (b.d_qm) (void)hipFree(b.d_qm);
  if (b.h_band) (void)hipHostFree(b.h_band);
  if (b.h_y) (void)hipHostFree(b.h_y);
  b = BandLevel();
}

typedef void (*dct_fn)(od_coeff *, int, const od_coeff *, int);
const dct_fn kPerCallFdct[ODHIP_NBSIZES] = {
  od_bin_fdct4x4_hip, od_bin_fdct8x8_hip, od_bin_fdct16x16_hip, od_bin_fdct32x32_hip,
  od_bin_fdct64x64_hip
};

template <int BS>
void cached_fdct(od_coeff *out, int out_stride, const od_coeff *in, int in_stride) {
  odhip_frame_cache *c = g_current;
  if (c && odhip_cache_lookup(c, in, in_stride, BS, out, out_stride) == 1) return;
  kPerCallFdct[BS](out, out_stride, in, in_stride);
}

}  // namespace

extern "C" {

odhip_frame_cache *odhip_cache_create(void) {
  odhip_frame_cache *c = (odhip_frame_cache *)calloc(1, sizeof(*c));
  if (!c) return nullptr;
  if (hipStreamCreate(&c->stream) != hipSuccess) {
    free(c);
    return nullptr;
  }
  const char *e = getenv("ODHIP_CACHE_CHECK");
  c->check = e && e[0] == '1';
  int dev = 0;
  (void)hipGetDevice(&dev);
  c->ctx = odhip_create(dev);
  return c;
}

void odhip_cache_destroy(odhip_frame_cache *c) {
  if (!c) return;
  if (g_current == c) g_current = nullptr;
  for (int i = 0; i < 4; i++) {
    odhip_frame_cache::Plane &p = c->planes[i];
    if (p.h_px) (void)hipHostFree(p.h_px);
    if (p.d_px) (void)hipFree(p.d_px);
    for (int l = 0; l < ODHIP_NBSIZES; l++) {
      if (p.h_levels[l]) (void)hipHostFree(p.h_levels[l]);
      if (p.d_levels[l]) (void)hipFree(p.d_levels[l]);
    }
    if (p.bands) {
      for (int l = 0; l < ODHIP_NBSIZES; l++) band_level_free(p.bands[l]);
      delete[] p.bands;
    }
  }
  if (c->ctx) odhip_destroy(c->ctx);
  (void)hipStreamDestroy(c->stream);
  free(c);
}

void odhip_cache_set_picture(odhip_frame_cache *c, int pic_w, int pic_h) {
  c->pic_w = pic_w;
  c->pic_h = pic_h;
}

void odhip_cache_make_current(odhip_frame_cache *c) {
  g_current = c;
}

void odhip_cache_stats(const odhip_frame_cache *c, long *hits, long *misses) {
  if (hits) *hits = c->hits;
  if (misses) *misses = c->misses;
}

/* The 8-bit source samples of a loaded plane, on the host (pinned) and on the device: what
   od_ref_buf_to_coeff made the coefficients of (valid until the next load of that slot). */
int odhip_cache_plane_pixels(const odhip_frame_cache *c, int pli, const uint8_t **h_px, const uint8_t **d_px,
 int *w, int *h) {
  if (!c || pli < 0 || pli >= 4 || !c->planes[pli].valid) return ODHIP_EINVAL;
  if (h_px) *h_px = c->planes[pli].h_px;
  if (d_px) *d_px = c->planes[pli].d_px;
  if (w) *w = c->planes[pli].w;
  if (h) *h = c->planes[pli].h;
  return ODHIP_SUCCESS;
}

int odhip_cache_load_plane(odhip_frame_cache *c, int pli, const od_coeff *coef, int stride,
 int w, int h, int dec) {
  if (!c || pli < 0 || pli >= 4 || !coef || stride != w || (dec != 0 && dec != 1)) {
    return ODHIP_EINVAL;
  }
  const int tile = 64 >> dec;
  if (w <= 0 || h <= 0 || w % tile || h % tile || c->pic_w <= 0 || c->pic_h <= 0) {
    return ODHIP_EINVAL;
  }
  odhip_frame_cache::Plane &p = c->planes[pli];
  const bool same_shape = p.valid && p.w == w && p.h == h && p.dec == dec && p.base == coef
   && p.pic_w == c->pic_w && p.pic_h == c->pic_h;
  const int was_bands = p.bands_valid;
  p.valid = 0;
  p.bands_valid = 0;
  int rc = plane_reserve(p, w, h, dec);
  if (rc) return rc;
  /* od_ref_buf_to_coeff wrote (p - 128) << 4 (src/state.c:1233): recover p. */
  const size_t n = (size_t)w*h;
  /* the exact test for "the same pixels again" (below): compared while h_px still holds
     the previous load's */
  bool same_px = same_shape;
  for (size_t i = 0; i < n; i++) {
    const int v = coef[i];
    const int px = (v >> 4) + 128;
    if ((v & 15) || px < 0 || px > 255) return ODHIP_EINVAL;  /* not a fresh 8-bit plane */
    same_px = same_px && p.h_px[i] == (uint8_t)px;
    p.h_px[i] = (uint8_t)px;
  }
  /* The encoder converts and laps the same input once per RDO pass (src/encode.c:
     2560-2572 runs for OD_ENCODE_RDO and for OD_ENCODE_REAL): the second load of
     identical pixels keeps the pyramid (and the band stage) it already has. */
  if (same_px && !c->check) {
    p.valid = 1;
    p.bands_valid = was_bands;
    c->reloads_skipped++;
    return ODHIP_SUCCESS;
  }
  p.pic_w = c->pic_w;
  p.pic_h = c->pic_h;
  ODHIP_TRY(hipMemcpyAsync(p.d_px, p.h_px, n, hipMemcpyHostToDevice, c->stream));
  od_coeff *levels[ODHIP_NBSIZES];
  for (int i = 0; i < ODHIP_NBSIZES; i++) levels[i] = p.d_levels[i];
  rc = odhip_forward_pyramid(levels, p.d_px, w, (long)n, 1, w, h, dec, c->pic_w, c->pic_h,
   c->stream);
  if (rc) return rc;
  for (int i = 0; i <= 4 - dec; i++) {
    ODHIP_TRY(hipMemcpyAsync(p.h_levels[i], p.d_levels[i], n*sizeof(od_coeff),
     hipMemcpyDeviceToHost, c->stream));
  }
  ODHIP_TRY(hipStreamSynchronize(c->stream));
  p.base = coef;
  p.valid = 1;
  return ODHIP_SUCCESS;
}

int odhip_cache_lookup(odhip_frame_cache *c, const od_coeff *in, int in_stride, int bs,
 od_coeff *out, int out_stride) {
  if (!c || bs < 0 || bs >= ODHIP_NBSIZES) return 0;
  const int n = 4 << bs;
  for (int i = 0; i < 4; i++) {
    odhip_frame_cache::Plane &p = c->planes[i];
    if (!p.valid || in_stride != p.w || in < p.base || in >= p.base + (size_t)p.w*p.h) continue;
    const size_t off = (size_t)(in - p.base);
    const int y = (int)(off / p.w);
    const int x = (int)(off % p.w);
    if (bs > 4 - p.dec || x % n || y % n || x + n > p.w || y + n > p.h) break;
    const od_coeff *src = p.h_levels[bs] + off;
    if (c->check) {
      od_coeff tmp[64*64];
      kPerCallFdct[bs](tmp, n, in, in_stride);
      for (int r = 0; r < n; r++) {
        if (memcmp(tmp + r*n, src + (size_t)r*p.w, n*sizeof(od_coeff)) != 0) {
          fprintf(stderr, "libdaalahip: frame cache mismatch: plane %d bs %d at (%d,%d)\n", i, bs,
           x, y);
          abort();
        }
      }
    }
    for (int r = 0; r < n; r++) {
      memcpy(out + (size_t)r*out_stride, src + (size_t)r*p.w, n*sizeof(od_coeff));
    }
    c->hits++;
    return 1;
  }
  c->misses++;
  return 0;
}

int odhip_cache_load_bands(odhip_frame_cache *c, int pli, const odhip_quant *qt,
 double pvq_norm_lambda) {
  if (!c || pli < 0 || pli >= 4 || !qt || !c->ctx) return ODHIP_EINVAL;
  odhip_frame_cache::Plane &p = c->planes[pli];
  if (!p.valid) return ODHIP_EINVAL;
  /* kept by a skipped reload of the same pixels: still valid if the quantiser set-up is
     the one it was made with */
  unsigned long long qm_hash = 1469598103934665603ULL;
  for (int i = 0; i < ODHIP_QM_BUFFER_SIZE; i++) qm_hash = (qm_hash ^ (uint16_t)qt->qm[i])*1099511628211ULL;
  if (p.bands_valid && p.bands && p.bands_qm_hash == qm_hash && p.bands_quantizer == qt->quantizer && p.bands_lambda == pvq_norm_lambda
   && memcmp(p.bands_qm4, qt->pvq_qm_q4[pli > 2 ? 2 : pli], ODHIP_QM_SIZE) == 0
   && p.bands_masking == qt->use_masking && !c->check) {
    return ODHIP_SUCCESS;
  }
  p.bands_valid = 0;
  p.bands_quantizer = qt->quantizer;
  p.bands_lambda = pvq_norm_lambda;
  p.bands_masking = qt->use_masking;
  p.bands_qm_hash = qm_hash;
  memcpy(p.bands_qm4, qt->pvq_qm_q4[pli > 2 ? 2 : pli], ODHIP_QM_SIZE);
  if (!p.bands) p.bands = new BandLevel[ODHIP_NBSIZES];
  const int nlev = ODHIP_NBSIZES - p.dec;
  const int qpli = pli > 2 ? 2 : pli;
  odhip_pvq_job jobs[ODHIP_NBSIZES];
  memset(jobs, 0, sizeof(jobs));
  for (int bs = 0; bs < nlev; bs++) {
    BandLevel &b = p.bands[bs];
    const int n = 4 << bs;
    const long B = (long)(p.w/n)*(p.h/n);
    int nb = 0;
    int len = 0;
    odhip_pvq_band_layout(bs, &nb, b.off, &len);
    if (b.nblocks != B || b.nb != nb || b.len != len) {
      band_level_free(b);
      odhip_pvq_band_layout(bs, &nb, b.off, &len);
      ODHIP_TRY(hipMalloc((void **)&b.d_band, sizeof(odhip_pvq_band)*(size_t)B*nb));
      ODHIP_TRY(hipMalloc((void **)&b.d_y, sizeof(int16_t)*(size_t)2*B*len));
      ODHIP_TRY(hipMalloc((void **)&b.d_choice, sizeof(int32_t)*(size_t)B*nb*4));
      ODHIP_TRY(hipMalloc((void **)&b.d_qm, sizeof(int16_t)*len));
      ODHIP_TRY(hipHostMalloc((void **)&b.h_band, sizeof(odhip_pvq_band)*(size_t)B*nb,
       hipHostMallocDefault));
      ODHIP_TRY(hipHostMalloc((void **)&b.h_y, sizeof(int16_t)*(size_t)2*B*len, hipHostMallocDefault));
      ODHIP_TRY(hipMemsetAsync(b.d_y, 0, sizeof(int16_t)*(size_t)2*B*len, c->stream));
      b.nblocks = B;
      b.nb = nb;
      b.len = len;
    }
    if (odhip_quant_bands(qt, qpli, bs, b.q, b.beta) != nb) return ODHIP_EINVAL;
    /* the tables may change from frame to frame (quantiser, matrices): copy per load;
       the source is the caller's memory, so the copy completes before returning */
    ODHIP_TRY(hipMemcpy(b.d_qm, qt->qm + odhip_qm_offset(bs, p.dec), sizeof(int16_t)*len,
     hipMemcpyHostToDevice));
    odhip_pvq_job &j = jobs[bs];
    j.d_coef = p.d_levels[bs];
    j.nplanes = 1;
    j.w = p.w;
    j.h = p.h;
    j.bs = bs;
    j.d_qm = b.d_qm;
    j.q_band = b.q;
    j.beta_band = b.beta;
    j.cands.band = b.d_band;
    j.cands.y = b.d_y;
    j.cands.choice = b.d_choice;
  }
  odhip_ctx *prev = odhip_get_current();
  (void)odhip_make_current(c->ctx);
  int rc = odhip_pvq_noref_bands_multi(jobs, nlev, pvq_norm_lambda, c->stream);
  (void)odhip_make_current(prev);
  if (rc) return rc;
  for (int bs = 0; bs < nlev; bs++) {
    BandLevel &b = p.bands[bs];
    ODHIP_TRY(hipMemcpyAsync(b.h_band, b.d_band, sizeof(odhip_pvq_band)*(size_t)b.nblocks*b.nb,
     hipMemcpyDeviceToHost, c->stream));
    ODHIP_TRY(hipMemcpyAsync(b.h_y, b.d_y, sizeof(int16_t)*(size_t)2*b.nblocks*b.len,
     hipMemcpyDeviceToHost, c->stream));
  }
  ODHIP_TRY(hipStreamSynchronize(c->stream));
  p.bands_valid = 1;
  return ODHIP_SUCCESS;
}

int odhip_cache_band(odhip_frame_cache *c, int pli, int bs, int bx, int by, int band,
 const od_coeff *x0, odhip_band_cands *out) {
  if (!c || !out || pli < 0 || pli >= 4) return 0;
  odhip_frame_cache::Plane &p = c->planes[pli];
  if (!p.valid || !p.bands_valid || bs < 0 || bs > 4 - p.dec) {
    c->band_misses++;
    return 0;
  }
  const BandLevel &b = p.bands[bs];
  const int n = 4 << bs;
  if (bx < 0 || by < 0 || bx >= p.w/n || by >= p.h/n || band < 0 || band >= b.nb) {
    c->band_misses++;
    return 0;
  }
  const long blk = (long)by*(p.w/n) + bx;
  const odhip_pvq_band &r = b.h_band[blk*b.nb + band];
  const int off = b.off[band];
  const int nn = b.off[band + 1] - off;
  if (c->check && x0) {
    /* the band the encoder is about to code must be the band the batch coded */
    const od_coeff *src = p.h_levels[bs] + ((size_t)by*n*p.w + (size_t)bx*n);
    for (int i = 0; i < nn; i++) {
      if (x0[i] != src[(size_t)OD_SCAN_XY[off + i][1]*p.w + OD_SCAN_XY[off + i][0]]) {
        fprintf(stderr, "libdaalahip: band cache mismatch: plane %d bs %d block (%d,%d) band %d\n", pli,
         bs, bx, by, band);
        abort();
      }
    }
  }
  out->n = nn;
  out->q = b.q[band];
  out->beta = b.beta[band];
  out->cg = r.cg;
  out->dist0 = r.dist0;
  for (int s = 0; s < 2; s++) {
    out->gain[s] = r.gain[s];
    out->k[s] = r.k[s];
    out->flags[s] = r.flags[s];
    out->dist[s] = r.dist[s];
    out->y[s] = b.h_y + ((size_t)s*b.nblocks + blk)*b.len + off;
  }
  c->band_hits++;
  return 1;
}

void odhip_cache_band_stats(const odhip_frame_cache *c, long *hits, long *misses) {
  if (hits) *hits = c->band_hits;
  if (misses) *misses = c->band_misses;
}

void odhip_install_cached_dct_vtbl(odhip_dct_func_2d fdct_2d[ODHIP_NBSIZES],
 odhip_dct_func_2d idct_2d[ODHIP_NBSIZES]) {
  odhip_dct_func_2d tmp[ODHIP_NBSIZES];
  odhip_install_dct_vtbl(tmp, idct_2d);
  fdct_2d[0] = cached_fdct<0>;
  fdct_2d[1] = cached_fdct<1>;
  fdct_2d[2] = cached_fdct<2>;
  fdct_2d[3] = cached_fdct<3>;
  fdct_2d[4] = cached_fdct<4>;
}

}  /* extern "C" */
