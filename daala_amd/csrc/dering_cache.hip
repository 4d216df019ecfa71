/* dering_cache.hip - the deringing level search of a frame served from batched GPU
   passes (SURVEY.md 8(f) rank 1 at the frame level).

   After a frame is coded the reference searches, superblock by superblock, the
   deringing level: od_dering on the luma superblock for each of the five non-zero
   levels, a distortion for each, then od_dering on the three planes with the level
   chosen (src/encode.c:2697-2832: 2 550 luma + up to 1 530 chroma calls per 1080p
   frame, each a 64x64 / 32x32 superblock - 0.16 s of one core for the luma calls
   alone).  Every one of those calls reads the UNFILTERED copy of the frame
   (state->etmp, :2700-2707) and the block-skip map, which do not change during the
   search, and the threshold of a call depends only on the level.  So a call with a
   (plane, threshold) pair seen for the first time filters EVERY superblock of the
   plane in one launch (odhip_dering_planes) and keeps the filtered plane on the
   host; that call and the ~509 that follow with the same pair copy their superblock
   out.  The level decision, its entropy-coder cost and the adaptation stay in the
   encoder, untouched.

   The luma pass also yields the directions of every 8x8 block (od_dering writes
   dir[][] for pli == 0 and reads it for chroma, src/dering.c:282-290); the chroma
   passes read them on the device, the luma calls return them to the caller.

   No CPU implementation stands behind this: a call the cache cannot serve (partial
   superblock) goes to the per-call GPU path od_dering_hip. */
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/daala_hip.h"
#include "od_ctx.cuh"

struct odhip_dering_cache {
  struct Plane {
    const int16_t *base = nullptr;      /* host plane the device copy was made from */
    const unsigned char *skip_base = nullptr;
    int xstride = 0;
    int skip_stride = 0;
    int nhsb = 0;
    int nvsb = 0;
    int xdec = 0;
    bool loaded = false;
    int16_t *d_x = nullptr;
    uint8_t *d_skip = nullptr;
    int16_t *d_y = nullptr;
    size_t x_cap = 0;
    size_t skip_cap = 0;
  } planes[3];
  struct Result {
    int pli;
    int threshold;
    int overlap;
    int coeff_shift;
    bool valid;
    int16_t *h_y;        /* pinned, whole filtered plane, row stride = the plane's xstride */
    size_t cap;
    /* luma passes of a frame whose source picture is known (odhip_dering_cache_set_source): the
       distortion parts of every 8x8 block of (source, this filtered plane), and the per-superblock
       distortions finished from them on first request */
    double *h_parts = nullptr;
    size_t parts_cap = 0;
    bool have_parts = false;
    bool dist_ready = false;
    int dist_key[3] = {0, 0, 0};      /* use_masking, flat_qm, coded_quantizer of `dist` */
    std::vector<double> dist;
  };
  std::vector<Result> results;
  int32_t *d_dirs = nullptr;
  int32_t *h_dirs = nullptr;
  int32_t *d_thr = nullptr;
  size_t dirs_cap = 0;
  size_t thr_cap = 0;
  bool have_dirs = false;
  hipStream_t stream = nullptr;
  odhip_ctx *ctx = nullptr;
  long launches = 0;
  long served = 0;
  /* the source picture of the frame (luma, 8-bit): host copy for the identity check of a served
     distortion, device copy for the parts kernel */
  const uint8_t *src_h = nullptr;
  const uint8_t *src_d = nullptr;
  int src_stride = 0;
  int src_use_masking = 0;
  int src_flat = 0;
  double *d_parts = nullptr;
  size_t d_parts_cap = 0;
  long dist_served = 0;
};

namespace {

int grow(void **p, size_t *cap, size_t bytes) {
  if (bytes <= *cap) return ODHIP_SUCCESS;
  if (*p) ODHIP_TRY(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  ODHIP_TRY(hipMalloc(p, bytes));
  *cap = bytes;
  return ODHIP_SUCCESS;
}

/* The unfiltered plane and its skip map, once per frame. */
int load_plane(odhip_dering_cache *c, odhip_dering_cache::Plane &p, const int16_t *base, int xstride,
 const unsigned char *skip_base, int skip_stride, int nhsb, int nvsb, int xdec) {
  const int n = 64 >> xdec;
  const size_t xbytes = (size_t)nvsb*n*xstride*sizeof(int16_t);
  const size_t sbytes = (size_t)(nvsb << (4 - xdec))*skip_stride;
  {
    void *q = p.d_x;
    size_t cap = p.x_cap;
    if (xbytes > cap && p.d_y) {
      ODHIP_TRY(hipFree(p.d_y));
      p.d_y = nullptr;
    }
    const int rc = grow(&q, &cap, xbytes);
    if (rc) return rc;
    p.d_x = (int16_t *)q;
    p.x_cap = cap;
    if (!p.d_y) ODHIP_TRY(hipMalloc((void **)&p.d_y, cap));
  }
  {
    void *q = p.d_skip;
    const int rc = grow(&q, &p.skip_cap, sbytes);
    if (rc) return rc;
    p.d_skip = (uint8_t *)q;
  }
  ODHIP_TRY(hipMemcpyAsync(p.d_x, base, xbytes, hipMemcpyHostToDevice, c->stream));
  ODHIP_TRY(hipMemcpyAsync(p.d_skip, skip_base, sbytes, hipMemcpyHostToDevice, c->stream));
  /* the encoder owns the source: it must not be read after this call returns */
  ODHIP_TRY(hipStreamSynchronize(c->stream));
  p.base = base;
  p.skip_base = skip_base;
  p.xstride = xstride;
  p.skip_stride = skip_stride;
  p.nhsb = nhsb;
  p.nvsb = nvsb;
  p.xdec = xdec;
  p.loaded = true;
  return ODHIP_SUCCESS;
}

/* One launch: every superblock of plane pli with one threshold; the filtered plane (and,
   for luma, the directions) come back to the host. */
int run_pass(odhip_dering_cache *c, int pli, odhip_dering_cache::Result &r) {
  odhip_dering_cache::Plane &p = c->planes[pli];
  const int n = 64 >> p.xdec;
  const long nsb = (long)p.nhsb*p.nvsb;
  const size_t dbytes = sizeof(int32_t)*(size_t)nsb*64;
  if (dbytes > c->dirs_cap) {
    if (c->have_dirs) return ODHIP_EINVAL;     /* the geometry changed inside a frame */
    if (c->d_dirs) ODHIP_TRY(hipFree(c->d_dirs));
    if (c->h_dirs) ODHIP_TRY(hipHostFree(c->h_dirs));
    c->d_dirs = nullptr;
    c->h_dirs = nullptr;
    c->dirs_cap = 0;
    ODHIP_TRY(hipMalloc((void **)&c->d_dirs, dbytes));
    ODHIP_TRY(hipHostMalloc((void **)&c->h_dirs, dbytes, hipHostMallocDefault));
    c->dirs_cap = dbytes;
  }
  {
    void *q = c->d_thr;
    const int rc = grow(&q, &c->thr_cap, sizeof(int32_t)*(size_t)nsb);
    if (rc) return rc;
    c->d_thr = (int32_t *)q;
  }
  std::vector<int32_t> thr((size_t)nsb, r.threshold);
  ODHIP_TRY(hipMemcpyAsync(c->d_thr, thr.data(), sizeof(int32_t)*(size_t)nsb, hipMemcpyHostToDevice, c->stream));
  /* the staging vector must outlive the copy: pageable copies return after staging */
  const size_t ybytes = (size_t)p.nvsb*n*p.xstride*sizeof(int16_t);
  if (ybytes > r.cap) {
    if (r.h_y) ODHIP_TRY(hipHostFree(r.h_y));
    r.h_y = nullptr;
    r.cap = 0;
    ODHIP_TRY(hipHostMalloc((void **)&r.h_y, ybytes, hipHostMallocDefault));
    r.cap = ybytes;
  }
  const int rc = odhip_dering_planes(p.d_y, p.d_x, p.xstride, p.nhsb, p.nvsb, p.xdec, 1, c->d_dirs, pli,
   p.d_skip, p.skip_stride, 0, c->d_thr, 1, r.overlap, r.coeff_shift, c->stream);
  if (rc) return rc;
  ODHIP_TRY(hipMemcpyAsync(r.h_y, p.d_y, ybytes, hipMemcpyDeviceToHost, c->stream));
  if (pli == 0) ODHIP_TRY(hipMemcpyAsync(c->h_dirs, c->d_dirs, dbytes, hipMemcpyDeviceToHost, c->stream));
  r.have_parts = false;
  r.dist_ready = false;
  if (pli == 0 && c->src_d) {
    /* the level search compares every filtered superblock with the source (od_compute_dist,
       src/encode.c:2776-2801): the device part of all of them now, on the planes already here */
    const int w = p.nhsb*64;
    const int h = p.nvsb*64;
    const size_t pbytes = sizeof(double)*(size_t)3*(w >> 3)*(h >> 3);
    {
      void *q = c->d_parts;
      const int rcg = grow(&q, &c->d_parts_cap, pbytes);
      if (rcg) return rcg;
      c->d_parts = (double *)q;
    }
    if (pbytes > r.parts_cap) {
      if (r.h_parts) ODHIP_TRY(hipHostFree(r.h_parts));
      r.h_parts = nullptr;
      r.parts_cap = 0;
      ODHIP_TRY(hipHostMalloc((void **)&r.h_parts, pbytes, hipHostMallocDefault));
      r.parts_cap = pbytes;
    }
    const int rcd = odhip_dist_parts_px16(c->d_parts, c->src_d, c->src_stride, p.d_y, p.xstride, 1, w, h, 4,
     c->src_use_masking, c->src_flat, c->stream);
    if (rcd) return rcd;
    ODHIP_TRY(hipMemcpyAsync(r.h_parts, c->d_parts, pbytes, hipMemcpyDeviceToHost, c->stream));
    r.have_parts = true;
  }
  ODHIP_TRY(hipStreamSynchronize(c->stream));
  if (pli == 0) c->have_dirs = true;
  c->launches++;
  return ODHIP_SUCCESS;
}

}  // namespace

extern "C" odhip_dering_cache *odhip_dering_cache_create(void) {
  /* the cache OWNS its context (as the frame cache does): it may be used from another
     thread than its creator, and outlive that thread's default context */
  odhip_ctx *cur = odhip_ctx_current();
  if (!cur) return nullptr;
  odhip_ctx *ctx = odhip_create(cur->device);
  if (!ctx) return nullptr;
  odhip_dering_cache *c = new odhip_dering_cache();
  c->ctx = ctx;
  if (hipSetDevice(ctx->device) != hipSuccess
   || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    odhip_destroy(ctx);
    delete c;
    return nullptr;
  }
  return c;
}

extern "C" void odhip_dering_cache_destroy(odhip_dering_cache *c) {
  if (!c) return;
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  for (auto &p : c->planes) {
    if (p.d_x) (void)hipFree(p.d_x);
    if (p.d_y) (void)hipFree(p.d_y);
    if (p.d_skip) (void)hipFree(p.d_skip);
  }
  for (auto &r : c->results) {
    if (r.h_y) (void)hipHostFree(r.h_y);
    if (r.h_parts) (void)hipHostFree(r.h_parts);
  }
  if (c->d_parts) (void)hipFree(c->d_parts);
  if (c->d_dirs) (void)hipFree(c->d_dirs);
  if (c->h_dirs) (void)hipHostFree(c->h_dirs);
  if (c->d_thr) (void)hipFree(c->d_thr);
  if (c->ctx) {
    if (odhip_get_current() == c->ctx) (void)odhip_make_current(nullptr);
    odhip_destroy(c->ctx);
  }
  delete c;
}

extern "C" void odhip_dering_cache_begin(odhip_dering_cache *c) {
  if (!c) return;
  for (auto &p : c->planes) p.loaded = false;
  for (auto &r : c->results) {     /* the pinned buffers are kept for the next frame */
    r.valid = false;
    r.have_parts = false;
    r.dist_ready = false;
  }
  c->have_dirs = false;
  c->src_h = c->src_d = nullptr;
}

/* The frame's luma source picture (8-bit samples, host and device copies of the same plane, row
   stride in samples; the planes od_ref_buf_to_coeff converts, i.e. the padded input): from now
   until the next odhip_dering_cache_begin every luma pass also computes the distortion parts of its
   output against it.  use_masking / flat_qm: the encoder's od_compute_dist settings. */
extern "C" int odhip_dering_cache_set_source(odhip_dering_cache *c, const uint8_t *h_px, const uint8_t *d_px,
 int stride, int use_masking, int flat_qm) {
  if (!c || !h_px || !d_px || stride <= 0) return ODHIP_EINVAL;
  c->src_h = h_px;
  c->src_d = d_px;
  c->src_stride = stride;
  c->src_use_masking = use_masking != 0;
  c->src_flat = flat_qm != 0;
  return ODHIP_SUCCESS;
}

/* od_compute_dist(enc, x, y, 64) of the level search, served when x IS superblock (sbx, sby) of the
   source picture and y IS that superblock of the luma pass with this threshold (both verified
   sample by sample: nothing is assumed about the caller's call order): 1 and *dist, or 0 when the
   cache cannot vouch for it (the caller runs the C function). */
extern "C" int odhip_dering_cache_dist(odhip_dering_cache *c, const od_coeff *x, const od_coeff *y, int n, int sbx,
 int sby, int threshold, int use_masking, int flat_qm, int coded_quantizer, double *dist) {
  if (!c || !x || !y || !dist || n != 64 || !c->src_h) return 0;
  const odhip_dering_cache::Plane &p = c->planes[0];
  if (!p.loaded || sbx < 0 || sby < 0 || sbx >= p.nhsb || sby >= p.nvsb) return 0;
  if ((use_masking != 0) != (c->src_use_masking != 0) || (flat_qm != 0) != (c->src_flat != 0)) return 0;
  odhip_dering_cache::Result *hit = nullptr;
  for (auto &r : c->results) {
    if (r.valid && r.pli == 0 && r.threshold == threshold && r.have_parts) {
      hit = &r;
      break;
    }
  }
  if (!hit) return 0;
  const uint8_t *sp = c->src_h + (long)sby*64*c->src_stride + (long)sbx*64;
  const int16_t *fp = hit->h_y + (long)sby*64*p.xstride + (long)sbx*64;
  for (int i = 0; i < 64; i++) {
    for (int j = 0; j < 64; j++) {
      if (x[i*64 + j] != ((int)sp[(long)i*c->src_stride + j] - 128) << 4) return 0;
      if (y[i*64 + j] != fp[(long)i*p.xstride + j]) return 0;
    }
  }
  if (!hit->dist_ready || hit->dist_key[0] != (use_masking != 0) || hit->dist_key[1] != (flat_qm != 0)
   || hit->dist_key[2] != coded_quantizer) {
    hit->dist.resize((size_t)p.nhsb*p.nvsb);
    if (odhip_dist_finish(hit->dist.data(), hit->h_parts, 1, p.nhsb*64, p.nvsb*64, 4, use_masking, flat_qm,
     coded_quantizer) != ODHIP_SUCCESS) {
      return 0;
    }
    hit->dist_key[0] = use_masking != 0;
    hit->dist_key[1] = flat_qm != 0;
    hit->dist_key[2] = coded_quantizer;
    hit->dist_ready = true;
  }
  *dist = hit->dist[(size_t)sby*p.nhsb + sbx];
  c->dist_served++;
  return 1;
}

extern "C" long odhip_dering_cache_dist_served(const odhip_dering_cache *c) {
  return c ? c->dist_served : 0;
}

extern "C" void odhip_dering_cache_stats(const odhip_dering_cache *c, long *launches, long *served) {
  if (launches) *launches = c ? c->launches : 0;
  if (served) *served = c ? c->served : 0;
}

extern "C" int odhip_dering_cache_call(odhip_dering_cache *c, int16_t *y, int ystride, const int16_t *x,
 int xstride, int nhb, int nvb, int sbx, int sby, int nhsb, int nvsb, int xdec, int dir[8][8], int pli,
 unsigned char *bskip, int skip_stride, int threshold, int overlap, int coeff_shift) {
  if (!c || !y || !x || !dir || !bskip || pli < 0 || pli > 2 || (xdec != 0 && xdec != 1)
   || (pli == 0 && xdec != 0) || sbx < 0 || sby < 0 || sbx >= nhsb || sby >= nvsb) {
    return ODHIP_EINVAL;
  }
  if (nhb != 8 || nvb != 8) {
    /* a partial superblock: the per-call GPU path */
    od_dering_hip(y, ystride, x, xstride, nhb, nvb, sbx, sby, nhsb, nvsb, xdec, dir, pli, bskip, skip_stride,
     threshold, overlap, coeff_shift);
    return ODHIP_SUCCESS;
  }
  odhip_ctx *prev = odhip_get_current();
  odhip_make_current(c->ctx);
  struct Restore {
    odhip_ctx *p;
    ~Restore() { odhip_make_current(p); }
  } restore{prev};
  ODHIP_TRY(hipSetDevice(c->ctx->device));
  const int n = 64 >> xdec;
  const int16_t *base = x - ((long)sby*n*xstride + (long)sbx*n);
  const unsigned char *skip_base = bskip - ((long)(sby << (4 - xdec))*skip_stride + (sbx << (4 - xdec)));
  odhip_dering_cache::Plane &p = c->planes[pli];
  if (!p.loaded || p.base != base || p.xstride != xstride || p.skip_base != skip_base
   || p.skip_stride != skip_stride || p.nhsb != nhsb || p.nvsb != nvsb || p.xdec != xdec) {
    if (p.loaded) {
      /* another buffer under the same plane index within one frame: everything derived
         from the old one is stale */
      for (auto &r : c->results) {
        if (r.pli == pli) r.valid = false;
      }
      if (pli == 0) c->have_dirs = false;
    }
    const int rc = load_plane(c, p, base, xstride, skip_base, skip_stride, nhsb, nvsb, xdec);
    if (rc) return rc;
  }
  if (pli != 0 && !c->have_dirs) {
    /* chroma before any luma call of the frame: the directions of the other superblocks
       are unknown, so only this superblock can be filtered */
    od_dering_hip(y, ystride, x, xstride, nhb, nvb, sbx, sby, nhsb, nvsb, xdec, dir, pli, bskip, skip_stride,
     threshold, overlap, coeff_shift);
    return ODHIP_SUCCESS;
  }
  odhip_dering_cache::Result *hit = nullptr;
  odhip_dering_cache::Result *spare = nullptr;
  for (auto &r : c->results) {
    if (r.valid && r.pli == pli && r.threshold == threshold && r.overlap == overlap
     && r.coeff_shift == coeff_shift) {
      hit = &r;
      break;
    }
    if (!r.valid && !spare) spare = &r;
  }
  if (!hit) {
    if (!spare) {
      odhip_dering_cache::Result fresh;
      fresh.pli = pli;
      fresh.threshold = threshold;
      fresh.overlap = overlap;
      fresh.coeff_shift = coeff_shift;
      fresh.valid = false;
      fresh.h_y = nullptr;
      fresh.cap = 0;
      c->results.push_back(fresh);
      spare = &c->results.back();
    }
    spare->pli = pli;
    spare->threshold = threshold;
    spare->overlap = overlap;
    spare->coeff_shift = coeff_shift;
    const int rc = run_pass(c, pli, *spare);
    if (rc) return rc;
    spare->valid = true;
    hit = spare;
  }
  const int16_t *src = hit->h_y + (long)sby*n*xstride + (long)sbx*n;
  for (int i = 0; i < n; i++) memcpy(y + (long)i*ystride, src + (long)i*xstride, n*sizeof(int16_t));
  if (pli == 0) {
    const int32_t *d = c->h_dirs + ((long)sby*8)*(nhsb*8) + sbx*8;
    for (int by = 0; by < 8; by++) {
      for (int bx = 0; bx < 8; bx++) dir[by][bx] = d[(long)by*nhsb*8 + bx];
    }
  }
  c->served++;
  return ODHIP_SUCCESS;
}
