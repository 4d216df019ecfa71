/* tests/interpose/interpose.c - TEST INFRASTRUCTURE.

   Load-time replacement of the reference's lapping-filter drivers and of its
   PVQ search by libdaalahip, WITHOUT touching the reference: this library
   defines the reference's own symbol names (src/filter.h:80-87,
   src/pvq_encoder.c:93) and forwards them to the *_hip entry points.  Loaded
   with RTLD_GLOBAL before oracle/_ref/libdaalaref.so, the dynamic linker binds
   every call inside the unmodified reference encoder (src/encode.c:1489,1760,
   1789,2571,2675; src/pvq_encoder.c:542,589) to these definitions - the link-time
   override INTEGRATION.md sections 2 and 3 describe.  Counters prove the calls really went
   through. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/daala_hip.h"

/* ODHIP_INTERPOSE_PASSTHROUGH=1: forward to the reference's own definition
   (the next one in symbol search order) instead of the *_hip entry point; used
   with the frame cache to time "batched pyramid only" on large frames, where
   one GPU round trip per 4-tap filter call would dominate. */
static int passthrough(void) {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("ODHIP_INTERPOSE_PASSTHROUGH");
    v = e && e[0] == '1';
  }
  return v;
}
#include <stdio.h>
/* dlopen handle of the reference library (dlsym(RTLD_NEXT) does not see
   libraries outside a dlopen'ed object's own dependency scope). */
static void *g_reference;
void odhip_interpose_set_reference(void *handle) {
  g_reference = handle;
}
static void *next_sym(const char *name) {
  /* dlopen'ed reference (python harness): its handle; a binary linked against
     the reference with this library in LD_PRELOAD (encoder_example): the next
     definition in load order. */
  void *p = g_reference ? dlsym(g_reference, name) : dlsym(RTLD_NEXT, name);
  if (!p) {
    fprintf(stderr, "interpose: no next definition of %s (%s)\n", name, dlerror());
    abort();
  }
  return p;
}
#define NEXT(type, name) ((type)next_sym(name))

long odhip_interposed_calls[6];

/* ODHIP_INTERPOSE_REPORT=1: print the call counters on stderr at exit (for
   processes the test cannot ask, i.e. the reference's encoder_example). */
__attribute__((destructor)) static void interpose_report(void) {
  const char *e = getenv("ODHIP_INTERPOSE_REPORT");
  if (e && e[0] == '1') {
    fprintf(stderr, "odhip_interposed_calls %ld %ld %ld %ld %ld %ld\n", odhip_interposed_calls[0],
     odhip_interposed_calls[1], odhip_interposed_calls[2], odhip_interposed_calls[3],
     odhip_interposed_calls[4], odhip_interposed_calls[5]);
  }
}

/* od_state_opt_vtbl_init (src/state.c:346-352): the reference's backend
   dispatch.  With ODHIP_INTERPOSE_VTBL=1 this is the load-time form of the one
   line of glue in INTEGRATION.md section 1: the reference's own initialisation
   runs first, then libdaalahip's ten 2-D transforms are written into
   opt_vtbl.fdct_2d / idct_2d - the shape of od_state_opt_vtbl_init_x86
   (src/x86/x86state.c:39-97).  The slot offsets inside od_state are known to
   ref_state_set_dct_vtbl (oracle/ref_encoder_driver.c, compiled with the
   reference's headers), not to this file. */
void od_state_opt_vtbl_init(void *state) {
  typedef void (*init_fn)(void *);
  typedef void (*set_fn)(void *, void **, void **);
  static init_fn next;
  const char *e;
  if (!next) next = NEXT(init_fn, "od_state_opt_vtbl_init");
  next(state);
  e = getenv("ODHIP_INTERPOSE_VTBL");
  if (e && e[0] == '1') {
    odhip_dct_func_2d fd[5];
    odhip_dct_func_2d id[5];
    set_fn set;
    if (odhip_init(0) != 0) {
      fprintf(stderr, "interpose: odhip_init failed\n");
      abort();
    }
    odhip_install_dct_vtbl(fd, id);
    set = g_reference ? (set_fn)dlsym(g_reference, "ref_state_set_dct_vtbl")
     : (set_fn)dlsym(RTLD_DEFAULT, "ref_state_set_dct_vtbl");
    if (!set) {
      fprintf(stderr, "interpose: ref_state_set_dct_vtbl not found\n");
      abort();
    }
    set(state, (void **)fd, (void **)id);
  }
}

void od_prefilter_split(od_coeff *c0, int stride, int bs, int f, int hfilter, int vfilter) {
  odhip_interposed_calls[0]++;
  if (passthrough()) {
    typedef void (*fn)(od_coeff *, int, int, int, int, int);
    static fn next;
    if (!next) next = NEXT(fn, "od_prefilter_split");
    next(c0, stride, bs, f, hfilter, vfilter);
    return;
  }
  od_prefilter_split_hip(c0, stride, bs, f, hfilter, vfilter);
}

void od_postfilter_split(od_coeff *c0, int stride, int bs, int f, int q, unsigned char *skip,
 int skip_stride, int hfilter, int vfilter) {
  odhip_interposed_calls[1]++;
  if (passthrough()) {
    typedef void (*fn)(od_coeff *, int, int, int, int, unsigned char *, int, int, int);
    static fn next;
    if (!next) next = NEXT(fn, "od_postfilter_split");
    next(c0, stride, bs, f, q, skip, skip_stride, hfilter, vfilter);
    return;
  }
  od_postfilter_split_hip(c0, stride, bs, f, q, skip, skip_stride, hfilter, vfilter);
}

static void interpose_load(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec);

void od_apply_prefilter_frame_sbs(od_coeff *c, int stride, int nhsb, int nvsb, int xdec,
 int ydec) {
  odhip_interposed_calls[2]++;
  interpose_load(c, stride, nhsb, nvsb, xdec);
  if (passthrough()) {
    typedef void (*fn)(od_coeff *, int, int, int, int, int);
    static fn next;
    if (!next) next = NEXT(fn, "od_apply_prefilter_frame_sbs");
    next(c, stride, nhsb, nvsb, xdec, ydec);
    return;
  }
  od_apply_prefilter_frame_sbs_hip(c, stride, nhsb, nvsb, xdec, ydec);
}

/* ---- mode 4: GPU decode check -------------------------------------------------------
   Inside the REAL reference decoder, at the moment it has decoded every block of a frame
   (state.dtmp = the dequantised coefficients, state.bsize = the partition, state.ctmp =
   its own block-by-block reconstruction) and is about to lap across superblock edges
   (src/decode.c:988-996): odhip_inverse_partition reconstructs the plane from dtmp and
   bsize alone, the reference finishes its own (od_apply_postfilter_frame_sbs, then the
   pixel conversion of od_coeff_to_ref_buf, src/state.c:1296-1304, applied here to its
   coefficients), and the two planes of pixels must be identical. */
static void *g_dec;
static int g_decode_check;
long odhip_interposed_decode[3];     /* planes checked, pixels compared, pixels that differ */

void odhip_interpose_enable_decode_check(void) {
  g_decode_check = 1;
}

int daala_decode_packet_in(void *dec, const void *dp) {
  typedef int (*fn)(void *, const void *);
  static fn next;
  if (!next) next = NEXT(fn, "daala_decode_packet_in");
  g_dec = dec;
  return next(dec, dp);
}

static int interpose_decode_check(od_coeff *c, int stride, int nhsb, int nvsb, int xdec,
 void (*reference)(od_coeff *, int, int, int, int, int, int, unsigned char *, int), int ydec, int q,
 unsigned char *skip, int skip_stride) {
  typedef int (*view_fn)(void *, const od_coeff *, const od_coeff **, const unsigned char **, int *, int *,
   int *);
  view_fn view;
  const od_coeff *d;
  const unsigned char *bsize;
  unsigned char *gpu;
  int bstride;
  int pic_w;
  int pic_h;
  int w;
  int h;
  long i;
  view = (view_fn)(g_reference ? dlsym(g_reference, "ref_state_recon_view")
   : dlsym(RTLD_DEFAULT, "ref_state_recon_view"));
  if (!view) {
    fprintf(stderr, "interpose: ref_state_recon_view not found\n");
    abort();
  }
  /* not one of the decoder's planes (an encoder in the same process laps its own) */
  if (view(g_dec, c, &d, &bsize, &bstride, &pic_w, &pic_h) < 0) return 0;
  w = nhsb << 6 >> xdec;
  h = nvsb << 6 >> xdec;
  if (stride != w) abort();
  gpu = (unsigned char *)malloc((size_t)w*h);
  if (odhip_inverse_partition_host(gpu, w, d, w, h, xdec, bsize, bstride, pic_w, pic_h) != 0) {
    fprintf(stderr, "interpose: odhip_inverse_partition_host failed\n");
    abort();
  }
  reference(c, stride, nhsb, nvsb, xdec, ydec, q, skip, skip_stride);
  odhip_interposed_decode[0]++;
  for (i = 0; i < (long)w*h; i++) {
    int v;
    v = ((c[i] + 8) >> 4) + 128;
    v = v < 0 ? 0 : v > 255 ? 255 : v;
    odhip_interposed_decode[1]++;
    if (v != gpu[i]) odhip_interposed_decode[2]++;
  }
  free(gpu);
  return 1;
}

/* ---- the deringing level search from batched passes (odhip_dering_cache) ------------
   ODHIP_INTERPOSE_DERING_CACHE=1: every od_dering call of the encoder's level search
   (src/encode.c:2787,:2826) is served by odhip_dering_cache_call; the frame boundary is
   the superblock-edge postfilter the encoder runs just before the search (:2670-2677) -
   in a reference build that is one line at :2697. */
static odhip_dering_cache *g_dering_cache;
long odhip_interposed_dering[2];      /* batched launches, calls served */

static int dering_cache_enabled(void) {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("ODHIP_INTERPOSE_DERING_CACHE");
    v = e && e[0] == '1';
  }
  return v;
}

void odhip_interpose_dering_stats(void) {
  odhip_dering_cache_stats(g_dering_cache, &odhip_interposed_dering[0], &odhip_interposed_dering[1]);
}

void od_apply_postfilter_frame_sbs(od_coeff *c, int stride, int nhsb, int nvsb, int xdec,
 int ydec, int q, unsigned char *skip, int skip_stride) {
  odhip_interposed_calls[3]++;
  if (dering_cache_enabled() && g_dering_cache) odhip_dering_cache_begin(g_dering_cache);
  if (g_decode_check && g_dec) {
    typedef void (*fn)(od_coeff *, int, int, int, int, int, int, unsigned char *, int);
    static fn next;
    if (!next) next = NEXT(fn, "od_apply_postfilter_frame_sbs");
    if (interpose_decode_check(c, stride, nhsb, nvsb, xdec, next, ydec, q, skip, skip_stride)) return;
  }
  if (passthrough()) {
    typedef void (*fn)(od_coeff *, int, int, int, int, int, int, unsigned char *, int);
    static fn next;
    if (!next) next = NEXT(fn, "od_apply_postfilter_frame_sbs");
    next(c, stride, nhsb, nvsb, xdec, ydec, q, skip, skip_stride);
    return;
  }
  od_apply_postfilter_frame_sbs_hip(c, stride, nhsb, nvsb, xdec, ydec, q, skip, skip_stride);
}

double pvq_search_rdo_double(const int16_t *xcoeff, int n, int k, od_coeff *ypulse, double g2,
 double pvq_norm_lambda, int prev_k) {
  odhip_interposed_calls[4]++;
  if (passthrough()) {
    typedef double (*fn)(const int16_t *, int, int, od_coeff *, double, double, int);
    static fn next;
    if (!next) next = NEXT(fn, "pvq_search_rdo_double");
    return next(xcoeff, n, k, ypulse, g2, pvq_norm_lambda, prev_k);
  }
  return od_pvq_search_rdo_double_hip(xcoeff, n, k, ypulse, g2, pvq_norm_lambda, prev_k);
}

/* ---- mode 2: the frame cache ------------------------------------------------
   When enabled (odhip_interpose_enable_cache), the interposed
   od_apply_prefilter_frame_sbs - the moment the reference has just filled a
   plane with (p - 128) << 4, src/encode.c:2568-2571 - first hands the plane to
   odhip_cache_load_plane (one batched GPU pyramid), then laps it as before.
   With odhip_install_cached_dct_vtbl bound into od_state.opt_vtbl, every later
   fdct_2d call on that plane is served from the cache. */
static odhip_frame_cache *g_cache;
static const od_coeff *g_bases[4];
static int g_nbases;
static int g_bands_on;      /* mode 3: the batched band stage behind pvq_theta */
static void *g_enc;         /* the encoder whose frame is being coded */
static int g_bands_frame;   /* the current frame's luma bands are loaded */
long odhip_interposed_theta[4];   /* served from the batch / left to the reference (r0 not null) /
                                     left to the reference (other reason) / searches the batch saved */

void odhip_interpose_enable_cache(int pic_w, int pic_h) {
  if (!g_cache) g_cache = odhip_cache_create();
  odhip_cache_set_picture(g_cache, pic_w, pic_h);
  odhip_cache_make_current(g_cache);
}

void odhip_interpose_cache_stats(long *hits, long *misses) {
  odhip_cache_stats(g_cache, hits, misses);
}

#include <time.h>
double odhip_interposed_load_ms;   /* wall time spent in the batched GPU pass (incl. PCIe both ways) */
static void interpose_load_timed(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec);
static void interpose_load(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec) {
  struct timespec a;
  struct timespec b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  interpose_load_timed(c, stride, nhsb, nvsb, xdec);
  clock_gettime(CLOCK_MONOTONIC, &b);
  odhip_interposed_load_ms += (b.tv_sec - a.tv_sec)*1e3 + (b.tv_nsec - a.tv_nsec)*1e-6;
}

static void interpose_load_timed(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec) {
  int slot;
  if (!g_cache) return;
  for (slot = 0; slot < g_nbases; slot++) if (g_bases[slot] == c) break;
  if (slot == g_nbases) {
    if (g_nbases == 4) return;
    g_bases[g_nbases++] = c;
  }
  odhip_cache_load_plane(g_cache, slot, c, stride, nhsb << 6 >> xdec, nvsb << 6 >> xdec, xdec);
  if (g_bands_on && slot == 0 && xdec == 0) {
    /* keyframe luma: the PVQ band stage of every block of every level, now, in one
       batch, with the quantiser set-up this encoder uses for this frame */
    typedef int (*setup_fn)(const void *, int *, int *, double *, unsigned char *, int16_t *, int16_t *);
    static odhip_quant qt;
    setup_fn setup;
    double lambda;
    g_bands_frame = 0;
    setup = (setup_fn)(g_reference ? dlsym(g_reference, "ref_enc_band_setup")
     : dlsym(RTLD_DEFAULT, "ref_enc_band_setup"));
    if (!setup || !g_enc) {
      fprintf(stderr, "interpose: no encoder to take the quantiser set-up from\n");
      abort();
    }
    if (setup(g_enc, &qt.quantizer, &qt.use_masking, &lambda, &qt.pvq_qm_q4[0][0], qt.qm, qt.qm_inv)) {
      if (odhip_cache_load_bands(g_cache, 0, &qt, lambda) != 0) {
        fprintf(stderr, "interpose: odhip_cache_load_bands failed\n");
        abort();
      }
      g_bands_frame = 1;
    }
  }
}

void odhip_interpose_enable_bands(void) {
  g_bands_on = 1;
}

void odhip_interpose_band_stats(long *hits, long *misses) {
  odhip_cache_band_stats(g_cache, hits, misses);
}

/* daala_encode_img_in (include/daala/daalaenc.h:118): remembers which encoder the
   following plane loads and block encodes belong to. */
int daala_encode_img_in(void *enc, void *img, int duration) {
  typedef int (*fn)(void *, void *, int);
  static fn next;
  if (!next) next = NEXT(fn, "daala_encode_img_in");
  g_enc = enc;
  g_bands_frame = 0;
  return next(enc, img, duration);
}

/* daala_encode_free (include/daala/daalaenc.h): the planes of this encoder are gone - the
   next encoder's planes take the cache slots from the start (a second encoder in one
   process used to find the four slots taken by the first one's buffers). */
void daala_encode_free(void *enc) {
  typedef void (*fn)(void *);
  static fn next;
  if (!next) next = NEXT(fn, "daala_encode_free");
  if (enc == g_enc) g_enc = NULL;
  g_nbases = 0;
  g_bands_frame = 0;
  next(enc);
}

/* od_pvq_encode (src/pvq_encoder.h:46-49, the boundary symbol of BASELINE.json): the
   reference's own definition runs; this wrapper only notes WHICH block its pvq_theta
   calls belong to (bx, by in 4x4 units as src/encode.c:1264-1265 passes them). */
static __thread int t_pli, t_bs, t_bx, t_by, t_band;
int od_pvq_encode(void *enc, od_coeff *ref, const od_coeff *in, od_coeff *out, int q0, int pli, int bs,
 const int16_t *beta, int nodesync, int is_keyframe, int q_scaling, int bx, int by, const int16_t *qm,
 const int16_t *qm_inv, int speed) {
  typedef int (*fn)(void *, od_coeff *, const od_coeff *, od_coeff *, int, int, int, const int16_t *, int,
   int, int, int, int, const int16_t *, const int16_t *, int);
  static fn next;
  if (!next) next = NEXT(fn, "od_pvq_encode");
  t_pli = pli;
  t_bs = bs;
  t_bx = bx;
  t_by = by;
  t_band = 0;
  return next(enc, ref, in, out, q0, pli, bs, beta, nodesync, is_keyframe, q_scaling, bx, by, qm, qm_inv,
   speed);
}

/* pvq_theta (src/pvq_encoder.c:333-641; file-static in the reference, an ordinary
   symbol of the test build).  This is the glue INTEGRATION.md section 7 puts at the
   top of that function: a keyframe luma band whose reference vector is null takes
   the no-reference path only (:452 fails, :571-609 runs), and every quantity of that
   path that does not depend on the entropy coder's adaptive state was computed for
   the whole frame in one batch (odhip_cache_load_bands).  What is left is what the
   reference keeps on the host: price the candidates with od_pvq_rate on the LIVE
   state, apply `cost <= best_cost`, the skip rule, and synthesise the winner with the
   reference's own od_gain_expand / od_pvq_synthesis_partial.  Every other band goes
   to the reference's pvq_theta untouched. */
int pvq_theta(od_coeff *out, const od_coeff *x0, const od_coeff *r0, int n, int q0, od_coeff *y,
 int *itheta, int *max_theta, int *vk, int16_t beta, double *skip_diff, int nodesync, int is_keyframe,
 int pli, const void *adapt, const int16_t *qm, const int16_t *qm_inv, double pvq_norm_lambda,
 int speed) {
  typedef int (*fn)(od_coeff *, const od_coeff *, const od_coeff *, int, int, od_coeff *, int *, int *,
   int *, int16_t, double *, int, int, int, const void *, const int16_t *, const int16_t *, double, int);
  typedef double (*rate_fn)(int, int, int, int, const void *, const od_coeff *, int, int, int, int, int);
  typedef int32_t (*expand_fn)(int32_t, int, int16_t);
  typedef void (*synth_fn)(od_coeff *, const od_coeff *, const int16_t *, int, int, int32_t, int32_t,
   int, int, const int16_t *);
  static fn next;
  static rate_fn rate;
  static expand_fn gain_expand;
  static synth_fn synthesis;
  odhip_band_cands c;
  const int band = t_band++;
  int i;
  if (!next) next = NEXT(fn, "pvq_theta");
  if (g_bands_on && g_bands_frame && is_keyframe && pli == 0 && t_pli == 0 && n <= 128) {
    int null_ref = 1;
    for (i = 0; i < n; i++) {
      if (r0[i]) {
        null_ref = 0;
        break;
      }
    }
    if (!null_ref) odhip_interposed_theta[1]++;
    else if (!odhip_cache_band(g_cache, 0, t_bs, t_bx >> t_bs, t_by >> t_bs, band, x0, &c)
     || c.n != n || c.q != q0 || c.beta != beta || c.flags[0] == 2 || c.flags[1] == 2) {
      odhip_interposed_theta[2]++;
    }
    else {
      od_coeff y_tmp[128];
      double best_cost;
      double best_dist;
      double skip_dist;
      int qg;
      int best_k;
      int s;
      if (!rate) {
        rate = NEXT(rate_fn, "od_pvq_rate");
        gain_expand = NEXT(expand_fn, "od_gain_expand");
        synthesis = NEXT(synth_fn, "od_pvq_synthesis_partial");
      }
      odhip_interposed_theta[0]++;
      /* :415-421 with a null reference on a keyframe: the null candidate */
      qg = 0;
      best_dist = c.dist0;
      best_cost = c.dist0 + pvq_norm_lambda*rate(0, 0, -1, 0, adapt, NULL, 0, n, is_keyframe, pli, speed);
      best_k = 0;
      *itheta = -1;
      *max_theta = 0;
      for (i = 0; i < n; i++) y[i] = 0;
      skip_dist = c.dist0;        /* :439: the same expression as :417 on a keyframe */
      /* :578-609: the (at most two) no-reference candidates, in gain order.  At the default
         complexity (speed == 0) od_pvq_rate runs the codeword coder on a copy of the live
         context per candidate: both candidates are priced in one call of the library's
         batched routine instead (odhip_pvq_rate_batch16: rate-only range coder,
         copy-on-touch CDF rows, the same doubles - tests/test_rate_host.py) */
      {
        double rates[2];
        rates[0] = rates[1] = 0;
        if (speed == 0) {
          const int16_t *ys[2];
          int ks[2];
          int qgs[2];
          int thetas[2];
          int tss[2];
          int nc;
          int map[2];
          nc = 0;
          for (s = 0; s < 2; s++) {
            if (c.flags[s] != 1) continue;
            ys[nc] = c.y[s];
            ks[nc] = c.k[s];
            qgs[nc] = c.gain[s];
            thetas[nc] = -1;
            tss[nc] = 0;
            map[nc++] = s;
          }
          if (nc) {
            double out[2];
            /* &adapt->pvq.pvq_codeword_ctx is at offset 0 of od_adapt_ctx (src/state.h:141-143) */
            if (odhip_pvq_rate_batch16(out, (const odhip_pvq_codeword_ctx *)adapt, nc, ys, ks, qgs, thetas, tss, n, 0,
             is_keyframe, pli) != 0) {
              fprintf(stderr, "interpose: odhip_pvq_rate_batch16 failed\n");
              abort();
            }
            for (i = 0; i < nc; i++) rates[map[i]] = out[i];
          }
        }
        for (s = 0; s < 2; s++) {
          double cost;
          if (c.flags[s] != 1) continue;
          odhip_interposed_theta[3]++;
          for (i = 0; i < n; i++) y_tmp[i] = c.y[s][i];
          if (speed != 0) {
            rates[s] = rate(c.gain[s], 0, -1, 0, adapt, y_tmp, c.k[s], n, is_keyframe, pli, speed);
          }
          else if (getenv("ODHIP_RATE_CHECK")) {
            /* every batched price against the reference's own od_pvq_rate */
            const double want = rate(c.gain[s], 0, -1, 0, adapt, y_tmp, c.k[s], n, is_keyframe, pli, speed);
            if (memcmp(&want, &rates[s], sizeof(want)) != 0) {
              fprintf(stderr, "interpose: batched rate %.17g != od_pvq_rate %.17g\n", rates[s], want);
              abort();
            }
          }
          cost = c.dist[s] + pvq_norm_lambda*rates[s];
          if (cost <= best_cost) {
            best_cost = cost;
            best_dist = c.dist[s];
            qg = c.gain[s];
            best_k = c.k[s];
            for (i = 0; i < n; i++) y[i] = y_tmp[i];
          }
        }
      }
      /* :611-633: skip rule and the decoder's synthesis */
      if (qg == 0) for (i = 0; i < n; i++) out[i] = 0;
      else {
        int16_t r16[128];
        for (i = 0; i < n; i++) r16[i] = 0;
        synthesis(out, y, r16, n, 1, gain_expand(qg << 8, q0, beta), 0, 0, 1, qm_inv);
      }
      *vk = best_k;
      *skip_diff += skip_dist - best_dist;
      return qg;
    }
  }
  return next(out, x0, r0, n, q0, y, itheta, max_theta, vk, beta, skip_diff, nodesync, is_keyframe, pli,
   adapt, qm, qm_inv, pvq_norm_lambda, speed);
}

/* od_dering, src/dering.c:252 (call sites src/encode.c:2787,2826): the function
   table argument is dropped, the HIP kernel implements what it would dispatch to. */
void od_dering(const void *vtbl, int16_t *y, int ystride, const int16_t *x, int xstride, int nhb, int nvb,
 int sbx, int sby, int nhsb, int nvsb, int xdec, int dir[8][8], int pli, unsigned char *bskip,
 int skip_stride, int threshold, int overlap, int coeff_shift) {
  odhip_interposed_calls[5]++;
  if (dering_cache_enabled()) {
    if (!g_dering_cache) {
      if (odhip_init(0) != 0 || !(g_dering_cache = odhip_dering_cache_create())) {
        fprintf(stderr, "interpose: odhip_dering_cache_create failed\n");
        abort();
      }
    }
    if (odhip_dering_cache_call(g_dering_cache, y, ystride, x, xstride, nhb, nvb, sbx, sby, nhsb, nvsb, xdec,
     dir, pli, bskip, skip_stride, threshold, overlap, coeff_shift) != 0) {
      fprintf(stderr, "interpose: odhip_dering_cache_call failed (no CPU fallback)\n");
      abort();
    }
    if (getenv("ODHIP_DERING_CHECK")) {
      /* every served superblock against the reference's own od_dering */
      typedef void (*fn)(const void *, int16_t *, int, const int16_t *, int, int, int, int, int, int, int, int,
       int (*)[8], int, unsigned char *, int, int, int, int);
      static fn next;
      int16_t want[64*64];
      int wdir[8][8];
      const int n = 64 >> xdec;
      int i;
      int j;
      if (!next) next = NEXT(fn, "od_dering");
      for (i = 0; i < 8; i++) for (j = 0; j < 8; j++) wdir[i][j] = dir[i][j];
      next(vtbl, want, n, x, xstride, nhb, nvb, sbx, sby, nhsb, nvsb, xdec, wdir, pli, bskip, skip_stride,
       threshold, overlap, coeff_shift);
      for (i = 0; i < n; i++) {
        for (j = 0; j < n; j++) {
          if (want[i*n + j] != y[i*ystride + j]) {
            fprintf(stderr, "interpose: dering cache mismatch pli %d sb (%d, %d) thr %d at (%d, %d)\n", pli, sbx,
             sby, threshold, i, j);
            abort();
          }
        }
      }
      for (i = 0; i < 8; i++) for (j = 0; j < 8; j++) {
        if (wdir[i][j] != dir[i][j]) {
          fprintf(stderr, "interpose: dering cache direction mismatch\n");
          abort();
        }
      }
    }
    return;
  }
  if (passthrough()) {
    typedef void (*fn)(const void *, int16_t *, int, const int16_t *, int, int, int, int, int, int, int, int,
     int (*)[8], int, unsigned char *, int, int, int, int);
    static fn next;
    if (!next) next = NEXT(fn, "od_dering");
    next(vtbl, y, ystride, x, xstride, nhb, nvb, sbx, sby, nhsb, nvsb, xdec, dir, pli, bskip, skip_stride,
     threshold, overlap, coeff_shift);
    return;
  }
  od_dering_hip(y, ystride, x, xstride, nhb, nvb, sbx, sby, nhsb, nvsb, xdec, dir, pli, bskip, skip_stride,
   threshold, overlap, coeff_shift);
}
