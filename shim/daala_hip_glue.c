/* shim/daala_hip_glue.c - the reference-side binding of libdaalahip (see daala_hip_glue.h and
   README.md in this directory).

   Defines the reference's own symbol names for the surfaces of the block-transform path and
   forwards them to libdaalahip; which surfaces are bound is what the host configured with
   odhip_glue_configure().  Placed in front of libdaala in symbol search order (LD_PRELOAD, link
   order, or RTLD_GLOBAL before a dlopen of the reference), the dynamic linker binds every call
   inside the UNMODIFIED encoder / decoder to these definitions (src/encode.c:1489,1760,1789,
   2571,2675,2787,2826; src/pvq_encoder.c:542,589; src/state.c:346) - the link-time form of the
   glue INTEGRATION.md sections 1-3 and 7 spell out line by line; the shape of the backend
   installation is od_state_opt_vtbl_init_x86's (src/x86/x86state.c:39-97).

   Plain C, no HIP: everything device-side is behind include/daala_hip.h.  There is no CPU
   fallback in here: a bound surface whose GPU call fails aborts. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include <pthread.h>
#include "../include/daala_hip.h"
#include "daala_hip_glue.h"

/* ---- configuration ------------------------------------------------------------------- */
static odhip_glue_config g_cfg;
static int g_cfg_set;

void odhip_glue_default_config(odhip_glue_config *cfg) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->bind_filters = cfg->bind_search = cfg->bind_dering = cfg->bind_dct_vtbl = 1;
}

/* A host that embeds the glue in a library of its own (tests/interpose does) may define this
   to supply the configuration in force until odhip_glue_configure() is called. */
__attribute__((weak)) void odhip_glue_startup_config(odhip_glue_config *cfg);

static odhip_glue_config *cfg(void) {
  if (!g_cfg_set) {
    odhip_glue_default_config(&g_cfg);
    if (odhip_glue_startup_config) odhip_glue_startup_config(&g_cfg);
    g_cfg_set = 1;
  }
  return &g_cfg;
}

int odhip_glue_configure(const odhip_glue_config *c) {
  if (!c) return ODHIP_EINVAL;
  g_cfg = *c;
  g_cfg_set = 1;
  if (c->frame_cache || c->band_cache) odhip_glue_enable_frame_cache(c->pic_w, c->pic_h);
  /* the caches themselves are created by the thread that first loads a plane (thread_cache): a
     device that cannot be initialised is reported here, not by an abort inside a cache constructor */
  return odhip_init(c->device);
}

void odhip_glue_get_config(odhip_glue_config *c) {
  *c = *cfg();
}

/* dlopen handle of the reference library (dlsym(RTLD_NEXT) does not see
   libraries outside a dlopen'ed object's own dependency scope). */
static void *g_reference;
void odhip_glue_set_reference(void *handle) {
  g_reference = handle;
}
static void *next_sym(const char *name) {
  /* dlopen'ed reference (python harness): its handle; a binary linked against
     the reference with this library in LD_PRELOAD (encoder_example): the next
     definition in load order. */
  void *p = g_reference ? dlsym(g_reference, name) : dlsym(RTLD_NEXT, name);
  if (!p) {
    fprintf(stderr, "daala_hip_glue: no next definition of %s (%s)\n", name, dlerror());
    abort();
  }
  return p;
}
#define NEXT(type, name) ((type)next_sym(name))

/* Counters (odhip_glue_get_stats): proof that the calls really went through.  The hot ones are
   counted in thread-local copies and folded into the process totals at frame boundaries
   (odhip_glue_flush_stats): sixteen encoder threads incrementing one shared cache line a million
   times per frame ran at 5.4 frames/s instead of 13.6 (profiles/r4_encode_mode_300frames.json). */
long odhip_glue_calls[6];
static __thread long t_calls[6];
static __thread long t_theta[4];
static __thread long t_dist[2];
long odhip_glue_dist[2];       /* od_compute_dist calls served from the dering cache / left to the C function */
static __thread double t_dering_ms;
static __thread double t_theta_ms;
void odhip_glue_flush_stats(void);

/* Per-thread caches are freed when their thread exits (a host with short-lived encoder threads
   would otherwise leak their pinned and device memory): a pthread key whose destructor folds the
   thread's counters into the process totals and destroys its frame / dering cache. */
static pthread_key_t g_thread_key;
static pthread_once_t g_thread_key_once = PTHREAD_ONCE_INIT;
static void thread_teardown(void *unused);
static void thread_key_make(void) {
  (void)pthread_key_create(&g_thread_key, thread_teardown);
}
static void thread_register(void) {
  (void)pthread_once(&g_thread_key_once, thread_key_make);
  (void)pthread_setspecific(g_thread_key, (void *)1);
}
/* Every bound surface counts its calls per thread: the first one registers the thread, so that a
   thread that never creates a cache still folds its counters at exit. */
static __thread int t_registered;
#define COUNT_CALL(i) \
  do { \
    if (!t_registered) { \
      t_registered = 1; \
      thread_register(); \
    } \
    t_calls[i]++; \
  } while (0)
/* Running totals of this thread's caches as of its last flush (odhip_glue_flush_stats): fdct hits /
   misses, band hits / misses, dering launches / served. */
static __thread long t_cache_seen[6];
static void cache_seen_reset(int from, int to) {
  int i;
  for (i = from; i < to; i++) t_cache_seen[i] = 0;
}

/* od_state_opt_vtbl_init (src/state.c:346-352): the reference's backend
   dispatch.  With bind_dct_vtbl this is the load-time form of the one
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
  if (!next) next = NEXT(init_fn, "od_state_opt_vtbl_init");
  next(state);
  if (cfg()->bind_dct_vtbl) {
    odhip_dct_func_2d fd[5];
    odhip_dct_func_2d id[5];
    set_fn set;
    if (odhip_init(cfg()->device) != 0) {
      fprintf(stderr, "daala_hip_glue: odhip_init failed\n");
      abort();
    }
    odhip_install_dct_vtbl(fd, id);
    set = g_reference ? (set_fn)dlsym(g_reference, "ref_state_set_dct_vtbl")
     : (set_fn)dlsym(RTLD_DEFAULT, "ref_state_set_dct_vtbl");
    if (!set) {
      fprintf(stderr, "daala_hip_glue: ref_state_set_dct_vtbl not found\n");
      abort();
    }
    set(state, (void **)fd, (void **)id);
  }
}

void od_prefilter_split(od_coeff *c0, int stride, int bs, int f, int hfilter, int vfilter) {
  COUNT_CALL(0);
  if (!cfg()->bind_filters) {
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
  COUNT_CALL(1);
  if (!cfg()->bind_filters) {
    typedef void (*fn)(od_coeff *, int, int, int, int, unsigned char *, int, int, int);
    static fn next;
    if (!next) next = NEXT(fn, "od_postfilter_split");
    next(c0, stride, bs, f, q, skip, skip_stride, hfilter, vfilter);
    return;
  }
  od_postfilter_split_hip(c0, stride, bs, f, q, skip, skip_stride, hfilter, vfilter);
}

static void glue_load_plane(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec);

void od_apply_prefilter_frame_sbs(od_coeff *c, int stride, int nhsb, int nvsb, int xdec,
 int ydec) {
  COUNT_CALL(2);
  glue_load_plane(c, stride, nhsb, nvsb, xdec);
  if (!cfg()->bind_filters) {
    typedef void (*fn)(od_coeff *, int, int, int, int, int);
    static fn next;
    if (!next) next = NEXT(fn, "od_apply_prefilter_frame_sbs");
    next(c, stride, nhsb, nvsb, xdec, ydec);
    return;
  }
  od_apply_prefilter_frame_sbs_hip(c, stride, nhsb, nvsb, xdec, ydec);
}

/* A host that embeds the glue may take over od_apply_postfilter_frame_sbs for a plane (returns
   nonzero when it did): tests/interpose checks the decoder's reconstruction there. */
typedef void (*odhip_glue_postfilter_fn)(od_coeff *, int, int, int, int, int, int, unsigned char *, int);
__attribute__((weak)) int odhip_glue_hook_postfilter_frame(od_coeff *c, int stride, int nhsb, int nvsb, int xdec,
 int ydec, int q, unsigned char *skip, int skip_stride, odhip_glue_postfilter_fn reference);

/* ---- the deringing level search from batched passes (odhip_dering_cache) ------------
   dering_cache: every od_dering call of the encoder's level search
   (src/encode.c:2787,:2826) is served by odhip_dering_cache_call; the frame boundary is
   the superblock-edge postfilter the encoder runs just before the search (:2670-2677) -
   in a reference build that is one line at :2697. */
static __thread odhip_dering_cache *g_dering_cache;   /* per host thread: see "Threads" in daala_hip_glue.h */
static int dering_cache_enabled(void) {
  return cfg()->dering_cache;
}

void odhip_glue_enable_dering_cache(void) {
  cfg()->dering_cache = 1;
}

/* ---- the level search's distortions from the same passes (dist_cache) -------------------
   The six od_compute_dist calls per superblock of the level search (src/encode.c:2776-2801)
   compare the SOURCE picture with the unfiltered reconstruction and with od_dering's five outputs.
   The frame cache holds the source picture, the dering cache the outputs: at the frame boundary
   the luma source goes to odhip_dering_cache_set_source, every luma pass then also computes the
   distortion parts of its whole output, and odhip_glue_compute_dist - called in front of the
   reference's file-static od_compute_dist by the ten lines of oracle/Makefile's DISTGLUE (the glue
   a maintainer adds) - answers the calls the cache can vouch for:
   those whose x IS the source superblock and whose y IS the cached output of the od_dering call
   just before (both compared sample by sample inside odhip_dering_cache_dist).  Every other
   od_compute_dist call of the encoder (block-size RDO, the unfiltered candidate) runs the C code. */
static __thread odhip_frame_cache *g_cache;
static __thread void *g_enc;
static __thread int t_dd_valid, t_dd_sbx, t_dd_sby, t_dd_thr;

static odhip_dering_cache *thread_dering_cache(void) {
  if (!g_dering_cache) {
    if (odhip_init(cfg()->device) != 0 || !(g_dering_cache = odhip_dering_cache_create())) {
      fprintf(stderr, "daala_hip_glue: odhip_dering_cache_create failed (no CPU fallback)\n");
      abort();
    }
    cache_seen_reset(4, 6);
    thread_register();
  }
  return g_dering_cache;
}

static void glue_dist_source(void) {
  typedef void (*setup_fn)(const void *, int *, int *);
  static setup_fn setup;
  const uint8_t *h_px;
  const uint8_t *d_px;
  int w;
  int h;
  int masking;
  int flat;
  t_dd_valid = 0;
  if (!cfg()->dist_cache || !g_cache || !g_enc || !g_dering_cache) return;
  if (odhip_cache_plane_pixels(g_cache, 0, &h_px, &d_px, &w, &h) != 0) return;
  if (!setup) {
    setup = (setup_fn)(g_reference ? dlsym(g_reference, "ref_enc_dist_setup")
     : dlsym(RTLD_DEFAULT, "ref_enc_dist_setup"));
    if (!setup) return;
  }
  setup(g_enc, &masking, &flat);
  (void)odhip_dering_cache_set_source(g_dering_cache, h_px, d_px, w, masking, flat);
}

int odhip_glue_compute_dist(const od_coeff *x, const od_coeff *y, int n, int use_masking, int flat_qm,
 int coded_quantizer, double *out) {
  if (cfg()->dist_cache && g_dering_cache && t_dd_valid && n == 64) {
    t_dd_valid = 0;
    if (odhip_dering_cache_dist(g_dering_cache, x, y, n, t_dd_sbx, t_dd_sby, t_dd_thr, use_masking, flat_qm,
     coded_quantizer, out)) {
      t_dist[0]++;
      return 1;
    }
  }
  t_dist[1]++;
  return 0;
}

int odhip_glue_check_dist(void) {
  return cfg()->check_dist;
}

void od_apply_postfilter_frame_sbs(od_coeff *c, int stride, int nhsb, int nvsb, int xdec,
 int ydec, int q, unsigned char *skip, int skip_stride) {
  COUNT_CALL(3);
  if (dering_cache_enabled()) {
    /* the frame boundary of the level search (the encoder laps every plane just before it) */
    odhip_dering_cache_begin(thread_dering_cache());
    glue_dist_source();
  }
  if (odhip_glue_hook_postfilter_frame) {
    static odhip_glue_postfilter_fn next;
    if (!next) next = NEXT(odhip_glue_postfilter_fn, "od_apply_postfilter_frame_sbs");
    if (odhip_glue_hook_postfilter_frame(c, stride, nhsb, nvsb, xdec, ydec, q, skip, skip_stride, next)) return;
  }
  if (!cfg()->bind_filters) {
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
  COUNT_CALL(4);
  if (!cfg()->bind_search) {
    typedef double (*fn)(const int16_t *, int, int, od_coeff *, double, double, int);
    static fn next;
    if (!next) next = NEXT(fn, "pvq_search_rdo_double");
    return next(xcoeff, n, k, ypulse, g2, pvq_norm_lambda, prev_k);
  }
  return od_pvq_search_rdo_double_hip(xcoeff, n, k, ypulse, g2, pvq_norm_lambda, prev_k);
}

/* ---- the frame cache ----------------------------------------------------------
   When enabled (odhip_glue_enable_frame_cache), the interposed
   od_apply_prefilter_frame_sbs - the moment the reference has just filled a
   plane with (p - 128) << 4, src/encode.c:2568-2571 - first hands the plane to
   odhip_cache_load_plane (one batched GPU pyramid), then laps it as before.
   With odhip_install_cached_dct_vtbl bound into od_state.opt_vtbl, every later
   fdct_2d call on that plane is served from the cache. */
static __thread const od_coeff *g_bases[4];
static __thread int g_nbases;
#define g_bands_on (cfg()->band_cache)   /* the batched band stage behind pvq_theta */
/* (g_enc, declared with the distortion binding above: the encoder whose frame is being coded) */
static __thread int g_bands_frame;   /* the current frame's luma bands are loaded */
long odhip_glue_theta[4];   /* served from the batch / left to the reference (r0 not null) /
                               left to the reference (other reason) / searches the batch saved */

/* The calling thread's frame cache (created on its first use in that thread). */
static odhip_frame_cache *thread_cache(void) {
  if (!g_cache) {
    if (odhip_init(cfg()->device) != 0 || !(g_cache = odhip_cache_create())) {
      fprintf(stderr, "daala_hip_glue: odhip_cache_create failed (no CPU fallback)\n");
      abort();
    }
    odhip_cache_set_picture(g_cache, cfg()->pic_w, cfg()->pic_h);
    odhip_cache_make_current(g_cache);
    /* a new cache counts from zero: what was seen of an earlier one is forgotten HERE (not inferred
       from the totals going backwards, which a busy new cache never does) */
    cache_seen_reset(0, 4);
    thread_register();
  }
  return g_cache;
}

/* The cache itself is created by the thread that first loads a plane (glue_load_plane_timed): the
   configuring thread of a multi-threaded host (encode_job.py's main thread) never encodes. */
void odhip_glue_enable_frame_cache(int pic_w, int pic_h) {
  cfg()->frame_cache = 1;
  cfg()->pic_w = pic_w;
  cfg()->pic_h = pic_h;
  if (g_cache) odhip_cache_set_picture(g_cache, pic_w, pic_h);
}

/* The transforms to put into od_state.opt_vtbl when the frame cache is on: fdct_2d served from
   the batched pyramid, idct_2d the per-call HIP ones (src/state.h:128-129).  A host that keeps
   the reference's C idct_2d passes NULL for idct. */
void odhip_glue_cached_dct_vtbl(odhip_dct_func_2d fdct[5], odhip_dct_func_2d idct[5]) {
  odhip_install_cached_dct_vtbl(fdct, idct);
}

double odhip_glue_batch_ms;   /* wall time spent in the batched GPU passes (incl. PCIe both ways), all threads */
double odhip_glue_dering_ms;  /* ... inside odhip_dering_cache_call (launches, copies and served superblocks) */
double odhip_glue_theta_ms;   /* ... inside the pvq_theta calls served from the band cache (incl. batched pricing) */
static volatile int g_ms_lock;
static double ms_since(const struct timespec *a) {
  struct timespec b;
  clock_gettime(CLOCK_MONOTONIC, &b);
  return (b.tv_sec - a->tv_sec)*1e3 + (b.tv_nsec - a->tv_nsec)*1e-6;
}
static void ms_add(double *acc, const struct timespec *a) {
  const double ms = ms_since(a);
  while (__atomic_exchange_n(&g_ms_lock, 1, __ATOMIC_ACQUIRE)) {
  }
  *acc += ms;
  __atomic_store_n(&g_ms_lock, 0, __ATOMIC_RELEASE);
}
/* Cache figures of every thread, folded as deltas (a thread's caches keep running totals). */
static long g_cache_tot[6];            /* fdct hits / misses, band hits / misses, dering launches / served */
/* The calling thread's counters and timers into the process totals. */
void odhip_glue_flush_stats(void) {
  int i;
  long cur[6];
  memset(cur, 0, sizeof(cur));
  for (i = 0; i < 6; i++) cur[i] = t_cache_seen[i];
  if (g_cache) {
    odhip_cache_stats(g_cache, &cur[0], &cur[1]);
    odhip_cache_band_stats(g_cache, &cur[2], &cur[3]);
  }
  if (g_dering_cache) odhip_dering_cache_stats(g_dering_cache, &cur[4], &cur[5]);
  while (__atomic_exchange_n(&g_ms_lock, 1, __ATOMIC_ACQUIRE)) {
  }
  for (i = 0; i < 6; i++) {
    /* (t_cache_seen is zeroed where a cache is created or destroyed: cur >= seen always) */
    g_cache_tot[i] += cur[i] - t_cache_seen[i];
    t_cache_seen[i] = cur[i];
  }
  for (i = 0; i < 6; i++) {
    odhip_glue_calls[i] += t_calls[i];
    t_calls[i] = 0;
  }
  for (i = 0; i < 4; i++) {
    odhip_glue_theta[i] += t_theta[i];
    t_theta[i] = 0;
  }
  for (i = 0; i < 2; i++) {
    odhip_glue_dist[i] += t_dist[i];
    t_dist[i] = 0;
  }
  odhip_glue_dering_ms += t_dering_ms;
  odhip_glue_theta_ms += t_theta_ms;
  t_dering_ms = t_theta_ms = 0;
  __atomic_store_n(&g_ms_lock, 0, __ATOMIC_RELEASE);
}
static void glue_load_plane_locked(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec);
static void glue_load_plane(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec) {
  struct timespec a;
  struct timespec b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  glue_load_plane_locked(c, stride, nhsb, nvsb, xdec);
  (void)b;
  ms_add(&odhip_glue_batch_ms, &a);
  odhip_glue_flush_stats();
}

/* Advisory lock around the batched GPU pass (odhip_glue_config.gpu_pass_lock). */
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
/* flock() excludes open file descriptions, i.e. other PROCESSES; the threads of one process share
   the description, so they are excluded from each other by a mutex taken first (and released
   last: the flock is never dropped while a sibling thread is inside its pass). */
static int g_lock_fd = -1;
static pthread_once_t g_lock_once = PTHREAD_ONCE_INIT;
static pthread_mutex_t g_lock_mutex = PTHREAD_MUTEX_INITIALIZER;
static void gpu_pass_lock_open(void) {
  char path[64];
  snprintf(path, sizeof(path), "/tmp/odhip_glue_gpu%d.lock", cfg()->device);
  g_lock_fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
}
static void gpu_pass_lock(int on) {
  if (!cfg()->gpu_pass_lock) return;
  (void)pthread_once(&g_lock_once, gpu_pass_lock_open);
  if (on) {
    (void)pthread_mutex_lock(&g_lock_mutex);
    if (g_lock_fd >= 0) (void)flock(g_lock_fd, LOCK_EX);
  }
  else {
    if (g_lock_fd >= 0) (void)flock(g_lock_fd, LOCK_UN);
    (void)pthread_mutex_unlock(&g_lock_mutex);
  }
}

static void glue_load_plane_timed(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec);
static void glue_load_plane_locked(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec) {
  gpu_pass_lock(1);
  glue_load_plane_timed(c, stride, nhsb, nvsb, xdec);
  gpu_pass_lock(0);
}

static void glue_load_plane_timed(const od_coeff *c, int stride, int nhsb, int nvsb, int xdec) {
  int slot;
  if (!cfg()->frame_cache && !cfg()->band_cache) return;
  (void)thread_cache();
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
    static __thread odhip_quant qt;
    setup_fn setup;
    double lambda;
    g_bands_frame = 0;
    setup = (setup_fn)(g_reference ? dlsym(g_reference, "ref_enc_band_setup")
     : dlsym(RTLD_DEFAULT, "ref_enc_band_setup"));
    if (!setup || !g_enc) {
      fprintf(stderr, "daala_hip_glue: no encoder to take the quantiser set-up from\n");
      abort();
    }
    if (setup(g_enc, &qt.quantizer, &qt.use_masking, &lambda, &qt.pvq_qm_q4[0][0], qt.qm, qt.qm_inv)) {
      if (odhip_cache_load_bands(g_cache, 0, &qt, lambda) != 0) {
        fprintf(stderr, "daala_hip_glue: odhip_cache_load_bands failed\n");
        abort();
      }
      g_bands_frame = 1;
    }
  }
}

void odhip_glue_enable_band_cache(void) {
  cfg()->band_cache = 1;
}

void odhip_glue_get_stats(odhip_glue_stats *st) {
  int i;
  odhip_glue_flush_stats();
  memset(st, 0, sizeof(*st));
  for (i = 0; i < 6; i++) st->calls[i] = odhip_glue_calls[i];
  for (i = 0; i < 4; i++) st->theta[i] = odhip_glue_theta[i];
  st->fdct_hits = g_cache_tot[0];
  st->fdct_misses = g_cache_tot[1];
  st->band_hits = g_cache_tot[2];
  st->band_misses = g_cache_tot[3];
  st->dering_launches = g_cache_tot[4];
  st->dering_served = g_cache_tot[5];
  st->batch_ms = odhip_glue_batch_ms;
  st->dering_ms = odhip_glue_dering_ms;
  st->dist_served = odhip_glue_dist[0];
  st->dist_left = odhip_glue_dist[1];
  st->theta_ms = odhip_glue_theta_ms;
}

static void thread_teardown(void *unused) {
  (void)unused;
  odhip_glue_flush_stats();
  if (g_cache) {
    odhip_cache_destroy(g_cache);
    g_cache = NULL;
  }
  if (g_dering_cache) {
    odhip_dering_cache_destroy(g_dering_cache);
    g_dering_cache = NULL;
  }
  memset(t_cache_seen, 0, sizeof(t_cache_seen));
  g_nbases = 0;
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
  odhip_glue_flush_stats();
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

/* od_pvq_synthesis_partial (src/pvq.h:164): called per band by the decoder's pvq_decode_partition
   (src/pvq_decoder.c:87) and by pvq_theta (src/pvq_encoder.c:631). */
long odhip_glue_synth_calls;      /* calls bound to the GPU so far */
void od_pvq_synthesis_partial(od_coeff *xcoeff, const od_coeff *ypulse, const int16_t *r16, int n, int noref,
 int32_t g, int32_t theta, int m, int s, const int16_t *qm_inv) {
  typedef void (*fn)(od_coeff *, const od_coeff *, const int16_t *, int, int, int32_t, int32_t, int, int,
   const int16_t *);
  static fn next;
  if (!cfg()->bind_synthesis) {
    if (!next) next = NEXT(fn, "od_pvq_synthesis_partial");
    next(xcoeff, ypulse, r16, n, noref, g, theta, m, s, qm_inv);
    return;
  }
  __atomic_add_fetch(&odhip_glue_synth_calls, 1, __ATOMIC_RELAXED);
  od_pvq_synthesis_partial_hip(xcoeff, ypulse, r16, n, noref, g, theta, m, s, qm_inv);
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
typedef double (*theta_rate_fn)(int, int, int, int, const void *, const od_coeff *, int, int, int, int, int);
typedef int32_t (*theta_expand_fn)(int32_t, int, int16_t);
typedef void (*theta_synth_fn)(od_coeff *, const od_coeff *, const int16_t *, int, int, int32_t, int32_t,
 int, int, const int16_t *);
static theta_rate_fn g_theta_rate;
static theta_expand_fn g_theta_gain_expand;
static theta_synth_fn g_theta_synthesis;
static pthread_once_t g_theta_syms_once = PTHREAD_ONCE_INIT;
static void theta_syms_resolve(void) {
  g_theta_rate = NEXT(theta_rate_fn, "od_pvq_rate");
  g_theta_gain_expand = NEXT(theta_expand_fn, "od_gain_expand");
  g_theta_synthesis = NEXT(theta_synth_fn, "od_pvq_synthesis_partial");
}

int pvq_theta(od_coeff *out, const od_coeff *x0, const od_coeff *r0, int n, int q0, od_coeff *y,
 int *itheta, int *max_theta, int *vk, int16_t beta, double *skip_diff, int nodesync, int is_keyframe,
 int pli, const void *adapt, const int16_t *qm, const int16_t *qm_inv, double pvq_norm_lambda,
 int speed) {
  typedef int (*fn)(od_coeff *, const od_coeff *, const od_coeff *, int, int, od_coeff *, int *, int *,
   int *, int16_t, double *, int, int, int, const void *, const int16_t *, const int16_t *, double, int);
  static fn next;
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
    if (!null_ref) t_theta[1]++;
    else if (!odhip_cache_band(g_cache, 0, t_bs, t_bx >> t_bs, t_by >> t_bs, band, x0, &c)
     || c.n != n || c.q != q0 || c.beta != beta || c.flags[0] == 2 || c.flags[1] == 2) {
      t_theta[2]++;
    }
    else {
      od_coeff y_tmp[128];
      double best_cost;
      double best_dist;
      double skip_dist;
      int qg;
      int best_k;
      int s;
      /* three pointers, resolved together exactly once: sixteen encoder threads reach their first
         served band at the same time (a single `if (!rate)` guard let a second thread see `rate`
         set while the other two were still NULL) */
      (void)pthread_once(&g_theta_syms_once, theta_syms_resolve);
      const theta_rate_fn rate = g_theta_rate;
      const theta_expand_fn gain_expand = g_theta_gain_expand;
      const theta_synth_fn synthesis = g_theta_synthesis;
      struct timespec t_a;
      clock_gettime(CLOCK_MONOTONIC, &t_a);
      t_theta[0]++;
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
              fprintf(stderr, "daala_hip_glue: odhip_pvq_rate_batch16 failed\n");
              abort();
            }
            for (i = 0; i < nc; i++) rates[map[i]] = out[i];
          }
        }
        for (s = 0; s < 2; s++) {
          double cost;
          if (c.flags[s] != 1) continue;
          t_theta[3]++;
          for (i = 0; i < n; i++) y_tmp[i] = c.y[s][i];
          if (speed != 0) {
            rates[s] = rate(c.gain[s], 0, -1, 0, adapt, y_tmp, c.k[s], n, is_keyframe, pli, speed);
          }
          else if (cfg()->check_rates) {
            /* every batched price against the reference's own od_pvq_rate */
            const double want = rate(c.gain[s], 0, -1, 0, adapt, y_tmp, c.k[s], n, is_keyframe, pli, speed);
            if (memcmp(&want, &rates[s], sizeof(want)) != 0) {
              fprintf(stderr, "daala_hip_glue: batched rate %.17g != od_pvq_rate %.17g\n", rates[s], want);
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
      t_theta_ms += ms_since(&t_a);
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
  COUNT_CALL(5);
  if (dering_cache_enabled()) {
    (void)thread_dering_cache();
    struct timespec t_a;
    int rc_d;
    clock_gettime(CLOCK_MONOTONIC, &t_a);
    rc_d = odhip_dering_cache_call(g_dering_cache, y, ystride, x, xstride, nhb, nvb, sbx, sby, nhsb, nvsb, xdec,
     dir, pli, bskip, skip_stride, threshold, overlap, coeff_shift);
    t_dering_ms += ms_since(&t_a);
    /* the od_compute_dist call that follows a luma call of the level search compares its output */
    t_dd_valid = pli == 0 && rc_d == 0;
    t_dd_sbx = sbx;
    t_dd_sby = sby;
    t_dd_thr = threshold;
    if (rc_d != 0) {
      fprintf(stderr, "daala_hip_glue: odhip_dering_cache_call failed (no CPU fallback)\n");
      abort();
    }
    if (cfg()->check_dering) {
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
            fprintf(stderr, "daala_hip_glue: dering cache mismatch pli %d sb (%d, %d) thr %d at (%d, %d)\n", pli, sbx,
             sby, threshold, i, j);
            abort();
          }
        }
      }
      for (i = 0; i < 8; i++) for (j = 0; j < 8; j++) {
        if (wdir[i][j] != dir[i][j]) {
          fprintf(stderr, "daala_hip_glue: dering cache direction mismatch\n");
          abort();
        }
      }
    }
    return;
  }
  if (!cfg()->bind_dering) {
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
