/* tests/interpose/interpose.c - TEST INFRASTRUCTURE.

   The binding itself is the shim (shim/daala_hip_glue.c: the reference's own symbol names
   forwarded to libdaalahip, configured by explicit calls); it is included textually so that ONE
   preloadable library carries the shim and what only the tests need on top of it:

   * the ODHIP_INTERPOSE_* / ODHIP_*_CHECK environment switches of the test harness, mapped onto
     odhip_glue_configure() once at start-up (child processes such as the reference's
     encoder_example cannot be configured by a call);
   * the counters under the names the tests read, and a report on stderr at exit;
   * the GPU decode check inside the real decoder (odhip_glue_hook_postfilter_frame). */
#include "../../shim/daala_hip_glue.c"

static int env_on(const char *name) {
  const char *e = getenv(name);
  return e && e[0] == '1';
}

/* ODHIP_INTERPOSE_PASSTHROUGH=1: the per-call surfaces forward to the reference's own
   definitions (used with the frame cache on large frames, where one GPU round trip per 4-tap
   filter call would dominate); ODHIP_INTERPOSE_VTBL=1: the ten 2-D transforms are installed
   from od_state_opt_vtbl_init; ODHIP_INTERPOSE_DERING_CACHE=1: the deringing level search from
   batched passes; ODHIP_RATE_CHECK / ODHIP_DERING_CHECK: cross-checks against the reference. */
void odhip_glue_startup_config(odhip_glue_config *c) {
  const int pass = env_on("ODHIP_INTERPOSE_PASSTHROUGH");
  c->bind_filters = c->bind_search = c->bind_dering = !pass;
  c->bind_dct_vtbl = env_on("ODHIP_INTERPOSE_VTBL");
  /* ODHIP_INTERPOSE_SYNTHESIS=1: od_pvq_synthesis_partial per call on the GPU (encoder and decoder) */
  c->bind_synthesis = env_on("ODHIP_INTERPOSE_SYNTHESIS");
  c->dering_cache = env_on("ODHIP_INTERPOSE_DERING_CACHE");
  c->check_rates = getenv("ODHIP_RATE_CHECK") != NULL;
  c->check_dering = getenv("ODHIP_DERING_CHECK") != NULL;
  /* ODHIP_INTERPOSE_DIST_CACHE=1 (with the dering cache and the frame cache, and a reference build
     that carries the DISTGLUE lines): the level search's od_compute_dist calls from the batched
     passes; ODHIP_DIST_CHECK: every served value against the C function */
  c->dist_cache = env_on("ODHIP_INTERPOSE_DIST_CACHE");
  c->check_dist = getenv("ODHIP_DIST_CHECK") != NULL;
}

extern long odhip_interposed_calls[6] __attribute__((alias("odhip_glue_calls")));
extern long odhip_interposed_theta[4] __attribute__((alias("odhip_glue_theta")));
extern double odhip_interposed_load_ms __attribute__((alias("odhip_glue_batch_ms")));
extern long odhip_interposed_dist[2] __attribute__((alias("odhip_glue_dist")));
long odhip_interposed_dering[2];      /* batched launches, calls served */

/* ODHIP_INTERPOSE_REPORT=1: print the call counters on stderr at exit (for
   processes the test cannot ask, i.e. the reference's encoder_example). */
__attribute__((destructor)) static void interpose_report(void) {
  odhip_glue_flush_stats();
  if (env_on("ODHIP_INTERPOSE_REPORT")) {
    fprintf(stderr, "odhip_interposed_calls %ld %ld %ld %ld %ld %ld\n", odhip_glue_calls[0],
     odhip_glue_calls[1], odhip_glue_calls[2], odhip_glue_calls[3], odhip_glue_calls[4], odhip_glue_calls[5]);
  }
}

void odhip_interpose_set_reference(void *handle) {
  odhip_glue_set_reference(handle);
}

void odhip_interpose_enable_cache(int pic_w, int pic_h) {
  odhip_glue_enable_frame_cache(pic_w, pic_h);
}

void odhip_interpose_enable_bands(void) {
  odhip_glue_enable_band_cache();
}

void odhip_interpose_cache_stats(long *hits, long *misses) {
  odhip_glue_stats st;
  odhip_glue_get_stats(&st);
  *hits = st.fdct_hits;
  *misses = st.fdct_misses;
}

void odhip_interpose_band_stats(long *hits, long *misses) {
  odhip_glue_stats st;
  odhip_glue_get_stats(&st);
  *hits = st.band_hits;
  *misses = st.band_misses;
}

void odhip_interpose_dering_stats(void) {
  odhip_glue_stats st;
  odhip_glue_get_stats(&st);
  odhip_interposed_dering[0] = st.dering_launches;
  odhip_interposed_dering[1] = st.dering_served;
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

int odhip_glue_hook_postfilter_frame(od_coeff *c, int stride, int nhsb, int nvsb, int xdec, int ydec, int q,
 unsigned char *skip, int skip_stride, odhip_glue_postfilter_fn reference) {
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
  if (!g_decode_check || !g_dec) return 0;
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

