/* shim/daala_hip_glue.h - the reference-side binding of libdaalahip.

   daala_hip_glue.c defines the reference's OWN symbol names for the surfaces of the
   block-transform path (src/filter.h:80-87, src/pvq_encoder.c:93, :333, src/pvq_encoder.h:46,
   src/dering.h, src/state.c:346) and forwards them to libdaalahip (include/daala_hip.h).  Linked
   in front of libdaala - LD_PRELOAD for an unmodified binary such as examples/encoder_example, or
   listed before libdaala on the link line, or compiled into it with the C definitions renamed -
   every call inside the unmodified encoder / decoder binds to these definitions.  Nothing is
   switched by environment variables: the host says what it wants bound with
   odhip_glue_configure() (or the odhip_glue_enable_*() calls) before it creates its first
   encoder; the default binds the per-call surfaces only.  README.md in this directory shows
   the three ways of linking it. */
#ifndef DAALA_HIP_GLUE_H
#define DAALA_HIP_GLUE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct odhip_glue_config {
  int device;            /* HIP device ordinal (odhip_init) */
  /* Per-call surfaces: 1 = the call runs on the GPU through the *_hip entry point with the
     reference's own signature, 0 = it is forwarded to the reference's own C definition (the
     next definition of the symbol).  One round trip per call: these are correctness surfaces
     and the plumbing of BASELINE configs[0]; the batched bindings below are the fast ones. */
  int bind_filters;      /* od_prefilter_split, od_postfilter_split, od_apply_{pre,post}filter_frame_sbs */
  int bind_search;       /* pvq_search_rdo_double */
  int bind_dering;       /* od_dering (per call) */
  int bind_dct_vtbl;     /* od_state_opt_vtbl_init: fdct_2d / idct_2d <- od_bin_{f,i}dctNxN_hip */
  /* Batched bindings. */
  int frame_cache;       /* one batched pyramid per plane behind every fdct_2d call of a frame */
  int band_cache;        /* the PVQ band stage of keyframe luma behind pvq_theta, host pricing through
                            odhip_pvq_rate_batch16 at speed 0 */
  int dering_cache;      /* the deringing level search served from batched passes */
  int pic_w, pic_h;      /* picture size (frame_cache) */
  /* Self-checks against the reference's own C definition of the same call (abort on the first
     difference): for bring-up in a new host. */
  int check_rates;       /* every batched od_pvq_rate against od_pvq_rate */
  int check_dering;      /* every served od_dering superblock against od_dering */
  /* Several encoder processes sharing one GPU (bench.py --procs-per-gpu): 1 = the batched GPU
     pass of a frame (pyramids + band stage, ~5-10 ms alone) is taken under an advisory lock on
     /tmp/odhip_glue_gpu<device>.lock, one process at a time.  Measured SLOWER than letting the
     passes overlap (profiles/r4_encode_mode_300frames.json); kept as an option, off by default. */
  int gpu_pass_lock;
  /* With dering_cache and frame_cache: the level search's od_compute_dist calls (src/encode.c:
     2776-2801) from the same batched passes (odhip_dering_cache_set_source / _dist).  Needs the ten
     lines of glue in front of the file-static od_compute_dist that call odhip_glue_compute_dist
     (oracle/Makefile DISTGLUE; README.md); without them nothing is served and nothing changes. */
  int dist_cache;
  int check_dist;        /* every served distortion against the C function (abort on a difference) */
  /* (round 6; appended so that hosts built against the earlier layout keep their field offsets) */
  int bind_synthesis;    /* per call: od_pvq_synthesis_partial (encoder AND decoder: the data-parallel half of
                            od_pvq_decode's pvq_decode_partition) - off in the default configuration: one round
                            trip per coded band */
} odhip_glue_config;

/* Called by the reference's od_compute_dist (see dist_cache): 1 and *out when the call is one the
   dering cache can vouch for, else 0 (run the C function). */
int odhip_glue_compute_dist(const int32_t *x, const int32_t *y, int n, int use_masking, int flat_qm,
 int coded_quantizer, double *out);
int odhip_glue_check_dist(void);

/* Threads.  The configuration is per process; the state the batched bindings keep between calls
   (frame cache, band cache, dering cache, the encoder and the block a pvq_theta call belongs to) is
   per HOST THREAD, created on a thread's first use: any number of encoder contexts may run in
   different threads of one process and share its HIP context (16 encoder threads in one process:
   7.9 ms of GPU passes per frame, 16 processes: 50 ms - profiles/r4_encode_mode_300frames.json).
   One encoder per thread at a time; the hot counters are counted per thread and folded into the
   process totals at frame boundaries (odhip_glue_flush_stats), the cache hit / miss / dering
   figures with them (as deltas of each thread's caches), so odhip_glue_get_stats reports the whole
   process from any thread.  A thread's caches are created on its first plane load - not by the
   configuring thread - and destroyed when the thread exits (a pthread key destructor). */

/* All per-call surfaces bound, no batched binding, no checks, device 0. */
void odhip_glue_default_config(odhip_glue_config *cfg);
/* Takes effect for the calls that follow.  Returns 0, or a negative ODHIP_* code when the device
   cannot be initialised. */
int odhip_glue_configure(const odhip_glue_config *cfg);
void odhip_glue_get_config(odhip_glue_config *cfg);

/* When the reference library was dlopen()ed (a Python / JNI host) instead of linked: its handle,
   so that the reference's own definitions and its state accessors (below) can be found;
   otherwise they are looked up with dlsym(RTLD_NEXT / RTLD_DEFAULT). */
void odhip_glue_set_reference(void *dl_handle);

/* Convenience forms of odhip_glue_configure for the batched bindings. */
void odhip_glue_enable_frame_cache(int pic_w, int pic_h);
void odhip_glue_enable_band_cache(void);
void odhip_glue_enable_dering_cache(void);

typedef struct odhip_glue_stats {
  long calls[6];          /* od_prefilter_split, od_postfilter_split, od_apply_prefilter_frame_sbs,
                             od_apply_postfilter_frame_sbs, pvq_search_rdo_double, od_dering */
  long theta[4];          /* pvq_theta calls served from the batch / left to the reference because
                             the band has a reference vector / left for another reason / K-pulse
                             searches the batch saved */
  long fdct_hits, fdct_misses;       /* frame cache */
  long band_hits, band_misses;       /* band cache */
  long dering_launches, dering_served;
  double batch_ms;        /* wall time inside the batched GPU passes, PCIe both ways included */
  double dering_ms;       /* ... inside the dering cache calls (its launches, copies, served superblocks) */
  double theta_ms;        /* ... inside the pvq_theta calls served from the band cache (with their pricing) */
  long dist_served, dist_left;       /* od_compute_dist calls served from the dering cache / run in C */
} odhip_glue_stats;
void odhip_glue_get_stats(odhip_glue_stats *st);
/* Folds the calling thread's counters into the process totals (done at every frame's GPU pass, at
   daala_encode_free and by odhip_glue_get_stats; a host that reads the totals from another thread
   while encoders run sees them one frame late). */
void odhip_glue_flush_stats(void);

/* What the glue needs from the reference side beyond its public API: three accessors compiled
   with the reference's headers (they know the layout of od_state / daala_enc_ctx).  In this
   repository they live in oracle/ref_encoder_driver.c (built into oracle/_ref/libdaalaref.so); a
   maintainer adds them to src/state.c / src/encode.c.  README.md lists them. */
/*   void ref_state_set_dct_vtbl(od_state *state, void **fdct_2d, void **idct_2d);
     int  ref_enc_band_setup(const daala_enc_ctx *enc, int *quantizer, int *use_masking, double *lambda,
                             unsigned char *pvq_qm_q4, int16_t *qm, int16_t *qm_inv);              */

#ifdef __cplusplus
}
#endif
#endif
