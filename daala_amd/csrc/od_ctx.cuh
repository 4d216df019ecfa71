/* od_ctx.cuh - odhip_ctx: the owner of every piece of mutable library state.

   SURVEY.md 8(b) "proposed batched ABI": all device memory, side streams, events
   and job tables the batched entry points need between calls belong to a
   context; one call sequence may be in flight per context (on one caller
   stream), any number of contexts may be in flight at once - on different
   streams, host threads or devices.  Shaped after the reference's per-od_state
   backend installation (src/x86/x86state.c:39-97: nothing process-global).

   The batched odhip_* functions use the CURRENT context of the calling thread:
   the one given to odhip_make_current, else a default context the thread gets
   lazily for the HIP device that is current when it first calls in.  Modules keep
   their state in a slot of the context (odhip_ctx_state<T>), created on first use
   and destroyed with the context. */
#pragma once
#include <mutex>
#include "od_common.cuh"

enum {
  ODHIP_SLOT_LAPPED = 0,   /* lapped_kernels.hip: edge strips of the inverse stage */
  ODHIP_SLOT_BANDS,        /* pvq_bands.hip: no-reference band stage */
  ODHIP_SLOT_REFBANDS,     /* pvq_refbands.hip: with-reference band stage */
  ODHIP_SLOT_COUNT
};

struct odhip_ctx {
  int device;
  int serial;              /* odhip_ctx_set_serial: no internal side streams */
  int fpr;                 /* odhip_ctx_set_fpr: picture planes hold int16 samples at 12 bits */
  /* odhip_ctx_set_test_hooks (tests only; per context, nothing process-wide) */
  double theta_margin;     /* <= 0: the default margin of the device acos */
  int theta_perturb;       /* a deliberately wrong device theta for listed bands */
  double price_tol_scale;  /* <= 0: 1; multiplies the priced choice's decision margin */
  void *slot[ODHIP_SLOT_COUNT];
  void (*drop[ODHIP_SLOT_COUNT])(void *);
};

/* The calling thread's current context, bound to the HIP device that is current
   now; nullptr (after a message on stderr) when an explicitly selected context
   belongs to another device or no context can be made. */
odhip_ctx *odhip_ctx_current(void);

template <class T>
static inline T *odhip_ctx_state(odhip_ctx *c, int slot) {
  if (!c->slot[slot]) {
    c->slot[slot] = new T();
    c->drop[slot] = [](void *p) { delete static_cast<T *>(p); };
  }
  return static_cast<T *>(c->slot[slot]);
}

/* Environment switches.  A DEFAULT build of the library reads exactly three: ODHIP_PVQ_SERIAL (no
   side streams inside the band stages and the pipe: exclusive kernel times), ODHIP_PVQ_FORCE_SEQ
   (every greedy pulse of the pair / row searches by the literal left-to-right scan: the cross-check
   of their exact-argmax shortcuts) - both below, ctx.hip - and ODHIP_CACHE_CHECK (frame_cache.hip).
   Everything else - superseded kernel generations kept as A/B baselines, ablations, tuning knobs -
   exists only in a build with -DODHIP_EXPERIMENTS (daala_amd/build.py: lib/libdaalahip_exp.so,
   selected with ODHIP_LIB; tools/ and the cross-check tests use it): there ODHIP_EXP_ENV(name) is
   getenv(name), in a default build it is a null pointer and the code behind it is not compiled. */
int odhip_env_serial(void);
int odhip_env_force_seq(void);
#ifdef ODHIP_EXPERIMENTS
#define ODHIP_EXP_ENV(name) getenv(name)
#else
#define ODHIP_EXP_ENV(name) (static_cast<const char *>(nullptr))
#endif

#define ODHIP_CTX_OR_RETURN(var) \
  odhip_ctx *var = odhip_ctx_current(); \
  if (!var) return ODHIP_EINVAL

/* Per-device one-time initialisation of a module's constant tables
   (hipMemcpyToSymbol targets are per device). */
constexpr int kOdhipMaxDevices = 64;
struct odhip_device_once {
  std::mutex lock;
  bool done[kOdhipMaxDevices] = {};
};

template <typename F>
static inline int odhip_once_per_device(odhip_device_once &once, F upload) {
  int dev = 0;
  ODHIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kOdhipMaxDevices) return ODHIP_EINVAL;
  std::lock_guard<std::mutex> guard(once.lock);
  if (once.done[dev]) return ODHIP_SUCCESS;
  const int rc = upload();
  if (rc == ODHIP_SUCCESS) once.done[dev] = true;
  return rc;
}
