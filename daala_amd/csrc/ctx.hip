/* ctx.hip - odhip_create / odhip_destroy / odhip_make_current and the per-thread
   default contexts (see od_ctx.cuh).  Host code only. */
#include <stdlib.h>
#include <string.h>
#include "od_ctx.cuh"

namespace {

constexpr int kDefaultsPerDevice = 2;   /* odhip_pvq_ref_set_context(0 | 1) */

odhip_ctx *ctx_new(int device) {
  odhip_ctx *c = static_cast<odhip_ctx *>(calloc(1, sizeof(odhip_ctx)));
  if (c) c->device = device;
  return c;
}

void ctx_free(odhip_ctx *c) {
  if (!c) return;
  int prev = -1;
  const bool switched = hipGetDevice(&prev) == hipSuccess && prev != c->device
   && hipSetDevice(c->device) == hipSuccess;
  /* nothing of this context may still be running when its scratch goes away */
  (void)hipDeviceSynchronize();
  for (int i = 0; i < ODHIP_SLOT_COUNT; i++) {
    if (c->slot[i]) c->drop[i](c->slot[i]);
  }
  if (switched) (void)hipSetDevice(prev);
  free(c);
}

/* The calling thread's state: the explicitly selected context, and its default
   contexts (created on first use, destroyed when the thread ends). */
struct ThreadState {
  odhip_ctx *current = nullptr;
  int default_sel = 0;
  odhip_ctx *defaults[kOdhipMaxDevices][kDefaultsPerDevice] = {};
  ~ThreadState() {
    for (int d = 0; d < kOdhipMaxDevices; d++) {
      for (int i = 0; i < kDefaultsPerDevice; i++) ctx_free(defaults[d][i]);
    }
  }
};
thread_local ThreadState t_state;

}  // namespace

odhip_ctx *odhip_ctx_current(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kOdhipMaxDevices) {
    fprintf(stderr, "libdaalahip: no current HIP device (no CPU fallback)\n");
    return nullptr;
  }
  ThreadState &t = t_state;
  if (t.current) {
    if (t.current->device != dev) {
      fprintf(stderr, "libdaalahip: the current context belongs to device %d, the calling thread's "
       "HIP device is %d\n", t.current->device, dev);
      return nullptr;
    }
    return t.current;
  }
  odhip_ctx *&c = t.defaults[dev][t.default_sel];
  if (!c) c = ctx_new(dev);
  return c;
}

/* The two environment switches of a default build that the band stages and the pipe share
   (od_ctx.cuh).  ODHIP_PVQ_SERIAL is read once per process; ODHIP_PVQ_FORCE_SEQ at every call
   (the tests flip it inside one process). */
int odhip_env_serial(void) {
  static const int on = getenv("ODHIP_PVQ_SERIAL") != nullptr;
  return on;
}

int odhip_env_force_seq(void) {
  const char *e = getenv("ODHIP_PVQ_FORCE_SEQ");
  return e && e[0] == '1';
}

extern "C" odhip_ctx *odhip_create(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count
   || device >= kOdhipMaxDevices) {
    return nullptr;
  }
  return ctx_new(device);
}

extern "C" void odhip_destroy(odhip_ctx *ctx) {
  if (!ctx) return;
  if (t_state.current == ctx) t_state.current = nullptr;
  ctx_free(ctx);
}

extern "C" int odhip_make_current(odhip_ctx *ctx) {
  if (ctx) ODHIP_TRY(hipSetDevice(ctx->device));
  t_state.current = ctx;
  return ODHIP_SUCCESS;
}

extern "C" odhip_ctx *odhip_get_current(void) {
  return t_state.current;
}

extern "C" int odhip_ctx_set_serial(odhip_ctx *ctx, int serial) {
  if (!ctx) return ODHIP_EINVAL;
  ctx->serial = serial != 0;
  return ODHIP_SUCCESS;
}

/* Full-precision references (daala_info.full_precision_references, src/encode.c:212-213,
   src/state.c:256-258): see include/daala_hip.h. */
extern "C" int odhip_ctx_set_fpr(odhip_ctx *ctx, int on) {
  if (!ctx) return ODHIP_EINVAL;
  ctx->fpr = on != 0;
  return ODHIP_SUCCESS;
}

/* Test hooks of one context (NULL: the calling thread's current context). */
extern "C" int odhip_ctx_set_test_hooks(odhip_ctx *ctx, double theta_margin, int theta_perturb,
 double price_tol_scale) {
  if (!ctx) ctx = odhip_ctx_current();
  if (!ctx) return ODHIP_EINVAL;
  ctx->theta_margin = theta_margin;
  ctx->theta_perturb = theta_perturb != 0;
  ctx->price_tol_scale = price_tol_scale;
  return ODHIP_SUCCESS;
}

extern "C" int odhip_ctx_get_fpr(const odhip_ctx *ctx) {
  return ctx ? ctx->fpr : ODHIP_EINVAL;
}

extern "C" int odhip_ctx_device(const odhip_ctx *ctx) {
  return ctx ? ctx->device : ODHIP_EINVAL;
}

/* Compatibility with round 1's two fixed with-reference contexts: selects which of
   the calling thread's two DEFAULT contexts its calls use while no explicit context
   is current. */
extern "C" int odhip_pvq_ref_set_context(int ctx) {
  if (ctx < 0 || ctx >= kDefaultsPerDevice) return ODHIP_EINVAL;
  t_state.default_sel = ctx;
  return ODHIP_SUCCESS;
}
