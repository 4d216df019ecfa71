#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>
/* Interposer for an Amdahl bound (test infrastructure, never shipped): wall time the reference
   encoder spends inside pvq_search_rdo_double, split by whether the enclosing pvq_theta call has a
   reference vector (r0 != 0 anywhere) - the calls the batched band stage cannot serve. */
typedef double (*sfn)(const int16_t *, int, int, int32_t *, double, double, int);
typedef int (*tfn)(int32_t *, const int32_t *, const int32_t *, int, int, int32_t *, int *, int *, int *, int16_t, double *, int, int, int, const void *, const int16_t *, const int16_t *, double, int);
static void *g_ref;
void cnt_set_ref(void *h) { g_ref = h; }
double t_search[2]; long n_search[2]; double t_theta[2]; long n_theta[2];
static __thread int cur;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9*t.tv_nsec; }
double pvq_search_rdo_double(const int16_t *x, int n, int k, int32_t *y, double g2, double lam, int prev_k) {
  static sfn next;
  if (!next) next = (sfn)dlsym(g_ref, "pvq_search_rdo_double");
  double a = now();
  double r = next(x, n, k, y, g2, lam, prev_k);
  t_search[cur] += now() - a; n_search[cur]++;
  return r;
}
int pvq_theta(int32_t *out, const int32_t *x0, const int32_t *r0, int n, int q0, int32_t *y, int *itheta, int *max_theta, int *vk, int16_t beta, double *skip_diff, int nodesync, int is_keyframe, int pli, const void *adapt, const int16_t *qm, const int16_t *qm_inv, double lam, int speed) {
  static tfn next;
  if (!next) next = (tfn)dlsym(g_ref, "pvq_theta");
  int i, has = 0;
  for (i = 0; i < n; i++) if (r0[i]) { has = 1; break; }
  cur = has || pli != 0;
  double a = now();
  int r = next(out, x0, r0, n, q0, y, itheta, max_theta, vk, beta, skip_diff, nodesync, is_keyframe, pli, adapt, qm, qm_inv, lam, speed);
  t_theta[cur] += now() - a; n_theta[cur]++;
  return r;
}
