#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
typedef double (*fn)(const int16_t *, int, int, int32_t *, double, double, int);
static void *g_ref;
void cnt_set_ref(void *h) { g_ref = h; }
/* [n class 0..5: 8,15,32(31),128(127),other][0 searches,1 greedy pulses,2 tail pulses,3 chained searches, 4 k sum, 5 zero-pulse searches] */
long cnt[6][8];
long khist[6][64];
static int cls(int n) { return n == 31 ? 0 : n == 127 ? 1 : n == 32 ? 2 : n == 128 ? 3 : n == 15 ? 4 : 5; }
double pvq_search_rdo_double(const int16_t *x, int n, int k, int32_t *y, double g2, double lam, int prev_k) {
  static fn next;
  if (!next) next = (fn)dlsym(g_ref, "pvq_search_rdo_double");
  if (!next) abort();
  int c = cls(n);
  int i = 0;
  if (prev_k > 0 && prev_k <= k) { i = prev_k; cnt[c][3]++; }
  else if (k > 2) {
    double l1 = 0; int j;
    for (j = 0; j < n; j++) l1 += fabs((float)x[j]);
    double inv = 1./(l1 > 1e-100 ? l1 : 1e-100);
    for (j = 0; j < n; j++) { int v = (int)floor(k*fabs((float)x[j])*inv); i += v > 0 ? v : 0; }
  }
  int ng = k - (1 + k/4);
  int greedy = ng > i ? ng - i : 0;
  int tail = k - (i > ng ? i : ng);
  if (tail < 0) tail = 0;
  cnt[c][0]++; cnt[c][1] += greedy; cnt[c][2] += tail; cnt[c][4] += k; if (greedy + tail == 0) cnt[c][5]++;
  khist[c][k < 63 ? k : 63]++;
  return next(x, n, k, y, g2, lam, prev_k);
}
