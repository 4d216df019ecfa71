"""The generated split of the inverse lifting networks (tools/extract_lifting.py emit_device_inverse_split:
od_idctN_lift_part0/1 + od_idctN_lift_join0/1, N = 32, 64; used by k_idct64_split) compiled for the HOST and compared
with the generated full network and with the CPU oracle's table-driven interpreter on random vectors: the two parts
must depend on inputs of one parity only, the joins must reproduce every output bit for bit, and join H may read of the
other part's values only the entries kIdctNNeedIdxH lists (the kernel exchanges exactly those).  No GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <stdint.h>
#include <string.h>
#define __device__
#define __forceinline__ inline
struct W { int v; W() {} W(int x) : v(x) {} operator int() const { return v; } };
static inline W operator+(W a, W b) { return W((int)((unsigned)a.v + (unsigned)b.v)); }
static inline W operator-(W a, W b) { return W((int)((unsigned)a.v - (unsigned)b.v)); }
static inline W operator-(W a) { return W((int)(0u - (unsigned)a.v)); }
static inline W &operator+=(W &a, W b) { a = a + b; return a; }
static inline W &operator-=(W &a, W b) { a = a - b; return a; }
static inline W od_rs1(W a) { return W((int)(((unsigned)a.v >> 31) + (unsigned)a.v) >> 1); }
static inline W od_lift(W a, int c, int r, int s) { return W((int)((unsigned)a.v*(unsigned)c + (unsigned)r) >> s); }
#include "gen/od_lifting_gen.h"
#define SPLIT(N, M0, M1) \
extern "C" void split##N(int *out, const int *in, int poison) { \
  W e[N/2], o[N/2], m0[M0], m1[M1], f0[M0], f1[M1], lo[N/2], hi[N/2]; \
  for (int k = 0; k < N/2; k++) { e[k] = W(in[2*k]); o[k] = W(in[2*k + 1]); } \
  od_idct##N##_lift_part0(m0, e); \
  od_idct##N##_lift_part1(m1, o); \
  /* what each join may see of the OTHER part: only the listed entries (the rest poisoned) */ \
  for (int i = 0; i < M1; i++) f1[i] = W(poison); \
  for (int i = 0; i < M0; i++) f0[i] = W(poison); \
  for (int j = 0; j < kIdct##N##Need0; j++) f1[kIdct##N##NeedIdx0[j]] = m1[kIdct##N##NeedIdx0[j]]; \
  for (int j = 0; j < kIdct##N##Need1; j++) f0[kIdct##N##NeedIdx1[j]] = m0[kIdct##N##NeedIdx1[j]]; \
  od_idct##N##_lift_join0(lo, m0, f1); \
  od_idct##N##_lift_join1(hi, f0, m1); \
  for (int k = 0; k < N/2; k++) { out[k] = lo[k]; out[N/2 + k] = hi[k]; } \
} \
extern "C" void full##N(int *out, const int *in) { \
  W a[N], b[N]; \
  for (int k = 0; k < N; k++) a[k] = W(in[k]); \
  od_idct##N##_lift(b, a); \
  for (int k = 0; k < N; k++) out[k] = b[k]; \
}
SPLIT(32, kIdct32Mid0, kIdct32Mid1)
SPLIT(64, kIdct64Mid0, kIdct64Mid1)
extern "C" int need(int n, int half) { return n == 32 ? (half ? kIdct32Need1 : kIdct32Need0) : (half ? kIdct64Need1 : kIdct64Need0); }
"""


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("lift")
    src = d / "split.cc"
    src.write_text(SRC)
    so = d / "libsplit.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fwrapv", "-I", os.path.join(ROOT, "daala_amd", "csrc"),
                    "-o", str(so), str(src)], check=True, capture_output=True)
    return ctypes.CDLL(str(so))


@pytest.mark.parametrize("n", [32, 64])
def test_split_inverse_network_equals_full_network_and_oracle(lib, n):
    from _libs import P, oracle
    rng = np.random.RandomState(5 + n)
    ln = {32: 3, 64: 4}[n]
    split = getattr(lib, "split%d" % n)
    full = getattr(lib, "full%d" % n)
    o = oracle()
    assert lib.need(64, 0) == 32 and lib.need(64, 1) == 32        # the exchange area of k_idct64_split
    for trial in range(300):
        amp = [4080, 1 << 19, 1 << 30][trial % 3]
        x = rng.randint(-amp, amp + 1, size=n).astype(np.int32)
        if trial % 7 == 0:
            x[n // 2:] = 0                                           # the pruned inputs of the 64x64 luma leaves
        a = np.zeros(n, np.int32)
        b = np.zeros(n, np.int32)
        c = np.zeros(n, np.int32)
        full(P(a), P(x))
        # two different poisons: an output that read an unlisted value of the other part would change
        split(P(b), P(x), 0x12345678)
        split(P(c), P(x), -77)
        assert np.array_equal(a, b) and np.array_equal(a, c), (n, trial)
        want = np.zeros(n, np.int32)
        o.odo_idct_1d(ln, P(want), 1, P(x))
        assert np.array_equal(a, want), (n, trial)
