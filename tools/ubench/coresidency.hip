// Development aid (round 6, DESIGN.md "two chains on one GPU"): when do workgroups of two kernels launched on
// two streams share the compute units?  Kernel A is shaped like the K-pulse searches (many 64-thread
// workgroups - or WA threads -, LA bytes of LDS each, a chain of fp64 operations, no memory traffic),
// kernel B like the filter + DCT kernels (256-thread workgroups, LB bytes of LDS, streams 16-byte loads and
// stores).  Reported per configuration: A alone, B alone, both launched together (wall time from the first
// launch to the last completion, and when each of the two finished), so that
//   together ~ max(A, B)  -> the workgroups shared the CUs,
//   together ~ A + B      -> one kernel waited for the other's workgroups to drain.
// usage: coresidency [prio]      prio = 1: B's stream gets the highest priority, 2: A's
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// NV live doubles per lane: 4 -> 16 VGPRs (occupancy limited by wave slots or LDS), 28 -> 86 VGPRs (5 waves per SIMD,
// like the search kernels), 36 -> 110 (4 waves per SIMD)
template <int NV>
__global__ void k_valu(double *out, int iters, double seed) {
  extern __shared__ double lds_a[];
  double r[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) r[j] = seed + threadIdx.x + j;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < NV; j++) r[j] = r[j]*1.0000001 + r[(j + 1) % NV];
  }
  double sum = 0;
#pragma unroll
  for (int j = 0; j < NV; j++) sum += r[j];
  if (sum == 123.456) {
    lds_a[threadIdx.x] = sum;
    out[blockIdx.x] = sum + lds_a[0];
  }
}
template <int NV>
static void launch_valu(int nwg, int wa, int la, hipStream_t s, double *dout, int iters) {
  k_valu<NV><<<nwg, wa, la, s>>>(dout, iters*4/NV, 1.5);
}
static void launch_a(int nv, int nwg, int wa, int la, hipStream_t s, double *dout, int iters) {
  if (nv == 36) launch_valu<36>(nwg, wa, la, s, dout, iters);
  else if (nv == 28) launch_valu<28>(nwg, wa, la, s, dout, iters);
  else launch_valu<4>(nwg, wa, la, s, dout, iters);
}

__global__ __launch_bounds__(256) void k_mem(const int4 *in, int4 *out, long n_per_wg, int rounds) {
  extern __shared__ int lds_b[];
  const long base = (long)blockIdx.x*n_per_wg;
  int4 acc = make_int4(0, 0, 0, 0);
  for (int r = 0; r < rounds; r++) {
    for (long i = threadIdx.x; i < n_per_wg; i += 256) {
      int4 v = in[base + i];
      acc.x ^= v.x;
      v.y += r;
      out[base + i] = v;
    }
  }
  if (acc.x == 0x7fffffff) {
    lds_b[threadIdx.x] = acc.x;
    out[0].x = lds_b[0];
  }
}

int main(int argc, char **argv) {
  const int prio = argc > 1 ? atoi(argv[1]) : 0;
  int lo = 0, hi = 0;
  CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, prio == 2 ? hi : prio == 1 ? lo : 0));
  CHECK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, prio == 1 ? hi : prio == 2 ? lo : 0));
  const long nwg_b = 4080;
  const long n_per_wg = 8192;                 // int4 per workgroup: 128 KB read + 128 KB written
  int4 *in, *out;
  double *dout;
  CHECK(hipMalloc(&in, nwg_b*n_per_wg*16));
  CHECK(hipMalloc(&out, nwg_b*n_per_wg*16));
  CHECK(hipMalloc(&dout, 1 << 20));
  CHECK(hipMemset(in, 1, nwg_b*n_per_wg*16));
  CHECK(hipFuncSetAttribute((const void *)k_mem, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  CHECK(hipFuncSetAttribute((const void *)k_valu<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  CHECK(hipFuncSetAttribute((const void *)k_valu<36>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  CHECK(hipFuncSetAttribute((const void *)k_valu<28>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  hipEvent_t a0, a1, b0, b1;
  CHECK(hipEventCreate(&a0)); CHECK(hipEventCreate(&a1)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));
  printf("priority range %d..%d, mode %d\n", lo, hi, prio);
  printf("%3s %6s %6s %7s %7s | %8s %8s | %8s %8s %8s | %s\n", "NV", "WA", "LA", "LB", "nwgA", "A_alone", "B_alone", "together", "A_done", "B_done", "verdict");
  struct Cfg { int nv; int wa; int la; int lb; };
  const Cfg cfgs[] = {
    // A fills every wave slot
    {4, 64, 0, 0}, {4, 64, 0, 53760},
    // A limited by LDS (wave slots and registers left over), B without LDS / with more than is left
    {4, 64, 8192, 0}, {4, 64, 11264, 0}, {4, 64, 11264, 21504}, {4, 64, 11264, 53760}, {4, 64, 20480, 0},
    // A limited by registers (5 waves per SIMD), LDS left over: 160 KB, 160 - 20*4 = 80 KB, 160 - 20*6 = 40 KB
    {28, 64, 0, 0}, {28, 64, 0, 53760}, {28, 64, 4096, 53760}, {28, 64, 6144, 53760}, {28, 64, 6144, 21504},
    // 4 waves per SIMD by registers (110 VGPRs)
    {36, 64, 0, 53760}, {36, 64, 2048, 53760},
    // A in slots the size of B's workgroups: 256 threads, B's LDS
    {4, 256, 53760, 53760}, {28, 256, 53760, 53760}, {28, 256, 53760, 21504}, {28, 256, 53760, 0}, {28, 256, 40960, 53760},
    // one big A workgroup per CU that leaves room
    {28, 1024, 106496, 53760}, {28, 512, 106496, 53760},
  };
  for (const Cfg &c : cfgs) {
    // A: ~600 us alone.  ~ iters*16 fp64 ops per lane; waves = nwgA*WA/64
    const int waves_a = 64*1024;
    const int nwg_a = waves_a*64/c.wa;
    const int iters = 1500;
    auto la = [&]() { launch_a(c.nv, nwg_a, c.wa, c.la, sa, dout, iters); };
    auto lb = [&]() { k_mem<<<nwg_b, 256, c.lb, sb>>>(in, out, n_per_wg, 1); };
    float ta = 0, tb = 0, tab = 0, ta_done = 0, tb_done = 0;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(a0, sa)); la(); CHECK(hipEventRecord(a1, sa)); CHECK(hipEventSynchronize(a1));
      CHECK(hipEventElapsedTime(&ta, a0, a1));
      CHECK(hipEventRecord(b0, sb)); lb(); CHECK(hipEventRecord(b1, sb)); CHECK(hipEventSynchronize(b1));
      CHECK(hipEventElapsedTime(&tb, b0, b1));
      CHECK(hipDeviceSynchronize());
      // together: A first, B right behind it
      CHECK(hipEventRecord(a0, sa)); la(); CHECK(hipEventRecord(a1, sa));
      lb(); CHECK(hipEventRecord(b1, sb));
      CHECK(hipEventSynchronize(a1)); CHECK(hipEventSynchronize(b1));
      CHECK(hipEventElapsedTime(&ta_done, a0, a1));
      CHECK(hipEventElapsedTime(&tb_done, a0, b1));
      tab = ta_done > tb_done ? ta_done : tb_done;
    }
    const float mx = ta > tb ? ta : tb;
    const float share = (ta + tb - tab)/(ta + tb - mx);     // 1: perfect overlap, 0: serial
    printf("%3d %6d %6d %7d %7d | %8.3f %8.3f | %8.3f %8.3f %8.3f | overlap %.2f\n", c.nv, c.wa, c.la, c.lb, nwg_a, ta, tb, tab, ta_done, tb_done, share);
  }
  // B first, then A
  printf("B launched first:\n");
  for (const Cfg &c : cfgs) {
    const int waves_a = 64*1024;
    const int nwg_a = waves_a*64/c.wa;
    const int iters = 1500;
    auto la = [&]() { launch_a(c.nv, nwg_a, c.wa, c.la, sa, dout, iters); };
    auto lb = [&]() { k_mem<<<nwg_b, 256, c.lb, sb>>>(in, out, n_per_wg, 1); };
    float tab = 0, ta_done = 0, tb_done = 0;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(b0, sb)); lb(); CHECK(hipEventRecord(b1, sb));
      la(); CHECK(hipEventRecord(a1, sa));
      CHECK(hipEventSynchronize(a1)); CHECK(hipEventSynchronize(b1));
      CHECK(hipEventElapsedTime(&tb_done, b0, b1));
      CHECK(hipEventElapsedTime(&ta_done, b0, a1));
      tab = ta_done > tb_done ? ta_done : tb_done;
    }
    printf("%3d %6d %6d %7d | together %8.3f A_done %8.3f B_done %8.3f\n", c.nv, c.wa, c.la, c.lb, tab, ta_done, tb_done);
  }
  return 0;
}
