// Measurement aid (round 6): what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access
// patterns of this library's kernels.  The guide calibrates one case (a wide coalesced streaming read: FETCH_SIZE = half
// the bytes) and says "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own
// access pattern".  Every kernel here touches a KNOWN set of bytes of a 2 GiB buffer exactly once (far beyond the 256 MiB
// Infinity Cache, so nothing is absorbed on-die); the kernel names carry the pattern, the driver script prints per kernel the
// useful bytes, the 32-byte sectors / 64-byte / 128-byte lines touched, and the counters.
//   reads : r_coalesced16   16 B per lane, lanes contiguous                      (the streaming kernels)
//           r_coalesced4     4 B per lane, lanes contiguous
//           r_16_stride64   16 B per lane, one per 64-byte line                  (sorted-order gathers of 16-byte pieces)
//           r_16_stride128  16 B per lane, one per 128-byte line
//           r_4_stride32     4 B per lane, one per 32-byte sector                (the inverse's DC gathers at the 4x4 level)
//           r_4_stride64     4 B per lane, one per 64-byte line
//           r_64_stride448  64 B per lane (4 x 16 B), records 448 B apart        (the 64-byte band records, 7 bands per block)
//   writes: w_coalesced16, w_4_stride64 (k_edge_rows: 4 bytes of every 64-byte line), w_64_stride448, w_32_stride64
// usage: fetch_calib            (run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// element i of the pattern: BYTES bytes at byte offset i*STRIDE; every lane handles one element per step, consecutive
// lanes consecutive elements; the result is folded so the loads cannot be dropped
template <int BYTES, int STRIDE>
__device__ __forceinline__ void read_pattern(const char *buf, long n, int *sink) {
  int acc = 0;
  for (long i = (long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x*blockDim.x) {
    const char *p = buf + i*STRIDE;
    if (BYTES == 4) acc ^= *reinterpret_cast<const int *>(p);
    else {
#pragma unroll
      for (int q = 0; q < BYTES/16; q++) {
        const int4 v = reinterpret_cast<const int4 *>(p)[q];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  if (acc == 0x12345678) sink[0] = acc;
}
template <int BYTES, int STRIDE>
__device__ __forceinline__ void write_pattern(char *buf, long n) {
  for (long i = (long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x*blockDim.x) {
    char *p = buf + i*STRIDE;
    if (BYTES == 4) *reinterpret_cast<int *>(p) = (int)i;
    else {
#pragma unroll
      for (int q = 0; q < BYTES/16; q++) reinterpret_cast<int4 *>(p)[q] = make_int4((int)i, q, 0, 0);
    }
  }
}
#define RK(name, B, S) __global__ __launch_bounds__(256) void name(const char *buf, long n, int *sink) { read_pattern<B, S>(buf, n, sink); }
#define WK(name, B, S) __global__ __launch_bounds__(256) void name(char *buf, long n) { write_pattern<B, S>(buf, n); }
RK(r_coalesced16, 16, 16)
RK(r_coalesced4, 4, 4)
RK(r_16_stride64, 16, 64)
RK(r_16_stride128, 16, 128)
RK(r_4_stride32, 4, 32)
RK(r_4_stride64, 4, 64)
RK(r_64_stride448, 64, 448)
WK(w_coalesced16, 16, 16)
WK(w_4_stride64, 4, 64)
WK(w_64_stride448, 64, 448)
WK(w_32_stride64, 32, 64)

int main() {
  const long total = 2L << 30;
  char *buf;
  int *sink;
  CHECK(hipMalloc((void **)&buf, total));
  CHECK(hipMalloc((void **)&sink, 64));
  CHECK(hipMemset(buf, 1, total));
  CHECK(hipDeviceSynchronize());
  const int grid = 256*16;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
#define RUN(name, B, S, ...) do { \
    const long n = total/(S); \
    CHECK(hipEventRecord(e0)); \
    name<<<grid, 256>>>(__VA_ARGS__); \
    CHECK(hipEventRecord(e1)); \
    CHECK(hipEventSynchronize(e1)); \
    float ms; \
    CHECK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-16s elements %10ld bytes %3d stride %3d  %.3f ms  %.0f GB/s useful\n", #name, n, (B), (S), ms, n*(double)(B)/ms/1e6); \
  } while (0)
  for (int rep = 0; rep < 2; rep++) {
    RUN(r_coalesced16, 16, 16, buf, n, sink);
    RUN(r_coalesced4, 4, 4, buf, n, sink);
    RUN(r_16_stride64, 16, 64, buf, n, sink);
    RUN(r_16_stride128, 16, 128, buf, n, sink);
    RUN(r_4_stride32, 4, 32, buf, n, sink);
    RUN(r_4_stride64, 4, 64, buf, n, sink);
    RUN(r_64_stride448, 64, 448, buf, n, sink);
    RUN(w_coalesced16, 16, 16, buf, n);
    RUN(w_4_stride64, 4, 64, buf, n);
    RUN(w_64_stride448, 64, 448, buf, n);
    RUN(w_32_stride64, 32, 64, buf, n);
  }
  return 0;
}
