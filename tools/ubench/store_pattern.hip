// Write-pattern microbenchmark (development aid, DESIGN.md par. 4): what HBM sustains for the luma
// pyramid's store pattern - 16 frames x 5 raster planes of int32 1920x1088, each workgroup writing
// TW adjacent 64x64 tiles as row segments - against a linear fill of the same 668 MB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define W 1920
#define H 1088
#define F 16
// one workgroup of 256 threads per TW tiles; each store instruction of a wave covers 1 KB:
// SEG bytes contiguous per row (SEG = 256*TW), 1024/SEG rows
template <int TW>
__global__ __launch_bounds__(256) void k_tiles(int *out, int v) {
  const int xb = blockIdx.x*64*TW;
  const int y0 = blockIdx.y*64;
  const long plane = (long)W*H;
  const int f = blockIdx.z;
  constexpr int VPR = 16*TW;            // int4 vectors per row of the workgroup's tiles
  for (int lvl = 0; lvl < 5; lvl++) {
    int *p = out + ((long)lvl*F + f)*plane;
    for (int i = threadIdx.x; i < 64*VPR; i += 256) {
      const int y = i/VPR;
      const int x = (i % VPR)*4;
      *reinterpret_cast<int4 *>(p + (long)(y0 + y)*W + xb + x) = make_int4(v, i, y, x);
    }
  }
}
__global__ __launch_bounds__(256) void k_linear(int4 *out, long n, int v) {
  for (long i = blockIdx.x*256L + threadIdx.x; i < n; i += gridDim.x*256L) out[i] = make_int4(v, (int)i, 0, 0);
}
template <typename L> void run(const char *name, L launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; i++) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 5.0*F*W*H*4;
  printf("%-44s %7.1f us  %.2f TB/s\n", name, ms/20*1e3, bytes/(ms/20*1e-3)/1e12);
}
int main() {
  int *d; const long n = 5L*F*W*H;
  (void)hipMalloc(&d, n*4);
  run("linear fill, 16 B per lane", [&] { k_linear<<<256*8, 256>>>((int4 *)d, n/4, 3); });
  run("1 tile per workgroup (256 B row segments)", [&] { k_tiles<1><<<dim3(W/64, H/64, F), 256>>>(d, 3); });
  run("2 tiles per workgroup (512 B row segments)", [&] { k_tiles<2><<<dim3(W/128, H/64, F), 256>>>(d, 3); });
  run("6 tiles per workgroup (1536 B row segments)", [&] { k_tiles<6><<<dim3(W/384, H/64, F), 256>>>(d, 3); });
  run("30 tiles per workgroup (whole rows)", [&] { k_tiles<30><<<dim3(1, H/64, F), 256>>>(d, 3); });
  (void)hipFree(d);
  return 0;
}
