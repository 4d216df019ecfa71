// Development aid (DESIGN.md par. 4): do bursts of global stores overlap with VALU work of OTHER waves on
// the CU?  Workgroups shaped like the luma pyramid's (256 threads, 53 KB of LDS -> 3 per CU, 4080 of
// them), five phases each = `work` dependent-free VALU iterations followed by a burst of 8 x 16-byte
// stores per lane (two 64x64 int32 tiles as row segments).  Reported: stores only, arithmetic only, both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define W 1920
#define H 1088
#define F 16
template <bool STORE, bool SPREAD>
__global__ __launch_bounds__(256) void k(int *out, int work, int v) {
  extern __shared__ int lds[];
  const int xb = blockIdx.x*128;
  const int y0 = blockIdx.y*64;
  const long plane = (long)W*H;
  const int f = blockIdx.z;
  int a0 = v + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  for (int lvl = 0; lvl < 5; lvl++) {
    int *p = out + ((long)lvl*F + f)*plane;
    if (!SPREAD) {
      for (int it = 0; it < work; it++) {
        asm volatile("v_add_u32 %0, %0, %1\n\tv_mad_i32_i24 %2, %2, %1, %1\n\tv_ashrrev_i32 %3, 1, %3\n\tv_sub_u32 %4, %4, %1"
         : "+v"(a0), "+v"(v), "+v"(a1), "+v"(a2), "+v"(a3));
      }
      if (threadIdx.x == 0) lds[lvl] = a0;
      __syncthreads();
    }
    for (int k8 = 0; k8 < 8; k8++) {
      if (SPREAD) {
        for (int it = 0; it < work/8; it++) {
          asm volatile("v_add_u32 %0, %0, %1\n\tv_mad_i32_i24 %2, %2, %1, %1\n\tv_ashrrev_i32 %3, 1, %3\n\tv_sub_u32 %4, %4, %1"
           : "+v"(a0), "+v"(v), "+v"(a1), "+v"(a2), "+v"(a3));
        }
      }
      const int i = threadIdx.x + k8*256;
      const int y = i/32;
      const int x = (i % 32)*4;
      if (STORE) *reinterpret_cast<int4 *>(p + (long)(y0 + y)*W + xb + x) = make_int4(a0, a1, a2, a3);
    }
  }
  if (a0 == 0x12345678) out[0] = a1 ^ a2 ^ a3 ^ lds[0];
}
template <typename L> float run(L launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; i++) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms/20*1e3f;
}
int main() {
  int *d; const long n = 5L*F*W*H;
  (void)hipMalloc(&d, n*4);
  const dim3 g(W/128, H/64, F);
  const size_t lds = 53312;
  for (int work : {0, 100, 200, 300, 400, 600}) {
    float ts = run([&] { k<true, false><<<g, 256, lds>>>(d, 0, 3); });
    float tc = run([&] { k<false, false><<<g, 256, lds>>>(d, work, 3); });
    float tb = run([&] { k<true, false><<<g, 256, lds>>>(d, work, 3); });
    float tsp = run([&] { k<true, true><<<g, 256, lds>>>(d, work, 3); });
    printf("work %4d: stores only %6.1f us, arithmetic only %6.1f us, bursts after each phase %6.1f us, stores spread through the phase %6.1f us\n",
     work, ts, tc, tb, tsp);
  }
  (void)hipFree(d);
  return 0;
}
