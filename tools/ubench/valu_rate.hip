// Integer VALU issue-rate microbenchmark for gfx950 (development aid): inline
// asm so the compiler cannot fold the chains.  8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
#define CHAIN8(ASM) \
  for (int it = 0; it < ITERS; it++) { \
    asm volatile(ASM : "+v"(a0) : "v"(c)); asm volatile(ASM : "+v"(a1) : "v"(c)); \
    asm volatile(ASM : "+v"(a2) : "v"(c)); asm volatile(ASM : "+v"(a3) : "v"(c)); \
    asm volatile(ASM : "+v"(a4) : "v"(c)); asm volatile(ASM : "+v"(a5) : "v"(c)); \
    asm volatile(ASM : "+v"(a6) : "v"(c)); asm volatile(ASM : "+v"(a7) : "v"(c)); \
  }
template <int OP>
__global__ void k(int *out, int seed) {
  int a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int c = seed*7 + 23013;
  if (OP == 0) { CHAIN8("v_add_u32 %0, %0, %1") }
  if (OP == 1) { CHAIN8("v_mad_i32_i24 %0, %0, %1, %1") }
  if (OP == 2) { CHAIN8("v_mul_lo_u32 %0, %0, %1") }
  if (OP == 3) { CHAIN8("v_ashrrev_i32 %0, 1, %0") }
  if (OP == 4) { CHAIN8("v_mul_i32_i24 %0, %0, %1") }
  if (OP == 5) { CHAIN8("v_mul_hi_i32 %0, %0, %1") }
  if (OP == 6) { CHAIN8("v_add3_u32 %0, %0, %1, %1") }
  if (OP == 7) { CHAIN8("v_mad_u32_u24 %0, %0, %1, %1") }
  if (OP == 8) { CHAIN8("v_fma_f32 %0, %0, %1, %1") }
  if (OP == 9) { CHAIN8("v_lshl_add_u32 %0, %0, 1, %1") }
  if (OP == 10) { CHAIN8("v_mad_i32_i16 %0, %0, %1, %1") }
  if (OP == 11) { CHAIN8("v_pk_mul_lo_u16 %0, %0, %1") }
  if (OP == 12) { CHAIN8("v_mul_u32_u24 %0, %0, %1") }
  out[blockIdx.x*blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> void run(const char *name) {
  int *d; (void)hipMalloc(&d, 256*16*256*4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<OP><<<256*16, 256>>>(d, 3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<OP><<<256*16, 256>>>(d, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double waveinstr = 256.0*16*4*ITERS*8;
  printf("%-18s %.3f ms  %.2f cycles per wave-instr per SIMD (assuming 2.4 GHz, 1024 SIMDs)\n", name, ms,
   ms*1e-3*2.4e9*1024/waveinstr);
  (void)hipFree(d);
}
int main() {
  run<0>("v_add_u32"); run<1>("v_mad_i32_i24"); run<2>("v_mul_lo_u32"); run<3>("v_ashrrev_i32");
  run<4>("v_mul_i32_i24"); run<5>("v_mul_hi_i32"); run<6>("v_add3_u32"); run<7>("v_mad_u32_u24");
  run<8>("v_fma_f32"); run<9>("v_lshl_add_u32"); run<10>("v_mad_i32_i16"); run<11>("v_pk_mul_lo_u16");
  run<12>("v_mul_u32_u24");
  return 0;
}
