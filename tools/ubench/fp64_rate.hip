// fp64 / conversion / select issue-rate microbenchmark for gfx950 (development
// aid for the PVQ search loops): 8 independent chains per lane, inline asm.
// Also a "1 wave per SIMD" launch to see single-wave dependent-issue behaviour.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 1024
#define CHAIN8(ASM) \
  for (int it = 0; it < ITERS; it++) { \
    asm volatile(ASM : "+v"(a0) : "v"(c), "v"(ci)); asm volatile(ASM : "+v"(a1) : "v"(c), "v"(ci)); \
    asm volatile(ASM : "+v"(a2) : "v"(c), "v"(ci)); asm volatile(ASM : "+v"(a3) : "v"(c), "v"(ci)); \
    asm volatile(ASM : "+v"(a4) : "v"(c), "v"(ci)); asm volatile(ASM : "+v"(a5) : "v"(c), "v"(ci)); \
    asm volatile(ASM : "+v"(a6) : "v"(c), "v"(ci)); asm volatile(ASM : "+v"(a7) : "v"(c), "v"(ci)); \
  }
#define ICHAIN8(ASM) \
  for (int it = 0; it < ITERS; it++) { \
    asm volatile(ASM : "+v"(b0) : "v"(c), "v"(ci)); asm volatile(ASM : "+v"(b1) : "v"(c), "v"(ci)); \
    asm volatile(ASM : "+v"(b2) : "v"(c), "v"(ci)); asm volatile(ASM : "+v"(b3) : "v"(c), "v"(ci)); \
    asm volatile(ASM : "+v"(b4) : "v"(c), "v"(ci)); asm volatile(ASM : "+v"(b5) : "v"(c), "v"(ci)); \
    asm volatile(ASM : "+v"(b6) : "v"(c), "v"(ci)); asm volatile(ASM : "+v"(b7) : "v"(c), "v"(ci)); \
  }
#define CHAIN1(ASM) \
  for (int it = 0; it < ITERS*8; it++) { asm volatile(ASM : "+v"(a0) : "v"(c), "v"(ci)); }
template <int OP>
__global__ void k(double *out, int seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double c = seed*1e-9 + 1.0000001;
  int ci = seed + 5;
  int b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3, b4 = seed + 4, b5 = seed + 5, b6 = seed + 6, b7 = seed + 7;
  if (OP == 0) { CHAIN8("v_add_f64 %0, %0, %1") }
  if (OP == 1) { CHAIN8("v_mul_f64 %0, %0, %1") }
  if (OP == 2) { CHAIN8("v_fma_f64 %0, %0, %1, %1") }
  if (OP == 3) { CHAIN8("v_cvt_f64_u32 %0, %2") }
  if (OP == 4) { CHAIN8("v_cmp_gt_f64 vcc, %0, %1") }
  if (OP == 5) { ICHAIN8("v_cndmask_b32 %0, %0, %2, vcc") }
  if (OP == 15) { ICHAIN8("v_cndmask_b32_e64 %0, %0, %2, s[20:21]") }
  if (OP == 16) { for (int it = 0; it < ITERS; it++) { asm volatile("v_cmp_gt_i32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b0) : "v"(ci) : "vcc"); asm volatile("v_cmp_gt_i32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b1) : "v"(ci) : "vcc"); asm volatile("v_cmp_gt_i32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b2) : "v"(ci) : "vcc"); asm volatile("v_cmp_gt_i32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b3) : "v"(ci) : "vcc"); } }
  if (OP == 17) { for (int it = 0; it < ITERS; it++) { asm volatile("v_cmp_gt_i32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4) : "v"(ci) : "vcc"); asm volatile("v_cmp_gt_i32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc" : "+v"(b5), "+v"(b6), "+v"(b7), "+v"(b0), "+v"(b1) : "v"(ci) : "vcc"); } }
  if (OP == 19) { ICHAIN8("v_cndmask_b32_e64 %0, %0, %2, vcc") }
  if (OP == 20) { ICHAIN8("v_cndmask_b32 %0, %0, %2, vcc\n v_add_u32 %0, %0, %2") }
  if (OP == 21) { ICHAIN8("v_cndmask_b32 %0, %0, %2, vcc\n v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %2") }
  if (OP == 22) { ICHAIN8("v_addc_co_u32 %0, vcc, %0, %2, vcc") }
  if (OP == 23) { for (int it = 0; it < ITERS; it++) { asm volatile("v_cmp_gt_i32 vcc, %1, %0\n s_nop 0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc\n v_cndmask_b32 %5, %5, %1, vcc\n v_cndmask_b32 %6, %6, %1, vcc" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5) : "v"(ci) : "vcc"); } }
  if (OP == 24) { for (int it = 0; it < ITERS; it++) { asm volatile("v_cmp_gt_i32 s[20:21], %1, %0\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %1, s[20:21]\n v_cndmask_b32_e64 %3, %3, %1, s[20:21]\n v_cndmask_b32_e64 %4, %4, %1, s[20:21]\n v_cndmask_b32_e64 %5, %5, %1, s[20:21]\n v_cndmask_b32_e64 %6, %6, %1, s[20:21]\n s_nop 0" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5) : "v"(ci) : "s20", "s21"); } }
  if (OP == 18) { ICHAIN8("v_max_i32 %0, %0, %2") }
  if (OP == 6) { CHAIN8("v_cvt_f64_f32 %0, %2") }
  if (OP == 7) { CHAIN1("v_add_f64 %0, %0, %1") }
  if (OP == 8) { CHAIN1("v_mul_f64 %0, %0, %1") }
  if (OP == 9) { CHAIN8("v_mad_u64_u32 %0, vcc, %2, %2, %0") }
  if (OP == 10) { CHAIN8("v_max_f64 %0, %0, %1") }
  if (OP == 11) { ICHAIN8("v_lshrrev_b32 %0, 1, %0") }
  if (OP == 12) { CHAIN8("v_rcp_f64 %0, %0") }
  if (OP == 13) { CHAIN8("v_floor_f64 %0, %0") }
  if (OP == 14) { ICHAIN8("v_cvt_i32_f64 %0, %1") }
  out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7);
}
template <int OP> void run(const char *name, int wg_per_cu, int threads) {
  double *d; (void)hipMalloc(&d, 256*16*256*8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<OP><<<256*wg_per_cu, threads>>>(d, 3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<OP><<<256*wg_per_cu, threads>>>(d, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD
  double per_simd = (double)wg_per_cu*(threads/64)/4.0*ITERS*8;
  printf("%-22s wg/cu=%2d thr=%3d  %.3f ms  %.2f cycles per wave-instr per SIMD (2.4 GHz)\n", name, wg_per_cu, threads, ms,
   ms*1e-3*2.4e9/per_simd);
  (void)hipFree(d);
}
#define BOTH(OP, NAME) run<OP>(NAME, 16, 256); run<OP>(NAME, 1, 256);
int main() {
  BOTH(0, "v_add_f64") BOTH(1, "v_mul_f64") BOTH(2, "v_fma_f64") BOTH(3, "v_cvt_f64_u32")
  BOTH(4, "v_cmp_gt_f64") BOTH(5, "v_cndmask_b32") BOTH(6, "v_cvt_f64_f32")
  BOTH(7, "dep v_add_f64") BOTH(8, "dep v_mul_f64") BOTH(9, "v_mad_u64_u32") BOTH(10, "v_max_f64")
  BOTH(15, "cndmask_e64 sgpr") BOTH(16, "cmp+cndmask x4 (8 instr/it)") BOTH(17, "cmp+4cnd, cmp+3cnd (9/it)") BOTH(18, "v_max_i32") BOTH(19, "cndmask_e64 vcc") BOTH(20, "cnd+add (x2 instr)") BOTH(21, "cnd+3add (x4 instr)") BOTH(22, "v_addc vcc")
  BOTH(23, "cmp,nop,6cnd vcc (1 chain, 8/it)") BOTH(24, "cmp_e64,6cnd_e64 sgpr (8/it)")
  BOTH(11, "v_lshrrev_b32") BOTH(12, "v_rcp_f64") BOTH(13, "v_floor_f64") BOTH(14, "v_cvt_i32_f64")
  return 0;
}
