/* od_sel.cuh - compare / select spelled as VOP3 (e64) instructions. */
#pragma once

namespace {

/* Selects through an SGPR-pair mask (VOP3 encodings).  Measured on gfx950
   (tools/ubench/fp64_rate.hip): back-to-back VOP2 `v_cndmask_b32 ..., vcc` -
   what the compiler emits for `c ? a : b` on doubles, two per select - issue at
   ~16-19 cycles each instead of ~4.4; the e64 forms do not.  The search loops
   are five selects per candidate, so they are spelled out. */
__device__ __forceinline__ unsigned long long od_cmp_gt(double a, double b) {
  unsigned long long m;
  asm("v_cmp_gt_f64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  return m;
}

__device__ __forceinline__ int od_sel(unsigned long long m, int t, int f) {
  int d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(f), "v"(t), "s"(m));
  return d;
}

/* The candidate index as a running VGPR: `pos = od_sel(m, j, pos)` with j a
   literal would need one VGPR per distinct j for the asm operand (the compiler
   hoists all of them: 127 registers for n = 128). */
__device__ __forceinline__ void od_inc(int &j) {
  asm("v_add_u32 %0, 1, %0" : "+v"(j));
}

/* (double)v for a non-negative int, spelled so that the compiler cannot hoist it out of a
   loop: the searches convert |x_j| where they use it - a double copy of the band kept across
   the pulse loops costs two VGPRs per coefficient and an occupancy step. */
__device__ __forceinline__ double od_cvt_u(int v) {
  double d;
  asm volatile("v_cvt_f64_u32_e32 %0, %1" : "=v"(d) : "v"(v));
  return d;
}

__device__ __forceinline__ double od_sel(unsigned long long m, double t, double f) {
  const int lo = od_sel(m, __double2loint(t), __double2loint(f));
  const int hi = od_sel(m, __double2hiint(t), __double2hiint(f));
  return __hiloint2double(hi, lo);
}

}  // namespace
