/* od_lift.cuh - device-side 1-D lifting transforms and the 4-point lapping
   filter of the Daala block-transform path.

   The lifting networks themselves are generated (gen/od_lifting_gen.h, from
   tools/extract_lifting.py) and restate od_bin_fdct4..64 / od_bin_idct4..64
   (reference src/dct.c:87-790, 4219-4820).  This header supplies the two
   primitives they are written in and size-indexed dispatch.

   Integer multiply flavour.  CDNA4 issues v_mul_lo_u32 at quarter rate but
   v_mad_i32_i24 at full rate.  A 24-bit multiply returns the low 32 bits of
   the 48-bit product, which equals the reference's 32-bit `int` product
   whenever the multiplicand fits in 24 bits signed.  Inside the frame pipeline
   every multiplicand is bounded by construction (8-bit pixels << 4 through
   the lapping filters and transforms peak at 273 942 < 2^19, reference dcttest
   dynamic_range, SURVEY.md hard part 6), so OdMul24 is bit-exact there; the
   generic od_dct_func_2d surfaces, which must accept arbitrary od_coeff
   input, use OdMul32. */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

/* OD_DCT_RSHIFT(a, 1) == OD_UNBIASED_RSHIFT32(a, 1), src/filter.h:38-41. */
__device__ __forceinline__ int od_rs1_i32(int a) {
  return (int)(((unsigned)a >> 31) + (unsigned)a) >> 1;
}

/* A coefficient whose lifting multiply is the exact 32-bit wrapping product. */
struct OdMul32 {
  int v;
  __device__ __forceinline__ OdMul32() {}
  __device__ __forceinline__ OdMul32(int x) : v(x) {}
  __device__ __forceinline__ operator int() const { return v; }
};

/* A coefficient known to fit 24 bits: lifting multiply through v_mad_i32_i24. */
struct OdMul24 {
  int v;
  __device__ __forceinline__ OdMul24() {}
  __device__ __forceinline__ OdMul24(int x) : v(x) {}
  __device__ __forceinline__ operator int() const { return v; }
};

/* OdMul24 with OD_DCT_RSHIFT(a, 1) in TWO instructions.  (a + (a < 0)) >> 1 costs
   v_lshrrev 31 / v_add / v_ashrrev; when |a| < 2^23 the top byte of a is pure sign,
   so one SDWA subtract of the sign-extended byte 3 (0 or -1) adds the "+1 if
   negative" and only the shift remains.  The frame pipeline's values peak below
   2^19 (see above). */
struct OdMul24S {
  int v;
  __device__ __forceinline__ OdMul24S() {}
  __device__ __forceinline__ OdMul24S(int x) : v(x) {}
  __device__ __forceinline__ operator int() const { return v; }
};

__device__ __forceinline__ int od_rs1_sdwa(int a) {
  int t;
  asm("v_sub_u32_sdwa %0, %1, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
      : "=v"(t) : "v"(a));
  return t >> 1;
}

#define OD_DEF_COEF_OPS(T) \
  __device__ __forceinline__ T operator+(T a, T b) { return T(a.v + b.v); } \
  __device__ __forceinline__ T operator-(T a, T b) { return T(a.v - b.v); } \
  __device__ __forceinline__ T operator-(T a) { return T(-a.v); } \
  __device__ __forceinline__ T &operator+=(T &a, T b) { a.v += b.v; return a; } \
  __device__ __forceinline__ T &operator-=(T &a, T b) { a.v -= b.v; return a; }
#define OD_DEF_COEF_RS1(T, fn) \
  __device__ __forceinline__ T od_rs1(T a) { return T(fn(a.v)); }
OD_DEF_COEF_OPS(OdMul32)
OD_DEF_COEF_OPS(OdMul24)
OD_DEF_COEF_OPS(OdMul24S)
OD_DEF_COEF_RS1(OdMul32, od_rs1_i32)
OD_DEF_COEF_RS1(OdMul24, od_rs1_i32)
OD_DEF_COEF_RS1(OdMul24S, od_rs1_sdwa)

/* (a*C + R) >> S with arithmetic shift, the lifting step of src/dct.c. */
__device__ __forceinline__ OdMul32 od_lift(OdMul32 a, int c, int r, int s) {
  return OdMul32((int)((unsigned)a.v*(unsigned)c + (unsigned)r) >> s);
}

__device__ __forceinline__ OdMul24 od_lift(OdMul24 a, int c, int r, int s) {
  return OdMul24((__mul24(a.v, c) + r) >> s);
}

__device__ __forceinline__ OdMul24S od_lift(OdMul24S a, int c, int r, int s) {
  return OdMul24S((__mul24(a.v, c) + r) >> s);
}

#include "gen/od_lifting_gen.h"

/* Size-indexed dispatch: LN = log2(N) - 2. */
template <int LN, typename T>
__device__ __forceinline__ void od_fdct_lift(T (&out)[4 << LN], const T (&in)[4 << LN]) {
  if constexpr (LN == 0) od_fdct4_lift(out, in);
  else if constexpr (LN == 1) od_fdct8_lift(out, in);
  else if constexpr (LN == 2) od_fdct16_lift(out, in);
  else if constexpr (LN == 3) od_fdct32_lift(out, in);
  else od_fdct64_lift(out, in);
}

/* Half networks: out[k] = y[2k + PARITY].  An N-point forward transform splits
   after its first butterfly stage into two independent N/2-point halves (even
   outputs: asymmetric DCT of the sums, odd outputs: asymmetric DST of the
   differences), so two lanes can share one column. */
template <int LN, int PARITY, typename T>
__device__ __forceinline__ void od_fdct_lift_half(T (&out)[2 << LN], const T (&in)[4 << LN]) {
  static_assert(LN >= 2, "half networks exist for 16, 32 and 64 points");
  if constexpr (LN == 2) {
    if constexpr (PARITY == 0) od_fdct16_lift_even(out, in);
    else od_fdct16_lift_odd(out, in);
  }
  else if constexpr (LN == 3) {
    if constexpr (PARITY == 0) od_fdct32_lift_even(out, in);
    else od_fdct32_lift_odd(out, in);
  }
  else {
    if constexpr (PARITY == 0) od_fdct64_lift_even(out, in);
    else od_fdct64_lift_odd(out, in);
  }
}

/* Split inverse networks (round 5; generated: tools/extract_lifting.py emit_device_inverse_split): part P
   runs the sub-network fed by the inputs of parity P alone (in_P[k] = in[2k + P]) and leaves the values
   that cross into the joining stage in mid[]; join H produces outputs H*N/2 .. H*N/2 + N/2 - 1 from both
   parts' mid[] - of the other part's only the entries kIdctNNeedIdxH lists.  Two lanes can share one line:
   each runs one part, exports what the other's join needs, and joins its own half. */
template <int PART, typename T>
__device__ __forceinline__ void od_idct64_lift_part(T (&mid)[48], const T (&in)[32]) {
  static_assert(kIdct64Mid0 == 48 && kIdct64Mid1 == 48, "array sizes below");
  if constexpr (PART == 0) od_idct64_lift_part0(mid, in);
  else od_idct64_lift_part1(mid, in);
}

template <int HALF, typename T>
__device__ __forceinline__ void od_idct64_lift_join(T (&out)[32], const T (&m0)[48], const T (&m1)[48]) {
  if constexpr (HALF == 0) od_idct64_lift_join0(out, m0, m1);
  else od_idct64_lift_join1(out, m0, m1);
}

template <int LN, typename T>
__device__ __forceinline__ void od_idct_lift(T (&out)[4 << LN], const T (&in)[4 << LN]) {
  if constexpr (LN == 0) od_idct4_lift(out, in);
  else if constexpr (LN == 1) od_idct8_lift(out, in);
  else if constexpr (LN == 2) od_idct16_lift(out, in);
  else if constexpr (LN == 3) od_idct32_lift(out, in);
  else od_idct64_lift(out, in);
}

/* od_pre_filter4, src/filter.c:147-193, OD_FILTER_PARAMS4 = {85,75,-15,33}.
   In place on four samples t0..t3 (t0,t1 on one side of the edge). */
__device__ __forceinline__ void od_pre_filter4_dev(int &t0, int &t1, int &t2, int &t3) {
  int d30 = t0 - t3;
  int d21 = t1 - t2;
  int s1 = t1 - (d21 >> 1);
  int s0 = t0 - (d30 >> 1);
  d21 = d21*85 >> 6;
  d21 += d21 > 0;
  d30 = d30*75 >> 6;
  d30 += d30 > 0;
  d30 += (d21*-15 + 32) >> 6;
  d21 += (d30*33 + 32) >> 6;
  s0 += d30 >> 1;
  s1 += d21 >> 1;
  t0 = s0;
  t1 = s1;
  t2 = s1 - d21;
  t3 = s0 - d30;
}

/* od_post_filter4, src/filter.c:195-222: truncating divisions by 75 and 85. */
__device__ __forceinline__ void od_post_filter4_dev(int &t0, int &t1, int &t2, int &t3) {
  int d30 = t0 - t3;
  int d21 = t1 - t2;
  int s1 = t1 - (d21 >> 1);
  int s0 = t0 - (d30 >> 1);
  d21 -= (d30*33 + 32) >> 6;
  d30 -= (d21*-15 + 32) >> 6;
  d30 = d30*64/75;
  d21 = d21*64/85;
  s0 += d30 >> 1;
  s1 += d21 >> 1;
  t0 = s0;
  t1 = s1;
  t2 = s1 - d21;
  t3 = s0 - d30;
}

/* od_post_filter4 on values that fit 24 bits (the frame pipeline, see OdMul24): the two lifting
   multiplies through the 24-bit multiplier - written as plain C they compile to a 64-bit
   v_mad_u64_u32 each. */
__device__ __forceinline__ void od_post_filter4_dev24(int &t0, int &t1, int &t2, int &t3) {
  int d30 = t0 - t3;
  int d21 = t1 - t2;
  int s1 = t1 - (d21 >> 1);
  int s0 = t0 - (d30 >> 1);
  d21 -= (__mul24(d30, 33) + 32) >> 6;
  d30 -= (__mul24(d21, -15) + 32) >> 6;
  d30 = d30*64/75;
  d21 = d21*64/85;
  s0 += d30 >> 1;
  s1 += d21 >> 1;
  t0 = s0;
  t1 = s1;
  t2 = s1 - d21;
  t3 = s0 - d30;
}
