/* od_dpp.cuh - DPP row (16-lane) primitives: cross-lane moves that are VALU
   operations (quad_perm / row_half_mirror / row_mirror), no LDS traffic. */
#pragma once

#define OD_DPP_XOR1 0xB1        /* quad_perm [1,0,3,2] */
#define OD_DPP_XOR2 0x4E        /* quad_perm [2,3,0,1] */
#define OD_DPP_HALF_MIRROR 0x141
#define OD_DPP_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ int row_mov(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

template <int CTRL>
__device__ __forceinline__ double row_mov(double v) {
  const int lo = row_mov<CTRL>(__double2loint(v));
  const int hi = row_mov<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

/* Sum over the 16 lanes of a row, in every lane of the row. */
__device__ __forceinline__ int row_sum(int v) {
  v += row_mov<OD_DPP_XOR1>(v);
  v += row_mov<OD_DPP_XOR2>(v);
  v += row_mov<OD_DPP_HALF_MIRROR>(v);
  v += row_mov<OD_DPP_MIRROR>(v);
  return v;
}
