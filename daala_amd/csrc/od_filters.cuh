/* od_filters.cuh - the 4/8/16/32-point lapping filters of the reference
   (src/filter.c:147-1321, the TYPE3 variants its `#elif 1` chains select) as
   one compile-time-sized network: +1/-1 butterflies, Q6 scaling of the high
   half with the "+1 if positive" that makes it invertible (skipped for a factor
   of 64), the rotation ladder from the top pair down, butterflies back; the
   post-filter is the exact inverse, its scalings undone by C truncating
   divisions.  Fully unrolled: t[] lives in registers. */
#pragma once
#include "gen/od_filter_params.h"

template <int N>
struct OdFilterParams;
template <>
struct OdFilterParams<4> {
  __device__ static constexpr int get(int i) { return OD_FPARAMS4[i]; }
};
template <>
struct OdFilterParams<8> {
  __device__ static constexpr int get(int i) { return OD_FPARAMS8[i]; }
};
template <>
struct OdFilterParams<16> {
  __device__ static constexpr int get(int i) { return OD_FPARAMS16[i]; }
};
template <>
struct OdFilterParams<32> {
  __device__ static constexpr int get(int i) { return OD_FPARAMS32[i]; }
};

template <int N>
__device__ __forceinline__ void od_pre_filter_dev(int (&t)[N]) {
  constexpr int h = N/2;
  using P = OdFilterParams<N>;
#pragma unroll
  for (int i = 0; i < h; i++) t[N - 1 - i] = t[i] - t[N - 1 - i];
#pragma unroll
  for (int i = 0; i < h; i++) t[i] = t[i] - (t[N - 1 - i] >> 1);
#pragma unroll
  for (int i = 0; i < h; i++) {
    if (P::get(i) != 64) {
      t[h + i] = t[h + i]*P::get(i) >> 6;
      t[h + i] += t[h + i] > 0;
    }
  }
#pragma unroll
  for (int k = h - 2; k >= 0; k--) {
    t[h + k + 1] += (t[h + k]*P::get(h + k) + 32) >> 6;
    t[h + k] += (t[h + k + 1]*P::get(2*h - 1 + k) + 32) >> 6;
  }
#pragma unroll
  for (int i = 0; i < h; i++) {
    t[i] += t[N - 1 - i] >> 1;
    t[N - 1 - i] = t[i] - t[N - 1 - i];
  }
}

template <int N>
__device__ __forceinline__ void od_post_filter_dev(int (&t)[N]) {
  constexpr int h = N/2;
  using P = OdFilterParams<N>;
#pragma unroll
  for (int i = 0; i < h; i++) t[N - 1 - i] = t[i] - t[N - 1 - i];
#pragma unroll
  for (int i = 0; i < h; i++) t[i] = t[i] - (t[N - 1 - i] >> 1);
#pragma unroll
  for (int k = 0; k <= h - 2; k++) {
    t[h + k] -= (t[h + k + 1]*P::get(2*h - 1 + k) + 32) >> 6;
    t[h + k + 1] -= (t[h + k]*P::get(h + k) + 32) >> 6;
  }
#pragma unroll
  for (int i = 0; i < h; i++) {
    if (P::get(i) != 64) t[h + i] = t[h + i]*64/P::get(i);
  }
#pragma unroll
  for (int i = 0; i < h; i++) {
    t[i] += t[N - 1 - i] >> 1;
    t[N - 1 - i] = t[i] - t[N - 1 - i];
  }
}
