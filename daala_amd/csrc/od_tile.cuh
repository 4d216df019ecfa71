/* od_tile.cuh - separable 2-D transforms on a coefficient tile staged in LDS.

   A tile is TILE x TILE od_coeff (TILE = 64 for luma superblocks, 32 for 4:2:0
   chroma) stored row-major with pitch TILE + 4 words.  It carries a grid of
   (TILE/N)^2 transform blocks of side N = 4 << LN.

   * column pass: one lane per block column.  Lane addresses differ by one
     word, so ds_read_b32/ds_write_b32 are bank-conflict free at any pitch.
   * row pass: one lane per block row, moving 16 bytes per LDS instruction.
     Pitch = TILE + 4 puts row y at bank 4*y (mod 64): the sixteen lanes of
     every ds_read_b128 service group land on sixteen distinct 4-bank slots
     (MI355X LDS: 64 banks for b128, groups {0-3,12-15,20-27} ...), so the
     row pass is conflict free too; pitch TILE + 1 would break the 16-byte
     alignment b128 needs.

   The 2-D forward transform (od_bin_fdctNxN, reference src/dct.c:151-156,
   351-356, 792-798, 4890-4904) is columns-then-rows; the inverse
   (od_bin_idctNxN, :158-163, ...) is rows-then-columns. */
#pragma once
#include "od_lift.cuh"

template <int TILE>
struct OdTile {
  static constexpr int kPitch = TILE + 4;
  static constexpr int kWords = TILE*kPitch;
};

/* Column pass over every block of the tile: dst column <- transform(src
   column).  src == dst is allowed (each lane rewrites only what it read).
   `active(bx, by)` masks blocks (partial tiles, partition leaves). */
template <int TILE, int LN, bool INV, typename T, int NT, typename Pred, typename SrcE>
__device__ __forceinline__ void od_tile_cols(int *dst, const SrcE *src, int tid, Pred active) {
  constexpr int N = 4 << LN;
  constexpr int P = OdTile<TILE>::kPitch;
  constexpr int kTasks = TILE*(TILE/N);
  for (int t = tid; t < kTasks; t += NT) {
    const int x = t % TILE;
    const int by = t / TILE;
    if (!active(x / N, by)) continue;
    const int base = by*N*P + x;
    T in[N];
    T out[N];
#pragma unroll
    for (int r = 0; r < N; r++) in[r] = T(src[base + r*P]);
    if constexpr (INV) od_idct_lift<LN>(out, in);
    else od_fdct_lift<LN>(out, in);
#pragma unroll
    for (int r = 0; r < N; r++) dst[base + r*P] = out[r];
  }
}

/* Row pass over every block of the tile, in place or src -> dst. */
template <int TILE, int LN, bool INV, typename T, int NT, typename Pred>
__device__ __forceinline__ void od_tile_rows(int *dst, const int *src, int tid, Pred active) {
  constexpr int N = 4 << LN;
  constexpr int P = OdTile<TILE>::kPitch;
  constexpr int kTasks = TILE*(TILE/N);
  for (int t = tid; t < kTasks; t += NT) {
    const int y = t % TILE;
    const int bx = t / TILE;
    if (!active(bx, y / N)) continue;
    const int base = y*P + bx*N;
    T in[N];
    T out[N];
#pragma unroll
    for (int c = 0; c < N; c += 4) {
      const int4 v = *reinterpret_cast<const int4 *>(src + base + c);
      in[c] = T(v.x);
      in[c + 1] = T(v.y);
      in[c + 2] = T(v.z);
      in[c + 3] = T(v.w);
    }
    if constexpr (INV) od_idct_lift<LN>(out, in);
    else od_fdct_lift<LN>(out, in);
#pragma unroll
    for (int c = 0; c < N; c += 4) {
      *reinterpret_cast<int4 *>(dst + base + c) =
       make_int4(out[c], out[c + 1], out[c + 2], out[c + 3]);
    }
  }
}

struct OdAllBlocks {
  __device__ __forceinline__ bool operator()(int, int) const { return true; }
};
