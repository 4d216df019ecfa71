/* lapped_pyramid_exp.cuh - forward-pyramid kernel shapes that rounds 2-3 measured against the default
   (tools/pyr_variants.py, profiles/r3_pyramid_experiments.txt).  Compiled only into the experiments
   build (-DODHIP_EXPERIMENTS, ODHIP_PYR_VARIANT bit 2); included by lapped_kernels.hip inside its
   anonymous namespace, after the default kernels.  All variants are bit-identical to the default. */
#pragma once

/* ONE luma superblock per 128-thread workgroup, and after the 64-point level NO
   workgroup barrier at all: blocks of 32x32 and smaller never straddle the
   horizontal mid-line of the superblock, so each of the two waves owns one half
   (32 rows x 64 columns) of the tile and runs the 32-, 16-, 8- and 4-point levels of
   its half on its own - column pass, row pass, split pre-filters and stores - with
   only "my LDS operations have completed" between the phases (LDS operations of one
   wave execute in order; od_wave_sync waits for their data and keeps the compiler
   from moving accesses across it).  The two waves of a workgroup and the six
   workgroups of a CU drift apart, so the arithmetic of one overlaps the stores and
   LDS traffic of the others instead of all meeting at a barrier eighteen times per
   superblock. */
__device__ __forceinline__ void od_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

/* Levels LN <= 3 of rows [r0, r0 + 32) of a 64-wide tile, one wave. */
template <int LN, typename T>
__device__ __forceinline__ void half_level(short *t, int *z, const PyramidArgs &a, long plane_off,
 int x0, int y0, int r0, int lane) {
  constexpr int TILE = 64;
  constexpr int P = Geo<TILE>::kPitch;
  constexpr int ROWS = 32;
  if constexpr (LN == 0) {
    /* one 4x4 block per lane and iteration, both passes in registers */
    constexpr int NBX = TILE/4;
    for (int blk = lane; blk < NBX*(ROWS/4); blk += 64) {
      const int bx = blk % NBX;
      const int by = blk/NBX;
      T m[4][4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const short4 v = *reinterpret_cast<const short4 *>(t + (r0 + by*4 + r)*P + bx*4);
        m[r][0] = T(v.x);
        m[r][1] = T(v.y);
        m[r][2] = T(v.z);
        m[r][3] = T(v.w);
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        T in[4] = {m[0][c], m[1][c], m[2][c], m[3][c]};
        T out[4];
        od_fdct4_lift(out, in);
        m[0][c] = out[0];
        m[1][c] = out[1];
        m[2][c] = out[2];
        m[3][c] = out[3];
      }
      if (!a.levels[0]) continue;
      od_coeff *plane = a.levels[0] + plane_off;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        T out[4];
        od_fdct4_lift(out, m[r]);
        od_store_coef4(plane + (long)(y0 + r0 + by*4 + r)*a.w + x0 + bx*4, make_int4(out[0], out[1], out[2], out[3]));
      }
    }
  }
  else {
    constexpr int N = 4 << LN;
    /* column pass: lane = (column, block row of the half) */
    for (int k = lane; k < TILE*(ROWS/N); k += 64) {
      const int x = k % TILE;
      const int by = k/TILE;
      const int base = (r0 + by*N)*P + x;
      T in[N];
      T out[N];
#pragma unroll
      for (int r = 0; r < N; r++) in[r] = T(t[base + r*P]);
      od_fdct_lift<LN>(out, in);
#pragma unroll
      for (int r = 0; r < N; r++) z[base + r*P] = out[r];
    }
    od_wave_sync();
    /* row pass in place: lane = (row of the half, block column), 16 bytes per access */
    for (int k = lane; k < ROWS*(TILE/N); k += 64) {
      const int y = k % ROWS;
      const int bx = k/ROWS;
      const int base = (r0 + y)*P + bx*N;
      T in[N];
      T out[N];
#pragma unroll
      for (int c = 0; c < N; c += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(z + base + c);
        in[c] = T(v.x);
        in[c + 1] = T(v.y);
        in[c + 2] = T(v.z);
        in[c + 3] = T(v.w);
      }
      od_fdct_lift<LN>(out, in);
#pragma unroll
      for (int c = 0; c < N; c += 4) {
        *reinterpret_cast<int4 *>(z + base + c) = make_int4(out[c], out[c + 1], out[c + 2], out[c + 3]);
      }
    }
    /* od_prefilter_split of this level's blocks, column taps (the tile is free) */
    for (int k = lane; k < TILE*(ROWS/N); k += 64) {
      const int x = k % TILE;
      const int by = k/TILE;
      const int gbx = (x0 + x)/N;
      if ((gbx + 1)*N <= a.pic_w) lds_filter4<false>(t + (r0 + by*N + N/2 - 2)*P + x, P);
    }
    od_wave_sync();
    if (a.levels[LN]) {
      od_coeff *plane = a.levels[LN] + plane_off;
      for (int i = lane; i < ROWS*TILE/4; i += 64) {
        const int y = r0 + i/(TILE/4);
        const int x = (i % (TILE/4))*4;
        od_store_coef4(plane + (long)(y0 + y)*a.w + x0 + x, *reinterpret_cast<const int4 *>(z + y*P + x));
      }
    }
    /* ... and row taps */
    for (int k = lane; k < ROWS*(TILE/N); k += 64) {
      const int y = r0 + k % ROWS;
      const int bx = k/ROWS;
      const int gby = (y0 + y)/N;
      if ((gby + 1)*N <= a.pic_h) lds_filter4<false>(t + y*P + bx*N + N/2 - 2, 1);
    }
    od_wave_sync();
    half_level<LN - 1, T>(t, z, a, plane_off, x0, y0, r0, lane);
  }
}

template <typename T>
__global__ __launch_bounds__(128) void k_forward_pyramid_halves(PyramidArgs a) {
  constexpr int TILE = 64;
  constexpr int NT = 128;
  using G = Geo<TILE>;
  constexpr int P = G::kPitch;
  __shared__ __attribute__((aligned(16))) short t[G::kHaloWords];
  __shared__ __attribute__((aligned(16))) int z[TILE*P];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x*TILE;
  const int y0 = blockIdx.y*TILE;
  const uint8_t *px = pyr_plane(a, blockIdx.z);
  const long plane_off = (long)blockIdx.z*a.w*a.h;
  sb_load<TILE, NT>(&t, a, px, x0, y0, tid);
  od_lds_barrier();
  sb_edge_cols<TILE, NT>(t, a, x0, y0, tid);
  od_lds_barrier();
  sb_edge_rows<TILE, NT>(t, a, x0, tid);
  od_lds_barrier();
  /* 64-point level: the two waves run the even and the odd half network of every
     column, then of every row (workgroup barriers inside); its split pre-filter
     crosses the mid-line between the halves, so one more barrier follows it */
  pyramid_level_split64<4, T, NT>(t, z, a, plane_off, x0, y0, tid);
  half_level<3, T>(t, z, a, plane_off, x0, y0, (tid >> 6)*32, tid & 63);
}

