/* dering_kernels.hip - the deringing filter of the reference (src/dering.c),
   SURVEY.md 8(f) rank 1: direction search per 8x8 block (od_dir_find8),
   directional smoothing with taps {3,2,1} and the orthogonal pass, one
   superblock (64x64 luma / 32x32 chroma) per workgroup.

   A superblock reads only the UNFILTERED plane x (its own samples plus a 3-sample
   border, src/dering.c:270-279), so every superblock of every plane - and every
   candidate threshold the encoder's level search tries (src/encode.c:2785-2810)
   - is independent: the x tile and the direction search are done once per
   workgroup, the two filter passes once per candidate.

   LDS: tile A = x with border (30000 where the border is outside the frame, as
   the reference's OD_DERING_VERY_LARGE), tile B = the directional pass with
   A's border (the reference copies y back into `in`, :330-334).  Arithmetic is
   the reference's int16 arithmetic, casts included. */
#include "../../include/daala_hip.h"
#include <stdlib.h>
#include <string.h>
#include "od_common.cuh"

namespace {

constexpr int kBorder = 3;          /* OD_FILT_BORDER */
constexpr int kVeryLarge = 30000;   /* OD_DERING_VERY_LARGE */

/* OD_DIRECTION_OFFSETS_TABLE, src/dering.c:39-48, as (dy, dx) for k = 1..3. */
__constant__ signed char kDirStep[8][3][2] = {
  {{-1, 1}, {-2, 2}, {-3, 3}},
  {{0, 1}, {-1, 2}, {-1, 3}},
  {{0, 1}, {0, 2}, {0, 3}},
  {{0, 1}, {1, 2}, {1, 3}},
  {{1, 1}, {2, 2}, {3, 3}},
  {{1, 0}, {2, 1}, {3, 1}},
  {{1, 0}, {2, 0}, {3, 0}},
  {{1, 0}, {2, -1}, {3, -1}}
};

/* OD_THRESH_TABLE_Q8, src/dering.c:225-229. */
__constant__ short kThreshQ8[18] = {
  128, 134, 150, 168, 188, 210, 234, 262, 292, 327, 365, 408, 455, 509, 569, 635, 710, 768};

struct DeringArgs {
  const int16_t *x;
  int16_t *y;             /* [plane][cand][h][w] */
  int32_t *dirs;          /* [plane][nvsb*8][nhsb*8] */
  const uint8_t *bskip;   /* [plane][...][skip_stride] */
  const int32_t *thr;     /* [plane][cand][nvsb*nhsb] */
  long x_plane_stride;    /* samples */
  long bskip_plane_stride;
  int stride;             /* samples per row of x and y */
  int nhsb;
  int nvsb;
  int ncand;
  int pli;
  int skip_stride;
  int overlap;
  int coeff_shift;
  /* per-call mode: x is an (n + 6) x (n + 6) window with its border already in
     place (sentinels included), the superblock's frame position is given here */
  int window;
  int sbx;
  int sby;
  int fnhsb;
  int fnvsb;
};

/* od_dir_find8, src/dering.c:61-124, on an 8x8 block of tile A (pitch P). */
template <int P>
__device__ __forceinline__ int dir_find8(const short *img, int coeff_shift, int *var) {
  int partial[8][15];
#pragma unroll
  for (int d = 0; d < 8; d++) {
#pragma unroll
    for (int t = 0; t < 15; t++) partial[d][t] = 0;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int x = img[i*P + j] >> coeff_shift;
      partial[0][i + j] += x;
      partial[1][i + j/2] += x;
      partial[2][i] += x;
      partial[3][3 + i - j/2] += x;
      partial[4][7 + i - j] += x;
      partial[5][3 - i/2 + j] += x;
      partial[6][j] += x;
      partial[7][i/2 + j] += x;
    }
  }
  constexpr int DIV[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
  int cost[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    cost[2] += partial[2][i]*partial[2][i];
    cost[6] += partial[6][i]*partial[6][i];
  }
  cost[2] *= DIV[8];
  cost[6] *= DIV[8];
#pragma unroll
  for (int i = 0; i < 7; i++) {
    cost[0] += (partial[0][i]*partial[0][i] + partial[0][14 - i]*partial[0][14 - i])*DIV[i + 1];
    cost[4] += (partial[4][i]*partial[4][i] + partial[4][14 - i]*partial[4][14 - i])*DIV[i + 1];
  }
  cost[0] += partial[0][7]*partial[0][7]*DIV[8];
  cost[4] += partial[4][7]*partial[4][7]*DIV[8];
#pragma unroll
  for (int i = 1; i < 8; i += 2) {
#pragma unroll
    for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j]*partial[i][3 + j];
    cost[i] *= DIV[8];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      cost[i] += (partial[i][j]*partial[i][j] + partial[i][10 - j]*partial[i][10 - j])*DIV[2*j + 2];
    }
  }
  int best_cost = 0;
  int best_dir = 0;
  int orth = cost[4];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (cost[i] > best_cost) {
      best_cost = cost[i];
      best_dir = i;
      orth = cost[(i + 4) & 7];
    }
  }
  *var = (best_cost - orth) >> 10;
  return best_dir;
}

template <int XDEC>
__global__ __launch_bounds__(256) void k_dering(DeringArgs a) {
  constexpr int N = 64 >> XDEC;           /* superblock side */
  constexpr int BS = 3 - XDEC;            /* log2 of the block side: 8x8 luma, 4x4 chroma */
  constexpr int P = N + 2*kBorder + 2;    /* LDS pitch in shorts (even) */
  constexpr int NPX = N*N/256;            /* pixels per thread */
  __shared__ short A[(N + 2*kBorder)*P];
  __shared__ short B[(N + 2*kBorder)*P];
  __shared__ int s_dir[64];
  __shared__ int s_var[64];
  __shared__ int s_thr[64];
  __shared__ unsigned char s_skip[64];
  __shared__ __attribute__((aligned(16))) int s_off[64][4];   /* LDS offsets of the three directional taps; [3] = orthogonal step */
  const int tid = threadIdx.x;
  const int plane = blockIdx.z;
  const int sbx = a.window ? a.sbx : blockIdx.x;
  const int sby = a.window ? a.sby : blockIdx.y;
  const int nhsb = a.window ? a.fnhsb : a.nhsb;
  const int nvsb = a.window ? a.fnvsb : a.nvsb;
  const int16_t *xp = a.x + plane*a.x_plane_stride;
  /* tile A: the superblock and its border (src/dering.c:270-279) */
  for (int t = tid; t < (N + 2*kBorder)*(N + 2*kBorder); t += 256) {
    const int i = t/(N + 2*kBorder) - kBorder;
    const int j = t%(N + 2*kBorder) - kBorder;
    short v = kVeryLarge;
    if (a.window) v = xp[(long)(i + kBorder)*a.stride + j + kBorder];
    else {
      const bool in_y = i >= -kBorder*(sby != 0) && i < N + kBorder*(sby != nvsb - 1);
      const bool in_x = j >= -kBorder*(sbx != 0) && j < N + kBorder*(sbx != nhsb - 1);
      if (in_y && in_x) v = xp[(long)(sby*N + i)*a.stride + sbx*N + j];
    }
    A[(i + kBorder)*P + j + kBorder] = v;
    B[(i + kBorder)*P + j + kBorder] = v;
  }
  __syncthreads();
  const short *Ai = A + kBorder*P + kBorder;
  short *Bi = B + kBorder*P + kBorder;
  int32_t *dirs = a.dirs + (long)plane*a.nvsb*8*a.nhsb*8;
  const long dir_row = (long)a.nhsb*8;
  /* one thread per block: direction (luma) / stored direction (chroma), skip test */
  if (tid < 64) {
    const int by = tid >> 3;
    const int bx = tid & 7;
    int dir;
    int var = 0;
    const long dpos = a.window ? tid : ((long)sby*8 + by)*dir_row + sbx*8 + bx;
    if (a.pli == 0) {
      dir = dir_find8<P>(Ai + (by*8)*P + bx*8, a.coeff_shift, &var);
      dirs[dpos] = dir;
    }
    else dir = dirs[dpos];
    s_dir[tid] = dir;
    s_var[tid] = var;
    /* a __constant__ table indexed by a per-lane direction would be read with
       up to eight different addresses per wavefront: resolve it once per block */
#pragma unroll
    for (int k = 0; k < 3; k++) s_off[tid][k] = kDirStep[dir][k][0]*P + kDirStep[dir][k][1];
    s_off[tid][3] = dir > 0 && dir < 4 ? P : 1;
    /* src/dering.c:306-325 */
    int xstart = 0;
    int ystart = 0;
    int xend = 2 >> XDEC;
    int yend = 2 >> XDEC;
    if (a.overlap) {
      xstart -= sbx != 0;
      ystart -= sby != 0;
      xend += sbx != nhsb - 1;
      yend += sby != nvsb - 1;
    }
    const uint8_t *bs = a.bskip + plane*a.bskip_plane_stride;
    if (!a.window) bs += (long)(sby << (4 - XDEC))*a.skip_stride + (sbx << (4 - XDEC));
    int skip = 1;
    for (int i = ystart; i < yend; i++) {
      for (int j = xstart; j < xend; j++) {
        skip = skip && bs[(long)((by << 1 >> XDEC) + i)*a.skip_stride + (bx << 1 >> XDEC) + j];
      }
    }
    s_skip[tid] = (unsigned char)skip;
  }
  __syncthreads();
  const long nsb = (long)a.nhsb*a.nvsb;
  for (int c = 0; c < a.ncand; c++) {
    const int threshold = a.window ? a.thr[c] : a.thr[((long)plane*a.ncand + c)*nsb + (long)sby*a.nhsb + sbx];
    if (tid < 64) {
      int th = threshold;
      if (a.pli == 0) {
        /* od_compute_thresh, src/dering.c:237-250 */
        int v1 = s_var[tid] >> 6;
        v1 = v1 < 32767 ? v1 : 32767;
        const int il = v1 ? 32 - __clz(v1) : 0;
        th = (threshold*kThreshQ8[il] + 128) >> 8;
      }
      s_thr[tid] = s_skip[tid] ? 0 : th;
    }
    __syncthreads();
    /* directional pass, src/dering.c:132-159 */
    short out[NPX];
#pragma unroll
    for (int t = 0; t < NPX; t++) {
      const int px = t*256 + tid;
      const int i = px/N;
      const int j = px%N;
      const int blk = (i >> BS)*8 + (j >> BS);
      const int th = s_thr[blk];
      const int4 offs = *reinterpret_cast<const int4 *>(s_off[blk]);
      const int off3[3] = {offs.x, offs.y, offs.z};
      const short xx = Ai[i*P + j];
      short sum = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int o = off3[k];
        const short p0 = (short)(Ai[i*P + j + o] - xx);
        const short p1 = (short)(Ai[i*P + j - o] - xx);
        if (abs((int)p0) < th) sum = (short)(sum + (3 - k)*p0);
        if (abs((int)p1) < th) sum = (short)(sum + (3 - k)*p1);
      }
      out[t] = (short)(xx + ((sum + 8) >> 4));
    }
    __syncthreads();   /* previous candidate's orthogonal pass has finished reading B */
#pragma unroll
    for (int t = 0; t < NPX; t++) {
      const int px = t*256 + tid;
      Bi[(px/N)*P + px%N] = out[t];
    }
    __syncthreads();
    /* orthogonal pass, src/dering.c:172-208 */
    int16_t *yp = a.y + ((long)plane*a.ncand + c)*(a.window ? (long)N*N : a.x_plane_stride);
#pragma unroll
    for (int t = 0; t < NPX; t++) {
      const int px = t*256 + tid;
      const int i = px/N;
      const int j = px%N;
      const int blk = (i >> BS)*8 + (j >> BS);
      const int th = s_thr[blk];
      const int offset = s_off[blk][3];
      const short yy = Bi[i*P + j];
      const int tt = th/3 + abs((int)yy - (int)Ai[i*P + j]);
      const short athresh = (short)(th < tt ? th : tt);
      short sum = 0;
      short p = (short)(Bi[i*P + j + offset] - yy);
      if (abs((int)p) < athresh) sum = (short)(sum + p);
      p = (short)(Bi[i*P + j - offset] - yy);
      if (abs((int)p) < athresh) sum = (short)(sum + p);
      p = (short)(Bi[i*P + j + 2*offset] - yy);
      if (abs((int)p) < athresh) sum = (short)(sum + p);
      p = (short)(Bi[i*P + j - 2*offset] - yy);
      if (abs((int)p) < athresh) sum = (short)(sum + p);
      const short r = (short)(yy + ((3*sum + 8) >> 4));
      if (a.window) yp[i*N + j] = r;
      else yp[(long)(sby*N + i)*a.stride + sbx*N + j] = r;
    }
  }
}

}  // namespace

extern "C" int odhip_dering_planes(int16_t *d_y, const int16_t *d_x, int stride, int nhsb, int nvsb,
 int xdec, int nplanes, int32_t *d_dirs, int pli, const uint8_t *d_bskip, int skip_stride,
 long bskip_plane_stride, const int32_t *d_thresholds, int ncand, int overlap, int coeff_shift,
 odhip_stream stream) {
  if (nplanes == 0 || ncand == 0) return ODHIP_SUCCESS;
  if (!d_y || !d_x || !d_dirs || !d_bskip || !d_thresholds || nhsb <= 0 || nvsb <= 0 || nplanes < 0
   || ncand < 0 || (xdec != 0 && xdec != 1) || stride < (nhsb*64 >> xdec) || coeff_shift < 0
   || coeff_shift > 15 || (pli == 0 && xdec != 0)) {   /* the direction search is 8x8 on luma */
    return ODHIP_EINVAL;
  }
  DeringArgs a;
  memset(&a, 0, sizeof(a));
  a.x = d_x;
  a.y = d_y;
  a.dirs = d_dirs;
  a.bskip = d_bskip;
  a.thr = d_thresholds;
  a.x_plane_stride = (long)stride*(nvsb*64 >> xdec);
  a.bskip_plane_stride = bskip_plane_stride;
  a.stride = stride;
  a.nhsb = nhsb;
  a.nvsb = nvsb;
  a.ncand = ncand;
  a.pli = pli;
  a.skip_stride = skip_stride;
  a.overlap = overlap;
  a.coeff_shift = coeff_shift;
  const dim3 grid(nhsb, nvsb, nplanes);
  if (xdec) k_dering<1><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else k_dering<0><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return odhip_check_launch();
}

/* Per-call surface with od_dering's argument list (src/dering.h:64-69) minus
   the function table: host pointers, synchronous. */
extern "C" void od_dering_hip(int16_t *y, int ystride, const int16_t *x, int xstride, int nhb, int nvb,
 int sbx, int sby, int nhsb, int nvsb, int xdec, int dir[8][8], int pli, unsigned char *bskip,
 int skip_stride, int threshold, int overlap, int coeff_shift) {
  const int n = 64 >> xdec;
  const int wn = n + 2*kBorder;
  if (nhb != 8 || nvb != 8 || (xdec != 0 && xdec != 1) || (pli == 0 && xdec != 0)) {
    fprintf(stderr, "libdaalahip: od_dering_hip supports full superblocks only (nhb = nvb = 8)\n");
    abort();
  }
  /* marshal the window the reference builds in `inbuf` (data movement only) */
  int16_t win[(64 + 6)*(64 + 6)];
  for (int i = -kBorder; i < n + kBorder; i++) {
    for (int j = -kBorder; j < n + kBorder; j++) {
      const bool in_y = i >= -kBorder*(sby != 0) && i < n + kBorder*(sby != nvsb - 1);
      const bool in_x = j >= -kBorder*(sbx != 0) && j < n + kBorder*(sbx != nhsb - 1);
      win[(i + kBorder)*wn + j + kBorder] = in_y && in_x ? x[(long)i*xstride + j] : (int16_t)kVeryLarge;
    }
  }
  /* the skip flags the test can touch: rows/cols -1 .. 16 >> xdec of the map */
  const int sn = (16 >> xdec) + 2;
  unsigned char skipwin[18*18];
  memset(skipwin, 1, sizeof(skipwin));
  for (int i = -(sby != 0); i < (16 >> xdec) + (sby != nvsb - 1); i++) {
    for (int j = -(sbx != 0); j < (16 >> xdec) + (sbx != nhsb - 1); j++) {
      skipwin[(i + 1)*sn + j + 1] = bskip[(long)i*skip_stride + j];
    }
  }
  int16_t *d_win;
  int16_t *d_y;
  int32_t *d_dir;
  uint8_t *d_skip;
  int32_t *d_thr;
  const size_t wbytes = (size_t)wn*wn*sizeof(int16_t);
  const size_t ybytes = (size_t)n*n*sizeof(int16_t);
  char *buf;
  if (hipMalloc((void **)&buf, wbytes + ybytes + 64*4 + sizeof(skipwin) + 16 + 64) != hipSuccess) {
    fprintf(stderr, "libdaalahip: od_dering_hip: hipMalloc failed (no CPU fallback)\n");
    abort();
  }
  size_t o = 0;
  d_win = (int16_t *)(buf + o);
  o += (wbytes + 15) & ~(size_t)15;
  d_y = (int16_t *)(buf + o);
  o += (ybytes + 15) & ~(size_t)15;
  d_dir = (int32_t *)(buf + o);
  o += 64*4;
  d_thr = (int32_t *)(buf + o);
  o += 16;
  d_skip = (uint8_t *)(buf + o);
  int dflat[64];
  for (int i = 0; i < 64; i++) dflat[i] = dir[i >> 3][i & 7];
  bool ok = hipMemcpy(d_win, win, wbytes, hipMemcpyHostToDevice) == hipSuccess
   && hipMemcpy(d_dir, dflat, sizeof(dflat), hipMemcpyHostToDevice) == hipSuccess
   && hipMemcpy(d_thr, &threshold, sizeof(int), hipMemcpyHostToDevice) == hipSuccess
   && hipMemcpy(d_skip, skipwin, sizeof(skipwin), hipMemcpyHostToDevice) == hipSuccess;
  DeringArgs a;
  memset(&a, 0, sizeof(a));
  a.x = d_win;
  a.y = d_y;
  a.dirs = d_dir;
  a.bskip = d_skip + sn + 1;   /* entry (0, 0) of the superblock's skip map */
  a.thr = d_thr;
  a.stride = wn;
  a.nhsb = 1;
  a.nvsb = 1;
  a.ncand = 1;
  a.pli = pli;
  a.skip_stride = sn;
  a.overlap = overlap;
  a.coeff_shift = coeff_shift;
  a.window = 1;
  a.sbx = sbx;
  a.sby = sby;
  a.fnhsb = nhsb;
  a.fnvsb = nvsb;
  if (ok) {
    if (xdec) k_dering<1><<<dim3(1, 1, 1), 256>>>(a);
    else k_dering<0><<<dim3(1, 1, 1), 256>>>(a);
    int16_t ytmp[64*64];
    ok = hipGetLastError() == hipSuccess
     && hipMemcpy(ytmp, d_y, ybytes, hipMemcpyDeviceToHost) == hipSuccess
     && hipMemcpy(dflat, d_dir, sizeof(dflat), hipMemcpyDeviceToHost) == hipSuccess;
    if (ok) {
      for (int i = 0; i < n; i++) memcpy(y + (long)i*ystride, ytmp + i*n, n*sizeof(int16_t));
      for (int i = 0; i < 64; i++) dir[i >> 3][i & 7] = dflat[i];
    }
  }
  (void)hipFree(buf);
  if (!ok) {
    fprintf(stderr, "libdaalahip: od_dering_hip failed on the device (no CPU fallback)\n");
    abort();
  }
}
