// 4-wide vector access + column-wise block reduction helpers shared by the HBM-bound kernels.
#pragma once
#include "common.h"
#include <cstdlib>

// ---- 4-wide vector helpers -------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void ld4(const T* p, float v[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float v[4]) { float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
template <> __device__ __forceinline__ void ld4<bf16>(const bf16* p, float v[4]) {
  uint2 t = *(const uint2*)p;
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float v[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, const float v[4]) {
  uint2 t; t.x = f32x2_to_bf16x2(v[0], v[1]);
  t.y = f32x2_to_bf16x2(v[2], v[3]);
  *(uint2*)p = t;
}

// ---- 8-wide (16 B of bf16) helpers ---------------------------------------------------------------
template <typename T> __device__ __forceinline__ void ld8(const T* p, float v[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float v[8]) { ld4<float>(p, v); ld4<float>(p + 4, v + 4); }
template <> __device__ __forceinline__ void ld8<bf16>(const bf16* p, float v[8]) {
  uint4 t = *(const uint4*)p; const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float v[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float v[8]) { st4<float>(p, v); st4<float>(p + 4, v + 4); }
template <> __device__ __forceinline__ void st8<bf16>(bf16* p, const float v[8]) {
  uint4 t;
  t.x = f32x2_to_bf16x2(v[0], v[1]); t.y = f32x2_to_bf16x2(v[2], v[3]);
  t.z = f32x2_to_bf16x2(v[4], v[5]); t.w = f32x2_to_bf16x2(v[6], v[7]);
  *(uint4*)p = t;
}

// 8 elements as they lie in memory (the conversion is deferred: loads of several chunks can be issued before any arithmetic)
__device__ __forceinline__ void bf8_to_f32(const uint4& t, float v[8]) {
  const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}
template <typename T> struct Raw8;
template <> struct Raw8<bf16> { uint4 t; __device__ __forceinline__ void load(const bf16* p) { t = *(const uint4*)p; } __device__ __forceinline__ void get(float v[8]) const { bf8_to_f32(t, v); } };
template <> struct Raw8<float> { float4 a, b; __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
  __device__ __forceinline__ void get(float v[8]) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; } };
#define DISPATCH_T(dtype, ...) do { if ((dtype) == AVEC_BF16) { typedef bf16 T; __VA_ARGS__; } else { typedef float T; __VA_ARGS__; } } while (0)


// =============================================================================================
// Two-pass column reduction.  Device-scope float atomics retire at only ~5 per ns on MI355X (whatever the address spread), so a
// launch of `nslots` blocks that each add `ncols` column sums atomically is bound by nslots * ncols atomics; in-kernel "last block
// reduces" schemes need device-scope fences, which on the 8-XCD part write back / invalidate whole L2s (measured 4x slower).
// With a workspace the blocks store their sums to partial[colblock][slot][NV][W] and a second tiny kernel (col_finalize) adds the
// slots: 16-column x 128-slot tiles, one atomic per column per tile (128x fewer).  ws.partial == nullptr: plain atomics.
// The workspace is registered per device with avec_set_reduce_workspace() and must not be shared by concurrent streams.
// =============================================================================================
struct ColWs { float* partial; };
ColWs avec_reduce_ws(size_t partial_floats, hipStream_t st);     // api.hip; the workspace registered for stream `st` (else the device default); {nullptr} when none / too small

__device__ __forceinline__ float* ws_slot(const ColWs& ws, unsigned colblock, unsigned slot, unsigned nslots, int ncols) {
  return ws.partial + ((size_t)colblock * nslots + slot) * ncols;
}
static constexpr int FIN_MAXNV = 20;
struct ColFin { const float* partial; float* dst[FIN_MAXNV]; int NV, W, C, nslots, dstride; };   // dstride: element stride of the destination columns (1 = dense)
// grid (ceil(NV*W/16), colblocks, ceil(nslots/128)); thread = (column cw = tid & 15, slot lane sl = tid >> 4)
__global__ __launch_bounds__(256) void col_finalize_kernel(ColFin f);     // norm.hip
int colsum_launch(int dtype, const void* x, long long ld, float* out, long long M, int N, bool use_ws, hipStream_t st);   // norm.hip
int col_finalize(const ColWs& ws, unsigned colblocks, unsigned nslots, int NV, int W, float* const* dst, int C, hipStream_t st, int dstride = 1);   // norm.hip

// =============================================================================================
// column-wise block reduction helper: block = 32 column groups (x4) x 8 row lanes
// =============================================================================================
template <int NV>
__device__ __forceinline__ void colreduce_atomic(float (&part)[NV][4], float* const (&dst)[NV], int col, int C, const ColWs& ws = ColWs{nullptr}) {
  __shared__ float red[8][32][4];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float* mine = ws.partial ? ws_slot(ws, blockIdx.x, blockIdx.y, gridDim.y, NV * 128) : nullptr;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    for (int e = 0; e < 4; ++e) red[ty][tx][e] = part[n][e];
    __syncthreads();
    if (ty == 0) {
      for (int e = 0; e < 4; ++e) {
        float s = 0.f;
        for (int y = 0; y < 8; ++y) s += red[y][tx][e];
        if (mine) mine[n * 128 + tx * 4 + e] = s;
        else if (col + e < C && dst[n]) atomicAdd(dst[n] + col + e, s);
      }
    }
    __syncthreads();
  }
}
// workspace for a col_grid launch with NV reduced quantities (finish with col_finalize(ws, grid.x, grid.y, NV, 128, dst, C, st))
static inline ColWs col_ws(dim3 grid, int NV, hipStream_t st) { return avec_reduce_ws((size_t)grid.x * grid.y * NV * 128, st); }
// ... only when the one-pass version would issue many atomics (the second pass costs a launch)
// AVEC_COLWS_MIN_ATOMICS: one-pass below this many atomics (default 16384).  Round 4 measured the step with 70 000 (the BatchNorm reductions of the conformer
// convolution modules lose their second-pass launch, ~4.7 us each inside a dependent chain): 20.00 vs 19.90 ms; with 270 000: 21.2 ms -- the contended fp32 atomics
// cost more than the launch they replace.
static inline long long col_ws_min_atomics() { return 16384; }
static inline ColWs col_ws_if(dim3 grid, int NV, int C, hipStream_t st) {
  const long long atomics = (long long)grid.y * NV * C;
  return atomics > col_ws_min_atomics() ? col_ws(grid, NV, st) : ColWs{nullptr};
}

static inline dim3 col_grid(long long M, int C) {
  unsigned gx = (unsigned)((C / 4 + 31) / 32);
  long long gy = (M + 63) / 64; long long cap = 2048 / gx; if (cap < 1) cap = 1; if (gy > cap) gy = cap; if (gy < 1) gy = 1;
  return dim3(gx, (unsigned)gy);
}

// =============================================================================================
// flat column mapping for C % 8 == 0, C <= 2048: the 256 threads of a block form R = 256/L rows of L = C/8 lanes, every lane owning
// 8 consecutive channels (one 16 B bf16 access); all lanes are busy for narrow C (C = 64: 32 rows x 8 lanes per pass).
// =============================================================================================
struct Col8 { int L, R, r, l; bool active; };
__device__ __forceinline__ Col8 col8_map(int C) {
  Col8 m; m.L = C >> 3; m.R = 256 / m.L; m.r = threadIdx.x / m.L; m.l = threadIdx.x - m.r * m.L; m.active = m.r < m.R; return m;
}
template <int NV>
__device__ __forceinline__ void colreduce8_atomic(float (&part)[NV][8], float* const (&dst)[NV], const Col8& m, const ColWs& ws = ColWs{nullptr}) {
  __shared__ float red8[NV * 8][256];
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int e = 0; e < 8; ++e) red8[n * 8 + e][threadIdx.x] = m.active ? part[n][e] : 0.f;
  __syncthreads();
  const int C = m.L * 8;
  float* mine = ws.partial ? ws_slot(ws, 0, blockIdx.x, gridDim.x, NV * C) : nullptr;
  for (int o = threadIdx.x; o < NV * 8 * m.L; o += 256) {
    const int q = o / m.L, l = o - q * m.L; float s = 0.f;
    for (int rr = 0; rr < m.R; ++rr) s += red8[q][rr * m.L + l];
    if (mine) mine[(q >> 3) * C + l * 8 + (q & 7)] = s;
    else {
      float* d = nullptr;
#pragma unroll
      for (int n = 0; n < NV; ++n) if ((q >> 3) == n) d = dst[n];
      if (d) atomicAdd(d + l * 8 + (q & 7), s);
    }
  }
}
static inline bool col8_ok(int C) { return C % 8 == 0 && C <= 2048; }
#ifndef AVEC_COL8_CAP
#define AVEC_COL8_CAP 1024
#endif
static inline long long col8_cap() { return AVEC_COL8_CAP; }      /* re-swept at the end of round 3 (tools/bench_bn.py): 1024 beats 2048 on every ResNet stage but the first (equal there) */
static inline unsigned col8_blocks(long long M, int C) { const int R = 256 / (C / 8); long long nb = (M + R - 1) / R; if (nb > col8_cap()) nb = col8_cap(); return (unsigned)nb; }
// grid size + workspace of a flat 8-wide launch (finish with col_finalize(ws, 1, nb, NV, C, dst, C, st)); without a workspace the
// block count is kept low (every block issues NV*C atomics)
static inline unsigned col8_cfg(long long M, int C, int NV, ColWs* ws, hipStream_t st) {
  unsigned nb = col8_blocks(M, C);
  if ((long long)nb * NV * C <= col_ws_min_atomics()) { *ws = ColWs{nullptr}; return nb; }      // small reductions: one pass, atomics
  *ws = avec_reduce_ws((size_t)nb * NV * C, st);
  if (!ws->partial && nb > 256) nb = 256;
  return nb;
}
