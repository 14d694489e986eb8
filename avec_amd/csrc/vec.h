// 4-wide vector access + column-wise block reduction helpers shared by the HBM-bound kernels.
#pragma once
#include "common.h"

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
  uint2 t; t.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  t.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  *(uint2*)p = t;
}

#define DISPATCH_T(dtype, ...) do { if ((dtype) == AVEC_BF16) { typedef bf16 T; __VA_ARGS__; } else { typedef float T; __VA_ARGS__; } } while (0)


// =============================================================================================
// column-wise block reduction helper: block = 32 column groups (x4) x 8 row lanes
// =============================================================================================
template <int NV>
__device__ __forceinline__ void colreduce_atomic(float (&part)[NV][4], float* const (&dst)[NV], int col, int C) {
  __shared__ float red[8][32][4];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    for (int e = 0; e < 4; ++e) red[ty][tx][e] = part[n][e];
    __syncthreads();
    if (ty == 0) {
      for (int e = 0; e < 4; ++e) {
        float s = 0.f;
        for (int y = 0; y < 8; ++y) s += red[y][tx][e];
        if (col + e < C && dst[n]) atomicAdd(dst[n] + col + e, s);
      }
    }
    __syncthreads();
  }
}

static inline dim3 col_grid(long long M, int C) {
  unsigned gx = (unsigned)((C / 4 + 31) / 32);
  long long gy = (M + 63) / 64; long long cap = 2048 / gx; if (cap < 1) cap = 1; if (gy > cap) gy = cap; if (gy < 1) gy = 1;
  return dim3(gx, (unsigned)gy);
}

