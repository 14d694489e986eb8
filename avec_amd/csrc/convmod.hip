// Conformer convolution-module middle section (nnet/modules.py:375-376 + layers.Conv1d :82-198):
//   g = GLU(u) = u[:, :C] * sigmoid(u[:, C:]);   c[b][t'][ch] = bias[ch] + sum_k w[k][ch] * g[b][t'*s + k - (K-1)/2][ch]
// depthwise, zero "same" padding ((K-1)//2, K//2), stride s; channels-last throughout.  The depthwise
// weight lives tap-major ([K][C]) so channel vectors are contiguous.  BatchNorm batch statistics of
// the conv output are accumulated in the same pass.
#include "vec.h"
#include "avec_hip.h"

static constexpr int KMAX = 16;

template <typename T>
__device__ __forceinline__ void glu4(const T* u, long long row, int C, int col, float g[4]) {
  float a[4], b[4]; ld4<T>(u + row * 2 * C + col, a); ld4<T>(u + row * 2 * C + C + col, b);
  for (int e = 0; e < 4; ++e) g[e] = a[e] * sigmoidf_(b[e]);
}

template <typename T>
__global__ __launch_bounds__(256) void glu_dwconv_fwd_kernel(const T* __restrict__ u, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                                                             float* stats, int B, int Tn, int C, int K, int stride, int To, ColWs ws) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  const int padl = (K - 1) / 2; const long long M = (long long)B * To;
  float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (col < C) {
    float bb[4] = {0.f, 0.f, 0.f, 0.f}; if (bias) ld4<float>(bias + col, bb);
    for (long long row = (long long)blockIdx.y * 8 + ty; row < M; row += (long long)gridDim.y * 8) {
      const int to = (int)(row % To); const long long b = row / To;
      float acc[4] = {bb[0], bb[1], bb[2], bb[3]};
      for (int k = 0; k < K; ++k) {
        const int t = to * stride + k - padl;
        if (t < 0 || t >= Tn) continue;
        float g[4], ww[4]; glu4<T>(u, b * Tn + t, C, col, g); ld4<float>(w + (long long)k * C + col, ww);
        for (int e = 0; e < 4; ++e) acc[e] += ww[e] * g[e];
      }
      st4<T>(out + row * C + col, acc);
      for (int e = 0; e < 4; ++e) { part[0][e] += acc[e]; part[1][e] += acc[e] * acc[e]; }
    }
  }
  float* const dst[2] = {stats, stats ? stats + C : nullptr};
  if (stats) colreduce_atomic<2>(part, dst, col, C, ws);
}

// du (act [B*T][2C]) from dc (act [B*To][C])
template <typename T>
__global__ __launch_bounds__(256) void dwconv_glu_bwd_input_kernel(const T* __restrict__ dc, const T* __restrict__ u, const float* __restrict__ w, T* __restrict__ du,
                                                                   int B, int Tn, int C, int K, int stride, int To) {
  const int padl = (K - 1) / 2; const long long n4 = (long long)B * Tn * (C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int col = (int)(i % (C / 4)) * 4; const long long row = i / (C / 4); const int t = (int)(row % Tn); const long long b = row / Tn;
    float dg[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
      const int tt = t + padl - k; if (tt < 0) break;
      const int to = tt / stride; if (to * stride != tt || to >= To) continue;
      float d[4], ww[4]; ld4<T>(dc + (b * To + to) * C + col, d); ld4<float>(w + (long long)k * C + col, ww);
      for (int e = 0; e < 4; ++e) dg[e] += ww[e] * d[e];
    }
    float a[4], bq[4], o1[4], o2[4]; ld4<T>(u + row * 2 * C + col, a); ld4<T>(u + row * 2 * C + C + col, bq);
    for (int e = 0; e < 4; ++e) { const float s = sigmoidf_(bq[e]); o1[e] = dg[e] * s; o2[e] = dg[e] * a[e] * s * (1.f - s); }
    st4<T>(du + row * 2 * C + col, o1); st4<T>(du + row * 2 * C + C + col, o2);
  }
}

// dw[k][c] += sum dc * g(shifted);  dbias[c] += sum dc
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ dc, const T* __restrict__ u, float* dw, float* dbias,
                                                                int B, int Tn, int C, int K, int stride, int To, ColWs ws) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  const int padl = (K - 1) / 2; const long long M = (long long)B * To;
  float part[KMAX + 1][4];
#pragma unroll
  for (int k = 0; k <= KMAX; ++k) for (int e = 0; e < 4; ++e) part[k][e] = 0.f;
  if (col < C) {
    for (long long row = (long long)blockIdx.y * 8 + ty; row < M; row += (long long)gridDim.y * 8) {
      const int to = (int)(row % To); const long long b = row / To;
      float d[4]; ld4<T>(dc + row * C + col, d);
      for (int e = 0; e < 4; ++e) part[KMAX][e] += d[e];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k >= K) break;
        const int t = to * stride + k - padl;
        if (t < 0 || t >= Tn) continue;
        float g[4]; glu4<T>(u, b * Tn + t, C, col, g);
        for (int e = 0; e < 4; ++e) part[k][e] += d[e] * g[e];
      }
    }
  }
  float* dst[KMAX + 1];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) dst[k] = (k < K) ? dw + (long long)k * C : nullptr;
  dst[KMAX] = dbias;
  float* const (&cdst)[KMAX + 1] = dst;
  colreduce_atomic<KMAX + 1>(part, cdst, col, C, ws);
}

extern "C" int avec_glu_dwconv_fwd(int dtype, const void* u, const float* w, const float* bias, void* out, float* stats,
                                   int B, int T_, int C, int K, int stride, hipStream_t st) {
  AVEC_CHECK_ARG(u && w && out && B > 0 && T_ > 0 && C > 0 && C % 4 == 0 && K > 0 && K <= KMAX && stride > 0, "glu_dwconv_fwd: bad arguments (C=%d K=%d)", C, K);
  const int To = (T_ - 1) / stride + 1;
  dim3 grid = col_grid((long long)B * To, C); ColWs ws = stats ? col_ws_if(grid, 2, C) : ColWs{nullptr};
  DISPATCH_T(dtype, hipLaunchKernelGGL(glu_dwconv_fwd_kernel<T>, grid, dim3(256), 0, st, (const T*)u, w, bias, (T*)out, stats, B, T_, C, K, stride, To, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {stats, stats + C}; return col_finalize(ws, grid.x, grid.y, 2, 128, dst, C, st); }
  return 0;
}
extern "C" int avec_dwconv_glu_bwd(int dtype, const void* dc, const void* u, const float* w, void* du, float* dw, float* dbias,
                                   int B, int T_, int C, int K, int stride, hipStream_t st) {
  AVEC_CHECK_ARG(dc && u && w && du && dw && B > 0 && T_ > 0 && C > 0 && C % 4 == 0 && K > 0 && K <= KMAX && stride > 0, "dwconv_glu_bwd: bad arguments");
  const int To = (T_ - 1) / stride + 1;
  long long n4 = (long long)B * T_ * (C / 4); long long nb = (n4 + 255) / 256; if (nb > 4096) nb = 4096;
  dim3 grid = col_grid((long long)B * To, C); ColWs ws = col_ws_if(grid, KMAX + 1, C);
  DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv_glu_bwd_input_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)dc, (const T*)u, w, (T*)du, B, T_, C, K, stride, To);
             hipLaunchKernelGGL(dwconv_bwd_weight_kernel<T>, grid, dim3(256), 0, st, (const T*)dc, (const T*)u, dw, dbias, B, T_, C, K, stride, To, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) {
    float* dst[KMAX + 1];
    for (int k = 0; k < KMAX; ++k) dst[k] = (k < K) ? dw + (long long)k * C : nullptr;
    dst[KMAX] = dbias;
    return col_finalize(ws, grid.x, grid.y, KMAX + 1, 128, dst, C, st);
  }
  return 0;
}
