// Stand-alone forms of layers that the hot path only runs fused (nnet/layers.py MaxPool3d:839-915, Upsample:1013-1043): small HBM-bound kernels on channels-last tensors.
// On the hot path the max pool lives inside the stem convolution (stem3p.hip) and the up-sampling inside the patch-attention un-pooling (norm.hip); these are what the
// layer classes execute when a user builds them on their own.
#include "vec.h"
#include "avec_hip.h"

struct PoolG { long long Fr; int H, W, C, KH, KW, SH, SW, P0H, P0W, OH, OW; };

// out[fr][oh][ow][c] = max over the KH x KW window at (oh*SH - P0H, ow*SW - P0W); positions outside the image hold the constant 0 of the reference's ConstantPad3d
// ("same" = zero padding, then a valid max pool): idx = window slot kh*KW + kw of the winner (first one on ties), 255 when the zero padding wins
template <typename T>
__global__ __launch_bounds__(256) void maxpool_hw_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, unsigned char* __restrict__ idx, PoolG g) {
  const long long n4 = g.Fr * g.OH * g.OW * (g.C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int C4 = g.C >> 2; const int c = (int)(i % C4) * 4; long long r = i / C4;
    const int ow = (int)(r % g.OW); r /= g.OW; const int oh = (int)(r % g.OH); const long long fr = r / g.OH;
    float best[4]; unsigned bi[4]; bool any_pad = false;
    for (int e = 0; e < 4; ++e) { best[e] = -INFINITY; bi[e] = 255u; }
    for (int kh = 0; kh < g.KH; ++kh)
      for (int kw = 0; kw < g.KW; ++kw) {
        const int h = oh * g.SH - g.P0H + kh, w = ow * g.SW - g.P0W + kw;
        if (h < 0 || h >= g.H || w < 0 || w >= g.W) { any_pad = true; continue; }
        float v[4]; ld4<T>(x + ((fr * g.H + h) * g.W + w) * g.C + c, v);
        for (int e = 0; e < 4; ++e) if (v[e] > best[e]) { best[e] = v[e]; bi[e] = (unsigned)(kh * g.KW + kw); }
      }
    if (any_pad) for (int e = 0; e < 4; ++e) if (!(best[e] > 0.f)) { best[e] = 0.f; bi[e] = 255u; }
    st4<T>(out + i * 4, best);
    if (idx) *(uint32_t*)(idx + i * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
  }
}
// dx[fr][h][w][c] = sum of dy over the windows whose winner is (h, w)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_hw_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx, T* __restrict__ dx, PoolG g) {
  const long long n4 = g.Fr * g.H * g.W * (g.C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int C4 = g.C >> 2; const int c = (int)(i % C4) * 4; long long r = i / C4;
    const int w = (int)(r % g.W); r /= g.W; const int h = (int)(r % g.H); const long long fr = r / g.H;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < g.KH; ++kh) {
      const int t = h + g.P0H - kh; if (t < 0 || t % g.SH) continue; const int oh = t / g.SH; if (oh >= g.OH) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        const int u = w + g.P0W - kw; if (u < 0 || u % g.SW) continue; const int ow = u / g.SW; if (ow >= g.OW) continue;
        const long long o = ((fr * g.OH + oh) * g.OW + ow) * g.C + c;
        const uint32_t sel = *(const uint32_t*)(idx + o); float gq[4]; ld4<T>(dy + o, gq);
        for (int e = 0; e < 4; ++e) if (((sel >> (8 * e)) & 255u) == (unsigned)(kh * g.KW + kw)) acc[e] += gq[e];
      }
    }
    st4<T>(dx + i * 4, acc);
  }
}
static int pool_geom(PoolG& g, long long frames, int H, int W, int C, int KH, int KW, int SH, int SW, int P0H, int P0W, int P1H, int P1W) {
  AVEC_CHECK_ARG(frames > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && KH * KW < 255 && P0H >= 0 && P0W >= 0 && P1H >= 0 && P1W >= 0,
                 "maxpool_hw: bad arguments (C %% 4 == 0, window < 255 slots)");
  g.Fr = frames; g.H = H; g.W = W; g.C = C; g.KH = KH; g.KW = KW; g.SH = SH; g.SW = SW; g.P0H = P0H; g.P0W = P0W;
  g.OH = (H + P0H + P1H - KH) / SH + 1; g.OW = (W + P0W + P1W - KW) / SW + 1;
  AVEC_CHECK_ARG(g.OH > 0 && g.OW > 0, "maxpool_hw: window larger than the padded image");
  return 0;
}
extern "C" int avec_maxpool_hw_fwd(int dtype, const void* x, void* out, unsigned char* idx, long long frames, int H, int W, int C, int KH, int KW, int SH, int SW,
                                   int pad0_h, int pad0_w, int pad1_h, int pad1_w, hipStream_t st) {
  AVEC_CHECK_ARG(x && out, "maxpool_hw_fwd: null pointer");
  PoolG g; if (int r = pool_geom(g, frames, H, W, C, KH, KW, SH, SW, pad0_h, pad0_w, pad1_h, pad1_w)) return r;
  long long n4 = frames * g.OH * g.OW * (C / 4); long long nb = (n4 + 255) / 256; if (nb > 8192) nb = 8192;
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_hw_fwd_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, (T*)out, idx, g));
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_maxpool_hw_bwd(int dtype, const void* dy, const unsigned char* idx, void* dx, long long frames, int H, int W, int C, int KH, int KW, int SH, int SW,
                                   int pad0_h, int pad0_w, int pad1_h, int pad1_w, hipStream_t st) {
  AVEC_CHECK_ARG(dy && idx && dx, "maxpool_hw_bwd: null pointer");
  PoolG g; if (int r = pool_geom(g, frames, H, W, C, KH, KW, SH, SW, pad0_h, pad0_w, pad1_h, pad1_w)) return r;
  long long n4 = frames * H * W * (C / 4); long long nb = (n4 + 255) / 256; if (nb > 8192) nb = 8192;
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_hw_bwd_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)dy, idx, (T*)dx, g));
  AVEC_LAUNCH_CHECK(); return 0;
}

// nearest-neighbour up-sampling of rows by an integer factor: y[b][t][:] = x[b][t / P][:]  (backward != 0: x <- sum of the P rows of y)
template <typename T>
__global__ __launch_bounds__(256) void upsample_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, long long B, int T_, int D, int P, int backward) {
  const int D4 = D >> 2;
  const long long n4 = B * (long long)T_ * (backward ? 1 : P) * D4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % D4) * 4; const long long row = i / D4;
    float v[4];
    if (!backward) { const long long b = row / ((long long)T_ * P); const long long t = row - b * T_ * P; ld4<T>(src + (b * T_ + t / P) * D + c, v); }
    else {
      v[0] = v[1] = v[2] = v[3] = 0.f;
      for (int p = 0; p < P; ++p) { float u[4]; ld4<T>(src + (row * P + p) * D + c, u); for (int e = 0; e < 4; ++e) v[e] += u[e]; }
    }
    st4<T>(dst + row * D + c, v);
  }
}
extern "C" int avec_upsample_rows(int dtype, const void* src, void* dst, long long B, int T_, int D, int P, int backward, hipStream_t st) {
  AVEC_CHECK_ARG(src && dst && B > 0 && T_ > 0 && D > 0 && D % 4 == 0 && P > 0, "upsample_rows: bad arguments (D %% 4 == 0)");
  long long n4 = B * (long long)T_ * (backward ? 1 : P) * (D / 4); long long nb = (n4 + 255) / 256; if (nb > 8192) nb = 8192;
  DISPATCH_T(dtype, hipLaunchKernelGGL(upsample_rows_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)src, (T*)dst, B, T_, D, P, backward));
  AVEC_LAUNCH_CHECK(); return 0;
}

// lengths after a strided layer (nnet/preprocessing.py:77, nnet/modules.py:127-128, nnet/networks.py:298,302): out = (in - sub) / div + add on int64, ONE launch
// instead of the three ATen element-wise launches (sub, floor-div, add) that each such point of the forward pass used to put on the dependent chain
__global__ void len_affine_kernel(const long long* __restrict__ in, long long* __restrict__ out, int n, long long sub, long long div, long long add) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) { const long long v = in[i] - sub; const long long q = v / div; out[i] = ((v % div != 0) && ((v < 0) != (div < 0)) ? q - 1 : q) + add; }      // floor division
}
extern "C" int avec_len_affine(const long long* in, long long* out, int n, long long sub, long long div, long long add, hipStream_t st) {
  AVEC_CHECK_ARG(in && out && n > 0 && div != 0, "len_affine: bad arguments");
  hipLaunchKernelGGL(len_affine_kernel, dim3((n + 63) / 64), dim3(64), 0, st, in, out, n, sub, div, add);
  AVEC_LAUNCH_CHECK(); return 0;
}
