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

// Workgroup = (128-channel slab, batch b, chunk of DW_TT output frames).  GLU(u) of the frames the chunk touches is computed ONCE into
// LDS (fp32, zero outside the sequence = the "same" padding), then every output frame reads its K taps from LDS.
static constexpr int DW_TT = 32;
template <typename T>
__device__ __forceinline__ void stage_glu(float* gs, const T* u, long long b, int Tn, int C, int col, int t_base, int nrows) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // six rows per pass: all twelve loads are in flight before the first sigmoid (one load latency per pass instead of one per row)
  for (int r0 = ty; r0 < nrows; r0 += 48) {
    float a[6][4], bq[6][4];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int r = r0 + 8 * q, t = t_base + r;
      const bool ok = r < nrows && col < C && t >= 0 && t < Tn;
      const long long row = ok ? b * Tn + t : b * Tn;          // clamped: the loads stay unconditional
      const int cc = col < C ? col : 0;
      ld4<T>(u + row * 2 * C + cc, a[q]); ld4<T>(u + row * 2 * C + C + cc, bq[q]);
      if (!ok) { a[q][0] = a[q][1] = a[q][2] = a[q][3] = 0.f; }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int r = r0 + 8 * q;
      if (r < nrows) *(float4*)(gs + r * 128 + tx * 4) = make_float4(a[q][0] * sigmoidf_(bq[q][0]), a[q][1] * sigmoidf_(bq[q][1]), a[q][2] * sigmoidf_(bq[q][2]), a[q][3] * sigmoidf_(bq[q][3]));
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void glu_dwconv_fwd_kernel(const T* __restrict__ u, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                                                             float* stats, int B, int Tn, int C, int K, int stride, int To, int padl, int nchunks, ColWs ws) {
  extern __shared__ __attribute__((aligned(16))) float gs[];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  const int b = blockIdx.y / nchunks, to0 = (blockIdx.y - b * nchunks) * DW_TT; const int nto = min(DW_TT, To - to0);
  // the taps of this thread's four channels in registers, requested in front of the staging pass (one memory round trip for both): the tap loop used to fetch
  // w[k][col] from memory once per (row, tap) -- 60 dependent L1 / L2 round trips per thread, most of the kernel's 10 us (tools/block_trace.py)
  const bool wreg = K <= KMAX;
  float ww[KMAX][4], bb[4] = {0.f, 0.f, 0.f, 0.f};
  if (wreg && col < C) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { if (k < K) ld4<float>(w + (long long)k * C + col, ww[k]); else ww[k][0] = ww[k][1] = ww[k][2] = ww[k][3] = 0.f; }
  }
  if (col < C && bias) ld4<float>(bias + col, bb);
  stage_glu<T>(gs, u, b, Tn, C, col, to0 * stride - padl, (nto - 1) * stride + K);
  __syncthreads();
  float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (col < C && wreg) {
    for (int r = ty; r < nto; r += 8) {
      float acc[4] = {bb[0], bb[1], bb[2], bb[3]};
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const float4 g = *(const float4*)(gs + (r * stride + k) * 128 + tx * 4);
          acc[0] += ww[k][0] * g.x; acc[1] += ww[k][1] * g.y; acc[2] += ww[k][2] * g.z; acc[3] += ww[k][3] * g.w;
        }
      }
      st4<T>(out + ((long long)b * To + to0 + r) * C + col, acc);
      for (int e = 0; e < 4; ++e) { part[0][e] += acc[e]; part[1][e] += acc[e] * acc[e]; }
    }
  } else
  if (col < C) {
    for (int r = ty; r < nto; r += 8) {
      float acc[4] = {bb[0], bb[1], bb[2], bb[3]};
      for (int k = 0; k < K; ++k) {
        const float4 g = *(const float4*)(gs + (r * stride + k) * 128 + tx * 4); float ww[4]; ld4<float>(w + (long long)k * C + col, ww);
        acc[0] += ww[0] * g.x; acc[1] += ww[1] * g.y; acc[2] += ww[2] * g.z; acc[3] += ww[3] * g.w;
      }
      st4<T>(out + ((long long)b * To + to0 + r) * C + col, acc);
      for (int e = 0; e < 4; ++e) { part[0][e] += acc[e]; part[1][e] += acc[e] * acc[e]; }
    }
  }
  float* const dst[2] = {stats, stats ? stats + C : nullptr};
  if (stats) colreduce_atomic<2>(part, dst, col, C, ws);
}

// du (act [B*T][2C]) from dc (act [B*To][C])
template <typename T>
__global__ __launch_bounds__(256) void dwconv_glu_bwd_input_kernel(const T* __restrict__ dc, const T* __restrict__ u, const float* __restrict__ w, T* __restrict__ du,
                                                                   int B, int Tn, int C, int K, int stride, int To, int padl) {
  const long long n4 = (long long)B * Tn * (C / 4);
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

// dw[k][c] += sum dc * g(shifted);  dbias[c] += sum dc        (same tiling as the forward: GLU staged once per chunk in LDS)
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ dc, const T* __restrict__ u, float* dw, float* dbias,
                                                                int B, int Tn, int C, int K, int stride, int To, int padl, int nchunks, ColWs ws) {
  extern __shared__ __attribute__((aligned(16))) float gs[];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  const int b = blockIdx.y / nchunks, to0 = (blockIdx.y - b * nchunks) * DW_TT; const int nto = min(DW_TT, To - to0);
  stage_glu<T>(gs, u, b, Tn, C, col, to0 * stride - padl, (nto - 1) * stride + K);
  __syncthreads();
  float part[KMAX + 1][4];
#pragma unroll
  for (int k = 0; k <= KMAX; ++k) for (int e = 0; e < 4; ++e) part[k][e] = 0.f;
  if (col < C) {
    float dq[DW_TT / 8][4];                      // this thread's rows of dc, all requested before the first one is used
#pragma unroll
    for (int q = 0; q < DW_TT / 8; ++q) {
      const int r = ty + 8 * q;
      ld4<T>(dc + ((long long)b * To + to0 + (r < nto ? r : 0)) * C + col, dq[q]);
      if (r >= nto) { dq[q][0] = dq[q][1] = dq[q][2] = dq[q][3] = 0.f; }
    }
#pragma unroll
    for (int q = 0; q < DW_TT / 8; ++q) {
      const int r = ty + 8 * q;
      if (r >= nto) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) part[KMAX][e] += dq[q][e];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const float4 g = *(const float4*)(gs + (r * stride + k) * 128 + tx * 4);
          part[k][0] += dq[q][0] * g.x; part[k][1] += dq[q][1] * g.y; part[k][2] += dq[q][2] * g.z; part[k][3] += dq[q][3] * g.w;
        }
      }
    }
  }
  // reduce the 17 partial vectors over the 8 row groups of the workgroup in ONE pass (the generic colreduce_atomic walks them one at a time: 34 barriers):
  // the two row groups of a wave meet by a shuffle, the four waves through the (now dead) GLU image
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k <= KMAX; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[k][e] += __shfl_xor(part[k][e], 32, 64);
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k <= KMAX; ++k) *(float4*)(gs + ((wv * (KMAX + 1) + k) * 32 + tx) * 4) = make_float4(part[k][0], part[k][1], part[k][2], part[k][3]);
  }
  __syncthreads();
  float* mine = ws.partial ? ws_slot(ws, blockIdx.x, blockIdx.y, gridDim.y, (KMAX + 1) * 128) : nullptr;
  for (int i = threadIdx.x; i < (KMAX + 1) * 128; i += 256) {
    const int k = i >> 7, c = i & 127;
    const float v = gs[(0 * (KMAX + 1) + k) * 128 + c] + gs[(1 * (KMAX + 1) + k) * 128 + c] + gs[(2 * (KMAX + 1) + k) * 128 + c] + gs[(3 * (KMAX + 1) + k) * 128 + c];
    if (mine) mine[i] = v;
    else {
      const int cc = blockIdx.x * 128 + c;
      float* d = k == KMAX ? dbias : (k < K ? dw + (long long)k * C : nullptr);
      if (d && cc < C) atomicAdd(d + cc, v);
    }
  }
}

// Both halves of the backward pass in ONE launch (stride 1: every conformer block but the two stage boundaries): the workgroup of the weight-gradient kernel also
// stages the dc rows its 32 frames can see (K - 1 rows of halo, fp32, zero outside the sequence) and produces du for those frames -- one launch, one pass over u and dc
// instead of two (the input-gradient kernel re-read K dc rows per element from L2).  LDS: (31 + K) x 128 floats twice.
// BatchNorm(+Swish) backward folded into the staging pass (round 6): with bnb.c set, `dc` is the gradient of the BatchNorm + Swish OUTPUT and the rows staged in LDS are
//   dc = gamma rstd (d - mean(d) - xhat mean(d xhat)),  d = da * swish'(scale c + shift)          (avec_bn_bwd_apply's arithmetic, act = Swish)
// from the reduced sums `dstats` -- the apply pass and its tensor disappear from the conformer block's chain; workgroup (0, 0) adds dgamma / dbeta.
struct DwBnb { const void* c; const float* ss; const float* gamma; const float* dstats; float inv_n; float* dgamma; float* dbeta; };
template <typename T>
__global__ __launch_bounds__(256) void dwconv_glu_bwd_fused_kernel(const T* __restrict__ dc, const T* __restrict__ u, const float* __restrict__ w, T* __restrict__ du, float* dw, float* dbias,
                                                                   int B, int Tn, int C, int K, int padl, int nchunks, ColWs ws, DwBnb bnb = DwBnb{nullptr}) {
  extern __shared__ __attribute__((aligned(16))) float gs[];
  const int NR = DW_TT - 1 + K;
  float* const ds = gs + NR * 128;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  const int b = blockIdx.y / nchunks, t0 = (blockIdx.y - b * nchunks) * DW_TT; const int nto = min(DW_TT, Tn - t0);
  stage_glu<T>(gs, u, b, Tn, C, col, t0 - padl, nto - 1 + K);
  {                                                            // dc rows of frames t0 + padl - (K - 1) + j, j < nto - 1 + K
    const int f0 = t0 + padl - (K - 1), nrows = nto - 1 + K, cc = col < C ? col : 0;
    float bA[4], bM1[4], bM2[4], bMu[4], bRs[4], bSc[4], bSh[4];
    if (bnb.c) {
      float g4[4], s1[4], s2[4];
      ld4<float>(bnb.ss + cc, bSc); ld4<float>(bnb.ss + C + cc, bSh); ld4<float>(bnb.ss + 2 * C + cc, bMu); ld4<float>(bnb.ss + 3 * C + cc, bRs);
      ld4<float>(bnb.gamma + cc, g4); ld4<float>(bnb.dstats + cc, s1); ld4<float>(bnb.dstats + C + cc, s2);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bA[e] = g4[e] * bRs[e]; bM1[e] = s1[e] * bnb.inv_n; bM2[e] = s2[e] * bnb.inv_n; }
      if (blockIdx.y == 0 && ty == 0 && col < C && bnb.dgamma) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { atomicAdd(bnb.dgamma + col + e, s2[e]); atomicAdd(bnb.dbeta + col + e, s1[e]); }
      }
    }
    for (int r0 = ty; r0 < nrows; r0 += 48) {
      float d[6][4];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int r = r0 + 8 * q, f = f0 + r; const bool ok = r < nrows && col < C && f >= 0 && f < Tn;
        ld4<T>(dc + ((long long)b * Tn + (ok ? f : 0)) * C + cc, d[q]);
        if (bnb.c) {
          float cv[4]; ld4<T>((const T*)bnb.c + ((long long)b * Tn + (ok ? f : 0)) * C + cc, cv);
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float dd = d[q][e] * dswishf_(cv[e] * bSc[e] + bSh[e]); d[q][e] = bA[e] * (dd - bM1[e] - (cv[e] - bMu[e]) * bRs[e] * bM2[e]); }
        }
        if (!ok) d[q][0] = d[q][1] = d[q][2] = d[q][3] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) { const int r = r0 + 8 * q; if (r < nrows) *(float4*)(ds + r * 128 + tx * 4) = make_float4(d[q][0], d[q][1], d[q][2], d[q][3]); }
    }
  }
  __syncthreads();
  float part[KMAX + 1][4];
#pragma unroll
  for (int k = 0; k <= KMAX; ++k) for (int e = 0; e < 4; ++e) part[k][e] = 0.f;
  if (col < C) {
    float ww[KMAX][4];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { if (k < K) ld4<float>(w + (long long)k * C + col, ww[k]); else ww[k][0] = ww[k][1] = ww[k][2] = ww[k][3] = 0.f; }
    const int own = K - 1 - padl;                              // ds row of this chunk's frame 0
#pragma unroll
    for (int q = 0; q < DW_TT / 8; ++q) {
      const int r = ty + 8 * q;
      if (r >= nto) break;
      const long long row = (long long)b * Tn + t0 + r;
      float a[4], bq[4]; ld4<T>(u + row * 2 * C + col, a); ld4<T>(u + row * 2 * C + C + col, bq);      // (requested before the tap loop: L2 hits, the staging pass has just read them)
      const float4 dq = *(const float4*)(ds + (own + r) * 128 + tx * 4);
      part[KMAX][0] += dq.x; part[KMAX][1] += dq.y; part[KMAX][2] += dq.z; part[KMAX][3] += dq.w;
      float dg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const float4 g = *(const float4*)(gs + (r + k) * 128 + tx * 4);
          part[k][0] += dq.x * g.x; part[k][1] += dq.y * g.y; part[k][2] += dq.z * g.z; part[k][3] += dq.w * g.w;
          const float4 dd = *(const float4*)(ds + (r + K - 1 - k) * 128 + tx * 4);
          dg[0] += ww[k][0] * dd.x; dg[1] += ww[k][1] * dd.y; dg[2] += ww[k][2] * dd.z; dg[3] += ww[k][3] * dd.w;
        }
      }
      float o1[4], o2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float sg = sigmoidf_(bq[e]); o1[e] = dg[e] * sg; o2[e] = dg[e] * a[e] * sg * (1.f - sg); }
      st4<T>(du + row * 2 * C + col, o1); st4<T>(du + row * 2 * C + C + col, o2);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k <= KMAX; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[k][e] += __shfl_xor(part[k][e], 32, 64);
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k <= KMAX; ++k) *(float4*)(gs + ((wv * (KMAX + 1) + k) * 32 + tx) * 4) = make_float4(part[k][0], part[k][1], part[k][2], part[k][3]);
  }
  __syncthreads();
  float* mine = ws.partial ? ws_slot(ws, blockIdx.x, blockIdx.y, gridDim.y, (KMAX + 1) * 128) : nullptr;
  for (int i = threadIdx.x; i < (KMAX + 1) * 128; i += 256) {
    const int k = i >> 7, c = i & 127;
    const float v = gs[(0 * (KMAX + 1) + k) * 128 + c] + gs[(1 * (KMAX + 1) + k) * 128 + c] + gs[(2 * (KMAX + 1) + k) * 128 + c] + gs[(3 * (KMAX + 1) + k) * 128 + c];
    if (mine) mine[i] = v;
    else {
      const int cc = blockIdx.x * 128 + c;
      float* d = k == KMAX ? dbias : (k < K ? dw + (long long)k * C : nullptr);
      if (d && cc < C) atomicAdd(d + cc, v);
    }
  }
}

extern "C" int avec_glu_dwconv_fwd(int dtype, const void* u, const float* w, const float* bias, void* out, float* stats,
                                   int B, int T_, int C, int K, int stride, int pad_left, hipStream_t st) {
  AVEC_CHECK_ARG(u && w && out && B > 0 && T_ > 0 && C > 0 && C % 4 == 0 && K > 0 && K <= KMAX && stride > 0 && pad_left >= 0 && pad_left < K, "glu_dwconv_fwd: bad arguments (C=%d K=%d pad_left=%d)", C, K, pad_left);
  const int To = (T_ - 1) / stride + 1;
  const int nchunks = (To + DW_TT - 1) / DW_TT;
  dim3 grid((unsigned)((C / 4 + 31) / 32), (unsigned)(B * nchunks)); ColWs ws = stats ? col_ws_if(grid, 2, C, st) : ColWs{nullptr};
  const size_t lds = (size_t)((DW_TT - 1) * stride + K) * 128 * sizeof(float);
  AVEC_CHECK_ARG(lds <= 64 * 1024, "glu_dwconv_fwd: stride %d too large", stride);
  DISPATCH_T(dtype, hipLaunchKernelGGL(glu_dwconv_fwd_kernel<T>, grid, dim3(256), lds, st, (const T*)u, w, bias, (T*)out, stats, B, T_, C, K, stride, To, pad_left, nchunks, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {stats, stats + C}; return col_finalize(ws, grid.x, grid.y, 2, 128, dst, C, st); }
  return 0;
}
// BatchNorm finalize (norm.hip bn_finalize_kernel, training mode) reading the two-pass column-reduction partials of glu_dwconv_fwd_kernel directly: the second pass
// (col_finalize) and the finalize were two ~4.8 us launches back to back in every conformer block's dependent chain (round 6).  block = 16 channels x 16 slot lanes;
// partial[(colblock * nslots + slot) * 256 + n * 128 + w], channel c = 128 colblock + w, n = 0 sum / 1 sum of squares.
__global__ __launch_bounds__(256) void bn_finalize_ws_kernel(const float* __restrict__ partial, int nslots, float count, const float* gamma, const float* beta, float* rmean, float* rvar,
                                                             long long* nbt, float momentum, float eps, float* ss, int C) {
  __shared__ float red[2][16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool lead = rl == 0 && c < C;
  float gm = 0.f, bt = 0.f, rm = 0.f, rv = 0.f;
  if (lead) { gm = gamma[c]; bt = beta[c]; if (rmean) { rm = rmean[c]; rv = rvar[c]; } }
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float* p = partial + (size_t)(c >> 7) * nslots * 256 + (c & 127);
    for (int r = rl; r < nslots; r += 16) { s1 += p[(size_t)r * 256]; s2 += p[(size_t)r * 256 + 128]; }
  }
  red[0][rl][cl] = s1; red[1][rl][cl] = s2;
  __syncthreads();
  if (!lead) return;
  s1 = 0.f; s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) { s1 += red[0][k][cl]; s2 += red[1][k][cl]; }
  const float n = count;
  const float mean = s1 / n, var = fmaxf(s2 / n - mean * mean, 0.f);
  if (rmean && isfinite(mean) && isfinite(var)) {
    rmean[c] = (1.f - momentum) * rm + momentum * mean;
    rvar[c] = (1.f - momentum) * rv + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    if (c == 0 && nbt) *nbt += 1;
  }
  const float rs = rsqrtf(var + eps);
  ss[c] = gm * rs; ss[C + c] = bt - mean * gm * rs; ss[2 * C + c] = mean; ss[3 * C + c] = rs;
}

// avec_glu_dwconv_fwd + avec_bn_finalize (training mode, local batch statistics over count = B * To rows) as TWO launches instead of three; `stats` ([2C], zeroed) is used
// only when the reduction runs on atomics (no workspace registered / few partials)
extern "C" int avec_glu_dwconv_fwd_bn(int dtype, const void* u, const float* w, const float* bias, void* out, float* stats, int B, int T_, int C, int K, int stride, int pad_left,
                                      const float* gamma, const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                      float* ss, hipStream_t st) {
  AVEC_CHECK_ARG(u && w && out && stats && gamma && beta && ss && B > 0 && T_ > 0 && C > 0 && C % 4 == 0 && K > 0 && K <= KMAX && stride > 0 && pad_left >= 0 && pad_left < K,
                 "glu_dwconv_fwd_bn: bad arguments (C=%d K=%d pad_left=%d)", C, K, pad_left);
  const int To = (T_ - 1) / stride + 1;
  const int nchunks = (To + DW_TT - 1) / DW_TT;
  dim3 grid((unsigned)((C / 4 + 31) / 32), (unsigned)(B * nchunks)); ColWs ws = col_ws_if(grid, 2, C, st);
  const size_t lds = (size_t)((DW_TT - 1) * stride + K) * 128 * sizeof(float);
  AVEC_CHECK_ARG(lds <= 64 * 1024, "glu_dwconv_fwd_bn: stride %d too large", stride);
  DISPATCH_T(dtype, hipLaunchKernelGGL(glu_dwconv_fwd_kernel<T>, grid, dim3(256), lds, st, (const T*)u, w, bias, (T*)out, stats, B, T_, C, K, stride, To, pad_left, nchunks, ws));
  AVEC_LAUNCH_CHECK();
  const float count = (float)((long long)B * To);
  if (ws.partial) {
    hipLaunchKernelGGL(bn_finalize_ws_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const float*)ws.partial, (int)grid.y, count, gamma, beta, running_mean, running_var,
                       num_batches_tracked, momentum, eps, ss, C);
    AVEC_LAUNCH_CHECK();
    return 0;
  }
  return avec_bn_finalize(stats, 1, nullptr, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, ss, C, 1, st);
}
// avec_bn_bwd_apply (act = Swish, local batch statistics over B * T rows, dstats = the reduced (sum d, sum d xhat) of avec_bn_bwd_reduce) + avec_dwconv_glu_bwd in ONE launch
// (stride 1): da = gradient of the BatchNorm + Swish output, c = the BatchNorm input (= the depthwise convolution's output); dgamma / dbeta += dstats.
extern "C" int avec_dwconv_glu_bwd_bn(int dtype, const void* da, const void* c, const float* ss, const float* gamma, const float* dstats, float count, const void* u, const float* w,
                                      void* du, float* dw, float* dbias, float* dgamma, float* dbeta, int B, int T_, int C, int K, int pad_left, hipStream_t st) {
  AVEC_CHECK_ARG(da && c && ss && gamma && dstats && count > 0.f && u && w && du && dw && B > 0 && T_ > 0 && C > 0 && C % 4 == 0 && K > 0 && K <= KMAX && pad_left >= 0 && pad_left < K,
                 "dwconv_glu_bwd_bn: bad arguments");
  const int nchunks = (T_ + DW_TT - 1) / DW_TT;
  dim3 grid((unsigned)((C / 4 + 31) / 32), (unsigned)(B * nchunks)); ColWs ws = col_ws_if(grid, KMAX + 1, C, st);
  size_t l2 = (size_t)2 * (DW_TT - 1 + K) * 128 * sizeof(float);
  if (l2 < (size_t)4 * (KMAX + 1) * 128 * sizeof(float)) l2 = (size_t)4 * (KMAX + 1) * 128 * sizeof(float);
  AVEC_CHECK_ARG(l2 <= 64 * 1024, "dwconv_glu_bwd_bn: kernel size %d too large", K);
  DwBnb bnb{c, ss, gamma, dstats, 1.f / count, dgamma, dbeta};
  DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv_glu_bwd_fused_kernel<T>, grid, dim3(256), l2, st, (const T*)da, (const T*)u, w, (T*)du, dw, dbias, B, T_, C, K, pad_left, nchunks, ws, bnb));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) {
    float* dst[KMAX + 1];
    for (int k = 0; k < KMAX; ++k) dst[k] = (k < K) ? dw + (long long)k * C : nullptr;
    dst[KMAX] = dbias;
    return col_finalize(ws, grid.x, grid.y, KMAX + 1, 128, dst, C, st);
  }
  return 0;
}
extern "C" int avec_dwconv_glu_bwd(int dtype, const void* dc, const void* u, const float* w, void* du, float* dw, float* dbias,
                                   int B, int T_, int C, int K, int stride, int pad_left, hipStream_t st) {
  AVEC_CHECK_ARG(dc && u && w && du && dw && B > 0 && T_ > 0 && C > 0 && C % 4 == 0 && K > 0 && K <= KMAX && stride > 0 && pad_left >= 0 && pad_left < K, "dwconv_glu_bwd: bad arguments");
  const int To = (T_ - 1) / stride + 1;
  long long n4 = (long long)B * T_ * (C / 4); long long nb = (n4 + 255) / 256; if (nb > 4096) nb = 4096;
  const int nchunks = (To + DW_TT - 1) / DW_TT;
  dim3 grid((unsigned)((C / 4 + 31) / 32), (unsigned)(B * nchunks)); ColWs ws = col_ws_if(grid, KMAX + 1, C, st);
  size_t lds = (size_t)((DW_TT - 1) * stride + K) * 128 * sizeof(float);
  if (lds < (size_t)4 * (KMAX + 1) * 128 * sizeof(float)) lds = (size_t)4 * (KMAX + 1) * 128 * sizeof(float);      // the reduction image of the weight-gradient kernel
  AVEC_CHECK_ARG(lds <= 64 * 1024, "dwconv_glu_bwd: stride %d too large", stride);
  static const bool no_fused = false;
  if (stride == 1 && !no_fused) {
    size_t l2 = (size_t)2 * (DW_TT - 1 + K) * 128 * sizeof(float); if (l2 < lds) l2 = lds;
    DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv_glu_bwd_fused_kernel<T>, grid, dim3(256), l2, st, (const T*)dc, (const T*)u, w, (T*)du, dw, dbias, B, T_, C, K, pad_left, nchunks, ws));
  } else
  DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv_glu_bwd_input_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)dc, (const T*)u, w, (T*)du, B, T_, C, K, stride, To, pad_left);
             hipLaunchKernelGGL(dwconv_bwd_weight_kernel<T>, grid, dim3(256), lds, st, (const T*)dc, (const T*)u, dw, dbias, B, T_, C, K, stride, To, pad_left, nchunks, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) {
    float* dst[KMAX + 1];
    for (int k = 0; k < KMAX; ++k) dst[k] = (k < K) ? dw + (long long)k * C : nullptr;
    dst[KMAX] = dbias;
    return col_finalize(ws, grid.x, grid.y, KMAX + 1, 128, dst, C, st);
  }
  return 0;
}
