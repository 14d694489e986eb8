// Front-end kernels.
//   audio : framing+Hann window (feeds an exact-fp32 MFMA DFT GEMM), |.|^2 -> 80 HTK mel filters -> log,
//           SpecAugment masks, Conv2d(1->180,3x3,s2)+BatchNorm2d+Swish stem written directly in the
//           (B, T', C*F') layout the 7200->180 Linear consumes            (nnet/preprocessing.py:57-130,
//           nnet/networks.py:356-377, nnet/modules.py:70-130)
//   video : BatchNorm3d+ReLU+MaxPool3d((1,3,3),s(1,2,2)) fused after the Conv3d stem GEMM, forward and
//           backward                                                        (nnet/networks.py:459-473)
#include "vec.h"
#include "avec_hip.h"

// ---------------------------------------------------------------------------------------------
// mel: frames[m][n] = w[n] * reflect_pad(audio)[f*hop + n + (n_fft-win)/2 - n_fft/2],  n in [0, win)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mel_frames_kernel(const float* __restrict__ audio, const float* __restrict__ window, float* __restrict__ frames,
                                                         int B, long long L, int F, int n_fft, int win, int hop) {
  const long long total = (long long)B * F * win;
  const int off = (n_fft - win) / 2 - n_fft / 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i % win); const long long r = i / win; const int f = (int)(r % F); const long long b = r / F;
    long long s = (long long)f * hop + n + off;
    if (s < 0) s = -s; else if (s >= L) s = 2 * (L - 1) - s;   // reflect (no edge repeat)
    frames[i] = audio[b * L + s] * window[n];
  }
}
// spec[m] = [re(0..nb-1) | im(0..nb-1)];  out[b][mel][f] = log(sum_k (re^2+im^2) fb[k][mel] + 1e-9)
__global__ __launch_bounds__(128) void mel_power_log_kernel(const float* __restrict__ spec, const float* __restrict__ fb, float* __restrict__ out,
                                                            int F, int nb, int n_mels) {
  extern __shared__ float pw[];
  const long long m = blockIdx.x; const int f = (int)(m % F); const long long b = m / F;
  for (int k = threadIdx.x; k < nb; k += 128) { const float re = spec[m * 2 * nb + k], im = spec[m * 2 * nb + nb + k]; pw[k] = re * re + im * im; }
  __syncthreads();
  for (int j = threadIdx.x; j < n_mels; j += 128) {
    float s = 0.f;
    for (int k = 0; k < nb; ++k) s += pw[k] * fb[k * n_mels + j];
    out[(b * n_mels + j) * F + f] = logf(s + 1e-9f);
  }
}
extern "C" int avec_mel_frames(const float* audio, const float* window, float* frames, int B, long long L, int n_fft, int win, int hop, hipStream_t st) {
  AVEC_CHECK_ARG(audio && window && frames && B > 0 && L > n_fft / 2 && win <= n_fft && hop > 0, "mel_frames: bad arguments");
  const int F = (int)(L / hop) + 1; long long total = (long long)B * F * win; long long nb = (total + 255) / 256; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(mel_frames_kernel, dim3((unsigned)nb), dim3(256), 0, st, audio, window, frames, B, L, F, n_fft, win, hop);
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_mel_power_log(const float* spec, const float* fb, float* out, int B, int F, int n_bins, int n_mels, hipStream_t st) {
  AVEC_CHECK_ARG(spec && fb && out && B > 0 && F > 0 && n_bins > 0 && n_mels > 0, "mel_power_log: bad arguments");
  hipLaunchKernelGGL(mel_power_log_kernel, dim3((unsigned)((long long)B * F)), dim3(128), n_bins * sizeof(float), st, spec, fb, out, F, n_bins, n_mels);
  AVEC_LAUNCH_CHECK(); return 0;
}

// ---------------------------------------------------------------------------------------------
// SpecAugment (nnet/preprocessing.py:115-130): mF batch-shared frequency masks (width U[0,Fp)), mT per-sample
// time masks (width U[0, int(pS*len_b))) inside the valid length; masked bins set to 0.  mel: [B][n_mels][F].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void specaug_kernel(float* __restrict__ mel, const long long* __restrict__ lens, int B, int n_mels, int F, int mF, int Fp, int mT, float pS,
                                                      const unsigned long long* rng, unsigned stream) {
  const unsigned long long seed = rng[0] + 0x9e3779b97f4a7c15ull * rng[1];
  const long long total = (long long)B * n_mels * F;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % F); const long long r = i / F; const int m = (int)(r % n_mels); const int b = (int)(r / n_mels);
    bool masked = false;
    for (int q = 0; q < mF && !masked; ++q) {
      const float val = rng_uniform(seed, stream, 2 * q) * Fp; const float mn = rng_uniform(seed, stream, 2 * q + 1) * (n_mels - val);
      const int s0 = (int)mn, s1 = (int)mn + (int)val; masked = (m >= s0 && m < s1);
    }
    const int len = lens ? (int)lens[b] : F; const int Tp = (int)(pS * len);
    if (f < len) for (int q = 0; q < mT && !masked; ++q) {
      const unsigned long long id = 1000ull + (unsigned long long)b * 64 + 2 * q;
      const float val = rng_uniform(seed, stream, id) * Tp; const float mn = rng_uniform(seed, stream, id + 1) * (len - val);
      const int s0 = (int)mn, s1 = (int)mn + (int)val; masked = (f >= s0 && f < s1);
    }
    if (masked) mel[i] = 0.f;
  }
}
extern "C" int avec_specaugment(float* mel, const long long* lens, int B, int n_mels, int F, int mF, int Fparam, int mT, float pS,
                                const unsigned long long* rng, unsigned rng_stream, hipStream_t st) {
  AVEC_CHECK_ARG(mel && rng && B > 0 && n_mels > 0 && F > 0, "specaugment: bad arguments");
  long long total = (long long)B * n_mels * F; long long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(specaug_kernel, dim3((unsigned)nb), dim3(256), 0, st, mel, lens, B, n_mels, F, mF, Fparam, mT, pS, rng, rng_stream);
  AVEC_LAUNCH_CHECK(); return 0;
}

__global__ void debug_rng_uniform_kernel(const unsigned long long* rng, unsigned stream, const long long* ids, int n, float* out) {
  const unsigned long long seed = rng[0] + 0x9e3779b97f4a7c15ull * rng[1];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = rng_uniform(seed, stream, (unsigned long long)ids[i]);
}
extern "C" int avec_debug_rng_uniform(const unsigned long long* rng, unsigned rng_stream, const long long* ids, int n, float* out, hipStream_t st) {
  AVEC_CHECK_ARG(rng && ids && out && n > 0, "debug_rng_uniform: bad arguments");
  hipLaunchKernelGGL(debug_rng_uniform_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rng, rng_stream, ids, n, out);
  AVEC_LAUNCH_CHECK(); return 0;
}

// ---------------------------------------------------------------------------------------------
// audio stem: y[b][to][c*Fo + fo] = bias[c] + sum_{kh,kw} w[c][kh][kw] * mel[b][2fo+kh-1][2to+kw-1]
// ---------------------------------------------------------------------------------------------
struct StemA { int B, NM, F, C, Fo, To; };

// add per-thread partials (a, b) of channel c to dst[c], dst[C + c]: LDS atomics inside the block (a 256-thread block spans <= 8 channels),
// then one global atomic per channel per block
__device__ __forceinline__ void block_channel_add2(float* dst, int C, int c, bool live, float a, float b) {
  __shared__ float red[2][16]; __shared__ int cbase;
  if (threadIdx.x == 0) cbase = c;
  if (threadIdx.x < 32) red[threadIdx.x >> 4][threadIdx.x & 15] = 0.f;
  __syncthreads();
  const int k = c - cbase;
  if (live && k >= 0 && k < 16) { atomicAdd(&red[0][k], a); atomicAdd(&red[1][k], b); }
  else if (live) { atomicAdd(dst + c, a); atomicAdd(dst + C + c, b); }
  __syncthreads();
  if (threadIdx.x < 16 && cbase + (int)threadIdx.x < C && (red[0][threadIdx.x] != 0.f || red[1][threadIdx.x] != 0.f)) {
    atomicAdd(dst + cbase + threadIdx.x, red[0][threadIdx.x]); atomicAdd(dst + C + cbase + threadIdx.x, red[1][threadIdx.x]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void audio_stem_conv_kernel(const float* __restrict__ mel, const float* __restrict__ w, const float* __restrict__ bias,
                                                              T* __restrict__ y, float* stats, StemA s) {
  int j = blockIdx.x * 256 + threadIdx.x; const int J = s.C * s.Fo;
  const bool live = j < J; if (!live) j = J - 1;          // keep every thread for the block-level reduction
  const int c = j / s.Fo, fo = j % s.Fo;
  float wk[9]; for (int q = 0; q < 9; ++q) wk[q] = w[c * 9 + q];
  const float bb = bias ? bias[c] : 0.f;
  float sum = 0.f, sq = 0.f;
  const long long M = (long long)s.B * s.To;
  for (long long row = blockIdx.y; row < M; row += gridDim.y) {
    const int to = (int)(row % s.To); const long long b = row / s.To;
    float acc = bb;
    for (int kh = 0; kh < 3; ++kh) { const int fi = 2 * fo + kh - 1; if (fi < 0 || fi >= s.NM) continue;
      for (int kw = 0; kw < 3; ++kw) { const int ti = 2 * to + kw - 1; if (ti < 0 || ti >= s.F) continue;
        acc += wk[kh * 3 + kw] * mel[(b * s.NM + fi) * s.F + ti]; } }
    if (live) { stf(y + row * J + j, acc); sum += acc; sq += acc * acc; }
  }
  if (stats) block_channel_add2(stats, s.C, c, live, sum, sq);
}
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_act_kernel(const T* __restrict__ y, const float* __restrict__ ss, T* __restrict__ a, long long M, int J, int Fo, int C) {
  const long long total = M * J;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)((unsigned)(i % (long long)J)) / Fo; stf(a + i, swishf_(ldf(y + i) * ss[c] + ss[C + c]));
  }
}
// 8 consecutive frequency bins (one channel) per thread: 16-byte accesses, 32-bit index arithmetic (Fo % 8 == 0, fewer than 2^31 chunks)
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_act8_kernel(const T* __restrict__ y, const float* __restrict__ ss, T* __restrict__ a, unsigned n8, unsigned J8, int Fo, int C) {
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n8; i += gridDim.x * 256) {
    const int c = (int)((i % J8) * 8) / Fo; const float sc = ss[c], sh = ss[C + c];
    Raw8<T> r; r.load(y + (long long)i * 8); float v[8]; r.get(v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = swishf_(v[e] * sc + sh);
    st8<T>(a + (long long)i * 8, v);
  }
}
// pass 1: dstats[c] += sum dr, dstats[C+c] += sum dr*yhat   with dr = da * swish'(pre)
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_bwd_reduce_kernel(const T* __restrict__ da, const T* __restrict__ y, const float* __restrict__ ss, float* dstats, StemA s) {
  int j = blockIdx.x * 256 + threadIdx.x; const int J = s.C * s.Fo;
  const bool live = j < J; if (!live) j = J - 1;
  const int c = j / s.Fo; const float sc = ss[c], sh = ss[s.C + c], mu = ss[2 * s.C + c], rs = ss[3 * s.C + c];
  float s1 = 0.f, s2 = 0.f; const long long M = (long long)s.B * s.To;
  if (live) for (long long row = blockIdx.y; row < M; row += gridDim.y) {
    const float yy = ldf(y + row * J + j); const float dr = ldf(da + row * J + j) * dswishf_(yy * sc + sh);
    s1 += dr; s2 += dr * (yy - mu) * rs;
  }
  block_channel_add2(dstats, s.C, c, live, s1, s2);
}
// pass 2: dy = gamma*rstd*(dr - s1/n - yhat*s2/n);  dw[c][kh][kw] += sum dy*mel(patch);  dbias[c] += sum dy; block row 0 adds dgamma/dbeta
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_bwd_params_kernel(const T* __restrict__ da, const T* __restrict__ y, const float* __restrict__ mel, const float* __restrict__ ss,
                                                                    const float* __restrict__ gamma, const float* __restrict__ dstats, const float* count_ptr, float count,
                                                                    float* dw, float* dbias, float* dgamma, float* dbeta, StemA s) {
  int j = blockIdx.x * 256 + threadIdx.x; const int J = s.C * s.Fo;
  const bool live = j < J; if (!live) j = J - 1;
  const int c = j / s.Fo, fo = j % s.Fo; const float inv_n = 1.f / (count_ptr ? *count_ptr : count);
  const float sc = ss[c], sh = ss[s.C + c], mu = ss[2 * s.C + c], rs = ss[3 * s.C + c], g = gamma[c];
  const float m1 = dstats[c] * inv_n, m2 = dstats[s.C + c] * inv_n;
  if (live && blockIdx.y == 0 && fo == 0 && dgamma) { atomicAdd(dgamma + c, dstats[s.C + c]); atomicAdd(dbeta + c, dstats[c]); }
  float aw[9]; for (int q = 0; q < 9; ++q) aw[q] = 0.f; float ab = 0.f;
  const long long M = (long long)s.B * s.To;
  if (live) for (long long row = blockIdx.y; row < M; row += gridDim.y) {
    const int to = (int)(row % s.To); const long long b = row / s.To;
    const float yy = ldf(y + row * J + j); const float dr = ldf(da + row * J + j) * dswishf_(yy * sc + sh);
    const float dy = g * rs * (dr - m1 - (yy - mu) * rs * m2);
    ab += dy;
    for (int kh = 0; kh < 3; ++kh) { const int fi = 2 * fo + kh - 1; if (fi < 0 || fi >= s.NM) continue;
      for (int kw = 0; kw < 3; ++kw) { const int ti = 2 * to + kw - 1; if (ti < 0 || ti >= s.F) continue;
        aw[kh * 3 + kw] += dy * mel[(b * s.NM + fi) * s.F + ti]; } }
  }
  // block-level reduction: dw is [C][9] = per-channel groups of 9 -> reuse the 2-vector helper on (tap q, tap q+1) pairs via a flat [C*9] view
  __shared__ float wred[16][10]; __shared__ int wbase;
  if (threadIdx.x == 0) wbase = c;
  if (threadIdx.x < 160) wred[threadIdx.x / 10][threadIdx.x % 10] = 0.f;
  __syncthreads();
  const int kq = c - wbase;
  if (live && kq >= 0 && kq < 16) { for (int q = 0; q < 9; ++q) atomicAdd(&wred[kq][q], aw[q]); atomicAdd(&wred[kq][9], ab); }
  else if (live) { for (int q = 0; q < 9; ++q) atomicAdd(dw + c * 9 + q, aw[q]); if (dbias) atomicAdd(dbias + c, ab); }
  __syncthreads();
  if (threadIdx.x < 160) {
    const int cc = wbase + threadIdx.x / 10, q = threadIdx.x % 10; const float v = wred[threadIdx.x / 10][q];
    if (cc < s.C && v != 0.f) { if (q < 9) atomicAdd(dw + cc * 9 + q, v); else if (dbias) atomicAdd(dbias + cc, v); }
  }
}
// ---- 8-wide audio stem (Fo % 8 == 0): workgroup = AS_ROWS consecutive output rows (b, to); the 3 mel columns a row touches are staged
// zero-padded in LDS ([NM + 2][3]), thread = 8 consecutive fo of one channel (one 16-byte store); per-channel sums accumulate in LDS
// over the block's rows and leave through the two-pass reduction workspace (vec.h) when one is registered.
#ifndef AVEC_AS_ROWS
#define AVEC_AS_ROWS 16
#endif
static constexpr int AS_ROWS = AVEC_AS_ROWS;
__device__ __forceinline__ void stage_mel_patch(float* patch, const float* mel, long long b, int to, const StemA& s) {
  for (int idx = threadIdx.x; idx < (s.NM + 2) * 3; idx += 256) {
    const int fi = idx / 3 - 1, kw = idx - (fi + 1) * 3; const int ti = 2 * to + kw - 1;
    patch[idx] = (fi >= 0 && fi < s.NM && ti >= 0 && ti < s.F) ? mel[(b * s.NM + fi) * s.F + ti] : 0.f;
  }
}
__device__ __forceinline__ void as_commit(const float* lsum, int n, float* const* dst, int ndst, int W, const ColWs& ws) {   // lsum: [ndst][W] in LDS
  if (ws.partial) { float* mine = ws_slot(ws, 0, blockIdx.x, gridDim.x, n); for (int i = threadIdx.x; i < n; i += 256) mine[i] = lsum[i]; }
  else for (int i = threadIdx.x; i < n; i += 256) { const int q = i / W; if (dst[q] && lsum[i] != 0.f) atomicAdd(dst[q] + (i - q * W), lsum[i]); }
}
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_conv8_kernel(const float* __restrict__ mel, const float* __restrict__ w, const float* __restrict__ bias,
                                                               T* __restrict__ y, float* stats, StemA s, ColWs ws) {
  extern __shared__ float asm_[];
  float* patch = asm_; float* lw = patch + (s.NM + 2) * 3; float* lsum = lw + s.C * 10;      // lw: [C][9 taps | bias], lsum: [2][C]
  const int J = s.C * s.Fo, F8 = s.Fo >> 3, chunks = s.C * F8; const long long M = (long long)s.B * s.To;
  for (int i = threadIdx.x; i < s.C * 10; i += 256) { const int c = i / 10, q = i - c * 10; lw[i] = q < 9 ? w[c * 9 + q] : (bias ? bias[c] : 0.f); }
  for (int i = threadIdx.x; i < 2 * s.C; i += 256) lsum[i] = 0.f;
  for (int rr = 0; rr < AS_ROWS; ++rr) {
    const long long row = (long long)blockIdx.x * AS_ROWS + rr; if (row >= M) break;
    const int to = (int)(row % s.To); const long long b = row / s.To;
    __syncthreads();
    stage_mel_patch(patch, mel, b, to, s);
    __syncthreads();
    for (int ch = threadIdx.x; ch < chunks; ch += 256) {
      const int c = ch / F8, fo0 = (ch - c * F8) * 8; const float* wk = lw + c * 10;
      float acc[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* pp = patch + (2 * (fo0 + e)) * 3; float a = wk[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) a += wk[q] * pp[q];          // patch rows 2fo .. 2fo+2 are contiguous: [kh][kw]
        acc[e] = a; s1 += a; s2 += a * a;
      }
      st8<T>(y + row * J + c * s.Fo + fo0, acc);
      if (stats) { atomicAdd(lsum + c, s1); atomicAdd(lsum + s.C + c, s2); }
    }
  }
  __syncthreads();
  if (stats) { float* const dst[2] = {stats, stats + s.C}; as_commit(lsum, 2 * s.C, dst, 2, s.C, ws); }
}
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_bwd_reduce8_kernel(const T* __restrict__ da, const T* __restrict__ y, const float* __restrict__ ss, float* dstats, StemA s, ColWs ws) {
  extern __shared__ float asm_[];
  float* lsum = asm_;
  const int J = s.C * s.Fo, F8 = s.Fo >> 3, chunks = s.C * F8; const long long M = (long long)s.B * s.To;
  for (int i = threadIdx.x; i < 2 * s.C; i += 256) lsum[i] = 0.f;
  __syncthreads();
  for (int rr = 0; rr < AS_ROWS; ++rr) {
    const long long row = (long long)blockIdx.x * AS_ROWS + rr; if (row >= M) break;
    for (int ch = threadIdx.x; ch < chunks; ch += 256) {
      const int c = ch / F8, fo0 = (ch - c * F8) * 8; const float sc = ss[c], sh = ss[s.C + c], mu = ss[2 * s.C + c], rs = ss[3 * s.C + c];
      float d[8], v[8], s1 = 0.f, s2 = 0.f; ld8<T>(da + row * J + c * s.Fo + fo0, d); ld8<T>(y + row * J + c * s.Fo + fo0, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dr = d[e] * dswishf_(v[e] * sc + sh); s1 += dr; s2 += dr * (v[e] - mu) * rs; }
      atomicAdd(lsum + c, s1); atomicAdd(lsum + s.C + c, s2);
    }
  }
  __syncthreads();
  float* const dst[2] = {dstats, dstats + s.C}; as_commit(lsum, 2 * s.C, dst, 2, s.C, ws);
}
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_bwd_params8_kernel(const T* __restrict__ da, const T* __restrict__ y, const float* __restrict__ mel, const float* __restrict__ ss,
                                                                     const float* __restrict__ gamma, const float* __restrict__ dstats, const float* count_ptr, float count,
                                                                     float* dw, float* dbias, float* dgamma, float* dbeta, StemA s, ColWs ws) {
  extern __shared__ float asm_[];
  float* patch = asm_; float* lsum = patch + (s.NM + 2) * 3;                  // lsum: [10][C] = 9 taps, then the bias gradient
  const int J = s.C * s.Fo, F8 = s.Fo >> 3, chunks = s.C * F8; const long long M = (long long)s.B * s.To;
  const float inv_n = 1.f / (count_ptr ? *count_ptr : count);
  if (blockIdx.x == 0 && dgamma) for (int c = threadIdx.x; c < s.C; c += 256) { atomicAdd(dgamma + c, dstats[s.C + c]); atomicAdd(dbeta + c, dstats[c]); }
  for (int i = threadIdx.x; i < 10 * s.C; i += 256) lsum[i] = 0.f;
  for (int rr = 0; rr < AS_ROWS; ++rr) {
    const long long row = (long long)blockIdx.x * AS_ROWS + rr; if (row >= M) break;
    const int to = (int)(row % s.To); const long long b = row / s.To;
    __syncthreads();
    stage_mel_patch(patch, mel, b, to, s);
    __syncthreads();
    for (int ch = threadIdx.x; ch < chunks; ch += 256) {
      const int c = ch / F8, fo0 = (ch - c * F8) * 8;
      const float sc = ss[c], sh = ss[s.C + c], mu = ss[2 * s.C + c], rs = ss[3 * s.C + c], g = gamma[c];
      const float m1 = dstats[c] * inv_n, m2 = dstats[s.C + c] * inv_n;
      float d[8], v[8], aw[9], ab = 0.f; ld8<T>(da + row * J + c * s.Fo + fo0, d); ld8<T>(y + row * J + c * s.Fo + fo0, v);
#pragma unroll
      for (int q = 0; q < 9; ++q) aw[q] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dr = d[e] * dswishf_(v[e] * sc + sh); const float dy = g * rs * (dr - m1 - (v[e] - mu) * rs * m2);
        ab += dy; const float* pp = patch + (2 * (fo0 + e)) * 3;
#pragma unroll
        for (int q = 0; q < 9; ++q) aw[q] += dy * pp[q];
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) atomicAdd(lsum + q * s.C + c, aw[q]);
      atomicAdd(lsum + 9 * s.C + c, ab);
    }
  }
  __syncthreads();
  if (ws.partial) { float* mine = ws_slot(ws, 0, blockIdx.x, gridDim.x, 10 * s.C); for (int i = threadIdx.x; i < 10 * s.C; i += 256) mine[i] = lsum[i]; }
  else for (int i = threadIdx.x; i < 10 * s.C; i += 256) { const int q = i / s.C, c = i - q * s.C; if (q < 9) atomicAdd(dw + c * 9 + q, lsum[i]); else if (dbias) atomicAdd(dbias + c, lsum[i]); }
}
// ---- the same three passes for chunks <= 4 * 256 (180 channels x 40 bins = 900): a thread OWNS its (up to) four chunks for all rows of the workgroup.  The first
// version ran ~10x below both the HBM and the VALU time of these passes: every chunk iteration was load -> wait -> ~600 dependent instructions -> 10 LDS atomics,
// with 1.5 workgroups per CU to hide it.  Here a row's global loads are issued together before any arithmetic, the 51 patch values of a chunk arrive as 13 ds_read_b128
// (were 72 ds_read_b32), per-chunk constants live in registers, and the per-channel sums stay in registers until the workgroup is through (one LDS atomic per sum).
constexpr int AS_NIT = 4;
__device__ __forceinline__ void as_patch51(const float* patch, int fo0, float p[52]) {      // patch rows 2 fo0 .. 2 fo0 + 16 (x 3 taps): 51 contiguous floats, 16-byte aligned
  const float4* q = (const float4*)(patch + 6 * fo0);
#pragma unroll
  for (int k = 0; k < 13; ++k) { const float4 t = q[k]; p[4 * k] = t.x; p[4 * k + 1] = t.y; p[4 * k + 2] = t.z; p[4 * k + 3] = t.w; }
}
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_conv8x_kernel(const float* __restrict__ mel, const float* __restrict__ w, const float* __restrict__ bias,
                                                                T* __restrict__ y, float* stats, StemA s, ColWs ws) {
  extern __shared__ __attribute__((aligned(16))) float asm_[];
  float* patch = asm_; float* lsum = patch + (((s.NM + 2) * 3 + 4 + 3) & ~3);             // (+4: the last chunk's 13th float4 reaches one float past the patch)
  const int J = s.C * s.Fo, F8 = s.Fo >> 3, chunks = s.C * F8; const long long M = (long long)s.B * s.To;
  int cc[AS_NIT], fo[AS_NIT]; float wk[AS_NIT][10], s1[AS_NIT], s2[AS_NIT];
#pragma unroll
  for (int it = 0; it < AS_NIT; ++it) {
    const int ch = threadIdx.x + 256 * it; const bool ok = ch < chunks; cc[it] = ok ? ch / F8 : -1; fo[it] = ok ? (ch - cc[it] * F8) * 8 : 0; s1[it] = s2[it] = 0.f;
#pragma unroll
    for (int q = 0; q < 10; ++q) wk[it][q] = ok ? (q < 9 ? w[cc[it] * 9 + q] : (bias ? bias[cc[it]] : 0.f)) : 0.f;
  }
  for (int i = threadIdx.x; i < 2 * s.C; i += 256) lsum[i] = 0.f;
  for (int rr = 0; rr < AS_ROWS; ++rr) {
    const long long row = (long long)blockIdx.x * AS_ROWS + rr; if (row >= M) break;
    const int to = (int)(row % s.To); const long long b = row / s.To;
    __syncthreads();
    stage_mel_patch(patch, mel, b, to, s);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < AS_NIT; ++it) {
      if (cc[it] < 0) continue;
      float p[52], acc[8]; as_patch51(patch, fo[it], p);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = wk[it][9];
#pragma unroll
        for (int q = 0; q < 9; ++q) a += wk[it][q] * p[6 * e + q];
        acc[e] = a; s1[it] += a; s2[it] += a * a;
      }
      st8<T>(y + row * J + cc[it] * s.Fo + fo[it], acc);
    }
  }
  if (!stats) return;
#pragma unroll
  for (int it = 0; it < AS_NIT; ++it) if (cc[it] >= 0) { atomicAdd(lsum + cc[it], s1[it]); atomicAdd(lsum + s.C + cc[it], s2[it]); }
  __syncthreads();
  float* const dst[2] = {stats, stats + s.C}; as_commit(lsum, 2 * s.C, dst, 2, s.C, ws);
}
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_bwd_reduce8x_kernel(const T* __restrict__ da, const T* __restrict__ y, const float* __restrict__ ss, float* dstats, StemA s, ColWs ws) {
  extern __shared__ __attribute__((aligned(16))) float asm_[];
  float* lsum = asm_;
  const int J = s.C * s.Fo, F8 = s.Fo >> 3, chunks = s.C * F8; const long long M = (long long)s.B * s.To;
  int cc[AS_NIT], off[AS_NIT]; float sc[AS_NIT], sh[AS_NIT], mu[AS_NIT], rs[AS_NIT], s1[AS_NIT], s2[AS_NIT];
#pragma unroll
  for (int it = 0; it < AS_NIT; ++it) {
    const int ch = threadIdx.x + 256 * it; const bool ok = ch < chunks; const int c = ok ? ch / F8 : 0; cc[it] = ok ? c : -1; off[it] = c * s.Fo + (ch - c * F8) * 8;
    sc[it] = ss[c]; sh[it] = ss[s.C + c]; mu[it] = ss[2 * s.C + c]; rs[it] = ss[3 * s.C + c]; s1[it] = s2[it] = 0.f;
  }
  for (int i = threadIdx.x; i < 2 * s.C; i += 256) lsum[i] = 0.f;
  for (int rr = 0; rr < AS_ROWS; ++rr) {
    const long long row = (long long)blockIdx.x * AS_ROWS + rr; if (row >= M) break;
    Raw8<T> rd[AS_NIT], rv[AS_NIT];
#pragma unroll
    for (int it = 0; it < AS_NIT; ++it) if (cc[it] >= 0) { rd[it].load(da + row * J + off[it]); rv[it].load(y + row * J + off[it]); }
#pragma unroll
    for (int it = 0; it < AS_NIT; ++it) {
      if (cc[it] < 0) continue;
      float d[8], v[8]; rd[it].get(d); rv[it].get(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dr = d[e] * dswishf_(v[e] * sc[it] + sh[it]); s1[it] += dr; s2[it] += dr * (v[e] - mu[it]) * rs[it]; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < AS_NIT; ++it) if (cc[it] >= 0) { atomicAdd(lsum + cc[it], s1[it]); atomicAdd(lsum + s.C + cc[it], s2[it]); }
  __syncthreads();
  float* const dst[2] = {dstats, dstats + s.C}; as_commit(lsum, 2 * s.C, dst, 2, s.C, ws);
}
template <typename T>
__global__ __launch_bounds__(256) void audio_stem_bwd_params8x_kernel(const T* __restrict__ da, const T* __restrict__ y, const float* __restrict__ mel, const float* __restrict__ ss,
                                                                      const float* __restrict__ gamma, const float* __restrict__ dstats, const float* count_ptr, float count,
                                                                      float* dw, float* dbias, float* dgamma, float* dbeta, StemA s, ColWs ws) {
  extern __shared__ __attribute__((aligned(16))) float asm_[];
  float* patch = asm_; float* lsum = patch + (((s.NM + 2) * 3 + 4 + 3) & ~3);             // lsum: [10][C] = 9 taps, then the bias gradient
  const int J = s.C * s.Fo, F8 = s.Fo >> 3, chunks = s.C * F8; const long long M = (long long)s.B * s.To;
  const float inv_n = 1.f / (count_ptr ? *count_ptr : count);
  if (blockIdx.x == 0 && dgamma) for (int c = threadIdx.x; c < s.C; c += 256) { atomicAdd(dgamma + c, dstats[s.C + c]); atomicAdd(dbeta + c, dstats[c]); }
  int cc[AS_NIT], fo[AS_NIT]; float sc[AS_NIT], sh[AS_NIT], mu[AS_NIT], A[AS_NIT], m1[AS_NIT], rm2[AS_NIT], aw[AS_NIT][10];
#pragma unroll
  for (int it = 0; it < AS_NIT; ++it) {
    const int ch = threadIdx.x + 256 * it; const bool ok = ch < chunks; const int c = ok ? ch / F8 : 0; cc[it] = ok ? c : -1; fo[it] = ok ? (ch - c * F8) * 8 : 0;
    const float rs = ss[3 * s.C + c];
    sc[it] = ss[c]; sh[it] = ss[s.C + c]; mu[it] = ss[2 * s.C + c]; A[it] = gamma[c] * rs; m1[it] = dstats[c] * inv_n; rm2[it] = rs * dstats[s.C + c] * inv_n;
#pragma unroll
    for (int q = 0; q < 10; ++q) aw[it][q] = 0.f;
  }
  for (int i = threadIdx.x; i < 10 * s.C; i += 256) lsum[i] = 0.f;
  for (int rr = 0; rr < AS_ROWS; ++rr) {
    const long long row = (long long)blockIdx.x * AS_ROWS + rr; if (row >= M) break;
    const int to = (int)(row % s.To); const long long b = row / s.To;
    Raw8<T> rd[AS_NIT], rv[AS_NIT];                                  // this row's gradients / conv outputs: requested before the patch is staged
#pragma unroll
    for (int it = 0; it < AS_NIT; ++it) if (cc[it] >= 0) { const long long o = row * J + cc[it] * s.Fo + fo[it]; rd[it].load(da + o); rv[it].load(y + o); }
    __syncthreads();
    stage_mel_patch(patch, mel, b, to, s);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < AS_NIT; ++it) {
      if (cc[it] < 0) continue;
      float d[8], v[8], p[52]; rd[it].get(d); rv[it].get(v); as_patch51(patch, fo[it], p);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dr = d[e] * dswishf_(v[e] * sc[it] + sh[it]); const float dy = A[it] * (dr - m1[it] - (v[e] - mu[it]) * rm2[it]);
        aw[it][9] += dy;
#pragma unroll
        for (int q = 0; q < 9; ++q) aw[it][q] += dy * p[6 * e + q];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < AS_NIT; ++it) if (cc[it] >= 0) {
#pragma unroll
    for (int q = 0; q < 10; ++q) atomicAdd(lsum + q * s.C + cc[it], aw[it][q]);
  }
  __syncthreads();
  if (ws.partial) { float* mine = ws_slot(ws, 0, blockIdx.x, gridDim.x, 10 * s.C); for (int i = threadIdx.x; i < 10 * s.C; i += 256) mine[i] = lsum[i]; }
  else for (int i = threadIdx.x; i < 10 * s.C; i += 256) { const int q = i / s.C, c = i - q * s.C; if (q < 9) atomicAdd(dw + c * 9 + q, lsum[i]); else if (dbias) atomicAdd(dbias + c, lsum[i]); }
}
static bool as8x_ok(const StemA& s) { return s.C * (s.Fo >> 3) <= AS_NIT * 256; }
static bool as8_ok(const StemA& s) { return s.Fo % 8 == 0 && ((size_t)(s.NM + 2) * 3 + (size_t)s.C * 12) * 4 <= 60 * 1024; }
static unsigned as8_blocks(const StemA& s) { return (unsigned)(((long long)s.B * s.To + AS_ROWS - 1) / AS_ROWS); }
static StemA stemA(int B, int NM, int F, int C) { StemA s; s.B = B; s.NM = NM; s.F = F; s.C = C; s.Fo = (NM - 1) / 2 + 1; s.To = (F - 1) / 2 + 1; return s; }
static dim3 stem_grid(const StemA& s) { long long M = (long long)s.B * s.To; unsigned gx = (s.C * s.Fo + 255) / 256; long long gy = 2048 / gx; if (gy > M) gy = M; if (gy < 1) gy = 1; return dim3(gx, (unsigned)gy); }

extern "C" int avec_audio_stem_conv_fwd(int dtype, const float* mel, const float* w, const float* bias, void* y, float* stats, int B, int n_mels, int F, int C, hipStream_t st) {
  AVEC_CHECK_ARG(mel && w && y && B > 0 && n_mels > 0 && F > 0 && C > 0, "audio_stem_conv_fwd: bad arguments");
  StemA s = stemA(B, n_mels, F, C);
  if (as8_ok(s)) {
    const unsigned nb = as8_blocks(s); ColWs ws = stats ? avec_reduce_ws((size_t)nb * 2 * C, st) : ColWs{nullptr};
    const size_t lds = ((size_t)(n_mels + 2) * 3 + (size_t)C * 12) * 4;
    static const bool no_x = false;
    if (as8x_ok(s) && !no_x) { const size_t l2 = ((size_t)(((n_mels + 2) * 3 + 4 + 3) & ~3) + (size_t)C * 2) * 4;
      DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_conv8x_kernel<T>, dim3(nb), dim3(256), l2, st, mel, w, bias, (T*)y, stats, s, ws)); }
    else
    DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_conv8_kernel<T>, dim3(nb), dim3(256), lds, st, mel, w, bias, (T*)y, stats, s, ws));
    AVEC_LAUNCH_CHECK();
    if (ws.partial) { float* const dst[2] = {stats, stats + C}; return col_finalize(ws, 1, nb, 2, C, dst, C, st); }
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_conv_kernel<T>, stem_grid(s), dim3(256), 0, st, mel, w, bias, (T*)y, stats, s));
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_audio_stem_act_fwd(int dtype, const void* y, const float* ss, void* a, int B, int n_mels, int F, int C, hipStream_t st) {
  AVEC_CHECK_ARG(y && ss && a, "audio_stem_act_fwd: null pointer");
  StemA s = stemA(B, n_mels, F, C); long long M = (long long)B * s.To; long long nb = (M * C * s.Fo + 255) / 256; if (nb > 4096) nb = 4096;
  if (s.Fo % 8 == 0 && M * C * s.Fo / 8 < (1ll << 31)) {
    const long long n8 = M * C * s.Fo / 8; long long nb8 = (n8 + 255) / 256; if (nb8 > 8192) nb8 = 8192;
    DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_act8_kernel<T>, dim3((unsigned)nb8), dim3(256), 0, st, (const T*)y, ss, (T*)a, (unsigned)n8, (unsigned)(C * s.Fo / 8), s.Fo, C));
    AVEC_LAUNCH_CHECK(); return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_act_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)y, ss, (T*)a, M, C * s.Fo, s.Fo, C));
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_audio_stem_bwd(int dtype, const void* da, const void* y, const float* mel, const float* ss, const float* gamma, float* dstats,
                                   const float* count_ptr, float count, int phase, float* dw, float* dbias, float* dgamma, float* dbeta,
                                   int B, int n_mels, int F, int C, hipStream_t st) {
  AVEC_CHECK_ARG(da && y && mel && ss && gamma && dstats && (phase == 0 || (dw != nullptr)), "audio_stem_bwd: bad arguments");
  StemA s = stemA(B, n_mels, F, C);
  if (as8_ok(s)) {
    const unsigned nb = as8_blocks(s);
    if (phase == 0) {
      ColWs ws = avec_reduce_ws((size_t)nb * 2 * C, st);
      static const bool no_x = false;
      if (as8x_ok(s) && !no_x) DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_bwd_reduce8x_kernel<T>, dim3(nb), dim3(256), (size_t)2 * C * 4, st, (const T*)da, (const T*)y, ss, dstats, s, ws));
      else
      DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_bwd_reduce8_kernel<T>, dim3(nb), dim3(256), (size_t)2 * C * 4, st, (const T*)da, (const T*)y, ss, dstats, s, ws));
      AVEC_LAUNCH_CHECK();
      if (ws.partial) { float* const dst[2] = {dstats, dstats + C}; return col_finalize(ws, 1, nb, 2, C, dst, C, st); }
    } else {
      ColWs ws = avec_reduce_ws((size_t)nb * 10 * C, st);
      const size_t lds = ((size_t)(n_mels + 2) * 3 + (size_t)C * 10) * 4;
      static const bool no_x = false;
      if (as8x_ok(s) && !no_x) { const size_t l2 = ((size_t)(((n_mels + 2) * 3 + 4 + 3) & ~3) + (size_t)C * 10) * 4;
        DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_bwd_params8x_kernel<T>, dim3(nb), dim3(256), l2, st, (const T*)da, (const T*)y, mel, ss, gamma, dstats, count_ptr, count,
                                             dw, dbias, dgamma, dbeta, s, ws)); }
      else
      DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_bwd_params8_kernel<T>, dim3(nb), dim3(256), lds, st, (const T*)da, (const T*)y, mel, ss, gamma, dstats, count_ptr, count,
                                           dw, dbias, dgamma, dbeta, s, ws));
      AVEC_LAUNCH_CHECK();
      if (ws.partial) {      // partial rows are [10][C]: taps 0..8 go to dw[c*9 + q] (element stride 9), row 9 to dbias
        float* dst[10]; for (int q = 0; q < 9; ++q) dst[q] = dw + q; dst[9] = nullptr;
        if (int r = col_finalize(ws, 1, nb, 10, C, dst, C, st, 9)) return r;
        if (dbias) { for (int q = 0; q < 9; ++q) dst[q] = nullptr; dst[9] = dbias; return col_finalize(ws, 1, nb, 10, C, dst, C, st, 1); }
      }
    }
    return 0;
  }
  if (phase == 0) { DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_bwd_reduce_kernel<T>, stem_grid(s), dim3(256), 0, st, (const T*)da, (const T*)y, ss, dstats, s)); }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL(audio_stem_bwd_params_kernel<T>, stem_grid(s), dim3(256), 0, st, (const T*)da, (const T*)y, mel, ss, gamma, dstats, count_ptr, count, dw, dbias, dgamma, dbeta, s)); }
  AVEC_LAUNCH_CHECK(); return 0;
}

// ---------------------------------------------------------------------------------------------
// video stem tail: a = relu(y*scale+shift) on [Fr][H][W][C]; 3x3 stride-2 max pool with zero pad 1 -> [Fr][H/2][W/2][C]
// idx (uint8) = window slot kh*3+kw of the maximum (255: the zero padding won / dead ReLU)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ ss, T* __restrict__ out, unsigned char* __restrict__ idx,
                                                            T* __restrict__ ymax, long long Fr, int H, int W, int C, int OH, int OW) {
  const long long n4 = Fr * OH * OW * (C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const unsigned iu = (unsigned)i, C4 = (unsigned)C >> 2;      // host-checked: fewer than 2^31 work items -> 32-bit divisions
    const int c = (int)(iu % C4) * 4; unsigned r = iu / C4; const int ow = (int)(r % (unsigned)OW); r /= (unsigned)OW; const int oh = (int)(r % (unsigned)OH); const long long fr = r / (unsigned)OH;
    float sc[4], sh[4]; ld4<float>(ss + c, sc); ld4<float>(ss + C + c, sh);
    float best[4] = {0.f, 0.f, 0.f, 0.f}, braw[4] = {0.f, 0.f, 0.f, 0.f}; unsigned bi[4] = {255u, 255u, 255u, 255u};
    float v[9][4]; bool ok[9];                          // unconditional (clamped) loads first: loads under divergent `continue`s are serialised
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int kh = q / 3, kw = q - kh * 3; const int h = 2 * oh + kh - 1, w = 2 * ow + kw - 1;
      ok[q] = h >= 0 && h < H && w >= 0 && w < W;
      ld4<T>(y + ((fr * H + (ok[q] ? h : 0)) * W + (ok[q] ? w : 0)) * C + c, v[q]);
    }
#pragma unroll
    for (int q = 0; q < 9; ++q)
      for (int e = 0; e < 4; ++e) { const float a = v[q][e] * sc[e] + sh[e]; if (ok[q] && a > best[e]) { best[e] = a; bi[e] = q; braw[e] = v[q][e]; } }
    st4<T>(out + i * 4, best);
    if (ymax) st4<T>(ymax + i * 4, braw);              // the conv output under the winning tap: lets the backward statistics pass run over the pooled domain
    *(uint32_t*)(idx + i * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
  }
}
// dr at input position (h,w): sum over the (<=4) windows that selected it
template <typename T>
__device__ __forceinline__ void stem_dr(const T* dp, const unsigned char* idx, long long fr, int h, int w, int c, int C, int OH, int OW, float dr[4]) {
  dr[0] = dr[1] = dr[2] = dr[3] = 0.f;
  for (int kh = 0; kh < 3; ++kh) { const int t = h + 1 - kh; if (t < 0 || (t & 1)) continue; const int oh = t >> 1; if (oh >= OH) continue;
    for (int kw = 0; kw < 3; ++kw) { const int u = w + 1 - kw; if (u < 0 || (u & 1)) continue; const int ow = u >> 1; if (ow >= OW) continue;
      const long long o = ((fr * OH + oh) * OW + ow) * C + c;
      const uint32_t sel = *(const uint32_t*)(idx + o); float g[4]; ld4<T>(dp + o, g);
      for (int e = 0; e < 4; ++e) if (((sel >> (8 * e)) & 255u) == (unsigned)(kh * 3 + kw)) dr[e] += g[e]; } }
}
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bwd_reduce_kernel(const T* __restrict__ dp, const unsigned char* __restrict__ idx, const T* __restrict__ y, const float* __restrict__ ss,
                                                                   float* dstats, long long Fr, int H, int W, int C, int OH, int OW, ColWs ws) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const long long M = Fr * H * W;
  if (col < C) {
    float mu[4], rs[4]; ld4<float>(ss + 2 * C + col, mu); ld4<float>(ss + 3 * C + col, rs);
    for (long long row = (long long)blockIdx.y * 8 + ty; row < M; row += (long long)gridDim.y * 8) {
      const int w = (int)(row % W); long long r = row / W; const int h = (int)(r % H); const long long fr = r / H;
      float dr[4], v[4]; stem_dr<T>(dp, idx, fr, h, w, col, C, OH, OW, dr); ld4<T>(y + row * C + col, v);
      for (int e = 0; e < 4; ++e) { part[0][e] += dr[e]; part[1][e] += dr[e] * (v[e] - mu[e]) * rs[e]; }
    }
  }
  float* const dst[2] = {dstats, dstats + C};
  colreduce_atomic<2>(part, dst, col, C, ws);
}
// 8-channel variants (C % 8 == 0): 16 B accesses, every lane busy for C = 64
// Input row h lies in the 3-row windows oh = h>>1 (slot kh = 1 for even h, 2 for odd h) and, for odd h, oh = (h>>1)+1 (kh = 0); likewise for
// columns: at most 4 candidate windows.  All four (idx, dpool) pairs are fetched unconditionally at clamped addresses and masked afterwards
// -- loads issued under divergent `continue`s are serialised by the compiler (one s_waitcnt vmcnt(0) per probe).
template <typename T>
__device__ __forceinline__ void stem_dr8(const T* dp, const unsigned char* idx, long long fr, int h, int w, int c, int C, int OH, int OW, float dr[8]) {
  const int ohA = h >> 1, owA = w >> 1, khA = 1 + (h & 1), kwA = 1 + (w & 1);
  int oh[2], ow[2], kh[2], kw[2]; bool vh[2], vw[2];
  oh[0] = ohA; kh[0] = khA; vh[0] = ohA < OH; oh[1] = ohA + 1; kh[1] = 0; vh[1] = (h & 1) && ohA + 1 < OH;
  ow[0] = owA; kw[0] = kwA; vw[0] = owA < OW; ow[1] = owA + 1; kw[1] = 0; vw[1] = (w & 1) && owA + 1 < OW;
  uint2 sel[4]; float g[4][8]; unsigned slot[4]; bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = q >> 1, b = q & 1;
    ok[q] = vh[a] && vw[b]; slot[q] = (unsigned)(kh[a] * 3 + kw[b]);
    const long long o = ((fr * OH + (vh[a] ? oh[a] : 0)) * OW + (vw[b] ? ow[b] : 0)) * C + c;
    sel[q] = *(const uint2*)(idx + o); ld8<T>(dp + o, g[q]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) dr[e] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ok[q] && ((sel[q].x >> (8 * e)) & 255u) == slot[q]) dr[e] += g[q][e];
      if (ok[q] && ((sel[q].y >> (8 * e)) & 255u) == slot[q]) dr[4 + e] += g[q][4 + e];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bwd_reduce8_kernel(const T* __restrict__ dp, const unsigned char* __restrict__ idx, const T* __restrict__ y, const float* __restrict__ ss,
                                                                    float* dstats, long long Fr, int H, int W, int C, int OH, int OW, ColWs ws) {
  const Col8 m = col8_map(C);
  float part[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) part[0][e] = part[1][e] = 0.f;
  const long long M = Fr * H * W;
  if (m.active) {
    const int c = m.l * 8;
    float mu[8], rs[8]; ld8<float>(ss + 2 * C + c, mu); ld8<float>(ss + 3 * C + c, rs);
    for (long long row = (long long)blockIdx.x * m.R + m.r; row < M; row += (long long)gridDim.x * m.R) {
      const unsigned ru = (unsigned)row; const int w = (int)(ru % (unsigned)W); const unsigned r = ru / (unsigned)W; const int h = (int)(r % (unsigned)H); const long long fr = r / (unsigned)H;
      float dr[8], v[8]; stem_dr8<T>(dp, idx, fr, h, w, c, C, OH, OW, dr); ld8<T>(y + row * C + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { part[0][e] += dr[e]; part[1][e] += dr[e] * (v[e] - mu[e]) * rs[e]; }
    }
  }
  float* const dst[2] = {dstats, dstats + C};
  colreduce8_atomic<2>(part, dst, m, ws);
}
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bwd_apply8_kernel(const T* __restrict__ dp, const unsigned char* __restrict__ idx, const T* __restrict__ y, const float* __restrict__ ss,
                                                                   const float* __restrict__ gamma, const float* __restrict__ dstats, const float* count_ptr, float count,
                                                                   T* __restrict__ dy, float* dgamma, float* dbeta, long long Fr, int H, int W, int C, int OH, int OW) {
  const float inv_n = 1.f / (count_ptr ? *count_ptr : count);
  if (blockIdx.x == 0 && dgamma) for (int c = threadIdx.x; c < C; c += 256) { atomicAdd(dgamma + c, dstats[C + c]); atomicAdd(dbeta + c, dstats[c]); }
  const int C8 = C / 8; const long long n8 = Fr * H * W * C8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const unsigned iu = (unsigned)i; const int c = (int)(iu % (unsigned)C8) * 8; const unsigned rowu = iu / (unsigned)C8; const long long row = rowu;
    const int w = (int)(rowu % (unsigned)W); const unsigned r = rowu / (unsigned)W; const int h = (int)(r % (unsigned)H); const long long fr = r / (unsigned)H;
    float dr[8], v[8], mu[8], rs[8], g[8], s1[8], s2[8], o[8];
    stem_dr8<T>(dp, idx, fr, h, w, c, C, OH, OW, dr); ld8<T>(y + row * C + c, v);
    ld8<float>(ss + 2 * C + c, mu); ld8<float>(ss + 3 * C + c, rs); ld8<float>(gamma + c, g); ld8<float>(dstats + c, s1); ld8<float>(dstats + C + c, s2);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float yh = (v[e] - mu[e]) * rs[e]; o[e] = g[e] * rs[e] * (dr[e] - s1[e] * inv_n - yh * s2[e] * inv_n); }
    st8<T>(dy + row * C + c, o);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bwd_apply_kernel(const T* __restrict__ dp, const unsigned char* __restrict__ idx, const T* __restrict__ y, const float* __restrict__ ss,
                                                                  const float* __restrict__ gamma, const float* __restrict__ dstats, const float* count_ptr, float count,
                                                                  T* __restrict__ dy, float* dgamma, float* dbeta, long long Fr, int H, int W, int C, int OH, int OW) {
  const float inv_n = 1.f / (count_ptr ? *count_ptr : count);
  if (blockIdx.x == 0 && dgamma) for (int c = threadIdx.x; c < C; c += 256) { atomicAdd(dgamma + c, dstats[C + c]); atomicAdd(dbeta + c, dstats[c]); }
  const long long n4 = Fr * H * W * (C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % (C / 4)) * 4; const long long row = i / (C / 4);
    const int w = (int)(row % W); long long r = row / W; const int h = (int)(r % H); const long long fr = r / H;
    float dr[4], v[4], mu[4], rs[4], g[4], s1[4], s2[4], o[4];
    stem_dr<T>(dp, idx, fr, h, w, c, C, OH, OW, dr); ld4<T>(y + row * C + c, v);
    ld4<float>(ss + 2 * C + c, mu); ld4<float>(ss + 3 * C + c, rs); ld4<float>(gamma + c, g); ld4<float>(dstats + c, s1); ld4<float>(dstats + C + c, s2);
    for (int e = 0; e < 4; ++e) { const float yh = (v[e] - mu[e]) * rs[e]; o[e] = g[e] * rs[e] * (dr[e] - s1[e] * inv_n - yh * s2[e] * inv_n); }
    st4<T>(dy + row * C + c, o);
  }
}
// im2col for the Cin=1 Conv3d stem (5x7x7, stride (1,2,2), pad (2,3,3)): one workgroup = one output row (clip, frame, oh).
// The 5x7 input rows it touches are staged zero-padded in LDS (row pitch WP, chosen == 10 mod 32 to spread the banks); then
// 32 lanes serve one output pixel, lane q writing the 16 B chunk k = 8q..8q+7 of its im2col row (contiguous 496 B rows).
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ video, T* __restrict__ A, int T3, int H, int W, int OH, int OW, int ldk, int WP) {
  extern __shared__ float slab[];              // [35][WP]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long blk = blockIdx.x; const int oh = (int)(blk % OH); const long long ft = blk / OH; const int fr = (int)(ft % T3); const long long clip = ft / T3;
  const float* src = video + clip * (long long)T3 * H * W;
  for (int rw = wv; rw < 35; rw += 4) {
    const int kd = rw / 7, kh = rw - kd * 7; const int it = fr + kd - 2, ih = oh * 2 - 3 + kh;
    const bool rv = it >= 0 && it < T3 && ih >= 0 && ih < H;
    const float* rp = src + ((long long)(rv ? it : 0) * H + (rv ? ih : 0)) * W;
    for (int x = lane; x < WP; x += 64) { const int iw = x - 3; slab[rw * WP + x] = (rv && iw >= 0 && iw < W) ? rp[iw] : 0.f; }
  }
  const int q = threadIdx.x & 31, pg = threadIdx.x >> 5;
  int off[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const int kk = q * 8 + e; const int kw = kk % 7, t2 = kk / 7; off[e] = kk < 245 ? t2 * WP + kw : -1; }
  __syncthreads();
  const int nq = ldk / 8;
  T* Arow = A + blk * (long long)OW * ldk;
  for (int ow = pg; ow < OW; ow += 8) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = off[e] >= 0 ? slab[off[e] + 2 * ow] : 0.f;
    if (q < nq) st8<T>(Arow + (long long)ow * ldk + q * 8, v);
  }
}
extern "C" int avec_stem_im2col(int dtype, const float* video, void* A, long long clips, int T_, int H, int W, int ldk, hipStream_t st) {
  AVEC_CHECK_ARG(video && A && clips > 0 && T_ > 0 && H > 0 && W > 0 && (ldk == 248 || ldk == 256), "stem_im2col: bad arguments (ldk must be 248 or 256)");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  int WP = W + 6; while (WP % 32 != 10) ++WP;
  const size_t lds = (size_t)35 * WP * 4;
  AVEC_CHECK_ARG(lds <= 64 * 1024, "stem_im2col: frame width %d too large", W);
  const long long nb = clips * T_ * OH;
  AVEC_CHECK_ARG(nb < (1ll << 31), "stem_im2col: too many output rows");
  DISPATCH_T(dtype, hipLaunchKernelGGL(stem_im2col_kernel<T>, dim3((unsigned)nb), dim3(256), lds, st, video, (T*)A, T_, H, W, OH, OW, ldk, WP));
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_stem_pool_fwd(int dtype, const void* y, const float* ss, void* out, unsigned char* idx, void* ymax, long long frames, int H, int W, int C, hipStream_t st) {
  AVEC_CHECK_ARG(y && ss && out && idx && frames > 0 && H > 0 && W > 0 && C % 4 == 0 && frames * H * W * (C / 4) < (1ll << 31), "stem_pool_fwd: bad arguments");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1; long long n4 = frames * OH * OW * (C / 4); long long nb = (n4 + 255) / 256; if (nb > 8192) nb = 8192;
  DISPATCH_T(dtype, hipLaunchKernelGGL(stem_pool_fwd_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)y, ss, (T*)out, idx, (T*)ymax, frames, H, W, C, OH, OW));
  AVEC_LAUNCH_CHECK(); return 0;
}
// BatchNorm-backward statistics of the stem over the POOLED domain: the gradient of the BN output is non-zero only where a max-pool window picked its winner, and
// the forward pass kept that winner's pre-BN value (ymax): dstats[c] += sum dp, dstats[C + c] += sum dp * (ymax - mean) * rstd over the pooled elements with a winner.
// Reads dp + idx + ymax (0.5 GB at the bench shape) instead of the whole 0.8 GB conv output plus the gathered gradients.
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bwd_reduce_pooled_kernel(const T* __restrict__ dp, const unsigned char* __restrict__ idx, const T* __restrict__ ymax,
                                                                          const float* __restrict__ ss, float* dstats, long long P, int C, ColWs ws) {
  const Col8 m = col8_map(C);
  float part[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) part[0][e] = part[1][e] = 0.f;
  if (m.active) {
    const int c = m.l * 8;
    float mu[8], rs[8]; ld8<float>(ss + 2 * C + c, mu); ld8<float>(ss + 3 * C + c, rs);
    for (long long row = (long long)blockIdx.x * m.R + m.r; row < P; row += (long long)gridDim.x * m.R) {
      float g[8], v[8]; ld8<T>(dp + row * C + c, g); ld8<T>(ymax + row * C + c, v);
      const uint2 sel = *(const uint2*)(idx + row * C + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned q = ((e < 4 ? sel.x : sel.y) >> (8 * (e & 3))) & 255u;
        const float d = q != 255u ? g[e] : 0.f;
        part[0][e] += d; part[1][e] += d * (v[e] - mu[e]) * rs[e];
      }
    }
  }
  float* const dst[2] = {dstats, dstats + C};
  colreduce8_atomic<2>(part, dst, m, ws);
}
extern "C" int avec_stem_pool_bwd_reduce_pooled(int dtype, const void* dpool, const unsigned char* idx, const void* ymax, const float* ss, float* dstats,
                                                long long frames, int H, int W, int C, hipStream_t st) {
  AVEC_CHECK_ARG(dpool && idx && ymax && ss && dstats && frames > 0 && C % 8 == 0 && C <= 2048, "stem_pool_bwd_reduce_pooled: bad arguments (C %% 8 == 0, C <= 2048)");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long P = frames * OH * OW;
  ColWs ws; const unsigned nb8 = col8_cfg(P, C, 2, &ws, st);
  DISPATCH_T(dtype, hipLaunchKernelGGL(stem_pool_bwd_reduce_pooled_kernel<T>, dim3(nb8), dim3(256), 0, st, (const T*)dpool, idx, (const T*)ymax, ss, dstats, P, C, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {dstats, dstats + C}; return col_finalize(ws, 1, nb8, 2, C, dst, C, st); }
  return 0;
}

extern "C" int avec_stem_pool_bwd(int dtype, const void* dpool, const unsigned char* idx, const void* y, const float* ss, const float* gamma, float* dstats,
                                  const float* count_ptr, float count, int phase, void* dy, float* dgamma, float* dbeta, long long frames, int H, int W, int C, hipStream_t st) {
  AVEC_CHECK_ARG(dpool && idx && y && ss && gamma && dstats && (phase == 0 || dy) && frames > 0 && C % 4 == 0 && frames * H * W * (C / 4) < (1ll << 31), "stem_pool_bwd: bad arguments");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  if (C % 8 == 0 && C <= 2048) {
    if (phase == 0) {
      ColWs ws; const unsigned nb8 = col8_cfg(frames * H * W, C, 2, &ws, st);
      DISPATCH_T(dtype, hipLaunchKernelGGL(stem_pool_bwd_reduce8_kernel<T>, dim3(nb8), dim3(256), 0, st, (const T*)dpool, idx, (const T*)y, ss, dstats,
                                           frames, H, W, C, OH, OW, ws));
      AVEC_LAUNCH_CHECK();
      if (ws.partial) { float* const dst[2] = {dstats, dstats + C}; return col_finalize(ws, 1, nb8, 2, C, dst, C, st); }
      return 0;
    } else {
      long long n8 = frames * H * W * (C / 8); long long nb = (n8 + 255) / 256; if (nb > 16384) nb = 16384;
      DISPATCH_T(dtype, hipLaunchKernelGGL(stem_pool_bwd_apply8_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)dpool, idx, (const T*)y, ss, gamma, dstats, count_ptr, count,
                                           (T*)dy, dgamma, dbeta, frames, H, W, C, OH, OW));
    }
    AVEC_LAUNCH_CHECK(); return 0;
  }
  if (phase == 0) {
    dim3 grid = col_grid(frames * H * W, C); ColWs ws = col_ws_if(grid, 2, C, st);
    DISPATCH_T(dtype, hipLaunchKernelGGL(stem_pool_bwd_reduce_kernel<T>, grid, dim3(256), 0, st, (const T*)dpool, idx, (const T*)y, ss, dstats, frames, H, W, C, OH, OW, ws));
    AVEC_LAUNCH_CHECK();
    if (ws.partial) { float* const dst[2] = {dstats, dstats + C}; return col_finalize(ws, grid.x, grid.y, 2, 128, dst, C, st); }
    return 0;
  } else {
    long long n4 = frames * H * W * (C / 4); long long nb = (n4 + 255) / 256; if (nb > 8192) nb = 8192;
    DISPATCH_T(dtype, hipLaunchKernelGGL(stem_pool_bwd_apply_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)dpool, idx, (const T*)y, ss, gamma, dstats, count_ptr, count,
                                         (T*)dy, dgamma, dbeta, frames, H, W, C, OH, OW));
  }
  AVEC_LAUNCH_CHECK(); return 0;
}
