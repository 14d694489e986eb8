// Direct (no im2col) MFMA kernels for the visual stem Conv3d(1 -> 64, kernel (5,7,7), stride (1,2,2), padding (2,3,3))
// (nnet/networks.py:459-470, nnet/layers.py Conv3d), bf16.  Cin = 1 means an im2col row is 245 scattered scalars, so a generic
// implicit-GEMM loader cannot fetch 16-byte operand chunks; materialising the (6.2 M x 248) matrix costs 3 GB of HBM writes and two
// 3 GB reads per step (forward + weight gradient were both HBM-bound on it).  Here a workgroup stages the input it needs ONCE in LDS
// as bf16 (5 frames x the rows of half an output frame, zero padded) and builds MFMA operands from it with 2-byte LDS gathers:
//   forward : y[m][c] = bias[c] + sum_k A[m][k] W[c][k]   -- A fragment: lane = output pixel, 8 consecutive taps;  W held in registers
//   wgrad   : dW[c][k] += sum_m dy[m][c] A[m][k]          -- A^T fragment: lane = tap, 8 consecutive pixels;  dy tiles by ds_read_b64_tr_b16
// k = (kd*7 + kh)*7 + kw.  BatchNorm statistics of y are accumulated per workgroup and leave through the two-pass reduction workspace.
#include "vec.h"
#include "avec_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef short v4s_t __attribute__((ext_vector_type(4)));

static constexpr int S3_HALVES = 2;      // an output frame is processed as two bands of rows
static constexpr int S3_C = 64;          // output channels
static constexpr int S3_K = 245;         // taps

struct Stem3 { int T3, H, W, OH, OW, ohn, SH, WP; long long items; };     // ohn: output rows per band; SH: staged input rows per frame; WP: staged row pitch

__device__ __forceinline__ f32x16 mma16(const chunk16& a, const chunk16& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ chunk16 tr8s(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) v4s_t* lp_t;
  const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p1);
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  chunk16 f; f.w[0] = ua.x; f.w[1] = ua.y; f.w[2] = ub.x; f.w[3] = ub.y; return f;
}

// stage the band's input: slab[kd][r][x] (bf16) = video[clip][fr + kd - 2][2*oh0 - 3 + r][x - 3], zero outside the clip / frame.
// Loads are issued in batches of 8 independent (clamped, unconditional) accesses per thread before any is consumed: a load-store loop
// serialises on the global latency (measured 70 us per workgroup).
__device__ __forceinline__ void stage_slab(unsigned short* slab, const float* video, long long clip, int fr, int oh0, const Stem3& G) {
  const float* src = video + clip * (long long)G.T3 * G.H * G.W;
  const int nrow = 5 * G.SH, padw = G.WP - G.W;
  for (int idx = threadIdx.x; idx < nrow * padw; idx += 256) {            // left / right zero columns
    const int row = idx / padw, j = idx - row * padw; slab[row * G.WP + (j < 3 ? j : G.W + j)] = 0;
  }
  if ((G.W & 3) == 0 && (((size_t)video) & 15) == 0) {
    const int W4 = G.W >> 2, total = nrow * W4;
    for (int base = threadIdx.x; base < total; base += 256 * 8) {
      float4 v[8]; int dst[8]; bool rv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256; const int ii = idx < total ? idx : 0;
        const int row = ii / W4, c4 = ii - row * W4; const int kd = row / G.SH, r = row - kd * G.SH; const int it = fr + kd - 2, ih = 2 * oh0 - 3 + r;
        rv[u] = it >= 0 && it < G.T3 && ih >= 0 && ih < G.H;
        v[u] = *(const float4*)(src + ((long long)(rv[u] ? it : 0) * G.H + (rv[u] ? ih : 0)) * G.W + 4 * c4);
        dst[u] = idx < total ? row * G.WP + 3 + 4 * c4 : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (dst[u] < 0) continue;
        unsigned short* d = slab + dst[u];
        d[0] = rv[u] ? f32_to_bf16(v[u].x) : (unsigned short)0; d[1] = rv[u] ? f32_to_bf16(v[u].y) : (unsigned short)0;
        d[2] = rv[u] ? f32_to_bf16(v[u].z) : (unsigned short)0; d[3] = rv[u] ? f32_to_bf16(v[u].w) : (unsigned short)0;
      }
    }
  } else {
    const int total = nrow * G.W;
    for (int base = threadIdx.x; base < total; base += 256 * 8) {
      float v[8]; int dst[8]; bool rv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256; const int ii = idx < total ? idx : 0;
        const int row = ii / G.W, x = ii - row * G.W; const int kd = row / G.SH, r = row - kd * G.SH; const int it = fr + kd - 2, ih = 2 * oh0 - 3 + r;
        rv[u] = it >= 0 && it < G.T3 && ih >= 0 && ih < G.H;
        v[u] = src[((long long)(rv[u] ? it : 0) * G.H + (rv[u] ? ih : 0)) * G.W + x];
        dst[u] = idx < total ? row * G.WP + 3 + x : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) if (dst[u] >= 0) slab[dst[u]] = rv[u] ? f32_to_bf16(v[u]) : (unsigned short)0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem3_fwd_kernel(const float* __restrict__ video, const bf16* __restrict__ wsh, int ldw, const float* __restrict__ bias,
                                                           bf16* __restrict__ y, int want_stats, Stem3 G, ColWs ws, float* stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned short s3[];
  unsigned short* slab = s3;                                       // [5][SH][WP]
  const int slab_elems = (5 * G.SH * G.WP + 7) & ~7;
  unsigned short* stage = slab + slab_elems;                       // [4 waves][32 px][72]
  float* lsum = (float*)(stage + 4 * 32 * 72);                     // [2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, pl = lane & 31;
  const long long item = blockIdx.x; const int hh = (int)(item % S3_HALVES); const long long cf = item / S3_HALVES;
  const int fr = (int)(cf % G.T3); const long long clip = cf / G.T3;
  const int oh0 = hh * G.ohn; const int nrows = min(G.ohn, G.OH - oh0); const int npx = nrows * G.OW;
  if (tid < 128) lsum[tid] = 0.f;
  stage_slab(slab, video, clip, fr, oh0, G);
  // The reduction runs over (kd,kh) rows of 8 slots (7 taps kw + one zero-weight slot): the 8 operand elements of a lane are then 8
  // CONSECUTIVE bf16 of one staged row = four aligned 32-bit LDS reads (an element-by-element gather costs 8 reads + 8 selects + 4 packs).
  // k-step s covers rows 2s (lane half g = 0) and 2s + 1 (g = 1); w8 = [64][36 rows][8] with row 35 and slot 7 zero.
  chunk16 wf[18][2];
#pragma unroll
  for (int s = 0; s < 18; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[s][j] = ldg16(wsh + (long long)(32 * j + pl) * ldw + (2 * s + g) * 8);
  int rowoff[35];
#pragma unroll
  for (int r = 0; r < 35; ++r) rowoff[r] = ((r / 7) * G.SH + (r % 7)) * G.WP;
  const float b0 = bias ? bias[pl] : 0.f, b1 = bias ? bias[32 + pl] : 0.f;
  float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};
  __syncthreads();
  const long long mbase = (cf * G.OH + oh0) * (long long)G.OW;     // first output row of this band in y
  for (int t0 = 0; t0 < npx; t0 += 128) {
    const int p = t0 + 32 * wave + pl; const int pc = p < npx ? p : 0;
    const int ohl = pc / G.OW, ow = pc - ohl * G.OW;
    const unsigned short* pix = slab + (2 * ohl) * G.WP + 2 * ow;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    // operands of 3 k-steps are requested ahead of the MFMAs that consume them (one wave per SIMD: nothing else hides the LDS latency)
    chunk16 fa[18];
    auto ldfrag = [&](int s) {
      const int offA = rowoff[2 * s > 34 ? 34 : 2 * s], offB = rowoff[2 * s + 1 > 34 ? 34 : 2 * s + 1];
      const uint32_t* rp = (const uint32_t*)(pix + (g ? offB : offA));          // even element index: 4-byte aligned
      chunk16 f; f.w[0] = rp[0]; f.w[1] = rp[1]; f.w[2] = rp[2]; f.w[3] = rp[3]; return f;
    };
#pragma unroll
    for (int s = 0; s < 3; ++s) fa[s] = ldfrag(s);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      if (s + 3 < 18) fa[s + 3] = ldfrag(s + 3);
      asm volatile("" ::: "memory");
      acc[0] = mma16(fa[s], wf[s][0], acc[0]);
      acc[1] = mma16(fa[s], wf[s][1], acc[1]);
    }
    // epilogue: + bias, statistics, bf16, staged through LDS for 16-byte row stores
    unsigned short* stw = stage + wave * 32 * 72;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bb = j ? b1 : b0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        const float v = acc[j][r] + bb;
        if (t0 + 32 * wave + row < npx) { st1[j] += v; st2[j] += v * v; }
        stw[row * 72 + 32 * j + pl] = f32_to_bf16(v);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ch = it * 64 + lane, row = ch >> 3, part = ch & 7;
      if (t0 + 32 * wave + row < npx) *(chunk16*)(y + (mbase + t0 + 32 * wave + row) * S3_C + part * 8) = *(const chunk16*)(stw + row * 72 + part * 8);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (!want_stats) return;
#pragma unroll
  for (int j = 0; j < 2; ++j) { st1[j] += __shfl_xor(st1[j], 32, 64); st2[j] += __shfl_xor(st2[j], 32, 64); }
  if (g == 0) { atomicAdd(lsum + pl, st1[0]); atomicAdd(lsum + 32 + pl, st1[1]); atomicAdd(lsum + 64 + pl, st2[0]); atomicAdd(lsum + 96 + pl, st2[1]); }
  __syncthreads();
  if (tid < 128) {
    if (ws.partial) ws_slot(ws, 0, blockIdx.x, gridDim.x, 128)[tid] = lsum[tid];
    else atomicAdd(stats + tid, lsum[tid]);
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: persistent workgroups, each accumulates D[k][c] over its bands in registers
//   wave w owns taps [64 w, 64 w + 64) (2 tiles of 32) x all 64 channels (2 tiles)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void stem3_wgrad_kernel(const float* __restrict__ video, const bf16* __restrict__ dy, float* dw, Stem3 G, ColWs ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned short s3[];
  unsigned short* slab = s3;
  const int slab_elems = (5 * G.SH * G.WP + 7) & ~7;
  char* dyt = (char*)(slab + slab_elems);                          // [128 px][64 c] bf16, 128-byte rows, chunks swizzled by 4*((m>>1)&1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, pl = lane & 31;
  // this lane's two taps (A^T rows) and their slab offsets
  int offk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int k = 64 * wave + 32 * i + pl; const int r = k / 7 > 34 ? 34 : k / 7; offk[i] = ((r / 7) * G.SH + (r % 7)) * G.WP + k % 7; }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // transposed-read offsets of the dy tile: B operand (col c = 32 j + 16 (g4&1) + t, 8 consecutive pixels)
  const int g4 = lane >> 4, t = lane & 15;
  int offb[2][2];
#pragma unroll
  for (int hh2 = 0; hh2 < 2; ++hh2) {
    const int row = 8 * (g4 >> 1) + 4 * hh2 + (t >> 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int cb = 32 * j + 16 * (g4 & 1); offb[j][hh2] = row * 128 + ((((cb >> 3) + ((t & 3) >> 1)) ^ (4 * ((row >> 1) & 1))) << 4) + (t & 1) * 8; }
  }
  for (long long item = blockIdx.x; item < G.items; item += gridDim.x) {
    const int hh = (int)(item % S3_HALVES); const long long cf = item / S3_HALVES; const int fr = (int)(cf % G.T3); const long long clip = cf / G.T3;
    const int oh0 = hh * G.ohn; const int nrows = min(G.ohn, G.OH - oh0); const int npx = nrows * G.OW;
    const long long mbase = (cf * G.OH + oh0) * (long long)G.OW;
    __syncthreads();                                               // previous band's slab no longer read
    stage_slab(slab, video, clip, fr, oh0, G);
    for (int t0 = 0; t0 < npx; t0 += 128) {
      __syncthreads();                                             // slab staged / previous dy tile consumed
      // dy tile: 128 px x 8 chunks of 16 B -> 4 per thread; rows beyond the band are zero
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ch = u * 256 + tid, row = ch >> 3, pcx = ch & 7;
        chunk16 v; v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0u;
        if (t0 + row < npx) v = *(const chunk16*)(dy + (mbase + t0 + row) * S3_C + ((pcx ^ (4 * ((row >> 1) & 1))) << 3));
        *(chunk16*)(dyt + row * 128 + (pcx << 4)) = v;
      }
      __syncthreads();
      // this lane's first pixel of the tile (pixels of a step: 16 s + 8 g + e); (row, column) advance without divisions (OW >= 16 assumed by the host check)
      int p0 = t0 + 8 * g; int ohl0 = p0 / G.OW, ow0 = p0 - ohl0 * G.OW;
#pragma unroll
      for (int s = 0; s < 8; ++s) {                               // 16 pixels per step
        chunk16 fa[2];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          int pb[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            int owe = ow0 + e + q, ohe = ohl0; if (owe >= G.OW) { owe -= G.OW; ++ohe; }
            if (ohe >= nrows) { ohe = 0; owe = 0; }               // beyond the band: the dy rows are zero, any staged address will do
            pb[q] = (2 * ohe) * G.WP + 2 * owe;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[i].w[e >> 1] = (uint32_t)slab[pb[0] + offk[i]] | ((uint32_t)slab[pb[1] + offk[i]] << 16);
        }
        ow0 += 16; if (ow0 >= G.OW) { ow0 -= G.OW; ++ohl0; }
        const char* db = dyt + (16 * s) * 128;
        const chunk16 fb0 = tr8s(db + offb[0][0], db + offb[0][1]), fb1 = tr8s(db + offb[1][0], db + offb[1][1]);
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[i][0] = mma16(fa[i], fb0, acc[i][0]); acc[i][1] = mma16(fa[i], fb1, acc[i][1]); }
      }
    }
  }
  // D[k][c] -> dw[c][245] (fp32): through the workspace partial ([64][245] per workgroup) or atomics
  float* mine = ws.partial ? ws_slot(ws, 0, blockIdx.x, gridDim.x, S3_C * S3_K) : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 64 * wave + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * g, c = 32 * j + pl;
        if (k < S3_K) { if (mine) mine[c * S3_K + k] = acc[i][j][r]; else atomicAdd(dw + c * S3_K + k, acc[i][j][r]); }
      }
}

// ------------------------------------------------------------------------------------------------
static bool stem3_geom(Stem3& G, long long clips, int T_, int H, int W, size_t* slab_bytes) {
  G.T3 = T_; G.H = H; G.W = W; G.OH = (H - 1) / 2 + 1; G.OW = (W - 1) / 2 + 1; G.ohn = (G.OH + S3_HALVES - 1) / S3_HALVES;
  G.SH = 2 * (G.ohn - 1) + 7; G.WP = W + 6 + ((W + 6) & 1);       // even pitch: 4-byte aligned pairs
  G.items = clips * T_ * S3_HALVES;
  *slab_bytes = (size_t)((5 * G.SH * G.WP + 7) & ~7) * 2;
  return *slab_bytes <= 60 * 1024 && G.items < (1ll << 31);
}
template <typename K> static int s3_set_lds(K kern, size_t bytes) {
  static const void* done[4]; static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
  if (ndone < 4) done[ndone++] = (const void*)kern;
  return 0;
}

extern "C" int avec_stem3d_supported(long long clips, int T_, int H, int W) {
  Stem3 G; size_t sb; return clips > 0 && T_ > 0 && H > 6 && W >= 31 && stem3_geom(G, clips, T_, H, W, &sb) ? 1 : 0;     // OW >= 16: the pixel walk wraps at most once per step
}

extern "C" int avec_stem3d_fwd(const float* video, const void* w_shadow, int ldw, const float* bias, void* y, float* stats, long long clips, int T_, int H, int W, hipStream_t st) {
  AVEC_CHECK_ARG(video && w_shadow && y && ldw == 288, "stem3d_fwd: the weight must be repacked as bf16 [64][36][8] (ldw = 288; row 35 and slot 7 zero)");
  Stem3 G; size_t sb;
  AVEC_CHECK_ARG(stem3_geom(G, clips, T_, H, W, &sb), "stem3d_fwd: frame %dx%d too large for the LDS band (use avec_stem_im2col + avec_gemm_nt)", H, W);
  const size_t lds = sb + (size_t)4 * 32 * 72 * 2 + 128 * 4;
  if (int r = s3_set_lds(stem3_fwd_kernel, lds)) return r;
  ColWs ws = stats ? avec_reduce_ws((size_t)G.items * 128, st) : ColWs{nullptr};
  hipLaunchKernelGGL(stem3_fwd_kernel, dim3((unsigned)G.items), dim3(256), lds, st, video, (const bf16*)w_shadow, ldw, bias, (bf16*)y, stats ? 1 : 0, G, ws, stats);
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {stats, stats + S3_C}; return col_finalize(ws, 1, (unsigned)G.items, 2, S3_C, dst, S3_C, st); }
  return 0;
}

extern "C" int avec_stem3d_wgrad(const float* video, const void* dy, float* dw, long long clips, int T_, int H, int W, hipStream_t st) {
  AVEC_CHECK_ARG(video && dy && dw, "stem3d_wgrad: null pointer");
  Stem3 G; size_t sb;
  AVEC_CHECK_ARG(stem3_geom(G, clips, T_, H, W, &sb), "stem3d_wgrad: frame %dx%d too large for the LDS band", H, W);
  const size_t lds = sb + (size_t)128 * 128;
  if (int r = s3_set_lds(stem3_wgrad_kernel, lds)) return r;
  unsigned nb = 512; if ((long long)nb > G.items) nb = (unsigned)G.items;
  ColWs ws = avec_reduce_ws((size_t)nb * S3_C * S3_K, st);
  hipLaunchKernelGGL(stem3_wgrad_kernel, dim3(nb), dim3(256), lds, st, video, (const bf16*)dy, dw, G, ws);
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[1] = {dw}; return col_finalize(ws, 1, nb, 1, S3_C * S3_K, dst, S3_C * S3_K, st); }
  return 0;
}
