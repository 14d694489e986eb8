// MFMA GEMM family for gfx950 (CDNA4, wave64).
//
//   gemm_nt :  C[m][n] = epi( sum_k A[m][k] * W[n][k] )          (Linear fwd / dX, conv fwd / bwd-data)
//   gemm_tn :  O[i][j] += sum_m P[m][i] * Q[m][j]                 (weight gradients, split over m, fp32 atomics)
//
// A / Q operands are produced by "row loaders": plain row-major (with optional strided row remap and
// fp32->bf16 conversion), NHWC implicit-GEMM im2col (forward and transposed/backward-data), and the
// Cin=1 Conv3d stem gather.  One LDS image serves both dtypes: each tile row holds 128 bytes of K
// (64 bf16 / 32 fp32) + 16 bytes padding (row stride 144 B = 9 x 16 B => ds_read_b128 conflict-free).
// MFMA: v_mfma_f32_32x32x16_bf16 (bf16) / v_mfma_f32_32x32x2_f32 (fp32, exact fp32 for parity tests).
#include "common.h"
#include "avec_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static constexpr int BKB = 128;      // bytes of K per LDS tile row
static constexpr int LDS_ROW = 144;  // padded LDS row stride in bytes

enum { MODE_PLAIN = 0, MODE_CONV_FWD = 1, MODE_CONV_BWD = 2, MODE_STEM3D = 3 };

struct RowSrc {
  const void* ptr;
  long long ld;                       // plain: row stride in elements
  int rows_out, rows_in, step;        // plain: src_row = (m / rows_out) * rows_in + (m % rows_out) * step  (step<=1: identity)
  int H, W, C, KH, KW, stride, pad, OH, OW;  // conv geometry (see loaders)
  int T3;                             // stem: frames per clip
};

struct RowInfo { long long base; int a, b; int valid; };

// ---- row decomposition (constant across the K loop) ----
template <int MODE>
__device__ __forceinline__ RowInfo row_info(const RowSrc& s, long long m, long long M) {
  RowInfo r; r.valid = m < M; r.base = 0; r.a = 0; r.b = 0;
  if (!r.valid) return r;
  if (MODE == MODE_PLAIN) {
    long long row = m;
    if (s.step > 1) row = (m / s.rows_out) * (long long)s.rows_in + (m % s.rows_out) * (long long)s.step;
    r.base = row * s.ld;
  } else if (MODE == MODE_CONV_FWD) {      // m -> (img, oh, ow); source x[img][H][W][C]
    int ow = (int)(m % s.OW); long long t = m / s.OW; int oh = (int)(t % s.OH); long long img = t / s.OH;
    r.base = img * (long long)s.H * s.W * s.C; r.a = oh * s.stride - s.pad; r.b = ow * s.stride - s.pad;
  } else if (MODE == MODE_CONV_BWD) {      // m -> (img, ih, iw) over HxW; source dy[img][OH][OW][C]
    int iw = (int)(m % s.W); long long t = m / s.W; int ih = (int)(t % s.H); long long img = t / s.H;
    r.base = img * (long long)s.OH * s.OW * s.C; r.a = ih + s.pad; r.b = iw + s.pad;
  } else {                                  // stem: m -> (clip*T3 + t, oh, ow); source video[clip][T3][H][W] fp32
    int ow = (int)(m % s.OW); long long t = m / s.OW; int oh = (int)(t % s.OH); long long ft = t / s.OH;
    int fr = (int)(ft % s.T3); long long clip = ft / s.T3;
    r.base = clip * (long long)s.T3 * s.H * s.W; r.a = oh * 2 - 3; r.b = ow * 2 - 3; r.valid = 1 + fr;  // frame index kept in valid-1
  }
  return r;
}

// ---- fetch VEC consecutive K-elements of row r starting at k (zero outside) as one 16-byte chunk ----
template <typename T, int MODE, bool SRC_F32>
__device__ __forceinline__ chunk16 fetch_chunk(const RowSrc& s, const RowInfo& r, int k, int K) {
  constexpr int VEC = Elt<T>::VEC;
  chunk16 out; out.w[0] = out.w[1] = out.w[2] = out.w[3] = 0u;
  if (!r.valid || k >= K) return out;
  if (MODE == MODE_STEM3D) {
    const float* src = (const float*)s.ptr + r.base;
    int fr = r.valid - 1;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      int kk = k + e; float x = 0.f;
      if (kk < K) {
        int kw = kk % 7; int t2 = kk / 7; int kh = t2 % 7; int kd = t2 / 7;
        int it = fr + kd - 2, ih = r.a + kh, iw = r.b + kw;
        if (it >= 0 && it < s.T3 && ih >= 0 && ih < s.H && iw >= 0 && iw < s.W)
          x = src[((long long)it * s.H + ih) * s.W + iw];
      }
      v[e] = x;
    }
    if (sizeof(T) == 4) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) out.w[e] = __float_as_uint(v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < VEC / 2; ++e) out.w[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
    }
    return out;
  }
  long long off;
  if (MODE == MODE_PLAIN) {
    off = r.base + k;
  } else if (MODE == MODE_CONV_FWD) {
    int tap = k / s.C, c = k - tap * s.C; int kh = tap / s.KW, kw = tap - kh * s.KW;
    int ih = r.a + kh, iw = r.b + kw;
    if (ih < 0 || ih >= s.H || iw < 0 || iw >= s.W) return out;
    off = r.base + ((long long)ih * s.W + iw) * s.C + c;
  } else {  // MODE_CONV_BWD
    int tap = k / s.C, c = k - tap * s.C; int kh = tap / s.KW, kw = tap - kh * s.KW;
    int th = r.a - kh, tw = r.b - kw;
    if (th < 0 || tw < 0) return out;
    int oh = th / s.stride, ow = tw / s.stride;
    if (oh * s.stride != th || ow * s.stride != tw || oh >= s.OH || ow >= s.OW) return out;
    off = r.base + ((long long)oh * s.OW + ow) * s.C + c;
  }
  if (SRC_F32 && sizeof(T) == 2) {      // fp32 source converted to bf16 while staging
    const float* p = (const float*)s.ptr + off;
    float v[8];
    if (k + 8 <= K) {
      chunk16 lo = ldg16(p), hi = ldg16(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(lo.w[e]); v[4 + e] = __uint_as_float(hi.w[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (k + e < K) ? p[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) out.w[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
    return out;
  }
  const T* p = (const T*)s.ptr + off;
  if (k + VEC <= K && (sizeof(T) == 4 || ((off & 1) == 0))) return ldg16(p);   // vector path needs dword alignment
  if (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) if (k + e < K) out.w[e] = ((const uint32_t*)p)[e];
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) if (k + e < K) out.w[e >> 1] |= ((uint32_t)((const bf16_raw*)p)[e]) << (16 * (e & 1));
  }
  return out;
}

struct Epi {
  void* out; long long ldo; int out_f32;
  void* out_pre; long long ldpre;
  const float* bias;
  int act;                         // 0 none, 1 swish, 2 relu (forward activation)
  float drop_p; const unsigned long long* rng; unsigned stream;
  const void* res; long long ldres; float alpha; int res_act;
  const void* dact_z; long long ldz; int dact;   // multiply by act'(z) (1 swish, 2 relu)
  float* colsum;                   // += column sums of v (bias gradient)
  float* stats;                    // += [N] sum, [N] sum of squares of v (BatchNorm batch statistics)
};

struct GemmArgs { RowSrc a; const void* W; long long ldw; long long M; int N, K; Epi e; };

template <typename T> struct Mma;
template <> struct Mma<bf16> {
  __device__ static __forceinline__ void run(const chunk16& a, const chunk16& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(const chunk16& a, const chunk16& b, f32x16& c) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w[t]), __uint_as_float(b.w[t]), c, 0, 0, 0);
  }
};

template <typename T, int BM, int BN, int MODE, bool SRC_F32>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  constexpr int VEC = Elt<T>::VEC;
  constexpr int KE = BKB / (int)sizeof(T);   // K elements per tile
  constexpr int NCA = BM * 8 / 256, NCB = BN * 8 / 256;
  constexpr int MT = BM / 64, NT = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem; char* Bs = smem + BM * LDS_ROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long m0 = (long long)blockIdx.x * BM; const int n0 = blockIdx.y * BN;
  const int kc = (tid & 7) * VEC;            // this thread's K offset inside a tile
  const int r0 = tid >> 3;                   // first tile row handled by this thread (+32 per extra chunk)

  RowInfo ra[NCA]; RowInfo rb[NCB];
  RowSrc ws; ws.ptr = g.W; ws.ld = g.ldw; ws.step = 0; ws.rows_out = ws.rows_in = 1;
#pragma unroll
  for (int i = 0; i < NCA; ++i) ra[i] = row_info<MODE>(g.a, m0 + r0 + i * 32, g.M);
#pragma unroll
  for (int i = 0; i < NCB; ++i) rb[i] = row_info<MODE_PLAIN>(ws, n0 + r0 + i * 32, g.N);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  chunk16 ca[NCA], cb[NCB];
  const int KT = (g.K + KE - 1) / KE;
#pragma unroll
  for (int i = 0; i < NCA; ++i) ca[i] = fetch_chunk<T, MODE, SRC_F32>(g.a, ra[i], kc, g.K);
#pragma unroll
  for (int i = 0; i < NCB; ++i) cb[i] = fetch_chunk<T, MODE_PLAIN, false>(ws, rb[i], kc, g.K);

  const int frag_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
  for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
    for (int i = 0; i < NCA; ++i) *(chunk16*)(As + (r0 + i * 32) * LDS_ROW + (tid & 7) * 16) = ca[i];
#pragma unroll
    for (int i = 0; i < NCB; ++i) *(chunk16*)(Bs + (r0 + i * 32) * LDS_ROW + (tid & 7) * 16) = cb[i];
    __syncthreads();
    if (kt + 1 < KT) {
      const int k = (kt + 1) * KE + kc;
#pragma unroll
      for (int i = 0; i < NCA; ++i) ca[i] = fetch_chunk<T, MODE, SRC_F32>(g.a, ra[i], k, g.K);
#pragma unroll
      for (int i = 0; i < NCB; ++i) cb[i] = fetch_chunk<T, MODE_PLAIN, false>(ws, rb[i], k, g.K);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      chunk16 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *(const chunk16*)(As + (wm * (BM / 2) + i * 32) * LDS_ROW + frag_off + kk * 32);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *(const chunk16*)(Bs + (wn * (BN / 2) + j * 32) * LDS_ROW + frag_off + kk * 32);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (64-row passes) -> coalesced 4-wide rows with fused bias/act/dropout/residual/stats ----
  const Epi& e = g.e;
  constexpr int CLD = BN + 4;                 // fp32 row stride of the staged C tile
  constexpr int TPR = BN / 4;                 // threads per tile row
  float* Cs = (float*)smem;
  const int cg = (tid % TPR) * 4;             // this thread's 4 columns inside the tile (fixed across rows)
  const int col = n0 + cg;
  const bool vec_ok = (col + 3 < g.N);
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (e.bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) if (col + c < g.N) bias4[c] = e.bias[col + c];
  }
  float csum[4] = {0.f, 0.f, 0.f, 0.f}, csq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int pass = 0; pass < BM / 64; ++pass) {
    if (BM == 64 || wm == pass) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lr = (BM == 64 ? wm * 32 : 0) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Cs[lr * CLD + wn * (BN / 2) + j * 32 + (lane & 31)] = acc[i][j][r];
          }
    }
    __syncthreads();
#pragma unroll 1
    for (int lr = tid / TPR; lr < 64; lr += 256 / TPR) {
      const long long row = m0 + pass * 64 + lr;
      if (row >= g.M || col >= g.N) continue;
      float v[4];
      { const float4 t = *(const float4*)(Cs + lr * CLD + cg); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (col + c >= g.N) { v[c] = 0.f; continue; }
        float x = v[c] + bias4[c];
        if (e.out_pre) stf((T*)e.out_pre + row * e.ldpre + col + c, x);
        if (e.act == 1) x = swishf_(x); else if (e.act == 2) x = fmaxf(x, 0.f);
        if (e.drop_p > 0.f) x *= drop_scale(e.rng, e.stream, (unsigned long long)row * g.N + col + c, e.drop_p);
        if (e.dact) {
          const float z = ldf((const T*)e.dact_z + row * e.ldz + col + c);
          x *= (e.dact == 1) ? dswishf_(z) : (z > 0.f ? 1.f : 0.f);
        }
        csum[c] += x; csq[c] += x * x;
        x *= e.alpha;
        if (e.res) x += e.res_act ? ldf((const T*)e.res + row * e.ldres + col + c) : ((const float*)e.res)[row * e.ldres + col + c];
        v[c] = x;
      }
      if (e.out_f32) {
        float* o = (float*)e.out + row * e.ldo + col;
        if (vec_ok && ((e.ldo & 3) == 0)) *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
        else { for (int c = 0; c < 4; ++c) if (col + c < g.N) o[c] = v[c]; }
      } else {
        T* o = (T*)e.out + row * e.ldo + col;
        for (int c = 0; c < 4; ++c) if (col + c < g.N) stf(o + c, v[c]);
      }
    }
    __syncthreads();
  }
  if (e.colsum || e.stats) {
    // workgroup-level reduction in LDS (the staged C tile is dead now), then ONE atomic per column per workgroup;
    // BatchNorm statistics additionally spread over AVEC_STAT_REPLICAS copies to cut same-address contention.
    float* red = (float*)smem;                 // [2][BN]
    for (int c = tid; c < 2 * BN; c += 256) red[c] = 0.f;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) { atomicAdd(red + cg + c, csum[c]); atomicAdd(red + BN + cg + c, csq[c]); }
    __syncthreads();
    if (tid < BN && n0 + tid < g.N) {
      if (e.colsum) atomicAdd(e.colsum + n0 + tid, red[tid]);
      if (e.stats) {
        float* rep = e.stats + (long long)(blockIdx.x % AVEC_STAT_REPLICAS) * 2 * g.N;
        atomicAdd(rep + n0 + tid, red[tid]); atomicAdd(rep + g.N + n0 + tid, red[BN + tid]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN: O[i][j] += sum_m P[m][i] * Q[m][j];   P plain [M][I] (T), Q via row loader ([M][J], plain or im2col)
// LDS images are [i][m] / [j][m] (reduction index contiguous) filled by transposing stores.
// ------------------------------------------------------------------------------------------------
struct TnArgs { const void* P; long long ldp; RowSrc q; float* O; long long ldo; long long M; int I, J; int m_per_block; };

template <typename T>
__device__ __forceinline__ void store_transposed(char* S, int col0, int mloc, const chunk16& c) {
  // writes the VEC elements of chunk (tile columns col0..col0+VEC-1, reduction row mloc) into S[col][mloc]
  constexpr int VEC = Elt<T>::VEC;
  if (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) *(uint32_t*)(S + (col0 + e) * LDS_ROW + mloc * 4) = c.w[e];
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) *(bf16_raw*)(S + (col0 + e) * LDS_ROW + mloc * 2) = (bf16_raw)(c.w[e >> 1] >> (16 * (e & 1)));
  }
}

template <typename T, int BI, int BJ, int MODE, bool Q_F32>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs g) {
  constexpr int VEC = Elt<T>::VEC;
  constexpr int KE = BKB / (int)sizeof(T);     // reduction rows (m) per tile
  constexpr int CPR_I = BI / VEC, CPR_J = BJ / VEC;          // chunks per m-row
  constexpr int NCI = KE * CPR_I / 256, NCJ = KE * CPR_J / 256;  // chunks per thread
  constexpr int MT = BI / 64, NT = BJ / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ps = smem; char* Qs = smem + BI * LDS_ROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = blockIdx.x * BI, j0 = blockIdx.y * BJ;
  const long long mb = (long long)blockIdx.z * g.m_per_block;
  long long me = mb + g.m_per_block; if (me > g.M) me = g.M;

  RowSrc ps; ps.ptr = g.P; ps.ld = g.ldp; ps.step = 0; ps.rows_out = ps.rows_in = 1;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  chunk16 cp[NCI], cq[NCJ];
  auto fetch = [&](long long mt0) {
#pragma unroll
    for (int u = 0; u < NCI; ++u) {
      int c = tid + u * 256; int mloc = c / CPR_I, ci = c % CPR_I;
      long long m = mt0 + mloc;
      RowInfo r = row_info<MODE_PLAIN>(ps, m, me);
      cp[u] = fetch_chunk<T, MODE_PLAIN, false>(ps, r, i0 + ci * VEC, g.I);
    }
#pragma unroll
    for (int u = 0; u < NCJ; ++u) {
      int c = tid + u * 256; int mloc = c / CPR_J, cj = c % CPR_J;
      long long m = mt0 + mloc;
      RowInfo r = row_info<MODE>(g.q, m, me);
      cq[u] = fetch_chunk<T, MODE, Q_F32>(g.q, r, j0 + cj * VEC, g.J);
    }
  };
  const int frag_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
  if (mb < me) fetch(mb);
  for (long long mt0 = mb; mt0 < me; mt0 += KE) {
#pragma unroll
    for (int u = 0; u < NCI; ++u) { int c = tid + u * 256; store_transposed<T>(Ps, (c % CPR_I) * VEC, c / CPR_I, cp[u]); }
#pragma unroll
    for (int u = 0; u < NCJ; ++u) { int c = tid + u * 256; store_transposed<T>(Qs, (c % CPR_J) * VEC, c / CPR_J, cq[u]); }
    __syncthreads();
    if (mt0 + KE < me) fetch(mt0 + KE);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      chunk16 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *(const chunk16*)(Ps + (wm * (BI / 2) + i * 32) * LDS_ROW + frag_off + kk * 32);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *(const chunk16*)(Qs + (wn * (BJ / 2) + j * 32) * LDS_ROW + frag_off + kk * 32);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = j0 + wn * (BJ / 2) + j * 32 + (lane & 31);
    if (col >= g.J) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * (BI / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.I) atomicAdd(g.O + (long long)row * g.ldo + col, acc[i][j][r]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// host launchers (C ABI)
// ------------------------------------------------------------------------------------------------
static RowSrc make_src(const void* ptr, const avec_rows_t* d) {
  RowSrc s; s.ptr = ptr; s.ld = d->ld; s.rows_out = d->rows_out; s.rows_in = d->rows_in; s.step = d->step;
  s.H = d->H; s.W = d->W; s.C = d->C; s.KH = d->KH; s.KW = d->KW; s.stride = d->stride; s.pad = d->pad; s.OH = d->OH; s.OW = d->OW; s.T3 = d->T3;
  return s;
}

template <typename T, int BM, int BN>
static void launch_nt_mode(const GemmArgs& g, int mode, int src_f32, hipStream_t st) {
  dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)((g.N + BN - 1) / BN));
  size_t lds = (size_t)(BM + BN) * LDS_ROW;
#define L(MODE, F) hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, MODE, F>), grid, dim3(256), lds, st, g)
  if (mode == MODE_PLAIN) { if (src_f32) L(MODE_PLAIN, true); else L(MODE_PLAIN, false); }
  else if (mode == MODE_CONV_FWD) L(MODE_CONV_FWD, false);
  else if (mode == MODE_CONV_BWD) L(MODE_CONV_BWD, false);
  else L(MODE_STEM3D, true);
#undef L
}

template <typename T>
static void launch_nt(const GemmArgs& g, int mode, int src_f32, hipStream_t st) {
  // tile choice: big tiles only when they still fill the chip (256 CUs)
  long long t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128);
  if (g.N > 64 && t128 >= 384) launch_nt_mode<T, 128, 128>(g, mode, src_f32, st);
  else if (((g.M + 127) / 128) * ((g.N + 63) / 64) >= 384) launch_nt_mode<T, 128, 64>(g, mode, src_f32, st);
  else launch_nt_mode<T, 64, 64>(g, mode, src_f32, st);
}

extern "C" int avec_gemm_nt(int dtype, const void* A, const avec_rows_t* a_rows, int a_mode, int a_f32,
                            const void* W, long long ldw, long long M, int N, int K,
                            const avec_epilogue_t* ep, hipStream_t stream) {
  AVEC_CHECK_ARG(dtype == AVEC_F32 || dtype == AVEC_BF16, "gemm_nt: bad dtype %d", dtype);
  AVEC_CHECK_ARG(A && W && ep && ep->out && a_rows, "gemm_nt: null pointer");
  AVEC_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_nt: bad dims M=%lld N=%d K=%d", M, N, K);
  AVEC_CHECK_ARG(a_mode >= 0 && a_mode <= 3, "gemm_nt: bad a_mode %d", a_mode);
  AVEC_CHECK_ARG(!(a_mode != MODE_PLAIN && a_mode != MODE_STEM3D && a_rows->C % (dtype == AVEC_BF16 ? 8 : 4)), "gemm_nt: conv C=%d not a multiple of the vector width", a_rows->C);
  GemmArgs g; g.a = make_src(A, a_rows); g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
  Epi& e = g.e;
  e.out = ep->out; e.ldo = ep->ldo; e.out_f32 = ep->out_f32; e.out_pre = ep->out_pre; e.ldpre = ep->ldpre; e.bias = ep->bias;
  e.act = ep->act; e.drop_p = ep->drop_p; e.rng = (const unsigned long long*)ep->rng; e.stream = ep->rng_stream;
  e.res = ep->res; e.ldres = ep->ldres; e.alpha = ep->alpha; e.res_act = ep->res_act; e.dact_z = ep->dact_z; e.ldz = ep->ldz; e.dact = ep->dact;
  e.colsum = ep->colsum; e.stats = ep->stats;
  AVEC_CHECK_ARG(!(e.drop_p > 0.f) || e.rng, "gemm_nt: dropout without rng state");
  if (dtype == AVEC_BF16) launch_nt<bf16>(g, a_mode, a_f32, stream); else launch_nt<float>(g, a_mode, a_f32, stream);
  AVEC_LAUNCH_CHECK();
  return 0;
}

template <typename T, int BI, int BJ>
static void launch_tn_tile(TnArgs g, int mode, int q_f32, hipStream_t st) {
  constexpr int KE = BKB / (int)sizeof(T);
  int tiles = ((g.I + BI - 1) / BI) * ((g.J + BJ - 1) / BJ);
  long long ksteps = (g.M + KE - 1) / KE;
  long long split = (1024 + tiles - 1) / tiles; if (split > ksteps) split = ksteps; if (split < 1) split = 1;
  long long per = ((ksteps + split - 1) / split) * KE;
  split = (g.M + per - 1) / per;
  g.m_per_block = (int)per;
  dim3 grid((g.I + BI - 1) / BI, (g.J + BJ - 1) / BJ, (unsigned)split);
  size_t lds = (size_t)(BI + BJ) * LDS_ROW;
#define L(MODE, F) hipLaunchKernelGGL((gemm_tn_kernel<T, BI, BJ, MODE, F>), grid, dim3(256), lds, st, g)
  if (mode == MODE_PLAIN) { if (q_f32) L(MODE_PLAIN, true); else L(MODE_PLAIN, false); }
  else if (mode == MODE_CONV_FWD) L(MODE_CONV_FWD, false);
  else L(MODE_STEM3D, true);
#undef L
}

extern "C" int avec_gemm_tn(int dtype, const void* P, long long ldp, const void* Q, const avec_rows_t* q_rows, int q_mode, int q_f32,
                            float* O, long long ldo, long long M, int I, int J, hipStream_t stream) {
  AVEC_CHECK_ARG(dtype == AVEC_F32 || dtype == AVEC_BF16, "gemm_tn: bad dtype %d", dtype);
  AVEC_CHECK_ARG(P && Q && O && q_rows, "gemm_tn: null pointer");
  AVEC_CHECK_ARG(M > 0 && I > 0 && J > 0, "gemm_tn: bad dims");
  AVEC_CHECK_ARG(q_mode == MODE_PLAIN || q_mode == MODE_CONV_FWD || q_mode == MODE_STEM3D, "gemm_tn: bad q_mode %d", q_mode);
  TnArgs g; g.P = P; g.ldp = ldp; g.q = make_src(Q, q_rows); g.O = O; g.ldo = ldo; g.M = M; g.I = I; g.J = J; g.m_per_block = 0;
  bool big = (I >= 128 && J >= 128);
  if (dtype == AVEC_BF16) { if (big) launch_tn_tile<bf16, 128, 128>(g, q_mode, q_f32, stream); else launch_tn_tile<bf16, 64, 64>(g, q_mode, q_f32, stream); }
  else { if (big) launch_tn_tile<float, 128, 128>(g, q_mode, q_f32, stream); else launch_tn_tile<float, 64, 64>(g, q_mode, q_f32, stream); }
  AVEC_LAUNCH_CHECK();
  return 0;
}
