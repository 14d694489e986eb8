// bf16 MFMA kernels for the relative-position attention (RelPos1dMultiHeadAttention.forwardQKV, nnet/attentions.py:280-323;
// rel_to_abs :234-278).  One workgroup = 2 waves = 64 queries of one (batch, head); every wave owns 32 queries and sees all keys.
//
// Everything is computed TRANSPOSED so that a query is a lane (column of the 32x32 MFMA result) and keys run over registers:
//     S^T[j][i]   = K_j . Q_i                          mfma(A = K rows, B = Q rows)
//     R^T[e][i]   = E_e . Q_i   (e over the window)    mfma(A = E rows, B = Q rows);   S^T[j][i] += R^T[j + 31 - il][i]   ("skew")
//     softmax over j = over registers (+ one cross-half shuffle), P^T stays in registers as the B operand of
//     O^T[c][i]   = sum_j V^T[c][j] P^T[j][i]          mfma(A = V^T via ds_read_b64_tr_b16, B = P^T registers)
// The reference materialises Q.E^T as (B,H,T,2T-1) and re-indexes it with pad/reshape; here the re-indexing is the skewed read of a
// 2-tile fp32 ring in LDS (row e = j + 31 - il of the wave's window, conflict-free: bank = il).
// LDS images of K, V, E, Q are [row][pitch] bf16 with pitch 128 B (d <= 64) or 256 B (d <= 128) and 16-byte chunks XOR-swizzled by a
// bijection of the row bits chosen so that BOTH the ds_read_b128 operand reads (lanes = rows) and the transposed reads (4 rows x 32 B
// blocks) are bank-conflict free:  pitch 128: chunk ^= ((row>>1)&1)<<2 | (row>>2)&3;   pitch 256: chunk ^= (row&3)<<2 | (row>>2)&3.
#include "attention.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef short v4s_t __attribute__((ext_vector_type(4)));

static constexpr int NTMAX = 7;      // key tiles of 32 held in registers (T <= 224)
static constexpr int CTMAX = 3;      // channel tiles of 32 (d <= 96)

template <int PITCH> __device__ __forceinline__ int aswz(int row) {
  return PITCH == 128 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : (((row & 3) << 2) | ((row >> 2) & 3));
}
__device__ __forceinline__ f32x16 mma(const chunk16& a, const chunk16& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ chunk16 tr8(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) v4s_t* lp_t;
  const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p1);
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  chunk16 f; f.w[0] = ua.x; f.w[1] = ua.y; f.w[2] = ub.x; f.w[3] = ub.y; return f;
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16); }

// stage `nrows` rows x d channels of a bf16 matrix into the swizzled LDS image (zero outside [0, limit) and beyond d)
template <int PITCH, int NTHR>
__device__ __forceinline__ void stage_rows(char* dst, const bf16* src, long long ld, int row0, int nrows, int limit, int d, int dpad) {
  const int CH = dpad >> 3;
  const bool al4 = ((((size_t)src) | ((size_t)ld * 2)) & 3) == 0;
  for (int idx = threadIdx.x; idx < nrows * CH; idx += NTHR) {
    const int r = idx / CH, c = idx - r * CH; const int gr = row0 + r;
    chunk16 v; v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0u;
    if (gr >= 0 && gr < limit) {
      const bf16* p = src + (long long)gr * ld + c * 8;
      if (c * 8 + 8 <= d && al4) v = ldg16(p);
      else {
        uint32_t h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (c * 8 + e < d) ? (uint32_t)p[e].v : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) v.w[e] = h[2 * e] | (h[2 * e + 1] << 16);
      }
    }
    *(chunk16*)(dst + r * PITCH + ((c ^ aswz<PITCH>(r)) << 4)) = v;
  }
}

struct MfmaGeom { int Tp, NT, NE, dpad, DKS, CT; size_t offK, offV, offE, offQ, offS, total; };
static MfmaGeom mfma_geom(int T, int d, int pitch, bool bwd) {
  MfmaGeom g; g.Tp = (T + 31) / 32 * 32; g.NT = g.Tp / 32; g.NE = g.Tp + 64; g.dpad = (d + 15) / 16 * 16; g.DKS = g.dpad / 16; g.CT = (d + 31) / 32;
  g.offK = 0; g.offV = g.offK + (size_t)g.Tp * pitch; g.offE = g.offV + (size_t)g.Tp * pitch; g.offQ = g.offE + (size_t)g.NE * pitch;
  g.offS = g.offQ + (size_t)64 * pitch * (bwd ? 2 : 1);            // bwd: Q and dO tiles
  g.total = g.offS + (size_t)2 * 8192;                             // per wave: 2 x [32][32] fp32 skew ring (also the output staging)
  return g;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int PITCH>
__global__ __launch_bounds__(128) void attn_mfma_fwd_kernel(AttnArgs a, MfmaGeom G) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* Ks = sm + G.offK; char* Vs = sm + G.offV; char* Es = sm + G.offE; char* Qs = sm + G.offQ;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hf = lane >> 5, il = lane & 31;
  float* ring = (float*)(sm + G.offS + w * 8192);
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int Tn = a.T, d = a.d, i0 = blockIdx.x * 64;
  const bf16* qp = (const bf16*)a.q + (long long)b * Tn * a.ld + h * d;
  const bf16* kp = (const bf16*)a.k + (long long)b * Tn * a.ld + h * d;
  const bf16* vp = (const bf16*)a.v + (long long)b * Tn * a.ld + h * d;
  const bf16* ep = (const bf16*)a.e + h * d;
  stage_rows<PITCH, 128>(Ks, kp, a.ld, 0, G.Tp, Tn, d, G.dpad);
  stage_rows<PITCH, 128>(Vs, vp, a.ld, 0, G.Tp, Tn, d, G.dpad);
  stage_rows<PITCH, 128>(Es, ep, a.lde, (Tn - 1) - (i0 + 63), G.NE, 2 * Tn - 1, d, G.dpad);
  stage_rows<PITCH, 128>(Qs, qp, a.ld, i0, 64, Tn, d, G.dpad);
  __syncthreads();

  const int i = i0 + 32 * w + il;                        // this lane's query
  const int swr = aswz<PITCH>(il);                       // swizzle of operand row (tile base is a multiple of 32: same bits)
  const char* qrow = Qs + (32 * w + il) * PITCH;
  const int klen = a.lens ? (int)(a.lens[b] / a.len_div) : Tn;
  const bool row_masked = i >= a.q_full;

  f32x16 S[NTMAX];
#pragma unroll
  for (int jt = 0; jt < NTMAX; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) S[jt][r] = 0.f;

  // R^T tile et -> ring slot et & 1 (rows e_local = 32 et + ..., this wave's window starts 32 (1 - w) rows into the staged E window)
  auto rel_tile = [&](int et) {
    f32x16 R;
#pragma unroll
    for (int r = 0; r < 16; ++r) R[r] = 0.f;
    const char* erow = Es + (32 * et + 32 * (1 - w) + il) * PITCH;
    for (int kk = 0; kk < G.DKS; ++kk) {
      const chunk16 fa = *(const chunk16*)(erow + (((2 * kk + hf) ^ swr) << 4));
      const chunk16 fb = *(const chunk16*)(qrow + (((2 * kk + hf) ^ swr) << 4));
      R = mma(fa, fb, R);
    }
    float* slot = ring + (et & 1) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) slot[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + il] = R[r];
  };
  rel_tile(0);
#pragma unroll
  for (int jt = 0; jt < NTMAX; ++jt) {
    if (jt < G.NT) {
      const char* krow = Ks + (32 * jt + il) * PITCH;
      for (int kk = 0; kk < G.DKS; ++kk) {
        const chunk16 fa = *(const chunk16*)(krow + (((2 * kk + hf) ^ swr) << 4));
        const chunk16 fb = *(const chunk16*)(qrow + (((2 * kk + hf) ^ swr) << 4));
        S[jt] = mma(fa, fb, S[jt]);
      }
      rel_tile(jt + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the ring writes of this wave are visible to its own reads (LDS is in-order per wave)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jl = (r & 3) + 8 * (r >> 2) + 4 * hf;            // key within the tile
        const int el = 32 * jt + jl + 31 - il;                     // window row; ring holds rows [32 jt, 32 jt + 64)
        S[jt][r] += ring[(el & 63) * 32 + il];
      }
      asm volatile("" ::: "memory");
    }
  }
  // scale, masks, softmax over keys (registers + the other half-wave)
  float mx = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < NTMAX; ++jt) {
    if (jt < G.NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * hf;
        float s = S[jt][r] * a.scale;
        if (row_masked || j >= klen) s += -1e9f;
        s = j < Tn ? s : -INFINITY;
        S[jt][r] = s; mx = fmaxf(mx, s);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float lsum = 0.f;
#pragma unroll
  for (int jt = 0; jt < NTMAX; ++jt) {
    if (jt < G.NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float p = __expf(S[jt][r] - mx); S[jt][r] = p; lsum += p; }
    }
  }
  lsum += __shfl_xor(lsum, 32, 64);
  if (hf == 0 && i < Tn) { a.lse[((long long)bh * Tn + i) * 2] = mx; a.lse[((long long)bh * Tn + i) * 2 + 1] = lsum; }

  // O^T = V^T P^T
  f32x16 O[CTMAX];
#pragma unroll
  for (int ct = 0; ct < CTMAX; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[ct][r] = 0.f;
  const int g4 = lane >> 4, t = lane & 15;
  int offv[CTMAX][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int key = 8 * hh + 4 * (g4 >> 1) + (t >> 2);   // + 32 jt + 16 s: the swizzle bits of the key do not change
#pragma unroll
    for (int ct = 0; ct < CTMAX; ++ct) {
      const int cb = 32 * ct + 16 * (g4 & 1);
      offv[ct][hh] = key * PITCH + ((((cb >> 3) + ((t & 3) >> 1)) ^ aswz<PITCH>(key)) << 4) + (t & 1) * 8;
    }
  }
#pragma unroll
  for (int jt = 0; jt < NTMAX; ++jt) {
    if (jt < G.NT) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        chunk16 pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) pb.w[e] = pack2(S[jt][8 * s + 2 * e], S[jt][8 * s + 2 * e + 1]);
        const char* vb = Vs + (32 * jt + 16 * s) * PITCH;
#pragma unroll
        for (int ct = 0; ct < CTMAX; ++ct) if (ct < G.CT) O[ct] = mma(tr8(vb + offv[ct][0], vb + offv[ct][1]), pb, O[ct]);
      }
    }
  }
  // stage O (bf16, [il][c]) through this wave's ring, then coalesced row stores
  const float inv = 1.f / lsum;
  bf16* ost = (bf16*)ring; const int OP = G.dpad + 4;          // 8 KB ring >= 32 * (128 + 4) * 2 B
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int ct = 0; ct < CTMAX; ++ct) {
    if (ct < G.CT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * ct + 8 * q + 4 * hf;
        uint2 u; u.x = pack2(O[ct][4 * q] * inv, O[ct][4 * q + 1] * inv); u.y = pack2(O[ct][4 * q + 2] * inv, O[ct][4 * q + 3] * inv);
        if (c0 < G.dpad) *(uint2*)(ost + il * OP + c0) = u;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  bf16* op = (bf16*)a.o + ((long long)b * Tn + i0 + 32 * w) * a.ldo + h * d;
  const int nrow = min(32, Tn - (i0 + 32 * w));
  const bool pair = (d & 1) == 0 && (((size_t)op | ((size_t)a.ldo * 2)) & 3) == 0;
  if (pair) {
    const int dh = d >> 1;
    for (int idx = lane; idx < nrow * dh; idx += 64) { const int r = idx / dh, c = (idx - r * dh) * 2; *(uint32_t*)(op + (long long)r * a.ldo + c) = *(const uint32_t*)(ost + r * OP + c); }
  } else {
    for (int idx = lane; idx < nrow * d; idx += 64) { const int r = idx / d, c = idx - r * d; op[(long long)r * a.ldo + c] = ost[r * OP + c]; }
  }
}

static const bool g_mfma_off = getenv("AVEC_NO_MFMA_ATTN") != nullptr;

template <typename K> static int mfma_set_lds(K kern, size_t bytes) {
  static const void* done[8]; static size_t done_bytes[8]; static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern && done_bytes[i] >= bytes) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
  if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
  if (ndone < 8) { done[ndone] = (const void*)kern; done_bytes[ndone] = 160 * 1024; ++ndone; }
  return 0;
}

int attn_mfma_fwd(const AttnArgs& a, hipStream_t st) {
  if (g_mfma_off || a.mask || a.d > 96 || a.T > 32 * NTMAX) return 1;
  const int pitch = a.d <= 64 ? 128 : 256;
  const MfmaGeom G = mfma_geom(a.T, a.d, pitch, false);
  if (G.total > 160 * 1024) return 1;
  dim3 grid((a.T + 63) / 64, a.B * a.H);
  if (pitch == 128) { if (int r = mfma_set_lds(attn_mfma_fwd_kernel<128>, G.total)) return r; hipLaunchKernelGGL(attn_mfma_fwd_kernel<128>, grid, dim3(128), G.total, st, a, G); }
  else { if (int r = mfma_set_lds(attn_mfma_fwd_kernel<256>, G.total)) return r; hipLaunchKernelGGL(attn_mfma_fwd_kernel<256>, grid, dim3(128), G.total, st, a, G); }
  AVEC_LAUNCH_CHECK(); return 0;
}

int attn_mfma_bwd_rows(const AttnArgs& a, hipStream_t st) { (void)a; (void)st; return 1; }
