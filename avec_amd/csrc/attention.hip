// Relative-position multi-head self-attention (RelPos1dMultiHeadAttention.forwardQKV,
// nnet/attentions.py:280-323) fused per (batch, head):
//     scores[i][j] = (Q_i.K_j + Q_i.E_{i-j}) / sqrt(d)  (+ -1e9 on masked keys)  -> softmax -> P.V
// The reference materialises Q.E^T as (B,H,T,2T-1) and re-indexes it with the pad/reshape trick
// (rel_to_abs, :234-278); here E rows are addressed directly: row r = (T-1) - (i-j).  Keys are
// streamed in tiles of 64 through LDS with an online softmax, so nothing of size TxT is stored;
// the backward recomputes probabilities from the saved log-sum-exp.  All arithmetic is fp32 VALU
// (the attention bmm's are 0.5 % of the model's FLOPs); I/O is the activation dtype.
//   fwd  : grid (ceil(T/32), B*H), lanes = keys (scores) then lanes = channels (P.V)
//   bwd1 : dQ, same tiling as fwd
//   bwd2 : dK, dV, dE; grid (key tiles, B*H, query segments of 64), lanes = keys, register accumulators
#include "attention.h"

static constexpr int TQ = 32;    // query rows per workgroup (fwd / bwd1): 4 waves x RPW rows
static constexpr int RPW = TQ / 4;
static constexpr int TK = 64;    // keys per tile = one per lane
static constexpr int QC = 32;    // queries staged per chunk in the column pass (dK/dV/dE)
static constexpr int NCS = 3;    // output channels per lane in the row passes (lane, lane + 64, lane + 128): head widths up to 192 (grouped attention: 3 x 45 + 1)

// cooperative load of `rows` rows x d channels (act dtype -> fp32 LDS, zero outside [0, limit)).
// VW elements per access: 16 B when the head width allows it (d % 8 == 0 for bf16, d % 4 == 0 for fp32), 4/8 B for even d,
// scalar otherwise (d = 45).  Threads are laid out as (row, chunk) with a power-of-two chunk count, so no integer division.
template <typename T, int VW>
__device__ __forceinline__ void load_vec(const T* p, float* v) {
  constexpr int NB = VW * (int)sizeof(T);
  if constexpr (NB < 4) { v[0] = ldf(p); }
  else {
    uint32_t w[NB / 4];
    if constexpr (NB == 16) { const chunk16 c = ldg16(p); w[0] = c.w[0]; w[1] = c.w[1]; w[2] = c.w[2]; w[3] = c.w[3]; }
    else {
#pragma unroll
      for (int k = 0; k < NB / 4; ++k) w[k] = ((const uint32_t*)p)[k];
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int k = 0; k < NB / 4; ++k) v[k] = __uint_as_float(w[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NB / 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    }
  }
}
template <typename T, int VW>
__device__ __forceinline__ void load_rows_v(float* dst, int DP, const T* src, long long ld, int row0, int rows, int limit, int d) {
  const int cpr = d / VW;
  int sh = 0; while ((1 << sh) < cpr && sh < 8) ++sh;
  const int tpr = 1 << sh, rl = threadIdx.x >> sh, cl = threadIdx.x & (tpr - 1), rpp = 256 >> sh;
  for (int r = rl; r < rows; r += rpp) {
    const int gr = row0 + r; const bool ok = gr >= 0 && gr < limit;
    const T* rp = src + (long long)(ok ? gr : 0) * ld;
    for (int cc = cl; cc < cpr; cc += tpr) {
      float v[VW];
      load_vec<T, VW>(rp + cc * VW, v);
#pragma unroll
      for (int k = 0; k < VW; ++k) dst[r * DP + cc * VW + k] = ok ? v[k] : 0.f;
    }
  }
}
template <typename T>
__device__ __forceinline__ void load_rows(float* dst, int DP, const T* src, long long ld, int row0, int rows, int limit, int d) {
  constexpr int V16 = 16 / (int)sizeof(T);
  const bool al4 = ((ld * (long long)sizeof(T)) & 3) == 0 && (((size_t)src) & 3) == 0;
  if (d % V16 == 0 && al4) load_rows_v<T, V16>(dst, DP, src, ld, row0, rows, limit, d);
  else if (d % 2 == 0 && al4) load_rows_v<T, 2>(dst, DP, src, ld, row0, rows, limit, d);
  else load_rows_v<T, 1>(dst, DP, src, ld, row0, rows, limit, d);
}

// ------------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int d = a.d, DP = d | 1, Tn = a.T, Tk = a.Tk;
  float* Ks = sm; float* Vs = Ks + TK * DP; float* Es = Vs + TK * DP; float* Qs = Es + (TQ + TK - 1) * DP;
  float* Gs = Qs + TQ * DP;                  // BWD: dO rows
  float* Ps = Gs + (BWD ? TQ * DP : 0);      // [4][64]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int i0 = blockIdx.x * TQ;
  const T* qp = (const T*)a.q + (long long)b * Tn * a.ld + h * d;
  const T* kp = (const T*)a.k + (long long)b * Tk * a.ld + h * d;
  const T* vp = (const T*)a.v + (long long)b * Tk * a.ld + h * d;
  const T* ep = (const T*)a.e + h * d;
  load_rows<T>(Qs, DP, qp, a.ld, i0, TQ, Tn, d);
  if (BWD) load_rows<T>(Gs, DP, (const T*)a.dout + (long long)b * Tn * a.ldo + h * d, a.ldo, i0, TQ, Tn, d);
  __syncthreads();

  float m_run[RPW], l_run[RPW], acc[RPW][NCS], Li[RPW], Il[RPW], dl[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) { m_run[rr] = -INFINITY; l_run[rr] = 0.f; for (int u = 0; u < NCS; ++u) acc[rr][u] = 0.f; Li[rr] = 0.f; Il[rr] = 0.f; dl[rr] = 0.f; }
  if (BWD) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int ri = w * RPW + rr, i = i0 + ri;
      float s = 0.f;
      if (i < Tn) {
        const T* op = (const T*)a.o + ((long long)b * Tn + i) * a.ldo + h * d;
        for (int c = lane; c < d; c += 64) s += Gs[ri * DP + c] * ldf(op + c);
        Li[rr] = a.lse[((long long)bh * Tn + i) * 2]; Il[rr] = 1.f / a.lse[((long long)bh * Tn + i) * 2 + 1];
      }
      dl[rr] = wave_sum(s);
    }
  }

  for (int j0 = 0; j0 < Tk; j0 += TK) {
    __syncthreads();
    load_rows<T>(Ks, DP, kp, a.ld, j0, TK, Tk, d);
    load_rows<T>(Vs, DP, vp, a.ld, j0, TK, Tk, d);
    const int rbase = (Tn - 1) - (i0 + TQ - 1) + j0;
    load_rows<T>(Es, DP, ep, a.lde, rbase, TQ + TK - 1, Tk + Tn - 1, d);
    __syncthreads();
    const int j = j0 + lane;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int ri = w * RPW + rr, i = i0 + ri;
      const bool iv = i < Tn, jv = j < Tk;
      const float* qrow = Qs + ri * DP; const float* krow = Ks + lane * DP; const float* erow = Es + (TQ - 1 - ri + lane) * DP;
      float s = 0.f, dp = 0.f;
      if (BWD) { const float* grow = Gs + ri * DP; const float* vrow = Vs + lane * DP;
        for (int c = 0; c < d; ++c) { s += qrow[c] * (krow[c] + erow[c]); dp += grow[c] * vrow[c]; } }
      else { for (int c = 0; c < d; ++c) s += qrow[c] * (krow[c] + erow[c]); }
      s *= a.scale;
      if (iv && jv && !key_keep<T>(a, b, i, j)) s += -1e9f;
      float pval;
      if (!BWD) {
        float sm_ = (iv && jv) ? s : -INFINITY;
        float tmax = wave_max(sm_);
        float m_new = fmaxf(m_run[rr], tmax);
        float alpha = (m_run[rr] == -INFINITY) ? 0.f : __expf(m_run[rr] - m_new);
        pval = (iv && jv) ? __expf(s - m_new) : 0.f;
        l_run[rr] = l_run[rr] * alpha + wave_sum(pval);
        for (int u = 0; u < NCS; ++u) acc[rr][u] *= alpha;
        m_run[rr] = m_new;
      } else {
        float p = (iv && jv) ? __expf(s - Li[rr]) * Il[rr] : 0.f;
        pval = p * (dp - dl[rr]) * a.scale;   // dS
        if (iv && jv) {
          const long long o = ((long long)bh * Tn + i) * a.ldt + j; stf((T*)a.pbuf + o, p); stf((T*)a.dsbuf + o, pval);
          if (a.dsrel) stf((T*)a.dsrel + ((long long)h * a.B * Tn + (long long)b * Tn + i) * a.ldr + (j + Tn - 1 - i), pval);
        }
      }
      Ps[w * 64 + lane] = pval;
      __syncthreads();
      if (!BWD) {
        for (int jj = 0; jj < TK; ++jj) {
          const float pj = Ps[w * 64 + jj];
#pragma unroll
          for (int u = 0; u < NCS; ++u) if (lane + 64 * u < d) acc[rr][u] += pj * Vs[jj * DP + lane + 64 * u];
        }
      } else {
        const float* ebase = Es + (TQ - 1 - ri) * DP;
        for (int jj = 0; jj < TK; ++jj) {
          const float dsj = Ps[w * 64 + jj];
#pragma unroll
          for (int u = 0; u < NCS; ++u) if (lane + 64 * u < d) acc[rr][u] += dsj * (Ks[jj * DP + lane + 64 * u] + ebase[jj * DP + lane + 64 * u]);
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int i = i0 + w * RPW + rr;
    if (i >= Tn) continue;
    if (!BWD) {
      const float inv = 1.f / l_run[rr];
      T* op = (T*)a.o + ((long long)b * Tn + i) * a.ldo + h * d;
#pragma unroll
      for (int u = 0; u < NCS; ++u) if (lane + 64 * u < d) stf(op + lane + 64 * u, acc[rr][u] * inv);
      if (lane == 0) { a.lse[((long long)bh * Tn + i) * 2] = m_run[rr]; a.lse[((long long)bh * Tn + i) * 2 + 1] = l_run[rr]; }
    } else {
      T* op = (T*)a.dq + ((long long)b * Tn + i) * a.lddq + h * d;
#pragma unroll
      for (int u = 0; u < NCS; ++u) if (lane + 64 * u < d) stf(op + lane + 64 * u, acc[rr][u]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// column pass of the backward: from the stored P / dS matrices,
//   key tiles   (blockIdx.x <  ktiles): lane = key j     dK_j = sum_i dS_ij q_i,   dV_j = sum_i P_ij dO_i
//   rel tiles   (blockIdx.x >= ktiles): lane = E row r   dE_r = sum_i dS_{i, i-(T-1)+r} q_i        (the skewed sum of rel_to_abs)
// register accumulators per lane, q_i / dO_i broadcast from LDS, 4 waves split the queries; no atomics in the loop.
// ------------------------------------------------------------------------------------------------
template <typename T, int DPAD>
__global__ __launch_bounds__(256) void attn_bwd_cols_kernel(AttnArgs a, int ktiles) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int d = a.d, DP = d | 1, Tn = a.T;
  float* Qs = sm; float* Gs = Qs + QC * DP; float* R0 = Gs + QC * DP; float* R1 = R0 + 64 * DP;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const bool rel = (int)blockIdx.x >= ktiles;
  const int c0 = (rel ? (int)blockIdx.x - ktiles : (int)blockIdx.x) * 64;
  const int col = c0 + lane;
  const T* qp = (const T*)a.q + (long long)b * Tn * a.ld + h * d;
  const T* gp = (const T*)a.dout + (long long)b * Tn * a.ldo + h * d;
  const T* P = (const T*)a.pbuf + (long long)bh * Tn * a.ldt; const T* S = (const T*)a.dsbuf + (long long)bh * Tn * a.ldt;
  float acc0[DPAD], acc1[DPAD];
#pragma unroll
  for (int c = 0; c < DPAD; ++c) { acc0[c] = 0.f; acc1[c] = 0.f; }
  for (int ic = 0; ic < Tn; ic += QC) {
    __syncthreads();
    load_rows<T>(Qs, DP, qp, a.ld, ic, QC, Tn, d);
    if (!rel) load_rows<T>(Gs, DP, gp, a.ldo, ic, QC, Tn, d);
    __syncthreads();
    for (int qi = w; qi < QC; qi += 4) {
      const int i = ic + qi;
      if (i >= Tn) break;                                   // wave-uniform
      const float* qrow = Qs + qi * DP;
      if (!rel) {
        float ds = 0.f, p = 0.f;
        if (col < Tn) { ds = ldf(S + (long long)i * a.ldt + col); p = ldf(P + (long long)i * a.ldt + col); }
        const float* grow = Gs + qi * DP;
#pragma unroll
        for (int c = 0; c < DPAD; ++c) if (c < d) { acc0[c] += ds * qrow[c]; acc1[c] += p * grow[c]; }
      } else {
        const int j = i - (Tn - 1) + col;
        const float ds = (col < 2 * Tn - 1 && j >= 0 && j < Tn) ? ldf(S + (long long)i * a.ldt + j) : 0.f;
#pragma unroll
        for (int c = 0; c < DPAD; ++c) if (c < d) acc0[c] += ds * qrow[c];
      }
    }
  }
  __syncthreads();
  for (int ww = 0; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int c = 0; c < DPAD; ++c) if (c < d) {
        if (ww == 0) { R0[lane * DP + c] = acc0[c]; R1[lane * DP + c] = acc1[c]; }
        else { R0[lane * DP + c] += acc0[c]; R1[lane * DP + c] += acc1[c]; }
      }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < 64 * d; idx += 256) {
    const int r = idx / d, c = idx - r * d; const int cc = c0 + r;
    if (!rel) {
      if (cc < Tn) { stf((T*)a.dk + ((long long)b * Tn + cc) * a.ldd + h * d + c, R0[r * DP + c]); stf((T*)a.dv + ((long long)b * Tn + cc) * a.ldd + h * d + c, R1[r * DP + c]); }
    } else if (cc < 2 * Tn - 1) {
      atomicAdd(a.de + (long long)cc * a.ldde + h * d + c, R0[r * DP + c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
static int fill_args(AttnArgs& a, const avec_attn_t* p) {
  a.q = p->q; a.k = p->k; a.v = p->v; a.ld = p->ld; a.e = p->e; a.lde = p->lde; a.lens = p->lens; a.len_div = p->len_div > 0 ? p->len_div : 1; a.q_full = p->q_full > 0 ? p->q_full : p->T;
  a.mask = p->mask; a.mask_bstride = p->mask_bstride; a.o = p->o; a.ldo = p->ldo; a.lse = p->lse; a.dout = p->dout;
  a.dq = p->dq; a.dk = p->dk; a.dv = p->dv; a.lddq = p->lddq; a.ldd = p->ldd; a.de = p->de; a.ldde = p->ldde; a.pbuf = p->pbuf; a.dsbuf = p->dsbuf; a.ldt = p->ldt; a.dsrel = p->dsrel; a.ldr = p->ldr;
  a.B = p->B; a.H = p->H; a.T = p->T; a.d = p->d; a.scale = p->scale; a.Tk = p->Tk > 0 ? p->Tk : p->T;
  AVEC_CHECK_ARG(a.q && a.k && a.v && a.e && a.o && a.lse, "attention: null pointer");
  AVEC_CHECK_ARG(a.B > 0 && a.H > 0 && a.T > 0 && a.d > 0 && a.d <= 64 * NCS && a.Tk >= a.T, "attention: bad dims B=%d H=%d T=%d Tk=%d d=%d (d <= %d, Tk >= T)", a.B, a.H, a.T, a.Tk, a.d, 64 * NCS);
  return 0;
}
template <typename K> static int set_lds(K kern, size_t bytes) {
  if (bytes > 160 * 1024) { avec_set_error("attention: %zu bytes of LDS requested (> 160 KiB)", bytes); return -1; }
  static const void* done[16]; static size_t done_bytes[16]; static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern && done_bytes[i] >= bytes) return 0;
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
    if (ndone < 16) { done[ndone] = (const void*)kern; done_bytes[ndone] = bytes; ++ndone; }
  }
  return 0;
}

extern "C" int avec_relpos_attention_fwd(int dtype, const avec_attn_t* p, hipStream_t st) {
  AttnArgs a; AVEC_CHECK_ARG(p, "attention_fwd: null args"); if (int r = fill_args(a, p)) return r;
  if (dtype == AVEC_BF16) { const int r = attn_mfma_fwd(a, st); if (r != 1) return r; }      // MFMA path; 1 = not applicable, fall through
  const int DP = a.d | 1; size_t lds = (size_t)(2 * TK + (TQ + TK - 1) + TQ) * DP * 4 + 4 * 64 * 4;
  dim3 grid((a.T + TQ - 1) / TQ, a.B * a.H);
  if (dtype == AVEC_BF16) { if (int r = set_lds(attn_rows_kernel<bf16, false>, lds)) return r; hipLaunchKernelGGL((attn_rows_kernel<bf16, false>), grid, dim3(256), lds, st, a); }
  else { if (int r = set_lds(attn_rows_kernel<float, false>, lds)) return r; hipLaunchKernelGGL((attn_rows_kernel<float, false>), grid, dim3(256), lds, st, a); }
  AVEC_LAUNCH_CHECK(); return 0;
}

template <typename T> static int launch_bwd(AttnArgs& a, hipStream_t st) {
  const int DP = a.d | 1;
  size_t lds1 = (size_t)(2 * TK + (TQ + TK - 1) + 2 * TQ) * DP * 4 + 4 * 64 * 4;
  dim3 grid1((a.T + TQ - 1) / TQ, a.B * a.H);
  bool rows_done = false;
  if (sizeof(T) == 2) { const int r = attn_mfma_bwd_rows(a, st); if (r == 0) rows_done = true; else if (r != 1) return r; }   // bf16 MFMA row pass
  if (!rows_done) {
    if (int r = set_lds(attn_rows_kernel<T, true>, lds1)) return r;
    hipLaunchKernelGGL((attn_rows_kernel<T, true>), grid1, dim3(256), lds1, st, a);
  }
  if (a.dsrel || a.Tk != a.T) return 0;            // (Tk != T: key/value cache attached -- only the probability / dQ row pass is defined)  dK/dV/dE are computed by the caller with avec_gemm_tn_batched on P / dS / dSrel (MFMA path)
  size_t lds2 = (size_t)(2 * QC + 2 * 64) * DP * 4;
  const int ktiles = (a.T + 63) / 64, rtiles = (2 * a.T - 1 + 63) / 64;
  dim3 grid2(ktiles + rtiles, a.B * a.H);
#define LB(DPAD) do { if (int r = set_lds(attn_bwd_cols_kernel<T, DPAD>, lds2)) return r; hipLaunchKernelGGL((attn_bwd_cols_kernel<T, DPAD>), grid2, dim3(256), lds2, st, a, ktiles); } while (0)
  AVEC_CHECK_ARG(a.d <= 96, "attention_bwd: the register-accumulating column pass serves d <= 96 (d=%d): pass dsrel and use avec_gemm_tn_batched", a.d);
  if (a.d <= 48) LB(48); else if (a.d <= 64) LB(64); else LB(96);
#undef LB
  return 0;
}

extern "C" int avec_relpos_attention_bwd(int dtype, const avec_attn_t* p, hipStream_t st) {
  AttnArgs a; AVEC_CHECK_ARG(p, "attention_bwd: null args"); if (int r = fill_args(a, p)) return r;
  AVEC_CHECK_ARG(a.dout && a.dq && a.pbuf && a.dsbuf && a.ldt >= a.Tk && (a.dsrel ? a.ldr >= a.Tk + a.T - 1 : (a.Tk != a.T || (a.dk && a.dv && a.de))), "attention_bwd: null gradient / scratch pointer");
  int r = (dtype == AVEC_BF16) ? launch_bwd<bf16>(a, st) : launch_bwd<float>(a, st);
  if (r) return r;
  AVEC_LAUNCH_CHECK(); return 0;
}
