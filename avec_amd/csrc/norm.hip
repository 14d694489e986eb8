// Normalisation / elementwise / reduction kernels (HBM-bound; 4-wide vector access, wave64 shuffles).
//   LayerNorm fwd/bwd (wave per row), BatchNorm (channels-last) stats finalize / apply / backward,
//   grad_prep (dropout+scale+cast+bias-grad), softmax fwd/bwd, casts, dropout, patch pool / unpool,
//   global average pool.
#include "common.h"
#include "avec_hip.h"

#include "vec.h"

// second pass of the two-pass column reductions (vec.h): dst[n][colblock*W + w] += sum over slots of partial[colblock][slot][n][w]
__global__ __launch_bounds__(256) void col_finalize_kernel(ColFin f) {
  __shared__ float red[16][17];
  const int cw = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int ncols = f.NV * f.W;
  const int o = blockIdx.x * 16 + cw;
  const unsigned cb = blockIdx.y;
  const int s0 = blockIdx.z * 128;
  float s = 0.f;
  if (o < ncols) {
    const float* p = f.partial + ((size_t)cb * f.nslots) * ncols + o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int si = s0 + sl + 16 * k; const int sc = si < f.nslots ? si : f.nslots - 1;
      const float v = p[(size_t)sc * ncols];
      s += si < f.nslots ? v : 0.f;
    }
  }
  red[sl][cw] = s;
  __syncthreads();
  if (sl == 0 && o < ncols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cw];
    const int n = o / f.W, w = o - n * f.W; const int col = (int)cb * f.W + w;
    float* d = nullptr;
#pragma unroll
    for (int i = 0; i < FIN_MAXNV; ++i) if (i == n) d = f.dst[i];
    if (d && col < f.C) atomicAdd(d + (long long)col * f.dstride, t);
  }
}
int col_finalize(const ColWs& ws, unsigned colblocks, unsigned nslots, int NV, int W, float* const* dst, int C, hipStream_t st, int dstride) {
  ColFin f; f.partial = ws.partial; f.NV = NV; f.W = W; f.C = C; f.nslots = (int)nslots; f.dstride = dstride;
  for (int i = 0; i < FIN_MAXNV; ++i) f.dst[i] = i < NV ? dst[i] : nullptr;
  dim3 grid((unsigned)((NV * W + 15) / 16), colblocks, (nslots + 127) / 128);
  hipLaunchKernelGGL(col_finalize_kernel, grid, dim3(256), 0, st, f);
  AVEC_LAUNCH_CHECK(); return 0;
}

// =============================================================================================
// LayerNorm: x fp32 [M][D]; one wave per row.   (nn.LayerNorm(eps=1e-6): nnet/modules.py:278,302,373; nnet/blocks.py:267)
// =============================================================================================
template <typename TO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                     TO* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long long M, int D, float eps) {
  const int lane = threadIdx.x & 63; const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + row * D;
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) { float v[4]; ld4<float>(xr + c, v); s += v[0] + v[1] + v[2] + v[3]; }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
  for (int c = lane * 4; c < D; c += 256) { float v[4]; ld4<float>(xr + c, v); for (int e = 0; e < 4; ++e) { float d = v[e] - mu; q += d * d; } }
  const float rs = rsqrtf(wave_sum(q) / D + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  for (int c = lane * 4; c < D; c += 256) {
    float v[4], gg[4], bb[4], o[4]; ld4<float>(xr + c, v); ld4<float>(g + c, gg); ld4<float>(b + c, bb);
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - mu) * rs * gg[e] + bb[e];
    st4<TO>(y + row * D + c, o);
  }
}

// dx (+)= LN backward; dgamma/dbeta: per-lane partial sums over the wave's rows, reduced per block in LDS, then the two-pass column reduction.
// NG = column groups of 256 per lane-quad (D <= 256 NG): the conformer widths (180..360) need 2, so the row pieces of the NEXT row are requested before the
// two wave reductions of the current one (a wave walks 4..8 rows; un-pipelined, every row costs a full load latency).
template <typename TG, int NG>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TG* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ g, float* dx, const float* dres,
                                                     float* __restrict__ dg, float* __restrict__ db, long long M, int D, ColWs ws) {
  const int lane = threadIdx.x & 63; const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6); const int nwaves = gridDim.x * 4;
  float pg[NG][4], pb[NG][4], gg[NG][4];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int c = lane * 4 + i * 256;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pg[i][e] = 0.f; pb[i][e] = 0.f; gg[i][e] = 0.f; }
    if (c < D) ld4<float>(g + c, gg[i]);
  }
  struct RowData { float d[NG][4], v[NG][4], o[NG][4], mu, rs; };
  auto fetch = [&](long long row, RowData& r) {
    r.mu = mean[row]; r.rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int c = lane * 4 + i * 256;
      if (c < D) {
        ld4<TG>(dy + row * D + c, r.d[i]); ld4<float>(x + row * D + c, r.v[i]);
        if (dres) ld4<float>(dres + row * D + c, r.o[i]); else { r.o[i][0] = r.o[i][1] = r.o[i][2] = r.o[i][3] = 0.f; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { r.d[i][e] = 0.f; r.v[i][e] = r.mu; r.o[i][e] = 0.f; }
      }
    }
  };
  RowData cur, nxt;
  long long row = wave_id;
  if (row < M) fetch(row, cur);
  for (; row < M; row += nwaves) {
    const bool more = row + nwaves < M;
    if (more) fetch(row + nwaves, nxt);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (cur.v[i][e] - cur.mu) * cur.rs, t = cur.d[i][e] * gg[i][e];
        cur.v[i][e] = xh; s1 += t; s2 += t * xh; pg[i][e] += cur.d[i][e] * xh; pb[i][e] += cur.d[i][e];
      }
    s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int c = lane * 4 + i * 256; if (c >= D) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) cur.o[i][e] += cur.rs * (cur.d[i][e] * gg[i][e] - s1 - cur.v[i][e] * s2);
      st4<float>(dx + row * D + c, cur.o[i]);
    }
    if (more) cur = nxt;
  }
  // block-level reduction of the 4 waves' partials through LDS, then one atomic per column per block
  extern __shared__ float lnred[];           // [2][D]
  for (int c = threadIdx.x; c < 2 * D; c += 256) lnred[c] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    int c = lane * 4 + i * 256; if (c >= D) break;
    for (int e = 0; e < 4; ++e) { atomicAdd(lnred + c + e, pg[i][e]); atomicAdd(lnred + D + c + e, pb[i][e]); }
  }
  __syncthreads();
  if (!ws.partial) { for (int c = threadIdx.x; c < D; c += 256) { atomicAdd(dg + c, lnred[c]); atomicAdd(db + c, lnred[D + c]); } return; }
  float* mine = ws_slot(ws, 0, blockIdx.x, gridDim.x, 2 * D);
  for (int c = threadIdx.x; c < 2 * D; c += 256) mine[c] = lnred[c];
}

extern "C" int avec_layernorm_fwd(int dtype, const float* x, const float* gamma, const float* beta, void* y, int y_f32,
                                  float* mean, float* rstd, long long M, int D, float eps, hipStream_t st) {
  AVEC_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  AVEC_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0, "layernorm_fwd: D=%d must be a positive multiple of 4", D);
  dim3 grid((unsigned)((M + 3) / 4));
  if (y_f32 || dtype == AVEC_F32) hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, dim3(256), 0, st, x, gamma, beta, (float*)y, mean, rstd, M, D, eps);
  else hipLaunchKernelGGL(ln_fwd_kernel<bf16>, grid, dim3(256), 0, st, x, gamma, beta, (bf16*)y, mean, rstd, M, D, eps);
  AVEC_LAUNCH_CHECK(); return 0;
}

// dx only (dgamma / dbeta are produced later by avec_layernorm_param_grads_grouped): one wave per row, every row in flight at once -- no serial walk over
// rows, no column reduction on the dependent chain of the backward pass.
// Optional second output (avec_layernorm_bwd_prep): prep = act(palpha * dropmask * dx) -- what the module in front of this one would compute from dx with a
// grad_prep launch of its own at the start of ITS backward (out = res + alpha * Dropout(.): nnet/blocks.py:292-301).
struct LnPrep { void* out; float alpha, p; const unsigned long long* rng; unsigned stream; int f32; };
template <typename TG, int NG>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const TG* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ g, float* __restrict__ dx,
                                                          const float* __restrict__ dres, long long M, int D, LnPrep pr) {
  const int lane = threadIdx.x & 63; const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float mu = mean[row], rs = rstd[row];
  DropKey dk; dk.k0 = 0u; dk.thr = 0u; dk.scale = 1.f;
  if (pr.out) dk = drop_key(pr.rng, pr.stream, pr.p);          // {seed, step} requested with the row, not behind the reductions (a memory round trip of its own there)
  float d[NG][4], v[NG][4], o[NG][4], gg[NG][4];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < D) {
      ld4<TG>(dy + row * D + c, d[i]); ld4<float>(x + row * D + c, v[i]); ld4<float>(g + c, gg[i]);
      if (dres) ld4<float>(dres + row * D + c, o[i]); else { o[i][0] = o[i][1] = o[i][2] = o[i][3] = 0.f; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[i][e] = 0.f; v[i][e] = mu; o[i][e] = 0.f; gg[i][e] = 0.f; }
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NG; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float xh = (v[i][e] - mu) * rs, t = d[i][e] * gg[i][e]; v[i][e] = xh; s1 += t; s2 += t * xh; }
  s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int c = lane * 4 + i * 256; if (c >= D) break;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[i][e] += rs * (d[i][e] * gg[i][e] - s1 - v[i][e] * s2);
    st4<float>(dx + row * D + c, o[i]);
    if (pr.out) {
      float q[4];
      float ds[4]; drop4(dk, (unsigned long long)row * D + c, ds);      // (D % 4 == 0, c % 4 == 0)
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = o[i][e] * (pr.alpha * ds[e]);
      if (pr.f32) st4<float>((float*)pr.out + row * D + c, q); else st4<bf16>((bf16*)pr.out + row * D + c, q);
    }
  }
}

// ---- two consecutive LayerNorms per launch (round 4): the block-closing norm and the next block's first pre-norm; one wave per row, the row in registers (D <= 512) ----
template <typename TO>
__global__ __launch_bounds__(256) void ln_fwd2_kernel(const float* __restrict__ x, const float* __restrict__ g1, const float* __restrict__ b1, float eps1,
                                                      float* __restrict__ y1, float* __restrict__ mean1, float* __restrict__ rstd1,
                                                      const float* __restrict__ g2, const float* __restrict__ b2, float eps2,
                                                      TO* __restrict__ h2, float* __restrict__ mean2, float* __restrict__ rstd2, long long M, int D) {
  const int lane = threadIdx.x & 63; const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int c = lane * 4 + i * 256; if (c < D) ld4<float>(x + row * D + c, v[i]); else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; }
  auto norm = [&](const float* g, const float* b, float eps, float* mean, float* rstd) {
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) sm += v[i][0] + v[i][1] + v[i][2] + v[i][3];      // (columns beyond D hold zeros)
    const float mu = wave_sum(sm) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int c = lane * 4 + i * 256; if (c < D) for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; } }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane * 4 + i * 256; if (c >= D) break;
      float gg[4], bb[4]; ld4<float>(g + c, gg); ld4<float>(b + c, bb);
      for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mu) * rs * gg[e] + bb[e];
    }
  };
  norm(g1, b1, eps1, mean1, rstd1);
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int c = lane * 4 + i * 256; if (c < D) st4<float>(y1 + row * D + c, v[i]); }
  norm(g2, b2, eps2, mean2, rstd2);
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int c = lane * 4 + i * 256; if (c < D) st4<TO>(h2 + row * D + c, v[i]); }
}
extern "C" int avec_layernorm_fwd2(int dtype, const float* x, const float* gamma1, const float* beta1, float eps1, float* y1, float* mean1, float* rstd1,
                                   const float* gamma2, const float* beta2, float eps2, void* h2, float* mean2, float* rstd2, long long M, int D, hipStream_t st) {
  AVEC_CHECK_ARG(x && gamma1 && beta1 && y1 && mean1 && rstd1 && gamma2 && beta2 && h2 && mean2 && rstd2, "layernorm_fwd2: null pointer");
  AVEC_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 512, "layernorm_fwd2: D=%d must be a multiple of 4 in [4, 512]", D);
  dim3 grid((unsigned)((M + 3) / 4));
  if (dtype == AVEC_F32) hipLaunchKernelGGL(ln_fwd2_kernel<float>, grid, dim3(256), 0, st, x, gamma1, beta1, eps1, y1, mean1, rstd1, gamma2, beta2, eps2, (float*)h2, mean2, rstd2, M, D);
  else hipLaunchKernelGGL(ln_fwd2_kernel<bf16>, grid, dim3(256), 0, st, x, gamma1, beta1, eps1, y1, mean1, rstd1, gamma2, beta2, eps2, (bf16*)h2, mean2, rstd2, M, D);
  AVEC_LAUNCH_CHECK(); return 0;
}
template <typename TG>
__global__ __launch_bounds__(256) void ln_bwd_rows2_kernel(const TG* __restrict__ dy2, const float* __restrict__ x2, const float* __restrict__ mean2, const float* __restrict__ rstd2,
                                                           const float* __restrict__ g2, const float* __restrict__ dres2, float* __restrict__ dx2,
                                                           const float* __restrict__ x1, const float* __restrict__ mean1, const float* __restrict__ rstd1,
                                                           const float* __restrict__ g1, float* __restrict__ dx1, long long M, int D, LnPrep pr) {
  const int lane = threadIdx.x & 63; const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float d[2][4], v[2][4], o[2][4], gg[2][4], w1[2][4], gg1[2][4];
  const float mu2 = mean2[row], rs2 = rstd2[row], mu1 = mean1[row], rs1 = rstd1[row];
  DropKey dk; dk.k0 = 0u; dk.thr = 0u; dk.scale = 1.f;
  if (pr.out) dk = drop_key(pr.rng, pr.stream, pr.p);          // (requested with the rows: see ln_bwd_rows_kernel)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < D) {
      ld4<TG>(dy2 + row * D + c, d[i]); ld4<float>(x2 + row * D + c, v[i]); ld4<float>(g2 + c, gg[i]); ld4<float>(x1 + row * D + c, w1[i]); ld4<float>(g1 + c, gg1[i]);
      if (dres2) ld4<float>(dres2 + row * D + c, o[i]); else { o[i][0] = o[i][1] = o[i][2] = o[i][3] = 0.f; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[i][e] = 0.f; v[i][e] = mu2; o[i][e] = 0.f; gg[i][e] = 0.f; w1[i][e] = mu1; gg1[i][e] = 0.f; }
    }
  }
  // the same arithmetic, in the same order, as two ln_bwd_rows_kernel launches
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float xh = (v[i][e] - mu2) * rs2, t = d[i][e] * gg[i][e]; v[i][e] = xh; s1 += t; s2 += t * xh; }
  s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane * 4 + i * 256;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[i][e] += rs2 * (d[i][e] * gg[i][e] - s1 - v[i][e] * s2);
    if (c < D) st4<float>(dx2 + row * D + c, o[i]); else { o[i][0] = o[i][1] = o[i][2] = o[i][3] = 0.f; }
  }
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float xh = (w1[i][e] - mu1) * rs1, t = o[i][e] * gg1[i][e]; w1[i][e] = xh; t1 += t; t2 += t * xh; }
  t1 = wave_sum(t1) / D; t2 = wave_sum(t2) / D;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane * 4 + i * 256; if (c >= D) break;
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = rs1 * (o[i][e] * gg1[i][e] - t1 - w1[i][e] * t2);
    st4<float>(dx1 + row * D + c, r);
    if (pr.out) {
      float q[4], ds[4]; drop4(dk, (unsigned long long)row * D + c, ds);
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = r[e] * (pr.alpha * ds[e]);
      if (pr.f32) st4<float>((float*)pr.out + row * D + c, q); else st4<bf16>((bf16*)pr.out + row * D + c, q);
    }
  }
}
extern "C" int avec_layernorm_bwd2(int dtype, const void* dy2, const float* x2, const float* mean2, const float* rstd2, const float* gamma2, const float* dres2, float* dx2,
                                   const float* x1, const float* mean1, const float* rstd1, const float* gamma1, float* dx1,
                                   void* prep, float prep_alpha, float prep_drop_p, const unsigned long long* rng, unsigned rng_stream, long long M, int D, hipStream_t st) {
  AVEC_CHECK_ARG(dy2 && x2 && mean2 && rstd2 && gamma2 && dx2 && x1 && mean1 && rstd1 && gamma1 && dx1, "layernorm_bwd2: null pointer");
  AVEC_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 512 && (!prep || prep_drop_p <= 0.f || rng), "layernorm_bwd2: D=%d must be a multiple of 4 in [4, 512]; dropout needs rng", D);
  LnPrep pr; pr.out = prep; pr.alpha = prep_alpha; pr.p = prep ? prep_drop_p : 0.f; pr.rng = rng; pr.stream = rng_stream; pr.f32 = dtype == AVEC_F32;
  dim3 grid((unsigned)((M + 3) / 4));
  if (dtype == AVEC_F32) hipLaunchKernelGGL(ln_bwd_rows2_kernel<float>, grid, dim3(256), 0, st, (const float*)dy2, x2, mean2, rstd2, gamma2, dres2, dx2, x1, mean1, rstd1, gamma1, dx1, M, D, pr);
  else hipLaunchKernelGGL(ln_bwd_rows2_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)dy2, x2, mean2, rstd2, gamma2, dres2, dx2, x1, mean1, rstd1, gamma1, dx1, M, D, pr);
  AVEC_LAUNCH_CHECK(); return 0;
}

// ---- grouped LayerNorm parameter gradients: dgamma_k[c] += sum_m dy_k[m][c] * xhat_k[m][c], dbeta_k[c] += sum_m dy_k[m][c] for up to AVEC_LN_GROUP_MAX layers in ONE launch.
// They only feed the optimizer, so the caller queues them (with the weight-gradient GEMMs) instead of paying a column reduction inside every LayerNorm backward.
// Workgroup = 128 rows of one layer; thread = 4 consecutive columns of every R-th row; LDS reduction over the row lanes, then one atomic per column per workgroup.
struct LnItem { const void* dy; const float* x; const float* mean; const float* rstd; float* dg; float* db; int M, D, dy_f32, first; };
struct LnGroup { LnItem it[AVEC_LN_GROUP_MAX]; int n; };
static constexpr int LNG_ROWS = 128;
__global__ __launch_bounds__(256) void ln_param_grads_grouped_kernel(LnGroup grp) {
  __shared__ float red[2][256][4];
  const int w = blockIdx.x;
  int lo = 0, hi = grp.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (grp.it[mid].first <= w) lo = mid; else hi = mid - 1; }
  const LnItem& t = grp.it[lo];
  const int D = t.D, L = D >> 2, R = 256 / L;                 // L lanes of 4 columns per row, R rows per pass (D <= 1024)
  const int r = threadIdx.x / L, l = threadIdx.x - r * L;
  const int m0 = (w - t.first) * LNG_ROWS; int m1 = m0 + LNG_ROWS; if (m1 > t.M) m1 = t.M;
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
  if (r < R) {
    constexpr int U = 8;                         // rows in flight per thread: one row at a time is one load latency per row (32 of them per workgroup at D = 256: 57 us per launch)
    for (int m = m0 + r; m < m1; m += R * U) {
      float d[U][4], v[U][4], mu[U], rs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mm = m + u * R; const bool ok = mm < m1; const long long mi = ok ? mm : m;
        if (t.dy_f32) ld4<float>((const float*)t.dy + mi * D + l * 4, d[u]); else ld4<bf16>((const bf16*)t.dy + mi * D + l * 4, d[u]);
        ld4<float>(t.x + mi * D + l * 4, v[u]); mu[u] = t.mean[mi]; rs[u] = ok ? t.rstd[mi] : 0.f;
        if (!ok) d[u][0] = d[u][1] = d[u][2] = d[u][3] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { pg[e] += d[u][e] * (v[u][e] - mu[u]) * rs[u]; pb[e] += d[u][e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = pg[e]; red[1][threadIdx.x][e] = pb[e]; }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * D; o += 256) {
    const int which = o >= D, c = o - which * D; float s = 0.f;
    for (int rr = 0; rr < R; ++rr) s += red[which][rr * L + (c >> 2)][c & 3];
    atomicAdd((which ? t.db : t.dg) + c, s);
  }
}
extern "C" int avec_layernorm_param_grads_grouped(int dtype, const avec_ln_item_t* items, int n, hipStream_t st) {
  AVEC_CHECK_ARG(items && n > 0 && n <= AVEC_LN_GROUP_MAX, "layernorm_param_grads_grouped: need 1..%d items (got %d)", AVEC_LN_GROUP_MAX, n);
  LnGroup grp; grp.n = n; int first = 0;
  for (int k = 0; k < n; ++k) {
    const avec_ln_item_t& s = items[k];
    AVEC_CHECK_ARG(s.dy && s.x && s.mean && s.rstd && s.dgamma && s.dbeta && s.M > 0 && s.M < (1ll << 31) && s.D >= 4 && s.D % 4 == 0 && s.D <= 1024,
                   "layernorm_param_grads_grouped: item %d: null pointer or D=%d not a multiple of 4 in [4, 1024]", k, s.D);
    LnItem& t = grp.it[k];
    t.dy = s.dy; t.x = s.x; t.mean = s.mean; t.rstd = s.rstd; t.dg = s.dgamma; t.db = s.dbeta; t.M = (int)s.M; t.D = s.D;
    t.dy_f32 = (s.dy_f32 || dtype == AVEC_F32) ? 1 : 0; t.first = first;
    first += (int)((s.M + LNG_ROWS - 1) / LNG_ROWS);
  }
  hipLaunchKernelGGL(ln_param_grads_grouped_kernel, dim3((unsigned)first), dim3(256), 0, st, grp);
  AVEC_LAUNCH_CHECK(); return 0;
}

static int layernorm_bwd_impl(int dtype, const void* dy, int dy_f32, const float* x, const float* mean, const float* rstd, const float* gamma,
                              float* dx, const float* dres, float* dgamma, float* dbeta, long long M, int D, const LnPrep& pr, hipStream_t st) {
  AVEC_CHECK_ARG(dy && x && mean && rstd && gamma && dx && (!dgamma == !dbeta), "layernorm_bwd: null pointer");
  AVEC_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 1536, "layernorm_bwd: D=%d must be a multiple of 4 and <= 1536", D);
  AVEC_CHECK_ARG(!pr.out || (!dgamma && (pr.p <= 0.f || pr.rng)), "layernorm_bwd_prep: the second output comes with the dx-only kernel (dgamma == NULL); dropout needs rng");
  if (!dgamma) {        // input gradient only (parameter gradients deferred to avec_layernorm_param_grads_grouped)
    const dim3 grid((unsigned)((M + 3) / 4)); const bool f32in = dy_f32 || dtype == AVEC_F32;
#define AVEC_LN_ROWS(TG, NG) hipLaunchKernelGGL((ln_bwd_rows_kernel<TG, NG>), grid, dim3(256), 0, st, (const TG*)dy, x, mean, rstd, gamma, dx, dres, M, D, pr)
    if (D <= 512) { if (f32in) AVEC_LN_ROWS(float, 2); else AVEC_LN_ROWS(bf16, 2); }
    else { if (f32in) AVEC_LN_ROWS(float, 6); else AVEC_LN_ROWS(bf16, 6); }
#undef AVEC_LN_ROWS
    AVEC_LAUNCH_CHECK(); return 0;
  }
  long long nb = (M + 15) / 16; if (nb > 256) nb = 256; if (nb < 1) nb = 1;
  ColWs ws = avec_reduce_ws((size_t)nb * 2 * D, st);
  if (!ws.partial && nb > 128) nb = 128;
  const size_t lds = (size_t)2 * D * sizeof(float);
#define AVEC_LN_BWD(TG, NG) hipLaunchKernelGGL((ln_bwd_kernel<TG, NG>), dim3((unsigned)nb), dim3(256), lds, st, (const TG*)dy, x, mean, rstd, gamma, dx, dres, dgamma, dbeta, M, D, ws)
  const bool f32in = dy_f32 || dtype == AVEC_F32;
  if (D <= 512) { if (f32in) AVEC_LN_BWD(float, 2); else AVEC_LN_BWD(bf16, 2); }
  else { if (f32in) AVEC_LN_BWD(float, 6); else AVEC_LN_BWD(bf16, 6); }
#undef AVEC_LN_BWD
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {dgamma, dbeta}; return col_finalize(ws, 1, (unsigned)nb, 2, D, dst, D, st); }
  return 0;
}
extern "C" int avec_layernorm_bwd(int dtype, const void* dy, int dy_f32, const float* x, const float* mean, const float* rstd, const float* gamma,
                                  float* dx, const float* dres, float* dgamma, float* dbeta, long long M, int D, hipStream_t st) {
  LnPrep pr; pr.out = nullptr; pr.alpha = 1.f; pr.p = 0.f; pr.rng = nullptr; pr.stream = 0; pr.f32 = 0;
  return layernorm_bwd_impl(dtype, dy, dy_f32, x, mean, rstd, gamma, dx, dres, dgamma, dbeta, M, D, pr, st);
}
extern "C" int avec_layernorm_bwd_prep(int dtype, const void* dy, int dy_f32, const float* x, const float* mean, const float* rstd, const float* gamma,
                                       float* dx, const float* dres, void* prep, float prep_alpha, float prep_drop_p, const unsigned long long* rng, unsigned rng_stream,
                                       long long M, int D, hipStream_t st) {
  AVEC_CHECK_ARG(prep, "layernorm_bwd_prep: null prep buffer");
  LnPrep pr; pr.out = prep; pr.alpha = prep_alpha; pr.p = prep_drop_p; pr.rng = rng; pr.stream = rng_stream; pr.f32 = dtype == AVEC_F32;
  return layernorm_bwd_impl(dtype, dy, dy_f32, x, mean, rstd, gamma, dx, dres, nullptr, nullptr, M, D, pr, st);
}

// =============================================================================================
// grad_prep: dacc[m][n] = alpha * dropmask * dout[m][n]   (fp32 -> act)   and   dbias[n] += sum_m dacc
//   backward of   out = res + alpha * Dropout(acc + bias)   (nnet/blocks.py:292-301, nnet/modules.py:286-288)
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void grad_prep_kernel(const float* __restrict__ dout, long long ldd, T* __restrict__ dacc, float alpha, float p,
                                                        const unsigned long long* rng, unsigned stream, float* dbias, long long M, int N, ColWs ws) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + tx) * 4;
  float part[1][4] = {{0.f, 0.f, 0.f, 0.f}};
  const DropKey dk = drop_key(rng, stream, p);
  if (col < N) {
    for (long long row = (long long)blockIdx.y * 8 + ty; row < M; row += (long long)gridDim.y * 8) {
      float v[4]; ld4<float>(dout + row * ldd + col, v);
      float ds[4]; drop4(dk, (unsigned long long)row * N + col, ds);
      for (int e = 0; e < 4; ++e) { v[e] *= alpha * ds[e]; part[0][e] += v[e]; }
      st4<T>(dacc + row * N + col, v);
    }
  }
  float* const dst[1] = {dbias};
  if (dbias) colreduce_atomic<1>(part, dst, col, N, ws);
}

// flat variant without the bias-gradient column sums (those now come out of the weight-gradient GEMM): plain elementwise grid
template <typename T>
__global__ __launch_bounds__(256) void grad_prep_flat_kernel(const float* __restrict__ dout, long long ldd, T* __restrict__ dacc, float alpha, float p,
                                                             const unsigned long long* rng, unsigned stream, long long M, int N) {
  const int N4 = N >> 2; const long long n4 = M * N4;
  const DropKey dk = drop_key(rng, stream, p);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long row = n4 < (1ll << 31) ? (long long)((unsigned)i / (unsigned)N4) : i / N4; const int col = (int)(i - row * N4) * 4;
    float v[4]; ld4<float>(dout + row * ldd + col, v);
    float ds[4]; drop4(dk, (unsigned long long)row * N + col, ds);
    for (int e = 0; e < 4; ++e) v[e] *= alpha * ds[e];
    st4<T>(dacc + row * N + col, v);
  }
}
extern "C" int avec_grad_prep(int dtype, const float* dout, long long ld, void* dacc, float alpha, float drop_p, const unsigned long long* rng,
                              unsigned rng_stream, float* dbias, long long M, int N, hipStream_t st) {
  AVEC_CHECK_ARG(dout && dacc && M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "grad_prep: bad arguments (N=%d ld=%lld)", N, ld);
  AVEC_CHECK_ARG(!(drop_p > 0.f) || rng, "grad_prep: dropout without rng");
  if (!dbias) {
    long long nb = (M * (N / 4) + 255) / 256; if (nb > 8192) nb = 8192;
    DISPATCH_T(dtype, hipLaunchKernelGGL(grad_prep_flat_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, dout, ld, (T*)dacc, alpha, drop_p, rng, rng_stream, M, N));
    AVEC_LAUNCH_CHECK(); return 0;
  }
  dim3 grid = col_grid(M, N); ColWs ws = dbias ? col_ws_if(grid, 1, N, st) : ColWs{nullptr};
  DISPATCH_T(dtype, hipLaunchKernelGGL(grad_prep_kernel<T>, grid, dim3(256), 0, st, dout, ld, (T*)dacc, alpha, drop_p, rng, rng_stream, dbias, M, N, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[1] = {dbias}; return col_finalize(ws, grid.x, grid.y, 1, 128, dst, N, st); }
  return 0;
}

// column sums of an act matrix:  out[n] += sum_m x[m][n]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long long ld, float* out, long long M, int N, ColWs ws) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  float part[1][4] = {{0.f, 0.f, 0.f, 0.f}};
  if (col < N) for (long long row = (long long)blockIdx.y * 8 + ty; row < M; row += (long long)gridDim.y * 8) {
    float v[4]; ld4<T>(x + row * ld + col, v); for (int e = 0; e < 4; ++e) part[0][e] += v[e];
  }
  float* const dst[1] = {out};
  colreduce_atomic<1>(part, dst, col, N, ws);
}
extern "C" int avec_colsum(int dtype, const void* x, long long ld, float* out, long long M, int N, hipStream_t st) { return colsum_launch(dtype, x, ld, out, M, N, true, st); }
// use_ws = false: plain atomics (callers that may run concurrently with other users of the reduction workspace, e.g. weight gradients on the side stream)
int colsum_launch(int dtype, const void* x, long long ld, float* out, long long M, int N, bool use_ws, hipStream_t st) {
  AVEC_CHECK_ARG(x && out && M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "colsum: bad arguments");
  dim3 grid = col_grid(M, N); if (!use_ws && grid.y > 16) grid.y = 16; ColWs ws = use_ws ? col_ws_if(grid, 1, N, st) : ColWs{nullptr};
  DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, st, (const T*)x, ld, out, M, N, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[1] = {out}; return col_finalize(ws, grid.x, grid.y, 1, 128, dst, N, st); }
  return 0;
}

// =============================================================================================
// BatchNorm over channels-last [M][C]   (nnet/normalizations.py:42-170; SyncBatchNorm :172-249 when the
// caller all-reduces `stats` across ranks between the statistics pass and bn_finalize)
//   ss = [scale | shift | mean | rstd], each [C]
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ y, float* stats, long long M, int C, ColWs ws) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (col < C) for (long long row = (long long)blockIdx.y * 8 + ty; row < M; row += (long long)gridDim.y * 8) {
    float v[4]; ld4<T>(y + row * C + col, v); for (int e = 0; e < 4; ++e) { part[0][e] += v[e]; part[1][e] += v[e] * v[e]; }
  }
  float* const dst[2] = {stats, stats + C};
  colreduce_atomic<2>(part, dst, col, C, ws);
}
extern "C" int avec_bn_stats(int dtype, const void* y, float* stats, long long M, int C, hipStream_t st) {
  AVEC_CHECK_ARG(y && stats && M > 0 && C > 0 && C % 4 == 0, "bn_stats: bad arguments");
  dim3 grid = col_grid(M, C); ColWs ws = col_ws_if(grid, 2, C, st);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_stats_kernel<T>, grid, dim3(256), 0, st, (const T*)y, stats, M, C, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {stats, stats + C}; return col_finalize(ws, grid.x, grid.y, 2, 128, dst, C, st); }
  return 0;
}

// block = 16 channels x 16 replica lanes (the 64 replicated partial sums of the GEMM epilogue are read in parallel, 4 per lane)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* stats, int nrep, const float* count_ptr, float count, const float* gamma, const float* beta, float* rmean, float* rvar,
                                                          long long* nbt, float momentum, float eps, float* ss, int C, int training) {
  __shared__ float red[2][16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  // everything the finishing lanes need is requested together with the partial sums: ONE memory round trip in a kernel that sits in the conformer's launch chain
  // (it used to take three: sums, then count / running statistics, then gamma / beta -- 4.4 us for 256 channels, tools/block_trace.py)
  const bool lead = rl == 0 && c < C;
  float gm = 0.f, bt = 0.f, rm = 0.f, rv = 0.f, n = count;
  if (lead) { gm = gamma[c]; bt = beta[c]; if (rmean) { rm = rmean[c]; rv = rvar[c]; } if (training && count_ptr) n = *count_ptr; }
  float s1 = 0.f, s2 = 0.f;
  if (training && c < C) for (int r = rl; r < nrep; r += 16) { s1 += stats[(long long)r * 2 * C + c]; s2 += stats[(long long)r * 2 * C + C + c]; }
  red[0][rl][cl] = s1; red[1][rl][cl] = s2;
  __syncthreads();
  if (!lead) return;
  float mean, var;
  if (training) {
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s1 += red[0][k][cl]; s2 += red[1][k][cl]; }
    mean = s1 / n; var = fmaxf(s2 / n - mean * mean, 0.f);
    if (rmean && isfinite(mean) && isfinite(var)) {      // (a SyncBatchNorm exchange that lost a rank delivers NaN sums: the running statistics must survive that step)
      rmean[c] = (1.f - momentum) * rm + momentum * mean;
      rvar[c] = (1.f - momentum) * rv + momentum * var * (n / fmaxf(n - 1.f, 1.f));
      if (c == 0 && nbt) *nbt += 1;
    }
  } else { mean = rm; var = rv; }
  const float rs = rsqrtf(var + eps);
  ss[c] = gm * rs; ss[C + c] = bt - mean * gm * rs; ss[2 * C + c] = mean; ss[3 * C + c] = rs;
}
// SyncBatchNorm helpers (one launch each instead of three small framework kernels per layer and pass):
//   collapse : out[0..2C) = sum over the n_replicas partial [sum | sumsq] vectors, out[2C] = count      (the vector that is all-reduced)
//   affine   : dgamma += dstats[C..2C), dbeta += dstats[0..C)                                            (LOCAL sums, before dstats is all-reduced)
__global__ __launch_bounds__(256) void bn_collapse_kernel(const float* __restrict__ stats, int nrep, float count, float* __restrict__ out, int C2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C2) { float s = 0.f; for (int r = 0; r < nrep; ++r) s += stats[(long long)r * C2 + c]; out[c] = s; }
  if (c == 0) out[C2] = count;
}
__global__ __launch_bounds__(256) void bn_affine_grads_kernel(const float* __restrict__ dstats, float* dgamma, float* dbeta, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) { dgamma[c] += dstats[C + c]; dbeta[c] += dstats[c]; }
}
extern "C" int avec_bn_collapse(const float* stats, int n_replicas, float count, float* out, int C, hipStream_t st) {
  AVEC_CHECK_ARG(stats && out && n_replicas > 0 && C > 0, "bn_collapse: bad arguments");
  hipLaunchKernelGGL(bn_collapse_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, stats, n_replicas, count, out, 2 * C);
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_bn_affine_grads(const float* dstats, float* dgamma, float* dbeta, int C, hipStream_t st) {
  AVEC_CHECK_ARG(dstats && dgamma && dbeta && C > 0, "bn_affine_grads: bad arguments");
  hipLaunchKernelGGL(bn_affine_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, st, dstats, dgamma, dbeta, C);
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_bn_finalize(const float* stats, int n_replicas, const float* count_ptr, float count, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, long long* num_batches_tracked, float momentum, float eps, float* ss, int C, int training, hipStream_t st) {
  AVEC_CHECK_ARG(gamma && beta && ss && C > 0 && (training ? (stats != nullptr) : (running_mean && running_var)), "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, stats, n_replicas > 0 ? n_replicas : 1, count_ptr, count, gamma, beta, running_mean, running_var,
                     num_batches_tracked, momentum, eps, ss, C, training);
  AVEC_LAUNCH_CHECK(); return 0;
}

// measurement aid: one lane writes the 100 MHz wall clock into slot `i` of a device array (tools/step_stamps.py: where a graph-replayed step spends its time, per stream, without a profiler)
__global__ void stamp_kernel(unsigned long long* out, int i) { if (threadIdx.x == 0) out[i] = wall_clock64(); }
extern "C" int avec_stamp(unsigned long long* out, int slot, hipStream_t st) {
  AVEC_CHECK_ARG(out && slot >= 0, "stamp: bad arguments");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, st, out, slot);
  AVEC_LAUNCH_CHECK(); return 0;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ stats, int nrep, const float* __restrict__ ss, float* __restrict__ dstats, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, sy = 0.f;
  for (int r = 0; r < nrep; ++r) { s1 += stats[(long long)r * 2 * C + c]; sy += stats[(long long)r * 2 * C + C + c]; }
  dstats[c] = s1; dstats[C + c] = ss[3 * C + c] * (sy - ss[2 * C + c] * s1);
}
extern "C" int avec_bn_bwd_finalize(const float* stats, int n_replicas, const float* ss, float* dstats, int C, hipStream_t st) {
  AVEC_CHECK_ARG(stats && ss && dstats && C > 0 && n_replicas > 0, "bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, stats, n_replicas, ss, dstats, C);
  AVEC_LAUNCH_CHECK(); return 0;
}

// out = act(y*scale + shift (+ residual))
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ ss, const T* __restrict__ res, int act,
                                                       T* __restrict__ out, long long n4, int C) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long off = i * 4; const int c = (int)(off % C);
    float v[4], sc[4], sh[4]; ld4<T>(y + off, v); ld4<float>(ss + c, sc); ld4<float>(ss + C + c, sh);
    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
    if (res) { float r[4]; ld4<T>(res + off, r); for (int e = 0; e < 4; ++e) v[e] += r[e]; }
    if (act == 1) { for (int e = 0; e < 4; ++e) v[e] = swishf_(v[e]); } else if (act == 2) { for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f); }
    st4<T>(out + off, v);
  }
}
// 8-wide variant (C % 8 == 0, < 2^31 chunks): 16-byte accesses.  The grid is sized so that the grid stride is a whole number of rows (bn8_blocks): a thread
// keeps ITS eight channels for the whole loop and the per-channel coefficients live in registers (the first version reloaded 2..7 x 32 B of them and took a
// 32-bit modulo per 16 B of payload -- the 3-operand backward pass ran at 3.8 TB/s on L1 traffic); U chunks per thread are in flight per trip.
static inline unsigned bn8_blocks(long long n8, int C, unsigned cap) {
  const unsigned C8 = (unsigned)C >> 3; unsigned g = C8, t = 256; while (t) { const unsigned r = g % t; g = t; t = r; }      // g = gcd(C8, 256)
  const unsigned q = C8 / g;                                              // the block count must be a multiple of q
  long long nb = (n8 + 255) / 256; if (nb > cap) nb = cap;
  nb = nb / q * q; if (nb < q) nb = q;
  return (unsigned)nb;
}
constexpr int BN8_U = 2;        // chunks in flight per thread in the element-wise passes (3 and 4: no gain)
#ifndef AVEC_BN8_UR
#define AVEC_BN8_UR 1
#endif
constexpr int BN8_UR = AVEC_BN8_UR;       // ... in the reduction pass (2 measured 4-6 % slower there)
constexpr unsigned BN8_CAP_FWD = 8192, BN8_CAP_BWD = 3072;      // grid caps: the 3-operand backward pass likes fewer, longer-running blocks (115200 x 256: 41.5 -> 36.7 us,
                                                                // 28800 x 512: 28.0 -> 21.9 us), the forward pass the opposite (32.9 vs 33.9 us); tools/bench_bn.py

template <typename T>
__global__ __launch_bounds__(256) void bn_apply8_kernel(const T* __restrict__ y, const float* __restrict__ ss, const T* __restrict__ res, int act,
                                                        T* __restrict__ out, unsigned n8, int C, unsigned char* __restrict__ mask = nullptr,
                                                        const float* __restrict__ res_ss = nullptr) {
  const unsigned C8 = (unsigned)C >> 3, stride = gridDim.x * 256, i0 = blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(i0 % C8) * 8;
  float sc[8], sh[8]; ld8<float>(ss + c, sc); ld8<float>(ss + C + c, sh);
  float rsc[8], rsh[8];                                   // res_ss: the residual is a raw convolution output with its own BatchNorm coefficients (projection shortcut)
  if (res_ss) { ld8<float>(res_ss + c, rsc); ld8<float>(res_ss + C + c, rsh);
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[e] += rsh[e]; }
  for (unsigned i = i0; i < n8; i += stride * BN8_U) {
    Raw8<T> ry[BN8_U], rr[BN8_U];
#pragma unroll
    for (int u = 0; u < BN8_U; ++u) { const unsigned j = i + u * stride; if (j < n8) { ry[u].load(y + (long long)j * 8); if (res) rr[u].load(res + (long long)j * 8); } }
#pragma unroll
    for (int u = 0; u < BN8_U; ++u) {
      const unsigned j = i + u * stride; if (j >= n8) break;
      float v[8]; ry[u].get(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
      if (res) { float r[8]; rr[u].get(r);
        if (res_ss) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e] * rsc[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e]; } }
      if (act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = swishf_(v[e]);
      } else if (act == 2) {
        if (mask) {                      // one bit per element: out > 0 (what the backward pass needs of `out`)
          unsigned b = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) b |= (v[e] > 0.f ? 1u : 0u) << e;
          mask[j] = (unsigned char)b;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      st8<T>(out + (long long)j * 8, v);
    }
  }
}
extern "C" int avec_bn_apply_fwd_mask(int dtype, const void* y, const float* ss, const void* residual, const float* residual_ss, void* out, unsigned char* mask, long long M, int C,
                                      hipStream_t st) {
  AVEC_CHECK_ARG(y && ss && out && mask && M > 0 && C > 0 && C % 8 == 0 && M * C / 8 < (1ll << 31) && (residual || !residual_ss), "bn_apply_fwd_mask: bad arguments (C %% 8 == 0, M*C < 2^34)");
  const long long n8 = M * C / 8; const unsigned nb8 = bn8_blocks(n8, C, BN8_CAP_FWD);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply8_kernel<T>, dim3((unsigned)nb8), dim3(256), 0, st, (const T*)y, ss, (const T*)residual, 2, (T*)out, (unsigned)n8, C, mask, residual_ss));
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_bn_apply_fwd(int dtype, const void* y, const float* ss, const void* residual, int act, void* out, long long M, int C, hipStream_t st) {
  AVEC_CHECK_ARG(y && ss && out && M > 0 && C > 0 && C % 4 == 0, "bn_apply_fwd: bad arguments");
  if (C % 8 == 0 && M * C / 8 < (1ll << 31)) {
    const long long n8 = M * C / 8; const unsigned nb8 = bn8_blocks(n8, C, BN8_CAP_FWD);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply8_kernel<T>, dim3((unsigned)nb8), dim3(256), 0, st, (const T*)y, ss, (const T*)residual, act, (T*)out, (unsigned)n8, C));
    AVEC_LAUNCH_CHECK(); return 0;
  }
  long long n4 = M * C / 4; long long nb = (n4 + 255) / 256; if (nb > 4096) nb = 4096;
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)y, ss, (const T*)residual, act, (T*)out, n4, C));
  AVEC_LAUNCH_CHECK(); return 0;
}

// dr = dout * act'(.)   (ReLU: mask from saved output `out`; Swish: from pre = y*scale+shift)
template <typename T>
__device__ __forceinline__ void bn_dr(const T* dout, const T* y, const T* out, const float* ss, int act, long long off, int c, int C, float dr[4], float yh[4]) {
  float d[4], v[4], mu[4], rs[4]; ld4<T>(dout + off, d); ld4<T>(y + off, v); ld4<float>(ss + 2 * C + c, mu); ld4<float>(ss + 3 * C + c, rs);
  for (int e = 0; e < 4; ++e) yh[e] = (v[e] - mu[e]) * rs[e];
  if (act == 2) {      // ReLU: mask from the saved output when there is one (residual added before the ReLU), else from the recomputed pre-activation
    if (out) { float o[4]; ld4<T>(out + off, o); for (int e = 0; e < 4; ++e) dr[e] = o[e] > 0.f ? d[e] : 0.f; }
    else { float sc[4], sh[4]; ld4<float>(ss + c, sc); ld4<float>(ss + C + c, sh); for (int e = 0; e < 4; ++e) dr[e] = (v[e] * sc[e] + sh[e]) > 0.f ? d[e] : 0.f; }
  }
  else if (act == 1) { float sc[4], sh[4]; ld4<float>(ss + c, sc); ld4<float>(ss + C + c, sh); for (int e = 0; e < 4; ++e) dr[e] = d[e] * dswishf_(v[e] * sc[e] + sh[e]); }
  else { for (int e = 0; e < 4; ++e) dr[e] = d[e]; }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ y, const T* __restrict__ out, const float* __restrict__ ss,
                                                            int act, float* dstats, long long M, int C, ColWs ws) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; const int col = (blockIdx.x * 32 + tx) * 4;
  float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (col < C) for (long long row = (long long)blockIdx.y * 8 + ty; row < M; row += (long long)gridDim.y * 8) {
    float dr[4], yh[4]; bn_dr<T>(dout, y, out, ss, act, row * C + col, col, C, dr, yh);
    for (int e = 0; e < 4; ++e) { part[0][e] += dr[e]; part[1][e] += dr[e] * yh[e]; }
  }
  float* const dst[2] = {dstats, dstats + C};
  colreduce_atomic<2>(part, dst, col, C, ws);
}
// 8-wide flat variant (C % 8 == 0): every lane busy for narrow C, 16 B accesses
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce8_kernel(const T* __restrict__ dout, const T* __restrict__ y, const T* __restrict__ out, const float* __restrict__ ss,
                                                             int act, float* dstats, long long M, int C, ColWs ws, const unsigned char* __restrict__ mask = nullptr) {
  const Col8 m = col8_map(C);
  float part[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) part[0][e] = part[1][e] = 0.f;
  if (m.active) {
    const int c = m.l * 8;
    float mu[8], rs[8], sc[8], sh[8]; ld8<float>(ss + 2 * C + c, mu); ld8<float>(ss + 3 * C + c, rs);
    if (act == 1 || (act == 2 && !out && !mask)) { ld8<float>(ss + c, sc); ld8<float>(ss + C + c, sh); }
    const long long rstride = (long long)gridDim.x * m.R;
    for (long long row = (long long)blockIdx.x * m.R + m.r; row < M; row += rstride * BN8_UR) {
      Raw8<T> rd[BN8_UR], rv[BN8_UR], ro[BN8_UR]; unsigned mb[BN8_UR];
#pragma unroll
      for (int u = 0; u < BN8_UR; ++u) { const long long r = row + u * rstride; if (r < M) { const long long off = r * C + c; rd[u].load(dout + off); rv[u].load(y + off);
        if (mask) mb[u] = mask[off >> 3]; else if (act == 2 && out) ro[u].load(out + off); } }
#pragma unroll
      for (int u = 0; u < BN8_UR; ++u) {
        if (row + u * rstride >= M) break;
        float d[8], v[8]; rd[u].get(d); rv[u].get(v);
        if (mask) {
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] = (mb[u] >> e) & 1u ? d[e] : 0.f; }
        else if (act == 2 && out) { float o[8]; ro[u].get(o);
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f; }
        else if (act == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] = (v[e] * sc[e] + sh[e]) > 0.f ? d[e] : 0.f; }
        else if (act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] *= dswishf_(v[e] * sc[e] + sh[e]); }
#pragma unroll
        for (int e = 0; e < 8; ++e) { part[0][e] += d[e]; part[1][e] += d[e] * (v[e] - mu[e]) * rs[e]; }
      }
    }
  }
  float* const dst[2] = {dstats, dstats + C};
  colreduce8_atomic<2>(part, dst, m, ws);
}
extern "C" int avec_bn_bwd_reduce_mask(int dtype, const void* dout, const void* y, const unsigned char* mask, const float* ss, float* dstats, long long M, int C, hipStream_t st) {
  AVEC_CHECK_ARG(dout && y && mask && ss && dstats && M > 0 && col8_ok(C), "bn_bwd_reduce_mask: bad arguments");
  ColWs ws; const unsigned nb = col8_cfg(M, C, 2, &ws, st);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_reduce8_kernel<T>, dim3(nb), dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)nullptr, ss, 2, dstats, M, C, ws, mask));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {dstats, dstats + C}; return col_finalize(ws, 1, nb, 2, C, dst, C, st); }
  return 0;
}
extern "C" int avec_bn_bwd_reduce(int dtype, const void* dout, const void* y, const void* out, const float* ss, int act, float* dstats, long long M, int C, hipStream_t st) {
  AVEC_CHECK_ARG(dout && y && ss && dstats && M > 0 && C % 4 == 0, "bn_bwd_reduce: bad arguments");
  if (col8_ok(C)) {
    ColWs ws; const unsigned nb = col8_cfg(M, C, 2, &ws, st);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_reduce8_kernel<T>, dim3(nb), dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)out, ss, act, dstats, M, C, ws));
    AVEC_LAUNCH_CHECK();
    if (ws.partial) { float* const dst[2] = {dstats, dstats + C}; return col_finalize(ws, 1, nb, 2, C, dst, C, st); }
    return 0;
  }
  dim3 grid = col_grid(M, C); ColWs ws = col_ws_if(grid, 2, C, st);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, grid, dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)out, ss, act, dstats, M, C, ws));
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {dstats, dstats + C}; return col_finalize(ws, grid.x, grid.y, 2, 128, dst, C, st); }
  return 0;
}
// dy = gamma*rstd*(dr - mean(dr) - yhat*mean(dr*yhat)); optional dres = dr; block 0 adds dgamma/dbeta
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ y, const T* __restrict__ out, const float* __restrict__ ss,
                                                           const float* __restrict__ gamma, const float* __restrict__ dstats, const float* count_ptr, float count, int act,
                                                           T* __restrict__ dy, T* __restrict__ dres, float* dgamma, float* dbeta, long long n4, int C) {
  const float inv_n = 1.f / (count_ptr ? *count_ptr : count);
  if (blockIdx.x == 0 && dgamma) for (int c = threadIdx.x; c < C; c += 256) { atomicAdd(dgamma + c, dstats[C + c]); atomicAdd(dbeta + c, dstats[c]); }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long off = i * 4; const int c = (int)(off % C);
    float dr[4], yh[4], g[4], rs[4], s1[4], s2[4], o[4];
    bn_dr<T>(dout, y, out, ss, act, off, c, C, dr, yh);
    ld4<float>(gamma + c, g); ld4<float>(ss + 3 * C + c, rs); ld4<float>(dstats + c, s1); ld4<float>(dstats + C + c, s2);
    for (int e = 0; e < 4; ++e) o[e] = g[e] * rs[e] * (dr[e] - s1[e] * inv_n - yh[e] * s2[e] * inv_n);
    st4<T>(dy + off, o);
    if (dres) st4<T>(dres + off, dr);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(const T* __restrict__ dout, const T* __restrict__ y, const T* __restrict__ out, const float* __restrict__ ss,
                                                            const float* __restrict__ gamma, const float* __restrict__ dstats, const float* count_ptr, float count, int act,
                                                            T* __restrict__ dy, T* __restrict__ dres, float* dgamma, float* dbeta, unsigned n8, int C,
                                                            const unsigned char* __restrict__ mask = nullptr) {
  const float inv_n = 1.f / (count_ptr ? *count_ptr : count);
  if (blockIdx.x == 0 && dgamma) for (int c = threadIdx.x; c < C; c += 256) { atomicAdd(dgamma + c, dstats[C + c]); atomicAdd(dbeta + c, dstats[c]); }
  const unsigned C8 = (unsigned)C >> 3, stride = gridDim.x * 256, i0 = blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(i0 % C8) * 8;
  // dy = A (d - m1 - (v - mu) rstd m2)   with A = gamma rstd, m1 = mean(d), m2 = mean(d yhat)
  float mu[8], rs[8], A[8], m1[8], m2[8], sc[8], sh[8];
  { float g[8], s1[8], s2[8]; ld8<float>(ss + 2 * C + c, mu); ld8<float>(ss + 3 * C + c, rs); ld8<float>(gamma + c, g); ld8<float>(dstats + c, s1); ld8<float>(dstats + C + c, s2);
#pragma unroll
    for (int e = 0; e < 8; ++e) { A[e] = g[e] * rs[e]; m1[e] = s1[e] * inv_n; m2[e] = s2[e] * inv_n; } }
  const bool recompute = act != 0 && !(act == 2 && (out || mask));
  if (recompute) { ld8<float>(ss + c, sc); ld8<float>(ss + C + c, sh); }
  for (unsigned i = i0; i < n8; i += stride * BN8_U) {
    Raw8<T> rd[BN8_U], rv[BN8_U], ro[BN8_U]; unsigned mb[BN8_U];
#pragma unroll
    for (int u = 0; u < BN8_U; ++u) { const unsigned j = i + u * stride; if (j < n8) { const long long off = (long long)j * 8; rd[u].load(dout + off); rv[u].load(y + off);
      if (mask) mb[u] = mask[j]; else if (act == 2 && out) ro[u].load(out + off); } }
#pragma unroll
    for (int u = 0; u < BN8_U; ++u) {
      const unsigned j = i + u * stride; if (j >= n8) break;
      const long long off = (long long)j * 8;
      float d[8], v[8], o[8]; rd[u].get(d); rv[u].get(v);
      if (mask) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = (mb[u] >> e) & 1u ? d[e] : 0.f;
      } else if (act == 2 && out) { float q[8]; ro[u].get(q);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = q[e] > 0.f ? d[e] : 0.f;
      } else if (act != 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float pre = v[e] * sc[e] + sh[e]; d[e] = act == 1 ? d[e] * dswishf_(pre) : (pre > 0.f ? d[e] : 0.f); }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = A[e] * (d[e] - m1[e] - (v[e] - mu[e]) * rs[e] * m2[e]);
      st8<T>(dy + off, o);
      if (dres) st8<T>(dres + off, d);
    }
  }
}
extern "C" int avec_bn_bwd_apply_mask(int dtype, const void* dout, const void* y, const unsigned char* mask, const float* ss, const float* gamma, const float* dstats,
                                      const float* count_ptr, float count, void* dy, void* dres, float* dgamma, float* dbeta, long long M, int C, hipStream_t st) {
  AVEC_CHECK_ARG(dout && y && mask && ss && gamma && dstats && dy && M > 0 && C % 8 == 0 && M * C / 8 < (1ll << 31), "bn_bwd_apply_mask: bad arguments");
  const long long n8 = M * C / 8; const unsigned nb8 = bn8_blocks(n8, C, BN8_CAP_BWD);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply8_kernel<T>, dim3((unsigned)nb8), dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)nullptr, ss, gamma, dstats,
                                       count_ptr, count, 2, (T*)dy, (T*)dres, dgamma, dbeta, (unsigned)n8, C, mask));
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_bn_bwd_apply(int dtype, const void* dout, const void* y, const void* out, const float* ss, const float* gamma, const float* dstats,
                                 const float* count_ptr, float count, int act, void* dy, void* dres, float* dgamma, float* dbeta, long long M, int C, hipStream_t st) {
  AVEC_CHECK_ARG(dout && y && ss && gamma && dstats && dy && M > 0 && C % 4 == 0, "bn_bwd_apply: bad arguments");
  if (C % 8 == 0 && M * C / 8 < (1ll << 31)) {
    const long long n8 = M * C / 8; const unsigned nb8 = bn8_blocks(n8, C, BN8_CAP_BWD);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply8_kernel<T>, dim3((unsigned)nb8), dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)out, ss, gamma, dstats,
                                         count_ptr, count, act, (T*)dy, (T*)dres, dgamma, dbeta, (unsigned)n8, C));
    AVEC_LAUNCH_CHECK(); return 0;
  }
  long long n4 = M * C / 4; long long nb = (n4 + 255) / 256; if (nb > 4096) nb = 4096;
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)out, ss, gamma, dstats,
                                       count_ptr, count, act, (T*)dy, (T*)dres, dgamma, dbeta, n4, C));
  AVEC_LAUNCH_CHECK(); return 0;
}

// =============================================================================================
// softmax over the last dim (InterCTC residual, nnet/modules.py:395-400): logits fp32 -> probs act
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, T* __restrict__ p, long long M, int V) {
  const int lane = threadIdx.x & 63; const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, x[row * V + c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += __expf(x[row * V + c] - mx);
  s = 1.f / wave_sum(s);
  for (int c = lane; c < V; c += 64) stf(p + row * V + c, __expf(x[row * V + c] - mx) * s);
}
// dlogits (+)= p * (dp - sum(dp*p));  probabilities recomputed from the fp32 logits
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ dp, const float* __restrict__ x, float* dx, const float* dadd, long long M, int V) {
  const int lane = threadIdx.x & 63; const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, x[row * V + c]);
  mx = wave_max(mx);
  float s = 0.f, dot = 0.f;
  for (int c = lane; c < V; c += 64) { float e = __expf(x[row * V + c] - mx); s += e; dot += e * ldf(dp + row * V + c); }
  s = wave_sum(s); dot = wave_sum(dot) / s;
  for (int c = lane; c < V; c += 64) {
    float pr = __expf(x[row * V + c] - mx) / s; float g = pr * (ldf(dp + row * V + c) - dot);
    dx[row * V + c] = dadd ? dadd[row * V + c] + g : g;
  }
}
extern "C" int avec_softmax_fwd(int dtype, const float* logits, void* probs, long long M, int V, hipStream_t st) {
  AVEC_CHECK_ARG(logits && probs && M > 0 && V > 0, "softmax_fwd: bad arguments");
  DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_fwd_kernel<T>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, logits, (T*)probs, M, V));
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_softmax_bwd(int dtype, const void* dprobs, const float* logits, float* dlogits, const float* dadd, long long M, int V, hipStream_t st) {
  AVEC_CHECK_ARG(dprobs && logits && dlogits && M > 0 && V > 0, "softmax_bwd: bad arguments");
  DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_bwd_kernel<T>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const T*)dprobs, logits, dlogits, dadd, M, V));
  AVEC_LAUNCH_CHECK(); return 0;
}

// =============================================================================================
// casts / copies / dropout on the fp32 stream
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, long long lds_, T* __restrict__ dst, long long ldd, long long M, int N) {
  const long long n4 = M * (N / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long row = i / (N / 4); const int c = (int)(i % (N / 4)) * 4;
    float v[4]; ld4<float>(src + row * lds_ + c, v); st4<T>(dst + row * ldd + c, v);
  }
}
extern "C" int avec_cast_rows(int dtype, const float* src, long long ld_src, void* dst, long long ld_dst, long long M, int N, hipStream_t st) {
  AVEC_CHECK_ARG(src && dst && M > 0 && N > 0 && N % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0, "cast_rows: bad arguments");
  long long nb = (M * (N / 4) + 255) / 256; if (nb > 4096) nb = 4096;
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_rows_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, src, ld_src, (T*)dst, ld_dst, M, N));
  AVEC_LAUNCH_CHECK(); return 0;
}
template <typename T>
__global__ __launch_bounds__(256) void to_f32_rows_kernel(const T* __restrict__ src, long long lds_, float* __restrict__ dst, long long ldd, long long M, int N, int accum) {
  const long long n4 = M * (N / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long row = i / (N / 4); const int c = (int)(i % (N / 4)) * 4;
    float v[4]; ld4<T>(src + row * lds_ + c, v);
    if (accum) { float o[4]; ld4<float>(dst + row * ldd + c, o); for (int e = 0; e < 4; ++e) v[e] += o[e]; }
    st4<float>(dst + row * ldd + c, v);
  }
}
extern "C" int avec_to_f32_rows(int dtype, const void* src, long long ld_src, float* dst, long long ld_dst, long long M, int N, int accum, hipStream_t st) {
  AVEC_CHECK_ARG(src && dst && M > 0 && N > 0 && N % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0, "to_f32_rows: bad arguments");
  long long nb = (M * (N / 4) + 255) / 256; if (nb > 4096) nb = 4096;
  DISPATCH_T(dtype, hipLaunchKernelGGL(to_f32_rows_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, (const T*)src, ld_src, dst, ld_dst, M, N, accum));
  AVEC_LAUNCH_CHECK(); return 0;
}

__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ y, float p, const unsigned long long* rng, unsigned stream, long long n) {
  const DropKey dk = drop_key(rng, stream, p);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = x[i] * drop_one(dk, (unsigned long long)i);
}
extern "C" int avec_dropout_f32(const float* x, float* y, float p, const unsigned long long* rng, unsigned rng_stream, long long n, hipStream_t st) {
  AVEC_CHECK_ARG(x && y && n > 0 && rng, "dropout_f32: bad arguments");
  long long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(dropout_f32_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, y, p, rng, rng_stream, n);
  AVEC_LAUNCH_CHECK(); return 0;
}

// ---- stand-alone activations (nnet/activations.py:39-69): on the hot path they live in GEMM / BatchNorm epilogues; these entries serve a bare nnet.Swish()(x),
// nnet.ReLU()(x), nnet.GLU(dim=-1)(x) on fp32 tensors.  act: 1 Swish, 2 ReLU, 3 GLU (x = [rows][2*C] -> y = [rows][C], a * sigmoid(b)) ----
__global__ __launch_bounds__(256) void act_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int act, long long rows, int C, int bwd) {
  const long long n = rows * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    if (act == 3) {
      const long long r = i / C; const int c = (int)(i - r * C);
      const float a = x[r * 2 * C + c], b = x[r * 2 * C + C + c], sg = sigmoidf_(b);
      if (!bwd) out[i] = a * sg;
      else { const float g = dy[i]; out[r * 2 * C + c] = g * sg; out[r * 2 * C + C + c] = g * a * sg * (1.f - sg); }
    } else {
      const float v = x[i];
      if (!bwd) out[i] = act == 1 ? swishf_(v) : fmaxf(v, 0.f);
      else out[i] = dy[i] * (act == 1 ? dswishf_(v) : (v > 0.f ? 1.f : 0.f));
    }
  }
}
extern "C" int avec_act_f32(int act, const float* x, const float* dy, float* out, long long rows, int C, int backward, hipStream_t st) {
  AVEC_CHECK_ARG(x && out && rows > 0 && C > 0 && act >= 1 && act <= 3 && (!backward || dy), "act_f32: bad arguments (act=%d)", act);
  long long nb = (rows * C + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(act_f32_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, dy, out, act, rows, C, backward);
  AVEC_LAUNCH_CHECK(); return 0;
}

// =============================================================================================
// patch attention helpers (nnet/attentions.py:348-382): avg-pool by P with zero padding (divisor P),
// nearest up-sample x P sliced to T fused with dropout + residual add.
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void patch_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Tn, int D, int P, int Tp) {
  const long long n4 = (long long)B * Tp * (D / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % (D / 4)) * 4; const long long r = i / (D / 4); const int tp = (int)(r % Tp); const int b = (int)(r / Tp);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < P; ++k) { int t = tp * P + k; if (t < Tn) { float v[4]; ld4<T>(x + ((long long)b * Tn + t) * D + c, v); for (int e = 0; e < 4; ++e) a[e] += v[e]; } }
    for (int e = 0; e < 4; ++e) a[e] /= P;
    st4<T>(y + r * D + c, a);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void patch_pool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Tn, int D, int P, int Tp) {
  const long long n4 = (long long)B * Tn * (D / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % (D / 4)) * 4; const long long r = i / (D / 4); const int t = (int)(r % Tn); const int b = (int)(r / Tn);
    float v[4]; ld4<T>(dy + ((long long)b * Tp + t / P) * D + c, v); for (int e = 0; e < 4; ++e) v[e] /= P;
    st4<T>(dx + r * D + c, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void patch_unpool_add_kernel(const T* __restrict__ o, const float* __restrict__ res, float* __restrict__ out, float p,
                                                               const unsigned long long* rng, unsigned stream, int B, int Tn, int D, int P, int Tp) {
  const long long n4 = (long long)B * Tn * (D / 4);
  const DropKey dk = drop_key(rng, stream, p);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % (D / 4)) * 4; const long long r = i / (D / 4); const int t = (int)(r % Tn); const int b = (int)(r / Tn);
    float v[4], q[4]; ld4<T>(o + ((long long)b * Tp + t / P) * D + c, v); ld4<float>(res + r * D + c, q);
    float ds[4]; drop4(dk, (unsigned long long)r * D + c, ds);
    for (int e = 0; e < 4; ++e) q[e] += v[e] * ds[e];
    st4<float>(out + r * D + c, q);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void patch_unpool_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dob, float p, const unsigned long long* rng, unsigned stream,
                                                               int B, int Tn, int D, int P, int Tp) {
  const long long n4 = (long long)B * Tp * (D / 4);
  const DropKey dk = drop_key(rng, stream, p);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % (D / 4)) * 4; const long long r = i / (D / 4); const int tp = (int)(r % Tp); const int b = (int)(r / Tp);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < P; ++k) {
      int t = tp * P + k; if (t >= Tn) break;
      const long long rr = (long long)b * Tn + t; float v[4]; ld4<float>(dout + rr * D + c, v);
      float ds[4]; drop4(dk, (unsigned long long)rr * D + c, ds);
      for (int e = 0; e < 4; ++e) a[e] += v[e] * ds[e];
    }
    st4<T>(dob + r * D + c, a);
  }
}
#define PATCH_LAUNCH(kern, total, ...) do { long long nb = ((total) + 255) / 256; if (nb > 4096) nb = 4096; if (nb < 1) nb = 1; \
  DISPATCH_T(dtype, hipLaunchKernelGGL(kern<T>, dim3((unsigned)nb), dim3(256), 0, st, __VA_ARGS__)); AVEC_LAUNCH_CHECK(); return 0; } while (0)
extern "C" int avec_patch_pool_fwd(int dtype, const void* x, void* y, int B, int T_, int D, int P, hipStream_t st) {
  AVEC_CHECK_ARG(x && y && B > 0 && T_ > 0 && D % 4 == 0 && P > 0, "patch_pool_fwd: bad arguments"); const int Tp = (T_ + P - 1) / P;
  PATCH_LAUNCH(patch_pool_fwd_kernel, (long long)B * Tp * (D / 4), (const T*)x, (T*)y, B, T_, D, P, Tp);
}
extern "C" int avec_patch_pool_bwd(int dtype, const void* dy, void* dx, int B, int T_, int D, int P, hipStream_t st) {
  AVEC_CHECK_ARG(dy && dx && B > 0 && T_ > 0 && D % 4 == 0 && P > 0, "patch_pool_bwd: bad arguments"); const int Tp = (T_ + P - 1) / P;
  PATCH_LAUNCH(patch_pool_bwd_kernel, (long long)B * T_ * (D / 4), (const T*)dy, (T*)dx, B, T_, D, P, Tp);
}
extern "C" int avec_patch_unpool_add(int dtype, const void* o, const float* res, float* out, float drop_p, const unsigned long long* rng, unsigned rng_stream,
                                     int B, int T_, int D, int P, hipStream_t st) {
  AVEC_CHECK_ARG(o && res && out && B > 0 && T_ > 0 && D % 4 == 0 && P > 0 && (drop_p <= 0.f || rng), "patch_unpool_add: bad arguments"); const int Tp = (T_ + P - 1) / P;
  PATCH_LAUNCH(patch_unpool_add_kernel, (long long)B * T_ * (D / 4), (const T*)o, res, out, drop_p, rng, rng_stream, B, T_, D, P, Tp);
}
extern "C" int avec_patch_unpool_bwd(int dtype, const float* dout, void* dob, float drop_p, const unsigned long long* rng, unsigned rng_stream,
                                     int B, int T_, int D, int P, hipStream_t st) {
  AVEC_CHECK_ARG(dout && dob && B > 0 && T_ > 0 && D % 4 == 0 && P > 0 && (drop_p <= 0.f || rng), "patch_unpool_bwd: bad arguments"); const int Tp = (T_ + P - 1) / P;
  PATCH_LAUNCH(patch_unpool_bwd_kernel, (long long)B * Tp * (D / 4), dout, (T*)dob, drop_p, rng, rng_stream, B, T_, D, P, Tp);
}

// =============================================================================================
// global average pool over HW (nnet/layers.py:1328-1342), channels-last [N][HW][C] -> [N][C]
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long N, int HW, int C) {
  const long long n4 = N * (C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % (C / 4)) * 4; const long long n = i / (C / 4);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < HW; ++s) { float v[4]; ld4<T>(x + (n * HW + s) * C + c, v); for (int e = 0; e < 4; ++e) a[e] += v[e]; }
    for (int e = 0; e < 4; ++e) a[e] /= HW;
    st4<T>(y + n * C + c, a);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, long long N, int HW, int C) {
  const long long n4 = N * HW * (C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % (C / 4)) * 4; const long long r = i / (C / 4); const long long n = r / HW;
    float v[4]; ld4<T>(dy + n * C + c, v); for (int e = 0; e < 4; ++e) v[e] /= HW;
    st4<T>(dx + r * C + c, v);
  }
}
extern "C" int avec_avgpool_fwd(int dtype, const void* x, void* y, long long N, int HW, int C, hipStream_t st) {
  AVEC_CHECK_ARG(x && y && N > 0 && HW > 0 && C % 4 == 0, "avgpool_fwd: bad arguments");
  PATCH_LAUNCH(avgpool_fwd_kernel, N * (C / 4), (const T*)x, (T*)y, N, HW, C);
}
extern "C" int avec_avgpool_bwd(int dtype, const void* dy, void* dx, long long N, int HW, int C, hipStream_t st) {
  AVEC_CHECK_ARG(dy && dx && N > 0 && HW > 0 && C % 4 == 0, "avgpool_bwd: bad arguments");
  PATCH_LAUNCH(avgpool_bwd_kernel, N * HW * (C / 4), (const T*)dy, (T*)dx, N, HW, C);
}

// dx[b][t*step] += src[b][t]  (backward of the strided k=1 conv_res rows, nnet/blocks.py:273-277)
__global__ __launch_bounds__(256) void strided_rows_add_kernel(float* __restrict__ dx, const float* __restrict__ src, int B, int Tn, int To, int D, int step) {
  const long long n4 = (long long)B * To * (D / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % (D / 4)) * 4; const long long r = i / (D / 4); const int to = (int)(r % To); const long long b = r / To;
    float v[4], o[4]; ld4<float>(src + r * D + c, v); float* dst = dx + (b * Tn + (long long)to * step) * D + c; ld4<float>(dst, o);
    for (int e = 0; e < 4; ++e) o[e] += v[e];
    st4<float>(dst, o);
  }
}
extern "C" int avec_strided_rows_add(float* dx, const float* src, int B, int T_, int To, int D, int step, hipStream_t st) {
  AVEC_CHECK_ARG(dx && src && B > 0 && T_ > 0 && To > 0 && D % 4 == 0 && step > 0 && (long long)(To - 1) * step < T_, "strided_rows_add: bad arguments");
  long long n4 = (long long)B * To * (D / 4); long long nb = (n4 + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(strided_rows_add_kernel, dim3((unsigned)nb), dim3(256), 0, st, dx, src, B, T_, To, D, step);
  AVEC_LAUNCH_CHECK(); return 0;
}
