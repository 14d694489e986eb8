// Visual stem without the pre-pool activation in HBM (round 3).
//   reference: Conv3d(1 -> 64, (5,7,7), stride (1,2,2), 'same') + BatchNorm3d + ReLU + MaxPool3d((1,3,3), stride (1,2,2))   (nnet/networks.py:459-470,
//   nnet/layers.py:326-503, 839-915, nnet/normalizations.py:90-170)
// The conv output z (3200 x 44 x 44 x 64 bf16 = 793 MB at the bench shape) used to be written once and read three times (pool forward, BatchNorm backward,
// weight gradient operand).  BatchNorm + ReLU is a per-channel MONOTONE map of z (increasing for gamma >= 0, decreasing for gamma < 0; gamma is known before
// the batch statistics are), so  maxpool(relu(bn(z))) = relu(bn(pool_s(z)))  with pool_s = max for gamma >= 0 and min for gamma < 0: the forward kernel pools
// the RAW conv output inside the workgroup and writes only the pooled winners zp (198 MB), their window slot idx (99 MB) and the batch statistics of z; a plain
// BatchNorm-apply over the pooled tensor finishes the forward pass.  The backward kernel recomputes z tiles from the input (MFMA, same code) and forms
//   dz = gamma * rstd * (route(dpool) - mean(dy) - xhat * mean(dy * xhat))
// without ever reading z from memory.
//
// Input: the clip as bf16 [clips][T][H][W] (W % 8 == 0: rows are whole 16-byte chunks).  A workgroup handles one band (frame, NB-th of the pooled rows): the 5 x SH
// input rows it needs are fetched by LDS-DMA (global_load_lds, 16 B per lane) into rows of [8 zero px | W px | 8 zero px] -- every chunk is 16-byte aligned on both
// sides, the left / right zero padding is a chunk sourced from a zero page.  Reduction order k = (kd, kh, slot), slot 0 = zero weight, slot 1..7 = kw 0..6: the 8
// operand elements of output pixel ow are the 8 consecutive staged pixels 2*ow - 4 .. 2*ow + 3 (four aligned 32-bit LDS reads), weights live in registers.
#include <type_traits>
#include "vec.h"
#include "avec_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

static __device__ __attribute__((aligned(64))) unsigned char s3p_zero16[64];

static constexpr int S3P_C = 64;
#ifndef S3P_STAMPS
#define S3P_STAMPS 0        // 1 (tools builds): shader-clock totals per phase of the forward kernels under abl & 64
#endif

struct S3P {
  int T3, H, W, OH, OW, PH, PW;
  int NB, pn;              // bands per frame, pooled rows per band
  int SH;                  // staged input rows per frame slot: 2 * (2 * pn + 1 - 1) + 7
  int CPR, pitch;          // 16-byte chunks per staged row (W / 8 + 2), row pitch in bytes
  int nchunks, slab_bytes; // 5 * SH * CPR chunks; LDS bytes (whole DMA passes of 256 chunks)
  int RING;                // forward: pixels of the conv-output ring (3 rows + one 128-pixel tile)
  long long items;         // clips * T3 * NB
};

__device__ __forceinline__ f32x16 s3p_mma(const chunk16& a, const chunk16& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// LDS-DMA of the band's input: slab[(kd * SH + r) * pitch + 16 * j] <- chunk j of input row ir0 + r of frame fr + kd - 2 (zero page outside the clip / frame, and
// for the two padding chunks).  Every lane issues slab_bytes / 4096 unconditional loads.
__device__ __forceinline__ void s3p_issue_slab(char* slab, const bf16* __restrict__ vb, long long clip, int fr, int ir0, const S3P& G) {
  const int tid = threadIdx.x, wave = tid >> 6;
  const bf16* src = vb + clip * (long long)G.T3 * G.H * G.W;
  const int iters = G.slab_bytes >> 12;
  for (int it = 0; it < iters; ++it) {
    const int q = it * 256 + tid;
    const int row = q / G.CPR, j = q - row * G.CPR;
    const int kd = row / G.SH, r = row - kd * G.SH;
    const int itf = fr + kd - 2, ih = ir0 + r;
    const bool ok = q < G.nchunks && itf >= 0 && itf < G.T3 && ih >= 0 && ih < G.H && j >= 1 && j <= G.CPR - 2;
    const void* s = ok ? (const void*)(src + ((long long)itf * G.H + ih) * G.W + (j - 1) * 8) : (const void*)s3p_zero16;
    __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(slab + (it * 256 + wave * 64) * 16), 16, 0, 0);
  }
}

// z tile: 32 pixels (this wave's, pixel p of the band for this lane) x 64 channels, K = 36 rows of 8 slots.  acc[j][r]: channel 32 j + (lane & 31), pixel row
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the wave's 32.
__device__ __forceinline__ void s3p_conv_tile(const char* slab, int ohl, int ow, int g, const chunk16 (&wf)[18][2], const int (&rowoff)[35], const S3P& G, f32x16 (&acc)[2]) {
  const char* pix = slab + (2 * ohl) * G.pitch + 4 * ow + 8;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  chunk16 fa[18];
  auto ldfrag = [&](int s) {
    const int offA = rowoff[2 * s > 34 ? 34 : 2 * s], offB = rowoff[2 * s + 1 > 34 ? 34 : 2 * s + 1];
    const uint32_t* rp = (const uint32_t*)(pix + (g ? offB : offA));
    chunk16 f; f.w[0] = rp[0]; f.w[1] = rp[1]; f.w[2] = rp[2]; f.w[3] = rp[3]; return f;
  };
#pragma unroll
  for (int s = 0; s < 3; ++s) fa[s] = ldfrag(s);
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    if (s + 3 < 18) fa[s + 3] = ldfrag(s + 3);
    asm volatile("" ::: "memory");
    acc[0] = s3p_mma(fa[s], wf[s][0], acc[0]);
    acc[1] = s3p_mma(fa[s], wf[s][1], acc[1]);
  }
}

// ------------------------------------------------------------------------------------------------
// forward: conv + batch statistics + window winners of the raw output
//   zp  [clips*T][PH][PW][64] bf16 : s * max_window(s * z), s = +1 (gamma >= 0) / -1
//   idx [clips*T][PH][PW][64] u8   : window slot kh * 3 + kw of the winner (first one in scan order on ties, like torch's max_pool)
// Persistent workgroups (weights stay in registers) walk over the bands.  A band's conv rows are produced 128 pixels (4 waves x 32) at a time into an LDS ring of
// 6 conv rows, CHANNEL-major ([64][6 * OW] bf16: a lane of the MFMA result owns one channel and 4 consecutive pixels per register quad = one 8-byte LDS store);
// a pooled row is reduced as soon as its three conv rows are complete.  Pool stage: thread = (channel, quarter of the pooled row); a 32-bit LDS word holds the
// column pair (2k, 2k+1), window pw = {hi(word pw-1), lo(word pw), hi(word pw)} per row; the 9 candidates are reduced with ONE unsigned max each on
// (order-preserving key of the bf16 value << 16 | 15 - slot): value and first-winner index in one go.
// Bands overlap by one conv row (the row above the first window): it is recomputed, not counted in the statistics.
// ------------------------------------------------------------------------------------------------
static constexpr int S3P_PWG = 6;        // pooled columns per pool-stage thread (4 threads per channel): PW <= 24

__device__ __forceinline__ uint32_t s3p_key(uint32_t x, uint32_t qinv) {       // x: bf16 value in the high half (low half ignored)
  const uint32_t m = (uint32_t)((int)x >> 31);
  return ((x ^ (m | 0x80000000u)) & 0xffff0000u) | qinv;
}
// the same order-preserving key for BOTH bf16 halves of a word (round 6: the ring holds keys, made once per conv output by the epilogue instead of once per window candidate
// -- every output is a candidate of 2.25 windows -- by the pool stage, whose per-candidate work is then one v_and_or / v_lshl_or)
typedef short s3p_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t s3p_key2(uint32_t w) {
  const s3p_v2s m = __builtin_bit_cast(s3p_v2s, w) >> 15;                       // 0xffff for a negative half
  return w ^ (__builtin_bit_cast(uint32_t, m) | 0x80008000u);
}

__global__ __launch_bounds__(256, 2) void stem3p_fwd_kernel(const bf16* __restrict__ vb, const bf16* __restrict__ wsh, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                             bf16* __restrict__ zp, unsigned char* __restrict__ idx, int want_stats, S3P G, ColWs ws, float* stats, int abl) {
  extern __shared__ __attribute__((aligned(16))) char s3[];
  char* slab = s3;
  char* ring = s3 + G.slab_bytes;                                   // [64 channels][RING pixels] bf16
  float* lsum = (float*)(ring + G.RING * 128);                      // [2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, pl = lane & 31;
  if (tid < 128) lsum[tid] = 0.f;
  chunk16 wf[18][2];
#pragma unroll
  for (int s = 0; s < 18; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[s][j] = ldg16(wsh + (long long)(32 * j + pl) * 288 + (2 * s + g) * 8);
  int rowoff[35];
#pragma unroll
  for (int r = 0; r < 35; ++r) rowoff[r] = ((r / 7) * G.SH + (r % 7)) * G.pitch;
  const float b0 = bias ? bias[pl] : 0.f, b1 = bias ? bias[32 + pl] : 0.f;
  const uint32_t fe0 = gamma[pl] < 0.f ? 0x80008000u : 0u, fe1 = gamma[32 + pl] < 0.f ? 0x80008000u : 0u;      // epilogue: sign flip of this lane's two channels
  // pool stage: this thread's channel, its sign flip (both halves of a word) and its pooled columns
  const int pc_ = tid & 63, qg = tid >> 6;
  const uint32_t flip2 = gamma[pc_] < 0.f ? 0x80008000u : 0u;
  const int pw0 = qg * S3P_PWG;
  float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};
  // measurement aid (abl & 64): shader-clock totals per phase of workgroup 0 / wave 0 -> stats[4096 + k] (tools/bench_stem_abl.py prints them)
  long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define S3P_T() ((S3P_STAMPS && (abl & 64)) ? (long long)clock64() : 0ll)
  for (long long item = blockIdx.x; item < G.items; item += gridDim.x) {
    const long long tq0 = S3P_T();
    const int band = (int)(item % G.NB); const long long cf = item / G.NB;
    const int fr = (int)(cf % G.T3); const long long clip = cf / G.T3;
    const int ph0 = band * G.pn; const int pnb = min(G.pn, G.PH - ph0);
    const int cr0 = max(0, 2 * ph0 - 1), cr1 = min(G.OH - 1, 2 * (ph0 + pnb - 1) + 1);
    const int ncr = cr1 - cr0 + 1, npx = ncr * G.OW;
    const int own0 = (2 * ph0 - cr0) * G.OW;                        // pixels before this one belong to the previous band's statistics
    __syncthreads();                                                // the previous band's slab and ring are no longer read
    // (round 6, measured and dropped: consecutive frames of a band with a ring of six frame slots and the next frame's rows prefetched -- the wave that waits here 17 % of its
    //  time, S3P_STAMPS, is covered by the CU's other workgroup: 562-566 us either way)
    if (!(abl & 16)) s3p_issue_slab(slab, vb, clip, fr, 2 * cr0 - 3, G);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ph_t[0] += S3P_T() - tq0;
    int rb = 32 * wave; if (rb >= G.RING) rb -= G.RING;            // ring slot of the wave's first pixel of the current tile
    int next_pool = 0;
    for (int t0 = 0; t0 < npx; t0 += 128) {
      const long long tq1 = S3P_T();
      const int p = t0 + 32 * wave + pl; const int pc = p < npx ? p : 0;
      const int ohl = pc / G.OW, ow = pc - ohl * G.OW;
      f32x16 acc[2];
      if (abl & 2) { for (int r = 0; r < 16; ++r) { acc[0][r] = (float)(t0 + r); acc[1][r] = (float)(lane + r); } }
      else s3p_conv_tile(slab, ohl, ow, g, wf, rowoff, G, acc);
      const long long tq2 = S3P_T();
      __syncthreads();                                              // the previous iteration's pool reads are done: ring slots may be overwritten
      const long long tq3 = S3P_T();
      if (abl & 4) { if (acc[0][3] + acc[1][5] == 1234.5f) st1[0] += 1.f; } else
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float bb = j ? b1 : b0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {                               // register quad a = pixels 8 a + 4 g .. + 3 of the wave's 32 (npx, own0, RING are multiples of 4)
          const int q4 = 8 * a + 4 * g; const int pr = t0 + 32 * wave + q4;
          const float v0 = acc[j][4 * a] + bb, v1 = acc[j][4 * a + 1] + bb, v2 = acc[j][4 * a + 2] + bb, v3 = acc[j][4 * a + 3] + bb;
          if (pr < npx) {
            if (pr >= own0) { st1[j] += (v0 + v1) + (v2 + v3); st2[j] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3); }
            int slot = rb + q4; if (slot >= G.RING) slot -= G.RING;
            const uint32_t fe = j ? fe1 : fe0;
            *(uint2*)(ring + ((32 * j + pl) * G.RING + slot) * 2) = make_uint2(s3p_key2(f32x2_to_bf16x2(v0, v1) ^ fe), s3p_key2(f32x2_to_bf16x2(v2, v3) ^ fe));
          }
        }
      }
      rb += 128; while (rb >= G.RING) rb -= G.RING;
      const long long tq4 = S3P_T();
      __syncthreads();
      const long long tq5 = S3P_T();
      const int done = min(t0 + 128, npx);
      while (!(abl & 8) && next_pool < pnb && min((2 * (ph0 + next_pool) + 2 - cr0) * G.OW, npx) <= done) {
        const int ph = ph0 + next_pool;
        uint32_t best[S3P_PWG];
#pragma unroll
        for (int i = 0; i < S3P_PWG; ++i) best[i] = 0u;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int c = 2 * ph + kh - 1;
          if (c < 0 || c >= G.OH) continue;                         // (uniform)
          int rr = c - cr0; while (rr >= 6) rr -= 6;
          const uint32_t* rowp = (const uint32_t*)(ring + (pc_ * G.RING + rr * G.OW) * 2);
          uint32_t D[S3P_PWG + 1];                                  // D[i] = columns (2 (pw0 - 1 + i), + 1)
#pragma unroll
          for (int i = 0; i <= S3P_PWG; ++i) { const int k = pw0 - 1 + i; D[i] = rowp[(k >= 0 && 2 * k < G.OW) ? k : 0]; }      // (keys: s3p_key2 in the epilogue)
#pragma unroll
          for (int i = 0; i < S3P_PWG; ++i) {
            if (pw0 + i > 0) best[i] = max(best[i], (D[i] & 0xffff0000u) | (15u - (kh * 3)));              // column 2 pw - 1
            best[i] = max(best[i], (D[i + 1] << 16) | (15u - (kh * 3 + 1)));                                 // column 2 pw
            best[i] = max(best[i], (D[i + 1] & 0xffff0000u) | (15u - (kh * 3 + 2)));                         // column 2 pw + 1
          }
        }
        const long long po = ((cf * G.PH + ph) * (long long)G.PW + pw0) * S3P_C + pc_;
#pragma unroll
        for (int i = 0; i < S3P_PWG; ++i) {
          if (pw0 + i >= G.PW) break;
          const uint32_t k = best[i] & 0xffff0000u, q = 15u - (best[i] & 15u);
          const uint32_t m = ~(uint32_t)((int)k >> 31);             // the key's top bit is set for values that were >= +0
          const uint32_t x = (k ^ (m | 0x80000000u)) ^ flip2;
          ((unsigned short*)zp)[po + (long long)i * S3P_C] = (unsigned short)(x >> 16);
          idx[po + (long long)i * S3P_C] = (unsigned char)q;
        }
        ++next_pool;
      }
      if (S3P_STAMPS && (abl & 64)) { const long long tq6 = S3P_T(); ph_t[1] += tq2 - tq1; ph_t[2] += tq3 - tq2; ph_t[3] += tq4 - tq3; ph_t[4] += tq5 - tq4; ph_t[5] += tq6 - tq5; ph_t[6] += 1; }
    }
  }
#undef S3P_T
  if (S3P_STAMPS && (abl & 64) && blockIdx.x == 0 && tid == 0 && stats) for (int k = 0; k < 8; ++k) stats[4096 + k] = (float)ph_t[k];
  if (!want_stats) return;
#pragma unroll
  for (int j = 0; j < 2; ++j) { st1[j] += __shfl_xor(st1[j], 32, 64); st2[j] += __shfl_xor(st2[j], 32, 64); }
  __syncthreads();
  if (g == 0) { atomicAdd(lsum + pl, st1[0]); atomicAdd(lsum + 32 + pl, st1[1]); atomicAdd(lsum + 64 + pl, st2[0]); atomicAdd(lsum + 96 + pl, st2[1]); }
  __syncthreads();
  if (tid < 128) {
    if (ws.partial) ws_slot(ws, 0, blockIdx.x, gridDim.x, 128)[tid] = lsum[tid];
    else atomicAdd(stats + tid, lsum[tid]);
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dz[m][c] = gamma rstd ( dr - mean(dy) - xhat mean(dy xhat) ),  dr = the pooled gradients routed to the window winners (ReLU mask already applied to dpool),
// xhat from the RECOMPUTED conv output.  Same tiles as the forward kernel (bands without the overlap row); stage 2 (thread = pixel x 8 channels) gathers the <= 4
// candidate windows from dpool / idx and writes 16 bytes of dz.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void s3p_dr8(const bf16* __restrict__ dp, const unsigned char* __restrict__ idx, long long fr, int h, int w, int c, int PH, int PW, float dr[8]) {
  const int ohA = h >> 1, owA = w >> 1, khA = 1 + (h & 1), kwA = 1 + (w & 1);
  int oh[2], ow[2], kh[2], kw[2]; bool vh[2], vw[2];
  oh[0] = ohA; kh[0] = khA; vh[0] = ohA < PH; oh[1] = ohA + 1; kh[1] = 0; vh[1] = (h & 1) && ohA + 1 < PH;
  ow[0] = owA; kw[0] = kwA; vw[0] = owA < PW; ow[1] = owA + 1; kw[1] = 0; vw[1] = (w & 1) && owA + 1 < PW;
  uint2 sel[4]; float gq[4][8]; unsigned slot[4]; bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = q >> 1, b = q & 1;
    ok[q] = vh[a] && vw[b]; slot[q] = (unsigned)(kh[a] * 3 + kw[b]);
    const long long o = ((fr * PH + (vh[a] ? oh[a] : 0)) * PW + (vw[b] ? ow[b] : 0)) * S3P_C + c;
    sel[q] = *(const uint2*)(idx + o); ld8<bf16>(dp + o, gq[q]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) dr[e] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ok[q] && ((sel[q].x >> (8 * e)) & 255u) == slot[q]) dr[e] += gq[q][e];
      if (ok[q] && ((sel[q].y >> (8 * e)) & 255u) == slot[q]) dr[4 + e] += gq[q][4 + e];
    }
}

// z^T tile (channels x pixels): the same operands with the roles swapped -- acc[j][r]: channel 32 j + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), pixel (lane & 31) of the
// wave's 32: a lane owns 4 consecutive channels per register quad (one 8-byte store into a [pixel][channel] tile)
__device__ __forceinline__ void s3p_conv_tile_t(const char* slab, int ohl, int ow, int g, const chunk16 (&wf)[18][2], const int (&rowoff)[35], const S3P& G, f32x16 (&acc)[2]) {
  const char* pix = slab + (2 * ohl) * G.pitch + 4 * ow + 8;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  chunk16 fa[18];
  auto ldfrag = [&](int s) {
    const int offA = rowoff[2 * s > 34 ? 34 : 2 * s], offB = rowoff[2 * s + 1 > 34 ? 34 : 2 * s + 1];
    const uint32_t* rp = (const uint32_t*)(pix + (g ? offB : offA));
    chunk16 f; f.w[0] = rp[0]; f.w[1] = rp[1]; f.w[2] = rp[2]; f.w[3] = rp[3]; return f;
  };
#pragma unroll
  for (int s = 0; s < 3; ++s) fa[s] = ldfrag(s);
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    if (s + 3 < 18) fa[s + 3] = ldfrag(s + 3);
    asm volatile("" ::: "memory");
    acc[0] = s3p_mma(wf[s][0], fa[s], acc[0]);
    acc[1] = s3p_mma(wf[s][1], fa[s], acc[1]);
  }
}

__global__ __launch_bounds__(256, 2) void stem3p_dz_kernel(const bf16* __restrict__ vb, const bf16* __restrict__ wsh, const float* __restrict__ bias, const bf16* __restrict__ dpm,
                                                            const unsigned char* __restrict__ idx, const float* __restrict__ ss, const float* __restrict__ gamma,
                                                            const float* __restrict__ dstats, const float* count_ptr, float count, bf16* __restrict__ dz,
                                                            float* dgamma, float* dbeta, S3P G) {
  extern __shared__ __attribute__((aligned(16))) char s3[];
  char* slab = s3;
  char* tile = s3 + G.slab_bytes;                                   // [128 px][64 ch] bf16 (conv output WITHOUT the bias), 16-byte chunk c of row m at slot c ^ ((m >> 1) & 7)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, pl = lane & 31;
  if (blockIdx.x == 0 && dgamma && tid < S3P_C) { atomicAdd(dgamma + tid, dstats[S3P_C + tid]); atomicAdd(dbeta + tid, dstats[tid]); }
  chunk16 wf[18][2];
#pragma unroll
  for (int s = 0; s < 18; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[s][j] = ldg16(wsh + (long long)(32 * j + pl) * 288 + (2 * s + g) * 8);
  int rowoff[35];
#pragma unroll
  for (int r = 0; r < 35; ++r) rowoff[r] = ((r / 7) * G.SH + (r % 7)) * G.pitch;
  // stage-2 constants of this thread's 8 channels: dz = A (dr - B - (acc - mu') Cc),  mu' = mean - bias
  const int cg = tid & 7;
  const float inv_n = 1.f / (count_ptr ? *count_ptr : count);
  float cA[8], cB[8], cC[8], cMu[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cg * 8 + e; const float rs = ss[3 * S3P_C + c];
    cA[e] = gamma[c] * rs; cB[e] = dstats[c] * inv_n; cC[e] = dstats[S3P_C + c] * inv_n * rs; cMu[e] = ss[2 * S3P_C + c] - (bias ? bias[c] : 0.f);
  }
  const int trow = 32 * wave + pl;                                  // this lane's pixel row of the tile (epilogue)
  char* const tw = tile + trow * 128; const int tsw = (trow >> 1) & 7;
  for (long long item = blockIdx.x; item < G.items; item += gridDim.x) {
    const int band = (int)(item % G.NB); const long long cf = item / G.NB;
    const int fr = (int)(cf % G.T3); const long long clip = cf / G.T3;
    const int ph0 = band * G.pn; const int pnb = min(G.pn, G.PH - ph0);
    const int cr0 = 2 * ph0, cr1 = min(G.OH - 1, 2 * (ph0 + pnb) - 1);
    const int ncr = cr1 - cr0 + 1, npx = ncr * G.OW;
    __syncthreads();                                                // the previous band's slab is no longer read
    s3p_issue_slab(slab, vb, clip, fr, 2 * cr0 - 3, G);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long mbase = (cf * G.OH + cr0) * (long long)G.OW;    // first row of this band in dz
    for (int t0 = 0; t0 < npx; t0 += 128) {
      const int p = t0 + trow; const int pc = p < npx ? p : 0;
      const int ohl = pc / G.OW, ow = pc - ohl * G.OW;
      f32x16 acc[2];
      s3p_conv_tile_t(slab, ohl, ow, g, wf, rowoff, G, acc);
      __syncthreads();                                              // the previous tile's stage 2 is done
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < 4; ++a)                                 // channels 32 j + 8 a + 4 g .. + 3 = half g of chunk 4 j + a
          *(uint2*)(tw + (((4 * j + a) ^ tsw) << 4) + 8 * g) = make_uint2(f32x2_to_bf16x2(acc[j][4 * a], acc[j][4 * a + 1]), f32x2_to_bf16x2(acc[j][4 * a + 2], acc[j][4 * a + 3]));
      __syncthreads();
#pragma unroll 1
      for (int u = 0; u < 4; u += 2) {                              // two pixels' gathers in flight per thread
        float dr[2][8]; int prow[2]; bool pv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = ((u + k) * 256 + tid) >> 3; prow[k] = row;
          const int pp = t0 + row; pv[k] = pp < npx; const int ppc = pv[k] ? pp : 0;
          const int hl = ppc / G.OW, w = ppc - hl * G.OW;
          s3p_dr8(dpm, idx, cf, cr0 + hl, w, cg * 8, G.PH, G.PW, dr[k]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (!pv[k]) continue;
          const uint4 t = *(const uint4*)(tile + prow[k] * 128 + ((cg ^ ((prow[k] >> 1) & 7)) << 4)); const uint32_t wv[4] = {t.x, t.y, t.z, t.w};
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float zl = __uint_as_float(wv[e] << 16), zh = __uint_as_float(wv[e] & 0xffff0000u);
            o[2 * e] = cA[2 * e] * (dr[k][2 * e] - cB[2 * e] - (zl - cMu[2 * e]) * cC[2 * e]);
            o[2 * e + 1] = cA[2 * e + 1] * (dr[k][2 * e + 1] - cB[2 * e + 1] - (zh - cMu[2 * e + 1]) * cC[2 * e + 1]);
          }
          st8<bf16>(dz + (mbase + t0 + prow[k]) * S3P_C + cg * 8, o);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fused backward (round 3): weight gradient of the stem straight from the pooled gradients -- neither z nor dz ever exist in memory.
//   dW[c][k] = sum_m dz[m][c] patch[m][k],   dz = gamma rstd (route(dpool) - mean(dy) - xhat mean(dy xhat)),  xhat from the RECOMPUTED conv output
// One 8-wave workgroup per CU, persistent over units (clip, band, chunk of CH consecutive frames):
//   * conv part: wave w computes z^T of pixels 32 (w >> 1) .. + 32 x channels 32 (w & 1) .. + 32 of a 128-pixel tile by MFMA (its half of the weights in registers:
//     the whole set costs 144 VGPRs, too many beside the weight-gradient accumulators) -> bf16 tile in LDS ([pixel][channel], the layout the transposed LDS reads want);
//   * stage 2 (thread = pixel x 8 channels) turns the tile into dz in place: the <= 4 candidate windows of every pixel are looked up in the band's pooled
//     gradients / window slots, which were copied to LDS by the DMA one frame ahead;
//   * wgrad part: D[k][c] += A^T (patches, 2-byte LDS gathers) x dz tile (ds_read_b64_tr_b16), wave w owns taps [32 w, 32 w + 32) x 64 channels in registers.
//   The conv of tile t + 1 is issued beside the weight-gradient product of tile t (two tiles ping-pong; two waves per SIMD fill each other's gaps).
// Frames of a unit are consecutive: the 5-frame input window lives in a ring of 6 frame slots, the DMA of frame f + 3 and of the pooled band of frame f + 1 is issued
// while frame f is processed (raw s_barrier + lgkmcnt waits inside a frame: a __syncthreads() would drain the DMA).
// ------------------------------------------------------------------------------------------------
struct S3W { int SLOT, DPB, IXB, CH, NCH; long long units; };      // bytes of a frame slot / pooled-gradient buffer / window-slot buffer; frames per unit; chunks per clip

#define S3P_BAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

__device__ __forceinline__ chunk16 s3p_tr8(const char* p0, const char* p1) {
  typedef short v4s_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4s_t* lp_t;
  const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p1);
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  chunk16 f; f.w[0] = ua.x; f.w[1] = ua.y; f.w[2] = ub.x; f.w[3] = ub.y; return f;
}

// LDS-DMA hidden from hipcc (cdna_hip_programming.md 5.7): the compiler otherwise drains the queue (s_waitcnt vmcnt(0)) in front of the next LDS access it cannot
// prove disjoint from the DMA target, i.e. right behind the prefetch.  Completion is counted by hand: one vmcnt(0) + barrier at the end of a frame.
__device__ __forceinline__ void s3p_glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// Pixel order inside a band (all three parts agree on it): a STEP is 16 pixels = 4 rows x 4 columns of the conv output, k = 4 rho + kappa; step S covers rows 4 R .. 4 R + 3 and
// columns 4 Cg .. 4 Cg + 3, (R, Cg) = (S / ncg, S % ncg), ncg = OW / 4; a tile is 8 steps (tile row m = 16 s + k).  With this order the gather address of a lane's 8 pixels of a
// step is  (tap offset + g * 4 pitch)  +  (R * 8 pitch + Cg * 16: wave-uniform)  +  ((e >> 2) * 2 pitch + (e & 3) * 4: immediates when the pitch is a compile-time constant):
// one VALU add per step instead of ~50 (at ~4 cycles per wave64 VALU instruction and 2 waves per SIMD only ~8 VALU instructions hide under one MFMA).
// (The eight-wave lockstep version of this kernel -- conv -> barrier -> stage 2 -> barrier -> product, all waves together: 1 018 us against 936 us with the two wave roles
// below -- was removed in round 6)
template <int PITCH_CT>
__global__ __launch_bounds__(512) void stem3p_wgrad_roles_kernel(const bf16* __restrict__ vb, const bf16* __restrict__ wsh, const float* __restrict__ bias, const bf16* __restrict__ dpm,
                                                            const unsigned char* __restrict__ idx, const float* __restrict__ ss, const float* __restrict__ gamma,
                                                            const float* __restrict__ dstats, const float* count_ptr, float count, float* dw, float* dgamma, float* dbeta,
                                                            S3P G, S3W Q, ColWs ws, int abl) {
  extern __shared__ __attribute__((aligned(16))) char s3[];
  char* const slab = s3;                                            // [6 slots][SLOT]: rows of pitch bytes, SH rows per frame
  char* const dpl = slab + 6 * Q.SLOT;                              // [2][DPB]: pooled gradients (masked) of the band's pn + 1 pooled rows, [row][pw][64] bf16
  char* const ixl = dpl + 2 * Q.DPB;                                // [2][IXB]: their window slots, [row][pw][64] u8
  char* const tiles = ixl + 2 * Q.IXB;                              // [2][128 px][64 ch] bf16, chunk c of row m at slot c ^ (4 * ((m >> 1) & 1))
  const int pitch = PITCH_CT ? PITCH_CT : G.pitch;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5, pl = lane & 31;
  // Round 5: two wave ROLES instead of eight waves walking through conv -> barrier -> stage 2 -> barrier -> product together (ablation: the three parts take 272 / 314 /
  // 247 us of 1002 and ADD UP -- the barriers serialise an MFMA phase, a VALU phase and another MFMA phase).  Waves 0-3 (producers) each own 32 pixels of a tile over
  // all 64 channels: recomputed convolution (both weight halves in registers) -> z^T rows in LDS -> stage 2 on their OWN rows (a wave-local dependency: no barrier);
  // waves 4-7 (consumers) each own 64 taps x 64 channels of the weight gradient and multiply the tile the producers finished one barrier ago.  One barrier per tile;
  // a SIMD holds one wave of each role, so the producers' VALU phase runs under the consumers' MFMAs.
  const bool producer = wave < 4;
  const int pgp = wave & 3;                                         // producer: pixel group; consumer: tap group pair
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)s3;
  if (blockIdx.x == 0 && dgamma && tid < S3P_C) { atomicAdd(dgamma + tid, dstats[S3P_C + tid]); atomicAdd(dbeta + tid, dstats[tid]); }
  if (tid < S3P_C) {
    float* const cst0 = (float*)(ixl + 2 * Q.IXB + 2 * 16384);
    const float inv_n0 = 1.f / (count_ptr ? *count_ptr : count);
    const float rs = ss[3 * S3P_C + tid];
    cst0[tid] = gamma[tid] * rs; cst0[64 + tid] = dstats[tid] * inv_n0; cst0[128 + tid] = dstats[S3P_C + tid] * inv_n0 * rs; cst0[192 + tid] = ss[2 * S3P_C + tid] - (bias ? bias[tid] : 0.f);
  }
  __syncthreads();
  // Each role's state lives inside ITS instantiation of the main loop: declared at kernel scope, the producers' 144 weight registers and the consumers' 64
  // accumulators are both live everywhere (the compiler cannot tell that `wave < 4` is the same condition at every use) and spill.
  auto role_main = [&](auto prodc) {
  constexpr bool PROD = decltype(prodc)::value;
  // ---- conv part: weights of all 64 channels (MFMA A operand: row = channel) ----
  chunk16 wf[18][2];
  if constexpr (PROD) {
#pragma unroll
    for (int s = 0; s < 18; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[s][j] = ldg16(wsh + (long long)(32 * j + pl) * 288 + (2 * s + g) * 8);
  }
  // ---- wgrad part: accumulators (this lane's tap x 64 channels), transposed-read offsets of the dz tile ----
  f32x16 acc[2][2];                                                 // consumer: [tap group][channel half]
  if constexpr (!PROD) {
#pragma unroll
    for (int tg = 0; tg < 2; ++tg)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tg][j][r] = 0.f;
  }
  int tkd[2], toff[2];                                              // tap -> frame offset kd and byte offset inside a frame slot (row kh, element kw + 1 behind the 8-byte left pad, rows 2 g, 2 g + 1 of the step)
#pragma unroll
  for (int tg = 0; tg < 2; ++tg) { const int k = 64 * pgp + 32 * tg + pl; const int r = k / 7 > 34 ? 34 : k / 7; tkd[tg] = r / 7; toff[tg] = (r % 7) * pitch + (k % 7 + 1) * 2 + 8 + g * 4 * pitch; }
  int eoff[8];                                                      // (generic pitch) offsets of the lane's 8 pixels inside a step
#pragma unroll
  for (int e = 0; e < 8; ++e) eoff[e] = (e >> 2) * 2 * pitch + (e & 3) * 4;
  const int g4 = lane >> 4, t16 = lane & 15;
  int offb[2][2];
#pragma unroll
  for (int hh2 = 0; hh2 < 2; ++hh2) {
    const int row = 8 * (g4 >> 1) + 4 * hh2 + (t16 >> 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int cb = 32 * j + 16 * (g4 & 1); offb[j][hh2] = row * 128 + ((((cb >> 3) + ((t16 & 3) >> 1)) ^ (4 * ((row >> 1) & 1))) << 4) + (t16 & 1) * 8; }
  }
  // ---- stage-2 constants of this thread's 8 channels: dz = A (dr - B - (acc - mu') Cc),  mu' = mean - bias ----
  const int cg = tid & 7;
  // (the four per-channel constants live in LDS -- cst[4][64] behind the tiles, written once by the first 64 threads -- and are fetched at the start of every stage 2:
  // 32 registers the convolution part, which holds 144 weight registers, does not have)
  float* const cst = (float*)(tiles + 2 * 16384);
  const int trow = 32 * pgp + pl;                                   // conv part: this lane's row of a tile = step 2 pgp + (pl >> 4), k = pl & 15
  const int tsw = 4 * ((trow >> 1) & 1);
  const int ck = pl & 15, crho = ck >> 2, ckap = ck & 3, cstep = 2 * pgp + (pl >> 4);
  // stage 2: task u of wave w handles the 8 tile rows of parity class c (row parity rp, column parity cp) in steps 2 q, 2 q + 1; the classes of a wave's two tasks are
  // complementary ((0,0) + (1,1): 1 + 4 candidate windows, (0,1) + (1,0): 2 + 2), so every wave does about the same work and the window loops are wave-uniform
  const int ncg = G.OW >> 2;
  const unsigned ncg_m = ((1u << 20) + (unsigned)ncg - 1u) / (unsigned)ncg;      // S / ncg = (S * ncg_m) >> 20 for the step indices of a band (S < 4096, ncg <= 64: exact) -- a per-lane integer division is ~35 VALU instructions

  // one frame slot: chunk q = tid (SLOT = 512 chunks >= SH * CPR) <- input row ir0 + q / CPR of frame `itf`, chunk q % CPR (zero page: padding chunks, rows outside the frame)
  auto dma_frame = [&](long long clip, int itf, int ir0, int slot) {
    const int row = tid / G.CPR, j = tid - row * G.CPR; const int ih = ir0 + row;
    const bool ok = row < G.SH && itf >= 0 && itf < G.T3 && ih >= 0 && ih < G.H && j >= 1 && j <= G.CPR - 2;
    const void* sp = ok ? (const void*)(vb + ((clip * G.T3 + itf) * (long long)G.H + ih) * G.W + (j - 1) * 8) : (const void*)s3p_zero16;
    s3p_glds16(sp, lds0 + slot * Q.SLOT + wave * 1024);
  };
  // the band's pooled rows ph0 .. ph0 + pn of frame cf: contiguous in memory; rows beyond PH come from the zero page
  auto dma_pooled = [&](long long cf, int ph0, int buf) {
    const int nrow = G.pn + 1;
    const long long dp0 = ((cf * G.PH + ph0) * (long long)G.PW) * S3P_C;     // element offset of the first pooled pixel (bf16 / u8 alike)
    const int valid_px = (min(G.PH, ph0 + nrow) - ph0) * G.PW;
    for (int q0 = wave * 64; q0 < Q.DPB / 16; q0 += 512) {          // 8 chunks of 16 B per pooled pixel
      const int q = q0 + lane;
      const void* sp = (q >> 3) < valid_px ? (const void*)(dpm + dp0 + (long long)q * 8) : (const void*)s3p_zero16;
      s3p_glds16(sp, lds0 + 6 * Q.SLOT + buf * Q.DPB + q0 * 16);
    }
    for (int q0 = wave * 64; q0 < Q.IXB / 16; q0 += 512) {          // 4 chunks per pooled pixel
      const int q = q0 + lane;
      const void* sp = (q >> 2) < valid_px ? (const void*)(idx + dp0 + (long long)q * 16) : (const void*)s3p_zero16;
      s3p_glds16(sp, lds0 + 6 * Q.SLOT + 2 * Q.DPB + buf * Q.IXB + q0 * 16);
    }
  };

  for (long long unit = blockIdx.x; unit < Q.units; unit += gridDim.x) {
    const int chunk = (int)(unit % Q.NCH); const int band = (int)((unit / Q.NCH) % G.NB); const long long clip = unit / ((long long)Q.NCH * G.NB);
    const int f0 = chunk * Q.CH, f1 = min(G.T3, f0 + Q.CH);
    const int ph0 = band * G.pn; const int pnb = min(G.pn, G.PH - ph0);
    const int cr0 = 2 * ph0, cr1 = min(G.OH - 1, 2 * (ph0 + pnb) - 1);
    const int ncr = cr1 - cr0 + 1, NS = ((ncr + 3) >> 2) * ncg, ntile = (NS + 7) >> 3;
    const int ir0 = 2 * cr0 - 3;
    __syncthreads();                                                // the previous unit is done with every buffer (nothing is in flight: it ended with vmcnt(0))
#pragma unroll 1
    for (int d = -2; d <= 2; ++d) { int sl = (f0 + d) % 6; if (sl < 0) sl += 6; dma_frame(clip, f0 + d, ir0, sl); }
    dma_pooled(clip * G.T3 + f0, ph0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int f = f0; f < f1; ++f) {
      const long long cf = clip * G.T3 + f;
      const int pb = (f - f0) & 1;
      if (f + 1 < f1 && !(abl & 8)) { dma_frame(clip, f + 3, ir0, (f + 3) % 6); dma_pooled(cf + 1, ph0, pb ^ 1); }      // one frame ahead (their buffers were last read in frame f - 1)
      int fb[5];                                                    // byte offset of the slot of frame f + kd - 2
#pragma unroll
      for (int kd = 0; kd < 5; ++kd) { int sl = (f + kd - 2) % 6; if (sl < 0) sl += 6; fb[kd] = sl * Q.SLOT; }
      const char* dpb = dpl + pb * Q.DPB; const char* ixb = ixl + pb * Q.IXB;

      auto conv_tile = [&](int t) {                                 // raw z^T of tile t -> tiles[t & 1]
        int S = 8 * t + cstep; if (S >= NS) S = 0;
        const int R = (int)(((unsigned)S * ncg_m) >> 20), Cg = S - R * ncg;
        const char* pix = slab + (2 * (4 * R + crho)) * pitch + 4 * (4 * Cg + ckap) + 8;
        f32x16 z[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { z[0][r] = 0.f; z[1][r] = 0.f; }
        chunk16 fa[18];
        auto ldfrag = [&](int s) {
          const int ra = 2 * s > 34 ? 34 : 2 * s, rb_ = 2 * s + 1 > 34 ? 34 : 2 * s + 1;
          const int offA = fb[ra / 7] + (ra % 7) * pitch, offB = fb[rb_ / 7] + (rb_ % 7) * pitch;
          const uint32_t* rp = (const uint32_t*)(pix + (g ? offB : offA));
          chunk16 fr; fr.w[0] = rp[0]; fr.w[1] = rp[1]; fr.w[2] = rp[2]; fr.w[3] = rp[3]; return fr;
        };
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[s] = ldfrag(s);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
          if (s + 4 < 18) fa[s + 4] = ldfrag(s + 4);
          asm volatile("" ::: "memory");
          asm volatile("" : "+v"(fa[s].w[0]), "+v"(fa[s].w[1]), "+v"(fa[s].w[2]), "+v"(fa[s].w[3]));      // (the use stays below the requests of step s + 4)
          z[0] = s3p_mma(wf[s][0], fa[s], z[0]);
          z[1] = s3p_mma(wf[s][1], fa[s], z[1]);
        }
        char* tw = tiles + (t & 1) * 16384 + trow * 128;
#pragma unroll
        for (int chh = 0; chh < 2; ++chh)
#pragma unroll
          for (int a = 0; a < 4; ++a)                               // channels 32 chh + 8 a + 4 g .. + 3 = half g of chunk 4 chh + a
            *(uint2*)(tw + (((4 * chh + a) ^ tsw) << 4) + 8 * g) = make_uint2(f32x2_to_bf16x2(z[chh][4 * a], z[chh][4 * a + 1]), f32x2_to_bf16x2(z[chh][4 * a + 2], z[chh][4 * a + 3]));
      };

      auto stage2 = [&](int t) {                                    // tiles[t & 1]: z -> dz in place (rows beyond the band: zero)
        char* tb = tiles + (t & 1) * 16384;
        float cA[8], cB[8], cC[8], cMu[8];
        { const float4* c4 = (const float4*)(cst + cg * 8);
          const float4 a0 = c4[0], a1 = c4[1], b0 = c4[16], b1 = c4[17], c0 = c4[32], c1 = c4[33], m0 = c4[48], m1 = c4[49];
          cA[0] = a0.x; cA[1] = a0.y; cA[2] = a0.z; cA[3] = a0.w; cA[4] = a1.x; cA[5] = a1.y; cA[6] = a1.z; cA[7] = a1.w;
          cB[0] = b0.x; cB[1] = b0.y; cB[2] = b0.z; cB[3] = b0.w; cB[4] = b1.x; cB[5] = b1.y; cB[6] = b1.z; cB[7] = b1.w;
          cC[0] = c0.x; cC[1] = c0.y; cC[2] = c0.z; cC[3] = c0.w; cC[4] = c1.x; cC[5] = c1.y; cC[6] = c1.z; cC[7] = c1.w;
          cMu[0] = m0.x; cMu[1] = m0.y; cMu[2] = m0.z; cMu[3] = m0.w; cMu[4] = m1.x; cMu[5] = m1.y; cMu[6] = m1.z; cMu[7] = m1.w; }
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {                               // the four parity classes of this producer's own 32 rows (steps 2 pgp, 2 pgp + 1)
          const int c = u; const int rp = c >> 1, cp = c & 1;       // (wave-uniform)
          const int q = pgp;
          const int j = lane >> 3;
          const int s = 2 * q + (j >> 2), rho = rp + 2 * ((j >> 1) & 1), kap = cp + 2 * (j & 1);
          const int row = 16 * s + 4 * rho + kap;
          const int S = 8 * t + s; const int R = (int)(((unsigned)S * ncg_m) >> 20), Cg = S - R * ncg;
          const int hl = 4 * R + rho, w = 4 * Cg + kap; const int h = cr0 + hl;
          const bool pv = S < NS && hl < ncr;
          char* zp_ = tb + row * 128 + ((cg ^ (4 * ((row >> 1) & 1))) << 4);
          float o[8];
          if (pv) {
            // candidate windows: rows oh = h >> 1 (slot kh = 1 + rp) and, for odd h, oh + 1 (kh = 0); likewise for columns
            const int ohA = h >> 1, owA = w >> 1;
            float dr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) dr[e] = 0.f;
            for (int a = 0; a <= rp; ++a)
              for (int b = 0; b <= cp; ++b) {
                const int oh = ohA + a, ow_ = owA + b;
                const bool ok = oh < G.PH && ow_ < G.PW;
                const unsigned slotq = (unsigned)((a ? 0 : 1 + rp) * 3 + (b ? 0 : 1 + cp));
                const int lp = ok ? ((oh - ph0) * G.PW + ow_) : 0;
                const uint2 sel = *(const uint2*)(ixb + lp * 64 + cg * 8);
                const uint4 gq = *(const uint4*)(dpb + lp * 128 + cg * 16); const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const unsigned sq = ((e < 4 ? sel.x : sel.y) >> (8 * (e & 3))) & 255u;
                  const float gv = (e & 1) ? __uint_as_float(gw[e >> 1] & 0xffff0000u) : __uint_as_float(gw[e >> 1] << 16);
                  if (ok && sq == slotq) dr[e] += gv;
                }
              }
            const uint4 tz = *(const uint4*)zp_; const uint32_t wv[4] = {tz.x, tz.y, tz.z, tz.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float zl = __uint_as_float(wv[e] << 16), zh = __uint_as_float(wv[e] & 0xffff0000u);
              o[2 * e] = cA[2 * e] * (dr[2 * e] - cB[2 * e] - (zl - cMu[2 * e]) * cC[2 * e]);
              o[2 * e + 1] = cA[2 * e + 1] * (dr[2 * e + 1] - cB[2 * e + 1] - (zh - cMu[2 * e + 1]) * cC[2 * e + 1]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;
          }
          *(uint4*)zp_ = make_uint4(f32x2_to_bf16x2(o[0], o[1]), f32x2_to_bf16x2(o[2], o[3]), f32x2_to_bf16x2(o[4], o[5]), f32x2_to_bf16x2(o[6], o[7]));
        }
      };

      auto wgrad_tile = [&](int t) {                                // D += A^T x dz of tiles[t & 1]
        const char* tb = tiles + (t & 1) * 16384;
        const char* gb[2] = {slab + fb[tkd[0]] + toff[0], slab + fb[tkd[1]] + toff[1]};      // this lane's taps (and row pair 2 g) at pixel (0, 0) of the band
        const int S0 = 8 * t; int R = S0 / ncg, Cg = S0 - R * ncg;  // (wave-uniform: scalar registers)
        uint32_t gl[2][2][8];
        auto gather = [&](int s, uint32_t (&o)[2][8]) {
          int Rs = R, Cs = Cg + s; while (Cs >= ncg) { Cs -= ncg; ++Rs; }
          const int so = Rs * 8 * pitch + Cs * 16;
#pragma unroll
          for (int tg = 0; tg < 2; ++tg)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[tg][e] = *(const unsigned short*)(gb[tg] + so + (PITCH_CT ? (e >> 2) * 2 * PITCH_CT + (e & 3) * 4 : eoff[e]));
        };
        const int nst = min(8, NS - S0);
        gather(0, gl[0]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {                               // 16 pixels per step: this lane's 8 = rows 2 g, 2 g + 1 x 4 columns
          if (s < nst) {
            const char* db = tb + (16 * s) * 128;
            const chunk16 fb0 = s3p_tr8(db + offb[0][0], db + offb[0][1]), fb1 = s3p_tr8(db + offb[1][0], db + offb[1][1]);
            if (s + 1 < 8 && s + 1 < nst) gather(s + 1, gl[(s + 1) & 1]);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
              for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(gl[s & 1][tg][e]));      // (packing stays below the next step's requests: one LDS round trip per step, not per pair)
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
              chunk16 fa2;
#pragma unroll
              for (int e = 0; e < 8; e += 2) fa2.w[e >> 1] = gl[s & 1][tg][e] | (gl[s & 1][tg][e + 1] << 16);
              acc[tg][0] = s3p_mma(fa2, fb0, acc[tg][0]); acc[tg][1] = s3p_mma(fa2, fb1, acc[tg][1]);
            }
          }
        }
      };

      auto produce = [&](int t) {
        if (!(abl & 1)) conv_tile(t);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's own rows: stage 2 reads what the same wave just wrote
        if (!(abl & 2)) stage2(t);
      };
      if constexpr (PROD) produce(0);
      S3P_BAR();
      for (int t = 0; t < ntile; ++t) {
        if constexpr (PROD) { if (t + 1 < ntile) produce(t + 1); }
        else { if (!(abl & 4)) wgrad_tile(t); }
        S3P_BAR();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // next frame's slab slot and pooled band have landed
      S3P_BAR();
    }
  }
  // D[k][c] -> dw[c][245] (fp32): through the workspace partial ([64][245] per workgroup) or atomics
  {
    float* mine = ws.partial ? ws_slot(ws, 0, blockIdx.x, gridDim.x, S3P_C * 245) : nullptr;
    if constexpr (!PROD) {
#pragma unroll
      for (int tg = 0; tg < 2; ++tg)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = 64 * pgp + 32 * tg + (r & 3) + 8 * (r >> 2) + 4 * g, c = 32 * j + pl;
            if (k < 245) { if (mine) mine[c * 245 + k] = acc[tg][j][r]; else atomicAdd(dw + c * 245 + k, acc[tg][j][r]); }
          }
    }
  }
  };
  if (producer) role_main(std::integral_constant<bool, true>{}); else role_main(std::integral_constant<bool, false>{});
}

// BatchNorm-backward statistics over the pooled domain with the ReLU mask taken from the pooled winner itself: live = scale * zp + shift > 0.
//   dstats[c] += sum d, dstats[C + c] += sum d * (zp - mean) * rstd,  d = live ? dp : 0;  dp is overwritten with d (the routing of stem3p_dz_kernel then needs no mask)
__global__ __launch_bounds__(256) void stem3p_reduce_kernel(bf16* __restrict__ dp, const bf16* __restrict__ zp, const float* __restrict__ ss, float* dstats, long long P, int C, ColWs ws) {
  const Col8 m = col8_map(C);
  float part[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) part[0][e] = part[1][e] = 0.f;
  if (m.active) {
    const int c = m.l * 8;
    float sc[8], sh[8], mu[8], rs[8]; ld8<float>(ss + c, sc); ld8<float>(ss + C + c, sh); ld8<float>(ss + 2 * C + c, mu); ld8<float>(ss + 3 * C + c, rs);
    for (long long row = (long long)blockIdx.x * m.R + m.r; row < P; row += (long long)gridDim.x * m.R) {
      float gq[8], v[8]; ld8<bf16>(dp + row * C + c, gq); ld8<bf16>(zp + row * C + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (v[e] * sc[e] + sh[e] > 0.f) ? gq[e] : 0.f;
        gq[e] = d; part[0][e] += d; part[1][e] += d * (v[e] - mu[e]) * rs[e];
      }
      st8<bf16>(dp + row * C + c, gq);
    }
  }
  float* const dst[2] = {dstats, dstats + C};
  colreduce8_atomic<2>(part, dst, m, ws);
}

// ------------------------------------------------------------------------------------------------
static bool s3p_geom(S3P& G, long long clips, int T_, int H, int W, size_t* lds_fwd, size_t* lds_bwd) {
  if (clips <= 0 || T_ <= 0 || H < 8 || W < 32 || (W & 7)) return false;
  G.T3 = T_; G.H = H; G.W = W; G.OH = (H - 1) / 2 + 1; G.OW = (W - 1) / 2 + 1; G.PH = (G.OH - 1) / 2 + 1; G.PW = (G.OW - 1) / 2 + 1;
  G.CPR = W / 8 + 2; G.pitch = G.CPR * 16;
  for (int nb = 1; nb <= G.PH; ++nb) {                              // fewest bands whose workgroup fits twice into a CU's LDS
    G.NB = nb; G.pn = (G.PH + nb - 1) / nb;
    if ((G.NB - 1) * G.pn >= G.PH) continue;                        // (an empty last band)
    G.SH = 2 * (2 * G.pn) + 7;
    G.nchunks = 5 * G.SH * G.CPR; G.slab_bytes = ((G.nchunks + 255) / 256) * 4096;
    G.RING = 6 * G.OW;                                             // >= 3 * OW + 128 checked below
    *lds_fwd = (size_t)G.slab_bytes + (size_t)G.RING * 128 + 512;
    *lds_bwd = (size_t)G.slab_bytes + (size_t)128 * 128;
    if (*lds_fwd <= 80 * 1024 - 512 && G.RING >= 3 * G.OW + 128 && G.PW <= 4 * S3P_PWG) break;
    if (nb == G.PH) return false;
  }
  G.items = clips * T_ * G.NB;
  return G.items < (1ll << 31) && clips * T_ * (long long)G.OH * G.OW * S3P_C < (1ll << 40);
}
template <typename K> static int s3p_set_lds(K kern) {
  static const void* done[4]; static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
  if (ndone < 4) done[ndone++] = (const void*)kern;
  return 0;
}

extern "C" int avec_stem3p_supported(long long clips, int T_, int H, int W) {
  S3P G; size_t a, b; return s3p_geom(G, clips, T_, H, W, &a, &b) ? 1 : 0;
}

extern "C" int avec_stem3p_fwd(const void* video_bf16, const void* w_shadow, const float* bias, const float* gamma, void* zp, unsigned char* idx, float* stats,
                               long long clips, int T_, int H, int W, hipStream_t st) {
  AVEC_CHECK_ARG(video_bf16 && w_shadow && gamma && zp && idx && (((size_t)video_bf16) & 15) == 0, "stem3p_fwd: null / unaligned pointer");
  S3P G; size_t lf, lb;
  AVEC_CHECK_ARG(s3p_geom(G, clips, T_, H, W, &lf, &lb), "stem3p_fwd: frame %dx%d not supported (W %% 8 == 0, W >= 32; use avec_stem_im2col + avec_gemm_nt)", H, W);
  if (int r = s3p_set_lds(stem3p_fwd_kernel)) return r;
  unsigned nb = 512; if ((long long)nb > G.items) nb = (unsigned)G.items;      // persistent: two workgroups per CU
  ColWs ws = stats ? avec_reduce_ws((size_t)nb * 128, st) : ColWs{nullptr};
  avec_note_kernel("stem3p_fwd_kernel");
  static const int abl = getenv("AVEC_S3P_ABL") ? atoi(getenv("AVEC_S3P_ABL")) : 0;      // kernel ablation (measurement only): 2 no conv tile, 4 no ring writes / statistics, 8 no pool stage, 16 no slab DMA
  hipLaunchKernelGGL(stem3p_fwd_kernel, dim3(nb), dim3(256), lf, st, (const bf16*)video_bf16, (const bf16*)w_shadow, bias, gamma, (bf16*)zp, idx, stats ? 1 : 0, G, ws, stats, abl);
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {stats, stats + S3P_C}; return col_finalize(ws, 1, nb, 2, S3P_C, dst, S3P_C, st); }
  return 0;
}

extern "C" int avec_stem3p_reduce(void* dpool, const void* zp, const float* ss, float* dstats, long long frames, int PH, int PW, hipStream_t st) {
  AVEC_CHECK_ARG(dpool && zp && ss && dstats && frames > 0 && PH > 0 && PW > 0, "stem3p_reduce: bad arguments");
  const long long P = frames * PH * PW;
  ColWs ws; const unsigned nb8 = col8_cfg(P, S3P_C, 2, &ws, st);
  hipLaunchKernelGGL(stem3p_reduce_kernel, dim3(nb8), dim3(256), 0, st, (bf16*)dpool, (const bf16*)zp, ss, dstats, P, S3P_C, ws);
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[2] = {dstats, dstats + S3P_C}; return col_finalize(ws, 1, nb8, 2, S3P_C, dst, S3P_C, st); }
  return 0;
}

static bool s3w_geom(const S3P& G, long long clips, S3W& Q, size_t* lds) {
  if (G.SH * G.CPR > 512 || (G.OW & 3)) return false;              // one DMA instruction per thread and frame slot; steps of 4 x 4 pixels
  Q.SLOT = 8192;
  Q.DPB = (((G.pn + 1) * G.PW * 128) + 1023) & ~1023; Q.IXB = (((G.pn + 1) * G.PW * 64) + 1023) & ~1023;
  Q.CH = 13; if (Q.CH > G.T3) Q.CH = G.T3;
  Q.NCH = (G.T3 + Q.CH - 1) / Q.CH;
  Q.units = clips * G.NB * Q.NCH;
  *lds = (size_t)6 * Q.SLOT + 2 * (size_t)(Q.DPB + Q.IXB) + 2 * 16384;
  return *lds <= 160 * 1024;
}

extern "C" int avec_stem3p_wgrad_supported(long long clips, int T_, int H, int W) {
  S3P G; S3W Q; size_t a, b, c; return (s3p_geom(G, clips, T_, H, W, &a, &b) && s3w_geom(G, clips, Q, &c)) ? 1 : 0;
}

extern "C" int avec_stem3p_wgrad(const void* video_bf16, const void* w_shadow, const float* bias, const void* dpool_masked, const unsigned char* idx, const float* ss,
                                 const float* gamma, const float* dstats, const float* count_ptr, float count, float* dw, float* dgamma, float* dbeta,
                                 long long clips, int T_, int H, int W, hipStream_t st) {
  AVEC_CHECK_ARG(video_bf16 && w_shadow && dpool_masked && idx && ss && gamma && dstats && dw, "stem3p_wgrad: null pointer");
  AVEC_CHECK_ARG(((((size_t)video_bf16) | ((size_t)dpool_masked) | ((size_t)idx)) & 15) == 0, "stem3p_wgrad: operands must be 16-byte aligned");
  S3P G; S3W Q; size_t lf, lb, lw;
  AVEC_CHECK_ARG(s3p_geom(G, clips, T_, H, W, &lf, &lb) && s3w_geom(G, clips, Q, &lw), "stem3p_wgrad: frame %dx%d not supported", H, W);
  unsigned nb = 256; if ((long long)nb > Q.units) nb = (unsigned)Q.units;
  ColWs ws = avec_reduce_ws((size_t)nb * S3P_C * 245, st);
  avec_note_kernel("stem3p_wgrad_roles_kernel");
  static const int abl = getenv("AVEC_S3W_ABL") ? atoi(getenv("AVEC_S3W_ABL")) : 0;      // kernel ablation (measurement only): 1 no conv part, 2 no stage 2, 4 no weight-gradient part, 8 no DMA prefetch
  { static bool attr2 = false;
    if (!attr2) {
      hipError_t e = hipFuncSetAttribute((const void*)stem3p_wgrad_roles_kernel<208>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem3p_wgrad_roles_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
      attr2 = true;
    }
    const size_t lw2 = lw + 1024;             // + the stage-2 constant table
    AVEC_CHECK_ARG(lw2 <= 160 * 1024, "stem3p_wgrad: LDS");
    if (G.pitch == 208)      // (W = 88: the gather offsets of a step are instruction immediates)
      hipLaunchKernelGGL(stem3p_wgrad_roles_kernel<208>, dim3(nb), dim3(512), lw2, st, (const bf16*)video_bf16, (const bf16*)w_shadow, bias, (const bf16*)dpool_masked, idx, ss, gamma,
                         dstats, count_ptr, count, dw, dgamma, dbeta, G, Q, ws, abl);
    else
      hipLaunchKernelGGL(stem3p_wgrad_roles_kernel<0>, dim3(nb), dim3(512), lw2, st, (const bf16*)video_bf16, (const bf16*)w_shadow, bias, (const bf16*)dpool_masked, idx, ss, gamma,
                         dstats, count_ptr, count, dw, dgamma, dbeta, G, Q, ws, abl);
  }
  AVEC_LAUNCH_CHECK();
  if (ws.partial) { float* const dst[1] = {dw}; return col_finalize(ws, 1, nb, 1, S3P_C * 245, dst, S3P_C * 245, st); }
  return 0;
}

extern "C" int avec_stem3p_dz(const void* video_bf16, const void* w_shadow, const float* bias, const void* dpool_masked, const unsigned char* idx, const float* ss,
                              const float* gamma, const float* dstats, const float* count_ptr, float count, void* dz, float* dgamma, float* dbeta,
                              long long clips, int T_, int H, int W, hipStream_t st) {
  AVEC_CHECK_ARG(video_bf16 && w_shadow && dpool_masked && idx && ss && gamma && dstats && dz, "stem3p_dz: null pointer");
  S3P G; size_t lf, lb;
  AVEC_CHECK_ARG(s3p_geom(G, clips, T_, H, W, &lf, &lb), "stem3p_dz: frame %dx%d not supported", H, W);
  if (int r = s3p_set_lds(stem3p_dz_kernel)) return r;
  avec_note_kernel("stem3p_dz_kernel");
  unsigned nb = 512; if ((long long)nb > G.items) nb = (unsigned)G.items;
  hipLaunchKernelGGL(stem3p_dz_kernel, dim3(nb), dim3(256), lb, st, (const bf16*)video_bf16, (const bf16*)w_shadow, bias, (const bf16*)dpool_masked, idx, ss, gamma,
                     dstats, count_ptr, count, (bf16*)dz, dgamma, dbeta, G);
  AVEC_LAUNCH_CHECK(); return 0;
}
