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
#include "vec.h"
#include "avec_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

static __device__ __attribute__((aligned(64))) unsigned char s3p_zero16[64];

static constexpr int S3P_C = 64;

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
  // pool stage: this thread's channel, its sign flip (both halves of a word) and its pooled columns
  const int pc_ = tid & 63, qg = tid >> 6;
  const uint32_t flip2 = gamma[pc_] < 0.f ? 0x80008000u : 0u;
  const int pw0 = qg * S3P_PWG;
  float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};
  for (long long item = blockIdx.x; item < G.items; item += gridDim.x) {
    const int band = (int)(item % G.NB); const long long cf = item / G.NB;
    const int fr = (int)(cf % G.T3); const long long clip = cf / G.T3;
    const int ph0 = band * G.pn; const int pnb = min(G.pn, G.PH - ph0);
    const int cr0 = max(0, 2 * ph0 - 1), cr1 = min(G.OH - 1, 2 * (ph0 + pnb - 1) + 1);
    const int ncr = cr1 - cr0 + 1, npx = ncr * G.OW;
    const int own0 = (2 * ph0 - cr0) * G.OW;                        // pixels before this one belong to the previous band's statistics
    __syncthreads();                                                // the previous band's slab and ring are no longer read
    if (!(abl & 16)) s3p_issue_slab(slab, vb, clip, fr, 2 * cr0 - 3, G);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int rb = 32 * wave; if (rb >= G.RING) rb -= G.RING;            // ring slot of the wave's first pixel of the current tile
    int next_pool = 0;
    for (int t0 = 0; t0 < npx; t0 += 128) {
      const int p = t0 + 32 * wave + pl; const int pc = p < npx ? p : 0;
      const int ohl = pc / G.OW, ow = pc - ohl * G.OW;
      f32x16 acc[2];
      if (abl & 2) { for (int r = 0; r < 16; ++r) { acc[0][r] = (float)(t0 + r); acc[1][r] = (float)(lane + r); } }
      else s3p_conv_tile(slab, ohl, ow, g, wf, rowoff, G, acc);
      __syncthreads();                                              // the previous iteration's pool reads are done: ring slots may be overwritten
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
            *(uint2*)(ring + ((32 * j + pl) * G.RING + slot) * 2) = make_uint2(f32x2_to_bf16x2(v0, v1), f32x2_to_bf16x2(v2, v3));
          }
        }
      }
      rb += 128; while (rb >= G.RING) rb -= G.RING;
      __syncthreads();
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
          for (int i = 0; i <= S3P_PWG; ++i) { const int k = pw0 - 1 + i; D[i] = rowp[(k >= 0 && 2 * k < G.OW) ? k : 0] ^ flip2; }
#pragma unroll
          for (int i = 0; i < S3P_PWG; ++i) {
            if (pw0 + i > 0) best[i] = max(best[i], s3p_key(D[i] & 0xffff0000u, 15u - (kh * 3)));          // column 2 pw - 1
            best[i] = max(best[i], s3p_key(D[i + 1] << 16, 15u - (kh * 3 + 1)));                             // column 2 pw
            best[i] = max(best[i], s3p_key(D[i + 1] & 0xffff0000u, 15u - (kh * 3 + 2)));                     // column 2 pw + 1
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
    }
  }
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
