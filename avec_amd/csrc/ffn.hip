// Fused macaron feed-forward module for gfx950 (bf16 MFMA, fp32 accumulate):  nnet/modules.py:257-289 + its residual nnet/blocks.py:292,301
//
//   forward :  y = x + alpha * Drop2( W2 * Drop1(Swish(W1 * LN(x) + b1)) + b2 )                    (saves mean, rstd, h0 = LN(x), z, h1)
//   backward:  dx = dy + LN'( (alpha * mask2 * dy) W2 * Swish'(z) * mask1 ) W1 )                   (saves dacc, dz, dh0 for the weight / LayerNorm gradients)
//
// One workgroup (4 waves) owns 64 rows of the residual stream and runs the whole chain on them; the hidden activations of a row tile never
// leave the chip between the two GEMMs.  Both GEMMs stream their weights through ONE ring of 16 KB LDS slots filled by the LDS-DMA
// (global_load_lds, 16 B per lane), in hidden-dimension chunks of 128:
//     a-slot [128 n][64 k]  : 128 hidden columns x 64 model-dim inputs     (GEMM-a, computed transposed: C^T = W * A0^T)
//     b-slot [ 64 n][128 k] : 64 model-dim outputs x the chunk's 128 hidden inputs   (GEMM-b)
// The 64 x D input tile (LN(x) / the prepared output gradient) lives in REGISTERS as MFMA operand fragments (<= 96 VGPRs), so does the 64 x 128
// chunk of hidden activations between the two GEMMs, and the 64 x D result (<= 96 accumulator VGPRs): every MFMA reads exactly one 1 KB weight
// fragment from LDS (128 B/clk/CU, half of the LDS rate).  GEMM-a is computed transposed so that a lane ends up with 4 consecutive hidden
// columns of one row = one 8-byte piece of the GEMM-b operand tile (no cross-lane packing).
// Synchronisation: ring of 4 slots, 3 in flight, counted `s_waitcnt vmcnt`, one raw s_barrier per slot (the DMA stays in flight across barriers).
// Inside the slot loop the waves issue no ordinary global loads (hipcc drains the DMA queue behind them); the per-chunk stores of z / h1 / dz
// are issued right after a barrier so that they retire before the next counted wait (the counts only include newer LOADS, which is safe whatever
// order stores and loads retire in).
#include "vec.h"
#include "avec_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 ffn_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ffn_f32x16;

__device__ __attribute__((aligned(64))) unsigned char ffn_zero16[64];
__device__ long long ffn_ts[8];        // s_memtime stamps of workgroup 0 / wave 0 when FfnArgs.dbg has bit 32 set (tools/bench_ffn.py)
#define FFN_STAMP(i) do { if ((g.dbg & 32) && blockIdx.x == 0 && tid == 0) ffn_ts[i] = __builtin_readcyclecounter(); } while (0)

struct FfnArgs {
  long long M; int D, F;
  const float* x; const float* ln_g; const float* ln_b; float eps;
  const bf16* Wa; long long lda;            // GEMM-a weight rows over F, K over D   (fwd: W1 [F][D];      bwd: W2^T [F][D])
  const bf16* Wb; long long ldb;            // GEMM-b weight rows over D, K over F   (fwd: W2 [D][F];      bwd: W1^T [D][F])
  const float* ba; const float* bb;         // fwd: b1 [F], b2 [D]
  float alpha, drop_p; const unsigned long long* rng; unsigned sid1, sid2;
  float* y;                                 // fwd: y [M][D];  bwd: dx [M][D]
  float* mean; float* rstd;                 // fwd: out;  bwd: in
  bf16* h0;                                 // fwd: LN(x) [M][D] out;  bwd: dh0 [M][D] out
  bf16* z;                                  // [M][F]  fwd: out;  bwd: in
  bf16* h1;                                 // [M][F]  fwd: h1 out;  bwd: dz out
  const float* dy;                          // bwd: [M][D]
  bf16* dacc;                               // bwd: [M][D] out
  int dbg;                                  // AVEC_FFN_DBG & 32: workgroup 0 records s_memtime stamps of its phases (tools/ffn_stamps.py)
};

static constexpr int FFN_BM = 64, FFN_SLOT = 16384, FFN_NSLOT = 4;
static constexpr int FFN_RING = FFN_NSLOT * FFN_SLOT;            // 64 KB
static constexpr int FFN_ZT = FFN_RING, FFN_HT = FFN_RING + 16384;  // z / h chunk tiles [64 m][128 k] bf16, 256-byte rows
static constexpr int FFN_BIAS = FFN_HT + 16384;                  // b1 (fwd), up to 2048 floats
static constexpr int FFN_LDS = FFN_BIAS + 8192;                  // 104 KB

// dropout: 4 keep-factors from one hash chain (16 random bits per element): element idx4*4 + e, e = 0..3
__device__ __forceinline__ void ffn_drop4(unsigned long long seed, unsigned sid, unsigned long long idx4, float p, float inv_keep, float (&s)[4]) {
  uint32_t a = mix32((uint32_t)idx4 ^ (uint32_t)seed);
  uint32_t b = mix32((uint32_t)(idx4 >> 32) + sid * 0x9e3779b9u + (uint32_t)(seed >> 32));
  uint32_t h0 = mix32(a ^ (b + 0x85ebca6bu + (a << 6) + (a >> 2)));
  uint32_t h1 = mix32(h0 + 0x632be5abu);
  const uint32_t thr = (uint32_t)(p * 65536.f);
  s[0] = (h0 & 0xffffu) >= thr ? inv_keep : 0.f; s[1] = (h0 >> 16) >= thr ? inv_keep : 0.f;
  s[2] = (h1 & 0xffffu) >= thr ? inv_keep : 0.f; s[3] = (h1 >> 16) >= thr ? inv_keep : 0.f;
}

__device__ __forceinline__ ffn_f32x16 ffn_mfma(const chunk16& a, const chunk16& b, const ffn_f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ffn_bf16x8, a), __builtin_bit_cast(ffn_bf16x8, b), c, 0, 0, 0);
}

// LDS reads the compiler must not see as LDS reads: hipcc orders every ds_read it cannot disambiguate from an in-flight LDS-DMA behind `s_waitcnt vmcnt(0)`
// (measured here: the bias and chunk-tile reads drained the 3-slot prefetch twice per chunk, 10x slower).  The caller waits with FFN_WAIT_LGKM0().
typedef __attribute__((ext_vector_type(4))) uint32_t ffn_u32x4;
__device__ __forceinline__ void ffn_lds_read16(uint32_t lds_addr, chunk16& out) {
  ffn_u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  out = __builtin_bit_cast(chunk16, v);
}
__device__ __forceinline__ void ffn_lds_read8(uint32_t lds_addr, uint2& out) {
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  out = __builtin_bit_cast(uint2, v);
}

__device__ __forceinline__ void ffn_lds_write8(uint32_t lds_addr, const uint2& val) {      // (same reason: a ds_write the compiler sees is ordered behind the DMA queue too)
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  const u32x2 v = {val.x, val.y};
  asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}

#define FFN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define FFN_WAIT_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// NKT = ceil(D / 64): k-tiles of GEMM-a = 64-column tiles of GEMM-b
template <int NKT, bool BWD>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, gh = lane >> 5;
  const int wa = wave >> 1, wb = wave & 1;            // GEMM-a: hidden-column half, row half
  const int wm = wave & 1, wn = wave >> 1;            // GEMM-b: row half, output-column half of a 64-column tile
  const long long m0 = (long long)blockIdx.x * FFN_BM;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const int D = g.D, F = g.F;
  const int NCH = (F + 127) >> 7;
  const unsigned long long seed = g.drop_p > 0.f ? g.rng[0] + 0x9e3779b97f4a7c15ull * g.rng[1] : 0ull;
  const float inv_keep = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  constexpr int DP = NKT * 64 + 4;                     // fp32 row pitch of the 64 x D result tile staged in LDS by the final epilogue

  FFN_STAMP(0);
  // ---------------- prologue: the 64 x D operand tile A0 (bf16) -> LDS in k-tile format -> this wave's MFMA fragments ----------------
  // fwd: A0 = LN(x) (also stored as h0, with mean / rstd);  bwd: A0 = alpha * mask2 * dy (also stored as dacc).  A wave owns 16 rows, 8 at a time in flight.
  {
    const float* src = BWD ? g.dy : g.x;
    float gg[2][4], bb[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane * 4 + i * 256;
#pragma unroll
      for (int e = 0; e < 4; ++e) { gg[i][e] = 0.f; bb[i][e] = 0.f; }
      if (!BWD && c < D) { ld4<float>(g.ln_g + c, gg[i]); ld4<float>(g.ln_b + c, bb[i]); }
    }
#pragma unroll 1
    for (int rb = 0; rb < 2; ++rb) {
      float v[8][2][4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long long row = m0 + wave * 16 + rb * 8 + j;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = lane * 4 + i * 256;
          if (row < g.M && c < D) ld4<float>(src + row * D + c, v[j][i]); else { v[j][i][0] = v[j][i][1] = v[j][i][2] = v[j][i][3] = 0.f; }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = wave * 16 + rb * 8 + j;
        const long long row = m0 + r;
        const bool rok = row < g.M;
        if (!BWD) {
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < 2; ++i) s += v[j][i][0] + v[j][i][1] + v[j][i][2] + v[j][i][3];
          const float mu = wave_sum(s) / D;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < 2; ++i) { const int c = lane * 4 + i * 256; if (c < D) for (int e = 0; e < 4; ++e) { const float d = v[j][i][e] - mu; q += d * d; } }
          const float rs = rsqrtf(wave_sum(q) / D + g.eps);
          if (lane == 0 && rok) { g.mean[row] = mu; g.rstd[row] = rs; }
#pragma unroll
          for (int i = 0; i < 2; ++i) { const int c = lane * 4 + i * 256; if (c < D) for (int e = 0; e < 4; ++e) v[j][i][e] = (v[j][i][e] - mu) * rs * gg[i][e] + bb[i][e]; }
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < D) {
              float ds[4] = {1.f, 1.f, 1.f, 1.f};
              if (g.drop_p > 0.f) ffn_drop4(seed, g.sid2, ((unsigned long long)row * D + c) >> 2, g.drop_p, inv_keep, ds);
              for (int e = 0; e < 4; ++e) v[j][i][e] *= g.alpha * ds[e];
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = lane * 4 + i * 256;
          if (c < NKT * 64) {
            uint2 pk; pk.x = f32x2_to_bf16x2(v[j][i][0], v[j][i][1]); pk.y = f32x2_to_bf16x2(v[j][i][2], v[j][i][3]);
            *(uint2*)(smem + (c >> 6) * 8192 + r * 128 + (((((c & 63) >> 3)) ^ ((r >> 1) & 7)) << 4) + (c & 7) * 2) = pk;
            if (rok && c < D) *(uint2*)((BWD ? g.dacc : g.h0) + row * D + c) = pk;
          }
        }
      }
    }
    if (!BWD) for (int c = tid; c < NCH * 128; c += 256) ((float*)(smem + FFN_BIAS))[c] = c < F ? g.ba[c] : 0.f;
  }
  __syncthreads();
  chunk16 a0f[NKT][4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wb * 32 + l31;
      a0f[kt][q] = *(const chunk16*)(smem + kt * 8192 + r * 128 + ((((2 * q + gh)) ^ ((r >> 1) & 7)) << 4));
    }
  __syncthreads();                                    // A0's LDS image is dead: the ring may be filled
  FFN_STAMP(1);

  // ---------------- slot stream ----------------
  constexpr int PC = 2 * NKT + (BWD ? 1 : 0);          // slots per chunk: NKT a-slots, (bwd: the z tile), NKT b-slots
  const int S = NCH * PC;
  int is_ch = 0, is_pos = 0, is_n = 0;                 // issue cursor (chunk, position inside the chunk, flat slot number)
  const int a_row = tid >> 3, a_lc = (tid & 7) ^ ((tid >> 4) & 7);       // a-slot: pass i covers rows i*32 + a_row; logical 16-byte k-chunk a_lc
  const int b_row = tid >> 4, b_lc = (tid & 15) ^ ((tid >> 4) & 15);     // b-slot / z tile: pass i covers rows i*16 + b_row; logical k-chunk b_lc
  // per-thread byte offsets of the four DMA pieces of a slot (32-bit: the weight matrices are < 4 GB), the slot's own position is wave-uniform
  uint32_t aoff[4], boff[4], zoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = (uint32_t)(((long long)(i * 32 + a_row) * g.lda + a_lc * 8) * 2);
    boff[i] = (uint32_t)(((long long)(i * 16 + b_row) * g.ldb + b_lc * 8) * 2);
    zoff[i] = (uint32_t)(((i * 16 + b_row) * F + b_lc * 8) * 2);
  }
  const char* zbase = BWD ? (const char*)(g.z + m0 * F) : nullptr;
  auto issue = [&]() {
    if (is_n >= S) return;
    char* dst = smem + (is_n & (FFN_NSLOT - 1)) * FFN_SLOT + wave * 1024;
    const int c0 = is_ch * 128;
    if (is_pos < NKT) {                                // a-slot: Wa rows c0 .. c0+127, K = is_pos*64 ..
      const char* base = (const char*)g.Wa + ((long long)c0 * g.lda + is_pos * 64) * 2;
      const bool tail = (c0 + 128 > F) || (is_pos * 64 + 64 > D);
      if (!tail) {
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(base + aoff[i]), (lptr_t)(dst + i * 4096), 16, 0, 0);
      } else {
        const bool kok = is_pos * 64 + a_lc * 8 < D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const void* src = (kok && c0 + i * 32 + a_row < F) ? (const void*)(base + aoff[i]) : (const void*)ffn_zero16;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + i * 4096), 16, 0, 0);
        }
      }
    } else if (BWD && is_pos == NKT) {                 // z tile of this chunk: rows m0 .., columns c0 ..
      const char* base = zbase + c0 * 2;
      const bool kok = c0 + b_lc * 8 < F;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const void* src = (kok && m0 + i * 16 + b_row < g.M) ? (const void*)(base + zoff[i]) : (const void*)ffn_zero16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + i * 4096), 16, 0, 0);
      }
    } else {                                           // b-slot: Wb rows nt*64 .., K = c0 ..
      const int nt = is_pos - NKT - (BWD ? 1 : 0);
      const char* base = (const char*)g.Wb + ((long long)nt * 64 * g.ldb + c0) * 2;
      const bool tail = (nt * 64 + 64 > D) || (c0 + 128 > F);
      if (!tail) {
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(base + boff[i]), (lptr_t)(dst + i * 4096), 16, 0, 0);
      } else {
        const bool kok = c0 + b_lc * 8 < F;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const void* src = (kok && nt * 64 + i * 16 + b_row < D) ? (const void*)(base + boff[i]) : (const void*)ffn_zero16;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + i * 4096), 16, 0, 0);
        }
      }
    }
    ++is_n; if (++is_pos == PC) { is_pos = 0; ++is_ch; }
  };
  int cs = 0;                                          // slot being consumed
  auto step = [&]() -> const char* {                   // wait for slot cs, barrier, refill the slot freed by the previous step
    const int newer = min(FFN_NSLOT - 2, S - 1 - cs);
    if (newer <= 0) FFN_WAIT_VM(0); else if (newer == 1) FFN_WAIT_VM(4); else FFN_WAIT_VM(8);
    FFN_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
    const char* T = smem + (cs & (FFN_NSLOT - 1)) * FFN_SLOT;
    ++cs;
    return T;
  };
#pragma unroll
  for (int st = 0; st < FFN_NSLOT - 1; ++st) issue();

  ffn_f32x16 acc2[NKT];
#pragma unroll
  for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.f;
  const int swzA = (l31 >> 1) & 7, swzB = l31 & 15;
  const int mrow = wb * 32 + l31;                      // this lane's row in the transposed GEMM-a result / the z, h tiles

#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 128;
    ffn_f32x16 acc1[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[s][r] = 0.f;
    // ---- GEMM-a (transposed): acc1[s][n][m] += Wa[c0 + wa*64 + s*32 + n][k] * A0[m][k]
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const char* T = step();
      chunk16 wf[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) wf[q][s] = *(const chunk16*)(T + (wa * 64 + s * 32 + l31) * 128 + (((2 * q + gh) ^ swzA) << 4));
      asm volatile("" ::: "memory");                  // all eight fragment reads are requested before the first MFMA (hipcc otherwise pairs each read with its consumer: one LDS latency per MFMA)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc1[s] = ffn_mfma(wf[q][s], a0f[kt][q], acc1[s]);
    }
    // ---- epilogue-a: hidden activations of this chunk -> h tile (LDS) [+ z tile]; lane = row mrow, 4 consecutive columns per (s, g4)
    const char* Zt = BWD ? step() : nullptr;            // bwd: the z tile arrives through the ring
    const long long row = m0 + mrow;
    {
      chunk16 b4[2][4]; uint2 zin[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n = wa * 64 + s * 32 + 8 * g4 + 4 * gh;
          if (!BWD) ffn_lds_read16(lds0 + FFN_BIAS + (c0 + n) * 4, b4[s][g4]);
          else ffn_lds_read8(lds0 + (uint32_t)(Zt - smem) + mrow * 256 + ((((n >> 3)) ^ (mrow & 15)) << 4) + (n & 7) * 2, zin[s][g4]);
        }
      FFN_WAIT_LGKM0();
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n = wa * 64 + s * 32 + 8 * g4 + 4 * gh;       // column inside the chunk (multiple of 4)
          const int f = c0 + n;
          const int toff = mrow * 256 + ((((n >> 3)) ^ (mrow & 15)) << 4) + (n & 7) * 2;
          float v[4], ds[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc1[s][4 * g4 + e];
          if (g.drop_p > 0.f) ffn_drop4(seed, g.sid1, ((unsigned long long)row * F + f) >> 2, g.drop_p, inv_keep, ds);
          if (!BWD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(b4[s][g4].w[e]);
            uint2 zp; zp.x = f32x2_to_bf16x2(v[0], v[1]); zp.y = f32x2_to_bf16x2(v[2], v[3]);
            ffn_lds_write8(lds0 + FFN_ZT + toff, zp);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = swishf_(v[e]) * ds[e];
          } else {
            const uint2 zp = zin[s][g4];
            const float zz[4] = {__uint_as_float(zp.x << 16), __uint_as_float(zp.x & 0xffff0000u), __uint_as_float(zp.y << 16), __uint_as_float(zp.y & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= dswishf_(zz[e]) * ds[e];
          }
          uint2 hp; hp.x = f32x2_to_bf16x2(v[0], v[1]); hp.y = f32x2_to_bf16x2(v[2], v[3]);
          ffn_lds_write8(lds0 + FFN_HT + toff, hp);
        }
    }
    // ---- GEMM-b: acc2[nt][m][n] += H[m][k] * Wb[nt*64 + wn*32 + n][c0 + k]
    chunk16 a1f[8];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
      const char* T = step();
      chunk16 wf[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) wf[q] = *(const chunk16*)(T + (wn * 32 + l31) * 256 + (((2 * q + gh) ^ swzB) << 4));
      if (nt == 0) {
        const int r = wm * 32 + l31;
        chunk16 ch_[4], cz[4];
#pragma unroll
        for (int q = 0; q < 8; ++q) ffn_lds_read16(lds0 + FFN_HT + r * 256 + (((2 * q + gh) ^ (r & 15)) << 4), a1f[q]);
        // copy the chunk tiles out (coalesced 16-byte pieces): fwd z and h1, bwd dz
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int o = (i * 16 + b_row) * 256 + ((tid & 15) << 4);
          ffn_lds_read16(lds0 + FFN_HT + o, ch_[i]);
          if (!BWD) ffn_lds_read16(lds0 + FFN_ZT + o, cz[i]);
        }
        FFN_WAIT_LGKM0();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = i * 16 + b_row; const long long grow = m0 + rr; const int k = c0 + b_lc * 8;
          if (grow < g.M && k < F) {
            *(chunk16*)(g.h1 + grow * F + k) = ch_[i];
            if (!BWD) *(chunk16*)(g.z + grow * F + k) = cz[i];
          }
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) acc2[nt] = ffn_mfma(a1f[q], wf[q], acc2[nt]);
    }
  }
  FFN_WAIT_VM(0);
  __syncthreads();
  FFN_STAMP(2);

  // ---------------- final epilogue: the 64 x D fp32 result tile -> LDS [64][DP], then row by row (a wave owns 16 rows, 4 consecutive columns per lane) ----------------
  float* Ls = (float*)smem;
#pragma unroll
  for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ls[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * gh) * DP + nt * 64 + wn * 32 + l31] = acc2[nt][r];
  __syncthreads();
  if (!BWD) {
    // y = x + alpha * drop2(acc + b2)
    float b2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int c = lane * 4 + i * 256; if (c < D) ld4<float>(g.bb + c, b2[i]); }
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int r = wave * 16 + j;
      const long long row = m0 + r;
      if (row >= g.M) break;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < D) {
          const float4 t = *(const float4*)(Ls + r * DP + c);
          float xv[4], ds[4] = {1.f, 1.f, 1.f, 1.f};
          ld4<float>(g.x + row * D + c, xv);
          if (g.drop_p > 0.f) ffn_drop4(seed, g.sid2, ((unsigned long long)row * D + c) >> 2, g.drop_p, inv_keep, ds);
          const float a4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[e] += g.alpha * ds[e] * (a4[e] + b2[i][e]);
          st4<float>(g.y + row * D + c, xv);
        }
      }
    }
  } else {
    // LayerNorm backward: dx = dy + rstd * (dh0*gamma - mean(dh0*gamma) - xhat * mean(dh0*gamma*xhat)); dh0 is stored (bf16) for the deferred d(gamma), d(beta)
    float gg[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int c = lane * 4 + i * 256; if (c < D) ld4<float>(g.ln_g + c, gg[i]); }
#pragma unroll 1
    for (int rb = 0; rb < 4; ++rb) {
      float xh[4][2][4], o[4][2][4], mu[4], rs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long row = m0 + wave * 16 + rb * 4 + j;
        const bool rok = row < g.M;
        mu[j] = rok ? g.mean[row] : 0.f; rs[j] = rok ? g.rstd[row] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = lane * 4 + i * 256;
          if (rok && c < D) { ld4<float>(g.x + row * D + c, xh[j][i]); ld4<float>(g.dy + row * D + c, o[j][i]); }
          else { for (int e = 0; e < 4; ++e) { xh[j][i][e] = 0.f; o[j][i][e] = 0.f; } }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wave * 16 + rb * 4 + j;
        const long long row = m0 + r;
        if (row >= g.M) continue;
        float dh[2][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = lane * 4 + i * 256;
          if (c < D) {
            const float4 t = *(const float4*)(Ls + r * DP + c); dh[i][0] = t.x; dh[i][1] = t.y; dh[i][2] = t.z; dh[i][3] = t.w;
            uint2 pk; pk.x = f32x2_to_bf16x2(dh[i][0], dh[i][1]); pk.y = f32x2_to_bf16x2(dh[i][2], dh[i][3]);
            *(uint2*)(g.h0 + row * D + c) = pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) { xh[j][i][e] = (xh[j][i][e] - mu[j]) * rs[j]; const float t2 = dh[i][e] * gg[i][e]; s1 += t2; s2 += t2 * xh[j][i][e]; }
          }
        }
        s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = lane * 4 + i * 256;
          if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[j][i][e] += rs[j] * (dh[i][e] * gg[i][e] - s1 - xh[j][i][e] * s2);
            st4<float>(g.y + row * D + c, o[j][i]);
          }
        }
      }
    }
  }
  FFN_STAMP(3);
}

static int ffn_check(long long M, int D, int F) {
  AVEC_CHECK_ARG(M > 0 && D >= 64 && D <= 384 && D % 8 == 0 && F >= 128 && F <= 2048 && F % 8 == 0, "ffn_fused: unsupported shape M=%lld D=%d F=%d (need D %% 8 == 0 in [64, 384], F %% 8 == 0 in [128, 2048])", M, D, F);
  return 0;
}
extern "C" int avec_ffn_debug_stamps(long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(ffn_ts), sizeof(long long) * 8); }
extern "C" int avec_ffn_fused_supported(int dtype, long long M, int D, int F) {
  return dtype == AVEC_BF16 && M > 0 && D >= 64 && D <= 384 && D % 8 == 0 && F >= 128 && F <= 2048 && F % 8 == 0;
}

template <bool BWD>
static int ffn_launch(FfnArgs g, hipStream_t st) {
  static const int dbg_env = getenv("AVEC_FFN_DBG") ? atoi(getenv("AVEC_FFN_DBG")) : 0;
  g.dbg = dbg_env;
  const int nkt = (g.D + 63) / 64;
  const dim3 grid((unsigned)((g.M + FFN_BM - 1) / FFN_BM));
#define FFN_L(N_) do { static bool once = false; if (!once) { hipError_t e = hipFuncSetAttribute((const void*)ffn_fused_kernel<N_, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS); \
      if (e != hipSuccess) { avec_set_error("ffn_fused: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; } once = true; } \
    hipLaunchKernelGGL((ffn_fused_kernel<N_, BWD>), grid, dim3(256), FFN_LDS, st, g); } while (0)
  switch (nkt) { case 1: FFN_L(1); break; case 2: FFN_L(2); break; case 3: FFN_L(3); break; case 4: FFN_L(4); break; case 5: FFN_L(5); break; default: FFN_L(6); break; }
#undef FFN_L
  AVEC_LAUNCH_CHECK();
  return 0;
}

extern "C" int avec_ffn_fused_fwd(const float* x, const float* ln_g, const float* ln_b, float eps, const void* w1, long long ldw1, const float* b1,
                                  const void* w2, long long ldw2, const float* b2, float alpha, float drop_p, const unsigned long long* rng,
                                  unsigned sid1, unsigned sid2, float* y, float* mean, float* rstd, void* h0, void* z, void* h1,
                                  long long M, int D, int F, hipStream_t stream) {
  AVEC_CHECK_ARG(x && ln_g && ln_b && w1 && b1 && w2 && b2 && y && mean && rstd && h0 && z && h1, "ffn_fused_fwd: null pointer");
  if (int r = ffn_check(M, D, F)) return r;
  AVEC_CHECK_ARG(ldw1 % 8 == 0 && ldw2 % 8 == 0 && (((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)z | (uintptr_t)h1 | (uintptr_t)h0) & 15) == 0, "ffn_fused_fwd: weights / activations must be 16-byte aligned with row strides %% 8 == 0");
  AVEC_CHECK_ARG(!(drop_p > 0.f) || rng, "ffn_fused_fwd: dropout without rng state");
  FfnArgs g = {}; g.M = M; g.D = D; g.F = F; g.x = x; g.ln_g = ln_g; g.ln_b = ln_b; g.eps = eps; g.Wa = (const bf16*)w1; g.lda = ldw1; g.Wb = (const bf16*)w2; g.ldb = ldw2;
  g.ba = b1; g.bb = b2; g.alpha = alpha; g.drop_p = drop_p; g.rng = rng; g.sid1 = sid1; g.sid2 = sid2; g.y = y; g.mean = mean; g.rstd = rstd;
  g.h0 = (bf16*)h0; g.z = (bf16*)z; g.h1 = (bf16*)h1;
  return ffn_launch<false>(g, stream);
}

extern "C" int avec_ffn_fused_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* ln_g, const void* w2t, long long ldw2t,
                                  const void* w1t, long long ldw1t, const void* z, float alpha, float drop_p, const unsigned long long* rng,
                                  unsigned sid1, unsigned sid2, float* dx, void* dacc, void* dz, void* dh0, long long M, int D, int F, hipStream_t stream) {
  AVEC_CHECK_ARG(dy && x && mean && rstd && ln_g && w2t && w1t && z && dx && dacc && dz && dh0, "ffn_fused_bwd: null pointer");
  if (int r = ffn_check(M, D, F)) return r;
  AVEC_CHECK_ARG(ldw2t % 8 == 0 && ldw1t % 8 == 0 && (((uintptr_t)w2t | (uintptr_t)w1t | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)dacc | (uintptr_t)dh0) & 15) == 0, "ffn_fused_bwd: weights / activations must be 16-byte aligned with row strides %% 8 == 0");
  AVEC_CHECK_ARG(!(drop_p > 0.f) || rng, "ffn_fused_bwd: dropout without rng state");
  FfnArgs g = {}; g.M = M; g.D = D; g.F = F; g.dy = dy; g.x = x; g.mean = (float*)mean; g.rstd = (float*)rstd; g.ln_g = ln_g;
  g.Wa = (const bf16*)w2t; g.lda = ldw2t; g.Wb = (const bf16*)w1t; g.ldb = ldw1t; g.z = (bf16*)z; g.alpha = alpha; g.drop_p = drop_p; g.rng = rng; g.sid1 = sid1; g.sid2 = sid2;
  g.y = dx; g.dacc = (bf16*)dacc; g.h1 = (bf16*)dz; g.h0 = (bf16*)dh0;
  return ffn_launch<true>(g, stream);
}
