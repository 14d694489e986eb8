// Row-resident CHAINS of the conformer modules for gfx950 (CDNA4, wave64), bf16 MFMA, fp32 accumulate.
//
// The conformer stacks are ~400 products of M = 800 .. 6400 rows per step: launched one by one they are bound by launch + ring fill + epilogue (3 % of the MFMA
// roofline), and every LayerNorm / activation between them is a launch of its own.  Here a workgroup keeps a 64-row tile resident across a chain
//
//   prologue (LayerNorm of the fp32 rows | gradient preparation)  ->  product A  ->  element-wise middle  ->  product B  ->  fp32 atomic add into the output
//
// and the hidden width is SPLIT over workgroups (split-F): workgroup (row tile t, slice s) computes only columns [256 s, 256 s + 256) of the hidden activation and
// its partial contribution to the output, so M/64 x F/256 workgroups each stream 2 x 128 KB of weights (one workgroup per row tile would stream the whole 1 MB and
// lose to the three-launch sequence: DESIGN.md section 11.6).  The partial products are added with fp32 atomics into a PRE-ZEROED output; slice 0 also adds the
// residual and the output bias.
//
//   MODE 0  macaron feed-forward module, forward (nnet/modules.py:257-289 + the residual of nnet/blocks.py:292,301):
//             out += [x + alpha * Drop2(b2)]_{slice 0} + alpha * Drop2( Drop1(Swish(LN(x) W1_s^T + b1_s)) W2_s^T )
//           saved: mean, rstd, h0 = LN(x) (slice 0); z = pre-activation in ACCUMULATOR order (each lane stores what it holds: 256 B per store instruction)
//   MODE 1  the same module, backward:   dacc = alpha * mask2 * dy;   dz_s = (dacc W2_s) * mask1 * Swish'(z_s);   out(dh0) += dz_s W1_s
//           written for the weight-gradient products: dacc (slice 0), dz and h1 = Drop1(Swish(z)) (recomputed), row-major bf16
//   MODE 2  LayerNorm + ONE product (Q|K|V projection, first pointwise convolution):  o1 = LN(x) Wa_s^T + ba_s   (bf16, row-major)
//
// LDS: the A operand of a phase is resident ([K/64] sub-tiles of 64 rows x 128 B, same swizzle as gemm_nt_plain_kernel); the weight tiles of both products ride one
// ring fed by LDS-DMA (global_load_lds_dwordx4, scalar base + per-lane offsets fixed at entry, counted vmcnt).  Product A: 4 waves x (64 rows x 64 columns);
// product B: 4 waves x (64 rows x 32 NTB columns), NTB = 2 (D <= 256) or 3 (D <= 384).
#include "vec.h"
#include "avec_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

template <int OFF> __device__ __forceinline__ u32x4 ch_lds_read(unsigned lds_addr) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory"); return v; }
// four LDS-DMA instructions (16 B per lane each) as one group: scalar base + per-lane 32-bit byte offsets; destinations lds0 + 4096 i (wave-uniform, through M0)
__device__ __forceinline__ void ch_glds4(unsigned v0, unsigned v1, unsigned v2, unsigned v3, const void* sbase, unsigned lds0) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
               "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2\n\t"
               "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %2\n\t"
               "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %2\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "s"(sbase), "s"(lds0), "v"(v1), "v"(v2), "v"(v3) : "memory", "scc");
}
template <int V> struct IC { static constexpr int value = V; };
__device__ __forceinline__ int ch_swz(int row) { return (row >> 1) & 7; }      // = glds_swz<128> of gemm.hip

struct ChainArgs {
  const float* xin;                      // MODE 0 / 2: x [M][D] fp32;  MODE 1: dy [M][D] fp32
  const float* ln_w; const float* ln_b; float eps;
  float* mean; float* rstd;              // MODE 0 / 2 (slice 0): LayerNorm statistics
  bf16* h0;                              // MODE 0 / 2 (slice 0): LN(x);  MODE 1 (slice 0): dacc          [M][D] row-major
  const bf16* Wa; long long ldwa; const float* ba; int Na;      // product A: weight rows = hidden / output index [Na][D]
  const bf16* Wb; long long ldwb; const float* bb;              // product B: weight rows = output index [D][Na], the slice takes columns 256 s ...
  unsigned* zbuf;                        // z in accumulator order (MODE 0 writes, MODE 1 reads)
  bf16* o1; long long ldo1;              // MODE 1: dz [M][Na];  MODE 2: out [M][Na]
  bf16* o2; long long ldo2;              // MODE 1: h1 [M][Na]
  float* out; long long ldo;             // MODE 0 / 1: [M][D] fp32, pre-zeroed, atomically accumulated
  long long M; int D, KTA;               // KTA = ceil(D / 64)
  float alpha, p; const unsigned long long* rng; unsigned sid1, sid2;
};

constexpr int SUB = 8192;                // one resident A sub-tile: 64 rows x 128 B
constexpr int TILE_A = 256 * 128;        // weight tile of product A: 256 rows x 128 B
constexpr int RING = 3 * TILE_A;         // 96 KB: 3 stages of product A; product B: 3 x 32 KB (NTB = 2) or 2 x 48 KB (NTB = 3)

// one product phase: acc[2][NT] += Ares[64][64 KT] . W[rows of this wave][64 KT]^T, A resident at lds_a (sub-tile kt), W tiles through the ring
template <int NT, int STAGES, int TILE>
struct Phase {
  static constexpr int NPASS = TILE / 4096;                      // DMA instructions per thread and tile
  unsigned off[NPASS];                                           // per-lane byte offsets of the tile rows this thread fetches (chunk swizzle folded in)
  const char* wbase;                                             // scalar: W + column offset of this slice / K origin
  unsigned lds_ring, lds_a, wslot;
  unsigned aad[4], bad[4];                                       // fragment addresses of K-substep q (stage / sub-tile offsets are added as scalars)

  __device__ __forceinline__ void plan(const void* W, long long ldw, int row0, int nrows_valid, long long col0_bytes, int tid, unsigned lds0_ring, unsigned lds0_a, int wave, int lane, int brow0) {
    wbase = (const char*)W + col0_bytes;
    lds_ring = lds0_ring; lds_a = lds0_a; wslot = (unsigned)wave * 1024u;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int r = (tid >> 3) + 32 * i; int n = row0 + r; n = n < nrows_valid ? n : nrows_valid - 1;
      const int kc = (tid & 7) ^ ch_swz(r);
      off[i] = (unsigned)(((long long)n * ldw + kc * 8) * 2);
    }
    const int g = lane >> 5, ra = lane & 31, rb = brow0 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      aad[q] = lds0_a + (unsigned)(ra * 128 + (((2 * q + g) ^ ch_swz(ra)) << 4));
      bad[q] = lds0_ring + (unsigned)(rb * 128 + (((2 * q + g) ^ ch_swz(rb)) << 4));
    }
  }
  template <int S> __device__ __forceinline__ void issue(int kt) const {
    const char* src = wbase + (long long)kt * 128;
    const unsigned dst = lds_ring + S * TILE + wslot;
#pragma unroll
    for (int i = 0; i < NPASS; i += 4) ch_glds4(off[i], off[i + 1], off[i + 2], off[i + 3], src, dst + i * 4096);
  }
  template <int S> __device__ __forceinline__ void step(int kt, int KT, f32x16 (&acc)[2][NT]) const {
    const int rem = KT - 1 - kt;                                   // tiles issued after kt that may still be in flight: min(rem, STAGES - 2)
    if (STAGES >= 3 && rem >= 1) {
      if (STAGES >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPASS) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPASS) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned asub = (unsigned)kt * SUB;
    // (the immediate offset field of ds_read holds 16 bits: a stage offset beyond that goes into the address register)
    constexpr bool FAR = S * TILE + (NT - 1) * 4096 > 65535;
    constexpr int IMM = FAR ? 0 : S * TILE;
    u32x4 fa[4][2], fb[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fa[q][0] = ch_lds_read<0>(aad[q] + asub);
      fa[q][1] = ch_lds_read<4096>(aad[q] + asub);
      const unsigned bq = bad[q] + (FAR ? (unsigned)(S * TILE) : 0u);
      fb[q][0] = ch_lds_read<IMM>(bq);
      if constexpr (NT > 1) fb[q][1] = ch_lds_read<IMM + 4096>(bq);
      if constexpr (NT > 2) fb[q][2] = ch_lds_read<IMM + 8192>(bq);
    }
    if (kt + STAGES - 1 < KT) issue<(S + STAGES - 1) % STAGES>(kt + STAGES - 1);      // into the slot everybody finished reading before this barrier
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * (2 + NT)) : "memory");
      else if (q == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (2 + NT)) : "memory");
      else if (q == 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 + NT) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(fa[q][0])); asm volatile("" : "+v"(fa[q][1]));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[q][j]));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[q][i]), __builtin_bit_cast(bf16x8_t, fb[q][j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __device__ __forceinline__ void prologue(int KT) const {
    issue<0>(0);
    if (STAGES >= 3 && KT > 1) issue<1 % STAGES>(1);
    if (STAGES >= 4 && KT > 2) issue<2 % STAGES>(2);
  }
  __device__ __forceinline__ void run(int KT, f32x16 (&acc)[2][NT]) const {
#pragma unroll 1
    for (int kt = 0; kt < KT; kt += STAGES) {
      step<0>(kt, KT, acc);
      if (STAGES > 1 && kt + 1 < KT) step<1 % STAGES>(kt + 1, KT, acc);
      if (STAGES > 2 && kt + 2 < KT) step<2 % STAGES>(kt + 2, KT, acc);
      if (STAGES > 3 && kt + 3 < KT) step<3 % STAGES>(kt + 3, KT, acc);
    }
  }
};

// resident-tile address of element (row, k) (k = column of the 64-row operand): sub-tile k / 64, chunk (k % 64) / 8 swizzled by the row
__device__ __forceinline__ unsigned res_addr(int row, int k) { return (unsigned)((k >> 6) * SUB + row * 128 + ((((k & 63) >> 3) ^ ch_swz(row)) << 4) + (k & 7) * 2); }

template <int NTB, int MODE>
__global__ __launch_bounds__(256, 1) void chain_kernel(ChainArgs g) {
  constexpr int STB = NTB == 2 ? 3 : 2, TILE_B = NTB * 128 * 128;
  constexpr int AREG = (NTB == 2 ? 4 : 6) * SUB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long m0 = (long long)blockIdx.x * 64; const int slice = blockIdx.y; const int f0 = slice * 256;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  char* const Areg = smem; char* const Ring = smem + AREG;
  const unsigned lds_a = lds0, lds_ring = lds0 + AREG;
  const int D = g.D, KTA = g.KTA;

  Phase<2, 3, TILE_A> pa;
  pa.plan(g.Wa, g.ldwa, f0, g.Na, 0, tid, lds_ring, lds_a, wave, lane, wave * 64);
  pa.prologue(KTA);                                   // the first weight tiles travel while the prologue computes the A operand

  // ---- prologue: the resident A operand (64 rows x KTA*64, bf16) ----
  {
#pragma unroll 1
    for (int rr = 0; rr < 16; rr += 2) {
      // two rows per trip (their loads are issued together); a wave owns rows 16 w .. 16 w + 15
      float v[2][2][4]; long long mrow[2]; int rloc[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        rloc[u] = wave * 16 + rr + u; const long long m = m0 + rloc[u]; mrow[u] = m < g.M ? m : g.M - 1;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          const int c = lane * 4 + gq * 256;
          if (c < D) ld4<float>(g.xin + mrow[u] * D + c, v[u][gq]); else { v[u][gq][0] = v[u][gq][1] = v[u][gq][2] = v[u][gq][3] = 0.f; }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float o[2][4];
        if (MODE == 1) {
#pragma unroll
          for (int gq = 0; gq < 2; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = lane * 4 + gq * 256 + e;
              o[gq][e] = v[u][gq][e] * g.alpha * drop_scale(g.rng, g.sid2, (unsigned long long)mrow[u] * D + c, g.p);
            }
        } else {
          float s = 0.f;
#pragma unroll
          for (int gq = 0; gq < 2; ++gq) s += v[u][gq][0] + v[u][gq][1] + v[u][gq][2] + v[u][gq][3];
          const float mu = wave_sum(s) / D;
          float q = 0.f;
#pragma unroll
          for (int gq = 0; gq < 2; ++gq) { const int c = lane * 4 + gq * 256; if (c < D) for (int e = 0; e < 4; ++e) { const float d = v[u][gq][e] - mu; q += d * d; } }
          const float rs = rsqrtf(wave_sum(q) / D + g.eps);
          if (slice == 0 && lane == 0 && m0 + rloc[u] < g.M) { g.mean[mrow[u]] = mu; g.rstd[mrow[u]] = rs; }
#pragma unroll
          for (int gq = 0; gq < 2; ++gq) {
            const int c = lane * 4 + gq * 256;
            float gg[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
            if (c < D) { ld4<float>(g.ln_w + c, gg); ld4<float>(g.ln_b + c, bb); }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[gq][e] = (v[u][gq][e] - mu) * rs * gg[e] + bb[e];
          }
        }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          const int c = lane * 4 + gq * 256;
          if (c >= KTA * 64) continue;
          uint2 t;
          if (c < D) { t.x = f32x2_to_bf16x2(o[gq][0], o[gq][1]); t.y = f32x2_to_bf16x2(o[gq][2], o[gq][3]); } else { t.x = 0u; t.y = 0u; }
          *(uint2*)(Areg + res_addr(rloc[u], c)) = t;
          if (slice == 0 && g.h0 && c < D && m0 + rloc[u] < g.M) *(uint2*)(g.h0 + mrow[u] * D + c) = t;
        }
      }
    }
  }
  // (the first barrier of phase A publishes the A operand)

  // ---- product A: hidden[64][256 of this slice] ----
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  pa.run(KTA, acc);
  __syncthreads();                                    // every wave is done with the A operand and the ring

  Phase<NTB, STB, TILE_B> pb;
  if (MODE != 2) {
    pb.plan(g.Wb, g.ldwb, 0, D, (long long)f0 * 2, tid, lds_ring, lds_a, wave, lane, wave * NTB * 32);
    pb.prologue(4);                                   // the first tile(s) of product B travel under the middle part
  }

  // ---- middle: element-wise work on the hidden tile; the result becomes the resident A operand of product B ----
  const long long zblk = ((long long)blockIdx.x * gridDim.y + blockIdx.y) * 8192;
  char* const Stage = Ring + (STB - 1) * TILE_B;     // MODE 1: row-major staging of h1 (the last ring slot: not a target of pb.prologue)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cl = wave * 64 + j * 32 + (lane & 31); const int f = f0 + cl; const bool fok = f < g.Na;
      const float bias = (MODE != 1 && fok && g.ba) ? g.ba[f] : 0.f;
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        float a2[2], hh[2]; unsigned zpair = 0u;
        const long long zi = zblk + ((((wave * 2 + i) * 2 + j) * 8 + rp) << 6) + lane;
        if (MODE == 1) zpair = g.zbuf[zi];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = 2 * rp + h; const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const long long m = m0 + rl;
          float v = acc[i][j][r] + bias;
          if (MODE == 0) {
            const unsigned zb = f32_to_bf16(v);
            zpair |= zb << (16 * h);
            v = swishf_(v) * drop_scale(g.rng, g.sid1, (unsigned long long)m * g.Na + f, g.p);
            a2[h] = fok ? v : 0.f;
          } else if (MODE == 1) {
            const float z = __uint_as_float((zpair >> (16 * h)) << 16);
            const float ds = drop_scale(g.rng, g.sid1, (unsigned long long)m * g.Na + f, g.p);
            a2[h] = fok ? v * ds * dswishf_(z) : 0.f;
            hh[h] = fok ? swishf_(z) * ds : 0.f;
          } else a2[h] = v;
          *(unsigned short*)(Areg + res_addr(rl, cl)) = f32_to_bf16(a2[h]);
          if (MODE == 1) *(unsigned short*)(Stage + res_addr(rl, cl)) = f32_to_bf16(hh[h]);
        }
        if (MODE == 0) g.zbuf[zi] = zpair;
      }
    }
  __syncthreads();
  if (MODE != 0) {
    // row-major copies for the weight-gradient products (MODE 1: dz, h1) / the output (MODE 2): 64 rows x 32 chunks of 16 B
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 256 + tid; const int row = idx >> 5, cc = idx & 31; const int col = f0 + cc * 8;
      if (m0 + row >= g.M || col >= g.Na) continue;
      const unsigned a = (unsigned)((cc >> 3) * SUB + row * 128 + (((cc & 7) ^ ch_swz(row)) << 4));
      *(uint4*)(g.o1 + (m0 + row) * g.ldo1 + col) = *(const uint4*)(Areg + a);
      if (MODE == 1) *(uint4*)(g.o2 + (m0 + row) * g.ldo2 + col) = *(const uint4*)(Stage + a);
    }
  }
  if (MODE == 2) return;
  __syncthreads();                                    // (MODE 1) the staging slot is free again before product B's ring reaches it

  // ---- product B: out[64][D] += hidden[64][256] . Wb[:, slice]^T ----
  f32x16 acc2[2][NTB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
  pb.run(4, acc2);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
      const int col = wave * NTB * 32 + j * 32 + (lane & 31);
      if (col >= D) continue;
      const float bias = (MODE == 0 && slice == 0 && g.bb) ? g.bb[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); const long long m = m0 + rl;
        if (m >= g.M) continue;
        float v = acc2[i][j][r];
        if (MODE == 0) {
          v = (v + bias) * drop_scale(g.rng, g.sid2, (unsigned long long)m * D + col, g.p) * g.alpha;
          if (slice == 0) v += g.xin[m * D + col];
        }
        atomicAdd(g.out + m * g.ldo + col, v);
      }
    }
}

template <typename K> int chain_lds(K kern, size_t bytes) {
  static const void* done[16]; static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
  if (ndone < 16) done[ndone++] = (const void*)kern;
  return 0;
}

template <int MODE> int chain_launch(const ChainArgs& g, hipStream_t st) {
  const dim3 grid((unsigned)((g.M + 63) / 64), (unsigned)((g.Na + 255) / 256));
  if (g.D <= 256) {
    const size_t lds = 4 * SUB + RING;
    avec_note_kernel("chain_kernel<2,%d>", MODE);
    if (int r = chain_lds(chain_kernel<2, MODE>, lds)) return r;
    hipLaunchKernelGGL((chain_kernel<2, MODE>), grid, dim3(256), lds, st, g);
  } else {
    const size_t lds = 6 * SUB + RING;
    avec_note_kernel("chain_kernel<3,%d>", MODE);
    if (int r = chain_lds(chain_kernel<3, MODE>, lds)) return r;
    hipLaunchKernelGGL((chain_kernel<3, MODE>), grid, dim3(256), lds, st, g);
  }
  AVEC_LAUNCH_CHECK(); return 0;
}

bool chain_dims_ok(long long M, int D, int N) { return M > 0 && D >= 64 && D <= 384 && D % 8 == 0 && N >= 8 && N % 8 == 0; }

}  // namespace

extern "C" int avec_chain_supported(long long M, int D, int N) { return chain_dims_ok(M, D, N) ? 1 : 0; }
extern "C" long long avec_ffn_chain_zbuf_bytes(long long M, int F) { return ((M + 63) / 64) * ((F + 255) / 256) * 8192LL * 4; }

extern "C" int avec_ffn_chain_fwd(const float* x, const float* ln_g, const float* ln_b, float eps, const void* w1, long long ldw1, const float* b1,
                                  const void* w2, long long ldw2, const float* b2, float alpha, float drop_p, const unsigned long long* rng,
                                  unsigned sid1, unsigned sid2, float* y, float* mean, float* rstd, void* h0, void* zbuf,
                                  long long M, int D, int F, hipStream_t st) {
  AVEC_CHECK_ARG(chain_dims_ok(M, D, F), "ffn_chain_fwd: unsupported dims M=%lld D=%d F=%d (64 <= D <= 384, D %% 8 == 0, F %% 8 == 0)", M, D, F);
  AVEC_CHECK_ARG(x && ln_g && ln_b && w1 && w2 && y && mean && rstd && h0 && zbuf && (drop_p <= 0.f || rng), "ffn_chain_fwd: null pointer");
  AVEC_CHECK_ARG(ldw1 >= D && ldw2 >= F && ldw1 % 8 == 0 && ldw2 % 8 == 0, "ffn_chain_fwd: weight row strides");
  ChainArgs g{};
  g.xin = x; g.ln_w = ln_g; g.ln_b = ln_b; g.eps = eps; g.mean = mean; g.rstd = rstd; g.h0 = (bf16*)h0;
  g.Wa = (const bf16*)w1; g.ldwa = ldw1; g.ba = b1; g.Na = F; g.Wb = (const bf16*)w2; g.ldwb = ldw2; g.bb = b2;
  g.zbuf = (unsigned*)zbuf; g.out = y; g.ldo = D; g.M = M; g.D = D; g.KTA = (D + 63) / 64;
  g.alpha = alpha; g.p = drop_p; g.rng = rng; g.sid1 = sid1; g.sid2 = sid2;
  return chain_launch<0>(g, st);
}

extern "C" int avec_ffn_chain_bwd(const float* dy, const void* w2t, long long ldw2t, const void* w1t, long long ldw1t, const void* zbuf,
                                  float alpha, float drop_p, const unsigned long long* rng, unsigned sid1, unsigned sid2,
                                  void* dacc, void* dz, void* h1, float* dh0, long long M, int D, int F, hipStream_t st) {
  AVEC_CHECK_ARG(chain_dims_ok(M, D, F), "ffn_chain_bwd: unsupported dims M=%lld D=%d F=%d", M, D, F);
  AVEC_CHECK_ARG(dy && w2t && w1t && zbuf && dacc && dz && h1 && dh0 && (drop_p <= 0.f || rng), "ffn_chain_bwd: null pointer");
  AVEC_CHECK_ARG(ldw2t >= D && ldw1t >= F && ldw2t % 8 == 0 && ldw1t % 8 == 0, "ffn_chain_bwd: weight row strides");
  ChainArgs g{};
  g.xin = dy; g.h0 = (bf16*)dacc;
  g.Wa = (const bf16*)w2t; g.ldwa = ldw2t; g.Na = F; g.Wb = (const bf16*)w1t; g.ldwb = ldw1t;
  g.zbuf = (unsigned*)zbuf; g.o1 = (bf16*)dz; g.ldo1 = F; g.o2 = (bf16*)h1; g.ldo2 = F;
  g.out = dh0; g.ldo = D; g.M = M; g.D = D; g.KTA = (D + 63) / 64;
  g.alpha = alpha; g.p = drop_p; g.rng = rng; g.sid1 = sid1; g.sid2 = sid2;
  return chain_launch<1>(g, st);
}

extern "C" int avec_ln_gemm(const float* x, const float* ln_g, const float* ln_b, float eps, const void* w, long long ldw, const float* bias,
                            void* out, long long ldo, float* mean, float* rstd, void* h0, long long M, int D, int N, hipStream_t st) {
  AVEC_CHECK_ARG(chain_dims_ok(M, D, N), "ln_gemm: unsupported dims M=%lld D=%d N=%d", M, D, N);
  AVEC_CHECK_ARG(x && ln_g && ln_b && w && out && mean && rstd && ldw >= D && ldw % 8 == 0 && ldo >= N && ldo % 8 == 0, "ln_gemm: bad arguments");
  ChainArgs g{};
  g.xin = x; g.ln_w = ln_g; g.ln_b = ln_b; g.eps = eps; g.mean = mean; g.rstd = rstd; g.h0 = (bf16*)h0;
  g.Wa = (const bf16*)w; g.ldwa = ldw; g.ba = bias; g.Na = N; g.o1 = (bf16*)out; g.ldo1 = ldo;
  g.M = M; g.D = D; g.KTA = (D + 63) / 64;
  return chain_launch<2>(g, st);
}
