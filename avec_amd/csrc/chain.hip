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
  const float* pin; long long pstride; int npart;      // MODE 0 / 2: the true input is xin + sum_{s < npart} pin[s * pstride + ...] (the partial outputs of a chain kernel
  float* xsum;                           // in front); slice 0 then stores the materialised sum here (npart > 0)
  const float* ln_w; const float* ln_b; float eps;
  float* mean; float* rstd;              // MODE 0 / 2 (slice 0): LayerNorm statistics
  bf16* h0;                              // MODE 0 / 2 (slice 0): LN(x);  MODE 1 (slice 0): dacc          [M][D] row-major
  const bf16* Wa; long long ldwa; const float* ba; int Na;      // product A: weight rows = hidden / output index [Na][D]
  const bf16* Wb; long long ldwb; const float* bb;              // product B: weight rows = output index [D][Na], the slice takes columns 256 s ...
  unsigned* zbuf;                        // z in accumulator order (MODE 0 writes, MODE 1 reads)
  bf16* o1; long long ldo1;              // MODE 1: dz [M][Na];  MODE 2: out [M][Na]
  bf16* o2; long long ldo2;              // MODE 1: h1 [M][Na]
  float* out; long long ostride;         // MODE 0 / 1: partial outputs [slices][M][D] fp32 (slice s at out + s * ostride), plain stores
  long long M; int D, KTA;               // KTA = ceil(D / 64)
  float alpha, p; const unsigned long long* rng; unsigned sid1, sid2;
  unsigned long long* dbg;               // measurement aid (NULL in the product): s_memtime stamps of workgroup (0, 0) and of the last workgroup, 8 each
};

constexpr int SUB = 8192;                // one resident A sub-tile: 64 rows x 128 B
constexpr int TILE_A = 256 * 128;        // weight tile of product A: 256 rows x 128 B
constexpr int RING = 3 * TILE_A;         // 96 KB: 3 stages of product A; product B: 3 x 32 KB (NTB = 2) or 2 x 48 KB (NTB = 3)

// wave-wide sum by DPP (full-rate VALU, no LDS round trips): the total in every lane
#define CH_DPP(v, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += CH_DPP(v, 0xB1, 0xf);             // quad_perm [1,0,3,2]
  v += CH_DPP(v, 0x4E, 0xf);             // quad_perm [2,3,0,1]
  v += CH_DPP(v, 0x141, 0xf);            // row_half_mirror
  v += CH_DPP(v, 0x140, 0xf);            // row_mirror: every lane of a 16-lane row holds the row's sum
  v += CH_DPP(v, 0x142, 0xa);            // row_bcast:15 into rows 1 and 3
  v += CH_DPP(v, 0x143, 0xc);            // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// 16 independent sums, step-major: the six DPP levels of different values interleave (a lone chain stalls on its own latency at one wave per SIMD)
__device__ __forceinline__ void wave_sum_dpp16(float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += CH_DPP(v[r], 0xB1, 0xf);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += CH_DPP(v[r], 0x4E, 0xf);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += CH_DPP(v[r], 0x141, 0xf);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += CH_DPP(v[r], 0x140, 0xf);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += CH_DPP(v[r], 0x142, 0xa);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += CH_DPP(v[r], 0x143, 0xc);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[r]), 63));
}
__device__ __forceinline__ float ch_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }      // (v_rcp_f32: 1 ulp; the IEEE division sequence is ~10 instructions per element)
__device__ __forceinline__ float ch_swish(float x) { return x * ch_sigmoid(x); }
__device__ __forceinline__ float ch_dswish(float x) { const float sg = ch_sigmoid(x); return sg * (1.f + x * (1.f - sg)); }
// dropout scales of 4 consecutive elements whose pair index fits 32 bits (host-checked: the tensor has < 2^32 elements)
__device__ __forceinline__ void drop4_32(const DropKey& k, unsigned idx, float (&s)[4]) {
  const unsigned h0 = mix32((idx >> 1) ^ k.k0), h1 = mix32(((idx >> 1) + 1u) ^ k.k0);
  s[0] = (h0 & 0xffffu) >= k.thr ? k.scale : 0.f; s[1] = (h0 >> 16) >= k.thr ? k.scale : 0.f;
  s[2] = (h1 & 0xffffu) >= k.thr ? k.scale : 0.f; s[3] = (h1 >> 16) >= k.thr ? k.scale : 0.f;
}

// one product phase: acc[i][j] (+)= tile (A row block i, W row block j), TRANSPOSED in the registers: lane & 31 = A row (32 i + ...), register r = W row
// 32 j + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- a lane holds 4 consecutive W rows (= output columns) per register quad, i.e. 8 / 16 contiguous bytes of a row-major
// output row.  A resident at lds_a (sub-tile kt), W tiles through the ring.
template <int NT, int STAGES, int TILE>
struct Phase {
  static constexpr int NPASS = TILE / 4096;                      // DMA instructions per thread and tile
  unsigned off[NPASS];                                           // per-lane byte offsets of the tile rows this thread fetches (chunk swizzle folded in)
  unsigned offl[NPASS];                                          // ... for the LAST K tile, relative to wbase: a chunk at or beyond K re-reads the row's first chunk instead
                                                                 // (finite data that meets a zero of the A operand -- never bytes behind the matrix)
  int KT;                                                        // K tiles of 64
  const char* wbase;                                             // scalar: W + column offset of this slice / K origin
  unsigned lds_ring, lds_a, wslot;
  unsigned aad[4], bad[4];                                       // fragment addresses of K-substep q (stage / sub-tile offsets are added as scalars)

  __device__ __forceinline__ void plan(const void* W, long long ldw, int row0, int nrows_valid, long long col0_bytes, int Kvalid, int tid, unsigned lds0_ring, unsigned lds0_a, int wave, int lane, int brow0) {
    wbase = (const char*)W + col0_bytes;
    lds_ring = lds0_ring; lds_a = lds0_a; wslot = (unsigned)wave * 1024u;
    KT = (Kvalid + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int r = (tid >> 3) + 32 * i; int n = row0 + r; n = n < nrows_valid ? n : nrows_valid - 1;
      const int kc = (tid & 7) ^ ch_swz(r);
      off[i] = (unsigned)(((long long)n * ldw + kc * 8) * 2);
      offl[i] = ((KT - 1) * 64 + kc * 8 < Kvalid) ? off[i] + (unsigned)(KT - 1) * 128u : (unsigned)((long long)n * ldw * 2);
    }
    const int g = lane >> 5, ra = lane & 31, rb = brow0 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      aad[q] = lds0_a + (unsigned)(ra * 128 + (((2 * q + g) ^ ch_swz(ra)) << 4));
      bad[q] = lds0_ring + (unsigned)(rb * 128 + (((2 * q + g) ^ ch_swz(rb)) << 4));
    }
  }
  template <int S> __device__ __forceinline__ void issue(int kt) const {
    const unsigned dst = lds_ring + S * TILE + wslot;
    if (kt < KT - 1) {
      const char* src = wbase + (long long)kt * 128;
#pragma unroll
      for (int i = 0; i < NPASS; i += 4) ch_glds4(off[i], off[i + 1], off[i + 2], off[i + 3], src, dst + i * 4096);
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; i += 4) ch_glds4(offl[i], offl[i + 1], offl[i + 2], offl[i + 3], wbase, dst + i * 4096);
    }
  }
  template <int S> __device__ __forceinline__ void step(int kt, f32x16 (&acc)[2][NT]) const {
    const int rem = KT - 1 - kt;                                   // tiles issued after kt that may still be in flight: min(rem, STAGES - 2)
    if (STAGES >= 3 && rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPASS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        for (int j = 0; j < NT; ++j)      // W fragment first: the accumulator comes out transposed (see above)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[q][j]), __builtin_bit_cast(bf16x8_t, fa[q][i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __device__ __forceinline__ void prologue() const {
    issue<0>(0);
    if (STAGES >= 3 && KT > 1) issue<1 % STAGES>(1);
  }
  __device__ __forceinline__ void run(f32x16 (&acc)[2][NT]) const {
#pragma unroll 1
    for (int kt = 0; kt < KT; kt += STAGES) {
      step<0>(kt, acc);
      if (STAGES > 1 && kt + 1 < KT) step<1 % STAGES>(kt + 1, acc);
      if (STAGES > 2 && kt + 2 < KT) step<2 % STAGES>(kt + 2, acc);
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
  const bool dbg = g.dbg && tid == 0 && ((blockIdx.x == 0 && blockIdx.y == 0) || (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1));
  unsigned long long* const dslot = g.dbg + ((blockIdx.x == 0 && blockIdx.y == 0) ? 0 : 8);
#define CH_STAMP(i) do { if (dbg) dslot[i] = __builtin_amdgcn_s_memtime(); } while (0)
  CH_STAMP(0);

  // ---- prologue: the resident A operand (64 rows x KTA*64, bf16) ----
  // A wave owns rows 16 w .. 16 w + 15 and requests ALL of them before any arithmetic, and before the weight DMA and the RNG state (one memory round trip for all three:
  // at one wave per SIMD nothing else hides a second one); loads are unconditional (clamped column): hipcc waits on the spot for a load issued under a divergent branch.
  constexpr int NG = NTB == 2 ? 1 : 2;               // column groups of 256 per lane
  float v[16][NG][4]; long long mrow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long m = m0 + wave * 16 + r; mrow[r] = m < g.M ? m : g.M - 1;
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      const int c = lane * 4 + gq * 256;
      ld4<float>(g.xin + mrow[r] * D + (c < D ? c : D - 4), v[r][gq]);
    }
  }
  unsigned long long rng0 = 0ull, rng1 = 0ull;
  if (g.p > 0.f) { rng0 = g.rng[0]; rng1 = g.rng[1]; }      // {seed, step} read ONCE (behind a store the compiler would reload them per element)
  float lng[NG][4], lnb[NG][4];
  if (MODE != 1) {
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) { const int c = lane * 4 + gq * 256; ld4<float>(g.ln_w + (c < D ? c : D - 4), lng[gq]); ld4<float>(g.ln_b + (c < D ? c : D - 4), lnb[gq]); }
  }
  asm volatile("" ::: "memory");

  Phase<2, 3, TILE_A> pa;
  pa.plan(g.Wa, g.ldwa, f0, g.Na, 0, D, tid, lds_ring, lds_a, wave, lane, wave * 64);
  pa.prologue();                                   // the first weight tiles travel while the prologue computes the A operand

  DropKey dk1, dk2;
  {
    const unsigned long long seed = rng0 + 0x9e3779b97f4a7c15ull * rng1;
    const float t = g.p * 65536.f + 0.5f; const unsigned thr = g.p > 0.f ? (t >= 65535.f ? 65535u : (unsigned)t) : 0u; const float sc = g.p > 0.f ? 1.f / (1.f - g.p) : 1.f;
    dk1.k0 = mix32((unsigned)seed + g.sid1 * 0x9e3779b9u) ^ mix32((unsigned)(seed >> 32) ^ 0x85ebca6bu); dk1.thr = thr; dk1.scale = sc;      // = drop_key() of common.h
    dk2.k0 = mix32((unsigned)seed + g.sid2 * 0x9e3779b9u) ^ mix32((unsigned)(seed >> 32) ^ 0x85ebca6bu); dk2.thr = thr; dk2.scale = sc;
  }
  {
    if (MODE != 1 && g.npart > 0) {                  // the input is a sum of partial outputs: add them pass by pass (16 rows in flight each), slice 0 stores the sum
#pragma unroll 1
      for (int s = 0; s < g.npart; ++s) {
#pragma unroll
        for (int hf = 0; hf < NG; ++hf) {             // (16 / NG rows in flight per round trip: the register budget)
          constexpr int RH = 16 / NG;
          float t[RH][NG][4];
#pragma unroll
          for (int r = 0; r < RH; ++r)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) { const int c = lane * 4 + gq * 256; ld4<float>(g.pin + s * g.pstride + mrow[hf * RH + r] * D + (c < D ? c : D - 4), t[r][gq]); }
#pragma unroll
          for (int r = 0; r < RH; ++r)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
#pragma unroll
              for (int e = 0; e < 4; ++e) v[hf * RH + r][gq][e] += t[r][gq][e];
        }
      }
      if (slice == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) { const int c = lane * 4 + gq * 256; if (c < D && m0 + wave * 16 + r < g.M) st4<float>(g.xsum + mrow[r] * D + c, v[r][gq]); }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) { const int c = lane * 4 + gq * 256; if (c >= D) { v[r][gq][0] = v[r][gq][1] = v[r][gq][2] = v[r][gq][3] = 0.f; } }
    if (MODE != 1) {
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) { const int c = lane * 4 + gq * 256; if (c >= D) { for (int e = 0; e < 4; ++e) { lng[gq][e] = 0.f; lnb[gq][e] = 0.f; } } }
      const float invD = 1.f / D;
      float mu[16], rs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float t = 0.f;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) t += v[r][gq][0] + v[r][gq][1] + v[r][gq][2] + v[r][gq][3];
        mu[r] = t;
      }
      wave_sum_dpp16(mu);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mu[r] *= invD;
        float q = 0.f;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) { const int c = lane * 4 + gq * 256; if (c < D) for (int e = 0; e < 4; ++e) { const float d = v[r][gq][e] - mu[r]; q += d * d; } }
        rs[r] = q;
      }
      wave_sum_dpp16(rs);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rs[r] = rsqrtf(rs[r] * invD + g.eps);
#pragma unroll
        for (int gq = 0; gq < NG; ++gq)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[r][gq][e] = (v[r][gq][e] - mu[r]) * rs[r] * lng[gq][e] + lnb[gq][e];
      }
      if (slice == 0 && lane < 16) {                 // lane r stores the statistics of row r (they are wave-uniform values)
        float a = mu[0], b = rs[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) { a = lane == r ? mu[r] : a; b = lane == r ? rs[r] : b; }
        const long long m = m0 + wave * 16 + lane;
        if (m < g.M) { g.mean[m] = a; g.rstd[m] = b; }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          float ds[4] = {1.f, 1.f, 1.f, 1.f};
          if (g.p > 0.f) drop4_32(dk2, (unsigned)(mrow[r] * D) + lane * 4 + gq * 256, ds);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[r][gq][e] *= g.alpha * ds[e];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = wave * 16 + r;
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        const int c = lane * 4 + gq * 256;
        if (c >= KTA * 64) continue;
        uint2 t;
        t.x = f32x2_to_bf16x2(v[r][gq][0], v[r][gq][1]); t.y = f32x2_to_bf16x2(v[r][gq][2], v[r][gq][3]);      // (zeros beyond D)
        *(uint2*)(Areg + res_addr(rl, c)) = t;
        if (slice == 0 && g.h0 && c < D && m0 + rl < g.M) *(uint2*)(g.h0 + mrow[r] * D + c) = t;
      }
    }
  }
  // (the first barrier of phase A publishes the A operand)
  CH_STAMP(1);

  // ---- product A: hidden[64][256 of this slice]; wave w owns hidden columns 64 w .. 64 w + 63: acc[i][j] = rows 32 i + (lane & 31), columns 64 w + 32 j + quad ----
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  pa.run(acc);
  __syncthreads();                                  // every wave is done with the A operand and the ring
  CH_STAMP(2);

  Phase<NTB, STB, TILE_B> pb;
  if (MODE != 2) {
    pb.plan(g.Wb, g.ldwb, 0, D, (long long)f0 * 2, (g.Na - f0 < 256 ? g.Na - f0 : 256), tid, lds_ring, lds_a, wave, lane, wave * NTB * 32);
    pb.prologue();                                   // the first tile(s) of product B travel under the middle part
  }

  // ---- middle: element-wise work on the hidden tile; the result becomes the resident A operand of product B ----
  // a lane holds, per (i, j, g): row 32 i + (lane & 31), the 4 consecutive hidden columns cl .. cl + 3, cl = 64 w + 32 j + 8 g + 4 (lane >> 5)
  const long long zblk = ((long long)blockIdx.x * gridDim.y + blockIdx.y) * 8192;
  char* const Stage = Ring + (STB - 1) * TILE_B;     // MODE 1: h1 staging (the last ring slot: not a target of pb.prologue)
  uint2 zreg[2][2][4];
  if (MODE == 1) {                                   // every z word of this lane requested before the arithmetic starts
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) zreg[i][j][gq] = *(const uint2*)(g.zbuf + zblk + (((((wave * 2 + i) * 2 + j) * 4 + gq) << 6) + lane) * 2);
  }
  float bias4[2][4][4];
  if (MODE != 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int f = f0 + wave * 64 + j * 32 + gq * 8 + 4 * (lane >> 5);
        if (g.ba) ld4<float>(g.ba + (f + 3 < g.Na ? f : g.Na - 4), bias4[j][gq]); else { bias4[j][gq][0] = bias4[j][gq][1] = bias4[j][gq][2] = bias4[j][gq][3] = 0.f; }
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = i * 32 + (lane & 31);
    const unsigned rowidx = (unsigned)((m0 + rl) * g.Na);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int cl = wave * 64 + j * 32 + gq * 8 + 4 * (lane >> 5); const int f = f0 + cl; const bool fok = f < g.Na;      // (Na % 8 == 0: the quad is inside or outside as a whole)
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (MODE != 2 && g.p > 0.f) drop4_32(dk1, rowidx + (unsigned)f, ds);
        float a2[4], hh[4]; uint2 zp = MODE == 1 ? zreg[i][j][gq] : make_uint2(0u, 0u);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][4 * gq + e];
          if (MODE == 0) {
            v += bias4[j][gq][e];
            const unsigned zb = f32_to_bf16(v);
            if (e < 2) zp.x |= zb << (16 * e); else zp.y |= zb << (16 * (e - 2));
            a2[e] = fok ? ch_swish(v) * ds[e] : 0.f;
          } else if (MODE == 1) {
            const unsigned zw = e < 2 ? zp.x : zp.y;
            const float z = __uint_as_float((zw >> (16 * (e & 1))) << 16);
            a2[e] = fok ? v * ds[e] * ch_dswish(z) : 0.f;
            hh[e] = fok ? ch_swish(z) * ds[e] : 0.f;
          } else a2[e] = v + bias4[j][gq][e];
        }
        uint2 t; t.x = f32x2_to_bf16x2(a2[0], a2[1]); t.y = f32x2_to_bf16x2(a2[2], a2[3]);
        *(uint2*)(Areg + res_addr(rl, cl)) = t;
        if (MODE == 1) { uint2 u; u.x = f32x2_to_bf16x2(hh[0], hh[1]); u.y = f32x2_to_bf16x2(hh[2], hh[3]); *(uint2*)(Stage + res_addr(rl, cl)) = u; }
        if (MODE == 0) zreg[i][j][gq] = zp;
      }
  }
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) *(uint2*)(g.zbuf + zblk + (((((wave * 2 + i) * 2 + j) * 4 + gq) << 6) + lane) * 2) = zreg[i][j][gq];
  }
  __syncthreads();
  CH_STAMP(3);
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
  CH_STAMP(4);
  if (MODE == 2) return;
  __syncthreads();                                  // (MODE 1) the staging slot is free again before product B's ring reaches it

  // ---- product B: partial out[64][D] = hidden[64][256] . Wb[:, slice]^T; wave w owns output columns 32 NTB w ... ----
  f32x16 acc2[2][NTB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
  pb.run(acc2);
  CH_STAMP(5);
  float* const outp = g.out + (long long)slice * g.ostride;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long long m = m0 + i * 32 + (lane & 31);
    const unsigned rowidx = (unsigned)(m * D);
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int col = wave * NTB * 32 + j * 32 + gq * 8 + 4 * (lane >> 5);
        if (col >= D || m >= g.M) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc2[i][j][4 * gq + e];
        if (MODE == 0) {
          float ds[4] = {1.f, 1.f, 1.f, 1.f};
          if (g.p > 0.f) drop4_32(dk2, rowidx + (unsigned)col, ds);
          float b4[4] = {0.f, 0.f, 0.f, 0.f};
          if (slice == 0 && g.bb) ld4<float>(g.bb + col, b4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] + b4[e]) * ds[e] * g.alpha;
        }
        st4<float>(outp + m * D + col, v);
      }
  }
  CH_STAMP(6);
#undef CH_STAMP
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

// (the dropout indices are 32-bit inside the kernel: M * max(D, N) < 2^32)
bool chain_dims_ok(long long M, int D, int N) { return M > 0 && D >= 64 && D <= 384 && D % 8 == 0 && N >= 8 && N % 8 == 0 && M * (long long)(N > D ? N : D) < (1ll << 32); }

static unsigned long long* g_chain_dbg = nullptr;

}  // namespace

extern "C" int avec_chain_debug_stamps(unsigned long long* dev16) { g_chain_dbg = dev16; return 0; }
extern "C" int avec_chain_supported(long long M, int D, int N) { return chain_dims_ok(M, D, N) ? 1 : 0; }
extern "C" int avec_chain_slices(int N) { return (N + 255) / 256; }
extern "C" long long avec_ffn_chain_zbuf_bytes(long long M, int F) { return ((M + 63) / 64) * ((F + 255) / 256) * 8192LL * 4; }

extern "C" int avec_ffn_chain_fwd(const float* x, const float* ln_g, const float* ln_b, float eps, const void* w1, long long ldw1, const float* b1,
                                  const void* w2, long long ldw2, const float* b2, float alpha, float drop_p, const unsigned long long* rng,
                                  unsigned sid1, unsigned sid2, float* yparts, float* mean, float* rstd, void* h0, void* zbuf,
                                  long long M, int D, int F, hipStream_t st) {
  AVEC_CHECK_ARG(chain_dims_ok(M, D, F), "ffn_chain_fwd: unsupported dims M=%lld D=%d F=%d (64 <= D <= 384, D %% 8 == 0, F %% 8 == 0)", M, D, F);
  AVEC_CHECK_ARG(x && ln_g && ln_b && w1 && w2 && yparts && mean && rstd && h0 && zbuf && (drop_p <= 0.f || rng), "ffn_chain_fwd: null pointer");
  AVEC_CHECK_ARG(ldw1 >= D && ldw2 >= F && ldw1 % 8 == 0 && ldw2 % 8 == 0, "ffn_chain_fwd: weight row strides");
  ChainArgs g{};
  g.xin = x; g.ln_w = ln_g; g.ln_b = ln_b; g.eps = eps; g.mean = mean; g.rstd = rstd; g.h0 = (bf16*)h0;
  g.Wa = (const bf16*)w1; g.ldwa = ldw1; g.ba = b1; g.Na = F; g.Wb = (const bf16*)w2; g.ldwb = ldw2; g.bb = b2;
  g.zbuf = (unsigned*)zbuf; g.out = yparts; g.ostride = M * D; g.M = M; g.D = D; g.KTA = (D + 63) / 64;
  g.alpha = alpha; g.p = drop_p; g.rng = rng; g.sid1 = sid1; g.sid2 = sid2;
  g.dbg = g_chain_dbg;
  return chain_launch<0>(g, st);
}

extern "C" int avec_ffn_chain_bwd(const float* dy, const void* w2t, long long ldw2t, const void* w1t, long long ldw1t, const void* zbuf,
                                  float alpha, float drop_p, const unsigned long long* rng, unsigned sid1, unsigned sid2,
                                  void* dacc, void* dz, void* h1, float* dh0parts, long long M, int D, int F, hipStream_t st) {
  AVEC_CHECK_ARG(chain_dims_ok(M, D, F), "ffn_chain_bwd: unsupported dims M=%lld D=%d F=%d", M, D, F);
  AVEC_CHECK_ARG(dy && w2t && w1t && zbuf && dacc && dz && h1 && dh0parts && (drop_p <= 0.f || rng), "ffn_chain_bwd: null pointer");
  AVEC_CHECK_ARG(ldw2t >= D && ldw1t >= F && ldw2t % 8 == 0 && ldw1t % 8 == 0, "ffn_chain_bwd: weight row strides");
  ChainArgs g{};
  g.xin = dy; g.h0 = (bf16*)dacc;
  g.Wa = (const bf16*)w2t; g.ldwa = ldw2t; g.Na = F; g.Wb = (const bf16*)w1t; g.ldwb = ldw1t;
  g.zbuf = (unsigned*)zbuf; g.o1 = (bf16*)dz; g.ldo1 = F; g.o2 = (bf16*)h1; g.ldo2 = F;
  g.out = dh0parts; g.ostride = M * D; g.M = M; g.D = D; g.KTA = (D + 63) / 64;
  g.alpha = alpha; g.p = drop_p; g.rng = rng; g.sid1 = sid1; g.sid2 = sid2;
  g.dbg = g_chain_dbg;
  return chain_launch<1>(g, st);
}

extern "C" int avec_ln_gemm(const float* x, const float* xparts, int nparts, float* xsum, const float* ln_g, const float* ln_b, float eps, const void* w, long long ldw,
                            const float* bias, void* out, long long ldo, float* mean, float* rstd, void* h0, long long M, int D, int N, hipStream_t st) {
  AVEC_CHECK_ARG(chain_dims_ok(M, D, N), "ln_gemm: unsupported dims M=%lld D=%d N=%d", M, D, N);
  AVEC_CHECK_ARG(x && ln_g && ln_b && w && out && mean && rstd && ldw >= D && ldw % 8 == 0 && ldo >= N && ldo % 8 == 0, "ln_gemm: bad arguments");
  AVEC_CHECK_ARG(nparts == 0 || (xparts && xsum && nparts > 0), "ln_gemm: partial inputs need xparts and xsum");
  ChainArgs g{};
  g.xin = x; g.pin = xparts; g.pstride = M * D; g.npart = nparts; g.xsum = xsum;
  g.ln_w = ln_g; g.ln_b = ln_b; g.eps = eps; g.mean = mean; g.rstd = rstd; g.h0 = (bf16*)h0;
  g.Wa = (const bf16*)w; g.ldwa = ldw; g.ba = bias; g.Na = N; g.o1 = (bf16*)out; g.ldo1 = ldo;
  g.M = M; g.D = D; g.KTA = (D + 63) / 64;
  g.dbg = g_chain_dbg;
  return chain_launch<2>(g, st);
}
