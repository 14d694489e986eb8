// Weight gradient of the wide 3x3 / stride-1 / pad-1 layers of the ResNet (stages 2..4: C = 128 / 256 / 512 channels on 11x11 / 6x6 / 3x3 images;
// nnet/blocks.py:29-91 ResNetBlock, nnet/networks.py:32-146), "pair" formulation (round 4).
//
//     dW[co][kh][kw][ci] = sum over images, oy, ox of  dy[img][oy][ox][co] * x[img][oy + kh - 1][ox + kw - 1][ci]        (zero outside the image)
//
// The slab kernel of round 2/3 (conv3x3.hip: wgrad3x3_wide_kernel) flattens an image to rows of pitch W+1 and multiplies shifted views of it: every MFMA of a
// K-step takes its own transposed x fragment (1.1 fragment reads per MFMA: the LDS pipe is the limit) and 16-44 % of the reduction rows are padding (3x3 images use
// 9 of 16).  Here the reduction index is the IMAGE ROW  r = img * H + oy  (contiguous in memory, no padding), and the pixel position inside the row is unrolled:
// for an output column ox and a tap (kh, kw) with 0 <= ox + kw - 1 < W
//     A_ox[r][co]         = dy[(r * W + ox) * C + co]
//     B_{x',kh}[r][ci]    = x[((r + kh - 1) * W + x') * C + ci],   x' = ox + kw - 1        (row r + kh - 1 must belong to the same image: k-rows with oy = 0
//                                                                                            (kh = 0) or oy = H-1 (kh = 2) are masked out of the fragment)
//     acc[kh][kw]        += A_ox^T  B_{x',kh}
// One K-step (16 image rows) needs W A fragments and 3 W B fragments for 3 (3 W - 2) MFMAs: 0.44 fragment reads per MFMA, and the column pairs that fall outside
// the image are never issued (3x3 images: 21 of the 27 nominal (ox, tap) products per row; the slab kernel issued 16/9 x 27).
// Workgroup = 8 waves = 64 output channels x 128 input channels x 9 taps (wave: one 32 x 32 block, nine accumulator tiles), a slice of the image rows; LDS = a ring of
// two stages of KS K-steps (dy rows [r, r + 16 KS) x 64 channels, x rows [r - 1, r + 16 KS + 1) x 128 channels) fed by LDS-DMA; fp32 atomics at the end, several
// layers per launch (avec_wgrad3x3_c128_grouped).
#include "common.h"
#include "vec.h"
#include "avec_hip.h"

#ifndef WP_ABL
#define WP_ABL 0          // timing experiments: 1 no LDS-DMA after the first stage, 2 no MFMA, 4 no final atomics, 8 no row masks
#endif
typedef __attribute__((ext_vector_type(16))) float wp_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 wp_bf16x8;
typedef short wp_v4s __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(64))) unsigned char wp_zero16[64];

__device__ __forceinline__ void wp_glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ uint2 wp_tr(const char* p) {
  typedef __attribute__((address_space(3))) wp_v4s* lp_t;
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p));
}
struct WpFrag { uint2 lo, hi; };      // k-elements 0..3 | 4..7 of the lane's eight
__device__ __forceinline__ wp_f32x16 wp_mma(const WpFrag& a, const WpFrag& b, wp_f32x16 c) {
  chunk16 fa, fb; fa.w[0] = a.lo.x; fa.w[1] = a.lo.y; fa.w[2] = a.hi.x; fa.w[3] = a.hi.y; fb.w[0] = b.lo.x; fb.w[1] = b.lo.y; fb.w[2] = b.hi.x; fb.w[3] = b.hi.y;
  if (WP_ABL & 2) { asm volatile("" :: "v"(fa.w[0]), "v"(fa.w[1]), "v"(fa.w[2]), "v"(fa.w[3]), "v"(fb.w[0]), "v"(fb.w[1]), "v"(fb.w[2]), "v"(fb.w[3])); return c; }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wp_bf16x8, fa), __builtin_bit_cast(wp_bf16x8, fb), c, 0, 0, 0);
}

struct WpArgs { const bf16* x; const bf16* dy; float* dw; int N, C, H, W; };

template <int H_, int W_, int KS_> struct WpGeom {
  static constexpr int H = H_, W = W_, KS = KS_;
  static constexpr int RWD = W | 1;                       // dy slab: pixels per image row, odd (128-byte pixels: consecutive k-rows must alternate between the two bank halves)
  static constexpr int RWX = W;                           // x slab (256-byte pixels: the swizzle alone spreads the banks)
  static constexpr int ROWS = 16 * KS;                    // image rows (K) per stage
  static constexpr int DYB = ROWS * RWD * 128, XB = (ROWS + 2) * RWX * 256;
  static constexpr int DYI = (DYB + 1023) / 1024, XI = (XB + 1023) / 1024;      // DMA instructions (1 KB each) per stage
  static constexpr int DPL = (DYI + 7) / 8, XPL = (XI + 7) / 8;                 // ... per wave
  static constexpr int STAGE = (DYI + XI) * 1024;
  static constexpr int PAIRS = 3 * (3 * W - 2);           // MFMAs per wave and K-step
};

// LDS-DMA with a scalar base and a per-lane 32-bit byte offset (no VALU on the issue path)
__device__ __forceinline__ void wp_glds16_s(unsigned voff, const void* sbase, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst_uniform) : "memory");
}
template <int OFF> __device__ __forceinline__ uint2 wp_tr_o(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  uint2 v; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory"); return v;
}

// ---- one stage of the reduction as KS * W software-pipelined iterations.  Iteration I = (K-step s, x column xp) runs its <= 9 MFMAs tap row by tap row
// (kh = 0, 1, 2) and re-issues the transposed reads of a tap row's x fragment for iteration I + 1 as soon as its three MFMAs are out (inline asm: the compiler
// neither reorders them nor waits behind them); every group waits for its OWN fragment by count (the LDS returns in order), so a fragment has the time of six
// MFMAs to arrive.  Ablation of the first version (compiler-scheduled reads, waited for right where they were issued; 9 layers grouped, 936 us): the reads / masks /
// barriers alone took 258 us, the MFMAs on top of them 579 us (400 us at the full rate), the DMA waits 100 us -- the parts added up.
struct WpRegs { WpFrag A[4]; WpFrag B[3]; };      // A: dy fragments of columns g - 1 .. g + 2 (ring, g = s W + ox); B: x fragments of the three tap rows

template <typename G, int I> struct WpIt {
  static constexpr int W = G::W, s = I / W, xp = I % W, g = s * W + xp;
  static constexpr int N0 = 2 + 2 * ((xp == 0 ? 1 : 0) + (xp + 1 < W ? 1 : 0));      // reads of the first group: B0 and the new dy fragments
  static __device__ __forceinline__ void issue0(WpRegs& R, const unsigned (&aA)[2], const unsigned (&aB)[3][2]) {
    if constexpr (xp == 0) { R.A[g & 3].lo = wp_tr_o<(16 * s * G::RWD) * 128>(aA[0]); R.A[g & 3].hi = wp_tr_o<(16 * s * G::RWD) * 128>(aA[1]); }
    if constexpr (xp + 1 < W) { R.A[(g + 1) & 3].lo = wp_tr_o<(16 * s * G::RWD + xp + 1) * 128>(aA[0]); R.A[(g + 1) & 3].hi = wp_tr_o<(16 * s * G::RWD + xp + 1) * 128>(aA[1]); }
    R.B[0].lo = wp_tr_o<(16 * s * G::RWX + xp) * 256>(aB[0][0]); R.B[0].hi = wp_tr_o<(16 * s * G::RWX + xp) * 256>(aB[0][1]);
  }
  template <int KH> static __device__ __forceinline__ void issueB(WpRegs& R, const unsigned (&aB)[3][2]) {
    R.B[KH].lo = wp_tr_o<(16 * s * G::RWX + xp) * 256>(aB[KH][0]); R.B[KH].hi = wp_tr_o<(16 * s * G::RWX + xp) * 256>(aB[KH][1]);
  }
  static __device__ __forceinline__ void arrived0(WpRegs& R) {      // pins the consumers of the fragments below the counted wait
    if constexpr (xp == 0) asm volatile("" : "+v"(R.A[g & 3].lo), "+v"(R.A[g & 3].hi));
    if constexpr (xp + 1 < W) asm volatile("" : "+v"(R.A[(g + 1) & 3].lo), "+v"(R.A[(g + 1) & 3].hi));
    asm volatile("" : "+v"(R.B[0].lo), "+v"(R.B[0].hi));
  }
};

template <typename G, int I>
__device__ __forceinline__ void wp_iter(WpRegs& R, wp_f32x16 (&acc)[9], unsigned (&m0)[4], unsigned (&m2)[4], const int oyb, const bool hi_half, const unsigned (&aA)[2], const unsigned (&aB)[3][2]) {
  typedef WpIt<G, I> It;
  constexpr int H = G::H, W = G::W, NIT = G::KS * W, g = It::g;
  constexpr bool more = I + 1 < NIT;
  constexpr int N0N = more ? WpIt<G, more ? I + 1 : I>::N0 : 0;      // reads of the next iteration's first group
  if constexpr (It::xp == 0) {
    // k-element e of this lane is image row oy = (oyb + 8 (lane >> 5) + 16 s + e) % H of its image (oyb: the stage's first row, wave-uniform): kh = 0 pairs it with
    // row oy - 1 (invalid: oy = 0), kh = 2 with row oy + 1 (invalid: oy = H - 1).  The invalid elements are e0, e0 + H, ... with e0 = (-first row) mod H, and one
    // element earlier for kh = 2.  Both half-waves' masks are built on the SCALAR unit and selected per lane (built per lane they were ~50 VALU instructions per
    // K-step: 90 of the launch's 740 us without the DMA).
    constexpr unsigned PAT = 1u | (1u << H) | (2 * H < 16 ? 1u << ((2 * H) & 15) : 0u) | (3 * H < 16 ? 1u << ((3 * H) & 15) : 0u) | (4 * H < 16 ? 1u << ((4 * H) & 15) : 0u);      // bits k H < 16
    unsigned ml0[2][4], ml2[2][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      int bs = oyb + (8 * hh + 16 * It::s) % H; bs = bs >= H ? bs - H : bs;
      const int e0 = bs == 0 ? 0 : H - bs, e2 = e0 == 0 ? H - 1 : e0 - 1;
      const unsigned b0 = (WP_ABL & 8) ? 0u : (PAT << e0), b2 = (WP_ABL & 8) ? 0u : (PAT << e2);
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        ml0[hh][dd] = ~((((b0 >> (2 * dd)) & 1u) * 0xffffu) | (((b0 >> (2 * dd + 1)) & 1u) * 0xffff0000u));
        ml2[hh][dd] = ~((((b2 >> (2 * dd)) & 1u) * 0xffffu) | (((b2 >> (2 * dd + 1)) & 1u) * 0xffff0000u));
      }
    }
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) { m0[dd] = hi_half ? ml0[1][dd] : ml0[0][dd]; m2[dd] = hi_half ? ml2[1][dd] : ml2[0][dd]; }
  }
  // tap (kh, kw) pairs x column xp with dy column ox = xp + 1 - kw
  // ---- kh = 0: in flight behind this group's fragments: B1, B2 of this iteration
  asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  It::arrived0(R);
  R.B[0].lo.x &= m0[0]; R.B[0].lo.y &= m0[1]; R.B[0].hi.x &= m0[2]; R.B[0].hi.y &= m0[3];
  if constexpr (It::xp + 1 < W) acc[0] = wp_mma(R.A[(g + 1) & 3], R.B[0], acc[0]);
  acc[1] = wp_mma(R.A[g & 3], R.B[0], acc[1]);
  if constexpr (It::xp >= 1) acc[2] = wp_mma(R.A[(g - 1) & 3], R.B[0], acc[2]);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (more) WpIt<G, more ? I + 1 : I>::issue0(R, aA, aB);
  // ---- kh = 1: behind B1: B2 and the next iteration's first group
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 + N0N) : "memory");
  asm volatile("" : "+v"(R.B[1].lo), "+v"(R.B[1].hi));
  if constexpr (It::xp + 1 < W) acc[3] = wp_mma(R.A[(g + 1) & 3], R.B[1], acc[3]);
  acc[4] = wp_mma(R.A[g & 3], R.B[1], acc[4]);
  if constexpr (It::xp >= 1) acc[5] = wp_mma(R.A[(g - 1) & 3], R.B[1], acc[5]);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (more) WpIt<G, more ? I + 1 : I>::template issueB<1>(R, aB);
  // ---- kh = 2: behind B2: the next iteration's first group and its B1
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(more ? N0N + 2 : 0) : "memory");
  asm volatile("" : "+v"(R.B[2].lo), "+v"(R.B[2].hi));
  R.B[2].lo.x &= m2[0]; R.B[2].lo.y &= m2[1]; R.B[2].hi.x &= m2[2]; R.B[2].hi.y &= m2[3];
  if constexpr (It::xp + 1 < W) acc[6] = wp_mma(R.A[(g + 1) & 3], R.B[2], acc[6]);
  acc[7] = wp_mma(R.A[g & 3], R.B[2], acc[7]);
  if constexpr (It::xp >= 1) acc[8] = wp_mma(R.A[(g - 1) & 3], R.B[2], acc[8]);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (more) { WpIt<G, more ? I + 1 : I>::template issueB<2>(R, aB); wp_iter<G, more ? I + 1 : I>(R, acc, m0, m2, oyb, hi_half, aA, aB); }
}

// stages [s_begin, s_end) of the reduction for one kind (64 output x 128 input channels) of layer `a`; adds the result to a.dw
// (not inlined: the register allocation of each geometry's loop stays its own -- inlined three times into the work-sharing loops of the kernel it spilled)
template <typename G>
__device__ __attribute__((noinline)) void wp_body(const WpArgs a_, char* const smem_, const int kind_, const int s_begin_, const int s_end_) {
  constexpr int H = G::H, W = G::W, RWD = G::RWD, RWX = G::RWX, ROWS = G::ROWS;
  // (arguments of a real call arrive in vector registers: make the uniform ones scalar again)
  auto uni = [](const void* p) { const unsigned long long v = (unsigned long long)p;
    return (const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v)); };
  WpArgs a; a.x = (const bf16*)uni(a_.x); a.dy = (const bf16*)uni(a_.dy); a.dw = (float*)uni(a_.dw);
  a.N = __builtin_amdgcn_readfirstlane(a_.N); a.C = __builtin_amdgcn_readfirstlane(a_.C); a.H = a_.H; a.W = a_.W;
  char* const smem = (char*)uni(smem_);
  const int kind = __builtin_amdgcn_readfirstlane(kind_), s_begin = __builtin_amdgcn_readfirstlane(s_begin_), s_end = __builtin_amdgcn_readfirstlane(s_end_);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = a.C;
  const int nci = C >> 7, cog = kind / nci, cig = kind - cog * nci;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const int NR = a.N * H;                                  // image rows in the tensor

  // ---- DMA plan (the same for every stage): slot S = (wave + 8 k) * 64 + lane of the dy / x slab -> byte offset from the stage's first dy row / first x slab row
  //      (slots that nothing reads -- the pad pixel of an odd row pitch, rows beyond the slab -- fetch offset 0) ----
  unsigned dpl[G::DPL], xpl[G::XPL];
  auto dslot = [&](const int k, int& rr, bool& used) {
    const int S = (wave + 8 * k) * 64 + lane, pix = S >> 3, cpos = S & 7;
    rr = pix / RWD; const int xx = pix - rr * RWD, c = cpos ^ (4 * ((rr >> 1) & 1));
    used = rr < ROWS && xx < W;
    return (unsigned)(((rr * W + xx) * C + cog * 64 + c * 8) * 2);
  };
  auto xslot = [&](const int k, int& rr, bool& used) {
    const int S = (wave + 8 * k) * 64 + lane, pix = S >> 4, cpos = S & 15;
    rr = pix / RWX; const int xx = pix - rr * RWX, c = cpos ^ (4 * (rr & 3));
    used = rr < ROWS + 2;
    return (unsigned)(((rr * W + xx) * C + cig * 128 + c * 8) * 2);      // slab row rr = image row (stage row - 1 + rr)
  };
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) { int rr; bool u; const unsigned o = dslot(k, rr, u); dpl[k] = u ? o : 0u; }
#pragma unroll
  for (int k = 0; k < G::XPL; ++k) { int rr; bool u; const unsigned o = xslot(k, rr, u); xpl[k] = u ? o : 0u; }
  auto load_stage = [&](const int q, const int par) {
    const int rs = q * ROWS;
    const unsigned d0 = lds0 + par * G::STAGE, x0 = d0 + G::DYI * 1024;
    const char* dyb = (const char*)(a.dy + (long long)rs * W * C); const char* xb = (const char*)(a.x + (long long)(rs - 1) * W * C);
    if (rs >= 1 && rs + ROWS + 1 <= NR) {                                         // every row of both slabs is inside the tensor (all stages but the first and the last)
#pragma unroll
      for (int k = 0; k < G::DPL; ++k) { if (wave + 8 * k >= G::DYI) break; wp_glds16_s(dpl[k], dyb, d0 + (wave + 8 * k) * 1024); }      // (wave-uniform)
#pragma unroll
      for (int k = 0; k < G::XPL; ++k) { if (wave + 8 * k >= G::XI) break; wp_glds16_s(xpl[k], xb, x0 + (wave + 8 * k) * 1024); }
    } else {                                                                      // rows outside the tensor come from the zero page
#pragma unroll
      for (int k = 0; k < G::DPL; ++k) {
        if (wave + 8 * k >= G::DYI) break;
        int rr; bool u; const unsigned o = dslot(k, rr, u);
        wp_glds16((u && rs + rr < NR) ? (const void*)(dyb + o) : (const void*)wp_zero16, d0 + (wave + 8 * k) * 1024);
      }
#pragma unroll
      for (int k = 0; k < G::XPL; ++k) {
        if (wave + 8 * k >= G::XI) break;
        int rr; bool u; const unsigned o = xslot(k, rr, u); const int r = rs - 1 + rr;
        wp_glds16((u && r >= 0 && r < NR) ? (const void*)(xb + o) : (const void*)wp_zero16, x0 + (wave + 8 * k) * 1024);
      }
    }
  };

  // ---- fragment addresses (stage 0, K-step 0, column 0): transposed reads, lane (16-lane group g4, t) fetches 8 bytes of k-row 8 (g4 >> 1) + 4 h + (t >> 2) ----
  const int cot = wave & 1, ciq = wave >> 1;
  const int g4 = lane >> 4, t = lane & 15;
  unsigned aA[2], aB[3][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int krow = 8 * (g4 >> 1) + 4 * h + (t >> 2);
    aA[h] = lds0 + krow * RWD * 128 + (((2 * cot + (g4 & 1)) ^ (2 * ((krow >> 1) & 1))) << 5) + (t & 3) * 8;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int rr = krow + kh;
      aB[kh][h] = lds0 + G::DYI * 1024 + rr * RWX * 256 + (((2 * ciq + (g4 & 1)) ^ (2 * (rr & 3))) << 5) + (t & 3) * 8;
    }
  }
  wp_f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  int q = s_begin;
  int par = 0;
  load_stage(q, 0);
  for (; q < s_end; ++q, par ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // this stage has landed for every wave / every wave is done with the other one
    asm volatile("" ::: "memory");
    WpRegs R; unsigned m0[4], m2[4];
    WpIt<G, 0>::issue0(R, aA, aB); WpIt<G, 0>::template issueB<1>(R, aB); WpIt<G, 0>::template issueB<2>(R, aB);
    if (q + 1 < s_end && !(WP_ABL & 1)) load_stage(q + 1, par ^ 1);
    const int oyb = (int)(((unsigned)q * (unsigned)ROWS) % (unsigned)H);       // image row (inside its image) of the stage's first row (wave-uniform)
    wp_iter<G, 0>(R, acc, m0, m2, oyb, lane >= 32, aA, aB);
    const unsigned dlt = par ? (unsigned)-G::STAGE : (unsigned)G::STAGE;      // the other stage of the ring
#pragma unroll
    for (int h = 0; h < 2; ++h) { aA[h] += dlt; aB[0][h] += dlt; aB[1][h] += dlt; aB[2][h] += dlt; }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cog * 64 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = cig * 128 + ciq * 32 + (lane & 31);
      if (WP_ABL & 4) { if (acc[j][r] == 1234.5f) a.dw[(long long)co * 9 * C + j * C + ci] = acc[j][r]; } else
      atomicAdd(a.dw + (long long)co * 9 * C + j * C + ci, acc[j][r]);
    }
  __builtin_amdgcn_s_barrier();                      // (the next segment of this workgroup starts with a DMA into stage 0 of the ring)
}

typedef WpGeom<11, 11, 1> WpG11;
typedef WpGeom<6, 6, 2> WpG6;
typedef WpGeom<3, 3, 4> WpG3;
static constexpr size_t wp_max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
static constexpr size_t WP_LDS = wp_max3(2 * WpG11::STAGE, 2 * WpG6::STAGE, 2 * WpG3::STAGE);
static_assert(WP_LDS <= 160 * 1024, "two stages must fit the LDS");

// Several layers per launch, the work shared out EVENLY: the (layer, K-range, kind, stage) quadruples form one sequence weighted by the cost of a stage; a workgroup takes
// one of gridDim.x equal shares -- a contiguous run of stages that may cross kind / range / layer boundaries (accumulators flushed by fp32 atomics at each boundary).
// The first version gave every layer a whole number of workgroups per kind: the 32 kinds of the 512-channel layers could only get 32 or 64 workgroups, the most
// loaded workgroup had 28 % more than the average.  Order and placement serve the L2: a layer's reduction is cut into R ranges with all kinds of a range next to
// each other in the sequence, and consecutive shares go to workgroups of ONE XCD (ids x, x + 8, x + 16, ...), which run at the same time: the kinds that read the
// same dy / x rows fetch them once per XCD (with kind-major order every kind streamed its rows from memory on its own: 174 us of DMA waits in 907 us).
struct WpGroup { WpArgs it[AVEC_WGRAD_GROUP_MAX]; long long cum[AVEC_WGRAD_GROUP_MAX + 1]; int stages[AVEC_WGRAD_GROUP_MAX], wt[AVEC_WGRAD_GROUP_MAX], ranges[AVEC_WGRAD_GROUP_MAX]; int n; };
__global__ __launch_bounds__(512) void wgrad3x3_pairs_grouped_kernel(WpGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const long long T = grp.cum[grp.n];
  const int nb = (int)gridDim.x, per = nb >> 3;
  const int slot = (nb & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);      // XCD x = workgroup ids x + 8 j owns the shares [x per, (x + 1) per)
  const long long lo = T * (long long)slot / nb, hi = T * (long long)(slot + 1) / nb;
  for (int i = 0; i < grp.n; ++i) {
    const long long c0 = grp.cum[i], c1 = grp.cum[i + 1];
    if (hi <= c0 || lo >= c1) continue;
    const WpArgs& a = grp.it[i];
    const int nst = grp.stages[i], wt = grp.wt[i], R = grp.ranges[i], kinds = (a.C >> 6) * (a.C >> 7);
    // item-local unit index u in [0, kinds nst): range r = stages [r nst / R, (r + 1) nst / R), inside a range kind-major
    long long u = ((lo > c0 ? lo : c0) - c0) / wt; const long long u1 = ((hi < c1 ? hi : c1) - c0) / wt;
    while (u < u1) {
      int r = (int)((u / kinds) * R / nst); if (r >= R) r = R - 1;
      while (r + 1 < R && (long long)kinds * (int)((long long)(r + 1) * nst / R) <= u) ++r;
      while (r > 0 && (long long)kinds * (int)((long long)r * nst / R) > u) --r;
      const int rs0 = (int)((long long)r * nst / R), rs1 = (int)((long long)(r + 1) * nst / R), len = rs1 - rs0;
      const long long v = u - (long long)kinds * rs0;
      const int k = (int)(v / len), sb = rs0 + (int)(v % len);
      long long take = rs1 - sb; if (take > u1 - u) take = u1 - u;
      const int se = sb + (int)take;
      if (a.W == 11) wp_body<WpG11>(a, smem, k, sb, se);
      else if (a.W == 6) wp_body<WpG6>(a, smem, k, sb, se);
      else wp_body<WpG3>(a, smem, k, sb, se);
      u += take;
    }
  }
}

bool wgrad3x3_pairs_supported(int H, int W, int C) {
  static const bool off = getenv("AVEC_NO_WGRAD_PAIRS") != nullptr;
  return !off && C >= 128 && C % 128 == 0 && C <= 1024 && ((H == 11 && W == 11) || (H == 6 && W == 6) || (H == 3 && W == 3));
}

// items: every one wgrad3x3_pairs_supported.  One workgroup per CU (the ring fills the LDS)
template <typename G> static constexpr int wp_stage_weight() {      // cycles of a CU per stage / 64: MFMAs (two waves per SIMD) + transposed reads (the two do not overlap) + barrier / DMA issue
  return (G::KS * (G::PAIRS * 64 + 8 * G::W * 32) + 512) / 64;
}
int wgrad3x3_pairs_grouped(const avec_wgrad3x3_item_t* items, int n, hipStream_t st) {
  WpGroup g; g.n = n; g.cum[0] = 0;
  long long total_stages = 0;
  for (int i = 0; i < n; ++i) {
    const avec_wgrad3x3_item_t& t = items[i];
    WpArgs& a = g.it[i]; a.x = (const bf16*)t.x; a.dy = (const bf16*)t.dy; a.dw = t.dw; a.N = (int)t.images; a.C = t.C; a.H = t.H; a.W = t.W;
    AVEC_CHECK_ARG(t.images * t.H < (1ll << 26) && 70ll * t.W * t.C * 2 < (1ll << 31), "wgrad3x3_pairs: item %d too large", i);
    const int rows = t.W == 11 ? WpG11::ROWS : t.W == 6 ? WpG6::ROWS : WpG3::ROWS;
    g.wt[i] = t.W == 11 ? wp_stage_weight<WpG11>() : t.W == 6 ? wp_stage_weight<WpG6>() : wp_stage_weight<WpG3>();
    g.stages[i] = (int)((t.images * t.H + rows - 1) / rows);
    const int kinds = (t.C / 64) * (t.C / 128);
    g.cum[i + 1] = g.cum[i] + (long long)kinds * g.stages[i] * g.wt[i];
    total_stages += (long long)kinds * g.stages[i];
  }
  static const int wgs_env = 256;
  const int grid = (int)(total_stages < wgs_env ? total_stages : wgs_env);
  static const int ranges_env = 0;      // A/B: 1 = kind-major order (one range per layer)
  for (int i = 0; i < n; ++i) {      // ranges: one share ~ one kind over one range
    const int kinds = (items[i].C / 64) * (items[i].C / 128);
    const double shares = (double)grid * (double)(g.cum[i + 1] - g.cum[i]) / (double)g.cum[n];
    int R = (int)(shares / kinds + 0.5); if (R < 1) R = 1; if (R > g.stages[i]) R = g.stages[i];
    g.ranges[i] = ranges_env > 0 ? (ranges_env < g.stages[i] ? ranges_env : g.stages[i]) : R;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_pairs_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WP_LDS);
    if (e != hipSuccess) { avec_set_error("wgrad3x3_pairs: cannot reserve %zu bytes of LDS: %s", (size_t)WP_LDS, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  avec_note_kernel("wgrad3x3_pairs_grouped_kernel");
  hipLaunchKernelGGL(wgrad3x3_pairs_grouped_kernel, dim3((unsigned)grid), dim3(512), WP_LDS, st, g);
  AVEC_LAUNCH_CHECK();
  return 0;
}

#ifdef WP_PROBE
extern __shared__ __attribute__((aligned(16))) char wp_probe_smem[];
__global__ __launch_bounds__(512) void wp_probe11(WpArgs a) { wp_body<WpG11>(a, wp_probe_smem, blockIdx.x, 0, gridDim.x); }
__global__ __launch_bounds__(512) void wp_probe6(WpArgs a) { wp_body<WpG6>(a, wp_probe_smem, blockIdx.x, 0, gridDim.x); }
__global__ __launch_bounds__(512) void wp_probe3(WpArgs a) { wp_body<WpG3>(a, wp_probe_smem, blockIdx.x, 0, gridDim.x); }
#endif
