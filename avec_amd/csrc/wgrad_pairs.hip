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

// bid / nwg: this workgroup's index among the nwg workgroups that share the layer
template <typename G>
__device__ __forceinline__ void wp_body(const WpArgs& a, const int bid, const int nwg) {
  constexpr int H = G::H, W = G::W, KS = G::KS, RWD = G::RWD, RWX = G::RWX, ROWS = G::ROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = a.C;
  const int nci = C >> 7, kinds = (C >> 6) * nci, kind = bid % kinds, cog = kind / nci, cig = kind - cog * nci;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const long long NR = (long long)a.N * H;                 // image rows in the tensor
  const long long nstages = (NR + ROWS - 1) / ROWS;
  const int wgs = nwg / kinds;                             // workgroups per kind

  // ---- DMA plan (the same for every stage): slot S = (wave + 8 k) * 64 + lane of the dy / x slab -> (slab row << 24) | element offset from the stage's first row ----
  int dpl[G::DPL], xpl[G::XPL];
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) {
    const int S = (wave + 8 * k) * 64 + lane, pix = S >> 3, cpos = S & 7;
    const int rr = pix / RWD, xx = pix - rr * RWD, c = cpos ^ (4 * ((rr >> 1) & 1));
    dpl[k] = (wave + 8 * k < G::DYI && rr < ROWS && xx < W) ? (rr << 24) | ((rr * W + xx) * C + cog * 64 + c * 8) : -1;
  }
#pragma unroll
  for (int k = 0; k < G::XPL; ++k) {
    const int S = (wave + 8 * k) * 64 + lane, pix = S >> 4, cpos = S & 15;
    const int rr = pix / RWX, xx = pix - rr * RWX, c = cpos ^ (4 * (rr & 3));
    xpl[k] = (wave + 8 * k < G::XI && rr < ROWS + 2) ? (rr << 24) | ((rr * W + xx) * C + cig * 128 + c * 8) : -1;      // slab row rr = image row (stage row - 1 + rr)
  }
  auto load_stage = [&](const long long q, const int par) {
    const long long rs = q * ROWS;
    const unsigned d0 = lds0 + par * G::STAGE, x0 = d0 + G::DYI * 1024;
    const bf16* dyb = a.dy + rs * W * C; const bf16* xb = a.x + (rs - 1) * W * C;
#pragma unroll
    for (int k = 0; k < G::DPL; ++k) {
      if (wave + 8 * k >= G::DYI) break;                                        // wave-uniform
      const int e = dpl[k];
      const void* src = (e >= 0 && rs + (e >> 24) < NR) ? (const void*)(dyb + (e & 0xffffff)) : (const void*)wp_zero16;
      wp_glds16(src, d0 + (wave + 8 * k) * 1024);
    }
#pragma unroll
    for (int k = 0; k < G::XPL; ++k) {
      if (wave + 8 * k >= G::XI) break;
      const int e = xpl[k]; const long long r = rs - 1 + (e >> 24);
      const void* src = (e >= 0 && r >= 0 && r < NR) ? (const void*)(xb + (e & 0xffffff)) : (const void*)wp_zero16;
      wp_glds16(src, x0 + (wave + 8 * k) * 1024);
    }
  };

  // ---- fragment addresses (stage 0, K-step 0, column 0): transposed reads, lane (16-lane group g4, t) fetches 8 bytes of k-row 8 (g4 >> 1) + 4 h + (t >> 2) ----
  const int cot = wave & 1, ciq = wave >> 1;
  const int g4 = lane >> 4, t = lane & 15;
  int adA[2], adB[3][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int krow = 8 * (g4 >> 1) + 4 * h + (t >> 2);
    adA[h] = krow * RWD * 128 + (((2 * cot + (g4 & 1)) ^ (2 * ((krow >> 1) & 1))) << 5) + (t & 3) * 8;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int rr = krow + kh;
      adB[kh][h] = G::DYI * 1024 + rr * RWX * 256 + (((2 * ciq + (g4 & 1)) ^ (2 * (rr & 3))) << 5) + (t & 3) * 8;
    }
  }
  wp_f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  long long q = bid / kinds;
  int par = 0;
  if (q < nstages) load_stage(q, 0);
  for (; q < nstages; q += wgs, par ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // this stage has landed for every wave / every wave is done with the other one
    asm volatile("" ::: "memory");
    if (q + wgs < nstages) load_stage(q + wgs, par ^ 1);
    const char* sb = smem + par * G::STAGE;
    const int oyb = (int)((q * ROWS + 8 * (lane >> 5)) % H);       // image row (inside its image) of this lane's first k-element of K-step 0
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // k-element e of this lane is image row oy = (oyb + 16 s + e) % H of its image: kh = 0 pairs it with row oy - 1, kh = 2 with row oy + 1
      unsigned m0[4], m2[4];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        const int o0 = (oyb + 16 * s + 2 * dd) % H, o1 = (oyb + 16 * s + 2 * dd + 1) % H;
        m0[dd] = (o0 == 0 ? 0u : 0xffffu) | (o1 == 0 ? 0u : 0xffff0000u);
        m2[dd] = (o0 == H - 1 ? 0u : 0xffffu) | (o1 == H - 1 ? 0u : 0xffff0000u);
      }
      auto ldA = [&](const int ox) { WpFrag f; f.lo = wp_tr(sb + adA[0] + (16 * s * RWD + ox) * 128); f.hi = wp_tr(sb + adA[1] + (16 * s * RWD + ox) * 128); return f; };
      auto ldB = [&](const int xp, const int kh) { WpFrag f; f.lo = wp_tr(sb + adB[kh][0] + (16 * s * RWX + xp) * 256); f.hi = wp_tr(sb + adB[kh][1] + (16 * s * RWX + xp) * 256); return f; };
      WpFrag Ap, Ac = ldA(0), An;
      Ap = Ac;
#pragma unroll
      for (int xp = 0; xp < W; ++xp) {
        if (xp + 1 < W) An = ldA(xp + 1);
        WpFrag B0 = ldB(xp, 0), B1 = ldB(xp, 1), B2 = ldB(xp, 2);
        B0.lo.x &= m0[0]; B0.lo.y &= m0[1]; B0.hi.x &= m0[2]; B0.hi.y &= m0[3];
        B2.lo.x &= m2[0]; B2.lo.y &= m2[1]; B2.hi.x &= m2[2]; B2.hi.y &= m2[3];
        // tap (kh, kw) pairs x column xp with dy column ox = xp + 1 - kw
        if (xp + 1 < W) { acc[0] = wp_mma(An, B0, acc[0]); acc[3] = wp_mma(An, B1, acc[3]); acc[6] = wp_mma(An, B2, acc[6]); }
        acc[1] = wp_mma(Ac, B0, acc[1]); acc[4] = wp_mma(Ac, B1, acc[4]); acc[7] = wp_mma(Ac, B2, acc[7]);
        if (xp >= 1) { acc[2] = wp_mma(Ap, B0, acc[2]); acc[5] = wp_mma(Ap, B1, acc[5]); acc[8] = wp_mma(Ap, B2, acc[8]); }
        Ap = Ac; Ac = An;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cog * 64 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = cig * 128 + ciq * 32 + (lane & 31);
      atomicAdd(a.dw + (long long)co * 9 * C + j * C + ci, acc[j][r]);
    }
}

typedef WpGeom<11, 11, 1> WpG11;
typedef WpGeom<6, 6, 2> WpG6;
typedef WpGeom<3, 3, 4> WpG3;
static constexpr size_t wp_max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
static constexpr size_t WP_LDS = wp_max3(2 * WpG11::STAGE, 2 * WpG6::STAGE, 2 * WpG3::STAGE);
static_assert(WP_LDS <= 160 * 1024, "two stages must fit the LDS");

struct WpGroup { WpArgs it[AVEC_WGRAD_GROUP_MAX]; int first[AVEC_WGRAD_GROUP_MAX + 1]; int n; };
__global__ __launch_bounds__(512) void wgrad3x3_pairs_grouped_kernel(WpGroup grp) {
  int i = 0;
  while (i + 1 < grp.n && (int)blockIdx.x >= grp.first[i + 1]) ++i;
  const WpArgs& a = grp.it[i];
  const int bid = (int)blockIdx.x - grp.first[i], nwg = grp.first[i + 1] - grp.first[i];
  if (a.W == 11) wp_body<WpG11>(a, bid, nwg);
  else if (a.W == 6) wp_body<WpG6>(a, bid, nwg);
  else wp_body<WpG3>(a, bid, nwg);
}

bool wgrad3x3_pairs_supported(int H, int W, int C) {
  static const bool off = getenv("AVEC_NO_WGRAD_PAIRS") != nullptr;
  return !off && C >= 128 && C % 128 == 0 && C <= 1024 && ((H == 11 && W == 11) || (H == 6 && W == 6) || (H == 3 && W == 3));
}

// items: every one wgrad3x3_pairs_supported.  One workgroup per CU (the ring fills the LDS): 256 workgroups shared out by work, a whole set of kinds at a time
int wgrad3x3_pairs_grouped(const avec_wgrad3x3_item_t* items, int n, hipStream_t st) {
  WpGroup g; g.n = n;
  long long cost[AVEC_WGRAD_GROUP_MAX], stages[AVEC_WGRAD_GROUP_MAX]; int kinds[AVEC_WGRAD_GROUP_MAX], nwg[AVEC_WGRAD_GROUP_MAX], total = 0;
  for (int i = 0; i < n; ++i) {
    const avec_wgrad3x3_item_t& t = items[i];
    WpArgs& a = g.it[i]; a.x = (const bf16*)t.x; a.dy = (const bf16*)t.dy; a.dw = t.dw; a.N = (int)t.images; a.C = t.C; a.H = t.H; a.W = t.W;
    AVEC_CHECK_ARG((long long)t.images * t.H * t.W * t.C < (1ll << 40) && 70ll * t.W * t.C < (1 << 24), "wgrad3x3_pairs: item %d too large", i);
    const int rows = t.W == 11 ? WpG11::ROWS : t.W == 6 ? WpG6::ROWS : WpG3::ROWS, pairs = t.W == 11 ? WpG11::PAIRS * WpG11::KS : t.W == 6 ? WpG6::PAIRS * WpG6::KS : WpG3::PAIRS * WpG3::KS;
    kinds[i] = (t.C / 64) * (t.C / 128);
    stages[i] = (t.images * t.H + rows - 1) / rows;
    cost[i] = stages[i] * (long long)(pairs + 12);                          // MFMAs of one kind's reduction (+ the per-stage barrier / DMA issue)
    nwg[i] = kinds[i]; total += kinds[i];
  }
  AVEC_CHECK_ARG(total <= 4096, "wgrad3x3_pairs: too many tiles");
  const int budget = 256;
  for (;;) {
    int best = -1; double worst = 0.0;
    for (int i = 0; i < n; ++i) {
      if (total + kinds[i] > budget || nwg[i] / kinds[i] >= stages[i]) continue;
      const double load = (double)cost[i] / (double)(nwg[i] / kinds[i]);
      if (load > worst) { worst = load; best = i; }
    }
    if (best < 0) break;
    nwg[best] += kinds[best]; total += kinds[best];
  }
  g.first[0] = 0; for (int i = 0; i < n; ++i) g.first[i + 1] = g.first[i] + nwg[i];
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_pairs_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WP_LDS);
    if (e != hipSuccess) { avec_set_error("wgrad3x3_pairs: cannot reserve %zu bytes of LDS: %s", (size_t)WP_LDS, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  avec_note_kernel("wgrad3x3_pairs_grouped_kernel");
  hipLaunchKernelGGL(wgrad3x3_pairs_grouped_kernel, dim3((unsigned)total), dim3(512), WP_LDS, st, g);
  AVEC_LAUNCH_CHECK();
  return 0;
}
