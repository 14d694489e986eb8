// 3x3 / stride-2 / pad-1 convolution (forward and backward-data), bf16, as a shifted-window implicit GEMM over the four PARITY CLASSES of the fine
// (H x W) grid -- the stage boundaries of the ResNet-18 front-end (reference: nnet/blocks.py:64-82, nnet/networks.py:32-146).
//
// Index space: "coarse pixels" p = (img, i, j), i < OH, j < OW (the forward output grid).  Fine pixel (2 i + pr, 2 j + pc) is element p of class plane (pr, pc).
//   forward      y[p] = sum_taps plane_(pr,pc)(x)[p + s] . W[tap],  (pr, pc) = ((kh + 1) & 1, (kw + 1) & 1),  s = dh * OW + dw,  dh = -(kh == 0), dw = -(kw == 0)
//                -> per 32-channel chunk FOUR windows (one per class, gathered from the NHWC tensor by the LDS-DMA: consecutive coarse pixels are two fine pixels
//                apart) of BM + 16 rows serve 4 + 1 + 2 + 2 taps; every window crosses L2 -> LDS once per chunk instead of once per tap.
//   backward-data  dx[plane_(pr,pc)][p] = sum over the taps of that class of dy[p + s] . W[tap]^T,  s = a * OW + b,  a = (kh == 0), b = (kw == 0)
//                -> a workgroup owns BM coarse pixels of ONE class (blockIdx.z); its window is BM + 16 consecutive rows of dy, reused by the class's 4 / 2 / 2 / 1 taps;
//                results are scattered to the fine grid by the register-direct epilogue (row map), the projection shortcut's gradient (res_cls0) is added to class 0.
// Loop, LDS image (64-byte rows, XOR swizzle on the source side), transposed product and epilogue are those of conv3x3_shift_kernel (gemm.hip).  The window ring has three
// buffers (the window two groups ahead is requested at the first tap of a group), the weight ring three stages; both positions are run-time (scalar) offsets.
// Taps that leave the image are zeroed on the A fragments (4 v_cndmask per fragment: no zero blocks in LDS, 75 KB per workgroup = two workgroups per CU).
#include "gemm_dev.h"

#include <stdint.h>
#include <stdlib.h>

namespace {

constexpr int S2_HALO = 16;

// LDS-DMA of one 16-byte piece per lane (lanes outside `on` idle): wave-uniform LDS base through M0
__device__ __forceinline__ void s2_glds16(const void* sbase, unsigned voff, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst_uniform) : "memory");
}

// MODE_CONV_FWD: CLS unused.  MODE_CONV_BWD: CLS = pr * 2 + pc of the fine pixels this workgroup produces.
template <int BM, int BN, int MODE, int CLS>
__device__ __forceinline__ void conv_s2_body(const GemmArgs& g, char* smem) {
  constexpr bool FWD = MODE == MODE_CONV_FWD;
  constexpr int RB = 64, KE = 32;
  constexpr int NCB = BN / 64;                                // DMA passes (64 rows x 64 B) per weight tile
  constexpr int WR = BM + S2_HALO, NAF = BM / 64;             // window rows; full DMA passes (+ one 16-row piece: lanes 0-15 of every wave)
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int BTILE = BN * RB, AWIN = WR * RB;
  constexpr int PR = CLS >> 1, PC = CLS & 1;
  // taps of a group, in step order
  constexpr int NTAPS = FWD ? 9 : (CLS == 3 ? 4 : (CLS == 0 ? 1 : 2));
  char* const Bring = smem; char* const Awin = smem + 3 * BTILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const long long m0 = (long long)blockIdx.x * BM; const int n0 = blockIdx.y * BN;
  const int H = g.a.H, Wd = g.a.W, OH = g.a.OH, OW = g.a.OW, C = g.a.C;
  const long long P = FWD ? g.M : g.M / ((long long)H * Wd) * OH * OW;       // coarse pixels
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned bring0 = (unsigned)(uintptr_t)(lptr_t)Bring, awin0 = (unsigned)(uintptr_t)(lptr_t)Awin;
  const unsigned wslot = (unsigned)wave * 1024u;

  // ---- DMA plan.  Window row wr <-> coarse pixel pw = m0 - HALO + wr (forward) / m0 + wr (backward), clamped into [0, P).  Pass i: row 64 i + tid / 4, physical slot tid % 4
  // carrying logical chunk slot ^ swz(row); the 16-row piece: row 64 NAF + (16 wave + lane) / 4, slot lane % 4.
  unsigned aoff[NAF + 1];
  const unsigned amax = (unsigned)(((FWD ? (P / ((long long)OH * OW)) * H * Wd : P) * (long long)C - 8) * 2);       // last 16-byte piece of the source tensor
  auto plan = [&](const int wr, const int slot) -> unsigned {
    long long pw = (FWD ? m0 - S2_HALO : m0) + wr; pw = pw < 0 ? 0 : (pw >= P ? P - 1 : pw);
    const unsigned kc = (unsigned)((slot ^ glds_swz<RB>(wr)) * 8);
    if (FWD) {
      const int j = (int)(pw % OW); const long long t = pw / OW; const int i = (int)(t % OH); const long long img = t / OH;
      return (unsigned)((((img * H + 2 * i) * Wd + 2 * j) * C + kc) * 2);        // class (0,0) pixel; class (pr,pc) adds (pr W + pc) C elements (clamped at issue)
    }
    return (unsigned)((pw * C + kc) * 2);
  };
#pragma unroll
  for (int i = 0; i < NAF; ++i) aoff[i] = plan(i * 64 + (tid >> 2), tid & 3);
  aoff[NAF] = plan(NAF * 64 + ((wave * 16 + (lane & 15)) >> 2), lane & 3);
  unsigned boff[NCB];
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int row = i * 64 + (tid >> 2); const int n = n0 + row < g.N ? n0 + row : g.N - 1;
    boff[i] = (unsigned)(((long long)n * g.ldw + (((tid & 3) ^ glds_swz<RB>(row)) * 8)) * 2);
  }
  const int C2 = C * 2;

  // ---- tap validity per fragment row (bit t = tap t may be used): forward top / left (the shifted element belongs to the previous row / image) and, for odd sizes,
  // bottom / right (the class plane has no such pixel); backward the taps that read dy beyond the last output row / column
  unsigned amask[MT]; long long prow[MT]; bool pvalid[MT];
#pragma unroll
  for (int f = 0; f < MT; ++f) {
    const long long p = m0 + wm * (BM / 2) + f * 32 + (lane & 31);
    const long long pc_ = p < P ? p : P - 1;
    const int j = (int)(pc_ % OW); const long long t = pc_ / OW; const int i = (int)(t % OH); const long long img = t / OH;
    unsigned mk = 0u;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        bool ok;
        if (FWD) { const int y = 2 * i + kh - 1, x = 2 * j + kw - 1; ok = y >= 0 && x >= 0 && y < H && x < Wd; }
        else { const int a = kh == 0 ? 1 : 0, b = kw == 0 ? 1 : 0; ok = i + a < OH && j + b < OW; }
        mk |= (ok ? 1u : 0u) << (kh * 3 + kw);
      }
    amask[f] = mk;
    if (FWD) { prow[f] = pc_; pvalid[f] = p < P; }
    else { const int y = 2 * i + PR, x = 2 * j + PC; pvalid[f] = p < P && y < H && x < Wd; prow[f] = (img * H + (y < H ? y : H - 1)) * (long long)Wd + (x < Wd ? x : Wd - 1); }
  }

  // ---- fragment addresses inside window buffer 0 / ring stage 0: A per (tap, K-substep), fragment f adds 2048 f; B per K-substep, fragment j adds 2048 j
  const int gsel = lane >> 5;
  unsigned aad[9][2], bad[2];
  { const int arow0 = wm * (BM / 2) + (lane & 31) + (FWD ? S2_HALO : 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int kh = t / 3, kw = t % 3;
      const int s = FWD ? -((kh == 0 ? OW : 0) + (kw == 0 ? 1 : 0)) : ((kh == 0 ? OW : 0) + (kw == 0 ? 1 : 0));
      const int w0 = arow0 + s;
#pragma unroll
      for (int q = 0; q < 2; ++q) aad[t][q] = awin0 + (unsigned)w0 * RB + ((((unsigned)(q * 2 + gsel)) ^ ((unsigned)(w0 >> 2) & 3u)) << 4);
    }
    const int row = wn * (BN / 2) + (lane & 31);
#pragma unroll
    for (int q = 0; q < 2; ++q) bad[q] = bring0 + (unsigned)(row * RB + (((q * 2 + gsel) ^ glds_swz<RB>(row)) << 4)); }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int NC = C / KE;
  const char* const Ab = (const char*)g.a.ptr; const char* const Wb = (const char*)g.W;

  // ---- the step list of one chunk (compile-time): tap, first-of-group flag, class of the group's window (forward)
  struct Plan {
    static constexpr int tap(int k) {
      if (FWD) { constexpr int o[9] = {0, 2, 6, 8, 4, 1, 7, 3, 5}; return o[k]; }
      if (CLS == 3) { constexpr int o[4] = {0, 2, 6, 8}; return o[k]; }
      if (CLS == 2) { constexpr int o[2] = {1, 7}; return o[k]; }
      if (CLS == 1) { constexpr int o[2] = {3, 5}; return o[k]; }
      return 4;
    }
    static constexpr bool first(int k) { return FWD ? (k == 0 || k == 4 || k == 5 || k == 7) : k == 0; }
    static constexpr int ngroups() { return FWD ? 4 : 1; }
    static constexpr int group(int k) { return FWD ? (k < 4 ? 0 : k < 5 ? 1 : k < 7 ? 2 : 3) : 0; }
    static constexpr int gclass(int gi) { return FWD ? (gi == 0 ? 3 : gi == 1 ? 0 : gi == 2 ? 2 : 1) : CLS; }      // window class of group gi: (1,1), (0,0), (1,0), (0,1)
  };
  constexpr int NG = Plan::ngroups();
  const int total_groups = NC * NG, total_steps = NC * NTAPS;

  // window of global group gq -> buffer gq % 3 (passed in as a byte offset)
  auto issue_window = [&](const int gq, const unsigned bufoff) {
    const int cc = gq / NG, gi = gq - cc * NG;
    unsigned delta = 0u;
    if (FWD) { const int cl = gi == 0 ? 3 : gi == 1 ? 0 : gi == 2 ? 2 : 1; delta = (unsigned)(((cl >> 1) * Wd + (cl & 1)) * C2); }
    const char* const src = Ab + (long long)cc * (KE * 2);
    unsigned v[NAF];
#pragma unroll
    for (int i = 0; i < NAF; ++i) { const unsigned o = aoff[i] + delta; v[i] = o < amax ? o : amax; }
    glds16_group<NAF>(v, src, awin0 + bufoff + wslot);
    if (lane < 16) { const unsigned o = aoff[NAF] + delta; s2_glds16(src, o < amax ? o : amax, awin0 + bufoff + NAF * 4096u + (unsigned)wave * 256u); }
  };
  // weight tile of global step sq -> ring stage sq % 3
  auto issue_b = [&](const int cc, const int tap, const unsigned stageoff) {
    glds16_group<NCB>(boff, Wb + (long long)tap * C2 + (long long)cc * (KE * 2), bring0 + stageoff + wslot);
  };

#define S2_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  // prologue: W(0), W(1), B(0), B(1)
  issue_window(0, 0u);
  if (total_groups > 1) issue_window(1, (unsigned)AWIN);
  issue_b(0, Plan::tap(0), 0u);
  if (total_steps > 1) { if (NTAPS > 1) issue_b(0, Plan::tap(1 % NTAPS), (unsigned)BTILE); else issue_b(1, Plan::tap(0), (unsigned)BTILE); }

  unsigned wbuf = 0u;          // byte offset of the current group's window buffer
  unsigned bst = 0u;           // byte offset of the current step's weight stage
  int gq = 0, sq = 0;          // global group / step counters
  bool prev_first_issued = false;      // did the previous step request a window?  (it is newer than this step's weight tile)

  auto step = [&](auto kc, const int cc) {
    constexpr int K_ = decltype(kc)::value;
    constexpr int TAP = Plan::tap(K_);
    constexpr bool FIRST = Plan::first(K_);
    // loads newer than B(sq): B(sq + 1) (if it exists) and the window the previous step requested (if any)
    const bool has_next_b = sq + 1 < total_steps;
    if (prev_first_issued) { if (has_next_b) S2_WAIT_VM(NCB + NAF + 1); else S2_WAIT_VM(NAF + 1); }
    else { if (has_next_b) S2_WAIT_VM(NCB); else S2_WAIT_VM(0); }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    u32x4 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned ar = aad[TAP][q] + wbuf, br = bad[q] + bst;
      fa[q][0] = lds_read128o<0>(ar); fa[q][1] = lds_read128o<2048>(ar);
      if (MT > 2) { fa[q][2 % MT] = lds_read128o<4096>(ar); fa[q][3 % MT] = lds_read128o<6144>(ar); }
      fb[q][0] = lds_read128o<0>(br);
      if (NT > 1) fb[q][1 % NT] = lds_read128o<2048>(br);
    }
    // requests two groups / two steps ahead go out while the fragment reads are in flight (the buffers they overwrite were last read before this barrier)
    bool issued_w = false;
    if (FIRST) {
      if (gq + 2 < total_groups) { unsigned nb = wbuf + 2u * AWIN; nb = nb >= 3u * AWIN ? nb - 3u * AWIN : nb; issue_window(gq + 2, nb); issued_w = true; }
    }
    if (sq + 2 < total_steps) {
      constexpr int K2 = (K_ + 2) % NTAPS; const int cc2 = cc + (K_ + 2 >= NTAPS ? ((K_ + 2) / NTAPS) : 0);
      unsigned nb = bst + 2u * BTILE; nb = nb >= 3u * BTILE ? nb - 3u * BTILE : nb;
      issue_b(cc2, Plan::tap(K2), nb);
    }
    prev_first_issued = issued_w;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[q][i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[q][j]));
      if (!(FWD && TAP == 4) && !(!FWD && CLS == 0)) {          // (the centre tap never leaves the image)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const bool ok = (amask[i] >> TAP) & 1u;
          fa[q][i][0] = ok ? fa[q][i][0] : 0u; fa[q][i][1] = ok ? fa[q][i][1] : 0u; fa[q][i][2] = ok ? fa[q][i][2] : 0u; fa[q][i][3] = ok ? fa[q][i][3] : 0u;
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[q][j]), __builtin_bit_cast(bf16x8_t, fa[q][i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // advance the rings
    ++sq; bst += BTILE; bst = bst >= 3u * BTILE ? 0u : bst;
    constexpr bool LAST_OF_GROUP = (K_ + 1 == NTAPS) || Plan::first((K_ + 1) % NTAPS);
    if (LAST_OF_GROUP) { ++gq; wbuf += AWIN; wbuf = wbuf >= 3u * AWIN ? 0u : wbuf; }
  };

#pragma unroll 1
  for (int cc = 0; cc < NC; ++cc) {
    step(IntC<0>{}, cc);
    if constexpr (NTAPS > 1) step(IntC<1 % NTAPS>{}, cc);
    if constexpr (NTAPS > 2) { step(IntC<2 % NTAPS>{}, cc); step(IntC<3 % NTAPS>{}, cc); }
    if constexpr (NTAPS > 4) { step(IntC<4 % NTAPS>{}, cc); step(IntC<5 % NTAPS>{}, cc); step(IntC<6 % NTAPS>{}, cc); step(IntC<7 % NTAPS>{}, cc); step(IntC<8 % NTAPS>{}, cc); }
  }
#undef S2_WAIT_VM
  __syncthreads();
  const bool use_res = g.e.res != nullptr && (!g.e.res_cls0 || (!FWD && CLS == 0));
  long long rrow[MT];
#pragma unroll
  for (int f = 0; f < MT; ++f) {
    const long long p = m0 + wm * (BM / 2) + f * 32 + (lane & 31);
    rrow[f] = g.e.res_cls0 ? (p < P ? p : P - 1) : prow[f];
  }
  bool full = m0 + BM <= P;
  if (!FWD && ((H & 1) | (Wd & 1))) full = false;            // odd sizes: some coarse pixels have no fine pixel in this class
  conv_epilogue_tr<BM, BN, MT, NT>(g, acc, smem, prow, pvalid, use_res ? (const bf16*)g.e.res : nullptr, rrow, full, n0, tid, lane, wm, wn);
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_fwd_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_s2_body<BM, BN, MODE_CONV_FWD, 0>(g, smem);
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_bwd_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // heaviest class first: blockIdx.z 0..3 -> classes (1,1), (1,0), (0,1), (0,0)
  const int z = (int)blockIdx.z;
  if (z == 0) conv_s2_body<BM, BN, MODE_CONV_BWD, 3>(g, smem);
  else if (z == 1) conv_s2_body<BM, BN, MODE_CONV_BWD, 2>(g, smem);
  else if (z == 2) conv_s2_body<BM, BN, MODE_CONV_BWD, 1>(g, smem);
  else conv_s2_body<BM, BN, MODE_CONV_BWD, 0>(g, smem);
}

bool s2_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <typename K> int s2_want_lds(K kern, size_t bytes) {
  static const void* done[16]; static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
  if (ndone < 16) done[ndone++] = (const void*)kern;
  return 0;
}

}  // namespace

// host side: 1 = not applicable (the caller continues with the general implicit-GEMM kernels), 0 = launched, other = error
int avec_launch_conv_s2(const GemmArgs& g_in, int mode, hipStream_t st) {
  static const bool off = getenv("AVEC_NO_CONV_S2") != nullptr;
  const RowSrc& a = g_in.a; const Epi& e = g_in.e;
  if (off || mode == MODE_PLAIN || a.KH != 3 || a.KW != 3 || a.stride != 2 || a.pad != 1 || a.OH != (a.H + 1) / 2 || a.OW != (a.W + 1) / 2 || a.OW > 15 || a.OW < 2 || a.C % 32 != 0) return 1;
  if (!s2_aligned16(a.ptr) || !s2_aligned16(g_in.W) || g_in.ldw % 8 != 0 || (long long)g_in.N * g_in.ldw * 2 >= (1ll << 32) || g_in.N < 64 || g_in.N % 8 != 0) return 1;
  // register-direct epilogue only: bf16 output, nothing but alpha / bf16 residual / BatchNorm statistics fused
  if (e.out_f32 || e.out_pre || e.bias || e.act != 0 || e.drop_p > 0.f || e.dact || e.colsum || e.bnb_y || e.ldo % 8 != 0 || !s2_aligned16(e.out)) return 1;
  if (e.res && (!e.res_act || e.ldres % 8 != 0 || !s2_aligned16(e.res))) return 1;
  const long long fine = (long long)a.H * a.W, coarse = (long long)a.OH * a.OW;
  long long imgs, P;
  if (mode == MODE_CONV_FWD) { if (g_in.M % coarse) return 1; imgs = g_in.M / coarse; if (e.res) return 1; }
  else { if (g_in.M % fine) return 1; imgs = g_in.M / fine; }
  P = imgs * coarse;
  if (imgs * fine * a.C * 2 >= (1ll << 32) || P * a.C * 2 >= (1ll << 32) || P < 1) return 1;
  GemmArgs g = g_in; g.perm2 = 0;
#define S2(BM, BN) do { \
    const size_t lds = (size_t)3 * BN * 64 + (size_t)3 * (BM + S2_HALO) * 64; \
    if (mode == MODE_CONV_FWD) { dim3 grid((unsigned)((P + BM - 1) / BM), (unsigned)((g.N + BN - 1) / BN)); \
      avec_note_kernel("conv3x3_s2_fwd_kernel<%d,%d>", BM, BN); if (int r = s2_want_lds(conv3x3_s2_fwd_kernel<BM, BN>, lds)) return r; \
      hipLaunchKernelGGL((conv3x3_s2_fwd_kernel<BM, BN>), grid, dim3(256), lds, st, g); } \
    else { dim3 grid((unsigned)((P + BM - 1) / BM), (unsigned)((g.N + BN - 1) / BN), 4); \
      avec_note_kernel("conv3x3_s2_bwd_kernel<%d,%d>", BM, BN); if (int r = s2_want_lds(conv3x3_s2_bwd_kernel<BM, BN>, lds)) return r; \
      hipLaunchKernelGGL((conv3x3_s2_bwd_kernel<BM, BN>), grid, dim3(256), lds, st, g); } \
    return 0; } while (0)
  if (g.N >= 128) S2(256, 128);
  S2(256, 64);
#undef S2
  return 0;
}
