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

// MODE_CONV_FWD: CLS unused.  MODE_CONV_BWD: CLS = pr * 2 + pc of the fine pixels this workgroup produces.  EVEN: H and W even (no class plane is short of a row / column).
// (bx, by): tile coordinates (the caller maps workgroup ids XCD-aware)
template <int BM, int BN, int MODE, int CLS, bool EVEN>
__device__ __forceinline__ void conv_s2_body(const GemmArgs& g, char* smem, const int bx, const int by) {
  constexpr bool FWD = MODE == MODE_CONV_FWD;
  constexpr int RB = 64, KE = 32;
  constexpr int NCB = BN / 64;                                // DMA passes (64 rows x 64 B) per weight tile
  constexpr int WR = BM + S2_HALO, NAF = BM / 64;             // window rows; full DMA passes (+ one 16-row piece: lanes 0-15 of every wave)
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int BTILE = BN * RB, AWIN = WR * RB;
  constexpr int PR = CLS >> 1, PC = CLS & 1;
  constexpr int NTAPS = FWD ? 9 : (CLS == 3 ? 4 : (CLS == 0 ? 1 : 2));      // steps per chunk
  constexpr int NG = FWD ? 4 : 1;                                            // window groups per chunk
  char* const Bring = smem; char* const Awin = smem + 3 * BTILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int H = g.a.H, Wd = g.a.W, OH = g.a.OH, OW = g.a.OW, C = g.a.C;
  const unsigned P = (unsigned)(FWD ? g.M : g.M / ((long long)H * Wd) * OH * OW);       // coarse pixels (host: < 2^31)
  const unsigned m0 = (unsigned)bx * BM; const int n0 = by * BN;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned bring0 = (unsigned)(uintptr_t)(lptr_t)Bring, awin0 = (unsigned)(uintptr_t)(lptr_t)Awin;
  const unsigned wslot = (unsigned)wave * 1024u;

  // ---- DMA plan.  Window row wr <-> coarse pixel pw = m0 - HALO + wr (forward) / m0 + wr (backward), clamped into [0, P).  Pass i: row 64 i + tid / 4, physical slot tid % 4
  // carrying logical chunk slot ^ swz(row); the 16-row piece: row 64 NAF + (16 wave + lane) / 4, slot lane % 4.
  unsigned aoff[NAF + 1];
  const unsigned amax = (unsigned)(((FWD ? (long long)(P / (unsigned)(OH * OW)) * H * Wd : (long long)P) * C - C + 24) * 2);  // last 16-byte piece of chunk 0 of the LAST pixel: the chunk
                                                                               // step (cc 64 bytes) is added to the base pointer after this clamp, so the clamp must leave room for it
  auto plan = [&](const int wr, const int slot) -> unsigned {
    int pws = (int)m0 + wr - (FWD ? S2_HALO : 0); pws = pws < 0 ? 0 : pws;
    const unsigned pw = (unsigned)pws >= P ? P - 1 : (unsigned)pws;
    const unsigned kc = (unsigned)((slot ^ glds_swz<RB>(wr)) * 8);
    if (FWD) {
      const unsigned t = pw / (unsigned)OW, j = pw - t * OW, img = t / (unsigned)OH, i = t - img * OH;
      return (((img * H + 2 * i) * Wd + 2 * j) * C + kc) * 2;        // class (0,0) pixel; class (pr,pc) adds (pr W + pc) C elements (clamped at issue)
    }
    return (pw * C + kc) * 2;
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

  // ---- tap validity per fragment row as AND masks (all ones = keep).  Forward: m0_ = "not the first row" (taps kh = 0 read the previous row), m1_ = "not the first column";
  // odd sizes also m2_ = "fine row 2 i + 1 exists", m3_ = "fine column 2 j + 1 exists".  Backward: m0_ = "dy row i + 1 exists" (taps kh = 0), m1_ = "dy column j + 1 exists".
  unsigned m0_[MT], m1_[MT], m2_[EVEN ? 1 : MT], m3_[EVEN ? 1 : MT];
#pragma unroll
  for (int f = 0; f < MT; ++f) {
    const unsigned p = m0 + wm * (BM / 2) + f * 32 + (lane & 31);
    const unsigned pc_ = p < P ? p : P - 1;
    const unsigned t = pc_ / (unsigned)OW, j = pc_ - t * OW, i = t % (unsigned)OH;
    if (FWD) {
      m0_[f] = i > 0 ? ~0u : 0u; m1_[f] = j > 0 ? ~0u : 0u;
      if (!EVEN) { m2_[f] = (int)(2 * i + 1) < H ? ~0u : 0u; m3_[f] = (int)(2 * j + 1) < Wd ? ~0u : 0u; }
    } else {
      m0_[f] = (int)(i + 1) < OH ? ~0u : 0u; m1_[f] = (int)(j + 1) < OW ? ~0u : 0u;
    }
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

  // ---- the step list of one chunk (compile-time).  Forward group order (1,1) | (0,0) | (1,0) | (0,1): 4 + 1 + 2 + 2 taps
  struct Plan {
    static constexpr int tap(int k) {
      if (FWD) { constexpr int o[9] = {0, 2, 6, 8, 4, 1, 7, 3, 5}; return o[k]; }
      if (CLS == 3) { constexpr int o[4] = {0, 2, 6, 8}; return o[k]; }
      if (CLS == 2) { constexpr int o[2] = {1, 7}; return o[k]; }
      if (CLS == 1) { constexpr int o[2] = {3, 5}; return o[k]; }
      return 4;
    }
    static constexpr bool first(int k) { return FWD ? (k == 0 || k == 4 || k == 5 || k == 7) : k == 0; }
    static constexpr int group(int k) { return FWD ? (k < 4 ? 0 : k < 5 ? 1 : k < 7 ? 2 : 3) : 0; }
    static constexpr int gclass(int gi) { return gi == 0 ? 3 : gi == 1 ? 0 : gi == 2 ? 2 : 1; }      // forward: window class of group gi
    // ring positions for chunk phase R = cc % 3
    static constexpr int wbuf(int R, int gi) { return FWD ? (R + gi) % 3 : R % 3; }
    static constexpr int bstage(int R, int k) { return (NTAPS * R + k) % 3; }
  };

  // window of group gi of chunk cc into buffer WB (forward: class of the group; backward: rows of dy)
  auto issue_window = [&](const int cc, auto gic, auto wbc) {
    constexpr int GI = decltype(gic)::value, WB = decltype(wbc)::value;
    unsigned delta = 0u;
    if (FWD) { constexpr int cl = Plan::gclass(GI); delta = (unsigned)(((cl >> 1) * Wd + (cl & 1)) * C2); }
    const char* const src = Ab + (long long)cc * (KE * 2);
    unsigned v[NAF];
#pragma unroll
    for (int i = 0; i < NAF; ++i) { const unsigned o = aoff[i] + delta; v[i] = (FWD && GI != 1) ? (o < amax ? o : amax) : o; }
    glds16_group<NAF>(v, src, awin0 + WB * AWIN + wslot);
    if (lane < 16) { const unsigned o = aoff[NAF] + delta; s2_glds16(src, (FWD && GI != 1) ? (o < amax ? o : amax) : o, awin0 + WB * AWIN + NAF * 4096u + (unsigned)wave * 256u); }
  };
  auto issue_b = [&](const int cc, const int tap, auto stc) {
    constexpr int ST = decltype(stc)::value;
    glds16_group<NCB>(boff, Wb + (long long)tap * C2 + (long long)cc * (KE * 2), bring0 + ST * BTILE + wslot);
  };

#define S2_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  // prologue: W(0), W(1), B(0), B(1)
  issue_window(0, IntC<0>{}, IntC<Plan::wbuf(0, 0)>{});
  if (FWD) issue_window(0, IntC<1 % NG>{}, IntC<Plan::wbuf(0, 1 % NG)>{});
  else if (NC > 1) issue_window(1, IntC<0>{}, IntC<Plan::wbuf(1, 0)>{});
  issue_b(0, Plan::tap(0), IntC<0>{});
  if (NTAPS > 1) issue_b(0, Plan::tap(1 % NTAPS), IntC<1>{});
  else if (NC > 1) issue_b(1, Plan::tap(0), IntC<Plan::bstage(1, 0)>{});

  auto step = [&](auto kc, auto rc, const int cc) {
    constexpr int K_ = decltype(kc)::value, R = decltype(rc)::value;
    constexpr int TAP = Plan::tap(K_);
    constexpr bool FIRST = Plan::first(K_);
    constexpr int GI = Plan::group(K_);
    constexpr int WB = Plan::wbuf(R, GI), ST = Plan::bstage(R, K_);
    const bool more = cc + 1 < NC;
    // loads newer than this step's weight tile: the next step's (if there is one) and the window the previous step asked for (if it did)
    const bool next_b = (K_ + 1 < NTAPS) || more;
    bool prev_w;
    if (K_ > 0) prev_w = Plan::first(K_ > 0 ? K_ - 1 : 0) && (FWD ? (Plan::group(K_ > 0 ? K_ - 1 : 0) + 2 < NG || more) : (cc + 2 < NC));
    else prev_w = (NTAPS == 1) && cc > 0 && more;               // (one-tap class: the previous chunk's only step asked for chunk cc + 1)
    if (prev_w) { if (next_b) S2_WAIT_VM(NCB + NAF + 1); else S2_WAIT_VM(NAF + 1); }
    else { if (next_b) S2_WAIT_VM(NCB); else S2_WAIT_VM(0); }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    u32x4 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      fa[q][0] = lds_read128o<WB * AWIN>(aad[TAP][q]); fa[q][1] = lds_read128o<WB * AWIN + 2048>(aad[TAP][q]);
      if (MT > 2) { fa[q][2 % MT] = lds_read128o<WB * AWIN + 4096>(aad[TAP][q]); fa[q][3 % MT] = lds_read128o<WB * AWIN + 6144>(aad[TAP][q]); }
      fb[q][0] = lds_read128o<ST * BTILE>(bad[q]);
      if (NT > 1) fb[q][1 % NT] = lds_read128o<ST * BTILE + 2048>(bad[q]);
    }
    // requests two groups / two steps ahead go out while the fragment reads are in flight (the buffers they overwrite were last read before this barrier)
    if (FIRST) {
      if (FWD) {
        if (GI + 2 < NG) issue_window(cc, IntC<(GI + 2) % NG>{}, IntC<Plan::wbuf(R, (GI + 2) % NG)>{});
        else if (more) issue_window(cc + 1, IntC<(GI + 2) % NG>{}, IntC<Plan::wbuf(R + 1, (GI + 2) % NG)>{});
      } else if (cc + 2 < NC) issue_window(cc + 2, IntC<0>{}, IntC<Plan::wbuf(R + 2, 0)>{});
    }
    { constexpr int K2 = (K_ + 2) % NTAPS, DC = (K_ + 2) / NTAPS;
      if (DC == 0 || cc + DC < NC) issue_b(cc + DC, Plan::tap(K2), IntC<(ST + 2) % 3>{}); }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[q][i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[q][j]));
      { constexpr int kh = TAP / 3, kw = TAP % 3;
        constexpr bool u0 = kh == 0, u1 = kw == 0, u2 = FWD && !EVEN && kh == 2, u3 = FWD && !EVEN && kw == 2;
        if (u0 || u1 || u2 || u3) {
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            unsigned m = ~0u;
            if (u0) m &= m0_[i];
            if (u1) m &= m1_[i];
            if (u2) m &= m2_[EVEN ? 0 : i];
            if (u3) m &= m3_[EVEN ? 0 : i];
            fa[q][i][0] &= m; fa[q][i][1] &= m; fa[q][i][2] &= m; fa[q][i][3] &= m;
          }
        } }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[q][j]), __builtin_bit_cast(bf16x8_t, fa[q][i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto chunk = [&](auto rc, const int cc) {
    step(IntC<0>{}, rc, cc);
    if constexpr (NTAPS > 1) step(IntC<1 % NTAPS>{}, rc, cc);
    if constexpr (NTAPS > 2) { step(IntC<2 % NTAPS>{}, rc, cc); step(IntC<3 % NTAPS>{}, rc, cc); }
    if constexpr (NTAPS > 4) { step(IntC<4 % NTAPS>{}, rc, cc); step(IntC<5 % NTAPS>{}, rc, cc); step(IntC<6 % NTAPS>{}, rc, cc); step(IntC<7 % NTAPS>{}, rc, cc); step(IntC<8 % NTAPS>{}, rc, cc); }
  };
#pragma unroll 1
  for (int cc = 0; cc < NC; cc += 3) {
    chunk(IntC<0>{}, cc);
    if (cc + 1 < NC) chunk(IntC<1>{}, cc + 1);
    if (cc + 2 < NC) chunk(IntC<2>{}, cc + 2);
  }
#undef S2_WAIT_VM
  __syncthreads();
  // ---- epilogue rows: forward the coarse pixel itself; backward its fine pixel of this class (none for odd sizes at the last row / column)
  const bool use_res = g.e.res != nullptr && (!g.e.res_cls0 || (!FWD && CLS == 0));
  long long prow[MT], rrow[MT]; bool pvalid[MT];
#pragma unroll
  for (int f = 0; f < MT; ++f) {
    const unsigned p = m0 + wm * (BM / 2) + f * 32 + (lane & 31);
    const unsigned pc_ = p < P ? p : P - 1;
    if (FWD) { prow[f] = pc_; pvalid[f] = p < P; }
    else {
      const unsigned t = pc_ / (unsigned)OW, j = pc_ - t * OW, img = t / (unsigned)OH, i = t - img * OH;
      const int y = 2 * (int)i + PR, x = 2 * (int)j + PC;
      pvalid[f] = p < P && y < H && x < Wd;
      prow[f] = ((long long)img * H + (y < H ? y : H - 1)) * (long long)Wd + (x < Wd ? x : Wd - 1);
    }
    rrow[f] = g.e.res_cls0 ? (long long)pc_ : prow[f];
  }
  bool full = m0 + BM <= P;
  if (!FWD && !EVEN) full = false;            // odd sizes: some coarse pixels have no fine pixel in this class
  conv_epilogue_tr<BM, BN, MT, NT>(g, acc, smem, prow, pvalid, use_res ? (const bf16*)g.e.res : nullptr, rrow, full, n0, tid, lane, wm, wn);
}

// XCD-aware tile order: XCD x takes a contiguous range of logical ids, column tile fastest (the column tiles of a row tile share its windows in ONE L2)
template <int BM, int BN, bool EVEN>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_fwd_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ny = (int)gridDim.y;
  const int lid = xcd_logical((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  conv_s2_body<BM, BN, MODE_CONV_FWD, 0, EVEN>(g, smem, lid / ny, lid % ny);
}

template <int BM, int BN, bool EVEN>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_bwd_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // heaviest class first: slice blockIdx.z = 0..3 -> classes (1,1), (1,0), (0,1), (0,0); the XCD-aware order inside a slice (every XCD works through all four classes: their
  // tiles cost 4 : 2 : 2 : 1).  A slice may start on any XCD: a rotation of the physical ids, each still gets one contiguous logical range.  (Measured and dropped: the four
  // class tiles of a row range side by side in one L2 -- class fastest in the logical order -- 170 / 109 / 87 -> 183 / 126 / 106 us on the three stage boundaries.)
  const int nx = (int)gridDim.x, ny = (int)gridDim.y, per = nx * ny;
  const int z = (int)blockIdx.z;
  const int rem = xcd_logical((int)(blockIdx.y * gridDim.x + blockIdx.x), per), bx = rem / ny, by = rem - bx * ny;
  if (z == 0) conv_s2_body<BM, BN, MODE_CONV_BWD, 3, EVEN>(g, smem, bx, by);
  else if (z == 1) conv_s2_body<BM, BN, MODE_CONV_BWD, 2, EVEN>(g, smem, bx, by);
  else if (z == 2) conv_s2_body<BM, BN, MODE_CONV_BWD, 1, EVEN>(g, smem, bx, by);
  else conv_s2_body<BM, BN, MODE_CONV_BWD, 0, EVEN>(g, smem, bx, by);
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
  if (imgs * fine * a.C * 2 >= (1ll << 32) || P * a.C * 2 >= (1ll << 32) || P < 1 || P + 1024 >= (1ll << 31) || imgs * fine >= (1ll << 31)) return 1;
  GemmArgs g = g_in; g.perm2 = 0;
  const bool even = !((a.H | a.W) & 1);
#define S2(BM, BN, EV) do { \
    const size_t lds = (size_t)3 * BN * 64 + (size_t)3 * (BM + S2_HALO) * 64; \
    if (mode == MODE_CONV_FWD) { dim3 grid((unsigned)((P + BM - 1) / BM), (unsigned)((g.N + BN - 1) / BN)); \
      avec_note_kernel("conv3x3_s2_fwd_kernel<%d,%d,%d>", BM, BN, (int)EV); if (int r = s2_want_lds(conv3x3_s2_fwd_kernel<BM, BN, EV>, lds)) return r; \
      hipLaunchKernelGGL((conv3x3_s2_fwd_kernel<BM, BN, EV>), grid, dim3(256), lds, st, g); } \
    else { dim3 grid((unsigned)((P + BM - 1) / BM), (unsigned)((g.N + BN - 1) / BN), 4); \
      avec_note_kernel("conv3x3_s2_bwd_kernel<%d,%d,%d>", BM, BN, (int)EV); if (int r = s2_want_lds(conv3x3_s2_bwd_kernel<BM, BN, EV>, lds)) return r; \
      hipLaunchKernelGGL((conv3x3_s2_bwd_kernel<BM, BN, EV>), grid, dim3(256), lds, st, g); } \
    return 0; } while (0)
  if (g.N >= 128) { if (even) S2(256, 128, true); else S2(256, 128, false); }
  if (even) S2(256, 64, true); else S2(256, 64, false);
#undef S2
  return 0;
}
