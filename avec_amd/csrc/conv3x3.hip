// 3x3, stride-1, 64 -> 64 channel convolution of the first ResNet-18 stage (nnet/networks.py ResNet stage 1, 22x22 images): slab kernel.
//
// Why not the implicit-GEMM kernel here: with N = 64 output channels there is one column tile, so every workgroup re-fetches the whole 72 KB
// weight matrix, and every input pixel travels L2 -> LDS nine times (once per tap): 2.7 GB of LDS-DMA per launch for 0.4 GB of tensors, which is
// what bounds that kernel at ~450 TFLOP/s on these four layers (and their backward-data twins).
// Here a persistent workgroup (8 waves) keeps the WHOLE weight matrix in LDS (64 x 576 bf16, 80-chunk rows, XOR-swizzled) and loads one image slab
// (a zero row above and below, one zero column shared by neighbouring rows: (H+2)*(W+1) + 1 pixels x 128 B, XOR-swizzled pixel pairs) per iteration by LDS-DMA: every input byte crosses L2 -> LDS once, the nine taps
// are nine LDS offsets.  Per image: M = H*W pixels (padded to 512 = 8 waves x 64), N = 64, K = 576 -> 36 k-steps of v_mfma_f32_32x32x16_bf16.
// The product is computed transposed (C^T = W . X^T): a lane then owns ONE pixel and groups of 4 consecutive channels, so the bf16 output leaves
// the registers as 8-byte pieces of its NHWC row and the BatchNorm statistics are per-register sums reduced once at the end.
// LDS: 81 920 (weights) + 73 728 (slab) = 155 648 B -> one workgroup per CU, so there is no room for a second slab.  Round 5: the slab of image n+1 is
// PREFETCHED INTO REGISTERS (nine 16-byte pieces per lane, plain loads issued at the start of image n's MFMA phase) and written to the LDS slab by ds_write_b128
// between the two barriers that separate the images: the HBM round trip of an image (62 KB per CU at the ~19 GB/s a CU gets when all 256 stream: the slab's
// LDS-DMA used to be waited for with nothing to overlap, ~7 of the ~13 us per image) now runs under the 36 k-steps of the previous image.  The zero border is
// written once.  The kernel is instantiated per (statistics, residual) pair: the forward launch carries no residual pieces, the backward one no statistics
// registers -- which is where the 36 prefetch registers come from.
// forward:   y[p][co]  = sum_{tap,ci} x[p + tap - 1][ci] * Wf[co][tap][ci]          (flip = 0, weight = forward shadow  [Cout][9][Cin])
// backward:  dx[p][ci] = sum_{tap,co} dy[p - tap + 1][co] * Wb[ci][tap][co] (+ res)  (flip = 1, weight = backward shadow [Cin][9][Cout])
#include "common.h"
#include "vec.h"
#include "avec_hip.h"

typedef __attribute__((ext_vector_type(16))) float c3_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 c3_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned c3_u32x4;

#define C3_WPITCH 1280                      // bytes per weight row in LDS: 72 data chunks in 80 slots (5 groups of 16)
#define C3_WBYTES (64 * C3_WPITCH)          // 81 920
#define C3_MAXPIX 576                       // slab pixels ((H+2)*(W+2), even)
#define C3_SBYTES (C3_MAXPIX * 128)         // 73 728

__device__ __attribute__((aligned(64))) unsigned char c3_zero16[64];     // source of the zero border (LDS-DMA has no immediate fill)

struct C3Args { const bf16* x; const bf16* w; bf16* y; const bf16* res; float* stats; int N, H, W, flip; const unsigned char* rmask; };      // rmask: one bit per element of res (added where set)

__device__ __forceinline__ int c3_wslot(int n, int c) { return (c & ~15) | ((c & 15) ^ (n & 15)); }
// slab: 256-byte rows of two pixels (16 chunks), chunk index XOR (pair index & 7): a ds_read_b128 lane group that reads 16 CONSECUTIVE pixels (any parity of the
// first) touches 16 different 16-byte bank slots.  Output pixel (oy, ox) and tap (kh, kw) read slab pixel oy*(W+1) + ox + kh*(W+1) + kw: consecutive pixels stay
// consecutive except for a +1 step at the end of an image row (one 2-way conflict in the lane groups that contain it): the slab keeps ONE zero column per row,
// shared by the right edge of row y and the left edge of row y+1.
__device__ __forceinline__ int c3_saddr(int pix, int c) { const int pr = pix >> 1; return pr * 256 + (((((pix & 1) << 3) | c) ^ (pr & 7)) << 4); }
// lane -> pixel of a 32-pixel tile such that the two b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} each own 16 consecutive pixels
__device__ __forceinline__ int c3_lane_pixel(int l) { return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l; }

#ifndef C3S_ABL
#define C3S_ABL 0         // timing experiments on the slab kernel: 1 no MFMA, 2 no slab prefetch / ds_write, 4 no fragment reads, 8 no fold / pack / stores, 16 no output stores
#endif
template <bool STATS, bool RES>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(C3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem; char* Sl = smem + C3_WBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W, PW = W + 1, HW = H * W, NPIX = (H + 2) * PW + 1;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // ---- weights -> LDS once (plain loads + ds_write: 9 chunks per thread) ----
  for (int q = tid; q < 64 * 72; q += 512) {
    const int n = q / 72, c = q - n * 72;
    *(chunk16*)(Ws + n * C3_WPITCH + c3_wslot(n, c) * 16) = ldg16(a.w + (long long)n * 576 + c * 8);
  }
  // ---- slab DMA plan: piece q = wave + 8*i (i < 9) covers slots [q*64, q*64+64); slot s of pair-row pr holds logical chunk s ^ (pr & 15) ----
  int soff[9];                               // element offset inside the image, or -1 = zero (border / beyond the slab)
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int S = (wave + 8 * i) * 64 + lane, pr = S >> 4, c2 = (S & 15) ^ (pr & 7);
    const int pix = pr * 2 + (c2 >> 3), ch = c2 & 7;          // slab pixel = 1 + y' * (W+1) + x: rows 0 and H+1 are zero, column W is the zero column shared by two rows
    const int py = (pix - 1) / PW, px = (pix - 1) - py * PW;
    const bool in = pix >= 1 && pix < NPIX && py >= 1 && py <= H && px < W;
    soff[i] = in ? ((py - 1) * W + px) * 64 + ch * 8 : -1;
  }
  // ---- fragment addressing ----
  // pixel operand (MFMA B): lane -> pixel row (lane & 31) of m-tile i, k-half lane >> 5;  weight operand (MFMA A): lane -> channel (lane & 31) of n-tile j
  const int kh2 = lane >> 5;
  int p0[2]; bool pvalid[2]; int pm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = wave * 64 + i * 32 + c3_lane_pixel(lane & 31);
    pvalid[i] = m < HW; if (m >= HW) m = HW - 1;
    pm[i] = m;
    const int oy = m / W;
    p0[i] = oy * PW + (m - oy * W);          // slab pixel of tap offset (0, 0); offset (dh, dw) adds dh*PW + dw
  }
  const int wrow[2] = {(lane & 31) * C3_WPITCH, (32 + (lane & 31)) * C3_WPITCH};
  const int wkey = lane & 15;                // (n & 15) for both n-tiles

  float ssum[STATS ? 2 : 1][16], ssq[STATS ? 2 : 1][16];      // BatchNorm statistics of this lane's channels over its pixels, all images of the workgroup
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { ssum[j][r] = 0.f; ssq[j][r] = 0.f; }
  }

  c3_f32x16 acc[2][2];                       // [n-tile j][m-tile i], C^T layout: column = pixel (lane & 31), rows = channels (r&3) + 8*(r>>2) + 4*(lane>>5)
  // Results leave in two steps so that nobody waits for a store: right after the MFMAs of image n the accumulators are folded into the statistics, added to the
  // residual and packed to bf16 (32 registers); those 8-byte pieces are stored at the START of image n+1's MFMA phase, after the wait for its slab, and
  // complete under that phase.  The residual pieces of image n are requested at the start of its own MFMA phase and consumed after it.
  // A lane owns channels {0-3, 8-11, 16-19, 24-27} (+4 for the upper half-wave) of its pixel: v_permlane32_swap trades the odd 4-channel group of the lower
  // half-wave for the even group of the upper one, after which every lane holds 8 consecutive channels = one 16-byte piece per (n-tile, 16-channel block).
  // Round 5: every global access of the image loop is inline asm with hand-counted waits.  Left to the compiler, the wait for the prefetched pieces (issued before
  // the CONDITIONAL stores of the previous image) comes out as a drain of everything, store acknowledgements included: the in-kernel stamps showed 4.8 of 18 thousand
  // cycles per image in that wait and 2.5 more where two 1 KB stores per tap throttled taps 1-4.  Issue order per image and wave (all 16-byte pieces per lane):
  //   tap t = 0..3: next image's pieces 2t, 2t+1 | residual pieces 2t, 2t+1 (backward) | store t of the previous image;  tap 4: piece 8, store 4;  taps 5-7: stores 5-7
  // so after tap 8 `s_waitcnt vmcnt(3)` covers every load and leaves the three youngest stores in flight (vmcnt counts in issue order; no stores in the first image: 0).
  c3_u32x4 outp[2][2][2], resp[RES ? 2 : 1][2][2];
  long long prev = -1;                       // image whose packed results are still in outp
  auto swap_pair = [&](uint2& x, uint2& y) {   // (X, Y) = (group 2k, group 2k+1) <-> (channels 0-7 | 8-15 of the 16-block): an involution
    auto r0 = __builtin_amdgcn_permlane32_swap(x.x, y.x, false, false); x.x = r0[0]; y.x = r0[1];
    auto r1 = __builtin_amdgcn_permlane32_swap(x.y, y.y, false, false); x.y = r1[0]; y.y = r1[1];
  };
  const int chq = kh2 * 8;                   // this lane's 8-channel piece inside a 16-channel block after the swap
  unsigned ldoff[9], zmask = 0u;             // byte offset of piece i inside an image (border slots read offset 0 and are zeroed on their way into the slab: bit i of zmask)
#pragma unroll
  for (int i = 0; i < 9; ++i) { ldoff[i] = soff[i] < 0 ? 0u : (unsigned)soff[i] * 2u; if (soff[i] < 0) zmask |= 1u << i; }
  // invalid lanes (pixels beyond the image) were clamped to the last pixel: they compute and store ITS value again (same address, same bytes): no predicate on the stores
  const unsigned stoff[2] = {(unsigned)pm[0] * 128u + (unsigned)chq * 2u, (unsigned)pm[1] * 128u + (unsigned)chq * 2u};
// ("+v": the destination is the loop-carried register itself -- with "=v" the compiler loads into a temporary and COPIES it into the loop variable before the data lands)
// hazards the compiler does not pad inside an asm string: a base SGPR fresh from the scalar ALU needs 5 states before a global_* reads it (s_nop 4 in front);
// a 16-byte store reads its data registers for two more states (s_nop 1 behind it)
#define C3_GLOAD(dst, voff, sbase, OFF) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(OFF))
#define C3_GSTORE(voff, data, sbase, OFF) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" :: "v"(voff), "v"(data), "s"(sbase), "n"(OFF))
  auto issue_stores = [&]() {
    if (prev < 0 || (C3S_ABL & (8 | 16))) return;
    bf16* yo = a.y + prev * HW * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!pvalid[i]) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) *(c3_u32x4*)(yo + (long long)pm[i] * 64 + j * 32 + k * 16 + chq) = outp[j][i][k];
    }
  };

  c3_u32x4 pf[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) pf[i] = c3_u32x4{0u, 0u, 0u, 0u};
  if (RES) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) resp[RES ? j : 0][i][k] = c3_u32x4{0u, 0u, 0u, 0u};
  }
  if ((long long)blockIdx.x < a.N && !(C3S_ABL & 2)) {
    const bf16* x0 = a.x + (long long)blockIdx.x * HW * 64;
#pragma unroll
    for (int i = 0; i < 9; ++i) C3_GLOAD(pf[i], ldoff[i], x0, 0);
    asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(pf[i]));
  }
#ifdef C3S_TRACE
  // phase stamps of workgroup C3S_TRACE_WG, wave 0 (STATS builds): a.stats + 8192 + 8 * image ordinal: loop top | after the slab is written | taps 0-4 | taps 5-8 | fold (core clock) | 100 MHz clock
  int tr_k = 0;
  float* const trbuf = STATS ? a.stats + 8192 : (float*)(a.y + (long long)a.N * HW * 64);      // (backward: behind the output tensor, which the trace script allocates longer)
#define C3S_STAMP(slot) do { if (blockIdx.x == C3S_TRACE_WG && tid == 0 && tr_k < 16) { trbuf[8 * tr_k + (slot)] = (float)(clock64() & 0xffffff); if ((slot) == 0) trbuf[8 * tr_k + 6] = (float)(wall_clock64() & 0xffffff); } } while (0)
#else
#define C3S_STAMP(slot) do {} while (0)
#endif
  for (long long n = blockIdx.x; n < a.N; n += gridDim.x) {
    C3S_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)");      // the youngest request is piece 8 (tap 8): everything has to have landed
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(pf[i]));
    __syncthreads();                         // every wave is done reading the slab of the previous image (and the weights are in place)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (C3S_ABL & 2) break;
      c3_u32x4 v = pf[i];
      if (zmask & (1u << i)) v = c3_u32x4{0u, 0u, 0u, 0u};
      *(c3_u32x4*)(Sl + ((wave + 8 * i) * 64 + lane) * 16) = v;
    }
    __syncthreads();
    C3S_STAMP(1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    // 36 k-steps (9 taps x 4 chunks of 16 channels); the fragments of step s+1 are requested before the MFMAs of step s
    // (the taps are unrolled: without an opaque input per image the compiler hoists every tap's loop-invariant fragment addresses out of the image loop and spills them)
    int p0o[2] = {p0[0], p0[1]}, wkeyo = wkey;
    asm volatile("" : "+v"(p0o[0]), "+v"(p0o[1]), "+v"(wkeyo));
    auto tap_addr = [&](int tap, int& ax0, int& ax1, int& aw0, int& aw1, bool& z0, bool& z1) {
      const int th = tap / 3, tw = tap - th * 3;
      const int dh = (a.flip & 1) ? 2 - th : th, dw = (a.flip & 1) ? 2 - tw : tw;
      const int d = dh * PW + dw;
      ax0 = c3_saddr(p0o[0] + d, kh2); ax1 = c3_saddr(p0o[1] + d, kh2);
      const int wt = (((tap >> 1) << 4) | ((((tap & 1) << 3) | kh2) ^ wkeyo)) << 4;
      aw0 = wrow[0] + wt; aw1 = wrow[1] + wt;
      z0 = z1 = false;
    };
    auto load4 = [&](int ax0, int ax1, int aw0, int aw1, bool z0, bool z1, int q, chunk16& fx0, chunk16& fx1, chunk16& fw0, chunk16& fw1) {
      if (C3S_ABL & 4) { fx0.w[0] = fx0.w[1] = fx0.w[2] = fx0.w[3] = ax0 + q; fx1 = fx0; fw0 = fx0; fw1 = fx0; return; }
      fx0 = *(const chunk16*)(Sl + (ax0 ^ (q << 5)));
      fx1 = *(const chunk16*)(Sl + (ax1 ^ (q << 5)));
      fw0 = *(const chunk16*)(Ws + (aw0 ^ (q << 5)));
      fw1 = *(const chunk16*)(Ws + (aw1 ^ (q << 5)));
    };
    // (the loads are unconditional and sit in straight-line code: a conditional or a switch around an asm with an output makes the compiler load into a temporary
    // and copy it into the loop-carried register before the data lands; the last image of a workgroup prefetches itself again)
    const bool has_prev = prev >= 0 && !(C3S_ABL & (8 | 16)), has_next = n + gridDim.x < a.N;
    const bf16* const yprev = a.y + (has_prev ? prev : 0) * HW * 64;
    const bf16* const xnext = a.x + (has_next ? n + gridDim.x : n) * HW * 64;
    const bf16* const rcur = RES ? a.res + n * HW * 64 : a.x;
    auto one_tap = [&](int tap) {
      int ax0, ax1, aw0, aw1; bool z0, z1;
      tap_addr(tap, ax0, ax1, aw0, aw1, z0, z1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        chunk16 cx0, cx1, cw0, cw1;
        load4(ax0, ax1, aw0, aw1, z0, z1, q, cx0, cx1, cw0, cw1);
        if (C3S_ABL & 1) { asm volatile("" :: "v"(cx0.w[0]), "v"(cx0.w[3]), "v"(cx1.w[0]), "v"(cx1.w[3]), "v"(cw0.w[0]), "v"(cw0.w[3]), "v"(cw1.w[0]), "v"(cw1.w[3])); continue; }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, cw0), __builtin_bit_cast(c3_bf16x8, cx0), acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, cw0), __builtin_bit_cast(c3_bf16x8, cx1), acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, cw1), __builtin_bit_cast(c3_bf16x8, cx0), acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, cw1), __builtin_bit_cast(c3_bf16x8, cx1), acc[1][1], 0, 0, 0);
      }
    };
    // piece k of a pixel tile: k = 4 i + 2 j + kk  ->  outp / resp [j][i][kk] at byte offset 64 j + 32 kk of the lane's row piece
#define C3_RS(I, J, K) do { if (RES) C3_GLOAD(resp[RES ? (J) : 0][I][K], stoff[I], rcur, (J) * 64 + (K) * 32); } while (0)
#define C3_ST(I, J, K) do { if (has_prev) C3_GSTORE(stoff[I], outp[J][I][K], yprev, (J) * 64 + (K) * 32); } while (0)
#define C3_LD1(I) do { if (!(C3S_ABL & 2)) C3_GLOAD(pf[I], ldoff[I], xnext, 0); } while (0)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // one piece of the next image and one store of the previous one per tap (waited for at the top of the next image); backward: the eight residual pieces of
      // THIS image in front of them in taps 0-3, waited for before the fold with everything issued from tap 4 on (5 pieces + 4 stores) still in flight
      switch (tap) {
        case 0: C3_LD1(0); C3_ST(0, 0, 0); break;
        case 1: C3_LD1(1); C3_ST(0, 0, 1); break;
        case 2: C3_LD1(2); C3_ST(0, 1, 0); break;
        case 3: C3_LD1(3); C3_ST(0, 1, 1); break;
        case 4: C3_LD1(4); C3_ST(1, 0, 0); break;
        case 5: C3_RS(0, 0, 0); C3_RS(0, 0, 1); C3_RS(0, 1, 0); C3_RS(0, 1, 1); C3_RS(1, 0, 0); C3_RS(1, 0, 1); C3_RS(1, 1, 0); C3_RS(1, 1, 1); C3_LD1(5); C3_ST(1, 0, 1); break;
        case 6: C3_LD1(6); C3_ST(1, 1, 0); break;
        case 7: C3_LD1(7); C3_ST(1, 1, 1); break;
        default: C3_LD1(8); break;
      }
      one_tap(tap);
      if (tap == 4) C3S_STAMP(2);
    }
#undef C3_LD1
#undef C3_RS
#undef C3_ST
    if (RES) { if (C3S_ABL & 2) asm volatile("s_waitcnt vmcnt(0)"); else if (has_prev) asm volatile("s_waitcnt vmcnt(7)"); else asm volatile("s_waitcnt vmcnt(4)"); }
    if (RES) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(resp[RES ? j : 0][i][k]));
    }
    C3S_STAMP(3);
    // fold + pack
    if (C3S_ABL & 8) {
      float t = 0.f;
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += acc[j][i][r];
      if (t == 1234.5f) outp[0][0][0][0] = 1u;
      asm volatile("" :: "v"(t));
      prev = n; continue;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          uint2 rx = make_uint2(0u, 0u), ry = make_uint2(0u, 0u);
          if (RES) { const c3_u32x4 rr = resp[RES ? j : 0][i][k]; rx = make_uint2(rr[0], rr[1]); ry = make_uint2(rr[2], rr[3]); swap_pair(rx, ry); }   // back to (group 2k, group 2k+1)
          uint2 o[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int g = 2 * k + e;
            float v[4] = {acc[j][i][g * 4 + 0], acc[j][i][g * 4 + 1], acc[j][i][g * 4 + 2], acc[j][i][g * 4 + 3]};
            if (STATS && pvalid[i]) {
#pragma unroll
              for (int t = 0; t < 4; ++t) { ssum[j][g * 4 + t] += v[t]; ssq[j][g * 4 + t] += v[t] * v[t]; }
            }
            if (RES) {
              const uint2 r = e ? ry : rx;
              v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u); v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
            }
            o[e] = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
          }
          swap_pair(o[0], o[1]);
          outp[j][i][k] = c3_u32x4{o[0].x, o[0].y, o[1].x, o[1].y};
        }
    prev = n;
    C3S_STAMP(4);
#ifdef C3S_TRACE
    ++tr_k;
#endif
  }
  issue_stores();
  if (STATS) {
    // lanes sharing lane >> 5 hold the same channels for different pixels: butterfly over the 32 pixel lanes, then over the 8 waves through LDS
    __syncthreads();
    float* red = (float*)Sl;                 // [8 waves][2 (sum, sq)][64 channels]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = ssum[j][r], q = ssq[j][r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if ((lane & 31) == 0) {
          const int ch = j * 32 + (r >> 2) * 8 + kh2 * 4 + (r & 3);
          red[(wave * 2 + 0) * 64 + ch] = s; red[(wave * 2 + 1) * 64 + ch] = q;
        }
      }
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
      for (int w = 0; w < 8; ++w) t += red[w * 128 + tid];
      float* rep = a.stats + (long long)(blockIdx.x % AVEC_STAT_REPLICAS) * 128;
      atomicAdd(rep + tid, t);               // [sum 64 | sumsq 64]
    }
  }
}

// The backward-data launches (residual gradient added): same register-staged slab, but every global access left to the compiler -- the next image's nine pieces
// requested in one go behind the barrier, the previous image's stores two per tap under taps 1-4, the eight residual pieces at tap 5.  Their residual pieces are
// 16-byte fragments of 32 different 128-byte lines per instruction, and spread over the taps by hand (as in the forward kernel above) they cost more issue time than
// the finer overlap gains: 189-210 us against 175 us in this form and 197 us with the slab by LDS-DMA (forward: 166 -> 149 us; profiles/r05_slab_phases.txt).
__device__ __forceinline__ unsigned c3_mask2(unsigned b, int d) {      // bits 2 d, 2 d + 1 -> AND mask of the two bf16 halves of dword d
  return ((0u - ((b >> (2 * d)) & 1u)) & 0xffffu) | ((0u - ((b >> (2 * d + 1)) & 1u)) & 0xffff0000u);
}
__global__ __launch_bounds__(512) void conv3x3_c64_res_kernel(C3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem; char* Sl = smem + C3_WBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W, PW = W + 1, HW = H * W, NPIX = (H + 2) * PW + 1;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // ---- weights -> LDS once (plain loads + ds_write: 9 chunks per thread) ----
  for (int q = tid; q < 64 * 72; q += 512) {
    const int n = q / 72, c = q - n * 72;
    *(chunk16*)(Ws + n * C3_WPITCH + c3_wslot(n, c) * 16) = ldg16(a.w + (long long)n * 576 + c * 8);
  }
  // ---- slab DMA plan: piece q = wave + 8*i (i < 9) covers slots [q*64, q*64+64); slot s of pair-row pr holds logical chunk s ^ (pr & 15) ----
  int soff[9];                               // element offset inside the image, or -1 = zero (border / beyond the slab)
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int S = (wave + 8 * i) * 64 + lane, pr = S >> 4, c2 = (S & 15) ^ (pr & 7);
    const int pix = pr * 2 + (c2 >> 3), ch = c2 & 7;          // slab pixel = 1 + y' * (W+1) + x: rows 0 and H+1 are zero, column W is the zero column shared by two rows
    const int py = (pix - 1) / PW, px = (pix - 1) - py * PW;
    const bool in = pix >= 1 && pix < NPIX && py >= 1 && py <= H && px < W;
    soff[i] = in ? ((py - 1) * W + px) * 64 + ch * 8 : -1;
  }
  // ---- fragment addressing ----
  // pixel operand (MFMA B): lane -> pixel row (lane & 31) of m-tile i, k-half lane >> 5;  weight operand (MFMA A): lane -> channel (lane & 31) of n-tile j
  const int kh2 = lane >> 5;
  int p0[2]; bool pvalid[2]; int pm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = wave * 64 + i * 32 + c3_lane_pixel(lane & 31);
    pvalid[i] = m < HW; if (m >= HW) m = HW - 1;
    pm[i] = m;
    const int oy = m / W;
    p0[i] = oy * PW + (m - oy * W);          // slab pixel of tap offset (0, 0); offset (dh, dw) adds dh*PW + dw
  }
  const int wrow[2] = {(lane & 31) * C3_WPITCH, (32 + (lane & 31)) * C3_WPITCH};
  const int wkey = lane & 15;                // (n & 15) for both n-tiles


  c3_f32x16 acc[2][2];                       // [n-tile j][m-tile i], C^T layout: column = pixel (lane & 31), rows = channels (r&3) + 8*(r>>2) + 4*(lane>>5)
  // Results leave in two steps so that nobody waits for a store: right after the MFMAs of image n the accumulators are folded into the statistics, added to the
  // residual and packed to bf16 (32 registers); those 8-byte pieces are stored at the START of image n+1's MFMA phase, after the wait for its slab, and
  // complete under that phase.  The residual pieces of image n are requested at the start of its own MFMA phase and consumed after it.
  // A lane owns channels {0-3, 8-11, 16-19, 24-27} (+4 for the upper half-wave) of its pixel: v_permlane32_swap trades the odd 4-channel group of the lower
  // half-wave for the even group of the upper one, after which every lane holds 8 consecutive channels = one 16-byte piece per (n-tile, 16-channel block).
  uint4 outp[2][2][2], resp[2][2][2];
  long long prev = -1;                       // image whose packed results are still in outp
  auto swap_pair = [&](uint2& x, uint2& y) {   // (X, Y) = (group 2k, group 2k+1) <-> (channels 0-7 | 8-15 of the 16-block): an involution
    auto r0 = __builtin_amdgcn_permlane32_swap(x.x, y.x, false, false); x.x = r0[0]; y.x = r0[1];
    auto r1 = __builtin_amdgcn_permlane32_swap(x.y, y.y, false, false); x.y = r1[0]; y.y = r1[1];
  };
  const int chq = kh2 * 8;                   // this lane's 8-channel piece inside a 16-channel block after the swap
  auto issue_stores = [&]() {
    if (prev < 0) return;
    bf16* yo = a.y + prev * HW * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!pvalid[i]) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) *(uint4*)(yo + (long long)pm[i] * 64 + j * 32 + k * 16 + chq) = outp[j][i][k];
    }
  };

  // branch-free: border slots load element 0 of the image and are replaced by zeros on their way into the slab
  chunk16 pf[9];
  if ((long long)blockIdx.x < a.N) {
    const bf16* x0 = a.x + (long long)blockIdx.x * HW * 64;
#pragma unroll
    for (int i = 0; i < 9; ++i) pf[i] = ldg16(x0 + (soff[i] < 0 ? 0 : soff[i]));
  }
  for (long long n = blockIdx.x; n < a.N; n += gridDim.x) {
    __syncthreads();                         // every wave is done reading the slab of the previous image (and the weights are in place)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      chunk16 v = pf[i];
      if (soff[i] < 0) v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0u;
      *(chunk16*)(Sl + ((wave + 8 * i) * 64 + lane) * 16) = v;
    }
    __syncthreads();
    if (n + gridDim.x < a.N) {               // the next image's pieces: in flight under this image's 36 k-steps (wave-uniform branch)
      const bf16* xn = a.x + (n + gridDim.x) * HW * 64;
#pragma unroll
      for (int i = 0; i < 9; ++i) pf[i] = ldg16(xn + (soff[i] < 0 ? 0 : soff[i]));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    // 36 k-steps (9 taps x 4 chunks of 16 channels); the fragments of step s+1 are requested before the MFMAs of step s
    auto tap_addr = [&](int tap, int& ax0, int& ax1, int& aw0, int& aw1, bool& z0, bool& z1) {
      const int th = tap / 3, tw = tap - th * 3;
      const int dh = (a.flip & 1) ? 2 - th : th, dw = (a.flip & 1) ? 2 - tw : tw;
      const int d = dh * PW + dw;
      ax0 = c3_saddr(p0[0] + d, kh2); ax1 = c3_saddr(p0[1] + d, kh2);
      const int wt = (((tap >> 1) << 4) | ((((tap & 1) << 3) | kh2) ^ wkey)) << 4;
      aw0 = wrow[0] + wt; aw1 = wrow[1] + wt;
      z0 = z1 = false;
    };
    auto load4 = [&](int ax0, int ax1, int aw0, int aw1, bool z0, bool z1, int q, chunk16& fx0, chunk16& fx1, chunk16& fw0, chunk16& fw1) {
      fx0 = *(const chunk16*)(Sl + (ax0 ^ (q << 5)));
      fx1 = *(const chunk16*)(Sl + (ax1 ^ (q << 5)));
      fw0 = *(const chunk16*)(Ws + (aw0 ^ (q << 5)));
      fw1 = *(const chunk16*)(Ws + (aw1 ^ (q << 5)));
    };
    // A CU moves only ~11 B/clk to or from HBM (its share of the chip's bandwidth): issued in one go, the 8 stores of the previous image block the wave
    // for thousands of cycles.  They leave two per tap under the MFMAs of taps 1..4 instead; the residual pieces of this image are requested at tap 5
    // (two separate loops, so that the registers of the packed results and of the residual pieces can be the same ones).
    bf16* yprev = prev >= 0 ? a.y + prev * HW * 64 : nullptr;
    auto one_tap = [&](int tap) {
      int ax0, ax1, aw0, aw1; bool z0, z1;
      tap_addr(tap, ax0, ax1, aw0, aw1, z0, z1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        chunk16 cx0, cx1, cw0, cw1;
        load4(ax0, ax1, aw0, aw1, z0, z1, q, cx0, cx1, cw0, cw1);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, cw0), __builtin_bit_cast(c3_bf16x8, cx0), acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, cw0), __builtin_bit_cast(c3_bf16x8, cx1), acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, cw1), __builtin_bit_cast(c3_bf16x8, cx0), acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, cw1), __builtin_bit_cast(c3_bf16x8, cx1), acc[1][1], 0, 0, 0);
      }
    };
#define C3_ST(I, J, K) if (pvalid[I]) *(uint4*)(yprev + (long long)pm[I] * 64 + J * 32 + K * 16 + chq) = outp[J][I][K]
#pragma unroll 1
    for (int tap = 0; tap < 5; ++tap) {
      if (yprev) {
        switch (tap) {
          case 1: C3_ST(0, 0, 0); C3_ST(0, 0, 1); break;
          case 2: C3_ST(0, 1, 0); C3_ST(0, 1, 1); break;
          case 3: C3_ST(1, 0, 0); C3_ST(1, 0, 1); break;
          case 4: C3_ST(1, 1, 0); C3_ST(1, 1, 1); break;
          default: break;
        }
      }
      one_tap(tap);
    }
#undef C3_ST
    if (a.res) {
      const bf16* ro = a.res + n * HW * 64;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int k = 0; k < 2; ++k) resp[j][i][k] = *(const uint4*)(ro + (long long)pm[i] * 64 + j * 32 + k * 16 + chq);
      if (a.rmask) {                         // the residual passes a ReLU mask (one byte per 8-channel piece): the masked gradient is never a tensor of its own
        const unsigned char* mo = a.rmask + n * HW * 8;
        unsigned mb[2][2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) mb[j][i][k] = mo[((long long)pm[i] * 64 + j * 32 + k * 16 + chq) >> 3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const unsigned b = mb[j][i][k];
              resp[j][i][k].x &= c3_mask2(b, 0); resp[j][i][k].y &= c3_mask2(b, 1); resp[j][i][k].z &= c3_mask2(b, 2); resp[j][i][k].w &= c3_mask2(b, 3);
            }
      }
    }
#pragma unroll 1
    for (int tap = 5; tap < 9; ++tap) one_tap(tap);
    // fold + pack
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          uint2 rx = make_uint2(0u, 0u), ry = make_uint2(0u, 0u);
          if (a.res) { rx = make_uint2(resp[j][i][k].x, resp[j][i][k].y); ry = make_uint2(resp[j][i][k].z, resp[j][i][k].w); swap_pair(rx, ry); }   // back to (group 2k, group 2k+1)
          uint2 o[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int g = 2 * k + e;
            float v[4] = {acc[j][i][g * 4 + 0], acc[j][i][g * 4 + 1], acc[j][i][g * 4 + 2], acc[j][i][g * 4 + 3]};
            if (a.res) {
              const uint2 r = e ? ry : rx;
              v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u); v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
            }
            o[e] = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
          }
          swap_pair(o[0], o[1]);
          outp[j][i][k] = make_uint4(o[0].x, o[0].y, o[1].x, o[1].y);
        }
    prev = n;
  }
  issue_stores();
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the same layers: dW[co][tap][ci] = sum over images and pixels of dy[p][co] * x[p + tap - 1][ci].
// The implicit-GEMM TN kernel is fabric-bound here (one 64-wide row tile: dy is re-read by every column tile, x once per tap: 1.95 GB fetched per
// launch at 6 TB/s for 0.4 GB of tensors).  Slab version: a persistent 8-wave workgroup owns the WHOLE 64 x 576 fp32 result in registers (36 tiles of
// 32 x 32 spread 5/4 over the waves) and streams images through LDS: the x slab (as in the forward kernel) and the dy image laid out with the same
// row pitch W+1 (one zero column per image row), so that the reduction index k' = oy*(W+1) + ox is linear and tap (dh, dw) is the plain row offset
// dh*(W+1) + dw into the slab.  Both MFMA operands are [k'][channel] arrays read with ds_read_b64_tr_b16 (the transposing read of gemm_tn_tr_kernel);
// every input byte crosses L2 -> LDS once.  One atomic add per result element per workgroup at the end.
// ------------------------------------------------------------------------------------------------
typedef short c3_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ chunk16 c3_tr_read8(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) c3_v4s* lp_t;
  const c3_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p1);
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  chunk16 f; f.w[0] = ua.x; f.w[1] = ua.y; f.w[2] = ub.x; f.w[3] = ub.y; return f;
}
__device__ __forceinline__ void c3_glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
typedef __attribute__((ext_vector_type(2))) unsigned c3_u32x2;
__device__ __forceinline__ c3_u32x2 c3_lds_tr(unsigned lds_addr) { c3_u32x2 v; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory"); return v; }
#ifndef C3W_ABL
#define C3W_ABL 0         // timing experiments on the 64-channel weight gradient (tools/build_abl_c3w.sh): 1 no MFMA, 2 no LDS-DMA, 4 no fragment reads, 8 no final atomics
#endif
#define C3W_KROWS 512                       // reduction rows per image: H*(W+1) <= 512
#define C3W_DBYTES (C3W_KROWS * 128)        // 65 536: dy image with the slab's row pitch

struct C3WArgs { const bf16* x; const bf16* dy; float* dw; int N, H, W; };
#if C3W_ABL & 64
__device__ float c3w_dbg[128];               // per workgroup (first 16): cycles of wave 0 in the four waits of an image and in the k-step groups
extern "C" int avec_c3w_debug(float* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3w_dbg), sizeof(float) * 128); }
#endif

// bid / nwg: this workgroup's index among the nwg workgroups that share the product
__device__ __forceinline__ void c3w_body(const C3WArgs& a, const int bid, const int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem; char* Ds = smem + C3_SBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, PW = W + 1, HW = H * W, NPIX = (H + 2) * PW + 1;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // 128-byte rows, 16-byte chunk index XOR 4*((row >> 1) & 1): the 4 rows of a transposing read fall on distinct banks (as in gemm_tn_tr_kernel)
  int xoff[9], doff[8];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int S = (wave + 8 * i) * 64 + lane, row = S >> 3, c = (S & 7) ^ (4 * ((row >> 1) & 1));
    const int py = (row - 1) / PW, px = (row - 1) - py * PW;
    const bool in = row >= 1 && row < NPIX && py >= 1 && py <= H && px < W;
    xoff[i] = in ? ((py - 1) * W + px) * 64 + c * 8 : -1;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int S = (wave + 8 * i) * 64 + lane, row = S >> 3, c = (S & 7) ^ (4 * ((row >> 1) & 1));
    const int oy = row / PW, ox = row - oy * PW;
    doff[i] = (oy < H && ox < W) ? (oy * W + ox) * 64 + c * 8 : -1;
  }
  // this wave's tiles: co half (wave >> 1) & 1, ci half wave & 1, taps (wave >> 2) + 2 j (j < 4) and, for waves 0..3, tap 8
  const int cohalf = (wave >> 1) & 1, cihalf = wave & 1, tap0 = wave >> 2;
  const bool five = wave < 4;
  const int g4 = lane >> 4, t = lane & 15;
  int offa[2], offb[5][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int krow = 8 * (g4 >> 1) + 4 * h + (t >> 2);
    const int cbA = cohalf * 32 + 16 * (g4 & 1), cbB = cihalf * 32 + 16 * (g4 & 1);
    offa[h] = krow * 128 + ((((cbA >> 3) + ((t & 3) >> 1)) ^ (4 * ((krow >> 1) & 1))) << 4) + (t & 1) * 8;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int tap = j < 4 ? tap0 + 2 * j : 8;
      const int rowd = krow + (tap / 3) * PW + (tap % 3);
      offb[j][h] = rowd * 128 + ((((cbB >> 3) + ((t & 3) >> 1)) ^ (4 * ((rowd >> 1) & 1))) << 4) + (t & 1) * 8;
    }
  }
  c3_f32x16 acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const unsigned xs0 = (unsigned)(uintptr_t)(lptr_t)Xs, ds0 = (unsigned)(uintptr_t)(lptr_t)Ds;
  for (long long n = bid; n < a.N; n += nwg) {
    __syncthreads();                         // every wave is done with the previous image
    const bf16* xi = a.x + n * HW * 64; const bf16* di = a.dy + n * HW * 64;
    if (!(C3W_ABL & 2)) {
#pragma unroll
      for (int i = 0; i < 9; ++i) c3_glds16(xoff[i] >= 0 ? (const void*)(xi + xoff[i]) : (const void*)c3_zero16, xs0 + (wave + 8 * i) * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i) c3_glds16(doff[i] >= 0 ? (const void*)(di + doff[i]) : (const void*)c3_zero16, ds0 + (wave + 8 * i) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (compiler-scheduled reads, two steps unrolled: for this 4-5 MFMA step the hand-made ladder of wgrad3x3_wide_kernel measured 20 % slower)
#pragma unroll 2
    for (int s = 0; s < C3W_KROWS / 16; ++s) {
      const int so = s * 2048;
      chunk16 fa, fb[5];
      if (!(C3W_ABL & 4)) {
        fa = c3_tr_read8(Ds + offa[0] + so, Ds + offa[1] + so);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = c3_tr_read8(Xs + offb[j][0] + so, Xs + offb[j][1] + so);
        if (five) fb[4] = c3_tr_read8(Xs + offb[4][0] + so, Xs + offb[4][1] + so);
      } else {
        fa.w[0] = fa.w[1] = fa.w[2] = fa.w[3] = (unsigned)(so + lane);
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[j].w[0] = fb[j].w[1] = fb[j].w[2] = fb[j].w[3] = (unsigned)(so * (j + 2) + lane);
      }
      if (!(C3W_ABL & 1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fa), __builtin_bit_cast(c3_bf16x8, fb[j]), acc[j], 0, 0, 0);
        if (five) acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fa), __builtin_bit_cast(c3_bf16x8, fb[4]), acc[4], 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) { if (j == 4 && !five) break; acc[j][0] += __uint_as_float(fa.w[0] ^ fb[j].w[0]); acc[j][1] += __uint_as_float(fa.w[3] ^ fb[j].w[3]); }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (j == 4 && !five) break;
    const int tap = j < 4 ? tap0 + 2 * j : 8;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cohalf * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = cihalf * 32 + (lane & 31);
      if (!(C3W_ABL & 8) || acc[j][r] == 123.456f) atomicAdd(a.dw + (long long)co * 576 + tap * 64 + ci, acc[j][r]);
    }
  }
}

// Round 5: the same product with the next image ROLLING into the slab behind the k-steps.  In-kernel cycle counts of c3w_body (C3W_ABL 64, profiles/r05_c64_wgrad_ablation.txt):
// 42 % of the kernel is the wait for the image's LDS-DMA (at 6.3 TB/s over the chip -- the rate is fine, nothing runs under it), 58 % the 32 k-steps.  A second pair of
// buffers does not fit (139 KB), but k-step s only reads dy rows [16 s, 16 s + 16) and slab rows [16 s, 16 s + 16 + 2 (W+1) + 2): after the g-th group of 8 steps the rows
// below 128 (g + 1) are dead, and the next image's rows can land there.  Four batches per image -- rows [128 g, 128 g + 128) of both regions, the last one the slab's
// rows [384, 576) -- each issued at the barrier behind the group that frees its rows and awaited (counted vmcnt, batches land in issue order) two to three groups later:
//   barrier before G0(n): needs b0(n), b1(n)   [b2(n) may be in flight: vmcnt(4)]   then issues b3(n)     (rows >= 384 were read by G3 of the previous image)
//   barrier before G1(n): needs b2(n)          [b3(n): vmcnt(5)]                    then issues b0(n+1)
//   barrier before G2(n): needs b3(n)          [b0(n+1): vmcnt(4)]                  then issues b1(n+1)
//   barrier before G3(n): needs nothing new                                         then issues b2(n+1)
// Every piece is issued by every wave whatever its lanes hold (border lanes fetch the shared zero chunk), so the counts are the same for all waves.
template <int V> struct C3I { static constexpr int value = V; };
__device__ __forceinline__ void c3w_roll_body(const C3WArgs& a, const int bid, const int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem; char* Ds = smem + C3_SBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, PW = W + 1, HW = H * W, NPIX = (H + 2) * PW + 1;
  typedef __attribute__((address_space(3))) void* lptr_t;
  int xoff[9], doff[8];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int S = (wave + 8 * i) * 64 + lane, row = S >> 3, c = (S & 7) ^ (4 * ((row >> 1) & 1));
    const int py = (row - 1) / PW, px = (row - 1) - py * PW;
    const bool in = row >= 1 && row < NPIX && py >= 1 && py <= H && px < W;
    xoff[i] = in ? ((py - 1) * W + px) * 64 + c * 8 : -1;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int S = (wave + 8 * i) * 64 + lane, row = S >> 3, c = (S & 7) ^ (4 * ((row >> 1) & 1));
    const int oy = row / PW, ox = row - oy * PW;
    doff[i] = (oy < H && ox < W) ? (oy * W + ox) * 64 + c * 8 : -1;
  }
  const int cohalf = (wave >> 1) & 1, cihalf = wave & 1, tap0 = wave >> 2;
  const bool five = wave < 4;
  const int g4 = lane >> 4, t = lane & 15;
  int offa[2], offb[5][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int krow = 8 * (g4 >> 1) + 4 * h + (t >> 2);
    const int cbA = cohalf * 32 + 16 * (g4 & 1), cbB = cihalf * 32 + 16 * (g4 & 1);
    offa[h] = krow * 128 + ((((cbA >> 3) + ((t & 3) >> 1)) ^ (4 * ((krow >> 1) & 1))) << 4) + (t & 1) * 8;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int tap = j < 4 ? tap0 + 2 * j : 8;
      const int rowd = krow + (tap / 3) * PW + (tap % 3);
      offb[j][h] = rowd * 128 + ((((cbB >> 3) + ((t & 3) >> 1)) ^ (4 * ((rowd >> 1) & 1))) << 4) + (t & 1) * 8;
    }
  }
  c3_f32x16 acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const unsigned xs0 = (unsigned)(uintptr_t)(lptr_t)Xs, ds0 = (unsigned)(uintptr_t)(lptr_t)Ds;
  auto batch = [&](auto gc, const long long n) {          // batch G of image n: slab pieces [2G, 2G + 2) (G = 3: [6, 9)) and dy pieces [2G, 2G + 2) of this wave
    constexpr int G = decltype(gc)::value;
    const bf16* xi = a.x + n * HW * 64; const bf16* di = a.dy + n * HW * 64;
#pragma unroll
    for (int i = 2 * G; i < (G == 3 ? 9 : 2 * G + 2); ++i) c3_glds16(xoff[i] >= 0 ? (const void*)(xi + xoff[i]) : (const void*)c3_zero16, xs0 + (wave + 8 * i) * 1024);
#pragma unroll
    for (int i = 2 * G; i < 2 * G + 2; ++i) c3_glds16(doff[i] >= 0 ? (const void*)(di + doff[i]) : (const void*)c3_zero16, ds0 + (wave + 8 * i) * 1024);
  };
  auto group = [&](auto gc) {                              // k-steps [8G, 8G + 8)
    constexpr int G = decltype(gc)::value;
#pragma unroll 2
    for (int s = 8 * G; s < 8 * G + 8; ++s) {
      const int so = s * 2048;
      const chunk16 fa = c3_tr_read8(Ds + offa[0] + so, Ds + offa[1] + so);
      chunk16 fb[5];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = c3_tr_read8(Xs + offb[j][0] + so, Xs + offb[j][1] + so);
      if (five) fb[4] = c3_tr_read8(Xs + offb[4][0] + so, Xs + offb[4][1] + so);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fa), __builtin_bit_cast(c3_bf16x8, fb[j]), acc[j], 0, 0, 0);
      if (five) acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fa), __builtin_bit_cast(c3_bf16x8, fb[4]), acc[4], 0, 0, 0);
    }
  };
  long long n = bid;
  if (n < a.N) { batch(C3I<0>{}, n); batch(C3I<1>{}, n); batch(C3I<2>{}, n); }
#if C3W_ABL & 64
  unsigned long long tw[4] = {0, 0, 0, 0}, tg = 0;
#define C3W_T0 const unsigned long long c_0 = __builtin_readcyclecounter();
#define C3W_TW(k) { const unsigned long long c_1 = __builtin_readcyclecounter(); tw[k] += c_1 - c_0; }
#define C3W_TG0 const unsigned long long c_2 = __builtin_readcyclecounter();
#define C3W_TG1 { asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[3][15])); tg += __builtin_readcyclecounter() - c_2; }
#else
#define C3W_T0
#define C3W_TW(k)
#define C3W_TG0
#define C3W_TG1
#endif
  for (; n < a.N; n += nwg) {
    const long long nx = n + nwg; const bool more = nx < a.N;
    { C3W_T0 asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __syncthreads(); C3W_TW(0) }
    batch(C3I<3>{}, n);
    { C3W_TG0 group(C3I<0>{}); C3W_TG1 }
    { C3W_T0 asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); __syncthreads(); C3W_TW(1) }
    if (more) batch(C3I<0>{}, nx);
    { C3W_TG0 group(C3I<1>{}); C3W_TG1 }
    { C3W_T0 if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads(); C3W_TW(2) }
    if (more) batch(C3I<1>{}, nx);
    { C3W_TG0 group(C3I<2>{}); C3W_TG1 }
    { C3W_T0 __syncthreads(); C3W_TW(3) }
    if (more) batch(C3I<2>{}, nx);
    { C3W_TG0 group(C3I<3>{}); C3W_TG1 }
  }
#if C3W_ABL & 64
  if (tid == 0 && bid < 16) { for (int k = 0; k < 4; ++k) c3w_dbg[bid * 8 + k] = (float)tw[k]; c3w_dbg[bid * 8 + 4] = (float)tg; }
#endif
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (j == 4 && !five) break;
    const int tap = j < 4 ? tap0 + 2 * j : 8;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cohalf * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = cihalf * 32 + (lane & 31);
      atomicAdd(a.dw + (long long)co * 576 + tap * 64 + ci, acc[j][r]);
    }
  }
}
#ifndef C3W_ROLL
#define C3W_ROLL 1        // 0: one image per iteration, its DMA awaited with nothing under it (c3w_body)
#endif
__global__ __launch_bounds__(512) void wgrad3x3_c64_kernel(C3WArgs a) { if (C3W_ROLL) c3w_roll_body(a, (int)blockIdx.x, (int)gridDim.x); else c3w_body(a, (int)blockIdx.x, (int)gridDim.x); }
// several 64-channel layers' weight gradients as ONE grid: the workgroups are shared out evenly, the final atomics (256 x 36 864 sums per launch) are paid once
struct C3WGroup { C3WArgs it[AVEC_WGRAD_GROUP_MAX]; int first[AVEC_WGRAD_GROUP_MAX + 1]; int n; };
__global__ __launch_bounds__(512) void wgrad3x3_c64_grouped_kernel(C3WGroup grp) {
  int i = 0;
  while (i + 1 < grp.n && (int)blockIdx.x >= grp.first[i + 1]) ++i;
  if (C3W_ROLL) c3w_roll_body(grp.it[i], (int)blockIdx.x - grp.first[i], grp.first[i + 1] - grp.first[i]);
  else c3w_body(grp.it[i], (int)blockIdx.x - grp.first[i], grp.first[i + 1] - grp.first[i]);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the wider 3x3 layers (ResNet stages 2..4: C = 128 / 256 / 512 channels, 11x11 / 6x6 / 3x3 images): same slab idea, the C x 9C fp32
// result split over (C/64) x (C/128) kinds of workgroup -- 64 output channels x 128 input channels x 9 taps = 72 tiles of 32 x 32, nine per wave (one
// (co tile, ci quarter) pair and all nine taps).  Several images per iteration: x slabs of RS rows x 256 B (this kind's 128 input channels) and dy images
// of KP rows x 128 B (its 64 output channels), all with the row pitch W+1; 352 x-rows and 288 dy-rows of LDS hold 2..10 images (as two halves:
// see the round-3 note below).  x is read C/64 times and dy C/128 times in total -- against once per column tile and
// once per tap for the implicit-GEMM TN kernel (0.65 GB fetched per launch for 0.2 GB of tensors at C = 128).
// ------------------------------------------------------------------------------------------------
#ifndef C3_ABL
#define C3_ABL 0          // timing experiments: 1 no MFMA, 2 no LDS-DMA, 4 no fragment reads
#endif
#define C3X_ROWS 352                        // x rows in LDS (two halves)
#define C3X_KROWS 288                       // dy rows in LDS (two halves)
#define C3X_XPL 6                           // DMA passes of one wave over a half: x (<= 176 rows x 16 chunks / 64 lanes / 8 waves), dy (<= 144 rows x 8 / 64 / 8)
#define C3X_DPL 3
struct C3WWArgs { const bf16* x; const bf16* dy; float* dw; int N, H, W, C, KP, RS, IT; };   // KP: reduction rows per image (16-multiple), RS: x rows per image, IT: images per HALF

// Round 3: (1) the LDS is a ring of two halves of IT images each: the LDS-DMA of the next half runs under the MFMAs of the current one (one barrier per half;
// the first version loaded, waited, computed); (2) the DMA plan -- which image / element each 16-byte LDS slot receives -- is the same for every half and
// lives in nine registers per lane instead of 32 KB of LDS read back before every DMA; (3) a step's 20 transposed reads go out together and the nine MFMAs
// follow a counted wait ladder (before: every tap's reads sat next to their MFMA behind lgkmcnt(0), ~45 address additions per step).
// bid / nwg: this workgroup's index among the nwg workgroups that share the product (a launch of its own: blockIdx.x / gridDim.x)
__device__ __forceinline__ void c3ww_body(const C3WWArgs& a, const int bid, const int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem; char* Ds = smem + C3X_ROWS * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, C = a.C, PW = W + 1, HW = H * W, NPIX = (H + 2) * PW + 1;
  const int nci = C >> 7, kinds = (C >> 6) * nci, kind = bid % kinds, cog = kind / nci, cig = kind - cog * nci;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned xs0 = (unsigned)(uintptr_t)(lptr_t)Xs, ds0 = (unsigned)(uintptr_t)(lptr_t)Ds;
  const int xhalf = a.IT * a.RS * 256, dhalf = a.IT * a.KP * 128;          // bytes of one half (multiples of 4096 / 2048: whole DMA passes)
  const int xpasses = xhalf >> 10, dpasses = dhalf >> 10;                   // 1 KB = one DMA instruction of a wave
  // plan entry: (image inside the half) * HW * C + element offset inside the image, or -1 = zero
  int xpl[C3X_XPL], dpl[C3X_DPL];
#pragma unroll
  for (int k = 0; k < C3X_XPL; ++k) {
    const int P = wave + 8 * k, S = P * 64 + lane;
    const int R = S >> 4, c = (S & 15) ^ (4 * (R & 3)), im = R / a.RS, row = R - im * a.RS;
    const int py = (row - 1) / PW, px = (row - 1) - py * PW;
    const bool in = P < xpasses && row >= 1 && row < NPIX && py >= 1 && py <= H && px < W;
    xpl[k] = in ? (im << 24) | (((py - 1) * W + px) * C + cig * 128 + c * 8) : -1;
  }
#pragma unroll
  for (int k = 0; k < C3X_DPL; ++k) {
    const int P = wave + 8 * k, S = P * 64 + lane;
    const int R = S >> 3, c = (S & 7) ^ (4 * ((R >> 1) & 1)), im = R / a.KP, row = R - im * a.KP;
    const int oy = row / PW, ox = row - oy * PW;
    dpl[k] = (P < dpasses && oy < H && ox < W) ? (im << 24) | ((oy * W + ox) * C + cog * 64 + c * 8) : -1;
  }
  const int cot = wave & 1, ciq = wave >> 1;
  const int g4 = lane >> 4, t = lane & 15;
  unsigned ada0[2], adb0[9][2];                                            // fragment addresses of step 0 of half 0
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int krow = 8 * (g4 >> 1) + 4 * h + (t >> 2);
    const int cbA = cot * 32 + 16 * (g4 & 1), cbB = ciq * 32 + 16 * (g4 & 1);
    ada0[h] = ds0 + krow * 128 + ((((cbA >> 3) + ((t & 3) >> 1)) ^ (4 * ((krow >> 1) & 1))) << 4) + (t & 1) * 8;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int rowd = krow + (j / 3) * PW + (j % 3);
      adb0[j][h] = xs0 + rowd * 256 + ((((cbB >> 3) + ((t & 3) >> 1)) ^ (4 * (rowd & 3))) << 4) + (t & 1) * 8;
    }
  }
  c3_f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const long long nhalves = ((long long)a.N + a.IT - 1) / a.IT;            // halves of IT images in the whole tensor
  const int wgs = nwg / kinds;                  // workgroups per kind (the launch rounds the count to a multiple of `kinds`)
  const int nsteps = a.KP >> 4;
  const int xjump = a.RS * 256 - (nsteps - 1) * 4096, djump = a.KP * 128 - (nsteps - 1) * 2048;      // last step of an image -> first step of the next one
  // this workgroup's halves: q = blockIdx.x / kinds + k * wgs; the ring alternates between the two LDS halves
  auto load_half = [&](long long q, int par) {
    if (C3_ABL & 2) return;
    const long long img0 = q * a.IT;
#pragma unroll
    for (int k = 0; k < C3X_XPL; ++k) {
      if (wave + 8 * k >= xpasses) break;                                     // wave-uniform
      const int e = xpl[k]; const long long img = img0 + (e >> 24);
      const void* src = (e >= 0 && img < a.N) ? (const void*)(a.x + img * HW * C + (e & 0xffffff)) : (const void*)c3_zero16;
      c3_glds16(src, xs0 + par * xhalf + (wave + 8 * k) * 1024);
    }
#pragma unroll
    for (int k = 0; k < C3X_DPL; ++k) {
      if (wave + 8 * k >= dpasses) break;
      const int e = dpl[k]; const long long img = img0 + (e >> 24);
      const void* src = (e >= 0 && img < a.N) ? (const void*)(a.dy + img * HW * C + (e & 0xffffff)) : (const void*)c3_zero16;
      c3_glds16(src, ds0 + par * dhalf + (wave + 8 * k) * 1024);
    }
  };
  long long q = bid / kinds;
  int par = 0;
  if (q < nhalves) load_half(q, 0);
  for (; q < nhalves; q += wgs, par ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // this half has landed for every wave / every wave is done with the other half
    asm volatile("" ::: "memory");
    if (q + wgs < nhalves) load_half(q + wgs, par ^ 1);
    unsigned ada[2], adb[9][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { ada[h] = ada0[h] + par * dhalf;
#pragma unroll
      for (int j = 0; j < 9; ++j) adb[j][h] = adb0[j][h] + par * xhalf; }
    int s = 0;
#pragma unroll 1
    for (int u = a.IT * nsteps; u > 0; --u) {
      c3_u32x2 ra[2], rb[9][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) ra[h] = (C3_ABL & 4) ? c3_u32x2{(unsigned)u, 1u} : c3_lds_tr(ada[h]);
#pragma unroll
      for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) rb[j][h] = (C3_ABL & 4) ? c3_u32x2{(unsigned)u, 2u} : c3_lds_tr(adb[j][h]);
      const bool wrap = ++s == nsteps; if (wrap) s = 0;
      const int dx = wrap ? xjump : 4096, dd = wrap ? djump : 2048;
#pragma unroll
      for (int h = 0; h < 2; ++h) { ada[h] += dd;
#pragma unroll
        for (int j = 0; j < 9; ++j) adb[j][h] += dx; }
#define C3_STEP(j) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((j) == 0 ? 15 : 2 * (8 - (j))) : "memory"); if ((j) == 0) asm volatile("" : "+v"(ra[0]), "+v"(ra[1])); asm volatile("" : "+v"(rb[j][0]), "+v"(rb[j][1])); \
        chunk16 fa, fb; fa.w[0] = ra[0].x; fa.w[1] = ra[0].y; fa.w[2] = ra[1].x; fa.w[3] = ra[1].y; fb.w[0] = rb[j][0].x; fb.w[1] = rb[j][0].y; fb.w[2] = rb[j][1].x; fb.w[3] = rb[j][1].y; \
        if (C3_ABL & 1) asm volatile("" :: "v"(fa.w[0]), "v"(fa.w[1]), "v"(fa.w[2]), "v"(fa.w[3]), "v"(fb.w[0]), "v"(fb.w[1]), "v"(fb.w[2]), "v"(fb.w[3])); else \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fa), __builtin_bit_cast(c3_bf16x8, fb), acc[j], 0, 0, 0); } while (0)
      C3_STEP(0); C3_STEP(1); C3_STEP(2); C3_STEP(3); C3_STEP(4); C3_STEP(5); C3_STEP(6); C3_STEP(7); C3_STEP(8);
#undef C3_STEP
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cog * 64 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = cig * 128 + ciq * 32 + (lane & 31);
      if (C3_ABL & 8) { if (acc[j][r] == 1234.5f) a.dw[(long long)co * 9 * C + j * C + ci] = acc[j][r]; } else
      atomicAdd(a.dw + (long long)co * 9 * C + j * C + ci, acc[j][r]);
    }
}

__global__ __launch_bounds__(512) void wgrad3x3_wide_kernel(C3WWArgs a) { c3ww_body(a, (int)blockIdx.x, (int)gridDim.x); }
// several layers' weight gradients as ONE grid (the final fp32 atomics are ~30 % of a launch of its own: 256 workgroups x 73 728 sums whatever the layer; grouped,
// the 256 workgroups are shared out by work and the atomics are paid once for all the layers)
struct C3WWGroup { C3WWArgs it[AVEC_WGRAD_GROUP_MAX]; int first[AVEC_WGRAD_GROUP_MAX + 1]; int n; };
__global__ __launch_bounds__(512) void wgrad3x3_wide_grouped_kernel(C3WWGroup grp) {
  int i = 0;
  while (i + 1 < grp.n && (int)blockIdx.x >= grp.first[i + 1]) ++i;
  c3ww_body(grp.it[i], (int)blockIdx.x - grp.first[i], grp.first[i + 1] - grp.first[i]);
}

static bool c3_wide_geometry(int H, int W, int C, int& KP, int& RS, int& IT) {
  const int PW = W + 1;
  KP = (H * PW + 15) / 16 * 16; RS = (KP + 2 * PW + 2 + 15) / 16 * 16;
  if (C < 128 || C % 128 || C > 1024 || H < 1 || W < 2 || 2 * KP > C3X_KROWS || 2 * RS > C3X_ROWS || (H + 2) * PW + 1 > RS || H * W * C >= (1 << 24)) return false;
  IT = C3X_ROWS / 2 / RS; if (C3X_KROWS / 2 / KP < IT) IT = C3X_KROWS / 2 / KP;            // images per half of the LDS ring
  if (IT > 64) IT = 64;
  return IT >= 1;
}

extern "C" int avec_wgrad3x3_c128_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride) {
  int KP, RS, IT;
  return Cin == Cout && KH == 3 && KW == 3 && stride == 1 && c3_wide_geometry(H, W, Cin, KP, RS, IT);
}

// wgrad_pairs.hip: the pair formulation (round 4) for the model's geometries (11x11 / 6x6 / 3x3 images); everything else stays on the slab kernel above
bool wgrad3x3_pairs_supported(int H, int W, int C);
int wgrad3x3_pairs_grouped(const avec_wgrad3x3_item_t* items, int n, hipStream_t st);

extern "C" int avec_wgrad3x3_c128(const void* x, const void* dy, float* dw, long long images, int C, int H, int W, hipStream_t st) {
  AVEC_CHECK_ARG(x && dy && dw && images > 0, "wgrad3x3_c128: null buffer");
  if (wgrad3x3_pairs_supported(H, W, C)) {
    avec_wgrad3x3_item_t it; it.x = x; it.dy = dy; it.dw = dw; it.images = images; it.C = C; it.H = H; it.W = W; it.reserved = 0;
    return wgrad3x3_pairs_grouped(&it, 1, st);
  }
  C3WWArgs a; a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.dw = dw; a.N = (int)images; a.H = H; a.W = W; a.C = C;
  AVEC_CHECK_ARG(c3_wide_geometry(H, W, C, a.KP, a.RS, a.IT), "wgrad3x3_c128: %d channels, %dx%d images do not fit the slabs", C, H, W);
  static bool attr_set = false;
  const size_t lds = (size_t)C3X_ROWS * 256 + C3X_KROWS * 128;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { avec_set_error("wgrad3x3_c128: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const int kinds = (C / 64) * (C / 128);
  const long long groups = (images + a.IT - 1) / a.IT;
  long long per_kind = 256 / kinds; if (per_kind < 1) per_kind = 1; if (per_kind > groups) per_kind = groups;
  avec_note_kernel("wgrad3x3_wide_kernel");
  hipLaunchKernelGGL(wgrad3x3_wide_kernel, dim3((unsigned)(per_kind * kinds)), dim3(512), lds, st, a);
  AVEC_LAUNCH_CHECK();
  return 0;
}

extern "C" int avec_wgrad3x3_c64_grouped(const avec_wgrad3x3_item_t* items, int n, hipStream_t st) {
  AVEC_CHECK_ARG(items && n > 0 && n <= AVEC_WGRAD_GROUP_MAX, "wgrad3x3_c64_grouped: 1..%d items", AVEC_WGRAD_GROUP_MAX);
  C3WGroup g; g.n = n;
  long long total_images = 0;
  for (int i = 0; i < n; ++i) {
    const avec_wgrad3x3_item_t& t = items[i];
    AVEC_CHECK_ARG(t.x && t.dy && t.dw && t.images > 0 && t.C == 64, "wgrad3x3_c64_grouped: bad item %d", i);
    AVEC_CHECK_ARG(avec_conv3x3_c64_supported(t.H, t.W, 64, 64, 3, 3, 1) && t.H * (t.W + 1) <= C3W_KROWS, "wgrad3x3_c64_grouped: item %d: %dx%d images do not fit the slab", i, t.H, t.W);
    C3WArgs& a = g.it[i]; a.x = (const bf16*)t.x; a.dy = (const bf16*)t.dy; a.dw = t.dw; a.N = (int)t.images; a.H = t.H; a.W = t.W;
    total_images += t.images;
  }
  // one workgroup per CU (the two slabs fill the LDS): 256 workgroups shared out by image count, at least one per item
  int used = 0; g.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    long long w = (256 * items[i].images + total_images / 2) / total_images; if (w < 1) w = 1; if (w > items[i].images) w = items[i].images;
    if (used + w > 256 + n) w = 1;
    used += (int)w; g.first[i + 1] = used;
  }
  static bool attr_set = false;
  const size_t lds = C3_SBYTES + C3W_DBYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_c64_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { avec_set_error("wgrad3x3_c64_grouped: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  avec_note_kernel("wgrad3x3_c64_grouped_kernel");
  hipLaunchKernelGGL(wgrad3x3_c64_grouped_kernel, dim3((unsigned)used), dim3(512), lds, st, g);
  AVEC_LAUNCH_CHECK();
  return 0;
}

extern "C" int avec_wgrad3x3_c128_grouped(const avec_wgrad3x3_item_t* items, int n, hipStream_t st) {
  AVEC_CHECK_ARG(items && n > 0 && n <= AVEC_WGRAD_GROUP_MAX, "wgrad3x3_c128_grouped: 1..%d items", AVEC_WGRAD_GROUP_MAX);
  {   // the model's geometries go to the pair kernel (one launch), the rest to the slab kernel (another)
    avec_wgrad3x3_item_t fast[AVEC_WGRAD_GROUP_MAX], slow[AVEC_WGRAD_GROUP_MAX]; int nf = 0, ns = 0;
    for (int i = 0; i < n; ++i) {
      AVEC_CHECK_ARG(items[i].x && items[i].dy && items[i].dw && items[i].images > 0, "wgrad3x3_c128_grouped: null buffer in item %d", i);
      if (wgrad3x3_pairs_supported(items[i].H, items[i].W, items[i].C)) fast[nf++] = items[i]; else slow[ns++] = items[i];
    }
    if (nf) {
      if (int r = wgrad3x3_pairs_grouped(fast, nf, st)) return r;
      if (!ns) return 0;
      return avec_wgrad3x3_c128_grouped(slow, ns, st);
    }
  }
  C3WWGroup g; g.n = n;
  long long cost[AVEC_WGRAD_GROUP_MAX]; int kinds[AVEC_WGRAD_GROUP_MAX], nwg[AVEC_WGRAD_GROUP_MAX], total = 0;
  for (int i = 0; i < n; ++i) {
    const avec_wgrad3x3_item_t& t = items[i];
    AVEC_CHECK_ARG(t.x && t.dy && t.dw && t.images > 0, "wgrad3x3_c128_grouped: null buffer in item %d", i);
    C3WWArgs& a = g.it[i]; a.x = (const bf16*)t.x; a.dy = (const bf16*)t.dy; a.dw = t.dw; a.N = (int)t.images; a.H = t.H; a.W = t.W; a.C = t.C;
    AVEC_CHECK_ARG(c3_wide_geometry(t.H, t.W, t.C, a.KP, a.RS, a.IT), "wgrad3x3_c128_grouped: item %d: %d channels, %dx%d images do not fit the slabs", i, t.C, t.H, t.W);
    kinds[i] = (t.C / 64) * (t.C / 128);
    const long long halves = (t.images + a.IT - 1) / a.IT;
    cost[i] = halves * (long long)(a.IT * (a.KP >> 4) + 2);              // steps of one kind's reduction (+ the per-half barrier / DMA issue)
    nwg[i] = kinds[i]; total += kinds[i];
  }
  AVEC_CHECK_ARG(total <= 4096, "wgrad3x3_c128_grouped: too many tiles");
  // one workgroup per CU (the slabs fill the LDS): hand the remaining workgroups to the item with the longest per-workgroup reduction, a whole set of kinds at a time
  const int budget = 256;
  for (;;) {
    int best = -1; double worst = 0.0;
    for (int i = 0; i < n; ++i) {
      const long long halves = (g.it[i].N + g.it[i].IT - 1) / g.it[i].IT;
      if (total + kinds[i] > budget || nwg[i] / kinds[i] >= halves) continue;
      const double load = (double)cost[i] / (double)(nwg[i] / kinds[i]);
      if (load > worst) { worst = load; best = i; }
    }
    if (best < 0) break;
    nwg[best] += kinds[best]; total += kinds[best];
  }
  g.first[0] = 0; for (int i = 0; i < n; ++i) g.first[i + 1] = g.first[i] + nwg[i];
  static bool attr_set = false;
  const size_t lds = (size_t)C3X_ROWS * 256 + C3X_KROWS * 128;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_wide_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { avec_set_error("wgrad3x3_c128_grouped: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  avec_note_kernel("wgrad3x3_wide_grouped_kernel");
  hipLaunchKernelGGL(wgrad3x3_wide_grouped_kernel, dim3((unsigned)total), dim3(512), lds, st, g);
  AVEC_LAUNCH_CHECK();
  return 0;
}

extern "C" int avec_conv3x3_c64_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride) {
  return Cin == 64 && Cout == 64 && KH == 3 && KW == 3 && stride == 1 && H * W <= 512 && (H + 2) * (W + 1) + 1 <= C3_MAXPIX && H >= 1 && W >= 2;
}

static int conv3x3_c64_impl(const void* x, const void* w, void* y, const void* res, const unsigned char* rmask, float* stats, long long images, int H, int W, int flip, hipStream_t st);
extern "C" int avec_conv3x3_c64(const void* x, const void* w, void* y, const void* res, float* stats, long long images, int H, int W, int flip, hipStream_t st) {
  return conv3x3_c64_impl(x, w, y, res, nullptr, stats, images, H, W, flip, st);
}
extern "C" int avec_conv3x3_c64_res_masked(const void* x, const void* w, void* y, const void* res, const unsigned char* res_mask, long long images, int H, int W, int flip, hipStream_t st) {
  AVEC_CHECK_ARG(res && res_mask, "conv3x3_c64_res_masked: null residual / mask");
  return conv3x3_c64_impl(x, w, y, res, res_mask, nullptr, images, H, W, flip, st);
}
static int conv3x3_c64_impl(const void* x, const void* w, void* y, const void* res, const unsigned char* rmask, float* stats, long long images, int H, int W, int flip, hipStream_t st) {
  AVEC_CHECK_ARG(x && w && y && images > 0, "conv3x3_c64: null buffer");
  AVEC_CHECK_ARG(avec_conv3x3_c64_supported(H, W, 64, 64, 3, 3, 1), "conv3x3_c64: %dx%d images do not fit the slab", H, W);
  static bool attr_set = false;
  const size_t lds = C3_WBYTES + C3_SBYTES;
  if (!attr_set) {
    const void* kerns[3] = {(const void*)conv3x3_c64_kernel<false, false>, (const void*)conv3x3_c64_kernel<true, false>, (const void*)conv3x3_c64_res_kernel};
    for (const void* k : kerns) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { avec_set_error("conv3x3_c64: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)e; }
    }
    attr_set = true;
  }
  static const int wgs_env = 256;
  C3Args a; a.x = (const bf16*)x; a.w = (const bf16*)w; a.y = (bf16*)y; a.res = (const bf16*)res; a.stats = stats; a.N = (int)images; a.H = H; a.W = W; a.flip = flip; a.rmask = rmask;
  const int grid = (int)(images < wgs_env ? images : wgs_env);
  avec_note_kernel(stats ? "conv3x3_c64_kernel<true,false>" : res ? "conv3x3_c64_res_kernel" : "conv3x3_c64_kernel<false,false>");
  // (statistics AND a residual in one launch would need 36 + 32 + 64 registers beside the accumulators: it spills, and a spilled register of an in-flight asm load is
  // wrong code -- no caller needs the pair: forward launches carry statistics, backward-data launches the residual)
  AVEC_CHECK_ARG(!(stats && res), "conv3x3_c64: statistics and a residual in the same launch are not supported");
  if (stats) hipLaunchKernelGGL((conv3x3_c64_kernel<true, false>), dim3(grid), dim3(512), lds, st, a);
  else if (res) hipLaunchKernelGGL(conv3x3_c64_res_kernel, dim3(grid), dim3(512), lds, st, a);
  else hipLaunchKernelGGL((conv3x3_c64_kernel<false, false>), dim3(grid), dim3(512), lds, st, a);
  AVEC_LAUNCH_CHECK();
  return 0;
}

extern "C" int avec_wgrad3x3_c64(const void* x, const void* dy, float* dw, long long images, int H, int W, hipStream_t st) {
  AVEC_CHECK_ARG(x && dy && dw && images > 0, "wgrad3x3_c64: null buffer");
  AVEC_CHECK_ARG(avec_conv3x3_c64_supported(H, W, 64, 64, 3, 3, 1) && H * (W + 1) <= C3W_KROWS, "wgrad3x3_c64: %dx%d images do not fit the slab", H, W);
  static bool attr_set = false;
  const size_t lds = C3_SBYTES + C3W_DBYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { avec_set_error("wgrad3x3_c64: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  C3WArgs a; a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.dw = dw; a.N = (int)images; a.H = H; a.W = W;
  const int grid = (int)(images < 256 ? images : 256);
  avec_note_kernel("wgrad3x3_c64_kernel");
  hipLaunchKernelGGL(wgrad3x3_c64_kernel, dim3(grid), dim3(512), lds, st, a);
  AVEC_LAUNCH_CHECK();
  return 0;
}
