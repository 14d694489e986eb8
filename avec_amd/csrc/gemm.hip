#include "gemm_dev.h"

template <typename T, int BM, int BN, int MODE, bool SRC_F32, bool A16>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  constexpr int VEC = Elt<T>::VEC;
  constexpr int KE = BKB / (int)sizeof(T);   // K elements per tile
  constexpr int NCA = BM * 8 / 256, NCB = BN * 8 / 256;
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int TILE = (BM + BN) * LDS_ROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long m0 = (long long)blockIdx.x * BM; const int n0 = blockIdx.y * BN;
  const int kc = (tid & 7) * VEC;            // this thread's K offset inside a tile
  const int r0 = tid >> 3;                   // first tile row handled by this thread (+32 per extra chunk)

  RowInfo ra[NCA]; RowInfo rb[NCB];
  RowSrc ws; ws.ptr = g.W; ws.ld = g.ldw; ws.step = 0; ws.rows_out = ws.rows_in = 1;
#pragma unroll
  for (int i = 0; i < NCA; ++i) ra[i] = row_info<MODE>(g.a, m0 + r0 + i * 32, g.M);
#pragma unroll
  for (int i = 0; i < NCB; ++i) rb[i] = row_info<MODE_PLAIN>(ws, n0 + r0 + i * 32, g.N);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Pend pa[NCA], pb[NCB];
  const int KT = (g.K + KE - 1) / KE;
  auto issue = [&](int kt) {
    const int k = kt * KE + kc;
    if (MODE == MODE_PLAIN) {
#pragma unroll
      for (int i = 0; i < NCA; ++i) pa[i] = issue_load<T, SRC_F32, A16>(g.a.ptr, ra[i].valid ? ra[i].base + k : -1, k, g.K);
    } else {
      int tap, c;
      if (g.a.C % KE == 0) { const int k0 = kt * KE; tap = k0 / g.a.C; c = k0 - tap * g.a.C + kc; }   // whole K-step inside one tap: wave-uniform
      else { tap = k / g.a.C; c = k - tap * g.a.C; }
      const int kh = tap / g.a.KW, kw = tap - kh * g.a.KW;
#pragma unroll
      for (int i = 0; i < NCA; ++i) pa[i] = issue_load<T, false, A16>(g.a.ptr, conv_offset<MODE == MODE_PLAIN ? MODE_CONV_FWD : MODE>(g.a, ra[i], kh, kw, c), k, g.K);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) pb[i] = issue_load<T, false, A16>(g.W, rb[i].valid ? rb[i].base + k : -1, k, g.K);
  };
  auto stage = [&](int buf) {
    char* As = smem + buf * TILE; char* Bs = As + BM * LDS_ROW;
#pragma unroll
    for (int i = 0; i < NCA; ++i) *(chunk16*)(As + (r0 + i * 32) * LDS_ROW + (tid & 7) * 16) = finish_load<T, SRC_F32 && MODE == MODE_PLAIN>(pa[i]);
#pragma unroll
    for (int i = 0; i < NCB; ++i) *(chunk16*)(Bs + (r0 + i * 32) * LDS_ROW + (tid & 7) * 16) = finish_load<T, false>(pb[i]);
  };

  const int frag_off = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
  issue(0);
  stage(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) issue(kt + 1);                                        // loads in flight during the MFMAs below
    mma_tile<T, BM, BN, MT, NT>(smem + cur * TILE, smem + cur * TILE + BM * LDS_ROW, wm, wn, frag_off, acc);
    if (kt + 1 < KT) stage(cur ^ 1);                                       // other buffer: last read two barriers ago
    __syncthreads();
  }

  nt_epilogue<T, BM, BN, MT, NT>(g, acc, smem, m0, n0, tid, lane, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// NT kernel, LDS-DMA variant (global_load_lds, 16 B per lane): the tile goes HBM/L2 -> LDS without a VGPR round trip, without
// ds_write and without per-chunk finishing VALU.  The DMA writes lane-linearly (wave-uniform base + lane*16), so LDS rows are
// unpadded 128-byte rows and bank conflicts are removed by an XOR swizzle applied on the SOURCE side: the lane that fills physical
// 16-byte slot p of tile row r fetches logical K-chunk p ^ ((r >> 1) & 7); the MFMA fragment reads apply the same involution
// (conflict-free for ds_read_b128's 16-lane groups).  Invalid rows / taps / K-tails fetch 16 zero bytes from `avec_zero16`.
// Requires every chunk address 16-byte aligned and K % VEC == 0 (host-checked: the A16 case).
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(64))) unsigned char avec_zero16[64];


// RB: bytes of K per LDS row (128: 8 chunks, swizzle (row>>1)&7;  64: 4 chunks, swizzle (row>>2)&3 -- half the ring, twice the resident workgroups)
template <typename T, int BM, int BN, int MODE, int STAGES, bool FASTC = false, int RB = 128>
__global__ __launch_bounds__(256, (BM * BN > 128 * 256) ? 1 : 2) void gemm_nt_glds_kernel(GemmArgs g) {
  static_assert(STAGES >= 2 && STAGES <= 8, "ring depth");
  constexpr int VEC = Elt<T>::VEC;
  constexpr int KE = RB / (int)sizeof(T);
  constexpr int CPR = RB / 16, RPP = 256 / CPR;      // chunks per row, rows per DMA pass
  constexpr int NCA = BM * CPR / 256, NCB = BN * CPR / 256;
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int TILE = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  constexpr bool PERM = MODE == MODE_CONV_BWD && FASTC;        // parity-class order possible (g.perm2 says whether it is on)
  const bool perm = PERM && g.perm2;
  const int cls = perm ? ((int)blockIdx.x >= g.pTs[2] ? ((int)blockIdx.x >= g.pTs[3] ? 3 : 2) : ((int)blockIdx.x >= g.pTs[1] ? 1 : 0)) : 0;
  const long long m0 = perm ? (long long)((int)blockIdx.x - g.pTs[cls]) * BM : (long long)blockIdx.x * BM; const int n0 = blockIdx.y * BN;
  const long long pMc = perm ? perm2_count(g.a, cls, g.pImgs) : 0;
  // chunk c = tid + i*256 of a tile -> row c>>3, physical slot c&7; it carries logical K-chunk (c&7) ^ ((row>>1)&7)
  RowInfo ra[NCA]; RowInfo rb[NCB]; int ka[NCA], kb[NCB];
  RowSrc ws; ws.ptr = g.W; ws.ld = g.ldw; ws.step = 0; ws.rows_out = ws.rows_in = 1;
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    const int row = tid / CPR + i * RPP;
    if (perm) {
      RowInfo r; r.valid = m0 + row < pMc; r.base = 0; r.a = 0; r.b = 0;
      if (r.valid) { long long img; int ih, iw; perm2_pixel(g.a, cls, m0 + row, img, ih, iw); r.base = img * (long long)g.a.OH * g.a.OW * g.a.C; r.a = ih + g.a.pad; r.b = iw + g.a.pad; }
      ra[i] = r;
    } else ra[i] = row_info<MODE>(g.a, m0 + row, g.M);
    ka[i] = ((tid % CPR) ^ glds_swz<RB>(row)) * VEC;
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) { const int row = tid / CPR + i * RPP; rb[i] = row_info<MODE_PLAIN>(ws, n0 + row, g.N); kb[i] = ((tid % CPR) ^ glds_swz<RB>(row)) * VEC; }
  // fast convolution addressing (host-checked: C % KE == 0 so a K-step lies inside one tap, <= 32 taps, < 2^31 source elements, backward
  // with stride 1 or 2): per row a 32-bit origin offset and a bit mask of the taps that fall inside the image; per K-step the tap and its
  // offset are wave-uniform, so a chunk address costs an add and a select instead of the bounds arithmetic of conv_offset().
  int rofs[NCA]; unsigned rmask[NCA];
  constexpr bool fast = MODE != MODE_PLAIN && FASTC;
  if (fast) {
    const int IW = MODE == MODE_CONV_FWD ? g.a.W : g.a.OW;      // row pitch of the source image
    // backward with stride 2: tap kh is valid iff (a - kh) is even and then oh = (a - kh)/2 = (a >> 1) - (kh >> 1): still linear in the tap
    const int sh = (MODE == MODE_CONV_BWD && g.a.stride == 2) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      rofs[i] = (int)ra[i].base + ((ra[i].a >> sh) * IW + (ra[i].b >> sh)) * g.a.C + ka[i];
      unsigned mk = 0u;
      for (int kh = 0; kh < g.a.KH; ++kh)
        for (int kw = 0; kw < g.a.KW; ++kw) {
          bool ok;
          if (MODE == MODE_CONV_FWD) { const int y = ra[i].a + kh, x = ra[i].b + kw; ok = y >= 0 && x >= 0 && y < g.a.H && x < g.a.W; }
          else { const int ty = ra[i].a - kh, tx = ra[i].b - kw; ok = ty >= 0 && tx >= 0 && !((ty | tx) & sh) && (ty >> sh) < g.a.OH && (tx >> sh) < g.a.OW; }
          mk |= ((ok && ra[i].valid) ? 1u : 0u) << (kh * g.a.KW + kw);
        }
      rmask[i] = mk;
    }
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  int KT = (g.K + KE - 1) / KE;
  int f_tap = 0, f_kh = 0, f_kw = 0, f_c0 = 0;       // fast path: running (tap, kh, kw, channel offset) of the next K-step to be issued (issue() is called with kt = 0, 1, 2, ...)
  unsigned tapmask = 0xffffffffu;                    // parity classes: the taps this tile's class can reach (wave-uniform); the others are skipped
  if (perm) {
    tapmask = 0u; int ntap = 0;
    for (int kh = 0; kh < g.a.KH; ++kh)
      for (int kw = 0; kw < g.a.KW; ++kw)
        if (!((((cls >> 1) + g.a.pad - kh) | ((cls & 1) + g.a.pad - kw)) & 1)) { tapmask |= 1u << (kh * g.a.KW + kw); ++ntap; }
    KT = ntap * (g.a.C / KE);
    while (f_tap < g.a.KH * g.a.KW && !((tapmask >> f_tap) & 1u)) { ++f_tap; if (++f_kw >= g.a.KW) { f_kw = 0; ++f_kh; } }
  }
  auto issue = [&](int kt, int buf) {
    if (AVEC_ABL & 2) return;
    char* As = smem + buf * TILE; char* Bs = As + BM * RB;
    if (fast) {
      const int IW = MODE == MODE_CONV_FWD ? g.a.W : g.a.OW;
      const int sh = (MODE == MODE_CONV_BWD && g.a.stride == 2) ? 1 : 0;
      const int tapoff = (MODE == MODE_CONV_FWD ? (f_kh * IW + f_kw) : -((f_kh >> sh) * IW + (f_kw >> sh))) * g.a.C + f_c0;
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        const T* sp = (const T*)g.a.ptr + (rofs[i] + tapoff);
        const void* src = ((rmask[i] >> f_tap) & 1u) ? (const void*)sp : (const void*)avec_zero16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (i * 256 + wave * 64) * 16), 16, 0, 0);
      }
    } else {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int k = kt * KE + ka[i];
      long long off;
      if (MODE == MODE_PLAIN) {
        off = (ra[i].valid && k < g.K) ? ra[i].base + k : -1;
        // K = 8n + 4: the last chunk of a row runs 8 bytes into the next row (fixed up in LDS below); the last row of the matrix re-reads its own last 16 bytes instead
        if (g.ktail && off >= 0 && k + VEC > g.K && m0 + (tid / CPR + i * RPP) == g.M - 1) off = ra[i].base + g.K - VEC;
      }
      else {
        int tap, c;
        if (g.a.C % KE == 0) { const int k0 = kt * KE; tap = k0 / g.a.C; c = k0 - tap * g.a.C + ka[i]; } else { tap = k / g.a.C; c = k - tap * g.a.C; }
        const int kh = tap / g.a.KW, kw = tap - kh * g.a.KW;
        off = (k < g.K) ? conv_offset<MODE == MODE_PLAIN ? MODE_CONV_FWD : MODE>(g.a, ra[i], kh, kw, c) : -1;
      }
      const void* src = off >= 0 ? (const void*)((const T*)g.a.ptr + off) : (const void*)avec_zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
    }
    const int kbase = fast ? f_tap * g.a.C + f_c0 : kt * KE;       // (fast: the K offset follows the tap actually issued)
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int k = kbase + kb[i];
      long long boffs = rb[i].base + k;
      if (MODE == MODE_PLAIN && g.ktail && k + VEC > g.K && n0 + (tid / CPR + i * RPP) == g.N - 1) boffs = rb[i].base + g.K - VEC;
      const void* src = (rb[i].valid && k < g.K) ? (const void*)((const T*)g.W + boffs) : (const void*)avec_zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
    if (fast) {
      f_c0 += KE;
      if (f_c0 >= g.a.C) {
        f_c0 = 0;
        do { ++f_tap; if (++f_kw >= g.a.KW) { f_kw = 0; ++f_kh; } } while (PERM && f_tap < 32 && !((tapmask >> f_tap) & 1u));
      }
    }
  };
  // fragment addressing: lane (row = lane&31 within a 32-row block, k-half g = lane>>5), K-substep kk: logical chunk 2*kk + g
  int offa[MT], swa[MT], offb[NT], swb[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) { const int row = wm * (BM / 2) + i * 32 + (lane & 31); offa[i] = row * RB; swa[i] = glds_swz<RB>(row); }
#pragma unroll
  for (int j = 0; j < NT; ++j) { const int row = wn * (BN / 2) + j * 32 + (lane & 31); offb[j] = row * RB; swb[j] = glds_swz<RB>(row); }
  const int gsel = lane >> 5;

  // STAGES-deep ring of LDS buffers, counted vmcnt waits, ONE raw barrier per K-step (the DMA stays in flight across barriers):
  //   wait(tile kt landed: only the newer S-2 tiles may still be outstanding) -> barrier -> issue tile kt+S-1 into the buffer everybody
  //   finished reading before this barrier -> MFMAs on tile kt.
  constexpr int LPT = NCA + NCB;             // DMA instructions per thread per tile
#define AVEC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st) if (st < KT) issue(st, st);
  for (int kt = 0; kt < KT; ++kt) {
    const int newer = min(STAGES - 2, KT - 1 - kt);     // tiles issued after kt that may still be in flight
    if (STAGES <= 2 || newer <= 0) AVEC_WAIT_VM(0);
    else if (newer == 1) AVEC_WAIT_VM(LPT);
    else if (newer == 2) AVEC_WAIT_VM(2 * LPT);
    else if (newer == 3) AVEC_WAIT_VM(3 * LPT);
    else if (newer == 4) AVEC_WAIT_VM(4 * LPT);
    else if (newer == 5) AVEC_WAIT_VM(5 * LPT);
    else AVEC_WAIT_VM(6 * LPT);
    if (!(AVEC_ABL & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < KT) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    if (MODE == MODE_PLAIN && sizeof(T) == 2 && g.ktail && kt == KT - 1) {
      // the chunk that holds elements K-4 .. K-1 also holds 4 elements of the next row: zero them (A rows, then W rows; the last row of each matrix was
      // fetched 8 bytes early: move its upper half down first).  Nothing is in flight any more (vmcnt(0) above).
      char* tile = smem + (kt % STAGES) * TILE;
      const int ctail = ((g.K - 4) % KE) / VEC;
      if (tid < BM + BN) {
        const int row = tid < BM ? tid : tid - BM;
        char* p = tile + (tid < BM ? 0 : BM * RB) + row * RB + ((ctail ^ glds_swz<RB>(row)) << 4);
        const bool last = tid < BM ? (m0 + row == g.M - 1) : (n0 + row == g.N - 1);
        if (last) *(uint2*)p = *(const uint2*)(p + 8);
        *(uint2*)(p + 8) = make_uint2(0u, 0u);
      }
      __syncthreads();
    }
    const char* As = smem + (kt % STAGES) * TILE; const char* Bs = As + BM * RB;
    // all operand fragments of the step are requested before the first MFMA (small tiles otherwise wait one LDS latency per MFMA:
    // the compiler keeps the reads next to their consumer); at most 2 K-substeps of fragments are held at a time for the big tiles
    constexpr int KK = RB / 32, KG = (MT * NT == 1) ? KK : (KK < 2 ? KK : 2);
#pragma unroll
    for (int k0 = 0; k0 < KK; k0 += KG) {
      chunk16 fa[KG][MT], fb[KG][NT];
#pragma unroll
      for (int q = 0; q < KG; ++q) {
#pragma unroll
        for (int i = 0; i < MT; ++i) { if (AVEC_ABL & 4) { fa[q][i].w[0] = fa[q][i].w[1] = fa[q][i].w[2] = fa[q][i].w[3] = kt + i; } else fa[q][i] = *(const chunk16*)(As + offa[i] + ((((k0 + q) * 2 + gsel) ^ swa[i]) << 4)); }
#pragma unroll
        for (int j = 0; j < NT; ++j) { if (AVEC_ABL & 4) { fb[q][j].w[0] = fb[q][j].w[1] = fb[q][j].w[2] = fb[q][j].w[3] = kt + j; } else fb[q][j] = *(const chunk16*)(Bs + offb[j] + ((((k0 + q) * 2 + gsel) ^ swb[j]) << 4)); }
      }
      asm volatile("" ::: "memory");            // keeps the reads above the MFMAs (the scheduler otherwise sinks each pair next to its consumer)
      // ... and makes every fragment opaque HERE: a register-only consumer (the MFMA) may be hoisted above a memory fence, which left one LDS round trip
      // per K-substep in the 64x64 kernel (read, wait, MFMA, read, wait, MFMA); with the values pinned the waits become one counted ladder per group
#pragma unroll
      for (int q = 0; q < KG; ++q) {
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[q][i].w[0]), "+v"(fa[q][i].w[1]), "+v"(fa[q][i].w[2]), "+v"(fa[q][i].w[3]));
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[q][j].w[0]), "+v"(fb[q][j].w[1]), "+v"(fb[q][j].w[2]), "+v"(fb[q][j].w[3]));
      }
#pragma unroll
      for (int q = 0; q < KG; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (AVEC_ABL & 1) { asm volatile("" :: "v"(fa[q][i].w[0]), "v"(fa[q][i].w[1]), "v"(fa[q][i].w[2]), "v"(fa[q][i].w[3]), "v"(fb[q][j].w[0]), "v"(fb[q][j].w[1]), "v"(fb[q][j].w[2]), "v"(fb[q][j].w[3])); }
            else Mma<T>::run(fa[q][i], fb[q][j], acc[i][j]);
          }
    }
  }
#undef AVEC_WAIT_VM
  __syncthreads();                            // every wave is done with the ring before the epilogue reuses the LDS
  nt_epilogue<T, BM, BN, MT, NT>(g, acc, smem, m0, n0, tid, lane, wm, wn, cls);
}


// ------------------------------------------------------------------------------------------------
// 3x3 / stride-1 / pad-1 convolution (forward and backward-data), bf16, as a SHIFTED-WINDOW implicit GEMM.
// The tile rows are BM consecutive pixels p of the NHWC tensor; tap (kh, kw) of pixel p reads pixel p + s, s = (kh-1)*W + (kw-1)
// (backward-data: p - s), so all nine A tiles of one 32-channel chunk are shifted views of ONE window of BM + 2(W+1) pixels.  The
// window is fetched once per chunk (LDS-DMA, double-buffered) instead of once per tap: the A operand crosses L2 -> LDS 1.2x instead
// of 9x (the im2col gather of gemm_nt_glds_kernel: ablation showed that kernel bound by the LDS-DMA, 135-160 us of loads alone for
// 95-103 us of MFMA + fragment reads).  Taps that fall outside the image are removed by a per-row 9-bit mask applied to the A
// fragments (4 v_cndmask per fragment).  K order = chunk-major, tap-minor; B tiles ([BN][32 channels] of tap t) ride a 3-stage ring.
// ------------------------------------------------------------------------------------------------

// Round 3: the loop is bound by instruction ISSUE, not by a data path.  Ablation (512-channel stage, 147 us): without MFMA, LDS-DMA and fragment reads the loop
// skeleton alone -- ~190 instructions per K-step (address arithmetic, tap bookkeeping, M0 traffic, ring-slot modulo, branches) -- took 72 us, the MFMAs alone
// need 63, and the two ADD: a wave issues one instruction per 4-cycle slot whatever its type, and two waves per SIMD do not hide each other's scalar and vector
// bookkeeping under the MFMAs.  So the nine taps of a channel chunk are unrolled (two chunks per trip for the window parity): tap, ring slot, window buffer and
// fragment offsets are compile-time, every ds_read_b128 takes one of 18 + 2 precomputed addresses plus an immediate, the DMA sources are a scalar base + a
// per-lane 32-bit offset fixed at entry, one M0 save / restore per DMA group.  Rows outside the tensor are clamped instead of redirected to a zero page: every tap
// that could read them is masked (it lies outside its image), and tile rows / columns beyond M / N are never stored.
template <int BM, int BN, int MODE, bool TR = false>      // TR: transposed product (weights as the MFMA A operand) + register-direct epilogue (conv_epilogue_tr)
__global__ __launch_bounds__(256, 2) void conv3x3_shift_kernel(GemmArgs g) {
  typedef bf16 T;
  constexpr int RB = 64, KE = 32, STAGES = 3;
  constexpr int NCB = BN / 64, NA = (BM + 64) / 64;           // DMA passes (64 rows of 64 B each) per B tile / per A window
  constexpr int WROWS = BM + 64;                              // window rows: BM + 2 * halo, halo = W + 1 <= 32
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int BTILE = BN * RB, AWIN = WROWS * RB;
  constexpr int SGN = MODE == MODE_CONV_FWD ? 1 : -1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Bring = smem; char* const Awin = smem + STAGES * BTILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // Tile order (g.pTs[0] = 1 for this kernel, which has no parity classes: XCD-aware).  In launch order (row tile fastest, workgroups dealt round-robin to the 8 XCDs) the column tiles of a row tile run a
  // whole grid column apart, on any XCD: every one re-fetches its window from the fabric (stage 3: 2 x, stage 4: 4 x the activation), and every XCD streams every weight
  // tile.  Here XCD x takes a contiguous range of logical ids; inside a group of two column tiles the column is the fastest index, so a window is fetched by ONE L2 (the
  // partner tile follows on the same XCD) and the two weight tiles of the group (<= 2.4 MB) stay resident in it.
  int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  if (g.pTs[0] && gridDim.y > 1 && !(gridDim.y & 1)) {
    const int lid = xcd_logical((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int per = (int)gridDim.x * 2, cg = lid / per, rem = lid - cg * per;
    bx = rem >> 1; by = cg * 2 + (rem & 1);
  }
  const long long m0 = (long long)bx * BM; const int n0 = by * BN;
  const int Wd = g.a.W, H = g.a.H, C = g.a.C, halo = Wd + 1;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // DMA plan: window row wr = i*64 + tid/4 holds source pixel m0 - halo + wr (clamped into the tensor); physical slot tid%4 carries logical chunk slot ^ swz(row)
  unsigned aoff[NA], boff[NCB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int wr = i * 64 + (tid >> 2); long long p = m0 - halo + wr;
    p = p < 0 ? 0 : (p >= g.M ? g.M - 1 : p);
    aoff[i] = (unsigned)((p * C + (((tid & 3) ^ glds_swz<RB>(wr)) * 8)) * 2);
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int row = i * 64 + (tid >> 2); const int n = n0 + row < g.N ? n0 + row : g.N - 1;
    boff[i] = (unsigned)(((long long)n * g.ldw + (((tid & 3) ^ glds_swz<RB>(row)) * 8)) * 2);
  }
  const unsigned bring0 = (unsigned)(uintptr_t)(lptr_t)Bring, awin0 = (unsigned)(uintptr_t)(lptr_t)Awin;
  const unsigned wslot = (unsigned)wave * 1024u;
  const unsigned zaddr = (awin0 + 2 * AWIN + 255u) & ~255u;   // zero blocks (256 B each, 256-aligned) at zaddr + 2048 i behind the two windows
  if (tid < 16 * MT) { *(uint4*)(smem + (zaddr - bring0) + (tid >> 4) * 2048 + (tid & 15) * 16) = make_uint4(0u, 0u, 0u, 0u); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  const int C2 = C * 2;                                       // bytes between the B tiles of consecutive taps

  // fragment rows and the taps each of them may use
  unsigned amask[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int r = wm * (BM / 2) + i * 32 + (lane & 31);
    const long long p = m0 + r;
    unsigned mk = 0u;
    if (p < g.M) {
      const int x = (int)(p % Wd), y = (int)((p / Wd) % H);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int yy = y + SGN * (kh - 1), xx = x + SGN * (kw - 1);
          if (yy >= 0 && yy < H && xx >= 0 && xx < Wd) mk |= 1u << (kh * 3 + kw);
        }
    }
    amask[i] = mk;
  }
  // LDS addresses of the fragment reads: A per (tap, K-substep) -- fragment i adds 2048 i, the second window buffer AWIN (rows 32 apart keep the swizzle
  // (w >> 2) & 3); B per K-substep -- fragment j adds 2048 j, the ring slot BTILE * slot
  const int gsel = lane >> 5;
  unsigned aad[9][2], bad[2];
  { const int arow0 = wm * (BM / 2) + (lane & 31) + halo;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int w0 = arow0 + SGN * ((t / 3 - 1) * Wd + (t % 3 - 1));
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
#define AVEC_WAIT_VM(n) do { if (!(AVEC_ABL & 16)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); } while (0)
  // issue order: A(0), B(0), B(1); then in step ks (after its barrier): [A(cc+1) when t == 0], B(ks+2)
  if (!(AVEC_ABL & (2 | 64))) glds16_group<NA>(aoff, Ab, awin0 + wslot);
  if (!(AVEC_ABL & (2 | 128))) { glds16_group<NCB>(boff, Wb, bring0 + wslot); glds16_group<NCB>(boff, Wb + C2, bring0 + BTILE + wslot); }

  // one K-step: tap TAP of chunk cc, whose window sits in buffer PAR
  auto step = [&](auto tapc, auto parc, const int cc) {
    constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value;
    const bool last_chunk = cc + 1 >= NC;
    if (TAP == 8) { if (last_chunk) AVEC_WAIT_VM(0); else AVEC_WAIT_VM(NCB); }
    else if (TAP == 1) { if (last_chunk) AVEC_WAIT_VM(NCB); else AVEC_WAIT_VM(NCB + NA); }       // the window of the next chunk went out in the previous step (newer than B(ks))
    else AVEC_WAIT_VM(NCB);
    if (!(AVEC_ABL & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // all fragment reads of the step are requested at once (inline asm: in-order returns, counted waits); the first MFMA group
    // waits only for its own K-substep while the second one's reads are still in flight
    u32x4 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (AVEC_ABL & 4) { for (int i = 0; i < MT; ++i) fa[q][i] = u32x4{(unsigned)cc, 1u, 2u, 3u}; for (int j = 0; j < NT; ++j) fb[q][j] = u32x4{(unsigned)cc, 5u, 6u, 7u}; continue; }
      // rows whose tap lies outside the image read zeros instead of masking the operand (4 selects per fragment + the wait states behind them): four 256-byte zero
      // blocks 2048 B apart behind the windows (fragment i adds 2048 i as an immediate), every lane at the bank group of its regular address (no extra conflicts).
      // The centre tap is never masked.  (The pin keeps the 72 loop-invariant selected addresses from being hoisted into registers the kernel does not have.)
      unsigned ar = aad[TAP][q] + (unsigned)(PAR * AWIN), az = (aad[TAP][q] & 0xF0u) | zaddr;
      asm volatile("" : "+v"(az));
#define AVEC_AADDR(i) ((TAP == 4 || (amask[i] & (1u << TAP))) ? ar : az)
      fa[q][0] = lds_read128o<0>(AVEC_AADDR(0)); fa[q][1] = lds_read128o<2048>(AVEC_AADDR(1));
      if (MT > 2) { fa[q][2 % MT] = lds_read128o<4096>(AVEC_AADDR(2 % MT)); fa[q][3 % MT] = lds_read128o<6144>(AVEC_AADDR(3 % MT)); }
#undef AVEC_AADDR
      fb[q][0] = lds_read128o<(TAP % 3) * BTILE>(bad[q]);
      if (NT > 1) fb[q][1 % NT] = lds_read128o<(TAP % 3) * BTILE + 2048>(bad[q]);
    }
    // the DMA of the tiles two steps ahead goes out while the fragment reads are in flight
    if (TAP == 0 && !last_chunk && !(AVEC_ABL & (2 | 64))) glds16_group<NA>(aoff, Ab + (long long)(cc + 1) * (KE * 2), awin0 + (PAR ^ 1) * AWIN + wslot);
    if ((TAP < 7 || !last_chunk) && !(AVEC_ABL & (2 | 128)))
      glds16_group<NCB>(boff, Wb + (long long)((TAP + 2) % 9) * C2 + (long long)(cc + (TAP >= 7 ? 1 : 0)) * (KE * 2), bring0 + ((TAP + 2) % 3) * BTILE + wslot);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (AVEC_ABL & 32) {} else            // (timing experiments only: the MFMAs then read whatever the registers hold)
      if (q == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[q][i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[q][j]));
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (AVEC_ABL & 1) asm volatile("" :: "v"(fa[q][i]), "v"(fb[q][j])); else
          if (TR) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[q][j]), __builtin_bit_cast(bf16x8_t, fa[q][i]), acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[q][i]), __builtin_bit_cast(bf16x8_t, fb[q][j]), acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);          // (keeps this group's MFMAs above the next group's wait)
    }
  };
  auto chunk = [&](auto parc, const int cc) {
    step(IntC<0>{}, parc, cc); step(IntC<1>{}, parc, cc); step(IntC<2>{}, parc, cc); step(IntC<3>{}, parc, cc); step(IntC<4>{}, parc, cc);
    step(IntC<5>{}, parc, cc); step(IntC<6>{}, parc, cc); step(IntC<7>{}, parc, cc); step(IntC<8>{}, parc, cc);
  };
#pragma unroll 1
  for (int cc = 0; cc < NC; cc += 2) {
    chunk(IntC<0>{}, cc);
    if (cc + 1 < NC) chunk(IntC<1>{}, cc + 1);
  }
#undef AVEC_WAIT_VM
  __syncthreads();
  if (TR) {
    long long row[MT]; bool valid[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { row[i] = m0 + wm * (BM / 2) + i * 32 + (lane & 31); valid[i] = row[i] < g.M; if (!valid[i]) row[i] = g.M - 1; }
    conv_epilogue_tr<BM, BN, MT, NT>(g, acc, smem, row, valid, (const bf16*)g.e.res, row, m0 + BM <= g.M, n0, tid, lane, wm, wn);
  } else nt_epilogue<T, BM, BN, MT, NT>(g, acc, smem, m0, n0, tid, lane, wm, wn);
}


// ------------------------------------------------------------------------------------------------
// Plain bf16 NT product with 64- or 128-row tiles x 64 columns: the conformer-sized products (a few hundred tiles, K = 256 .. 1440), which are bound by the
// instructions the K loop issues, not by a data path -- 64 x 64 tiles run 4 MFMAs per wave and K-step; the general kernel above spends ~150 instructions per
// step on them (bounds / tail / tap logic per DMA chunk, fragment addresses, ring-slot modulo).  Here the ring position is compile-time (steps unrolled by the
// ring depth), every fragment read is one of eight precomputed addresses plus an immediate, the DMA sources are a scalar base + per-lane 32-bit offsets fixed at
// entry; only the last, partial K tile takes per-lane selects (K % 8 == 0: a 16-byte chunk is inside K or not at all).  Rows beyond M / N are clamped (their
// results are not stored).  Same LDS image, swizzle and epilogue as gemm_nt_glds_kernel<bf16, BM, 64, MODE_PLAIN, 4, false, 128>.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16_v64(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
// STAGES = 4: all of K <= 256 in flight, two workgroups per CU (64 KB).  STAGES = 2 (AVEC_NT_S2): 32 KB, four workgroups per CU -- for products whose 64 x 64 tiling
// exceeds the 512 slots of the deep ring (3200 x 1024 x 256: 800 tiles) one round of workgroups that hide each other's DMA latency instead of two rounds.
template <int BM, int BN, int STAGES = 4, bool TR = false>      // TR: transposed product + register-direct epilogue (plain_epilogue_tr)
__global__ __launch_bounds__(256, 2) void gemm_nt_plain_kernel(const void* pa_ptr, const void* pa_w, long long pa_lda, long long pa_ldw, long long pa_M, int pa_N, int pa_K, int pa_ktail, GemmArgs g_unused) {
  // The leading arguments repeat the fields of `g` that the operand DMAs need (13 dwords): with the kernel-argument preload of the build they are in SGPRs when the
  // first wave starts, and the first tiles go out without waiting for the ~400-byte argument block (that block misses every cache between two
  // launches of the same kernel; where it lives is worth 1 us per launch).  Everything else is read from `g` as before.
  typedef bf16 T;
  constexpr int RB = 128, KE = 64;
  static_assert(STAGES == 2 || STAGES == 4, "ring of 2 or 4 K tiles");
  constexpr int NCA = BM / 32, NCB = BN / 32, LPT = NCA + NCB;         // DMA passes (32 rows x 128 B) per tile
  constexpr int MT = BM / 64, NT = BN / 64, KK = 4;                    // K-substeps of 16 per tile
  constexpr int TILE = (BM + BN) * RB;
  static_assert(NT == 1, "64-column tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroup ids go round-robin to the 8 XCDs (each with its own L2), so in launch order every XCD pulls ALL of A and W (PMC: 3.4x the
  // algorithmic bytes per launch).  XCD x takes a contiguous range of logical ids, column tile fastest: the tiles of one row block share their A rows in ONE L2.
#if AVEC_NT_XCD
  const int lid = xcd_logical((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const long long m0 = (long long)(lid / (int)gridDim.y) * BM; const int n0 = (lid % (int)gridDim.y) * BN;
#else
  const long long m0 = (long long)blockIdx.x * BM; const int n0 = blockIdx.y * BN;
#endif
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem, wslot = (unsigned)wave * 1024u;
  const int K = pa_K, KT = (K + KE - 1) / KE, KF = K / KE;
  // DMA plan: pass i, thread tid -> tile row tid/8 + 32 i, physical slot tid%8 carrying logical K-chunk (tid%8) ^ swz(row)
  unsigned aoff[NCA], boff[NCB]; int kca, kcb[NCB];
  unsigned lastrow = 0;                                              // bit i / bit 8 + i: DMA pass i of A / W fetches the LAST row of its matrix (K = 8n + 4 only, see below)
  kca = 0;
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    const int row = (tid >> 3) + i * 32; const long long m = m0 + row < pa_M ? m0 + row : pa_M - 1;
    const int kc = (tid & 7) ^ glds_swz<RB>(row);
    aoff[i] = (unsigned)((m * pa_lda + kc * 8) * 2);                 // (plain rows only: strided row maps take the general kernel)
    if (i == 0) kca = kc;                                            // (rows 32 apart share the swizzle)
    if (m == pa_M - 1) lastrow |= 1u << i;
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int row = (tid >> 3) + i * 32; const int n = n0 + row < pa_N ? n0 + row : pa_N - 1;
    kcb[i] = (tid & 7) ^ glds_swz<RB>(row);
    boff[i] = (unsigned)(((long long)n * pa_ldw + kcb[i] * 8) * 2);
    if (n == pa_N - 1) lastrow |= 256u << i;
  }
  // K = 8n + 4 (the 180- / 540-wide audio stage): the chunk that holds elements K-4 .. K-1 also holds 4 elements of the next row -- zeroed in LDS once the last tile has
  // landed (step()); the last row of a matrix fetches that chunk 8 bytes early (nothing is read behind the matrix) and its upper half is moved down first
  const bool ktail = pa_ktail != 0;
  const char* const Ab = (const char*)pa_ptr; const char* const Wb = (const char*)pa_w;
  auto issue = [&](const int kt, auto stagec) {
    constexpr int S = decltype(stagec)::value;
    if (AVEC_ABL & 2) return;
    const unsigned la = lds0 + S * TILE + wslot, lb = la + BM * RB;
    if (kt < KF) { glds16_group<NCA>(aoff, Ab + (long long)kt * (KE * 2), la); glds16_group<NCB>(boff, Wb + (long long)kt * (KE * 2), lb); }
    else {                                                           // the partial last tile: chunks at or beyond K come from the zero page
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        const int k0 = kt * KE + kca * 8; const long long early = (ktail && k0 + 8 > K && ((lastrow >> i) & 1u)) ? 8 : 0;
        glds16_v64(k0 < K ? (const void*)(Ab + aoff[i] + (long long)kt * (KE * 2) - early) : (const void*)avec_zero16, la + i * 4096);
      }
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        const int k0 = kt * KE + kcb[i] * 8; const long long early = (ktail && k0 + 8 > K && ((lastrow >> (8 + i)) & 1u)) ? 8 : 0;
        glds16_v64(k0 < K ? (const void*)(Wb + boff[i] + (long long)kt * (KE * 2) - early) : (const void*)avec_zero16, lb + i * 4096);
      }
    }
  };
  // fragment addresses inside a tile: K-substep q reads logical chunk 2 q + g of its row; A fragment i adds 4096 i (32 rows), the ring stage S * TILE
  const int gsel = lane >> 5;
  unsigned aad[2][KK], bad[2][KK];           // [ring stages 0-1 | 2-3]: the immediate offset field holds 16 bits
  { const int ra = wm * (BM / 2) + (lane & 31), rb = wn * (BN / 2) + (lane & 31);
#pragma unroll
    for (int q = 0; q < KK; ++q) {
      aad[0][q] = lds0 + (unsigned)(ra * RB + (((2 * q + gsel) ^ glds_swz<RB>(ra)) << 4));
      bad[0][q] = lds0 + (unsigned)(BM * RB + rb * RB + (((2 * q + gsel) ^ glds_swz<RB>(rb)) << 4));
      aad[1][q] = aad[0][q] + 2 * TILE; bad[1][q] = bad[0][q] + 2 * TILE;
    } }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

#define AVEC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  issue(0, IntC<0>{});
  if (STAGES > 2 && KT > 1) issue(1, IntC<1 % STAGES>{});
  if (STAGES > 2 && KT > 2) issue(2, IntC<2 % STAGES>{});
  // the argument block itself is read only now, through a pointer the compiler cannot see through: its scalar loads (and the wait for them, which the register
  // allocator otherwise drags to the top of the kernel with an SGPR spill) stay behind the DMAs above
  struct PlainKA { const void* a; const void* w; long long lda, ldw, M; int N, K, ktail; GemmArgs g; };
  typedef const __attribute__((address_space(4))) char* kseg_t;
  typedef const __attribute__((address_space(4))) GemmArgs* kgp_t;
  kgp_t gp = (kgp_t)((kseg_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(PlainKA, g));
  asm volatile("" : "+s"(gp));
  { const unsigned long long u = (unsigned long long)(uintptr_t)gp;      // (an asm output counts as divergent: say it is uniform, or the block is read by vector loads)
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    gp = (kgp_t)(uintptr_t)(((unsigned long long)hi << 32) | lo); }
  GemmArgs g;
  { static_assert(sizeof(GemmArgs) % 8 == 0, "copied as 64-bit words");
    typedef const __attribute__((address_space(4))) unsigned long long* kq_t;
    const kq_t q = (kq_t)gp; unsigned long long buf[sizeof(GemmArgs) / 8];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(GemmArgs) / 8); ++i) buf[i] = q[i];      // constant address space + uniform pointer: scalar loads
    __builtin_memcpy(&g, buf, sizeof(GemmArgs)); }
  auto step = [&](const int kt, auto stagec) {
    constexpr int S = decltype(stagec)::value;
    const int rem = KT - 1 - kt;                                     // tiles issued after kt: min(rem, STAGES - 2) may still be in flight
    if (STAGES > 2 && rem >= 2) AVEC_WAIT_VM(2 * LPT); else if (STAGES > 2 && rem == 1) AVEC_WAIT_VM(LPT); else AVEC_WAIT_VM(0);
    if (!(AVEC_ABL & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ktail && kt == KT - 1) {                                     // (workgroup-uniform; nothing is in flight: vmcnt(0) above)
      if (tid < BM + BN) {
        const int row = tid < BM ? tid : tid - BM; const int ctail = ((K - 4) % KE) >> 3;
        char* p = smem + S * TILE + (tid < BM ? 0 : BM * RB) + row * RB + ((ctail ^ glds_swz<RB>(row)) << 4);
        const bool last = tid < BM ? (m0 + row >= g.M - 1) : (n0 + row >= g.N - 1);
        if (last) *(uint2*)p = *(const uint2*)(p + 8);
        *(uint2*)(p + 8) = make_uint2(0u, 0u);
      }
      __syncthreads();
    }
    u32x4 fa[KK][MT], fb[KK];
#pragma unroll
    for (int q = 0; q < KK; ++q) {
      if (AVEC_ABL & 4) { for (int i = 0; i < MT; ++i) fa[q][i] = u32x4{(unsigned)kt, 1u, 2u, 3u}; fb[q] = u32x4{(unsigned)kt, 5u, 6u, 7u}; continue; }
      fa[q][0] = lds_read128o<(S & 1) * TILE>(aad[S >> 1][q]);
      if (MT > 1) fa[q][1 % MT] = lds_read128o<(S & 1) * TILE + 4096>(aad[S >> 1][q]);
      fb[q] = lds_read128o<(S & 1) * TILE>(bad[S >> 1][q]);
    }
    if (kt + STAGES - 1 < KT) issue(kt + STAGES - 1, IntC<(S + STAGES - 1) % STAGES>{});        // into the slot everybody finished reading before this barrier
#pragma unroll
    for (int q = 0; q < KK; ++q) {
      if (q == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * (MT + 1)) : "memory");
      else if (q == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (MT + 1)) : "memory");
      else if (q == 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + 1) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[q][i]));
      asm volatile("" : "+v"(fb[q]));
#pragma unroll
      for (int i = 0; i < MT; ++i)
        if (AVEC_ABL & 1) asm volatile("" :: "v"(fa[q][i]), "v"(fb[q])); else
        if (TR) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[q]), __builtin_bit_cast(bf16x8_t, fa[q][i]), acc[i][0], 0, 0, 0); else
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[q][i]), __builtin_bit_cast(bf16x8_t, fb[q]), acc[i][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll 1
  for (int kt = 0; kt < KT; kt += STAGES) {
    step(kt, IntC<0>{});
    if (kt + 1 < KT) step(kt + 1, IntC<1>{});
    if (STAGES > 2 && kt + 2 < KT) step(kt + 2, IntC<2 % STAGES>{});
    if (STAGES > 2 && kt + 3 < KT) step(kt + 3, IntC<3 % STAGES>{});
  }
#undef AVEC_WAIT_VM
  if constexpr (TR) {
    static_assert(BM == 64 && BN == 64, "register-direct epilogue: 64 x 64 tiles");
    plain_epilogue_tr(g, acc[0][0], m0, n0, lane, wm, wn);
  } else {
    __syncthreads();                          // every wave is done with the ring before the epilogue reuses the LDS
    nt_epilogue<T, BM, BN, MT, NT>(g, acc, smem, m0, n0, tid, lane, wm, wn, 0);
  }
}

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (forward, or backward-data incl. the parity-class order of stride-2 layers) with the loop of gemm_nt_plain_kernel: 128-row tiles,
// 64-byte LDS rows (32 channels per K-step, always inside one tap: C % 32 == 0), 3-stage ring unrolled so that the ring position is compile-time, fragment reads
// from four precomputed addresses + immediates, the weight tiles by scalar-base DMA; only the gathered A chunks keep per-lane work (64-bit row pointer + the
// tap's scalar offset, redirected to the zero page where the tap leaves the image).  The stride-2 and 1x1 layers of the ResNet (the general kernel spends ~110
// instructions per K-step of 8 MFMAs on them).  Host-checked like the fast path of gemm_nt_glds_kernel (fast_conv); same LDS image, swizzle and epilogue.
// ------------------------------------------------------------------------------------------------
template <int BN, int MODE, bool TR = false>      // TR: transposed product + register-direct epilogue, as in conv3x3_shift_kernel
__global__ __launch_bounds__(256, 2) void gemm_nt_conv_lean_kernel(GemmArgs g) {
  typedef bf16 T;
  constexpr int BM = 128, RB = 64, KE = 32, STAGES = 3, CPR = 4;
  constexpr int NCA = BM * CPR / 256, NCB = BN * CPR / 256, LPT = NCA + NCB;
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int TILE = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr bool PERM = MODE == MODE_CONV_BWD;
  const bool perm = PERM && g.perm2;
  const int cls = perm ? ((int)blockIdx.x >= g.pTs[2] ? ((int)blockIdx.x >= g.pTs[3] ? 3 : 2) : ((int)blockIdx.x >= g.pTs[1] ? 1 : 0)) : 0;
  const long long m0 = perm ? (long long)((int)blockIdx.x - g.pTs[cls]) * BM : (long long)blockIdx.x * BM; const int n0 = blockIdx.y * BN;
  const long long pMc = perm ? perm2_count(g.a, cls, g.pImgs) : 0;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem, wslot = (unsigned)wave * 1024u;
  const int IW = MODE == MODE_CONV_FWD ? g.a.W : g.a.OW;      // row pitch of the source image
  const int sh = (MODE == MODE_CONV_BWD && g.a.stride == 2) ? 1 : 0;
  // A plan: pass i, thread tid -> tile row tid/4 + 64 i, physical slot tid%4 carrying logical chunk (tid%4) ^ swz(row): 64-bit pointer of the row's origin + tap mask
  const T* aptr[NCA]; unsigned rmask[NCA];
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    const int row = (tid >> 2) + i * 64;
    RowInfo r;
    if (perm) {
      r.valid = m0 + row < pMc; r.base = 0; r.a = 0; r.b = 0;
      if (r.valid) { long long img; int ih, iw; perm2_pixel(g.a, cls, m0 + row, img, ih, iw); r.base = img * (long long)g.a.OH * g.a.OW * g.a.C; r.a = ih + g.a.pad; r.b = iw + g.a.pad; }
    } else r = row_info<MODE>(g.a, m0 + row, g.M);
    const int ka = ((tid & 3) ^ glds_swz<RB>(row)) * 8;
    aptr[i] = (const T*)g.a.ptr + (r.base + (long long)((r.a >> sh) * IW + (r.b >> sh)) * g.a.C + ka);
    unsigned mk = 0u;
    for (int kh = 0; kh < g.a.KH; ++kh)
      for (int kw = 0; kw < g.a.KW; ++kw) {
        bool ok;
        if (MODE == MODE_CONV_FWD) { const int y = r.a + kh, x = r.b + kw; ok = y >= 0 && x >= 0 && y < g.a.H && x < g.a.W; }
        else { const int ty = r.a - kh, tx = r.b - kw; ok = ty >= 0 && tx >= 0 && !((ty | tx) & sh) && (ty >> sh) < g.a.OH && (tx >> sh) < g.a.OW; }
        mk |= ((ok && r.valid) ? 1u : 0u) << (kh * g.a.KW + kw);
      }
    rmask[i] = mk;
  }
  unsigned boff[NCB];
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int row = (tid >> 2) + i * 64; const int n = n0 + row < g.N ? n0 + row : g.N - 1;
    boff[i] = (unsigned)(((long long)n * g.ldw + (((tid & 3) ^ glds_swz<RB>(row)) * 8)) * 2);
  }
  const char* const Wb = (const char*)g.W;
  // taps of this tile (parity classes reach a subset), K-steps
  unsigned tapmask = 0xffffffffu; int KT = (g.K + KE - 1) / KE;
  int f_tap = 0, f_kh = 0, f_kw = 0, f_c0 = 0;                 // running (tap, kh, kw, channel offset) of the next K-step to be issued
  if (perm) {
    tapmask = 0u; int ntap = 0;
    for (int kh = 0; kh < g.a.KH; ++kh)
      for (int kw = 0; kw < g.a.KW; ++kw)
        if (!((((cls >> 1) + g.a.pad - kh) | ((cls & 1) + g.a.pad - kw)) & 1)) { tapmask |= 1u << (kh * g.a.KW + kw); ++ntap; }
    KT = ntap * (g.a.C / KE);
    while (f_tap < g.a.KH * g.a.KW && !((tapmask >> f_tap) & 1u)) { ++f_tap; if (++f_kw >= g.a.KW) { f_kw = 0; ++f_kh; } }
  }
  auto issue = [&](auto stagec) {
    constexpr int S = decltype(stagec)::value;
    const unsigned la = lds0 + S * TILE + wslot, lb = la + BM * RB;
    const long long tapoff = (long long)((MODE == MODE_CONV_FWD ? (f_kh * IW + f_kw) : -((f_kh >> sh) * IW + (f_kw >> sh))) * g.a.C + f_c0);
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const void* src = ((rmask[i] >> f_tap) & 1u) ? (const void*)(aptr[i] + tapoff) : (const void*)avec_zero16;
      glds16_v64(src, la + i * 4096);
    }
    glds16_group<NCB>(boff, Wb + (long long)(f_tap * g.a.C + f_c0) * 2, lb);
    f_c0 += KE;
    if (f_c0 >= g.a.C) {
      f_c0 = 0;
      do { ++f_tap; if (++f_kw >= g.a.KW) { f_kw = 0; ++f_kh; } } while (PERM && f_tap < 32 && !((tapmask >> f_tap) & 1u));
    }
  };
  const int gsel = lane >> 5;
  unsigned aad[2], bad[2];
  { const int ra = wm * (BM / 2) + (lane & 31), rb = wn * (BN / 2) + (lane & 31);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      aad[q] = lds0 + (unsigned)(ra * RB + (((2 * q + gsel) ^ glds_swz<RB>(ra)) << 4));
      bad[q] = lds0 + (unsigned)(BM * RB + rb * RB + (((2 * q + gsel) ^ glds_swz<RB>(rb)) << 4));
    } }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define AVEC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  issue(IntC<0>{});
  if (KT > 1) issue(IntC<1>{});
  auto step = [&](const int kt, auto stagec) {
    constexpr int S = decltype(stagec)::value;
    if (kt + 1 < KT) AVEC_WAIT_VM(LPT); else AVEC_WAIT_VM(0);        // tile kt + 1 may still be in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    u32x4 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      fa[q][0] = lds_read128o<S * TILE>(aad[q]); fa[q][1] = lds_read128o<S * TILE + 2048>(aad[q]);
      fb[q][0] = lds_read128o<S * TILE>(bad[q]);
      if (NT > 1) fb[q][1 % NT] = lds_read128o<S * TILE + 2048>(bad[q]);
    }
    if (kt + 2 < KT) issue(IntC<(S + 2) % STAGES>{});                 // into the slot everybody finished reading before this barrier
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[q][i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[q][j]));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (TR) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[q][j]), __builtin_bit_cast(bf16x8_t, fa[q][i]), acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[q][i]), __builtin_bit_cast(bf16x8_t, fb[q][j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll 1
  for (int kt = 0; kt < KT; kt += STAGES) {
    step(kt, IntC<0>{});
    if (kt + 1 < KT) step(kt + 1, IntC<1>{});
    if (kt + 2 < KT) step(kt + 2, IntC<2>{});
  }
#undef AVEC_WAIT_VM
  __syncthreads();
  if (TR) {
    // parity-class order: the tile's rows are class-local indices; the output row is the pixel's; a class-0-only residual (res_cls0) is indexed class-locally
    long long row[MT], rrow[MT]; bool valid[MT];
    const long long Mc = perm ? pMc : g.M;
    const bool use_res = g.e.res != nullptr && (!g.e.res_cls0 || cls == 0);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      long long mc = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
      valid[i] = mc < Mc; if (!valid[i]) mc = Mc - 1;
      row[i] = mc;
      if (perm) { long long img; int ih, iw; perm2_pixel(g.a, cls, mc, img, ih, iw); row[i] = (img * g.a.H + ih) * (long long)g.a.W + iw; }
      rrow[i] = g.e.res_cls0 ? mc : row[i];
    }
    conv_epilogue_tr<BM, BN, MT, NT>(g, acc, smem, row, valid, use_res ? (const bf16*)g.e.res : nullptr, rrow, m0 + BM <= Mc, n0, tid, lane, wm, wn);
  } else nt_epilogue<T, BM, BN, MT, NT>(g, acc, smem, m0, n0, tid, lane, wm, wn, cls);
}

// ------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) NT product: A [M][K] and W [N][K] one byte per element, per-tensor scales, fp32 accumulate on
// v_mfma_scale_f32_32x32x64_f8f6f4 (block scales fixed at 2^0).  Same LDS-DMA ring as gemm_nt_glds_kernel with 128-byte rows (= 128 K values
// = two MFMA K-steps); a lane's 32-byte operand is two adjacent 16-byte chunks of its row -- A and B fragments pick the same chunks, so the
// pairing of K indices inside the instruction does not matter.  The accumulators are multiplied by scale_a * scale_w and then go through the
// shared epilogue (bias / activation / dropout / residual / pre-activation copy), output in bf16 or fp32 as for the bf16 kernel.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int BM, int BN, int STAGES>
__global__ __launch_bounds__(256, 2) void gemm_nt_fp8_kernel(GemmArgs g, const float* __restrict__ amax_a, const float* __restrict__ amax_w) {
  typedef bf16 T;                               // dtype of the "act" buffers the epilogue reads / writes
  constexpr int RB = 128, KE = 128;             // bytes = K elements per LDS row
  constexpr int NCA = BM * 8 / 256, NCB = BN * 8 / 256;
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int TILE = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long m0 = (long long)blockIdx.x * BM; const int n0 = blockIdx.y * BN;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  long long aoff[NCA], boff[NCB]; int ka[NCA], kb[NCB];
#pragma unroll
  for (int i = 0; i < NCA; ++i) { const int row = (tid >> 3) + i * 32; aoff[i] = (m0 + row < g.M) ? (m0 + row) * g.a.ld : -1; ka[i] = ((tid & 7) ^ glds_swz<RB>(row)) * 16; }
#pragma unroll
  for (int i = 0; i < NCB; ++i) { const int row = (tid >> 3) + i * 32; boff[i] = (n0 + row < g.N) ? (long long)(n0 + row) * g.ldw : -1; kb[i] = ((tid & 7) ^ glds_swz<RB>(row)) * 16; }
  auto issue = [&](int kt, int buf) {
    char* As = smem + buf * TILE; char* Bs = As + BM * RB;
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int k = kt * KE + ka[i];
      const void* src = (aoff[i] >= 0 && k < g.K) ? (const void*)((const char*)g.a.ptr + aoff[i] + k) : (const void*)avec_zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int k = kt * KE + kb[i];
      const void* src = (boff[i] >= 0 && k < g.K) ? (const void*)((const char*)g.W + boff[i] + k) : (const void*)avec_zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
  };
  int offa[MT], swa[MT], offb[NT], swb[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) { const int row = wm * (BM / 2) + i * 32 + (lane & 31); offa[i] = row * RB; swa[i] = glds_swz<RB>(row); }
#pragma unroll
  for (int j = 0; j < NT; ++j) { const int row = wn * (BN / 2) + j * 32 + (lane & 31); offb[j] = row * RB; swb[j] = glds_swz<RB>(row); }
  const int gsel = lane >> 5;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int KT = (g.K + KE - 1) / KE;
  constexpr int LPT = NCA + NCB;
#define AVEC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st) if (st < KT) issue(st, st);
  for (int kt = 0; kt < KT; ++kt) {
    const int newer = min(STAGES - 2, KT - 1 - kt);
    if (STAGES <= 2 || newer <= 0) AVEC_WAIT_VM(0);
    else if (newer == 1) AVEC_WAIT_VM(LPT);
    else AVEC_WAIT_VM(2 * LPT);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < KT) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    const char* As = smem + (kt % STAGES) * TILE; const char* Bs = As + BM * RB;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {           // two K = 64 steps per 128-byte row; this lane's chunks: 4*s2 + 2*gsel, +1
      chunk16 fa[MT][2], fb[NT][2];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) fa[i][h] = *(const chunk16*)(As + offa[i] + (((4 * s2 + 2 * gsel + h) ^ swa[i]) << 4));
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) fb[j][h] = *(const chunk16*)(Bs + offb[j] + (((4 * s2 + 2 * gsel + h) ^ swb[j]) << 4));
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const i32x8 a = {(int)fa[i][0].w[0], (int)fa[i][0].w[1], (int)fa[i][0].w[2], (int)fa[i][0].w[3], (int)fa[i][1].w[0], (int)fa[i][1].w[1], (int)fa[i][1].w[2], (int)fa[i][1].w[3]};
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const i32x8 b = {(int)fb[j][0].w[0], (int)fb[j][0].w[1], (int)fb[j][0].w[2], (int)fb[j][0].w[3], (int)fb[j][1].w[0], (int)fb[j][1].w[1], (int)fb[j][1].w[2], (int)fb[j][1].w[3]};
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i][j], 0, 0, 0, 127, 0, 127);
        }
      }
    }
  }
#undef AVEC_WAIT_VM
  const float sc = (fmaxf(*amax_a, 1e-20f) * (1.0f / 448.0f)) * (fmaxf(*amax_w, 1e-20f) * (1.0f / 448.0f));
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] *= sc;
  __syncthreads();
  nt_epilogue<T, BM, BN, MT, NT>(g, acc, smem, m0, n0, tid, lane, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// TN: O[i][j] += sum_m P[m][i] * Q[m][j];   P plain [M][I] (T), Q via row loader ([M][J], plain or im2col)
// LDS images are [i][m] / [j][m] (reduction index contiguous) filled by transposing stores.
// ------------------------------------------------------------------------------------------------
struct TnArgs { const void* P; long long ldp; RowSrc q; float* O; void* Oact; long long ldo; long long M; int I, J, Iq, Jq; int m_per_block; float* pcs;   // Oact: when set, the result is STORED in the activation dtype (one workgroup per tile, no split) instead of added to O;   // pcs: optional column sums of P (bias gradient), transposed-read kernel only
                  // Iq/Jq: load bounds (>= I/J when rows are padded)
                int split, nb_inner, xcd_map, q_ohw, q_mg, q_nwrap; long long sPo, sPi, sQo, sQi, sOo, sOi; };   // q_*: the gathered operand's pixel bookkeeping (tn_tr_body); batching: blockIdx.z = batch * split + k-slice; batch = outer * nb_inner + inner; element strides

// LDS image of a TN operand: [col][word], word = reduction-row pair (bf16: rows 2p,2p+1 packed in 32 bits) or row (fp32), 32 words
// per 144-byte row.  Word index XOR-swizzled by 16 * parity(col bits 2..4): with lanes mapped (16 pairs x 4 column chunks) both the
// ds_write_b32 of the transposing store and the ds_read_b128 of the MFMA fragments are bank-conflict free (brute-forced offline).
__device__ __forceinline__ int tn_swz(int col) { return (((col >> 2) ^ (col >> 3) ^ (col >> 4)) & 1) << 4; }

template <typename T>
__device__ __forceinline__ void store_transposed(char* S, int col0, int p, const chunk16& c0, const chunk16& c1) {
  constexpr int VEC = Elt<T>::VEC;
  if (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) *(uint32_t*)(S + (col0 + e) * LDS_ROW + ((p ^ tn_swz(col0 + e)) << 2)) = c0.w[e];
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const uint32_t w = (e & 1) ? __builtin_amdgcn_perm(c1.w[e >> 1], c0.w[e >> 1], 0x07060302u) : __builtin_amdgcn_perm(c1.w[e >> 1], c0.w[e >> 1], 0x05040100u);
      *(uint32_t*)(S + (col0 + e) * LDS_ROW + ((p ^ tn_swz(col0 + e)) << 2)) = w;
    }
  }
}

template <typename T, int BM, int BN, int MT, int NT>
__device__ __forceinline__ void mma_tile_swz(const char* As, const char* Bs, int wm, int wn, int lane, f32x16 (&acc)[MT][NT]) {
  int offa[MT], offb[NT], lowa[MT], lowb[NT];     // row base and swizzled in-row byte offset (kept apart: the k-step is XOR-ed into the latter)
#pragma unroll
  for (int i = 0; i < MT; ++i) { const int row = wm * (BM / 2) + i * 32 + (lane & 31); offa[i] = row * LDS_ROW; lowa[i] = ((lane >> 5) * 16) ^ (tn_swz(row) << 2); }
#pragma unroll
  for (int j = 0; j < NT; ++j) { const int row = wn * (BN / 2) + j * 32 + (lane & 31); offb[j] = row * LDS_ROW; lowb[j] = ((lane >> 5) * 16) ^ (tn_swz(row) << 2); }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    chunk16 fa[MT], fb[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = *(const chunk16*)(As + offa[i] + (lowa[i] ^ (kk * 32)));
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = *(const chunk16*)(Bs + offb[j] + (lowb[j] ^ (kk * 32)));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
  }
}

template <typename T, int BI, int BJ, int MODE, bool Q_F32, bool A16>
__device__ __forceinline__ void gemm_tn_body(TnArgs g, const int bx, const int by, const int bz) {
  constexpr int VEC = Elt<T>::VEC;
  constexpr int RPT = sizeof(T) == 2 ? 2 : 1;   // reduction rows per task (bf16: a pair packed into one 32-bit LDS word)
  constexpr int KE = 32 * RPT;                  // reduction rows (m) per tile = 128 bytes of LDS row
  constexpr int CPR_I = BI / VEC, CPR_J = BJ / VEC;          // column chunks per row
  constexpr int NTI = 32 * CPR_I / 256, NTJ = 32 * CPR_J / 256;  // tasks per thread
  constexpr int MT = BI / 64, NT = BJ / 64;
  constexpr int TILE = (BI + BJ) * LDS_ROW;
  constexpr bool Q_CONV = (MODE == MODE_CONV_FWD);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = bx * BI, j0 = by * BJ;
  const int zb = bz / g.split, zs = bz - zb * g.split;
  const long long bo = zb / g.nb_inner, bi = zb - bo * g.nb_inner;
  g.P = (const T*)g.P + bo * g.sPo + bi * g.sPi;
  g.q.ptr = (Q_F32 && sizeof(T) == 2) ? (const void*)((const float*)g.q.ptr + bo * g.sQo + bi * g.sQi) : (const void*)((const T*)g.q.ptr + bo * g.sQo + bi * g.sQi);
  g.O += bo * g.sOo + bi * g.sOi;
  T* Oact = g.Oact ? (T*)g.Oact + bo * g.sOo + bi * g.sOi : nullptr;
  const long long mb = (long long)zs * g.m_per_block;
  long long me = mb + g.m_per_block; if (me > g.M) me = g.M;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // task t of this thread: (pair p, column chunk ci); lanes: 16 pairs x 4 chunks, the remaining index bits alternate pair-half / chunk-group
  auto task_p = [&](int u) { const int hi = (tid >> 6) + 4 * u; return (tid & 15) + 16 * (hi & 1); };
  auto task_c = [&](int u) { const int hi = (tid >> 6) + 4 * u; return ((tid >> 4) & 3) + 4 * (hi >> 1); };
  int qkh[NTJ], qkw[NTJ], qc[NTJ], qoh[NTJ][RPT], qow[NTJ][RPT]; long long qimg[NTJ][RPT];
  if (Q_CONV) {
#pragma unroll
    for (int u = 0; u < NTJ; ++u) {
      const int j = j0 + task_c(u) * VEC; const int tap = j / g.q.C; qc[u] = j - tap * g.q.C; qkh[u] = tap / g.q.KW; qkw[u] = tap - qkh[u] * g.q.KW;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const long long m = mb + task_p(u) * RPT + r; qow[u][r] = (int)(m % g.q.OW); const long long t = m / g.q.OW; qoh[u][r] = (int)(t % g.q.OH); qimg[u][r] = t / g.q.OH;
      }
    }
  }
  Pend pp[NTI][RPT], pq[NTJ][RPT];
  auto issue = [&](long long mt0) {
#pragma unroll
    for (int u = 0; u < NTI; ++u) {
      const int col = i0 + task_c(u) * VEC;
#pragma unroll
      for (int r = 0; r < RPT; ++r) { const long long m = mt0 + task_p(u) * RPT + r; pp[u][r] = issue_load<T, false, A16>(g.P, m < me ? m * g.ldp + col : -1, col, g.Iq); }
    }
#pragma unroll
    for (int u = 0; u < NTJ; ++u) {
      const int col = j0 + task_c(u) * VEC;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const long long m = mt0 + task_p(u) * RPT + r;
        if (Q_CONV) {
          RowInfo ri; ri.valid = m < me; ri.base = qimg[u][r] * (long long)g.q.H * g.q.W * g.q.C;
          ri.a = qoh[u][r] * g.q.stride - g.q.pad; ri.b = qow[u][r] * g.q.stride - g.q.pad;
          pq[u][r] = issue_load<T, false, A16>(g.q.ptr, conv_offset<MODE_CONV_FWD>(g.q, ri, qkh[u], qkw[u], qc[u]), col, g.J);
          qow[u][r] += KE;                                       // advance this row by one reduction tile
          while (qow[u][r] >= g.q.OW) { qow[u][r] -= g.q.OW; if (++qoh[u][r] >= g.q.OH) { qoh[u][r] = 0; ++qimg[u][r]; } }
        } else {
          long long row = m;
          if (g.q.step > 1) row = (m / g.q.rows_out) * (long long)g.q.rows_in + (m % g.q.rows_out) * (long long)g.q.step;
          pq[u][r] = issue_load<T, Q_F32, A16>(g.q.ptr, m < me ? row * g.q.ld + col : -1, col, g.Jq);
        }
      }
    }
  };
  auto stage = [&](int buf) {
    char* Ps = smem + buf * TILE; char* Qs = Ps + BI * LDS_ROW;
#pragma unroll
    for (int u = 0; u < NTI; ++u) store_transposed<T>(Ps, task_c(u) * VEC, task_p(u), finish_load<T, false>(pp[u][0]), finish_load<T, false>(pp[u][RPT - 1]));
#pragma unroll
    for (int u = 0; u < NTJ; ++u) store_transposed<T>(Qs, task_c(u) * VEC, task_p(u), finish_load<T, Q_F32 && !Q_CONV>(pq[u][0]), finish_load<T, Q_F32 && !Q_CONV>(pq[u][RPT - 1]));
  };
  if (mb < me) {
    issue(mb);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (long long mt0 = mb; mt0 < me; mt0 += KE, cur ^= 1) {
      const bool more = mt0 + KE < me;
      if (more) issue(mt0 + KE);
      mma_tile_swz<T, BI, BJ, MT, NT>(smem + cur * TILE, smem + cur * TILE + BI * LDS_ROW, wm, wn, lane, acc);
      if (more) stage(cur ^ 1);
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = j0 + wn * (BJ / 2) + j * 32 + (lane & 31);
    if (col >= g.J) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * (BI / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.I) { if (Oact) stf(Oact + (long long)row * g.ldo + col, acc[i][j][r]); else atomicAdd(g.O + (long long)row * g.ldo + col, acc[i][j][r]); }
      }
  }
}

template <typename T, int BI, int BJ, int MODE, bool Q_F32, bool A16>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs g) { gemm_tn_body<T, BI, BJ, MODE, Q_F32, A16>(g, blockIdx.x, blockIdx.y, blockIdx.z); }

// Up to three independent batched products of the small-tile kind in ONE launch (dK, dV and dE of an attention layer's backward pass: three launches of ~5 us each
// on 50-400 workgroups): workgroup w belongs to the problem whose [first, first + count) range contains it.
#define AVEC_TN_MULTI_MAX 3
struct TnMulti { TnArgs g[AVEC_TN_MULTI_MAX]; int first[AVEC_TN_MULTI_MAX + 1], gx[AVEC_TN_MULTI_MAX], gy[AVEC_TN_MULTI_MAX], n; };
__global__ __launch_bounds__(256, 2) void gemm_tn_multi_kernel(TnMulti m) {
  const int w = blockIdx.x;
  int p = 0; if (m.n > 1 && w >= m.first[1]) p = 1; if (m.n > 2 && w >= m.first[2]) p = 2;
  int l = w - m.first[p]; const int bx = l % m.gx[p]; l /= m.gx[p];
  gemm_tn_body<bf16, 64, 64, MODE_PLAIN, false, false>(m.g[p], bx, l % m.gy[p], l / m.gy[p]);
}

// ------------------------------------------------------------------------------------------------
// TN, bf16, gfx950 transposed-read variant.  The reduction tiles are copied AS THEY LIE in memory ([m][i] and [m][j], 64 rows) by the
// LDS-DMA (no register staging, no transposing stores) and the MFMA operands are fetched with ds_read_b64_tr_b16: a 16-lane group reads a
// [4 m][16 col] block and every lane receives 4 consecutive m of its column = half of a 32x32x16 operand.  16-byte chunks are XOR-swizzled
// (source side: lane L of a DMA fetches the logical chunk that belongs in physical slot L) so that the 4 rows of a block fall on
// distinct banks: chunk ^= 4*(row & 3) for 256-byte rows, ^= 4*((row >> 1) & 1) for 128-byte rows.
// ------------------------------------------------------------------------------------------------
typedef short v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ chunk16 tr_read8(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) v4s_t* lp_t;
  const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p1);
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  chunk16 f; f.w[0] = ua.x; f.w[1] = ua.y; f.w[2] = ub.x; f.w[3] = ub.y; return f;
}
template <int BC> __device__ __forceinline__ int tr_swz(int row) { return BC == 128 ? 4 * (row & 3) : 4 * ((row >> 1) & 1); }

template <int BI, int BJ, int MODE, int STAGES, bool Q32 = false, int KT = 64>      // Q32: the im2col source has < 2^31 elements: 32-bit offsets; KT: reduction rows per tile
__device__ __forceinline__ void tn_tr_body(const TnArgs& g, const int bx, const int by, const int bz) {
  typedef bf16 T;
  constexpr int CPI = BI / 8, CPJ = BJ / 8;                    // 16-byte chunks per LDS row
  constexpr int RI = 256 / CPI, RJ = 256 / CPJ;                // rows covered by one DMA pass of the workgroup
  constexpr int NLI = KT / RI, NLJ = KT / RJ;                  // DMA instructions per thread per tile
  constexpr int MTI = BI / 64, MTJ = BJ / 64;
  constexpr int PBYTES = KT * BI * 2, QBYTES = KT * BJ * 2, TILE = PBYTES + QBYTES;
  constexpr bool Q_CONV = (MODE == MODE_CONV_FWD);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int i0 = bx * BI, j0 = by * BJ;
  const int zb = bz / g.split, zs = bz - zb * g.split;
  const long long bo = zb / g.nb_inner, bi = zb - bo * g.nb_inner;
  const T* Pp = (const T*)g.P + bo * g.sPo + bi * g.sPi;
  const T* Qp = (const T*)g.q.ptr + bo * g.sQo + bi * g.sQi;
  float* Op = g.O + bo * g.sOo + bi * g.sOi;
  const long long mb = (long long)zs * g.m_per_block;
  long long me = mb + g.m_per_block; if (me > g.M) me = g.M;

  f32x16 acc[MTI][MTJ];
#pragma unroll
  for (int i = 0; i < MTI; ++i)
#pragma unroll
    for (int j = 0; j < MTJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA task u: LDS slot u*256 + tid = (row tid/CP + u*R, physical chunk tid%CP); the swizzle term does not depend on u (R = 16 or 32 rows)
  const int prow0 = tid / CPI, pcol = i0 + (((tid % CPI) ^ tr_swz<BI>(prow0)) << 3);
  const int qrow0 = tid / CPJ, qcol = j0 + (((tid % CPJ) ^ tr_swz<BJ>(qrow0)) << 3);
  const bool pok = pcol < g.Iq, qok = qcol < (Q_CONV ? g.J : g.Jq);
  // gathered operand: a DMA slot follows ONE reduction row per tile; its pixel is kept as (index inside the image qpx, element offset of the image qimg) and advanced by KT
  // pixels per tile with q_nwrap compare / subtract rounds (no data-dependent loop: the old row / column / image carry chain with its divergent while loop was most of
  // what a wave issued per tile); row and column follow from a multiply-shift that the host has checked for every pixel index (q_mg, 20-bit shift)
  int qkh = 0, qkw = 0, qc = 0, qpx[NLJ]; long long qimg[NLJ];
  const int q_hwc = Q_CONV ? g.q.H * g.q.W * g.q.C : 0;
  if (Q_CONV) {
    const int tap = qcol / g.q.C; qc = qcol - tap * g.q.C; qkh = tap / g.q.KW; qkw = tap - qkh * g.q.KW;
#pragma unroll
    for (int u = 0; u < NLJ; ++u) {
      const long long m = mb + qrow0 + u * RJ; const long long im = m / g.q_ohw; qpx[u] = (int)(m - im * g.q_ohw); qimg[u] = im * (long long)q_hwc;
    }
  }
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // The DMA goes out through inline asm: behind the builtin the compiler drains the queue (s_waitcnt vmcnt(0), visible in the ISA of this loop) in front of the first
  // transposed read it cannot prove disjoint, i.e. right behind the prefetch of the next tile (DESIGN.md 17.2 rule 3); completion is counted by hand below (AVEC_WAIT_VM).
  // (The gathered operand only gained from it once its per-tile address arithmetic had been cut down -- with the old carry chain the asm form was 5-25 % slower.)
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  auto tn_dma = [&](const void* src, unsigned off) {
    if ((!Q_CONV || AVEC_TN_CONV_ASM) && !(AVEC_TN_BUILTIN_DMA)) glds16_v64(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + off)));
    else __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + off), 16, 0, 0);
  };
  // plain products, whole tiles: scalar base (first row of the tile) + per-lane 32-bit offsets fixed at entry, one M0 save / restore per operand (glds16_group) --
  // the per-lane 64-bit address arithmetic of the general path below (~25 instructions per DMA, 8 DMAs per tile) was most of what a wave issued between two
  // groups of MFMAs.  Lanes beyond the operand's width fetch the tile's first column instead of the zero page: whatever they load only reaches result rows /
  // columns that are not stored.  The last, partial tile (rows >= me must be ZERO) and strided rows take the general path.
  unsigned poff[NLI], qoff[NLJ];
  bool lean = false;
  if constexpr (!Q_CONV) {
    lean = g.q.step <= 1 && !(AVEC_TN_BUILTIN_DMA);
#pragma unroll
    for (int u = 0; u < NLI; ++u) poff[u] = (unsigned)(((long long)(prow0 + u * RI) * g.ldp + (pok ? pcol : i0)) * 2);
#pragma unroll
    for (int u = 0; u < NLJ; ++u) qoff[u] = (unsigned)(((long long)(qrow0 + u * RJ) * g.q.ld + (qok ? qcol : j0)) * 2);
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](long long mt0, int buf) {
    if constexpr (!Q_CONV) {
      if (lean && mt0 + KT <= me) {
        const unsigned l0 = lds0 + (unsigned)(buf * TILE) + (unsigned)wave_u * 1024u;
        glds16_group<NLI>(poff, (const char*)Pp + mt0 * g.ldp * 2, l0);
        glds16_group<NLJ>(qoff, (const char*)Qp + mt0 * g.q.ld * 2, l0 + PBYTES);
        return;
      }
    }
#pragma unroll
    for (int u = 0; u < NLI; ++u) {
      const long long m = mt0 + prow0 + u * RI;
      const void* src = (pok && m < me) ? (const void*)(Pp + m * g.ldp + pcol) : (const void*)avec_zero16;
      tn_dma(src, (unsigned)(buf * TILE + (u * 256 + wave * 64) * 16));
    }
#pragma unroll
    for (int u = 0; u < NLJ; ++u) {
      const long long m = mt0 + qrow0 + u * RJ;
      const void* src = (const void*)avec_zero16;
      if (Q_CONV) {
        const int oh = (int)(((unsigned)qpx[u] * (unsigned)g.q_mg) >> 20), ow = qpx[u] - oh * g.q.OW;
        const int ih = oh * g.q.stride - g.q.pad + qkh, iw = ow * g.q.stride - g.q.pad + qkw;
        const bool ok = qok && m < me && (unsigned)ih < (unsigned)g.q.H && (unsigned)iw < (unsigned)g.q.W;
        const T* sp = Q32 ? Qp + ((int)qimg[u] + (ih * g.q.W + iw) * g.q.C + qc) : Qp + qimg[u] + (ih * g.q.W + iw) * g.q.C + qc;
        src = ok ? (const void*)sp : src;
        qpx[u] += KT;                                            // advance this row by one reduction tile
        for (int r = 0; r < g.q_nwrap; ++r) { const bool w = qpx[u] >= g.q_ohw; qpx[u] -= w ? g.q_ohw : 0; qimg[u] += w ? q_hwc : 0; }
      } else {
        long long row = m;
        if (g.q.step > 1) row = (m / g.q.rows_out) * (long long)g.q.rows_in + (m % g.q.rows_out) * (long long)g.q.step;
        src = (qok && m < me) ? (const void*)(Qp + row * g.q.ld + qcol) : src;
      }
      tn_dma(src, (unsigned)(buf * TILE + PBYTES + (u * 256 + wave * 64) * 16));
    }
  };
  // fragment addressing: lane = (16-lane group g4, t); operand row/col 16*(g4&1) + t of the 32-wide block, k-group g4>>1; read h fetches
  // reduction rows 8*(g4>>1) + 4h + (t>>2), 8-byte piece (t&3) of the 32-byte column block
  const int g4 = lane >> 4, t = lane & 15;
  int offa[MTI][2], offb[MTJ][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * (g4 >> 1) + 4 * h + (t >> 2);
#pragma unroll
    for (int i = 0; i < MTI; ++i) { const int cb = wi * (BI / 2) + i * 32 + 16 * (g4 & 1); offa[i][h] = row * (BI * 2) + ((((cb >> 3) + ((t & 3) >> 1)) ^ tr_swz<BI>(row)) << 4) + (t & 1) * 8; }
#pragma unroll
    for (int j = 0; j < MTJ; ++j) { const int cb = wj * (BJ / 2) + j * 32 + 16 * (g4 & 1); offb[j][h] = PBYTES + row * (BJ * 2) + ((((cb >> 3) + ((t & 3) >> 1)) ^ tr_swz<BJ>(row)) << 4) + (t & 1) * 8; }
  }
  // optional column sums of P (= bias gradient of the layer whose weight gradient this is), by the workgroups of the first J tile: lane =
  // (physical chunk pc, row class r4 = m & 3, row subgroup rs); rows of one class share the swizzle, so a lane always sees the same 8 columns
  constexpr int CS_RF = 64 / (CPI * 4), CS_N = (KT / 4) / (4 * CS_RF);
  const bool do_cs = g.pcs != nullptr && by == 0;
  const int cs_pc = lane % CPI, cs_r4 = (lane / CPI) & 3, cs_rs = lane / (CPI * 4);
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int LPT = NLI + NLJ;
#define AVEC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  const int KTN = (int)((me - mb + KT - 1) / KT);
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st) if (st < KTN) issue(mb + (long long)st * KT, st);
  for (int kt = 0; kt < KTN; ++kt) {
    const int newer = min(STAGES - 2, KTN - 1 - kt);
    if (STAGES <= 2 || newer <= 0) AVEC_WAIT_VM(0);
    else if (newer == 1) AVEC_WAIT_VM(LPT);
    else AVEC_WAIT_VM(2 * LPT);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < KTN) issue(mb + (long long)(kt + STAGES - 1) * KT, (kt + STAGES - 1) % STAGES);
    const char* S = smem + (kt % STAGES) * TILE;
    if (do_cs) {
#pragma unroll
      for (int n = 0; n < CS_N; ++n) {
        const int m = cs_r4 + 4 * (wave * CS_RF + cs_rs + 4 * CS_RF * n);
        const chunk16 c = *(const chunk16*)(S + m * (BI * 2) + (cs_pc << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[2 * e] += __uint_as_float(c.w[e] << 16); cs[2 * e + 1] += __uint_as_float(c.w[e] & 0xffff0000u); }
      }
    }
#pragma unroll
    for (int kk = 0; kk < KT / 16; ++kk) {
      chunk16 fa[MTI], fb[MTJ];
#pragma unroll
      for (int i = 0; i < MTI; ++i) fa[i] = tr_read8(S + offa[i][0] + kk * 16 * BI * 2, S + offa[i][1] + kk * 16 * BI * 2);
#pragma unroll
      for (int j = 0; j < MTJ; ++j) fb[j] = tr_read8(S + offb[j][0] + kk * 16 * BJ * 2, S + offb[j][1] + kk * 16 * BJ * 2);
#pragma unroll
      for (int i = 0; i < MTI; ++i)
#pragma unroll
        for (int j = 0; j < MTJ; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
    }
  }
#undef AVEC_WAIT_VM
  if (do_cs) {                                           // workgroup-uniform: reduce the 16 partials per column in LDS, then one global atomic per column
    float* red = (float*)smem;
    __syncthreads();                                     // the ring is free
    if (tid < BI) red[tid] = 0.f;
    __syncthreads();
    const int cl = (cs_pc ^ tr_swz<BI>(cs_r4)) << 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(red + cl + e, cs[e]);
    __syncthreads();
    if (tid < BI && i0 + tid < g.I) atomicAdd(g.pcs + i0 + tid, red[tid]);
  }
#pragma unroll
  for (int j = 0; j < MTJ; ++j) {
    const int col = j0 + wj * (BJ / 2) + j * 32 + (lane & 31);
    if (col >= g.J) continue;
#pragma unroll
    for (int i = 0; i < MTI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wi * (BI / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.I) atomicAdd(Op + (long long)row * g.ldo + col, acc[i][j][r]);
      }
  }
}

template <int BI, int BJ, int MODE, int STAGES, bool Q32 = false, int KT = 64>
__global__ __launch_bounds__(256, 2) void gemm_tn_tr_kernel(TnArgs g) {
  if (g.xcd_map) {                             // (tile x fastest, then tile y, then batch / reduction slice: the tiles of one slice are neighbours)
    int l = xcd_logical(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int bx = l % gridDim.x; l /= gridDim.x;
    tn_tr_body<BI, BJ, MODE, STAGES, Q32, KT>(g, bx, l % gridDim.y, l / gridDim.y);
  } else tn_tr_body<BI, BJ, MODE, STAGES, Q32, KT>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- grouped launch: up to AVEC_TN_GROUP_MAX independent plain bf16 products (the weight gradients of one or more conformer blocks) as ONE grid.
// The items travel BY VALUE in the kernel argument block (captured by a hipGraph node like any other argument; no device-side table to keep alive);
// workgroup w belongs to the item whose [first, first + count) range contains it.
struct TnItem { const void* P; const void* Q; float* O; float* pcs; int ldp, ldq, ldo, M, I, J, Iq, Jq, m_per_block, split, gx, gy, rows_out, rows_in, step, first; };
struct TnGroup { TnItem it[AVEC_TN_GROUP_MAX]; int n, total, xcd_map; };
template <int BT, int STG = 2, int KT = 32>
__global__ __launch_bounds__(256, 2) void gemm_tn_tr_grouped_kernel(TnGroup grp) {
  const int w = grp.xcd_map ? xcd_logical(blockIdx.x, grp.total) : (int)blockIdx.x;
  int lo = 0, hi = grp.n - 1;                   // last item with first <= w
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (grp.it[mid].first <= w) lo = mid; else hi = mid - 1; }
  const TnItem& t = grp.it[lo];
  TnArgs g; g.P = t.P; g.ldp = t.ldp; g.q.ptr = t.Q; g.q.ld = t.ldq; g.q.rows_out = t.rows_out; g.q.rows_in = t.rows_in; g.q.step = t.step;
  g.q.H = g.q.W = g.q.C = g.q.KH = g.q.KW = g.q.stride = g.q.pad = g.q.OH = g.q.OW = 0;
  g.O = t.O; g.Oact = nullptr; g.ldo = t.ldo; g.M = t.M; g.I = t.I; g.J = t.J; g.Iq = t.Iq; g.Jq = t.Jq; g.m_per_block = t.m_per_block; g.pcs = t.pcs;
  g.split = t.split; g.nb_inner = 1; g.xcd_map = 0; g.q_ohw = 1; g.q_mg = 0; g.q_nwrap = 0; g.sPo = g.sPi = g.sQo = g.sQi = g.sOo = g.sOi = 0;
  int l = w - t.first; const int bx = l % t.gx; l /= t.gx; const int by = l % t.gy; const int bz = l / t.gy;
  tn_tr_body<BT, BT, MODE_PLAIN, STG, false, KT>(g, bx, by, bz);
}

// ------------------------------------------------------------------------------------------------
// host launchers (C ABI)
// ------------------------------------------------------------------------------------------------
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static RowSrc make_src(const void* ptr, const avec_rows_t* d) {
  RowSrc s; s.ptr = ptr; s.ld = d->ld; s.rows_out = d->rows_out; s.rows_in = d->rows_in; s.step = d->step;
  s.H = d->H; s.W = d->W; s.C = d->C; s.KH = d->KH; s.KW = d->KW; s.stride = d->stride; s.pad = d->pad; s.OH = d->OH; s.OW = d->OW;
  return s;
}

template <typename K> static int want_lds(K kern, size_t bytes) {
  static const void* done[256]; static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
  if (ndone < 256) done[ndone++] = (const void*)kern;
  return 0;
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// register-direct epilogue (conv_epilogue_tr, transposed product): bf16 output and residual in whole 16-byte pieces, nothing but alpha / residual / BatchNorm statistics
// to fuse.  AVEC_NO_EPI_TR=1: the staged epilogue everywhere (A/B runs).
static bool epi_tr_ok(const GemmArgs& g) {
  static const bool off = getenv("AVEC_NO_EPI_TR") != nullptr;
  const Epi& e = g.e;
  return !off && !e.out_f32 && !e.out_pre && !e.bias && e.act == 0 && !(e.drop_p > 0.f) && !e.dact && !e.colsum && !e.bnb_y && g.N % 8 == 0 &&
         e.ldo % 8 == 0 && aligned16(e.out) && (!e.res || (e.res_act && e.ldres % 8 == 0 && aligned16(e.res)));
}

// register-direct epilogue of the plain 64 x 64 product: whole 8-column pieces, 16-byte aligned rows of every operand, no column reduction.  AVEC_NO_PLAIN_TR=1: off
static bool plain_tr_ok(const GemmArgs& g) {
  static const bool off = getenv("AVEC_NO_PLAIN_TR") != nullptr || getenv("AVEC_NO_EPI_TR") != nullptr;
  const Epi& e = g.e;
  const long long ob = e.out_f32 ? 4 : 2;
  return !off && !e.colsum && !e.stats && !e.bnb_y && g.N % 8 == 0 && g.N >= 8 && aligned16(e.out) && (e.ldo * ob) % 16 == 0 &&
         (!e.out_pre || (aligned16(e.out_pre) && e.ldpre % 8 == 0)) && (!e.bias || aligned16(e.bias)) && (!e.dact || (aligned16(e.dact_z) && e.ldz % 8 == 0)) &&
         (!e.res || (aligned16(e.res) && (e.ldres * (e.res_act ? 2 : 4)) % 16 == 0));
}

template <typename T, int BM, int BN>
static int launch_nt_mode(const GemmArgs& g_in, int mode, int src_f32, hipStream_t st) {
  GemmArgs g = g_in;
  dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)((g.N + BN - 1) / BN));
  size_t lds = (size_t)2 * (BM + BN) * LDS_ROW;
  constexpr int VEC = Elt<T>::VEC;
  const bool f32src = src_f32 && sizeof(T) == 2;
  // every chunk address 16-byte aligned?  (then each chunk is one global_load_dwordx4 instead of two dwordx2)
  const bool a16 = aligned16(g.a.ptr) && aligned16(g.W) && g.K % VEC == 0 && g.ldw % VEC == 0 &&
                   (mode != MODE_PLAIN || (g.a.ld % (f32src ? 4 : VEC) == 0));
  static const bool use_glds_ = true;
  static const bool no_ktail = false;
  // plain bf16 products take the LDS-DMA kernel whatever their alignment (the DMA takes any source address); K = 8n + 4 with the in-LDS tail fix-up
  const bool plain_any = sizeof(T) == 2 && mode == MODE_PLAIN && !f32src && use_glds_ && !no_ktail && (g.K % 8 == 0 || g.K % 8 == 4) && g.K >= 8 && g.ldw >= g.K && g.a.ld >= g.K &&
                         g.a.step <= 1;
  g.ktail = (plain_any && g.K % 8 == 4) ? 4 : 0;
  if (g.perm2 && mode == MODE_CONV_BWD && g.fast_conv && a16 && !f32src && use_glds_) {     // parity-class order: only the fast LDS-DMA kernel knows it
    g.pTs[0] = 0; for (int c = 0; c < 4; ++c) g.pTs[c + 1] = g.pTs[c] + (int)((perm2_count(g.a, c, g.pImgs) + BM - 1) / BM);
    grid.x = (unsigned)g.pTs[4];
  } else {
    if (g.e.res_cls0) { avec_set_error("gemm_nt: res_cls0: this launch cannot run in parity-class order (alignment)"); return -1; }
    g.perm2 = 0;
  }
  constexpr int STG = (BM + BN) <= 128 ? 4 : 2;      // ring depth: deep for the small latency-bound tiles; the big tiles keep 3 workgroups per CU instead (measured)
  static const bool rb_env_set = getenv("AVEC_NT_RB") != nullptr;
  static const int rb_env = rb_env_set ? atoi(getenv("AVEC_NT_RB")) : 128;
  const size_t epi_lds = (size_t)64 * (BN + 4) * 4 + 10 * BN * 4;
#define G2(MODE, FC, RB_) do { const size_t l2 = (size_t)STG * (BM + BN) * RB_ > epi_lds ? (size_t)STG * (BM + BN) * RB_ : epi_lds; \
    avec_note_kernel("gemm_nt_glds_kernel<%s,%d,%d,%d,%d,%d,%d>", (sizeof(T) == 2 ? "bf16" : "float"), BM, BN, MODE, STG, (int)FC, RB_); if (int r = want_lds(gemm_nt_glds_kernel<T, BM, BN, MODE, STG, FC, RB_>, l2)) return r; hipLaunchKernelGGL((gemm_nt_glds_kernel<T, BM, BN, MODE, STG, FC, RB_>), grid, dim3(256), l2, st, g); return 0; } while (0)
  // 64-byte rows (K-step 32): half the ring, 4 resident workgroups per CU instead of 2 -- measured +4..18 % on the implicit-GEMM layers with
  // thousands of tiles, -16 % on the deep-K / few-tile ones (512-channel 3x3 stage): chosen by tile count.  AVEC_NT_RB=64/128 forces it.
  const long long ntiles = (long long)grid.x * grid.y;
  const bool rb64 = sizeof(T) == 2 && (rb_env_set ? rb_env == 64 : ntiles >= 1536);
  static const int stg_env = 3;     // ring depth of the fast implicit-GEMM kernels with 64-byte rows: 3 measured +2..6 % over 2, 4 is -5..10 %
#define G3(MODE, S_) do { const size_t l2 = (size_t)S_ * (BM + BN) * 64 > epi_lds ? (size_t)S_ * (BM + BN) * 64 : epi_lds; \
    avec_note_kernel("gemm_nt_glds_kernel<%s,%d,%d,%d,%d,1,64>", (sizeof(T) == 2 ? "bf16" : "float"), BM, BN, MODE, S_); if (int r = want_lds(gemm_nt_glds_kernel<T, BM, BN, MODE, S_, true, 64>, l2)) return r; hipLaunchKernelGGL((gemm_nt_glds_kernel<T, BM, BN, MODE, S_, true, 64>), grid, dim3(256), l2, st, g); return 0; } while (0)
#define G(MODE) do { if (MODE != MODE_PLAIN && g.fast_conv) { if (rb64 && stg_env == 3 && (BM + BN) > 128) G3(MODE, 3); if (rb64 && stg_env == 4 && (BM + BN) > 128) G3(MODE, 4); \
    if (rb64) G2(MODE, true, 64); else G2(MODE, true, 128); } G2(MODE, false, 128); } while (0)
  static const bool use_glds = true;
  static const bool no_lean = false;
  if constexpr (sizeof(T) == 2 && BN == 64 && BM == 64) {       // (128 x 64 with the 4-stage ring leaves one workgroup per CU: slower than the general kernel's 2-stage ring)
    // the lean plain kernel: whole 16-byte K-chunks, 32-bit byte offsets into both operands
    const long long arows = g.a.step > 1 ? (g.M / (g.a.rows_out > 0 ? g.a.rows_out : 1) + 1) * (long long)g.a.rows_in : g.M;
    static const bool no_lean_tail = false;
    if (mode == MODE_PLAIN && !f32src && use_glds && !no_lean && g.a.step <= 1 && (g.K % 8 == 0 || (g.ktail && !no_lean_tail)) && g.K >= 8 && g.ldw >= g.K && g.a.ld >= g.K &&
        arows * g.a.ld * 2 < (1ll << 32) && (long long)g.N * g.ldw * 2 < (1ll << 32)) {
      // two-stage ring for products with more than 512 tiles (the slots of the deep ring) and at most 6 K tiles: 3200 x 1024 x 256 + Swish 10.4 -> 8.2 us,
      // 3200 x 768 x 256 8.9 -> 6.4 us, 1600 x 1440 x 360 11.7 -> 9.4 us; step 19.31 -> 19.18 ms (tools/gpu/r4_s2.sh; with 256: no further gain).  AVEC_NT_S2=0: off
      static const int s2_min = 512;
      const bool tr = plain_tr_ok(g);
      if (s2_min > 0 && (long long)grid.x * grid.y > s2_min && !g.ktail && g.K <= 384) {
        const size_t l2s = (size_t)2 * (BM + BN) * 128 > epi_lds ? (size_t)2 * (BM + BN) * 128 : epi_lds;
        if (tr) {
          const size_t lt = (size_t)2 * (BM + BN) * 128;
          avec_note_kernel("gemm_nt_plain_kernel<%d,%d,2,tr>", BM, BN);
          if (int r = want_lds(gemm_nt_plain_kernel<BM, BN, 2, true>, lt)) return r;
          hipLaunchKernelGGL((gemm_nt_plain_kernel<BM, BN, 2, true>), grid, dim3(256), lt, st, g.a.ptr, g.W, g.a.ld, g.ldw, g.M, g.N, g.K, g.ktail, g); return 0;
        }
        avec_note_kernel("gemm_nt_plain_kernel<%d,%d,2,false>", BM, BN);
        if (int r = want_lds(gemm_nt_plain_kernel<BM, BN, 2>, l2s)) return r;
        hipLaunchKernelGGL((gemm_nt_plain_kernel<BM, BN, 2>), grid, dim3(256), l2s, st, g.a.ptr, g.W, g.a.ld, g.ldw, g.M, g.N, g.K, g.ktail, g); return 0;
      }
      const size_t l2 = (size_t)4 * (BM + BN) * 128 > epi_lds ? (size_t)4 * (BM + BN) * 128 : epi_lds;
      if (tr) {
        const size_t lt = (size_t)4 * (BM + BN) * 128;
        avec_note_kernel("gemm_nt_plain_kernel<%d,%d,4,tr>", BM, BN);
        if (int r = want_lds(gemm_nt_plain_kernel<BM, BN, 4, true>, lt)) return r;
        hipLaunchKernelGGL((gemm_nt_plain_kernel<BM, BN, 4, true>), grid, dim3(256), lt, st, g.a.ptr, g.W, g.a.ld, g.ldw, g.M, g.N, g.K, g.ktail, g); return 0;
      }
      avec_note_kernel("gemm_nt_plain_kernel<%d,%d,4,false>", BM, BN);
      if (int r = want_lds(gemm_nt_plain_kernel<BM, BN>, l2)) return r;
      hipLaunchKernelGGL((gemm_nt_plain_kernel<BM, BN>), grid, dim3(256), l2, st, g.a.ptr, g.W, g.a.ld, g.ldw, g.M, g.N, g.K, g.ktail, g); return 0;
    }
  }
  if constexpr (sizeof(T) == 2 && BM == 128 && (BN == 128 || BN == 64)) {
    static const bool no_clean = getenv("AVEC_NO_LEAN_CONV") != nullptr;
    if (mode != MODE_PLAIN && g.fast_conv && a16 && !f32src && use_glds && !no_clean && g.a.C % 32 == 0 && (long long)g.N * g.ldw * 2 < (1ll << 32)) {
      const size_t l2 = (size_t)3 * (BM + BN) * 64 > epi_lds ? (size_t)3 * (BM + BN) * 64 : epi_lds;
      if (epi_tr_ok(g)) {
        const size_t lt = (size_t)3 * (BM + BN) * 64;
        avec_note_kernel("gemm_nt_conv_lean_kernel<%d,%d,tr>", BN, mode);
        if (mode == MODE_CONV_FWD) { if (int r = want_lds(gemm_nt_conv_lean_kernel<BN, MODE_CONV_FWD, true>, lt)) return r; hipLaunchKernelGGL((gemm_nt_conv_lean_kernel<BN, MODE_CONV_FWD, true>), grid, dim3(256), lt, st, g); }
        else { if (int r = want_lds(gemm_nt_conv_lean_kernel<BN, MODE_CONV_BWD, true>, lt)) return r; hipLaunchKernelGGL((gemm_nt_conv_lean_kernel<BN, MODE_CONV_BWD, true>), grid, dim3(256), lt, st, g); }
        return 0;
      }
      if (g.e.res_mask) { avec_set_error("gemm_nt: res_mask needs the register-direct epilogue"); return -1; }
      avec_note_kernel("gemm_nt_conv_lean_kernel<%d,%d>", BN, mode);
      if (mode == MODE_CONV_FWD) { if (int r = want_lds(gemm_nt_conv_lean_kernel<BN, MODE_CONV_FWD>, l2)) return r; hipLaunchKernelGGL((gemm_nt_conv_lean_kernel<BN, MODE_CONV_FWD>), grid, dim3(256), l2, st, g); }
      else { if (int r = want_lds(gemm_nt_conv_lean_kernel<BN, MODE_CONV_BWD>, l2)) return r; hipLaunchKernelGGL((gemm_nt_conv_lean_kernel<BN, MODE_CONV_BWD>), grid, dim3(256), l2, st, g); }
      return 0;
    }
  }
  if (g.e.res_mask) { avec_set_error("gemm_nt: res_mask needs the register-direct epilogue (no kernel with it takes this product)"); return -1; }
  if ((a16 || plain_any) && !f32src && use_glds) { if (mode == MODE_PLAIN) G(MODE_PLAIN); else if (mode == MODE_CONV_FWD) G(MODE_CONV_FWD); else G(MODE_CONV_BWD); }
#undef G2
#undef G3
#undef G
#define L(MODE, F, A) do { avec_note_kernel("gemm_nt_kernel<%s,%d,%d,%d,%d,%d>", (sizeof(T) == 2 ? "bf16" : "float"), BM, BN, MODE, (int)F, (int)A); if (int r = want_lds(gemm_nt_kernel<T, BM, BN, MODE, F, A>, lds)) return r; hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, MODE, F, A>), grid, dim3(256), lds, st, g); } while (0)
  if (mode == MODE_PLAIN) {
    if (f32src) { if (a16) L(MODE_PLAIN, true, true); else L(MODE_PLAIN, true, false); }
    else { if (a16) L(MODE_PLAIN, false, true); else L(MODE_PLAIN, false, false); }
  } else if (mode == MODE_CONV_FWD) { if (a16) L(MODE_CONV_FWD, false, true); else L(MODE_CONV_FWD, false, false); }
  else { if (a16) L(MODE_CONV_BWD, false, true); else L(MODE_CONV_BWD, false, false); }
#undef L
  return 0;
}

// host side: 1 = not applicable (caller continues with the generic kernels), 0 = launched, other = error
static int launch_conv_shift(const GemmArgs& g_in, int mode, hipStream_t st) {
  static const bool off = getenv("AVEC_NO_CONV_SHIFT") != nullptr;
  const RowSrc& a = g_in.a;
  if (off || mode == MODE_PLAIN || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.H != a.OH || a.W != a.OW || a.W > 31 || a.C % 32 != 0) return 1;
  if (!aligned16(a.ptr) || !aligned16(g_in.W) || g_in.ldw % 8 != 0 || g_in.M * a.C >= (1ll << 31) || (long long)g_in.N * g_in.ldw >= (1ll << 31) || g_in.N < 64) return 1;
  static const bool xcd_order = true;
  GemmArgs g = g_in; g.perm2 = 0; g.pTs[0] = xcd_order ? 1 : 0;        // (no parity classes here: pTs[0] is this kernel's tile-order switch)
#define S(BM, BN, MODE) do { const size_t ring = (size_t)3 * BN * 64 + (size_t)2 * (BM + 64) * 64 + 512 + (BM / 64 - 1) * 2048, epi = (size_t)64 * (BN + 4) * 4 + 10 * BN * 4; const size_t lds = ring > epi ? ring : epi; \
    dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)((g.N + BN - 1) / BN)); \
    avec_note_kernel("conv3x3_shift_kernel<%d,%d,%d>", BM, BN, MODE); if (int r = want_lds(conv3x3_shift_kernel<BM, BN, MODE>, lds)) return r; hipLaunchKernelGGL((conv3x3_shift_kernel<BM, BN, MODE>), grid, dim3(256), lds, st, g); return 0; } while (0)
  // 256-row tiles halve the weight-tile DMA per FLOP (measured 5-15 % on the 3200-image ResNet stages 2-3, slower once fewer than ~3 tiles per CU remain)
  static const int bm_env = getenv("AVEC_SHIFT_BM") ? atoi(getenv("AVEC_SHIFT_BM")) : 0;
  // tile height by wave quantisation: workgroups / (rounds * resident slots), slots = 256 CUs x 3 (128 rows, 136 VGPRs) or x 2 (256 rows, 237 VGPRs);
  // e.g. the 512-channel stage (28 800 rows x 512): 900 tiles of 128 rows = 1.17 rounds (59 %), 452 of 256 rows = 0.88 round (88 %)
  const long long t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128), t256 = ((g.M + 255) / 256) * ((g.N + 127) / 128);
  const double e128 = (double)t128 / (double)(((t128 + 767) / 768) * 768), e256 = 1.08 * (double)t256 / (double)(((t256 + 511) / 512) * 512);
  const bool tr_ok = epi_tr_ok(g) && !g.e.res_cls0;
#define ST(BM, BN, MODE) do { const size_t ring = (size_t)3 * BN * 64 + (size_t)2 * (BM + 64) * 64 + 512 + (BM / 64 - 1) * 2048; const size_t lds = ring; \
    dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)((g.N + BN - 1) / BN)); \
    avec_note_kernel("conv3x3_shift_kernel<%d,%d,%d,tr>", BM, BN, MODE); if (int r = want_lds(conv3x3_shift_kernel<BM, BN, MODE, true>, lds)) return r; hipLaunchKernelGGL((conv3x3_shift_kernel<BM, BN, MODE, true>), grid, dim3(256), lds, st, g); return 0; } while (0)
  if (g.N >= 128 && (bm_env == 256 || (bm_env == 0 && e256 > e128) || (g.e.res_mask && tr_ok))) {
    if (tr_ok) { if (mode == MODE_CONV_FWD) ST(256, 128, MODE_CONV_FWD); else ST(256, 128, MODE_CONV_BWD); }
    if (mode == MODE_CONV_FWD) S(256, 128, MODE_CONV_FWD); else S(256, 128, MODE_CONV_BWD);
  }
#undef ST
  if (g.e.res_mask) { avec_set_error("gemm_nt: res_mask needs the register-direct epilogue (N >= 128, bf16 output and residual in 16-byte pieces)"); return -1; }
  if (g.N >= 128) { if (mode == MODE_CONV_FWD) S(128, 128, MODE_CONV_FWD); else S(128, 128, MODE_CONV_BWD); }
  if (mode == MODE_CONV_FWD) S(128, 64, MODE_CONV_FWD); else S(128, 64, MODE_CONV_BWD);
#undef S
  return 0;
}

template <typename T>
static int launch_nt(const GemmArgs& g, int mode, int src_f32, hipStream_t st) {
  // tile choice: big tiles only when they still fill the chip (256 CUs)
  long long t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128);
  if (g.N > 64 && t128 >= 384) return launch_nt_mode<T, 128, 128>(g, mode, src_f32, st);
  // plain bf16 products with whole 16-byte K-chunks: the lean 64 x 64 kernel beats the general 128 x 64 one up to the sizes the model has (3200 x 1024 x 256: 11.3 vs 11.8 us)
  static const bool no_lean = false;
  const bool lean = sizeof(T) == 2 && mode == MODE_PLAIN && !src_f32 && !no_lean && g.K % 8 == 0 && ((g.M + 63) / 64) * ((g.N + 63) / 64) <= 4096;      // (K = 8n + 4 with many tiles: the general 128 x 64 kernel is faster, 14.3 vs 16.0 us at 6400 x 720 x 180; the lean kernel takes those shapes only where 64 x 64 tiles are chosen anyway)
  if (!lean && ((g.M + 127) / 128) * ((g.N + 63) / 64) >= 384) return launch_nt_mode<T, 128, 64>(g, mode, src_f32, st);
  return launch_nt_mode<T, 64, 64>(g, mode, src_f32, st);
}

extern "C" int avec_gemm_nt(int dtype, const void* A, const avec_rows_t* a_rows, int a_mode, int a_f32,
                            const void* W, long long ldw, long long M, int N, int K,
                            const avec_epilogue_t* ep, hipStream_t stream) {
  AVEC_CHECK_ARG(dtype == AVEC_F32 || dtype == AVEC_BF16, "gemm_nt: bad dtype %d", dtype);
  AVEC_CHECK_ARG(A && W && ep && ep->out && a_rows, "gemm_nt: null pointer");
  AVEC_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_nt: bad dims M=%lld N=%d K=%d", M, N, K);
  AVEC_CHECK_ARG(a_mode >= 0 && a_mode <= 2, "gemm_nt: bad a_mode %d", a_mode);
  const int vec = dtype == AVEC_BF16 ? 8 : 4;
  AVEC_CHECK_ARG(K >= vec && (dtype == AVEC_F32 || (K % 2 == 0 && ldw % 2 == 0)), "gemm_nt: K=%d / ldw=%lld: need K >= %d and (bf16) even K, ldw (dword-aligned chunks)", K, ldw, vec);
  AVEC_CHECK_ARG(a_mode != AVEC_ROWS_PLAIN || dtype == AVEC_F32 || a_rows->ld % 2 == 0, "gemm_nt: lda=%lld must be even", a_rows->ld);
  AVEC_CHECK_ARG(!(a_f32 && dtype == AVEC_BF16) || K % 4 == 0, "gemm_nt: fp32-source staging needs K %% 4 == 0 (K=%d)", K);
  AVEC_CHECK_ARG(a_mode == AVEC_ROWS_PLAIN || (a_rows->C % vec == 0 && K == a_rows->KH * a_rows->KW * a_rows->C),
                 "gemm_nt: conv C=%d must be a multiple of %d and K = KH*KW*C", a_rows->C, vec);
  GemmArgs g; g.a = make_src(A, a_rows); g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
  g.fast_conv = 0; g.perm2 = 0; g.pImgs = 0; g.ktail = 0; for (int c = 0; c < 5; ++c) g.pTs[c] = 0;
  if (a_mode != AVEC_ROWS_PLAIN) {
    const int KE = dtype == AVEC_BF16 ? 64 : 32;
    const long long imgs = a_mode == MODE_CONV_FWD ? (M + (long long)a_rows->OH * a_rows->OW - 1) / ((long long)a_rows->OH * a_rows->OW) : (M + (long long)a_rows->H * a_rows->W - 1) / ((long long)a_rows->H * a_rows->W);
    const long long src_elems = imgs * (a_mode == MODE_CONV_FWD ? (long long)a_rows->H * a_rows->W : (long long)a_rows->OH * a_rows->OW) * a_rows->C;
    static const bool no_fast = false;
    g.fast_conv = !no_fast && a_rows->C % KE == 0 && a_rows->KH * a_rows->KW <= 32 && src_elems + (long long)(a_rows->W + a_rows->OW + 2) * a_rows->C * 4 < (1ll << 31) &&
                  (a_mode == MODE_CONV_FWD || a_rows->stride == 1 || a_rows->stride == 2);
    static const bool no_perm = getenv("AVEC_NO_PERM2") != nullptr;
    if (!no_perm && dtype == AVEC_BF16 && g.fast_conv && a_mode == MODE_CONV_BWD && a_rows->stride == 2 && M == imgs * (long long)a_rows->H * a_rows->W) {
      g.perm2 = 1; g.pImgs = imgs;
    }
  }
  Epi& e = g.e;
  e.out = ep->out; e.ldo = ep->ldo; e.out_f32 = ep->out_f32; e.out_pre = ep->out_pre; e.ldpre = ep->ldpre; e.bias = ep->bias;
  e.act = ep->act; e.drop_p = ep->drop_p; e.rng = (const unsigned long long*)ep->rng; e.stream = ep->rng_stream;
  e.res = ep->res; e.ldres = ep->ldres; e.alpha = ep->alpha; e.res_act = ep->res_act; e.dact_z = ep->dact_z; e.ldz = ep->ldz; e.dact = ep->dact;
  e.colsum = ep->colsum; e.stats = ep->stats;
  e.bnb_y = ep->bnb_y; e.ldby = ep->ldby; e.bnb_ss = ep->bnb_ss; e.bnb_mask = ep->bnb_mask; e.res_cls0 = ep->res_cls0; e.res_mask = ep->res_mask;
  AVEC_CHECK_ARG(!e.res_mask || (dtype == AVEC_BF16 && !a_f32 && a_mode != AVEC_ROWS_PLAIN && e.res && e.res_act && !e.res_cls0 && e.ldres % 8 == 0),
                 "gemm_nt: res_mask needs a bf16 convolution product with a bf16 residual whose rows are whole 8-element pieces");
  AVEC_CHECK_ARG(!e.res_cls0 || (g.perm2 && e.res), "gemm_nt: res_cls0 needs a bf16 stride-2 backward-data product that runs in parity-class order");
  AVEC_CHECK_ARG(!e.bnb_y || (e.stats && N % 4 == 0 && !(e.ldo & 3) && !(e.ldby & 3) && !(e.ldres & 3) && !(e.ldz & 3) && !(e.ldpre & 3) && (!e.bnb_mask || e.bnb_ss) && (e.bnb_mask || e.dact == 2 || e.dact == 0)),
                 "gemm_nt: the BatchNorm-backward fusion needs stats, N %% 4 == 0 and row strides that are multiples of 4");
  AVEC_CHECK_ARG(!(e.drop_p > 0.f) || e.rng, "gemm_nt: dropout without rng state");
  int r = 1;
  if (dtype == AVEC_BF16 && !a_f32 && a_mode != AVEC_ROWS_PLAIN) r = launch_conv_shift(g, a_mode, stream);      // 3x3 / stride 1: shifted-window kernel
  if (r == 1 && dtype == AVEC_BF16 && !a_f32 && a_mode != AVEC_ROWS_PLAIN) r = avec_launch_conv_s2(g, a_mode, stream);   // 3x3 / stride 2: shifted windows over the parity classes
  if (r == 1) r = (dtype == AVEC_BF16) ? launch_nt<bf16>(g, a_mode, a_f32, stream) : launch_nt<float>(g, a_mode, a_f32, stream);
  if (r) return r;
  AVEC_LAUNCH_CHECK();
  return 0;
}


/* fp8 forward Linear product (include/avec_hip.h) */
extern "C" int avec_gemm_nt_fp8(const void* A, long long lda, const void* W, long long ldw, long long M, int N, int K,
                                const float* amax_a, const float* amax_w, const avec_epilogue_t* ep, hipStream_t stream) {
  AVEC_CHECK_ARG(A && W && ep && ep->out && amax_a && amax_w, "gemm_nt_fp8: null pointer");
  AVEC_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 16 == 0 && lda % 16 == 0 && ldw % 16 == 0 && aligned16(A) && aligned16(W),
                 "gemm_nt_fp8: M=%lld N=%d K=%d lda=%lld ldw=%lld: K and the row strides must be multiples of 16 bytes, operands 16-byte aligned", M, N, K, lda, ldw);
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.a.ptr = A; g.a.ld = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
  Epi& e = g.e;
  e.out = ep->out; e.ldo = ep->ldo; e.out_f32 = ep->out_f32; e.out_pre = ep->out_pre; e.ldpre = ep->ldpre; e.bias = ep->bias;
  e.act = ep->act; e.drop_p = ep->drop_p; e.rng = (const unsigned long long*)ep->rng; e.stream = ep->rng_stream;
  e.res = ep->res; e.ldres = ep->ldres; e.alpha = ep->alpha; e.res_act = ep->res_act; e.dact_z = ep->dact_z; e.ldz = ep->ldz; e.dact = ep->dact;
  e.colsum = ep->colsum; e.stats = ep->stats;
  AVEC_CHECK_ARG(!(e.drop_p > 0.f) || e.rng, "gemm_nt_fp8: dropout without rng state");
#define F8(BM, BN, S_) do { dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN)); \
    const size_t ring = (size_t)S_ * (BM + BN) * 128, epi = (size_t)64 * (BN + 4) * 4 + 10 * BN * 4; const size_t lds = ring > epi ? ring : epi; \
    if (int r = want_lds(gemm_nt_fp8_kernel<BM, BN, S_>, lds)) return r; \
    hipLaunchKernelGGL((gemm_nt_fp8_kernel<BM, BN, S_>), grid, dim3(256), lds, stream, g, amax_a, amax_w); } while (0)
  const long long t128 = ((M + 127) / 128) * ((N + 127) / 128);
  if (N > 64 && t128 >= 384) F8(128, 128, 2);
  else if (((M + 127) / 128) * ((N + 63) / 64) >= 384) F8(128, 64, 2);
  else F8(64, 64, 4);
#undef F8
  AVEC_LAUNCH_CHECK();
  return 0;
}

static thread_local TnMulti* tn_collect = nullptr;      // set while avec_gemm_tn_batched_multi gathers its problems
template <typename T, int BI, int BJ>
static int launch_tn_tile(TnArgs g, int mode, int q_f32, int nbatch, hipStream_t st) {
  constexpr int KE = BKB / (int)sizeof(T);
  int tiles = ((g.I + BI - 1) / BI) * ((g.J + BJ - 1) / BJ) * nbatch;
  long long ksteps = (g.M + KE - 1) / KE;
  // split the reduction so that ~512 workgroups exist, but keep >= 4 K-steps per workgroup: every split costs BI*BJ fp32 atomics
  // measured on MI355X: the implicit-GEMM weight gradients keep improving up to ~2048 workgroups (8 per CU; 183 -> 255 TFLOP/s on the
  // 64->64 3x3 layer), the plain small-K products are best around 512
  static const long long wg_env = getenv("AVEC_TN_WGS") ? atoll(getenv("AVEC_TN_WGS")) : 0;
  // re-measured per ResNet stage (tools/bench_gemm.py, AVEC_TN_WGS sweep): few output tiles (stage 2: 9) are best at ~1024 workgroups (232 vs 255 us), 36 tiles
  // (stage 3) at 2048, 144 tiles (stage 4) at 4096 (358 vs 378 us): the split has to cover the chip several times over, but every workgroup pays BI*BJ atomics
  // (end of round 3, after the gather loop lost its carry chain: the split should fill the ~1024 resident slots ONCE -- 4 workgroups of 32 KB per CU -- and no more: 72 tiles
  //  x 12 slices = 864 workgroups take 127 us on the 512-channel stride-2 layer, 72 x 15 = 1080 take 154 us, the old 4096 target 227 us; tools/gpu/r3_tnkt.sh)
  const long long wg_target = wg_env > 0 ? wg_env : (mode != MODE_PLAIN ? 896 : 512);
  long long split = (wg_target + tiles - 1) / tiles; if (split > ksteps / 4) split = ksteps / 4; if (split < 1 || g.Oact) split = 1;
  long long per = ((ksteps + split - 1) / split) * KE;
  split = (g.M + per - 1) / per;
  g.m_per_block = (int)per; g.split = (int)split;
  dim3 grid((g.I + BI - 1) / BI, (g.J + BJ - 1) / BJ, (unsigned)(split * nbatch));
  size_t lds = (size_t)2 * (BI + BJ) * LDS_ROW;
  constexpr int VEC = Elt<T>::VEC;
  const bool f32src = q_f32 && sizeof(T) == 2;
  const bool a16 = nbatch == 1 && aligned16(g.P) && aligned16(g.q.ptr) && g.Iq % VEC == 0 && g.Jq % VEC == 0 && g.ldp % VEC == 0 &&
                   (mode != MODE_PLAIN || (g.q.ld % (f32src ? 4 : VEC) == 0));
  if (sizeof(T) == 2 && a16 && !f32src && !g.Oact && (mode == MODE_PLAIN || g.q_mg != 0)) {        // bf16: LDS-DMA + transposed-read kernel (gathered operand: images of <= 4096 output pixels)
    static const bool use_tr = true;
    if (use_tr) {
      // reduction rows per LDS tile: 32 (two 16 KB stages for 128x128) keeps 4 workgroups resident per CU; measured on the ResNet weight
      // gradients 1.2-2.1x over 64-row tiles (2 per CU), and 4-stage rings (1 per CU) are 1.5-2x slower: occupancy hides the HBM latency
      static const int kt_env = getenv("AVEC_TN_KT") ? atoi(getenv("AVEC_TN_KT")) : 32;
      const long long q_elems = mode == MODE_PLAIN ? 0 : ((g.M + (long long)g.q.OH * g.q.OW - 1) / ((long long)g.q.OH * g.q.OW) + 1) * g.q.H * g.q.W * g.q.C;
      const bool q32 = q_elems < (1ll << 31);
#define LT(MODE, Q32, KT_) do { const size_t l2 = (size_t)2 * KT_ * (BI + BJ) * 2; g.q_nwrap = (KT_ + g.q_ohw - 1) / g.q_ohw; avec_note_kernel("gemm_tn_tr_kernel<%d,%d,%d,2,%d,%d>", BI, BJ, MODE, (int)Q32, KT_); if (int r = want_lds(gemm_tn_tr_kernel<BI, BJ, MODE, 2, Q32, KT_>, l2)) return r; \
        hipLaunchKernelGGL((gemm_tn_tr_kernel<BI, BJ, MODE, 2, Q32, KT_>), grid, dim3(256), l2, st, g); return 0; } while (0)
#define LK(KT_) do { if (mode == MODE_PLAIN) LT(MODE_PLAIN, false, KT_); else if (q32) LT(MODE_CONV_FWD, true, KT_); else LT(MODE_CONV_FWD, false, KT_); } while (0)
      if (kt_env == 64) LK(64); else LK(32);
#undef LK
#undef LT
    }
  }
  if (g.pcs) {      // kernels without the fused column sums: a separate pass (plain atomics only when weight gradients run on their own stream)
    static const bool side_wgrad = false;
    if (int r = colsum_launch(sizeof(T) == 2 ? AVEC_BF16 : AVEC_F32, g.P, g.ldp, g.pcs, g.M, g.I, !side_wgrad, st)) return r;
  }     // kernels without the fused column sums
#define L(MODE, F, A) do { avec_note_kernel("gemm_tn_kernel<%s,%d,%d,%d,%d,%d>", (sizeof(T) == 2 ? "bf16" : "float"), BI, BJ, MODE, (int)F, (int)A); if (int r = want_lds(gemm_tn_kernel<T, BI, BJ, MODE, F, A>, lds)) return r; hipLaunchKernelGGL((gemm_tn_kernel<T, BI, BJ, MODE, F, A>), grid, dim3(256), lds, st, g); } while (0)
  if (tn_collect && sizeof(T) == 2 && BI == 64 && BJ == 64 && mode == MODE_PLAIN && !f32src && !a16 && !g.pcs && tn_collect->n < AVEC_TN_MULTI_MAX) {
    TnMulti& m = *tn_collect; const int k = m.n++;           // (avec_gemm_tn_batched_multi: this problem joins the common launch)
    m.g[k] = g; m.gx[k] = (int)grid.x; m.gy[k] = (int)grid.y; m.first[k + 1] = m.first[k] + (int)(grid.x * grid.y * grid.z);
    return 0;
  }
  if (mode == MODE_PLAIN) {
    if (f32src) { if (a16) L(MODE_PLAIN, true, true); else L(MODE_PLAIN, true, false); }
    else { if (a16) L(MODE_PLAIN, false, true); else L(MODE_PLAIN, false, false); }
  } else { if (a16) L(MODE_CONV_FWD, false, true); else L(MODE_CONV_FWD, false, false); }
#undef L
  return 0;
}

static int gemm_tn_impl(int dtype, const void* P, long long ldp, const void* Q, const avec_rows_t* q_rows, int q_mode, int q_f32,
                        float* O, void* Oact, long long ldo, long long M, int I, int J, int nb_outer, int nb_inner, const long long* strides, float* p_colsum, hipStream_t stream) {
  AVEC_CHECK_ARG(dtype == AVEC_F32 || dtype == AVEC_BF16, "gemm_tn: bad dtype %d", dtype);
  AVEC_CHECK_ARG(P && Q && (O || Oact) && q_rows, "gemm_tn: null pointer");
  AVEC_CHECK_ARG(M > 0 && I > 0 && J > 0 && nb_outer > 0 && nb_inner > 0, "gemm_tn: bad dims");
  AVEC_CHECK_ARG(q_mode == MODE_PLAIN || q_mode == MODE_CONV_FWD, "gemm_tn: bad q_mode %d", q_mode);
  const int vec = dtype == AVEC_BF16 ? 8 : 4;
  // rows padded to a multiple of the vector width (stem im2col matrix J=245 in rows of 248; attention P/dS): load the padding, bound the output by I / J
  const int Jq = (q_mode == MODE_PLAIN && q_rows->ld >= (J + vec - 1) / vec * vec) ? (J + vec - 1) / vec * vec : J;
  const int Iq = (ldp >= (I + vec - 1) / vec * vec) ? (I + vec - 1) / vec * vec : I;
  AVEC_CHECK_ARG(Iq >= vec && Jq >= vec && (dtype == AVEC_F32 || (Iq % 2 == 0 && Jq % 2 == 0 && ldp % 2 == 0)), "gemm_tn: I=%d, J=%d: need >= %d and (bf16) even I, J, ldp", I, J, vec);
  AVEC_CHECK_ARG(q_mode != MODE_PLAIN || dtype == AVEC_F32 || q_rows->ld % 2 == 0, "gemm_tn: ldq must be even");
  AVEC_CHECK_ARG(q_mode == MODE_PLAIN || q_rows->C % vec == 0, "gemm_tn: conv C=%d must be a multiple of %d", q_rows->C, vec);
  AVEC_CHECK_ARG(!(q_f32 && dtype == AVEC_BF16) || Jq % 4 == 0, "gemm_tn: fp32-source staging needs J %% 4 == 0");
  TnArgs g; g.P = P; g.ldp = ldp; g.q = make_src(Q, q_rows); g.O = O; g.Oact = Oact; g.ldo = ldo; g.M = M; g.I = I; g.J = J; g.Iq = Iq; g.Jq = Jq; g.m_per_block = 0; g.pcs = p_colsum;
  g.split = 1; g.nb_inner = nb_inner;
  g.q_ohw = 1; g.q_mg = 0; g.q_nwrap = 0;
  if (q_mode != MODE_PLAIN && q_rows->OW > 0 && q_rows->OH > 0 && (long long)q_rows->OH * q_rows->OW <= 4096) {      // pixel -> (row, column) by multiply-shift, verified here
    const int ohw = q_rows->OH * q_rows->OW; const unsigned mg = ((1u << 20) + (unsigned)q_rows->OW - 1u) / (unsigned)q_rows->OW;
    bool exact = true; for (int px = 0; px < ohw && exact; ++px) exact = (int)(((unsigned)px * mg) >> 20) == px / q_rows->OW;
    g.q_ohw = ohw; g.q_mg = exact ? (int)mg : 0;
  }
  { static const bool no_xcd = false; g.xcd_map = no_xcd ? 0 : 1; }
  g.sPo = strides ? strides[0] : 0; g.sPi = strides ? strides[1] : 0; g.sQo = strides ? strides[2] : 0; g.sQi = strides ? strides[3] : 0;
  g.sOo = strides ? strides[4] : 0; g.sOi = strides ? strides[5] : 0;
  // P batches stay dword aligned; Q batches may start on any element (heads of odd width, d = 45: the plain-load kernels issue unaligned dword loads, which gfx950 serves)
  if (dtype == AVEC_BF16 && strides) for (int i = 0; i < 2; ++i) AVEC_CHECK_ARG(strides[i] % 2 == 0, "gemm_tn: bf16 batch strides of P must be even");
  const int nbatch = nb_outer * nb_inner;
  // few output tiles AND a short reduction (conformer weight gradients): 64x64 tiles fill the chip with less atomic traffic; long reductions
  // (conv weight gradients, M ~ 1e5..1e6) amortise the atomics and prefer the more efficient 128x128 tile
  bool big = (I >= 128 && J >= 128) && ((long long)((I + 127) / 128) * ((J + 127) / 128) * nbatch >= 48 || M >= 32768);
  int r;
  // narrow-I long reductions (64-channel conv layers, the stem): a 64x128 tile re-reads P half as often as 64x64 (these launches are bound by L2/HBM traffic)
  const bool wide = !big && I <= 64 && J >= 128 && M >= 32768 && dtype == AVEC_BF16 && !q_f32 && nbatch == 1 && aligned16(P) && aligned16(Q) && Iq % 8 == 0 && Jq % 8 == 0 &&
                    ldp % 8 == 0 && (q_mode != MODE_PLAIN || q_rows->ld % 8 == 0);      // (only the transposed-read kernel is instantiated for this shape in practice)
  if (dtype == AVEC_BF16) r = big ? launch_tn_tile<bf16, 128, 128>(g, q_mode, q_f32, nbatch, stream) : wide ? launch_tn_tile<bf16, 64, 128>(g, q_mode, q_f32, nbatch, stream)
                                                                                                    : launch_tn_tile<bf16, 64, 64>(g, q_mode, q_f32, nbatch, stream);
  else r = big ? launch_tn_tile<float, 128, 128>(g, q_mode, q_f32, nbatch, stream) : launch_tn_tile<float, 64, 64>(g, q_mode, q_f32, nbatch, stream);
  if (r) return r;
  AVEC_LAUNCH_CHECK();
  return 0;
}

extern "C" int avec_gemm_tn(int dtype, const void* P, long long ldp, const void* Q, const avec_rows_t* q_rows, int q_mode, int q_f32,
                            float* O, long long ldo, long long M, int I, int J, hipStream_t stream) {
  return gemm_tn_impl(dtype, P, ldp, Q, q_rows, q_mode, q_f32, O, nullptr, ldo, M, I, J, 1, 1, nullptr, nullptr, stream);
}
extern "C" int avec_gemm_tn_bias(int dtype, const void* P, long long ldp, const void* Q, const avec_rows_t* q_rows, int q_mode, int q_f32,
                                 float* O, long long ldo, float* p_colsum, long long M, int I, int J, hipStream_t stream) {
  AVEC_CHECK_ARG(!p_colsum || (I % 4 == 0 && ldp % 4 == 0), "gemm_tn_bias: column sums need I %% 4 == 0 and ldp %% 4 == 0");
  return gemm_tn_impl(dtype, P, ldp, Q, q_rows, q_mode, q_f32, O, nullptr, ldo, M, I, J, 1, 1, nullptr, p_colsum, stream);
}

extern "C" int avec_gemm_tn_batched(int dtype, const void* P, long long ldp, const void* Q, long long ldq, float* O, long long ldo, long long M, int I, int J,
                                    int nb_outer, int nb_inner, const long long* strides6, hipStream_t stream) {
  avec_rows_t rows = {}; rows.ld = ldq;
  AVEC_CHECK_ARG(strides6, "gemm_tn_batched: null strides");
  return gemm_tn_impl(dtype, P, ldp, Q, &rows, MODE_PLAIN, 0, O, nullptr, ldo, M, I, J, nb_outer, nb_inner, strides6, nullptr, stream);
}
extern "C" int avec_gemm_tn_batched_store(int dtype, const void* P, long long ldp, const void* Q, long long ldq, void* O_act, long long ldo, long long M, int I, int J,
                                          int nb_outer, int nb_inner, const long long* strides6, hipStream_t stream) {
  avec_rows_t rows = {}; rows.ld = ldq;
  AVEC_CHECK_ARG(strides6 && O_act, "gemm_tn_batched_store: null pointer");
  return gemm_tn_impl(dtype, P, ldp, Q, &rows, MODE_PLAIN, 0, nullptr, O_act, ldo, M, I, J, nb_outer, nb_inner, strides6, nullptr, stream);
}

extern "C" int avec_gemm_tn_batched_multi(int dtype, const avec_tn_batched_t* items, int n, hipStream_t stream) {
  AVEC_CHECK_ARG(items && n >= 1 && n <= AVEC_TN_MULTI_MAX, "gemm_tn_batched_multi: need 1..%d problems (got %d)", AVEC_TN_MULTI_MAX, n);
  static const bool off = false;
  TnMulti m; m.n = 0; m.first[0] = 0;
  if (!off) tn_collect = &m;
  int rc = 0;
  for (int k = 0; k < n && rc == 0; ++k) {               // a problem that does not take the common small-tile kernel is launched on its own right here
    const avec_tn_batched_t& t = items[k];
    avec_rows_t rows = {}; rows.ld = t.ldq;
    if (!t.strides6 || (!t.O && !t.O_act)) { avec_set_error("gemm_tn_batched_multi: problem %d: null pointer", k); rc = -1; break; }
    rc = gemm_tn_impl(dtype, t.P, t.ldp, t.Q, &rows, MODE_PLAIN, 0, t.O_act ? nullptr : t.O, t.O_act, t.ldo, t.M, t.I, t.J, t.nb_outer, t.nb_inner, t.strides6, nullptr, stream);
  }
  tn_collect = nullptr;
  if (rc) return rc;
  if (m.n > 0) {
    const size_t lds = (size_t)2 * (64 + 64) * LDS_ROW;
    avec_note_kernel("gemm_tn_multi_kernel<%d>", m.n);
    if (int r = want_lds(gemm_tn_multi_kernel, lds)) return r;
    hipLaunchKernelGGL(gemm_tn_multi_kernel, dim3((unsigned)m.first[m.n]), dim3(256), lds, stream, m);
    AVEC_LAUNCH_CHECK();
  }
  return 0;
}

// ---- grouped weight gradients -------------------------------------------------------------------------------------------------------------
static bool tn_item_ok(int dtype, const avec_tn_item_t& t) {
  if (dtype != AVEC_BF16 || !t.P || !t.Q || !t.O || t.M <= 0 || t.I <= 0 || t.J <= 0) return false;
  // No alignment requirement: the LDS-DMA takes any source alignment (tools/glds_align_probe.hip).  Rows whose width is not a multiple of 8 elements are
  // read in whole 16-byte chunks up to the next multiple of 8 -- past the row end into the next row, and past the LAST row by up to 14 bytes, which the
  // caller must keep readable (include/avec_hip.h); whatever is read there only reaches output rows / columns >= I / J, which are not stored.
  if (t.ldp < t.I || t.ldq < t.J) return false;
  if (t.p_colsum && (t.I % 4 || t.ldp % 4)) return false;
  return t.M < (1ll << 31) && t.ldp < (1ll << 31) && t.ldq < (1ll << 31) && t.ldo < (1ll << 31);
}
extern "C" int avec_gemm_tn_grouped_ok(int dtype, const avec_tn_item_t* item) { return item && tn_item_ok(dtype, *item) ? 1 : 0; }

extern "C" int avec_gemm_tn_grouped(int dtype, const avec_tn_item_t* items, int n, hipStream_t stream) {
  AVEC_CHECK_ARG(items && n > 0 && n <= AVEC_TN_GROUP_MAX, "gemm_tn_grouped: need 1..%d items (got %d)", AVEC_TN_GROUP_MAX, n);
  for (int k = 0; k < n; ++k) AVEC_CHECK_ARG(tn_item_ok(dtype, items[k]), "gemm_tn_grouped: item %d is not eligible (bf16 operands, row strides >= widths, bias sums need I %% 4 == 0)", k);
  // tile size: 128x128 tiles read each operand byte half as often as 64x64 (these products are bound by L2 traffic), but a group must still cover the
  // chip: take the big tile when the group has enough of them
  static const int bt_env = 0;
  long long t128 = 0;
  for (int k = 0; k < n; ++k) t128 += (long long)((items[k].I + 127) / 128) * ((items[k].J + 127) / 128);
  const int BT = bt_env == 64 || bt_env == 128 ? bt_env : (t128 >= 96 ? 128 : 64);
  // common reduction-slice length: the largest multiple of 64 rows (>= 256) that still yields ~wg_target workgroups over the whole group
  static const long long wg_env = 0;
  const long long wg_target = wg_env > 0 ? wg_env : 512;       // every slice costs I*J atomics: 256-512 beat 1536 by 0.2 ms in the step; with the lean DMA loop 512 beats 384 (21.14 vs 21.24 ms, two same-box sweeps)
  long long per = 256;
  for (long long cand = 8192; cand >= 256; cand -= 64) {
    long long wgs = 0;
    for (int k = 0; k < n; ++k) wgs += (long long)((items[k].I + BT - 1) / BT) * ((items[k].J + BT - 1) / BT) * ((items[k].M + cand - 1) / cand);
    if (wgs >= wg_target) { per = cand; break; }
  }
  TnGroup grp; grp.n = n;
  int first = 0;
  for (int k = 0; k < n; ++k) {
    const avec_tn_item_t& s = items[k]; TnItem& t = grp.it[k];
    t.P = s.P; t.Q = s.Q; t.O = s.O; t.pcs = s.p_colsum; t.ldp = (int)s.ldp; t.ldq = (int)s.ldq; t.ldo = (int)s.ldo; t.M = (int)s.M; t.I = s.I; t.J = s.J;
    t.Iq = (s.I + 7) / 8 * 8; t.Jq = (s.J + 7) / 8 * 8;
    t.rows_out = s.q_rows_out > 0 ? s.q_rows_out : 1; t.rows_in = s.q_rows_in > 0 ? s.q_rows_in : 1; t.step = s.q_step;
    long long mp = per; if (mp > s.M) mp = (s.M + 63) / 64 * 64;
    t.m_per_block = (int)mp; t.split = (int)((s.M + mp - 1) / mp);
    t.gx = (s.I + BT - 1) / BT; t.gy = (s.J + BT - 1) / BT; t.first = first;
    first += t.gx * t.gy * t.split;
  }
  grp.total = first;
  static const bool no_xcd = false;
  grp.xcd_map = no_xcd ? 0 : 1;
  static const int kt_env = 64;       // reduction rows per LDS tile
  static const int stg_env = 2;
  const int KT = kt_env == 32 ? 32 : 64, STG = (stg_env == 4 && KT == 32) ? 4 : 2;       // (2 x 64 rows and 4 x 32 rows measure the same, 3 stages are slower: profiles/r03_tn_grouped.txt)
  const size_t lds = (size_t)STG * KT * (BT + BT) * 2;
  avec_note_kernel("gemm_tn_tr_grouped_kernel<%d,%d,%d>", BT, STG, KT);
#define TNG(BT_, S_, K_) do { if (BT == BT_ && STG == S_ && KT == K_) { if (int r = want_lds(gemm_tn_tr_grouped_kernel<BT_, S_, K_>, lds)) return r; \
    hipLaunchKernelGGL((gemm_tn_tr_grouped_kernel<BT_, S_, K_>), dim3((unsigned)first), dim3(256), lds, stream, grp); } } while (0)
  TNG(128, 2, 32); TNG(128, 2, 64); TNG(64, 2, 32); TNG(64, 2, 64); TNG(128, 4, 32); TNG(64, 4, 32);
#undef TNG
  AVEC_LAUNCH_CHECK();
  return 0;
}
