// Experiment (round 3): bf16 NT GEMM C[M][N] = A[M][K] * W[N][K]^T, 256x256 tile, BK = 64, 8 waves, one workgroup per CU, FOUR PHASES per K-tile:
//   LDS: 2 K-tile buffers x {A0, A1, B0, B1} half-tiles of 128 rows x 128 B (16 KB each) = 128 KB, rows XOR-swizzled ((row >> 1) & 7) on the DMA source side.
//   wave (wr, wc) owns rows 64 wr .. + 64 of BOTH A halves and columns 32 wc .. + 32 of BOTH B halves (8 accumulator tiles of 32x32): a phase multiplies one
//   (A half, B half) pair = 8 MFMAs (256 cycles); the fragments are read once per K-tile (A0 + B0 in phase 0, B1 in phase 1, A1 in phase 2, none in phase 3).
//   Every phase issues ONE half-tile of the next K-tile (LDS-DMA by inline asm: hipcc must not drain it), waits with a counted vmcnt for the half the next phase
//   needs, and the two wave groups (waves 0-3 / 4-7: one of each per SIMD) run one barrier interval apart: one group's read + DMA-issue section meets the other
//   group's MFMA section.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_gemm_p8 tools/ubench_gemm_p8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef unsigned short bf16raw;

__device__ __attribute__((aligned(64))) unsigned char zero16[64];

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ u32x4 lds_read128(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BAR() __builtin_amdgcn_s_barrier()

#ifndef P8_TRACE
#define P8_TRACE 0
#endif
#define TR_KT0 16
#define TR_NKT 6
#define TR_PTS 16
// VARIANT bit 5 (32): the half-tile DMA is issued BEHIND the MFMAs of its phase (the issue blocks the wave while the CU's vector-memory path is busy)
// VARIANT bit 0: no stagger; bit 1: no setprio; bit 2: no DMA inside the loop (stale buffers); bit 3: no fragment reads inside the loop; bit 4: no MFMA
// -DP8_TRACE=1: waves 0 and 4 of workgroup (0, 0) stamp s_memtime at four points of every phase of K-tiles TR_KT0 .. + TR_NKT (the stamp waits lgkmcnt(0): only where
// the loop does so anyway, or where nothing is outstanding)
template <int VARIANT>
__global__ __launch_bounds__(512, 1) void gemm_p8_kernel(const bf16raw* __restrict__ A, const bf16raw* __restrict__ W, bf16raw* __restrict__ C, int M, int N, int K, unsigned long long* __restrict__ trace) {
  constexpr int HALF = 128 * 128;            // bytes of a half-tile
  constexpr int BUF = 4 * HALF;              // A0 A1 B0 B1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 1, wc = (wave >> 1) & 3, grp = wave >> 2;
  // XCD-aware remap: consecutive tiles of one N-column stay on one XCD
  const int nbx = gridDim.x, nby = gridDim.y;
  int bid = blockIdx.y * nbx + blockIdx.x;
  const int nwg = nbx * nby, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  const int by = bid / nbx, bx = bid % nbx;
  const long long m0 = (long long)bx * 256; const int n0 = by * 256;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  // DMA plan: half h (0: A0, 1: B0, 2: B1, 3: A1 in issue order), instruction i (0, 1): chunk id = i * 512 + tid -> row id >> 3, physical chunk id & 7
  const bf16raw* src[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = i * 512 + tid, row = id >> 3, kc = ((id & 7) ^ swz(row)) * 8;
    const long long ra0 = m0 + row, ra1 = m0 + 128 + row; const int rb0 = n0 + row, rb1 = n0 + 128 + row;
    src[0][i] = ra0 < M ? A + ra0 * K + kc : nullptr;
    src[3][i] = ra1 < M ? A + ra1 * K + kc : nullptr;
    src[1][i] = rb0 < N ? W + (long long)rb0 * K + kc : nullptr;
    src[2][i] = rb1 < N ? W + (long long)rb1 * K + kc : nullptr;
  }
  // LDS offset of half h inside a buffer: A0 0, A1 1, B0 2, B1 3
  const bool tracing = P8_TRACE && blockIdx.x == 0 && blockIdx.y == 0 && (tid == 0 || tid == 256);
  auto stamp = [&](int kt, int pt) {
    if (P8_TRACE) { const unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (tracing && kt >= TR_KT0 && kt < TR_KT0 + TR_NKT) trace[((tid >> 8) * TR_NKT + (kt - TR_KT0)) * TR_PTS + pt] = t; }
  };
  auto issue = [&](int kt, int h) {          // h in issue order
    if ((VARIANT & 4) && kt > 0) return;
    const int slot = h == 0 ? 0 : h == 3 ? 1 : h == 1 ? 2 : 3;
    const unsigned base = lds0 + (kt & 1) * BUF + slot * HALF + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const void* s = src[h][i] ? (const void*)(src[h][i] + (long long)kt * 64) : (const void*)zero16;
      glds16(s, base + i * 8192);
    }
  };
  // fragment addresses (bytes inside a half): row-tile rt of this wave's 64 rows, k-substep ks
  unsigned fa_off[2], fb_off; int fa_sw[2], fb_sw;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) { const int row = 64 * wr + 32 * rt + (lane & 31); fa_off[rt] = row * 128; fa_sw[rt] = swz(row); }
  { const int row = 32 * wc + (lane & 31); fb_off = row * 128; fb_sw = swz(row); }
  const int gsel = lane >> 5;
  f32x16 acc[2][2][2];                       // [A half][B half][row tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][t][e] = 0.f;
  const int KT = K / 64;
  // prologue: the whole first K-tile
  issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3);
  WAIT_VM(0);
  BAR();
  if (!(VARIANT & 1) && grp == 1) BAR();      // the second wave group runs one barrier interval behind
  u32x4 fA[2][4], fB0[4], fB1[4];
  if (VARIANT & 8) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { fB0[ks] = lds_read128(lds0 + 2 * HALF + fb_off + (((2 * ks + gsel) ^ fb_sw) << 4)); fB1[ks] = lds_read128(lds0 + 3 * HALF + fb_off + (((2 * ks + gsel) ^ fb_sw) << 4));
      fA[0][ks] = lds_read128(lds0 + fa_off[0] + (((2 * ks + gsel) ^ fa_sw[0]) << 4)); fA[1][ks] = lds_read128(lds0 + fa_off[1] + (((2 * ks + gsel) ^ fa_sw[1]) << 4)); }
    WAIT_LGKM0();
  }
  for (int kt = 0; kt < KT; ++kt) {
    const unsigned buf = lds0 + (kt & 1) * BUF;
    const bool more = kt + 1 < KT;
    // ---------------- phase 0: A0 x B0 ----------------
    stamp(kt, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) if (!(VARIANT & 8)) fB0[ks] = lds_read128(buf + 2 * HALF + fb_off + (((2 * ks + gsel) ^ fb_sw) << 4));
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) if (!(VARIANT & 8)) fA[rt][ks] = lds_read128(buf + 0 * HALF + fa_off[rt] + (((2 * ks + gsel) ^ fa_sw[rt]) << 4));
    if (VARIANT & 32) WAIT_VM(2);               // (issue after the MFMAs: B1(kt), A1(kt) outstanding -> B1 has landed)
    else if (more) { issue(kt + 1, 0); WAIT_VM(4); } else WAIT_VM(2);      // B1(kt) has landed (only A1(kt) [+ the half just issued] may still be in flight)
    if (P8_TRACE == 2) stamp(kt, 1);
    BAR();
    WAIT_LGKM0();
    stamp(kt, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VARIANT & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        if (VARIANT & 16) asm volatile("" :: "v"(fA[rt][ks]), "v"(fB0[ks])); else
        acc[0][0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[rt][ks]), __builtin_bit_cast(bf16x8_t, fB0[ks]), acc[0][0][rt], 0, 0, 0);
    stamp(kt, 3);
    if (!(VARIANT & 2)) __builtin_amdgcn_s_setprio(0);
    if ((VARIANT & 32) && more) issue(kt + 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    BAR();
    // ---------------- phase 1: A0 x B1 ----------------
    stamp(kt, 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) if (!(VARIANT & 8)) fB1[ks] = lds_read128(buf + 3 * HALF + fb_off + (((2 * ks + gsel) ^ fb_sw) << 4));
    if (VARIANT & 32) { if (more) WAIT_VM(2); else WAIT_VM(0); }      // (A1(kt), A0(kt+1) outstanding -> A1 has landed)
    else if (more) { issue(kt + 1, 1); WAIT_VM(4); } else WAIT_VM(0);      // A1(kt) has landed
    if (P8_TRACE == 2) stamp(kt, 5);
    BAR();
    WAIT_LGKM0();
    stamp(kt, 6);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VARIANT & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        if (VARIANT & 16) asm volatile("" :: "v"(fA[rt][ks]), "v"(fB1[ks])); else
        acc[0][1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[rt][ks]), __builtin_bit_cast(bf16x8_t, fB1[ks]), acc[0][1][rt], 0, 0, 0);
    stamp(kt, 7);
    if (!(VARIANT & 2)) __builtin_amdgcn_s_setprio(0);
    if ((VARIANT & 32) && more) issue(kt + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    BAR();
    // ---------------- phase 2: A1 x B1 ----------------
    stamp(kt, 8);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) if (!(VARIANT & 8)) fA[rt][ks] = lds_read128(buf + 1 * HALF + fa_off[rt] + (((2 * ks + gsel) ^ fa_sw[rt]) << 4));
    if (!(VARIANT & 32) && more) issue(kt + 1, 2);               // (nothing new is needed by phase 3)
    if (P8_TRACE == 2) stamp(kt, 9);
    BAR();
    WAIT_LGKM0();
    stamp(kt, 10);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VARIANT & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        if (VARIANT & 16) asm volatile("" :: "v"(fA[rt][ks]), "v"(fB1[ks])); else
        acc[1][1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[rt][ks]), __builtin_bit_cast(bf16x8_t, fB1[ks]), acc[1][1][rt], 0, 0, 0);
    stamp(kt, 11);
    if (!(VARIANT & 2)) __builtin_amdgcn_s_setprio(0);
    if ((VARIANT & 32) && more) issue(kt + 1, 2);
    __builtin_amdgcn_sched_barrier(0);
    BAR();
    // ---------------- phase 3: A1 x B0 ----------------
    stamp(kt, 12);
    if (VARIANT & 32) { if (more) WAIT_VM(2); }                        // (A0, B0, B1 of kt+1 outstanding -> A0, B0 have landed)
    else if (more) { issue(kt + 1, 3); WAIT_VM(4); }                        // A0(kt+1), B0(kt+1) have landed (B1, A1 of kt+1 may be in flight)
    if (P8_TRACE == 2) stamp(kt, 13);
    BAR();
    stamp(kt, 14);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VARIANT & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        if (VARIANT & 16) asm volatile("" :: "v"(fA[rt][ks]), "v"(fB0[ks])); else
        acc[1][0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[rt][ks]), __builtin_bit_cast(bf16x8_t, fB0[ks]), acc[1][0][rt], 0, 0, 0);
    stamp(kt, 15);
    if (!(VARIANT & 2)) __builtin_amdgcn_s_setprio(0);
    if ((VARIANT & 32) && more) issue(kt + 1, 3);
    __builtin_amdgcn_sched_barrier(0);
    BAR();
  }
  if (!(VARIANT & 1) && grp == 0) BAR();      // (balance the barrier count of the two groups)
  // plain epilogue (experiment): bf16 stores straight from the accumulators
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long row = m0 + 128 * a + 64 * wr + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int col = n0 + 128 * b + 32 * wc + (lane & 31);
          if (row < M && col < N) { const unsigned u = __float_as_uint(acc[a][b][t][e]); C[row * N + col] = (bf16raw)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
        }
}

static bf16raw f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (bf16raw)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(bf16raw h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int VARIANT>
static void run(int M, int N, int K, bool zeros = false) {
  std::vector<bf16raw> hA((size_t)M * K), hW((size_t)N * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = zeros ? 0 : f2bf(rnd());
  for (auto& v : hW) v = zeros ? 0 : f2bf(rnd());
  bf16raw *dA, *dW, *dC; unsigned long long* dT; hipMalloc(&dT, 2 * TR_NKT * TR_PTS * 8); hipMemset(dT, 0, 2 * TR_NKT * TR_PTS * 8);
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  const size_t lds = 128 * 1024;
  auto kern = gemm_p8_kernel<VARIANT>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((M + 255) / 256, (N + 255) / 256);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, dA, dW, dC, M, N, K, dT);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, dA, dW, dC, M, N, K, dT);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  std::vector<bf16raw> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int t = 0; t < 400; ++t) {
    const int m = (int)((t * 7919LL + 13) % M), n = (int)((t * 104729LL + 7) % N);
    double ref = 0; for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
    maxerr = fmax(maxerr, fabs(ref - bf2f(hC[(size_t)m * N + n]))); maxref = fmax(maxref, fabs(ref));
  }
  printf("variant %d  M %6d N %5d K %5d : %8.1f us  %7.1f TF   max err %.3g (ref max %.3g) %s\n", VARIANT, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, maxerr, maxref,
         hipGetLastError() == hipSuccess ? "" : "LAUNCH ERROR");
  if (P8_TRACE) {
    unsigned long long hT[2 * TR_NKT * TR_PTS]; hipMemcpy(hT, dT, sizeof(hT), hipMemcpyDeviceToHost);
    for (int g = 0; g < 2; ++g) for (int k = 0; k < TR_NKT; ++k) {
      printf("  group %d kt %2d:", g, TR_KT0 + k);
      for (int ph = 0; ph < 4; ++ph) { const unsigned long long* t = hT + (g * TR_NKT + k) * TR_PTS + 4 * ph;
        const unsigned long long nxt = ph < 3 ? t[4] : (k + 1 < TR_NKT ? hT[(g * TR_NKT + k + 1) * TR_PTS] : t[3]);
        if (P8_TRACE == 2) printf("  | issue+vm %4lld bar+lgkm %4lld mfma %4lld bar %4lld", (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]), (long long)(nxt - t[3]));
        else printf("  | rd+vm+bar+lgkm %4lld mfma %4lld bar %4lld", (long long)(t[2] - t[0]), (long long)(t[3] - t[2]), (long long)(nxt - t[3])); }
      printf("   (tile %lld)\n", (long long)(k + 1 < TR_NKT ? hT[(g * TR_NKT + k + 1) * TR_PTS] - hT[(g * TR_NKT + k) * TR_PTS] : 0));
    }
  }
  hipFree(dA); hipFree(dW); hipFree(dC); hipFree(dT);
}

int main() {
#if P8_TRACE
  run<0>(512, 512, 4096); run<0>(4096, 4096, 4096);
#else
  // the same 64 K-tiles per workgroup on 4 / 16 / 64 / 256 CUs: per-CU limits keep the time, memory-system limits and the power cap do not
  run<0>(512, 512, 4096); run<0>(1024, 1024, 4096); run<0>(2048, 2048, 4096); run<0>(4096, 2048, 4096);
  run<4>(512, 512, 4096); run<12>(512, 512, 4096); run<24>(512, 512, 4096); run<16>(512, 512, 4096); run<8>(512, 512, 4096); run<28>(512, 512, 4096);
  run<4>(2048, 2048, 4096); run<12>(2048, 2048, 4096); run<24>(2048, 2048, 4096); run<16>(2048, 2048, 4096); run<8>(2048, 2048, 4096); run<28>(2048, 2048, 4096);
  run<0>(4096, 4096, 4096); run<1>(4096, 4096, 4096); run<2>(4096, 4096, 4096); run<32>(4096, 4096, 4096); run<33>(4096, 4096, 4096); run<34>(4096, 4096, 4096);
  run<0>(4096, 4096, 4096, true); run<32>(4096, 4096, 4096, true);
  run<4>(4096, 4096, 4096); run<8>(4096, 4096, 4096); run<16>(4096, 4096, 4096); run<12>(4096, 4096, 4096); run<20>(4096, 4096, 4096); run<24>(4096, 4096, 4096); run<28>(4096, 4096, 4096);
  run<0>(8192, 8192, 8192); run<32>(8192, 8192, 8192);
  run<0>(115200, 256, 2304); run<0>(28800, 512, 4608); run<32>(115200, 256, 2304); run<32>(28800, 512, 4608);
#endif
  return 0;
}
