// Experiment: bf16 NT GEMM C[M][N] = A[M][K] * W[N][K]^T with a 256x256 tile, 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA 32x32x16 tiles), one workgroup per CU.
// LDS-DMA ring of 64-byte rows (BK = 32), source-side swizzle, counted vmcnt, one raw barrier per K-tile; all 12 fragment reads of a K-tile are issued at
// once (inline asm) and the MFMA groups wait with counted lgkmcnt.  VARIANT 1: the second wave group runs half a K-tile behind the first (split barrier).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_gemm8 tools/ubench_gemm8.hip
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef unsigned short bf16raw;

__device__ __attribute__((aligned(64))) unsigned char zero16[64];
__device__ __forceinline__ u32x4 lds_read128(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }

template <int STAGES>
__global__ __launch_bounds__(512, 1) void gemm8_kernel(const bf16raw* __restrict__ A, const bf16raw* __restrict__ W, bf16raw* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 256, BN = 256, RB = 64, KE = 32;
  constexpr int TILE = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  // XCD-aware remap: consecutive tiles of one N-column stay on one XCD
  const int nbx = gridDim.x, nby = gridDim.y;
  int bid = blockIdx.y * nbx + blockIdx.x;
  const int nwg = nbx * nby, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  const int by = bid / nbx, bx = bid % nbx;
  const long long m0 = (long long)bx * BM; const int n0 = by * BN;
  long long aoff[2], boff[2]; int kc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 2) + i * 128;
    aoff[i] = (m0 + row < M) ? (m0 + row) * (long long)K : -1;
    boff[i] = (n0 + row < N) ? (long long)(n0 + row) * K : -1;
    kc[i] = ((tid & 3) ^ swz(row)) * 8;
  }
  auto issue = [&](int kt, int buf) {
    char* As = smem + buf * TILE; char* Bs = As + BM * RB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const void* src = aoff[i] >= 0 ? (const void*)(A + aoff[i] + kt * KE + kc[i]) : (const void*)zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (i * 512 + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const void* src = boff[i] >= 0 ? (const void*)(W + boff[i] + kt * KE + kc[i]) : (const void*)zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (i * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  unsigned fa_off[4], fb_off[2]; int fa_sw[4], fb_sw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int row = wr * 128 + i * 32 + (lane & 31); fa_off[i] = row * RB; fa_sw[i] = swz(row); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int row = wc * 64 + j * 32 + (lane & 31); fb_off[j] = BM * RB + row * RB; fb_sw[j] = swz(row); }
  const int gsel = lane >> 5;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int KT = K / KE;
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st) if (st < KT) issue(st, st);
  for (int kt = 0; kt < KT; ++kt) {
    const int newer = min(STAGES - 2, KT - 1 - kt);
    if (newer <= 0) WAIT_VM(0); else if (newer == 1) WAIT_VM(4); else if (newer == 2) WAIT_VM(8); else WAIT_VM(12);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned base = lds0 + (kt % STAGES) * TILE;
    u32x4 fb[2][2], fa[2][4];
#pragma unroll
    for (int qk = 0; qk < 2; ++qk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[qk][j] = lds_read128(base + fb_off[j] + (((qk * 2 + gsel) ^ fb_sw[j]) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[qk][i] = lds_read128(base + fa_off[i] + (((qk * 2 + gsel) ^ fa_sw[i]) << 4));
    }
    if (kt + STAGES - 1 < KT) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int qk = 0; qk < 2; ++qk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // reads issued: per qk 2 B + 4 A = 6; group (qk, i) needs everything up to A_i of qk: outstanding allowed = 12 - (qk * 6 + 2 + i + 1)
        constexpr int dummy = 0; (void)dummy;
        const int allowed = 12 - (qk * 6 + 3 + i);
        if (allowed == 9) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); else if (allowed == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else if (allowed == 7) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); else if (allowed == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else if (allowed == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); else if (allowed == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (allowed == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(fa[qk][i]), "+v"(fb[qk][0]), "+v"(fb[qk][1]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[qk][i]), __builtin_bit_cast(bf16x8_t, fb[qk][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
#undef WAIT_VM
  // plain epilogue (experiment): bf16 stores straight from the accumulators
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = m0 + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = n0 + wc * 64 + j * 32 + (lane & 31);
        if (row < M && col < N) { const unsigned u = __float_as_uint(acc[i][j][e]); C[row * N + col] = (bf16raw)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
      }
}

template <int STAGES, int ABLV>
__global__ __launch_bounds__(512, 1) void gemm8pp_kernel(const bf16raw* __restrict__ A, const bf16raw* __restrict__ W, bf16raw* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 256, BN = 256, RB = 64, KE = 32;
  constexpr int TILE = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  // XCD-aware remap: consecutive tiles of one N-column stay on one XCD
  const int nbx = gridDim.x, nby = gridDim.y;
  int bid = blockIdx.y * nbx + blockIdx.x;
  const int nwg = nbx * nby, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  const int by = bid / nbx, bx = bid % nbx;
  const long long m0 = (long long)bx * BM; const int n0 = by * BN;
  long long aoff[2], boff[2]; int kc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 2) + i * 128;
    aoff[i] = (m0 + row < M) ? (m0 + row) * (long long)K : -1;
    boff[i] = (n0 + row < N) ? (long long)(n0 + row) * K : -1;
    kc[i] = ((tid & 3) ^ swz(row)) * 8;
  }
  auto issue = [&](int kt, int buf) {
    char* As = smem + buf * TILE; char* Bs = As + BM * RB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const void* src = aoff[i] >= 0 ? (const void*)(A + aoff[i] + kt * KE + kc[i]) : (const void*)zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (i * 512 + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const void* src = boff[i] >= 0 ? (const void*)(W + boff[i] + kt * KE + kc[i]) : (const void*)zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (i * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto issue1 = [&](int kt, int buf, int p) {
    char* As = smem + buf * TILE; char* Bs = As + BM * RB;
    const int ii = p & 1;
    if (p < 2) { const void* src = aoff[ii] >= 0 ? (const void*)(A + aoff[ii] + kt * KE + kc[ii]) : (const void*)zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (ii * 512 + wave * 64) * 16), 16, 0, 0); }
    else { const void* src = boff[ii] >= 0 ? (const void*)(W + boff[ii] + kt * KE + kc[ii]) : (const void*)zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (ii * 512 + wave * 64) * 16), 16, 0, 0); }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  unsigned fa_off[4], fb_off[2]; int fa_sw[4], fb_sw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int row = wr * 128 + i * 32 + (lane & 31); fa_off[i] = row * RB; fa_sw[i] = swz(row); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int row = wc * 64 + j * 32 + (lane & 31); fb_off[j] = BM * RB + row * RB; fb_sw[j] = swz(row); }
  const int gsel = lane >> 5;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int KT = K / KE;
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st) if (st < KT) issue(st, st);
  // tile 0 must have landed before the first read: every wave waits for its share, then one barrier
  { const int newer = min(STAGES - 2, KT - 1); if (newer <= 0) WAIT_VM(0); else if (newer == 1) WAIT_VM(4); else if (newer == 2) WAIT_VM(8); else WAIT_VM(12); }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();          // the second wave group runs one phase behind: its load phase meets the first group's MFMA phase
  for (int kt = 0; kt < KT; ++kt) {
    // ---- load phase: fragments of tile kt, DMA of tile kt + STAGES - 1 ----
    const unsigned base = lds0 + (kt % STAGES) * TILE;
    u32x4 fb[2][2], fa[2][4];
#pragma unroll
    for (int qk = 0; qk < 2; ++qk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { if (ABLV & 2) { fb[qk][j] = (u32x4){(unsigned)kt, 1u, 2u, 3u}; } else fb[qk][j] = lds_read128(base + fb_off[j] + (((qk * 2 + gsel) ^ fb_sw[j]) << 4)); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { if (ABLV & 2) { fa[qk][i] = (u32x4){(unsigned)kt, 1u, 2u, (unsigned)i}; } else fa[qk][i] = lds_read128(base + fa_off[i] + (((qk * 2 + gsel) ^ fa_sw[i]) << 4)); }
    }
    if (!(ABLV & 4) && !(ABLV & 16) && kt + STAGES - 1 < KT) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // tile kt + 1 complete (this wave's share) before the barrier that precedes anybody's read of it
    { const int newer = min(STAGES - 2, KT - 2 - kt); if (newer <= 0) WAIT_VM(0); else if (newer == 1) WAIT_VM(4); else if (newer == 2) WAIT_VM(8); else WAIT_VM(12); }
    __builtin_amdgcn_s_barrier();
    // ---- MFMA phase ----
#pragma unroll
    for (int qk = 0; qk < 2; ++qk)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fa[qk][i]), "+v"(fb[qk][0]), "+v"(fb[qk][1]));
    if (!(ABLV & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int qk = 0; qk < 2; ++qk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (ABLV & 1) asm volatile("" :: "v"(fa[qk][i]), "v"(fb[qk][j])); else
          { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[qk][i]), __builtin_bit_cast(bf16x8_t, fb[qk][j]), acc[i][j], 0, 0, 0);
            if ((ABLV & 16) && (i & 1) && j == 1 && kt + STAGES - 1 < KT) { issue1(kt + STAGES - 1, (kt + STAGES - 1) % STAGES, qk * 2 + (i >> 1)); __builtin_amdgcn_sched_barrier(0); } }
    if (!(ABLV & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();          // (balance the barrier count of the two groups)
#undef WAIT_VM
  // plain epilogue (experiment): bf16 stores straight from the accumulators
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = m0 + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = n0 + wc * 64 + j * 32 + (lane & 31);
        if (row < M && col < N) { const unsigned u = __float_as_uint(acc[i][j][e]); C[row * N + col] = (bf16raw)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
      }
}

static bf16raw f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (bf16raw)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(bf16raw h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int STAGES, int PP, int ABLV = 0>
static void run(int M, int N, int K) {
  std::vector<bf16raw> hA((size_t)M * K), hW((size_t)N * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = f2bf(rnd());
  for (auto& v : hW) v = f2bf(rnd());
  bf16raw *dA, *dW, *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  const size_t lds = (size_t)STAGES * 512 * 64;
  auto kern = PP ? gemm8pp_kernel<STAGES, ABLV> : gemm8_kernel<STAGES>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((M + 255) / 256, (N + 255) / 256);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, dA, dW, dC, M, N, K);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, dA, dW, dC, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  std::vector<bf16raw> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int t = 0; t < 200; ++t) {
    const int m = (int)((t * 7919LL + 13) % M), n = (int)((t * 104729LL + 7) % N);
    double ref = 0; for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
    maxerr = fmax(maxerr, fabs(ref - bf2f(hC[(size_t)m * N + n]))); maxref = fmax(maxref, fabs(ref));
  }
  printf("pp %d abl %d stages %d  M %6d N %5d K %5d : %8.1f us  %7.1f TF   max err %.3g (ref max %.3g) %s\n", PP, ABLV, STAGES, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, maxerr, maxref,
         hipGetLastError() == hipSuccess ? "" : "LAUNCH ERROR");
  hipFree(dA); hipFree(dW); hipFree(dC);
}

int main() {
  run<4, 1, 0>(4096, 4096, 4096); run<4, 1, 8>(4096, 4096, 4096); run<4, 1, 16>(4096, 4096, 4096); run<4, 1, 24>(4096, 4096, 4096); run<3, 1, 16>(4096, 4096, 4096);
  return 0;
}
