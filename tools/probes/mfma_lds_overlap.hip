// Probe (round 4): do MFMAs and LDS reads of one CU overlap?  DESIGN 20.5 found T ~ T_MFMA + T_LDS in every MFMA kernel of this repository.
// A workgroup of W waves loops over: R x ds_read_b128 (conflict-free, results unused) + M x v_mfma_f32_32x32x16_bf16 on M independent accumulators, with the
// accumulators in VGPRs or pinned to AGPRs ("a" constraint), reads waited for once per iteration (one iteration behind).  Prints cycles per iteration for
// MFMA only / reads only / both.   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_overlap.hip -o tools/_bin/mfma_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, bool AGPR, int R, int M>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = 0x3f803f80u;      // bf16 1.0 pairs (64 KB)
  __syncthreads();
  typedef __attribute__((address_space(3))) void* lptr_t;
  unsigned base = (unsigned)(unsigned long long)(lptr_t)smem + lane * 16 + (tid >> 6) * 4096;
  f32x16 acc[M];
#pragma unroll
  for (int j = 0; j < M; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  u32x4 f[2][R];
#pragma unroll
  for (int k = 0; k < R; ++k) { f[0][k] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; f[1][k] = f[0][k]; }
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll 1
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (MODE != 0) {
#pragma unroll
        for (int k = 0; k < R; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[h][k]) : "v"(base), "n"(k * 1024 % 4096) : "memory");
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(R) : "memory");       // the previous half's reads have landed
#pragma unroll
        for (int k = 0; k < R; ++k) asm volatile("" : "+v"(f[h ^ 1][k]));
      }
      if (MODE != 1) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(f[h ^ 1][j % R]));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(f[h ^ 1][j % R]));
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < M; ++j) s += acc[j][0] + acc[j][7];
#pragma unroll
  for (int k = 0; k < R; ++k) s += __uint_as_float(f[0][k].x) + __uint_as_float(f[1][k].y);
  if (s == 1234.5f) out[0] = s;
}

// the iteration of wgrad_pairs.hip: three tap-row groups of 3 MFMAs; transposed reads (two per fragment) re-issued behind each group, counted waits, masks on two of the
// three x fragments.  MODE 0: MFMAs only, 1: reads (+ masks) only, 2: both
template <int MODE, int VAR>      // VAR bits: 1 no masks, 2 no ring rotation, 4 one group of nine MFMAs behind one wait (x fragments double-buffered)
__global__ __launch_bounds__(512) void probe_pairs(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = 0x3f803f80u;
  __syncthreads();
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned base = (unsigned)(unsigned long long)(lptr_t)smem + (lane & 15) * 8 + (lane >> 4) * 128 + (tid >> 6) * 4096;
  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  u32x2 A[4][2], B[3][2], B2[3][2];
#pragma unroll
  for (int k = 0; k < 4; ++k) A[k][0] = A[k][1] = u32x2{0x3f803f80u, 0x3f803f80u};
#pragma unroll
  for (int k = 0; k < 3; ++k) B[k][0] = B[k][1] = B2[k][0] = B2[k][1] = u32x2{0x3f803f80u, 0x3f803f80u};
  const unsigned msk = tid == 9999 ? 0u : 0xffffffffu;
  auto rd = [&](u32x2& d, int off) { if (MODE != 0) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(base), "n"(0) : "memory"); (void)off; };
  auto mm = [&](f32x16& c, const u32x2 (&a)[2], const u32x2 (&b)[2]) {
    if (MODE == 1) return;
    const u32x4 va = {a[0].x, a[0].y, a[1].x, a[1].y}, vb = {b[0].x, b[0].y, b[1].x, b[1].y};
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(va), "v"(vb));
  };
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (VAR & 4) {
      // reads of the NEXT iteration first (into the other buffer), then one wait for this iteration's eight, then nine MFMAs
      rd(B2[0][0], 0); rd(B2[0][1], 0); rd(B2[1][0], 0); rd(B2[1][1], 0); rd(B2[2][0], 0); rd(B2[2][1], 0); rd(A[2][0], 0); rd(A[2][1], 0);
      if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      asm volatile("" : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[2][0]), "+v"(B[2][1]), "+v"(A[1][0]), "+v"(A[1][1]));
      if (!(VAR & 1)) { B[0][0].x &= msk; B[0][0].y &= msk; B[0][1].x &= msk; B[0][1].y &= msk; B[2][0].x &= msk; B[2][0].y &= msk; B[2][1].x &= msk; B[2][1].y &= msk; }
      mm(acc[0], A[1], B[0]); mm(acc[1], A[0], B[0]); mm(acc[2], A[3], B[0]);
      mm(acc[3], A[1], B[1]); mm(acc[4], A[0], B[1]); mm(acc[5], A[3], B[1]);
      mm(acc[6], A[1], B[2]); mm(acc[7], A[0], B[2]); mm(acc[8], A[3], B[2]);
#pragma unroll
      for (int k = 0; k < 3; ++k) { const u32x2 t0 = B[k][0], t1 = B[k][1]; B[k][0] = B2[k][0]; B[k][1] = B2[k][1]; B2[k][0] = t0; B2[k][1] = t1; }
    } else {
    if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    asm volatile("" : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(A[1][0]), "+v"(A[1][1]));
    if (!(VAR & 1)) { B[0][0].x &= msk; B[0][0].y &= msk; B[0][1].x &= msk; B[0][1].y &= msk; }
    mm(acc[0], A[1], B[0]); mm(acc[1], A[0], B[0]); mm(acc[2], A[3], B[0]);
    rd(B[0][0], 0); rd(B[0][1], 0); rd(A[2][0], 0); rd(A[2][1], 0);
    if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    asm volatile("" : "+v"(B[1][0]), "+v"(B[1][1]));
    mm(acc[3], A[1], B[1]); mm(acc[4], A[0], B[1]); mm(acc[5], A[3], B[1]);
    rd(B[1][0], 0); rd(B[1][1], 0);
    if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    asm volatile("" : "+v"(B[2][0]), "+v"(B[2][1]));
    if (!(VAR & 1)) { B[2][0].x &= msk; B[2][0].y &= msk; B[2][1].x &= msk; B[2][1].y &= msk; }
    mm(acc[6], A[1], B[2]); mm(acc[7], A[0], B[2]); mm(acc[8], A[3], B[2]);
    rd(B[2][0], 0); rd(B[2][1], 0);
    }
    if (!(VAR & 2)) { const u32x2 t0 = A[0][0], t1 = A[0][1]; A[0][0] = A[1][0]; A[0][1] = A[1][1]; A[1][0] = A[2][0]; A[1][1] = A[2][1]; A[3][0] = t0; A[3][1] = t1; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 9; ++j) s += acc[j][0] + acc[j][7];
  s += __uint_as_float(A[0][0].x) + __uint_as_float(A[1][1].y) + __uint_as_float(A[2][0].x) + __uint_as_float(B[0][0].x) + __uint_as_float(B[1][0].x) + __uint_as_float(B[2][1].y) + __uint_as_float(B2[0][0].x) + __uint_as_float(B2[1][1].y) + __uint_as_float(B2[2][0].x);
  if (s == 1234.5f) out[0] = s;
}
template <int MODE, int VAR> static double run_pairs(int waves, int iters) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto k = probe_pairs<MODE, VAR>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, out, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms * 1e-3 * 2.4e9 / iters;
}

template <int MODE, bool AGPR, int R, int M>
static double run(int waves, int iters) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto k = probe<MODE, AGPR, R, M>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, out, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms * 1e-3 * 2.4e9 / iters;      // cycles per iteration at 2.4 GHz
}

template <bool AGPR, int R, int M> static void row(int waves) {
  const int iters = 20000;
  const double m = run<0, AGPR, R, M>(waves, iters), l = run<1, AGPR, R, M>(waves, iters), b = run<2, AGPR, R, M>(waves, iters);
  printf("%d waves/WG  acc in %s  %d reads + %d MFMAs per iteration: MFMA only %7.1f  reads only %7.1f  both %7.1f cycles   (sum %7.1f, max %7.1f)\n",
         waves, AGPR ? "AGPR" : "VGPR", R, M, m, l, b, m + l, m > l ? m : l);
}

int main() {
#define PAIRS_ROW(VAR, WHAT) for (int waves : {8, 4}) { \
    const double m = run_pairs<0, VAR>(waves, 20000), l = run_pairs<1, VAR>(waves, 20000), b = run_pairs<2, VAR>(waves, 20000); \
    printf("%d waves/WG  pairs-kernel iteration (9 MFMAs, 8 transposed reads) %-44s: MFMA only %7.1f  reads only %7.1f  both %7.1f cycles\n", waves, WHAT, m, l, b); }
  PAIRS_ROW(0, "as in the kernel (+ ring rotation moves)");
  PAIRS_ROW(2, "no ring rotation moves");
  PAIRS_ROW(3, "no masks, no rotation");
  PAIRS_ROW(6, "one group of nine, masks");
  PAIRS_ROW(7, "one group of nine, no masks");
  row<false, 8, 9>(8); row<true, 8, 9>(8);
  row<false, 8, 9>(4); row<true, 8, 9>(4);
  row<false, 4, 9>(8); row<true, 4, 9>(8);
  row<false, 8, 8>(8); row<true, 8, 8>(8);
  return 0;
}
