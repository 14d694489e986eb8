// Per-CU load-path micro-benchmark (gfx950): how fast can one workgroup stream an L2-resident buffer into LDS / registers?
//   mode 0: global_load_lds (LDS-DMA, 16 B per lane), DEPTH 16 KB slots in flight
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging), DEPTH slots in flight
//   mode 2: global_load_dwordx4 -> VGPR only (consumed by an xor), DEPTH slots in flight
// Each workgroup (256 threads) walks `bytes` of the buffer `iters` times in 16 KB slots.  Prints GB/s per workgroup and in aggregate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_load tools/ubench_load.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ buf, long long bytes, int iters, unsigned* sink, int same_region) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const long long nslots = bytes / 16384;
  const long long start = same_region ? 0 : (blockIdx.x * 7) % nslots;       // different workgroups start at different slots
  uint4 acc = {0, 0, 0, 0};
  long long total = (long long)iters * nslots;
  if (MODE == 0) {
    for (long long s = 0; s < total + DEPTH - 1; ++s) {
      if (s < total) {
        const char* src = buf + ((start + s) % nslots) * 16384 + tid * 16;
        char* dst = smem + (s % DEPTH) * 16384 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 4096), (lptr_t)(dst + i * 4096), 16, 0, 0);
      }
      if (s >= DEPTH - 1) {
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc.x = *(const unsigned*)(smem + tid * 4);
  } else {
    uint4 r[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) r[d][i] = *(const uint4*)(buf + ((start + d) % nslots) * 16384 + tid * 16 + i * 4096);
    for (long long s0 = 0; s0 < total; s0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const long long s = s0 + d;
        const long long sn = s + DEPTH - 1;
        constexpr int dn_base = DEPTH - 1;
        const int dn = (d + dn_base) % DEPTH;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[dn][i] = *(const uint4*)(buf + ((start + sn) % nslots) * 16384 + tid * 16 + i * 4096);
        if (MODE == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) *(uint4*)(smem + (d % 2) * 16384 + (i * 256 + tid) * 16) = r[d][i];
          __builtin_amdgcn_s_barrier();
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc.x ^= r[d][i].x; acc.y ^= r[d][i].y; acc.z ^= r[d][i].z; acc.w ^= r[d][i].w; }
        }
      }
    }
    if (MODE == 1) { __syncthreads(); acc.x ^= *(const unsigned*)(smem + tid * 4); }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// mode 3: LDS-DMA of a [rows][RB bytes] tile taken from a row-major matrix with row stride ST bytes (how a GEMM operand tile is fetched); 16 KB per slot
template <int RB, int DEPTH>
__global__ __launch_bounds__(256) void tile_kernel(const char* __restrict__ buf, long long st, int iters, int nslots, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  constexpr int CPR = RB / 16, RPP = 256 / CPR, ROWS = 16384 / RB;
  const int row0 = tid / CPR, cb = (tid % CPR) * 16;
  const long long total = (long long)iters * nslots;
  for (long long s = 0; s < total + DEPTH - 1; ++s) {
    if (s < total) {
      // slot q of the matrix: rows 0..ROWS-1, byte columns q*RB .. (q+1)*RB  (successive K-steps of one row block)
      const long long q = (s + blockIdx.x) % nslots;
      const char* src = buf + q * RB + cb;
      char* dst = smem + (s % DEPTH) * 16384 + wave * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + (long long)(row0 + i * RPP) * st), (lptr_t)(dst + i * 4096), 16, 0, 0);
    }
    if (s >= DEPTH - 1) {
      if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (*(const unsigned*)(smem + tid * 4) == 0x12345678u) sink[0] = 1;
}
template <int RB, int DEPTH>
static void run_tile(const char* buf, long long st, int wgs, unsigned* sink) {
  const int iters = 64; const int nslots = (int)(st / RB);
  size_t lds = (size_t)DEPTH * 16384;
  hipFuncSetAttribute((const void*)tile_kernel<RB, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((tile_kernel<RB, DEPTH>), dim3(wgs), dim3(256), lds, 0, buf, st, 2, nslots, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((tile_kernel<RB, DEPTH>), dim3(wgs), dim3(256), lds, 0, buf, st, iters, nslots, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double gb = 16384.0 * iters * nslots * wgs / 1e9;
  printf("tile rows of %3d B, row stride %5lld B, depth %d, wgs %4d: %8.1f us  %7.1f GB/s per WG  %8.1f GB/s total\n", RB, st, DEPTH, wgs, ms * 1e3, gb / wgs / (ms * 1e-3), gb / (ms * 1e-3));
}

template <int MODE, int DEPTH>
static void run(const char* buf, long long bytes, int wgs, unsigned* sink, int same) {
  const int iters = 64;
  size_t lds = MODE == 0 ? (size_t)DEPTH * 16384 : 32768;
  hipFuncSetAttribute((const void*)stream_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(wgs), dim3(256), lds, 0, buf, bytes, 2, sink, same);
  hipEventRecord(e0);
  hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(wgs), dim3(256), lds, 0, buf, bytes, iters, sink, same);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double gb = (double)bytes * iters * wgs / 1e9;
  printf("mode %d depth %d buf %5lld KB wgs %4d same %d: %8.1f us  %7.1f GB/s per WG  %8.1f GB/s total\n", MODE, DEPTH, bytes / 1024, wgs, same, ms * 1e3, gb / wgs / (ms * 1e-3), gb / (ms * 1e-3));
}

int main() {
  char* buf; unsigned* sink;
  hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20); hipMalloc(&sink, 4);
  for (int wgs : {50, 256}) {
    for (long long st : {512ll, 720ll, 2048ll, 2880ll, 4096ll, 8192ll}) { run_tile<128, 4>(buf, st, wgs, sink); if (st >= 256) run_tile<256, 4>(buf, st, wgs, sink); }
  }
  for (long long bytes : {1ll << 20}) {
    for (int wgs : {50, 256}) {
      for (int same : {1, 0}) {
        run<0, 2>(buf, bytes, wgs, sink, same);
        run<0, 4>(buf, bytes, wgs, sink, same);
        run<0, 8>(buf, bytes, wgs, sink, same);
        run<1, 2>(buf, bytes, wgs, sink, same);
        run<1, 4>(buf, bytes, wgs, sink, same);
        run<2, 2>(buf, bytes, wgs, sink, same);
        run<2, 4>(buf, bytes, wgs, sink, same);
      }
    }
  }
  return 0;
}
