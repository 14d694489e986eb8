// How fast can ONE workgroup per CU stream a private slice of a large (HBM-resident) buffer into LDS?  (gfx950; profiles/r05_hbm_stream_ubench.txt)
//   mode 0: global_load_lds_dwordx4 (LDS-DMA) into a two-unit LDS ring: the unit after next is requested behind the barrier that frees its buffer, counted vmcnt
//   mode 1: global_load_dwordx4 -> VGPR (two register sets, one unit ahead) -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR only (xor-consumed)
// Workgroups of NT threads, unit = UNIT bytes, every workgroup walks its own contiguous slice once.   usage: ubench_hbm_stream [GB]
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_hbm_stream tools/ubench_hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

__device__ __attribute__((aligned(64))) unsigned char zero16[64];
template <int MODE, int NT, int UNIT, int PAT = 0>
__global__ __launch_bounds__(NT) void stream_kernel(const char* __restrict__ buf, long long slice, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PER = UNIT / (NT * 16);                       // 16-byte pieces per thread and unit
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const char* base = buf + (long long)blockIdx.x * slice;
  const int units = (int)(slice / UNIT);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  uint4 acc = {0, 0, 0, 0};
  if (MODE == 0) {
    auto issue = [&](int u) {
      const long long uoff = PAT == 4 ? ((long long)u * gridDim.x + blockIdx.x) * UNIT - (long long)blockIdx.x * slice : (long long)u * UNIT;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int piece = wave + (NT / 64) * i;
        const int row = piece * 8 + (lane >> 3); int c = lane & 7;
        if (PAT == 1 || PAT == 2) c ^= 4 * ((row >> 1) & 1);
        const char* src = base + uoff + (long long)row * 128 + c * 16;
        if (PAT == 3 && (i & 1)) src += slice * (gridDim.x / 2) ;           // second stream: the other half of the buffer (workgroups use the first half only)
        if (PAT == 2 && row % 23 == 22) src = (const char*)zero16;
        glds16(src, lds0 + (u & 1) * UNIT + piece * 1024);
      }
    };
    issue(0);
    for (int u = 0; u < units; ++u) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (u + 1 < units) issue(u + 1);
      acc.x ^= *(const unsigned*)(smem + (u & 1) * UNIT + tid * 4);      // (touch the unit)
    }
  } else {
    uint4 ra[PER], rb[PER];
    auto load = [&](uint4 (&r)[PER], int u) {
      const char* src = base + (long long)u * UNIT + tid * 16;
#pragma unroll
      for (int i = 0; i < PER; ++i) r[i] = *(const uint4*)(src + (long long)i * NT * 16);
    };
    auto use = [&](uint4 (&r)[PER], int u) {
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < PER; ++i) *(uint4*)(smem + (u & 1) * UNIT + tid * 16 + i * NT * 16) = r[i];
        __syncthreads();
        acc.x ^= *(const unsigned*)(smem + (u & 1) * UNIT + ((tid * 4 + 64) % UNIT));
      } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) { acc.x ^= r[i].x; acc.y ^= r[i].y; acc.z ^= r[i].z; acc.w ^= r[i].w; }
      }
    };
    load(ra, 0);
    for (int u = 0; u < units; u += 2) {
      if (u + 1 < units) load(rb, u + 1);
      use(ra, u);
      if (u + 2 < units) load(ra, u + 2);
      if (u + 1 < units) use(rb, u + 1);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <int MODE, int NT, int UNIT, int PAT = 0>
void run(const char* name, const char* buf, long long bytes, unsigned* sink, int wgs) {
  const long long slice = (PAT == 3 ? bytes / 2 : bytes) / wgs / UNIT * UNIT;
  const size_t lds = MODE == 2 ? 0 : 2 * UNIT;
  CK(hipFuncSetAttribute((const void*)stream_kernel<MODE, NT, UNIT, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<MODE, NT, UNIT, PAT>), dim3(wgs), dim3(NT), lds, 0, buf, slice, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1) printf("%-58s %4d wg x %4d thr, unit %3d KB: %8.1f us  %6.2f TB/s  (%5.1f GB/s per workgroup)\n", name, wgs, NT, UNIT / 1024, ms * 1e3, slice * wgs / ms / 1e9, slice / ms / 1e6);
  }
}

int main(int argc, char** argv) {
  const long long bytes = (long long)((argc > 1 ? atof(argv[1]) : 2.0) * (1 << 30));
  char* buf; unsigned* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 1, bytes));
  run<0, 512, 65536>("LDS-DMA, 8 waves, 2 x 64 KB ring", buf, bytes, sink, 256);
  run<0, 1024, 65536>("LDS-DMA, 16 waves, 2 x 64 KB ring", buf, bytes, sink, 256);
  run<0, 256, 32768>("LDS-DMA, 4 waves, 2 x 32 KB ring, 2 wg / CU", buf, bytes, sink, 512);
  run<0, 256, 16384>("LDS-DMA, 4 waves, 2 x 16 KB ring, 4 wg / CU", buf, bytes, sink, 1024);
  run<0, 512, 65536, 1>("LDS-DMA, 8 waves: chunks of odd row pairs XOR-swizzled", buf, bytes, sink, 256);
  run<0, 512, 65536, 2>("LDS-DMA, 8 waves: swizzled + every 23rd row from a zero chunk", buf, bytes, sink, 256);
  run<0, 512, 65536, 3>("LDS-DMA, 8 waves: two streams, alternating pieces", buf, bytes, sink, 256);
  run<0, 512, 65536, 4>("LDS-DMA, 8 waves: units strided over the workgroups", buf, bytes, sink, 256);
  run<1, 512, 65536>("loads -> VGPR -> ds_write, 8 waves, 64 KB units", buf, bytes, sink, 256);
  run<1, 1024, 65536>("loads -> VGPR -> ds_write, 16 waves, 64 KB units", buf, bytes, sink, 256);
  run<1, 256, 32768>("loads -> VGPR -> ds_write, 4 waves, 32 KB units, 2 wg / CU", buf, bytes, sink, 512);
  run<2, 512, 65536>("loads -> VGPR only, 8 waves, 64 KB units", buf, bytes, sink, 256);
  run<2, 1024, 65536>("loads -> VGPR only, 16 waves", buf, bytes, sink, 256);
  run<2, 256, 16384>("loads -> VGPR only, 4 waves, 16 KB units, 8 wg / CU", buf, bytes, sink, 2048);
  run<2, 256, 16384>("loads -> VGPR only, 4 waves, 16 KB units, 16 wg / CU", buf, bytes, sink, 4096);
  return 0;
}
