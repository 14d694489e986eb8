// Probe (round 4): what does ONE dependent kernel node of a captured hipGraph cost on this part, whatever the kernel does?
// The conformer part of the step is a chain of ~25 dependent launches per block and pass; tools/bench_small_gemm.py measures 5-12 us per product inside a graph.
// This probe replays chains of N dependent nodes of
//   null    : <<<G, 256, lds>>>, one thread increments a counter (the floor: dispatch + end-of-kernel fence + dependency)
//   touch   : every workgroup reads 16 KB and writes 8 KB of bf16 (the traffic of a 64 x 64 x 256 tile), no arithmetic
//   ldsdma  : every workgroup pulls 48 KB into LDS by global_load_lds, waits, reads it once, writes 8 KB (the first three ring stages of gemm_nt_plain_kernel)
// for several grid sizes and LDS footprints (= workgroups per CU), as a graph and as plain stream launches.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/graph_chain.hip -o tools/_bin/graph_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_null(unsigned* ctr) {
  extern __shared__ char smem[];
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[0] += 1u;
}
__global__ __launch_bounds__(256) void k_touch(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned nsrc16, unsigned* ctr) {
  extern __shared__ char smem[];
  // 16 KB = 1024 x 16 B per workgroup: 4 per thread
  const unsigned b = blockIdx.x * 1024u;
  uint4 a = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) { const uint4 t = src[(b + i * 256 + threadIdx.x) % nsrc16]; a.x ^= t.x; a.y ^= t.y; a.z ^= t.z; a.w ^= t.w; }
  dst[blockIdx.x * 512u + threadIdx.x] = a; dst[blockIdx.x * 512u + 256 + threadIdx.x] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[0] += 1u;
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__global__ __launch_bounds__(256) void k_ldsdma(const char* __restrict__ src, uint4* __restrict__ dst, unsigned nsrc, unsigned* ctr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned b = (blockIdx.x * 49152u) % nsrc;
#pragma unroll
  for (int i = 0; i < 12; ++i) glds16(src + b + (i * 4 + wave) * 1024u + lane * 16u, lds0 + (i * 4 + wave) * 1024u);     // 12 x 4 KB = 48 KB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  uint4 a = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 12; ++i) { const uint4 t = *(const uint4*)(smem + i * 4096 + threadIdx.x * 16); a.x ^= t.x; a.y ^= t.y; a.z ^= t.z; a.w ^= t.w; }
  dst[blockIdx.x * 512u + threadIdx.x] = a; dst[blockIdx.x * 512u + 256 + threadIdx.x] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[0] += 1u;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 200, REP = 20;
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned* ctr; CK(hipMalloc(&ctr, 4)); CK(hipMemset(ctr, 0, 4));
  const size_t SRC = 16u << 20; char* src; CK(hipMalloc(&src, SRC + 65536)); CK(hipMemset(src, 1, SRC + 65536));
  uint4* dst; CK(hipMalloc(&dst, (size_t)4096 * 8192));
  CK(hipFuncSetAttribute((const void*)k_null, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_ldsdma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { int kind, G, lds; };
  std::vector<Cfg> cfgs;
  for (int G : {1, 256, 512, 800, 1600}) for (int lds : {0, 65536}) cfgs.push_back({0, G, lds});
  for (int G : {200, 512, 800, 1600}) for (int lds : {0, 65536}) cfgs.push_back({1, G, lds});
  for (int G : {200, 512, 800, 1600}) for (int lds : {49152, 65536}) cfgs.push_back({2, G, lds});
  const char* names[3] = {"null", "touch", "ldsdma"};
  printf("%-8s %6s %7s | graph us/node | stream us/launch\n", "kernel", "grid", "lds");
  for (const Cfg& c : cfgs) {
    auto launch = [&]() {
      if (c.kind == 0) hipLaunchKernelGGL(k_null, dim3(c.G), dim3(256), c.lds, st, ctr);
      else if (c.kind == 1) hipLaunchKernelGGL(k_touch, dim3(c.G), dim3(256), c.lds, st, (const uint4*)src, dst, (unsigned)(SRC / 16), ctr);
      else hipLaunchKernelGGL(k_ldsdma, dim3(c.G), dim3(256), c.lds, st, (const char*)src, dst, (unsigned)SRC, ctr);
    };
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < REP; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float msg; CK(hipEventElapsedTime(&msg, e0, e1));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) for (int i = 0; i < N; ++i) launch();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float mss; CK(hipEventElapsedTime(&mss, e0, e1));
    printf("%-8s %6d %7d | %8.2f      | %8.2f\n", names[c.kind], c.G, c.lds, msg * 1e3 / (REP * N), mss * 1e3 / (5 * N));
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
  }
  return 0;
}
