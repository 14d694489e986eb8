// Does ds_read_b128 serve 4-byte-aligned LDS addresses on gfx950 (SH_MEM_CONFIG alignment mode), and at what cost?  hipcc --offload-arch=gfx950 -O3 lds_unaligned_probe.hip -o lds_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void probe(unsigned* out, int shift, int iters, int mode) {
  __shared__ unsigned lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  unsigned a = base + (threadIdx.x * 16 + shift * 4) % 16384;      // 16-byte stride per lane, shifted by `shift` dwords
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 v;
    if (mode == 0) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    else { unsigned x, y, z, w; asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(w) : "v"(a) : "memory"); v = u32x4{x, y, z, w}; }
    acc += v; a = base + ((a - base) + 64) % 16000;
  }
  for (int e = 0; e < 4; ++e) out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + e] = acc[e];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 256 * 16);
  for (int shift = 0; shift < 4; ++shift)
    for (int mode = 0; mode < 2; ++mode) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, shift, 1, mode); hipDeviceSynchronize();
      std::vector<unsigned> h(256); hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
      bool ok = true; for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (h[l * 4 + e] != (unsigned)((l * 4 + shift + e) % 4096)) ok = false;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d, shift, 2000, mode); hipDeviceSynchronize();
      hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d, shift, 20000, mode); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("shift %d dwords  %s  values %s  %.3f ms for 20000 reads per lane\n", shift, mode == 0 ? "ds_read_b128   " : "4 x ds_read_b32", ok ? "OK" : "WRONG", ms);
    }
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
