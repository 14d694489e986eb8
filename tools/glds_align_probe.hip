// Probe: does global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) accept source addresses that are only 8- or 4-byte aligned?  Copies 1 KB through LDS from src + off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const unsigned char* src, unsigned char* dst, int off) {
  __shared__ __attribute__((aligned(16))) unsigned char buf[1024];
  __builtin_amdgcn_global_load_lds((gptr_t)(src + off + threadIdx.x * 16), (lptr_t)buf, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) dst[i] = buf[i];
}
int main() {
  std::vector<unsigned char> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (unsigned char)(i * 7 + 3);
  unsigned char *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 1024); hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
  for (int off : {0, 8, 4, 24, 2}) {
    hipMemset(d, 0, 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, d, off);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned char> o(1024); hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += o[i] != h[off + i];
    printf("offset %2d: %s, %d wrong bytes\n", off, hipGetErrorString(e), bad);
  }
  return 0;
}
