// Probe of ds_read_b64_tr_b16 semantics on gfx950: prints, for every lane and element j, which LDS element index it received
// when lane l supplies the address of elements [4l, 4l+4).   hipcc --offload-arch=gfx950 tools/tr16_probe.hip -o tools/_bin/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + threadIdx.x * 4));
  out[threadIdx.x * 4 + 0] = r.x; out[threadIdx.x * 4 + 1] = r.y; out[threadIdx.x * 4 + 2] = r.z; out[threadIdx.x * 4 + 3] = r.w;
}
int main() {
  short* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int exp = (l & 15) + j * 16 + (l >> 4) * 64; if (h[l * 4 + j] != exp) ok = 0; }
  for (int l = 0; l < 20; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  printf("matches lds[(l&15) + j*16 + (l>>4)*64]: %s\n", ok ? "yes" : "NO");
  return 0;
}
