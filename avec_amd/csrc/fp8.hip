// OCP e4m3 operands for the forward Linear products (BASELINE config 5's "fp8 GEMMs"): per-tensor scaling, quantizers and the bookkeeping kernels.
// The GEMM itself (v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate) lives in gemm.hip beside the epilogue it shares with the bf16 kernels.
//
// Scaling is "current", not delayed: an absolute-maximum pass writes amax[slot] (atomic max on the bit pattern of |x|), the quantizer and the GEMM both
// derive scale = max(amax, tiny) / 448 from it, so there is no history and no first-step special case; the caller zeroes the amax array once per pass.
#include "vec.h"
#include "avec_hip.h"

__device__ __forceinline__ float fp8_scale_of(float amax) { return fmaxf(amax, 1e-20f) * (1.0f / 448.0f); }

// two floats -> two e4m3 bytes (round to nearest even; inputs pre-clamped to +-448 so the conversion never meets its overflow rule)
__device__ __forceinline__ unsigned fp8_pack2(float a, float b, unsigned old, bool hi) {
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  return hi ? (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true) : (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false);
}

__device__ __forceinline__ void amax_commit(float local, float* amax, float* red /* >= 4 floats of LDS */) {
  for (int o = 32; o > 0; o >>= 1) local = fmaxf(local, __shfl_xor(local, o, 64));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
    if (m > 0.f) atomicMax((unsigned*)amax, __float_as_uint(m));          // non-negative floats order like their bit patterns
  }
}

// ---- activations: |x| maximum of a [M][K] matrix (row stride ldx), then x / scale -> e4m3 [M][K] (row stride ldq) ----------------------------
template <typename T>
__global__ __launch_bounds__(256) void fp8_absmax_kernel(const T* __restrict__ x, long long ldx, long long M, int K, float* amax) {
  __shared__ float red[4];
  const int kc = K / 8;                                    // 8-element chunks per row (host: K % 16 == 0)
  const long long total = M * kc;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / kc; const int c = (int)(i - r * kc);
    float v[8]; ld8<T>(x + r * ldx + c * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
  }
  amax_commit(m, amax, red);
}

template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_kernel(const T* __restrict__ x, long long ldx, unsigned char* __restrict__ q, long long ldq, long long M, int K,
                                                        const float* __restrict__ amax) {
  const float inv = 1.0f / fp8_scale_of(*amax);
  const int kc = (K + 15) / 16;                            // 16-element (16-byte) output chunks per row; K % 8 == 0: the last one may be half zeros
  const long long total = M * kc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / kc; const int c = (int)(i - r * kc);
    float v[16]; ld8<T>(x + r * ldx + c * 16, v);
    if (c * 16 + 8 < K) ld8<T>(x + r * ldx + c * 16 + 8, v + 8);
    else { for (int e = 8; e < 16; ++e) v[e] = 0.f; }
    uint4 o; unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { unsigned t = fp8_pack2(v[4 * k] * inv, v[4 * k + 1] * inv, 0u, false); w[k] = fp8_pack2(v[4 * k + 2] * inv, v[4 * k + 3] * inv, t, true); }
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *(uint4*)(q + r * ldq + c * 16) = o;
  }
}

extern "C" int avec_fp8_quantize(int src_dtype, const void* x, long long ldx, void* q, long long ldq, long long M, int K, float* amax, int compute_amax, hipStream_t stream) {
  AVEC_CHECK_ARG(src_dtype == AVEC_F32 || src_dtype == AVEC_BF16, "fp8_quantize: bad dtype %d", src_dtype);
  AVEC_CHECK_ARG(x && q && amax && M > 0 && K > 0, "fp8_quantize: null pointer / empty matrix");
  AVEC_CHECK_ARG(K % 8 == 0 && ldx % 8 == 0 && ldq % 16 == 0 && ldq >= (K + 15) / 16 * 16 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 15) == 0,
                 "fp8_quantize: K=%d must be a multiple of 8, ldx=%lld of 8, ldq=%lld a multiple of 16 and >= K rounded up to 16, pointers 16-byte aligned", K, ldx, ldq);
  const long long chunks = M * ((K + 15) / 16);
  const unsigned blocks = (unsigned)((chunks + 255) / 256 > 2048 ? 2048 : (chunks + 255) / 256);
  if (compute_amax) DISPATCH_T(src_dtype, hipLaunchKernelGGL((fp8_absmax_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)x, ldx, M, K, amax));
  DISPATCH_T(src_dtype, hipLaunchKernelGGL((fp8_quant_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)x, ldx, (unsigned char*)q, ldq, M, K, (const float*)amax));
  AVEC_LAUNCH_CHECK();
  return 0;
}

// ---- weights: a device table of fp32 master matrices -> e4m3 shadows, all in two launches --------------------------------------------------
struct Fp8Item { const float* src; unsigned char* dst; long long n; int slot; int K, Kp; int pad; };     // n = rows * K source elements; dst rows are Kp bytes apart (zero padded)
static_assert(sizeof(Fp8Item) == sizeof(avec_fp8_item_t), "avec_fp8_item_t layout");

__global__ __launch_bounds__(256) void fp8_table_absmax_kernel(const Fp8Item* __restrict__ tab, float* amax) {
  __shared__ float red[4];
  const Fp8Item it = tab[blockIdx.y];
  float m = 0.f;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < it.n; i += (long long)gridDim.x * 1024) {
    float v[4]; ld4<float>(it.src + i, v);
    m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(m, fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  amax_commit(m, amax + it.slot, red);
}

__global__ __launch_bounds__(256) void fp8_table_quant_kernel(const Fp8Item* __restrict__ tab, const float* __restrict__ amax) {
  const Fp8Item it = tab[blockIdx.y];
  const float inv = 1.0f / fp8_scale_of(amax[it.slot]);
  const long long nq = it.n / it.K * it.Kp;               // destination bytes
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < nq; i += (long long)gridDim.x * 1024) {
    const long long r = i / it.Kp; const int c = (int)(i - r * it.Kp);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < it.K) ld4<float>(it.src + r * it.K + c, v);
    const unsigned t = fp8_pack2(v[0] * inv, v[1] * inv, 0u, false);
    *(unsigned*)(it.dst + i) = fp8_pack2(v[2] * inv, v[3] * inv, t, true);
  }
}

extern "C" int avec_fp8_weights_refresh(const avec_fp8_item_t* table, int n_items, int blocks_per_item, float* amax, hipStream_t stream) {
  AVEC_CHECK_ARG(table && amax && n_items > 0 && blocks_per_item > 0 && blocks_per_item <= 1024, "fp8_weights_refresh: bad arguments");
  hipLaunchKernelGGL(fp8_table_absmax_kernel, dim3(blocks_per_item, n_items), dim3(256), 0, stream, (const Fp8Item*)table, amax);
  hipLaunchKernelGGL(fp8_table_quant_kernel, dim3(blocks_per_item, n_items), dim3(256), 0, stream, (const Fp8Item*)table, (const float*)amax);
  AVEC_LAUNCH_CHECK();
  return 0;
}
