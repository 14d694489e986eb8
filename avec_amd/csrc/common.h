// Shared device helpers for the AVEC gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AVEC_F32 0
#define AVEC_BF16 1

typedef unsigned short bf16_raw;  // storage type for bfloat16

struct bf16 { bf16_raw v; };

__device__ __forceinline__ float bf16_to_f32(bf16_raw h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32); the compiler emits it for a __bf16 cast
// (a software RNE costs 5 VALU per element, which dominated the epilogues of the small kernels)
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) { return __builtin_bit_cast(bf16_raw, (__bf16)f); }
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {      // one v_cvt_pk_bf16_f32
  typedef float f32x2_ __attribute__((ext_vector_type(2))); typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}

template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int VEC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elt<bf16> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16* p) { return bf16_to_f32(p->v); }
  __device__ static __forceinline__ void st(bf16* p, float v) { p->v = f32_to_bf16(v); }
};

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return Elt<T>::ld(p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { Elt<T>::st(p, v); }

struct __attribute__((aligned(16))) chunk16 { uint32_t w[4]; };           // 16 B register/LDS chunk (ds_read/write_b128)
struct __attribute__((packed, aligned(4))) chunk16u { uint32_t w[4]; };  // 16 B global access, only dword alignment assumed
__device__ __forceinline__ chunk16 ldg16(const void* p) { chunk16u t = *(const chunk16u*)p; chunk16 o; o.w[0] = t.w[0]; o.w[1] = t.w[1]; o.w[2] = t.w[2]; o.w[3] = t.w[3]; return o; }

// ---- wave64 reductions by DPP (full-rate VALU; round 4): six levels, the result read from lane 63 and therefore wave-uniform.  The first version walked
// __shfl_xor = ds_bpermute_b32, an LDS round trip per level: in the one-wave-per-row kernels (LayerNorm forward / backward, softmax) the two dependent chains of six
// round trips were most of a row's latency.
#define AVEC_DPP_F(v, old, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
  v += AVEC_DPP_F(v, 0.f, 0xB1, 0xf);        // quad_perm [1,0,3,2]
  v += AVEC_DPP_F(v, 0.f, 0x4E, 0xf);        // quad_perm [2,3,0,1]
  v += AVEC_DPP_F(v, 0.f, 0x141, 0xf);       // row_half_mirror
  v += AVEC_DPP_F(v, 0.f, 0x140, 0xf);       // row_mirror: every lane of a 16-lane row holds the row's sum
  v += AVEC_DPP_F(v, 0.f, 0x142, 0xa);       // row_bcast:15 into rows 1 and 3
  v += AVEC_DPP_F(v, 0.f, 0x143, 0xc);       // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, AVEC_DPP_F(v, v, 0xB1, 0xf));
  v = fmaxf(v, AVEC_DPP_F(v, v, 0x4E, 0xf));
  v = fmaxf(v, AVEC_DPP_F(v, v, 0x141, 0xf));
  v = fmaxf(v, AVEC_DPP_F(v, v, 0x140, 0xf));
  v = fmaxf(v, AVEC_DPP_F(v, v, 0x142, 0xa));
  v = fmaxf(v, AVEC_DPP_F(v, v, 0x143, 0xc));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- counter-based RNG for dropout / SpecAugment: stateless, reproducible in backward ----
// u = hash(seed, stream, idx) in [0,1); keep <=> u >= p.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint32_t stream, uint64_t idx) {
  uint32_t a = mix32((uint32_t)idx ^ (uint32_t)seed);
  uint32_t b = mix32((uint32_t)(idx >> 32) + stream * 0x9e3779b9u + (uint32_t)(seed >> 32));
  uint32_t h = mix32(a ^ (b + 0x85ebca6bu + (a << 6) + (a >> 2)));
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// ---- dropout masks: ONE 32-bit hash per PAIR of consecutive elements, 16 bits each (p is resolved to 1/65536) ----
// The epilogues that apply dropout are VALU-bound on exactly this arithmetic (integer multiplies are quarter rate): the first version hashed every element with three
// mix32 rounds (6 multiplies) after re-reading {seed, step} from memory; now the key is built once per thread and a pair of elements costs one mix32 (2 multiplies).
// Element idx takes the low (even idx) / high (odd idx) half of hash(idx >> 1): forward and backward, fused and unfused kernels agree as long as they use these helpers.
struct DropKey { uint32_t k0, thr; float scale; };
__device__ __forceinline__ DropKey drop_key(const unsigned long long* rng, uint32_t stream, float p) {
  DropKey k; k.k0 = 0u; k.thr = 0u; k.scale = 1.f;
  if (p > 0.f) {
    const uint64_t seed = rng[0] + 0x9e3779b97f4a7c15ull * rng[1];
    k.k0 = mix32((uint32_t)seed + stream * 0x9e3779b9u) ^ mix32((uint32_t)(seed >> 32) ^ 0x85ebca6bu);
    // p is resolved to thr / 65536 and the scale follows the QUANTISED probability (E[mask * scale] = 1 exactly); p >= 1 drops everything with scale 0 like torch
    // (thr = 65536 is above every 16-bit hash value) instead of letting the 1 / 65536 survivors through at 1 / (1 - p) = inf
    const float t = p * 65536.f + 0.5f; k.thr = t >= 65536.f ? 65536u : (uint32_t)t;
    k.scale = k.thr >= 65536u ? 0.f : 65536.f / (float)(65536u - k.thr);
  }
  return k;
}
__device__ __forceinline__ uint32_t drop_hash(const DropKey& k, uint64_t pair) {
  uint32_t x = (uint32_t)pair ^ k.k0;
  const uint32_t hi = (uint32_t)(pair >> 32);
  if (hi) x ^= hi * 0x9e3779b1u;                      // (tensors beyond 2^33 elements only)
  return mix32(x);
}
// scale factors (0 or 1/(1-p)) of the elements 2 pair and 2 pair + 1
__device__ __forceinline__ void drop_pair(const DropKey& k, uint64_t pair, float& s0, float& s1) {
  const uint32_t h = drop_hash(k, pair);
  s0 = (h & 0xffffu) >= k.thr ? k.scale : 0.f; s1 = (h >> 16) >= k.thr ? k.scale : 0.f;
}
// ... of the 4 consecutive elements idx .. idx + 3 (idx even)
__device__ __forceinline__ void drop4(const DropKey& k, uint64_t idx, float (&s)[4]) { drop_pair(k, idx >> 1, s[0], s[1]); drop_pair(k, (idx >> 1) + 1, s[2], s[3]); }
__device__ __forceinline__ float drop_one(const DropKey& k, uint64_t idx) {
  const uint32_t h = drop_hash(k, idx >> 1);
  return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= k.thr ? k.scale : 0.f;
}
// single-element form (builds the key every call: keep it out of inner loops)
__device__ __forceinline__ float drop_scale(const unsigned long long* rng, uint32_t stream, uint64_t idx, float p) {
  if (p <= 0.f) return 1.f;
  const DropKey k = drop_key(rng, stream, p);
  return drop_one(k, idx);
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU instructions): Swish / GLU / their derivatives sit in VALU-bound epilogues
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float dswishf_(float x) { float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }

// ---- host-side error plumbing (api.hip) ----
extern "C" const char* avec_last_error();
void avec_set_error(const char* fmt, ...);
void avec_note_kernel(const char* fmt, ...);      // api.hip: remembers which kernel instance an entry point chose (avec_last_kernel)
#define AVEC_CHECK_ARG(cond, ...) do { if (!(cond)) { avec_set_error(__VA_ARGS__); return -1; } } while (0)
#define AVEC_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { avec_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return (int)e_; } } while (0)
