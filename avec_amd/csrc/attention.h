// Shared argument block of the relative-position attention kernels (attention.hip: fp32 VALU kernels; attention_mfma.hip: bf16 MFMA kernels).
#pragma once
#include "common.h"
#include "avec_hip.h"

struct AttnArgs {
  const void *q, *k, *v; long long ld;      // act, row stride (elements); head h occupies columns [h*d, (h+1)*d)
  const void* e; long long lde;             // act [2T-1][lde]
  const long long* lens; int len_div;       // key j kept iff j < lens[b] / len_div  (null: all kept)
  int q_full;                               // query rows i >= q_full see every key masked (the zero-padded last patch, nnet/attentions.py:152-154,357-362)
  const float* mask; long long mask_bstride; // optional dense mask [Bm][T][T] (1 = keep); overrides lens
  void* o; long long ldo;                    // act [B*T][ldo]
  float* lse;                                // [B*H][T][2] = (row max m, row sum l): kept apart, m + log l loses log l when every key is masked (m = -1e9)
  const void* dout;                          // act [B*T][ldo]   (backward)
  void *dq, *dk, *dv; long long lddq, ldd;   // dq: act, row stride lddq; dk/dv: row stride ldd
  float* de; long long ldde;                 // fp32 [2T-1][ldde], atomically accumulated
  void *pbuf, *dsbuf; long long ldt;         // backward scratch, act [B*H][T][ldt]: probabilities and dS (written by the dQ pass)
  void* dsrel; long long ldr;                // optional act [H][B*T][ldr]: dS re-indexed by E row r = j + (T-1) - i (zero elsewhere; caller zero-fills)
  int B, H, T, d; float scale;
  int Tk;                                    // keys / values per batch element (= T unless a key/value cache is attached: decoding, forward + probability pass only)
};

template <typename T>
__device__ __forceinline__ bool key_keep(const AttnArgs& a, int b, int i, int j) {
  if (a.mask) return a.mask[(long long)b * a.mask_bstride + (long long)i * a.Tk + j] != 0.f;
  if (i >= a.q_full) return false;
  if (a.lens) return j < (int)(a.lens[b] / a.len_div);
  return true;
}


// bf16 MFMA path (attention_mfma.hip): 0 = launched, 1 = shape/mask not supported (caller falls back to the VALU kernels), other = error
int attn_mfma_fwd(const AttnArgs& a, hipStream_t st);
int attn_mfma_bwd_rows(const AttnArgs& a, hipStream_t st);
