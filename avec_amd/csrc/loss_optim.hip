// CTC loss (log-softmax + alpha/beta recursion + gradient) and the flat-arena optimizer step.
//   CTC      : nnet/losses.py:311-334  (log_softmax -> nn.CTCLoss(blank=0, reduction="none", zero_infinity))
//   Adam     : nnet/optimizers.py:61-93 over torch.optim.Adam (coupled L2 weight decay), lr from a device scalar
//   shadows  : compute-dtype copies of the GEMM weights in the two layouts the NT kernels want
#include <atomic>
#include "vec.h"
#include "avec_hip.h"

__device__ __forceinline__ float logaddexpf_(float a, float b) {
  if (a == -INFINITY) return b; if (b == -INFINITY) return a;
  const float m = fmaxf(a, b); return m + log1pf(__expf(-fabsf(a - b)));
}

// one wave per utterance; states s = 0..S-1 (S = 2L+1) strided over lanes.
// ws: per utterance [T][S] alphas followed by [T] frame log-normalisers.
__global__ __launch_bounds__(64) void ctc_kernel(const float* __restrict__ logits, const long long* __restrict__ in_lens, const long long* __restrict__ targets,
                                                 const long long* __restrict__ tgt_lens, float* __restrict__ nll, float* __restrict__ mean_out, float* __restrict__ grad,
                                                 float* __restrict__ ws, int B, int T, int V, int Lmax, int blank, int zero_inf) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int Smax = 2 * Lmax + 1;
  float* beta0 = sm; float* beta1 = sm + Smax; float* occ = beta1 + Smax;      // occ[V]
  int* ext = (int*)(occ + V);                                                   // ext[Smax]
  const int Tb = min((int)in_lens[b], T), L = min((int)tgt_lens[b], Lmax), S = 2 * L + 1;
  float* alpha = ws + (long long)b * ((long long)T * Smax + T); float* lnorm = alpha + (long long)T * Smax;
  const float* lg = logits + (long long)b * T * V;
  float* gr = grad ? grad + (long long)b * T * V : nullptr;
  for (int s = lane; s < S; s += 64) ext[s] = (s & 1) ? (int)targets[(long long)b * Lmax + (s >> 1)] : blank;
  // frame log-normalisers
  for (int t = 0; t < Tb; ++t) {
    float mx = -INFINITY; for (int v = lane; v < V; v += 64) mx = fmaxf(mx, lg[t * V + v]);
    mx = wave_max(mx);
    float se = 0.f; for (int v = lane; v < V; v += 64) se += __expf(lg[t * V + v] - mx);
    se = wave_sum(se);
    if (lane == 0) lnorm[t] = mx + __logf(se);
  }
  __syncthreads();
  // alpha
  float ll = -INFINITY;
  if (Tb > 0) {
    for (int s = lane; s < S; s += 64) alpha[s] = (s < 2) ? lg[ext[s]] - lnorm[0] : -INFINITY;
    __syncthreads();
    for (int t = 1; t < Tb; ++t) {
      const float* ap = alpha + (long long)(t - 1) * Smax; float* an = alpha + (long long)t * Smax;
      for (int s = lane; s < S; s += 64) {
        float a = ap[s];
        if (s >= 1) a = logaddexpf_(a, ap[s - 1]);
        if (s >= 2 && ext[s] != blank && ext[s] != ext[s - 2]) a = logaddexpf_(a, ap[s - 2]);
        an[s] = a + lg[t * V + ext[s]] - lnorm[t];
      }
      __syncthreads();
    }
    const float* al = alpha + (long long)(Tb - 1) * Smax;
    ll = al[S - 1]; if (S > 1) ll = logaddexpf_(ll, al[S - 2]);
  } else if (L == 0) ll = 0.f;
  float loss = -ll;
  const bool inf = !(loss < INFINITY);     // inf or nan
  if (inf && zero_inf) loss = 0.f;
  if (lane == 0) { nll[b] = loss; if (mean_out) atomicAdd(mean_out, loss / B); }
  if (!gr) return;
  // gradient w.r.t. logits:  softmax - occupancy   (zero where t >= Tb or the alignment is infeasible)
  for (int i = lane; i < (T - Tb) * V; i += 64) gr[(long long)Tb * V + i] = 0.f;
  if (inf || Tb == 0) { for (int i = lane; i < Tb * V; i += 64) gr[i] = 0.f; return; }
  float* bc = beta0; float* bn = beta1;
  for (int t = Tb - 1; t >= 0; --t) {
    for (int s = lane; s < S; s += 64) {
      float bv;
      if (t == Tb - 1) bv = (s >= S - 2) ? 0.f : -INFINITY;
      else {
        bv = bn[s];
        if (s + 1 < S) bv = logaddexpf_(bv, bn[s + 1]);
        if (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != ext[s]) bv = logaddexpf_(bv, bn[s + 2]);
      }
      bc[s] = bv + lg[t * V + ext[s]] - lnorm[t];
    }
    for (int v = lane; v < V; v += 64) occ[v] = 0.f;
    __syncthreads();
    const float* at = alpha + (long long)t * Smax;
    for (int s = lane; s < S; s += 64) {
      const float lp = lg[t * V + ext[s]] - lnorm[t];
      const float term = __expf(at[s] + bc[s] - lp - ll);
      atomicAdd(occ + ext[s], term);
    }
    __syncthreads();
    for (int v = lane; v < V; v += 64) gr[t * V + v] = __expf(lg[t * V + v] - lnorm[t]) - occ[v];
    __syncthreads();
    float* tmp = bc; bc = bn; bn = tmp;
  }
}

// LDS-resident variant (used when 3*T*S + T + 4*V floats fit in 64 KB): 256 threads per utterance.
//   phase 1  frame log-normalisers, one frame per wave, then log p(ext[s] | t) for every (t, s) in parallel
//   phase 2  alpha (wave 0) and beta (wave 1) recursions run concurrently and wave-synchronously (no workgroup barrier per frame), LDS only
//   phase 3  gradient rows, one frame per wave (occupancy scatter into a per-wave LDS histogram)
__device__ __forceinline__ void ctc_lds_body(const float* __restrict__ logits, const long long* __restrict__ in_lens, const long long* __restrict__ targets,
                                             const long long* __restrict__ tgt_lens, float* __restrict__ nll, float* __restrict__ mean_out, float* __restrict__ grad,
                                             int B, int T, int V, int Lmax, int blank, int zero_inf, int b, float* total = nullptr, float wtot = 0.f) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NT = blockDim.x, NW = NT >> 6;      // 256 or 1024 threads (CTC_WAVES of the launch)
  const int Smax = 2 * Lmax + 1;
  float* lpe = sm; float* alpha = lpe + T * Smax; float* beta = alpha + T * Smax; float* lnorm = beta + T * Smax;
  float* occ = lnorm + T; int* ext = (int*)(occ + NW * V);
  const int Tb = min((int)in_lens[b], T), L = min((int)tgt_lens[b], Lmax), S = 2 * L + 1;
  const float* lg = logits + (long long)b * T * V;
  float* gr = grad ? grad + (long long)b * T * V : nullptr;
  for (int s = tid; s < S; s += NT) ext[s] = (s & 1) ? (int)targets[(long long)b * Lmax + (s >> 1)] : blank;
  // frame log-normalisers: four frames per wave in flight (each is a load -> max -> exp -> sum chain of its own)
  for (int t0 = wv * 4; t0 < Tb; t0 += 4 * NW) {
    float mx[4], se[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int t = min(t0 + q, Tb - 1); mx[q] = -INFINITY; for (int v = lane; v < V; v += 64) mx[q] = fmaxf(mx[q], lg[t * V + v]); }
#pragma unroll
    for (int q = 0; q < 4; ++q) mx[q] = wave_max(mx[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int t = min(t0 + q, Tb - 1); se[q] = 0.f; for (int v = lane; v < V; v += 64) se[q] += __expf(lg[t * V + v] - mx[q]); }
#pragma unroll
    for (int q = 0; q < 4; ++q) se[q] = wave_sum(se[q]);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (t0 + q < Tb) lnorm[t0 + q] = mx[q] + __logf(se[q]);
    }
  }
  __syncthreads();
  for (int i = tid; i < Tb * S; i += NT) { const int t = i / S, s = i - t * S; lpe[t * Smax + s] = lg[t * V + ext[s]] - lnorm[t]; }
  __syncthreads();
  float ll = -INFINITY;
  if (Tb > 0) {
    // wave 0 runs the forward recursion, wave 1 the backward one, each on its own rows: LDS operations of one wave complete in order, so a step needs no workgroup
    // barrier (it was ~1 us per frame with 128 threads per direction and __syncthreads; the fence keeps the compiler from moving a row's reads above its writes)
    const bool fw = wv == 0; const int s0 = lane;
    if (wv < 2) for (int k = 0; k < Tb; ++k) {
      if (fw) {
        const int t = k; float* an = alpha + t * Smax; const float* ap = an - Smax;
        for (int s = s0; s < S; s += 64) {
          float a;
          if (t == 0) a = (s < 2) ? 0.f : -INFINITY;
          else {
            a = ap[s];
            if (s >= 1) a = logaddexpf_(a, ap[s - 1]);
            if (s >= 2 && ext[s] != blank && ext[s] != ext[s - 2]) a = logaddexpf_(a, ap[s - 2]);
          }
          an[s] = a + lpe[t * Smax + s];
        }
      } else {
        const int t = Tb - 1 - k; float* bc = beta + t * Smax; const float* bn = bc + Smax;
        for (int s = s0; s < S; s += 64) {
          float bv;
          if (k == 0) bv = (s >= S - 2) ? 0.f : -INFINITY;
          else {
            bv = bn[s];
            if (s + 1 < S) bv = logaddexpf_(bv, bn[s + 1]);
            if (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != ext[s]) bv = logaddexpf_(bv, bn[s + 2]);
          }
          bc[s] = bv + lpe[t * Smax + s];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const float* al = alpha + (Tb - 1) * Smax;
    ll = al[S - 1]; if (S > 1) ll = logaddexpf_(ll, al[S - 2]);
  } else if (L == 0) ll = 0.f;
  float loss = -ll;
  const bool inf = !(loss < INFINITY);
  if (inf && zero_inf) loss = 0.f;
  if (tid == 0) { nll[b] = loss; if (mean_out) atomicAdd(mean_out, loss / B); if (total) atomicAdd(total, wtot * loss / B); }
  if (!gr) return;
  for (int i = tid; i < (T - Tb) * V; i += NT) gr[(long long)Tb * V + i] = 0.f;
  if (inf || Tb == 0) { for (int i = tid; i < Tb * V; i += NT) gr[i] = 0.f; return; }
  // gradient rows: one frame per wave per pass; the occupancy histogram is private to the wave and LDS operations of one wave complete in order,
  // so the passes need no workgroup barrier
  float* oc = occ + wv * V;
  for (int t = wv; t < Tb; t += NW) {
    for (int v = lane; v < V; v += 64) oc[v] = 0.f;
    for (int s = lane; s < S; s += 64) atomicAdd(oc + ext[s], __expf(alpha[t * Smax + s] + beta[t * Smax + s] - lpe[t * Smax + s] - ll));
    for (int v = lane; v < V; v += 64) gr[t * V + v] = __expf(lg[t * V + v] - lnorm[t]) - oc[v];
  }
}

__global__ __launch_bounds__(1024) void ctc_lds_kernel(const float* __restrict__ logits, const long long* __restrict__ in_lens, const long long* __restrict__ targets,
                                                      const long long* __restrict__ tgt_lens, float* __restrict__ nll, float* __restrict__ mean_out, float* __restrict__ grad,
                                                      int B, int T, int V, int Lmax, int blank, int zero_inf) {
  ctc_lds_body(logits, in_lens, targets, tgt_lens, nll, mean_out, grad, B, T, V, Lmax, blank, zero_inf, blockIdx.x);
}

// Several CTC heads over the same batch and labels in ONE launch (the InterCTC model evaluates six of them back to back, 32 workgroups each: together they fill
// 192 CUs instead of 32 six times).  blockIdx.x = head * B + utterance.
#define AVEC_CTC_MAX_HEADS 8
struct CtcHeads { const float* logits[AVEC_CTC_MAX_HEADS]; const long long* in_lens[AVEC_CTC_MAX_HEADS]; float* nll[AVEC_CTC_MAX_HEADS]; float* mean_out[AVEC_CTC_MAX_HEADS];
                  float* grad[AVEC_CTC_MAX_HEADS]; int T[AVEC_CTC_MAX_HEADS]; float w[AVEC_CTC_MAX_HEADS]; float* total; };
__global__ __launch_bounds__(1024) void ctc_lds_multi_kernel(CtcHeads h, const long long* __restrict__ targets, const long long* __restrict__ tgt_lens,
                                                            int B, int V, int Lmax, int blank, int zero_inf) {
  const int head = blockIdx.x / B, b = blockIdx.x - head * B;
  ctc_lds_body(h.logits[head], h.in_lens[head], targets, tgt_lens, h.nll[head], h.mean_out[head], h.grad[head], B, h.T[head], V, Lmax, blank, zero_inf, b, h.total, h.w[head]);
}

// Long-utterance variant (used when the three [T][S] arrays above do not fit: 15 s clips have T = 376 frames at the inter-CTC heads): only alpha lives in LDS.
//   phase 1  frame log-normalisers; the gradient rows are pre-set to 0
//   phase 2  alpha recursion over the frames (one barrier per frame; the emission of the next frame is fetched from the L2-resident logits one frame ahead)
//   phase 3  beta recursion with two rolling rows; each state adds its occupancy exp(alpha + beta - emission - ll) to its label's gradient entry with a
//            fire-and-forget global atomic
//   phase 4  gradient rows in parallel: softmax - occupancy
__global__ __launch_bounds__(256) void ctc_alpha_lds_kernel(const float* __restrict__ logits, const long long* __restrict__ in_lens, const long long* __restrict__ targets,
                                                            const long long* __restrict__ tgt_lens, float* __restrict__ nll, float* __restrict__ mean_out, float* __restrict__ grad,
                                                            int B, int T, int V, int Lmax, int blank, int zero_inf) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int Smax = 2 * Lmax + 1;
  float* alpha = sm; float* lnorm = alpha + (size_t)T * Smax; float* brow = lnorm + T; int* ext = (int*)(brow + 2 * Smax);
  const int Tb = min((int)in_lens[b], T), L = min((int)tgt_lens[b], Lmax), S = 2 * L + 1;
  const float* lg = logits + (long long)b * T * V;
  float* gr = grad ? grad + (long long)b * T * V : nullptr;
  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? (int)targets[(long long)b * Lmax + (s >> 1)] : blank;
  for (int t = wv; t < Tb; t += 4) {
    float mx = -INFINITY; for (int v = lane; v < V; v += 64) mx = fmaxf(mx, lg[t * V + v]);
    mx = wave_max(mx);
    float se = 0.f; for (int v = lane; v < V; v += 64) se += __expf(lg[t * V + v] - mx);
    se = wave_sum(se);
    if (lane == 0) lnorm[t] = mx + __logf(se);
  }
  if (gr) for (int i = tid; i < T * V; i += 256) gr[i] = 0.f;
  __syncthreads();
  float ll = -INFINITY;
  if (Tb > 0) {
    // forward
    const int NS = (S + 255) / 256;
    for (int t = 0; t < Tb; ++t) {
      float* an = alpha + (size_t)t * Smax; const float* ap = an - Smax;
      for (int q = 0; q < NS; ++q) {
        const int s = tid + q * 256; if (s >= S) break;
        float a;
        if (t == 0) a = (s < 2) ? 0.f : -INFINITY;
        else {
          a = ap[s];
          if (s >= 1) a = logaddexpf_(a, ap[s - 1]);
          if (s >= 2 && ext[s] != blank && ext[s] != ext[s - 2]) a = logaddexpf_(a, ap[s - 2]);
        }
        an[s] = a + (lg[t * V + ext[s]] - lnorm[t]);
      }
      __syncthreads();
    }
    const float* al = alpha + (size_t)(Tb - 1) * Smax;
    ll = al[S - 1]; if (S > 1) ll = logaddexpf_(ll, al[S - 2]);
  } else if (L == 0) ll = 0.f;
  float loss = -ll;
  const bool inf = !(loss < INFINITY);
  if (inf && zero_inf) loss = 0.f;
  if (tid == 0) { nll[b] = loss; if (mean_out) atomicAdd(mean_out, loss / B); }
  if (!gr || inf || Tb == 0) return;          // gradient rows already zero
  {
    const int NS = (S + 255) / 256;
    for (int k = 0; k < Tb; ++k) {
      const int t = Tb - 1 - k; float* bc = brow + (k & 1) * Smax; const float* bn = brow + ((k & 1) ^ 1) * Smax;
      for (int q = 0; q < NS; ++q) {
        const int s = tid + q * 256; if (s >= S) break;
        float bv;
        if (k == 0) bv = (s >= S - 2) ? 0.f : -INFINITY;
        else {
          bv = bn[s];
          if (s + 1 < S) bv = logaddexpf_(bv, bn[s + 1]);
          if (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != ext[s]) bv = logaddexpf_(bv, bn[s + 2]);
        }
        const float e = lg[t * V + ext[s]] - lnorm[t];
        bv += e;
        bc[s] = bv;
        const float w = __expf(alpha[(size_t)t * Smax + s] + bv - e - ll);
        if (w > 0.f) atomicAdd(gr + t * V + ext[s], w);
      }
      __syncthreads();
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < Tb * V; i += 256) { const int t = i / V; gr[i] = __expf(lg[i] - lnorm[t]) - gr[i]; }
}

extern "C" long long avec_ctc_workspace_floats(int B, int T, int Lmax) { return (long long)B * ((long long)T * (2 * Lmax + 1) + T); }

// Waves per utterance of the all-LDS kernels: the log-normaliser and gradient phases work frame by frame, one frame per wave and pass -- 16 waves when their
// occupancy histograms still fit the 64 KB (each is V floats), else 4
static int ctc_waves(size_t lds4, int V) {
  static const int want = 16;
  if (want < 16 || lds4 + (size_t)12 * V * 4 > 128 * 1024) return 4;
  // (more than 64 KB of dynamic LDS has to be asked for once per kernel AND device; the call is cheap, the state is per device and published with release / acquire
  //  so that a second device or a concurrent first call from the autograd thread never launches without it)
  static std::atomic<int> state[16];                        // per device: 0 unknown, 1 granted, 2 refused
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  int stt = state[dev].load(std::memory_order_acquire);
  if (stt == 0) {
    const bool ok = hipFuncSetAttribute((const void*)ctc_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
                    hipFuncSetAttribute((const void*)ctc_lds_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    stt = ok ? 1 : 2;
    state[dev].store(stt, std::memory_order_release);
  }
  const bool attr_ok = stt == 1;
  return (attr_ok || lds4 + (size_t)12 * V * 4 <= 64 * 1024) ? 16 : 4;
}
extern "C" int avec_ctc_loss(const float* logits, const long long* in_lens, const long long* targets, const long long* tgt_lens, float* nll, float* mean_out,
                             float* grad, float* workspace, int B, int T, int V, int Lmax, int blank, int zero_infinity, hipStream_t st) {
  AVEC_CHECK_ARG(logits && in_lens && targets && tgt_lens && nll && workspace, "ctc_loss: null pointer");
  AVEC_CHECK_ARG(B > 0 && T > 0 && V > 0 && Lmax >= 0 && blank >= 0 && blank < V, "ctc_loss: bad dims B=%d T=%d V=%d Lmax=%d", B, T, V, Lmax);
  const size_t Smax = 2 * (size_t)Lmax + 1;
  const size_t lds_fast = (3 * (size_t)T * Smax + T + 4 * (size_t)V + Smax) * 4;
  if (lds_fast <= 64 * 1024) {
    const int nw = ctc_waves(lds_fast, V);
    hipLaunchKernelGGL(ctc_lds_kernel, dim3(B), dim3(64 * nw), lds_fast + (size_t)(nw - 4) * V * 4, st, logits, in_lens, targets, tgt_lens, nll, mean_out, grad, B, T, V, Lmax, blank, zero_infinity);
    AVEC_LAUNCH_CHECK(); return 0;
  }
  const size_t lds_alpha = ((size_t)T * Smax + T + 3 * Smax) * 4;
  static const bool no_alpha = getenv("AVEC_CTC_NO_ALPHA_LDS") != nullptr;
  if (lds_alpha <= 150 * 1024 && !no_alpha) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)ctc_alpha_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      if (e != hipSuccess) { avec_set_error("ctc_loss: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
      attr_set = true;
    }
    hipLaunchKernelGGL(ctc_alpha_lds_kernel, dim3(B), dim3(256), lds_alpha, st, logits, in_lens, targets, tgt_lens, nll, mean_out, grad, B, T, V, Lmax, blank, zero_infinity);
    AVEC_LAUNCH_CHECK(); return 0;
  }
  size_t lds = (size_t)(2 * (2 * Lmax + 1) + V) * 4 + (size_t)(2 * Lmax + 1) * 4;
  AVEC_CHECK_ARG(lds <= 60 * 1024, "ctc_loss: label length %d too long for the LDS state buffers", Lmax);
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(64), lds, st, logits, in_lens, targets, tgt_lens, nll, mean_out, grad, workspace, B, T, V, Lmax, blank, zero_infinity);
  AVEC_LAUNCH_CHECK(); return 0;
}

extern "C" int avec_ctc_loss_multi(int n_heads, const float* const* logits, const long long* const* in_lens, const int* T, float* const* nll, float* const* mean_out, float* const* grad,
                                   const long long* targets, const long long* tgt_lens, const float* weights, float* total, int B, int V, int Lmax, int blank, int zero_infinity, hipStream_t st) {
  AVEC_CHECK_ARG(!total || weights, "ctc_loss_multi: a weighted total needs the weights");
  AVEC_CHECK_ARG(n_heads >= 1 && n_heads <= AVEC_CTC_MAX_HEADS && logits && in_lens && T && nll && mean_out && grad && targets && tgt_lens, "ctc_loss_multi: bad arguments (%d heads)", n_heads);
  AVEC_CHECK_ARG(B > 0 && V > 0 && Lmax >= 0 && blank >= 0 && blank < V, "ctc_loss_multi: bad dims");
  CtcHeads h; size_t lds = 0; h.total = total;
  const size_t Smax = 2 * (size_t)Lmax + 1;
  for (int i = 0; i < AVEC_CTC_MAX_HEADS; ++i) {
    const int k = i < n_heads ? i : 0;
    AVEC_CHECK_ARG(logits[k] && in_lens[k] && nll[k] && T[k] > 0, "ctc_loss_multi: null buffer in head %d", k);
    h.logits[i] = logits[k]; h.in_lens[i] = in_lens[k]; h.nll[i] = nll[k]; h.mean_out[i] = mean_out[k]; h.grad[i] = grad[k]; h.T[i] = T[k]; h.w[i] = (weights && i < n_heads) ? weights[i] : 0.f;
    const size_t need = (3 * (size_t)T[k] * Smax + T[k] + 4 * (size_t)V + Smax) * 4;
    if (need > lds) lds = need;
  }
  AVEC_CHECK_ARG(lds <= 64 * 1024, "ctc_loss_multi: a head does not fit the all-LDS kernel (use avec_ctc_loss per head)");
  const int nw = ctc_waves(lds, V);
  hipLaunchKernelGGL(ctc_lds_multi_kernel, dim3((unsigned)(n_heads * B)), dim3(64 * nw), lds + (size_t)(nw - 4) * V * 4, st, h, targets, tgt_lens, B, V, Lmax, blank, zero_infinity);
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_ctc_loss_multi_fits(int T, int V, int Lmax) { const size_t S = 2 * (size_t)Lmax + 1; return (3 * (size_t)T * S + T + 4 * (size_t)V + S) * 4 <= 64 * 1024; }

// nn.CrossEntropyLoss(ignore_index, reduction='none') as losses.SoftmaxCrossEntropy uses it (nnet/losses.py:258-290): one wave per row,
//   loss[m] = logsumexp(x[m]) - x[m][y[m]]  (0 when y[m] == ignore_index);  mean_out += loss[m] / M  (Reduction('mean') = mean over ALL rows);
//   grad[m] = softmax(x[m]) - onehot(y[m])  (zero row when ignored)
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ x, const long long* __restrict__ y, long long ignore, float* __restrict__ loss,
                                                         float* mean_out, float* __restrict__ grad, long long M, int V) {
  const int lane = threadIdx.x & 63; const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + row * V; const long long t = y[row]; const bool ign = t == ignore || t < 0 || t >= V;
  float mx = -INFINITY; for (int c = lane; c < V; c += 64) mx = fmaxf(mx, xr[c]);
  mx = wave_max(mx);
  float se = 0.f; for (int c = lane; c < V; c += 64) se += __expf(xr[c] - mx);
  se = wave_sum(se);
  const float lse = mx + __logf(se);
  const float l = ign ? 0.f : lse - xr[t];
  if (lane == 0) { loss[row] = l; if (mean_out) atomicAdd(mean_out, l / (float)M); }
  if (grad) for (int c = lane; c < V; c += 64) grad[row * V + c] = ign ? 0.f : (__expf(xr[c] - lse) - (c == (int)t ? 1.f : 0.f));
}
extern "C" int avec_softmax_ce(const float* logits, const long long* targets, long long ignore_index, float* loss, float* mean_out, float* grad, long long M, int V, hipStream_t st) {
  AVEC_CHECK_ARG(logits && targets && loss && M > 0 && V > 0, "softmax_ce: bad arguments");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, logits, targets, ignore_index, loss, mean_out, grad, M, V);
  AVEC_LAUNCH_CHECK(); return 0;
}

// out = g * (*s) * mul    (upstream scalar gradient applied to the saved CTC gradient)
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ g, const float* __restrict__ s, float mul, float* __restrict__ out, long long n) {
  const float f = (s ? *s : 1.f) * mul;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = g[i] * f;
}
extern "C" int avec_scale_by_scalar(const float* g, const float* scalar_dev, float mul, float* out, long long n, hipStream_t st) {
  AVEC_CHECK_ARG(g && out && n > 0, "scale_by_scalar: bad arguments");
  long long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)nb), dim3(256), 0, st, g, scalar_dev, mul, out, n);
  AVEC_LAUNCH_CHECK(); return 0;
}

// the same for up to AVEC_CTC_MAX_HEADS tensors with their own factors in ONE launch (backward of the multi-head CTC loss: blockIdx.y = tensor)
struct ScaleMulti { const float* g[AVEC_CTC_MAX_HEADS]; float* out[AVEC_CTC_MAX_HEADS]; long long n[AVEC_CTC_MAX_HEADS]; float mul[AVEC_CTC_MAX_HEADS]; };
__global__ __launch_bounds__(256) void scale_by_scalar_multi_kernel(ScaleMulti m, const float* __restrict__ s) {
  const int k = blockIdx.y;
  const float f = (s ? *s : 1.f) * m.mul[k];
  const float* __restrict__ g = m.g[k]; float* __restrict__ out = m.out[k]; const long long n = m.n[k];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = g[i] * f;
}
extern "C" int avec_scale_by_scalar_multi(int n_tensors, const float* const* g, float* const* out, const long long* numel, const float* mul, const float* scalar_dev, hipStream_t st) {
  AVEC_CHECK_ARG(n_tensors >= 1 && n_tensors <= AVEC_CTC_MAX_HEADS && g && out && numel && mul, "scale_by_scalar_multi: bad arguments (%d tensors)", n_tensors);
  ScaleMulti m; long long nmax = 0;
  for (int i = 0; i < AVEC_CTC_MAX_HEADS; ++i) {
    const int k = i < n_tensors ? i : 0;
    AVEC_CHECK_ARG(g[k] && out[k] && numel[k] > 0, "scale_by_scalar_multi: null / empty tensor %d", k);
    m.g[i] = g[k]; m.out[i] = out[k]; m.n[i] = numel[k]; m.mul[i] = mul[k];
    if (numel[k] > nmax) nmax = numel[k];
  }
  long long nb = (nmax + 255) / 256; if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(scale_by_scalar_multi_kernel, dim3((unsigned)nb, (unsigned)n_tensors), dim3(256), 0, st, m, scalar_dev);
  AVEC_LAUNCH_CHECK(); return 0;
}

// argmax over the last dim (greedy CTC decoding, nnet/decoders.py:97-120): first maximal index, like torch.argmax
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long long* __restrict__ out, long long M, int V) {
  const int lane = threadIdx.x & 63; const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int v = lane; v < V; v += 64) { const float z = x[row * V + v]; if (z > best || (z == best && v < bi)) { best = z; bi = v; } }
  for (int o = 32; o > 0; o >>= 1) { const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64); if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; } }
  if (lane == 0) out[row] = bi;
}
extern "C" int avec_argmax_rows(const float* x, long long* out, long long M, int V, hipStream_t st) {
  AVEC_CHECK_ARG(x && out && M > 0 && V > 0, "argmax_rows: bad arguments");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, out, M, V);
  AVEC_LAUNCH_CHECK(); return 0;
}

// ---------------------------------------------------------------------------------------------
// Adam over the flat fp32 arenas.  state[0] = step (float, already incremented by the host scheduler mirror), state[1] = lr.
// g <- g*gscale; g += wd*p; m,v updates; p -= lr * mhat / (sqrt(v)/sqrt(bc2) + eps);  optionally zero the gradient arena.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ state,
                                                   float beta1, float beta2, float eps, float wd, float gscale, int zero_grad, long long n4, const int* __restrict__ skip_flag) {
  // skip_flag (optional, device): non-zero = this step's gradients are poisoned (a SyncBatchNorm peer exchange timed out and produced NaN sums):
  // parameters and moments stay untouched, the gradient arena is still cleared so that the next step starts clean
  if (skip_flag && *skip_flag != 0) {
    if (zero_grad) for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) { const float z[4] = {0.f, 0.f, 0.f, 0.f}; st4<float>(g + i * 4, z); }
    return;
  }
  const float step = state[0], lr = state[1];
  const float bc1 = 1.f - powf(beta1, step), bc2s = sqrtf(1.f - powf(beta2, step));
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float pp[4], gg[4], mm[4], vv[4]; ld4<float>(p + i * 4, pp); ld4<float>(g + i * 4, gg); ld4<float>(m + i * 4, mm); ld4<float>(v + i * 4, vv);
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale + wd * pp[e];
      mm[e] = mm[e] + (1.f - beta1) * (gr - mm[e]);            // lerp, as torch's single-tensor Adam
      vv[e] = beta2 * vv[e] + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[e]) / bc2s + eps;
      pp[e] -= step_size * (mm[e] / denom);
      gg[e] = 0.f;
    }
    st4<float>(p + i * 4, pp); st4<float>(m + i * 4, mm); st4<float>(v + i * 4, vv);
    if (zero_grad) st4<float>(g + i * 4, gg);
  }
}
extern "C" int avec_adam_step_guarded(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* state_dev, float beta1, float beta2, float eps,
                                      float weight_decay, float grad_scale, int zero_grad, long long n, const int* skip_flag, hipStream_t st) {
  AVEC_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && state_dev && n > 0 && n % 4 == 0, "adam_step: bad arguments (n=%lld must be a multiple of 4)", n);
  long long n4 = n / 4; long long nb = (n4 + 255) / 256; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, state_dev, beta1, beta2, eps, weight_decay, grad_scale, zero_grad, n4, skip_flag);
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* state_dev, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, int zero_grad, long long n, hipStream_t st) {
  return avec_adam_step_guarded(params, grads, exp_avg, exp_avg_sq, state_dev, beta1, beta2, eps, weight_decay, grad_scale, zero_grad, n, nullptr, st);
}

// ---------------------------------------------------------------------------------------------
// shadow refresh: for every GEMM weight (master fp32, logical [A][Tm][C]) write
//   fwd shadow  (act) = same order                         -> NT forward   (rows = A, K = Tm*C)
//   bwd shadow  (act) [C][Tm][A] (axes 0 and 2 swapped)    -> NT backward-data (rows = C, K = Tm*A)
// table entry (10 x int64): src_off, fwd_off (-1: none), bwd_off (-1: none), A, Tm, C, first_block, n_blocks, C_pad, bwd row pitch (0: Tm*A)
// ---------------------------------------------------------------------------------------------
// One block = one 64 (a) x 64 (c) tile of one tap t: the master is read along c (float4, coalesced) and written to the fwd shadow in the same order (4 elements
// per thread); the tile is transposed through LDS and written to the bwd shadow along a, again 4 elements per thread.  (Round 2: 32 x 32 tiles, one element per
// thread: 2-byte stores, 2.9 TB/s.)  Rows / offsets that are not multiples of 4 elements (the stem's 245-tap rows) take the element-wise path of the same block.
// n_blocks of an entry = Tm * ceil(A/64) * ceil(C/64).  block_base: first block of a partial refresh (table entries [first, first + n) only).
template <typename T>
__global__ __launch_bounds__(256) void shadow_kernel(const float* __restrict__ master, T* __restrict__ shadow, const long long* __restrict__ table, int n_entries, long long block_base) {
  __shared__ float tile[64][65];
  const long long blk = block_base + blockIdx.x;
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (table[mid * 10 + 6] <= blk) lo = mid; else hi = mid - 1; }
  const long long* e = table + lo * 10;
  const long long src = e[0], fwd = e[1], bwd = e[2]; const int A = (int)e[3], Tm = (int)e[4], C = (int)e[5]; const long long Cp = e[8];
  const long long ldb = e[9] > 0 ? e[9] : (long long)Tm * A;       // row pitch of the bwd shadow (> Tm*A: weights fused side by side, e.g. Q|K|V)
  const bool padded = Cp > C && Tm == 1;
  const long long ldf = padded ? Cp : (long long)Tm * C;           // row pitch of the fwd shadow
  const int ta = (A + 63) >> 6, tc = (C + 63) >> 6;
  int b = (int)(blk - e[6]);
  const int ct = b % tc; b /= tc; const int at = b % ta; const int t = b / ta;
  const int x = threadIdx.x & 15, yy = threadIdx.x >> 4;
  const bool vin = ((C | src) & 3) == 0, vfw = vin && ((fwd | ldf) & 3) == 0, vbw = ((A | bwd | ldb) & 3) == 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int a = at * 64 + yy + 16 * r, c = ct * 64 + x * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (a < A && c < C) {
      const float* mp = master + src + ((long long)a * Tm + t) * C + c;
      if (vin) ld4<float>(mp, v);
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (c + q < C) v[q] = mp[q]; }
      if (fwd >= 0) {
        T* fp = shadow + fwd + (long long)a * ldf + (padded ? 0 : (long long)t * C) + c;
        if (vfw) st4<T>(fp, v);
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (c + q < C) stf(fp + q, v[q]); }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[yy + 16 * r][x * 4 + q] = v[q];
  }
  if (bwd < 0) return;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = ct * 64 + yy + 16 * r, a = at * 64 + x * 4;
    if (a >= A || c >= C) continue;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = tile[x * 4 + q][yy + 16 * r];
    T* bp = shadow + bwd + (long long)c * ldb + (long long)t * A + a;
    if (vbw) st4<T>(bp, v);
    else {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (a + q < A) stf(bp + q, v[q]); }
  }
}
extern "C" int avec_shadow_refresh_range(int dtype, const float* master, void* shadow, const long long* table_dev, int n_entries, long long first_block, long long n_blocks, hipStream_t st) {
  AVEC_CHECK_ARG(master && shadow && table_dev && n_entries > 0 && first_block >= 0 && n_blocks > 0, "shadow_refresh: bad arguments");
  DISPATCH_T(dtype, hipLaunchKernelGGL(shadow_kernel<T>, dim3((unsigned)n_blocks), dim3(256), 0, st, master, (T*)shadow, table_dev, n_entries, first_block));
  AVEC_LAUNCH_CHECK(); return 0;
}
extern "C" int avec_shadow_refresh(int dtype, const float* master, void* shadow, const long long* table_dev, int n_entries, long long total_blocks, hipStream_t st) {
  return avec_shadow_refresh_range(dtype, master, shadow, table_dev, n_entries, 0, total_blocks, st);
}
