// One-shot all-reduce(sum) of a short fp32 vector between the GPUs of one node by direct peer writes over xGMI (SyncBatchNorm statistics:
// nnet/normalizations.py:172-249 exchanges (2C+1)-float vectors 45 times forward and 45 times backward per step; through RCCL each costs a
// latency-bound collective of 20-50 us plus host work, and keeps the step out of a hipGraph).
//
// Every rank owns an exchange buffer that all ranks map (HIP IPC).  For exchange site s and epoch e (the e-th visit of that site):
//   page(s, e & 1) = [world][n] granules, granule = 8 bytes {payload bits, epoch tag} written by ONE 64-bit system-scope store,
// rank r writes its vector into slot r of that page in EVERY rank's buffer, then polls its own page until all `world` slots carry tag e
// and adds them in rank order (same order on every rank: bit-identical results, as an all-reduce must give).  A granule needs no fence or
// flag: payload and tag arrive together (MI355X_MICROARCH.md, hand-off rows).  Two pages per site: a rank cannot write epoch e+2 before every
// rank wrote e+1, which every rank does only after it finished reading e.  No host involvement: capturable into a hipGraph (the epoch
// counter lives in device memory).  A poll that lasts longer than `timeout_ms` raises the error flag instead of hanging the GPU.
#include "common.h"
#include "avec_hip.h"
#include <string.h>

struct PeerArgs {
  const float* in; float* out; int n;
  unsigned long long* pages[AVEC_PEER_MAX_WORLD];   // base of this site's two pages in every rank's buffer (index = rank)
  long long page_stride;                            // granules between the two parity pages
  int rank, world;
  unsigned* epoch;                                  // this site's visit counter (device memory, this rank)
  int* err;
  long long timeout_spins;                          // polls (each followed by s_sleep 8, >= ~0.25 us) before a slot is given up
  // fused producer work (round 4: one launch instead of two per SyncBatchNorm exchange): the exchanged vector is  v[i] = sum_{r < nrep} in[r * n_in + i]  for i < n_in
  // (the statistic replicas of the producing kernel), v[n_in] = tail when has_tail (the local element count);  and, before the exchange, the LOCAL sums are added to the
  // affine gradients: dbeta[c] += v[c], dgamma[c] += v[C + c]  (backward exchange: torch's SyncBatchNorm all-reduces the statistics, not the parameter gradients)
  int nrep, n_in, has_tail; float tail;
  float* dgamma; float* dbeta; int C;
};

__global__ __launch_bounds__(256) void peer_exchange_sum_kernel(PeerArgs a) {
  __shared__ unsigned s_epoch;
  if (threadIdx.x == 0) { s_epoch = *a.epoch + 1u; *a.epoch = s_epoch; }
  __syncthreads();
  const unsigned e = s_epoch;
  const long long poff = (long long)(e & 1u) * a.page_stride;
  // publish: slot `rank` of the page in every rank's buffer (own buffer included)
  for (int i = threadIdx.x; i < a.n; i += 256) {
    float v;
    if (a.has_tail && i == a.n_in) v = a.tail;
    else { v = 0.f; for (int r = 0; r < a.nrep; ++r) v += a.in[(long long)r * a.n_in + i]; }
    if (a.dgamma && i < 2 * a.C) { if (i < a.C) a.dbeta[i] += v; else a.dgamma[i - a.C] += v; }
    const unsigned long long gr = ((unsigned long long)e << 32) | (unsigned long long)__float_as_uint(v);
    for (int r = 0; r < a.world; ++r)
      __hip_atomic_store(a.pages[r] + poff + (long long)a.rank * a.n + i, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // gather: own page, all slots, in rank order
  const unsigned long long* mine = a.pages[a.rank] + poff;
  for (int i = threadIdx.x; i < a.n; i += 256) {
    float s = 0.f;
    bool lost = false;
    for (int r = 0; r < a.world; ++r) {
      unsigned long long gr;
      long long spins = 0;
      while (true) {
        gr = __hip_atomic_load(mine + (long long)r * a.n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(gr >> 32) == e) break;
        __builtin_amdgcn_s_sleep(8);
        if (++spins > a.timeout_spins) { *a.err = 1; lost = true; break; }      // counted in polls, not by the cycle counter: s_memtime is not monotonic for a wave that was
                                                                  // context-switched out and resumed elsewhere (two processes sharing a GPU), and a spurious
                                                                  // time-out here means a wrong sum
      }
      s += __uint_as_float((unsigned)gr);
    }
    a.out[i] = lost ? __uint_as_float(0x7fc00000u) : s;        // a missing peer must not look like a sum: NaN propagates into the loss
  }
}

static int peer_exchange_impl(const float* in, int nrep, int n_in, int has_tail, float tail, float* dgamma, float* dbeta, int C, float* out, void* const* pages,
                              long long page_stride_granules, int rank, int world, unsigned* epoch, int* err_flag, int timeout_ms, hipStream_t stream) {
  const int n = n_in + (has_tail ? 1 : 0);
  AVEC_CHECK_ARG(in && out && pages && epoch && err_flag && n_in > 0 && nrep >= 1 && world >= 1 && world <= AVEC_PEER_MAX_WORLD && rank >= 0 && rank < world,
                 "peer_exchange_sum: bad arguments (n=%d rank=%d world=%d)", n, rank, world);
  AVEC_CHECK_ARG(!dgamma == !dbeta && (!dgamma || (C > 0 && 2 * C <= n_in)), "peer_exchange_sum: affine-gradient outputs need both pointers and 2 C <= n");
  PeerArgs a; a.in = in; a.out = out; a.n = n; a.nrep = nrep; a.n_in = n_in; a.has_tail = has_tail; a.tail = tail; a.dgamma = dgamma; a.dbeta = dbeta; a.C = C; a.page_stride = page_stride_granules; a.rank = rank; a.world = world; a.epoch = epoch; a.err = err_flag; a.timeout_spins = (long long)(timeout_ms > 0 ? timeout_ms : 20000) * 4000ll;
  for (int r = 0; r < AVEC_PEER_MAX_WORLD; ++r) a.pages[r] = r < world ? (unsigned long long*)pages[r] : nullptr;
  for (int r = 0; r < world; ++r) AVEC_CHECK_ARG(a.pages[r] && (((uintptr_t)a.pages[r]) & 7) == 0, "peer_exchange_sum: page pointer of rank %d is null / not 8-byte aligned", r);
  hipLaunchKernelGGL(peer_exchange_sum_kernel, dim3(1), dim3(256), 0, stream, a);
  AVEC_LAUNCH_CHECK();
  return 0;
}

extern "C" int avec_peer_exchange_sum(const float* in, float* out, int n, void* const* pages, long long page_stride_granules, int rank, int world,
                                      unsigned* epoch, int* err_flag, int timeout_ms, hipStream_t stream) {
  return peer_exchange_impl(in, 1, n, 0, 0.f, nullptr, nullptr, 0, out, pages, page_stride_granules, rank, world, epoch, err_flag, timeout_ms, stream);
}
extern "C" int avec_peer_exchange_sum_fused(const float* in, int n_replicas, int n_in, int has_tail, float tail, float* dgamma, float* dbeta, int C, float* out,
                                            void* const* pages, long long page_stride_granules, int rank, int world, unsigned* epoch, int* err_flag, int timeout_ms,
                                            hipStream_t stream) {
  return peer_exchange_impl(in, n_replicas, n_in, has_tail, tail, dgamma, dbeta, C, out, pages, page_stride_granules, rank, world, epoch, err_flag, timeout_ms, stream);
}

// hipDeviceEnablePeerAccess for the current device (kernels here dereference memory of the peer GPUs that was mapped through HIP IPC)
extern "C" int avec_enable_peer_access(int peer_device) {
  hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
  if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return 0; }
  if (e != hipSuccess) { avec_set_error("hipDeviceEnablePeerAccess(%d) failed: %s", peer_device, hipGetErrorString(e)); (void)hipGetLastError(); return (int)e; }
  return 0;
}

// ---- the exchange buffer itself ------------------------------------------------------------------------------------------------------------
// The one exception to "the library never allocates": the buffer must be UNCACHED / fine-grained device memory (stores from a peer GPU have to be
// visible to a kernel that is already running here; ordinary hipMalloc memory is only coherent at kernel boundaries) and exportable through HIP IPC,
// which a framework allocator does not offer.  The caller owns the returned pointers (free / close them with the functions below).
extern "C" int avec_peer_buffer_alloc(void** ptr, long long bytes, void* ipc_handle_64b) {
  AVEC_CHECK_ARG(ptr && bytes > 0 && ipc_handle_64b, "peer_buffer_alloc: bad arguments");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) { avec_set_error("peer_buffer_alloc: hipExtMallocWithFlags failed: %s", hipGetErrorString(e)); (void)hipGetLastError(); return (int)e; }
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle((hipIpcMemHandle_t*)ipc_handle_64b, p);
  if (e != hipSuccess) { avec_set_error("peer_buffer_alloc: memset / hipIpcGetMemHandle failed: %s", hipGetErrorString(e)); (void)hipGetLastError(); (void)hipFree(p); return (int)e; }
  *ptr = p;
  return 0;
}
extern "C" int avec_peer_buffer_open(const void* ipc_handle_64b, void** ptr) {
  AVEC_CHECK_ARG(ptr && ipc_handle_64b, "peer_buffer_open: bad arguments");
  hipIpcMemHandle_t h; memcpy(&h, ipc_handle_64b, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { avec_set_error("peer_buffer_open: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e)); (void)hipGetLastError(); return (int)e; }
  return 0;
}
extern "C" int avec_peer_buffer_close(void* ptr) { hipError_t e = hipIpcCloseMemHandle(ptr); (void)hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }
extern "C" int avec_peer_buffer_free(void* ptr) { hipError_t e = hipFree(ptr); (void)hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }
