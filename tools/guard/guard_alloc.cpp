// Guard-page allocator for PyTorch (torch.cuda.memory.CUDAPluggableAllocator): test infrastructure, not part of the product.
// Every allocation gets its own virtual range with an UNMAPPED granule on both sides (hipMemAddressReserve / hipMemCreate / hipMemMap); the block sits flush against the end
// (GUARD_MODE=tail, default) or the start (GUARD_MODE=head) of the mapped part, so a kernel that reads or writes past that side of a tensor takes a memory fault instead of
// silently touching a neighbour.  tests/test_gpu_bounds.py runs a whole training pass on it.  ONLY THE FAULT IS EVIDENCE in this mode: on hipMemCreate memory, on this ROCm
// stack, plain stores of any kernel (torch's own included) are lost run to run -- losses differ between identical runs -- so values computed here mean nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

struct Rec { char* va; size_t va_size; size_t map_size; hipMemGenericAllocationHandle_t h; };
static std::map<void*, Rec> g_live;
static std::mutex g_mu;
static size_t g_gran = 0, g_gap = 0;      // g_gap: unmapped bytes on each side of a block (GUARD_GAP_MB, default one granule)
static long long g_count = 0, g_bytes = 0;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "guard_alloc: %s failed: %s\n", #x, hipGetErrorString(e_)); abort(); } } while (0)

static int plain_mode() { static int m = -1; if (m < 0) { const char* e = getenv("GUARD_MODE"); m = (e && !strcmp(e, "plain")) ? 1 : 0; } return m; }
// GUARD_MODE=plain: no guards, one hipMalloc per tensor, optionally filled with the byte GUARD_FILL (255: NaN) -- no block is ever handed out twice with its old contents, which
// tells a result that depends on memory nobody wrote (varies / turns NaN here, repeats bit for bit on the caching allocator) from a property of hipMemCreate memory
extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t) {
  if (size <= 0) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  CK(hipSetDevice(device));
  if (plain_mode()) {
    void* p = nullptr; CK(hipMalloc(&p, (size_t)size));
    if (const char* f = getenv("GUARD_FILL")) { CK(hipMemset(p, atoi(f), (size_t)size)); CK(hipDeviceSynchronize()); }
    ++g_count; g_bytes += (long long)size;
    return p;
  }
  hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
  if (!g_gran) {
    CK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
    const char* gm = getenv("GUARD_GAP_MB"); g_gap = gm ? (size_t)atoi(gm) << 20 : g_gran; g_gap = (g_gap + g_gran - 1) / g_gran * g_gran;
  }
  const size_t msz = ((size_t)size + g_gran - 1) / g_gran * g_gran, vsz = msz + 2 * g_gap;
  Rec r; void* va = nullptr;
  CK(hipMemAddressReserve(&va, vsz, g_gran, nullptr, 0));
  r.va = (char*)va; r.va_size = vsz; r.map_size = msz;
  CK(hipMemCreate(&r.h, msz, &prop, 0));
  CK(hipMemMap(r.va + g_gap, msz, 0, r.h, 0));
  hipMemAccessDesc d; memset(&d, 0, sizeof(d)); d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(r.va + g_gap, msz, &d, 1));
  const char* mode = getenv("GUARD_MODE");
  const size_t align = (size_t)size >= (1u << 20) ? 256 : 16;          // (the library asks for 256-byte aligned workspaces; everything else needs 16)
  const size_t sz_al = ((size_t)size + align - 1) / align * align;
  char* p = (mode && !strcmp(mode, "head")) ? r.va + g_gap : r.va + g_gap + msz - sz_al;
  g_live[p] = r; ++g_count; g_bytes += (long long)msz;
  return p;
}
extern "C" void guard_free(void* ptr, ssize_t, int device, hipStream_t) {
  if (!ptr) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (plain_mode()) { CK(hipSetDevice(device)); CK(hipDeviceSynchronize()); CK(hipFree(ptr)); return; }
  auto it = g_live.find(ptr);
  if (it == g_live.end()) { fprintf(stderr, "guard_alloc: free of unknown pointer %p\n", ptr); abort(); }
  CK(hipSetDevice(device));
  CK(hipDeviceSynchronize());                     // the tensor died on the host; work that uses it may still be queued on any stream
  Rec r = it->second; g_live.erase(it);
  CK(hipMemUnmap(r.va + g_gap, r.map_size));
  CK(hipMemRelease(r.h));
  CK(hipMemAddressFree(r.va, r.va_size));
}
extern "C" long long guard_stats(int what) { return what == 0 ? g_count : what == 1 ? g_bytes : (long long)g_gran; }
