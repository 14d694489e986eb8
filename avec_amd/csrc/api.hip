// Error plumbing + version for libavec_hip.so
#include "common.h"
#include "avec_hip.h"
#include "vec.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void avec_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* avec_last_error() { return g_err; }

// name of the kernel instance the last GEMM-family entry point launched on this thread (bench.py's per-kernel roofline rows use it as the key)
static thread_local char g_kname[128] = "";
void avec_note_kernel(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_kname, sizeof(g_kname), fmt, ap); va_end(ap);
}
extern "C" const char* avec_last_kernel() { return g_kname; }
extern "C" int avec_version() { return AVEC_ABI_VERSION; }
extern "C" int avec_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(avec_rows_t); case 1: return (int)sizeof(avec_epilogue_t); case 2: return (int)sizeof(avec_attn_t); case 3: return (int)sizeof(avec_tn_item_t);
    case 4: return (int)sizeof(avec_tn_batched_t); case 5: return (int)sizeof(avec_ln_item_t); case 6: return (int)sizeof(avec_fp8_item_t); case 7: return (int)sizeof(avec_wgrad3x3_item_t);
    default: return -1;
  }
}

// ---------------------------------------------------------------------------------------------
// reduction workspace (vec.h: two-pass column reductions).  One registration per device.
// ---------------------------------------------------------------------------------------------
static constexpr int WS_MAX_DEV = 64, WS_MAX_STREAMS = 4;
static struct { void* base; size_t bytes; } g_ws[WS_MAX_DEV];
static struct { hipStream_t st; int dev; void* base; size_t bytes; } g_ws_stream[WS_MAX_STREAMS];      // extra workspaces bound to specific streams
static int g_n_ws_stream = 0;

extern "C" int avec_set_reduce_workspace(void* base, long long bytes) {
  int dev = 0; hipError_t e = hipGetDevice(&dev);
  AVEC_CHECK_ARG(e == hipSuccess && dev >= 0 && dev < WS_MAX_DEV, "set_reduce_workspace: no current device");
  AVEC_CHECK_ARG((base == nullptr && bytes == 0) || (base != nullptr && bytes >= (1 << 16) && ((size_t)base & 255) == 0),
                 "set_reduce_workspace: need a 256-byte aligned buffer of at least 64 KB (or NULL, 0 to unregister)");
  g_ws[dev].base = base; g_ws[dev].bytes = (size_t)bytes;
  return 0;
}
extern "C" int avec_set_reduce_workspace_stream(void* base, long long bytes, hipStream_t stream) {
  int dev = 0; hipError_t e = hipGetDevice(&dev);
  AVEC_CHECK_ARG(e == hipSuccess && base != nullptr && bytes >= (1 << 16) && ((size_t)base & 255) == 0, "set_reduce_workspace_stream: bad arguments");
  for (int i = 0; i < g_n_ws_stream; ++i) if (g_ws_stream[i].st == stream && g_ws_stream[i].dev == dev) { g_ws_stream[i].base = base; g_ws_stream[i].bytes = (size_t)bytes; return 0; }
  AVEC_CHECK_ARG(g_n_ws_stream < WS_MAX_STREAMS, "set_reduce_workspace_stream: at most %d stream-bound workspaces", WS_MAX_STREAMS);
  g_ws_stream[g_n_ws_stream].st = stream; g_ws_stream[g_n_ws_stream].dev = dev; g_ws_stream[g_n_ws_stream].base = base; g_ws_stream[g_n_ws_stream].bytes = (size_t)bytes; ++g_n_ws_stream;
  return 0;
}
ColWs avec_reduce_ws(size_t partial_floats, hipStream_t st) {
  ColWs ws{nullptr};
  static const bool off = false;
  if (off) return ws;
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WS_MAX_DEV) return ws;
  for (int i = 0; i < g_n_ws_stream; ++i) if (g_ws_stream[i].st == st && g_ws_stream[i].dev == dev) {
    if (partial_floats * sizeof(float) <= g_ws_stream[i].bytes) ws.partial = (float*)g_ws_stream[i].base;
    return ws;
  }
  if (!g_ws[dev].base || partial_floats * sizeof(float) > g_ws[dev].bytes) return ws;
  ws.partial = (float*)g_ws[dev].base;
  return ws;
}
