// Library-level entry points: version, error string, per-family HIP-event profiler.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace mvg {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// CUs the work-distribution planners leave to someone else (the RCCL kernels of a data-parallel run)
static int g_reserved_cus = 0;
int compute_cus() {
  int cus = mvg_device_cus();
  if (cus <= 0) cus = 256;
  cus -= g_reserved_cus;
  return cus > 8 ? cus : 8;
}

// Workspaces the CALLER registered per (device, stream) with mvg_set_scratch: the library never allocates device
// memory.  A kernel sequence that wants scratch and finds none (or too little) takes its scratch-free form
// (plain launches instead of stream-K, a one-level bn_finalize).
struct Scratch {
  int dev;
  hipStream_t st;
  float *ptr;
  size_t floats;
};
static std::mutex g_scratch_mu;
static Scratch g_scratch[32];
float *stream_scratch(hipStream_t st, size_t floats) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  for (auto &e : g_scratch)
    if (e.ptr && e.st == st && e.dev == dev) return e.floats >= floats ? e.ptr : nullptr;
  return nullptr;
}

struct ProfRec {
  int fam;
  hipEvent_t a, b;
};
static bool g_prof = false;
static std::mutex g_mu;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static mvg_prof_entry g_acc[MVG_K_FAMILIES];
static hipEvent_t g_open[MVG_K_FAMILIES];

bool prof_on() { return g_prof; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

void prof_begin(int fam, hipStream_t s, double flops, double bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t a = get_event();
  hipEventRecord(a, s);
  g_open[fam] = a;
  g_acc[fam].launches += 1;
  g_acc[fam].flops += flops;
  g_acc[fam].bytes += bytes;
}
void prof_end(int fam, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t b = get_event();
  hipEventRecord(b, s);
  g_recs.push_back({fam, g_open[fam], b});
}

static const char *kNames[MVG_K_FAMILIES] = {
    "conv_fprop", "conv_dgrad", "conv_wgrad", "wgrad_reduce", "bn_finalize", "bn_apply",
    "bn_bwd_reduce", "bn_bwd_apply", "pool", "layout", "linear_fprop", "linear_dgrad",
    "linear_wgrad", "rotcat", "colsum", "loss", "geometry", "elementwise"};

}  // namespace mvg

extern "C" {

int mvg_abi_version(void) { return MVG_ABI_VERSION; }
const char *mvg_last_error(void) { return mvg::g_err; }

void *mvg_stream_create_low_priority(void) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
    (void)hipGetLastError();
    least = 0;
  }
  hipStream_t st = nullptr;
  if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) {
    (void)hipGetLastError();
    mvg::set_error("hipStreamCreateWithPriority failed");
    return nullptr;
  }
  return (void *)st;
}

int mvg_set_reserved_cus(int n) {
  MVG_REQUIRE(n >= 0 && n < 4096, "reserved CUs must be >= 0");
  mvg::g_reserved_cus = n;
  return 0;
}

int mvg_set_scratch(void *ptr, size_t bytes, void *stream) {
  std::lock_guard<std::mutex> lk(mvg::g_scratch_mu);
  int dev = 0;
  MVG_REQUIRE(hipGetDevice(&dev) == hipSuccess, "set_scratch: no current device");
  MVG_REQUIRE(((uintptr_t)ptr & 15) == 0, "set_scratch: the workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  mvg::Scratch *slot = nullptr;
  for (auto &e : mvg::g_scratch)
    if (e.ptr && e.st == st && e.dev == dev) slot = &e;
  if (!slot && ptr)
    for (auto &e : mvg::g_scratch)
      if (!e.ptr) {
        slot = &e;
        break;
      }
  if (!ptr) {                           // unregister
    if (slot) slot->ptr = nullptr;
    return 0;
  }
  MVG_REQUIRE(slot != nullptr, "set_scratch: more than 32 (device, stream) workspaces registered");
  slot->dev = dev;
  slot->st = st;
  slot->ptr = (float *)ptr;
  slot->floats = bytes / sizeof(float);
  return 0;
}

size_t mvg_scratch_bytes(void) {
  // the largest user: stream-K pieces of the fp32-MFMA kernels, 2 slots of a 128 x 128 fp32 tile per persistent
  // workgroup, at most four workgroups per CU
  int cus = mvg_device_cus();
  if (cus <= 0) cus = 256;
  return (size_t)cus * 4 * 2 * 128 * 128 * sizeof(float);
}

int mvg_device_cus(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return cus;
}

int mvg_prof_enable(int on) {
  mvg::g_prof = on != 0;
  return 0;
}
int mvg_prof_reset(void) {
  std::lock_guard<std::mutex> lk(mvg::g_mu);
  for (auto &r : mvg::g_recs) {
    mvg::g_pool.push_back(r.a);
    mvg::g_pool.push_back(r.b);
  }
  mvg::g_recs.clear();
  memset(mvg::g_acc, 0, sizeof(mvg::g_acc));
  return 0;
}
int mvg_prof_collect(mvg_prof_entry *out) {
  std::lock_guard<std::mutex> lk(mvg::g_mu);
  for (auto &r : mvg::g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) {
      mvg::set_error("prof_collect: event sync failed");
      return 1;
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    mvg::g_acc[r.fam].ms += ms;
    mvg::g_pool.push_back(r.a);
    mvg::g_pool.push_back(r.b);
  }
  mvg::g_recs.clear();
  memcpy(out, mvg::g_acc, sizeof(mvg::g_acc));
  return 0;
}
const char *mvg_prof_family_name(int f) { return (f >= 0 && f < MVG_K_FAMILIES) ? mvg::kNames[f] : "?"; }

}  // extern "C"
