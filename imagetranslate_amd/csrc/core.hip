// Library-level helpers: version, thread-local error string, optional per-launch HIP-event profiler.
#include <stdarg.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "common.hpp"

static thread_local char g_err[512] = "";

void imt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int imt_version(void) { return 100; }
extern "C" const char* imt_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ profiler
// Off by default (zero overhead: one relaxed flag test per launch site).  When enabled by the bench harness,
// every kernel launch site brackets its launch with two hipEvents recorded ON THE LAUNCH STREAM; the report
// gives, per kernel kind, launch count, summed device time and the algorithmic FLOPs / bytes the callers declared.
namespace {
struct Rec { const char* kind; double flops, bytes; hipEvent_t e0, e1; };
std::mutex g_mu;
bool g_prof_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
hipEvent_t take_event() {
  if (g_pool_next == g_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_pool.push_back(e);
  }
  return g_pool[g_pool_next++];
}
}  // namespace

bool imt_prof_enabled() { return g_prof_on; }

// "kind MxNxK" strings with process lifetime (profiling only)
const char* imt_prof_intern(const char* kind, int M, int N, int K) {
  static std::map<std::string, std::string*> table;
  char buf[96];
  snprintf(buf, sizeof(buf), "%s %dx%dx%d", kind, M, N, K);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = table.find(buf);
  if (it == table.end()) it = table.emplace(buf, new std::string(buf)).first;
  return it->second->c_str();
}

void* imt_prof_begin_launch(const char* kind, double flops, double bytes, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_prof_on) return nullptr;
  Rec r{kind, flops, bytes, take_event(), take_event()};
  if (!r.e0 || !r.e1) return nullptr;
  (void)hipEventRecord(r.e0, st);
  g_recs.push_back(r);
  return reinterpret_cast<void*>(g_recs.size());  // 1-based index
}
void imt_prof_end_launch(void* tok, hipStream_t st) {
  if (!tok) return;
  std::lock_guard<std::mutex> lk(g_mu);
  const size_t i = reinterpret_cast<size_t>(tok) - 1;
  if (i < g_recs.size()) (void)hipEventRecord(g_recs[i].e1, st);
}

extern "C" int imt_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_on = (on != 0);
  if (on) { g_recs.clear(); g_pool_next = 0; }
  return IMT_OK;
}

// Synchronises the recorded events (host-side wait) and writes up to max_kinds rows; returns the number of kinds.
extern "C" int imt_prof_report(imt_prof_row* rows, int max_kinds) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, imt_prof_row> agg;
  for (const Rec& r : g_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    imt_prof_row& a = agg[r.kind];
    if (a.launches == 0) { memset(&a, 0, sizeof(a)); strncpy(a.kind, r.kind, sizeof(a.kind) - 1); }
    a.launches += 1; a.total_ms += ms; a.flops += r.flops; a.bytes += r.bytes;
  }
  int n = 0;
  for (auto& kv : agg) {
    if (n >= max_kinds) break;
    rows[n++] = kv.second;
  }
  g_recs.clear();
  g_pool_next = 0;
  return n;
}

// ------------------------------------------------------------------------------------------------ test utility
// Occupies `blocks` workgroups (threads, lds_bytes each) for ~`cycles` shader clocks: stands in for a long-running
// communication kernel when measuring how the step's kernels behave with fewer free CUs (tools/comm_pressure.py).
namespace {
__global__ void spin_kernel(long long cycles, int lds_bytes) {
  extern __shared__ char spin_lds[];
  if (lds_bytes > 0 && threadIdx.x == 0) spin_lds[0] = 1;
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

extern "C" int imt_debug_spin(int blocks, int threads, int lds_bytes, int64_t cycles, void* stream) {
  IMT_CHECK_ARG(blocks > 0 && threads > 0 && threads <= 1024 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && cycles >= 0, "debug_spin: bad args");
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, (long long)cycles, lds_bytes);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_abi_sizeof(const char* name) {
  if (!name) return -1;
#define IMT_SZ(T) if (strcmp(name, #T) == 0) return (int)sizeof(T)
  IMT_SZ(imt_gemm_args); IMT_SZ(imt_attn_args); IMT_SZ(imt_prof_row); IMT_SZ(imt_attn_block); IMT_SZ(imt_layer_desc);
  IMT_SZ(imt_stack_desc); IMT_SZ(imt_stack_io); IMT_SZ(imt_attn_decode_args); IMT_SZ(imt_decode_io); IMT_SZ(imt_beam_args);
  IMT_SZ(imt_mass_args);
#undef IMT_SZ
  return -1;
}
