// Library-level entry points of lib3dvnet_hip.so (see include/v3d.h).
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "v3d_common.h"

extern "C" int v3d_version(void) { return 6; }

extern "C" const char* v3d_last_error(void) { return v3d::err_buf(); }

namespace {
struct Span { std::string name; hipEvent_t a, b; };
std::mutex g_mu;
bool g_on = false;
std::vector<Span> g_spans;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace

namespace v3d {
bool timing_enabled() { return g_on; }
void timing_begin(const char* name, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  Span sp{name, take_event(), take_event()};
  (void)hipEventRecord(sp.a, s);
  g_spans.push_back(sp);
}
void timing_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_spans.empty()) (void)hipEventRecord(g_spans.back().b, s);
}
}  // namespace v3d

namespace {
struct OptDef { const char* name; int def; };
const OptDef kOptDefs[v3d::kOptCount] = {{"psv_kernel", 0}, {"psv_threads", 64}, {"c12_march", 1}, {"c12_nseg", 0}, {"c9_kernel", 0},
                                         {"conv_vec", 1}, {"stop_after", 99}, {"gemm_rounds", 1}, {"gemm_round_rows", 0},
                                         {"gemm_pipe", 1}, {"tail_streams", 1}, {"tail_from", 3}, {"tail_to", 8}, {"prop_fused", 1}};
std::atomic<int> g_opt[v3d::kOptCount];
std::atomic<bool> g_opt_init{false};
void opt_init() {
  if (!g_opt_init.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_opt_init.load()) {
      for (int i = 0; i < v3d::kOptCount; ++i) g_opt[i].store(kOptDefs[i].def);
      g_opt_init.store(true, std::memory_order_release);
    }
  }
}
}  // namespace

int v3d::option(v3d::Option o) {
  opt_init();
  return g_opt[o].load(std::memory_order_relaxed);
}

extern "C" int v3d_set_option(const char* name, int value) {
  V3D_REQUIRE(name, V3D_ERR_BAD_ARG, "v3d_set_option: null name");
  opt_init();
  for (int i = 0; i < v3d::kOptCount; ++i)
    if (!strcmp(name, kOptDefs[i].name)) {
#ifndef V3D_EXPERIMENTS
      V3D_REQUIRE(!(i == v3d::kOptC9Kernel && value == 1), V3D_ERR_UNSUPPORTED,
                  "v3d_set_option: c9_kernel = 1 (csrc/conv9z.hip) needs a library built with -DV3D_EXPERIMENTS");
#endif
      g_opt[i].store(value);
      return V3D_OK;
    }
  return v3d::fail(V3D_ERR_BAD_ARG, "v3d_set_option: unknown option '%s'", name);
}

extern "C" int v3d_get_option(const char* name, int* value) {
  V3D_REQUIRE(name && value, V3D_ERR_BAD_ARG, "v3d_get_option: null argument");
  opt_init();
  for (int i = 0; i < v3d::kOptCount; ++i)
    if (!strcmp(name, kOptDefs[i].name)) { *value = g_opt[i].load(); return V3D_OK; }
  return v3d::fail(V3D_ERR_BAD_ARG, "v3d_get_option: unknown option '%s'", name);
}

extern "C" int v3d_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return V3D_OK;
}

// Synchronises the recorded events, aggregates by kernel name, clears the log.  Returns the number
// of distinct kernels (<= max_entries written).
extern "C" int v3d_timing_collect(int max_entries, char* names, int name_stride, float* total_ms,
                                  int* launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, std::pair<float, int>> agg;
  std::vector<std::string> order;
  for (auto& sp : g_spans) {
    float ms = 0.f;
    if (hipEventSynchronize(sp.b) == hipSuccess) (void)hipEventElapsedTime(&ms, sp.a, sp.b);
    if (!agg.count(sp.name)) order.push_back(sp.name);
    agg[sp.name].first += ms;
    agg[sp.name].second += 1;
    g_pool.push_back(sp.a);
    g_pool.push_back(sp.b);
  }
  g_spans.clear();
  int n = 0;
  for (auto& k : order) {
    if (n >= max_entries) break;
    if (names && name_stride > 0) {
      strncpy(names + (size_t)n * name_stride, k.c_str(), name_stride - 1);
      names[(size_t)n * name_stride + name_stride - 1] = 0;
    }
    if (total_ms) total_ms[n] = agg[k].first;
    if (launches) launches[n] = agg[k].second;
    ++n;
  }
  return n;
}
