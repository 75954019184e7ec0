// runtime.hip -- error reporting, ABI version and the per-family hipEvent profiler.
#include <stdlib.h>
#include <functional>
#include <mutex>
#include <vector>
#include "common.h"

namespace yt8m {
thread_local char g_err[512] = "";

static bool g_prof_on = false;
static std::mutex g_prof_mu;
struct Pending { int fam; hipEvent_t e0, e1; };
static std::vector<Pending> g_pending;
static int64_t g_launches[F_COUNT];
static double g_ms[F_COUNT];
static double g_flops[F_COUNT];
static double g_bytes[F_COUNT];

ProfScope::ProfScope(int family, hipStream_t stream, double algorithmic_flops, double algorithmic_bytes)
    : fam(family), s(stream), on(g_prof_on), flops(algorithmic_flops), bytes(algorithmic_bytes) {
  if (!on) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
  (void)hipEventRecord(e0, s);
}
ProfScope::~ProfScope() {
  if (!on) return;
  (void)hipEventRecord(e1, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_flops[fam] += flops;
  g_bytes[fam] += bytes;
  g_pending.push_back({fam, e0, e1});
}

bool prof_enabled() { return g_prof_on; }

// ---- hipGraph replay of launch-bound kernel chains (the LSTM recurrence: 75-300 steps x 1-2 tiny kernels) ------------
// A chain is captured once per distinct argument tuple (pointers + shapes) on the caller's stream, instantiated and kept in a
// small LRU cache; later calls with the same arguments replay it with ONE hipGraphLaunch.  PyTorch's caching allocator
// hands the same addresses to the same allocation sequence every training step, so steady-state steps hit the cache.
// Off while the hipEvent profiler is on (its event records would be captured) or when YT8M_NO_GRAPH is set.
struct GraphEntry {
  GraphKey key;
  hipGraphExec_t exec;
  uint64_t last;
};
static std::mutex g_graph_mu;
static std::vector<GraphEntry> g_graphs;
static uint64_t g_graph_tick = 0;
static int64_t g_graph_hits = 0, g_graph_captures = 0, g_graph_fallbacks = 0;
static const size_t GRAPH_CACHE = 96;

static bool graphs_disabled() {
  static const bool off = getenv("YT8M_NO_GRAPH") != nullptr;
  return off;
}

int run_chain(const GraphKey& key, hipStream_t s, const std::function<int()>& launch_all) {
  if (graphs_disabled() || g_prof_on) return launch_all();
  {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (auto& e : g_graphs) {
      if (memcmp(&e.key, &key, sizeof(GraphKey)) == 0) {
        e.last = ++g_graph_tick;
        ++g_graph_hits;
        if (hipGraphLaunch(e.exec, s) == hipSuccess) return YT8M_OK;
        (void)hipGetLastError();
        break;                                                       // stale exec: fall through to a direct launch
      }
    }
  }
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    ++g_graph_fallbacks;
    return launch_all();
  }
  const int rc = launch_all();
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(s, &graph);
  if (rc != YT8M_OK || ec != hipSuccess || !graph) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    ++g_graph_fallbacks;
    return rc != YT8M_OK ? rc : launch_all();                        // nothing ran during the failed capture
  }
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) {
    (void)hipGetLastError();
    (void)hipGraphDestroy(graph);
    ++g_graph_fallbacks;
    return launch_all();
  }
  (void)hipGraphDestroy(graph);
  if (hipGraphLaunch(exec, s) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipGraphExecDestroy(exec);
    ++g_graph_fallbacks;
    return launch_all();
  }
  std::lock_guard<std::mutex> lk(g_graph_mu);
  ++g_graph_captures;
  if (g_graphs.size() >= GRAPH_CACHE) {                              // evict the least recently used entry
    size_t victim = 0;
    for (size_t i = 1; i < g_graphs.size(); ++i)
      if (g_graphs[i].last < g_graphs[victim].last) victim = i;
    // an evicted executable may still be running: synchronise its stream's work before destroying it
    (void)hipDeviceSynchronize();
    (void)hipGraphExecDestroy(g_graphs[victim].exec);
    g_graphs.erase(g_graphs.begin() + victim);
  }
  g_graphs.push_back({key, exec, ++g_graph_tick});
  return YT8M_OK;
}

static void drain() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& p : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      g_launches[p.fam] += 1;
      g_ms[p.fam] += ms;
    }
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  g_pending.clear();
}
}  // namespace yt8m

// ABI 2 (round 3 changes): the persistent-recurrence workspace starts with a sticky error word the host zeroes once
// (control block 128 B further in, yt8m_lstm_persist_workspace_bytes larger), yt8m_lstm_persist_status clears the word it reads,
// yt8m_gemm_problem.lda / ldb of the image products are K-block strides (0 = dense).  Round 4 adds entry points only.
extern "C" int yt8m_abi_version(void) { return 4; }
extern "C" const char* yt8m_last_error(void) { return yt8m::g_err; }
extern "C" const char* yt8m_built_arch(void) { return "gfx950"; }

extern "C" int yt8m_prof_enable(int on) {
  yt8m::g_prof_on = on != 0;
  return YT8M_OK;
}
extern "C" int yt8m_graph_cache_stats(int64_t* hits, int64_t* captures, int64_t* fallbacks, int64_t* entries) {
  std::lock_guard<std::mutex> lk(yt8m::g_graph_mu);
  if (hits) *hits = yt8m::g_graph_hits;
  if (captures) *captures = yt8m::g_graph_captures;
  if (fallbacks) *fallbacks = yt8m::g_graph_fallbacks;
  if (entries) *entries = (int64_t)yt8m::g_graphs.size();
  return YT8M_OK;
}
extern "C" int yt8m_graph_cache_clear(void) {
  std::lock_guard<std::mutex> lk(yt8m::g_graph_mu);
  (void)hipDeviceSynchronize();
  for (auto& e : yt8m::g_graphs) (void)hipGraphExecDestroy(e.exec);
  yt8m::g_graphs.clear();
  return YT8M_OK;
}
extern "C" int yt8m_prof_reset(void) {
  yt8m::drain();
  for (int i = 0; i < yt8m::F_COUNT; ++i) { yt8m::g_launches[i] = 0; yt8m::g_ms[i] = 0.0; yt8m::g_flops[i] = 0.0; yt8m::g_bytes[i] = 0.0; }
  return YT8M_OK;
}
extern "C" int yt8m_prof_get_flops(int family, double* flops) {
  using namespace yt8m;
  YT8M_REQUIRE(family >= 0 && family < F_COUNT && flops, YT8M_E_BADARG, "bad family / null out");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  *flops = g_flops[family];
  return YT8M_OK;
}
extern "C" int yt8m_prof_get_bytes(int family, double* bytes) {
  using namespace yt8m;
  YT8M_REQUIRE(family >= 0 && family < F_COUNT && bytes, YT8M_E_BADARG, "bad family / null out");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  *bytes = g_bytes[family];
  return YT8M_OK;
}
extern "C" int yt8m_prof_get(int family, int64_t* launches, double* total_ms) {
  using namespace yt8m;
  YT8M_REQUIRE(family >= 0 && family < F_COUNT && launches && total_ms, YT8M_E_BADARG, "bad family / null out");
  drain();
  *launches = g_launches[family];
  *total_ms = g_ms[family];
  return YT8M_OK;
}
