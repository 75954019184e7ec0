// runtime.hip -- error reporting, ABI version and the per-family hipEvent profiler.
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

ProfScope::ProfScope(int family, hipStream_t stream) : fam(family), s(stream), on(g_prof_on) {
  if (!on) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
  (void)hipEventRecord(e0, s);
}
ProfScope::~ProfScope() {
  if (!on) return;
  (void)hipEventRecord(e1, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_pending.push_back({fam, e0, e1});
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

extern "C" int yt8m_abi_version(void) { return 1; }
extern "C" const char* yt8m_last_error(void) { return yt8m::g_err; }
extern "C" const char* yt8m_built_arch(void) { return "gfx950"; }

extern "C" int yt8m_prof_enable(int on) {
  yt8m::g_prof_on = on != 0;
  return YT8M_OK;
}
extern "C" int yt8m_prof_reset(void) {
  yt8m::drain();
  for (int i = 0; i < yt8m::F_COUNT; ++i) { yt8m::g_launches[i] = 0; yt8m::g_ms[i] = 0.0; }
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
