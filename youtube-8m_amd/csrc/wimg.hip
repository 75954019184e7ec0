// wimg.hip -- registry of RESIDENT operand images of the weight matrices (round 5; VERDICT r4 #3a).
//
// Every large fp32 product of the step runs on the bf16 matrix pipe from "operand images" of its two matrices (csrc/gemm_x3.hip:
// three bf16 planes in LDS-DMA block order; one plane in --compute_dtype=bfloat16).  Activations change every step, so their
// images are made by a split pass per use.  Weights change in exactly one place -- the Adam update (W/train.py:459-466,
// tf.train.AdamOptimizer.apply_gradients) -- yet rounds 2-4 re-split every weight matrix every step, once per orientation
// (forward x . W: the transposed image; dx = dz . W^T: the plain one): 4 B read + 6 B written per element and pass, and one
// launch each on the step's critical chain.
//
// Round 5: the Adam update of a matrix that owns images runs as a 64 x 64 tile kernel (optim.hip adam_tile_kernel: the same
// per-element arithmetic as the chunk kernel, bitwise the same weights / slots) that stores the updated tile into the matrix's
// images in the same pass.  This file is the host side of it: a process-wide table
//     (source pointer, rows, cols, row pitch, orientation, planes, scale) -> image pointer
// that the consumers which used to split consult first (yt8m_gemm_auto_grouped, yt8m_lstm_stack_fwd / _bwd, the Python helpers);
// a miss is what it always was -- a split into the caller's scratch.  Demand recording (yt8m_wimg_watch) tells the owner of an arena
// which splits of ITS memory a step performs, so that it allocates exactly those images (youtube-8m_amd/variables.py).
// Validity is the owner's contract: an image is current iff every write to its source went through yt8m_adam_tiles (do_adam = 1)
// or was followed by a refresh (do_adam = 0); the Python owner watches torch's version counter of the arena for the second case.
#include <mutex>
#include <vector>
#include "common.h"

using namespace yt8m;

namespace {

struct Entry {
  const float* src;
  int64_t R, C, ld;
  int trans, planes;
  float scale;
  void* image;
};
std::mutex g_mu;
std::vector<Entry> g_reg;
std::vector<yt8m_wimg_demand> g_dem;
struct Range { const char* lo; const char* hi; int64_t noted; };   // noted: demands ever recorded inside the range (monotonic)
std::vector<Range> g_watch;                               // parameter arenas whose splits are noted as demands
constexpr size_t MAX_DEMANDS = 4096;

bool same(const Entry& e, const float* src, int64_t R, int64_t C, int64_t ld, int trans, int planes, float scale) {
  return e.src == src && e.R == R && e.C == C && e.ld == ld && e.trans == trans && e.planes == planes && e.scale == scale;
}

}  // namespace

// The image of src[R, C] (row pitch ld) -- plain (trans = 0: an [R rows, K = C] operand) or transposed (trans = 1: [C rows, K = R])
// -- with `planes` bf16 planes and every element multiplied by `scale` lives at `image` and is kept current by its owner.
extern "C" int yt8m_wimg_register(const float* src, int64_t R, int64_t C, int64_t ld, int trans, int planes, float scale, void* image) {
  YT8M_REQUIRE(src && image && R >= 1 && C >= 1 && ld >= C, YT8M_E_BADARG, "bad image registration");
  YT8M_REQUIRE((planes == 1 || planes == 2 || planes == 3) && (trans == 0 || trans == 1), YT8M_E_BADARG, "planes in {1, 2, 3}, trans in {0, 1}");
  YT8M_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, YT8M_E_BADARG, "images must be 16-byte aligned");
  std::lock_guard<std::mutex> lk(g_mu);
  for (Entry& e : g_reg)
    if (same(e, src, R, C, ld, trans, planes, scale)) { e.image = image; return YT8M_OK; }
  g_reg.push_back({src, R, C, ld, trans, planes, scale, image});
  return YT8M_OK;
}

// Drops every entry whose source starts inside [lo, hi) (an arena being released); lo == hi == NULL: all.  Returns the number dropped.
extern "C" int64_t yt8m_wimg_unregister(const void* lo, const void* hi) {
  std::lock_guard<std::mutex> lk(g_mu);
  int64_t n = 0;
  for (size_t i = g_reg.size(); i-- > 0;) {
    const char* p = reinterpret_cast<const char*>(g_reg[i].src);
    if ((!lo && !hi) || (p >= static_cast<const char*>(lo) && p < static_cast<const char*>(hi))) {
      g_reg.erase(g_reg.begin() + (long)i);
      ++n;
    }
  }
  return n;
}

extern "C" void* yt8m_wimg_lookup(const float* src, int64_t R, int64_t C, int64_t ld, int trans, int planes, float scale) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (const Entry& e : g_reg)
    if (same(e, src, R, C, ld, trans, planes, scale)) return e.image;
  return nullptr;
}

extern "C" int64_t yt8m_wimg_count(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int64_t)g_reg.size();
}

// Demand recording: every image a split entry point (yt8m_x3_split, yt8m_bf16_image) is asked to make of memory inside a WATCHED
// range is noted once (source, shape, orientation, planes, scale).  on = 1 starts watching [lo, hi) (a parameter arena), on = 0 stops
// and forgets the range's demands.
extern "C" int yt8m_wimg_watch(const void* lo, const void* hi, int on) {
  YT8M_REQUIRE(lo && hi && lo < hi, YT8M_E_BADARG, "bad range");
  std::lock_guard<std::mutex> lk(g_mu);
  const char* l = static_cast<const char*>(lo);
  const char* h = static_cast<const char*>(hi);
  for (size_t i = g_watch.size(); i-- > 0;)
    if (g_watch[i].lo == l && g_watch[i].hi == h) g_watch.erase(g_watch.begin() + (long)i);
  if (on) {
    g_watch.push_back({l, h, 0});
  } else {
    for (size_t i = g_dem.size(); i-- > 0;) {
      const char* p = reinterpret_cast<const char*>(g_dem[i].src);
      if (p >= l && p < h) g_dem.erase(g_dem.begin() + (long)i);
    }
  }
  return YT8M_OK;
}

extern "C" int yt8m_wimg_note_demand(const float* src, int64_t R, int64_t C, int64_t ld, int trans, int planes, float scale) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_watch.empty() || !src || g_dem.size() >= MAX_DEMANDS) return YT8M_OK;
  bool watched = false;
  for (const Range& r : g_watch) watched = watched || (reinterpret_cast<const char*>(src) >= r.lo && reinterpret_cast<const char*>(src) < r.hi);
  if (!watched) return YT8M_OK;
  for (const yt8m_wimg_demand& d : g_dem)
    if (d.src == src && d.R == R && d.C == C && d.ld == ld && d.trans == trans && d.planes == planes && d.scale == scale) return YT8M_OK;
  yt8m_wimg_demand d;
  d.src = src; d.R = R; d.C = C; d.ld = ld; d.trans = trans; d.planes = planes; d.scale = scale; d.pad = 0;
  g_dem.push_back(d);
  for (Range& r : g_watch)
    if (reinterpret_cast<const char*>(src) >= r.lo && reinterpret_cast<const char*>(src) < r.hi) ++r.noted;
  return YT8M_OK;
}

// How many demands were EVER noted inside the watched range [lo, hi) (monotonic while the range is watched; -1: not watched).  The
// owner of one arena compares this with what it has examined: demands of other arenas, or another owner that stops watching, do not
// move it (the process-wide count of yt8m_wimg_demands does both).
extern "C" int64_t yt8m_wimg_demand_generation(const void* lo, const void* hi) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (const Range& r : g_watch)
    if (r.lo == static_cast<const char*>(lo) && r.hi == static_cast<const char*>(hi)) return r.noted;
  return -1;
}

// Copies up to `max` recorded demands to `out` (may be NULL to query the count); returns how many there are.
extern "C" int64_t yt8m_wimg_demands(yt8m_wimg_demand* out, int64_t max) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (out)
    for (int64_t i = 0; i < max && i < (int64_t)g_dem.size(); ++i) out[i] = g_dem[(size_t)i];
  return (int64_t)g_dem.size();
}
