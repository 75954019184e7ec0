// lstm_stack.hip -- the whole MultiRNNCell([BasicLSTMCell] * L) under tf.nn.dynamic_rnn as TWO library calls (forward, backward):
// W/all_frame_models/lstm_model.py:34-47 (lstm_memory_model.py:36-52, its DropoutWrapper included since round 6) and the gradient tf.gradients
// builds through them (W/train.py:435-466), SURVEY.md section 8(b) last row, VERDICT r2 #9.
//
// Everything the measured headline step does for its recurrent stack lives here, behind the C ABI: the time partition (one
// forward launch per layer, four backward parts 2 : 2 : 1 : 1), the stream layout (one HIGH-priority stream per layer for its projection /
// recurrence / dx chain, one stream for the weight-gradient products), the events between them, the operand images of the
// bf16-pipe GEMMs (csrc/gemm_x3.hip) and the choice of product form.  A host binds three functions and owns two buffers
// (tape = activations kept for the backward pass, scratch = everything else); nothing is allocated in here.
//
// Data path of one training step (B videos, F frames, D features, H cells; rows are TIME-major m = f * B + b):
//   forward   raw uint8 frames -> ONE conversion pass: (q - 128) as a one-plane bf16 operand image + row norms r (csrc/u8proj.hip;
//             the dequantise + l2-normalise of W/readers.py:178-187 / W/train.py:343-344 never materialises an fp32 copy)
//             layer 0:  z = r (.) ((q - 128) . (alpha W_x) + beta colsum(W_x)) + b     three exact bf16 products   (x1x3)
//             layer l:  z = out_{l-1} . W_x + b                                        six bf16 products           (x3)
//             recurrence: one persistent launch per (layer, chunk)                     (csrc/lstm_persist.hip)
//   backward  per time part, last to first, top layer first:
//             recurrence (half chip) -> dz;  dx = dz . W_x^T (x3, releases the layer below);  on the weight-gradient stream:
//             dW_h += h^T dz, dW_x += out_{l-1}^T dz (x3, K ranges of whole-sequence transposed images made ONCE per step while the
//             first lone recurrence runs), db += colsum(dz);
//             layer 0 on uint8 frames: dW_x += alpha ((q - 128)^T . (r (.) dz) + (beta / alpha) colsum(r (.) dz))   (x1x3: three
//             products instead of six, and the fp32 time-major copy of the frames the round-2 path kept for it is gone).
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <vector>
#include "common.h"

using namespace yt8m;

namespace {

constexpr int MAXL = 8, MAXP = 64;
constexpr float U8_ALPHA = 4.0f / 255.0f;
constexpr float U8_BETA = 128.0f * (4.0f / 255.0f) + (4.0f / 512.0f - 2.0f);   // dequantise(q) = alpha (q - 128) + beta
constexpr int64_t X3_MIN_ROWS = 1024;                      // F * B below which the fp32-MFMA kernel's smaller tiles win (seq_ops.py)
constexpr int64_t STEP_IMAGES_MAX_BYTES = 8LL << 30;       // per layer; larger launches keep the two-image exchange
constexpr float H2_S = 8192.0f;                            // static scale of h2 images of operands bounded by 1 (LSTM outputs / states)

int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }

// scheduling knobs (tuning aids; defaults are the measured best)
int knob(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

struct Part { int64_t t0, T; };
int chunks(int64_t F, int n, Part* out) {
  if (n < 1) n = 1;
  if (n > F) n = (int)F;
  if (n > MAXP) n = MAXP;
  const int64_t step = (F + n - 1) / n;
  int k = 0;
  for (int64_t t0 = 0; t0 < F; t0 += step) out[k++] = {t0, std::min(step, F - t0)};
  return k;
}

struct Plan {
  int64_t B, F, D, H, FB;
  int L, u8, need_dx;
  int bf16;                                                // hoisted products on ONE-plane bf16 operand images (input_u8 bit 1)
  int nf, nb;
  Part fp[MAXP], bp[MAXP];
  // tape (bytes from its base)
  int64_t z[MAXL], cs[MAXL], hs[MAXL], out[MAXL], rrow, tape_bytes;
  // scratch
  int64_t pws[MAXL], pws_bytes;                            // persistent-recurrence workspaces: FIRST in the scratch (zero once)
  int64_t gws[MAXL + 1], gws_bytes;                        // split-K workspace per layer stream + weight-gradient stream
  int64_t gwx[MAXL], gws2;                                 // ... per dx stream, second weight-gradient stream
  int64_t qimg, w3t, wcs, wxt3[MAXL], xi[MAXL];            // forward operand images
  int64_t dz[MAXL], dbuf[MAXL], work[MAXL];                // backward: dz [F,B,4H], dout of the layer below [F,B,H], running (dh, dc)
  int64_t dz3[MAXL], wx3[MAXL], dzT3[MAXL], dzT3s, csr, dbdummy; // chunk images (dzT3 per layer: the chains may run on two streams)
  int64_t xT[MAXL], hT[MAXL];                              // whole-sequence transposed images (layer input / h_{t-1})
  int64_t cpart[MAXL], cparts;                             // [F B / 64, 4H] per-tile column sums of dz (plain; layer 0 on uint8: r-weighted)
  bool colparts;                                           // every backward part is a multiple of 64 frame rows: the sums ride on the split
  int img_rows;                                            // > 0: the backward recurrences write dz's operand images themselves (round 4)
  int h2;                                                  // layers >= 1: projection and weight gradients as three f16 products (round 5)
  int64_t hsc;                                             // ... their device-side scale words: 256 B per layer
  int64_t hx0;                                             // ... and the float input's
  int64_t hrow, hrow_stride;                               // ... per-row scales of dz for dx (S then 1 / S), per layer
  float keep;                                              // DropoutWrapper(input_keep_prob) around every layer (0: none); round 6
  uint64_t seed[MAXL];
  int emit_max;                                            // the backward recurrences measure max |dz| per frame row and per part themselves
  int64_t rmax, rmax_stride;                               // ... [F B] words per layer
  int64_t cimg[MAXL], cimgs;                               // their column-sum partials: [img_rows x launches][4H] per layer
  int64_t scratch_bytes;
};

int64_t x3_bytes(int64_t rows, int64_t K) { return yt8m_x3_image_bytes(rows, K); }
int64_t x1_bytes(int64_t rows, int64_t K) { return ((rows + 31) / 32) * ((K + 15) / 16) * 1024; }

// why a description is not covered (NULL: it is)
const char* plan(const yt8m_lstm_stack_desc* d, Plan* P) {
  if (!d) return "null description";
  if (d->L < 1 || d->L > MAXL) return "1..8 layers";
  if (d->B < 1 || d->F < 1 || d->D < 1 || d->H < 1) return "empty dimension";
  Plan& p = *P;
  p.B = d->B; p.F = d->F; p.D = d->D; p.H = d->H; p.L = d->L; p.u8 = (d->input_u8 & 1) != 0; p.need_dx = d->need_dx != 0;
  p.bf16 = (d->input_u8 & 2) != 0;
  p.FB = p.F * p.B;
  if (p.FB >= (1LL << 31) - 64) return "too many frame rows";
  if (!yt8m_lstm_persist_supported(p.B, p.H) || !yt8m_lstm_persist_bwd_supported(p.B, p.H))
    return "shape not covered by the persistent recurrence kernels";
  if (p.FB < X3_MIN_ROWS || p.H < 128) return "too small for the bf16-pipe products";
  if (p.u8 && (p.D % 16 != 0 || p.D > 2048)) return "uint8 input needs D % 16 == 0 and D <= 2048";
  if (p.u8 && p.need_dx) return "no gradient with respect to uint8 frames";
  if (p.FB % 16 != 0) return "F * B must be a multiple of 16 (K ranges of the transposed operand images)";
  p.nf = chunks(p.F, d->fwd_chunks > 0 ? d->fwd_chunks : 1, p.fp);
  p.nb = chunks(p.F, d->bwd_chunks > 0 ? d->bwd_chunks : 3, p.bp);
  // The library's own backward partition (bwd_chunks == 0): parts of relative length 2 : 2 : 1 : 1 in forward-time order (round 3:
  // 3 : 2 : 1).  The
  // backward pass runs them last to first: SHORT first parts (the top layer's recurrence runs alone on half the chip while the
  // first one lasts, and the weight-gradient stream has nothing to do yet) and long last ones.  Round 3 (profiles/
  // r3_sched_knobs.md): 23.5-23.6 ms/step for 3:2:1, 7:4:2, 8:5:3, 5:3:1 against 24.0 for three equal parts.
  // Round 4 (profiles/r4_sched_knobs.md, with the column sums taken from the split pass and the rotated backward epilogue): four
  // parts 2 : 2 : 1 : 1 run at 22.30-22.35 ms/step against 23.17 for 3 : 2 : 1 (22.5-22.9 for five / six parts and 1:1:1:1).
  static const char* dflt_parts = "2,2,1,1";
  const char* spec = getenv("YT8M_STACK_BWD_PARTS");
  if (!spec && d->bwd_chunks <= 0 && p.F >= 12) spec = dflt_parts;
  if (spec) {                                                   // relative lengths in forward-time order
    double w[MAXP], tot = 0;
    int n = 0;
    for (const char* q = spec; *q && n < MAXP;) { w[n] = atof(q); tot += w[n++]; while (*q && *q != ',') ++q; if (*q) ++q; }
    if (n >= 1 && tot > 0) {
      int64_t t0 = 0; double acc = 0; int k = 0;
      Part pp[MAXP];
      for (int i = 0; i < n; ++i) {
        acc += w[i];
        int64_t t1 = i + 1 == n ? p.F : (int64_t)(p.F * acc / tot + 0.5);
        if (t1 > t0) { pp[k++] = {t0, t1 - t0}; t0 = t1; }
      }
      bool ok = k >= 1;
      for (int i = 0; i < k; ++i) ok = ok && (pp[i].t0 * p.B) % 16 == 0 && (pp[i].T * p.B) % 16 == 0;
      if (ok) { p.nb = k; for (int i = 0; i < k; ++i) p.bp[i] = pp[i]; }        // else: the equal parts above
    }
  }
  int64_t tmax = 0;
  for (int c = 0; c < p.nf; ++c) {
    if (p.u8 && (p.fp[c].t0 * p.B) % 32 != 0) return "a forward chunk does not start on a 32-row group of the frame image";
    tmax = std::max(tmax, p.fp[c].T);
  }
  for (int c = 0; c < p.nb; ++c) {
    if ((p.bp[c].t0 * p.B) % 16 != 0 || (p.bp[c].T * p.B) % 16 != 0) return "a backward part is not a multiple of 16 frame rows";
    tmax = std::max(tmax, p.bp[c].T);
  }
  const int64_t BH = p.B * p.H, FBH = p.FB * p.H;
  int64_t o = 0;
  for (int l = 0; l < p.L; ++l) {
    p.z[l] = o; o += up256(FBH * 4 * 4);
    p.cs[l] = o; o += up256((FBH + BH) * 4);
    p.hs[l] = o; o += up256((FBH + BH) * 4);
    p.out[l] = o; o += up256(FBH * 4);
  }
  p.rrow = o; o += up256(p.FB * 4);
  p.tape_bytes = o;
  // scratch
  int64_t pb = yt8m_lstm_persist_workspace_bytes_steps(p.B, p.H, tmax);
  const bool per_step = pb <= STEP_IMAGES_MAX_BYTES;       // one exchange image per step of a launch (else: two alternating ones)
  if (!per_step) pb = yt8m_lstm_persist_workspace_bytes(p.B, p.H);
  p.pws_bytes = up256(pb);
  o = 0;
  for (int l = 0; l < p.L; ++l) { p.pws[l] = o; o += p.pws_bytes; }
  p.gws_bytes = up256(yt8m_gemm_workspace_bytes());
  for (int l = 0; l <= p.L; ++l) { p.gws[l] = o; o += p.gws_bytes; }
  for (int l = 0; l < p.L; ++l) { p.gwx[l] = o; o += p.gws_bytes; }
  p.gws2 = o; o += p.gws_bytes;
  auto ib = [&](int64_t rows, int64_t K) { return p.bf16 ? x1_bytes(rows, K) : x3_bytes(rows, K); };
  const int64_t H4 = 4 * p.H;
  p.qimg = o; o += p.u8 ? up256(x1_bytes(p.FB, p.D)) : 0;
  p.w3t = o; o += p.u8 ? up256(ib(H4, p.D)) : 0;
  p.wcs = o; o += up256(H4 * 4);
  int64_t fmax = 0, bmax = 0;
  for (int c = 0; c < p.nf; ++c) fmax = std::max(fmax, p.fp[c].T * p.B);
  for (int c = 0; c < p.nb; ++c) bmax = std::max(bmax, p.bp[c].T * p.B);
  for (int l = 0; l < p.L; ++l) {
    const int64_t Din = l ? p.H : p.D;
    const bool x1 = l == 0 && p.u8;
    p.wxt3[l] = o; o += x1 ? 0 : up256(ib(H4, Din));
    p.xi[l] = o; o += x1 ? 0 : up256(ib(fmax, Din));
    p.dz[l] = o; o += up256(FBH * 4 * 4);
    p.dbuf[l] = o; o += l + 1 < p.L ? up256(FBH * 4) : 0;
    p.work[l] = o; o += up256(4 * BH * 4);
    const bool dxl = l > 0 || p.need_dx;
    p.dz3[l] = o; o += dxl ? up256(ib(bmax, H4)) : 0;
    p.wx3[l] = o; o += dxl ? up256(ib(Din, H4)) : 0;
    p.xT[l] = o; o += x1 ? up256(x1_bytes(p.D, p.FB)) : up256(ib(Din, p.FB));
    p.hT[l] = o; o += up256(ib(p.H, p.FB));
  }
  // Images written by the recurrences (img_rows > 0) are consumed by products on OTHER streams that may lag a whole part behind, so
  // every launch gets its own K range of a whole-sequence image (944 MB per layer at the headline shape) instead of one re-used
  // part-sized buffer that stream order used to protect.
  // Built, bit-exact (tests/test_gpu_round4.py) and measured SLOWER (profiles/r4_sched_knobs.md: 23.35 against 22.45 ms per headline
  // step): the extra stores and ~600 VALU operations per item slow the recurrence itself by more than the split passes cost -- the
  // same verdict as round 3's attempt in the team epilogue.  Off by default; YT8M_STACK_FUSED_IMAGES=1 opts in.
  // Round 5: the hoisted products of the layers above the first whose result is a sum over frame rows or whose operand rows share one
  // magnitude -- the input projection (outputs of the layer below: |h| < 1) and both weight gradients (h^T / out^T against dz^T) -- run
  // as THREE f16 products of two-plane half images (gemm_h2q_kernel: 1.6x the six-product kernel) with a static scale 2^13 on the
  // bounded operand and a device-measured scale on the weights / on each backward part's dz.  dx = dz . W_x^T takes the same form with
  // ONE POWER OF TWO PER FRAME ROW of dz (YT8M_STACK_H2_DX, default 1: a time step whose gradient has decayed by decades against the
  // part's largest keeps its own 22 bits; =0 keeps dx on the six-product form); layer 0 on uint8 frames runs its own two- / three-product
  // forms (h1x2).  YT8M_STACK_H2=0 keeps every product on the bf16 split.
  p.h2 = (knob("YT8M_STACK_H2", 1) && !p.bf16) ? 1 : 0;
  // DropoutWrapper(cell, input_keep_prob) (W/all_frame_models/lstm_memory_model.py:36-44): the mask is applied where each layer's operand
  // images are built (yt8m_h2_split_dropout) and replayed on dx -- f16 product forms on a float input only; keep >= 0.3 keeps the
  // dropped operands (x / keep) inside the half range under the scales measured on the undropped ones
  p.keep = (d->input_keep_prob > 0.f && d->input_keep_prob < 1.f) ? d->input_keep_prob : 0.f;
  for (int l = 0; l < MAXL; ++l) p.seed[l] = d->dropout_seed[l];
  if (d->input_keep_prob < 0.f || d->input_keep_prob > 1.f) return "input_keep_prob must be in [0, 1]";
  if (p.keep > 0.f && (p.u8 || !p.h2 || p.keep < 0.3f || (p.D % 4) != 0 || (p.H % 4) != 0))
    return "input dropout: float input, f16 product forms (YT8M_STACK_H2), keep_prob >= 0.3, D % 4 == 0";
  p.img_rows = (knob("YT8M_STACK_FUSED_IMAGES", 0) && !p.bf16 && !p.h2) ? yt8m_lstm_persist_bwd_images_rows(p.B, p.H) : 0;
  const int64_t trows = p.img_rows ? p.FB : bmax;
  for (int l = 0; l < p.L; ++l) { p.dzT3[l] = o; o += up256(ib(H4, trows)); }
  p.dzT3s = o; o += p.u8 ? up256(ib(H4, trows)) : 0;
  const int64_t ci = p.img_rows ? up256((int64_t)p.img_rows * MAXP * 2 * H4 * 4) : 0;
  for (int l = 0; l < p.L; ++l) { p.cimg[l] = o; o += ci; }
  p.cimgs = o; o += p.u8 ? ci : 0;
  p.csr = o; o += up256(H4 * 4);
  p.dbdummy = o; o += up256(H4 * 4);
  p.hsc = o; o += 256 * MAXL;
  p.hx0 = o; o += 256;                                     // max |x| of a float input (layer 0's h2 operand scale)
  p.hrow_stride = up256(bmax * 4);
  p.hrow = o; o += p.h2 ? p.hrow_stride * 2 * MAXL : 0;   // per-row scales / inverses of a backward part's dz (dx operand), per layer
  // maxima of dz measured by the recurrence itself (yt8m_lstm_persist_bwd_ex): per frame row, per layer
  // (one absmax word per backward PART: words 1 .. 62 of the layer's block.  More parts than words, or sub-parts of a part -- the opt-in
  // YT8M_STACK_SUB0 knobs -- would make launches share a word that the next recurrence raises while a weight-gradient product still
  // reads it: the recurrences then leave the maxima to the separate passes / the sub-part knobs are ignored.  ADVICE r5)
  p.emit_max = (p.h2 && knob("YT8M_STACK_DZ_MAXIMA", 1) && yt8m_lstm_persist_bwd_images_rows(p.B, p.H) > 0 && per_step && p.nb <= 61) ? 1 : 0;
  p.rmax_stride = up256(p.FB * 4);
  p.rmax = o; o += p.emit_max ? p.rmax_stride * p.L : 0;
  p.colparts = knob("YT8M_STACK_COLPARTS", 1) != 0;
  for (int c = 0; c < p.nb; ++c) p.colparts = p.colparts && (p.bp[c].t0 * p.B) % 64 == 0 && (p.bp[c].T * p.B) % 64 == 0;
  const int64_t cp = p.colparts ? up256(((p.FB + 63) / 64) * H4 * 4) : 0;
  for (int l = 0; l < p.L; ++l) { p.cpart[l] = o; o += cp; }
  p.cparts = o; o += p.u8 ? cp : 0;
  p.scratch_bytes = o;
  return nullptr;
}

// ---- per-device streams and events (created once, never destroyed: they live as long as the library) ----------------------------
// Streams are created on demand and only as many as a step uses: the runtime multiplexes every stream of a process onto a few
// hardware queues (4 by default), and kernels of two streams that share a queue serialise -- sixteen idle high-priority streams
// created here once pushed the layer streams of the Python orchestration onto one queue (the bf16 variant of the bench: 19.9 ->
// 46 ms/step).  With L = 2 a step uses the caller's stream + 2 layer streams + 1 weight-gradient stream = 4.
struct DevState {
  hipStream_t rs[MAXL] = {nullptr};
  hipStream_t dxs[MAXL] = {nullptr};  // dz split + dx product of a layer (knob YT8M_STACK_DX_STREAM): frees the layer stream
  hipStream_t sw = nullptr, sw2 = nullptr;
  int prio = 0;
  bool prio_known = false;
  std::vector<hipEvent_t> ev[2];      // forward / backward pools
  size_t used[2] = {0, 0};
  hipEvent_t layer_done[MAXL] = {nullptr};   // most recent backward call: layer l's dW / db are final (yt8m_lstm_stack_layer_done_wait)
  int layers_done = 0;
};
std::mutex g_mu;
DevState g_dev[16];
// One-shot host callback of the NEXT yt8m_lstm_stack_bwd call of this thread (yt8m_lstm_stack_set_prep_hook).
thread_local yt8m_stream_hook g_prep_hook = nullptr;
thread_local void* g_prep_user = nullptr;
// ... and its early optimiser pass (yt8m_lstm_stack_set_early_optimizer): the descriptor and the host tables it points to, copied
struct EarlyOpt {
  bool set = false;
  yt8m_opt_ranges o;
  std::vector<int32_t> tcs, job_tensor;
  std::vector<int64_t> tile_base;
};
thread_local EarlyOpt g_early;

int high_stream(DevState& S, hipStream_t* s) {
  if (*s) return YT8M_OK;
  if (!S.prio_known) {
    int least = 0, greatest = 0;
    YT8M_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    S.prio = greatest;
    S.prio_known = true;
  }
  // HIGH priority: a persistent recurrence needs every workgroup resident and must take freed CUs before the queued workgroups
  // of a weight-gradient GEMM do ("Scheduling around them")
  YT8M_HIP_CHECK(hipStreamCreateWithPriority(s, hipStreamNonBlocking, S.prio));
  return YT8M_OK;
}
int plain_stream(hipStream_t* s) {
  if (*s) return YT8M_OK;
  YT8M_HIP_CHECK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
  return YT8M_OK;
}

int dev_state(int L, DevState** out) {
  int dev = 0;
  YT8M_HIP_CHECK(hipGetDevice(&dev));
  YT8M_REQUIRE(dev >= 0 && dev < 16, YT8M_E_BADARG, "device index out of range");
  DevState& S = g_dev[dev];
  for (int l = 0; l < L; ++l) { int rc = high_stream(S, &S.rs[l]); if (rc != YT8M_OK) return rc; }
  { int rc = plain_stream(&S.sw); if (rc != YT8M_OK) return rc; }
  if (knob("YT8M_STACK_DX_STREAM", 0))
    for (int l = 0; l < L; ++l) { int rc = high_stream(S, &S.dxs[l]); if (rc != YT8M_OK) return rc; }
  if (knob("YT8M_STACK_SW2", 0)) { int rc = plain_stream(&S.sw2); if (rc != YT8M_OK) return rc; }
  *out = &S;
  return YT8M_OK;
}

struct Ev {
  DevState& S;
  int dir;
  int rc = YT8M_OK;
  Ev(DevState& s, int d) : S(s), dir(d) { S.used[d] = 0; }
  hipEvent_t record(hipStream_t st) {
    if (S.used[dir] == S.ev[dir].size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rc = fail(YT8M_E_HIP, "hipEventCreate failed%s", ""); return nullptr; }
      S.ev[dir].push_back(e);
    }
    hipEvent_t e = S.ev[dir][S.used[dir]++];
    if (hipEventRecord(e, st) != hipSuccess) rc = fail(YT8M_E_HIP, "hipEventRecord failed%s", "");
    return e;
  }
  void wait(hipStream_t st, hipEvent_t e) {
    if (e && hipStreamWaitEvent(st, e, 0) != hipSuccess) rc = fail(YT8M_E_HIP, "hipStreamWaitEvent failed%s", "");
  }
};

#define RC(expr) do { int _rc = (expr); if (_rc != YT8M_OK) return _rc; } while (0)

template <typename T> T* at(void* base, int64_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }

}  // namespace

extern "C" int yt8m_lstm_stack_supported(const yt8m_lstm_stack_desc* desc) {
  Plan P;
  const char* why = plan(desc, &P);
  if (why) { snprintf(g_err, sizeof(g_err), "yt8m_lstm_stack: %s", why); return 0; }
  return 1;
}
extern "C" int64_t yt8m_lstm_stack_tape_bytes(const yt8m_lstm_stack_desc* desc) {
  Plan P;
  return plan(desc, &P) ? 0 : P.tape_bytes;
}
extern "C" int64_t yt8m_lstm_stack_scratch_bytes(const yt8m_lstm_stack_desc* desc) {
  Plan P;
  return plan(desc, &P) ? 0 : P.scratch_bytes;
}
extern "C" int yt8m_lstm_stack_partition(const yt8m_lstm_stack_desc* desc, int* fwd_chunks, int* bwd_chunks) {
  Plan P;
  const char* why = plan(desc, &P);
  YT8M_REQUIRE(!why, YT8M_E_SHAPE, why ? why : "");
  if (fwd_chunks) *fwd_chunks = P.nf;
  if (bwd_chunks) *bwd_chunks = P.nb;
  return YT8M_OK;
}

// The library's per-device streams (created on first use): L high-priority layer streams and the weight-gradient stream.  A host
// that runs its own orchestration of the per-call entry points beside the stack (the Python fallback paths do) should use THESE
// streams rather than create more: the runtime multiplexes all streams of a process onto a few hardware queues.
extern "C" int yt8m_lstm_stack_streams(int L, yt8m_stream_t* layer_streams, yt8m_stream_t* wgrad_stream) {
  YT8M_REQUIRE(L >= 1 && L <= MAXL && layer_streams, YT8M_E_BADARG, "1..8 layers, non-null output");
  std::lock_guard<std::mutex> lk(g_mu);
  DevState* S = nullptr;
  RC(dev_state(L, &S));
  for (int l = 0; l < L; ++l) layer_streams[l] = (yt8m_stream_t)S->rs[l];
  if (wgrad_stream) *wgrad_stream = (yt8m_stream_t)S->sw;
  return YT8M_OK;
}

// Views into the tape after yt8m_lstm_stack_fwd: which = 0 outputs of the layer [F,B,H] (time-major; zeros beyond num_frames),
// 1 final cell state c [B,H], 2 final hidden state h [B,H] (copy-through beyond num_frames), 3 gate activations [F,B,4H].
extern "C" int yt8m_lstm_stack_view(const yt8m_lstm_stack_desc* desc, void* tape, int layer, int which, float** out) {
  Plan P;
  const char* why = plan(desc, &P);
  YT8M_REQUIRE(!why, YT8M_E_SHAPE, why ? why : "");
  YT8M_REQUIRE(tape && out && layer >= 0 && layer < P.L && which >= 0 && which <= 3, YT8M_E_BADARG, "bad view");
  const int64_t FBH = P.FB * P.H;
  switch (which) {
    case 0: *out = at<float>(tape, P.out[layer]); break;
    case 1: *out = at<float>(tape, P.cs[layer]) + FBH; break;
    case 2: *out = at<float>(tape, P.hs[layer]) + FBH; break;
    default: *out = at<float>(tape, P.z[layer]); break;
  }
  return YT8M_OK;
}

// Makes `stream` wait until the weight and bias gradients of `layer` written by the most recent yt8m_lstm_stack_bwd call on the
// current device are final (they are, in the order L-1 ... 0, well before the call's own stream is released: layer l's last
// products precede those of the layers below on the weight-gradient stream).  For per-layer gradient all-reduces (W/train.py:
// 624-639 averages tower gradients after the whole backward pass; here layer L-1's bucket is on the wire while layer 0's last
// weight-gradient products still run).
extern "C" int yt8m_lstm_stack_layer_done_wait(int layer, yt8m_stream_t stream) {
  int dev = 0;
  YT8M_HIP_CHECK(hipGetDevice(&dev));
  YT8M_REQUIRE(dev >= 0 && dev < 16, YT8M_E_BADARG, "device index out of range");
  std::lock_guard<std::mutex> lk(g_mu);
  DevState& S = g_dev[dev];
  YT8M_REQUIRE(layer >= 0 && layer < S.layers_done && S.layer_done[layer], YT8M_E_BADARG,
               "no backward call on this device has recorded that layer");
  YT8M_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), S.layer_done[layer], 0));
  return YT8M_OK;
}

// Sticky time-out words of the stack's persistent-recurrence workspaces (include/yt8m_hip.h, yt8m_lstm_persist_status): YT8M_E_HIP
// if any launch of any layer since the previous status call gave up waiting.  Synchronises `stream`.
extern "C" int yt8m_lstm_stack_status(const yt8m_lstm_stack_desc* desc, void* scratch, yt8m_stream_t stream) {
  Plan P;
  const char* why = plan(desc, &P);
  YT8M_REQUIRE(!why, YT8M_E_SHAPE, why ? why : "");
  YT8M_REQUIRE(scratch, YT8M_E_BADARG, "null scratch");
  int rc = YT8M_OK;
  for (int l = 0; l < P.L; ++l) {
    const int r = yt8m_lstm_persist_status(at<char>(scratch, P.pws[l]), stream);
    if (r != YT8M_OK) rc = r;                              // read (= clear) every word
  }
  return rc;
}

// Registers a one-shot callback for the calling thread's NEXT yt8m_lstm_stack_bwd: invoked on the host, once, right after the first
// backward recurrence has been enqueued, with the library's weight-gradient stream (everything the callback enqueues there runs
// behind the operand-image preparation and before the first weight-gradient product).  NULL clears it.
extern "C" int yt8m_lstm_stack_set_prep_hook(yt8m_stream_hook hook, void* user) {
  g_prep_hook = hook;
  g_prep_user = user;
  return YT8M_OK;
}

// clip + Adam of the tensor ranges in `opt` enqueued by the calling thread's NEXT yt8m_lstm_stack_bwd on its weight-gradient stream,
// right after its first backward recurrence (see the hook above for the window).  Copies the descriptor and its host tables.
extern "C" int yt8m_lstm_stack_set_early_optimizer(const yt8m_opt_ranges* opt) {
  g_early.set = false;
  if (!opt) return YT8M_OK;
  YT8M_REQUIRE(opt->nranges >= 1 && opt->nranges <= 8 && opt->tensor_chunk_start_host, YT8M_E_BADARG, "1..8 ranges + the host chunk starts");
  int hi = 0;
  for (int r = 0; r < opt->nranges; ++r) {
    YT8M_REQUIRE(opt->range_lo[r] >= 0 && opt->range_hi[r] >= opt->range_lo[r], YT8M_E_BADARG, "bad tensor range");
    hi = std::max(hi, (int)opt->range_hi[r]);
  }
  YT8M_REQUIRE(opt->njobs >= 0 && (opt->njobs == 0 || (opt->job_tensor_host && opt->job_tile_base_host)), YT8M_E_BADARG, "job tables");
  g_early.o = *opt;
  g_early.tcs.assign(opt->tensor_chunk_start_host, opt->tensor_chunk_start_host + hi + 1);
  g_early.o.tensor_chunk_start_host = g_early.tcs.data();
  g_early.job_tensor.clear();
  g_early.tile_base.clear();
  if (opt->njobs) {
    g_early.job_tensor.assign(opt->job_tensor_host, opt->job_tensor_host + opt->njobs);
    g_early.tile_base.assign(opt->job_tile_base_host, opt->job_tile_base_host + opt->njobs + 1);
    g_early.o.job_tensor_host = g_early.job_tensor.data();
    g_early.o.job_tile_base_host = g_early.tile_base.data();
  }
  g_early.set = true;
  return YT8M_OK;
}

extern "C" int yt8m_lstm_stack_fwd(const yt8m_lstm_stack_desc* desc, const void* x, const int32_t* num_frames, const float* const* W,
                                   const float* const* b, void* tape, int64_t tape_bytes, void* scratch, int64_t scratch_bytes,
                                   yt8m_stream_t stream) {
  Plan P;
  const char* why = plan(desc, &P);
  YT8M_REQUIRE(!why, YT8M_E_SHAPE, why ? why : "");
  YT8M_REQUIRE(x && W && b && tape && scratch, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(tape_bytes >= P.tape_bytes && scratch_bytes >= P.scratch_bytes, YT8M_E_BADARG, "tape / scratch too small");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(tape) | reinterpret_cast<uintptr_t>(scratch)) & 255) == 0, YT8M_E_BADARG,
               "tape and scratch must be 256-byte aligned");
  for (int l = 0; l < P.L; ++l) YT8M_REQUIRE(W[l] && b[l], YT8M_E_BADARG, "null weights");
  std::lock_guard<std::mutex> lk(g_mu);
  DevState* S = nullptr;
  RC(dev_state(P.L, &S));
  Ev ev(*S, 0);
  hipStream_t main = as_stream(stream);
  const int64_t B = P.B, D = P.D, H = P.H, H4 = 4 * H, BH = B * H;
  // bf16-operand mode (--compute_dtype=bfloat16): the hoisted products take ONE-plane images (the bf16 roundings of their operands)
  // on the b1 kernel instead of three-plane images on the x3 kernel; the recurrence stays what it is (fp32-grade)
  const bool bf = P.bf16 != 0;
  auto split = [&](const float* src, int64_t R, int64_t C, int64_t ld, float scale, void* plain, void* trans, hipStream_t st) {
    return bf ? yt8m_bf16_image(src, R, C, ld, scale, plain, trans, (yt8m_stream_t)st) : yt8m_x3_split(src, R, C, ld, scale, plain, trans, (yt8m_stream_t)st);
  };
  hipEvent_t start = ev.record(main);
  for (int l = 0; l < P.L; ++l) ev.wait(S->rs[l], start);
  const void* wxt_img[MAXL] = {nullptr};                   // image of W_x^T per layer: resident, or split into the scratch below
  // per layer, once: zero initial state, operand images of the input weights
  for (int l = 0; l < P.L; ++l) {
    hipStream_t s = S->rs[l];
    const int64_t Din = l ? H : D;
    YT8M_HIP_CHECK(hipMemsetAsync(at<char>(tape, P.cs[l]), 0, (size_t)BH * 4, s));
    YT8M_HIP_CHECK(hipMemsetAsync(at<char>(tape, P.hs[l]), 0, (size_t)BH * 4, s));
    if (!bf) {                                             // max |W_h| (word 63 of the layer's scale words): the backward recurrence's f16 form
      YT8M_HIP_CHECK(hipMemsetAsync(at<char>(scratch, P.hsc + 256 * l + 252), 0, 4, s));
      RC(yt8m_h2_absmax(W[l] + Din * H4, H, H4, H4, at<char>(scratch, P.hsc + 256 * l + 252), (yt8m_stream_t)s));
    }
    if (l == 0 && P.u8) {
      if (P.h2) {
        // (q - 128) as a ONE-plane half image (exact), (alpha W_x)^T as an h2 image under a device-measured scale: two f16 products
        RC(yt8m_u8_frames_image_f16(static_cast<const uint8_t*>(x), num_frames, B, P.F, D, 1e-12f, at<char>(scratch, P.qimg), nullptr,
                                    at<float>(tape, P.rrow), s));
        float* word = at<float>(scratch, P.hsc);
        YT8M_HIP_CHECK(hipMemsetAsync(word, 0, 4, s));
        RC(yt8m_h2_absmax(W[0], D, H4, H4, word, (yt8m_stream_t)s));
        RC(yt8m_h2_split(W[0], D, H4, H4, U8_ALPHA, word, nullptr, at<char>(scratch, P.w3t), nullptr, (yt8m_stream_t)s));
        wxt_img[0] = at<char>(scratch, P.w3t);
      } else {
      RC(yt8m_u8_frames_image(static_cast<const uint8_t*>(x), num_frames, B, P.F, D, 1e-12f, at<char>(scratch, P.qimg), nullptr,
                              at<float>(tape, P.rrow), s));
      // (alpha W_x)^T: rows 4H, K = D -- resident when the optimiser pass keeps the weight's images current (csrc/wimg.hip)
      wxt_img[0] = yt8m_wimg_lookup(W[0], D, H4, H4, 1, bf ? 1 : 3, U8_ALPHA);
      if (!wxt_img[0]) {
        RC(split(W[0], D, H4, H4, U8_ALPHA, nullptr, at<char>(scratch, P.w3t), s));
        wxt_img[0] = at<char>(scratch, P.w3t);
      }
      }
      RC(yt8m_colsum_f32(W[0], D, H4, H4, at<float>(scratch, P.wcs), 0.f, at<char>(scratch, P.gws[0]), P.gws_bytes, s));
    } else {
      if (P.h2) {                                            // W_x^T as an h2 image under a scale measured on the device
        float* word = at<float>(scratch, P.hsc + 256 * l);   // max |W_x| as float bits (the forward's word of this layer)
        YT8M_HIP_CHECK(hipMemsetAsync(word, 0, 4, s));
        RC(yt8m_h2_absmax(W[l], Din, H4, H4, word, (yt8m_stream_t)s));
        RC(yt8m_h2_split(W[l], Din, H4, H4, 1.0f, word, nullptr, at<char>(scratch, P.wxt3[l]), nullptr, (yt8m_stream_t)s));
        wxt_img[l] = at<char>(scratch, P.wxt3[l]);
        if (l == 0) {                                        // a float input has no bound the library knows: its maximum, once per call
          YT8M_HIP_CHECK(hipMemsetAsync(at<char>(scratch, P.hx0), 0, 4, s));
          RC(yt8m_h2_absmax(static_cast<const float*>(x), P.FB, D, D, at<char>(scratch, P.hx0), (yt8m_stream_t)s));
        }
        continue;
      }
      wxt_img[l] = yt8m_wimg_lookup(W[l], Din, H4, H4, 1, bf ? 1 : 3, 1.0f);                       // W_x^T: rows 4H, K = Din
      if (!wxt_img[l]) {
        RC(split(W[l], Din, H4, H4, 1.0f, nullptr, at<char>(scratch, P.wxt3[l]), s));
        wxt_img[l] = at<char>(scratch, P.wxt3[l]);
      }
    }
  }
  std::vector<hipEvent_t> done((size_t)P.L * P.nf, nullptr);
  // Forward layer wavefront (knob YT8M_STACK_FWD_HALF, default 0): half-chip forward recurrences (128 workgroups of eight
  // 16-row tiles each on the f16 kernel) so that layer l's chunk c runs beside layer l - 1's chunk c + 1 (needs fwd_chunks > 1:
  // YT8M_LSTM_PERSIST_FWD_CHUNKS).  Measured: profiles/r6_sched_knobs.md.
  static const int fwd_half = knob("YT8M_STACK_FWD_HALF", 0);
  struct CapGuard {
    bool on;
    explicit CapGuard(bool o) : on(o) { if (on) yt8m_lstm_persist_set_cus(128, -1); }
    ~CapGuard() { if (on) yt8m_lstm_persist_set_cus(-1, -1); }
  } cap_guard(fwd_half != 0 && P.L >= 2 && P.nf >= 2);
  for (int c = 0; c < P.nf; ++c) {
    const int64_t t0 = P.fp[c].t0, T = P.fp[c].T, M = T * B;
    for (int l = 0; l < P.L; ++l) {
      hipStream_t s = S->rs[l];
      const int64_t Din = l ? H : D;
      if (l > 0) ev.wait(s, done[(size_t)(l - 1) * P.nf + c]);
      float* zc = at<float>(tape, P.z[l]) + t0 * B * H4;
      void* gw = at<char>(scratch, P.gws[l]);
      // products on the critical chain (forward projections, dx): their K parts are summed inside the launch -- the separate
      // fix-up pass is one more kernel (and two launch seams) between a recurrence and the next.  Measured: 22.07-22.11 ms/step with it
      // against 21.96-21.97 without -- off; knob YT8M_STACK_CHAIN_COMBINE=1
      static const int chain_combine = knob("YT8M_STACK_CHAIN_COMBINE", 0);
      if (chain_combine) yt8m_x3_set_combine(1);
      if (l == 0 && P.u8) {
        const char* qi = at<char>(scratch, P.qimg) + (t0 * B / 32) * (D / 16) * 1024;
        const int prc = P.h2 ? yt8m_gemm_h1x2_nt_ex(M, H4, D, qi, 0, wxt_img[0], 0, zc, H4, b[0], 1.0f, at<float>(scratch, P.hsc),
                                                    at<float>(tape, P.rrow) + t0 * B, at<float>(scratch, P.wcs), U8_BETA, 0.f, gw, P.gws_bytes, s)
                           : bf ? yt8m_gemm_b1_nt_ex(M, H4, D, qi, 0, wxt_img[0], 0, zc, H4, b[0], 1.0f, at<float>(tape, P.rrow) + t0 * B,
                                                at<float>(scratch, P.wcs), U8_BETA, 0.f, gw, P.gws_bytes, s)
                           : yt8m_gemm_x1x3_nt(M, H4, D, qi, wxt_img[0], zc, H4, b[0], at<float>(tape, P.rrow) + t0 * B,
                                               at<float>(scratch, P.wcs), U8_BETA, gw, P.gws_bytes, s);
        yt8m_x3_set_combine(0);
        RC(prc);
      } else {
        const float* src = l ? at<float>(tape, P.out[l - 1]) + t0 * B * H : static_cast<const float*>(x) + t0 * B * D;
        yt8m_gemm_problem pr = {M, H4, Din, at<char>(scratch, P.xi[l]), 0, wxt_img[l], 0, zc, H4, b[l], 0.0f};
        int grc;
        if (P.h2 && l >= 1) {                                // |out_{l-1}| < 1: static scale 2^13; the weights' inverse scale from the device
          if (P.keep > 0.f)                                  // DropoutWrapper: the image of tf.nn.dropout(out_{l-1}); |x / keep| < 4: 2^15 under 2^13
            RC(yt8m_h2_split_dropout(src, M, Din, H2_S, nullptr, at<char>(scratch, P.xi[l]), nullptr, P.keep, P.seed[l], t0 * B * Din, (yt8m_stream_t)s));
          else
          RC(yt8m_h2_split(src, M, Din, Din, H2_S, nullptr, at<char>(scratch, P.xi[l]), nullptr, nullptr, (yt8m_stream_t)s));
          const float alpha = 1.0f / H2_S;
          const float* dsb = at<float>(scratch, P.hsc + 256 * l);
          grc = yt8m_gemm_h2_nt_grouped(1, &pr, &alpha, nullptr, &dsb, gw, P.gws_bytes, (yt8m_stream_t)s);
        } else if (P.h2) {                                   // float input: both operands under device-measured scales
          const float* dsa = at<float>(scratch, P.hx0);
          const float* dsb = at<float>(scratch, P.hsc);
          if (P.keep > 0.f)
            RC(yt8m_h2_split_dropout(src, M, Din, 1.0f, dsa, at<char>(scratch, P.xi[0]), nullptr, P.keep, P.seed[0], t0 * B * Din, (yt8m_stream_t)s));
          else
          RC(yt8m_h2_split(src, M, Din, Din, 1.0f, dsa, at<char>(scratch, P.xi[0]), nullptr, nullptr, (yt8m_stream_t)s));
          const float alpha = 1.0f;
          grc = yt8m_gemm_h2_nt_grouped(1, &pr, &alpha, &dsa, &dsb, gw, P.gws_bytes, (yt8m_stream_t)s);
        } else {
          RC(split(src, M, Din, Din, 1.0f, at<char>(scratch, P.xi[l]), nullptr, s));
          grc = bf ? yt8m_gemm_b1_nt_grouped(1, &pr, gw, P.gws_bytes, s) : yt8m_gemm_x3_nt_grouped(1, &pr, gw, P.gws_bytes, s);
        }
        yt8m_x3_set_combine(0);
        RC(grc);
      }
      // bf16-operand mode: the recurrent product on one bf16 plane (knob YT8M_STACK_BF16_RECUR, default 1)
      static const int bf_recur_f = knob("YT8M_STACK_BF16_RECUR", 1);
      // fp32 configuration: the recurrent product as three f16 products of two-half-plane splits (yt8m_lstm_persist_fwd_h2; the word is this
      // layer's max |W_h| measured above, on this stream): 6.1 against 7.1 us/step, 17.24 -> 16.76 ms/step.  Follows YT8M_STACK_H2 (with
      // the h2 products off, the stack stays bit for bit what the Python orchestration of the same launches computes); knob
      // YT8M_STACK_H2_RECUR_FWD overrides.
      static const int h2_recur_f_env = knob("YT8M_STACK_H2_RECUR_FWD", -1);
      const int h2_recur_f = h2_recur_f_env >= 0 ? h2_recur_f_env : P.h2;
      if (!bf && h2_recur_f)
        RC(yt8m_lstm_persist_fwd_h2(at<float>(tape, P.z[l]), W[l] + Din * H4, H4, at<float>(tape, P.cs[l]), at<float>(tape, P.hs[l]),
                                    at<float>(tape, P.out[l]), num_frames, t0, T, B, H, desc->forget_bias,
                                    at<char>(scratch, P.hsc + 256 * l + 252), at<char>(scratch, P.pws[l]), P.pws_bytes, s));
      else
      RC((bf && bf_recur_f ? yt8m_lstm_persist_fwd_bf16 : yt8m_lstm_persist_fwd)(
          at<float>(tape, P.z[l]), W[l] + Din * H4, H4, at<float>(tape, P.cs[l]), at<float>(tape, P.hs[l]), at<float>(tape, P.out[l]),
          num_frames, t0, T, B, H, desc->forget_bias, at<char>(scratch, P.pws[l]), P.pws_bytes, s));
      done[(size_t)l * P.nf + c] = ev.record(s);
    }
  }
  for (int l = 0; l < P.L; ++l) ev.wait(main, done[(size_t)l * P.nf + P.nf - 1]);
  return ev.rc;
}

extern "C" int yt8m_lstm_stack_bwd(const yt8m_lstm_stack_desc* desc, const void* x, const int32_t* num_frames, const float* const* W,
                                   void* tape, int64_t tape_bytes, void* scratch, int64_t scratch_bytes, const float* dout_top,
                                   const float* const* dc_final, const float* const* dh_final, float* const* dW, float* const* db,
                                   const float* beta_W, const float* beta_b, float* dx, yt8m_stream_t stream) {
  Plan P;
  const char* why = plan(desc, &P);
  YT8M_REQUIRE(!why, YT8M_E_SHAPE, why ? why : "");
  YT8M_REQUIRE(x && W && tape && scratch && dW && db, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(tape_bytes >= P.tape_bytes && scratch_bytes >= P.scratch_bytes, YT8M_E_BADARG, "tape / scratch too small");
  YT8M_REQUIRE(!P.need_dx || dx, YT8M_E_BADARG, "need_dx is set but dx is NULL");
  for (int l = 0; l < P.L; ++l) {
    YT8M_REQUIRE(W[l], YT8M_E_BADARG, "null weights");
    YT8M_REQUIRE(!beta_W || beta_W[l] == 0.f || beta_W[l] == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
    YT8M_REQUIRE(!beta_b || beta_b[l] == 0.f || beta_b[l] == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  }
  std::lock_guard<std::mutex> lk(g_mu);
  DevState* S = nullptr;
  RC(dev_state(P.L, &S));
  Ev ev(*S, 1);
  hipStream_t main = as_stream(stream), sw = S->sw;
  static const int dx_stream = knob("YT8M_STACK_DX_STREAM", 0), two_sw = knob("YT8M_STACK_SW2", 0);
  const int fuse_dz = P.h2 ? 0 : knob("YT8M_STACK_FUSE_DZ_SPLIT", 0);
  const int64_t B = P.B, D = P.D, H = P.H, H4 = 4 * H, BH = B * H, FB = P.FB;
  const int64_t KBtot = FB / 16;
  const bool bf = P.bf16 != 0;                             // one-plane operand images + the b1 kernel (see yt8m_lstm_stack_fwd)
  const int64_t KBB = bf ? 1024 : 3072;                    // bytes of one K block of a 32-row group in an operand image
  auto split = [&](const float* src, int64_t R, int64_t C, int64_t ld, float scale, void* plain, void* trans, hipStream_t st) {
    return bf ? yt8m_bf16_image(src, R, C, ld, scale, plain, trans, (yt8m_stream_t)st) : yt8m_x3_split(src, R, C, ld, scale, plain, trans, (yt8m_stream_t)st);
  };
  auto gemm = [&](int n, const yt8m_gemm_problem* pr, void* ws, int64_t wsb, hipStream_t st) {
    return bf ? yt8m_gemm_b1_nt_grouped(n, pr, ws, wsb, (yt8m_stream_t)st) : yt8m_gemm_x3_nt_grouped(n, pr, ws, wsb, (yt8m_stream_t)st);
  };
  hipEvent_t start = ev.record(main);
  ev.wait(sw, start);
  if (two_sw) ev.wait(S->sw2, start);
  for (int l = 0; l < P.L; ++l) { ev.wait(S->rs[l], start); if (dx_stream) ev.wait(S->dxs[l], start); }
  if (P.h2)                                                // the parts' absmax words; word 0 of a layer (max |W_x|, the forward's) stays
    for (int l = 0; l < P.L; ++l) {                        // (on the layer's own stream: its recurrences may write them -- emit_max)
      YT8M_HIP_CHECK(hipMemsetAsync(at<char>(scratch, P.hsc + 256 * l + 4), 0, 248, S->rs[l]));   // words 1 .. 62
      if (P.emit_max && l >= 1) YT8M_HIP_CHECK(hipMemsetAsync(at<char>(scratch, P.rmax + l * P.rmax_stride), 0, (size_t)FB * 4, S->rs[l]));
    }
  // weight-gradient stream: the whole-sequence transposed operands (K = frame rows), made while the first recurrence runs alone
  for (int l = 0; l < P.L; ++l) {
    if (!dW[l]) continue;
    if (l == 0 && P.u8) {
      if (P.h2) RC(yt8m_u8_frames_image_t_f16(static_cast<const uint8_t*>(x), num_frames, B, P.F, D, at<char>(scratch, P.xT[0]), sw));
      else RC(yt8m_u8_frames_image_t(static_cast<const uint8_t*>(x), num_frames, B, P.F, D, at<char>(scratch, P.xT[0]), sw));
    } else {
      const float* src = l ? at<float>(tape, P.out[l - 1]) : static_cast<const float*>(x);
      const int64_t Din = l ? H : D;
      if (P.h2 && P.keep > 0.f)                            // dW_x = dropout(x)^T . dz: the mask of the forward pass, replayed
        RC(yt8m_h2_split_dropout(src, FB, Din, l ? H2_S : 1.0f, l ? nullptr : at<float>(scratch, P.hx0), nullptr, at<char>(scratch, P.xT[l]),
                                 P.keep, P.seed[l], 0, (yt8m_stream_t)sw));
      else if (P.h2 && l >= 1) RC(yt8m_h2_split(src, FB, Din, Din, H2_S, nullptr, nullptr, at<char>(scratch, P.xT[l]), nullptr, (yt8m_stream_t)sw));
      else if (P.h2) RC(yt8m_h2_split(src, FB, Din, Din, 1.0f, at<float>(scratch, P.hx0), nullptr, at<char>(scratch, P.xT[0]), nullptr, (yt8m_stream_t)sw));
      else RC(split(src, FB, Din, Din, 1.0f, nullptr, at<char>(scratch, P.xT[l]), sw));
    }
    if (P.h2) RC(yt8m_h2_split(at<float>(tape, P.hs[l]), FB, H, H, H2_S, nullptr, nullptr, at<char>(scratch, P.hT[l]), nullptr, (yt8m_stream_t)sw));
    else RC(split(at<float>(tape, P.hs[l]), FB, H, H, 1.0f, nullptr, at<char>(scratch, P.hT[l]), sw));            // h_{t-1}: hs[0 .. F)
  }
  // Host hook (yt8m_lstm_stack_set_prep_hook): work the caller wants on the weight-gradient stream in the window where that stream
  // is idle and half the chip is free -- behind the image preparation, while the top layer's first recurrence (enqueued below on
  // its high-priority stream, i.e. holding its CUs first) runs alone.  The training step puts clip + Adam of the variables whose
  // gradients are already final (the classifier head: 85 % of LstmModel's parameters) there.
  yt8m_stream_hook hook = g_prep_hook;
  void* hook_user = g_prep_user;
  g_prep_hook = nullptr;
  g_prep_user = nullptr;
  bool early = g_early.set;                                // one-shot, like the hook
  g_early.set = false;
  if (two_sw) ev.wait(S->sw2, ev.record(sw));              // layer 0's chain reads images made on sw
  int phase[MAXL];
  bool wx3_done[MAXL];
  const void* wx_img[MAXL] = {nullptr};
  hipEvent_t dzT_free[MAXL] = {nullptr};
  const bool fused_img = P.img_rows > 0 && !dx_stream && !fuse_dz;
  int64_t img_done_rows[MAXL] = {0};                       // frame rows whose images the layer's launches have written so far
  int img_launches[MAXL] = {0};
  for (int l = 0; l < P.L; ++l) {
    hipStream_t s = S->rs[l];
    float* work = at<float>(scratch, P.work[l]);
    const float* dh = dh_final ? dh_final[l] : nullptr;
    const float* dc = dc_final ? dc_final[l] : nullptr;
    if (dh) YT8M_HIP_CHECK(hipMemcpyAsync(work, dh, (size_t)BH * 4, hipMemcpyDeviceToDevice, s));
    else YT8M_HIP_CHECK(hipMemsetAsync(work, 0, (size_t)BH * 4, s));
    if (dc) YT8M_HIP_CHECK(hipMemcpyAsync(work + BH, dc, (size_t)BH * 4, hipMemcpyDeviceToDevice, s));
    else YT8M_HIP_CHECK(hipMemsetAsync(work + BH, 0, (size_t)BH * 4, s));
    phase[l] = 0;
    wx3_done[l] = false;
  }
  std::vector<hipEvent_t> last;
  // Sub-parts of the BOTTOM layer (knobs YT8M_STACK_SUB0 = "n0,n1,.." per backward part in forward-time order, or
  // YT8M_STACK_SUB0_LAST = n for the part that runs last; default 1 = none).  The bottom layer's last recurrence runs alone on half
  // the chip and its transposed dz image, two weight-gradient products and the bias pass are the tail of the backward pass; cutting
  // that part into sub-launches leaves only the last sub-part's products behind the recurrence -- measured (profiles/
  // r4_sched_knobs.md): 23.25 / 23.16 / 23.37 / 23.57 ms/step for 1 / 2 / 3 / 5 sub-parts, i.e. nothing: the weight-gradient stream
  // is busy from the first double phase to the end, what leaves the tail queues up in front of it.
  int sub0[MAXP];
  for (int c = 0; c < P.nb; ++c) sub0[c] = (c == 0 && P.L > 1 && P.nb > 1 && !getenv("YT8M_STACK_SUB0")) ? std::max(1, knob("YT8M_STACK_SUB0_LAST", 1)) : 1;   // (0 would skip the part)
  if (const char* spec = getenv("YT8M_STACK_SUB0")) {
    int n = 0;
    for (const char* q = spec; *q && n < P.nb;) { sub0[n++] = std::max(1, atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; }
  }
  for (int c = P.nb - 1; c >= 0; --c) {
    hipEvent_t dx_ev = nullptr;
    for (int l = P.L - 1; l >= 0; --l) {
     Part sp[MAXP];
     int nsub = (l == 0 && P.L > 1 && !P.emit_max) ? std::min(sub0[c], MAXP) : 1;   // (sub-parts would share the part's absmax word)
     if (nsub > 1) {                                         // equal sub-parts on 16-frame-row boundaries, else the whole part
       const int64_t step = (P.bp[c].T + nsub - 1) / nsub;
       int k = 0;
       bool ok = (step * B) % (P.colparts ? 64 : 16) == 0;
       for (int64_t u = 0; ok && u < P.bp[c].T; u += step) sp[k++] = {P.bp[c].t0 + u, std::min(step, P.bp[c].T - u)};
       nsub = ok ? k : 1;
     }
     if (nsub == 1) sp[0] = P.bp[c];
     for (int j = nsub - 1; j >= 0; --j) {
      const int64_t t0 = sp[j].t0, T = sp[j].T, M = T * B;
      const int64_t kb0 = t0 * B / 16;
      const bool first = c == P.nb - 1 && j == nsub - 1;
      hipStream_t s = S->rs[l];
      const int64_t Din = l ? H : D;
      if (dx_ev && j == nsub - 1) ev.wait(s, dx_ev);
      const float* dout = l == P.L - 1 ? dout_top : at<float>(scratch, P.dbuf[l]);
      float* dz = at<float>(scratch, P.dz[l]);
      // this launch's K range of the whole-sequence transposed image(s) (fused_img): (4H / 32) row groups x (rows / 16) K blocks
      const int64_t toff = fused_img ? (H4 / 32) * (img_done_rows[l] / 16) * 3072 : 0;            // (fused_img: three-plane mode only)
      char* dzT_img = at<char>(scratch, P.dzT3[l]) + toff;
      char* dzTs_img = at<char>(scratch, P.dzT3s) + toff;
      if (fused_img && img_launches[l] < 2 * MAXP) {
        const bool dxl = l > 0 || P.need_dx;
        yt8m_persist_bwd_images im;
        im.plain = dxl ? at<char>(scratch, P.dz3[l]) : nullptr;
        im.trans = dW[l] ? dzT_img : nullptr;
        const bool sc = dW[l] && l == 0 && P.u8;
        im.trans_scaled = sc ? dzTs_img : nullptr;
        im.rowscale = sc ? at<float>(tape, P.rrow) : nullptr;
        im.colpart = at<float>(scratch, P.cimg[l]) + (int64_t)img_launches[l] * P.img_rows * H4;
        im.colpart_scaled = sc ? at<float>(scratch, P.cimgs) + (int64_t)img_launches[l] * P.img_rows * H4 : nullptr;
        RC(yt8m_lstm_persist_bwd_images(at<float>(tape, P.z[l]), W[l] + Din * H4, H4, at<float>(tape, P.cs[l]), dout, dz,
                                        at<float>(scratch, P.work[l]), phase[l], num_frames, t0, T, B, H, at<char>(scratch, P.pws[l]),
                                        P.pws_bytes, &im, s));
        img_done_rows[l] += M;
        ++img_launches[l];
      } else {
        // (opt-in knobs YT8M_STACK_FUSED_IMAGES + YT8M_STACK_SUB0 together can ask for more launches per layer than there are
        // column-sum slots: the products below would then read images nobody wrote -- refuse instead)
        YT8M_REQUIRE(!fused_img, YT8M_E_SHAPE, "YT8M_STACK_FUSED_IMAGES: more backward launches per layer than image slots (YT8M_STACK_SUB0)");
        // bf16-operand mode: the recurrent product of the backward pass on one bf16 plane too (knob YT8M_STACK_BF16_RECUR, default 1;
        // the launch falls back to the fp32 form by itself where the shape cannot take it)
        static const int bf_recur = knob("YT8M_STACK_BF16_RECUR", 1);
        // fp32 configuration: the recurrent product as three f16 products of two-half-plane splits (yt8m_lstm_persist_bwd_h2: fp32-grade,
        // 14.8 instead of 18.3 us per step; knob YT8M_STACK_H2_RECUR, default 1; falls back by itself where the shape cannot take it)
        static const int h2_recur = knob("YT8M_STACK_H2_RECUR", 1);
        static const int h2_dx_rows = knob("YT8M_STACK_H2_DX", 1);
        if (P.emit_max)                                    // ... and measures max |dz| per frame row (dx operand) and of the part (dW operand)
          RC(yt8m_lstm_persist_bwd_ex(at<float>(tape, P.z[l]), W[l] + Din * H4, H4, at<float>(tape, P.cs[l]), dout, dz,
                                      at<float>(scratch, P.work[l]), phase[l], num_frames, t0, T, B, H,
                                      h2_recur ? at<char>(scratch, P.hsc + 256 * l + 252) : nullptr,
                                      (l >= 1 && h2_dx_rows) ? at<char>(scratch, P.rmax + l * P.rmax_stride) : nullptr,
                                      at<char>(scratch, P.hsc + 256 * l + 4 * (1 + std::min(c, 61))), at<char>(scratch, P.pws[l]), P.pws_bytes, s));
        else if (!bf && h2_recur)
          RC(yt8m_lstm_persist_bwd_h2(at<float>(tape, P.z[l]), W[l] + Din * H4, H4, at<float>(tape, P.cs[l]), dout, dz,
                                      at<float>(scratch, P.work[l]), phase[l], nullptr, num_frames, t0, T, B, H,
                                      at<char>(scratch, P.hsc + 256 * l + 252), at<char>(scratch, P.pws[l]), P.pws_bytes, s));
        else
        RC((bf && bf_recur ? yt8m_lstm_persist_bwd_bf16 : yt8m_lstm_persist_bwd)(
            at<float>(tape, P.z[l]), W[l] + Din * H4, H4, at<float>(tape, P.cs[l]), dout, dz, at<float>(scratch, P.work[l]), phase[l],
            nullptr, num_frames, t0, T, B, H, at<char>(scratch, P.pws[l]), P.pws_bytes, s));
      }
      phase[l] = (int)((phase[l] + T) % 2);
      hipEvent_t rb = ev.record(s);
      if (hook) {                                          // the very first recurrence is enqueued: now the caller's work on sw
        hook(hook_user, (yt8m_stream_t)S->sw);
        hook = nullptr;
      }
      if (early) {                                         // ... and the early clip + Adam pass (yt8m_lstm_stack_set_early_optimizer)
        early = false;
        RC(yt8m_optimizer_ranges(&g_early.o, (yt8m_stream_t)S->sw));
      }
      bool fused_t = false;
      const float* dzc = dz + t0 * B * H4;
      if (j == nsub - 1) dx_ev = nullptr;
      if (l > 0 || P.need_dx) {                            // dz as stored feeds dx: the critical path to the layer below
        hipStream_t sx = dx_stream ? S->dxs[l] : s;        // (on its own stream the layer's next part starts at once)
        if (dx_stream) ev.wait(sx, rb);
        // knob YT8M_STACK_FUSE_DZ_SPLIT: ONE pass over dz writes the plain image (dx, this stream) and the transposed one (the
        // weight-gradient stream then waits for this pass instead of reading dz again)
        fused_t = fuse_dz && dW[l] && !(l == 0 && P.u8);
        if (fused_t && dzT_free[l]) ev.wait(sx, dzT_free[l]);       // the previous part's products have read the image
        static const int h2_dx_knob = knob("YT8M_STACK_H2_DX", 1);
        const bool dx_h2 = P.h2 && l >= 1 && h2_dx_knob;
        if (dx_h2) {
          // dx = dz . W_x^T as three f16 products: dz split ROW by ROW (one power of two per frame row: a time step whose gradient
          // has decayed by decades keeps its own 22 bits), W_x under the scale of its absmax word (measured by the forward pass)
          float* rS = at<float>(scratch, P.hrow + 2 * l * P.hrow_stride);
          float* rI = at<float>(scratch, P.hrow + (2 * l + 1) * P.hrow_stride);
          if (P.emit_max) {                                  // the row maxima came with the recurrence: one pass over dz instead of two
            RC(yt8m_h2_split_rowmax(dzc, M, H4, H4, at<char>(scratch, P.rmax + l * P.rmax_stride + t0 * B * 4), rI, at<char>(scratch, P.dz3[l]),
                                    (yt8m_stream_t)sx));
          } else {
            RC(yt8m_h2_rowscales(dzc, M, H4, H4, rS, rI, (yt8m_stream_t)sx));
            RC(yt8m_h2_split_rows(dzc, M, H4, H4, rS, at<char>(scratch, P.dz3[l]), (yt8m_stream_t)sx));
          }
          const float* wword = at<float>(scratch, P.hsc + 256 * l);
          if (!wx3_done[l]) {
            RC(yt8m_h2_split(W[l], Din, H4, H4, 1.0f, wword, at<char>(scratch, P.wx3[l]), nullptr, nullptr, (yt8m_stream_t)sx));
            wx3_done[l] = true;
          }
          float* dst = at<float>(scratch, P.dbuf[l - 1]) + t0 * B * H;
          RC(yt8m_gemm_h2_nt_ex(M, Din, H4, at<char>(scratch, P.dz3[l]), 0, at<char>(scratch, P.wx3[l]), 0, dst, Din, nullptr, 1.0f, nullptr,
                                wword, rI, 0.0f, at<char>(scratch, dx_stream ? P.gwx[l] : P.gws[l]), P.gws_bytes, (yt8m_stream_t)sx));
          if (P.keep > 0.f)                                  // d dropout(x) / dx: the same mask on the gradient, in place
            RC(yt8m_dropout_f32(dst, dst, M * Din, P.keep, P.seed[l], t0 * B * Din, (yt8m_stream_t)sx));
          dx_ev = ev.record(sx);
          if (c == 0 && j == 0) last.push_back(dx_ev);
        } else {
        if (!fused_img)
          RC(split(dzc, M, H4, H4, 1.0f, at<char>(scratch, P.dz3[l]), fused_t ? at<char>(scratch, P.dzT3[l]) : nullptr, sx));
        if (fused_t) rb = ev.record(sx);
        if (!wx3_done[l]) {                                  // W_x: rows Din, K = 4H (resident image, or split once per step)
          wx_img[l] = yt8m_wimg_lookup(W[l], Din, H4, H4, 0, bf ? 1 : 3, 1.0f);
          if (!wx_img[l]) {
            RC(split(W[l], Din, H4, H4, 1.0f, at<char>(scratch, P.wx3[l]), nullptr, sx));
            wx_img[l] = at<char>(scratch, P.wx3[l]);
          }
          wx3_done[l] = true;
        }
        float* dst = l ? at<float>(scratch, P.dbuf[l - 1]) + t0 * B * H : dx + t0 * B * D;
        yt8m_gemm_problem pr = {M, Din, H4, at<char>(scratch, P.dz3[l]), 0, wx_img[l], 0, dst, Din, nullptr, 0.0f};
        static const int chain_combine_b = knob("YT8M_STACK_CHAIN_COMBINE", 0);
        if (chain_combine_b) yt8m_x3_set_combine(1);
        const int grc = gemm(1, &pr, at<char>(scratch, dx_stream ? P.gwx[l] : P.gws[l]), P.gws_bytes, sx);
        yt8m_x3_set_combine(0);
        RC(grc);
        if (P.keep > 0.f) RC(yt8m_dropout_f32(dst, dst, M * Din, P.keep, P.seed[l], t0 * B * Din, (yt8m_stream_t)sx));
        dx_ev = ev.record(sx);
        if (c == 0 && j == 0) last.push_back(dx_ev);
        }
      }
      // weight-gradient stream: transposed image(s) of this part's dz, the two products, the bias gradient
      hipStream_t sw = (two_sw && l == 0) ? S->sw2 : S->sw;       // (knob: layer 0's chain on a second weight-gradient stream)
      ev.wait(sw, rb);
      void* gw = at<char>(scratch, (two_sw && l == 0) ? P.gws2 : P.gws[P.L]);
      const float bW = first ? (beta_W ? beta_W[l] : 0.f) : 1.f;
      // Bias gradient and the rank-1 remainder of layer 0's uint8 product: ONE pass over the layer's whole dz after its last part
      // (c == 0) instead of a column sum per part -- three launches fewer per part on the chain that ends the backward pass.
      const bool lastpart = c == 0 && j == 0;
      if (dW[l]) {
        if (l == 0 && P.u8 && P.h2) {
          // layer 0 on uint8 frames as f16 products: ONE pass over this part's dz (after its absmax) writes dz^T and (r (.) dz)^T as h2
          // images + the per-tile column sums; dW_x = (q - 128)^T . (r (.) dz) on two products, dW_h = h^T . dz on three
          const float* rr = at<float>(tape, P.rrow) + t0 * B;
          float* word = at<float>(scratch, P.hsc) + 1 + std::min(c, 61);
          if (!P.emit_max) RC(yt8m_h2_absmax(dzc, M, H4, H4, word, (yt8m_stream_t)sw));
          float* cp = P.colparts ? at<float>(scratch, P.cpart[l]) + (t0 * B / 64) * H4 : nullptr;
          float* cps = P.colparts ? at<float>(scratch, P.cparts) + (t0 * B / 64) * H4 : nullptr;
          RC(yt8m_h2_split_ex(dzc, M, H4, H4, 1.0f, word, rr, nullptr, at<char>(scratch, P.dzT3[l]), at<char>(scratch, P.dzT3s), cp, cps,
                              (yt8m_stream_t)sw));
          RC(yt8m_gemm_h1x2_nt_ex(D, H4, M, at<char>(scratch, P.xT[0]) + kb0 * 1024, KBtot, at<char>(scratch, P.dzT3s), 0, dW[0], H4, nullptr,
                                  U8_ALPHA, word, nullptr, nullptr, 0.f, bW, gw, P.gws_bytes, (yt8m_stream_t)sw));
          yt8m_gemm_problem pr = {H, H4, M, at<char>(scratch, P.hT[0]) + kb0 * 2048, KBtot, at<char>(scratch, P.dzT3[l]), 0,
                                  dW[0] + D * H4, H4, nullptr, bW};
          const float alpha = 1.0f / H2_S;
          const float* dsb = word;
          RC(yt8m_gemm_h2_nt_grouped(1, &pr, &alpha, nullptr, &dsb, gw, P.gws_bytes, (yt8m_stream_t)sw));
        } else if (l == 0 && P.u8) {
          const float* rr = at<float>(tape, P.rrow) + t0 * B;
          if (fused_img) {                                   // images and column sums came with the recurrence
          } else if (P.colparts || bf) {                     // bias gradient + rank-1 remainder: per-tile sums from this pass
            float* cp = P.colparts ? at<float>(scratch, P.cpart[l]) + (t0 * B / 64) * H4 : nullptr;
            float* cps = P.colparts ? at<float>(scratch, P.cparts) + (t0 * B / 64) * H4 : nullptr;
            if (bf) RC(yt8m_bf16_image_colsum(dzc, M, H4, H4, 1.0f, rr, nullptr, at<char>(scratch, P.dzT3[l]), at<char>(scratch, P.dzT3s), cp, cps, sw));
            else RC(yt8m_x3_split_colsum(dzc, M, H4, H4, 1.0f, rr, nullptr, at<char>(scratch, P.dzT3[l]), at<char>(scratch, P.dzT3s), cp, cps, sw));
          } else
            RC(yt8m_x3_split_ex(dzc, M, H4, H4, 1.0f, rr, nullptr, at<char>(scratch, P.dzT3[l]), at<char>(scratch, P.dzT3s), sw));
          if (bf)
            RC(yt8m_gemm_b1_nt_ex(D, H4, M, at<char>(scratch, P.xT[0]) + kb0 * 1024, KBtot, dzTs_img, 0, dW[0], H4,
                                  nullptr, U8_ALPHA, nullptr, nullptr, 0.f, bW, gw, P.gws_bytes, sw));
          else
            RC(yt8m_gemm_x1x3_nt_ex(D, H4, M, at<char>(scratch, P.xT[0]) + kb0 * 1024, KBtot, dzTs_img, 0, dW[0], H4,
                                    nullptr, U8_ALPHA, nullptr, nullptr, 0.f, bW, gw, P.gws_bytes, sw));
          yt8m_gemm_problem pr = {H, H4, M, at<char>(scratch, P.hT[0]) + kb0 * KBB, KBtot, dzT_img, 0,
                                  dW[0] + D * H4, H4, nullptr, bW};
          RC(gemm(1, &pr, gw, P.gws_bytes, sw));
        } else if (P.h2) {
          // three f16 products: dz^T of this part under a scale measured on the device (a sum over the part's frame rows: one scale
          // serves it), h^T / out^T under the static 2^13; the bias gradient's per-tile column sums ride on the split as before
          float* word = at<float>(scratch, P.hsc + 256 * l) + 1 + std::min(c, 61);                             // one word per backward part (zeroed at the start)
          if (!P.emit_max) RC(yt8m_h2_absmax(dzc, M, H4, H4, word, (yt8m_stream_t)sw));
          float* cp = (P.colparts && db[l]) ? at<float>(scratch, P.cpart[l]) + (t0 * B / 64) * H4 : nullptr;
          RC(yt8m_h2_split(dzc, M, H4, H4, 1.0f, word, nullptr, at<char>(scratch, P.dzT3[l]), cp, (yt8m_stream_t)sw));
          yt8m_gemm_problem pr[2] = {
              {Din, H4, M, at<char>(scratch, P.xT[l]) + kb0 * 2048, KBtot, at<char>(scratch, P.dzT3[l]), 0, dW[l], H4, nullptr, bW},
              {H, H4, M, at<char>(scratch, P.hT[l]) + kb0 * 2048, KBtot, at<char>(scratch, P.dzT3[l]), 0, dW[l] + Din * H4, H4, nullptr, bW}};
          // (a float bottom-layer input sits under its device-measured scale instead of the static one)
          const float alphas[2] = {l ? 1.0f / H2_S : 1.0f, 1.0f / H2_S};
          const float* dsa[2] = {l ? nullptr : at<float>(scratch, P.hx0), nullptr};
          const float* dsb[2] = {word, word};
          RC(yt8m_gemm_h2_nt_grouped(2, pr, alphas, dsa, dsb, gw, P.gws_bytes, (yt8m_stream_t)sw));
        } else {
          if (fused_img) {
          } else if (!fused_t) {
            if (P.colparts && db[l]) {
              float* cp = at<float>(scratch, P.cpart[l]) + (t0 * B / 64) * H4;
              if (bf) RC(yt8m_bf16_image_colsum(dzc, M, H4, H4, 1.0f, nullptr, nullptr, at<char>(scratch, P.dzT3[l]), nullptr, cp, nullptr, sw));
              else RC(yt8m_x3_split_colsum(dzc, M, H4, H4, 1.0f, nullptr, nullptr, at<char>(scratch, P.dzT3[l]), nullptr, cp, nullptr, sw));
            } else {
              RC(split(dzc, M, H4, H4, 1.0f, nullptr, at<char>(scratch, P.dzT3[l]), sw));
            }
          }
          yt8m_gemm_problem pr[2] = {
              {Din, H4, M, at<char>(scratch, P.xT[l]) + kb0 * KBB, KBtot, dzT_img, 0, dW[l], H4, nullptr, bW},
              {H, H4, M, at<char>(scratch, P.hT[l]) + kb0 * KBB, KBtot, dzT_img, 0, dW[l] + Din * H4, H4, nullptr, bW}};
          RC(gemm(2, pr, gw, P.gws_bytes, sw));
          if (fused_t) dzT_free[l] = ev.record(sw);
        }
      }
      if (lastpart) {
        const float bb = beta_b ? beta_b[l] : 0.f;
        const int64_t ntile = (FB + 63) / 64;                 // rows of the per-tile partial sums (P.colparts)
        const int64_t nimg = (int64_t)img_launches[l] * P.img_rows;   // ... of the recurrences' own partial sums (fused_img)
        if (dW[l] && l == 0 && P.u8) {
          float* dbo = db[0] ? db[0] : at<float>(scratch, P.dbdummy);
          if (fused_img) {
            RC(yt8m_colsum_f32(at<float>(scratch, P.cimgs), nimg, H4, H4, at<float>(scratch, P.csr), 0.f, gw, P.gws_bytes, sw));
            if (db[0]) RC(yt8m_colsum_f32(at<float>(scratch, P.cimg[0]), nimg, H4, H4, db[0], bb, gw, P.gws_bytes, sw));
          } else if (P.colparts) {                                  // fixed-order sums of the partials the split passes left: 10 MB instead of
            RC(yt8m_colsum_f32(at<float>(scratch, P.cparts), ntile, H4, H4, at<float>(scratch, P.csr), 0.f, gw, P.gws_bytes, sw));   // 629
            if (db[0]) RC(yt8m_colsum_f32(at<float>(scratch, P.cpart[0]), ntile, H4, H4, db[0], bb, gw, P.gws_bytes, sw));
          } else {
            RC(yt8m_colsum_weighted_f32(dz, FB, H4, H4, at<float>(tape, P.rrow), dbo, db[0] ? bb : 0.f, at<float>(scratch, P.csr), gw,
                                        P.gws_bytes, sw));
          }
          RC(yt8m_rank1_add_rows_f32(dW[0], D, H4, H4, at<float>(scratch, P.csr), U8_BETA, sw));
        } else if (db[l]) {
          if (fused_img) RC(yt8m_colsum_f32(at<float>(scratch, P.cimg[l]), nimg, H4, H4, db[l], bb, gw, P.gws_bytes, sw));
          else if (P.colparts && dW[l] && !fuse_dz) RC(yt8m_colsum_f32(at<float>(scratch, P.cpart[l]), ntile, H4, H4, db[l], bb, gw, P.gws_bytes, sw));
          else RC(yt8m_colsum_f32(dz, FB, H4, H4, db[l], bb, gw, P.gws_bytes, sw));
        }
        // the layer's gradients are final here -- layer L-1 first, a whole last part of weight-gradient work before layer 0's: a
        // data-parallel host starts each layer's all-reduce from this point (yt8m_lstm_stack_layer_done_wait)
        S->layer_done[l] = ev.record(sw);
      }
     }
    }
  }
  S->layers_done = P.L;
  hipEvent_t fin = ev.record(sw);                          // sw waited for every recurrence part
  ev.wait(main, fin);
  if (two_sw) ev.wait(main, ev.record(S->sw2));
  for (hipEvent_t e : last) ev.wait(main, e);
  return ev.rc;
}
