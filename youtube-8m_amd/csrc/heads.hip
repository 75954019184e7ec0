// heads.hip -- whole-head entry points over the kernels of this library, the shape SURVEY.md section 8b proposes for the C ABI
// (yt8m_moe_fwd / yt8m_moe_bwd / yt8m_logistic_fwd_bwd): a non-Python host drives a full classifier head with two calls.
//   MoeModel.create_model       W/all_video_models/moe_model.py:12-65   (+ CrossEntropyLoss W/losses.py:110-130)
//   LogisticModel.create_model  W/all_video_models/logistic_model.py:12-26
// They only sequence existing launches (grouped GEMMs through the library's kernel dispatch -- csrc/gemm_auto.hip: the large products
// run on the bf16 pipe as six products of split operands when the workspace holds their images --, fused mixing + loss, column
// sums) on the caller's stream.
#include <stdlib.h>
#include <algorithm>
#include "common.h"

using namespace yt8m;

namespace {
int64_t up256h(int64_t v) { return (v + 255) / 256 * 256; }
// round 6: the logits product x . [W_g | W_e] declares the h2 role (include/yt8m_hip.h YT8M_GEMM_ROLE_H2), as the host mirror does
// (ops._moe_logits); YT8M_MOE_LOGITS_H2=0: the six-product form
// ... from YT8M_MOE_LOGITS_H2_MIN_ROWS rows on (default 512, as the host mirror): the weights' half-plane images are made per call or kept
// resident by the optimiser pass, which a B = 128 product does not pay back
int logits_role(int64_t B) {
  static const bool off = getenv("YT8M_MOE_LOGITS_H2") != nullptr && atoi(getenv("YT8M_MOE_LOGITS_H2")) == 0;
  static const int64_t min_rows = getenv("YT8M_MOE_LOGITS_H2_MIN_ROWS") ? atoll(getenv("YT8M_MOE_LOGITS_H2_MIN_ROWS")) : 512;
  return (off || B < min_rows) ? 0 : YT8M_GEMM_ROLE_H2;
}
// [ mix+xent partial sums | split-K workspace | operand images of the bf16-pipe products (whatever is left) ]
struct HeadWs { char* mix; char* gemm; int64_t gemm_bytes; char* img; int64_t img_bytes; };
HeadWs carve(void* workspace, int64_t workspace_bytes, int64_t mix_bytes) {
  HeadWs w;
  char* ws = static_cast<char*>(workspace);
  const int64_t off = up256h(mix_bytes), g = up256h(yt8m_gemm_workspace_bytes());
  w.mix = ws;
  w.gemm = ws + off;
  w.gemm_bytes = std::min<int64_t>(g, workspace_bytes - off);
  // keep the image region 256-byte aligned in ABSOLUTE address terms
  const uintptr_t base = reinterpret_cast<uintptr_t>(ws + off + g);
  const int64_t pad = (int64_t)((256 - (base & 255)) & 255);
  w.img = ws + off + g + pad;
  w.img_bytes = std::max<int64_t>(0, workspace_bytes - off - g - pad);
  if (w.img_bytes == 0) w.img = nullptr;
  return w;
}
}  // namespace

// scratch: [ mix+xent partial sums ][ GEMM split-K workspace ]
extern "C" int64_t yt8m_moe_workspace_bytes(int64_t B, int64_t V) {
  return ((yt8m_moe_mix_xent_workspace_bytes(B, V) + 255) / 256) * 256 + yt8m_gemm_workspace_bytes();
}
// The same plus room for the operand images of the head's products when they run on the bf16 pipe (the larger of the forward,
// dx and weight-gradient stages); with only yt8m_moe_workspace_bytes the head stays on the fp32-MFMA kernel.
extern "C" int64_t yt8m_moe_workspace_bytes_ex(int64_t B, int64_t D, int64_t V, int M) {
  if (B <= 0 || D <= 0 || V <= 0 || M < 1) return 0;
  const int64_t Ng = V * (M + 1), Ne = V * M;
  const yt8m_gemm_problem f[2] = {{B, Ng, D, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0.f}, {B, Ne, D, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0.f}};
  const yt8m_gemm_problem x[2] = {{B, D, Ng, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0.f}, {B, D, Ne, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 1.f}};
  const yt8m_gemm_problem w[2] = {{D, Ng, B, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0.f}, {D, Ne, B, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0.f}};
  const int64_t img = std::max(yt8m_gemm_auto_scratch_bytes(logits_role(B), 0, 2, f),
                               std::max(std::max(yt8m_gemm_auto_scratch_bytes(0, 1, 1, x), yt8m_gemm_auto_scratch_bytes(0, 1, 1, x + 1)),
                                        yt8m_gemm_auto_scratch_bytes(1 | YT8M_GEMM_ROLE_DW, 0, 2, w)));
  return up256h(yt8m_moe_mix_xent_workspace_bytes(B, V)) + up256h(yt8m_gemm_workspace_bytes()) + 256 + img;
}

// Zg [B, V(M+1)], Ze [B, VM]: logits out (kept by the caller for yt8m_moe_bwd).  labels NULL: p only (inference).
extern "C" int yt8m_moe_fwd(const float* x, const float* Wg, const float* We, const float* be, const void* labels,
                            int label_dtype, int64_t B, int64_t D, int64_t V, int M, float eps, float* Zg, float* Ze, float* p,
                            float* loss_out, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && D >= 0 && V >= 0 && M >= 1, YT8M_E_SHAPE, "bad shape");
  if (B * V == 0) return YT8M_OK;
  YT8M_REQUIRE(x && Wg && We && be && Zg && Ze && p && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(!labels || loss_out, YT8M_E_BADARG, "labels given but loss_out is NULL");
  YT8M_REQUIRE(workspace_bytes >= yt8m_moe_workspace_bytes(B, V), YT8M_E_BADARG, "workspace too small");
  const int64_t Ng = V * (M + 1), Ne = V * M;
  const HeadWs w = carve(workspace, workspace_bytes, yt8m_moe_mix_xent_workspace_bytes(B, V));
  yt8m_gemm_problem pr[2] = {{B, Ng, D, x, D, Wg, Ng, Zg, Ng, nullptr, 0.0f}, {B, Ne, D, x, D, We, Ne, Ze, Ne, be, 0.0f}};
  int rc = yt8m_gemm_auto_grouped(logits_role(B), 0, 2, pr, w.gemm, w.gemm_bytes, w.img, w.img_bytes, nullptr, stream);
  if (rc != YT8M_OK) return rc;
  if (labels) return yt8m_moe_mix_xent_fwd(Zg, Ze, labels, label_dtype, p, loss_out, B, V, M, eps, w.mix, stream);
  return yt8m_moe_mix_fwd(Zg, Ze, p, B, V, M, stream);
}

// Zg / Ze hold the forward logits on entry and dL/dZ on exit.  dWg, dWe, dbe: beta 0 (overwrite) or 1 (accumulate).
// dx [B,D] may be NULL (the head sits on input data).  upstream scales the loss gradient (e.g. 1 - support_loss_percent).
extern "C" int yt8m_moe_bwd(const float* x, const float* Wg, const float* We, float* Zg, float* Ze, const void* labels,
                            int label_dtype, int64_t B, int64_t D, int64_t V, int M, float eps, float upstream, float* dWg,
                            float* dWe, float* dbe, float beta, float* dx, void* workspace, int64_t workspace_bytes,
                            yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && D >= 0 && V >= 0 && M >= 1, YT8M_E_SHAPE, "bad shape");
  YT8M_REQUIRE(beta == 0.f || beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  if (B * V == 0) return YT8M_OK;
  YT8M_REQUIRE(x && Wg && We && Zg && Ze && labels && dWg && dWe && dbe && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(workspace_bytes >= yt8m_moe_workspace_bytes(B, V), YT8M_E_BADARG, "workspace too small");
  const int64_t Ng = V * (M + 1), Ne = V * M;
  const HeadWs w = carve(workspace, workspace_bytes, yt8m_moe_mix_xent_workspace_bytes(B, V));
  // the mixing backward leaves max |dZg| / max |dZe| in the first two words of the (otherwise idle) mix scratch: the weight-gradient
  // products' h2 split takes them instead of measuring them again
  float* zmax = reinterpret_cast<float*>(w.mix);
  int rc = yt8m_moe_mix_xent_bwd_absmax(Zg, Ze, labels, label_dtype, nullptr, B, V, M, eps, upstream, zmax, stream);
  if (rc != YT8M_OK) return rc;
  if (dx) {                                               // dx = dZg Wg^T + dZe We^T (two launches: the second accumulates)
    yt8m_gemm_problem px = {B, D, Ng, Zg, Ng, Wg, Ng, dx, D, nullptr, 0.0f};
    rc = yt8m_gemm_auto_grouped(0, 1, 1, &px, w.gemm, w.gemm_bytes, w.img, w.img_bytes, nullptr, stream);
    if (rc != YT8M_OK) return rc;
    yt8m_gemm_problem py = {B, D, Ne, Ze, Ne, We, Ne, dx, D, nullptr, 1.0f};
    rc = yt8m_gemm_auto_grouped(0, 1, 1, &py, w.gemm, w.gemm_bytes, w.img, w.img_bytes, nullptr, stream);
    if (rc != YT8M_OK) return rc;
  }
  yt8m_gemm_problem pw[2] = {{D, Ng, B, x, D, Zg, Ng, dWg, Ng, nullptr, beta}, {D, Ne, B, x, D, Ze, Ne, dWe, Ne, nullptr, beta}};
  const float* wordsB[2] = {zmax, zmax + 1};
  rc = yt8m_gemm_auto_grouped_ex(1 | YT8M_GEMM_ROLE_DW, 0, 2, pw, nullptr, wordsB, w.gemm, w.gemm_bytes, w.img, w.img_bytes, nullptr, stream);
  if (rc != YT8M_OK) return rc;
  // (the column sums' scratch starts behind the two words: the products above read them on this stream before the sums overwrite)
  return yt8m_colsum_f32(Ze, B, Ne, Ne, dbe, beta, w.mix, up256h(yt8m_moe_mix_xent_workspace_bytes(B, V)), stream);
}

// p = sigmoid(x W + b), loss = CrossEntropyLoss(p, labels), dW / db (beta 0/1), dx optional.  Z [B,V] scratch (holds dL/dz).
extern "C" int yt8m_logistic_fwd_bwd(const float* x, const float* W, const float* b, const void* labels, int label_dtype,
                                     int64_t B, int64_t D, int64_t V, float eps, float* p, float* loss_out, float* Z, float* dW,
                                     float* db, float beta, float* dx, void* workspace, int64_t workspace_bytes,
                                     yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && D >= 0 && V >= 0, YT8M_E_SHAPE, "bad shape");
  YT8M_REQUIRE(beta == 0.f || beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  if (B * V == 0) return YT8M_OK;
  YT8M_REQUIRE(x && W && b && p && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(!labels || (loss_out && Z && dW && db), YT8M_E_BADARG, "training call needs loss_out, Z, dW, db");
  const int64_t off = ((yt8m_xent_workspace_bytes(B, V) + 255) / 256) * 256;
  YT8M_REQUIRE(workspace_bytes >= off + yt8m_gemm_workspace_bytes(), YT8M_E_BADARG, "workspace too small");
  char* ws = static_cast<char*>(workspace);
  yt8m_gemm_problem pf = {B, V, D, x, D, W, V, p, V, b, 0.0f};
  int rc = yt8m_gemm_f32_grouped(0, 0, 1, &pf, ws + off, workspace_bytes - off, stream);
  if (rc != YT8M_OK) return rc;
  rc = yt8m_act_fwd_f32(YT8M_ACT_SIGMOID, p, p, B * V, stream);
  if (rc != YT8M_OK || !labels) return rc;
  rc = yt8m_xent_fwd_bwd(p, labels, label_dtype, nullptr, loss_out, Z, B, V, eps, 1.0f, ws, stream);      // Z <- dL/dp
  if (rc != YT8M_OK) return rc;
  rc = yt8m_act_bwd_f32(YT8M_ACT_SIGMOID, p, Z, Z, B * V, stream);                                         // Z <- dL/dz
  if (rc != YT8M_OK) return rc;
  yt8m_gemm_problem pw = {D, V, B, x, D, Z, V, dW, V, nullptr, beta};
  rc = yt8m_gemm_f32_grouped(1, 0, 1, &pw, ws + off, workspace_bytes - off, stream);
  if (rc != YT8M_OK) return rc;
  rc = yt8m_colsum_f32(Z, B, V, V, db, beta, ws, off, stream);
  if (rc != YT8M_OK || !dx) return rc;
  yt8m_gemm_problem px = {B, D, V, Z, V, W, V, dx, D, nullptr, 0.0f};
  return yt8m_gemm_f32_grouped(0, 1, 1, &px, ws + off, workspace_bytes - off, stream);
}
