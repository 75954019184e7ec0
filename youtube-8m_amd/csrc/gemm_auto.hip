// gemm_auto.hip -- fp32 products through the library's own choice of kernel (the x3-vs-fp32 dispatch lived in
// youtube-8m_amd/ops.py).  Every fp32 GEMM of the package that is not hand-placed (csrc/lstm_stack.hip places its own) comes through
// yt8m_gemm_auto_grouped: per PROBLEM -- never per group, so a product takes the same kernel and the same summation order
// whether it is launched alone or inside a group -- a cost estimate decides between
//   * six bf16 MFMA products of three-plane split operands (csrc/gemm_x3.hip: 256 x 256 tiles, fp32-grade error, ~2x the rate on
//     large shapes, plus one split pass per operand: 4 B read + 6 B written per element), and
//   * the exact fp32-MFMA kernel (csrc/gemm_f32.hip: 128 x 128 tiles, no preparation).
// The operand images live in a caller-provided scratch (yt8m_gemm_auto_scratch_bytes); an operand shared by several problems of a
// call (the activations of the MoE head's gate and expert products) is split once.  Too little scratch is never an error: the
// problems whose images do not fit stay on the fp32 kernel.
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "common.h"

using namespace yt8m;

namespace {

// measured: fp32-equivalent FLOP/s of either kernel, bytes/s of the split pass (profiles/r2_x3_check.txt)
constexpr double X3_RATE = 195e12, F32_RATE = 110e12, SPLIT_RATE = 3.2e12;

// rate_mul / split_bytes: the three-f16-product form of a declared-h2 product runs 1.65x the six-product kernel (profiles/r6_x3_check.txt)
// and its operand passes move 8 instead of 10 bytes per element
bool image_form_pays(int64_t M, int64_t N, int64_t K, double rate_mul, double split_bytes) {
  if (M <= 0 || N <= 0 || K <= 0) return false;
  const double fl = 2.0 * (double)M * (double)N * (double)K;
  const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256;
  const double eff = (double)M * (double)N / ((double)(tm * tn) * 65536.0);
  // K parts per tile the launch will cut (csrc/gemm_x3.hip x3_launch): up to 8, up to 16 for a very long reduction.  Measured
  // (tools/gemm_auto_probe.py, profiles/r3_plugin_step_times.txt): a product that needs more than 8 parts to fill the chip runs
  // at ~0.7 of the nominal rate (the NetVLAD hidden FC [1024 x 73728] . [73728 x 1024]: 1.12 ms against 1.44 on the fp32 kernel --
  // taken; the einsum-CNN weight gradients [3456 x 38400] . [38400 x 1024], 20 tiles: slower than the fp32 kernel once their split
  // passes are paid -- not taken; its 5- and 9-tile siblings [1152 | 2304 x 38400] . [38400 x 128] run faster on the fp32 kernel's own
  // deep K split than this estimate of it says -- kept there by the 16-tile floor); at K ~ 14 000 the 16-part form of an 18-tile
  // product loses outright (MoE-chain dx shapes).
  const double tiles = (double)(tm * tn);
  double occ = std::min(1.0, tiles * (double)std::max<int64_t>(1, std::min<int64_t>(8, K / 128)) / 256.0);
  if (occ < 1.0 && K >= 32768 && tiles >= 16.0) occ = std::max(occ, 0.7 * std::min(1.0, tiles * 16.0 / 256.0));
  const double tx3 = fl / (X3_RATE * rate_mul * eff * occ) + ((double)M * K + (double)N * K) * split_bytes / SPLIT_RATE + 2e-5;
  const double t32 = fl / (F32_RATE * std::min(1.0, (double)(((M + 127) / 128) * ((N + 127) / 128)) *
                                                        (double)std::max<int64_t>(1, std::min<int64_t>(8, K / 256)) / 768.0));
  return tx3 < 0.9 * t32;
}
bool x3_pays(int64_t M, int64_t N, int64_t K) { return image_form_pays(M, N, K, 1.0, 10.0); }

// h2: the product will take the three-f16-product form (declared role): priced at that kernel's rate -- the weight gradients of the
// MoE heads at 512 rows ([1152..2176 x 512] . [512 x 23 580]) lost to the fp32 kernel by 3 us on the six-product price and run 1.5x
// faster on the form they actually take (DeepCombineChainModel at B = 512; knob YT8M_GEMM_H2_PRICE=0: the six-product price for all)
bool x3_allowed(const yt8m_gemm_problem& q, bool h2 = false) {
  static const bool off = getenv("YT8M_GEMM_X3") != nullptr && atoi(getenv("YT8M_GEMM_X3")) == 0;
  static const bool h2_price = !(getenv("YT8M_GEMM_H2_PRICE") != nullptr && atoi(getenv("YT8M_GEMM_H2_PRICE")) == 0);
  // the x3 launch takes beta in {0, 1} and its split pass at most 64 * 65535 rows per operand
  return !off && (q.beta == 0.f || q.beta == 1.f) && std::max(q.M, std::max(q.N, q.K)) < 64LL * 65535 &&
         ((h2 && h2_price) ? image_form_pays(q.M, q.N, q.K, 1.65, 8.0) : x3_pays(q.M, q.N, q.K));
}

int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }

struct Img { const void* src; int64_t R, C, ld; bool trans; int64_t off; bool h2; const float* word; };   // word: the caller's absmax word (h2) or NULL

// Round 5: a weight gradient dW = x^T . dz (transA, !transB) sums over the rows of BOTH stored operands, neither of which is a weight:
// one power-of-two scale per operand, measured on the device, serves every term of every output element -- such a product runs as
// THREE f16 products of two-plane half images (gemm_h2q_kernel, 1.6x the six-product kernel).  Forward products and dx read a weight
// (whose six-product image is resident, csrc/wimg.hip) and keep the bf16 split.  YT8M_GEMM_H2=0 turns it off.
// Round 6: the role is DECLARED by the caller -- YT8M_GEMM_ROLE_DW or-ed into transA -- never inferred from the
// transposition flags: a generic transA product keeps the six-product / fp32 contract.  The h2 contract (include/yt8m_hip.h): every
// element of an operand is held to 2^-22 of THAT OPERAND's largest magnitude (one scale word per matrix), so an element 2^-k below
// the maximum carries 22 - k significant bits and elements more than 2^-38 below it flush to zero.
// ... for K >= YT8M_GEMM_H2_MINK (default 512): an h2 operand costs three launches (zero the word, absmax, split) against the six-product
// form's one, and at K = 128 -- the MoE head's weight gradient at the headline's B = 128 -- the nine tiny launches of three operands
// (230 us with their launch seams) outlast the product they prepare (130 us).
bool h2_role(int transA_flags, int transB, int64_t K) {
  static const bool off = getenv("YT8M_GEMM_H2") != nullptr && atoi(getenv("YT8M_GEMM_H2")) == 0;
  static const int64_t mink = getenv("YT8M_GEMM_H2_MINK") ? atoll(getenv("YT8M_GEMM_H2_MINK")) : 512;
  if (off || K < mink) return false;
  if ((transA_flags & YT8M_GEMM_ROLE_H2) != 0) return true;           // declared for this product whatever its orientation (round 6)
  return (transA_flags & YT8M_GEMM_ROLE_DW) != 0 && (transA_flags & 1) != 0 && transB == 0;
}
constexpr int64_t H2_SCALE_BYTES = 256;                  // the operand's absmax word (yt8m_h2_absmax), in front of its h2 image in the scratch

// operand image kept current by the optimiser pass (csrc/wimg.hip): no split, no scratch
const void* resident(const void* src, int64_t R, int64_t C, int64_t ld, bool trans) {
  return yt8m_wimg_lookup(static_cast<const float*>(src), R, C, ld, trans ? 1 : 0, 3, 1.0f);
}

}  // namespace

extern "C" int yt8m_gemm_x3_pays(int64_t M, int64_t N, int64_t K) { return x3_pays(M, N, K) ? 1 : 0; }

// bytes of image scratch with which every problem of the call that should run on the bf16 pipe does
extern "C" int64_t yt8m_gemm_auto_scratch_bytes(int transA_flags, int transB, int nprob, const yt8m_gemm_problem* probs) {
  if (nprob < 1 || !probs) return 0;
  const int transA = transA_flags & 1;
  int64_t n = 0;
  for (int i = 0; i < nprob; ++i) {
    const yt8m_gemm_problem& q = probs[i];
    const bool h2q = h2_role(transA_flags, transB, q.K) && (q.N % 4) == 0;
    if (!x3_allowed(q, h2q)) continue;
    if (h2q) {                                                    // (h2 images are 2/3 of these sizes; the scale words sit in front)
      if (!yt8m_wimg_lookup(static_cast<const float*>(q.A), transA ? q.K : q.M, transA ? q.M : q.K, q.lda, transA ? 1 : 0, 2, 0.0f))
        n += H2_SCALE_BYTES + up256(yt8m_x3_image_bytes(q.M, q.K));
      if (!yt8m_wimg_lookup(static_cast<const float*>(q.B), transB ? q.N : q.K, transB ? q.K : q.N, q.ldb, transB == 0 ? 1 : 0, 2, 0.0f))
        n += H2_SCALE_BYTES + up256(yt8m_x3_image_bytes(q.N, q.K));
      continue;
    }
    if (!resident(q.A, transA ? q.K : q.M, transA ? q.M : q.K, q.lda, transA != 0)) n += up256(yt8m_x3_image_bytes(q.M, q.K));
    if (!resident(q.B, transB ? q.N : q.K, transB ? q.K : q.N, q.ldb, transB == 0)) n += up256(yt8m_x3_image_bytes(q.N, q.K));
  }
  return n;
}

// C_i = op(A_i) . op(B_i) (+ bias_i) (+ C_i) for nprob problems sharing transA / transB (fp32 row-major operands, as
// yt8m_gemm_f32_grouped).  workspace: split-K scratch (yt8m_gemm_workspace_bytes, may be NULL); image_scratch: see above (may be
// NULL / small).  used_x3 (may be NULL): bit i set when problem i ran on the bf16 pipe.
extern "C" int yt8m_gemm_auto_grouped(int transA_flags, int transB, int nprob, const yt8m_gemm_problem* probs, void* workspace,
                                      int64_t workspace_bytes, void* image_scratch, int64_t image_scratch_bytes, uint64_t* used_x3,
                                      yt8m_stream_t stream) {
  return yt8m_gemm_auto_grouped_ex(transA_flags, transB, nprob, probs, nullptr, nullptr, workspace, workspace_bytes, image_scratch,
                                   image_scratch_bytes, used_x3, stream);
}

// yt8m_gemm_auto_grouped with absmax words the caller already has (round 6): absmaxA / absmaxB (NULL, or nprob entries, each NULL or a
// device word holding max |operand| as float bits, e.g. from yt8m_moe_mix_xent_bwd_absmax) -- an operand that takes the h2 form under
// such a word skips its memset + yt8m_h2_absmax pass.  The word must cover the whole stored operand matrix.
extern "C" int yt8m_gemm_auto_grouped_ex(int transA_flags, int transB, int nprob, const yt8m_gemm_problem* probs, const float* const* absmaxA,
                                         const float* const* absmaxB, void* workspace, int64_t workspace_bytes, void* image_scratch,
                                         int64_t image_scratch_bytes, uint64_t* used_x3, yt8m_stream_t stream) {
  YT8M_REQUIRE(nprob >= 1 && nprob <= 64 && probs, YT8M_E_BADARG, "1..64 problems per call");
  YT8M_REQUIRE((transA_flags & ~(1 | YT8M_GEMM_ROLE_DW | YT8M_GEMM_ROLE_H2)) == 0 && (transB & ~1) == 0, YT8M_E_BADARG,
               "transA: 0 / 1 (| YT8M_GEMM_ROLE_DW | YT8M_GEMM_ROLE_H2), transB: 0 / 1");
  const int transA = transA_flags & 1;
  YT8M_REQUIRE((reinterpret_cast<uintptr_t>(image_scratch) & 255) == 0, YT8M_E_BADARG, "image scratch must be 256-byte aligned");
  std::vector<Img> imgs;
  std::vector<yt8m_gemm_problem> px, p32, ph;
  std::vector<const float*> hda, hdb;                     // device inverse scales of the h2 problems' operands
  uint64_t mask = 0;
  int64_t off = 0;
  char* const base = static_cast<char*>(image_scratch);
  // *word: where the operand's absmax word lives (h2): the caller's, or the slot in front of the image in the scratch
  auto image_of = [&](const void* src, int64_t R, int64_t C, int64_t ld, bool trans, bool h2, const float* ext, const void** at,
                      const float** word) -> bool {
    if (!h2)
      if (const void* r = resident(src, R, C, ld, trans)) { *at = r; return true; }    // a weight matrix with a resident image
    if (h2 && !ext)                                        // ... or a resident half-plane image: its scale word sits in front of it
      if (const void* r = yt8m_wimg_lookup(static_cast<const float*>(src), R, C, ld, trans ? 1 : 0, 2, 0.0f)) {
        *at = r;
        *word = reinterpret_cast<const float*>(static_cast<const char*>(r) - H2_SCALE_BYTES);
        return true;
      }
    for (const Img& m : imgs)
      if (m.src == src && m.R == R && m.C == C && m.ld == ld && m.trans == trans && m.h2 == h2) {
        *at = base + m.off;
        *word = m.word ? m.word : reinterpret_cast<const float*>(base + m.off - H2_SCALE_BYTES);
        return true;
      }
    const int64_t bytes = up256(trans ? yt8m_x3_image_bytes(C, R) : yt8m_x3_image_bytes(R, C)) + (h2 ? H2_SCALE_BYTES : 0);
    if (!image_scratch || off + bytes > image_scratch_bytes) return false;
    imgs.push_back({src, R, C, ld, trans, off + (h2 ? H2_SCALE_BYTES : 0), h2, h2 ? ext : nullptr});      // (h2: [scale words | image])
    *at = base + off + (h2 ? H2_SCALE_BYTES : 0);
    *word = (h2 && ext) ? ext : reinterpret_cast<const float*>(base + off);
    off += bytes;
    return true;
  };
  for (int i = 0; i < nprob; ++i) {
    const yt8m_gemm_problem& q = probs[i];
    const bool h2 = h2_role(transA_flags, transB, q.K) && (q.N % 4) == 0;   // (the scaled epilogue stores float4: odd widths take the six-product form)
    bool x3 = q.M > 0 && q.N > 0 && x3_allowed(q, h2) && q.A && q.B;
    const void* ia = nullptr;
    const void* ib = nullptr;
    const float* wa = nullptr;
    const float* wb = nullptr;
    if (x3) {
      const int64_t mark = off;
      const size_t nimg = imgs.size();
      // op(A) as [M rows, K]: A is stored [M,K] (plain) or [K,M] (transA: the transposing split); op(B)^T as [N rows, K]
      x3 = image_of(q.A, transA ? q.K : q.M, transA ? q.M : q.K, q.lda, transA != 0, h2, absmaxA ? absmaxA[i] : nullptr, &ia, &wa) &&
           image_of(q.B, transB ? q.N : q.K, transB ? q.K : q.N, q.ldb, transB == 0, h2, absmaxB ? absmaxB[i] : nullptr, &ib, &wb);
      if (!x3) { imgs.resize(nimg); off = mark; }          // not enough scratch for this one: fp32 kernel
    }
    if (x3) {
      yt8m_gemm_problem t = q;
      t.A = ia; t.lda = 0;
      t.B = ib; t.ldb = 0;
      if (h2) {
        ph.push_back(t);
        hda.push_back(wa);
        hdb.push_back(wb);
      } else {
        px.push_back(t);
      }
      mask |= 1ULL << i;
    } else {
      p32.push_back(q);
    }
  }
  for (const Img& m : imgs) {
    void* dst = static_cast<char*>(image_scratch) + m.off;
    int rc;
    if (m.h2) {                                            // scale measured on the device, then the two-plane half image under it
      const float* word = m.word;                            // the caller's word, or measured here: max |src| as float bits
      if (!word) {
        // (a weight inside a watched parameter arena: its owner may keep this image resident from the next step on -- planes 2, scale 0
        //  = "under the device word in front of the image", csrc/wimg.hip)
        yt8m_wimg_note_demand(static_cast<const float*>(m.src), m.R, m.C, m.ld, m.trans ? 1 : 0, 2, 0.0f);
        float* w = reinterpret_cast<float*>(static_cast<char*>(dst) - H2_SCALE_BYTES);
        YT8M_HIP_CHECK(hipMemsetAsync(w, 0, 4, as_stream(stream)));
        rc = yt8m_h2_absmax(static_cast<const float*>(m.src), m.R, m.C, m.ld, w, stream);
        if (rc != YT8M_OK) return rc;
        word = w;
      }
      rc = yt8m_h2_split(static_cast<const float*>(m.src), m.R, m.C, m.ld, 1.0f, word, m.trans ? nullptr : dst, m.trans ? dst : nullptr, nullptr, stream);
    } else {
      rc = yt8m_x3_split(static_cast<const float*>(m.src), m.R, m.C, m.ld, 1.0f, m.trans ? nullptr : dst, m.trans ? dst : nullptr, stream);
    }
    if (rc != YT8M_OK) return rc;
  }
  for (size_t lo = 0; lo < ph.size(); lo += 4) {
    const int n = (int)std::min<size_t>(4, ph.size() - lo);
    int rc = yt8m_gemm_h2_nt_grouped(n, &ph[lo], nullptr, &hda[lo], &hdb[lo], workspace, workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
  }
  for (size_t lo = 0; lo < px.size(); lo += 4) {
    int rc = yt8m_gemm_x3_nt_grouped((int)std::min<size_t>(4, px.size() - lo), &px[lo], workspace, workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
  }
  for (size_t lo = 0; lo < p32.size(); lo += 4) {
    int rc = yt8m_gemm_f32_grouped(transA, transB, (int)std::min<size_t>(4, p32.size() - lo), &p32[lo], workspace, workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
  }
  if (used_x3) *used_x3 = mask;
  return YT8M_OK;
}
