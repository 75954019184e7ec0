/* yt8m_hip.h -- C ABI of libyt8m_hip.so: the MI355X (gfx950) hot path of wangheda/youtube-8m.
 *
 * The reference has NO native code and NO FFI: every entry point below replaces the
 * stock TensorFlow-1.0 kernels that the cited reference line invokes.  Paths are relative to
 * /root/reference/youtube-8m-wangheda/ (W).  All pointers are DEVICE pointers unless marked host;
 * matrices are row-major; sizes are int64_t; every launch goes to the caller's hipStream_t
 * (passed as void*; NULL = default stream).  Functions are re-entrant, never throw / abort, and return
 * 0 on success or a negative yt8m_status (message via yt8m_last_error(), thread-local).
 */
#ifndef YT8M_HIP_H
#define YT8M_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* yt8m_stream_t; /* hipStream_t */

enum yt8m_status { YT8M_OK = 0, YT8M_E_BADARG = -1, YT8M_E_SHAPE = -2, YT8M_E_HIP = -3, YT8M_E_RCCL = -4 };
enum yt8m_label_dtype { YT8M_LABEL_U8 = 0, YT8M_LABEL_F32 = 1 };

/* 2 since round 3: persistent-recurrence workspaces begin with a sticky error word the host zeroes once (yt8m_lstm_persist_status
 * reads AND clears it), the image GEMMs read yt8m_gemm_problem.lda / ldb as K-block strides.
 * 3 since round 5: yt8m_lstm_stack_desc.input_u8 is a bit field (bit 1 selects bf16 operand images: a truthy 2 from an older host
 * would change behaviour), yt8m_gemm_auto_grouped / yt8m_lstm_stack_* consult the resident weight-image table (yt8m_wimg_*).
 * 4 since round 6: yt8m_lstm_stack_desc grew (input_keep_prob, dropout_seed[8]: a smaller struct from an older host would be read past
 * its end); yt8m_gemm_auto_* read YT8M_GEMM_ROLE_DW in transA (without it a transA product no longer takes the three-f16-product form).
 * A host must check it. */
int yt8m_abi_version(void);
const char* yt8m_last_error(void);
/* name of the gfx target the device code was built for ("gfx950") */
const char* yt8m_built_arch(void);

/* ---- profiling hooks used by bench.py (roofline leg): per-kernel-family hipEvent timing -------- */
/* family ids: 0 gemm_f32 (and the plain bf16 GEMMs), 1 moe_head_fused, 2 elementwise, 3 optimizer, 4 lstm (forward recurrence),
 * 5 netvlad, 6 lstm backward recurrence, 7 gemm_x3 (six bf16 products per fp32 product), 8 gemm_x1x3 (three) */
int yt8m_prof_enable(int on);
int yt8m_prof_reset(void);
/* synchronises the device; returns launches and total ms for a family */
int yt8m_prof_get(int family, int64_t* launches, double* total_ms);
/* algorithmic FLOPs the launches of a family declared while the profiler was on (GEMM: 2 M N K per problem; recurrence:
 * 2 T B H 4H per launch); families: 0 gemm (fp32 MFMA), 1 moe_fused, 2 elementwise, 3 optimizer, 4 lstm_recurrence (forward),
 * 5 netvlad, 6 lstm_recurrence_bwd, 7 gemm_x3 (fp32 products on the bf16 pipe) */
int yt8m_prof_get_flops(int family, double* flops);
/* algorithmic HBM bytes declared the same way by the streaming kernels that are priced against the HBM roof: families 9 vlad_rows,
 * 10 vlad_cols (the two kernels of the fused NetVLAD pooling, timed inside family 5: uint8 frames + cT / agg) */
int yt8m_prof_get_bytes(int family, double* bytes);

/* hardware probes (measured ceilings of THIS box, printed next to the roofline numbers):
 * mfma: register-only v_mfma_f32_32x32x2_f32 loop; FLOPs = blocks*4*iters*32*4096.  copy: float4 stream, n%4==0. */
int yt8m_probe_mfma_f32(int iters, int blocks, float* sink, yt8m_stream_t stream);
/* 32x32x16 bf16: FLOPs = blocks*4*iters*32*32768; random_operands != 0: full-entropy operand bits (the clock the chip holds
 * under the data-dependent power of a real GEMM) instead of a few constant values */
int yt8m_probe_mfma_bf16(int iters, int blocks, int random_operands, float* sink, yt8m_stream_t stream);
int yt8m_probe_copy_f32(const float* src, float* dst, int64_t n, yt8m_stream_t stream);
/* placement: out[2 b] = XCC id, out[2 b + 1] = raw HW_ID of workgroup b (each spins spin_ticks of the 100 MHz clock). */
int yt8m_probe_placement(int* out, int blocks, int spin_ticks, yt8m_stream_t stream);
/* A stream whose dispatches are confined to the CUs of `mask` (hipExtStreamCreateWithCUMask; bit i of word i/32 = CU i in
 * the runtime's numbering).  The LSTM backward pass keeps the GEMMs that run beside a half-chip persistent recurrence on
 * such a stream so the recurrence always finds its CUs free.  No reference counterpart (TF places kernels itself). */
int yt8m_stream_create_cu_mask(const uint32_t* mask, int words, yt8m_stream_t* stream);
int yt8m_stream_destroy(yt8m_stream_t stream);

/* ---- GEMM: C[M,N] = op(A)[M,K] . op(B)[K,N] (+ bias[N]) (+ beta*C), exact fp32 on v_mfma_f32_32x32x2_f32.
 * Replaces tf.matmul / slim.fully_connected's MatMul+BiasAdd (W/all_video_models/moe_model.py:40-52,
 * logistic_model.py:23-25, deep_combine_chain_model.py:29-34, BasicLSTMCell _linear) and their
 * autodiff transposes.  transA=0: A is [M,K] (lda>=K); transA=1: A is [K,M] (lda>=M).
 * transB=0: B is [K,N] (ldb>=N); transB=1: B is [N,K] (ldb>=K).  beta must be 0 or 1.
 * bias may be NULL.  colsum (optional, [N]) : NULL, reserved. */
int yt8m_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                  const float* A, int64_t lda, const float* B, int64_t ldb,
                  float* C, int64_t ldc, const float* bias, float beta, yt8m_stream_t stream);

/* batched form (no bias): problem i uses A + i*strideA, B + i*strideB, C + i*strideC.  Used for the per-video
 * aggregation GEMMs (NetVLAD a^T.x, attention pooling w^T.out; SURVEY.md Appendix B,
 * W/all_frame_models/lstm_attention_max_pooling_model.py:63). */
int yt8m_gemm_f32_batched(int transA, int transB, int64_t M, int64_t N, int64_t K,
                          const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb, int64_t strideB,
                          float* C, int64_t ldc, int64_t strideC, float beta, int64_t batch, yt8m_stream_t stream);

/* grouped / persistent form: up to 4 problems with the same transA/transB share ONE launch of 768 resident
 * workgroups (3 per CU).  Whole rounds of tiles run data-parallel; the remainder tiles are split along K into
 * `workspace` and summed by a deterministic fix-up pass, so the chip has no wave-quantisation tail (the MoE head's
 * gate + expert GEMMs -- moe_model.py:40-52 -- are 888 + 592 tiles on 768 slots).  workspace >=
 * yt8m_gemm_workspace_bytes() enables the split (NULL: remainder tiles run whole). */
typedef struct yt8m_gemm_problem {
  int64_t M, N, K;
  const void* A; int64_t lda;   /* float (f32 entry points) or bf16 (yt8m_gemm_bf16_nt_grouped); ld in elements */
  const void* B; int64_t ldb;
  float* C; int64_t ldc;
  const float* bias;  /* may be NULL */
  float beta;         /* 0 or 1 */
} yt8m_gemm_problem;
int64_t yt8m_gemm_workspace_bytes(void);
int yt8m_gemm_f32_grouped(int transA, int transB, int nprob, const yt8m_gemm_problem* probs, void* workspace,
                          int64_t workspace_bytes, yt8m_stream_t stream);
/* bf16 operands, fp32 accumulate / output, "NT": C[M,N] = A[M,K] . B[N,K]^T with BOTH operands K-contiguous bf16
 * (v_mfma_f32_32x32x16_bf16; same persistent scheduler).  K, lda, ldb even; 16-byte aligned rows (lda % 8 == 0) take the
 * LDS-DMA path.  Serves the bf16 configuration (BASELINE config 5): x / W^T / dZ^T are kept as bf16 copies. */
int yt8m_gemm_bf16_nt_grouped(int nprob, const yt8m_gemm_problem* probs, void* workspace, int64_t workspace_bytes,
                              yt8m_stream_t stream);
/* fp32 GEMM on the bf16 matrix pipe (csrc/gemm_x3.hip): every fp32 operand element is split exactly into three bf16 terms
 * (a = a1 + a2 + a3) and C accumulates, in fp32, the six partial products of weight >= 2^-16 (a1b1, a1b2, a2b1, a1b3, a2b2,
 * a3b1); the dropped terms are <= 2^-23 |a b|, the rounding an fp32 FMA commits on the product.  Same role as yt8m_gemm_f32
 * (tf.matmul and its autodiff transposes) at up to 2.6x its rate.
 * yt8m_x3_split: fp32 src [R, C] (row stride ld, every element multiplied by scale first) -> "x3 images"
 *   [row][ceil(K/16)][3][16] bf16.  plain (rows = R, K = C) serves src as a K-contiguous operand, trans (rows = C, K = R)
 *   serves src^T; either may be NULL; sizes from yt8m_x3_image_bytes(rows, K); 16-byte aligned.
 * yt8m_gemm_x3_nt_grouped: C[M,N] (+)= A . B^T (+ bias); problem.A / .B are the x3 images of A ([M rows, K]) and B ([N rows, K]),
 *   lda / ldb are ignored, K is the logical K.  workspace as for yt8m_gemm_f32_grouped. */
int64_t yt8m_x3_image_bytes(int64_t rows, int64_t K);
int yt8m_x3_split(const float* src, int64_t R, int64_t C, int64_t ld, float scale, void* plain, void* trans, yt8m_stream_t stream);
int yt8m_gemm_x3_nt_grouped(int nprob, const yt8m_gemm_problem* probs, void* workspace, int64_t workspace_bytes,
                            yt8m_stream_t stream);
/* fp32 products as THREE f16 MFMA products (round 5; the same tf.matmul call sites as yt8m_gemm_x3_nt_grouped, at half its matrix time):
 * every operand element times a power-of-two scale S is split into two IEEE halves a S = hi + lo ("h2 image": the x3 image layout
 * with two half planes, yt8m_x3_image_bytes(rows, K) * 2 / 3 bytes) and C accumulates hi hi + hi lo + lo hi in fp32: 2^-21 |a b| per
 * term.  Half has five exponent bits, so the CALLER owns the scales: |scale . src| < 65504 (clamped), alpha = 1 / (S_a S_b) from the host
 * and / or dsa / dsb device words written by yt8m_h2_absmax.  Meant for products that sum over an operand's rows (weight
 * gradients) or whose rows share one magnitude (l2-normalised inputs, LSTM outputs); per-row dynamic range stays on the x3 form.
 *   yt8m_h2_absmax        : max |src| as float bits into a zeroed 32-bit device word (atomicMax).  The device-chosen scale of an operand
 *                           is S_d = the power of two with max S_d in [2^13, 2^14); split and product both derive it from the word.
 *   yt8m_h2_split         : src [R, C] -> plain ([R rows, K = C]) and / or trans ([C rows, K = R]) h2 images of S_d . scale . src
 *                           (dscale = the operand's absmax word, or NULL: S_d = 1); colpart as yt8m_x3_split_colsum (of the UNscaled source).
 *   yt8m_gemm_h2_nt_grouped: C_i (+)= alphas[i] / (S_d,a S_d,b) . A_i . B_i^T (+ bias); alphas / dsa / dsb (or entries) may be NULL
 *                           (dsa[i] / dsb[i] = the absmax words of operands split with a device-chosen scale). */
int yt8m_h2_split(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const float* dscale, void* plain, void* trans,
                  float* colpart, yt8m_stream_t stream);
int yt8m_h2_absmax(const float* src, int64_t R, int64_t C, int64_t ld, void* word, yt8m_stream_t stream);
/* Sticky degradation counters of the h2 split passes on the current device: counts[0] = elements CLAMPED (|x . S| beyond the largest
 * half: the operand outgrew its scale; results degrade), counts[1] = nonzero elements FLUSHED to a zero half (more than 2^-38 below
 * their scale's maximum).  Waits for `stream`; reset != 0 zeroes them.  A training host checks counts[0] == 0 now and then. */
int yt8m_h2_degraded(uint64_t* counts, int reset, yt8m_stream_t stream);
/* yt8m_h2_split of tf.nn.dropout(src, keep_prob) in one pass: the image(s) of x / keep_prob where the Philox stream of (seed, offset + row C
 * + col) keeps the element, else 0 -- bit for bit what yt8m_dropout_f32(src, ., R C, keep_prob, seed, offset) followed by yt8m_h2_split
 * writes.  src is [R, C] contiguous.  (DropoutWrapper(input_keep_prob) inside yt8m_lstm_stack_fwd / _bwd.) */
int yt8m_h2_split_dropout(const float* src, int64_t R, int64_t C, float scale, const float* dscale, void* plain, void* trans,
                          float keep_prob, uint64_t seed, int64_t offset, yt8m_stream_t stream);
/* yt8m_h2_rowscales + yt8m_h2_split_rows in one pass over src when the row maxima are already known (rowmax[r] = max |src[r, :]| as float
 * bits, e.g. from yt8m_lstm_persist_bwd_ex): the plain h2 image [R rows, K = C] of diag(S) . src, S[r] the power of two that brings the
 * row's maximum into [2^13, 2^14) (1 for an all-zero row), and inv[r] = 1 / S[r] (the rowscale of yt8m_gemm_h2_nt_ex). */
int yt8m_h2_split_rowmax(const float* src, int64_t R, int64_t C, int64_t ld, const void* rowmax, float* inv, void* plain,
                         yt8m_stream_t stream);
/* h2 forms of yt8m_x3_split_colsum and yt8m_gemm_x1x3_nt_ex -- the uint8 layer-0 projection and weight gradient of the recurrent stack
 * (readers.py:178-187 folded into lstm_model.py:34-47 and its gradient) as TWO f16 products: A1 = (q - 128) as a one-plane HALF image
 * (exact; yt8m_u8_frames_image_f16 / _t_f16: the half forms of yt8m_u8_frames_image / _t), B2 = an h2 image under the device-chosen
 * scale of the absmax word dsb.  C (+)= alpha . rowscale[m] . (A1 . B^T / S_b + colsum_scale . colsum[n]) + bias[n]. */
int yt8m_h2_split_ex(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const void* dscale, const float* rowscale, void* plain,
                     void* trans, void* trans_scaled, float* colpart, float* colpart_scaled, yt8m_stream_t stream);
int yt8m_gemm_h1x2_nt_ex(int64_t M, int64_t N, int64_t K, const void* A1, int64_t ska, const void* B2, int64_t skb, float* C, int64_t ldc,
                         const float* bias, float alpha, const void* dsb, const float* rowscale, const float* colsum, float colsum_scale,
                         float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* An h2 operand whose ROWS each keep their own precision (dx = dz . W_x^T of time steps whose gradients differ by decades): one power of
 * two per row -- S[r] with max |src[r, :]| S[r] in [2^13, 2^14), inv[r] = 1 / S[r] --, the plain image of diag(S) . src, and the product
 * with rowscale = inv. */
int yt8m_h2_rowscales(const float* src, int64_t R, int64_t C, int64_t ld, float* S, float* inv, yt8m_stream_t stream);
int yt8m_h2_split_rows(const float* src, int64_t R, int64_t C, int64_t ld, const float* S, void* plain, yt8m_stream_t stream);
int yt8m_gemm_h2_nt_ex(int64_t M, int64_t N, int64_t K, const void* A2, int64_t ska, const void* B2, int64_t skb, float* C, int64_t ldc,
                       const float* bias, float alpha, const void* dsa, const void* dsb, const float* rowscale, float beta, void* workspace,
                       int64_t workspace_bytes, yt8m_stream_t stream);
/* The max-pooled einsum CNN of W/all_frame_models/cnn_deep_combine_chain_model.py:60-82,100-106 on raw uint8 frames (csrc/cnn_pool.hip; the
 * dense products are yt8m_gemm_h1x2_nt_ex launches on the yt8m_u8_frames_image_f16 image, one per (filter, frame shift), at time-major row
 * offsets -- youtube-8m_amd/seq_ops.py _PooledCnnU8).
 *   yt8m_timepool_max_f32 : y [F B rows (t B + b), N] (row stride ldy) -> out [B, N] = tf.reduce_max over the frames, idx [B, N] = the FIRST
 *                           frame that attains it (row stride ldo for both).  N, ldy, ldo multiples of 4; 16-byte aligned operands.
 *   yt8m_u8_cnn_pool_dw   : the filter's gradient through the pooling: dW [fs D, N] (beta = 0 / 1: overwrite / accumulate)
 *                           dW[i D + d, n] (+)= sum_b g[b, n] x[idx[b, n] - i, b, d]   (terms with idx - i < 0 dropped),
 *                           x = the dequantised, l2-normalised, padding-masked frames of q [B, F, D] uint8 (W/utils.py:23-38 +
 *                           default_transformer.py:4-8) recomputed from the bytes: r_tm [F B] by row t B + b is yt8m_u8_frames_image_f16's
 *                           r_out.  g, idx: [B, N] with row stride ldg.  B gathered rows per column instead of a [D, F B] x [F B, N] product. */
int yt8m_timepool_max_f32(const float* y, int64_t F, int64_t B, int64_t N, int64_t ldy, float* out, int32_t* idx, int64_t ldo,
                          yt8m_stream_t stream);
 /* yt8m_timepool_shiftmax_f32: the pooling over PER-SHIFT partial outputs of ONE product for the whole CNN: z [F B rows, ldz], columns of
 * filter k (fs[k] shifts, ncol[k] columns, k < nfilt <= 8; host arrays) at sum_{j<k} fs[j] ncol[j] + i ncol[k] + n hold x[t, b] . W_k[i D : (i + 1) D][:, n];
 * cnn_output[t, b, k, n] = sum_i z[(t - i) B + b, .] (i ascending, t - i >= 0), out / idx [B, sum ncol] as yt8m_timepool_max_f32. */
int yt8m_timepool_shiftmax_f32(const float* z, int64_t F, int64_t B, int64_t ldz, int nfilt, const int32_t* fs, const int32_t* ncol, float* out,
                               int32_t* idx, int64_t ldo, yt8m_stream_t stream);
int yt8m_u8_cnn_pool_dw(const uint8_t* q, const float* r_tm, const int32_t* idx, const float* g, int64_t ldg, int64_t B, int64_t F, int64_t D,
                        int64_t N, int64_t fs, float* dW, float beta, yt8m_stream_t stream);
int yt8m_u8_frames_image_f16(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, float eps, void* image,
                             float* x_tm, float* r_out, yt8m_stream_t stream);
int yt8m_u8_frames_image_t_f16(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, void* image_t,
                               yt8m_stream_t stream);
int yt8m_gemm_h2_nt_grouped(int nprob, const yt8m_gemm_problem* probs, const float* alphas, const float* const* dsa,
                            const float* const* dsb, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* Where the K parts of a split tile are summed by the following x3 / x1x3 / b1 launches OF THE CALLING THREAD: 0 = process default
 * (separate fix-up pass; YT8M_X3_FUSED_COMBINE=1 flips it), 1 = by the last part to arrive, inside the launch (no second kernel on
 * the stream: for products on a critical chain), 2 = separate pass.  Same sums in the same order either way. */
int yt8m_x3_set_combine(int mode);
/* Main-loop schedule of the following x3 / x1x3 / b1 launches OF THE CALLING THREAD: 0 = process default (the interleaved
 * kernels; YT8M_X3_PIPE=0 / YT8M_B1_PIPE=0 flip it), 1 = interleaved (one LDS read or LDS-DMA request behind each MFMA, request
 * ring three steps deep), 2 = the round-3 kernels.  Same products in the same order: bit-identical results; kept for A/B runs. */
int yt8m_x3_set_schedule(int mode);
/* fp32 products through the LIBRARY's choice of kernel (csrc/gemm_auto.hip): per problem -- never per group, so a product takes
 * the same kernel and summation order alone or grouped -- a cost estimate (yt8m_gemm_x3_pays: tile efficiency at 256 x 256,
 * occupancy, the split passes) picks the six-product bf16-pipe kernel or the fp32-MFMA kernel.  Arguments as yt8m_gemm_f32_grouped
 * (up to 64 problems); image_scratch (yt8m_gemm_auto_scratch_bytes of the same arguments, 256-byte aligned) holds the operand
 * images -- an operand shared by several problems is split once; with too little of it the problems that do not fit stay on the
 * fp32 kernel.  used_x3 (may be NULL): bit i = problem i ran on the bf16 pipe.  YT8M_GEMM_X3=0 keeps everything on the fp32 kernel.
 * Replaces every slim.fully_connected / tf.matmul site outside the recurrent stack (W/all_video_models/moe_model.py:43-55, ...).
 * ROLE (round 6): transA may carry YT8M_GEMM_ROLE_DW (transA = 1 | YT8M_GEMM_ROLE_DW): the caller DECLARES the product a weight
 * gradient dW = x^T . dz -- neither stored operand is a weight, the sum runs over the rows of both -- and accepts the "h2" form for it:
 * three f16 MFMA products of two-half-plane images under ONE power-of-two scale per operand, measured on the device (K >= 512,
 * N % 4 == 0; YT8M_GEMM_H2=0 turns it off).  Its precision contract: every element of an operand is held to 2^-22 of THAT
 * operand's largest magnitude, i.e. an element 2^-k below the matrix maximum keeps 22 - k significant bits and elements more than
 * 2^-38 below it flush to zero; products accumulate in fp32.  Without the flag a product never takes that form, whatever its
 * transposition flags (six bf16 products of exact three-plane splits, or the fp32-MFMA kernel: fp32-grade for every element). */
#define YT8M_GEMM_ROLE_DW 0x100
/* Round 6: the caller accepts the same three-f16-product form (same contract: one device-measured power-of-two scale per operand
 * matrix) for a product of ANY orientation -- declared for the MoE head's logits x . [W_g | W_e] (W/all_video_models/moe_model.py:43-52),
 * whose input is l2-normalised and whose operands are each one weight matrix: K = 1152 terms of comparable magnitude.  The operands'
 * half-plane images are made per call (absmax + split; a weight's resident six-product image is not used). */
#define YT8M_GEMM_ROLE_H2 0x200
int yt8m_gemm_x3_pays(int64_t M, int64_t N, int64_t K);
int64_t yt8m_gemm_auto_scratch_bytes(int transA, int transB, int nprob, const yt8m_gemm_problem* probs);
int yt8m_gemm_auto_grouped(int transA, int transB, int nprob, const yt8m_gemm_problem* probs, void* workspace,
                           int64_t workspace_bytes, void* image_scratch, int64_t image_scratch_bytes, uint64_t* used_x3,
                           yt8m_stream_t stream);
/* Round 6: the same call with absmax words the caller already has -- absmaxA / absmaxB: NULL, or nprob entries, each NULL or a device word
 * holding max |stored operand| as float bits (what yt8m_h2_absmax writes; e.g. from yt8m_moe_mix_xent_bwd_absmax).  An operand that takes
 * the h2 form under such a word skips its memset + absmax pass; results are bitwise those of yt8m_gemm_auto_grouped. */
int yt8m_gemm_auto_grouped_ex(int transA, int transB, int nprob, const yt8m_gemm_problem* probs, const float* const* absmaxA,
                              const float* const* absmaxB, void* workspace, int64_t workspace_bytes, void* image_scratch,
                              int64_t image_scratch_bytes, uint64_t* used_x3, yt8m_stream_t stream);
/* The uint8 input projection on the same kernel ("readers.py uint8 -> float dequantise folded into the first GEMM",
 * W/readers.py:178-187 -> W/all_frame_models/lstm_model.py:34-47):
 *   C[M,N] = rowscale[m] * (A . B^T + colsum_scale * colsum[n]) + bias[n]
 * A1: ONE-plane image of (q - 128) (exact in bf16; yt8m_u8_frames_image), B3: x3 image of (4/255) W^T (yt8m_x3_split with
 * scale) -> three exact products per element pair.  rowscale / colsum NULL: plain product + bias. */
int yt8m_gemm_x1x3_nt(int64_t M, int64_t N, int64_t K, const void* A1, const void* B3, float* C, int64_t ldc, const float* bias,
                      const float* rowscale, const float* colsum, float colsum_scale, void* workspace, int64_t workspace_bytes,
                      yt8m_stream_t stream);
/* General form of the one-plane product: C[M,N] (+)= alpha * rowscale[m] * (A1 . B3^T + colsum_scale * colsum[n]) + bias[n];
 * rowscale / colsum may be NULL independently.  ska / skb: 16-wide K blocks between consecutive 32-row groups of either image
 * (0: the image is exactly K wide) -- a product may read a K RANGE of a larger image (A1 / B3 then point at the first block of the
 * range and K % 16 == 0).  The same convention holds for yt8m_gemm_problem.lda / .ldb in yt8m_gemm_x3_nt_grouped.
 * Use: the layer-0 weight gradient of the recurrent models on raw uint8 frames (W/readers.py:178-187 folded into the gradient
 * of W/all_frame_models/lstm_model.py:34-47): dW_x = alpha ((q - 128)^T . (r (.) dz) + (beta / alpha) colsum(r (.) dz)). */
int yt8m_gemm_x1x3_nt_ex(int64_t M, int64_t N, int64_t K, const void* A1, int64_t ska, const void* B3, int64_t skb, float* C,
                         int64_t ldc, const float* bias, float alpha, const float* rowscale, const float* colsum,
                         float colsum_scale, float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* bf16 operands as ONE-plane images (BASELINE configs[4] / --compute_dtype=bfloat16): yt8m_bf16_image rounds an fp32 matrix to
 * bfloat16 (nearest even) straight into the image layout ([rows / 32][K / 16][32 rows][2 halves][8] bf16 = yt8m_x3_image_bytes / 3
 * bytes; plain: rows = R, K = C; trans: rows = C, K = R) and yt8m_gemm_b1_nt_grouped multiplies two such images, C (+)= A . B^T
 * (+ bias), fp32 accumulate / output.  A wave's LDS-DMA instruction on an image moves 1 KiB of consecutive memory; the row-major
 * bf16 kernel (yt8m_gemm_bf16_nt_grouped) is bound by operand delivery at 64-128 bytes per row.  Problems as in
 * yt8m_gemm_x3_nt_grouped (lda / ldb = K-block strides, 0 = exact). */
int yt8m_bf16_image(const float* src, int64_t R, int64_t C, int64_t ld, float scale, void* plain, void* trans, yt8m_stream_t stream);
int yt8m_gemm_b1_nt_grouped(int nprob, const yt8m_gemm_problem* probs, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* Round 6: yt8m_gemm_b1_nt_grouped whose outputs flagged in c_bf16_mask (bit i = problem i) are bf16 matrices -- C points
 * at bf16 elements, ldc counts them (ldc % 4 == 0, C 16-byte aligned), beta must be 0; accumulation stays fp32, one rounding to nearest even
 * after the bias.  The MoE logits of the bf16 configuration (W/all_video_models/moe_model.py:54-64 under --compute_dtype=bfloat16): half
 * the bytes written by the product and read by yt8m_moe_mix_fwd_bf16z / yt8m_moe_mix_bwd_bf16_images_z16 behind it.  The host mirror takes
 * this form only with YT8M_Z16_LOGITS=1: -3 % on the configs[4] step, but one full-size gradient checksum of that configuration leaves the
 * golden replay's bf16 tolerance (DESIGN_LOG 11.7). */
int yt8m_gemm_b1_nt_grouped_bf16c(int nprob, const yt8m_gemm_problem* probs, unsigned c_bf16_mask, void* workspace, int64_t workspace_bytes,
                                  yt8m_stream_t stream);
/* p[B,V] = sum_m softmax(Zg[b,l,:])[m] sigmoid(Ze[b,l,m]) on bf16 logits Zg [B,3V], Ze [B,2V] (M == 2; B V % 4 == 0; 16-byte aligned). */
int yt8m_moe_mix_fwd_bf16z(const void* Zg, const void* Ze, float* p, int64_t B, int64_t V, int M, yt8m_stream_t stream);
/* yt8m_moe_mix_bwd_bf16_images reading bf16 logits. */
int yt8m_moe_mix_bwd_bf16_images_z16(const void* Zg, const void* Ze, const float* dp, const void* labels, int label_dtype, int64_t B,
                                     int64_t V, int M, float eps, float dscale, const float* upstream_dev, void* dZg_img, int64_t g_kb,
                                     void* dZg_t_img, int64_t gt_kb, void* dZe_img, int64_t e_kb, void* dZe_t_img, int64_t et_kb,
                                     float* be_part, yt8m_stream_t stream);
/* one-plane forms of yt8m_x3_split_colsum and yt8m_gemm_x1x3_nt_ex (the recurrent stack in bf16-operand mode: hoisted products on
 * bf16 roundings of their operands, the recurrence itself stays fp32-grade) */
int yt8m_bf16_image_colsum(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const float* rowscale, void* plain,
                           void* trans, void* trans_scaled, float* colpart, float* colpart_scaled, yt8m_stream_t stream);
int yt8m_gemm_b1_nt_ex(int64_t M, int64_t N, int64_t K, const void* A1, int64_t ska, const void* B1, int64_t skb, float* C, int64_t ldc,
                       const float* bias, float alpha, const float* rowscale, const float* colsum, float colsum_scale, float beta,
                       void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* yt8m_x3_split with a third output from the same pass: trans_scaled = x3 image of (diag(rowscale) . scale . src)^T
 * ([C rows, K = R]; rowscale [R]).  Any image may be NULL; rowscale and trans_scaled come together. */
int yt8m_x3_split_ex(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const float* rowscale, void* plain,
                     void* trans, void* trans_scaled, yt8m_stream_t stream);
/* the same pass with per-tile column sums: colpart / colpart_scaled (either may be NULL; [ceil(R / 64), C] floats each) receive the
 * sums over the 64-row tiles of scale * src, plain and rowscale-weighted.  The recurrent stack takes the bias gradient colsum(dz)
 * (BasicLSTMCell's biases, W/all_frame_models/lstm_model.py:34-40) and the rank-1 remainder colsum(r (.) dz) of the uint8 layer-0
 * weight gradient from the pass that writes dz's operand images: one yt8m_colsum_f32 over the partial matrix finishes them. */
int yt8m_x3_split_colsum(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const float* rowscale, void* plain,
                         void* trans, void* trans_scaled, float* colpart, float* colpart_scaled, yt8m_stream_t stream);
/* fp32 [rows, cols] (row stride ld) -> bf16 (round to nearest even); transpose != 0 writes dst as [cols, rows].
 * dst_ld: row stride of dst in bf16 elements (0 = dense).  The training path pads it to a multiple of 8 so that every bf16
 * row starts 16-byte aligned and the GEMMs stay on their LDS-DMA path (V*(M+1) = 14148 is not a multiple of 8). */
int yt8m_cast_f32_bf16(const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst, int64_t dst_ld, int transpose,
                       yt8m_stream_t stream);
/* both layouts from ONE pass over the fp32 source: dst_plain [rows, cols] and dst_trans [cols, rows] (8 B/element of HBM
 * traffic instead of 12).  Operands that are needed K-contiguous in one product and N-contiguous in another (dZ for dx and
 * dW; W for the forward and dx; x for the forward and dW) are cast once. */
int yt8m_cast_f32_bf16_dual(const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst_plain, int64_t plain_ld,
                            void* dst_trans, int64_t trans_ld, yt8m_stream_t stream);

/* ---- input transform ---------------------------------------------------------------------------
 * yt8m_l2norm_*: tf.nn.l2_normalize on the last axis (W/all_feature_transform/default_transformer.py:4-8,
 * deep_combine_chain_model.py:44): y = x * rsqrt(max(sum x^2, eps)).  bwd per SURVEY.md Appendix G. */
int yt8m_l2norm_fwd_f32(const float* x, float* y, int64_t rows, int64_t cols, float eps, yt8m_stream_t stream);
int yt8m_l2norm_bwd_f32(const float* x, const float* dy, float* dx, int64_t rows, int64_t cols, float eps,
                        yt8m_stream_t stream);
/* yt8m_dequant_l2norm_u8: readers.py:178-187 + utils.py:23-38 + default_transformer.py:7 in one pass over
 * the raw uint8 frame block q[B,F,D]: x = l2norm(q*(4/255) + (4/512-2)); rows f >= num_frames[b] are 0.
 * num_frames may be NULL (all F frames valid). */
int yt8m_dequant_l2norm_u8(const uint8_t* q, const int32_t* num_frames, float* x,
                           int64_t B, int64_t F, int64_t D, float eps, yt8m_stream_t stream);
/* video-level: mean over valid frames of the dequantised features (readers.py:69-71 "average of
 * dequantized values") then L2-normalise: q[B,F,D] -> x[B,D] */
int yt8m_dequant_mean_l2norm_u8(const uint8_t* q, const int32_t* num_frames, float* x,
                                int64_t B, int64_t F, int64_t D, float eps, yt8m_stream_t stream);

/* ---- MoE head (W/all_video_models/moe_model.py:54-64) ------------------------------------------
 * Zg [B, V*(M+1)] gate logits (label-major, mixture-minor; gate M = dummy expert), Ze [B, V*M] expert
 * logits (bias already added).  p[b,l] = sum_{m<M} softmax(Zg[b,l,:])[m] * sigmoid(Ze[b,l,m]). */
int yt8m_moe_mix_fwd(const float* Zg, const float* Ze, float* p, int64_t B, int64_t V, int M,
                     yt8m_stream_t stream);
/* in-place backward: Zg <- dL/dZg, Ze <- dL/dZe given dp = dL/dp [B,V].
 * (the expert-bias gradient is yt8m_colsum_f32 of the returned dL/dZe) */
int yt8m_moe_mix_bwd(float* Zg, float* Ze, const float* dp, int64_t B, int64_t V, int M,
                     yt8m_stream_t stream);

/* MoE mixing fused with CrossEntropyLoss (W/losses.py:110-130 applied to MoeModel's output; the reference lets a
 * model return its own "loss", W/train.py:384-385): fwd writes p AND loss = mean_b sum_l CE(p, y) in one pass over Z;
 * bwd forms dL/dp from the recomputed p and the labels in registers and overwrites Zg/Ze with dL/dZ (no dp tensor).
 * workspace >= yt8m_moe_mix_xent_workspace_bytes(B, V).  upstream_dev: device float[1] or NULL. */
int64_t yt8m_moe_mix_xent_workspace_bytes(int64_t B, int64_t V);
int yt8m_moe_mix_xent_fwd(const float* Zg, const float* Ze, const void* labels, int label_dtype, float* p,
                          float* loss_out, int64_t B, int64_t V, int M, float eps, void* workspace,
                          yt8m_stream_t stream);
int yt8m_moe_mix_xent_bwd(float* Zg, float* Ze, const void* labels, int label_dtype, const float* upstream_dev,
                          int64_t B, int64_t V, int M, float eps, float upstream, yt8m_stream_t stream);
/* yt8m_moe_mix_xent_bwd that also leaves max |dL/dZg| and max |dL/dZe| as float bits in absmax2[0..1] (8 bytes, zeroed by the call): the
 * scale words of the weight-gradient products' h2 operands (yt8m_gemm_auto_grouped_ex), measured while the gradients are written. */
int yt8m_moe_mix_xent_bwd_absmax(float* Zg, float* Ze, const void* labels, int label_dtype, const float* upstream_dev, int64_t B, int64_t V,
                                 int M, float eps, float upstream, void* absmax2, yt8m_stream_t stream);

/* ---- whole-head entry points (yt8m_moe_fwd / yt8m_moe_bwd / yt8m_logistic_fwd_bwd) ---------------
 * MoeModel.create_model (moe_model.py:12-65) + CrossEntropyLoss (losses.py:110-130) in two calls for a non-Python host:
 *   fwd: Zg = x.Wg, Ze = x.We + be (ONE persistent grouped GEMM launch) -> p [B,V] (+ loss when labels != NULL).
 *        Zg [B,V(M+1)] / Ze [B,VM] are caller-owned and must be handed to the backward unchanged.
 *   bwd: Zg/Ze <- dL/dZ in place, dWg = x^T dZg, dWe = x^T dZe (one grouped launch), dbe = colsum(dZe), dx optional;
 *        beta 0 overwrites / 1 accumulates the three gradients; upstream scales dL (e.g. 1 - support_loss_percent).
 * x [B,D], Wg [D,V(M+1)], We [D,VM], be [VM] row-major fp32; labels uint8 / float32 [B,V]; loss_out device float[1].
 * workspace >= yt8m_moe_workspace_bytes(B, V).  LogisticModel (logistic_model.py:12-26): p = sigmoid(x W + b), same
 * loss; Z [B,V] is scratch (dL/dz); labels == NULL -> forward only; workspace >= yt8m_moe_workspace_bytes(B, V). */
int64_t yt8m_moe_workspace_bytes(int64_t B, int64_t V);
/* ... plus room for the operand images of the head's three product stages on the bf16 pipe (csrc/gemm_auto.hip); with only
 * yt8m_moe_workspace_bytes the head's products stay on the fp32-MFMA kernel */
int64_t yt8m_moe_workspace_bytes_ex(int64_t B, int64_t D, int64_t V, int M);
int yt8m_moe_fwd(const float* x, const float* Wg, const float* We, const float* be, const void* labels, int label_dtype,
                 int64_t B, int64_t D, int64_t V, int M, float eps, float* Zg, float* Ze, float* p, float* loss_out,
                 void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
int yt8m_moe_bwd(const float* x, const float* Wg, const float* We, float* Zg, float* Ze, const void* labels, int label_dtype,
                 int64_t B, int64_t D, int64_t V, int M, float eps, float upstream, float* dWg, float* dWe, float* dbe,
                 float beta, float* dx, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
int yt8m_logistic_fwd_bwd(const float* x, const float* W, const float* b, const void* labels, int label_dtype, int64_t B,
                          int64_t D, int64_t V, float eps, float* p, float* loss_out, float* Z, float* dW, float* db,
                          float beta, float* dx, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);

/* ---- fully-connected layers with N <= 16 outputs over very many rows (csrc/gemm_skinny.hip) ------------------------
 * The attention logits of W/all_frame_models/lstm_attention_max_pooling_model.py:51-56 (slim.fully_connected on
 * [B*300, 1152+1024] -> 8).  Vector-ALU streaming kernels at the HBM rate instead of 94 %-padded MFMA tiles.  Row-major
 * fp32, K % 4 == 0, x / dx 16-byte aligned with ld % 4 == 0, beta in {0, 1}.
 *   fwd: y[M,N] (+)= x[M,K] . W[K,N] (+ bias)           (needs yt8m_skinny_supported(M,K,N): weights resident in LDS)
 *   dw : dW[K,N] (+)= x^T . dy   (workspace >= yt8m_skinny_workspace_bytes(); deterministic two-stage reduction)
 *   dx : dx[M,K] (+)= dy . W^T */
int yt8m_skinny_supported(int64_t M, int64_t K, int64_t N);
int64_t yt8m_skinny_workspace_bytes(int64_t M, int64_t K, int64_t N);
int yt8m_skinny_fwd_f32(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* y, int64_t ldy,
                        int64_t M, int64_t K, int64_t N, float beta, yt8m_stream_t stream);
int yt8m_skinny_dw_f32(const float* x, int64_t ldx, const float* dy, int64_t ldy, float* dW, int64_t lddw, int64_t M, int64_t K,
                       int64_t N, float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
int yt8m_skinny_dx_f32(const float* dy, int64_t ldy, const float* W, int64_t ldw, float* dx, int64_t lddx, int64_t M, int64_t K,
                       int64_t N, float beta, yt8m_stream_t stream);

/* BasicLSTM recurrence with bf16 operands for the recurrent product (csrc/lstm_bf16.hip; --compute_dtype=bfloat16): h_{t-1} /
 * dz_t and the packed W_h enter the matrix cores as bf16, accumulation, state, gates and gradients stay fp32.  H % 256 == 0
 * (yt8m_lstm_packed16_elems() > 0).  hs16 [F+1,B,H] bf16 mirrors hs (row t0 must hold the bf16 copy of hs[t0]: zeros at t0 = 0,
 * afterwards written by the previous chunk); dz16 [F,B,4H] bf16 is scratch that receives the bf16 copy of dz.  Other arguments
 * as yt8m_lstm_steps_fwd / _bwd. */
int64_t yt8m_lstm_packed16_elems(int64_t B, int64_t H);
int yt8m_lstm_pack_bf16(const float* Wh, int64_t ldw, int64_t H, void* Wp16, void* Wq16, yt8m_stream_t stream);
int yt8m_lstm_steps_fwd_bf16(float* z, const void* Wp16, float* cs, float* hs, void* hs16, float* out, const int32_t* num_frames,
                             int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias, yt8m_stream_t stream);
int yt8m_lstm_steps_bwd_bf16(const float* gates, const void* Wq16, const float* cs, const float* dout, float* dz, void* dz16,
                             float* work, int phase, const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H,
                             yt8m_stream_t stream);

/* attention pooling over the frame axis (lstm_attention_max_pooling_model.py:63, einsum "ijk,ijl->ikl") on the same streaming
 * kernels: w [B,F,A] (A <= 16), x [B,F,H] (H % 4 == 0), C [B,A,H] = w^T . x per video; dense row-major fp32.
 * bwd: dw [B,F,A] = x . dC^T and / or dx [B,F,H] = w . dC (either may be NULL). */
int yt8m_attn_pool_supported(int64_t B, int64_t F, int64_t A, int64_t H);
int yt8m_attn_pool_fwd(const float* w, const float* x, float* C, int64_t B, int64_t F, int64_t A, int64_t H, yt8m_stream_t stream);
int yt8m_attn_pool_bwd(const float* w, const float* x, const float* dC, float* dw, float* dx, int64_t B, int64_t F, int64_t A,
                       int64_t H, yt8m_stream_t stream);

/* The same layers straight from the reader's RAW uint8 frames (W/readers.py:178-187 hands uint8; W/utils.py:23-38 Dequantize and
 * W/all_feature_transform/default_transformer.py:4-8 l2-normalise + mask are folded in: no fp32 [B,F,D] tensor is written):
 *   x = diag(rs) (a0 q + c0 1 1^T),  a0 = 4/255, c0 = 4/512 - 2,  rs = yt8m_u8_frame_scales (1 / ||a0 q + c0||, 0 for padding frames)
 *   yt8m_skinny_fwd_u8    y[M,N] (+)= rs (.) (a0 q.W + c0 colsum_w) (+ bias)       colsum_w [N] = column sums of W[0:K]
 *   yt8m_skinny_dw_u8     dW[K,N] (+)= a0 q^T (rs (.) dy) + c0 1 (x) colsum(rs (.) dy)
 *   yt8m_attn_pool_fwd_u8 C[b] = (w[b] (.) rs[b])^T (a0 q[b] + c0)                 q [B,F,H] uint8, w [B,F,A], rs [B,F], C [B,A,H]
 *   yt8m_attn_pool_dw_u8  dw[b,f,a] = rs[b,f] (a0 q[b,f,:].dC[b,a,:] + c0 dCsum[b,a]),  dCsum [B,A] = sum_h dC
 * q rows 4-byte aligned, row stride % 4 == 0; rs may be NULL (= 1).  There is no dx: the frames are the input. */
int yt8m_u8_frame_scales(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, float eps, float* rs,
                         yt8m_stream_t stream);
int yt8m_skinny_fwd_u8(const uint8_t* q, int64_t ldq, const float* W, int64_t ldw, const float* bias, const float* rs,
                       const float* colsum_w, float* y, int64_t ldy, int64_t M, int64_t K, int64_t N, float beta, yt8m_stream_t stream);
int yt8m_skinny_dw_u8(const uint8_t* q, int64_t ldq, const float* dy, int64_t ldy, const float* rs, float* dW, int64_t lddw, int64_t M,
                      int64_t K, int64_t N, float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
int yt8m_attn_pool_fwd_u8(const float* w, const uint8_t* q, const float* rs, float* C, int64_t B, int64_t F, int64_t A, int64_t H,
                          yt8m_stream_t stream);
int yt8m_attn_pool_dw_u8(const uint8_t* q, const float* rs, const float* dC, const float* dCsum, float* dw, int64_t B, int64_t F,
                         int64_t A, int64_t H, yt8m_stream_t stream);

/* ---- GRUCell / LayerNormBasicLSTMCell layers (csrc/cells.hip), time-major, generic per-step form --------------------
 * tf.contrib.rnn.GRUCell under tf.nn.dynamic_rnn (W/all_frame_models/gru_pooling_model.py:34-47):
 *   zg [F,B,2H]: in = x.Wg[:in] + b_gates (hoisted by the caller), out = the gates r | u;
 *   zc [F,B,H] : in = x.Wc[:in] + b_cand, out = the candidate c;   Wg_h = Wg[in:] [H,2H], Wc_h = Wc[in:] [H,H];
 *   hs [F+1,B,H] with hs[0] = initial state (caller), rh [F,B,H] receives r*h (saved for the weight gradient), out [F,B,H]
 *   (optional) the dynamic_rnn outputs (zero past num_frames; state copied through).
 * Backward: dout [F,B,H] (optional), dh_final [B,H] (optional) -> dzg [F,B,2H], dzc [F,B,H]; work >= 3*B*H floats; the final
 * dh of step 0 is left in work[(F % 2) * B*H].  Weight / input gradients are hoisted GEMMs over dzg / dzc (caller). */
int yt8m_gru_layer_fwd(float* zg, float* zc, const float* Wg_h, int64_t ldg, const float* Wc_h, int64_t ldc, float* hs, float* rh,
                       float* out, const int32_t* num_frames, int64_t F, int64_t B, int64_t H, void* gemm_workspace,
                       int64_t gemm_workspace_bytes, yt8m_stream_t stream);
int yt8m_gru_layer_bwd(const float* zg, const float* zc, const float* Wg_h, int64_t ldg, const float* Wc_h, int64_t ldc,
                       const float* hs, const float* dout, const float* dh_final, float* dzg, float* dzc, float* work,
                       const int32_t* num_frames, int64_t F, int64_t B, int64_t H, void* gemm_workspace,
                       int64_t gemm_workspace_bytes, yt8m_stream_t stream);
/* tf.contrib.rnn.LayerNormBasicLSTMCell(H, forget_bias, dropout_keep_prob) (W/all_frame_models/layernorm_lstm_memory_model.py:37-50):
 *   z [F,B,4H]: in = x.W[:in] (no bias), stays RAW (+ h.Wh) for the backward pass; gamma / beta [5,H] in the order
 *   input, transform, forget, output, state; stats [F,B,10] receives (mean, rstd) of the five normalisations; cs / hs
 *   [F+1,B,H] with row 0 = initial state; keep_prob < 1 drops the candidate g (Philox element (t*B+b)*H+h under `seed`).
 * Backward writes dz [F,B,4H] (w.r.t. the raw pre-activations), dyb / dyg [F,B,5H] (column sums = dbeta / dgamma);
 * work >= 4*B*H floats.  H <= 2048. */
int yt8m_lnlstm_layer_fwd(float* z, const float* Wh, int64_t ldw, const float* gamma, const float* beta, float* stats, float* cs,
                          float* hs, float* out, const int32_t* num_frames, int64_t F, int64_t B, int64_t H, float forget_bias,
                          float keep_prob, uint64_t seed, void* gemm_workspace, int64_t gemm_workspace_bytes,
                          yt8m_stream_t stream);
int yt8m_lnlstm_layer_bwd(const float* z, const float* Wh, int64_t ldw, const float* gamma, const float* beta, const float* stats,
                          const float* cs, const float* dout, const float* dc_final, const float* dh_final, float* dz, float* dyb,
                          float* dyg, float* work, const int32_t* num_frames, int64_t F, int64_t B, int64_t H, float forget_bias,
                          float keep_prob, uint64_t seed, void* gemm_workspace, int64_t gemm_workspace_bytes,
                          yt8m_stream_t stream);

/* ---- counter-based random elementwise ops (csrc/random.hip) -------------------------------------
 * Philox4x32-10: element e of the logical tensor uses word (e & 3) of the block with counter (e >> 2), key = seed;
 * `offset` = logical index of x[0], so a chunk of a tensor draws the same numbers as a call over the whole tensor and the
 * backward pass regenerates the mask from (seed, offset) instead of storing it.
 * yt8m_dropout_f32: tf.nn.dropout (W/all_video_models/deep_combine_chain_model.py:57-58; DropoutWrapper input_keep_prob,
 *   W/all_frame_models/lstm_memory_model.py:36-45): y = floor(keep_prob + u) ? x / keep_prob : 0, u = (word >> 8) * 2^-24.
 *   In place (y == x) allowed; its own backward is the same call on the upstream gradient.
 * yt8m_add_noise_f32: y = x + stddev * N(0,1) (W/all_frame_models/lstm_memory_model.py:62-63), Box-Muller on word pairs. */
int yt8m_dropout_f32(const float* x, float* y, int64_t n, float keep_prob, uint64_t seed, int64_t offset, yt8m_stream_t stream);
int yt8m_add_noise_f32(const float* x, float* y, int64_t n, float stddev, uint64_t seed, int64_t offset, yt8m_stream_t stream);

/* MoE mixing backward straight into bf16 GEMM operands (csrc/moe_bf16.hip; --compute_dtype=bfloat16, num_mixtures == 2): one pass
 * over the fp32 logits Zg [B, 3V] / Ze [B, 2V] (dense) and either dL/dp [B,V] (dp) or the labels (CrossEntropyLoss fused, as
 * yt8m_moe_mix_xent_bwd: eps, dscale * upstream_dev[0]) writes dL/dZ as bf16 in both layouts -- dZg_b [B, 3V] (pitch gb_ld),
 * dZg_t [3V, B] (gt_ld), dZe_b [B, 2V] (eb_ld), dZe_t [2V, B] (et_ld) -- and be_part [yt8m_moe_mix_bwd_bf16_partial_rows(B), 2V]:
 * column sums of dZe per 64-row block (their column sum is the expert-bias gradient).  The fp32 logits are left untouched. */
int64_t yt8m_moe_mix_bwd_bf16_partial_rows(int64_t B);
int yt8m_moe_mix_bwd_bf16(const float* Zg, const float* Ze, const float* dp, const void* labels, int label_dtype, int64_t B, int64_t V,
                          int M, float eps, float dscale, const float* upstream_dev, void* dZg_b, int64_t gb_ld, void* dZg_t,
                          int64_t gt_ld, void* dZe_b, int64_t eb_ld, void* dZe_t, int64_t et_ld, float* be_part,
                          yt8m_stream_t stream);
/* the same pass with the four bf16 outputs as ONE-plane operand images of yt8m_gemm_b1_nt_grouped (plain images: rows = B,
 * K = V (M+1) resp. V M; transposed: rows = V (M+1) resp. V M, K = B); *_kb = K-block counts ceil(K / 16) of the images */
int yt8m_moe_mix_bwd_bf16_images(const float* Zg, const float* Ze, const float* dp, const void* labels, int label_dtype,
                                 int64_t B, int64_t V, int M, float eps, float dscale, const float* upstream_dev,
                                 void* dZg_img, int64_t g_kb, void* dZg_t_img, int64_t gt_kb, void* dZe_img, int64_t e_kb,
                                 void* dZe_t_img, int64_t et_kb, float* be_part, yt8m_stream_t stream);

/* ---- elementwise activations + column sums (bias gradients) ------------------------------------ */
enum yt8m_act { YT8M_ACT_SIGMOID = 0, YT8M_ACT_RELU = 1, YT8M_ACT_RELU6 = 2, YT8M_ACT_TANH = 3, YT8M_ACT_ELU = 4 };
int yt8m_act_fwd_f32(int act, const float* x, float* y, int64_t n, yt8m_stream_t stream);
/* dx = dy * act'(.) expressed from the OUTPUT y (sigmoid/tanh/relu/relu6/elu all admit it) */
int yt8m_act_bwd_f32(int act, const float* y, const float* dy, float* dx, int64_t n, yt8m_stream_t stream);
/* out[n] (beta=0) or out[n] += (beta=1): sum over rows of X[rows, cols]; deterministic (fixed summation order).
 * workspace (optional, may be NULL): >= yt8m_colsum_workspace_bytes() device bytes let tall-and-narrow inputs
 * ([B*F, 8..64] attention / cluster logits) be split over rows so that the whole chip is used. */
int64_t yt8m_colsum_workspace_bytes(int64_t rows, int64_t cols);
int yt8m_colsum_f32(const float* X, int64_t rows, int64_t cols, int64_t ldx, float* out, float beta, void* workspace,
                    int64_t workspace_bytes, yt8m_stream_t stream);
/* Two column sums from one pass: out[c] (+)= sum_r X[r][c] (beta 0 / 1) and out_weighted[c] = sum_r row_weights[r] X[r][c]
 * (overwritten).  workspace: 2 * yt8m_colsum_workspace_bytes(rows, cols), may be NULL. */
int yt8m_colsum_weighted_f32(const float* X, int64_t rows, int64_t cols, int64_t ldx, const float* row_weights, float* out, float beta,
                             float* out_weighted, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* C[r][c] += scale * v[c] for every row r (cols % 4 == 0): the rank-1 remainder beta * 1 (x) colsum(r (.) dz) of the layer-0 weight
 * gradient on uint8 frames, applied once per step */
int yt8m_rank1_add_rows_f32(float* C, int64_t rows, int64_t cols, int64_t ldc, const float* v, float scale, yt8m_stream_t stream);

/* ---- loss: CrossEntropyLoss (W/losses.py:110-130), probability space, eps = 1e-5 ---------------
 * loss = mean_b sum_l -[y log(p+eps) + (1-y) log(1-p+eps)] * (w_b);  dp = dloss/dp * upstream.
 * labels: uint8 {0,1} or float32 (smoothed labels, losses.py:46-54).  weights [B] optional (NULL).
 * loss_out: device float[1].  dp may be NULL (eval).  workspace: device, >= yt8m_xent_workspace_bytes(). */
int64_t yt8m_xent_workspace_bytes(int64_t B, int64_t V);
int yt8m_xent_fwd_bwd(const float* p, const void* labels, int label_dtype, const float* weights,
                      float* loss_out, float* dp, int64_t B, int64_t V, float eps, float upstream,
                      void* workspace, yt8m_stream_t stream);

/* backward only: dp = dL/dp * upstream * (upstream_dev ? *upstream_dev : 1)  (upstream_dev: device float[1]) */
int yt8m_xent_bwd(const float* p, const void* labels, int label_dtype, const float* weights,
                  const float* upstream_dev, float* dp, int64_t B, int64_t V, float eps, float upstream,
                  yt8m_stream_t stream);

/* ---- optimiser slice of build_graph (W/train.py:435-466, W/utils.py:164-174, tf.train.AdamOptimizer)
 * The parameters live in one flat fp32 arena; `chunks` (device, int32[nchunks*4]) describes it as
 * {offset, count, tensor_id, unused} with no chunk spanning two tensors; count <= 4096.
 * g_eff = g*gscale + l2[tensor]*w  (gscale = 1/world for the all-reduce mean; l2 = 1e-8 for
 * regularised weights, 0 otherwise: slim.l2_regularizer gradient).
 * Both calls may be given a SUB-RANGE of the chunk table (chunks pointer advanced, nchunks = range length) covering
 * the tensors [tensor_base, tensor_base + ntensors): the data-parallel step updates each gradient bucket as soon as
 * its all-reduce has landed, while later buckets are still on the wire.
 * sqnorm: norms[tensor] = sum g_eff^2  (deterministic two-stage reduction; partial: float[nchunks]).
 *         tensor_chunk_start (optional, device int32[all tensors + 1], ABSOLUTE chunk indices; chunk_base = absolute
 *         index of chunks[0]) lets the second stage read only its own partials instead of scanning the chunk table.
 * adam:   g_c = g_eff * clip/max(sqrt(norms[t]), clip) (clip<=0: no clipping);
 *         m = b1 m + (1-b1) g_c; v = b2 v + (1-b2) g_c^2; w -= lr_t * m / (sqrt(v) + eps)   [TF-1 form] */
int yt8m_sqnorm_multi(const float* w, const float* g, const int32_t* chunks, int64_t nchunks,
                      const float* l2, float gscale, float* partial, float* norms, int64_t tensor_base,
                      int64_t ntensors, const int32_t* tensor_chunk_start, int64_t chunk_base,
                      yt8m_stream_t stream);
/* tf.clip_by_norm of ONE tensor, the reference's own helper (W/utils.py:164-174 clip_gradient_norms): out = g * max_norm / max(||g||,
 * max_norm); out may alias g; workspace: 256 floats.  For hosts that keep their own gradient list -- the training step clips inside the
 * fused optimiser pass (yt8m_sqnorm_multi + yt8m_adam_multi_ex / yt8m_optimizer_ranges). */
int yt8m_clip_by_norm_f32(const float* g, float* out, int64_t n, float max_norm, float* workspace, yt8m_stream_t stream);
int yt8m_adam_multi(float* w, float* m, float* v, const float* g, const int32_t* chunks, int64_t nchunks,
                    const float* l2, float gscale, const float* norms, float clip,
                    float lr_t, float beta1, float beta2, float eps, yt8m_stream_t stream);

/* ---- Resident operand images of the weight matrices, kept current by the optimiser pass (csrc/wimg.hip, csrc/optim.hip; round 5).
 * Replaces: the per-step re-split of every weight matrix in front of its products -- in the reference the variable is read by
 * tf.matmul directly (W/all_video_models/moe_model.py:40-52, W/all_frame_models/lstm_model.py:34-47) and changes only in
 * tf.train.AdamOptimizer.apply_gradients (W/train.py:459-466); here the operand image (yt8m_x3_split / yt8m_bf16_image format) is the
 * form the matrix pipe reads, so it is rewritten exactly where the weight is.
 *
 * Registry (host side, process wide, thread safe): (src, R, C, ld, trans, planes, scale) -> image.  trans = 0: the image of
 * src[R, C] as an [R rows, K = C] operand ("plain" of yt8m_x3_split); trans = 1: of its transpose ([C rows, K = R]).  planes = 3
 * (x3 image) or 1 (bf16 image).  yt8m_gemm_auto_grouped, yt8m_lstm_stack_fwd / _bwd and the Python helpers look a weight operand up
 * before splitting it; a miss falls back to the split they always did.  The OWNER of the source keeps the image valid: every write to
 * the source goes through yt8m_adam_tiles(do_adam = 1) or is followed by yt8m_adam_tiles(do_adam = 0) before the next product, and
 * yt8m_wimg_unregister covers the source's memory before it is released.
 * Demand recording: yt8m_x3_split / yt8m_bf16_image note every (src, shape, orientation, planes, scale) they are asked for when src lies
 * in a watched range (yt8m_wimg_watch), so the owner of a parameter arena learns which images a step of its model needs. */
typedef struct yt8m_wimg_demand {
  const float* src;
  int64_t R, C, ld;
  int32_t trans, planes;
  float scale;
  int32_t pad;
} yt8m_wimg_demand;
int yt8m_wimg_register(const float* src, int64_t R, int64_t C, int64_t ld, int trans, int planes, float scale, void* image);
int64_t yt8m_wimg_unregister(const void* lo, const void* hi);      /* sources in [lo, hi); NULL, NULL: all.  Returns the number dropped */
void* yt8m_wimg_lookup(const float* src, int64_t R, int64_t C, int64_t ld, int trans, int planes, float scale);   /* NULL: not resident */
int64_t yt8m_wimg_count(void);
int yt8m_wimg_watch(const void* lo, const void* hi, int on);       /* 1: note the demands on memory in [lo, hi); 0: stop + forget them */
int yt8m_wimg_note_demand(const float* src, int64_t R, int64_t C, int64_t ld, int trans, int planes, float scale);
int64_t yt8m_wimg_demands(yt8m_wimg_demand* out, int64_t max);      /* copies <= max entries, returns how many were recorded */
int64_t yt8m_wimg_demand_generation(const void* lo, const void* hi); /* demands EVER noted inside the watched range [lo, hi) (monotonic;
                                                                        -1: not watched): what ONE arena's owner compares, unmoved by
                                                                        other arenas' demands or by another owner that stops watching */
/* One matrix of the parameter arena and the images it owns.  offset: floats from the arena base (w, m, v, g share it) to the
 * row-major contiguous matrix [R, C]; tensor: its index into l2 / norms.  Each spec covers the row window [row0, row0 + rows)
 * (row0 % 64 == 0; rows % 64 == 0 or the window ends with the matrix -- the input rows [0, Din) of an LSTM weight [Din + H, 4H]):
 * plain = image of the window as an [rows, K = C] operand, trans = image of its transpose ([C rows, K = rows]); either may be NULL;
 * every element times `scale`; planes 3 or 1 -- or 2 (round 6): two IEEE-half planes ("h2") under the power of two that the word 256 BYTES
 * IN FRONT of the image gives (max |w| as float bits, the contract of yt8m_h2_absmax / yt8m_gemm_h2_nt_grouped's dsa / dsb): the caller
 * allocates [256-byte header | image] and passes the image pointer; header word 1 receives the maximum the tile pass measures and
 * yt8m_adam_tiles moves it into word 0 before its next pass (first build: put max |w| into word 1, call with do_adam = 0).  Such an image
 * is registered / looked up with scale 0.  tile_base is filled by yt8m_wimg_jobs_layout. */
typedef struct yt8m_wimg_spec {
  void* plain;
  void* trans;
  int64_t row0, rows;
  float scale;
  int32_t planes;
} yt8m_wimg_spec;
typedef struct yt8m_wimg_job {
  int64_t offset;
  int64_t R, C;
  int32_t tensor;
  int32_t nspec;                /* 0..4 (0: plain Adam on 64 x 64 tiles) */
  int64_t tile_base;
  yt8m_wimg_spec spec[4];
} yt8m_wimg_job;
int64_t yt8m_wimg_jobs_layout(yt8m_wimg_job* jobs_host, int64_t njobs);   /* validates, fills tile_base; returns total tiles (< 0: status) */
/* clip + TF-Adam of the matrices jobs[0 .. njobs) (DEVICE copy of a laid-out array, or a sub-range of one: tile0 = tile_base of its
 * first job, ntiles = the range's tile count) with the same per-element arithmetic as yt8m_adam_multi -- bitwise the same w, m, v --
 * and, in the same pass, the images of every spec.  do_adam = 0: images only (first build, refresh after a host-side write).
 * yt8m_adam_multi_ex = yt8m_adam_multi that leaves the tensors flagged in skip_tensor (device uint8[all tensors]) to this call. */
int yt8m_adam_tiles(float* w, float* m, float* v, const float* g, const yt8m_wimg_job* jobs, int64_t njobs, int64_t tile0,
                    int64_t ntiles, const float* l2, float gscale, const float* norms, float clip, float lr_t, float beta1,
                    float beta2, float eps, int do_adam, yt8m_stream_t stream);
int yt8m_adam_multi_ex(float* w, float* m, float* v, const float* g, const int32_t* chunks, int64_t nchunks,
                       const float* l2, float gscale, const float* norms, float clip, float lr_t, float beta1, float beta2,
                       float eps, const uint8_t* skip_tensor, yt8m_stream_t stream);

/* ---- BasicLSTMCell gate block (tf.contrib.rnn.BasicLSTMCell via W/all_frame_models/lstm_model.py:34-47)
 * z [B,4H] = pre-activations in the order i, j, f, o.  live[b] = (t < num_frames[b]) implements
 * dynamic_rnn's copy-through (Z/rnn_residual.py:61-188): dead rows keep (c,h) and emit out = 0.
 * fwd: c_new, h_new, out (may alias h_new when no dead rows matter; out may be NULL).
 * z is overwritten with the ACTIVATED gates (sigmoid(i), tanh(j), sigmoid(f+fb), sigmoid(o)) for bwd. */
int yt8m_lstm_gates_fwd(float* z, const float* c_prev, const float* h_prev, float* c_new, float* h_new,
                        float* out, const int32_t* num_frames, int32_t t, int64_t B, int64_t H,
                        float forget_bias, yt8m_stream_t stream);
/* bwd of one step.  gates = activated gates saved by fwd; c_prev, c_new saved.  dh, dc are the incoming
 * gradients wrt the carried state (h_new, c_new); dout (may be NULL) is the gradient wrt the emitted
 * output of this step (only live rows emitted h_new; dead rows emitted the constant 0).
 * Outputs: dz [B,4H] (pre-activation grads, 0 on dead rows), dc_prev, dh_prev (dead rows: dh flows
 * straight through; live rows get 0 here and receive dh_prev from dz . W^T by the caller's GEMM, beta=1). */
int yt8m_lstm_gates_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh,
                        const float* dc, const float* dout, float* dz, float* dc_prev, float* dh_prev,
                        const int32_t* num_frames, int32_t t, int64_t B, int64_t H, yt8m_stream_t stream);

/* One BasicLSTM layer over all F steps of tf.nn.dynamic_rnn, TIME-MAJOR buffers.
 * z [F,B,4H]: in = hoisted input projection x_t.W_x + b (one GEMM over all steps); out = activated gates.
 * Wh: the recurrent rows of the cell's "weights" variable ([H,4H] block, row stride ldw).
 * cs, hs [F+1,B,H]: state history, slot 0 = initial state (caller zero-fills), slot t+1 = state after step t.
 * out [F,B,H] (may be NULL): emitted outputs, 0 on dead rows.  The time loop runs inside the library
 * (one recurrent GEMM accumulate + one gate kernel per step).  gemm_workspace (>= yt8m_gemm_workspace_bytes(), may be
 * NULL) lets the small per-step GEMMs ([B,H] x [H,4H]: 32 tiles) be split along K across the whole chip. */
int yt8m_lstm_layer_fwd(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out,
                        const int32_t* num_frames, int64_t F, int64_t B, int64_t H, float forget_bias,
                        void* gemm_workspace, int64_t gemm_workspace_bytes, yt8m_stream_t stream);
/* BPTT of one layer.  gates [F,B,4H] from fwd; dout [F,B,H] or NULL; dc_final/dh_final [B,H] or NULL (gradient
 * wrt the final carried state).  Writes dz [F,B,4H] (pre-activation gradients; the caller turns them into
 * dW_x, dW_h, db, dX with four hoisted GEMMs/column sums).  work: device scratch of 4*B*H floats. */
int yt8m_lstm_layer_bwd(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout,
                        const float* dc_final, const float* dh_final, float* dz, float* work,
                        const int32_t* num_frames, int64_t F, int64_t B, int64_t H,
                        void* gemm_workspace, int64_t gemm_workspace_bytes, yt8m_stream_t stream);

/* ---- persistent recurrence (csrc/lstm_persist.hip): ONE launch runs steps [t0, t0+T) of a BasicLSTM layer --------
 * Replaces the per-step launches of yt8m_lstm_steps_fwd / _bwd (tf.nn.dynamic_rnn's while_loop body,
 * W/all_frame_models/lstm_model.py:44-47) when yt8m_lstm_persist_supported(B, H): W_h stays in registers for all T
 * steps, h_t / dz_t are exchanged between workgroups through `workspace` (MFMA-fragment order, write-through stores,
 * per-tile arrival counters; no grid barrier).  Same arguments and semantics as the time-range forms; `workspace`
 * (yt8m_lstm_persist_workspace_bytes) is scratch owned by the caller for the duration of the call chain of one layer.
 * Launches of one device are chained behind each other inside the library (whole-chip residency).
 * The first 128 bytes of `workspace` hold a STICKY error word: zero them once when the buffer is allocated; no launch ever
 * clears them.  A launch that gives up waiting (bounded spins) sets the word and later launches on the same workspace leave it
 * alone, so yt8m_lstm_persist_status -- which synchronises `stream`, returns YT8M_E_HIP if ANY launch on `workspace` since the
 * previous status call gave up, and then clears the word -- cannot miss a time-out of an earlier launch.
 * yt8m_lstm_persist_debug_fault marks `workspace` exactly as a timed-out launch would (test hook for that path). */
int yt8m_lstm_persist_supported(int64_t B, int64_t H);
int64_t yt8m_lstm_persist_workspace_bytes(int64_t B, int64_t H);
/* A workspace of this size gives every step of a launch of up to T steps its own exchange image; each state byte is then
 * written once and fetched with plain loads (one fabric read per XCD, the other CUs hit that XCD's L2) instead of sc0 sc1
 * loads that cross the fabric for every CU.  Launches pick the protocol from the workspace_bytes they are given. */
int64_t yt8m_lstm_persist_workspace_bytes_steps(int64_t B, int64_t H, int64_t T);
/* 1 when a forward launch on such a workspace runs the recurrent product h.W_h as six bf16 MFMA products of exact three-plane
 * splits (fp32-grade results, 3/8 of the fp32-MFMA matrix time; H in {512, 1024}, >= 2 tiles per workgroup), 0: fp32 MFMA */
int yt8m_lstm_persist_fwd_on_bf16_pipe(int64_t B, int64_t H);
/* CUs the following forward / backward launches may occupy (0: whole chip, -1: environment / default = whole chip forward, 128
 * backward).  Two forward launches of neighbouring layers run side by side when each takes half the chip. */
int yt8m_lstm_persist_set_cus(int fwd_cus, int bwd_cus);
/* CUs the library leaves out of its "do these persistent launches fit the chip together" arithmetic: a data-parallel host reserves
 * room for the RCCL kernels of the gradient all-reduce that run beside the backward pass (launches that no longer fit side by side
 * are chained instead of spinning on a partly resident grid).  Env default: YT8M_PERSIST_RESERVED_CUS, else 0. */
int yt8m_lstm_persist_reserve_cus(int cus, int* previous);
int yt8m_lstm_persist_status(const void* workspace, yt8m_stream_t stream);
int yt8m_lstm_persist_debug_fault(void* workspace, yt8m_stream_t stream);
/* diagnostics: persistent launches / workgroups since the last reset on the current device and how many workgroups did not run on
 * the XCD their block index suggests (a launch that finds CUs busy is placed wherever some are free: correct, but the state fetch
 * loses its L2 sharing).  Synchronises the device. */
int yt8m_lstm_persist_placement_stats(int64_t* launches, int64_t* workgroups, int64_t* off_xcd, int reset);
int yt8m_lstm_persist_fwd(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out,
                          const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias,
                          void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* --compute_dtype=bfloat16 (our variant flag): the recurrent product h_{t-1} . W_h of the same forward recurrence
 * (W/all_frame_models/lstm_model.py:34-47) on ONE bf16 plane -- h and W_h rounded to nearest even, fp32 accumulation.  Same arguments.
 * A permission: launches that cannot take the bf16-pipe kernel run the fp32 form. */
int yt8m_lstm_persist_fwd_bf16(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out,
                               const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias,
                               void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* backward: steps t0+T-1 down to t0; (gates, cs, dout, dz, work, phase) exactly as yt8m_lstm_steps_bwd.  The backward of
 * dynamic_rnn's while_loop (tf.gradients through W/all_frame_models/lstm_model.py:44-47).  dbias_rows (may be NULL):
 * [B,4H] running sums of dz over the processed steps, accumulated IN PLACE (zero it before the first chunk); the bias
 * gradient is its column sum -- saves the pass over the whole [F*B,4H] dz.
 * work (round 6): a persistent launch reads the running (dh, dc) from the half `phase` names and leaves them in the half of
 * (phase + T) % 2 when it ends; the other half is scratch -- with one 16-row tile per epilogue wave the steps in between carry the
 * state in registers and do not touch it. */
/* Round 5: yt8m_lstm_persist_fwd with the recurrent product h_{t-1} . W_h (BasicLSTMCell._linear under dynamic_rnn,
 * W/all_frame_models/lstm_model.py:34-47) as THREE f16 products of two-half-plane splits instead of six bf16 products of three-plane splits:
 * the same fp32 grade, half the matrix instructions, 4 instead of 6 exchanged bytes per state element.  wh_absmax: device word with
 * max |W_h| as float bits (yt8m_h2_absmax).  Taken where yt8m_lstm_persist_fwd_on_bf16_pipe(B, H) holds; elsewhere (or YT8M_PERSIST_FWD_H2=0)
 * the launch is yt8m_lstm_persist_fwd. */
int yt8m_lstm_persist_fwd_h2(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out, const int32_t* num_frames,
                             int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias, const void* wh_absmax, void* workspace,
                             int64_t workspace_bytes, yt8m_stream_t stream);
int yt8m_lstm_persist_bwd_supported(int64_t B, int64_t H);
int yt8m_lstm_persist_bwd(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                          float* work, int phase, float* dbias_rows, const int32_t* num_frames, int64_t t0, int64_t T,
                          int64_t B, int64_t H, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* --compute_dtype=bfloat16 (our variant flag; the reference is fp32 throughout): the recurrent product dz_t . W_h^T of the same backward
 * pass (W/all_frame_models/lstm_model.py:34-47 through tf.gradients) on ONE bf16 plane -- dz_t and W_h rounded to nearest even, fp32
 * accumulation, dz as stored stays fp32.  Same arguments.  A permission, not a demand: launches that cannot take the bf16 form (H
 * other than 512 / 1024, fewer than four 16-row tiles per workgroup, no room for one exchange image per step) run the fp32 form. */
int yt8m_lstm_persist_bwd_bf16(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                               float* work, int phase, float* dbias_rows, const int32_t* num_frames, int64_t t0, int64_t T,
                               int64_t B, int64_t H, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* Round 5: the fp32 configuration's recurrent product dz_t . W_h^T of the same backward pass (W/all_frame_models/lstm_model.py:34-47
 * through tf.gradients) OFF the fp32 matrix pipe, fp32-grade: dz_t and W_h^T as two IEEE-half planes each (x S = hi + lo to 2^-22), three
 * products hi.hi + lo.hi + hi.lo of v_mfma_f32_16x16x32_f16, fp32 accumulation; every producer workgroup scales its 16 rows x 64 values of
 * dz_t by one power of two per row and publishes the inverse exponents beside the tile (no scale is guessed).  Results differ from
 * yt8m_lstm_persist_bwd by rounding of the fp32 grade only (2^-21 relative per product term).  wh_absmax: device word with max |W_h| as
 * float bits (yt8m_h2_absmax over the [H, 4H] block W_h; zero the word first).  A permission like the bf16 form: H other than 512 / 1024,
 * tiles per workgroup not a multiple of four, or a workspace smaller than yt8m_lstm_persist_workspace_bytes_steps(B, H, T) -> the fp32
 * form runs.  yt8m_lstm_persist_bwd_on_f16_pipe: 1 when the shape takes the f16 form.  YT8M_PERSIST_BWD_H2=0: never. */
int yt8m_lstm_persist_bwd_h2(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                             float* work, int phase, float* dbias_rows, const int32_t* num_frames, int64_t t0, int64_t T,
                             int64_t B, int64_t H, const void* wh_absmax, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
int yt8m_lstm_persist_bwd_on_f16_pipe(int64_t B, int64_t H);
/* Round 6: tf.contrib.rnn.GRUCell under tf.nn.dynamic_rnn as ONE launch per (layer, time range) and direction on the persistent
 * recurrences' exchange protocol (csrc/gru_persist.inl; reference W/all_frame_models/gru_pooling_model.py:34-47 -- replaces the
 * 2 + 3 launches per time step of yt8m_gru_layer_fwd / _bwd).  A GRU step is two dependent products, so a step is two half-steps of the
 * exchange: the workspace (yt8m_gru_persist_workspace_bytes) holds one exchange image per half-step.
 *   fwd: zg [F,B,2H], zc [F,B,H] = hoisted input projections + biases on entry, activations r|u, c on exit; Wg_h / Wc_h = the recurrent
 *        rows of gates/weights [H, >= 2H] and candidate/weights [H, >= H]; hs [F+1,B,H] with hs[t0] given; rh [F,B,H] = r * h_{t-1};
 *        out [F,B,H] or NULL; rows with t >= num_frames[b] copy their state through and emit zeros.
 *   bwd: steps t0 + T - 1 .. t0; work [B,H] = dL/dh entering the range's last step (in) / dL/dh_{t0-1} (out); writes dzg, dzc.
 * Results equal the per-step entry points up to the K summation order of the recurrent products and the v_exp / v_rcp gate functions
 * (<= ~1.5e-7 absolute per activation).  yt8m_lstm_persist_status(workspace) reports a timed-out launch, as for the LSTM kernels.
 * Measured at B = 128, H = 1024 (profiles/r6_gru_persist.txt): forward 13.9 us/step against 17.6 for the per-step launches (the host
 * mirror takes it by default), backward 29.0 against 20.3 (opt-in: YT8M_GRU_PERSIST_BWD=1 in the host mirror). */
int yt8m_gru_persist_supported(int64_t B, int64_t H);
int64_t yt8m_gru_persist_workspace_bytes(int64_t B, int64_t H, int64_t T);
int yt8m_gru_persist_fwd(float* zg, float* zc, const float* Wg_h, int64_t ldg, const float* Wc_h, int64_t ldc, float* hs, float* rh,
                         float* out, const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, void* workspace,
                         int64_t workspace_bytes, yt8m_stream_t stream);
int yt8m_gru_persist_bwd(const float* zg, const float* zc, const float* Wg_h, int64_t ldg, const float* Wc_h, int64_t ldc,
                         const float* hs, const float* dout, float* dzg, float* dzc, float* work, const int32_t* num_frames, int64_t t0,
                         int64_t T, int64_t B, int64_t H, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* Round 6: the f16 backward recurrence as K-SPLIT WORKGROUP PAIRS -- the two workgroups that share a line of gates each reduce HALF of
 * K for the pair's 32 units (half the dz a CU draws per step) and hand the partner its partial tile through tagged 8-byte granules.
 * Opt-in (stand-alone 13.7 vs 14.4 us/step, but slower inside the headline step where two recurrences and the dW products share the
 * chip: profiles/r6_recur_ab.txt block 4); available wherever the f16 form runs with <= 8 tiles per workgroup; results equal the unpaired form to fp32 rounding (one more
 * level in the fixed summation tree).  mode -1: environment (YT8M_PERSIST_BWD_PAIR, default 0), 0: off, 1: on.  Process-wide. */
int yt8m_lstm_persist_set_pair(int mode);
/* yt8m_lstm_persist_bwd (wh_absmax NULL) / yt8m_lstm_persist_bwd_h2 that also measures, while it writes dz, what the products after it
 * would otherwise measure in passes over dz (105-210 MB each at the headline shape): rowmax[t B + b] = max |dz[t, b, :]| as float bits ([F B]
 * words by absolute frame row, zeroed by the caller; the operand of yt8m_h2_split_rowmax) and / or partmax = max |dz| of the launch (one
 * zeroed word; the dscale / dsb operand of yt8m_h2_split / yt8m_gemm_h2_nt_grouped).  Either may be NULL.  Needs the rotated epilogue
 * (yt8m_lstm_persist_bwd_images_rows(B, H) > 0) on a workspace of yt8m_lstm_persist_workspace_bytes_steps, else YT8M_E_SHAPE. */
int yt8m_lstm_persist_bwd_ex(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz, float* work,
                             int phase, const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, const void* wh_absmax,
                             void* rowmax, void* partmax, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
/* The same launch with the operand images of dz[t0 .. t0 + T) written by the recurrence itself (round 4): what yt8m_x3_split /
 * yt8m_x3_split_colsum would make of that part of dz in separate passes -- bit for bit -- for the products that follow it in the
 * backward pass of BasicLSTMCell under dynamic_rnn (W/all_frame_models/lstm_model.py:34-47 through tf.gradients, W/train.py:461):
 *   plain        x3 image [T B rows, K = 4H]  (A operand of dx = dz . W_x^T)                    yt8m_x3_image_bytes(T B, 4H)
 *   trans        x3 image [4H rows, K = T B]  (B operand of dW = x^T dz, dW_h = h^T dz)         yt8m_x3_image_bytes(4H, T B)
 *   trans_scaled the same of diag(rowscale) dz (layer-0 weight gradient on uint8 frames); rowscale: [F B] by absolute frame row
 *   colpart / colpart_scaled: [yt8m_lstm_persist_bwd_images_rows(B, H)][4H] column sums of dz (of diag(rowscale) dz) over the
 *                launch; their sum over the rows is colsum(dz[t0 .. t0 + T)) -- the bias gradient / the rank-1 remainder
 * Any member may be NULL.  The launch still writes dz in the standard layout.  Needs yt8m_lstm_persist_bwd_images_rows(B, H) > 0
 * (B % 16 == 0, a multiple of four 16-row tiles per workgroup) and a workspace from yt8m_lstm_persist_workspace_bytes_steps. */
typedef struct yt8m_persist_bwd_images {
  void* plain;
  void* trans;
  void* trans_scaled;
  const float* rowscale;
  float* colpart;
  float* colpart_scaled;
} yt8m_persist_bwd_images;
int yt8m_lstm_persist_bwd_images_rows(int64_t B, int64_t H);
int yt8m_lstm_persist_bwd_images(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                                 float* work, int phase, const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H,
                                 void* workspace, int64_t workspace_bytes, const yt8m_persist_bwd_images* images, yt8m_stream_t stream);

/* ---- the whole recurrent stack as two calls (csrc/lstm_stack.hip; SURVEY.md 8(b): yt8m_lstm_fwd / yt8m_lstm_bwd) --------------
 * MultiRNNCell([BasicLSTMCell(H)] * L) under tf.nn.dynamic_rnn(sequence_length = num_frames) and its gradient
 * (W/all_frame_models/lstm_model.py:34-47, lstm_memory_model.py:36-52 with its DropoutWrapper: input_keep_prob below; W/train.py:435-466), with the
 * reader's dequantise + l2-normalise (W/readers.py:178-187, W/train.py:343-344) folded into the layer-0 products when the input
 * is the raw uint8 [B,F,D] batch.  The time partition, the stream layout (one high-priority stream per layer + one for the weight
 * gradients, created once per device inside the library), the bf16-pipe product forms and all operand images are the library's:
 * this is the path bench.py measures.  The caller owns two buffers:
 *   tape    (yt8m_lstm_stack_tape_bytes)    activations kept from forward to backward; read them with yt8m_lstm_stack_view
 *   scratch (yt8m_lstm_stack_scratch_bytes) everything else; ZERO IT ONCE after allocation (its head holds the sticky time-out
 *           words of the persistent recurrence, see yt8m_lstm_persist_status) and keep it for the life of the model
 * both 256-byte aligned.  All work is ordered after what `stream` holds at the call and `stream` waits for all of it before the
 * call returns (asynchronously: nothing synchronises the host).  W[l]: [Din_l + H, 4H] row-major (Din_0 = D, else H), b[l]: [4H].
 * yt8m_lstm_stack_supported: 1 if the description is covered (else 0 and yt8m_last_error says why; use the per-call entry points). */
typedef struct yt8m_lstm_stack_desc {
  int64_t B, F, D, H;     /* videos, frames, input features, cells per layer */
  int32_t L;              /* layers, 1..8 */
  int32_t input_u8;       /* BIT FIELD.  bit 0: x = raw uint8 [B,F,D] (batch-major, as the reader hands it over); clear: float [F,B,D]
                           * (time-major).  bit 1 (ABI >= 3): the hoisted products take ONE-plane bf16 operand images (the
                           * --compute_dtype=bfloat16 variant).  A host written against ABI 2 passes 0 / 1 only. */
  float forget_bias;
  int32_t fwd_chunks;     /* time partition of the forward pass; 0 = the library's (1: one persistent launch per layer) */
  int32_t bwd_chunks;     /* ... of the backward pass; 0 = the library's (3 parts) */
  int32_t need_dx;        /* backward also produces dL/dx [F,B,D] (float input only) */
  /* ABI >= 4 (a host written against ABI 3 must zero these): tf.contrib.rnn.DropoutWrapper(cell, input_keep_prob) around EVERY layer
   * (W/all_frame_models/lstm_memory_model.py:36-44).  input_keep_prob in (0, 1): the input of layer l (the frames for l = 0, the
   * outputs of layer l - 1 otherwise -- never the recurrent state) is tf.nn.dropout(., keep) under the Philox stream of
   * dropout_seed[l] with element index = position in the [F,B,Din_l] tensor: the mask yt8m_dropout_f32 draws.  It is applied where
   * the layer's operand images are built (forward projection, weight gradient) and replayed on dx; the dropped tensors are never
   * written.  0 or 1: no dropout.  Float input, the f16 product forms (the fp32 configuration's default) and keep >= 0.3 only
   * (yt8m_lstm_stack_supported says so). */
  float input_keep_prob;
  int32_t reserved0;
  uint64_t dropout_seed[8];
} yt8m_lstm_stack_desc;
int yt8m_lstm_stack_supported(const yt8m_lstm_stack_desc* desc);
int64_t yt8m_lstm_stack_tape_bytes(const yt8m_lstm_stack_desc* desc);
int64_t yt8m_lstm_stack_scratch_bytes(const yt8m_lstm_stack_desc* desc);
int yt8m_lstm_stack_partition(const yt8m_lstm_stack_desc* desc, int* fwd_chunks, int* bwd_chunks);
/* the library's per-device streams: L high-priority layer streams + the weight-gradient stream (created on first use, never
 * destroyed).  Hosts that orchestrate the per-call entry points themselves should reuse them: all streams of a process share a
 * few hardware queues, and kernels of streams that land on one queue serialise. */
int yt8m_lstm_stack_streams(int L, yt8m_stream_t* layer_streams, yt8m_stream_t* wgrad_stream);
int yt8m_lstm_stack_fwd(const yt8m_lstm_stack_desc* desc, const void* x, const int32_t* num_frames, const float* const* W,
                        const float* const* b, void* tape, int64_t tape_bytes, void* scratch, int64_t scratch_bytes,
                        yt8m_stream_t stream);
/* which: 0 outputs of `layer` [F,B,H] (time-major, zeros beyond num_frames), 1 final c [B,H], 2 final h [B,H], 3 gates [F,B,4H] */
int yt8m_lstm_stack_view(const yt8m_lstm_stack_desc* desc, void* tape, int layer, int which, float** out);
/* dout_top [F,B,H] (may be NULL), dc_final / dh_final: L pointers to [B,H] (arrays or entries may be NULL);
 * dW[l] [Din_l + H, 4H], db[l] [4H] (entries may be NULL), beta_W / beta_b: L host floats, 0 = overwrite, 1 = accumulate (NULL: 0);
 * dx [F,B,D] when need_dx.  x and num_frames: the forward call's. */
int yt8m_lstm_stack_bwd(const yt8m_lstm_stack_desc* desc, const void* x, const int32_t* num_frames, const float* const* W, void* tape,
                        int64_t tape_bytes, void* scratch, int64_t scratch_bytes, const float* dout_top,
                        const float* const* dc_final, const float* const* dh_final, float* const* dW, float* const* db,
                        const float* beta_W, const float* beta_b, float* dx, yt8m_stream_t stream);
/* YT8M_E_HIP if any persistent launch of the stack gave up waiting since the previous status call.  Synchronises `stream`. */
int yt8m_lstm_stack_status(const yt8m_lstm_stack_desc* desc, void* scratch, yt8m_stream_t stream);
/* Makes `stream` wait until the weight / bias gradients of `layer` from the most recent yt8m_lstm_stack_bwd call on the current
 * device are final -- layer L-1 first, a whole last time part of weight-gradient work before layer 0.  A data-parallel host starts
 * each layer's gradient all-reduce from this point instead of the end of the call (W/train.py:624-639 averages the tower
 * gradients after the whole backward pass). */
int yt8m_lstm_stack_layer_done_wait(int layer, yt8m_stream_t stream);
/* One-shot host callback of the calling thread's next yt8m_lstm_stack_bwd, invoked right after its first backward recurrence is
 * enqueued, with the library's weight-gradient stream: what the callback enqueues there runs while that recurrence holds half the
 * chip and the stream has nothing else to do yet.  hook == NULL clears.  (The training step's own use of this window -- clip + Adam
 * of the variables whose gradients are already final -- no longer needs a callback: yt8m_lstm_stack_set_early_optimizer.) */
typedef void (*yt8m_stream_hook)(void* user, yt8m_stream_t stream);
int yt8m_lstm_stack_set_prep_hook(yt8m_stream_hook hook, void* user);
/* clip + TF-Adam of tensor RANGES of a flat parameter arena (the arguments of yt8m_sqnorm_multi / yt8m_adam_multi_ex / yt8m_adam_tiles
 * for the whole arena, plus the ranges): W/train.py:459-466 applied to the variables lo .. hi-1 of every range.  tensor_chunk_start
 * comes twice (device for the kernels, host for the launch arithmetic); jobs / job_tensor_host / job_tile_base_host describe the
 * image-owning matrices (yt8m_wimg_jobs_layout; njobs = 0: none).
 *   yt8m_optimizer_ranges: enqueues the passes on `stream` -- what a host's end-of-step pass calls.
 *   yt8m_lstm_stack_set_early_optimizer: the SAME passes, enqueued by the calling thread's next yt8m_lstm_stack_bwd itself on its
 *     weight-gradient stream right after its first backward recurrence (the window the prep hook describes): the variables whose
 *     gradients are final before the recurrent stack's backward pass starts (LstmModel: the MoE head, 85 % of the parameters) are
 *     updated while that recurrence holds half the chip.  The descriptor is copied; NULL clears; consumed by one call.  W/train.py
 *     applies all gradients after the whole backward pass; the arithmetic per variable is the same, only its place in the step moves. */
typedef struct yt8m_opt_ranges {
  float* w;
  float* m;
  float* v;
  const float* g;
  const int32_t* chunks;                  /* device int32[nchunks * 4]: the whole chunk table */
  const int32_t* tensor_chunk_start;      /* device int32[ntensors + 1] */
  const int32_t* tensor_chunk_start_host; /* host copy */
  const float* l2;                        /* device float[ntensors] */
  float* partial;                         /* device float[nchunks] */
  float* norms;                           /* device float[ntensors] */
  const uint8_t* skip_tensor;             /* device uint8[ntensors] or NULL: tensors updated by the tile pass (image owners) */
  const yt8m_wimg_job* jobs;              /* device, laid out; NULL when njobs == 0 */
  const int32_t* job_tensor_host;         /* host int32[njobs]: tensor index of job j, ascending */
  const int64_t* job_tile_base_host;      /* host int64[njobs + 1] */
  int32_t njobs;
  int32_t nranges;                        /* 1..8 */
  int32_t range_lo[8], range_hi[8];
  float gscale, clip, lr_t, beta1, beta2, eps;
  yt8m_stream_t after_stream;             /* NULL, or a stream whose work enqueued so far must finish before the pass starts (the
                                           * stream the host computed some of these gradients on); honoured by both entry points */
} yt8m_opt_ranges;
int yt8m_optimizer_ranges(const yt8m_opt_ranges* opt, yt8m_stream_t stream);
int yt8m_lstm_stack_set_early_optimizer(const yt8m_opt_ranges* opt);

/* Time-range forms of the same recurrence: steps [t0, t0+T) of a layer (backward: t0+T-1 down to t0), with the
 * re-packed recurrent weights owned by the caller (yt8m_lstm_pack; yt8m_lstm_packed_floats() floats each for the forward
 * image Wp and the backward image Wq; 0 = shape not covered, pass NULL and the generic per-step GEMM path runs).
 * z / cs / hs / out / gates / dz / dout are the WHOLE-layer base pointers.  They let a multi-layer stack (MultiRNNCell,
 * lstm_model.py:34-40) be pipelined over time chunks on separate streams.  bwd: work [4,B,H] holds the running (dh, dc)
 * in work[0..1] (phase 0) or work[2..3] (phase 1); each step flips the phase, the caller carries (phase + T) % 2. */
int64_t yt8m_lstm_packed_floats(int64_t B, int64_t H);
int yt8m_lstm_pack(const float* Wh, int64_t ldw, int64_t H, float* Wp, float* Wq, yt8m_stream_t stream);
int yt8m_lstm_steps_fwd(float* z, const float* Wh, int64_t ldw, const float* Wp, float* cs, float* hs, float* out,
                        const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias,
                        void* gemm_workspace, int64_t gemm_workspace_bytes, yt8m_stream_t stream);
int yt8m_lstm_steps_bwd(const float* gates, const float* Wh, int64_t ldw, const float* Wq, const float* cs,
                        const float* dout, float* dz, float* work, int phase, const int32_t* num_frames, int64_t t0,
                        int64_t T, int64_t B, int64_t H, void* gemm_workspace, int64_t gemm_workspace_bytes,
                        yt8m_stream_t stream);

/* The packed-weight step chains above are launch-bound (75-300 steps x 1-2 kernels of ~20 us): each distinct argument tuple is
 * captured once on the caller's stream and replayed as ONE hipGraph launch afterwards (LRU cache inside the library; off
 * while yt8m_prof_enable(1) is active or with YT8M_NO_GRAPH set in the environment). */
int yt8m_graph_cache_stats(int64_t* hits, int64_t* captures, int64_t* fallbacks, int64_t* entries);
int yt8m_graph_cache_clear(void);

/* ---- masked softmax over frames + renormalise (lstm_attention_max_pooling_model.py:59-60) -------
 * act [B,F,A] -> w [B,F,A]: w = mask * softmax_F(act) / sum_F(mask * softmax_F(act)).  bwd: dact from dw. */
int yt8m_attn_softmax_fwd(const float* act, const int32_t* num_frames, float* w, int64_t B, int64_t F,
                          int64_t A, yt8m_stream_t stream);
int yt8m_attn_softmax_bwd(const float* w, const float* dw, const int32_t* num_frames, float* dact,
                          int64_t B, int64_t F, int64_t A, yt8m_stream_t stream);

/* ---- row softmax over the last axis with an optional frame mask (NetVLAD assignment, Appendix B) ---
 * s [rows, K] -> a = softmax_K(s) * (f < num_frames[b]); rows = B*F. */
int yt8m_softmax_rows_fwd(const float* s, const int32_t* num_frames, float* a, int64_t B, int64_t F,
                          int64_t K, yt8m_stream_t stream);
int yt8m_softmax_rows_bwd(const float* a, const float* da, const int32_t* num_frames, float* ds,
                          int64_t B, int64_t F, int64_t K, yt8m_stream_t stream);

/* ---- NetVLAD residual aggregation + intra-normalisation (not in the reference) -------------
 * agg [B,K,D] = a^T x per video (batched GEMM), a [B,F,K] masked assignment, centres [K,D]:
 * n = sum_f a;  vlad = l2norm_D(agg - n*c).  One pass over the rows.  n_out [B,K] is saved for the backward
 * (a == NULL: n_out already holds n on entry -- the fused uint8 pooling computes it).
 * bwd: dagg [B,K,D], dn [B,K] (to be broadcast over frames into da), dcentres [K,D] (beta 0/1; may be NULL). */
int yt8m_vlad_finish_fwd(const float* agg, const float* a, const float* centres, float* vlad, float* n_out,
                         int64_t B, int64_t F, int64_t K, int64_t D, float eps, yt8m_stream_t stream);
int yt8m_vlad_finish_bwd(const float* agg, const float* n_in, const float* centres, const float* dvlad, float* dagg,
                         float* dn, float* dcentres, float dcentres_beta, int64_t B, int64_t K, int64_t D, float eps,
                         yt8m_stream_t stream);
/* The same two passes with q_out [B,K] = ||vlad[b,k,:]||^2 (1 unless the row norm was clamped) and its gradient dq (may be NULL):
 * the l2-normalisation of the whole [K D] descriptor that follows then needs no pass over [B,K,D] -- its
 * scale rsqrt(max(sum_k q, eps)) is applied to the [B, hidden] output of the next layer.  D % 4 == 0, D <= 2048, 16-byte aligned
 * operands (yt8m_vlad_finish_q_supported). */
int yt8m_vlad_finish_q_supported(int64_t D);
int yt8m_vlad_finish_q_fwd(const float* agg, const float* a, const float* centres, float* vlad, float* n_out, float* q_out, int64_t B,
                           int64_t F, int64_t K, int64_t D, float eps, yt8m_stream_t stream);
int yt8m_vlad_finish_q_bwd(const float* agg, const float* n_in, const float* centres, const float* dvlad, const float* dq, float* dagg,
                           float* dn, float* dcentres, float dcentres_beta, int64_t B, int64_t K, int64_t D, float eps,
                           yt8m_stream_t stream);

/* ---- fused NetVLAD pooling on RAW uint8 frames (SURVEY.md section 8b "yt8m_netvlad_fwd/bwd", Appendix B) -----------
 * Replaces, for the NetVLAD plugin, the chain  Dequantize (W/utils.py:23-38) -> zero padding (W/readers.py:178-187) ->
 * l2_normalize (W/all_feature_transform/default_transformer.py:7) -> assignment GEMM -> masked softmax -> aggregation GEMM.
 *   q [B,F,D] uint8 (16-byte aligned), num_frames [B] int32 or NULL, Wc [D,K], bc [K]
 *   a [B,F,K] = softmax_K(x.Wc + bc) * (f < num_frames)          x = l2_normalize(dequantise(q)), never written
 *   cT_out [B,K,Fp] = a[f,k] / ||dequantise(q[f])||, frames contiguous, Fp = F rounded up to 32 (zero padded): the tensor
 *                     the backward needs (a = cT * ||.||); 16-byte aligned
 *   n_out [B,K] = sum_f a[f,k];   agg_out [B,K,D] = sum_f a[f,k] x[f,:]     (feed yt8m_vlad_finish_fwd with a = NULL)
 * bwd: cT (from the forward), dagg [B,K,D], dn [B,K] (both from yt8m_vlad_finish_bwd) -> dWc [D,K], dbc [K] (beta 0/1).
 * q carries no gradient.
 * nsplit = 2: fp32 operands enter the f16 matrix cores as hi + lo halves (fp32-class, ~1e-6 relative);
 * nsplit = 1: single f16 operand (the reduced-precision variant, ~5e-4 relative on the weights).
 * Supported when yt8m_netvlad_supported() != 0 (K == 64, D % 64 == 0); workspace from yt8m_netvlad_workspace_bytes(). */
int yt8m_netvlad_supported(int64_t B, int64_t F, int64_t D, int64_t K);
/* 1 when yt8m_netvlad_fwd_u8 takes the single-pass form for this shape (one workgroup per video: assignment, softmax and aggregation in
 * one launch, the frames read from HBM once; K = 64, D % 128 == 0, D <= 1152, F <= 320), 0 when it runs the rows + cols pair. */
int yt8m_netvlad_single_pass(int64_t B, int64_t F, int64_t D, int64_t K);
int yt8m_netvlad_set_single(int mode);      /* -1: YT8M_NETVLAD_SINGLE / default (on); 0: always the pair; 1: single pass where covered (A/B, tests) */
int64_t yt8m_netvlad_workspace_bytes(int64_t B, int64_t F, int64_t D, int64_t K);
int yt8m_netvlad_fwd_u8(const uint8_t* q, const int32_t* num_frames, const float* Wc, const float* bc, int64_t B,
                        int64_t F, int64_t D, int64_t K, int nsplit, float eps, float* cT_out, float* n_out,
                        float* agg_out, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);
int yt8m_netvlad_bwd_u8(const uint8_t* q, const int32_t* num_frames, const float* cT, const float* dagg,
                        const float* dn, int64_t B, int64_t F, int64_t D, int64_t K, int nsplit, float eps,
                        float* dWc, float dWc_beta, float* dbc, float dbc_beta, void* workspace,
                        int64_t workspace_bytes, yt8m_stream_t stream);

/* ---- gradient all-reduce: RCCL over xGMI (csrc/comm.hip) ---------------------------------------
 * Replaces the reference's asynchronous parameter-server traffic (tf.train.replica_device_setter + apply_gradients over
 * gRPC, W/train.py:624-639,731-776) by the synchronous data-parallel mean of SURVEY.md 8e.  One communicator per process /
 * GPU: rank 0 creates the 128-byte id (yt8m_comm_unique_id), the caller hands it to every rank, each rank calls
 * yt8m_comm_init with its HIP device current.  All-reduce / broadcast are in place on fp32 device buffers and
 * asynchronous on `stream`.  RCCL is bound with dlopen at first use; every failure here returns YT8M_E_RCCL. */
#define YT8M_COMM_ID_BYTES 128
int yt8m_comm_unique_id(void* id_out);
int yt8m_comm_init(int rank, int world, const void* unique_id, void** comm_out);
int yt8m_comm_size(void* comm, int* rank, int* world);
int yt8m_comm_allreduce_f32(void* comm, float* buf, int64_t n, int mean, yt8m_stream_t stream);
int yt8m_comm_allreduce_mean(void* comm, float* buf, int64_t n, yt8m_stream_t stream);
/* the same reduction as reduce-scatter + all-gather (two RCCL launches, in place; phase 0 = both, 1 = reduce-scatter only: rank r
 * then owns the reduced slice [r * (n / world), (r + 1) * (n / world)) plus the all-reduced n % world tail, 2 = all-gather only) */
int yt8m_comm_allreduce_rsag_f32(void* comm, float* buf, int64_t n, int mean, int phase, yt8m_stream_t stream);
int yt8m_comm_broadcast_f32(void* comm, float* buf, int64_t n, int root, yt8m_stream_t stream);
int yt8m_comm_destroy(void* comm);

/* ---- uint8 operand path of the hoisted input projection (csrc/u8proj.hip) -----------------------------------------
 * "readers.py uint8 -> float dequantise folded into the first GEMM" (W/readers.py:178-187, W/utils.py:23-38,
 * default_transformer.py:4-8 -> lstm_model.py:34-47):  x.W = r (.) ((q - 128).(alpha W) + beta colsum(W)),  (q - 128) exact
 * in bf16, alpha W split into three bf16 terms (exact to 2^-26), one bf16 NT product over the concatenated reduction.
 *   yt8m_u8_frames_to_bf16_tm: q [B,F,D] uint8 -> Qb [F*B, ldq] bf16 (TIME-major rows f*B+b, `copies` copies of (q-128)
 *     side by side), r_out [F*B] = 1/||dequantise(q_f)|| (0 for padding frames), optionally x_tm [F,B,D] fp32 = the
 *     transformed frames themselves (bit-identical to yt8m_dequant_l2norm_u8, time-major) for the weight-gradient product.
 *   yt8m_split3_bf16_t: W [K, ldw] fp32 -> out [N, ldo] bf16, out[n][j K + k] = term j of the split of scale * W[k][n].
 *   yt8m_rowscale_bias_f32: z[m][n] = r[m] (z[m][n] + beta cs[n]) + bias[n] in place (the affine remainder). */
int yt8m_u8_proj_supported(int64_t D);
int yt8m_u8_frames_to_bf16_tm(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, float eps,
                              int copies, void* Qb, int64_t ldq, float* x_tm, float* r_out, yt8m_stream_t stream);
/* same pass writing (q - 128) as the one-plane operand image of yt8m_gemm_x1x3_nt (ceil(B F / 32) * (D / 16) KiB; rows time-major
 * f * B + b; D % 16 == 0) instead of bf16 copies */
int yt8m_u8_frames_image(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, float eps, void* image,
                         float* x_tm, float* r_out, yt8m_stream_t stream);
/* (q - 128)^T as a ONE-plane x3 operand image: rows = features (D), K = time-major frame rows f * B + b (zeros for padding frames);
 * ceil(D / 32) * ceil(B F / 16) KiB.  The A operand of yt8m_gemm_x1x3_nt_ex for the layer-0 weight gradient. */
int yt8m_u8_frames_image_t(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, void* image_t,
                           yt8m_stream_t stream);
int yt8m_split3_bf16_t(const float* W, int64_t ldw, int64_t K, int64_t N, float scale, void* out, int64_t ldo,
                       yt8m_stream_t stream);
int yt8m_rowscale_bias_f32(float* z, int64_t M, int64_t N, int64_t ldz, const float* r, const float* cs, float beta,
                           const float* bias, yt8m_stream_t stream);

/* ---- DbofModel pieces (csrc/dbof.hip; W/all_frame_models/dbof_model.py:36-124, W/model_utils.py:23-95) -------
 * yt8m_sample_frames_*: SampleRandomFrames (mode 0: index = int(u[b,s] * num_frames[b])) / SampleRandomSequence (mode 1:
 *   start = int(u[b] * (max(num_frames - S, 0) + 1)), index = min(start + s, num_frames - 1)); u = Philox4x32-10 uniform
 *   of element b*S+s (mode 0) or b (mode 1) under `seed`; x [B,F,D] -> out [B,S,D], idx_out [B,S] (may be NULL).
 * yt8m_frame_pool_*: FramePooling over the S sampled frames, mode 0 max (gradient split equally between ties, as
 *   tf.reduce_max) / 1 average; x [B,S,C] -> out [B,C].
 * yt8m_batchnorm_*: slim.batch_norm(center=True, scale=True) over the rows of x [N,C]: training != 0 uses the batch
 *   moments (biased variance) and updates moving_mean / moving_var with `decay`; save_mean / save_rstd [C] feed the
 *   backward.  bwd: dx may be NULL (input is data); dgamma / dbeta written with beta_* = 0 or accumulated with 1. */
int yt8m_sample_frames_f32(const float* x, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, int64_t S, int mode,
                           uint64_t seed, float* out, int32_t* idx_out, yt8m_stream_t stream);
int yt8m_sample_frames_u8(const uint8_t* x, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, int64_t S, int mode,
                          uint64_t seed, uint8_t* out, int32_t* idx_out, yt8m_stream_t stream);
int yt8m_frame_pool_fwd(const float* x, int64_t B, int64_t S, int64_t C, int mode, float* out, yt8m_stream_t stream);
int yt8m_frame_pool_bwd(const float* x, const float* out, const float* dy, int64_t B, int64_t S, int64_t C, int mode,
                        float* dx, yt8m_stream_t stream);
int yt8m_batchnorm_fwd(const float* x, int64_t N, int64_t C, const float* gamma, const float* beta, float* moving_mean,
                       float* moving_var, int training, float eps, float decay, float* y, float* save_mean,
                       float* save_rstd, yt8m_stream_t stream);
int64_t yt8m_batchnorm_workspace_bytes(int64_t C);
int yt8m_batchnorm_bwd(const float* x, const float* dy, int64_t N, int64_t C, const float* gamma, const float* save_mean,
                       const float* save_rstd, int training, float* dx, float* dgamma, float dgamma_beta, float* dbeta,
                       float dbeta_beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);

/* ---- per-row top-k for the GAP@20 eval path (W/eval_util.py:123-165 top_k_triplets) -------------
 * p [B,V] -> vals [B,k] (descending), idx [B,k] int32; ties broken towards the LOWER class index. k<=64 */
int yt8m_topk_rows(const float* p, int64_t B, int64_t V, int k, float* vals, int32_t* idx,
                   yt8m_stream_t stream);

/* ---- per-row precision at equal recall rate (W/eval_util.py:74-99 calculate_precision_at_equal_recall_rate) ---
 * p [B,V] f32, labels [B,V] uint8/bool -> perr [B]: among the top-(#labels) classes of the row (stable descending
 * order), the fraction that are labels with a score > 0; 0 for a row without labels.  V <= 19200. */
int yt8m_perr_rows(const float* p, const uint8_t* labels, int64_t B, int64_t V, float* perr, yt8m_stream_t stream);

/* ---- native input reader (HOST buffers) ----------------------------------------------
 * TFRecord framing (u64 length | masked crc32c | payload | masked crc32c) + tf.train.SequenceExample / tf.train.Example
 * wire-format decode for YT8MFrameFeatureReader (W/readers.py:189-259) and YT8MAggregatedFeatureReader (:94-125).
 * Frame batches stay RAW uint8 (the device transform dequantises); rows >= num_frames are zero bytes; labels become a
 * uint8 multi-hot (duplicates / order irrelevant).  *n_read < max_records means end of file.  Errors: missing file or
 * feature, wrong feature size, corrupt CRC, malformed protobuf -> negative status + yt8m_last_error(). */
uint32_t yt8m_crc32c(const void* data, int64_t n);
uint32_t yt8m_crc32c_masked(const void* data, int64_t n);
int yt8m_tfrecord_open(const char* path, int check_crc, void** reader_out);
int yt8m_tfrecord_close(void* reader);
int yt8m_tfrecord_read_frame_batch(void* reader, const char* const* feature_names, const int32_t* feature_sizes, int nfeat,
                                   int64_t max_frames, int64_t num_classes, int64_t max_records, uint8_t* q,
                                   int32_t* num_frames, uint8_t* labels, char* video_ids, int64_t id_stride,
                                   int64_t* n_read);
int yt8m_tfrecord_read_video_batch(void* reader, const char* const* feature_names, const int32_t* feature_sizes, int nfeat,
                                   int64_t num_classes, int64_t max_records, float* x, uint8_t* labels, char* video_ids,
                                   int64_t id_stride, int64_t* n_read);

/* multi-threaded shard prefetcher (the reference's num_readers queue-runner threads + batch_join, W/train.py:199-209):
 * nthreads workers decode whole batches of `batch` records (short at shard ends, never spanning shards) into slots of pinned
 * host memory; acquire() lends the next ready batch (pointers valid until the next acquire / close; *n == 0: all shards
 * done).  One thread reproduces the sequential reader's batch sequence; more threads deliver every record exactly once in a
 * scheduling-dependent batch order (the reference shuffles).  frame_level: q uint8 [n,max_frames,D] + num_frames; else x
 * float [n,D].  A decode error in any worker surfaces from acquire() with that worker's message. */
int yt8m_prefetch_open(const char* const* paths, int npaths, int frame_level, const char* const* feature_names,
                       const int32_t* feature_sizes, int nfeat, int64_t max_frames, int64_t num_classes, int64_t batch,
                       int nthreads, int queue_depth, int check_crc, void** prefetcher_out);
int yt8m_prefetch_acquire(void* prefetcher, void** q_or_x, int32_t** num_frames, uint8_t** labels, char** video_ids,
                          int64_t* id_stride, int64_t* n, int* pinned);
int yt8m_prefetch_close(void* prefetcher);

/* prediction dump for the ensemble stage (W/inference-pre-ensemble.py:291-308): one tf.train.Example per video with
 * {"video_id", "labels" = nonzero(labels row), feature_name = predictions row (float list)}, TFRecord framed.  HOST buffers:
 * video_ids [n, id_stride] NUL-padded, labels [n, num_classes] uint8 multi-hot, predictions [n, num_classes] float32. */
int yt8m_tfrecord_write_predictions(const char* path, int64_t n, const char* video_ids, int64_t id_stride,
                                    const uint8_t* labels, const float* predictions, int64_t num_classes,
                                    const char* feature_name);

#ifdef __cplusplus
}
#endif
#endif /* YT8M_HIP_H */
