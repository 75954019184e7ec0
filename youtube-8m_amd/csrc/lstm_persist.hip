// lstm_persist.hip -- persistent BasicLSTM recurrence for gfx950: ONE launch runs T time steps of a layer, forward or backward
// (tf.contrib.rnn.BasicLSTMCell under tf.nn.dynamic_rnn and its gradient; W/all_frame_models/lstm_model.py:34-47, SURVEY.md K5,
// App. A.3-A.5, App. G).
//
// The per-step kernels (lstm_fused.hip) re-stream W_h from L2 for every step (128 MB / step at B = 128, H = 1024) and pay a
// launch boundary per step: 19-22 us against a 6.8 us MFMA bound.  Here the recurrent weights never move:
//   * forward: workgroup = (unit group of 8 hidden units = their 32 gate columns, row group).  K = H is split over 8 "matrix"
//     waves (wave w owns k in [w H/8, (w+1) H/8) = NQ = H/128 q-groups of 16 k); half of a wave's B fragments stay in registers,
//     half in LDS (64 KB per workgroup) -- no L2 traffic for weights in the step loop.  Backward: 16 units per workgroup
//     (K = 4H, output H wide), 128 KB of W_h in registers + 128 KB in LDS.
//   * the only per-step traffic is the state itself: h_t (forward) / dz_t (backward) is exchanged through a buffer laid out in
//     MFMA A-fragment order ([parity][16-row tile][q-group][16 rows][16 k] = 1 KB blocks): a producer writes the bytes it owns with
//     write-through 16-byte stores (sc0 sc1), a consumer wave fetches one fully coalesced 1 KB block per 4 MFMAs with sc0 sc1 loads
//     (coherent across the 8 XCD-private L2s; no acquire fence, no L1 involvement).
//   * no grid barrier: work is cut into ITEMS = (time step, 16-row tile).  Item (s, T) needs the state of tile T from step s - 1
//     only, so completion is tracked per tile with monotonic arrival counters (8 shards per tile, one 128-byte line each) and a
//     workgroup that owns several tiles has as many independent chains in flight; the A fragments of the next item(s) are
//     requested while the MFMAs of the current one run, after a speculative counter read whose round trip hides under MFMAs.
//   * 12 waves per workgroup in two roles, coupled only through LDS counters (no s_barrier in the step loop): 8 matrix waves
//     (64 v_mfma_f32_16x16x4_f32 per item forward, 128 backward, partial tiles -> LDS slot -> ds_add) and 4 epilogue waves
//     (s_setprio 3: reduce the 8 partial tiles in a fixed order, gate math, publish the new state FIRST -- write-through stores,
//     drain, one relaxed agent-scope atomic_add -- then write what the other pass / the caller needs in the standard layouts).
// Exact fp32 products (v_mfma_f32_16x16x4_f32 is an fmaf chain) with a fixed summation order: bitwise reproducible run to run.
// The gate non-linearities use v_exp_f32 / v_rcp_f32 (<= 1.5e-7 absolute; the libm forms were a 600-instruction dependent chain
// on the critical path).  Every spin is bounded (wall clock): on a time-out the control block's error word is set, all waits fall
// through and the launch ends; yt8m_lstm_persist_status reports YT8M_E_HIP.
//
// Deadlock note: a persistent launch needs its whole grid resident (1 workgroup per CU: 768 threads x ~160 VGPRs).  Two such
// launches on different streams could each hold part of the chip and wait for the rest forever, so the host side chains every
// persistent launch of a device behind the previous one with an event (PersistGate), whatever stream the caller passes.
// Measured behaviour, the s_memtime timeline tooling and what was tried and rejected: DESIGN.md section 7.1.
#include <algorithm>
#include <atomic>
#include <mutex>
#include "common.h"
#include "x3_image.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef YT8M_FWD_POLLQ
#define YT8M_FWD_POLLQ 2     // where in an item (quarters of its MFMA block) the state of the item PD ahead is requested
#endif
#ifndef YT8M_AUX_ST
#define YT8M_AUX_ST 17
#endif
#ifndef YT8M_EPI_PRIO
#define YT8M_EPI_PRIO 3
#endif
#ifndef YT8M_AUX_LD
#define YT8M_AUX_LD 17
#endif
// aux 17 = sc0 sc1: write-through store / coherent load (MI355X_MICROARCH.md, inter-workgroup visibility)
constexpr long long SPIN_TIMEOUT = 300000000;  // wall_clock64 ticks (100 MHz): 3 s
constexpr int CTL_HDR = 32;                    // control block: [0] error word, counters from word 32
constexpr int CTL_STICKY = 32;                 // words between the sticky error word (workspace word 0) and ctl[0]
#ifndef YT8M_PERSIST_SHARDS
#define YT8M_PERSIST_SHARDS 8
#endif
// Arrival counter shards per tile, one 128-byte line each (the single polling wave of a workgroup reads all of them with one
// load instruction).  Measured at B = 128, H = 1024: 8 and 16 shards run alike (10.6 us / step forward), 64 are slower (12.4:
// the poll costs more than the shorter add queues save).
constexpr int NSH = YT8M_PERSIST_SHARDS;

struct PersistFwdArgs {
  float* z;             // [F,B,4H] hoisted input projection + bias on entry, gate activations on exit
  const float* Wh;      // [H, ldw]
  long long ldw;
  float* cs;            // [F+1,B,H]
  float* hs;            // [F+1,B,H]
  float* out;           // [F,B,H] or null
  const int32_t* nf;    // [B] or null
  float* hx;            // exchange images [nimg][NT16][H/16][256]: one per step (SH) or two alternating ones
  unsigned* ctl;        // control block (zeroed before the launch)
  unsigned* stats;      // device-wide placement statistics (see note_placement)
  int t0, T, B, H;
  float fb;
  int NU, RB, NT16, per, pf;
  int nimg;             // exchange images in the workspace (>= T: one per step)
  const unsigned* wword;     // f16 form of the recurrent product (lstm_persist_fwd_x3_kernel<.., 2, true>): max |W_h| as float bits
  unsigned long long* dbg;   // timing variant (-DYT8M_PERSIST_TIMING): s_memtime stamps of workgroup 0
};

#ifdef YT8M_PERSIST_TIMING
#define STAMP(slot)                                                                                            \
  do {                                                                                                         \
    if (blockIdx.x == 0 && (w == 0 || w == 8) && lane == 0 && k < 256)                                          \
      a.dbg[((w >> 3) * 256 + k) * 8 + (slot)] = __builtin_readcyclecounter();                                 \
  } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ float4 as_f4(u32x4 v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// The kernels take "block b runs on XCD (b + k) % 8 for one k per launch" as a placement hint (row groups <-> XCD sets,
// line-sharing unit groups on one XCD).  A launch that finds part of the chip busy is placed wherever CUs are free; nothing breaks,
// but the state fetch loses its L2 sharing.  Counted so that a slow step can be told from an unlucky placement: workgroup 0 posts
// its offset k in the control block at the start, every workgroup compares its own with it at the end: stats[0] += mismatches.
__device__ __forceinline__ unsigned placement_offset() {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((xcc & 7u) - (blockIdx.x & 7u)) & 7u;
}
__device__ __forceinline__ void note_placement(unsigned* ctl) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(ctl + 2, placement_offset() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void check_placement(unsigned* ctl, unsigned* stats) {      // one lane, at the end of the kernel
  if (!stats) return;
  const unsigned k0 = __hip_atomic_load(ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (k0 != 0u && k0 - 1u != placement_offset()) __hip_atomic_fetch_add(stats, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sticky error word.  ctl[0] is the PER-LAUNCH flag (zeroed with the counters before every launch: a time-out makes the waits of
// THAT launch fall through and must not poison the next one).  The word the host reads -- yt8m_lstm_persist_status -- is ctl[-32],
// the first word of the workspace, outside the region a launch zeroes: every workgroup that leaves a timed-out launch ORs the
// flag into it (one lane, at the end of the kernel), and only the status call clears it.  A time-out in ANY launch since the last
// status call is therefore reported, whatever ran on the workspace afterwards.
// The wave that times out also writes the sticky word itself (wait_tile): the exit path below only covers workgroups whose
// wave 8 leaves after the flag was set.
__device__ __forceinline__ void propagate_error(unsigned* ctl) {                        // one lane, at the end of the kernel
  if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
    __hip_atomic_store(ctl - CTL_STICKY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum of lanes 0 .. NSH-1 (wave-uniform result)
__device__ __forceinline__ unsigned shard_sum(unsigned v) {
  unsigned tot = 0;
#pragma unroll
  for (int i = 0; i < NSH; ++i) tot += (unsigned)__builtin_amdgcn_readlane((int)v, i);
  return tot;
}

// Waits until the NSH shard counters of `tile` sum to >= target.  Lanes 0-7 poll one shard each with relaxed agent-scope
// loads; bounded: after SPIN_TIMEOUT the error word is set and every later wait falls through at once.
__device__ __forceinline__ void wait_tile(unsigned* ctl, int tile, unsigned target, int lane) {
  unsigned* c = ctl + CTL_HDR + (tile * NSH + (lane & (NSH - 1))) * 32;
  long long t_start = 0;
  for (unsigned spins = 0;; ++spins) {
    unsigned v = 0;
    if (lane < NSH) v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned tot = shard_sum(v);
    if (tot >= target) return;
    if (spins >= 16) {
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 255) == 16) {
        if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        const long long now = wall_clock64();
        if (t_start == 0) t_start = now;
        else if (now - t_start > SPIN_TIMEOUT) {
          if (lane == 0) {
            __hip_atomic_store(ctl - CTL_STICKY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky word first: the host's view
            __hip_atomic_store(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          return;
        }
      }
    }
  }
}

// hx parity 0 <- h_{t0-1} (standard layout hs[t0], written by the previous launch / the caller), so that step 0 of a launch
// reads its state exactly like every later step (one uniform, branch-free load path in the step loop).
__global__ __launch_bounds__(256) void hx_pack_kernel(const float* __restrict__ h, float* __restrict__ hx, int B, int H, int NT16) {
  const long long n = (long long)NT16 * 16 * H;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int j = (int)(e & 15), i = (int)((e >> 4) & 15);
    const long long blk = e >> 8;                      // T * (H/16) + q-group
    const int qg = (int)(blk % (H >> 4)), T = (int)(blk / (H >> 4));
    const int row = T * 16 + i;
    hx[e] = row < B ? h[(long long)row * H + qg * 16 + j] : 0.f;
  }
}

// Gate non-linearities of the persistent kernels: v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the libm-exact expf / tanhf of
// the per-step kernels -- the epilogue is a single wave's dependent chain on the critical path of the state exchange, and the
// exact forms cost ~10x the instructions.  Absolute error <= ~1.5e-7 per value (checked against the exact kernels and the fp64
// oracle by the parity tests; north_star tolerance 1e-3).
#ifdef YT8M_EPI_FAKE   // timing experiment only (wrong results): how much of the chain is the gate math?
__device__ __forceinline__ float fast_sigmoid(float x) { return 0.5f + 0.01f * x; }
__device__ __forceinline__ float fast_tanh(float x) { return 0.01f * x; }
#else
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}
#endif

// value of lane + N within a row of 16 lanes (DPP row_shl: one VALU op; __shfl_down is a ds_bpermute round trip through LDS)
template <int N>
__device__ __forceinline__ float row_shl(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xF, 0xF, true));
}

__device__ __forceinline__ unsigned lds_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// bounded spin on an LDS word (intra-workgroup hand-offs between the MFMA waves and the epilogue waves)
__device__ __forceinline__ void lds_wait_ge(const unsigned* p, unsigned target, unsigned* ctl) {
  for (unsigned spins = 0; lds_load(p) < target; ++spins) {
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 4095) == 4095 && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  }
}

constexpr int MAX_LOCAL_TILES = 64;   // 16-row tiles one workgroup may own (persist_geometry refuses larger batches)
constexpr int NSLOT = 4;     // partial-tile slots in LDS (items k, k+4, ... share slot k & 3)
constexpr int NEPI = 4;      // epilogue waves (item k is finished by epilogue wave k & 3)

// Workgroup = 12 waves in two roles, coupled only through LDS counters (no s_barrier in the step loop):
//   waves 0-7  "matrix" waves: K split 8-way, W_h slice in registers; per item: poll + request the A fragments of the NEXT item,
//              64 MFMAs on the current one, partial 16x32 tile -> LDS slot, ds_add on the slot's arrival counter.
//   waves 8-11 "epilogue" waves: wave 8 + (k & 3) finishes item k: waits for the 8 partial tiles, reduces them in a fixed order,
//              gate math + cell update + copy-through for the 128 (row, unit) pairs (2 per lane), stores, publishes h_t of the
//              tile (write-through stores -> drain -> arrival counter).  Four of them in rotation: an epilogue (~2 us with
//              its store drain) has four item times, so the matrix pipe never waits for it.
// PF: every workgroup owns >= 2 tiles -> the A fragments of item k + 1 are requested at the start of item k; otherwise each
// item waits for and fetches its own operands (tiny batches).
// PD = how many items ahead the A fragments are requested: 2 when every workgroup owns >= 4 tiles (three register buffers in
// rotation; the coherent loads take ~2 us under load, more than one item of matrix work), 1 for 2-3 tiles, 0 (each item waits
// for and fetches its own operands) for a single tile.
// SH: one exchange image per step.  An address is then written once and read only after its tile's arrival count is complete,
// so no cache can hold an older copy of it within the launch: the consumers fetch with PLAIN loads, the first one of an XCD
// brings a line into that XCD's L2 and the other 31 CUs hit it -- the fabric carries every state byte 8 times per step instead
// of 256 times (with two alternating images the loads must be sc0 sc1 = served by the fabric every time: both recurrences then
// ran at the same ~5.7 TB/s of cross-XCD reads, 11.1 / 23.3 us per step).
template <int NQ, int PD, bool SH>
__global__ __launch_bounds__(768) void lstm_persist_fwd_kernel(PersistFwdArgs a) {
  constexpr int HQ = NQ / 2;                             // q-groups per wave whose weights sit in registers (the rest: LDS)
  constexpr int NL = NQ - HQ;                            // ... in LDS (NQ = 1, H = 128: the single q-group; HQ = 0)
  constexpr int HQA = HQ > 0 ? HQ : 1;                   // array extents (no zero-length arrays)
  __shared__ __attribute__((aligned(16))) float red[NSLOT][8][2][4][64];   // [slot][wave][col half][acc reg][lane]: 64 KB
  __shared__ __attribute__((aligned(16))) float4 Wl[8][NL][2][64];         // LDS-resident half of W_h's slice: 8 * NL * 2 KB
  __shared__ unsigned lds_cnt[NSLOT], lds_free[NSLOT];
  // Only matrix wave 0 polls the arrival counters in memory; it posts what it has seen per local tile here and the other seven
  // matrix waves wait on LDS (eight waves x 256 workgroups polling the same lines was the bulk of the fabric's request traffic).
  __shared__ unsigned lds_seen[MAX_LOCAL_TILES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ug, g;
  {
    const int b = blockIdx.x;
    // row group <-> XCD set (speed only).  Within a row group the unit groups of an XCD come in runs of four: a unit group's z / c /
    // h accesses are 32-byte pieces of 128-byte lines, and the four workgroups that share a line then share an L2 (with unit
    // groups dealt round-robin over the XCDs every XCD fetched every line for a quarter of it: 3.4x the algorithmic HBM bytes).
    if (a.per > 0 && (a.NU % (4 * a.per)) == 0) {
      const int x = b & 7, sl = b >> 3;
      g = x / a.per;
      ug = ((sl >> 2) * a.per + (x % a.per)) * 4 + (sl & 3);
    } else if (a.per > 0) { const int x = b & 7; g = x / a.per; ug = (b >> 3) * a.per + (x % a.per); }
    else { g = b / a.NU; ug = b % a.NU; }
  }
  const int H = a.H, B = a.B, NT16 = a.NT16, RB = a.RB;
  const int n_it = (NT16 - g + RB - 1) / RB;            // tiles g, g + RB, ... of this workgroup
  const int total = n_it * a.T;
  const unsigned img_bytes = (unsigned)NT16 * (unsigned)H * 16u * 4u;
  const long long img_f = (long long)NT16 * H * 16;
  auto image = [&](int s) -> __amdgpu_buffer_rsrc_t { return make_rsrc(a.hx + (SH ? s : (s & 1)) * img_f, img_bytes); };
  const int QH = H >> 4;                                // q-groups per row
  const unsigned arrivals = (unsigned)a.NU;             // per (tile, step): one epilogue wave per workgroup
  // B fragment of v_mfma_f32_16x16x4_f32: lane (n = lane & 15, kq = lane >> 4) supplies B[k = kq][n]; a float4 covers the four
  // successive MFMAs e = 0..3 of a q-group (k = 16 q + 4 kq + e).  Column n of half ct <-> (unit 4 ct + n / 4, gate n % 4), so
  // the four gates of a unit are four neighbouring lanes of the result (a float4 of the LDS partial tile).
  auto w_frag = [&](int qg, int ct) -> float4 {
    const int i16 = lane & 15, kq = lane >> 4;
    const long long k = (long long)(w * NQ + qg) * 16 + kq * 4;
    const long long col = (long long)(i16 & 3) * H + ug * 8 + ct * 4 + (i16 >> 2);
    const float* p = a.Wh + k * a.ldw + col;
    return make_float4(p[0], p[a.ldw], p[2 * a.ldw], p[3 * a.ldw]);
  };
  note_placement(a.ctl);
  if (tid < NSLOT) { lds_cnt[tid] = 0; lds_free[tid] = 0; }
  for (int i = tid; i < MAX_LOCAL_TILES; i += 768) lds_seen[i] = 0;
  if (w < 8) {
#pragma unroll
    for (int qq = 0; qq < NL; ++qq) {
      Wl[w][qq][0][lane] = w_frag(HQ + qq, 0);
      Wl[w][qq][1][lane] = w_frag(HQ + qq, 1);
    }
  }
  __syncthreads();

  if (w < 8) {
    // =============================== matrix waves ===============================
    const int i16 = lane & 15, kq = lane >> 4;
    float4 Wr[HQA][2];
#pragma unroll
    for (int qg = 0; qg < HQ; ++qg) { Wr[qg][0] = w_frag(qg, 0); Wr[qg][1] = w_frag(qg, 1); }
    // A fragments of item (s, T), this wave's K range: exchange buffer parity s & 1, one coherent 1 KB block load per q-group
    const unsigned lane_off = (unsigned)(i16 * 16 + kq * 4) * 4u + (unsigned)(w * NQ) * 1024u;
    auto load_item = [&](float4 (&A)[NQ], int s, int T) {
      const __amdgpu_buffer_rsrc_t hxr = image(s);
      const unsigned base = (unsigned)(T * QH) * 1024u + lane_off;
#pragma unroll
      for (int qg = 0; qg < NQ; ++qg)
        A[qg] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(hxr, (int)(base + (unsigned)qg * 1024u), 0, SH ? 0 : YT8M_AUX_LD));
    };
    float4 A0[NQ], A1[NQ], A2[NQ];
    if (PD >= 1) load_item(A0, 0, g);                   // items 0 (and 1) read the packed initial state: nothing to wait for
    if (PD >= 2 && total > 1) load_item(A1, 0, g + RB);
    int s_cur = 0, it_cur = 0;                          // item k = (s_cur, it_cur)
    // A = this item's fragments, Areq = the buffer the item PD ahead goes to
    auto item = [&](float4 (&A)[NQ], float4 (&Areq)[NQ], int k) {
      const int s = s_cur, T = g + it_cur * RB;
      STAMP(0);
      int sr = s, Tr = T, itr = it_cur;                 // item k + PD (the last PD items re-request themselves: no branch
      unsigned pv = 0;                                  // around the loads, that data is long published)
      if (PD >= 1) {
        itr = it_cur + PD;
        while (itr >= n_it) { itr -= n_it; ++sr; }
        const bool have = k + PD < total;
        Tr = have ? g + itr * RB : T;
        itr = have ? itr : it_cur;
        sr = have ? sr : s;
        // speculative poll: the counter read travels under the first MFMAs
        if (w == 0 && lane < NSH)
          pv = __hip_atomic_load(a.ctl + CTL_HDR + (Tr * NSH + lane) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (w == 0) {
          wait_tile(a.ctl, T, (unsigned)s * arrivals, lane);
          if (lane == 0) __hip_atomic_store(&lds_seen[it_cur], (unsigned)s * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          lds_wait_ge(&lds_seen[it_cur], (unsigned)s * arrivals, a.ctl);
        }
        load_item(A, s, T);
      }
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qg = 0; qg < NQ; ++qg) {
        if (PD >= 1 && qg == (NQ * YT8M_FWD_POLLQ) / 4) {   // request point: the state of item k + PD must be complete now
          if (w == 0) {
            const unsigned tot = shard_sum(pv);
            if (tot < (unsigned)sr * arrivals) wait_tile(a.ctl, Tr, (unsigned)sr * arrivals, lane);
            if (lane == 0) __hip_atomic_store(&lds_seen[itr], (unsigned)sr * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            lds_wait_ge(&lds_seen[itr], (unsigned)sr * arrivals, a.ctl);
          }
          load_item(Areq, sr, Tr);
          STAMP(1);
        }
        const float4 av = A[qg];
        float4 b0, b1;
        if (qg < HQ) { b0 = Wr[qg < HQ ? qg : 0][0]; b1 = Wr[qg < HQ ? qg : 0][1]; }
        else { b0 = Wl[w][qg - HQ][0][lane]; b1 = Wl[w][qg - HQ][1][lane]; }
#ifdef YT8M_MFMA_CUT   // timing experiment only (wrong results): 3 of 8 MFMAs, the matrix time of six bf16 products
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z + av.w, b0.z + b1.w, acc0, 0, 0, 0);
#else
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, acc1, 0, 0, 0);
#endif
      }
      STAMP(2);
      const int slot = k & (NSLOT - 1);
      if (k >= NSLOT) lds_wait_ge(&lds_free[slot], (unsigned)(k / NSLOT), a.ctl);   // the epilogue of item k - 4 has read its tiles
      float* rw = &red[slot][w][0][0][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) { rw[r * 64] = acc0[r]; rw[256 + r * 64] = acc1[r]; }
      // LDS executes a wave's operations in order: the arrival count lands behind the tile, no wait needed in between
      if (lane == 0) __hip_atomic_fetch_add(&lds_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      STAMP(3);
      if (++it_cur == n_it) { it_cur = 0; ++s_cur; }
    };
    if (PD == 2) {
      for (int k = 0; k < total; k += 3) {
        item(A0, A2, k);
        if (k + 1 < total) item(A1, A0, k + 1);
        if (k + 2 < total) item(A2, A1, k + 2);
      }
    } else {
      for (int k = 0; k < total; k += 2) {
        item(A0, A1, k);
        if (k + 1 < total) item(A1, A0, k + 1);
      }
    }
    return;
  }

  // =============================== epilogue waves ===============================
  // Epilogue wave ew owns the tiles it = ew, ew + 4, ... of this workgroup for ALL steps: the c / h it reloads at step s + 1
  // are its own stores of step s (same wave: program order), never another wave's.
  const int ew = w - 8;
  const int eunit = lane & 7;
  __builtin_amdgcn_s_setprio(YT8M_EPI_PRIO);             // the epilogue is the latency-critical chain: win VALU issue arbitration
  for (int s = 0; s < a.T; ++s) {
    const int t = a.t0 + s;
    for (int it = ew; it < n_it; it += NEPI) {
      const int k = s * n_it + it;
      const int T = g + it * RB;
      STAMP(0);
      // pairs j = 0, 1: (row in tile = 8 j + lane / 8, unit in group = lane % 8).  Operands that do not depend on the product
      // are requested before the wait; rows >= B are clamped (never stored) so that the loads stay branch-free.
      float zpre[2][4], cpre[2], hpre[2];
      bool live[2], evalid[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int brow = T * 16 + 8 * j + (lane >> 3);
        evalid[j] = brow < B;
        const int br = evalid[j] ? brow : B - 1;
        const float* zr = a.z + ((long long)t * B + br) * 4 * H + ug * 8 + eunit;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) zpre[j][g4] = zr[g4 * H];
        const long long idx = ((long long)t * B + br) * H + ug * 8 + eunit;
        cpre[j] = a.cs[idx];
        hpre[j] = a.hs[idx];
        live[j] = a.nf ? (t < a.nf[br]) : true;
      }
      const int slot = k & (NSLOT - 1);
      lds_wait_ge(&lds_cnt[slot], 8u * (unsigned)(k / NSLOT + 1), a.ctl);
      STAMP(1);
      // C layout of the 16x16 tile: column = lane & 15, row = 4 (lane >> 4) + r.  (row, unit): the four gate columns of the unit
      // are lanes 16 (row / 4) + 4 (unit % 4) + {0..3} of register r = row % 4 in half ct = unit / 4 -> one float4 per matrix wave.
      float4 sum[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int erow = 8 * j + (lane >> 3);
        const int ct = eunit >> 2, r = erow & 3, l0 = (erow >> 2) * 16 + (eunit & 3) * 4;
        sum[j] = *reinterpret_cast<const float4*>(&red[slot][0][ct][r][l0]);
#pragma unroll
        for (int wv = 1; wv < 8; ++wv) {
          const float4 p = *reinterpret_cast<const float4*>(&red[slot][wv][ct][r][l0]);
          sum[j].x += p.x; sum[j].y += p.y; sum[j].z += p.z; sum[j].w += p.w;
        }
      }
      // (in-order LDS: the release of the slot is queued behind the reads above)
      if (lane == 0) __hip_atomic_fetch_add(&lds_free[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      STAMP(2);
      // gate block (BasicLSTMCell: i | j | f | o, forget_bias on f), branch-free; sigmoid / tanh on v_exp_f32 + v_rcp_f32
      float gi[2], gj[2], gf[2], go[2], cn[2], hn[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        gi[j] = fast_sigmoid(zpre[j][0] + sum[j].x);
        gj[j] = fast_tanh(zpre[j][1] + sum[j].y);
        gf[j] = fast_sigmoid(zpre[j][2] + sum[j].z + a.fb);
        go[j] = fast_sigmoid(zpre[j][3] + sum[j].w);
        const float c1 = cpre[j] * gf[j] + gi[j] * gj[j];
        const float h1 = fast_tanh(c1) * go[j];
        cn[j] = live[j] ? c1 : cpre[j];                  // dynamic_rnn copy-through: the state passes, the output is zero
        hn[j] = live[j] ? h1 : hpre[j];
        if (!evalid[j]) hn[j] = 0.f;                     // rows >= B publish zeros
      }
#ifdef YT8M_PERSIST_TIMING
      asm volatile("" ::"v"(hn[0]), "v"(hn[1]), "v"(cn[0]), "v"(cn[1]));   // the stamp below must not move above the gate math
#endif
      STAMP(5);
      if (s + 1 < a.T) {                                 // publish h_t of this tile first: it is what the other workgroups wait for
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float h1 = row_shl<1>(hn[j]), h2 = row_shl<2>(hn[j]), h3 = row_shl<3>(hn[j]);
          if ((eunit & 3) == 0) {
            u32x4 v;
            v.x = __float_as_uint(hn[j]); v.y = __float_as_uint(h1); v.z = __float_as_uint(h2); v.w = __float_as_uint(h3);
            const int erow = 8 * j + (lane >> 3);
            const unsigned off = ((unsigned)(T * QH + (ug >> 1)) * 256u + (unsigned)(erow * 16 + (ug & 1) * 8 + eunit)) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(v, image(s + 1), (int)off, 0, YT8M_AUX_ST);
          }
        }
        STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // drain the write-through stores, then count the arrival
        if (lane == 0)
          __hip_atomic_fetch_add(a.ctl + CTL_HDR + (T * NSH + (blockIdx.x & (NSH - 1))) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        STAMP(4);
      }
      // everything the backward pass / the caller needs, in the standard layouts (nobody inside this launch waits for these)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (evalid[j]) {
          const int brow = T * 16 + 8 * j + (lane >> 3);
          const long long idx1 = ((long long)(t + 1) * B + brow) * H + ug * 8 + eunit;
          if (live[j]) {
            float* zr = a.z + ((long long)t * B + brow) * 4 * H + ug * 8 + eunit;
            zr[0] = gi[j]; zr[H] = gj[j]; zr[2 * H] = gf[j]; zr[3 * H] = go[j];
          }
          a.cs[idx1] = cn[j];
          a.hs[idx1] = hn[j];
          if (a.out) a.out[((long long)t * B + brow) * H + ug * 8 + eunit] = live[j] ? hn[j] : 0.f;
        }
      }
    }
  }
  if (ew == 0 && lane == 0) { check_placement(a.ctl, a.stats); propagate_error(a.ctl); }
}

// =====================================================================================================================
// Forward recurrence with the recurrent product on the bf16 pipe (the x3 scheme of gemm_x3.hip): h_t and W_h are split exactly
// into three bf16 planes and h . W_h accumulates, in fp32, the six partial products of weight >= 2^-16 -- fp32-grade results at
// 3/8 of the fp32-MFMA matrix time (v_mfma_f32_16x16x32_bf16: 48 MFMAs of 16 cycles per item and wave instead of 64 of 32).  The
// fp32 kernel above is matrix-bound for 64 % of a step and its epilogue shares the SIMDs with those MFMAs; with the matrix time of
// this kernel (timing experiment on the fp32 kernel with 3 of 8 MFMAs) a step takes 7.1 us instead of 10.3.
//   * the producer splits: an epilogue lane splits its h value and the eight units of a (row, workgroup) are gathered with DPP
//     into ONE 16-byte store per plane (an MFMA A fragment of the 32-wide K block the workgroup's units belong to); the exchange
//     image per step is [tile][H / 32][plane][4 k-groups x 16 rows][8 bf16] = 1 KiB blocks, 1.5x the fp32 image.
//   * the [H x 32] slice of W_h is split once per launch: 24 B fragments per wave (H = 1024), 10 in registers, 14 in LDS.
//   * A fragments travel in half items (three half buffers in rotation): the state of item k + 1 is requested in the middle of
//     item k, its second half into the registers the first half of item k has just left.  Two partial-tile slots (LDS: 112 KB of
//     weights + 32 KB of partial tiles).
// Needs one exchange image per step (plain L2-shared fetch), >= 2 tiles per workgroup and H in {512, 1024}; otherwise the fp32
// kernel runs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x4 as_u4(float4 v) {
  u32x4 r;
  r.x = __float_as_uint(v.x); r.y = __float_as_uint(v.y); r.z = __float_as_uint(v.z); r.w = __float_as_uint(v.w);
  return r;
}
// value of lane (l + N) mod 16 within a row of 16 lanes (DPP row_ror): four of them make a row-wide all-reduce
template <int N>
__device__ __forceinline__ float row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, true));
}

__device__ __forceinline__ unsigned bf16_rn_bits(float x) {        // round to nearest even, finite x
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3_bits(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  h1 = bf16_rn_bits(x);
  const float r1 = x - __uint_as_float(h1 << 16);                   // exact
  h2 = bf16_rn_bits(r1);
  h3 = bf16_rn_bits(r1 - __uint_as_float(h2 << 16));
}
template <int N>
__device__ __forceinline__ unsigned row_shl_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xF, 0xF, true);
}

// F16 form (round 5): h_t and W_h as TWO IEEE-half planes each (csrc/x3_image.h split_h2), three products hi.hi + hi.lo + lo.hi of
// v_mfma_f32_16x16x32_f16 -- the same fp32 grade as the six-product bf16 split at half the matrix instructions and 2/3 of the exchange
// bytes.  |h| < 1: the static scale 2^13; W_h under the power of two that brings max |W_h| (a device word) into [2^13, 2^14).
constexpr float FWD_H2_S = 8192.f;
// image 0 of a launch <- h_{t0-1} (standard layout), split into the three planes
template <int NP, bool F16 = false>
__global__ __launch_bounds__(256) void hx_pack_x3_kernel(const float* __restrict__ h, u32x4* __restrict__ img, int B, int H, int NT16) {
  const int KBH = H >> 5;
  const long long n = (long long)NT16 * KBH * NP * 64;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int l = (int)(e & 63), p = (int)((e >> 6) % NP);
    const long long blk = e / (64 * NP);
    const int kbg = (int)(blk % KBH), T = (int)(blk / KBH);
    const int row = T * 16 + (l & 15), k0 = kbg * 32 + (l >> 4) * 8;
    unsigned hb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned h1, h2, h3 = 0;
      const float hv = row < B ? h[(long long)row * H + k0 + j] : 0.f;
      if constexpr (F16) yt8m_x3::split_h2(hv * FWD_H2_S, h1, h2);
      else split3_bits(hv, h1, h2, h3);
      hb[j] = p == 0 ? h1 : (p == 1 ? h2 : h3);
    }
    u32x4 v;
    v.x = hb[0] | (hb[1] << 16); v.y = hb[2] | (hb[3] << 16); v.z = hb[4] | (hb[5] << 16); v.w = hb[6] | (hb[7] << 16);
    img[e] = v;
  }
}

// NP = 1 (round 4, --compute_dtype=bfloat16): ONE plane -- h_t travels as bf16 (a third of the exchange), W_h is rounded once per
// launch and is register-resident in full, one MFMA per (K block, column half) instead of six.
template <int NKB, int NP = 3, bool F16 = false>
__global__ __launch_bounds__(768) void lstm_persist_fwd_x3_kernel(PersistFwdArgs a) {
  static_assert(!F16 || NP == 2, "the f16 form has two planes");
  constexpr int NS = 2;                                  // partial-tile slots
#ifndef YT8M_X3_EPW
#define YT8M_X3_EPW 2
#endif
  constexpr int EPW = YT8M_X3_EPW;                       // epilogue waves per item (1: a whole tile per wave, 2: half a tile each)
  constexpr int NF = NKB * 2 * NP;                       // B fragments of a wave: [K block][column half][plane]
  constexpr int NREG = F16 ? NF : (NF < 10 ? NF : 10), NLDS = NF - NREG;   // (the f16 form's 16 fragments all fit in registers)
  constexpr int HK = NKB / 2;                            // K blocks per half item
  static_assert(NKB % 2 == 0, "half items");
  __shared__ __attribute__((aligned(16))) float red[NS][8][2][4][64];                 // 32 KB
  __shared__ __attribute__((aligned(16))) u32x4 Wl[8][NLDS > 0 ? NLDS : 1][64];       // 112 KB at H = 1024
  __shared__ unsigned lds_cnt[NS], lds_free[NS];
  __shared__ unsigned lds_seen[MAX_LOCAL_TILES];
  // num_frames of this workgroup's rows, read ONCE: as a global load per item it sat last in the epilogue's load queue, and the gate math
  // of every item waited for it (tools/fwd_h2_check.py: 0.7 us of a 6.3 us step in the f16 form)
  __shared__ int lds_nf[MAX_LOCAL_TILES * 16];
  // Round 6 (f16 form: its LDS is nearly empty): c_t / h_t of a (row, unit) pair are read back one step later by the lane that wrote them
  // (a tile always meets the same epilogue wave) -- they travel through this lane-private LDS slot instead of two global reads per
  // pair and step (tools: fwd_noch variant, 6.05 -> 5.55 us/step with those reads gone; profiles/r6_pmc_recur_tcc.txt)
  constexpr int CARRY_T = F16 ? 4 : 1;                   // tiles per epilogue wave that can be carried
  __shared__ float lds_ch[F16 ? 4 : 1][CARRY_T][2][2][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ug, g;
  {
    const int b = blockIdx.x;                            // same placement as lstm_persist_fwd_kernel
    if (a.per > 0 && (a.NU % (4 * a.per)) == 0) {
      const int x = b & 7, sl = b >> 3;
      g = x / a.per;
      ug = ((sl >> 2) * a.per + (x % a.per)) * 4 + (sl & 3);
    } else if (a.per > 0) { const int x = b & 7; g = x / a.per; ug = (b >> 3) * a.per + (x % a.per); }
    else { g = b / a.NU; ug = b % a.NU; }
  }
  const int H = a.H, B = a.B, NT16 = a.NT16, RB = a.RB;
  const int n_it = (NT16 - g + RB - 1) / RB;
  const int total = n_it * a.T;
  const int KBH = H >> 5;                                // 32-wide K blocks per row
  const long long img_f = (long long)NT16 * KBH * NP * 256;
  const unsigned img_bytes = (unsigned)(img_f * 4);
  auto image = [&](int s) -> __amdgpu_buffer_rsrc_t { return make_rsrc(a.hx + s * img_f, img_bytes); };
  const unsigned arrivals = (unsigned)a.NU * EPW;       // per (tile, step): EPW epilogue waves per workgroup
  // B fragment of v_mfma_f32_16x16x32_bf16: lane (n = lane & 15, kg = lane >> 4) supplies B[k = 8 kg + j][n], j = 0..7.
  // Column n of half ct <-> (unit 4 ct + n / 4, gate n % 4), as in the fp32 kernel.
  float f16_sw = 1.f;
  if constexpr (F16) f16_sw = yt8m_x3::pow2_scale_for(__uint_as_float(a.wword[0]), 14);
  const float f16_inv = F16 ? 1.0f / (f16_sw * FWD_H2_S) : 1.0f;
  auto w_frag = [&](int kb, int ct, int p) -> u32x4 {
    const int n16 = lane & 15, kg = lane >> 4;
    const long long k = (long long)(w * NKB + kb) * 32 + kg * 8;
    const long long col = (long long)(n16 & 3) * H + ug * 8 + ct * 4 + (n16 >> 2);
    const float* q = a.Wh + k * a.ldw + col;
    unsigned hb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned h1, h2, h3 = 0;
      if constexpr (F16) yt8m_x3::split_h2(q[j * a.ldw] * f16_sw, h1, h2);
      else split3_bits(q[j * a.ldw], h1, h2, h3);
      hb[j] = p == 0 ? h1 : (p == 1 ? h2 : h3);
    }
    u32x4 v;
    v.x = hb[0] | (hb[1] << 16); v.y = hb[2] | (hb[3] << 16); v.z = hb[4] | (hb[5] << 16); v.w = hb[6] | (hb[7] << 16);
    return v;
  };
  note_placement(a.ctl);
  if (tid < NS) { lds_cnt[tid] = 0; lds_free[tid] = 0; }
  for (int i = tid; i < MAX_LOCAL_TILES; i += 768) lds_seen[i] = 0;
  for (int i = tid; i < n_it * 16; i += 768) {
    const int brow = (g + (i >> 4) * RB) * 16 + (i & 15);
    lds_nf[i] = (a.nf && brow < B) ? a.nf[brow] : 0x7fffffff;
  }
  if (w < 8) {
#pragma unroll
    for (int f = NREG; f < NF; ++f) Wl[w][f - NREG][lane] = w_frag(f / (2 * NP), (f / NP) & 1, f % NP);
  }
  __syncthreads();

  if (w < 8) {
    // =============================== matrix waves ===============================
    u32x4 Wr[NREG];
#pragma unroll
    for (int f = 0; f < NREG; ++f) Wr[f] = w_frag(f / (2 * NP), (f / NP) & 1, f % NP);
    const unsigned lane_off = (unsigned)lane * 16u + (unsigned)(w * NKB) * (NP * 1024u);
    // half `half` (K blocks half * HK ..) of the A fragments of item (s, T): plain loads, one 1 KiB block per K block and plane
    auto load_half = [&](u32x4 (&Hh)[HK][NP], int s, int T, int half) {
      const __amdgpu_buffer_rsrc_t hxr = image(s);
      const unsigned base = (unsigned)(T * KBH) * (NP * 1024u) + lane_off + (unsigned)(half * HK) * (NP * 1024u);
#pragma unroll
      for (int kbl = 0; kbl < HK; ++kbl)
#pragma unroll
        for (int p = 0; p < NP; ++p)
          Hh[kbl][p] = __builtin_amdgcn_raw_buffer_load_b128(hxr, (int)(base + (unsigned)(kbl * NP + p) * 1024u), 0, 0);
    };
    u32x4 H0[HK][NP], H1[HK][NP], H2[HK][NP];
    load_half(H0, 0, g, 0);                              // item 0 reads the packed initial state: nothing to wait for
    load_half(H1, 0, g, 1);
    int s_cur = 0, it_cur = 0;
    // Ha / Hb: the halves of this item; the state of the next item goes to Hc (first half) and Ha (second half)
    auto item = [&](u32x4 (&Ha)[HK][NP], u32x4 (&Hb)[HK][NP], u32x4 (&Hc)[HK][NP], int k) {
      const int s = s_cur, T = g + it_cur * RB;
      STAMP(0);
      int sr = s, itr = it_cur + 1;
      if (itr >= n_it) { itr -= n_it; ++sr; }
      const bool have = k + 1 < total;
      const int Tr = have ? g + itr * RB : T;
      itr = have ? itr : it_cur;
      sr = have ? sr : s;
      unsigned pv = 0;                                   // speculative poll: the counter read travels under the first MFMAs
      if (w == 0 && lane < NSH)
        pv = __hip_atomic_load(a.ctl + CTL_HDR + (Tr * NSH + lane) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
      u32x4 ls[2][NP];
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (p >= NREG) ls[0][p] = Wl[w][p >= NREG ? p - NREG : 0][lane];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb == HK) {                                  // request point: the state of item k + 1 must be complete now
          if (w == 0) {
            const unsigned tot = shard_sum(pv);
            if (tot < (unsigned)sr * arrivals) wait_tile(a.ctl, Tr, (unsigned)sr * arrivals, lane);
            if (lane == 0) __hip_atomic_store(&lds_seen[itr], (unsigned)sr * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            lds_wait_ge(&lds_seen[itr], (unsigned)sr * arrivals, a.ctl);
          }
          STAMP(1);                                      // (the 3 NKB loads of the request go out three per MFMA group below)
        }
        u32x4 av[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) av[p] = kb < HK ? Ha[kb < HK ? kb : 0][p] : Hb[kb >= HK ? kb - HK : 0][p];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          // LDS-resident weight fragments travel one six-MFMA group ahead of their use (two register sets in rotation): read in
          // front of their own group every group began with an exposed LDS round trip, in a kernel paced by its item time
          const int gi = kb * 2 + ct;
          if (kb >= HK) {                                // the state of item k + 1: first half -> Hc, second half -> Ha (the MFMAs
            const __amdgpu_buffer_rsrc_t hxr = image(sr);   // that read Ha have been issued), three 1 KiB blocks per group
            const unsigned base = (unsigned)(Tr * KBH) * (NP * 1024u) + lane_off;
#pragma unroll
            for (int j = NP * (gi - NKB); j < NP * (gi - NKB) + NP; ++j) {
              const int kbl = (j % (NP * HK)) / NP, pl = j % NP;
              const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(hxr, (int)(base + (unsigned)j * 1024u), 0, 0);
              if (j < NP * HK) Hc[kbl][pl] = v; else Ha[kbl][pl] = v;
            }
          }
          if (gi + 1 < 2 * NKB) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              const int f = (gi + 1) * NP + p;
              if (f >= NREG) ls[(gi + 1) & 1][p] = Wl[w][f >= NREG ? f - NREG : 0][lane];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          u32x4 bv[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const int f = gi * NP + p;
            bv[p] = f < NREG ? Wr[f < NREG ? f : 0] : ls[gi & 1][p];
          }
          auto mm = [&](const u32x4& x, const u32x4& y, f32x4 c) -> f32x4 {
            if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
          };
          if constexpr (NP == 3) {
            acc[ct][0] = mm(av[0], bv[0], acc[ct][0]);
            acc[ct][1] = mm(av[0], bv[1], acc[ct][1]);
            acc[ct][0] = mm(av[1], bv[0], acc[ct][0]);
            acc[ct][1] = mm(av[0], bv[2], acc[ct][1]);
            acc[ct][0] = mm(av[1], bv[1], acc[ct][0]);
            acc[ct][1] = mm(av[2], bv[0], acc[ct][1]);
          } else if constexpr (NP == 2) {                   // hi hi + hi lo + lo hi
            acc[ct][0] = mm(av[0], bv[0], acc[ct][0]);
            acc[ct][1] = mm(av[0], bv[1], acc[ct][1]);
            acc[ct][0] = mm(av[1], bv[0], acc[ct][0]);
          } else {
            acc[ct][kb & 1] = mm(av[0], bv[0], acc[ct][kb & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      STAMP(2);
      const int slot = k & (NS - 1);
      if (k >= NS) lds_wait_ge(&lds_free[slot], (unsigned)(EPW * (k / NS)), a.ctl);   // every epilogue wave of item k - NS has read
      float* rw = &red[slot][w][0][0][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (F16) { rw[r * 64] = (acc[0][0][r] + acc[0][1][r]) * f16_inv; rw[256 + r * 64] = (acc[1][0][r] + acc[1][1][r]) * f16_inv; }
        else { rw[r * 64] = acc[0][0][r] + acc[0][1][r]; rw[256 + r * 64] = acc[1][0][r] + acc[1][1][r]; }
      }
      if (lane == 0) __hip_atomic_fetch_add(&lds_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      STAMP(3);
      if (++it_cur == n_it) { it_cur = 0; ++s_cur; }
    };
    for (int k = 0; k < total; k += 3) {
      item(H0, H1, H2, k);
      if (k + 1 < total) item(H2, H0, H1, k + 1);
      if (k + 2 < total) item(H1, H2, H0, k + 2);
    }
    return;
  }

  // =============================== epilogue waves (as in lstm_persist_fwd_kernel; the publish splits) ===============================
  // EPW epilogue waves share an item (EPW = 2: wave pair (ew >> 1) owns the tiles it = pair, pair + 2, ...; its two waves take
  // rows 0-7 / 8-15, ONE (row, unit) pair per lane, so the dependent gate / split / store sequence of the publish is half as long)
  const int ew = w - 8;
  const int eunit = lane & 7;
  constexpr int JP = 2 / EPW;                            // (row, unit) pairs per lane
  __builtin_amdgcn_s_setprio(YT8M_EPI_PRIO);
#ifdef YT8M_FWD_NO_CARRY
  const bool carry = false;
#else
  const bool carry = F16 && (n_it + NEPI / EPW - 1) / (NEPI / EPW) <= CARRY_T;
#endif
  for (int s = 0; s < a.T; ++s) {
    const int t = a.t0 + s;
    for (int it = EPW == 2 ? (ew >> 1) : ew; it < n_it; it += NEPI / EPW) {
      const int li = carry ? it / (NEPI / EPW) : 0;       // this wave's local index of the tile
      const int k = s * n_it + it;
      const int T = g + it * RB;
      STAMP(0);
      float zpre[JP][4], cpre[JP], hpre[JP];
      bool live[JP], evalid[JP];
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = EPW == 2 ? (ew & 1) : jj;
        const int brow = T * 16 + 8 * j + (lane >> 3);
        evalid[jj] = brow < B;
        const int br = evalid[jj] ? brow : B - 1;
        const float* zr = a.z + ((long long)t * B + br) * 4 * H + ug * 8 + eunit;
        const long long idx = ((long long)t * B + br) * H + ug * 8 + eunit;
#if defined(YT8M_FWD_EPI_NOLOAD) || defined(YT8M_FWD_EPI_NOZ)     // timing experiments only (wrong results)
        for (int g4 = 0; g4 < 4; ++g4) zpre[jj][g4] = 0.1f * (float)(g4 + eunit);
#elif defined(YT8M_FWD_ZPACK_T)   // timing experiment only (wrong results): the four gates of a (row, unit) pair as ONE 16-byte read, a row's
        {                             // eight units of this workgroup = one full 128-byte line (layout [row][unit][gate] instead of [row][gate][unit])
          const float4 zq = *reinterpret_cast<const float4*>(a.z + ((long long)t * B + br) * 4 * H + (ug * 8 + eunit) * 4);
          zpre[jj][0] = zq.x; zpre[jj][1] = zq.y; zpre[jj][2] = zq.z; zpre[jj][3] = zq.w;
          (void)zr;
        }
#else
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) zpre[jj][g4] = zr[g4 * H];
#endif
#if defined(YT8M_FWD_EPI_NOLOAD) || defined(YT8M_FWD_EPI_NOCH)
        cpre[jj] = 0.2f; hpre[jj] = 0.1f;
#else
        if (carry && s > 0) {
          cpre[jj] = lds_ch[F16 ? ew : 0][li][0][jj][lane];
          hpre[jj] = lds_ch[F16 ? ew : 0][li][1][jj][lane];
        } else {
          cpre[jj] = a.cs[idx];
          hpre[jj] = a.hs[idx];
        }
#endif
        live[jj] = t < lds_nf[it * 16 + 8 * j + (lane >> 3)];
      }
#ifdef YT8M_FWD_ZPREFETCH   // (opt-in: measured no gain, profiles/r6_recur_ab.txt)
      // Round 6: z[t] (the hoisted projection, read exactly once) comes from HBM, and the four unit groups that share each of its 128-byte
      // lines (32 columns of one gate) run on four CUs of one XCD and ask for the line at the same moment: all four hold a slot of their CU's
      // vector-memory window for the whole HBM round trip (profiles/r6_pmc_recur_tcc.txt: without these reads the step is 26 % shorter;
      // a CU's 64-request window x latency is what paces the kernel).  Each sharer therefore touches a QUARTER of the next step's lines
      // one step ahead (rows with row % 4 == ug % 4: one dword per line and gate): three of four demand reads become L2 hits.
      float zpf = 0.f;
      if (s + 1 < a.T) {
        const int j = EPW == 2 ? (ew & 1) : 0;
        constexpr int NPF = (EPW == 2 ? 2 : 4) * 4;        // rows of this wave with row % 4 == ug % 4, times four gates
        if (lane < NPF) {
          const int prow = T * 16 + 8 * j + 4 * (lane >> 2) + (ug & 3);
          if (prow < B) zpf = a.z[((long long)(t + 1) * B + prow) * 4 * H + (long long)(lane & 3) * H + (ug & ~3) * 8];
        }
      }
#endif
      const int slot = k & (NS - 1);
      lds_wait_ge(&lds_cnt[slot], 8u * (unsigned)(k / NS + 1), a.ctl);
      STAMP(1);
      float4 sum[JP];
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = EPW == 2 ? (ew & 1) : jj;
        const int erow = 8 * j + (lane >> 3);
        const int ct = eunit >> 2, r = erow & 3, l0 = (erow >> 2) * 16 + (eunit & 3) * 4;
        sum[jj] = *reinterpret_cast<const float4*>(&red[slot][0][ct][r][l0]);
#pragma unroll
        for (int wv = 1; wv < 8; ++wv) {
          const float4 p = *reinterpret_cast<const float4*>(&red[slot][wv][ct][r][l0]);
          sum[jj].x += p.x; sum[jj].y += p.y; sum[jj].z += p.z; sum[jj].w += p.w;
        }
      }
      if (lane == 0) __hip_atomic_fetch_add(&lds_free[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      STAMP(2);
      float gi[JP], gj[JP], gf[JP], go[JP], cn[JP], hn[JP];
#pragma unroll
      for (int j = 0; j < JP; ++j) {
        gi[j] = fast_sigmoid(zpre[j][0] + sum[j].x);
        gj[j] = fast_tanh(zpre[j][1] + sum[j].y);
        gf[j] = fast_sigmoid(zpre[j][2] + sum[j].z + a.fb);
        go[j] = fast_sigmoid(zpre[j][3] + sum[j].w);
        const float c1 = cpre[j] * gf[j] + gi[j] * gj[j];
        const float h1 = fast_tanh(c1) * go[j];
        cn[j] = live[j] ? c1 : cpre[j];
        hn[j] = live[j] ? h1 : hpre[j];
        if (!evalid[j]) hn[j] = 0.f;
        if (carry) { lds_ch[F16 ? ew : 0][li][0][j][lane] = cn[j]; lds_ch[F16 ? ew : 0][li][1][j][lane] = hn[j]; }
      }
      if (s + 1 < a.T) {                                 // publish h_t of this tile first, as three bf16 planes
        const __amdgpu_buffer_rsrc_t hxr = image(s + 1);
#pragma unroll
        for (int jj = 0; jj < JP; ++jj) {
          const int j = EPW == 2 ? (ew & 1) : jj;
          unsigned hb[3];
          if constexpr (F16) yt8m_x3::split_h2(hn[jj] * FWD_H2_S, hb[0], hb[1]);
          else if constexpr (NP == 3) split3_bits(hn[jj], hb[0], hb[1], hb[2]);
          else hb[0] = bf16_rn_bits(hn[jj]);
          const int erow = 8 * j + (lane >> 3);
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            // units 0..7 of this row are lanes l .. l + 7: pairs first, then the four pair words into the lane of unit 0
            const unsigned d = hb[p] | (row_shl_u<1>(hb[p]) << 16);
            const unsigned d2 = row_shl_u<2>(d), d4 = row_shl_u<4>(d), d6 = row_shl_u<6>(d);
            if (eunit == 0) {
              u32x4 v;
              v.x = d; v.y = d2; v.z = d4; v.w = d6;
              const unsigned off = ((unsigned)((T * KBH + (ug >> 2)) * NP + p) * 256u + (unsigned)(((ug & 3) * 16 + erow) * 4)) * 4u;
              __builtin_amdgcn_raw_buffer_store_b128(v, hxr, (int)off, 0, YT8M_AUX_ST);
            }
          }
        }
        STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_fetch_add(a.ctl + CTL_HDR + (T * NSH + ((blockIdx.x * EPW + (ew & (EPW - 1))) & (NSH - 1))) * 32, 1u,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        STAMP(4);
      }
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = EPW == 2 ? (ew & 1) : jj;
        if (evalid[jj]) {
          const int brow = T * 16 + 8 * j + (lane >> 3);
          const long long idx1 = ((long long)(t + 1) * B + brow) * H + ug * 8 + eunit;
#ifndef YT8M_FWD_EPI_NOSTORE  // timing experiment only (wrong results)
          if (live[jj]) {
#ifdef YT8M_FWD_ZPACK_T
            *reinterpret_cast<float4*>(a.z + ((long long)t * B + brow) * 4 * H + (ug * 8 + eunit) * 4) = make_float4(gi[jj], gj[jj], gf[jj], go[jj]);
#else
            float* zr = a.z + ((long long)t * B + brow) * 4 * H + ug * 8 + eunit;
            zr[0] = gi[jj]; zr[H] = gj[jj]; zr[2 * H] = gf[jj]; zr[3 * H] = go[jj];
#endif
          }
          a.cs[idx1] = cn[jj];
          a.hs[idx1] = hn[jj];
          if (a.out) a.out[((long long)t * B + brow) * H + ug * 8 + eunit] = live[jj] ? hn[jj] : 0.f;
#endif
        }
      }
#ifdef YT8M_FWD_ZPREFETCH
      asm volatile("" ::"v"(zpf));                          // (keeps the touch alive; its value is never used)
#endif
    }
  }
  if (ew == 0 && lane == 0) { check_placement(a.ctl, a.stats); propagate_error(a.ctl); }
}

// =====================================================================================================================
// Backward recurrence.  Step t (t_hi down to t_lo):  dL/dh_{t-1} = base_t + dz_t . W_h^T  ([B,4H] x [4H,H], K = 4H), then the
// BasicLSTM gate backward of step t-1 turns (dL/dh_{t-1} + dout_{t-1}, dL/dc_{t-1}) into dz_{t-1} (sequence.hip
// lstm_gates_bwd_kernel).  The output is only H wide but the reduction is 4H long, so a workgroup owns 16
// hidden units (one 16-column MFMA tile) and its [16 x 4H] slice of W_h -- 256 KB: half of every wave's K range sits in
// registers, the other half in LDS (128 KB).  dz_t (4H wide: 4x the forward's state) is what travels: "dzx" in A-fragment order,
// same write-through / arrival-counter protocol as the forward pass, fetched through a 16-slot register ring that always runs
// half an item ahead (mid-item poll of the next item, as in the forward kernel).
//   matrix waves 0-7 : K = 4H split 8-way (NQB = H/32 q-groups of 16 k each); q-groups [0, NQB/2) multiply register-resident
//                      weights, [NQB/2, NQB) LDS-resident ones (ds_read_b128, lane-linear = conflict-free); two accumulators.
//   epilogue waves 8-11: ALL four finish EVERY item together (wave e: rows 4e..4e+3 of the 16-row tile x 16 units, one pair per
//                      lane), so a pair's running (base, dc) state is always re-read by the lane that wrote it.
struct PersistBwdArgs {
  const float* gates;   // [F,B,4H] saved i|j|f|o
  const float* Wh;      // [H, ldw]
  long long ldw;
  const float* cs;      // [F+1,B,H]
  const float* dout;    // [F,B,H] or null
  float* dz;            // [F,B,4H]
  float* work;          // [4,B,H]: running (dh, dc) in halves 0 / 1
  float* dbrows;        // [B,4H] or null: per-row running sum of dz over the steps (the bias gradient before its sum over rows)
  const int32_t* nf;
  float* dzx;           // exchange images [nimg][NT16][4H/16][256]
  unsigned* ctl;
  unsigned* stats;
  int t0, T, B, H, phase;
  int NUB, RB, NT16, per, pf;
  int nimg;
  int bf;               // one-plane bf16 recurrent product requested (taken when the launch has the rotated, prefetching, image-per-step form)
  int h2;               // two-plane f16 recurrent product (H2 form below) requested; taken under the same conditions + room for the scales
  const unsigned* wword;// H2: max |W_h| as float bits (yt8m_h2_absmax)
  unsigned* sc;         // H2: per (publish, tile, producer workgroup, row quad) one word of four inverse-scale exponents
  unsigned sc_bytes;
  unsigned* rowmax;     // rotated epilogue, or null: max |dz[t, b, :]| as float bits by absolute frame row t B + b (atomicMax over the 64
                        // producer workgroups of a row; zeroed by the caller) -- what yt8m_h2_rowscales would measure in a pass over dz
  unsigned* partmax;    // ... or null: max |dz| of the whole launch into one word (atomicMax; what yt8m_h2_absmax would measure)
  unsigned* px;         // P2 (K-split workgroup pairs): partial-tile hand-off slots [2 parities][NT16][NUB][64 lanes][8 dwords] = {value, tag} granules
  unsigned px_bytes;
  unsigned nonce;       // ... a number no earlier launch on this workspace used: the tags of this launch are nonce * 8191 + step + 1
  unsigned long long* dbg;
  // IMG (rotated epilogue only): the operand images of this launch's dz written by the epilogue itself -- what yt8m_x3_split would
  // make of dz[t0 .. t0 + T) in separate passes (csrc/gemm_x3.hip image layout: 1 KiB blocks of 32 rows x 16 k per plane).
  float* img_plain;     // [T B rows, K = 4H]: A operand of dx = dz . W_x^T                       (or null)
  float* img_trans;     // [4H rows, K = T B]: B operand of dW = x^T dz                           (or null)
  float* img_trans_s;   // the same of diag(rowscale) dz: layer-0 weight gradient on uint8 frames (or null)
  const float* rowscale;// [F B] by absolute frame row t B + b
  float* colpart;       // [4 RB rows][4H] column sums of dz over this launch, one row per (row group, epilogue wave)   (or null)
  float* colpart_s;     // ... of diag(rowscale) dz                                                                    (or null)
};

constexpr int NSLOT_B = 3;

// round-to-nearest-even bf16 bits / exact three-term split: the arithmetic of csrc/gemm_x3.hip's split pass, bit for bit
__device__ __forceinline__ unsigned p_bf16_rn_bits(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void p_split3(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  h1 = p_bf16_rn_bits(x);
  const float r1 = x - __uint_as_float(h1 << 16);
  h2 = p_bf16_rn_bits(r1);
  const float r2 = r1 - __uint_as_float(h2 << 16);
  h3 = p_bf16_rn_bits(r2);
  if ((__float_as_uint(x) & 0x7F800000u) == 0x7F800000u) {
    h1 = __float_as_uint(x) >> 16;
    h2 = h3 = 0;
  }
}


// ROT (round 4): the four epilogue waves take whole items IN ROTATION (wave e finishes the items of the workgroup's tiles e, e + 4,
// ...: one lane = one unit x four rows) instead of finishing every item together.  An epilogue is two memory round trips long (the
// saved activations in, the write-through drain of dz out: ~7k cycles of a ~12k-cycle item); as a team of four the waves ran them
// strictly one item after the other, so item k + 1's partial tiles queued behind item k's drain and the wait went onto the chain of
// tile k + 1 through all 64 unit groups.  In rotation each wave has four item times for its epilogue, requests its next item's
// operands as soon as it has published the previous one, and the matrix waves only ever wait for the partial-tile slot.  Needs
// every workgroup to own a multiple of four tiles (a tile then always meets the same wave: its running (dh, dc) are re-read by the
// lanes that wrote them); the host falls back to the team form otherwise (YT8M_BWD_ROT=0 forces it).
// BF (round 4, --compute_dtype=bfloat16 on the native stack): the recurrent product on ONE bf16 plane -- dz travels as bf16 (round
// to nearest even, A fragments of v_mfma_f32_16x16x32_bf16: [tile][4H / 32][4 k-groups x 16 rows][8 bf16] = 1 KiB blocks, half the
// fp32 exchange), the workgroup's [16 x 4H] slice of W_h^T is rounded once per launch and is register-resident in full (64 VGPRs),
// a wave's K range is NQB / 2 blocks of 32: NQB / 2 MFMAs of 16 cycles per item instead of 4 NQB of 32.  dz as STORED (the operand of
// dx and of the weight gradients, which round it themselves) stays fp32; fp32 accumulation and gate arithmetic as in the fp32 form.
// H2 (round 5): the fp32-GRADE recurrent product off the fp32 pipe -- dz and W_h^T as TWO IEEE-half planes each (csrc/x3_image.h
// split_h2: x S = hi + lo to 2^-22), three products hi.hi + lo.hi + hi.lo of v_mfma_f32_16x16x32_f16: 48 MFMAs of 16 cycles per item and
// wave instead of 128 of 32, with the fp32 form's footprint everywhere else (4 bytes per exchanged dz element, a 256 KiB weight slice:
// the hi plane in registers, the lo plane in LDS).  What made this form wait for round 5 is the scale a half plane needs: dz spans
// decades from row to row and step to step, and a row's maximum over all 4H columns is spread over 64 producer workgroups.  Here every
// PRODUCER scales its own 16 rows x 64 values by a power of two per row (the maximum is one DPP reduction over the 16 lanes of a row),
// publishes the four inverse exponents of a row quad as one word beside the tile, and the K order of the exchange is permuted so that a
// producer's 64 values are two whole 32-wide K blocks: a consumer wave runs the six MFMAs of a producer into a scratch accumulator and
// adds it, times the producer's inverse scale of each row, to the running one (4 FMAs per 6 MFMAs).  No scale is guessed or lagged.
//   K slot (block 2 p + uh, k-group kg, j) of the exchange = producer p's value (unit 8 uh + 2 kg + j / 4, gate j % 4): a lane of the
//   epilogue (one unit x four gates per row) and its neighbour fill one 16-byte A fragment piece -- one DPP move per dword instead of the
//   one-plane form's three-step gather.
// P2 (round 6): K-SPLIT WORKGROUP PAIRS of the H2 form.  What paces the H2 kernel is the CU's vector-memory window: 64 line requests in
// flight x ~270 cycles each = ~31 B/clk, and every workgroup draws the dz of its 64 rows -- 1 MiB per step -- through it
// (profiles/r6_pmc_recur_tcc.txt).  The two workgroups (pair, kh = 0 / 1) that share a 128-byte line of gates (same XCD) now share the
// WORK differently: both compute the partial dh of the pair's 32 units, each over HALF of K (kh's 2H of the 4H dz columns = the producers
// [32 kh, 32 kh + 32)) -- the same 256 KiB weight footprint ([2H x 32] instead of [4H x 16]: hi plane in registers, lo plane in LDS),
// the same 48 MFMAs per item and wave, HALF the dz bytes per CU.  The price is one hand-off per item: each workgroup finishes its own 16
// units and needs the partner's partial tile for them -- 1 KiB as sixty-four {value, tag} granule quads written with sc0 sc1 stores into
// a slot the partner's epilogue wave polls with sc0 sc1 loads (no flag, no fence: the tag IS the arrival; MI355X_MICROARCH.md
// "handoff-1to1").  Sum order: K-half 0 + K-half 1, whichever workgroup finishes the unit -- bitwise reproducible run to run.
template <int NQB, bool PF, bool SH, bool ROT, bool IMG = false, bool BF = false, bool H2 = false, bool P2 = false>
__global__ __launch_bounds__(768) void lstm_persist_bwd_kernel(PersistBwdArgs a) {
  static_assert(!P2 || (H2 && (NQB % 16) == 0), "K-split pairs are a form of the two-half-plane kernel");
  static_assert(!IMG || ROT, "the operand images are written by the rotated epilogue");
  static_assert(!BF || (PF && SH && ROT && !IMG && (NQB % 4) == 0), "the bf16-operand form: prefetching, one image per step, rotated epilogue");
  static_assert(!H2 || (PF && SH && ROT && !IMG && !BF && (NQB % 8) == 0), "the two-half-plane form: prefetching, one image per step, rotated epilogue");
  constexpr int HALF = NQB / 2;                          // q-groups per wave in registers (= in LDS = ring slots)
  constexpr unsigned EPW = ROT ? 1u : 4u;                // epilogue waves that read a partial-tile slot / publish a tile
  constexpr int NSL = P2 ? 2 : NSLOT_B;                    // partial-tile slots (P2: two tiles of 16 x 32 per wave -> 16 KB per slot)
  constexpr int NRR = P2 ? 8 : 4;                          // accumulator registers a wave leaves per item
  __shared__ __attribute__((aligned(16))) float4 Wl[8][BF ? 1 : (P2 ? HALF - 1 : HALF)][64];  // LDS-resident half of the weights: 8 * HALF KB (BF: none; P2: one fragment per wave moves to registers)
  __shared__ __attribute__((aligned(16))) float red[NSL][8][NRR][64];      // [slot][wave][acc reg][lane]: 24 KB (P2: 32 KB)
  __shared__ unsigned lds_cnt[NSLOT_B], lds_free[NSLOT_B];
  __shared__ unsigned lds_seen[MAX_LOCAL_TILES];           // see the forward kernel: only matrix wave 0 polls memory
  __shared__ int lds_nf[ROT ? (P2 ? 8 : MAX_LOCAL_TILES) * 16 : 1];   // rotated epilogue: num_frames of the workgroup's rows, read once (as the forward kernel)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ub, g;
  {
    const int b = blockIdx.x;
    // as in the forward kernel: the two 16-unit groups that share a 128-byte line of gates / cs / dz sit on one XCD
    if (a.per > 0 && (a.NUB % (2 * a.per)) == 0) {
      const int x = b & 7, sl = b >> 3;
      g = x / a.per;
      ub = ((sl >> 1) * a.per + (x % a.per)) * 2 + (sl & 1);
    } else if (a.per > 0) { const int x = b & 7; g = x / a.per; ub = (b >> 3) * a.per + (x % a.per); }
    else { g = b / a.NUB; ub = b % a.NUB; }
  }
  const int H = a.H, B = a.B, NT16 = a.NT16, RB = a.RB;
  const int n_it = (NT16 - g + RB - 1) / RB;
  const int total = n_it * a.T;
  const int QH4 = H >> 2;                                // q-groups per dz row (4H / 16)
  const unsigned img_bytes = (unsigned)NT16 * (unsigned)H * (BF ? 32u : 64u) * 4u;     // BF: 2 bytes per dz element
  const long long img_f = (long long)NT16 * H * (BF ? 32 : 64);
  auto image = [&](int s) -> __amdgpu_buffer_rsrc_t { return make_rsrc(a.dzx + (SH ? s : (s & 1)) * img_f, img_bytes); };
  constexpr int AUX_LD = SH ? 0 : YT8M_AUX_LD;
  const unsigned arrivals = (unsigned)a.NUB * EPW;       // per (tile, publish): EPW epilogue waves per workgroup
  const int i16 = lane & 15, kq = lane >> 4;
  note_placement(a.ctl);
  if (tid < NSLOT_B) { lds_cnt[tid] = 0; lds_free[tid] = 0; }
  for (int i = tid; i < MAX_LOCAL_TILES; i += 768) lds_seen[i] = 0;
  if constexpr (ROT) {
    for (int i = tid; i < n_it * 16; i += 768) {
      const int brow = (g + (i >> 4) * RB) * 16 + (i & 15);
      lds_nf[i] = (a.nf && brow < B) ? a.nf[brow] : 0x7fffffff;
    }
  }
  // H2: B fragments of v_mfma_f32_16x16x32_f16 in the permuted K order: lane (n = unit, kg) supplies, for local K block kbl of this wave
  // (global block kbp = w NQB / 2 + kbl: producer kbp / 2, unit half kbp % 2), W_h[16 ub + n][gate (j % 4) H + 16 producer + 8 uh + 2 kg + j / 4]
  float h2_sw = 1.f;
  if constexpr (H2) h2_sw = yt8m_x3::pow2_scale_for(__uint_as_float(a.wword[0]), 14);
  // P2: fragment f = 2 kbl + nt of a wave = K block kh 2 NQB + w NQB / 4 + kbl of the pair's unit tile nt (units 16 (2 pair + nt) ..)
  auto w_frag_h2 = [&](int kbl, u32x4& fhi, u32x4& flo) {
    const int kbp = P2 ? (ub & 1) * 2 * NQB + w * (NQB / 4) + (kbl >> 1) : w * (NQB / 2) + kbl;
    const int urow = P2 ? ((ub & ~1) + (kbl & 1)) * 16 + i16 : ub * 16 + i16;
    const float* q = a.Wh + (long long)urow * a.ldw + 16 * (kbp >> 1) + 8 * (kbp & 1) + 2 * kq;
    unsigned hb[8], lb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) yt8m_x3::split_h2(q[(long long)(j & 3) * H + (j >> 2)] * h2_sw, hb[j], lb[j]);
    fhi.x = hb[0] | (hb[1] << 16); fhi.y = hb[2] | (hb[3] << 16); fhi.z = hb[4] | (hb[5] << 16); fhi.w = hb[6] | (hb[7] << 16);
    flo.x = lb[0] | (lb[1] << 16); flo.y = lb[2] | (lb[3] << 16); flo.z = lb[4] | (lb[5] << 16); flo.w = lb[6] | (lb[7] << 16);
  };
  if constexpr (H2) {
    if (w < 8) {
#pragma unroll
      for (int kbl = 0; kbl < (P2 ? HALF - 1 : HALF); ++kbl) {   // the lo plane of the whole slice lives in LDS, the hi plane in registers
        u32x4 fhi, flo;
        w_frag_h2(kbl, fhi, flo);
        Wl[w][kbl][lane] = as_f4(flo);
      }
    }
  }
  if (!BF && !H2 && w < 8) {
    // B fragment: lane (n = unit, kq) supplies W_h[16 ub + n][k = 16 q + 4 kq + e], e = 0..3: a float4 of a W_h row
    const float* wrow = a.Wh + (long long)(ub * 16 + i16) * a.ldw + (long long)(w * NQB) * 16 + kq * 4;
#pragma unroll
    for (int qq = 0; qq < HALF; ++qq) Wl[w][qq][lane] = *reinterpret_cast<const float4*>(wrow + (HALF + qq) * 16);
  }
  __syncthreads();

  if constexpr (BF) {
    if (w < 8) {
      // =============================== matrix waves, one bf16 plane ===============================
      constexpr int KBW = NQB / 2, HB = KBW / 2;           // 32-wide K blocks per wave / per half item
      // B fragment of v_mfma_f32_16x16x32_bf16: lane (n = unit, kg) supplies W_h[16 ub + n][k = 32 kb + 8 kg + j], j = 0..7
      const float* wrow = a.Wh + (long long)(ub * 16 + i16) * a.ldw + (long long)(w * KBW) * 32 + kq * 8;
      u32x4 Wb[KBW];
#pragma unroll
      for (int kb = 0; kb < KBW; ++kb) {
        const float4 lo = *reinterpret_cast<const float4*>(wrow + kb * 32), hi = *reinterpret_cast<const float4*>(wrow + kb * 32 + 4);
        Wb[kb].x = p_bf16_rn_bits(lo.x) | (p_bf16_rn_bits(lo.y) << 16);
        Wb[kb].y = p_bf16_rn_bits(lo.z) | (p_bf16_rn_bits(lo.w) << 16);
        Wb[kb].z = p_bf16_rn_bits(hi.x) | (p_bf16_rn_bits(hi.y) << 16);
        Wb[kb].w = p_bf16_rn_bits(hi.z) | (p_bf16_rn_bits(hi.w) << 16);
      }
      const int QB = H >> 3;                               // 32-wide K blocks per dz row (4H / 32)
      const unsigned lane_off = (unsigned)lane * 16u + (unsigned)(w * KBW) * 1024u;
      auto blk = [&](int T) -> unsigned { return (unsigned)(T * QB) * 1024u + lane_off; };
      u32x4 ring[HB];
      int s_cur = 0, it_cur = 0, slot = 0, gen = 0;
      {
        if (w == 0) {
          wait_tile(a.ctl, g, arrivals, lane);
          if (lane == 0) __hip_atomic_store(&lds_seen[0], arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          lds_wait_ge(&lds_seen[0], arrivals, a.ctl);
        }
        const unsigned b0 = blk(g);
        const __amdgpu_buffer_rsrc_t dx0 = image(0);
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) ring[kb] = __builtin_amdgcn_raw_buffer_load_b128(dx0, (int)(b0 + (unsigned)kb * 1024u), 0, 0);
      }
      for (int k = 0; k < total; ++k) {
        const int s = s_cur, T = g + it_cur * RB;
        int s1 = s, it1 = it_cur + 1;
        if (it1 == n_it) { it1 = 0; ++s1; }
        const bool have1 = k + 1 < total;
        const int T1 = have1 ? g + it1 * RB : T;
        s1 = have1 ? s1 : s;
        unsigned pv = 0;
        const unsigned bcur = blk(T);
        const __amdgpu_buffer_rsrc_t dxr = image(s);
        if (w == 0 && lane < NSH)
          pv = __hip_atomic_load(a.ctl + CTL_HDR + (T1 * NSH + lane) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // same placement as the fp32 form: first half -- refill (this item's second half) ahead of its MFMA, into fresh registers;
        // second half -- the refill of entry kb - 1 (the next item's first half) behind MFMA kb
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) {
          const bf16x8 av = __builtin_bit_cast(bf16x8, ring[kb]), bv = __builtin_bit_cast(bf16x8, Wb[kb]);
          ring[kb] = __builtin_amdgcn_raw_buffer_load_b128(dxr, (int)(bcur + (unsigned)(HB + kb) * 1024u), 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (kb & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        const int it1l = have1 ? it1 : it_cur;
        if (w == 0) {
          const unsigned tot = shard_sum(pv);
          if (tot < (unsigned)(s1 + 1) * arrivals) wait_tile(a.ctl, T1, (unsigned)(s1 + 1) * arrivals, lane);
          if (lane == 0) __hip_atomic_store(&lds_seen[it1l], (unsigned)(s1 + 1) * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          lds_wait_ge(&lds_seen[it1l], (unsigned)(s1 + 1) * arrivals, a.ctl);
        }
        const unsigned bnext = blk(T1);
        const __amdgpu_buffer_rsrc_t dxn = image(s1);
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) {
          const bf16x8 av = __builtin_bit_cast(bf16x8, ring[kb]), bv = __builtin_bit_cast(bf16x8, Wb[HB + kb]);
          if (kb & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc0, 0, 0, 0);
          if (kb >= 1) {
            __builtin_amdgcn_sched_barrier(0);
            ring[kb - 1] = __builtin_amdgcn_raw_buffer_load_b128(dxn, (int)(bnext + (unsigned)(kb - 1) * 1024u), 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        ring[HB - 1] = __builtin_amdgcn_raw_buffer_load_b128(dxn, (int)(bnext + (unsigned)(HB - 1) * 1024u), 0, 0);
        if (gen > 0) lds_wait_ge(&lds_free[slot], EPW * (unsigned)gen, a.ctl);
        float* rw = &red[slot][w][0][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[r * 64] = acc0[r] + acc1[r];
        if (lane == 0) __hip_atomic_fetch_add(&lds_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (++slot == NSLOT_B) { slot = 0; ++gen; }
        if (++it_cur == n_it) { it_cur = 0; ++s_cur; }
      }
      return;
    }
  } else if constexpr (H2 && P2) {
    if (w < 8) {
      // =============================== matrix waves, two half planes, K-split pair ===============================
      constexpr int KW = NQB / 4, HK = KW / 2;             // K blocks per wave (8 at H = 1024) / per half item
      constexpr int NPW = KW / 2, PH = NPW / 2;            // producers per wave (a producer = 2 K blocks) / per half item
      static_assert(HK >= 2 && (HK % 2) == 0, "whole producers per half item");
      const int kh = ub & 1;
      u32x4 Wh_[HALF];                                     // hi plane: fragment f = 2 kbl + nt
      u32x4 Wlast;                                         // lo plane of fragment HALF - 1 (the one that left the LDS)
#pragma unroll
      for (int f = 0; f < HALF; ++f) { u32x4 flo; w_frag_h2(f, Wh_[f], flo); if (f == HALF - 1) Wlast = flo; }
      const float inv_sw = 1.0f / h2_sw;
      const int lane_off = lane * 16;
      const int kbp0 = kh * 2 * NQB + w * KW;              // first global K block of this wave
      auto blk = [&](int T) -> int { return __builtin_amdgcn_readfirstlane((T * QH4 + 2 * kbp0) * 1024); };
      const __amdgpu_buffer_rsrc_t scr = make_rsrc(a.sc, a.sc_bytes);
      const int kq_off = kq * 4;
      auto sc_off = [&](int s, int T, int pl) -> int {
        return __builtin_amdgcn_readfirstlane(((s * NT16 + T) * a.NUB + (kbp0 >> 1) + pl) * 16);
      };
      u32x4 ring[2 * HK];                                  // half an item: HK K blocks x 2 planes
      unsigned sc[PH];
      int s_cur = 0, it_cur = 0, slot = 0, gen = 0;
      auto seen_wait = [&](int it, int T, unsigned target) {
        if (w == 0) {
          wait_tile(a.ctl, T, target, lane);
          if (lane == 0) __hip_atomic_store(&lds_seen[it], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          lds_wait_ge(&lds_seen[it], target, a.ctl);
        }
      };
      {
        seen_wait(0, g, arrivals);
        const int b0 = blk(g);
        const __amdgpu_buffer_rsrc_t dx0 = image(0);
#pragma unroll
        for (int q = 0; q < 2 * HK; ++q) ring[q] = __builtin_amdgcn_raw_buffer_load_b128(dx0, lane_off, b0 + q * 1024, 0);
#pragma unroll
        for (int pl = 0; pl < PH; ++pl) sc[pl] = __builtin_amdgcn_raw_buffer_load_b32(scr, kq_off, sc_off(0, g, pl), 0);
      }
      auto lo_frag = [&](int f) -> u32x4 { return f == HALF - 1 ? Wlast : as_u4(Wl[w][f < HALF - 1 ? f : 0][lane]); };
      u32x4 wl0 = lo_frag(0), wl1 = lo_frag(1);            // lo fragments of the NEXT K block (both unit tiles), one block ahead
      auto rescale = [&](f32x4& acc, const f32x4& tmp, unsigned e4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(tmp[r], __uint_as_float(((e4 >> (8 * r)) & 0xFFu) << 23), acc[r]);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
      };
      for (int k = 0; k < total; ++k) {
        const int s = s_cur, T = g + it_cur * RB;
        STAMP(0);
        int s1 = s, it1 = it_cur + 1;
        if (it1 == n_it) { it1 = 0; ++s1; }
        const bool have1 = k + 1 < total;
        const int T1 = have1 ? g + it1 * RB : T;
        s1 = have1 ? s1 : s;
        unsigned pv = 0;
        const int bcur = blk(T);
        const __amdgpu_buffer_rsrc_t dxr = image(s);
        if (w == 0 && lane < NSH)
          pv = __hip_atomic_load(a.ctl + CTL_HDR + (T1 * NSH + lane) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 tmp0 = {0.f, 0.f, 0.f, 0.f}, tmp1 = {0.f, 0.f, 0.f, 0.f};
        unsigned e_cur = 0;
        // one K block: its A fragments (hi, lo plane) against both unit tiles -- six MFMAs -- then the in-place refill of its two ring
        // entries and the lo fragments of the next block
        auto kblock = [&](int kbl, int kr, const __amdgpu_buffer_rsrc_t& rsrc, int rbase, bool refill, int sc_s, int sc_T, int sc_pl) {
          const f16x8 ah = __builtin_bit_cast(f16x8, ring[2 * kr]), al = __builtin_bit_cast(f16x8, ring[2 * kr + 1]);
          const f16x8 bh0 = __builtin_bit_cast(f16x8, Wh_[2 * kbl]), bh1 = __builtin_bit_cast(f16x8, Wh_[2 * kbl + 1]);
          const f16x8 bl0 = __builtin_bit_cast(f16x8, wl0), bl1 = __builtin_bit_cast(f16x8, wl1);
          if ((kbl & 1) == 0) e_cur = sc[(kr >> 1)];
          __builtin_amdgcn_sched_barrier(0);
          if ((kbl & 1) == 0) {
            tmp0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            tmp1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          } else {
            tmp0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, tmp0, 0, 0, 0);
            tmp1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, tmp1, 0, 0, 0);
          }
          tmp0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh0, tmp0, 0, 0, 0);
          tmp1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh1, tmp1, 0, 0, 0);
          tmp0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl0, tmp0, 0, 0, 0);
          tmp1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl1, tmp1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          const int fn = (2 * (kbl + 1)) % HALF;             // (the last block of an item fetches block 0's of the next)
          wl0 = lo_frag(fn);
          wl1 = lo_frag(fn + 1);
          if (refill) {
            ring[2 * kr] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, rbase + (2 * kr) * 1024, 0);
            ring[2 * kr + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, rbase + (2 * kr + 1) * 1024, 0);
          }
          if (kbl & 1) {
            if (refill) sc[(kr >> 1)] = __builtin_amdgcn_raw_buffer_load_b32(scr, kq_off, sc_off(sc_s, sc_T, sc_pl), 0);
            rescale(acc0, tmp0, e_cur);
            rescale(acc1, tmp1, e_cur);
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        // first half: ring entries are refilled with this item's second half
#pragma unroll
        for (int kb = 0; kb < HK; ++kb) kblock(kb, kb, dxr, bcur + 2 * HK * 1024, true, s, T, PH + (kb >> 1));
        {
          const int it1l = have1 ? it1 : it_cur;
          if (w == 0) {
            const unsigned tot = shard_sum(pv);
            if (tot < (unsigned)(s1 + 1) * arrivals) wait_tile(a.ctl, T1, (unsigned)(s1 + 1) * arrivals, lane);
            if (lane == 0) __hip_atomic_store(&lds_seen[it1l], (unsigned)(s1 + 1) * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            lds_wait_ge(&lds_seen[it1l], (unsigned)(s1 + 1) * arrivals, a.ctl);
          }
          STAMP(1);
        }
        const int bnext = blk(T1);
        const __amdgpu_buffer_rsrc_t dxn = image(s1);
        // second half: refilled with the next item's first half
#pragma unroll
        for (int kb = HK; kb < KW; ++kb) kblock(kb, kb - HK, dxn, bnext, true, s1, T1, (kb - HK) >> 1);
        STAMP(2);
        if (gen > 0) lds_wait_ge(&lds_free[slot], EPW * (unsigned)gen, a.ctl);
        float* rw = &red[slot][w][0][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) { rw[r * 64] = acc0[r] * inv_sw; rw[(4 + r) * 64] = acc1[r] * inv_sw; }
        if (lane == 0) __hip_atomic_fetch_add(&lds_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        STAMP(3);
        if (++slot == NSL) { slot = 0; ++gen; }
        if (++it_cur == n_it) { it_cur = 0; ++s_cur; }
      }
      return;
    }
  } else if constexpr (H2) {
    if (w < 8) {
      // =============================== matrix waves, two half planes ===============================
      constexpr int NPW = NQB / 4, PH = NPW / 2;           // producers per wave / per half item (a producer = 4 ring blocks: 2 K blocks x 2 planes)
      u32x4 Wh_[HALF];                                     // hi plane of the wave's HALF = NQB / 2 K blocks
#pragma unroll
      for (int kbl = 0; kbl < HALF; ++kbl) { u32x4 flo; w_frag_h2(kbl, Wh_[kbl], flo); }
      const float inv_sw = 1.0f / h2_sw;
      // (addresses: the lane's 16 bytes in the VGPR offset, everything uniform -- tile, wave, block -- in the scalar offset: with the
      // block index folded into per-load VGPR offsets the register allocator keeps a dozen base registers alive)
      const int lane_off = lane * 16;
#ifdef YT8M_TWIN_TEST    // timing experiment only (wrong results): waves 2j and 2j + 1 fetch the SAME K blocks -- what a twin-wave N = 32 form would move
      auto blk = [&](int T) -> int { return __builtin_amdgcn_readfirstlane((T * QH4 + (w & ~1) * NQB) * 1024); };
#else
      auto blk = [&](int T) -> int { return __builtin_amdgcn_readfirstlane((T * QH4 + w * NQB) * 1024); };
#endif
      const __amdgpu_buffer_rsrc_t scr = make_rsrc(a.sc, a.sc_bytes);
      const int kq_off = kq * 4;
      auto sc_off = [&](int s, int T, int pl) -> int { return __builtin_amdgcn_readfirstlane(((s * NT16 + T) * a.NUB + w * NPW + pl) * 16); };
      u32x4 ring[HALF];
      unsigned sc[PH];
      int s_cur = 0, it_cur = 0, slot = 0, gen = 0;
      auto seen_wait = [&](int it, int T, unsigned target) {
        if (w == 0) {
          wait_tile(a.ctl, T, target, lane);
          if (lane == 0) __hip_atomic_store(&lds_seen[it], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          lds_wait_ge(&lds_seen[it], target, a.ctl);
        }
      };
      {
        seen_wait(0, g, arrivals);
        const int b0 = blk(g);
        const __amdgpu_buffer_rsrc_t dx0 = image(0);
#pragma unroll
        for (int q = 0; q < HALF; ++q) ring[q] = __builtin_amdgcn_raw_buffer_load_b128(dx0, lane_off, b0 + (q) * 1024, 0);
#pragma unroll
        for (int pl = 0; pl < PH; ++pl) sc[pl] = __builtin_amdgcn_raw_buffer_load_b32(scr, kq_off, sc_off(0, g, pl), 0);
      }
      u32x4 wlb[2] = {as_u4(Wl[w][0][lane]), as_u4(Wl[w][1][lane])};
      auto rescale = [&](f32x4& acc, const f32x4& tmp, unsigned e4) {          // acc[r] += tmp[r] 2^(e_r - 127), e_r = byte r of e4
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(tmp[r], __uint_as_float(((e4 >> (8 * r)) & 0xFFu) << 23), acc[r]);
        // (pinned: left alone the compiler sinks all eight folds of an item behind its last MFMA and keeps eight scratch tiles alive)
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
      };
      for (int k = 0; k < total; ++k) {
        const int s = s_cur, T = g + it_cur * RB;
        STAMP(0);
        int s1 = s, it1 = it_cur + 1;
        if (it1 == n_it) { it1 = 0; ++s1; }
        const bool have1 = k + 1 < total;
        const int T1 = have1 ? g + it1 * RB : T;
        s1 = have1 ? s1 : s;
        unsigned pv = 0;
        const int bcur = blk(T);
        const __amdgpu_buffer_rsrc_t dxr = image(s);
        if (w == 0 && lane < NSH)
          pv = __hip_atomic_load(a.ctl + CTL_HDR + (T1 * NSH + lane) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        f32x4 tmp = {0.f, 0.f, 0.f, 0.f};
        unsigned e_cur = 0;
        // first half: producers 0 .. PH - 1 (K blocks 0 .. HALF / 2 - 1); their ring blocks are refilled with this item's second half.
        // The lo-plane weight fragment of a K block is read from LDS TWO K blocks ahead of its MFMA (wlb[kb & 1], carried over the item
        // boundary): read one block ahead, every K block waited ~100 cycles for it (tools/persist_timeline.py: 1.6k of an 8.1k-cycle item).
#pragma unroll
        for (int kb = 0; kb < HALF / 2; ++kb) {
          const int pl = kb >> 1;
          const f16x8 ah = __builtin_bit_cast(f16x8, ring[2 * kb]), al = __builtin_bit_cast(f16x8, ring[2 * kb + 1]);
          const f16x8 bh = __builtin_bit_cast(f16x8, Wh_[kb]), bl = __builtin_bit_cast(f16x8, wlb[kb & 1]);
          if ((kb & 1) == 0) e_cur = sc[pl];
          __builtin_amdgcn_sched_barrier(0);
          if ((kb & 1) == 0) tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, tmp, 0, 0, 0);
          tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, tmp, 0, 0, 0);
          tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, tmp, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          // refills IN PLACE behind the MFMAs that read the registers (the fp32 form loads ahead into fresh registers: 8 more VGPRs than
          // this form has)
          wlb[kb & 1] = as_u4(Wl[w][kb + 2][lane]);
#ifndef YT8M_HALF_TEST   // timing experiment only (wrong results): the second half of an item's dz is never fetched -- the data flow of a K-split pair
          ring[2 * kb] = __builtin_amdgcn_raw_buffer_load_b128(dxr, lane_off, bcur + (HALF + 2 * kb) * 1024, 0);
          ring[2 * kb + 1] = __builtin_amdgcn_raw_buffer_load_b128(dxr, lane_off, bcur + (HALF + 2 * kb + 1) * 1024, 0);
#endif
          if ((kb & 1) == 0) sc[pl] = __builtin_amdgcn_raw_buffer_load_b32(scr, kq_off, sc_off(s, T, PH + pl), 0);
          if (kb & 1) rescale(acc, tmp, e_cur);              // the producer's six MFMAs: its tile times its rows' inverse scales
          __builtin_amdgcn_sched_barrier(0);
        }
        // mid-item: dz (and the scales) of the next item must be complete before its fetch starts
        {
          const int it1l = have1 ? it1 : it_cur;
          if (w == 0) {
            const unsigned tot = shard_sum(pv);
            if (tot < (unsigned)(s1 + 1) * arrivals) wait_tile(a.ctl, T1, (unsigned)(s1 + 1) * arrivals, lane);
            if (lane == 0) __hip_atomic_store(&lds_seen[it1l], (unsigned)(s1 + 1) * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            lds_wait_ge(&lds_seen[it1l], (unsigned)(s1 + 1) * arrivals, a.ctl);
          }
          STAMP(1);
        }
        const int bnext = blk(T1);
        const __amdgpu_buffer_rsrc_t dxn = image(s1);
        // second half: producers PH .. NPW - 1; the ring is refilled with the next item's first half
#pragma unroll
        for (int kb = HALF / 2; kb < HALF; ++kb) {
          const int kr = kb - HALF / 2, plr = kr >> 1;
          const f16x8 ah = __builtin_bit_cast(f16x8, ring[2 * kr]), al = __builtin_bit_cast(f16x8, ring[2 * kr + 1]);
          const f16x8 bh = __builtin_bit_cast(f16x8, Wh_[kb]), bl = __builtin_bit_cast(f16x8, wlb[kb & 1]);
          if ((kb & 1) == 0) e_cur = sc[plr];
          __builtin_amdgcn_sched_barrier(0);
          if ((kb & 1) == 0) tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, tmp, 0, 0, 0);
          tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, tmp, 0, 0, 0);
          tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, tmp, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          wlb[kb & 1] = as_u4(Wl[w][(kb + 2) % HALF][lane]);   // (the last two: K blocks 0 and 1 of the next item)
          ring[2 * kr] = __builtin_amdgcn_raw_buffer_load_b128(dxn, lane_off, bnext + (2 * kr) * 1024, 0);
          ring[2 * kr + 1] = __builtin_amdgcn_raw_buffer_load_b128(dxn, lane_off, bnext + (2 * kr + 1) * 1024, 0);
          if (kb & 1) {
            sc[plr] = __builtin_amdgcn_raw_buffer_load_b32(scr, kq_off, sc_off(s1, T1, plr), 0);
            rescale(acc, tmp, e_cur);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(2);
        if (gen > 0) lds_wait_ge(&lds_free[slot], EPW * (unsigned)gen, a.ctl);
        float* rw = &red[slot][w][0][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[r * 64] = acc[r] * inv_sw;
        if (lane == 0) __hip_atomic_fetch_add(&lds_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        STAMP(3);
        if (++slot == NSLOT_B) { slot = 0; ++gen; }
        if (++it_cur == n_it) { it_cur = 0; ++s_cur; }
      }
      return;
    }
  } else if (w < 8) {
    // =============================== matrix waves ===============================
    const float* wrow = a.Wh + (long long)(ub * 16 + i16) * a.ldw + (long long)(w * NQB) * 16 + kq * 4;
    float4 Wr[HALF];
#pragma unroll
    for (int qg = 0; qg < HALF; ++qg) Wr[qg] = *reinterpret_cast<const float4*>(wrow + qg * 16);
    const unsigned lane_off = (unsigned)(i16 * 16 + kq * 4) * 4u + (unsigned)(w * NQB) * 1024u;
    auto blk = [&](int T) -> unsigned { return (unsigned)(T * QH4) * 1024u + lane_off; };
    float4 ring[HALF];
    int s_cur = 0, it_cur = 0, slot = 0, gen = 0;
    auto seen_wait = [&](int it, int T, unsigned target) {   // wave 0 polls memory and posts; the others wait on LDS
      if (w == 0) {
        wait_tile(a.ctl, T, target, lane);
        if (lane == 0) __hip_atomic_store(&lds_seen[it], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        lds_wait_ge(&lds_seen[it], target, a.ctl);
      }
    };
    if (PF) {                                            // first half of item 0: after every workgroup's prologue publish
      seen_wait(0, g, arrivals);
      const unsigned b0 = blk(g);
      const __amdgpu_buffer_rsrc_t dx0 = image(0);
#pragma unroll
      for (int q = 0; q < HALF; ++q)
        ring[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(dx0, (int)(b0 + (unsigned)q * 1024u), 0, AUX_LD));
    }
    for (int k = 0; k < total; ++k) {
      const int s = s_cur, T = g + it_cur * RB;
      STAMP(0);
      int s1 = s, it1 = it_cur + 1;
      if (it1 == n_it) { it1 = 0; ++s1; }
      const bool have1 = k + 1 < total;
      const int T1 = have1 ? g + it1 * RB : T;
      s1 = have1 ? s1 : s;
      unsigned pv = 0;
      const unsigned bcur = blk(T);
      const __amdgpu_buffer_rsrc_t dxr = image(s);
      if (PF) {
        if (w == 0 && lane < NSH)
          pv = __hip_atomic_load(a.ctl + CTL_HDR + (T1 * NSH + lane) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        seen_wait(it_cur, T, (unsigned)(s + 1) * arrivals);
#pragma unroll
        for (int q = 0; q < HALF; ++q)
          ring[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(dxr, (int)(bcur + (unsigned)q * 1024u), 0, AUX_LD));
      }
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      // (sched_barrier on either side of every q-group: left to itself hipcc issues the half's 64 MFMAs first and its 16 refill
      // loads in a clump behind them -- the loaded registers are the MFMAs' own sources -- so the next half starts by waiting a whole
      // L2 round trip for loads issued a moment earlier.  Pinned, each load goes out ahead of its group's MFMAs into four fresh
      // registers and has half an item to land.)
#pragma unroll
      for (int q = 0; q < HALF; ++q) {                   // first half: register-resident weights; refill with this item's 2nd half
        const float4 av = ring[q], bv = Wr[q];
        ring[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(dxr, (int)(bcur + (unsigned)(HALF + q) * 1024u), 0, AUX_LD));
        __builtin_amdgcn_sched_barrier(0);
        if (q & 1) {
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc1, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      float4 bvn = Wl[w][0][lane];                       // LDS-resident weights, one q-group ahead of their MFMAs
      unsigned bnext = bcur;
      __amdgpu_buffer_rsrc_t dxn = dxr;
      if (PF) {                                          // mid-item: dz of the next item must be complete before its fetch starts
        const int it1l = have1 ? it1 : it_cur;
        if (w == 0) {
          const unsigned tot = shard_sum(pv);
          if (tot < (unsigned)(s1 + 1) * arrivals) wait_tile(a.ctl, T1, (unsigned)(s1 + 1) * arrivals, lane);
          if (lane == 0) __hip_atomic_store(&lds_seen[it1l], (unsigned)(s1 + 1) * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          lds_wait_ge(&lds_seen[it1l], (unsigned)(s1 + 1) * arrivals, a.ctl);
        }
        bnext = blk(T1);
        dxn = image(s1);
        STAMP(1);
      }
#pragma unroll
      for (int q = 0; q < HALF; ++q) {                   // second half: LDS-resident weights; refill with the next item's 1st half
        const float4 av = ring[q];
        const float4 bv = bvn;
        // the next group's weights are read behind this group's FIRST MFMA: three MFMAs and a refill ahead of their use (read in
        // front of their own group, every group began with an exposed LDS round trip)
        if (q & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < HALF) bvn = Wl[w][q + 1][lane];
        __builtin_amdgcn_sched_barrier(0);
        if (q & 1) {
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc1, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc0, 0, 0, 0);
        }
        if (PF && q >= 1) {                              // one group late: ring[q - 1]'s registers are free by now
          __builtin_amdgcn_sched_barrier(0);
          ring[q - 1] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(dxn, (int)(bnext + (unsigned)(q - 1) * 1024u), 0, AUX_LD));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (PF) ring[HALF - 1] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(dxn, (int)(bnext + (unsigned)(HALF - 1) * 1024u), 0, AUX_LD));
      STAMP(2);
      if (gen > 0) lds_wait_ge(&lds_free[slot], EPW * (unsigned)gen, a.ctl);  // its epilogue wave(s) have read item k - 3
      float* rw = &red[slot][w][0][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) rw[r * 64] = acc0[r] + acc1[r];
      if (lane == 0) __hip_atomic_fetch_add(&lds_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      STAMP(3);
      if (++slot == NSLOT_B) { slot = 0; ++gen; }
      if (++it_cur == n_it) { it_cur = 0; ++s_cur; }
    }
    return;
  }

  // =============================== epilogue waves ===============================
  const int ew = w - 8;
  __builtin_amdgcn_s_setprio(YT8M_EPI_PRIO);
  if constexpr (ROT) {
    // lane = (row quad rq, unit): its four pairs are rows 4 rq + r, r = 0..3, of the tile -- exactly the four accumulator registers
    // of that lane in every matrix wave's partial tile (C layout: column = lane & 15, row = 4 (lane >> 4) + r)
    const int rq = lane >> 4, eunit = lane & 15;
    const int t_hi = a.t0 + a.T - 1;
    const long long BH = (long long)B * H;
    int pend_T = -1;
    auto arrive = [&]() {
      if (pend_T < 0) return;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0)
        __hip_atomic_fetch_add(a.ctl + CTL_HDR + (pend_T * NSH + ((blockIdx.x * 4 + ew) & (NSH - 1))) * 32, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      pend_T = -1;
    };
    auto publish_stores = [&](int T, int pub, const float (&dzv)[4][4]) {
      const __amdgpu_buffer_rsrc_t dxr = image(pub);
      if constexpr (H2) {
        // this workgroup's 16 rows x 64 values of the tile under one power of two per ROW (its maximum over the 16 units x 4 gates
        // into [2^13, 2^14)): blocks 2 ub + (unit / 8) of the permuted K order, both planes; the inverse exponents of the row quad in
        // one word (lane unit 0).  An all-zero row (a video that has ended, a padded row) keeps scale 1.
        const __amdgpu_buffer_rsrc_t scr = make_rsrc(a.sc, a.sc_bytes);
        unsigned ex = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float m = fmaxf(fmaxf(fabsf(dzv[r][0]), fabsf(dzv[r][1])), fmaxf(fabsf(dzv[r][2]), fabsf(dzv[r][3])));
          m = fmaxf(m, row_ror<8>(m));
          m = fmaxf(m, row_ror<4>(m));
          m = fmaxf(m, row_ror<2>(m));
          m = fmaxf(m, row_ror<1>(m));
          const unsigned eb = __float_as_uint(m) >> 23;       // m >= 0: the biased exponent; m = f 2^(eb - 126), f in [0.5, 1)
          const bool ok = m > 7.9e-31f && m < 3.0e38f;        // (yt8m_x3::pow2_scale_for's guard)
          const float S = __uint_as_float((ok ? 267u - eb : 127u) << 23);      // 2^(14 - (eb - 126))
          ex |= (ok ? eb - 13u : 127u) << (8 * r);            // biased exponent of 1 / S
          unsigned hb[4], lb[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) yt8m_x3::split_h2(dzv[r][g4] * S, hb[g4], lb[g4]);
          const unsigned h0 = hb[0] | (hb[1] << 16), h1 = hb[2] | (hb[3] << 16), l0 = lb[0] | (lb[1] << 16), l1 = lb[2] | (lb[3] << 16);
          const unsigned h2_ = row_shl_u<1>(h0), h3 = row_shl_u<1>(h1), l2 = row_shl_u<1>(l0), l3 = row_shl_u<1>(l1);
          if ((eunit & 1) == 0) {
            u32x4 vh, vl;
            vh.x = h0; vh.y = h1; vh.z = h2_; vh.w = h3;
            vl.x = l0; vl.y = l1; vl.z = l2; vl.w = l3;
            const unsigned off = (((unsigned)(T * (H >> 3)) + 2u * (unsigned)ub + (unsigned)(eunit >> 3)) * 2u) * 1024u +
                                 ((unsigned)((eunit >> 1) & 3) * 16u + (unsigned)(4 * rq + r)) * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(vh, dxr, (int)off, 0, YT8M_AUX_ST);
            __builtin_amdgcn_raw_buffer_store_b128(vl, dxr, (int)(off + 1024u), 0, YT8M_AUX_ST);
          }
        }
        if (eunit == 0)
          __builtin_amdgcn_raw_buffer_store_b32(ex, scr, (int)((((unsigned)(pub * NT16 + T) * (unsigned)a.NUB + (unsigned)ub) * 4u + (unsigned)rq) * 4u), 0,
                                                YT8M_AUX_ST);
        pend_T = T;
        return;
      }
      if constexpr (BF) {
        // A fragments of v_mfma_f32_16x16x32_bf16: block (T, kbg) = [4 k-groups][16 rows][8 bf16]; this workgroup's 16 units of gate
        // g4 are k = g4 H + 16 ub + unit: K block g4 (H / 32) + ub / 2, k-groups 2 (ub & 1) + unit / 8.  Eight units of a row = one
        // 16-byte piece, gathered along the 16-lane DPP row (lane = unit) into the lanes with unit % 8 == 0.
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const unsigned hb = p_bf16_rn_bits(dzv[r][g4]);
            const unsigned d0 = hb | (row_shl_u<1>(hb) << 16);           // units (u, u + 1) on even lanes
            const unsigned d1 = row_shl_u<2>(d0);
            const unsigned d2 = row_shl_u<4>(d0), d3 = row_shl_u<4>(d1);
            if ((eunit & 7) == 0) {
              u32x4 v;
              v.x = d0; v.y = d1; v.z = d2; v.w = d3;
              const unsigned kbg = (unsigned)(g4 * (H >> 5) + (ub >> 1));
              const unsigned kgrp = (unsigned)(((ub & 1) << 1) + (eunit >> 3));
              const unsigned off = ((unsigned)(T * (H >> 3)) + kbg) * 1024u + (kgrp * 16u + (unsigned)(4 * rq + r)) * 16u;
              __builtin_amdgcn_raw_buffer_store_b128(v, dxr, (int)off, 0, YT8M_AUX_ST);
            }
          }
        }
        pend_T = T;
        return;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float v0 = dzv[r][g4];
          const float v1 = row_shl<1>(v0), v2 = row_shl<2>(v0), v3 = row_shl<3>(v0);
          if ((eunit & 3) == 0) {
            u32x4 v;
            v.x = __float_as_uint(v0); v.y = __float_as_uint(v1); v.z = __float_as_uint(v2); v.w = __float_as_uint(v3);
            const unsigned off = ((unsigned)(T * QH4 + g4 * (H >> 4) + ub) * 256u + (unsigned)((4 * rq + r) * 16 + eunit)) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(v, dxr, (int)off, 0, YT8M_AUX_ST);
          }
        }
      }
      pend_T = T;
    };
    struct GateIn { float gi, gj, gf, go, cp, cn, dout; bool live; };
    // Round 6: with ONE tile per epilogue wave (n_it == 4: the headline's B = 128) a lane meets the same four (row, unit) pairs every
    // step, so what it wrote or read the step before travels in registers instead of through memory: the running (dh, dc) of `work`
    // (two loads + two stores per pair and step) and c_{t+1} = the c_t it read one step earlier.  Every request a CU does not make
    // frees a slot of its 64-request vector-memory window for the dz stream (profiles/r6_pmc_recur_tcc.txt).
#ifdef YT8M_BWD_NO_CARRY
    const bool carry = false;
#else
    const bool carry = !IMG && (n_it == 4);                // (the image-writing epilogue has no registers to spare: it keeps the loads)
#endif
    float car_base[4] = {0.f, 0.f, 0.f, 0.f}, car_dc[4] = {0.f, 0.f, 0.f, 0.f}, car_cn[4] = {0.f, 0.f, 0.f, 0.f};
    auto gate_load = [&](int t1, int br, int lrow, bool have_cn = false, float cnv = 0.f) -> GateIn {       // lrow: the row's slot in lds_nf (16 it + row of the tile)
      GateIn q;
#ifdef YT8M_EPI_NOLOAD   // timing experiment only (wrong results): the epilogue's saved-activation reads never reach memory
      q.gi = 0.5f; q.gj = 0.3f; q.gf = 0.6f; q.go = 0.4f; q.cp = 0.1f * (float)eunit; q.cn = 0.2f; q.dout = 0.01f; q.live = t1 < lds_nf[lrow];
      return q;
#endif
      const float* gr = a.gates + ((long long)t1 * B + br) * 4 * H + ub * 16 + eunit;
      const long long idx = (long long)t1 * BH + (long long)br * H + ub * 16 + eunit;
#if defined(YT8M_EPI_LD_NT)          // experiments: cache policy of the saved-activation reads (each line is read once, from HBM)
#define EPI_LD(p) __builtin_nontemporal_load(p)
#elif defined(YT8M_EPI_LD_SC1)
#define EPI_LD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define EPI_LD(p) (*(p))
#endif
      q.gi = EPI_LD(gr); q.gj = EPI_LD(gr + H); q.gf = EPI_LD(gr + 2 * H); q.go = EPI_LD(gr + 3 * H);
      q.cp = EPI_LD(a.cs + idx);
      q.cn = have_cn ? cnv : EPI_LD(a.cs + idx + BH);
      q.dout = a.dout ? EPI_LD(a.dout + idx) : 0.f;
      q.live = t1 < lds_nf[lrow];
      return q;
    };
    auto gate_bwd = [&](const GateIn& q, float dh_in, float dc, float (&dzv)[4], float& dc_out, float& base_out) {
      const float tc = fast_tanh(q.cn);
      const float dht = dh_in + q.dout;
      const float dct = dc + dht * q.go * (1.0f - tc * tc);
      dzv[0] = q.live ? dct * q.gj * q.gi * (1.0f - q.gi) : 0.f;
      dzv[1] = q.live ? dct * q.gi * (1.0f - q.gj * q.gj) : 0.f;
      dzv[2] = q.live ? dct * q.cp * q.gf * (1.0f - q.gf) : 0.f;
      dzv[3] = q.live ? dht * tc * q.go * (1.0f - q.go) : 0.f;
      dc_out = q.live ? dct * q.gf : dc;
      base_out = q.live ? 0.f : dh_in;
    };
    auto store_std = [&](int t1, int brow, const float (&dzv)[4], float dc_out, float base_out, int half, bool with_work = true) {
#ifdef YT8M_EPI_NOSTORE  // timing experiment only (wrong results): no standard-layout dz / running-state stores
      return;
#endif
      float* dzr = a.dz + ((long long)t1 * B + brow) * 4 * H + ub * 16 + eunit;
#ifdef YT8M_EPI_ST_NT
      __builtin_nontemporal_store(dzv[0], dzr); __builtin_nontemporal_store(dzv[1], dzr + H);
      __builtin_nontemporal_store(dzv[2], dzr + 2 * H); __builtin_nontemporal_store(dzv[3], dzr + 3 * H);
#else
      dzr[0] = dzv[0]; dzr[H] = dzv[1]; dzr[2 * H] = dzv[2]; dzr[3 * H] = dzv[3];
#endif
      if (a.dbrows) {
        float* db = a.dbrows + (long long)brow * 4 * H + ub * 16 + eunit;
        db[0] += dzv[0]; db[H] += dzv[1]; db[2 * H] += dzv[2]; db[3 * H] += dzv[3];
      }
      if (!with_work) return;
      float* wk = a.work + (long long)(2 * half) * BH + (long long)brow * H + ub * 16 + eunit;
      wk[0] = base_out;
      wk[BH] = dc_out;
    };
    // Row maxima of dz for the products that follow (a.rowmax / a.partmax): this workgroup's 64 values of each of the lane's four rows,
    // reduced over the 16 lanes of the DPP row; one atomicMax per (row, workgroup) on the row's word -- the bit pattern of a non-negative
    // float orders like the float -- issued AFTER the tile's arrival (nothing on the state's chain waits for them), and a running
    // maximum per lane that becomes one atomicMax per wave at the end of the launch.
    float part_m = 0.f;
    auto emit_rowmax = [&](int t1, int T, const float (&dzv)[4][4]) {
      if (!a.rowmax && !a.partmax) return;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float m = fmaxf(fmaxf(fabsf(dzv[r][0]), fabsf(dzv[r][1])), fmaxf(fabsf(dzv[r][2]), fabsf(dzv[r][3])));
        m = fmaxf(m, row_ror<8>(m));
        m = fmaxf(m, row_ror<4>(m));
        m = fmaxf(m, row_ror<2>(m));
        m = fmaxf(m, row_ror<1>(m));
        const int brow = T * 16 + 4 * rq + r;
        if (!(m < 3.0e38f)) m = 0.f;                       // (inf / nan: the products clamp; the word keeps a finite maximum)
        part_m = fmaxf(part_m, m);
        if (a.rowmax && eunit == 0 && brow < B && m > 0.f)
          __hip_atomic_fetch_max(a.rowmax + (long long)t1 * B + brow, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    // Operand images of dz (IMG): the wave holds a whole 16-row x 16-unit tile of every gate after the gate backward -- one
    // 16-wide K block of the transposed image (K = frame rows) for 16 image rows, and 16 rows of one K block of the plain image
    // (K = gate columns).  Issued AFTER the tile's arrival: nothing on the state's dependency chain waits for these ~500 VALU
    // operations, the wave has four item times until its next tile.  Image geometry: csrc/gemm_x3.hip store_block.
    float csum[4] = {0.f, 0.f, 0.f, 0.f}, csum_s[4] = {0.f, 0.f, 0.f, 0.f};
    auto emit_images = [&](int t1, int T, const float (&dzv)[4][4]) {
      if constexpr (!IMG) return;
      const int MKB = (a.T * B) >> 4;                     // K blocks of the transposed image (K = the launch's frame rows)
      const int mrow0 = (t1 - a.t0) * B + T * 16;         // first frame row of the tile within the launch
      float rs[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.rowscale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int brow = T * 16 + 4 * rq + r;
          rs[r] = brow < B ? a.rowscale[(long long)t1 * B + brow] : 0.f;
        }
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        unsigned h[3][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p_split3(dzv[r][g4], h[0][r], h[1][r], h[2][r]);
          csum[g4] += dzv[r][g4];
        }
        // transposed image: image row = gate column g4 H + 16 ub + unit, K block = this tile-step; this lane's four rows are
        // 8 bytes of the 16-byte half (rows 0-7 / 8-15 of the tile): [(row & 31) * 32 B][half slot * 16 B][(rq & 1) * 8 B]
        const int irow = g4 * H + ub * 16 + eunit;
        const int ir = irow & 31, isw = (ir >> 3) & 1;
        const long long tblk = ((long long)(irow >> 5) * MKB + (mrow0 >> 4)) * 768;            // floats: 3 planes x 256
        const int toff = ir * 8 + 4 * ((rq >> 1) ^ isw) + 2 * (rq & 1);
        if (a.img_trans) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            uint2 v;
            v.x = h[pl][0] | (h[pl][1] << 16);
            v.y = h[pl][2] | (h[pl][3] << 16);
            *reinterpret_cast<uint2*>(a.img_trans + tblk + pl * 256 + toff) = v;
          }
        }
        if (a.img_trans_s) {
          unsigned hs[3][4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv = dzv[r][g4] * rs[r];
            p_split3(sv, hs[0][r], hs[1][r], hs[2][r]);
            csum_s[g4] += sv;
          }
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            uint2 v;
            v.x = hs[pl][0] | (hs[pl][1] << 16);
            v.y = hs[pl][2] | (hs[pl][3] << 16);
            *reinterpret_cast<uint2*>(a.img_trans_s + tblk + pl * 256 + toff) = v;
          }
        }
        // plain image: image row = frame row, K block = g4 H / 16 + ub, k = unit: eight units of a row = one 16-byte half,
        // gathered along the 16-lane DPP row (lane = unit) into the lanes with unit % 8 == 0
        if (a.img_plain) {
          const int kbp = g4 * (H >> 4) + ub;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mrow0 + 4 * rq + r;
            const int pr = m & 31, psw = (pr >> 3) & 1;
            float* pb = a.img_plain + ((long long)(m >> 5) * (H >> 2) + kbp) * 768 + pr * 8 + 4 * ((eunit >> 3) ^ psw);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              const unsigned d0 = h[pl][r] | (row_shl_u<1>(h[pl][r]) << 16);       // units (u, u + 1) on even lanes
              const unsigned d1 = row_shl_u<2>(d0);
              const unsigned d2 = row_shl_u<4>(d0), d3 = row_shl_u<4>(d1);
              if ((eunit & 7) == 0) {
                uint4 v;
                v.x = d0; v.y = d1; v.z = d2; v.w = d3;
                *reinterpret_cast<uint4*>(pb + pl * 256) = v;
              }
            }
          }
        }
      }
    };
    // ---- prologue: gate backward of step t_hi from the caller's running (dh, dc) in half `phase`; publish #1 of this wave's tiles
    for (int it = ew; it < n_it; it += 4) {
      const int T = g + it * RB;
      float dzv[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int brow = T * 16 + 4 * rq + r;
        const bool valid = brow < B;
        const int br = valid ? brow : B - 1;
        const float* wk = a.work + (long long)(2 * a.phase) * BH + (long long)br * H + ub * 16 + eunit;
        const float dh0 = wk[0], dc0 = wk[BH];
        const GateIn q = gate_load(t_hi, br, it * 16 + 4 * rq + r);
        float dc_out, base_out;
        gate_bwd(q, dh0, dc0, dzv[r], dc_out, base_out);
        if (!valid) { dzv[r][0] = dzv[r][1] = dzv[r][2] = dzv[r][3] = 0.f; }
        if (valid) store_std(t_hi, brow, dzv[r], dc_out, base_out, a.phase ^ 1, !carry || a.T == 0);
        car_base[r] = base_out; car_dc[r] = dc_out; car_cn[r] = q.cp;
      }
      publish_stores(T, 0, dzv);
      arrive();
      emit_images(t_hi, T, dzv);
      emit_rowmax(t_hi, T, dzv);
    }
    for (int s = 0; s < a.T; ++s) {
      const int t1 = t_hi - s - 1;
      const bool last = s + 1 == a.T;
      const int half = (a.phase + s + 1) & 1;
      for (int it = ew; it < n_it; it += 4) {
        const int k = s * n_it + it;                      // n_it % 4 == 0: item k is always this wave's
        const int slot = k % NSL, gen = k / NSL;
        const int T = g + it * RB;
        STAMP(0);
        float base[4], dc[4];
        GateIn q[4];
        bool valid[4];
        int brow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                     // everything this item needs from memory, requested before the wait
          brow[r] = T * 16 + 4 * rq + r;
          valid[r] = brow[r] < B;
          const int br = valid[r] ? brow[r] : B - 1;
          if (carry) {
            base[r] = car_base[r];
            dc[r] = car_dc[r];
          } else {
            const float* wk = a.work + (long long)(2 * half) * BH + (long long)br * H + ub * 16 + eunit;
            base[r] = wk[0];
            dc[r] = wk[BH];
          }
          if (!last) q[r] = gate_load(t1, br, it * 16 + 4 * rq + r, carry, car_cn[r]);
        }
        lds_wait_ge(&lds_cnt[slot], 8u * (unsigned)(gen + 1), a.ctl);
        STAMP(1);
        float p[4];
        if constexpr (P2) {
          // this workgroup's K half of BOTH unit tiles: keep the own tile's, hand the other to the partner, take the partner's half of
          // the own tile.  Slot of (step parity, tile, DESTINATION workgroup): 64 lanes x {p0, tag, p1, tag | p2, tag, p3, tag}.
          const int kh = ub & 1;
          float q[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[r] = red[slot][0][4 * kh + r][lane];
            q[r] = red[slot][0][4 * (kh ^ 1) + r][lane];
#pragma unroll
            for (int wv = 1; wv < 8; ++wv) { p[r] += red[slot][wv][4 * kh + r][lane]; q[r] += red[slot][wv][4 * (kh ^ 1) + r][lane]; }
          }
          if (lane == 0) __hip_atomic_fetch_add(&lds_free[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const __amdgpu_buffer_rsrc_t pxr = make_rsrc(a.px, a.px_bytes);
          const unsigned tag = a.nonce * 8191u + (unsigned)s + 1u;
          const unsigned slot_b = (unsigned)(((s & 1) * NT16 + T) * a.NUB) * 2048u + (unsigned)lane * 32u;
          {
            u32x4 v0, v1;
            v0.x = __float_as_uint(q[0]); v0.y = tag; v0.z = __float_as_uint(q[1]); v0.w = tag;
            v1.x = __float_as_uint(q[2]); v1.y = tag; v1.z = __float_as_uint(q[3]); v1.w = tag;
            const unsigned dst = slot_b + (unsigned)(ub ^ 1) * 2048u;
            __builtin_amdgcn_raw_buffer_store_b128(v0, pxr, (int)dst, 0, YT8M_AUX_ST);
            __builtin_amdgcn_raw_buffer_store_b128(v1, pxr, (int)(dst + 16u), 0, YT8M_AUX_ST);
          }
          const unsigned src = slot_b + (unsigned)ub * 2048u;
          long long t_start = 0;
          for (unsigned spins = 0;; ++spins) {
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(pxr, (int)src, 0, YT8M_AUX_LD);
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(pxr, (int)(src + 16u), 0, YT8M_AUX_LD);
            const bool ok = v0.y == tag && v0.w == tag && v1.y == tag && v1.w == tag;
            if (__all(ok)) {
              const float o0 = __uint_as_float(v0.x), o1 = __uint_as_float(v0.z), o2 = __uint_as_float(v1.x), o3 = __uint_as_float(v1.z);
              if (kh == 0) { p[0] += o0; p[1] += o1; p[2] += o2; p[3] += o3; }       // K half 0 + K half 1, on either side
              else { p[0] = o0 + p[0]; p[1] = o1 + p[1]; p[2] = o2 + p[2]; p[3] = o3 + p[3]; }
              break;
            }
            if (spins >= 4) {
              __builtin_amdgcn_s_sleep(2);
              if ((spins & 63) == 4) {
                if (__hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                const long long now = wall_clock64();
                if (t_start == 0) t_start = now;
                else if (now - t_start > SPIN_TIMEOUT) {
                  if (lane == 0) {
                    __hip_atomic_store(a.ctl - CTL_STICKY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  }
                  break;
                }
              }
            }
          }
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = red[slot][0][r][lane];
#pragma unroll
          for (int wv = 1; wv < 8; ++wv) p[r] += red[slot][wv][r][lane];
        }
        if (lane == 0) __hip_atomic_fetch_add(&lds_free[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        STAMP(2);
        if (last) {                                       // dL/dh_{t_lo - 1}: handed to the caller (next chunk / initial state)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (valid[r]) {
              float* wk = a.work + (long long)(2 * half) * BH + (long long)brow[r] * H + ub * 16 + eunit;
              wk[0] = base[r] + p[r];
              if (carry) wk[BH] = dc[r];                    // (the carried form never stored it on the way)
            }
          }
          continue;
        }
        float dzv[4][4], dc_out[4], base_out[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gate_bwd(q[r], base[r] + p[r], dc[r], dzv[r], dc_out[r], base_out[r]);
          if (!valid[r]) { dzv[r][0] = dzv[r][1] = dzv[r][2] = dzv[r][3] = 0.f; }
        }
        STAMP(3);
        publish_stores(T, s + 1, dzv);
        arrive();                                         // drain + arrival: nothing else of this wave is waiting behind it
        STAMP(4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (valid[r]) store_std(t1, brow[r], dzv[r], dc_out[r], base_out[r], half ^ 1, !carry);
          car_base[r] = base_out[r]; car_dc[r] = dc_out[r]; car_cn[r] = q[r].cp;
        }
        emit_images(t1, T, dzv);
        emit_rowmax(t1, T, dzv);
      }
    }
    arrive();
    if (a.partmax) {                                       // the launch's maximum: lanes of unit 0 hold their row quads' running maxima
      float m = fmaxf(part_m, __shfl_xor(part_m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if (lane == 0 && m > 0.f) __hip_atomic_fetch_max(a.partmax, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (IMG) {
      // column sums of this wave's tiles over the launch: the four row quads of a unit meet in the lane with rq == 0 (fixed order)
      if (a.colpart || a.colpart_s) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float c = csum[g4], cs_ = csum_s[g4];
          const float c1 = __shfl(c, lane + 16, 64), c2 = __shfl(c, lane + 32, 64), c3 = __shfl(c, lane + 48, 64);
          const float s1 = __shfl(cs_, lane + 16, 64), s2 = __shfl(cs_, lane + 32, 64), s3 = __shfl(cs_, lane + 48, 64);
          if (rq == 0) {
            const long long o = (long long)(g * 4 + ew) * 4 * H + g4 * H + ub * 16 + eunit;
            if (a.colpart) a.colpart[o] = ((c + c1) + c2) + c3;
            if (a.colpart_s) a.colpart_s[o] = ((cs_ + s1) + s2) + s3;
          }
        }
      }
    }
    if (ew == 0 && lane == 0) { check_placement(a.ctl, a.stats); propagate_error(a.ctl); }
    return;
  }
  const int er = lane >> 4, eunit = lane & 15;           // pair of this lane: (row 4 ew + er of the tile, unit eunit of the group)
  const int erow = 4 * ew + er;
  const int t_hi = a.t0 + a.T - 1;
  const long long BH = (long long)B * H;
  // gate backward of step t1 for this lane's pair of tile T: consumes (dh_in, dc) and produces dz (4 gates), dc', base'
  // Publishing dz of an item = (1) write-through stores of its image bytes, (2) wait until they have left this CU, (3) one
  // arrival on the tile's counter.  (2) sits right behind (1): ~3.6k cycles of waiting per item in which all four epilogue waves
  // do nothing (tools/persist_timeline.py: epilogue 1.0k reduce + 2.7k gate math + 3.6k drain per item).  Round 3 tried to hide
  // it (-DYT8M_BWD_DEFER_ARRIVE: go on with the standard-layout stores of item k and the operand loads of item k + 1, arrive for
  // item k when those loads are back or at once if item k + 1's partial tiles are not there yet): correct, and NO faster (21.5
  // vs 21.4 us/step stand-alone, 23.3 vs 23.2 ms for the training step) -- the step time is the chain of a tile through ALL 64
  // unit-group workgroups of its row group (the slowest of 64 publishes gates every consumer), not the epilogue's throughput.
  int pend_T = -1;
  auto publish_stores = [&](int T, int pub, const float (&dzv)[4]) {
    const __amdgpu_buffer_rsrc_t dxr = image(pub);
    // dzx block of gate g4 = q-group g4 * (H/16) + ub: [16 rows][16 units]; four neighbouring lanes -> one 16-byte store
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float v0 = dzv[g4];
      const float v1 = row_shl<1>(v0), v2 = row_shl<2>(v0), v3 = row_shl<3>(v0);
      if ((eunit & 3) == 0) {
        u32x4 v;
        v.x = __float_as_uint(v0); v.y = __float_as_uint(v1); v.z = __float_as_uint(v2); v.w = __float_as_uint(v3);
        const unsigned off = ((unsigned)(T * QH4 + g4 * (H >> 4) + ub) * 256u + (unsigned)(erow * 16 + eunit)) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(v, dxr, (int)off, 0, YT8M_AUX_ST);
      }
    }
    pend_T = T;
  };
  auto arrive = [&]() {                                   // every vector memory operation of this wave issued so far has completed
    if (pend_T < 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
      __hip_atomic_fetch_add(a.ctl + CTL_HDR + (pend_T * NSH + ((blockIdx.x * 4 + ew) & (NSH - 1))) * 32, 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    pend_T = -1;
  };
  struct GateIn { float gi, gj, gf, go, cp, cn, dout; bool live; };
  auto gate_load = [&](int t1, int br) -> GateIn {
    GateIn q;
    const float* gr = a.gates + ((long long)t1 * B + br) * 4 * H + ub * 16 + eunit;
    q.gi = gr[0]; q.gj = gr[H]; q.gf = gr[2 * H]; q.go = gr[3 * H];
    const long long idx = (long long)t1 * BH + (long long)br * H + ub * 16 + eunit;
    q.cp = a.cs[idx];
    q.cn = a.cs[idx + BH];
    q.dout = a.dout ? a.dout[idx] : 0.f;
    q.live = a.nf ? (t1 < a.nf[br]) : true;
    return q;
  };
  auto gate_bwd = [&](const GateIn& q, float dh_in, float dc, float (&dzv)[4], float& dc_out, float& base_out) {
    const float tc = fast_tanh(q.cn);
    const float dht = dh_in + q.dout;
    const float dct = dc + dht * q.go * (1.0f - tc * tc);
    dzv[0] = q.live ? dct * q.gj * q.gi * (1.0f - q.gi) : 0.f;
    dzv[1] = q.live ? dct * q.gi * (1.0f - q.gj * q.gj) : 0.f;
    dzv[2] = q.live ? dct * q.cp * q.gf * (1.0f - q.gf) : 0.f;
    dzv[3] = q.live ? dht * tc * q.go * (1.0f - q.go) : 0.f;
    dc_out = q.live ? dct * q.gf : dc;
    base_out = q.live ? 0.f : dh_in;
  };
  auto store_std = [&](int t1, int brow, const float (&dzv)[4], float dc_out, float base_out, int half) {
    float* dzr = a.dz + ((long long)t1 * B + brow) * 4 * H + ub * 16 + eunit;
    dzr[0] = dzv[0]; dzr[H] = dzv[1]; dzr[2 * H] = dzv[2]; dzr[3 * H] = dzv[3];
    if (a.dbrows) {                                      // this lane owns (row, unit) for every step: a private running sum
      float* db = a.dbrows + (long long)brow * 4 * H + ub * 16 + eunit;
      db[0] += dzv[0]; db[H] += dzv[1]; db[2 * H] += dzv[2]; db[3 * H] += dzv[3];
    }
    float* wk = a.work + (long long)(2 * half) * BH + (long long)brow * H + ub * 16 + eunit;
    wk[0] = base_out;
    wk[BH] = dc_out;
  };
  // ---- prologue: gate backward of step t_hi from the caller's running (dh, dc) in half `phase`; publish #1 of every tile
  for (int it = 0; it < n_it; ++it) {
    const int T = g + it * RB;
    const int brow = T * 16 + erow;
    const bool valid = brow < B;
    const int br = valid ? brow : B - 1;
    const float* wk = a.work + (long long)(2 * a.phase) * BH + (long long)br * H + ub * 16 + eunit;
    const float dh0 = wk[0], dc0 = wk[BH];
    const GateIn q = gate_load(t_hi, br);
    float dzv[4], dc_out, base_out;
    gate_bwd(q, dh0, dc0, dzv, dc_out, base_out);
    if (!valid) { dzv[0] = dzv[1] = dzv[2] = dzv[3] = 0.f; }
    publish_stores(T, 0, dzv);
    if (valid) store_std(t_hi, brow, dzv, dc_out, base_out, a.phase ^ 1);
    arrive();
  }
  int slot = 0, gen = 0;
  for (int s = 0; s < a.T; ++s) {
    const int t1 = t_hi - s - 1;                         // the step whose gate backward follows the product of step t_hi - s
    const bool last = s + 1 == a.T;
    const int half = (a.phase + s + 1) & 1;              // where this step's (base, dc) live
    for (int it = 0; it < n_it; ++it) {
      const int k = s * n_it + it;
      (void)k;
      const int T = g + it * RB;
      const int brow = T * 16 + erow;
      const bool valid = brow < B;
      const int br = valid ? brow : B - 1;
      STAMP(0);
      float* wk = a.work + (long long)(2 * half) * BH + (long long)br * H + ub * 16 + eunit;
      const float base = wk[0], dc = wk[BH];
      GateIn q;
      if (!last) q = gate_load(t1, br);
#ifdef YT8M_BWD_DEFER_ARRIVE
      // the previous item's arrival: at once when this item's partial tiles are not there yet (nothing to overlap the drain with)
      if (pend_T >= 0 && lds_load(&lds_cnt[slot]) < 8u * (unsigned)(gen + 1)) arrive();
#endif
      lds_wait_ge(&lds_cnt[slot], 8u * (unsigned)(gen + 1), a.ctl);
      STAMP(1);
      // C layout: column = lane & 15 (unit), row = 4 (lane >> 4) + r  ->  pair (row 4 ew + er, unit): register er, lane 16 ew + unit
      float p = red[slot][0][er][ew * 16 + eunit];
#pragma unroll
      for (int wv = 1; wv < 8; ++wv) p += red[slot][wv][er][ew * 16 + eunit];
      if (lane == 0) __hip_atomic_fetch_add(&lds_free[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (++slot == NSLOT_B) { slot = 0; ++gen; }
      STAMP(2);
      const float dh_in = base + p;                      // (base / dc / the gate operands: loads issued at the top of the item)
      arrive();                                          // previous item: its image stores are older than those loads -> complete
      if (last) {                                        // dL/dh_{t_lo - 1}: handed to the caller (next chunk / initial state)
        if (valid) wk[0] = dh_in;
        continue;
      }
      float dzv[4], dc_out, base_out;
      gate_bwd(q, dh_in, dc, dzv, dc_out, base_out);
      if (!valid) { dzv[0] = dzv[1] = dzv[2] = dzv[3] = 0.f; }
      STAMP(3);
      publish_stores(T, s + 1, dzv);
#ifndef YT8M_BWD_DEFER_ARRIVE
      arrive();                                          // drain right behind the image stores (default; see above)
#endif
      STAMP(4);
      if (valid) store_std(t1, brow, dzv, dc_out, base_out, half ^ 1);
    }
  }
  arrive();
  if (ew == 0 && lane == 0) { check_placement(a.ctl, a.stats); propagate_error(a.ctl); }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// Persistent launches of one device may overlap only while their grids fit the chip TOGETHER (each needs every one of its
// workgroups resident, one per CU; a third grid taking CUs two others still wait for would hang all three).  The gate keeps the
// last four launches (completion event, CUs); a new launch waits for every recorded one that does not fit beside it, newest
// first.  Two half-chip backward recurrences of neighbouring layers thus run side by side, whole-chip forward launches chain.
#include "gru_persist.inl"

struct PersistGate {
  std::mutex mu;
  static constexpr int R = 4;
  hipEvent_t ev[16][R];
  int cus[16][R];          // 0: empty slot
  int head[16];
  bool init[16];
  PersistGate() { memset(cus, 0, sizeof(cus)); memset(head, 0, sizeof(head)); memset(init, 0, sizeof(init)); }
  // call with mu held, before the launch: makes `s` wait for what must finish first
  int admit(int dev, int need, int total, hipStream_t s) {
    static const bool chain_all = getenv("YT8M_PERSIST_CHAIN") != nullptr;   // A/B: strict one-at-a-time chaining
    if (!init[dev]) {
      for (int i = 0; i < R; ++i) YT8M_HIP_CHECK(hipEventCreateWithFlags(&ev[dev][i], hipEventDisableTiming));
      init[dev] = true;
    }
    int room = chain_all ? 0 : total - need;
    for (int i = 0; i < R; ++i) {
      const int k = (head[dev] + R - 1 - i) % R;                      // newest first
      if (cus[dev][k] == 0) continue;
      const bool oldest = i == R - 1;                                 // its slot is about to be reused
      if (!oldest && cus[dev][k] <= room) { room -= cus[dev][k]; continue; }
      room = 0;                                                       // everything older waits too
      YT8M_HIP_CHECK(hipStreamWaitEvent(s, ev[dev][k], 0));
    }
    return YT8M_OK;
  }
  // after the launch
  int done(int dev, int used, hipStream_t s) {
    const int k = head[dev];
    YT8M_HIP_CHECK(hipEventRecord(ev[dev][k], s));
    cus[dev][k] = used;
    head[dev] = (k + 1) % R;
    return YT8M_OK;
  }
};
PersistGate g_gate;

// placement statistics: one device word per device, allocated on first use; launches / workgroups counted on the host
unsigned* g_stats[16] = {nullptr};
int64_t g_stat_launches[16] = {0}, g_stat_wgs[16] = {0};
unsigned* stats_ptr(int dev) {
  if (!g_stats[dev]) {
    if (hipMalloc(reinterpret_cast<void**>(&g_stats[dev]), 64) != hipSuccess) return nullptr;
    (void)hipMemset(g_stats[dev], 0, 64);
  }
  return g_stats[dev];
}

// CUs a persistent launch may occupy: the whole chip, or YT8M_PERSIST_CUS of them (the rest stays free for kernels of other
// streams -- the hoisted GEMMs of the layer pipeline -- to run beside the recurrence)
std::atomic<int> g_pair_mode{-1};        // yt8m_lstm_persist_set_pair
int g_cap_fwd = -1, g_cap_bwd = -1;      // yt8m_lstm_persist_set_cus (calling thread's choice for its next launches); -1: environment
// CUs the gate keeps out of its residency arithmetic (yt8m_lstm_persist_reserve_cus): under data parallelism the RCCL kernels of
// the gradient all-reduce occupy CUs during the backward recurrences -- two half-chip launches admitted side by side would then
// not both fit, and the second would spin on the CUs it got until the collective's kernels leave (a slowdown, never a deadlock:
// RCCL's kernels do not wait for ours).  With a reserve the gate chains such launches instead.
int g_reserved_cus = getenv("YT8M_PERSIST_RESERVED_CUS") ? atoi(getenv("YT8M_PERSIST_RESERVED_CUS")) : 0;
int persist_cu_budget(int cus, bool bwd = false) {
  static const int cap = getenv("YT8M_PERSIST_CUS") ? atoi(getenv("YT8M_PERSIST_CUS")) : 0;
  static const int cap_b = getenv("YT8M_PERSIST_CUS_BWD") ? atoi(getenv("YT8M_PERSIST_CUS_BWD")) : 128;
  int c = bwd ? cap_b : cap;
  if (bwd && g_cap_bwd >= 0) c = g_cap_bwd;
  if (!bwd && g_cap_fwd >= 0) c = g_cap_fwd;
  return (c > 0 && c < cus) ? c : cus;
}

int device_cus(int* dev_out) {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
  }
  if (dev_out) *dev_out = dev;
  return cus[dev];
}

struct Geometry { int NQ, NU, RB, NT16, per, pf; };

bool persist_geometry(int64_t B, int64_t H, Geometry* geo) {
  static const bool off = getenv("YT8M_NO_PERSIST") != nullptr;
  if (off || B < 1 || H < 128 || (H % 128) != 0 || H > 1024) return false;
  const int NQ = (int)(H / 128);
  // (NQ odd > 1 would split the register / LDS halves unevenly: not instantiated.  NQ = 1 -- H = 128, the audio stack of the
  // parallel rgb / audio models, lstm_parallel_finaloutput_model.py:13-73 -- keeps its single q-group per wave in LDS.)
  if (!(NQ == 1 || NQ == 2 || NQ == 4 || NQ == 6 || NQ == 8)) return false;
  int dev = 0;
  const int cus = device_cus(&dev);
  const int NU = (int)(H / 8);
  if (cus < NU) return false;
  const int NT16 = (int)((B + 15) / 16);
  int RB = persist_cu_budget(cus) / NU;
  if (RB > NT16) RB = NT16;
  if (RB > NT16 / 4) RB = NT16 / 4;                      // >= 4 tiles per workgroup when the batch allows: four independent chains
  if (RB < 1) RB = 1;                                    // hide the exchange latency (publish -> visible -> fetched ~ 2 item times)
  if ((NT16 + RB - 1) / RB > MAX_LOCAL_TILES) return false;   // lds_seen: tiles one workgroup may own
  const int nit_min = NT16 / RB;                          // the last row group has floor(NT16 / RB) or one more
  int per = 0;
  if (RB <= 8 && (8 % RB) == 0 && (NU % (8 / RB)) == 0) per = 8 / RB;
  // two items ahead only with >= 8 chains: with 4 the producer of item k + 2 has barely published when it is requested
  if (geo) *geo = {NQ, NU, RB, NT16, per, nit_min >= 8 ? 2 : (nit_min >= 2 ? 1 : 0)};
  return true;
}

template <int NQ, bool SH>
int launch_fwd_sh(const PersistFwdArgs& a, unsigned grid, hipStream_t s) {
  if (a.pf >= 2) hipLaunchKernelGGL((lstm_persist_fwd_kernel<NQ, 2, SH>), dim3(grid), dim3(768), 0, s, a);
  else if (a.pf == 1) hipLaunchKernelGGL((lstm_persist_fwd_kernel<NQ, 1, SH>), dim3(grid), dim3(768), 0, s, a);
  else hipLaunchKernelGGL((lstm_persist_fwd_kernel<NQ, 0, SH>), dim3(grid), dim3(768), 0, s, a);
  return yt8m::launch_status("lstm_persist_fwd_kernel");
}
template <int NQ>
int launch_fwd(const PersistFwdArgs& a, unsigned grid, hipStream_t s) {
  return a.nimg >= a.T ? launch_fwd_sh<NQ, true>(a, grid, s) : launch_fwd_sh<NQ, false>(a, grid, s);
}

constexpr int64_t DBG_BYTES = 65536;     // tail of the workspace: s_memtime stamps of the timing variant
// workspace head: [sticky error line, 128 B, never zeroed by a launch][control block: zeroed before every launch], padded to 256 B
constexpr int64_t STICKY_BYTES = CTL_STICKY * 4;
int64_t ctl_bytes(int NT16) { return (int64_t)(CTL_HDR + NT16 * NSH * 32) * 4; }
int64_t ctl_padded(int NT16) { return ((STICKY_BYTES + ctl_bytes(NT16) + 255) / 256) * 256; }
// exchange images (of `width` = H forward, 4H backward) that fit in a workspace
int images_in(int64_t workspace_bytes, int NT16, int64_t width) {
  const int64_t n = (workspace_bytes - ctl_padded(NT16) - DBG_BYTES) / ((int64_t)NT16 * 16 * width * 4);
  return (int)std::min<int64_t>(n, 1 << 20);
}

}  // namespace

using namespace yt8m;

// CUs the following forward / backward launches (and shape queries) may occupy: 0 = the whole chip, -1 = the environment's choice
// (YT8M_PERSIST_CUS, YT8M_PERSIST_CUS_BWD; defaults: whole chip forward, 128 backward).  Process-wide tuning state: a caller that
// wants two layers' forward recurrences side by side sets 128 before launching them.
extern "C" int yt8m_lstm_persist_set_cus(int fwd_cus, int bwd_cus) {
  YT8M_REQUIRE(fwd_cus >= -1 && bwd_cus >= -1, YT8M_E_BADARG, "CU counts must be >= -1");
  g_cap_fwd = fwd_cus;
  g_cap_bwd = bwd_cus;
  return YT8M_OK;
}

// CUs to leave out of the gate's "do these persistent launches fit the chip together" arithmetic (0: none; the data-parallel
// reducer reserves some for the RCCL kernels that run beside the backward pass).  Returns the previous value through *previous.
extern "C" int yt8m_lstm_persist_reserve_cus(int cus, int* previous) {
  YT8M_REQUIRE(cus >= 0 && cus <= 4096, YT8M_E_BADARG, "CU count out of range");
  std::lock_guard<std::mutex> lk(g_gate.mu);
  if (previous) *previous = g_reserved_cus;
  g_reserved_cus = cus;
  return YT8M_OK;
}

extern "C" int yt8m_lstm_persist_supported(int64_t B, int64_t H) { return persist_geometry(B, H, nullptr) ? 1 : 0; }

namespace {
bool fwd_x3_shape(int64_t H, int pf) {
  static const bool x3_off = getenv("YT8M_PERSIST_X3") != nullptr && atoi(getenv("YT8M_PERSIST_X3")) == 0;
  return !x3_off && (H == 512 || H == 1024) && pf >= 1;
}
}  // namespace

// 1 when a forward launch on a per-step-image workspace (yt8m_lstm_persist_workspace_bytes_steps) runs the recurrent product as
// six bf16 products (lstm_persist_fwd_x3_kernel) rather than on the fp32 MFMA pipe
extern "C" int yt8m_lstm_persist_fwd_on_bf16_pipe(int64_t B, int64_t H) {
  Geometry geo;
  return (persist_geometry(B, H, &geo) && fwd_x3_shape(H, geo.pf)) ? 1 : 0;
}

extern "C" int64_t yt8m_lstm_persist_workspace_bytes(int64_t B, int64_t H) {
  Geometry geo;
  if (!persist_geometry(B, H, &geo)) return 0;
  // control block + two alternating exchange images sized for the BACKWARD pass (dz is 4H wide): [2][NT16][4H/16][256] floats
  return ctl_padded(geo.NT16) + (int64_t)2 * geo.NT16 * 16 * 4 * H * 4 + DBG_BYTES;
}

// The workspace that gives every step of a T-step launch its own exchange image (forward and backward): the launches then take
// the XCD-L2-shared fetch path (see lstm_persist_fwd_kernel).  Smaller workspaces (>= yt8m_lstm_persist_workspace_bytes) run the
// two-image protocol.
extern "C" int64_t yt8m_lstm_persist_workspace_bytes_steps(int64_t B, int64_t H, int64_t T) {
  Geometry geo;
  if (!persist_geometry(B, H, &geo)) return 0;
  // (+ the scale words of the f16 form of the backward recurrence: yt8m_lstm_persist_bwd_h2)
  // ... + the hand-off slots of its K-split pair form (two parities x tiles x H / 16 workgroups x 2 KiB)
  return ctl_padded(geo.NT16) + std::max<int64_t>(T, 2) * geo.NT16 * 16 * 4 * H * 4 + ((std::max<int64_t>(T, 2) * geo.NT16 * (H / 16) * 16 + 255) / 256) * 256 +
         (int64_t)2 * geo.NT16 * (H / 16) * 2048 + DBG_BYTES;
}

// Since the last reset on the current device: persistent launches, their workgroups, and how many of those did not run on the XCD
// their block index suggests (blockIdx % 8).  Synchronises the device.  reset != 0 zeroes the counters afterwards.
extern "C" int yt8m_lstm_persist_placement_stats(int64_t* launches, int64_t* workgroups, int64_t* off_xcd, int reset) {
  int dev = 0;
  device_cus(&dev);
  unsigned v = 0;
  if (g_stats[dev]) YT8M_HIP_CHECK(hipMemcpy(&v, g_stats[dev], 4, hipMemcpyDeviceToHost));
  if (launches) *launches = g_stat_launches[dev];
  if (workgroups) *workgroups = g_stat_wgs[dev];
  if (off_xcd) *off_xcd = v;
  if (reset) {
    if (g_stats[dev]) YT8M_HIP_CHECK(hipMemset(g_stats[dev], 0, 64));
    g_stat_launches[dev] = 0;
    g_stat_wgs[dev] = 0;
  }
  return YT8M_OK;
}

extern "C" int yt8m_lstm_persist_status(const void* workspace, yt8m_stream_t stream) {
  YT8M_REQUIRE(workspace, YT8M_E_BADARG, "null workspace");
  unsigned err = 0;
  YT8M_HIP_CHECK(hipMemcpyAsync(&err, workspace, 4, hipMemcpyDeviceToHost, as_stream(stream)));
  YT8M_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  if (err != 0) {                                        // sticky word: reported once, then cleared
    YT8M_HIP_CHECK(hipMemsetAsync(const_cast<void*>(workspace), 0, 4, as_stream(stream)));
    return fail(YT8M_E_HIP, "a persistent LSTM launch on this workspace timed out waiting for a tile since the last status call%s", "");
  }
  return YT8M_OK;
}

namespace {
__global__ void persist_fault_kernel(unsigned* ctl) {   // what a timed-out wait + the kernel's exit path do
  __hip_atomic_store(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  propagate_error(ctl);
}
}  // namespace

extern "C" int yt8m_lstm_persist_debug_fault(void* workspace, yt8m_stream_t stream) {
  YT8M_REQUIRE(workspace, YT8M_E_BADARG, "null workspace");
  hipLaunchKernelGGL(persist_fault_kernel, dim3(1), dim3(1), 0, as_stream(stream), static_cast<unsigned*>(workspace) + CTL_STICKY);
  return launch_status("persist_fault_kernel");
}

namespace {
int persist_fwd_impl(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out, const int32_t* num_frames, int64_t t0,
                     int64_t T, int64_t B, int64_t H, float forget_bias, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream,
                     bool bf16, const void* h2_wword = nullptr);
}
// yt8m_lstm_persist_fwd with the recurrent product h_{t-1} . W_h as THREE f16 products of two-half-plane splits (csrc/x3_image.h) instead
// of six bf16 products of three-plane splits: the same fp32 grade (2^-21 relative per term), half the matrix instructions, 4 instead of 6
// exchanged bytes per state element.  wh_absmax: device word with max |W_h| as float bits (yt8m_h2_absmax over the [H, 4H] block).  Taken
// where the six-product form would be (yt8m_lstm_persist_fwd_on_bf16_pipe); elsewhere, or with YT8M_PERSIST_FWD_H2=0, the launch is
// yt8m_lstm_persist_fwd.
extern "C" int yt8m_lstm_persist_fwd_h2(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out,
                                        const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias,
                                        const void* wh_absmax, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(wh_absmax, YT8M_E_BADARG, "null absmax word");
  static const bool off = getenv("YT8M_PERSIST_FWD_H2") != nullptr && atoi(getenv("YT8M_PERSIST_FWD_H2")) == 0;
  return persist_fwd_impl(z, Wh, ldw, cs, hs, out, num_frames, t0, T, B, H, forget_bias, workspace, workspace_bytes, stream, false,
                          off ? nullptr : wh_absmax);
}
extern "C" int yt8m_lstm_persist_fwd(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out,
                                     const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias,
                                     void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  return persist_fwd_impl(z, Wh, ldw, cs, hs, out, num_frames, t0, T, B, H, forget_bias, workspace, workspace_bytes, stream, false);
}
// yt8m_lstm_persist_fwd with the recurrent product h_{t-1} . W_h on ONE bf16 plane (h and W_h rounded to nearest even, fp32
// accumulation; --compute_dtype=bfloat16).  A permission: launches that cannot take the bf16-pipe kernel (H other than 512 / 1024,
// fewer than two 16-row tiles per workgroup, no room for one exchange image per step) run the fp32 form.
extern "C" int yt8m_lstm_persist_fwd_bf16(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out,
                                          const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias,
                                          void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  return persist_fwd_impl(z, Wh, ldw, cs, hs, out, num_frames, t0, T, B, H, forget_bias, workspace, workspace_bytes, stream, true);
}
namespace {
int persist_fwd_impl(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out, const int32_t* num_frames, int64_t t0,
                     int64_t T, int64_t B, int64_t H, float forget_bias, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream,
                     bool bf16, const void* h2_wword) {
  using namespace yt8m;
  YT8M_REQUIRE(t0 >= 0 && T >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (T * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(z && Wh && cs && hs && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldw >= 4 * H, YT8M_E_SHAPE, "ldw < 4H");
  Geometry geo;
  YT8M_REQUIRE(persist_geometry(B, H, &geo), YT8M_E_SHAPE, "shape not supported by the persistent recurrence (see yt8m_lstm_persist_supported)");
  YT8M_REQUIRE(workspace_bytes >= yt8m_lstm_persist_workspace_bytes(B, H), YT8M_E_SHAPE, "workspace too small");
  YT8M_REQUIRE(T < (1 << 20), YT8M_E_SHAPE, "T too large");
  hipStream_t s = as_stream(stream);
  const int64_t cb = ctl_padded(geo.NT16);
  PersistFwdArgs a;
  a.z = z; a.Wh = Wh; a.ldw = ldw; a.cs = cs; a.hs = hs; a.out = out; a.nf = num_frames;
  a.ctl = static_cast<unsigned*>(workspace) + CTL_STICKY;
  a.hx = reinterpret_cast<float*>(static_cast<char*>(workspace) + cb);
  a.t0 = (int)t0; a.T = (int)T; a.B = (int)B; a.H = (int)H; a.fb = forget_bias;
  a.NU = geo.NU; a.RB = geo.RB; a.NT16 = geo.NT16; a.per = geo.per; a.pf = geo.pf;
  a.nimg = images_in(workspace_bytes, geo.NT16, H);
  a.dbg = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + workspace_bytes - DBG_BYTES);
  const unsigned grid = (unsigned)(geo.NU * geo.RB);
  int dev = 0;
  device_cus(&dev);
  a.stats = stats_ptr(dev);
  ++g_stat_launches[dev];
  g_stat_wgs[dev] += grid;
  ProfScope prof(F_LSTM, s, 2.0 * (double)T * (double)B * (double)H * 4.0 * (double)H);
  std::lock_guard<std::mutex> lk(g_gate.mu);
  const int total_cus = device_cus(nullptr);
  int grc = g_gate.admit(dev, (int)grid, std::max(0, total_cus - g_reserved_cus), s);
  if (grc != YT8M_OK) return grc;
  YT8M_HIP_CHECK(hipMemsetAsync(a.ctl, 0, (size_t)ctl_bytes(geo.NT16), s));
  // the recurrent product on the bf16 pipe (lstm_persist_fwd_x3_kernel) when its preconditions hold: an exchange image (1.5x the
  // fp32 one) per step, >= 2 tiles per workgroup, H in {512, 1024}
  const bool x3 = fwd_x3_shape(H, geo.pf) && images_in(workspace_bytes, geo.NT16, H + H / 2) >= T;
  int rc;
  a.wword = static_cast<const unsigned*>(h2_wword);
  if (x3 && !bf16 && h2_wword) {                         // two half planes: image per step of the fp32 image's size
    hipLaunchKernelGGL((hx_pack_x3_kernel<2, true>), dim3(256), dim3(256), 0, s, hs + t0 * B * H, reinterpret_cast<u32x4*>(a.hx), (int)B, (int)H,
                       geo.NT16);
    rc = launch_status("hx_pack_x3_kernel");
    if (rc != YT8M_OK) return rc;
    if (H == 1024) hipLaunchKernelGGL((lstm_persist_fwd_x3_kernel<4, 2, true>), dim3(grid), dim3(768), 0, s, a);
    else hipLaunchKernelGGL((lstm_persist_fwd_x3_kernel<2, 2, true>), dim3(grid), dim3(768), 0, s, a);
    rc = launch_status("lstm_persist_fwd_x3_kernel");
  } else if (x3 && bf16) {
    hipLaunchKernelGGL(hx_pack_x3_kernel<1>, dim3(256), dim3(256), 0, s, hs + t0 * B * H, reinterpret_cast<u32x4*>(a.hx), (int)B, (int)H,
                       geo.NT16);
    rc = launch_status("hx_pack_x3_kernel");
    if (rc != YT8M_OK) return rc;
    if (H == 1024) hipLaunchKernelGGL((lstm_persist_fwd_x3_kernel<4, 1>), dim3(grid), dim3(768), 0, s, a);
    else hipLaunchKernelGGL((lstm_persist_fwd_x3_kernel<2, 1>), dim3(grid), dim3(768), 0, s, a);
    rc = launch_status("lstm_persist_fwd_x3_kernel");
  } else if (x3) {
    hipLaunchKernelGGL(hx_pack_x3_kernel<3>, dim3(256), dim3(256), 0, s, hs + t0 * B * H, reinterpret_cast<u32x4*>(a.hx), (int)B, (int)H,
                       geo.NT16);
    rc = launch_status("hx_pack_x3_kernel");
    if (rc != YT8M_OK) return rc;
    if (H == 1024) hipLaunchKernelGGL((lstm_persist_fwd_x3_kernel<4>), dim3(grid), dim3(768), 0, s, a);
    else hipLaunchKernelGGL((lstm_persist_fwd_x3_kernel<2>), dim3(grid), dim3(768), 0, s, a);
    rc = launch_status("lstm_persist_fwd_x3_kernel");
  } else {
    hipLaunchKernelGGL(hx_pack_kernel, dim3(256), dim3(256), 0, s, hs + t0 * B * H, a.hx, (int)B, (int)H, geo.NT16);
    rc = launch_status("hx_pack_kernel");
    if (rc != YT8M_OK) return rc;
    switch (geo.NQ) {
      case 1: rc = launch_fwd<1>(a, grid, s); break;
      case 2: rc = launch_fwd<2>(a, grid, s); break;
      case 4: rc = launch_fwd<4>(a, grid, s); break;
      case 6: rc = launch_fwd<6>(a, grid, s); break;
      default: rc = launch_fwd<8>(a, grid, s); break;
    }
  }
  if (rc != YT8M_OK) return rc;
  return g_gate.done(dev, (int)grid, s);
}
}  // namespace

namespace {
struct GeometryB { int NQB, NUB, RB, NT16, per, pf; };

bool persist_geometry_bwd(int64_t B, int64_t H, GeometryB* geo) {
  static const bool off = getenv("YT8M_NO_PERSIST") != nullptr || getenv("YT8M_NO_PERSIST_BWD") != nullptr;
  if (off || B < 1 || H > 1024 || !(H == 128 || (H >= 256 && (H % 256) == 0))) return false;
  const int NQB = (int)(H / 32);
  if (!(NQB == 4 || NQB == 8 || NQB == 16 || NQB == 24 || NQB == 32)) return false;
  int dev = 0;
  const int cus = device_cus(&dev);
  const int NUB = (int)(H / 16);
  if (cus < NUB) return false;
  const int NT16 = (int)((B + 15) / 16);
  // default: HALF the chip.  With 16 units per workgroup the whole chip leaves two 16-row tiles (= two independent chains) per
  // workgroup at B = 128 and the exchange latency is exposed on every item (42 us / step measured); on 128 CUs a workgroup
  // owns four tiles, runs 24 us / step -- the per-step kernels' speed on the whole chip -- and the other 128 CUs stay free for
  // the weight-gradient GEMMs of the layer pipeline.
  int RB = persist_cu_budget(cus, true) / NUB;
  if (RB > NT16 / 2) RB = NT16 / 2;                      // >= 2 tiles per workgroup when the batch allows: one chain's exchange
  if (RB < 1) RB = 1;                                    // latency hides behind the other chain's matrix work
  if ((NT16 + RB - 1) / RB > MAX_LOCAL_TILES) return false;
  const int nit_min = NT16 / RB;
  int per = 0;
  if (RB <= 8 && (8 % RB) == 0 && (NUB % (8 / RB)) == 0) per = 8 / RB;
  if (geo) *geo = {NQB, NUB, RB, NT16, per, nit_min >= 2 ? 1 : 0};
  return true;
}

bool bwd_rot_off() {
  static const bool off = getenv("YT8M_BWD_ROT") != nullptr && atoi(getenv("YT8M_BWD_ROT")) == 0;
  return off;
}
// rotation needs a multiple of four tiles in EVERY workgroup (row group g owns tiles g, g + RB, ...) and the prefetching form
bool bwd_rot(int pf, int NT16, int RB) { return !bwd_rot_off() && pf && (NT16 % (4 * RB)) == 0; }

template <int NQB, bool SH>
int launch_bwd_sh(const PersistBwdArgs& a, unsigned grid, hipStream_t s) {
  const bool rot = bwd_rot(a.pf, a.NT16, a.RB);
  const bool img = a.img_plain || a.img_trans || a.img_trans_s || a.colpart || a.colpart_s;
  if (img) {
    if constexpr (SH) {
      if (!rot) return yt8m::fail(YT8M_E_SHAPE, "operand images need the rotated backward epilogue%s", "");
      hipLaunchKernelGGL((lstm_persist_bwd_kernel<NQB, true, true, true, true>), dim3(grid), dim3(768), 0, s, a);
      return yt8m::launch_status("lstm_persist_bwd_kernel");
    } else {
      return yt8m::fail(YT8M_E_SHAPE, "operand images need one exchange image per step%s", "");
    }
  }
  if constexpr (SH && (NQB == 16 || NQB == 32)) {
    if (a.bf && rot) {
      hipLaunchKernelGGL((lstm_persist_bwd_kernel<NQB, true, true, true, false, true>), dim3(grid), dim3(768), 0, s, a);
      return yt8m::launch_status("lstm_persist_bwd_kernel");
    }
  }
  if constexpr (SH && (NQB == 16 || NQB == 32)) {
    if (a.h2 && rot) {
      if (a.px) hipLaunchKernelGGL((lstm_persist_bwd_kernel<NQB, true, true, true, false, false, true, true>), dim3(grid), dim3(768), 0, s, a);
      else hipLaunchKernelGGL((lstm_persist_bwd_kernel<NQB, true, true, true, false, false, true>), dim3(grid), dim3(768), 0, s, a);
      return yt8m::launch_status("lstm_persist_bwd_kernel");
    }
  }
  if (rot) hipLaunchKernelGGL((lstm_persist_bwd_kernel<NQB, true, SH, true>), dim3(grid), dim3(768), 0, s, a);
  else if (a.pf) hipLaunchKernelGGL((lstm_persist_bwd_kernel<NQB, true, SH, false>), dim3(grid), dim3(768), 0, s, a);
  else hipLaunchKernelGGL((lstm_persist_bwd_kernel<NQB, false, SH, false>), dim3(grid), dim3(768), 0, s, a);
  return yt8m::launch_status("lstm_persist_bwd_kernel");
}
template <int NQB>
int launch_bwd(const PersistBwdArgs& a, unsigned grid, hipStream_t s) {
  return a.nimg >= a.T ? launch_bwd_sh<NQB, true>(a, grid, s) : launch_bwd_sh<NQB, false>(a, grid, s);
}
}  // namespace

// K-split workgroup pairs of the f16 backward recurrence (lstm_persist_bwd_kernel<.., P2>): -1 = the environment's choice
// (YT8M_PERSIST_BWD_PAIR, default OFF: 13.7 vs 14.4 us/step stand-alone, but +0.66 ms on the headline step -- profiles/r6_recur_ab.txt block 4), 0 = off, 1 = on where the launch can take it.  Process-wide; for A/B runs and tests.
extern "C" int yt8m_lstm_persist_set_pair(int mode) {
  g_pair_mode.store(mode < 0 ? -1 : (mode ? 1 : 0));
  return YT8M_OK;
}

extern "C" int yt8m_lstm_persist_bwd_supported(int64_t B, int64_t H) {
  return (persist_geometry(B, H, nullptr) && persist_geometry_bwd(B, H, nullptr)) ? 1 : 0;
}

// Rows of the column-sum partials a launch with operand images writes (4 per row group), 0 when the shape cannot take the images
// (they are written by the rotated epilogue: B % 16 == 0 and a multiple of four 16-row tiles per workgroup).
extern "C" int yt8m_lstm_persist_bwd_images_rows(int64_t B, int64_t H) {
  GeometryB geo;
  if (!persist_geometry(B, H, nullptr) || !persist_geometry_bwd(B, H, &geo)) return 0;
  if ((B % 16) != 0 || !bwd_rot(geo.pf, geo.NT16, geo.RB)) return 0;
  return 4 * geo.RB;
}

namespace {
int persist_bwd_impl(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                     float* work, int phase, float* dbias_rows, const int32_t* num_frames, int64_t t0, int64_t T,
                     int64_t B, int64_t H, void* workspace, int64_t workspace_bytes, const yt8m_persist_bwd_images* img,
                     yt8m_stream_t stream, bool bf16 = false, const void* h2_wword = nullptr, void* rowmax = nullptr, void* partmax = nullptr);
// bytes of the scale words an H2 launch of T steps publishes (one word per step, tile, producer workgroup and row quad), 256-aligned
int64_t h2_scale_bytes(int NT16, int64_t H, int64_t T) { return ((T * NT16 * (H / 16) * 16 + 255) / 256) * 256; }
}

extern "C" int yt8m_lstm_persist_bwd(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                                     float* work, int phase, float* dbias_rows, const int32_t* num_frames, int64_t t0, int64_t T,
                                     int64_t B, int64_t H, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  return persist_bwd_impl(gates, Wh, ldw, cs, dout, dz, work, phase, dbias_rows, num_frames, t0, T, B, H, workspace, workspace_bytes,
                          nullptr, stream);
}

// yt8m_lstm_persist_bwd with the recurrent product dz . W_h^T on ONE bf16 plane (dz and W_h rounded to nearest even, fp32 accumulation;
// --compute_dtype=bfloat16).  Shapes / launches that cannot take the bf16 form (H not 512 or 1024, fewer than four 16-row tiles per
// workgroup, no room for one exchange image per step) run the fp32 form: the request is a permission, never an error.
extern "C" int yt8m_lstm_persist_bwd_bf16(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                                          float* work, int phase, float* dbias_rows, const int32_t* num_frames, int64_t t0, int64_t T,
                                          int64_t B, int64_t H, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  return persist_bwd_impl(gates, Wh, ldw, cs, dout, dz, work, phase, dbias_rows, num_frames, t0, T, B, H, workspace, workspace_bytes,
                          nullptr, stream, true);
}

// yt8m_lstm_persist_bwd with the recurrent product dz . W_h^T as THREE f16 products of two-half-plane splits (fp32-grade: 2^-21 relative per
// term, like yt8m_gemm_h2_nt_grouped) instead of the fp32 matrix pipe: 48 MFMAs of 16 cycles per item and wave instead of 128 of 32.
// wh_absmax: device word holding max |W_h| as float bits (yt8m_h2_absmax over the [H, 4H] block).  Results differ from
// yt8m_lstm_persist_bwd by fp32 rounding only.  A permission like the bf16 form: launches that cannot take it (H not 512 / 1024, tiles per
// workgroup not a multiple of four, a workspace without one image per step + the scale words: yt8m_lstm_persist_workspace_bytes_steps
// has the room) run the fp32 form.  YT8M_PERSIST_BWD_H2=0 turns the request off process-wide.
extern "C" int yt8m_lstm_persist_bwd_h2(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                                        float* work, int phase, float* dbias_rows, const int32_t* num_frames, int64_t t0, int64_t T,
                                        int64_t B, int64_t H, const void* wh_absmax, void* workspace, int64_t workspace_bytes,
                                        yt8m_stream_t stream) {
  YT8M_REQUIRE(wh_absmax, YT8M_E_BADARG, "null absmax word");
  static const bool off = getenv("YT8M_PERSIST_BWD_H2") != nullptr && atoi(getenv("YT8M_PERSIST_BWD_H2")) == 0;
  return persist_bwd_impl(gates, Wh, ldw, cs, dout, dz, work, phase, dbias_rows, num_frames, t0, T, B, H, workspace, workspace_bytes,
                          nullptr, stream, false, off ? nullptr : wh_absmax);
}
// yt8m_lstm_persist_bwd / _h2 (wh_absmax NULL: the fp32-pipe form) that also measures what the products after it would measure in passes
// over dz: rowmax[t B + b] = max |dz[t, b, :]| (float bits; [F B] words by absolute frame row, ZEROED by the caller for the launch's
// frames) and / or partmax = max |dz| of the launch (one zeroed word; several launches may share it) -- the operands of
// yt8m_h2_split_rowmax and of yt8m_h2_split / yt8m_gemm_h2_nt_grouped (dscale / dsb).  Needs the rotated epilogue
// (yt8m_lstm_persist_bwd_images_rows(B, H) > 0), else YT8M_E_SHAPE.
extern "C" int yt8m_lstm_persist_bwd_ex(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                                        float* work, int phase, const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H,
                                        const void* wh_absmax, void* rowmax, void* partmax, void* workspace, int64_t workspace_bytes,
                                        yt8m_stream_t stream) {
  static const bool off = getenv("YT8M_PERSIST_BWD_H2") != nullptr && atoi(getenv("YT8M_PERSIST_BWD_H2")) == 0;
  return persist_bwd_impl(gates, Wh, ldw, cs, dout, dz, work, phase, nullptr, num_frames, t0, T, B, H, workspace, workspace_bytes,
                          nullptr, stream, false, off ? nullptr : wh_absmax, rowmax, partmax);
}
// 1 when yt8m_lstm_persist_bwd_h2 takes the f16 form for a T-step launch on a workspace of yt8m_lstm_persist_workspace_bytes_steps(B, H, T)
extern "C" int yt8m_lstm_persist_bwd_on_f16_pipe(int64_t B, int64_t H) {
  GeometryB geo;
  if (!persist_geometry(B, H, nullptr) || !persist_geometry_bwd(B, H, &geo)) return 0;
  return ((H == 512 || H == 1024) && bwd_rot(geo.pf, geo.NT16, geo.RB)) ? 1 : 0;
}

// yt8m_lstm_persist_bwd that also leaves the operand images of dz[t0 .. t0 + T) for the products that follow (include/yt8m_hip.h).
extern "C" int yt8m_lstm_persist_bwd_images(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                                            float* work, int phase, const int32_t* num_frames, int64_t t0, int64_t T, int64_t B,
                                            int64_t H, void* workspace, int64_t workspace_bytes, const yt8m_persist_bwd_images* images,
                                            yt8m_stream_t stream) {
  YT8M_REQUIRE(images, YT8M_E_BADARG, "null image description");
  YT8M_REQUIRE(yt8m_lstm_persist_bwd_images_rows(B, H) > 0, YT8M_E_SHAPE, "shape cannot take the operand images (yt8m_lstm_persist_bwd_images_rows)");
  YT8M_REQUIRE((images->rowscale != nullptr) == (images->trans_scaled != nullptr || images->colpart_scaled != nullptr), YT8M_E_BADARG,
               "rowscale comes with trans_scaled / colpart_scaled");
  YT8M_REQUIRE((((uintptr_t)images->plain | (uintptr_t)images->trans | (uintptr_t)images->trans_scaled) & 15) == 0, YT8M_E_BADARG,
               "images must be 16-byte aligned");
  return persist_bwd_impl(gates, Wh, ldw, cs, dout, dz, work, phase, nullptr, num_frames, t0, T, B, H, workspace, workspace_bytes, images,
                          stream);
}

namespace {
int persist_bwd_impl(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout, float* dz,
                     float* work, int phase, float* dbias_rows, const int32_t* num_frames, int64_t t0, int64_t T,
                     int64_t B, int64_t H, void* workspace, int64_t workspace_bytes, const yt8m_persist_bwd_images* img,
                     yt8m_stream_t stream, bool bf16, const void* h2_wword, void* rowmax, void* partmax) {
  using namespace yt8m;
  YT8M_REQUIRE(t0 >= 0 && T >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(phase == 0 || phase == 1, YT8M_E_BADARG, "phase must be 0 or 1");
  if (T * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(gates && Wh && cs && dz && work && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldw >= 4 * H && (ldw % 4) == 0 && (reinterpret_cast<uintptr_t>(Wh) & 15) == 0, YT8M_E_SHAPE,
               "W_h rows must be 16-byte aligned (ldw % 4 == 0)");
  GeometryB geo;
  YT8M_REQUIRE(persist_geometry(B, H, nullptr) && persist_geometry_bwd(B, H, &geo), YT8M_E_SHAPE,
               "shape not supported by the persistent backward recurrence (see yt8m_lstm_persist_bwd_supported)");
  YT8M_REQUIRE(workspace_bytes >= yt8m_lstm_persist_workspace_bytes(B, H), YT8M_E_SHAPE, "workspace too small");
  YT8M_REQUIRE(T < (1 << 20), YT8M_E_SHAPE, "T too large");
  hipStream_t s = as_stream(stream);
  const int64_t cb = ctl_padded(geo.NT16);
  PersistBwdArgs a;
  a.gates = gates; a.Wh = Wh; a.ldw = ldw; a.cs = cs; a.dout = dout; a.dz = dz; a.work = work; a.nf = num_frames;
  a.dbrows = dbias_rows;
  a.img_plain = img ? static_cast<float*>(img->plain) : nullptr;
  a.img_trans = img ? static_cast<float*>(img->trans) : nullptr;
  a.img_trans_s = img ? static_cast<float*>(img->trans_scaled) : nullptr;
  a.rowscale = img ? img->rowscale : nullptr;
  a.colpart = img ? img->colpart : nullptr;
  a.colpart_s = img ? img->colpart_scaled : nullptr;
  a.ctl = static_cast<unsigned*>(workspace) + CTL_STICKY;
  a.dzx = reinterpret_cast<float*>(static_cast<char*>(workspace) + cb);
  a.t0 = (int)t0; a.T = (int)T; a.B = (int)B; a.H = (int)H; a.phase = phase;
  a.NUB = geo.NUB; a.RB = geo.RB; a.NT16 = geo.NT16; a.per = geo.per; a.pf = geo.pf;
  a.nimg = images_in(workspace_bytes, geo.NT16, 4 * H);
  a.bf = (bf16 && !img) ? 1 : 0;
  a.dbg = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + workspace_bytes - DBG_BYTES);
  a.rowmax = static_cast<unsigned*>(rowmax);
  a.partmax = static_cast<unsigned*>(partmax);
  YT8M_REQUIRE(!(rowmax || partmax) || (!img && images_in(workspace_bytes, geo.NT16, 4 * H) >= T && bwd_rot(geo.pf, geo.NT16, geo.RB)), YT8M_E_SHAPE,
               "row / launch maxima need the rotated backward epilogue on a per-step workspace (yt8m_lstm_persist_bwd_images_rows)");
  // H2: the scale words sit between the last exchange image and the debug tail; taken only when T images still fit in front of them
  a.h2 = 0; a.wword = nullptr; a.sc = nullptr; a.sc_bytes = 0;
  a.px = nullptr; a.px_bytes = 0; a.nonce = 0;
  if (h2_wword && !img && !bf16 && (H == 512 || H == 1024)) {
    const int64_t scb = h2_scale_bytes(geo.NT16, H, T);
    if (scb < (1LL << 31) && images_in(workspace_bytes - scb, geo.NT16, 4 * H) >= T) {
      a.h2 = 1;
      a.wword = static_cast<const unsigned*>(h2_wword);
      a.sc = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + workspace_bytes - DBG_BYTES - scb);
      a.sc_bytes = (unsigned)scb;
      // K-split workgroup pairs (P2): the two workgroups that share a 128-byte line of gates sit on one XCD (the paired block -> unit
      // group map), every workgroup owns <= 8 tiles, and the hand-off slots fit in front of the scale words.  Opt-in: YT8M_PERSIST_BWD_PAIR=1
      static const bool pair_env_off = getenv("YT8M_PERSIST_BWD_PAIR") == nullptr || atoi(getenv("YT8M_PERSIST_BWD_PAIR")) == 0;
      const int pair_mode = g_pair_mode.load();
      const bool pair_off = pair_mode < 0 ? pair_env_off : pair_mode == 0;
      const int64_t pxb = (int64_t)2 * geo.NT16 * geo.NUB * 2048;
      const int n_it_max = (geo.NT16 + geo.RB - 1) / geo.RB;
      if (!pair_off && bwd_rot(geo.pf, geo.NT16, geo.RB) && geo.per > 0 && (geo.NUB % (2 * geo.per)) == 0 && n_it_max <= 8 &&
          images_in(workspace_bytes - scb - pxb, geo.NT16, 4 * H) >= T) {
        static std::atomic<unsigned> g_nonce{1};
        a.px = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + workspace_bytes - DBG_BYTES - scb - pxb);
        a.px_bytes = (unsigned)pxb;
        a.nonce = g_nonce.fetch_add(1) & 0x7FFFFu;           // (nonce * 8191 + step + 1 < 2^32; reuse after 2^19 launches)
        if (a.nonce == 0) a.nonce = g_nonce.fetch_add(1) & 0x7FFFFu;
      }
    }
  }
  const unsigned grid = (unsigned)(geo.NUB * geo.RB);
  int dev = 0;
  device_cus(&dev);
  a.stats = stats_ptr(dev);
  ++g_stat_launches[dev];
  g_stat_wgs[dev] += grid;
  ProfScope prof(F_LSTM_BWD, s, 2.0 * (double)T * (double)B * (double)H * 4.0 * (double)H);
  std::lock_guard<std::mutex> lk(g_gate.mu);
  const int total_cus = device_cus(nullptr);
  int grc = g_gate.admit(dev, (int)grid, std::max(0, total_cus - g_reserved_cus), s);
  if (grc != YT8M_OK) return grc;
  YT8M_HIP_CHECK(hipMemsetAsync(a.ctl, 0, (size_t)ctl_bytes(geo.NT16), s));
  int rc;
  switch (geo.NQB) {
    case 4: rc = launch_bwd<4>(a, grid, s); break;
    case 8: rc = launch_bwd<8>(a, grid, s); break;
    case 16: rc = launch_bwd<16>(a, grid, s); break;
    case 24: rc = launch_bwd<24>(a, grid, s); break;
    default: rc = launch_bwd<32>(a, grid, s); break;
  }
  if (rc != YT8M_OK) return rc;
  return g_gate.done(dev, (int)grid, s);
}
}  // namespace


// =====================================================================================================================
// GRUCell on the persistent protocol (gru_persist.inl).  One launch runs T steps = 2T half-steps of a layer; the workspace holds the
// control block and ONE exchange image per half-step (forward 2T + 1 images of [NT16 16, H]; backward T blocks of [NT16 16, 3H]).
extern "C" int yt8m_gru_persist_supported(int64_t B, int64_t H) {
  static const bool off = getenv("YT8M_GRU_PERSIST") != nullptr && atoi(getenv("YT8M_GRU_PERSIST")) == 0;
  Geometry geo;
  return (!off && persist_geometry(B, H, &geo) && geo.NQ >= 2) ? 1 : 0;
}

extern "C" int64_t yt8m_gru_persist_workspace_bytes(int64_t B, int64_t H, int64_t T) {
  Geometry geo;
  if (!persist_geometry(B, H, &geo)) return 0;
  return ctl_padded(geo.NT16) + (3 * std::max<int64_t>(T, 1) + 3) * geo.NT16 * 16 * H * 4 + DBG_BYTES;
}

namespace {
template <int NQ>
int launch_gru_fwd(const GruFwdArgs& a, unsigned grid, hipStream_t s) {
  if (a.pf >= 2) hipLaunchKernelGGL((gru_persist_fwd_kernel<NQ, 2>), dim3(grid), dim3(768), 0, s, a);
  else if (a.pf == 1) hipLaunchKernelGGL((gru_persist_fwd_kernel<NQ, 1>), dim3(grid), dim3(768), 0, s, a);
  else hipLaunchKernelGGL((gru_persist_fwd_kernel<NQ, 0>), dim3(grid), dim3(768), 0, s, a);
  return yt8m::launch_status("gru_persist_fwd_kernel");
}
}  // namespace

// Steps t0 .. t0 + T - 1 of one GRU layer.  zg [F,B,2H] / zc [F,B,H]: the hoisted input projections (+ biases) on entry, the
// activations r | u / c on exit (what yt8m_gru_layer_fwd leaves there); hs [F+1,B,H] with hs[t0] given; rh [F,B,H]; out optional.
// Same results as yt8m_gru_layer_fwd up to the K summation order of the recurrent products and the v_exp / v_rcp gate functions
// (<= ~1.5e-7 absolute per activation).
extern "C" int yt8m_gru_persist_fwd(float* zg, float* zc, const float* Wg_h, int64_t ldg, const float* Wc_h, int64_t ldc, float* hs,
                                    float* rh, float* out, const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H,
                                    void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  using namespace yt8m;
  YT8M_REQUIRE(t0 >= 0 && T >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (T * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(zg && zc && Wg_h && Wc_h && hs && rh && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldg >= 2 * H && ldc >= H, YT8M_E_SHAPE, "leading dimension too small");
  Geometry geo;
  YT8M_REQUIRE(yt8m_gru_persist_supported(B, H) && persist_geometry(B, H, &geo), YT8M_E_SHAPE,
               "shape not supported by the persistent GRU recurrence (see yt8m_gru_persist_supported)");
  YT8M_REQUIRE(workspace_bytes >= yt8m_gru_persist_workspace_bytes(B, H, T), YT8M_E_SHAPE, "workspace too small");
  YT8M_REQUIRE(T < (1 << 19), YT8M_E_SHAPE, "T too large");
  hipStream_t s = as_stream(stream);
  GruFwdArgs a;
  a.zg = zg; a.zc = zc; a.Wg = Wg_h; a.Wc = Wc_h; a.ldg = ldg; a.ldc = ldc; a.hs = hs; a.rh = rh; a.out = out; a.nf = num_frames;
  a.ctl = static_cast<unsigned*>(workspace) + CTL_STICKY;
  a.hx = reinterpret_cast<float*>(static_cast<char*>(workspace) + ctl_padded(geo.NT16));
  a.t0 = (int)t0; a.T = (int)T; a.B = (int)B; a.H = (int)H;
  a.NU = geo.NU; a.RB = geo.RB; a.NT16 = geo.NT16; a.per = geo.per; a.pf = geo.pf;
  const unsigned grid = (unsigned)(geo.NU * geo.RB);
  int dev = 0;
  device_cus(&dev);
  a.stats = stats_ptr(dev);
  ++g_stat_launches[dev];
  g_stat_wgs[dev] += grid;
  ProfScope prof(F_LSTM, s, 2.0 * (double)T * (double)B * (double)H * 3.0 * (double)H);
  std::lock_guard<std::mutex> lk(g_gate.mu);
  const int total_cus = device_cus(nullptr);
  int grc = g_gate.admit(dev, (int)grid, std::max(0, total_cus - g_reserved_cus), s);
  if (grc != YT8M_OK) return grc;
  YT8M_HIP_CHECK(hipMemsetAsync(a.ctl, 0, (size_t)ctl_bytes(geo.NT16), s));
  hipLaunchKernelGGL(hx_pack_kernel, dim3(256), dim3(256), 0, s, hs + t0 * B * H, a.hx, (int)B, (int)H, geo.NT16);
  int rc = launch_status("hx_pack_kernel");
  if (rc != YT8M_OK) return rc;
  switch (geo.NQ) {
    case 2: rc = launch_gru_fwd<2>(a, grid, s); break;
    case 4: rc = launch_gru_fwd<4>(a, grid, s); break;
    case 6: rc = launch_gru_fwd<6>(a, grid, s); break;
    default: rc = launch_gru_fwd<8>(a, grid, s); break;
  }
  if (rc != YT8M_OK) return rc;
  return g_gate.done(dev, (int)grid, s);
}

// Backward of the same steps, last to first.  work [B,H]: dL/dh flowing into step t0 + T - 1 from later steps on entry (zeros, or
// the final state's gradient), dL/dh_{t0 - 1} on exit -- launches over consecutive time ranges chain through it.  Writes dzg [F,B,2H]
// and dzc [F,B,H] of the range (the operands of the hoisted weight-gradient / dx products), as yt8m_gru_layer_bwd does.
extern "C" int yt8m_gru_persist_bwd(const float* zg, const float* zc, const float* Wg_h, int64_t ldg, const float* Wc_h, int64_t ldc,
                                    const float* hs, const float* dout, float* dzg, float* dzc, float* work, const int32_t* num_frames,
                                    int64_t t0, int64_t T, int64_t B, int64_t H, void* workspace, int64_t workspace_bytes,
                                    yt8m_stream_t stream) {
  using namespace yt8m;
  YT8M_REQUIRE(t0 >= 0 && T >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (T * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(zg && zc && Wg_h && Wc_h && hs && dzg && dzc && work && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldg >= 2 * H && ldc >= H, YT8M_E_SHAPE, "leading dimension too small");
  Geometry geo;
  YT8M_REQUIRE(yt8m_gru_persist_supported(B, H) && persist_geometry(B, H, &geo), YT8M_E_SHAPE,
               "shape not supported by the persistent GRU recurrence (see yt8m_gru_persist_supported)");
  YT8M_REQUIRE(workspace_bytes >= yt8m_gru_persist_workspace_bytes(B, H, T), YT8M_E_SHAPE, "workspace too small");
  YT8M_REQUIRE(T < (1 << 19), YT8M_E_SHAPE, "T too large");
  hipStream_t s = as_stream(stream);
  GruBwdArgs a;
  a.zg = zg; a.zc = zc; a.Wg = Wg_h; a.Wc = Wc_h; a.ldg = ldg; a.ldc = ldc; a.hs = hs; a.dout = dout; a.dzg = dzg; a.dzc = dzc;
  a.work = work; a.nf = num_frames;
  a.ctl = static_cast<unsigned*>(workspace) + CTL_STICKY;
  a.hx = reinterpret_cast<float*>(static_cast<char*>(workspace) + ctl_padded(geo.NT16));
  a.t0 = (int)t0; a.T = (int)T; a.B = (int)B; a.H = (int)H;
  a.NU = geo.NU; a.RB = geo.RB; a.NT16 = geo.NT16; a.per = geo.per; a.pf = geo.pf;
  // 16 units per workgroup (H / 16 unit groups, every result column of the MFMA tile used) on HALF the chip by default (YT8M_GRU_BWD_CUS,
  // the LSTM backward kernel's policy: four tiles = four chains per workgroup at B = 128, the other 128 CUs stay free for the hoisted
  // products): 29.0 us/step; the whole chip (two chains per workgroup) 31.5; YT8M_GRU_BWD_U=8, the forward geometry (half of every B
  // fragment zero) on 256 workgroups: 30.3 -- profiles/r6_gru_persist.txt.
  static const int u_env = getenv("YT8M_GRU_BWD_U") ? atoi(getenv("YT8M_GRU_BWD_U")) : 16;
  static const int cus_env = getenv("YT8M_GRU_BWD_CUS") ? atoi(getenv("YT8M_GRU_BWD_CUS")) : 128;
  const bool u16 = u_env == 16;
  if (u16) {
    const int cus = device_cus(nullptr);
    const int budget = (cus_env > 0 && cus_env < cus) ? cus_env : cus;
    a.NU = (int)(H / 16);
    int RB = std::max(1, budget / a.NU);
    if (RB > geo.NT16 / 2) RB = std::max(1, geo.NT16 / 2);
    a.RB = RB;
    a.per = (RB <= 8 && (8 % RB) == 0 && (a.NU % (8 / RB)) == 0) ? 8 / RB : 0;
    YT8M_REQUIRE((geo.NT16 + RB - 1) / RB <= MAX_LOCAL_TILES, YT8M_E_SHAPE, "batch too large for the persistent GRU recurrence");
  }
  const unsigned grid = (unsigned)(a.NU * a.RB);
  int dev = 0;
  device_cus(&dev);
  a.stats = stats_ptr(dev);
  ++g_stat_launches[dev];
  g_stat_wgs[dev] += grid;
  ProfScope prof(F_LSTM_BWD, s, 2.0 * (double)T * (double)B * (double)H * 3.0 * (double)H);
  std::lock_guard<std::mutex> lk(g_gate.mu);
  const int total_cus = device_cus(nullptr);
  int grc = g_gate.admit(dev, (int)grid, std::max(0, total_cus - g_reserved_cus), s);
  if (grc != YT8M_OK) return grc;
  YT8M_HIP_CHECK(hipMemsetAsync(a.ctl, 0, (size_t)ctl_bytes(geo.NT16), s));
  if (u16) {
    switch (geo.NQ) {
      case 2: hipLaunchKernelGGL((gru_persist_bwd_kernel<2, 16>), dim3(grid), dim3(768), 0, s, a); break;
      case 4: hipLaunchKernelGGL((gru_persist_bwd_kernel<4, 16>), dim3(grid), dim3(768), 0, s, a); break;
      case 6: hipLaunchKernelGGL((gru_persist_bwd_kernel<6, 16>), dim3(grid), dim3(768), 0, s, a); break;
      default: hipLaunchKernelGGL((gru_persist_bwd_kernel<8, 16>), dim3(grid), dim3(768), 0, s, a); break;
    }
  } else {
    switch (geo.NQ) {
      case 2: hipLaunchKernelGGL((gru_persist_bwd_kernel<2, 8>), dim3(grid), dim3(768), 0, s, a); break;
      case 4: hipLaunchKernelGGL((gru_persist_bwd_kernel<4, 8>), dim3(grid), dim3(768), 0, s, a); break;
      case 6: hipLaunchKernelGGL((gru_persist_bwd_kernel<6, 8>), dim3(grid), dim3(768), 0, s, a); break;
      default: hipLaunchKernelGGL((gru_persist_bwd_kernel<8, 8>), dim3(grid), dim3(768), 0, s, a); break;
    }
  }
  int rc = launch_status("gru_persist_bwd_kernel");
  if (rc != YT8M_OK) return rc;
  return g_gate.done(dev, (int)grid, s);
}
