// gru_persist.inl -- tf.contrib.rnn.GRUCell under tf.nn.dynamic_rnn as ONE launch per (layer, time range) and direction, on the
// exchange protocol of the persistent LSTM recurrences (W/all_frame_models/gru_pooling_model.py:34-47, gru_with_pooling_model.py:34-38;
// VERDICT r5 #3).  Included by lstm_persist.hip inside its anonymous namespace: the control block, the sharded arrival counters,
// wait_tile / lds_wait_ge, the residency gate and the block -> (unit group, row group) map are the LSTM kernels' own.
//
// A GRU step is TWO dependent products -- [r|u] = sigmoid(zg + h . Wg_h), then c = tanh(zc + (r*h) . Wc_h) -- so a time step is two
// HALF-STEPS of the LSTM kernel's machine: half-step m reads exchange image m and its epilogue publishes image m + 1,
//   forward   m = 2 s     : A = h_{t-1}  (K = H), columns r|u of the workgroup's 8 units -> publishes r*h_{t-1}
//             m = 2 s + 1 : A = r*h      (K = H), columns c                             -> publishes h_t
//   backward  m = 0       : (no product) the epilogue waves form dzc of the launch's last step from dh and publish it
//             m = 2 s + 1 : A = dzc_t    (K = H),  B = Wc_h^T rows of the 8 units -> d(r*h); publishes [dzr|dzu]_t
//             m = 2 s + 2 : A = dzg_t    (K = 2H), B = Wg_h^T rows                -> dh_{t-1}; publishes dzc_{t-1}
// one exchange image per half-step (an address is written once and read after its tile's count is complete: plain L2-shared loads,
// see lstm_persist_fwd_kernel).  The products run on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 out: bit-compatible in kind with the
// per-step kernels of cells.hip / lstm_fused.hip; only the K summation order and the v_exp / v_rcp gate functions differ).
// Workgroup = 8 matrix waves (K split 8 ways, weights resident: half registers, half LDS) + 4 epilogue waves, as the LSTM kernel.

struct GruFwdArgs {
  float* zg;            // [F,B,2H] hoisted x . Wg_x + bg on entry; r | u activations on exit
  float* zc;            // [F,B,H]  hoisted x . Wc_x + bc on entry; c = tanh(..) on exit
  const float* Wg;      // [H, ldg]  recurrent rows of gates/weights (columns r: 0 .. H-1, u: H .. 2H-1)
  const float* Wc;      // [H, ldc]  recurrent rows of candidate/weights
  long long ldg, ldc;
  float* hs;            // [F+1,B,H] hs[t] = state before step t
  float* rh;            // [F,B,H]   r * h_{t-1} (operand of the candidate's weight gradient)
  float* out;           // [F,B,H] or null
  const int32_t* nf;    // [B] or null
  float* hx;            // exchange images [2T+1][NT16][H/16][256]
  unsigned* ctl;
  unsigned* stats;
  int t0, T, B, H;
  int NU, RB, NT16, per, pf;
};

template <int NQ, int PD>
__global__ __launch_bounds__(768) void gru_persist_fwd_kernel(GruFwdArgs a) {
  constexpr int HQ = NQ / 2;
  constexpr int NL = NQ - HQ;
  constexpr int HQA = HQ > 0 ? HQ : 1;
  __shared__ __attribute__((aligned(16))) float red[NSLOT][8][4][64];      // [slot][wave][acc reg][lane]: 32 KB
  __shared__ __attribute__((aligned(16))) float4 Wl[8][NL][2][64];         // LDS half of the slice: [tile 0 = r|u, tile 1 = c]
  __shared__ unsigned lds_cnt[NSLOT], lds_free[NSLOT];
  __shared__ unsigned lds_seen[MAX_LOCAL_TILES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ug, g;
  {
    const int b = blockIdx.x;
    if (a.per > 0 && (a.NU % (4 * a.per)) == 0) {
      const int x = b & 7, sl = b >> 3;
      g = x / a.per;
      ug = ((sl >> 2) * a.per + (x % a.per)) * 4 + (sl & 3);
    } else if (a.per > 0) { const int x = b & 7; g = x / a.per; ug = (b >> 3) * a.per + (x % a.per); }
    else { g = b / a.NU; ug = b % a.NU; }
  }
  const int H = a.H, B = a.B, NT16 = a.NT16, RB = a.RB;
  const int n_it = (NT16 - g + RB - 1) / RB;
  const int HT = 2 * a.T;                                 // half-steps of this launch
  const int total = n_it * HT;
  const unsigned img_bytes = (unsigned)NT16 * (unsigned)H * 16u * 4u;
  const long long img_f = (long long)NT16 * H * 16;
  auto image = [&](int m) -> __amdgpu_buffer_rsrc_t { return make_rsrc(a.hx + m * img_f, img_bytes); };
  const int QH = H >> 4;
  const unsigned arrivals = (unsigned)a.NU;
  // B fragment of v_mfma_f32_16x16x4_f32: lane (n = lane & 15, kq = lane >> 4) supplies B[k = 16 q + 4 kq + e][n], e = 0..3 as a float4.
  // tile 0: column n <-> (unit n / 2, gate n % 2: r, u) -- the two gates of a unit are neighbouring result lanes;
  // tile 1: column n < 8 <-> candidate of unit n, columns 8-15 are zero.
  auto w_frag = [&](int qg, int ct) -> float4 {
    const int i16 = lane & 15, kq = lane >> 4;
    const long long k = (long long)(w * NQ + qg) * 16 + kq * 4;
    if (ct == 0) {
      const float* p = a.Wg + k * a.ldg + (long long)(i16 & 1) * H + ug * 8 + (i16 >> 1);
      return make_float4(p[0], p[a.ldg], p[2 * a.ldg], p[3 * a.ldg]);
    }
    if (i16 >= 8) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = a.Wc + k * a.ldc + ug * 8 + i16;
    return make_float4(p[0], p[a.ldc], p[2 * a.ldc], p[3 * a.ldc]);
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
    const unsigned lane_off = (unsigned)(i16 * 16 + kq * 4) * 4u + (unsigned)(w * NQ) * 1024u;
    auto load_item = [&](float4 (&A)[NQ], int m, int T) {
      const __amdgpu_buffer_rsrc_t hxr = image(m);
      const unsigned base = (unsigned)(T * QH) * 1024u + lane_off;
#pragma unroll
      for (int qg = 0; qg < NQ; ++qg) A[qg] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(hxr, (int)(base + (unsigned)qg * 1024u), 0, 0));
    };
    float4 A0[NQ], A1[NQ], A2[NQ];
    if (PD >= 1) load_item(A0, 0, g);
    if (PD >= 2 && total > 1) load_item(A1, n_it > 1 ? 0 : 1, n_it > 1 ? g + RB : g);   // (n_it >= 8 whenever PD == 2)
    int m_cur = 0, it_cur = 0;                             // item k = (half-step m_cur, local tile it_cur)
    auto item = [&](float4 (&A)[NQ], float4 (&Areq)[NQ], int k) {
      const int m = m_cur, T = g + it_cur * RB;
      const bool cand = (m & 1) != 0;
      int mr = m, Tr = T, itr = it_cur;
      unsigned pv = 0;
      if (PD >= 1) {
        itr = it_cur + PD;
        while (itr >= n_it) { itr -= n_it; ++mr; }
        const bool have = k + PD < total;
        Tr = have ? g + itr * RB : T;
        itr = have ? itr : it_cur;
        mr = have ? mr : m;
        if (w == 0 && lane < NSH)
          pv = __hip_atomic_load(a.ctl + CTL_HDR + (Tr * NSH + lane) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (w == 0) {
          wait_tile(a.ctl, T, (unsigned)m * arrivals, lane);
          if (lane == 0) __hip_atomic_store(&lds_seen[it_cur], (unsigned)m * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          lds_wait_ge(&lds_seen[it_cur], (unsigned)m * arrivals, a.ctl);
        }
        load_item(A, m, T);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qg = 0; qg < NQ; ++qg) {
        if (PD >= 1 && qg == NQ / 2) {                     // request point: the image of item k + PD must be complete now
          if (w == 0) {
            const unsigned tot = shard_sum(pv);
            if (tot < (unsigned)mr * arrivals) wait_tile(a.ctl, Tr, (unsigned)mr * arrivals, lane);
            if (lane == 0) __hip_atomic_store(&lds_seen[itr], (unsigned)mr * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            lds_wait_ge(&lds_seen[itr], (unsigned)mr * arrivals, a.ctl);
          }
          load_item(Areq, mr, Tr);
        }
        const float4 av = A[qg];
        float4 bv;
        if (qg < HQ) {
          const float4 b0 = Wr[qg < HQ ? qg : 0][0], b1 = Wr[qg < HQ ? qg : 0][1];
          bv = make_float4(cand ? b1.x : b0.x, cand ? b1.y : b0.y, cand ? b1.z : b0.z, cand ? b1.w : b0.w);
        } else {
          bv = Wl[w][qg - HQ][cand ? 1 : 0][lane];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
      }
      const int slot = k & (NSLOT - 1);
      if (k >= NSLOT) lds_wait_ge(&lds_free[slot], (unsigned)(k / NSLOT), a.ctl);
      float* rw = &red[slot][w][0][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) rw[r * 64] = acc[r];
      if (lane == 0) __hip_atomic_fetch_add(&lds_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (++it_cur == n_it) { it_cur = 0; ++m_cur; }
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
  // wave ew owns the local tiles ew, ew + 4, .. for ALL half-steps: what it reloads (u, h_{t-1}) are its own earlier stores.
  const int ew = w - 8;
  const int eunit = lane & 7;
  __builtin_amdgcn_s_setprio(YT8M_EPI_PRIO);
  for (int m = 0; m < HT; ++m) {
    const int s = m >> 1;
    const bool cand = (m & 1) != 0;
    const int t = a.t0 + s;
    for (int it = ew; it < n_it; it += NEPI) {
      const int k = m * n_it + it;
      const int T = g + it * RB;
      float z0[2], z1[2], hp[2];
      bool live[2], evalid[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int brow = T * 16 + 8 * j + (lane >> 3);
        evalid[j] = brow < B;
        const int br = evalid[j] ? brow : B - 1;
        const long long idx = ((long long)t * B + br) * H + ug * 8 + eunit;
        const float* zr = a.zg + ((long long)t * B + br) * 2 * H + ug * 8 + eunit;
        hp[j] = a.hs[idx];
        z1[j] = zr[H];                                     // gate half-step: u pre-activation; candidate half-step: u itself
        z0[j] = cand ? a.zc[idx] : zr[0];
        live[j] = a.nf ? (t < a.nf[br]) : true;
      }
      const int slot = k & (NSLOT - 1);
      lds_wait_ge(&lds_cnt[slot], 8u * (unsigned)(k / NSLOT + 1), a.ctl);
      // C layout of the 16x16 tile: column = lane & 15, row = 4 (lane >> 4) + reg.
      float s0[2], s1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int erow = 8 * j + (lane >> 3);
        const int r = erow & 3, lb = (erow >> 2) * 16;
        s0[j] = 0.f; s1[j] = 0.f;
        if (!cand) {
#pragma unroll
          for (int wv = 0; wv < 8; ++wv) {
            const float2 p = *reinterpret_cast<const float2*>(&red[slot][wv][r][lb + 2 * eunit]);
            s0[j] += p.x; s1[j] += p.y;
          }
        } else {
#pragma unroll
          for (int wv = 0; wv < 8; ++wv) s0[j] += red[slot][wv][r][lb + eunit];
        }
      }
      if (lane == 0) __hip_atomic_fetch_add(&lds_free[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      float pub[2], va[2], vb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!cand) {
          va[j] = fast_sigmoid(z0[j] + s0[j]);             // r
          vb[j] = fast_sigmoid(z1[j] + s1[j]);             // u
          pub[j] = va[j] * hp[j];                          // r * h_{t-1}
        } else {
          va[j] = fast_tanh(z0[j] + s0[j]);                // c
          const float u = z1[j];
          const float h1 = u * hp[j] + (1.0f - u) * va[j];
          vb[j] = live[j] ? h1 : hp[j];                    // dynamic_rnn copy-through
          pub[j] = vb[j];
        }
        if (!evalid[j]) pub[j] = 0.f;
      }
      if (m + 1 < HT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float h1 = row_shl<1>(pub[j]), h2 = row_shl<2>(pub[j]), h3 = row_shl<3>(pub[j]);
          if ((eunit & 3) == 0) {
            u32x4 v;
            v.x = __float_as_uint(pub[j]); v.y = __float_as_uint(h1); v.z = __float_as_uint(h2); v.w = __float_as_uint(h3);
            const int erow = 8 * j + (lane >> 3);
            const unsigned off = ((unsigned)(T * QH + (ug >> 1)) * 256u + (unsigned)(erow * 16 + (ug & 1) * 8 + eunit)) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(v, image(m + 1), (int)off, 0, YT8M_AUX_ST);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_fetch_add(a.ctl + CTL_HDR + (T * NSH + (blockIdx.x & (NSH - 1))) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (evalid[j]) {
          const int brow = T * 16 + 8 * j + (lane >> 3);
          const long long idx = ((long long)t * B + brow) * H + ug * 8 + eunit;
          if (!cand) {
            float* zr = a.zg + ((long long)t * B + brow) * 2 * H + ug * 8 + eunit;
            zr[0] = va[j]; zr[H] = vb[j];
            a.rh[idx] = va[j] * hp[j];
          } else {
            a.zc[idx] = va[j];
            a.hs[idx + (long long)B * H] = vb[j];
            if (a.out) a.out[idx] = live[j] ? vb[j] : 0.f;
          }
        }
      }
    }
  }
  if (ew == 0 && lane == 0) { check_placement(a.ctl, a.stats); propagate_error(a.ctl); }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward.  Per step t (descending), with dh = dL/dh_t accumulated so far (+ dout_t for live rows):
//   dzc = dh (1-u)(1-c^2)      dzu = dh (h_{t-1} - c) u (1-u)       dh_{t-1} <- dh u                (gru_bwd1_kernel)
//   drh = dzc . Wc_h^T         dzr = drh h_{t-1} r (1-r)            dh_{t-1} += drh r               (gru_bwd2_kernel)
//   dh_{t-1} += [dzr|dzu] . Wg_h^T
// rows with t >= num_frames pass dh through and leave zeros in dz.  Exchange images: even m (dzc, H wide) at float offset
// (m / 2) 3H S, odd m (dzg, 2H wide) behind it, S = NT16 * 16.
struct GruBwdArgs {
  const float* zg;      // [F,B,2H] r | u
  const float* zc;      // [F,B,H]  c
  const float* Wg;      // [H, ldg]
  const float* Wc;      // [H, ldc]
  long long ldg, ldc;
  const float* hs;      // [F+1,B,H]
  const float* dout;    // [F,B,H] or null
  float* dzg;           // [F,B,2H]
  float* dzc;           // [F,B,H]
  float* work;          // [B,H] dL/dh carried between launches (in: after step t0 + T; out: after step t0)
  const int32_t* nf;
  float* hx;
  unsigned* ctl;
  unsigned* stats;
  int t0, T, B, H;
  int NU, RB, NT16, per, pf;
};

// NQ = H / 128 q-groups of 16 k per wave in the candidate half-step (K = H), 2 NQ in the gate half-step (K = 2H).
// U = hidden units per workgroup: 16 fills the 16 result columns of the MFMA tile (the reduction is 3H long and the output H wide, as
// in the LSTM backward kernel: H / 16 unit groups, fewer and fuller workgroups per row group); 8 is the forward kernel's geometry (half
// of every B fragment is zero) and remains for A/B runs.
template <int NQ, int U>
__global__ __launch_bounds__(768) void gru_persist_bwd_kernel(GruBwdArgs a) {
  constexpr int LU = U == 16 ? 4 : 3;
  constexpr int NP = U / 4;                              // (row, unit) pairs per epilogue lane: rows (64 / U) j + lane / U
  constexpr int NQG = 2 * NQ;
  constexpr int NQT = 3 * NQ;                            // q-groups of weights per wave: [candidate NQ | gates 2 NQ]
  constexpr int HQ = NQT / 2;                            // ... of them in registers
  constexpr int NL = NQT - HQ;
  __shared__ __attribute__((aligned(16))) float red[NSLOT][8][4][64];
  __shared__ __attribute__((aligned(16))) float4 Wl[8][NL][64];
  __shared__ unsigned lds_cnt[NSLOT], lds_free[NSLOT];
  __shared__ unsigned lds_seen[MAX_LOCAL_TILES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ug, g;
  {
    const int b = blockIdx.x;
    if (a.per > 0 && (a.NU % (4 * a.per)) == 0) {
      const int x = b & 7, sl = b >> 3;
      g = x / a.per;
      ug = ((sl >> 2) * a.per + (x % a.per)) * 4 + (sl & 3);
    } else if (a.per > 0) { const int x = b & 7; g = x / a.per; ug = (b >> 3) * a.per + (x % a.per); }
    else { g = b / a.NU; ug = b % a.NU; }
  }
  const int H = a.H, B = a.B, NT16 = a.NT16, RB = a.RB;
  const int n_it = (NT16 - g + RB - 1) / RB;
  const int HT = 2 * a.T;                                 // product half-steps m = 1 .. HT (m = 0: the epilogue-only start)
  const int total = n_it * HT;
  const long long S16 = (long long)NT16 * 16;
  // image m: even -> dzc (width H), odd -> dzg (width 2H)
  auto image = [&](int m) -> __amdgpu_buffer_rsrc_t {
    const long long off = (long long)(m >> 1) * 3 * H * S16 + ((m & 1) ? (long long)H * S16 : 0);
    return make_rsrc(a.hx + off, (unsigned)(S16 * ((m & 1) ? 2 * H : H) * 4));
  };
  const unsigned arrivals = (unsigned)a.NU;
  // weights: q-group q of this wave, q < NQ: candidate (k = (w NQ + q) 16 + ..; B[k][n] = Wc[ug U + n][k]), else gates
  // (k over 2H = (w NQG + q - NQ) 16 + ..; B[k][n] = Wg[ug U + n][k]); columns n >= U are zero.
  auto w_frag = [&](int q) -> float4 {
    const int i16 = lane & 15, kq = lane >> 4;
    const int n = i16 < U ? i16 : 0;                        // (branch-free: lanes of the zero columns load unit 0 and mask it)
    const float msk = i16 < U ? 1.0f : 0.0f;
    const float* p = q < NQ ? a.Wc + (long long)(ug * U + n) * a.ldc + (long long)(w * NQ + q) * 16 + kq * 4
                            : a.Wg + (long long)(ug * U + n) * a.ldg + (long long)(w * NQG + (q - NQ)) * 16 + kq * 4;
    return make_float4(p[0] * msk, p[1] * msk, p[2] * msk, p[3] * msk);
  };
  note_placement(a.ctl);
  if (tid < NSLOT) { lds_cnt[tid] = 0; lds_free[tid] = 0; }
  for (int i = tid; i < MAX_LOCAL_TILES; i += 768) lds_seen[i] = 0;
  if (w < 8) {
#pragma unroll
    for (int qq = 0; qq < NL; ++qq) Wl[w][qq][lane] = w_frag(HQ + qq);
  }
  __syncthreads();

  if (w < 8) {
    // =============================== matrix waves ===============================
    const int i16 = lane & 15, kq = lane >> 4;
    float4 Wr[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) Wr[q] = w_frag(q);
    // ONE fragment buffer (16 float4 at H = 1024: two would not fit beside the resident weights at 12 waves per CU), refilled in
    // place: item k + 1's q-th fragment is requested into slot q right after item k's product has consumed it.  The image of item
    // k + 1 must be complete before the first refill: polled at the start of item k, waited for at q = RQ (the slots below RQ are
    // refilled there in one go).
    constexpr int RQ = NQ / 2 > 0 ? NQ / 2 : 1;
    float4 A[NQG];
    auto frag_addr = [&](int m, int T, bool gates_img) -> unsigned {
      const int QW = gates_img ? (H >> 3) : (H >> 4);
      return (unsigned)(T * QW + w * (gates_img ? NQG : NQ)) * 1024u + (unsigned)(i16 * 16 + kq * 4) * 4u;
    };
    int m_cur = 1, it_cur = 0;
    {                                                       // item 0 fetches its own operands
      if (w == 0) {
        wait_tile(a.ctl, g, arrivals, lane);
        if (lane == 0) __hip_atomic_store(&lds_seen[0], arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        lds_wait_ge(&lds_seen[0], arrivals, a.ctl);
      }
      const __amdgpu_buffer_rsrc_t hxr = image(0);
      const unsigned base = frag_addr(1, g, false);
#pragma unroll
      for (int q = 0; q < NQ; ++q) A[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(hxr, (int)(base + (unsigned)q * 1024u), 0, 0));
    }
    for (int k = 0; k < total; ++k) {
      const int m = m_cur;
      const bool gates = ((m - 1) & 1) != 0;
      int mr = m, itr = it_cur + 1;
      if (itr >= n_it) { itr = 0; ++mr; }
      const bool have = k + 1 < total;
      const int Tr = g + itr * RB;
      const bool gates_r = ((mr - 1) & 1) != 0;
      const int nq = gates ? NQG : NQ, nq_r = have ? (gates_r ? NQG : NQ) : 0;
      unsigned pv = 0;
      if (have && w == 0 && lane < NSH)
        pv = __hip_atomic_load(a.ctl + CTL_HDR + (Tr * NSH + lane) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const __amdgpu_buffer_rsrc_t hxr = image(have ? mr - 1 : 0);
      const unsigned rbase = frag_addr(mr, Tr, gates_r);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQG; ++q) {
        if (q == RQ && have) {
          if (w == 0) {
            const unsigned tot = shard_sum(pv);
            if (tot < (unsigned)mr * arrivals) wait_tile(a.ctl, Tr, (unsigned)mr * arrivals, lane);
            if (lane == 0) __hip_atomic_store(&lds_seen[itr], (unsigned)mr * arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            lds_wait_ge(&lds_seen[itr], (unsigned)mr * arrivals, a.ctl);
          }
        }
        if (q < nq) {
          const float4 av = A[q];
          // weights of q-group q: candidate slots [0, NQ), gate slots [NQ, 3 NQ) of the wave's table; slots below HQ are registers
          float4 bv;
          if (gates) {
            bv = (NQ + q) < HQ ? Wr[(NQ + q) < HQ ? (NQ + q) : 0] : Wl[w][(NQ + q) < HQ ? 0 : (NQ + q) - HQ][lane];
          } else {
            bv = q < HQ ? Wr[q < HQ ? q : 0] : Wl[w][q < HQ ? 0 : q - HQ][lane];
          }
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
        }
        if (q == RQ) {
#pragma unroll
          for (int q2 = 0; q2 < RQ; ++q2)
            if (q2 < nq_r) A[q2] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(hxr, (int)(rbase + (unsigned)q2 * 1024u), 0, 0));
        }
        if (q >= RQ && q < nq_r) A[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(hxr, (int)(rbase + (unsigned)q * 1024u), 0, 0));
      }
      const int slot = k & (NSLOT - 1);
      if (k >= NSLOT) lds_wait_ge(&lds_free[slot], (unsigned)(k / NSLOT), a.ctl);
      float* rw = &red[slot][w][0][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) rw[r * 64] = acc[r];
      if (lane == 0) __hip_atomic_fetch_add(&lds_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (++it_cur == n_it) { it_cur = 0; ++m_cur; }
    }
    return;
  }

  // =============================== epilogue waves ===============================
  // wave ew owns the local tiles ew, ew + 4, .. for all half-steps; the running dL/dh of its (row, unit) pairs and what the next
  // half-step of the same pair needs travel through `work` / dzg / dzc (its own stores, program order).
  const int ew = w - 8;
  const int eunit = lane & (U - 1);
  const int rsub = lane >> LU;
  const int ucol = ug * U + eunit;                         // this lane's hidden unit
  __builtin_amdgcn_s_setprio(YT8M_EPI_PRIO);
  // publishes one value per (row, unit) pair into image `m`, column block `cb` (in units of H): 16-byte pieces of four units
  auto publish = [&](int m, int T, const float (&v)[NP], int cb) {
    const int QW = (m & 1) ? (H >> 3) : (H >> 4);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const float h1 = row_shl<1>(v[j]), h2 = row_shl<2>(v[j]), h3 = row_shl<3>(v[j]);
      if ((eunit & 3) == 0) {
        u32x4 x;
        x.x = __float_as_uint(v[j]); x.y = __float_as_uint(h1); x.z = __float_as_uint(h2); x.w = __float_as_uint(h3);
        const int erow = (64 / U) * j + rsub;
        const unsigned off = ((unsigned)(T * QW + cb * (H >> 4) + (ucol >> 4)) * 256u + (unsigned)(erow * 16 + (ucol & 15))) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(x, image(m), (int)off, 0, YT8M_AUX_ST);
      }
    }
  };
  auto arrive = [&](int T) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
      __hip_atomic_fetch_add(a.ctl + CTL_HDR + (T * NSH + (blockIdx.x & (NSH - 1))) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // start of step t for a pair whose dL/dh_t (without dout_t) is `dh`: dzc_t, dzu_t, and dh u
  auto step_start = [&](int t, int br, bool valid, long long idx, float dh, float& dzc_v, float& dzu_v, float& dhu) {
    const bool live = a.nf ? (t < a.nf[br]) : true;
    const float u = a.zg[((long long)t * B + br) * 2 * H + H + ucol];
    const float c = a.zc[idx];
    const float hp = a.hs[idx];
    const float d = dh + ((a.dout && live) ? a.dout[idx] : 0.f);
    dzc_v = live ? d * (1.0f - u) * (1.0f - c * c) : 0.f;
    dzu_v = live ? d * (hp - c) * u * (1.0f - u) : 0.f;
    dhu = live ? d * u : d;
    if (!valid) { dzc_v = 0.f; dzu_v = 0.f; dhu = 0.f; }
  };
  // m = 0: the launch's last step from the incoming dL/dh.  dzu of a step is known when its dzc is: it goes into the NEXT image's second
  // column block right away (behind the arrival of this one), off the chain of the candidate half-step that completes that image.
  {
    const int t = a.t0 + a.T - 1;
    for (int it = ew; it < n_it; it += NEPI) {
      const int T = g + it * RB;
      float pub[NP], pu[NP], dhu[NP];
      bool valid[NP];
      int brr[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int brow = T * 16 + (64 / U) * j + rsub;
        valid[j] = brow < B;
        brr[j] = valid[j] ? brow : B - 1;
        const long long idx = ((long long)t * B + brr[j]) * H + ucol;
        step_start(t, brr[j], valid[j], idx, a.work[(long long)brr[j] * H + ucol], pub[j], pu[j], dhu[j]);
      }
      publish(0, T, pub, 0);
      arrive(T);
      publish(1, T, pu, 1);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        if (valid[j]) {
          const long long idx = ((long long)t * B + brr[j]) * H + ucol;
          a.dzc[idx] = pub[j];
          a.dzg[((long long)t * B + brr[j]) * 2 * H + H + ucol] = pu[j];
          a.work[(long long)brr[j] * H + ucol] = dhu[j];                              // dh_{t-1} so far
        }
      }
    }
  }
  for (int m = 1; m <= HT; ++m) {
    const int s = (m - 1) >> 1;
    const bool gates = ((m - 1) & 1) != 0;
    const int t = a.t0 + a.T - 1 - s;
    for (int it = ew; it < n_it; it += NEPI) {
      const int k = (m - 1) * n_it + it;
      const int T = g + it * RB;
      const int slot = k & (NSLOT - 1);
      // operands that do not depend on the product (candidate half-step: of step t; gate half-step: of step t - 1, whose start is
      // this half-step's epilogue -- its saved activations are HBM-cold, so they are requested before the wait, not behind it)
      float hp[NP], rv[NP], acc_dh[NP];
      float u1[NP], c1[NP], hp1[NP], do1[NP];
      bool valid[NP], live[NP], live1[NP];
      int brr[NP];
      const bool next = gates && m < HT;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int brow = T * 16 + (64 / U) * j + rsub;
        valid[j] = brow < B;
        brr[j] = valid[j] ? brow : B - 1;
        const long long idx = ((long long)t * B + brr[j]) * H + ucol;
        const int nfb = a.nf ? a.nf[brr[j]] : 0x7fffffff;
        live[j] = t < nfb;
        live1[j] = t - 1 < nfb;
        acc_dh[j] = a.work[(long long)brr[j] * H + ucol];
        hp[j] = 0.f; rv[j] = 0.f; u1[j] = 0.f; c1[j] = 0.f; hp1[j] = 0.f; do1[j] = 0.f;
        if (!gates) {
          hp[j] = a.hs[idx];
          rv[j] = a.zg[((long long)t * B + brr[j]) * 2 * H + ucol];
        } else if (next) {
          const long long idx1 = idx - (long long)B * H;
          u1[j] = a.zg[((long long)(t - 1) * B + brr[j]) * 2 * H + H + ucol];
          c1[j] = a.zc[idx1];
          hp1[j] = a.hs[idx1];
          do1[j] = a.dout ? a.dout[idx1] : 0.f;
        }
      }
      lds_wait_ge(&lds_cnt[slot], 8u * (unsigned)(k / NSLOT + 1), a.ctl);
      // C layout of the 16x16 tile: column = lane & 15, row = 4 (lane >> 4) + reg
      float sum[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int erow = (64 / U) * j + rsub;
        const int r = erow & 3, lb = (erow >> 2) * 16;
        sum[j] = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) sum[j] += red[slot][wv][r][lb + eunit];
      }
      if (lane == 0) __hip_atomic_fetch_add(&lds_free[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (!gates) {
        // sum = d(r*h): dzr, dh_{t-1} += drh r; dzr completes image m (dzu is already there)
        float pr[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const float drh = (live[j] && valid[j]) ? sum[j] : 0.f;
          pr[j] = drh * hp[j] * rv[j] * (1.0f - rv[j]);
          acc_dh[j] += drh * rv[j];
        }
        publish(m, T, pr, 0);
        arrive(T);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          if (valid[j]) {
            a.dzg[((long long)t * B + brr[j]) * 2 * H + ucol] = pr[j];
            a.work[(long long)brr[j] * H + ucol] = acc_dh[j];
          }
        }
      } else {
        // sum = dzg . Wg_h^T: dL/dh_{t-1} complete; start step t - 1 (or leave dh in `work` for the next launch)
        float pub[NP], pu[NP], keep[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const float dh = acc_dh[j] + ((live[j] && valid[j]) ? sum[j] : 0.f);
          const float d = dh + (live1[j] ? do1[j] : 0.f);
          pub[j] = (next && live1[j] && valid[j]) ? d * (1.0f - u1[j]) * (1.0f - c1[j] * c1[j]) : 0.f;
          pu[j] = (next && live1[j] && valid[j]) ? d * (hp1[j] - c1[j]) * u1[j] * (1.0f - u1[j]) : 0.f;
          keep[j] = next ? (live1[j] ? d * u1[j] : d) : dh;
        }
        if (next) {
          publish(m, T, pub, 0);
          arrive(T);
          publish(m + 1, T, pu, 1);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          if (valid[j]) {
            if (next) {
              const long long idx1 = ((long long)(t - 1) * B + brr[j]) * H + ucol;
              a.dzc[idx1] = pub[j];
              a.dzg[((long long)(t - 1) * B + brr[j]) * 2 * H + H + ucol] = pu[j];
            }
            a.work[(long long)brr[j] * H + ucol] = keep[j];
          }
        }
      }
    }
  }
  if (ew == 0 && lane == 0) { check_placement(a.ctl, a.stats); propagate_error(a.ctl); }
}
