// Backward of the small-graph time loop (N <= 6 objects), gfx950: the adjoint of gnn_small.hip in the same
// wave-per-node-row (half-wave-per-node-row in the kernels for five or six objects), register-chained form, split in two launches:
//   dyn_loop_bwd_small_k  walks the T-serial chain backwards (data gradients only: dz1, dzsup, dzsstd, dextra) and
//                         streams every layer's pre-activation gradient ("dY") to HBM;
//   gnn_dw_small_k        turns the saved layer inputs and the dY streams into weight gradients as a throughput
//                         GEMM: 16 consecutive (time, row) rows per MFMA tile, i.e. full tiles instead of the 3 useful
//                         rows of 16 the in-loop accumulation of gnn.hip had, and off the latency-critical chain.
// Formulas follow gnn_backward (gnn.hip, stages b1-b13), which is validated against the reference goldens.
#include "common.h"

namespace stove {

// ---- dY streams of one sequence (same row indexing as the activation streams, sm_act2) ----------------------
struct SmDy {
  float *dRES, *dO1, *dF3, *duF1, *dF1p, *dSD, *dH1p, *dEnc, *dP, *dR3, *E32, *dA2p, *dR1p, *dA1p, *dq;
};
__host__ __device__ inline size_t sm_dy_floats(int N, int Ts) {
  const size_t f = (size_t)Ts * ((size_t)N * (8 * 32 + 256) + (size_t)N * (N - 1) * (3 * 32 + 2 * 64 + 1));
  return (f + 3) & ~(size_t)3;       // every sequence's streams start 16-byte aligned
}
__device__ __forceinline__ SmDy sm_dy(float* seq, int N, int Ts, int ts) {
  SmDy d;
  const size_t nrows = (size_t)Ts * N, erows = (size_t)Ts * N * (N - 1);
  float* nb = seq + (size_t)ts * N * 32;
  d.dRES = nb; d.dO1 = nb + nrows * 32; d.dF3 = nb + 2 * nrows * 32; d.duF1 = nb + 3 * nrows * 32; d.dF1p = nb + 4 * nrows * 32;
  d.dSD = nb + 5 * nrows * 32; d.dH1p = nb + 6 * nrows * 32; d.dEnc = nb + 7 * nrows * 32;
  d.dP = seq + nrows * 8 * 32 + (size_t)ts * N * 256;
  float* eb = seq + nrows * (8 * 32 + 256);
  const size_t eo = (size_t)ts * N * (N - 1);
  d.dR3 = eb + eo * 32; d.E32 = eb + erows * 32 + eo * 32; d.dA2p = eb + 2 * erows * 32 + eo * 32;
  float* e64 = eb + 3 * erows * 32;
  d.dR1p = e64 + eo * 64; d.dA1p = e64 + erows * 64 + eo * 64;
  d.dq = e64 + 2 * erows * 64 + eo;
  return d;
}

// LDS of the backward kernel (floats): W^T images repacked for row-per-lane dots, vectors, exchange buffers
struct SmBLds {
  float *W, *V, *DSD, *EG, *DD, *POS, *DP;
};
template <int NMX>
constexpr int smb_lds_floats() { return W_END + V_END + NMX * 32 + SmShape<NMX>::NE * 128 + 2 * ((SmShape<NMX>::NE + 3) & ~3) + 2 * NMX * 4 + NMX * 256; }
template <int NMX>
__device__ __forceinline__ SmBLds smb_carve(float* base) {
  constexpr int NE = SmShape<NMX>::NE, NE4 = (NE + 3) & ~3;
  SmBLds L;
  L.W = base;
  L.V = L.W + W_END;
  L.DSD = L.V + V_END;          // [NMX][32]   dL/dPRED of every node row
  L.EG = L.DSD + NMX * 32;      // [NMX^2][dR1pre 64 | dA1pre 64] by edge row i*N + j
  L.DD = L.EG + NE * 128;       // [2][NMX^2]  dL/d dist of every edge: relation-chain part | attention-chain part
  L.POS = L.DD + 2 * NE4;       // [2][NMX][4] positions (two parities: a fast wave may already write the next step's)
  L.DP = L.POS + 2 * NMX * 4;   // [NMX][256]  dP rows for the edge-first transpose product
  return L;
}
// transposed layer image (rows = layer inputs, K = layer outputs) -> [K/4][rows][4]
__device__ __forceinline__ void smb_setup(const SmBLds& L, const float* __restrict__ P) {
  sm_copy_packed(L.W, P + P_WTPACK, W_END);         // the transposed layers, packed (image section 5)
  for (int i = threadIdx.x; i < V_END; i += blockDim.x) L.V[i] = P[2 * W_END + i];
}

// what a node wave needs of step ts, fetched one step ahead
struct SmBNodeIn {
  float RES, SIN, O1, F1, F2, H1, S;         // lane l (and l + 32): element l of the row
  float ep, ms, ss, gz, gmu_in, gsg_in, gzd_in, dpred;
};
__device__ __forceinline__ SmBNodeIn smb_node_load(const SmAct& a, int r, int l, size_t o, const float* __restrict__ eps,
                                                   const float* __restrict__ zsup, const float* __restrict__ zsstd,
                                                   const float* __restrict__ dz, const float* __restrict__ dzdyn,
                                                   const float* __restrict__ dmean, const float* __restrict__ dstd,
                                                   const float* __restrict__ dpred) {
  SmBNodeIn n;
  n.RES = a.RES[r * 32 + l];
  n.SIN = a.SIN[r * 32 + l];
  n.O1 = a.O1[r * 32 + l];
  n.F1 = a.F1[r * 32 + l];
  n.F2 = a.F2[r * 32 + l];
  n.H1 = a.H1[r * 32 + l];
  n.S = a.S[r * 32 + l];
  n.ep = n.gz = n.gmu_in = n.gsg_in = n.gzd_in = 0.0f;
  n.ms = 0.0f;
  n.ss = 1.0f;
  const int q = l < 16 ? l + 2 : l - 16;          // dim of the 18-vector owned by this lane (lanes 0..17)
  if (l < 18) {
    n.ep = eps[o * 18 + q];
    if (dz != nullptr) n.gz = dz[o * 18 + q];
    if (dmean != nullptr) n.gmu_in = dmean[o * 18 + q];
    if (dstd != nullptr) n.gsg_in = dstd[o * 18 + q];
    if (l < 4 || l >= 16) {
      n.ms = zsup[o * 6 + (l < 4 ? 2 + l : l - 16)];
      n.ss = zsstd[o * 6 + (l < 4 ? 2 + l : l - 16)];
    }
    if (l < 16 && dzdyn != nullptr) n.gzd_in = dzdyn[o * 16 + l];
  }
  n.dpred = dpred != nullptr ? dpred[o * 32 + l] : 0.0f;
  return n;
}

// ---- edge phase of the backward on the matrix cores (the adjoint of sm_edge_phase_mfma, gnn_small.hip) ------------------
// Columns = edges; wave 3 walks the relation chain backwards, wave 2 the attention chain, through the transposed weight images:
//     dq = (dSD_i . R3) att,  dR3 = dSD_i att,  E32 = (W_R2^T dR3 + dR3) phi'(R2),  dR1 = (W_R1^T E32) phi'(R1)          (wave 3)
//     dA2 = dq w_a2 phi'(A2),  dA1 = (W_A1^T dA2) phi'(A1)                                                              (wave 2)
// and each chain's share of dL/d dist = dX1 . w_d.  In the row-per-lane form every wave took its edges one after the other (two
// rounds for N = 3); here all edges are one 48- (32-) MFMA chain.  Saved forward values arrive in the accumulator layout
// (lane = (edge column, lane >> 4), four consecutive features per float4) straight from the activation streams.
struct SmBEdgeIn {
  smf4 r3[2], y2[2], x1[4];       // R3, R2 | A2, R1 | A1 of the lane's edge
  float att;
};
__device__ __forceinline__ SmBEdgeIn smb_edge_load(const SmAct& a, const SmEdgeLane& el, int h) {
  SmBEdgeIn in;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    in.r3[t] = sm_ld4(a.R3 + el.s32 + 16 * t);
    in.y2[t] = sm_ld4((h ? a.A2 : a.R2) + el.s32 + 16 * t);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) in.x1[t] = sm_ld4((h ? a.A1 : a.R1) + el.s64 + 16 * t);
  in.att = a.ATT[el.s1];
  return in;
}
// OT output tiles x ET column tiles through the transposed image Wl = [K/4][16 OT][4]: one fetch of the A operands, the column
// tiles' chains interleaved (see sm_mfma_layer_tiles, gnn_small.hip)
template <int KT, int OT, int ET>
__device__ __forceinline__ void smb_mfma_layer_t(const float* Wl, const smf4 (&x)[ET][KT], smf4 (&acc)[ET][OT], int lane) {
  constexpr int ROWS = 16 * OT;
  const int i = lane & 15, g = lane >> 4;
  float4 w[KT][OT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int u = 0; u < OT; ++u) w[t][u] = *reinterpret_cast<const float4*>(Wl + ((4 * t + g) * ROWS + 16 * u + i) * 4);
#pragma unroll
  for (int t = 0; t < KT; ++t) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int u = 0; u < OT; ++u) {
        const float wv = e == 0 ? w[t][u].x : (e == 1 ? w[t][u].y : (e == 2 ? w[t][u].z : w[t][u].w));
#pragma unroll
        for (int c = 0; c < ET; ++c) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, x[c][t][e], acc[c][u], 0, 0, 0);
      }
  }
}
// Round 5: the two transposed layers of each chain on half-piece MFMAs with column-normalised operands (gnn_small.hip
// sm_split_layer_t_norm: 6 + 12 MFMAs of 16 cycles instead of 16 + 32 fp32 MFMAs that ran at ~50 cycles each in their dependent
// chains); the hi / lo fragments of the transposed weights are built once per launch and stay in registers.
struct SmBChainW {
  bf16x8 h1[4], l1[4];      // W_R1^T (relation chain) / W_A1^T (attention chain): 32 -> 64
  bf16x8 h2[2], l2[2];      // W_R2^T: 32 -> 32 (relation chain)
};
__device__ __forceinline__ void smb_chain_wbuild(const SmBLds& L, SmBChainW& cw) {
  const int wv = wave_id(), lane = lane_id();
  if (wv < 2) return;
  const int h = wv == 2 ? 1 : 0;
  sm_chain_wfrag_t<4>(L.W + (h ? W_A1 : W_R1), lane, cw.h1, cw.l1);
  if (h == 0) sm_chain_wfrag_t<2>(L.W + W_R2, lane, cw.h2, cw.l2);
}
template <int ET>
__device__ __forceinline__ void smb_edge_phase_mfma(const SmBLds& L, const SmEdgeLane (&el)[ET], const SmBEdgeIn (&in)[ET], const SmDy& g,
                                                    int elu, int dd_stride, const SmBChainW& cw) {
  const int wv = wave_id(), lane = lane_id(), gq = lane >> 4;
  if (wv < 2) return;
  const int h = wv == 2 ? 1 : 0;
  const float* V = L.V;
  smf4 dsd[ET][2], y[ET][2], d1[ET][4];
  float dq[ET];
#pragma unroll
  for (int c = 0; c < ET; ++c) {
    float pq = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      dsd[c][t] = sm_ld4(L.DSD + el[c].i32 + 16 * t);
#pragma unroll
      for (int e = 0; e < 4; ++e) pq = fmaf(dsd[c][t][e], in[c].r3[t][e], pq);
    }
    pq = sum_xor32(sum_xor16(pq));
    dq[c] = pq * in[c].att;
  }
  if (h == 0) {
    smf4 dr3[ET][2], acc[ET][2];
#pragma unroll
    for (int c = 0; c < ET; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dr3[c][t] = dsd[c][t] * in[c].att;
        acc[c][t] = smf4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    sm_split_layer_t_norm<2, ET>(cw.h2, cw.l2, dr3, acc);
#pragma unroll
    for (int c = 0; c < ET; ++c) {
#pragma unroll
      for (int t = 0; t < 2; ++t) y[c][t] = (acc[c][t] + dr3[c][t]) * sm_dphi4(in[c].y2[t], elu);
      if (el[c].valid) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          sm_st4(g.dR3 + el[c].s32 + 16 * t, dr3[c][t]);
          sm_st4(g.E32 + el[c].s32 + 16 * t, y[c][t]);
        }
        if (gq == 0) g.dq[el[c].s1] = dq[c];
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < ET; ++c) {
#pragma unroll
      for (int t = 0; t < 2; ++t) y[c][t] = sm_ld4(V + V_WA2 + 16 * t + 4 * gq) * dq[c] * sm_dphi4(in[c].y2[t], elu);
      if (el[c].valid) {
#pragma unroll
        for (int t = 0; t < 2; ++t) sm_st4(g.dA2p + el[c].s32 + 16 * t, y[c][t]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < ET; ++c)
#pragma unroll
    for (int u = 0; u < 4; ++u) d1[c][u] = smf4{0.0f, 0.0f, 0.0f, 0.0f};
  sm_split_layer_t_norm<4, ET>(cw.h1, cw.l1, y, d1);
#pragma unroll
  for (int c = 0; c < ET; ++c) {
    float dd = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      d1[c][u] = d1[c][u] * sm_dphi4(in[c].x1[u], elu);
      const smf4 wd = sm_ld4(V + (h ? V_WDA : V_WDR) + 16 * u + 4 * gq);
#pragma unroll
      for (int e = 0; e < 4; ++e) dd = fmaf(d1[c][u][e], wd[e], dd);
    }
    dd = sum_xor32(sum_xor16(dd));
    if (el[c].valid) {
      float* eg = L.EG + el[c].e * 128 + 64 * h + 4 * gq;
      float* g1 = g.dR1p + (h ? g.dA1p - g.dR1p : (ptrdiff_t)0) + el[c].s64;      // (a select between the two pointers went through scratch memory)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sm_st4(eg + 16 * u, d1[c][u]);
        sm_st4(g1 + 16 * u, d1[c][u]);
      }
      if (gq == 0) L.DD[dd_stride * h + el[c].e] = dd;
    }
  }
}

template <int NMX, bool ELU, int NT, bool HEAD, bool STAMP = false>
__global__ __launch_bounds__(64 * kSmWaves) void dyn_loop_bwd_small_k(
    const float* __restrict__ zsup, const float* __restrict__ zsstd, const float* __restrict__ eps, const float* __restrict__ P,
    float* __restrict__ act, const float* __restrict__ dz, const float* __restrict__ dzdyn, const float* __restrict__ dmean,
    const float* __restrict__ dstd, const float* __restrict__ dpred, float* __restrict__ dz1, float* __restrict__ dzsup,
    float* __restrict__ dzsstd, float* __restrict__ dextra, float* __restrict__ dy, int B, int Ts, int N, int sin_dim, int lim_enc,
    int elu, LoopConst kc, long long* stamps, int ts0, int ts1, float* __restrict__ carry) {
  // steps ts1-1 .. ts0 of the Ts the tensors are laid out for (see dyn_loop_fwd_small_k).  The gradient that flows from step ts0
  // into the state before it leaves through `carry` (B, N, 18; layout of dz1) when ts0 > 0 and enters there when ts1 < Ts.
  constexpr int RP = SmShape<NMX>::RP, ET = SmShape<NMX>::ET, NE4 = (SmShape<NMX>::NE + 3) & ~3;
  constexpr bool HK = RP == 1;       // one node row per wave: the half-waves split K of the row-per-lane dots (gnn_small.hip sm_dotw)
  static_assert(NT <= NMX, "object count beyond what the kernel is built for");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // the chain is issue-latency bound: where another kernel's wave shares the SIMD (the table gradients that run underneath,
  // objspn_tablegrad_under_k) the arbiter picks this one whenever it is ready
  __builtin_amdgcn_s_setprio(3);
  const SmBLds L = smb_carve<NMX>(lds);
  const int b = blockIdx.x;
  const int wv = wave_id(), lane = lane_id(), l = lane & 31, h = lane >> 5;
  elu = ELU ? 1 : 0;      // compile-time activation (see dyn_loop_fwd_small_k)
  if (NT > 0) N = NT;
  if (!STAMP) stamps = nullptr;
  if (HEAD) {      // the training call of the plain model: all four state gradients present, no reward head, no extra inputs
    __builtin_assume(dz != nullptr);
    __builtin_assume(dzdyn != nullptr);
    __builtin_assume(dmean != nullptr);
    __builtin_assume(dstd != nullptr);
    dpred = nullptr;
    sin_dim = 16;
    lim_enc = 2;
  }
  SmCfg cf{N, sin_dim, lim_enc, elu};
  cf.stamps = nullptr;
  const int E = sin_dim - 16;
  const float* V = L.V;
  SmEdgeLane el[ET];
#pragma unroll
  for (int t = 0; t < ET; ++t) el[t] = sm_edge_lane(N, 1, t);
  smb_setup(L, P);
  float* aseq = act + (size_t)b * sm_act2_floats(N, Ts);
  float* dseq = dy + (size_t)b * sm_dy_floats(N, Ts);
  // the lane's node row (see sm_step, gnn_small.hip): one per wave, or one per half-wave
  const int r = RP == 2 ? wv + 4 * h : wv;
  const bool row = r < N;
  const bool own = RP == 2 ? row : (lane < 32 && row);
  const int rs = row ? r : wv;
  const bool node = wv < N;
  float car = 0.0f;                                   // lane d < 16: gradient carried into z[t][2 + d] from step t + 1
  SmBNodeIn nin{};
  if (row) {
    const SmAct a = sm_act2(aseq, N, Ts, ts1 - 1);
    nin = smb_node_load(a, r, l, ((size_t)b * Ts + ts1 - 1) * N + r, eps, zsup, zsstd, dz, dzdyn, dmean, dstd, dpred);
    if (ts1 < Ts && l < 16) car = carry[((size_t)b * N + r) * 18 + 2 + l];
  }
  WG_SYNC();
  SmBChainW cw;
  smb_chain_wbuild(L, cw);      // the edge chains' transposed weights as half-piece fragments: registers, for all time steps
  for (int ts = ts1 - 1; ts >= ts0; --ts) {
    cf.stamps = (ts == ts0 + 1 && stamps != nullptr) ? stamps + 64 : nullptr;       // second half of the debug buffer
    sm_stamp(cf, 0);
    const SmAct a = sm_act2(aseq, N, Ts, ts);
    const SmDy g = sm_dy(dseq, N, Ts, ts);
    const size_t o = ((size_t)b * Ts + ts) * N + r;
    const SmBNodeIn cur = nin;
    if (row && ts > ts0) {
      const SmAct an = sm_act2(aseq, N, Ts, ts - 1);
      nin = smb_node_load(an, r, l, o - N, eps, zsup, zsstd, dz, dzdyn, dmean, dstd, dpred);
    }
    // this wave's edges of the step: saved forward values, in flight across the node phase
    SmBEdgeIn ein[ET] = {};
    if (wv >= 2) {
#pragma unroll
      for (int t = 0; t < ET; ++t) ein[t] = smb_edge_load(a, el[t], wv == 2 ? 1 : 0);
    }
    float* pos = L.POS + (ts & 1) * (NMX * 4);
    float dS_o0 = 0.0f, dSD = 0.0f, pc = 0.0f;
    // ---- Q4: node rows: epilogue, output and affector MLPs backwards ----------------------------------------------
    if (node) {
      SmW<8> wa = sm_wload<8, HK>(L.W + W_O1, 32, l), wb;
      if (l < 2 && own) pos[r * 4 + l] = cur.S;
      // epilogue backward (dyn_loop_bwd_k of gnn.hip, per (row, q)): lane d < 16 owns q = d + 2, lanes 16/17 q = 0/1
      const float res_hi = from_xor16(cur.RES);
      const float gz = cur.gz + (l < 16 ? car : 0.0f);
      const float gmu = gz + cur.gmu_in, gsg = gz * cur.ep + cur.gsg_in;
      float lo = 0.0f, hi = 0.0f;
      if (l < 16) {
        const int d = l;
        // reciprocals by v_rcp_f32 (1 ulp) instead of IEEE division sequences (~10 instructions each, four of them on the
        // serial chain of every step): these values only scale gradients
        const float kd = std_scale(d, kc);
        const float m = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-cur.RES)) - 1.0f;
        const float sg_hi = __builtin_amdgcn_rcpf(1.0f + __expf(-res_hi));
        const float sd = kd * sg_hi;
        const float zd = m + (d < 2 ? cur.SIN : 0.0f);
        float gzd = cur.gzd_in, gsd;
        if (d < 4) {
          const float ms = cur.ms, ss = cur.ss;
          const float sd2 = sd * sd, ss2 = ss * ss, D = sd2 + ss2, iD = __builtin_amdgcn_rcpf(D);
          const float mu = (ss2 * zd + sd2 * ms) * iD;
          const float rD = rsqrtf(D);
          gzd += gmu * ss2 * iD;
          gsd = gmu * (ms - mu) * iD * 2.0f * sd + gsg * ss * ss2 * iD * rD;
          if (own) {
            dzsup[o * 6 + 2 + d] = gmu * sd2 * iD;
            dzsstd[o * 6 + 2 + d] = gmu * (zd - mu) * iD * 2.0f * ss + gsg * sd * sd2 * iD * rD;
          }
        } else {
          gzd += gmu;
          gsd = gsg;
        }
        if (d < 2) pc = gzd;                              // z_dyn position = previous position + delta
        lo = gzd * 0.5f * (1.0f - m * m);
        hi = gsd * sd * (1.0f - sg_hi);
      } else if (l < 18 && own) {
        dzsup[o * 6 + (l - 16)] = gmu;
        dzsstd[o * 6 + (l - 16)] = gsg;
      }
      const float hi_from = from_xor16(hi);      // lane l >= 16 takes hi of lane l - 16
      const float dres = l < 16 ? lo : hi_from;
      sm_stamp(cf, 6);
      // b1. out.1
      wb = sm_wload<8, HK>(L.W + W_O0, 64, l);
      const float db = (sm_dotw<8, HK>(wa, dres) + dres) * (1.0f - cur.O1 * cur.O1);
      // b2. out.0 on [F3 | S]
      wa = sm_wload<8, HK>(L.W + W_O0, 64, 32 + l);
      const float dF3 = sm_dotw<8, HK>(wb, db);
      wb = sm_wload<8, HK>(L.W + W_F2, 32, l);
      dS_o0 = sm_dotw<8, HK>(wa, db);
      // b3. affector.2
      wa = sm_wload<8, HK>(L.W + W_F1, 32, l);
      const float dF2 = sm_dotw<8, HK>(wb, dF3);
      const float th = cur.F2 - cur.F1;
      const float du = dF2 * (1.0f - th * th);
      // b4. affector.1
      wb = sm_wload<8, HK>(L.W + W_F0, 32, l);
      const float dbf = (sm_dotw<8, HK>(wa, du) + dF2) * (1.0f - cur.F1 * cur.F1);
      // b5. affector.0
      dSD = sm_dotw<8, HK>(wb, dbf) + cur.dpred;
      if (own) {
        L.DSD[r * 32 + l] = dSD;
        g.dRES[r * 32 + l] = dres;
        g.dO1[r * 32 + l] = db;
        g.dF3[r * 32 + l] = dF3;
        g.duF1[r * 32 + l] = du;
        g.dF1p[r * 32 + l] = dbf;
        g.dSD[r * 32 + l] = dSD;
      }
    }
    sm_stamp(cf, 1);
    WG_SYNC();
    sm_stamp(cf, 2);
    // ---- Q3: edges as the columns of the relation chain (wave 3) / the attention chain (wave 2), all column tiles together
    smb_edge_phase_mfma<ET>(L, el, ein, g, elu, NE4, cw);
    sm_stamp(cf, 3);
    WG_SYNC();
    sm_stamp(cf, 4);
    // ---- Q2: node rows: first edge layer, self-dynamics, encoder backwards ------------------------------------------
    if (node) {
      float dS_edge = 0.0f;
      // a row's dP (256 wide) and its product with the edge-first weights take all 64 lanes: the wave's rows one after the other
#pragma unroll
      for (int rq = 0; rq < RP; ++rq) {
        const int rw = wv + 4 * rq;
        if (rq > 0 && rw >= N) break;
        // dP[rw][c], c = lane + 64 g: relation / attention, rw as first (s_i) or second (s_j) argument
        float dp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = 0; j < N; ++j) {
          if (j == rw) continue;
          const float* e1 = L.EG + (rw * N + j) * 128;
          const float* e2 = L.EG + (j * N + rw) * 128;
          dp[0] += e1[lane];
          dp[1] += e2[lane];
          dp[2] += e1[64 + lane];
          dp[3] += e2[64 + lane];
        }
        float* dpr = L.DP + rw * 256;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          dpr[lane + 64 * gq] = dp[gq];
          g.dP[rw * 256 + lane + 64 * gq] = dp[gq];
        }
        sm_stamp(cf, 7);
        // dS from the edge layers: dP (256) W_ef (256 x 32); the two half-waves split the 256 terms
        v2f s0 = {0.0f, 0.0f}, s1 = {0.0f, 0.0f};
        // four rounds of 8 weight + 8 operand reads, all of a round in flight before its FMAs (at 4 per round the loop was
        // LDS latency: 3.0 k cycles of a 15.4 k step, profiles/r02_loop_stamps_bwd.txt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 w[8], x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            w[u] = *reinterpret_cast<const float4*>(L.W + W_EF + ((32 * h + 8 * q + u) * 32 + l) * 4);
            x[u] = *reinterpret_cast<const float4*>(dpr + 128 * h + 4 * (8 * q + u));
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            s0 = pk_fma(v2f{w[u].x, w[u].y}, v2f{x[u].x, x[u].y}, s0);
            s1 = pk_fma(v2f{w[u].z, w[u].w}, v2f{x[u].z, x[u].w}, s1);
          }
        }
        s0 += s1;
        float de = s0.x + s0.y;
        de = sum_xor32(de);
        if (RP == 1 || rq == h) dS_edge = de;           // element l of row rw: kept by the half-wave that carries the row
      }
      sm_stamp(cf, 8);
      SmW<8> wa = sm_wload<8, HK>(L.W + W_S1, 32, l);
      // b11. self.1:  SD = H1 W^T + b + H1
      SmW<8> wb = sm_wload<8, HK>(L.W + W_S0, 32, l);
      const float dH1p = (sm_dotw<8, HK>(wa, dSD) + dSD) * dphi_from_out(cur.H1, elu);
      // b12. self.0 ; total dS ; split into the encoder output part and the pass-through part
      wa = sm_wload<8, HK>(L.W + W_ENC, 32, l);
      float tot = sm_dotw<8, HK>(wb, dH1p) + dS_edge + dS_o0;
      if (l < 2) {
        float dd = 0.0f;
        for (int j = 0; j < N; ++j)
          if (j != rs) dd += 2.0f * (pos[rs * 4 + l] - pos[j * 4 + l]) * ((L.DD[rs * N + j] + L.DD[NE4 + rs * N + j]) + (L.DD[j * N + rs] + L.DD[NE4 + j * N + rs]));
        tot += dd;
      }
      const bool raw = l < lim_enc;
      const float dEnc = raw ? 0.0f : tot;
      // b13. encoder
      const float dsin = sm_dotw<8, HK>(wa, dEnc) + (raw ? tot : 0.0f);
      if (own) {
        g.dH1p[r * 32 + l] = dH1p;
        g.dEnc[r * 32 + l] = dEnc;
        if (l >= 16 && l < sin_dim) dextra[o * E + (l - 16)] = dsin;
      }
      car = (l < 16) ? dsin + (l < 2 ? pc : 0.0f) : 0.0f;
    }
    sm_stamp(cf, 5);
  }
  if (node) {
    const float shifted = sm_from_lane(car, (lane + 62) & 63);      // all lanes take part: a bpermute reads 0 from inactive source lanes
    float* dst = ts0 == 0 ? dz1 : carry;
    if (l < 18 && own) dst[((size_t)b * N + r) * 18 + l] = l < 2 ? 0.0f : shifted;
  }
}

// =================================================================================================
// weight gradients from the streams: one workgroup per sequence (the partial images are reduced by reduce_chunks_k)
// =================================================================================================
// 16 consecutive rows of every stream form one MFMA K-block.  A row block is fetched with coalesced float4 loads into
// registers while the previous block's MFMAs run, then dropped into padded LDS buffers, from which the validated
// dW_layer / vec_layer of gnn.hip (step-major interleaved MFMAs) accumulate -- full 16-row tiles instead of the 3 useful
// rows of 16 the in-loop accumulation has, and no per-tile global-load latency (a first version that fed the MFMAs
// straight from global memory spent 1 us per tile waiting: 0.98 ms per launch).
struct DwLds {
  float *SIN, *H1, *PRED, *F1, *F2, *O1, *CAT;                          // layer inputs, [16][LDN] ([16][LDC] for CAT = [F3 | S])
  float *dEnc, *dH1p, *dSD, *dF1p, *duF1, *dF3, *dO1, *dRES, *dP;       // dY, [16][LDN] ([16][LDP] for dP)
  float *R1, *A1, *dR1p, *dA1p;                                         // edge pass, [16][LDC]
  float *R2, *A2, *E32, *dA2p, *dR3;                                    // edge pass, [16][LDN]
  float *AUXN, *AUXE;                                                   // [16][16]
};
constexpr int kDwLdsFloats = 14 * 16 * LDN + 16 * LDC + 16 * LDP + 4 * 16 * LDC + 5 * 16 * LDN + 2 * 256;
__device__ __forceinline__ DwLds dw_carve(float* p) {
  DwLds L;
  auto take = [&](int n) { float* q = p; p += n; return q; };
  L.SIN = take(16 * LDN); L.H1 = take(16 * LDN); L.PRED = take(16 * LDN); L.F1 = take(16 * LDN); L.F2 = take(16 * LDN);
  L.O1 = take(16 * LDN); L.CAT = take(16 * LDC);
  L.dEnc = take(16 * LDN); L.dH1p = take(16 * LDN); L.dSD = take(16 * LDN); L.dF1p = take(16 * LDN); L.duF1 = take(16 * LDN);
  L.dF3 = take(16 * LDN); L.dO1 = take(16 * LDN); L.dRES = take(16 * LDN); L.dP = take(16 * LDP);
  L.R1 = take(16 * LDC); L.A1 = take(16 * LDC); L.dR1p = take(16 * LDC); L.dA1p = take(16 * LDC);
  L.R2 = take(16 * LDN); L.A2 = take(16 * LDN); L.E32 = take(16 * LDN); L.dA2p = take(16 * LDN); L.dR3 = take(16 * LDN);
  L.AUXN = take(256); L.AUXE = take(256);
  return L;
}
// float4 number `idx` of the 16-row block starting at row r0 of a row-major stream of width W (zero beyond nrows)
template <int W>
__device__ __forceinline__ float4 dw_fetch(const float* __restrict__ src, int r0, int nrows, int idx) {
  const int row = idx / (W / 4);
  if (r0 + row >= nrows) return float4{0.0f, 0.0f, 0.0f, 0.0f};
  return reinterpret_cast<const float4*>(src + (size_t)r0 * W)[idx];
}
template <int W>
__device__ __forceinline__ void dw_drop(float* dst, int ld, int col0, int idx, float4 v) {
  const int row = idx / (W / 4), c4 = idx % (W / 4);
  *reinterpret_cast<float4*>(dst + row * ld + col0 + c4 * 4) = v;
}

__global__ __launch_bounds__(256) void gnn_dw_small_k(const float* __restrict__ act, const float* __restrict__ dy, float* __restrict__ gpart,
                                                      int B, int Ts, int N) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwLds L = dw_carve(lds);
  const int b = blockIdx.x, wv = wave_id(), tid = threadIdx.x;
  const SmAct a = sm_act2(const_cast<float*>(act) + (size_t)b * sm_act2_floats(N, Ts), N, Ts, 0);
  const SmDy g = sm_dy(const_cast<float*>(dy) + (size_t)b * sm_dy_floats(N, Ts), N, Ts, 0);
  f32x4 acc[SL_END], vacc[VSLOTS];
#pragma unroll
  for (int k = 0; k < SL_END; ++k) acc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < VSLOTS; ++k) vacc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = tid; i < 256; i += 256) {
    L.AUXN[i] = (i & 15) == 0 ? 1.0f : 0.0f;
    L.AUXE[i] = (i & 15) == 0 ? 1.0f : 0.0f;
  }
  const int nrows = Ts * N, erows = Ts * N * (N - 1);
  // ---- node rows: 16 streams of width 32 (two per thread half: threads 0..127 the first, 128..255 the second) + dP (256)
  const float* n32[16] = {a.SIN, a.H1, a.PRED, a.F1, a.F2, a.O1, a.F3, a.S, g.dEnc, g.dH1p, g.dSD, g.dF1p, g.duF1, g.dF3, g.dO1, g.dRES};
  float* d32[16] = {L.SIN, L.H1, L.PRED, L.F1, L.F2, L.O1, L.CAT, L.CAT, L.dEnc, L.dH1p, L.dSD, L.dF1p, L.duF1, L.dF3, L.dO1, L.dRES};
  const int half = tid >> 7, t7 = tid & 127;
  float4 rg[12];
  auto node_fetch = [&](int r0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) rg[p] = dw_fetch<32>(n32[2 * p + half], r0, nrows, t7);
#pragma unroll
    for (int q = 0; q < 4; ++q) rg[8 + q] = dw_fetch<256>(g.dP, r0, nrows, tid + 256 * q);
  };
  auto node_drop = [&]() {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int sidx = 2 * p + half;
      const int ld = (sidx == 6 || sidx == 7) ? LDC : LDN;
      dw_drop<32>(d32[sidx], ld, sidx == 7 ? 32 : 0, t7, rg[p]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) dw_drop<256>(L.dP, LDP, 0, tid + 256 * q, rg[8 + q]);
  };
  node_fetch(0);
  WG_SYNC();
  for (int r0 = 0; r0 < nrows; r0 += 16) {
    node_drop();
    WG_SYNC();
    if (r0 + 16 < nrows) node_fetch(r0 + 16);
    const float* S = L.CAT + 32;
    dW_layer<32, 32, SL_ENC>(acc, L.dEnc, LDN, L.SIN, LDN, 1, wv);
    dW_layer<32, 32, SL_S0>(acc, L.dH1p, LDN, S, LDC, 1, wv);
    dW_layer<32, 32, SL_S1>(acc, L.dSD, LDN, L.H1, LDN, 1, wv);
    dW_layer<256, 32, SL_EF>(acc, L.dP, LDP, S, LDC, 1, wv);
    dW_layer<32, 32, SL_F0>(acc, L.dF1p, LDN, L.PRED, LDN, 1, wv);
    dW_layer<32, 32, SL_F1>(acc, L.duF1, LDN, L.F1, LDN, 1, wv);
    dW_layer<32, 32, SL_F2>(acc, L.dF3, LDN, L.F2, LDN, 1, wv);
    dW_layer<32, 64, SL_O0>(acc, L.dO1, LDN, L.CAT, LDC, 1, wv);
    dW_layer<32, 32, SL_O1>(acc, L.dRES, LDN, L.O1, LDN, 1, wv);
    vec_layer<VT_ENC, 2>(vacc, L.dEnc, LDN, L.AUXN, 1, wv);
    vec_layer<VT_S0, 2>(vacc, L.dH1p, LDN, L.AUXN, 1, wv);
    vec_layer<VT_S1, 2>(vacc, L.dSD, LDN, L.AUXN, 1, wv);
    vec_layer<VT_F0, 2>(vacc, L.dF1p, LDN, L.AUXN, 1, wv);
    vec_layer<VT_F1, 2>(vacc, L.duF1, LDN, L.AUXN, 1, wv);
    vec_layer<VT_F2, 2>(vacc, L.dF3, LDN, L.AUXN, 1, wv);
    vec_layer<VT_O0, 2>(vacc, L.dO1, LDN, L.AUXN, 1, wv);
    vec_layer<VT_O1, 2>(vacc, L.dRES, LDN, L.AUXN, 1, wv);
    WG_SYNC();
  }
  // ---- edge rows: four 64-wide streams (one float4 per thread each), five 32-wide (threads 0..127 / 128..255), dist and dq
  const float* e32[6] = {a.R2, a.A2, g.E32, g.dA2p, g.dR3, g.dR3};
  float* f32d[6] = {L.R2, L.A2, L.E32, L.dA2p, L.dR3, L.dR3};
  float aux1 = 0.0f, aux2 = 0.0f;
  auto edge_fetch = [&](int r0) {
    rg[0] = dw_fetch<64>(a.R1, r0, erows, tid);
    rg[1] = dw_fetch<64>(a.A1, r0, erows, tid);
    rg[2] = dw_fetch<64>(g.dR1p, r0, erows, tid);
    rg[3] = dw_fetch<64>(g.dA1p, r0, erows, tid);
#pragma unroll
    for (int p = 0; p < 3; ++p) rg[4 + p] = dw_fetch<32>(e32[2 * p + half], r0, erows, t7);
    if (tid < 16) {
      aux1 = (r0 + tid < erows) ? a.DIST[r0 + tid] : 0.0f;
      aux2 = (r0 + tid < erows) ? g.dq[r0 + tid] : 0.0f;
    }
  };
  auto edge_drop = [&]() {
    dw_drop<64>(L.R1, LDC, 0, tid, rg[0]);
    dw_drop<64>(L.A1, LDC, 0, tid, rg[1]);
    dw_drop<64>(L.dR1p, LDC, 0, tid, rg[2]);
    dw_drop<64>(L.dA1p, LDC, 0, tid, rg[3]);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      if (!(p == 2 && half == 1)) dw_drop<32>(f32d[2 * p + half], LDN, 0, t7, rg[4 + p]);
    if (tid < 16) {
      L.AUXE[tid * 16 + 1] = aux1;
      L.AUXE[tid * 16 + 2] = aux2;
    }
  };
  edge_fetch(0);
  for (int r0 = 0; r0 < erows; r0 += 16) {
    edge_drop();
    WG_SYNC();
    if (r0 + 16 < erows) edge_fetch(r0 + 16);
    dW_layer<32, 64, SL_R1>(acc, L.E32, LDN, L.R1, LDC, 1, wv);
    dW_layer<32, 64, SL_A1>(acc, L.dA2p, LDN, L.A1, LDC, 1, wv);
    dW_layer<32, 32, SL_R2>(acc, L.dR3, LDN, L.R2, LDN, 1, wv);
    vec_layer<VT_R0, 4>(vacc, L.dR1p, LDC, L.AUXE, 1, wv);
    vec_layer<VT_A0, 4>(vacc, L.dA1p, LDC, L.AUXE, 1, wv);
    vec_layer<VT_R1, 2>(vacc, L.E32, LDN, L.AUXE, 1, wv);
    vec_layer<VT_A1, 2>(vacc, L.dA2p, LDN, L.AUXE, 1, wv);
    vec_layer<VT_R2, 2>(vacc, L.dR3, LDN, L.AUXE, 1, wv);
    vec_layer<VT_WA2, 2>(vacc, L.A2, LDN, L.AUXE, 1, wv);
    vec_layer<VT_BA2, 1>(vacc, L.AUXE, 16, L.AUXE, 1, wv);
    WG_SYNC();
  }
  gnn_store_grads(acc, vacc, gpart + (size_t)blockIdx.x * kGnnGrads);
}

}  // namespace stove
