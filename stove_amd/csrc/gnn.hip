// Relational GNN dynamics core + the T-serial inference recursion, gfx950.
//
// Replaces the reference's
//   Dynamics.forward / core      (model/video_prediction/dynamics.py:181-265)
//   Dynamics.constrain_z_dyn     (dynamics.py:147-179)
//   the loop body of stove_forward + Stove.full_state   (model/video_prediction/stove.py:696-713, 103-170)
//   the loop body of Stove.rollout                      (stove.py:823-846)
//
// The recursion z[t] <- f(z[t-1], z_sup[t], eps[t]) is serial in t but independent across
// sequences, so ONE persistent workgroup owns G = floor(16/N) sequences for the whole time loop:
// no grid synchronisation, one launch for all T steps (the reference issues ~60 tiny ATen
// launches per step).  All dense layers run on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32, bit-for-bit an fma chain), 16 node rows (G*N padded) or
// 16-row edge tiles (G*N*N) per MFMA tile; activations live in LDS, weights stream from L2.
// The 65-wide edge input [s_i | s_j | d_ij] is never built: W0 [s_i|s_j|d] = W0a s_i + W0b s_j
// + w_d d, so the first edge layer is one per-node GEMM (32 -> 256) plus an elementwise gather.
// The backward recomputes the step's forward in LDS, then back-propagates; weight gradients
// accumulate in MFMA accumulators (22 tiles of 16x16 per wave) across ALL time steps and are
// written once per workgroup, then reduced in a fixed order (bitwise reproducible).
#include "common.h"

namespace stove {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// global store/load of the wave (vmcnt(0)); inside the time loop that would put the latency of the
// streaming output / activation stores on the critical path of every stage.  Nothing in these kernels
// hands global data from one wave to another inside a launch, so LDS ordering is all that is needed.
#define WG_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- parameter image (floats) ----------------------------------------------------------------
// P = [ W image | WT image | VEC ];  W: row-major [out][K];  WT: the transposes [K][out].
constexpr int W_ENC = 0, W_S0 = 1024, W_S1 = 2048, W_EF = 3072, W_R1 = 11264, W_A1 = 13312, W_R2 = 15360,
              W_F0 = 16384, W_F1 = 17408, W_F2 = 18432, W_O0 = 19456, W_O1 = 21504, W_END = 22528;
constexpr int V_ENC = 0, V_S0 = 32, V_S1 = 64, V_BR0 = 96, V_WDR = 160, V_BA0 = 224, V_WDA = 288, V_BR1 = 352,
              V_BA1 = 384, V_BR2 = 416, V_WA2 = 448, V_BA2 = 480, V_F0 = 512, V_F1 = 544, V_F2 = 576, V_O0 = 608,
              V_O1 = 640, V_END = 672;
// forward image: [W | W^T | vectors | W packed | W^T packed].  The two packed sections hold every layer of W / W^T once more in the
// [K/4][OUT][4] order the small-graph kernels keep in LDS (gnn_small.hip: sm_repack), so that their 256 workgroups fill LDS with
// straight float4 copies instead of each redoing the scattered repack of 90 KB (round 3: ~10 us off every launch of the recursion).
constexpr int P_WPACK = 2 * W_END + V_END, P_WTPACK = P_WPACK + W_END;
constexpr int kGnnParams = 2 * W_END + V_END + 2 * W_END;
constexpr int kGnnGrads = W_END + V_END;           // gradient image (W layout + VEC)

constexpr int LDN = 36, LDC = 68, LDP = 260, NEMAX = 80;

struct GnnShape {
  int N, G, NR, NE, ME;   // objects, sequences per workgroup, node rows, edge rows, edge tiles
  int sin_dim, lim_enc, elu;
  long long* stamps;      // debug: cycle stamps per stage (block 0, thread 0), normally null
};
__device__ __forceinline__ void gnn_stamp(const GnnShape& sh, int k) {
  if (sh.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) sh.stamps[k] = (long long)__builtin_readcyclecounter();
}

struct GnnLds {
  float *SIN, *H1, *SD, *PRED, *F1, *F2, *O1, *RES, *DA, *DB, *DC;   // [16][LDN]
  float *CAT, *DCAT;                                                  // [16][LDC]  CAT = [F3 | S]
  float* P;                                                           // [16][LDP]
  float *R1, *A1;                                                     // [NEMAX][LDC]
  float *R2, *A2, *R3, *E32;                                          // [NEMAX][LDN]
  float *ATT, *DIST, *DATT;                                           // [NEMAX]
  float* DDIST;                                                       // [16][2]
  float* PC;                                                          // [16][2] position carry of the time loop
  float* V;                                                           // [V_END] bias / vector parameters (copy of the VEC image)
  float* AUXN;                                                        // [16][16]    col 0 = 1                        (bias grads ride the MFMAs)
  float* AUXE;                                                        // [NEMAX][16] col 0 = 1, col 1 = dist_e, col 2 = dq_e
  int *EI, *EJ;                                                       // [NEMAX] node rows of edge e = (g, i, j); -1 for padding
  int *NG, *NI;                                                       // [16] first node row of the sequence of node r; object index of r
  float* X;                                                           // [16][40] epilogue scratch
};
constexpr int kGnnLdsFloats = 11 * 16 * LDN + 2 * 16 * LDC + 16 * LDP + 2 * NEMAX * LDC + 4 * NEMAX * LDN + 3 * NEMAX + 32 + 32 + V_END + 16 * 16 + NEMAX * 16 + 2 * NEMAX + 32 + 16 * 40;

__device__ __forceinline__ GnnLds carve(float* base) {
  GnnLds L;
  float* p = base;
  auto take = [&](int n) { float* q = p; p += n; return q; };
  L.SIN = take(16 * LDN); L.H1 = take(16 * LDN); L.SD = take(16 * LDN); L.PRED = take(16 * LDN);
  L.F1 = take(16 * LDN); L.F2 = take(16 * LDN); L.O1 = take(16 * LDN); L.RES = take(16 * LDN);
  L.DA = take(16 * LDN); L.DB = take(16 * LDN); L.DC = take(16 * LDN);
  L.CAT = take(16 * LDC); L.DCAT = take(16 * LDC);
  L.P = take(16 * LDP);
  L.R1 = take(NEMAX * LDC); L.A1 = take(NEMAX * LDC);
  L.R2 = take(NEMAX * LDN); L.A2 = take(NEMAX * LDN); L.R3 = take(NEMAX * LDN); L.E32 = take(NEMAX * LDN);
  L.ATT = take(NEMAX); L.DIST = take(NEMAX); L.DATT = take(NEMAX);
  L.DDIST = take(32);
  L.PC = take(32);
  L.V = take(V_END);
  L.AUXN = take(16 * 16);
  L.AUXE = take(NEMAX * 16);
  L.EI = reinterpret_cast<int*>(take(NEMAX));
  L.EJ = reinterpret_cast<int*>(take(NEMAX));
  L.NG = reinterpret_cast<int*>(take(16));
  L.NI = reinterpret_cast<int*>(take(16));
  L.X = take(16 * 40);
  return L;
}

// tanh through one v_exp_f32 and one v_rcp_f32 (absolute error ~2e-7); ocml's tanhf costs several hundred
// cycles of dependent latency per stage of the T-serial chain.
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
__device__ __forceinline__ float act_phi(float x, int elu) { return x > 0.0f ? x : (elu ? expm1f(x) : 0.01f * x); }
__device__ __forceinline__ float dphi_from_out(float y, int elu) { return y > 0.0f ? 1.0f : (elu ? y + 1.0f : 0.01f); }

// ---- MFMA tiles --------------------------------------------------------------------------------
// C(16x16) = A[16 x K] * B^T, A rows in LDS (lda floats), B rows = output columns, K contiguous.
// The K order inside a 16-block is permuted identically on both operands (lane group kq feeds
// k = 16 kb + 4 kq + m to MFMA m), so each lane reads one float4 per operand per 4 MFMAs.
template <int K>
__device__ __forceinline__ f32x4 tile_AB(const float* A, int lda, const float* __restrict__ B, int ldb) {
  const int l = lane_id(), i = l & 15, kq = l >> 4;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int kb = 0; kb < K / 16; ++kb) {
    const float4 a = *reinterpret_cast<const float4*>(A + i * lda + kb * 16 + 4 * kq);
    const float4 b = *reinterpret_cast<const float4*>(B + i * ldb + kb * 16 + 4 * kq);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  }
  return acc;
}
// The same tile with the weight fragment fetched ahead of time: wfrag_load is issued one stage early (the
// weights do not depend on anything), so its L2 latency is off the critical path of the stage that uses it.
template <int K>
struct WFrag {
  float4 b[K / 16];
};
template <int K>
__device__ __forceinline__ WFrag<K> wfrag_load(const float* __restrict__ B, int ldb) {
  const int l = lane_id(), i = l & 15, kq = l >> 4;
  WFrag<K> f;
#pragma unroll
  for (int kb = 0; kb < K / 16; ++kb) f.b[kb] = *reinterpret_cast<const float4*>(B + i * ldb + kb * 16 + 4 * kq);
  return f;
}
template <int K>
__device__ __forceinline__ f32x4 tile_AW(const float* A, int lda, const WFrag<K>& w) {
  const int l = lane_id(), i = l & 15, kq = l >> 4;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int kb = 0; kb < K / 16; ++kb) {
    const float4 a = *reinterpret_cast<const float4*>(A + i * lda + kb * 16 + 4 * kq);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.b[kb].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.b[kb].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.b[kb].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.b[kb].w, acc, 0, 0, 0);
  }
  return acc;
}
// acc[o][i] += sum_rows dO[row][o] * In[row][i]   (both operands row-strided in LDS)
__device__ __forceinline__ f32x4 tile_dW(const float* dO, int ldo, const float* In, int ldi, int row_tiles, f32x4 acc) {
  const int l = lane_id(), i = l & 15, kq = l >> 4;
  for (int kb = 0; kb < row_tiles; ++kb) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int r = kb * 16 + 4 * kq + m;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dO[r * ldo + i], In[r * ldi + i], acc, 0, 0, 0);
    }
  }
  return acc;
}
// element (row, col) of an accumulator tile held by this lane: row = m0 + 4*(lane>>4) + reg, col = n0 + (lane&15)
template <class F>
__device__ __forceinline__ void tile_each(f32x4 acc, int m0, int n0, F f) {
  const int l = lane_id();
  const int col = n0 + (l & 15), r0 = m0 + 4 * (l >> 4);
  f(r0, col, acc[0]);
  f(r0 + 1, col, acc[1]);
  f(r0 + 2, col, acc[2]);
  f(r0 + 3, col, acc[3]);
}

// one-time per-kernel setup: vector parameters and index tables into LDS (call after lds_zero + barrier)
__device__ __forceinline__ void gnn_setup(const GnnLds& L, const GnnShape& sh, const float* __restrict__ Vg) {
  const int tid = threadIdx.x;
  for (int i = tid; i < V_END; i += blockDim.x) L.V[i] = Vg[i];
  const int NN = sh.N * sh.N;
  if (tid < NEMAX) {
    int ei = -1, ej = -1;
    if (tid < sh.NE) {
      const int g = tid / NN, ij = tid % NN;
      ei = g * sh.N + ij / sh.N;
      ej = g * sh.N + ij % sh.N;
    }
    L.EI[tid] = ei;
    L.EJ[tid] = ej;
    L.AUXE[tid * 16] = tid < sh.NE ? 1.0f : 0.0f;
  }
  if (tid < 16) {
    L.NG[tid] = (tid / sh.N) * sh.N;
    L.NI[tid] = tid % sh.N;
    L.AUXN[tid * 16] = 1.0f;
  }
}

// Weight fragments of the first two stages, which every step of the time loop starts with: loaded ONCE per
// kernel and kept in registers (48 VGPRs), so the step never waits on L2 before its first MFMA.  All later
// stages fetch their fragments one or two stages ahead (WFrag prefetch), for the same reason.
struct FwdW {
  WFrag<32> enc;          // encoder tile (wave & 1)
  float4 ef0[4], ef1[4];  // edge-first column tiles wave + 4 q
  float4 s00, s01;        // self.0 tile (wave & 1)
};
__device__ __forceinline__ FwdW gnn_fwdw_load(const float* __restrict__ Wf) {
  const int wv = wave_id(), i = lane_id() & 15, kq = lane_id() >> 4;
  FwdW f;
  f.enc = wfrag_load<32>(Wf + W_ENC + (wv & 1) * 16 * 32, 32);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float* w = Wf + W_EF + (wv + 4 * q) * 16 * 32 + i * 32 + 4 * kq;
    f.ef0[q] = *reinterpret_cast<const float4*>(w);
    f.ef1[q] = *reinterpret_cast<const float4*>(w + 16);
  }
  const float* w = Wf + W_S0 + (wv & 1) * 16 * 32 + i * 32 + 4 * kq;
  f.s00 = *reinterpret_cast<const float4*>(w);
  f.s01 = *reinterpret_cast<const float4*>(w + 16);
  return f;
}

// =================================================================================================
// forward of one GNN step; input L.SIN (rows < NR, cols < sin_dim, rest zero), output L.RES, L.PRED
// =================================================================================================
__device__ __forceinline__ void gnn_forward(const GnnLds& L, const GnnShape& sh, const float* __restrict__ Wf, const FwdW& fw) {
  const int wv = wave_id();
  const int tid = threadIdx.x;
  const int w01 = wv & 1;
  const float* V = L.V;
  float* S = L.CAT + 32;   // S lives in CAT[:, 32:64]
  gnn_stamp(sh, 0);
  // 1. state encoder; raw positions (first lim_enc dims) are kept for the distances (dynamics.py:250)
  if (wv < 2) {
    const f32x4 acc = tile_AW<32>(L.SIN, LDN, fw.enc);
    tile_each(acc, 0, wv * 16, [&](int r, int c, float v) {
      S[r * LDC + c] = (c < sh.lim_enc) ? L.SIN[r * LDN + c] : v + V[V_ENC + c];
    });
  }
  WG_SYNC();
  gnn_stamp(sh, 1);
  // fragments of stages 3 and 4, in flight while stage 2 computes
  const WFrag<32> ws1 = wfrag_load<32>(Wf + W_S1 + w01 * 16 * 32, 32);
  const WFrag<64> w4 = wfrag_load<64>(Wf + (w01 ? W_A1 : W_R1) + ((wv >> 1) & 1) * 16 * 64, 64);
  // 2. self-dynamics layer 0 and the factorised first edge layer (rel_i | rel_j | att_i | att_j)
  {
    // every tile of this stage multiplies the same 16 x 32 activation block S: read its fragment once,
    // issue all weight loads of the wave's tiles up front, then run the independent MFMA chains interleaved
    // (wave w: edge-first column tiles w, w+4, w+8, w+12; waves 0/1 additionally self.0 tile w)
    const int i = lane_id() & 15, kq = lane_id() >> 4;
    const float4 a0 = *reinterpret_cast<const float4*>(S + i * LDC + 4 * kq);
    const float4 a1 = *reinterpret_cast<const float4*>(S + i * LDC + 16 + 4 * kq);
    float4 b0[5], b1[5];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      b0[q] = fw.ef0[q];
      b1[q] = fw.ef1[q];
    }
    b0[4] = fw.s00;
    b1[4] = fw.s01;
    f32x4 c[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) c[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#define STOVE_MMA_STEP(AV, BV)                                                          \
    _Pragma("unroll") for (int q = 0; q < 5; ++q)                                       \
      if (q < 4 || wv < 2) c[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV, BV[q], c[q], 0, 0, 0);
    float bx[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) bx[q] = b0[q].x;
    STOVE_MMA_STEP(a0.x, bx)
#pragma unroll
    for (int q = 0; q < 5; ++q) bx[q] = b0[q].y;
    STOVE_MMA_STEP(a0.y, bx)
#pragma unroll
    for (int q = 0; q < 5; ++q) bx[q] = b0[q].z;
    STOVE_MMA_STEP(a0.z, bx)
#pragma unroll
    for (int q = 0; q < 5; ++q) bx[q] = b0[q].w;
    STOVE_MMA_STEP(a0.w, bx)
#pragma unroll
    for (int q = 0; q < 5; ++q) bx[q] = b1[q].x;
    STOVE_MMA_STEP(a1.x, bx)
#pragma unroll
    for (int q = 0; q < 5; ++q) bx[q] = b1[q].y;
    STOVE_MMA_STEP(a1.y, bx)
#pragma unroll
    for (int q = 0; q < 5; ++q) bx[q] = b1[q].z;
    STOVE_MMA_STEP(a1.z, bx)
#pragma unroll
    for (int q = 0; q < 5; ++q) bx[q] = b1[q].w;
    STOVE_MMA_STEP(a1.w, bx)
#undef STOVE_MMA_STEP
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = wv + 4 * q;
      tile_each(c[q], 0, n * 16, [&](int r, int cc, float v) { L.P[r * LDP + cc] = v; });
    }
    if (wv < 2) tile_each(c[4], 0, wv * 16, [&](int r, int cc, float v) { L.H1[r * LDN + cc] = act_phi(v + V[V_S0 + cc], sh.elu); });
  }
  if (tid >= 192 && tid - 192 < sh.ME * 16) {          // squared distances (wave 3 has the fewest tiles)
    const int e = tid - 192;
    float d = 0.0f;
    if (L.EI[e] >= 0) {
      const int ni = L.EI[e], nj = L.EJ[e];
      const float dx = S[ni * LDC] - S[nj * LDC], dy = S[ni * LDC + 1] - S[nj * LDC + 1];
      d = dx * dx + dy * dy;
    }
    L.DIST[e] = d;
    L.AUXE[e * 16 + 1] = d;
  }
  if (tid >= 192 && sh.ME * 16 > 64 && tid - 192 + 64 < sh.ME * 16) {
    const int e = tid - 192 + 64;
    float d = 0.0f;
    if (L.EI[e] >= 0) {
      const int ni = L.EI[e], nj = L.EJ[e];
      const float dx = S[ni * LDC] - S[nj * LDC], dy = S[ni * LDC + 1] - S[nj * LDC + 1];
      d = dx * dx + dy * dy;
    }
    L.DIST[e] = d;
    L.AUXE[e * 16 + 1] = d;
  }
  WG_SYNC();
  gnn_stamp(sh, 2);
  // 3. edge pre-activations (gather) + self-dynamics layer 1
  const WFrag<32> w5 = wfrag_load<32>(Wf + W_R2 + w01 * 16 * 32, 32);
  {
    for (int idx = tid; idx < sh.ME * 16 * 64; idx += blockDim.x) {
      const int e = idx >> 6, c = idx & 63;
      float r1 = 0.0f, a1 = 0.0f;
      const int ni = L.EI[e];
      if (ni >= 0) {
        const int nj = L.EJ[e];
        const float d = L.DIST[e];
        r1 = act_phi(L.P[ni * LDP + c] + L.P[nj * LDP + 64 + c] + V[V_WDR + c] * d + V[V_BR0 + c], sh.elu);
        a1 = act_phi(L.P[ni * LDP + 128 + c] + L.P[nj * LDP + 192 + c] + V[V_WDA + c] * d + V[V_BA0 + c], sh.elu);
      }
      L.R1[e * LDC + c] = r1;
      L.A1[e * LDC + c] = a1;
    }
    if (wv < 2) {
      const f32x4 acc = tile_AW<32>(L.H1, LDN, ws1);
      tile_each(acc, 0, wv * 16, [&](int r, int c, float v) { L.SD[r * LDN + c] = v + V[V_S1 + c] + L.H1[r * LDN + c]; });
    }
  }
  WG_SYNC();
  gnn_stamp(sh, 3);
  // 4. second edge layers (64 -> 32), relation and attention: wave w owns (column tile (w>>1)&1, which = w&1)
  WFrag<32> wf0 = wfrag_load<32>(Wf + W_F0 + w01 * 16 * 32, 32);      // stage 7's fragment
  for (int t = wv; t < sh.ME * 4; t += 4) {
    const int m = t >> 2, n = (t >> 1) & 1, which = t & 1;
    if (which == 0) {
      const f32x4 acc = tile_AW<64>(L.R1 + m * 16 * LDC, LDC, w4);
      tile_each(acc, m * 16, n * 16, [&](int r, int c, float v) { L.R2[r * LDN + c] = act_phi(v + V[V_BR1 + c], sh.elu); });
    } else {
      const f32x4 acc = tile_AW<64>(L.A1 + m * 16 * LDC, LDC, w4);
      tile_each(acc, m * 16, n * 16, [&](int r, int c, float v) { L.A2[r * LDN + c] = act_phi(v + V[V_BA1 + c], sh.elu); });
    }
  }
  WG_SYNC();
  gnn_stamp(sh, 4);
  // 5. third edge layers: relation 32 -> 32 (+skip), attention 32 -> 1 -> exp
  if (tid < sh.ME * 16) {
    float q = V[V_BA2];
    for (int c = 0; c < 32; ++c) q = fmaf(L.A2[tid * LDN + c], V[V_WA2 + c], q);
    L.ATT[tid] = (tid < sh.NE) ? __expf(q) : 0.0f;
  }
  for (int t = wv; t < sh.ME * 2; t += 4) {
    const int m = t >> 1, n = t & 1;
    const f32x4 acc = tile_AW<32>(L.R2 + m * 16 * LDN, LDN, w5);       // n = t & 1 = wave & 1
    tile_each(acc, m * 16, n * 16, [&](int r, int c, float v) { L.R3[r * LDN + c] = v + V[V_BR2 + c] + L.R2[r * LDN + c]; });
  }
  WG_SYNC();
  gnn_stamp(sh, 5);
  // 6. masked, attention-weighted aggregation over the other objects
  for (int idx = tid; idx < 16 * 32; idx += blockDim.x) {
    const int r = idx >> 5, c = idx & 31;
    float v = 0.0f;
    if (r < sh.NR) {
      const int i = L.NI[r], e0 = r * sh.N;                  // edges (r -> j) are rows e0 .. e0+N-1
      v = L.SD[r * LDN + c];
      for (int j = 0; j < sh.N; ++j)
        if (j != i) v = fmaf(L.R3[(e0 + j) * LDN + c], L.ATT[e0 + j], v);
    }
    L.PRED[r * LDN + c] = v;
  }
  WG_SYNC();
  gnn_stamp(sh, 6);
  // 7-11. affector and output MLPs: a chain of five small layers on waves 0/1; each stage fetches the NEXT
  // layer's weight fragment before it starts computing.
  if (wv < 2) {
    const WFrag<32> wf1 = wfrag_load<32>(Wf + W_F1 + w01 * 16 * 32, 32);
    const f32x4 acc = tile_AW<32>(L.PRED, LDN, wf0);
    tile_each(acc, 0, wv * 16, [&](int r, int c, float v) { L.F1[r * LDN + c] = fast_tanh(v + V[V_F0 + c]); });
    wf0 = wf1;
  }
  WG_SYNC();
  gnn_stamp(sh, 7);
  WFrag<64> wo0;
  if (wv < 2) {
    const WFrag<32> wf2 = wfrag_load<32>(Wf + W_F2 + w01 * 16 * 32, 32);
    const f32x4 acc = tile_AW<32>(L.F1, LDN, wf0);
    tile_each(acc, 0, wv * 16, [&](int r, int c, float v) { L.F2[r * LDN + c] = fast_tanh(v + V[V_F1 + c]) + L.F1[r * LDN + c]; });
    wf0 = wf2;
  }
  WG_SYNC();
  gnn_stamp(sh, 8);
  if (wv < 2) {
    wo0 = wfrag_load<64>(Wf + W_O0 + w01 * 16 * 64, 64);
    const f32x4 acc = tile_AW<32>(L.F2, LDN, wf0);
    tile_each(acc, 0, wv * 16, [&](int r, int c, float v) { L.CAT[r * LDC + c] = v + V[V_F2 + c]; });
  }
  WG_SYNC();
  gnn_stamp(sh, 9);
  if (wv < 2) {
    wf0 = wfrag_load<32>(Wf + W_O1 + w01 * 16 * 32, 32);
    const f32x4 acc = tile_AW<64>(L.CAT, LDC, wo0);
    tile_each(acc, 0, wv * 16, [&](int r, int c, float v) { L.O1[r * LDN + c] = fast_tanh(v + V[V_O0 + c]); });
  }
  WG_SYNC();
  gnn_stamp(sh, 10);
  if (wv < 2) {
    const f32x4 acc = tile_AW<32>(L.O1, LDN, wf0);
    tile_each(acc, 0, wv * 16, [&](int r, int c, float v) { L.RES[r * LDN + c] = v + V[V_O1 + c] + L.O1[r * LDN + c]; });
  }
  WG_SYNC();
  gnn_stamp(sh, 11);
}

// =================================================================================================
// backward of one GNN step.  Requires the buffers left by gnn_forward of the same step.
// in : L.DA = dL/dRES (rows >= NR zero);  dpred_up (global, may be null) = dL/dPRED from outside
// out: L.DA = dL/dSIN;  weight gradients accumulate into acc[22] (MFMA tiles) and L.DV.
// =================================================================================================
template <int OUT, int IN, int SLOT0>
__device__ __forceinline__ void dW_layer(f32x4* acc, const float* dO, int ldo, const float* In, int ldi, int row_tiles, int wv) {
  constexpr int NT = (OUT / 16) * (IN / 16);
  static_assert(NT % 4 == 0, "tiles per layer must be a multiple of the wave count");
  if (row_tiles == 1) {
    // one 16-row block (every node layer; edge layers of small graphs): fetch the operands of ALL the wave's tiles
    // first, then issue the MFMAs step-major, so consecutive MFMAs hit different accumulators and the LDS latency is
    // paid once instead of once per tile (the tile-by-tile order made dW of the 256 x 32 edge-first layer cost 2.4k
    // cycles for 1k cycles of MFMA)
    const int l = lane_id(), i = l & 15, kq = l >> 4;
    float a[NT / 4][4], b[NT / 4][4];
#pragma unroll
    for (int k = 0; k < NT / 4; ++k) {
      const int t = wv + 4 * k;
      const int ot = t / (IN / 16), it = t % (IN / 16);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        a[k][m] = dO[(4 * kq + m) * ldo + ot * 16 + i];
        b[k][m] = In[(4 * kq + m) * ldi + it * 16 + i];
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int k = 0; k < NT / 4; ++k) acc[SLOT0 + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][m], b[k][m], acc[SLOT0 + k], 0, 0, 0);
    return;
  }
#pragma unroll
  for (int k = 0; k < NT / 4; ++k) {
    const int t = wv + 4 * k;
    const int ot = t / (IN / 16), it = t % (IN / 16);
    acc[SLOT0 + k] = tile_dW(dO + ot * 16, ldo, In + it * 16, ldi, row_tiles, acc[SLOT0 + k]);
  }
}
template <int OUT, int IN, int SLOT0>
__device__ __forceinline__ void dW_store(const f32x4* acc, float* __restrict__ img, int wv) {
  constexpr int NT = (OUT / 16) * (IN / 16);
#pragma unroll
  for (int k = 0; k < NT / 4; ++k) {
    const int t = wv + 4 * k;
    const int ot = t / (IN / 16), it = t % (IN / 16);
    tile_each(acc[SLOT0 + k], ot * 16, it * 16, [&](int r, int c, float v) { img[r * IN + c] = v; });
  }
}
constexpr int SL_ENC = 0, SL_S0 = 1, SL_S1 = 2, SL_EF = 3, SL_R1 = 11, SL_A1 = 13, SL_R2 = 15, SL_F0 = 16, SL_F1 = 17,
              SL_F2 = 18, SL_O0 = 19, SL_O1 = 21, SL_END = 22;

// Bias / vector gradients ride the matrix cores too: with an auxiliary operand whose column 0 is 1
// (column 1 the edge distance, column 2 dq), tile (o, j) of dO^T AUX holds sum_rows dO[row][o] * AUX[row][j].
// Vector tiles are numbered globally; tile t belongs to wave t & 3 and accumulator slot t >> 2.
constexpr int VT_ENC = 0, VT_S0 = 2, VT_S1 = 4, VT_F0 = 6, VT_F1 = 8, VT_F2 = 10, VT_O0 = 12, VT_O1 = 14,
              VT_R0 = 16, VT_A0 = 20, VT_R1 = 24, VT_A1 = 26, VT_R2 = 28, VT_WA2 = 30, VT_BA2 = 32, VT_END = 33;
constexpr int VSLOTS = (VT_END + 3) / 4;

template <int T0, int NT>
__device__ __forceinline__ void vec_layer(f32x4* vacc, const float* dO, int ldo, const float* aux, int row_tiles, int wv) {
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    constexpr int dummy = 0;
    (void)dummy;
    if (((T0 + k) & 3) == wv) vacc[(T0 + k) >> 2] = tile_dW(dO + k * 16, ldo, aux, 16, row_tiles, vacc[(T0 + k) >> 2]);
  }
}
// column `col` of vector tile t -> out[k*16 + o]
template <int T0, int NT>
__device__ __forceinline__ void vec_store(const f32x4* vacc, float* __restrict__ out, int col, int wv) {
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if (((T0 + k) & 3) == wv) {
      tile_each(vacc[(T0 + k) >> 2], k * 16, 0, [&](int r, int c, float v) {
        if (c == col) out[r] = v;
      });
    }
  }
}

// `o1t`: the W^T fragment of the first backward stage (out.1, tile wave & 1), loaded once per kernel by the caller.
// Every other stage's fragment is fetched one stage ahead, so no stage waits on L2 before its first MFMA.
__device__ __forceinline__ void gnn_backward(const GnnLds& L, const GnnShape& sh, const float* __restrict__ WT, const WFrag<32>& o1t,
                             f32x4* acc, f32x4* vacc, const float* dpred_up /* global or null */, size_t dpred_seq_stride) {
  const int wv = wave_id();
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int w01 = wv & 1;
  const float* V = L.V;
  float* S = L.CAT + 32;
  gnn_stamp(sh, 20);
  // b1. out.1:  RES = O1 W^T + b + O1
  const WFrag<32> o0t = wfrag_load<32>(WT + W_O0 + wv * 16 * 32, 32);
  dW_layer<32, 32, SL_O1>(acc, L.DA, LDN, L.O1, LDN, 1, wv);
  vec_layer<VT_O1, 2>(vacc, L.DA, LDN, L.AUXN, 1, wv);
  if (wv < 2) {
    const f32x4 t = tile_AW<32>(L.DA, LDN, o1t);
    tile_each(t, 0, wv * 16, [&](int r, int c, float v) {
      const float o = L.O1[r * LDN + c];
      L.DB[r * LDN + c] = (v + L.DA[r * LDN + c]) * (1.0f - o * o);       // d pre-tanh of out.0
    });
  }
  WG_SYNC();
  gnn_stamp(sh, 21);
  // b2. out.0 on CAT = [F3 | S]
  const WFrag<32> f2t = wfrag_load<32>(WT + W_F2 + w01 * 16 * 32, 32);
  dW_layer<32, 64, SL_O0>(acc, L.DB, LDN, L.CAT, LDC, 1, wv);
  vec_layer<VT_O0, 2>(vacc, L.DB, LDN, L.AUXN, 1, wv);
  {
    const f32x4 t = tile_AW<32>(L.DB, LDN, o0t);
    tile_each(t, 0, wv * 16, [&](int r, int c, float v) { L.DCAT[r * LDC + c] = v; });
  }
  WG_SYNC();
  gnn_stamp(sh, 22);
  // b3. affector.2:  F3 = F2 W^T + b      (dF3 = DCAT[:, :32])
  const WFrag<32> f1t = wfrag_load<32>(WT + W_F1 + w01 * 16 * 32, 32);
  dW_layer<32, 32, SL_F2>(acc, L.DCAT, LDC, L.F2, LDN, 1, wv);
  vec_layer<VT_F2, 2>(vacc, L.DCAT, LDC, L.AUXN, 1, wv);
  if (wv < 2) {
    const f32x4 t = tile_AW<32>(L.DCAT, LDC, f2t);
    tile_each(t, 0, wv * 16, [&](int r, int c, float v) {
      const float th = L.F2[r * LDN + c] - L.F1[r * LDN + c];              // tanh(u) of affector.1
      L.DA[r * LDN + c] = v;                                               // dF2 (skip path)
      L.DC[r * LDN + c] = v * (1.0f - th * th);                            // du
    });
  }
  WG_SYNC();
  gnn_stamp(sh, 23);
  // b4. affector.1:  F2 = tanh(F1 W^T + b) + F1
  const WFrag<32> f0t = wfrag_load<32>(WT + W_F0 + w01 * 16 * 32, 32);
  dW_layer<32, 32, SL_F1>(acc, L.DC, LDN, L.F1, LDN, 1, wv);
  vec_layer<VT_F1, 2>(vacc, L.DC, LDN, L.AUXN, 1, wv);
  if (wv < 2) {
    const f32x4 t = tile_AW<32>(L.DC, LDN, f1t);
    tile_each(t, 0, wv * 16, [&](int r, int c, float v) {
      const float f1 = L.F1[r * LDN + c];
      L.DB[r * LDN + c] = (v + L.DA[r * LDN + c]) * (1.0f - f1 * f1);      // d pre-tanh of affector.0
    });
  }
  WG_SYNC();
  gnn_stamp(sh, 24);
  // b5. affector.0:  F1 = tanh(PRED W^T + b)
  const WFrag<32> r2t = wfrag_load<32>(WT + W_R2 + w01 * 16 * 32, 32);
  dW_layer<32, 32, SL_F0>(acc, L.DB, LDN, L.PRED, LDN, 1, wv);
  vec_layer<VT_F0, 2>(vacc, L.DB, LDN, L.AUXN, 1, wv);
  if (wv < 2) {
    const f32x4 t = tile_AW<32>(L.DB, LDN, f0t);
    tile_each(t, 0, wv * 16, [&](int r, int c, float v) {
      float up = 0.0f;
      if (dpred_up != nullptr && r < sh.NR) up = dpred_up[(size_t)(r / sh.N) * dpred_seq_stride + (r % sh.N) * 32 + c];
      L.DC[r * LDN + c] = v + up;                                          // dPRED = dSD
    });
  }
  WG_SYNC();
  gnn_stamp(sh, 25);
  // b6a. d attention: one wave-half (32 lanes = 32 channels) per edge
  for (int e = wv * 2 + (lane >> 5); e < sh.ME * 16; e += 8) {
    const int ni = L.EI[e], c = lane & 31;
    float v = 0.0f;
    if (ni >= 0 && ni != L.EJ[e]) v = L.DC[ni * LDN + c] * L.R3[e * LDN + c];
    // sum over the 32 lanes of this half
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    if (c == 0) {
      const float dq = v * L.ATT[e];                                       // att = exp(q)
      L.DATT[e] = dq;
      L.AUXE[e * 16 + 2] = dq;
    }
  }
  WG_SYNC();
  gnn_stamp(sh, 26);
  // b6b. dR3 in place; attention output layer grads (A2 still holds the forward values)
  vec_layer<VT_WA2, 2>(vacc, L.A2, LDN, L.AUXE, sh.ME, wv);
  vec_layer<VT_BA2, 1>(vacc, L.AUXE, 16, L.AUXE, sh.ME, wv);
  for (int idx = tid; idx < sh.ME * 16 * 32; idx += blockDim.x) {
    const int e = idx >> 5, c = idx & 31;
    const int ni = L.EI[e];
    float v = 0.0f;
    if (ni >= 0 && ni != L.EJ[e]) v = L.DC[ni * LDN + c] * L.ATT[e];
    L.R3[e * LDN + c] = v;
  }
  WG_SYNC();
  gnn_stamp(sh, 27);
  // b7. rel.2:  R3 = R2 W^T + b + R2 ;  attention pre-activation grads in place in A2
  // b9's fragments: tile t = wave + 4k has which = wave & 1 and column tile (wave >> 1) + 2 (k & 1)
  const float* w9 = WT + (w01 ? W_A1 : W_R1) + (wv >> 1) * 16 * 32;
  const WFrag<32> r1ta = wfrag_load<32>(w9, 32);
  const WFrag<32> r1tb = wfrag_load<32>(w9 + 2 * 16 * 32, 32);
  dW_layer<32, 32, SL_R2>(acc, L.R3, LDN, L.R2, LDN, sh.ME, wv);
  vec_layer<VT_R2, 2>(vacc, L.R3, LDN, L.AUXE, sh.ME, wv);
  for (int t = wv; t < sh.ME * 2; t += 4) {
    const int m = t >> 1, n = t & 1;
    const f32x4 a = tile_AW<32>(L.R3 + m * 16 * LDN, LDN, r2t);            // n = wave & 1
    tile_each(a, m * 16, n * 16, [&](int r, int c, float v) {
      L.E32[r * LDN + c] = (v + L.R3[r * LDN + c]) * dphi_from_out(L.R2[r * LDN + c], sh.elu);
    });
  }
  for (int idx = tid; idx < sh.ME * 16 * 32; idx += blockDim.x) {
    const int e = idx >> 5, c = idx & 31;
    const float y = L.A2[e * LDN + c];
    L.A2[e * LDN + c] = L.DATT[e] * V[V_WA2 + c] * dphi_from_out(y, sh.elu);
  }
  WG_SYNC();
  gnn_stamp(sh, 28);
  // b8. rel.1 / att.1 weight grads (inputs R1 / A1 still intact)
  dW_layer<32, 64, SL_R1>(acc, L.E32, LDN, L.R1, LDC, sh.ME, wv);
  dW_layer<32, 64, SL_A1>(acc, L.A2, LDN, L.A1, LDC, sh.ME, wv);
  vec_layer<VT_R1, 2>(vacc, L.E32, LDN, L.AUXE, sh.ME, wv);
  vec_layer<VT_A1, 2>(vacc, L.A2, LDN, L.AUXE, sh.ME, wv);
  WG_SYNC();
  gnn_stamp(sh, 29);
  // b9. rel.1 / att.1 data grads, multiplied by phi'(first-layer output), in place in R1 / A1
  for (int t = wv; t < sh.ME * 8; t += 4) {
    const int m = t >> 3, n = (t >> 1) & 3, which = t & 1;
    const WFrag<32>& w = (n & 2) ? r1tb : r1ta;
    if (which == 0) {
      const f32x4 a = tile_AW<32>(L.E32 + m * 16 * LDN, LDN, w);
      tile_each(a, m * 16, n * 16, [&](int r, int c, float v) { L.R1[r * LDC + c] = v * dphi_from_out(L.R1[r * LDC + c], sh.elu); });
    } else {
      const f32x4 a = tile_AW<32>(L.A2 + m * 16 * LDN, LDN, w);
      tile_each(a, m * 16, n * 16, [&](int r, int c, float v) { L.A1[r * LDC + c] = v * dphi_from_out(L.A1[r * LDC + c], sh.elu); });
    }
  }
  WG_SYNC();
  gnn_stamp(sh, 30);
  // b10. first edge layer: scatter (as a gather) into dP, bias / distance-weight grads, d distance
  const WFrag<64> eft0 = wfrag_load<64>(WT + W_EF + wv * 64, 256);        // b11's fragments
  const WFrag<64> eft1 = wfrag_load<64>(WT + W_EF + 16 * 256 + wv * 64, 256);
  const WFrag<32> s1t = wfrag_load<32>(WT + W_S1 + w01 * 16 * 32, 32);
  vec_layer<VT_R0, 4>(vacc, L.R1, LDC, L.AUXE, sh.ME, wv);
  vec_layer<VT_A0, 4>(vacc, L.A1, LDC, L.AUXE, sh.ME, wv);
  {
    // dP[r][c] for the real node rows only (thread = column c), padded rows cleared with vector stores
    const int c = tid, blk = c >> 6, cc = c & 63;
    const float* src = (blk < 2) ? L.R1 : L.A1;
    for (int r = 0; r < sh.NR; ++r) {
      const int g0 = L.NG[r], i = L.NI[r];
      float v = 0.0f;
      if (blk & 1) {
        for (int j = 0; j < sh.N; ++j) v += src[((g0 + j) * sh.N + i) * LDC + cc];      // r as the second argument s_j
      } else {
        for (int j = 0; j < sh.N; ++j) v += src[(r * sh.N + j) * LDC + cc];             // r as the first argument s_i
      }
      L.P[r * LDP + c] = v;
    }
    for (int q = tid; q < (16 - sh.NR) * 64; q += blockDim.x)
      *reinterpret_cast<float4*>(L.P + (sh.NR + (q >> 6)) * LDP + (q & 63) * 4) = float4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  // dL/d dist_e: 16 lanes per edge, 4 channels each, row reduction
  for (int e0 = 0; e0 < sh.ME * 16; e0 += 16) {
    const int e = e0 + (tid >> 4), c4 = (tid & 15) * 4;
    const float4 r1 = *reinterpret_cast<const float4*>(L.R1 + e * LDC + c4), a1 = *reinterpret_cast<const float4*>(L.A1 + e * LDC + c4);
    const float4 wr = *reinterpret_cast<const float4*>(V + V_WDR + c4), wa = *reinterpret_cast<const float4*>(V + V_WDA + c4);
    float v = r1.x * wr.x + r1.y * wr.y + r1.z * wr.z + r1.w * wr.w + a1.x * wa.x + a1.y * wa.y + a1.z * wa.z + a1.w * wa.w;
    v = row_sum_lane15(v);
    if ((tid & 15) == 15) L.DATT[e] = v;
  }
  WG_SYNC();
  gnn_stamp(sh, 31);
  // b11. edge-first + self.1
  const WFrag<32> s0t = wfrag_load<32>(WT + W_S0 + w01 * 16 * 32, 32);
  dW_layer<256, 32, SL_EF>(acc, L.P, LDP, S, LDC, 1, wv);
  dW_layer<32, 32, SL_S1>(acc, L.DC, LDN, L.H1, LDN, 1, wv);
  vec_layer<VT_S1, 2>(vacc, L.DC, LDN, L.AUXN, 1, wv);
  {
    // dS from the edge layers: dP (16 x 256) Wef (256 x 32).  Each wave contracts one quarter of K for both
    // column tiles (two interleaved chains of 16 MFMAs instead of one chain of 64); b12 adds the 4 partials.
    float* part = (wv == 0) ? L.DA : (wv == 1) ? L.SD : (wv == 2) ? L.RES : L.O1;     // SD, RES, O1 are dead here
    const f32x4 t0 = tile_AW<64>(L.P + wv * 64, LDP, eft0);
    const f32x4 t1 = tile_AW<64>(L.P + wv * 64, LDP, eft1);
    tile_each(t0, 0, 0, [&](int r, int c, float v) { part[r * LDN + c] = v; });
    tile_each(t1, 0, 16, [&](int r, int c, float v) { part[r * LDN + c] = v; });
  }
  if (wv < 2) {
    const f32x4 t = tile_AW<32>(L.DC, LDN, s1t);
    tile_each(t, 0, wv * 16, [&](int r, int c, float v) {
      L.DB[r * LDN + c] = (v + L.DC[r * LDN + c]) * dphi_from_out(L.H1[r * LDN + c], sh.elu);   // d pre-act of self.0
    });
  }
  if (tid < 32) {
    const int r = tid >> 1, ax = tid & 1;
    float s = 0.0f;
    if (r < sh.NR) {
      const int g0 = L.NG[r], i = L.NI[r];
      for (int j = 0; j < sh.N; ++j) {
        const float diff = S[r * LDC + ax] - S[(g0 + j) * LDC + ax];
        s += 2.0f * diff * (L.DATT[r * sh.N + j] + L.DATT[(g0 + j) * sh.N + i]);
      }
    }
    L.DDIST[tid] = s;
  }
  WG_SYNC();
  gnn_stamp(sh, 32);
  // b12. self.0 ; total dS ; split into the encoder output part and the pass-through part
  const WFrag<32> enct = wfrag_load<32>(WT + W_ENC + w01 * 16 * 32, 32);
  dW_layer<32, 32, SL_S0>(acc, L.DB, LDN, S, LDC, 1, wv);
  vec_layer<VT_S0, 2>(vacc, L.DB, LDN, L.AUXN, 1, wv);
  if (wv < 2) {
    const f32x4 t = tile_AW<32>(L.DB, LDN, s0t);
    tile_each(t, 0, wv * 16, [&](int r, int c, float v) {
      float tot = v + ((L.DA[r * LDN + c] + L.SD[r * LDN + c]) + (L.RES[r * LDN + c] + L.O1[r * LDN + c])) + L.DCAT[r * LDC + 32 + c];
      if (c < 2) tot += L.DDIST[r * 2 + c];
      const bool raw = c < sh.lim_enc;
      L.DC[r * LDN + c] = raw ? 0.0f : tot;      // d encoder output
      L.F1[r * LDN + c] = raw ? tot : 0.0f;      // straight to SIN (F1 is dead by now)
    });
  }
  WG_SYNC();
  gnn_stamp(sh, 33);
  // b13. encoder
  dW_layer<32, 32, SL_ENC>(acc, L.DC, LDN, L.SIN, LDN, 1, wv);
  vec_layer<VT_ENC, 2>(vacc, L.DC, LDN, L.AUXN, 1, wv);
  if (wv < 2) {
    const f32x4 t = tile_AW<32>(L.DC, LDN, enct);
    tile_each(t, 0, wv * 16, [&](int r, int c, float v) { L.DA[r * LDN + c] = v + L.F1[r * LDN + c]; });
  }
  WG_SYNC();
  gnn_stamp(sh, 34);
}

__device__ __forceinline__ void gnn_store_grads(const f32x4* acc, const f32x4* vacc, float* __restrict__ gout) {
  const int wv = wave_id();
  dW_store<32, 32, SL_ENC>(acc, gout + W_ENC, wv);
  dW_store<32, 32, SL_S0>(acc, gout + W_S0, wv);
  dW_store<32, 32, SL_S1>(acc, gout + W_S1, wv);
  dW_store<256, 32, SL_EF>(acc, gout + W_EF, wv);
  dW_store<32, 64, SL_R1>(acc, gout + W_R1, wv);
  dW_store<32, 64, SL_A1>(acc, gout + W_A1, wv);
  dW_store<32, 32, SL_R2>(acc, gout + W_R2, wv);
  dW_store<32, 32, SL_F0>(acc, gout + W_F0, wv);
  dW_store<32, 32, SL_F1>(acc, gout + W_F1, wv);
  dW_store<32, 32, SL_F2>(acc, gout + W_F2, wv);
  dW_store<32, 64, SL_O0>(acc, gout + W_O0, wv);
  dW_store<32, 32, SL_O1>(acc, gout + W_O1, wv);
  float* gv = gout + W_END;
  vec_store<VT_ENC, 2>(vacc, gv + V_ENC, 0, wv);
  vec_store<VT_S0, 2>(vacc, gv + V_S0, 0, wv);
  vec_store<VT_S1, 2>(vacc, gv + V_S1, 0, wv);
  vec_store<VT_F0, 2>(vacc, gv + V_F0, 0, wv);
  vec_store<VT_F1, 2>(vacc, gv + V_F1, 0, wv);
  vec_store<VT_F2, 2>(vacc, gv + V_F2, 0, wv);
  vec_store<VT_O0, 2>(vacc, gv + V_O0, 0, wv);
  vec_store<VT_O1, 2>(vacc, gv + V_O1, 0, wv);
  vec_store<VT_R0, 4>(vacc, gv + V_BR0, 0, wv);
  vec_store<VT_R0, 4>(vacc, gv + V_WDR, 1, wv);
  vec_store<VT_A0, 4>(vacc, gv + V_BA0, 0, wv);
  vec_store<VT_A0, 4>(vacc, gv + V_WDA, 1, wv);
  vec_store<VT_R1, 2>(vacc, gv + V_BR1, 0, wv);
  vec_store<VT_A1, 2>(vacc, gv + V_BA1, 0, wv);
  vec_store<VT_R2, 2>(vacc, gv + V_BR2, 0, wv);
  vec_store<VT_WA2, 2>(vacc, gv + V_WA2, 2, wv);
  // ba2 = sum_e dq_e = element (row 0, col 2) of the AUXE^T AUXE tile; the rest of the 32-slot is padding
  if (((VT_BA2)&3) == wv) {
    tile_each(vacc[VT_BA2 >> 2], 0, 0, [&](int r, int c, float v) {
      if (c == 2 && r < 16) gv[V_BA2 + r] = (r == 0) ? v : 0.0f;
    });
  }
  if (wv == 1)
    for (int i = lane_id(); i < 16; i += 64) gv[V_BA2 + 16 + i] = 0.0f;
}

// sequences per workgroup: node rows <= 16 and edge rows <= NEMAX
__host__ __device__ inline int gnn_group_max(int N) {
  const int a = 16 / N, b = NEMAX / (N * N);
  const int g = a < b ? a : b;
  return g < 1 ? 1 : g;
}
// The recursion is latency-bound (stage count x per-stage latency), not throughput-bound: as long as
// there are CUs to spare, fewer sequences per workgroup means shorter elementwise loops and fewer
// tiles per stage, hence a shorter step.  Use the smallest group that still fits one workgroup per CU.
__host__ __device__ inline int gnn_group_for(int B, int N) {
  if (N <= 4) return 1;      // the small-graph time loop (gnn_small.hip) is one sequence per workgroup
  const int gmax = gnn_group_max(N);
  int g = (B + 255) / 256;
  if (g < 1) g = 1;
  return g > gmax ? gmax : g;
}

__device__ __forceinline__ GnnShape make_shape(int N, int G, int b0, int B, int sin_dim, int lim_enc, int elu) {
  GnnShape sh;
  sh.N = N;
  sh.G = (B - b0) < G ? (B - b0) : G;
  sh.NR = sh.G * N;
  sh.NE = sh.G * N * N;
  sh.ME = (sh.NE + 15) / 16;
  sh.sin_dim = sin_dim;
  sh.lim_enc = lim_enc;
  sh.elu = elu;
  sh.stamps = nullptr;
  return sh;
}

__device__ __forceinline__ void lds_zero(float* base, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) base[i] = 0.0f;
}

// =================================================================================================
// single step: Dynamics.forward(s) -> result (B,N,32), dynamic_pred (B,N,32)
// =================================================================================================
__global__ __launch_bounds__(256) void gnn_step_fwd_k(const float* __restrict__ sin, const float* __restrict__ P,
                                                      float* __restrict__ res, float* __restrict__ pred,
                                                      int B, int N, int G, int sin_dim, int lim_enc, int elu) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GnnLds L = carve(lds);
  const int b0 = blockIdx.x * G;
  const GnnShape sh = make_shape(N, G, b0, B, sin_dim, lim_enc, elu);
  lds_zero(lds, kGnnLdsFloats);
  WG_SYNC();
  gnn_setup(L, sh, P + 2 * W_END);
  WG_SYNC();
  for (int i = threadIdx.x; i < sh.NR * sin_dim; i += blockDim.x) {
    const int r = i / sin_dim, c = i % sin_dim;
    L.SIN[r * LDN + c] = sin[((size_t)b0 * N + r) * sin_dim + c];
  }
  const FwdW fw = gnn_fwdw_load(P);
  WG_SYNC();
  gnn_forward(L, sh, P, fw);
  for (int i = threadIdx.x; i < sh.NR * 32; i += blockDim.x) {
    const int r = i >> 5, c = i & 31;
    res[((size_t)b0 * N + r) * 32 + c] = L.RES[r * LDN + c];
    if (pred != nullptr) pred[((size_t)b0 * N + r) * 32 + c] = L.PRED[r * LDN + c];
  }
}

__global__ __launch_bounds__(256) void gnn_step_bwd_k(const float* __restrict__ sin, const float* __restrict__ P,
                                                      const float* __restrict__ dres, const float* __restrict__ dpred,
                                                      float* __restrict__ dsin, float* __restrict__ gpart,
                                                      int B, int N, int G, int sin_dim, int lim_enc, int elu, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GnnLds L = carve(lds);
  const int b0 = blockIdx.x * G;
  GnnShape sh = make_shape(N, G, b0, B, sin_dim, lim_enc, elu);
  lds_zero(lds, kGnnLdsFloats);
  WG_SYNC();
  gnn_setup(L, sh, P + 2 * W_END);
  WG_SYNC();
  for (int i = threadIdx.x; i < sh.NR * sin_dim; i += blockDim.x) {
    const int r = i / sin_dim, c = i % sin_dim;
    L.SIN[r * LDN + c] = sin[((size_t)b0 * N + r) * sin_dim + c];
  }
  f32x4 acc[SL_END], vacc[VSLOTS];
#pragma unroll
  for (int k = 0; k < SL_END; ++k) acc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < VSLOTS; ++k) vacc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const WFrag<32> o1t = wfrag_load<32>(P + W_END + W_O1 + (wave_id() & 1) * 16 * 32, 32);
  // debug stamps: time the THIRD pass (weights L2-warm, as every step of the time loop sees them); the extra
  // passes only happen with stamps != null (their weight gradients are then 3x, nobody reads them)
  const int reps = (stamps != nullptr) ? 3 : 1;
  for (int rep = 0; rep < reps; ++rep) {
    sh.stamps = (rep == reps - 1) ? stamps : nullptr;
    {
      const FwdW fw = gnn_fwdw_load(P);
      WG_SYNC();
      gnn_forward(L, sh, P, fw);
    }
    for (int i = threadIdx.x; i < 16 * 32; i += blockDim.x) {
      const int r = i >> 5, c = i & 31;
      L.DA[r * LDN + c] = (r < sh.NR) ? dres[((size_t)b0 * N + r) * 32 + c] : 0.0f;
    }
    WG_SYNC();
    gnn_backward(L, sh, P + W_END, o1t, acc, vacc, dpred != nullptr ? dpred + (size_t)b0 * N * 32 : nullptr, (size_t)N * 32);
    if (rep + 1 < reps) {        // restore the input for the next pass (DA aliases nothing the forward reads, SIN is intact)
      WG_SYNC();
    }
  }
  for (int i = threadIdx.x; i < sh.NR * sin_dim; i += blockDim.x) {
    const int r = i / sin_dim, c = i % sin_dim;
    dsin[((size_t)b0 * N + r) * sin_dim + c] = L.DA[r * LDN + c];
  }
  WG_SYNC();
  gnn_store_grads(acc, vacc, gpart + (size_t)blockIdx.x * kGnnGrads);
}

// =================================================================================================
// inference recursion (stove.py:696-713 + full_state :103-170 + constrain_z_dyn)
//   z1      (B,N,18)      state at t = skip-1, layout [sx, sy/sx, x, y, vx, vy, latent 12]
//   zsup    (B,Ts,N,6)    SuPAIR means   [sx, sy/sx, x, y, vx, vy]  for t = skip .. T-1 (Ts = T - skip)
//   zsstd   (B,Ts,N,6)    SuPAIR stds
//   eps     (B,Ts,N,18)   standard-normal draws
//   extra   (B,Ts,N,E)    per-step extra core inputs (action embedding, appearance) of step t-1, E = sin_dim-16
// outputs, all (B,Ts,N,.): z 18, zdyn 16, zdstd 16, mean 18, std 18, pred 32 (optional)
// =================================================================================================

// ---- saved activations of one step (what gnn_backward + the epilogue backward read) ---------------
// dense block per (workgroup, step): 7 node buffers [NRmax][32] (SIN,H1,PRED,F1,F2,O1,RES), CAT [NRmax][64],
// R1,A1 [NEmax][64], R2,A2,R3 [NEmax][32], ATT, DIST [NEmax]  with NRmax = G N, NEmax = G N N.
__host__ __device__ inline size_t gnn_act_floats(int N, int G) {
  const size_t nr = (size_t)G * N, ne = (size_t)G * N * N;
  return nr * (7 * 32 + 64) + ne * (2 * 64 + 3 * 32 + 2);
}
template <int Q, bool FROM_G>
__device__ __forceinline__ void seg_get(float* v, const float* lds, int ld, int shift, int n, const float* __restrict__ g) {
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = (int)threadIdx.x + q * 256;
    float x = 0.0f;
    if (i < n) x = FROM_G ? g[i] : lds[(i >> shift) * ld + (i & ((1 << shift) - 1))];
    v[q] = x;
  }
}
template <int Q, bool TO_G>
__device__ __forceinline__ void seg_put(const float* v, float* lds, int ld, int shift, int n, float* __restrict__ g) {
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = (int)threadIdx.x + q * 256;
    if (i < n) {
      if (TO_G) g[i] = v[q];
      else lds[(i >> shift) * ld + (i & ((1 << shift) - 1))] = v[q];
    }
  }
}
// Two-phase transfer: every source read (global loads for a restore, LDS reads for a save) is issued
// before the first dependent write, so a restore costs ONE memory latency instead of one per buffer.
// Q* = elements per thread of each buffer class (256 threads).
template <bool STORE, int QN, int QC, int Q64, int Q32>
__device__ __forceinline__ void gnn_act_xfer(const GnnLds& L, const GnnShape& sh, float* __restrict__ g, int nr_max, int ne_max) {
  constexpr int NV = 7 * QN + QC + 2 * Q64 + 3 * Q32 + 2;
  float v[NV];
  float* node32[7] = {L.SIN, L.H1, L.PRED, L.F1, L.F2, L.O1, L.RES};
  float* edge64[2] = {L.R1, L.A1};
  float* edge32[3] = {L.R2, L.A2, L.R3};
  float* edge1[2] = {L.ATT, L.DIST};
  const int n32 = sh.NR * 32, n64 = sh.NR * 64, e64 = sh.NE * 64, e32 = sh.NE * 32, e1 = sh.NE;
  {
    const float* gg = g;
    int o = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) { seg_get<QN, !STORE>(v + o, node32[k], LDN, 5, n32, gg); o += QN; gg += nr_max * 32; }
    seg_get<QC, !STORE>(v + o, L.CAT, LDC, 6, n64, gg); o += QC; gg += nr_max * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) { seg_get<Q64, !STORE>(v + o, edge64[k], LDC, 6, e64, gg); o += Q64; gg += ne_max * 64; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { seg_get<Q32, !STORE>(v + o, edge32[k], LDN, 5, e32, gg); o += Q32; gg += ne_max * 32; }
#pragma unroll
    for (int k = 0; k < 2; ++k) { seg_get<1, !STORE>(v + o, edge1[k], 1, 0, e1, gg); o += 1; gg += ne_max; }
  }
  {
    float* gg = g;
    int o = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) { seg_put<QN, STORE>(v + o, node32[k], LDN, 5, n32, gg); o += QN; gg += nr_max * 32; }
    seg_put<QC, STORE>(v + o, L.CAT, LDC, 6, n64, gg); o += QC; gg += nr_max * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) { seg_put<Q64, STORE>(v + o, edge64[k], LDC, 6, e64, gg); o += Q64; gg += ne_max * 64; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { seg_put<Q32, STORE>(v + o, edge32[k], LDN, 5, e32, gg); o += Q32; gg += ne_max * 32; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      seg_put<1, STORE>(v + o, edge1[k], 1, 0, e1, gg);
      if (!STORE && k == 1 && (int)threadIdx.x < e1) L.AUXE[threadIdx.x * 16 + 1] = v[o];   // dist also feeds the aux MFMA operand
      o += 1;
      gg += ne_max;
    }
  }
}
// shape classes with a register-resident transfer: 0: N<=3-ish, 1: up to G*N = 8 nodes / 36 edges, 2: none (recompute)
__device__ __forceinline__ int gnn_act_class(const GnnShape& sh) {
  if (sh.NR * 64 <= 256 && sh.NE * 64 <= 768) return 0;
  if (sh.NR * 64 <= 512 && sh.NE * 64 <= 2304) return 1;
  return 2;
}
template <bool STORE>
__device__ __forceinline__ bool gnn_act_any(const GnnLds& L, const GnnShape& sh, float* g, int nr_max, int ne_max) {
  const int cls = gnn_act_class(sh);
  if (cls == 0) gnn_act_xfer<STORE, 1, 1, 3, 2>(L, sh, g, nr_max, ne_max);
  else if (cls == 1) gnn_act_xfer<STORE, 1, 2, 9, 5>(L, sh, g, nr_max, ne_max);
  return cls < 2;
}

struct LoopConst {
  float pos_var, vel_std, lat_std;
};

__device__ __forceinline__ float std_scale(int d, const LoopConst& k) { return d < 2 ? k.pos_var : (d < 4 ? k.vel_std : k.lat_std); }

// per (row, d<16) forward epilogue; writes z/zdyn/... and the next SIN
__device__ __forceinline__ void loop_epilogue_fwd(const GnnLds& L, const GnnShape& sh, const LoopConst& kc, int b0, int Ts, int ts,
                                                  const float* __restrict__ zsup, const float* __restrict__ zsstd,
                                                  const float* __restrict__ eps, float* __restrict__ z,
                                                  float* __restrict__ zdyn, float* __restrict__ zdstd,
                                                  float* __restrict__ mean, float* __restrict__ stdv, float* Znew) {
  for (int idx = threadIdx.x; idx < sh.NR * 18; idx += blockDim.x) {
    const int r = idx / 18, q = idx % 18;                 // q: dim of the 18-vector
    const int b = b0 + r / sh.N, n = r % sh.N;
    const size_t o = ((size_t)b * Ts + ts) * sh.N + n;
    float mu, sg;
    if (q < 2) {
      mu = zsup[o * 6 + q];
      sg = zsstd[o * 6 + q];
    } else {
      const int d = q - 2;
      const float m = 2.0f * sigmoidf_(L.RES[r * LDN + d]) - 1.0f;
      const float sd = std_scale(d, kc) * sigmoidf_(L.RES[r * LDN + 16 + d]);
      const float zd = m + (d < 2 ? L.SIN[r * LDN + d] : 0.0f);
      zdyn[o * 16 + d] = zd;
      zdstd[o * 16 + d] = sd;
      if (d < 4) {
        const float ms = zsup[o * 6 + 2 + d], ss = zsstd[o * 6 + 2 + d];
        const float sd2 = sd * sd, ss2 = ss * ss, D = sd2 + ss2;
        mu = (ss2 * zd + sd2 * ms) / D;
        sg = sd * ss / sqrtf(D);
      } else {
        mu = zd;
        sg = sd;
      }
    }
    const float zv = fmaf(sg, eps[o * 18 + q], mu);
    z[o * 18 + q] = zv;
    mean[o * 18 + q] = mu;
    stdv[o * 18 + q] = sg;
    Znew[r * 20 + q] = zv;
  }
}

template <bool ELU, bool N6>
__global__ __launch_bounds__(256) void dyn_loop_fwd_k(
    const float* __restrict__ z1, const float* __restrict__ zsup, const float* __restrict__ zsstd,
    const float* __restrict__ eps, const float* __restrict__ extra, const float* __restrict__ P,
    float* __restrict__ z, float* __restrict__ zdyn, float* __restrict__ zdstd, float* __restrict__ mean,
    float* __restrict__ stdv, float* __restrict__ pred, float* __restrict__ act,
    int B, int Ts, int N, int G, int sin_dim, int lim_enc, int elu, LoopConst kc) {
  elu = ELU ? 1 : 0;       // compile-time activation (the ocml expm1f path of ELU otherwise sits in every phi)
  if (N6) {                // six objects, one sequence per workgroup: BASELINE.json's multibilliards shape
    N = 6;
    G = 1;
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GnnLds L = carve(lds);
  const int b0 = blockIdx.x * G;
  const GnnShape sh = make_shape(N, G, b0, B, sin_dim, lim_enc, elu);
  const int E = sin_dim - 16;
  const size_t act_stride = gnn_act_floats(N, G);
  lds_zero(lds, kGnnLdsFloats);
  WG_SYNC();
  gnn_setup(L, sh, P + 2 * W_END);
  WG_SYNC();
  float* Z = L.X;     // [16][20] current state z[t-1]
  for (int i = threadIdx.x; i < sh.NR * 18; i += blockDim.x) Z[(i / 18) * 20 + i % 18] = z1[(size_t)b0 * N * 18 + i];
  const FwdW fw = gnn_fwdw_load(P);
  WG_SYNC();
  for (int ts = 0; ts < Ts; ++ts) {
    for (int i = threadIdx.x; i < sh.NR * sin_dim; i += blockDim.x) {
      const int r = i / sin_dim, c = i % sin_dim;
      float v;
      if (c < 16) v = Z[r * 20 + 2 + c];
      else v = extra[(((size_t)(b0 + r / N) * Ts + ts) * N + r % N) * E + (c - 16)];
      L.SIN[r * LDN + c] = v;
    }
    WG_SYNC();
    gnn_forward(L, sh, P, fw);
    if (act != nullptr)      // keep this step's activations for the backward (instead of recomputing them there)
      gnn_act_any<true>(L, sh, act + ((size_t)blockIdx.x * Ts + ts) * act_stride, G * N, G * N * N);
    loop_epilogue_fwd(L, sh, kc, b0, Ts, ts, zsup, zsstd, eps, z, zdyn, zdstd, mean, stdv, Z);
    if (pred != nullptr) {
      for (int i = threadIdx.x; i < sh.NR * 32; i += blockDim.x) {
        const int r = i >> 5, c = i & 31;
        pred[(((size_t)(b0 + r / N) * Ts + ts) * N + r % N) * 32 + c] = L.PRED[r * LDN + c];
      }
    }
    WG_SYNC();
  }
}

// backward of the recursion.  Upstream gradients (any may be null): dz, dzdyn, dmean, dstd (B,Ts,N,.), dpred.
// Outputs: dz1 (B,N,18), dzsup, dzsstd (B,Ts,N,6), dextra (B,Ts,N,E), gpart[block][kGnnGrads].
template <bool ELU, bool N6>
__global__ __launch_bounds__(256) void dyn_loop_bwd_k(
    const float* __restrict__ z1, const float* __restrict__ zsup, const float* __restrict__ zsstd,
    const float* __restrict__ eps, const float* __restrict__ extra, const float* __restrict__ P,
    const float* __restrict__ z, const float* __restrict__ act,
    const float* __restrict__ dz, const float* __restrict__ dzdyn, const float* __restrict__ dmean,
    const float* __restrict__ dstd, const float* __restrict__ dpred,
    float* __restrict__ dz1, float* __restrict__ dzsup, float* __restrict__ dzsstd, float* __restrict__ dextra,
    float* __restrict__ gpart,
    int B, int Ts, int N, int G, int sin_dim, int lim_enc, int elu, LoopConst kc) {
  elu = ELU ? 1 : 0;
  if (N6) {
    N = 6;
    G = 1;
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GnnLds L = carve(lds);
  const int b0 = blockIdx.x * G;
  const GnnShape sh = make_shape(N, G, b0, B, sin_dim, lim_enc, elu);
  const int E = sin_dim - 16;
  lds_zero(lds, kGnnLdsFloats);
  f32x4 acc[SL_END], vacc[VSLOTS];
#pragma unroll
  for (int k = 0; k < SL_END; ++k) acc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < VSLOTS; ++k) vacc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float* CAR = L.X;   // [16][20] gradient carried into z[t] from step t+1
  const WFrag<32> o1t = wfrag_load<32>(P + W_END + W_O1 + (wave_id() & 1) * 16 * 32, 32);
  WG_SYNC();
  gnn_setup(L, sh, P + 2 * W_END);
  WG_SYNC();
  for (int ts = Ts - 1; ts >= 0; --ts) {
    bool restored = false;
    if (act != nullptr) {
      // forward activations of this step, saved by dyn_loop_fwd_k
      restored = gnn_act_any<false>(L, sh, const_cast<float*>(act) + ((size_t)blockIdx.x * Ts + ts) * gnn_act_floats(N, G), G * N, G * N * N);
      if (restored) WG_SYNC();
    }
    if (!restored) {
      // recompute them from the input state z[t-1]
      for (int i = threadIdx.x; i < sh.NR * sin_dim; i += blockDim.x) {
        const int r = i / sin_dim, c = i % sin_dim;
        const int b = b0 + r / N, n = r % N;
        float v;
        if (c < 16) v = (ts == 0) ? z1[((size_t)b * N + n) * 18 + 2 + c] : z[(((size_t)b * Ts + ts - 1) * N + n) * 18 + 2 + c];
        else v = extra[(((size_t)b * Ts + ts) * N + n) * E + (c - 16)];
        L.SIN[r * LDN + c] = v;
      }
      const FwdW fw = gnn_fwdw_load(P);
      WG_SYNC();
      gnn_forward(L, sh, P, fw);
    }
    // epilogue backward: per (row, d < 16) -> dRES in L.DA, SuPAIR grads, position carry in L.DDIST
    for (int idx = threadIdx.x; idx < 16 * 18; idx += blockDim.x) {
      const int r = idx / 18, q = idx % 18;
      if (r >= sh.NR) {
        if (q >= 2) {
          L.DA[r * LDN + q - 2] = 0.0f;
          L.DA[r * LDN + 16 + q - 2] = 0.0f;
        }
        continue;
      }
      const int b = b0 + r / N, n = r % N;
      const size_t o = ((size_t)b * Ts + ts) * N + n;
      const float ep = eps[o * 18 + q];
      const float gz = (dz != nullptr ? dz[o * 18 + q] : 0.0f) + CAR[r * 20 + q];
      const float gmu = gz + (dmean != nullptr ? dmean[o * 18 + q] : 0.0f);
      const float gsg = gz * ep + (dstd != nullptr ? dstd[o * 18 + q] : 0.0f);
      if (q < 2) {
        dzsup[o * 6 + q] = gmu;
        dzsstd[o * 6 + q] = gsg;
        continue;
      }
      const int d = q - 2;
      const float kd = std_scale(d, kc);
      const float m = 2.0f * sigmoidf_(L.RES[r * LDN + d]) - 1.0f;
      const float sd = kd * sigmoidf_(L.RES[r * LDN + 16 + d]);
      const float zd = m + (d < 2 ? L.SIN[r * LDN + d] : 0.0f);
      float gzd = (dzdyn != nullptr) ? dzdyn[o * 16 + d] : 0.0f;
      float gsd;
      if (d < 4) {
        const float ms = zsup[o * 6 + 2 + d], ss = zsstd[o * 6 + 2 + d];
        const float sd2 = sd * sd, ss2 = ss * ss, D = sd2 + ss2, iD = 1.0f / D;
        const float mu = (ss2 * zd + sd2 * ms) * iD;
        const float rD = rsqrtf(D);
        gzd += gmu * ss2 * iD;
        gsd = gmu * (ms - mu) * iD * 2.0f * sd + gsg * ss * ss2 * iD * rD;
        dzsup[o * 6 + 2 + d] = gmu * sd2 * iD;
        dzsstd[o * 6 + 2 + d] = gmu * (zd - mu) * iD * 2.0f * ss + gsg * sd * sd2 * iD * rD;
      } else {
        gzd += gmu;
        gsd = gsg;
      }
      if (d < 2) L.PC[r * 2 + d] = gzd;                                    // z_dyn position = previous position + delta
      L.DA[r * LDN + d] = gzd * 0.5f * (1.0f - m * m);                      // m = 2 sigmoid - 1
      L.DA[r * LDN + 16 + d] = gsd * sd * (1.0f - sd / kd);                 // sd = k sigmoid
    }
    WG_SYNC();
    gnn_backward(L, sh, P + W_END, o1t, acc, vacc,
                 dpred != nullptr ? dpred + ((size_t)b0 * Ts + ts) * N * 32 : nullptr, (size_t)Ts * N * 32);
    // new carry into z[t-1]
    for (int i = threadIdx.x; i < sh.NR * sin_dim; i += blockDim.x) {
      const int r = i / sin_dim, c = i % sin_dim;
      const int b = b0 + r / N, n = r % N;
      const float g = L.DA[r * LDN + c];
      if (c < 16) CAR[r * 20 + 2 + c] = g + (c < 2 ? L.PC[r * 2 + c] : 0.0f);
      else dextra[(((size_t)b * Ts + ts) * N + n) * E + (c - 16)] = g;
    }
    if (threadIdx.x < sh.NR * 2) CAR[(threadIdx.x >> 1) * 20 + (threadIdx.x & 1)] = 0.0f;
    WG_SYNC();
  }
  for (int i = threadIdx.x; i < sh.NR * 18; i += blockDim.x) dz1[(size_t)b0 * N * 18 + i] = CAR[(i / 18) * 20 + i % 18];
  WG_SYNC();
  gnn_store_grads(acc, vacc, gpart + (size_t)blockIdx.x * kGnnGrads);
}

// =================================================================================================
// generative rollout (stove.py:823-846), mean prediction, forward only
//   z_last (B,N,18) [sx, sy, x, y, vx, vy, latent]; extra (B,A,N,E) indexed (t-1) % A (A>=1) or null
//   z_pred (B,num,N,18); zstd (B,num,N,16) optional; pred (B,num,N,32) optional
// =================================================================================================
__global__ __launch_bounds__(256) void rollout_fwd_k(const float* __restrict__ z_last, const float* __restrict__ extra,
                                                     const float* __restrict__ P, float* __restrict__ z_pred,
                                                     float* __restrict__ zstd, float* __restrict__ pred,
                                                     int B, int num, int A, int N, int G, int sin_dim, int lim_enc, int elu, LoopConst kc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GnnLds L = carve(lds);
  const int b0 = blockIdx.x * G;
  const GnnShape sh = make_shape(N, G, b0, B, sin_dim, lim_enc, elu);
  const int E = sin_dim - 16;
  lds_zero(lds, kGnnLdsFloats);
  WG_SYNC();
  gnn_setup(L, sh, P + 2 * W_END);
  WG_SYNC();
  float* Z = L.X;
  for (int i = threadIdx.x; i < sh.NR * 18; i += blockDim.x) Z[(i / 18) * 20 + i % 18] = z_last[(size_t)b0 * N * 18 + i];
  const FwdW fw = gnn_fwdw_load(P);
  WG_SYNC();
  for (int t = 0; t < num; ++t) {
    for (int i = threadIdx.x; i < sh.NR * sin_dim; i += blockDim.x) {
      const int r = i / sin_dim, c = i % sin_dim;
      float v;
      if (c < 16) v = Z[r * 20 + 2 + c];
      else v = extra[(((size_t)(b0 + r / N) * A + (t % A)) * N + r % N) * E + (c - 16)];
      L.SIN[r * LDN + c] = v;
    }
    WG_SYNC();
    gnn_forward(L, sh, P, fw);
    for (int idx = threadIdx.x; idx < sh.NR * 18; idx += blockDim.x) {
      const int r = idx / 18, q = idx % 18;
      const size_t o = ((size_t)(b0 + r / N) * num + t) * N + r % N;
      float v;
      if (q < 2) {
        v = Z[r * 20 + q];                                  // scale stays constant
      } else {
        const int d = q - 2;
        v = 2.0f * sigmoidf_(L.RES[r * LDN + d]) - 1.0f + (d < 2 ? L.SIN[r * LDN + d] : 0.0f);
        if (zstd != nullptr) zstd[o * 16 + d] = std_scale(d, kc) * sigmoidf_(L.RES[r * LDN + 16 + d]);
      }
      z_pred[o * 18 + q] = v;
      Z[r * 20 + q] = v;
    }
    if (pred != nullptr) {
      for (int i = threadIdx.x; i < sh.NR * 32; i += blockDim.x) {
        const int r = i >> 5, c = i & 31;
        pred[(((size_t)(b0 + r / N) * num + t) * N + r % N) * 32 + c] = L.PRED[r * LDN + c];
      }
    }
    WG_SYNC();
  }
}

}  // namespace stove
