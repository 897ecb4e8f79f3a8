// Small-graph variant of the GNN time loop (N <= 6 objects, one sequence per workgroup), gfx950.
//
// Why a second formulation.  With N = 3 the MFMA path (gnn.hip) fills 3 of the 16 rows of every node tile and
// 9 of 16 of every edge tile, and -- what actually bounds the T-serial recursion -- every dense layer is a
// workgroup-wide stage: LDS write, barrier, LDS read, MFMA chain, epilogue; 11 such stages per step.  But the
// node MLPs (encoder, self-dynamics, affector, output) act on every node row independently, and the edge MLPs
// on every edge independently; rows only mix at the edge gather and at the aggregation.  So here
//   * one WAVE owns one node row for the whole step -- or two, one per HALF-wave, in the kernels built for five or six
//     objects (the node layers are 32 wide: the second half-wave, which otherwise mirrors the first, carries row wave + 4
//     through the same instructions);
//   * a 32-wide activation vector lives in ONE VGPR (lane k of the half holds x[k]); a dense layer sends it through a per-wave
//     LDS scratch, reads it back as 8 broadcast float4 and runs 16 v_pk_fma_f32 against the lane's weight row, which
//     streams from LDS as float4 ([K/4][OUT][4] layout, conflict-free) one layer ahead; layers chain inside the
//     wave with no workgroup barrier (a first version broadcast x with 32 v_readlane per layer: 391 vs 255 cycles);
//   * the edge MLPs and the self-dynamics run on the matrix cores with the edges (nodes) as the 16 columns of an MFMA chain,
//     one chain per wave (sm_edge_phase_mfma); 30 edges (N = 6) are two column tiles through the same weights;
//   * the workgroup synchronises exactly twice per step (before the edge phase, before the aggregation).
// All forward weights (90 KB) sit in LDS for the whole launch.
#include "common.h"
#include "split16.h"

namespace stove {

constexpr int kSmW = W_END;                                   // [layer][K/4][OUT][4]
constexpr int kSmWaves = 4;     // waves per workgroup; more than 4 halves the register budget (2 waves per SIMD) and the kernels spill

// A kernel is built for up to NMX objects: 4 (one node row per wave, one tile of edge columns) or 6 (two rows per wave, two tiles)
template <int NMX>
struct SmShape {
  static_assert(NMX == 4 || NMX == 6, "built for up to four or up to six objects");
  static constexpr int RP = NMX > 4 ? 2 : 1;                        // node rows per wave
  static constexpr int ET = (NMX * (NMX - 1) + 15) / 16;            // 16-column tiles of the edge chains
  static constexpr int NE = NMX * NMX;                              // edge rows i N + j of the LDS exchange buffers
  static constexpr int kLdsFloats = kSmW + V_END + NMX * 256 + NMX * 4 + NE * 32 + ((NE + 3) & ~3);
};

struct SmLds {
  float *W, *V, *PR, *POS, *R3, *ATT;
};
template <int NMX>
__device__ __forceinline__ SmLds sm_carve(float* base) {
  SmLds L;
  L.W = base;
  L.V = L.W + kSmW;
  L.PR = L.V + V_END;        // [NMX][256]  W_a s_i | W_b s_j | A_a s_i | A_b s_j of every node
  L.POS = L.PR + NMX * 256;  // [NMX][4]    encoder outputs 0, 1 (the positions the distances use)
  L.R3 = L.POS + NMX * 4;    // [NMX^2][32] relation outputs by edge row i*N + j
  L.ATT = L.R3 + SmShape<NMX>::NE * 32;    // [NMX^2]
  return L;
}

__device__ __forceinline__ void sm_copy_packed(float* dst, const float* __restrict__ src, int n) {
  for (int i = threadIdx.x; i < n / 4; i += blockDim.x) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
}
__device__ __forceinline__ void sm_setup(const SmLds& L, const float* __restrict__ P) {
  sm_copy_packed(L.W, P + P_WPACK, W_END);          // every layer already in [K/4][OUT][4] order (the image's packed section)
  for (int i = threadIdx.x; i < V_END; i += blockDim.x) L.V[i] = P[2 * W_END + i];
}

struct SmCfg {
  int N, sin_dim, lim_enc, elu;
  long long* stamps;     // debug: [4 waves][16] cycle stamps of workgroup 0 (normally null)
};
__device__ __forceinline__ void sm_stamp(const SmCfg& cf, int k) {
  if (cf.stamps != nullptr && blockIdx.x == 0 && lane_id() == 0) cf.stamps[wave_id() * 16 + k] = (long long)__builtin_readcyclecounter();
}
// Saved activations ("streams", sm_act2_floats: what the small-graph backward and its weight-gradient pass read): per sequence
// one stream per buffer over all steps -- row (t N + r) of a node buffer, row (t N(N-1) + q) of an edge buffer -- so that
// 16 consecutive rows of any layer input are one contiguous tile for the weight-gradient MFMAs.
struct SmAct {
  float *SIN, *H1, *PRED, *F1, *F2, *O1, *RES, *S, *F3, *R1, *A1, *R2, *A2, *R3, *ATT, *DIST;
};
constexpr int kSmNodeBufs = 9;      // SIN, S, H1, PRED, F1, F2, F3, O1, RES
__host__ __device__ inline size_t sm_act2_floats(int N, int Ts) {
  const size_t f = (size_t)Ts * ((size_t)N * kSmNodeBufs * 32 + (size_t)N * (N - 1) * (64 + 64 + 32 + 32 + 32 + 2));
  return (f + 3) & ~(size_t)3;       // every sequence's streams start 16-byte aligned
}
// stream pointers of one sequence, advanced to step ts
__device__ __forceinline__ SmAct sm_act2(float* seq, int N, int Ts, int ts) {
  SmAct a;
  const size_t nrows = (size_t)Ts * N, erows = (size_t)Ts * N * (N - 1);
  float* nb = seq + (size_t)ts * N * 32;
  a.SIN = nb; a.S = nb + nrows * 32; a.H1 = nb + 2 * nrows * 32; a.PRED = nb + 3 * nrows * 32; a.F1 = nb + 4 * nrows * 32;
  a.F2 = nb + 5 * nrows * 32; a.F3 = nb + 6 * nrows * 32; a.O1 = nb + 7 * nrows * 32; a.RES = nb + 8 * nrows * 32;
  float* eb = seq + nrows * kSmNodeBufs * 32;
  const size_t eo = (size_t)ts * N * (N - 1);
  a.R1 = eb + eo * 64; a.A1 = eb + erows * 64 + eo * 64;
  float* e32 = eb + erows * 128;
  a.R2 = e32 + eo * 32; a.A2 = e32 + erows * 32 + eo * 32; a.R3 = e32 + 2 * erows * 32 + eo * 32;
  float* e1 = e32 + 3 * erows * 32;
  a.ATT = e1 + eo; a.DIST = e1 + erows + eo;
  return a;
}

// A layer's weight rows for this lane, fetched from LDS ahead of use: the weights never depend on the data, so
// every layer's fetch is issued while the previous layer still computes and the LDS latency (~100+ cycles per
// read, which dominated the first version of this kernel) disappears from the serial chain.
template <int K4>
struct SmW {
  float4 w[K4];
};
// HK (round 4; kernels with ONE node row per wave, where the upper half-wave only mirrored the lower one): the two halves split K --
// the lower half takes k < K / 2, the upper half k >= K / 2 (its own half of the weight row: half the LDS reads and half the registers
// per layer) -- and sm_dotw adds the two partial sums across the halves with one v_permlane32_swap (new on gfx950).
template <int K4, bool HK = false>
__device__ __forceinline__ SmW<K4> sm_wload(const float* Wl, int OUT, int o, int k4_0 = 0) {
  SmW<K4> r;
  if (HK) {
    const int kh = k4_0 + (lane_id() >> 5) * (K4 / 2);
#pragma unroll
    for (int k4 = 0; k4 < K4 / 2; ++k4) r.w[k4] = *reinterpret_cast<const float4*>(Wl + ((kh + k4) * OUT + o) * 4);
    return r;
  }
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) r.w[k4] = *reinterpret_cast<const float4*>(Wl + ((k4_0 + k4) * OUT + o) * 4);
  return r;
}
// Packed fp32 FMA (v_pk_fma_f32: two lanes of fp32 per VGPR pair, full rate on CDNA3/4).  The operands are laid out
// so that every packed operand is a natural register pair -- (w.x, w.y) and (w.z, w.w) of a float4, an SGPR pair
// of consecutive readlanes -- otherwise the compiler pays two v_mov per packed instruction.
// (v2f / pk_fma live in common.h)

// y[o] = sum_k W[o][k] x[k].  The activation vector goes through a per-wave LDS scratch and comes back as K/4 broadcast
// float4 reads (every lane reads the same address: one LDS pass), then K/2 packed FMAs on four independent chains.
// Measured per 32 x 32 layer on one wave per SIMD (tools/ubench/dot_variants.hip): 255 cycles, against 391 for
// 32 v_readlane feeding the FMAs through SGPRs (every FMA then waits on the SGPR its readlane has just written).
// Each HALF-wave is a row of its own: lane k of the half holds x[k], lane o gets y[o] of its half's row (both halves use the
// same weight rows W[o]).  With one node row per wave the upper half mirrors the lower one and computes the same numbers.
template <int K4, bool HK = false>
__device__ __forceinline__ float sm_dotw(const SmW<K4>& W, float x) {
  __shared__ __attribute__((aligned(16))) float xb[kSmWaves][64];
  float* p = xb[wave_id()];
  p[lane_id()] = x;
  if (HK) {
    static_assert(K4 % 4 == 0, "two packed-FMA pairs per half");
    // both halves read the LOWER half's copy of x (lanes 0..31 are the row; what the upper lanes hold is not relied on)
    p += (lane_id() >> 5) * (2 * K4);              // the half's K / 2 = 2 K4 inputs
    float4 xh[K4 / 2];
#pragma unroll
    for (int k4 = 0; k4 < K4 / 2; ++k4) xh[k4] = *reinterpret_cast<const float4*>(p + 4 * k4);
    v2f a = {0.0f, 0.0f}, b = {0.0f, 0.0f}, c = {0.0f, 0.0f}, d = {0.0f, 0.0f};
#pragma unroll
    for (int k4 = 0; k4 < K4 / 2; k4 += 2) {
      a = pk_fma(v2f{W.w[k4].x, W.w[k4].y}, v2f{xh[k4].x, xh[k4].y}, a);
      b = pk_fma(v2f{W.w[k4].z, W.w[k4].w}, v2f{xh[k4].z, xh[k4].w}, b);
      c = pk_fma(v2f{W.w[k4 + 1].x, W.w[k4 + 1].y}, v2f{xh[k4 + 1].x, xh[k4 + 1].y}, c);
      d = pk_fma(v2f{W.w[k4 + 1].z, W.w[k4 + 1].w}, v2f{xh[k4 + 1].z, xh[k4 + 1].w}, d);
    }
    a += b;
    c += d;
    a += c;
    // the two halves' partial sums meet through one v_permlane32_swap (common.h sum_xor32)
    return sum_xor32(a.x + a.y);
  }
  p += lane_id() & 32;
  float4 xv[K4];
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) xv[k4] = *reinterpret_cast<const float4*>(p + 4 * k4);
  v2f a = {0.0f, 0.0f}, b = {0.0f, 0.0f}, c = {0.0f, 0.0f}, d = {0.0f, 0.0f};
#pragma unroll
  for (int k4 = 0; k4 < K4; k4 += 2) {
    a = pk_fma(v2f{W.w[k4].x, W.w[k4].y}, v2f{xv[k4].x, xv[k4].y}, a);
    b = pk_fma(v2f{W.w[k4].z, W.w[k4].w}, v2f{xv[k4].z, xv[k4].w}, b);
    c = pk_fma(v2f{W.w[k4 + 1].x, W.w[k4 + 1].y}, v2f{xv[k4 + 1].x, xv[k4 + 1].y}, c);
    d = pk_fma(v2f{W.w[k4 + 1].z, W.w[k4 + 1].w}, v2f{xv[k4 + 1].z, xv[k4 + 1].w}, d);
  }
  a += b;
  c += d;
  a += c;
  return a.x + a.y;
}

// ---- edge phase on the matrix cores -----------------------------------------------------------------------------------
// In the wave-per-row form the N(N-1) edges are 3-layer chains dealt out over four waves: with N = 3 two waves carry two edges
// each plus the self-dynamics of their node row, 8 dependent layers between the two barriers of a step -- half of the step
// (cycle stamps: profiles/r02_loop_stamps_before.txt).  Here the edge phase is column-parallel instead: ALL edges are the
// columns of one v_mfma_f32_16x16x4_f32 chain (exact fp32 FMA chains), one wave per chain:
//     wave 3  relation chain   R1 = phi(P_a[i] + P_b[j] + w_d d + b) -> R2 = phi(W R1 + b) -> R3 = W R2 + b + R2
//     wave 2  attention chain  A1 likewise -> A2 -> att = exp(w . A2 + b)
//     wave 1  self-dynamics of all node rows (columns = nodes)   H1 = phi(W S + b), SD = W H1 + b + H1
// A layer's result tile has output o = 16 t' + 4 (lane >> 4) + reg in register `reg` of lane (column, lane >> 4) -- exactly the
// B operand the next layer's MFMA wants for k-slot (lane >> 4) of k-step (t', reg): activations chain from accumulators to
// operands with no data movement, and the A operands W[16 t' + i][16 t + 4 g .. +3] are float4s of the [K/4][OUT][4] LDS
// weight image the row-per-lane dots already use.  3 dependent layers of 16-32 MFMAs instead of 8 layers of LDS round trips.
typedef __attribute__((ext_vector_type(4))) float smf4;

// acc[t'] += sum_k W[16 t' + i][k] x[k][column], k = 16 t + 4 g + s, for the two output tiles t' of a 32-wide layer
template <int KT>
struct SmMW {
  float4 w[KT][2];        // [input tile t][output tile t']: W[16 t' + i][16 t + 4 g .. + 3]
};
// a layer's A operands, fetched from the LDS weight image ahead of use (all of them at once: the reads overlap with whatever
// the wave does before the layer; fetched one input tile at a time, each MFMA group waited ~130 cycles for its weights)
template <int KT>
__device__ __forceinline__ SmMW<KT> sm_mfma_wload(const float* Wl, int lane) {
  SmMW<KT> m;
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    m.w[t][0] = *reinterpret_cast<const float4*>(Wl + ((4 * t + g) * 32 + i) * 4);
    m.w[t][1] = *reinterpret_cast<const float4*>(Wl + ((4 * t + g) * 32 + 16 + i) * 4);
  }
  return m;
}
// Four accumulation chains (output tile x parity of the input tile), their MFMAs issued round-robin: a chain's next MFMA
// comes four issues (128 cycles) after its previous one.  With two chains the dependent-accumulator latency showed: 58-86
// cycles per MFMA measured in place against the 32-cycle issue rate.  The two partial tiles are added at the end.
template <int KT>
__device__ __forceinline__ void sm_mfma_layer(const SmMW<KT>& m, const smf4 (&x)[KT], smf4 (&acc)[2]) {
  static_assert(KT % 2 == 0, "input tiles are taken in pairs");
  smf4 p0 = {0.0f, 0.0f, 0.0f, 0.0f}, p1 = p0;
#pragma unroll
  for (int t = 0; t < KT; t += 2) {
    const float4 a0 = m.w[t][0], a1 = m.w[t][1], b0 = m.w[t + 1][0], b1 = m.w[t + 1][1];
    const float a0s[4] = {a0.x, a0.y, a0.z, a0.w}, a1s[4] = {a1.x, a1.y, a1.z, a1.w};
    const float b0s[4] = {b0.x, b0.y, b0.z, b0.w}, b1s[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0s[e], x[t][e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1s[e], x[t][e], acc[1], 0, 0, 0);
      p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0s[e], x[t + 1][e], p0, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1s[e], x[t + 1][e], p1, 0, 0, 0);
    }
  }
  acc[0] += p0;
  acc[1] += p1;
}
// The same for ET column tiles at once (the 30 edges of six objects are two): one set of A operands, the tiles' chains interleaved
// MFMA by MFMA -- 4 ET independent accumulators, so the chain is bound by the issue rate instead of the accumulator latency.
template <int KT, int ET>
__device__ __forceinline__ void sm_mfma_layer_tiles(const SmMW<KT>& m, const smf4 (&x)[ET][KT], smf4 (&acc)[ET][2]) {
  static_assert(KT % 2 == 0, "input tiles are taken in pairs");
  smf4 p0[ET], p1[ET];
#pragma unroll
  for (int c = 0; c < ET; ++c) p0[c] = p1[c] = smf4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < KT; t += 2) {
    const float4 a0 = m.w[t][0], a1 = m.w[t][1], b0 = m.w[t + 1][0], b1 = m.w[t + 1][1];
    const float a0s[4] = {a0.x, a0.y, a0.z, a0.w}, a1s[4] = {a1.x, a1.y, a1.z, a1.w};
    const float b0s[4] = {b0.x, b0.y, b0.z, b0.w}, b1s[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < ET; ++c) {
        acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0s[e], x[c][t][e], acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1s[e], x[c][t][e], acc[c][1], 0, 0, 0);
        p0[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0s[e], x[c][t + 1][e], p0[c], 0, 0, 0);
        p1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1s[e], x[c][t + 1][e], p1[c], 0, 0, 0);
      }
  }
#pragma unroll
  for (int c = 0; c < ET; ++c) {
    acc[c][0] += p0[c];
    acc[c][1] += p1[c];
  }
}
// The general form (backward: transposed weight images with ROWS = the layer's input width as output rows): OT output tiles,
// one accumulation chain each, KT input tiles; weights fetched inside.  Wl = [K/4][ROWS][4] LDS image.
template <int KT, int OT>
__device__ __forceinline__ void sm_mfma_layer_t(const float* Wl, const smf4 (&x)[KT], smf4 (&acc)[OT], int lane) {
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
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, x[t][e], acc[u], 0, 0, 0);
      }
  }
}
// ---- the edge chains' dense layers on half-piece MFMAs (round 5) --------------------------------------------------------------------
// A 32-wide layer of the relation / attention chain as v_mfma_f32_16x16x32_f16 on IEEE-half hi / lo pieces (split16.h: three MFMAs per
// product, 2^-22 of the value per product -- an fp32-grade result) instead of v_mfma_f32_16x16x4_f32: a 64 -> 32 layer is 12
// MFMAs of 16 cycles on six independent accumulators instead of 32 dependent-chain MFMAs of 32 cycles (measured in place: 52
// cycles each), 32 -> 32 is 6 instead of 16.  The accumulator of a layer is still the B operand of the next: k-slot j of lane
// (column, g) in k-block b is feature 32 b + 4 g + j (j < 4) or 32 b + 16 + 4 g + (j - 4) -- the two accumulator tiles 2 b, 2 b + 1 the
// lane holds -- a sum over k does not care about the order as long as the weights use the same one.  The weight fragments (A
// operands, hi and lo) are built ONCE per launch from the fp32 LDS image and stay in registers over all time steps.
// Forward only: activations and weights sit inside half's range; the backward's gradients do not (gnn_small_bwd.hip keeps fp32 MFMAs).
struct SmChainW {
  bf16x8 h2[2][2], l2[2][2];      // second layer (64 -> 32): [k-block b][output tile u], hi and lo pieces
  bf16x8 h3[2], l3[2];            // third layer (32 -> 32) of the relation chain: [output tile u]
};
__device__ __forceinline__ void sm_frag_split(const float4 a, const float4 b, bf16x8& hi, bf16x8& lo) {
  u32x2 h0, l0, h1, l1;
  split4<2, true>(a, h0, l0);
  split4<2, true>(b, h1, l1);
  hi = __builtin_bit_cast(bf16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
  lo = __builtin_bit_cast(bf16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
}
// fragments of the [K/4][32][4] LDS weight image Wl of a layer with KB k-blocks of 32 inputs
template <int KB>
__device__ __forceinline__ void sm_chain_wfrag(const float* Wl, int lane, bf16x8 (*h)[2], bf16x8 (*l)[2]) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int b = 0; b < KB; ++b)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      sm_frag_split(*reinterpret_cast<const float4*>(Wl + ((8 * b + g) * 32 + 16 * u + i) * 4),
                    *reinterpret_cast<const float4*>(Wl + ((8 * b + 4 + g) * 32 + 16 * u + i) * 4), h[b][u], l[b][u]);
}
__device__ __forceinline__ void sm_chain_wbuild(const SmLds& L, SmChainW& cw) {
  const int wv = wave_id(), lane = lane_id();
  if (wv < 2) return;
  const int h = wv == 2 ? 1 : 0;
  sm_chain_wfrag<2>(L.W + (h ? W_A1 : W_R1), lane, cw.h2, cw.l2);
  if (h == 0) {
    bf16x8 h3[1][2], l3[1][2];
    sm_chain_wfrag<1>(L.W + W_R2, lane, h3, l3);
    cw.h3[0] = h3[0][0]; cw.h3[1] = h3[0][1];
    cw.l3[0] = l3[0][0]; cw.l3[1] = l3[0][1];
  }
}
__device__ __forceinline__ float4 sm_f4(smf4 v) { return float4{v[0], v[1], v[2], v[3]}; }
// acc[c][u] += W x for ET column tiles, KB k-blocks: x[c][2 b], x[c][2 b + 1] are the accumulator tiles of the previous layer
template <int KB, int ET>
__device__ __forceinline__ void sm_split_layer_tiles(const bf16x8 (*wh)[2], const bf16x8 (*wl)[2], const smf4 (&x)[ET][2 * KB], smf4 (&acc)[ET][2]) {
  f32x4 p[ET][2], q[ET][2];          // the two cross terms on accumulators of their own (six independent chains per column tile)
#pragma unroll
  for (int c = 0; c < ET; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) p[c][u] = q[c][u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int b = 0; b < KB; ++b)
#pragma unroll
    for (int c = 0; c < ET; ++c) {
      bf16x8 xh, xl;
      sm_frag_split(sm_f4(x[c][2 * b]), sm_f4(x[c][2 * b + 1]), xh, xl);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        p[c][u] = mfma16<true>(wl[b][u], xh, p[c][u]);
        q[c][u] = mfma16<true>(wh[b][u], xl, q[c][u]);
        acc[c][u] = mfma16<true>(wh[b][u], xh, acc[c][u]);
      }
    }
#pragma unroll
  for (int c = 0; c < ET; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[c][u] += p[c][u] + q[c][u];
}

// The same for operands of ANY magnitude (the backward's gradient vectors): every column is scaled by a power of two that brings
// its largest entry into [1, 2) before it is split into half pieces, and the result is scaled back -- exact, and the pieces of every
// entry down to 2^-14 of the column's largest keep their 22 bits (smaller ones add less than 2^-24 of the result).  The column's
// eight values per lane group and its four lane groups (lane ^ 16, lane ^ 32) meet through two v_permlane swaps.
__device__ __forceinline__ void sm_col_scale(const smf4 (&x)[2], float& s, float& inv) {
  float m = fmaxf(fmaxf(fmaxf(fabsf(x[0][0]), fabsf(x[0][1])), fmaxf(fabsf(x[0][2]), fabsf(x[0][3]))),
                  fmaxf(fmaxf(fabsf(x[1][0]), fabsf(x[1][1])), fmaxf(fabsf(x[1][2]), fabsf(x[1][3]))));
  m = max_xor32(max_xor16(m));
  unsigned e = (__float_as_uint(m) >> 23) & 0xffu;
  e = e < 1u ? 1u : (e > 253u ? 253u : e);
  s = __uint_as_float((254u - e) << 23);
  inv = __uint_as_float(e << 23);
}
// acc[c][u] += W^T x for OT output tiles of a layer with 32 inputs (ONE k-block: x[c][0], x[c][1] are the two accumulator tiles of
// the column), x column-normalised; wh / wl: the [OT] hi / lo fragments of the transposed weight image
template <int OT, int ET>
__device__ __forceinline__ void sm_split_layer_t_norm(const bf16x8 (&wh)[OT], const bf16x8 (&wl)[OT], const smf4 (&x)[ET][2], smf4 (&acc)[ET][OT]) {
#pragma unroll
  for (int c = 0; c < ET; ++c) {
    float s, inv;
    sm_col_scale(x[c], s, inv);
    bf16x8 xh, xl;
    sm_frag_split(sm_f4(x[c][0] * s), sm_f4(x[c][1] * s), xh, xl);
    f32x4 p[OT], q[OT], r[OT];
#pragma unroll
    for (int u = 0; u < OT; ++u) {
      const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
      p[u] = mfma16<true>(wl[u], xh, z);
      q[u] = mfma16<true>(wh[u], xl, z);
      r[u] = mfma16<true>(wh[u], xh, z);
    }
#pragma unroll
    for (int u = 0; u < OT; ++u) acc[c][u] += (r[u] + (p[u] + q[u])) * inv;
  }
}
// [OT] fragments of a transposed [32 inputs / 4][16 OT][4] LDS image
template <int OT>
__device__ __forceinline__ void sm_chain_wfrag_t(const float* Wl, int lane, bf16x8 (&h)[OT], bf16x8 (&l)[OT]) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int u = 0; u < OT; ++u)
    sm_frag_split(*reinterpret_cast<const float4*>(Wl + (g * (16 * OT) + 16 * u + i) * 4),
                  *reinterpret_cast<const float4*>(Wl + ((4 + g) * (16 * OT) + 16 * u + i) * 4), h[u], l[u]);
}

// d phi / d pre-activation from the activation's OUTPUT, four values
__device__ __forceinline__ smf4 sm_dphi4(smf4 y, int elu) {
  smf4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = y[e] > 0.0f ? 1.0f : (elu ? y[e] + 1.0f : 0.01f);
  return r;
}
// phi on four values; the branch on the (wave-uniform) nonlinearity is taken once, not per element
__device__ __forceinline__ smf4 sm_phi4(smf4 v, int elu) {
  smf4 r;
  if (elu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = v[e] > 0.0f ? v[e] : expm1f(v[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fmaxf(v[e], 0.01f * v[e]);      // leaky_relu(0.01): max(x, 0.01 x)
  }
  return r;
}
__device__ __forceinline__ smf4 sm_ld4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return smf4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void sm_st4(float* p, smf4 v) { *reinterpret_cast<float4*>(p) = float4{v[0], v[1], v[2], v[3]}; }

// What a lane of the MFMA edge phase keeps over the whole time loop (column -> edge / node, LDS and stream offsets): the
// integer divisions and address arithmetic behind them cost more than the 48 MFMAs of the relation chain when redone per step.
struct SmEdgeLane {
  int valid;                 // column < number of edges (chain waves) / of nodes (self-dynamics wave)
  int pi, pj;                // float offsets into L.PR of the two first-layer halves of the lane's edge (incl. 4 g)
  int pos_i, pos_j;          // float offsets into L.POS
  int e32;                   // e * 32 + 4 g        (L.R3 row of the edge)
  int e;                     // e = i N + j         (L.ATT slot)
  int s64, s32, s1;          // stream offsets: eg * 64 + 4 g, eg * 32 + 4 g, eg
  int node32;                // self-dynamics wave: r * 32 + 4 g
  int i32;                   // i * 32 + 4 g        (row of the edge's target node in [node][32] LDS buffers)
};
__device__ __forceinline__ SmEdgeLane sm_edge_lane(int N, int compact_streams, int tile = 0) {      // tile: columns 16 tile .. 16 tile + 15
  SmEdgeLane el;
  const int wv = wave_id(), lane = lane_id();
  const int c = 16 * tile + (lane & 15), g = lane >> 4;
  const int E = N * (N - 1);
  const int h = wv == 2 ? 1 : 0;
  const int q = c < E ? c : (E > 0 ? E - 1 : 0);
  const int nm1 = N > 1 ? N - 1 : 1;
  const int i = q / nm1, jj = q % nm1, j = jj + (jj >= i ? 1 : 0);
  el.e = i * N + j;
  const int eg = compact_streams ? q : el.e;
  el.valid = wv == 1 ? (c < N) : (c < E);
  el.pi = i * 256 + 128 * h + 4 * g;
  el.pj = j * 256 + 128 * h + 64 + 4 * g;
  el.pos_i = i * 4;
  el.pos_j = j * 4;
  el.e32 = el.e * 32 + 4 * g;
  el.s64 = eg * 64 + 4 * g;
  el.s32 = eg * 32 + 4 * g;
  el.s1 = eg;
  el.node32 = (c < N ? c : N - 1) * 32 + 4 * g;
  el.i32 = i * 32 + 4 * g;
  return el;
}

// Everything of the edge phase that does not depend on the step's data -- the A operands (weights) of the wave's chain and its
// bias / distance vectors -- is read from LDS BEFORE the first barrier of the step (wave 3 idles there for ~6 000 cycles, the
// others issue the reads behind their node work), so the chain starts on the barrier's release with its operands in registers.
struct SmEdgePre {
  SmMW<4> w1;            // self-dynamics wave: w1.w[0..1] = layer 0, w1.w[2..3] = layer 1
  smf4 wd[4], b0[4];     // chains: distance weights and biases of the factorised first layer
  smf4 bl2[2], bl3[2];   // biases of the wave's second / third layer (self-dynamics: layer 0 / layer 1)
};
__device__ __forceinline__ void sm_edge_prefetch(const SmLds& L, SmEdgePre& pre) {
  const int wv = wave_id(), lane = lane_id(), g = lane >> 4;
  const float* V = L.V;
  if (wv == 1) {
    const SmMW<2> a = sm_mfma_wload<2>(L.W + W_S0, lane), b = sm_mfma_wload<2>(L.W + W_S1, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      pre.w1.w[t][0] = a.w[t][0]; pre.w1.w[t][1] = a.w[t][1];
      pre.w1.w[2 + t][0] = b.w[t][0]; pre.w1.w[2 + t][1] = b.w[t][1];
      pre.bl2[t] = sm_ld4(V + V_S0 + 16 * t + 4 * g);
      pre.bl3[t] = sm_ld4(V + V_S1 + 16 * t + 4 * g);
    }
  } else if (wv >= 2) {
    const int h = wv == 2 ? 1 : 0;       // (the chain's weights are the half-piece fragments of SmChainW, resident since the launch began)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pre.wd[t] = sm_ld4(V + (h ? V_WDA : V_WDR) + 16 * t + 4 * g);
      pre.b0[t] = sm_ld4(V + (h ? V_BA0 : V_BR0) + 16 * t + 4 * g);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      pre.bl2[t] = sm_ld4(V + (h ? V_BA1 : V_BR1) + 16 * t + 4 * g);
      pre.bl3[t] = sm_ld4(V + (h ? V_WA2 : V_BR2) + 16 * t + 4 * g);       // attention: the 32 -> 1 weights
    }
  }
}

// The relation (wave 3, h = 0) / attention (wave 2, h = 1) chain: columns = edges (i -> j, i != j), ET tiles of 16 columns taken
// through every layer together (a kernel built for ET tiles is only used where the last tile has edges in it)
template <bool SAVE, int ET>
__device__ __forceinline__ void sm_edge_chain(const SmLds& L, const SmCfg& cf, const SmAct& act, const SmEdgeLane (&el)[ET],
                                              const SmEdgePre& pre, const SmChainW& cw, int h) {
  const int g = lane_id() >> 4;
  const float* V = L.V;
  float* s1p = h ? act.A1 : act.R1;
  float* s2p = h ? act.A2 : act.R2;
  float d[ET];
  smf4 x1[ET][4];
#pragma unroll
  for (int c = 0; c < ET; ++c) {
    const float dx = L.POS[el[c].pos_i] - L.POS[el[c].pos_j], dy = L.POS[el[c].pos_i + 1] - L.POS[el[c].pos_j + 1];
    d[c] = dx * dx + dy * dy;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const smf4 a = sm_ld4(L.PR + el[c].pi + 16 * t), b = sm_ld4(L.PR + el[c].pj + 16 * t);
      x1[c][t] = sm_phi4(a + b + pre.wd[t] * d[c] + pre.b0[t], cf.elu);
    }
  }
  sm_stamp(cf, 8);
  smf4 acc[ET][2], a2[ET][2];
#pragma unroll
  for (int c = 0; c < ET; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[c][t] = pre.bl2[t];
  sm_split_layer_tiles<2, ET>(cw.h2, cw.l2, x1, acc);
  if (SAVE) {          // the stores of the layer input go out behind the MFMAs that consumed it
#pragma unroll
    for (int c = 0; c < ET; ++c)
      if (el[c].valid) {
#pragma unroll
        for (int t = 0; t < 4; ++t) sm_st4(s1p + el[c].s64 + 16 * t, x1[c][t]);
      }
  }
  sm_stamp(cf, 9);
#pragma unroll
  for (int c = 0; c < ET; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t) a2[c][t] = sm_phi4(acc[c][t], cf.elu);
  sm_stamp(cf, 10);
  if (h == 0) {
#pragma unroll
    for (int c = 0; c < ET; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[c][t] = pre.bl3[t];
    {
      const bf16x8 wh[1][2] = {{cw.h3[0], cw.h3[1]}}, wl[1][2] = {{cw.l3[0], cw.l3[1]}};
      sm_split_layer_tiles<1, ET>(wh, wl, a2, acc);
    }
    sm_stamp(cf, 11);
#pragma unroll
    for (int c = 0; c < ET; ++c)
      if (el[c].valid) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const smf4 r3 = acc[c][t] + a2[c][t];
          sm_st4(L.R3 + el[c].e32 + 16 * t, r3);
          if (SAVE) {
            sm_st4(act.R3 + el[c].s32 + 16 * t, r3);
            sm_st4(s2p + el[c].s32 + 16 * t, a2[c][t]);
          }
        }
      }
  } else {
#pragma unroll
    for (int c = 0; c < ET; ++c) {
      float p = 0.0f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) p = fmaf(pre.bl3[t][s2], a2[c][t][s2], p);
      p = sum_xor32(sum_xor16(p));
      const float att = __expf(p + V[V_BA2]);
      if (el[c].valid) {
        if (SAVE) {
#pragma unroll
          for (int t = 0; t < 2; ++t) sm_st4(s2p + el[c].s32 + 16 * t, a2[c][t]);
        }
        if (g == 0) {
          L.ATT[el[c].e] = att;
          if (SAVE) {
            act.ATT[el[c].s1] = att;
            act.DIST[el[c].s1] = d[c];
          }
        }
      }
    }
  }
}

// sbuf [NMX][32]: encoder outputs S of the node rows (written in P1); sdx [NMX][32]: SD of the node rows (read in P4)
template <bool SAVE, int ET>
__device__ __forceinline__ void sm_edge_phase_mfma(const SmLds& L, const SmCfg& cf, const SmAct& act, const float* sbuf, float* sdx,
                                                   const SmEdgeLane (&el)[ET], const SmEdgePre& pre, const SmChainW& cw) {
  const int wv = wave_id();
  if (wv == 1) {
    // ---- self-dynamics, columns = node rows (at most 16: one tile)
    SmMW<2> ws0, ws1;
    smf4 x[2], acc[2], h1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ws0.w[t][0] = pre.w1.w[t][0]; ws0.w[t][1] = pre.w1.w[t][1];
      ws1.w[t][0] = pre.w1.w[2 + t][0]; ws1.w[t][1] = pre.w1.w[2 + t][1];
      x[t] = sm_ld4(sbuf + el[0].node32 + 16 * t);
      acc[t] = pre.bl2[t];
    }
    sm_mfma_layer<2>(ws0, x, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      h1[t] = sm_phi4(acc[t], cf.elu);
      acc[t] = pre.bl3[t];
    }
    sm_mfma_layer<2>(ws1, h1, acc);
    if (el[0].valid) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sm_st4(sdx + el[0].node32 + 16 * t, acc[t] + h1[t]);
        if (SAVE) sm_st4(act.H1 + el[0].node32 + 16 * t, h1[t]);
      }
    }
    return;
  }
  if ((wv != 2 && wv != 3) || cf.N < 2) return;
  sm_edge_chain<SAVE, ET>(L, cf, act, el, pre, cw, wv == 2 ? 1 : 0);
}

// One GNN step.  The lane's node row is r = wave (+ 4 for the upper half-wave of a two-rows-per-wave kernel); lane k of the half
// holds s_in[r][k] in `sinv` (zero beyond sin_dim) and gets RES[o] / PRED[o] of that row back in lane o.  With one row per wave
// the upper half mirrors the lower.  SAVE: write the activation block (act.* valid).
template <bool SAVE, int NMX>
__device__ __forceinline__ void sm_step(const SmLds& L, const SmCfg& cf, float sinv, const SmAct& act, float& res_out, float& pred_out,
                                        const SmEdgeLane (&el)[SmShape<NMX>::ET], const SmChainW& cw) {
  constexpr int RP = SmShape<NMX>::RP, ET = SmShape<NMX>::ET;
  constexpr bool HK = RP == 1;                                      // one row per wave: the half-waves split K (sm_wload / sm_dotw)
  __shared__ __attribute__((aligned(16))) float sbuf[NMX][32];     // S back as broadcast float4 reads (see sm_dotw)
  __shared__ __attribute__((aligned(16))) float sdx[NMX][32];      // SD of every node row, from the self-dynamics wave
  const int wv = wave_id();
  const int lane = lane_id();
  const int o = lane & 31;
  const int N = cf.N;
  const int r = RP == 2 ? wv + 4 * (lane >> 5) : wv;               // the lane's node row
  const bool own = RP == 2 ? r < N : (lane < 32 && r < N);          // one lane per element of a valid row: the stores
  const int rs = r < N ? r : wv;                                    // a row that exists, for the reads of the lanes without one
  const float* V = L.V;
  float S = 0.0f;
  SmW<8> wa, wb;
  sm_stamp(cf, 0);
  // ---- P1: node rows: encoder, factorised first edge layer ------------------------------------------------------
  if (wv < N) {
    wa = sm_wload<8, HK>(L.W + W_ENC, 32, o);
    const float benc = V[V_ENC + o];
    // first quarter of the edge-first weights (k4 = 0, 1; four column groups of 64) while the encoder computes; the other
    // quarters arrive one round ahead of their FMAs (two halves of 16 float4 held 128 registers of weights in a kernel that
    // already parks registers in AGPRs)
    float4 ef[2][4];
#pragma unroll
    for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
      for (int g = 0; g < 4; ++g) ef[k4][g] = *reinterpret_cast<const float4*>(L.W + W_EF + (k4 * 256 + g * 64 + lane) * 4);
    float e;
    if (HK) {
      e = sm_dotw<8, HK>(wa, sinv);          // (s_in is zero beyond sin_dim: the split form always takes the 32-wide dot, 8 packed FMAs a half)
    } else if (cf.sin_dim <= 16) {
      SmW<4> w4;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) w4.w[k4] = wa.w[k4];
      e = sm_dotw<4>(w4, sinv);
    } else {
      e = sm_dotw<8, HK>(wa, sinv);
    }
    S = (o < cf.lim_enc) ? sinv : e + benc;
    if (own) {
      sbuf[r][o] = S;
      if (SAVE) {
        act.SIN[r * 32 + o] = sinv;
        act.S[r * 32 + o] = S;
      }
      if (o < 2) L.POS[r * 4 + o] = S;
    }
    // the 256 first-layer outputs of a row take all 64 lanes: the wave's rows one after the other
#pragma unroll
    for (int rq = 0; rq < RP; ++rq) {
      const int row = wv + 4 * rq;
      if (rq > 0) {
        if (row >= N) break;
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
          for (int g = 0; g < 4; ++g) ef[k4][g] = *reinterpret_cast<const float4*>(L.W + W_EF + (k4 * 256 + g * 64 + lane) * 4);
      }
      float4 sx[8];
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) sx[k4] = *reinterpret_cast<const float4*>(&sbuf[row][4 * k4]);
      v2f p[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 nx[2][4];
        if (q < 3) {
#pragma unroll
          for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
            for (int g = 0; g < 4; ++g) nx[k4][g] = *reinterpret_cast<const float4*>(L.W + W_EF + ((2 * (q + 1) + k4) * 256 + g * 64 + lane) * 4);
        }
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4) {
          const float4 xq = sx[2 * q + k4];
          const v2f x01 = {xq.x, xq.y}, x23 = {xq.z, xq.w};
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            p[g] = pk_fma(v2f{ef[k4][g].x, ef[k4][g].y}, x01, p[g]);
            p[g] = pk_fma(v2f{ef[k4][g].z, ef[k4][g].w}, x23, p[g]);
          }
        }
        if (q < 3) {
#pragma unroll
          for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
            for (int g = 0; g < 4; ++g) ef[k4][g] = nx[k4][g];
        }
      }
      float* pr = L.PR + row * 256 + lane;
      pr[0] = p[0].x + p[0].y;
      pr[64] = p[1].x + p[1].y;
      pr[128] = p[2].x + p[2].y;
      pr[192] = p[3].x + p[3].y;
    }
  }
  // waves without a node row idle in front of the barrier: they fetch their chain's operands there; node waves (on the critical
  // path of P1) fetch theirs behind it
  SmEdgePre pre;
  if (wv >= N) sm_edge_prefetch(L, pre);
  sm_stamp(cf, 1);
  WG_SYNC();
  sm_stamp(cf, 2);
  if (wv < N) sm_edge_prefetch(L, pre);
  // ---- P3: edges (i -> j, i != j) as the columns of the relation chain (wave 3) and the attention chain (wave 2); the
  // self-dynamics of all node rows as the columns of wave 1's
  sm_edge_phase_mfma<SAVE, ET>(L, cf, act, &sbuf[0][0], &sdx[0][0], el, pre, cw);
  if (wv < N) wa = sm_wload<8, HK>(L.W + W_F0, 32, o);        // affector.0, in flight across the barrier
  sm_stamp(cf, 4);
  WG_SYNC();
  sm_stamp(cf, 5);
  // ---- P4: node rows: aggregation, affector, output ------------------------------------------------------------------
  if (wv < N) {
    wb = sm_wload<8, HK>(L.W + W_F1, 32, o);
    const float bf0 = V[V_F0 + o], bf1 = V[V_F1 + o], bf2 = V[V_F2 + o], bo0 = V[V_O0 + o], bo1 = V[V_O1 + o];
    float pred = sdx[rs][o];
    for (int j = 0; j < N; ++j)
      if (j != rs) pred = fmaf(L.R3[(rs * N + j) * 32 + o], L.ATT[rs * N + j], pred);
    sm_stamp(cf, 12);
    const float F1 = fast_tanh(sm_dotw<8, HK>(wa, pred) + bf0);
    sm_stamp(cf, 13);
    wa = sm_wload<8, HK>(L.W + W_F2, 32, o);
    const float F2 = fast_tanh(sm_dotw<8, HK>(wb, F1) + bf1) + F1;
    wb = sm_wload<8, HK>(L.W + W_O0, 32, o);
    const float F3 = sm_dotw<8, HK>(wa, F2) + bf2;
    wa = sm_wload<8, HK>(L.W + W_O0, 32, o, 8);
    float t = sm_dotw<8, HK>(wb, F3);
    wb = sm_wload<8, HK>(L.W + W_O1, 32, o);
    t += sm_dotw<8, HK>(wa, S);
    const float O1 = fast_tanh(t + bo0);
    sm_stamp(cf, 14);
    const float RES = sm_dotw<8, HK>(wb, O1) + bo1 + O1;
    sm_stamp(cf, 15);
    if (SAVE && own) {
      act.PRED[r * 32 + o] = pred;
      act.F1[r * 32 + o] = F1;
      act.F2[r * 32 + o] = F2;
      act.F3[r * 32 + o] = F3;
      act.O1[r * 32 + o] = O1;
      act.RES[r * 32 + o] = RES;
    }
    res_out = RES;
    pred_out = pred;
  }
  sm_stamp(cf, 6);
}

__device__ __forceinline__ float sm_from_lane(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane * 4, __builtin_bit_cast(int, v)));
}

// =================================================================================================
// inference recursion, same contract as dyn_loop_fwd_k (gnn.hip) with G = 1: grid = B sequences
// =================================================================================================
// SAVEM: 0 = inference, 2 = save the activations as per-sequence streams (gnn_small_bwd.hip); NT > 0: the number of objects at
// compile time (3: the headline shape); NMX: 4 or 6 (SmShape)
template <int SAVEM, int NMX, bool ELU, int NT, bool STAMP = false>      // STAMP: the phase stamps of tools/loop_stamps.py (one debug instantiation)
__global__ __launch_bounds__(64 * kSmWaves) void dyn_loop_fwd_small_k(
    const float* __restrict__ z1, const float* __restrict__ zsup, const float* __restrict__ zsstd,
    const float* __restrict__ eps, const float* __restrict__ extra, const float* __restrict__ P,
    float* __restrict__ z, float* __restrict__ zdyn, float* __restrict__ zdstd, float* __restrict__ mean,
    float* __restrict__ stdv, float* __restrict__ pred, float* __restrict__ act,
    int B, int Ts, int N, int sin_dim, int lim_enc, int elu, LoopConst kc, long long* stamps, int ts0, int ts1) {
  // steps [ts0, ts1) of the Ts the tensors are laid out for: a caller that pipelines the recursion against the scene likelihood of
  // the frames already inferred runs it in pieces; a piece that does not start at 0 takes the state the previous one left in z
  constexpr bool SAVE = SAVEM != 0;
  constexpr int RP = SmShape<NMX>::RP, ET = SmShape<NMX>::ET;
  static_assert(SAVEM == 0 || SAVEM == 2, "no saved activations, or the streams layout");
  static_assert(NT <= NMX, "object count beyond what the kernel is built for");
  if (!STAMP) stamps = nullptr;      // the 16 stamp sites of a step vanish (a run-time null check each was ~50 instructions per step)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const SmLds L = sm_carve<NMX>(lds);
  const int b = blockIdx.x;
  const int wv = wave_id(), lane = lane_id(), l = lane & 31;
  elu = ELU ? 1 : 0;      // compile-time activation: with a run-time flag every phi carried the ocml expm1f path (code, registers, branches)
  if (NT > 0) N = NT;       // loops over the objects unroll, their LDS reads go out together
  SmCfg cf{N, sin_dim, lim_enc, elu};
  cf.stamps = nullptr;
  const int E = sin_dim - 16;
  SmEdgeLane el[ET];
#pragma unroll
  for (int t = 0; t < ET; ++t) el[t] = sm_edge_lane(N, SAVE, t);
  sm_setup(L, P);
  // the lane's node row r: lane l of its half holds s_in[r][l]; dims 0..15 come from the running state z[t-1][2..17]
  float sinv = 0.0f;
  const int r = RP == 2 ? wv + 4 * (lane >> 5) : wv;
  const bool row = r < N;                                           // lanes of the upper half of a one-row wave mirror the lower
  const bool own = RP == 2 ? row : (lane < 32 && row);              // one lane per element of a valid row: the stores
  if (row) {
    if (l < 16) sinv = ts0 == 0 ? z1[((size_t)b * N + r) * 18 + 2 + l] : z[(((size_t)b * Ts + ts0 - 1) * N + r) * 18 + 2 + l];
    else if (l < sin_dim) sinv = extra[(((size_t)b * Ts + ts0) * N + r) * E + (l - 16)];
  }
  WG_SYNC();
  SmChainW cw;
  sm_chain_wbuild(L, cw);       // the edge chains' weight fragments: registers, for all time steps
  for (int ts = ts0; ts < ts1; ++ts) {
    const size_t o = ((size_t)b * Ts + ts) * N + r;
    // this step's epilogue inputs and the next step's extra inputs: issued now, consumed ~2 us later
    float ep = 0.0f, ms = 0.0f, ss = 1.0f, xnext = 0.0f;
    if (row) {
      if (l < 16) ep = eps[o * 18 + 2 + l];
      else if (l < 18) ep = eps[o * 18 + (l - 16)];
      if (l < 4) {
        ms = zsup[o * 6 + 2 + l];
        ss = zsstd[o * 6 + 2 + l];
      } else if (l >= 16 && l < 18) {
        ms = zsup[o * 6 + (l - 16)];
        ss = zsstd[o * 6 + (l - 16)];
      }
      if (l >= 16 && l < sin_dim && ts + 1 < Ts) xnext = extra[(((size_t)b * Ts + ts + 1) * N + r) * E + (l - 16)];
    }
    SmAct a{};
    if (SAVE) a = sm_act2(act + (size_t)b * sm_act2_floats(N, Ts), N, Ts, ts);
    cf.stamps = (ts == ts1 - 1) ? stamps : nullptr;
    float res = 0.0f, prd = 0.0f;
    sm_step<SAVE, NMX>(L, cf, sinv, a, res, prd, el, cw);
    if (wv < N) {
      // epilogue (stove.py:103-170 + constrain_z_dyn): lane d < 16 owns state dim d, lanes 16/17 the two scale dims
      const float res_s = from_xor16(res);      // RES[16 + d] for d < 16
      float zv;
      if (l < 16) {
        const int d = l;
        // v_rcp_f32 / v_rsq_f32 (1 ulp) for the two sigmoids' and the product-of-Gaussians' divisions: four IEEE divisions and a
        // square root were ~40 instructions of the step's serial tail (the backward's epilogue has had them since round 2)
        const float m = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-res)) - 1.0f;
        const float sd = std_scale(d, kc) * __builtin_amdgcn_rcpf(1.0f + __expf(-res_s));
        const float zd = m + (d < 2 ? sinv : 0.0f);
        float mu, sg;
        if (d < 4) {
          const float sd2 = sd * sd, ss2 = ss * ss, D = sd2 + ss2;
          mu = (ss2 * zd + sd2 * ms) * __builtin_amdgcn_rcpf(D);
          sg = sd * ss * __builtin_amdgcn_rsqf(D);
        } else {
          mu = zd;
          sg = sd;
        }
        zv = fmaf(sg, ep, mu);
        if (own) {
          zdyn[o * 16 + d] = zd;
          zdstd[o * 16 + d] = sd;
          z[o * 18 + 2 + d] = zv;
          mean[o * 18 + 2 + d] = mu;
          stdv[o * 18 + 2 + d] = sg;
        }
      } else {
        zv = xnext;                                   // becomes s_in[l] of the next step (0 beyond sin_dim)
        if (l < 18 && own) {
          const int q = l - 16;
          z[o * 18 + q] = fmaf(ss, ep, ms);
          mean[o * 18 + q] = ms;
          stdv[o * 18 + q] = ss;
        }
      }
      if (pred != nullptr && own) pred[o * 32 + l] = prd;
      sinv = zv;
    }
    sm_stamp(cf, 7);
  }
}

// =================================================================================================
// generative rollout, same contract as rollout_fwd_k (gnn.hip) with G = 1
// =================================================================================================
template <int NMX, bool ELU, int NT>
__global__ __launch_bounds__(64 * kSmWaves) void rollout_fwd_small_k(const float* __restrict__ z_last, const float* __restrict__ extra,
                                                           const float* __restrict__ P, float* __restrict__ z_pred,
                                                           float* __restrict__ zstd, float* __restrict__ pred,
                                                           int B, int num, int A, int N, int sin_dim, int lim_enc, int elu, LoopConst kc) {
  constexpr int RP = SmShape<NMX>::RP, ET = SmShape<NMX>::ET;
  static_assert(NT <= NMX, "object count beyond what the kernel is built for");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const SmLds L = sm_carve<NMX>(lds);
  const int b = blockIdx.x;
  const int wv = wave_id(), lane = lane_id(), l = lane & 31;
  elu = ELU ? 1 : 0;      // compile-time activation: with a run-time flag every phi carried the ocml expm1f path (code, registers, branches)
  if (NT > 0) N = NT;       // loops over the objects unroll, their LDS reads go out together
  SmCfg cf{N, sin_dim, lim_enc, elu};
  cf.stamps = nullptr;
  const int E = sin_dim - 16;
  SmEdgeLane el[ET];
#pragma unroll
  for (int t = 0; t < ET; ++t) el[t] = sm_edge_lane(N, 0, t);
  sm_setup(L, P);
  float sinv = 0.0f, scale = 0.0f;
  const int r = RP == 2 ? wv + 4 * (lane >> 5) : wv;
  const bool row = r < N;
  const bool own = RP == 2 ? row : (lane < 32 && row);
  if (row) {
    if (l < 16) sinv = z_last[((size_t)b * N + r) * 18 + 2 + l];
    else if (l < sin_dim) sinv = extra[(((size_t)b * A + 0) * N + r) * E + (l - 16)];
    if (l >= 16 && l < 18) scale = z_last[((size_t)b * N + r) * 18 + (l - 16)];       // sx, sy stay constant
  }
  WG_SYNC();
  SmChainW cw;
  sm_chain_wbuild(L, cw);
  for (int t = 0; t < num; ++t) {
    const size_t o = ((size_t)b * num + t) * N + r;
    float xnext = 0.0f;
    if (row && l >= 16 && l < sin_dim && t + 1 < num) xnext = extra[(((size_t)b * A + ((t + 1) % A)) * N + r) * E + (l - 16)];
    SmAct a{};
    float res = 0.0f, prd = 0.0f;
    sm_step<false, NMX>(L, cf, sinv, a, res, prd, el, cw);
    if (wv < N) {
      const float res_s = from_xor16(res);
      float zv;
      if (l < 16) {
        zv = 2.0f * sigmoidf_(res) - 1.0f + (l < 2 ? sinv : 0.0f);
        if (own) {
          z_pred[o * 18 + 2 + l] = zv;
          if (zstd != nullptr) zstd[o * 16 + l] = std_scale(l, kc) * sigmoidf_(res_s);
        }
      } else {
        zv = xnext;
        if (l < 18 && own) z_pred[o * 18 + (l - 16)] = scale;
      }
      if (pred != nullptr && own) pred[o * 32 + l] = prd;
      sinv = zv;
    }
  }
}

}  // namespace stove
