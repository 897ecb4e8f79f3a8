// Object RAT-SPN (depth-2 random region graph) -- forward / backward kernels for gfx950.
//
// Replaces the ATen op chains of the reference's RatSpn.forward for the object SPN
// (model/spn/rat_torch.py:83-109 leaf, :147-163 product, :202-222 sum, :333-357 sweep)
// as called from Supair.likelihood (model/video_prediction/supair.py:71-76).
//
// Mapping: one lane = one sample (glimpse patch), 64 samples per wave ("batch").  All
// structure and parameter reads are wave-uniform, so hipcc serves them from the scalar
// cache (s_load) and the VALU sees them as SGPR operands; per-lane data is only the
// patch pixels, read coalesced from a [batch][pixel][x|w][64] tile.
// A workgroup = 2R waves = (replica r, side) pairs of one batch; the two sides of a
// replica meet through LDS for the root product, the R replicas for the root sum.
//
// Leaf in expanded form: sum_p w_p (a x_p^2 + b x_p + c),  a=-1/(2v), b=mu/v,
// c=-mu^2/(2v) - 0.5 log(2 pi v)   (3 FMAs per pixel x gaussian).
// Sum layer as a bilinear form on max-shifted exponentials with linear softmax weights:
//   out_s = m1 + m2 + log sum_{j2,j1} E1[j1] E2[j2] W[j2*G+j1][s]   (20 exps instead of 1000).
#include "common.h"

namespace stove {

template <int N>
__device__ __forceinline__ float vmax(const float (&v)[N]) {
  float m = v[0];
#pragma unroll
  for (int i = 1; i < N; ++i) m = fmaxf(m, v[i]);
  return m;
}

// ---- wave-uniform weight tables -------------------------------------------------------------------
// The sum-layer weights of a (replica, side) are wave-uniform and each is used once per batch.  Three ways to feed them:
//  * plain scalar loads: the compiler hoists all ~1000 out of the batch loop, runs out of SGPRs and parks them in VGPR
//    lanes -- a v_writelane and a v_readlane per weight around ~2700 useful FMAs (objspn_bwd_k: 0.51 ms);
//  * lane-distributed in VGPRs (LaneTable): one v_readlane (+ a hazard s_nop) per weight feeding the FMA as an SGPR
//    operand: 2 VALU slots per weight (0.13 ms);
//  * the wave's table in LDS (used for the K-wide rows of the sum nodes): broadcast ds_read_b64 of two neighbouring
//    weights straight into a register pair, one v_pk_fma_f32 per pair: 0.5 VALU slots per weight, the reads on the LDS port.
template <int N>
struct LaneTable {
  float v[(N + 63) / 64];
  __device__ __forceinline__ void load(const float* __restrict__ p, int lane) {
#pragma unroll
    for (int i = 0; i < (N + 63) / 64; ++i) v[i] = (i * 64 + lane < N) ? p[i * 64 + lane] : 0.0f;
  }
  __device__ __forceinline__ float at(int idx) const {      // idx must fold to a constant (fully unrolled callers)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[idx >> 6]), idx & 63));
  }
};

// copy a wave-uniform table of N floats (N even) into the wave's LDS slice
template <int N>
__device__ __forceinline__ void lds_table_load(float* dst, const float* __restrict__ src, int lane) {
#pragma unroll
  for (int i = 0; i < (N + 63) / 64; ++i)
    if (i * 64 + lane < N) dst[i * 64 + lane] = src[i * 64 + lane];
}
__device__ __forceinline__ v2f lds_pair(const float* t, int idx) { return *reinterpret_cast<const v2f*>(t + idx); }

// ---- shared piece: leaves + sum node of one (replica, side) -------------------------------
template <int S, int G, int K>
struct SideState {
  float E1[G], E2[G];   // exp(leaf - max)
  float acc[K];         // sum_k E1 E2 W[k][s]
  float o[K];           // sum node outputs
};

template <int S, int G, int K>
__device__ __forceinline__ void side_forward(const float* __restrict__ tile, int lane,
                                             const int* __restrict__ scope, const float* __restrict__ coef,
                                             const float* W, SideState<S, G, K>& st) {   // W: the wave's LDS table
  float ell[2][G];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.0f;
    const int* sc = scope + l * S;
    const float* cf = coef + l * S * G * 3;
    // unrolled: as a rolled loop every pixel paid two dependent load latencies (scope index, then x / w) back to back
#pragma unroll 5
    for (int i = 0; i < S; ++i) {
      const int p = sc[i];
      const float x = tile[(p * 2) * 64 + lane];
      const float w = tile[(p * 2 + 1) * 64 + lane];
      const float wx = w * x, wxx = wx * x;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float* c3 = cf + (i * G + g) * 3;
        acc[g] = fmaf(wxx, c3[0], acc[g]);
        acc[g] = fmaf(wx, c3[1], acc[g]);
        acc[g] = fmaf(w, c3[2], acc[g]);
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) ell[l][g] = acc[g];
  }
  const float m1 = vmax<G>(ell[0]), m2 = vmax<G>(ell[1]);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    st.E1[g] = __expf(ell[0][g] - m1);
    st.E2[g] = __expf(ell[1][g] - m2);
  }
  static_assert(K % 2 == 0, "sum nodes are processed in pairs");
  v2f acc2[K / 2];
#pragma unroll
  for (int s = 0; s < K / 2; ++s) acc2[s] = v2f{0.0f, 0.0f};
#pragma unroll
  for (int j2 = 0; j2 < G; ++j2) {
#pragma unroll
    for (int j1 = 0; j1 < G; ++j1) {
      const float t = st.E1[j1] * st.E2[j2];
      const v2f tt = {t, t};
#pragma unroll
      for (int s = 0; s < K / 2; ++s) acc2[s] = pk_fma(tt, lds_pair(W, (j2 * G + j1) * K + 2 * s), acc2[s]);
    }
  }
#pragma unroll
  for (int s = 0; s < K / 2; ++s) {
    st.acc[2 * s] = acc2[s].x;
    st.acc[2 * s + 1] = acc2[s].y;
  }
#pragma unroll
  for (int s = 0; s < K; ++s) st.o[s] = m1 + m2 + __logf(st.acc[s]);
}

// ---- root pieces shared by the forward, the backward and the fused forward + unit backward ------------------------
// replica root (side-0 wave): sum_{j2, j1} EA[j1] EB[j2] wroot[j2 K + j1] on max-shifted exponentials
template <int K>
__device__ __forceinline__ float root_pair_sum(const float (&EA)[K], const float (&EB)[K], const float* __restrict__ wr) {
  float sr = 0.0f;
#pragma unroll
  for (int j2 = 0; j2 < K; ++j2)
#pragma unroll
    for (int j1 = 0; j1 < K; ++j1) sr = fmaf(EA[j1] * EB[j2], wr[j2 * K + j1], sr);
  return sr;
}
// root of the lane's sample from the R replicas' (max, sum) pairs in LDS: log-density = M + log Z.  The backward takes a
// replica's share of the root as exp(M_r - M) / Z from the SAME rounded M_r, M and Z (M_r - M is an exact subtraction), not as
// exp(M_r - value): the value's rounding (6e-8 of a log-density of 10^2..10^4) would sit on every gradient of the sample as a
// common relative error (round 5, 'stress' weight regime; spn_bg.hip bgspn_root_bwd_k has the numbers).
struct RootMZ {
  float M, Z;
  __device__ __forceinline__ float value() const { return M + __logf(Z); }
  __device__ __forceinline__ float share(float Mr) const { return __expf(Mr - M) / Z; }
};
template <int R>
__device__ __forceinline__ RootMZ root_mz(const float* part, int lane) {
  float M = part[lane];
#pragma unroll
  for (int q = 1; q < R; ++q) M = fmaxf(M, part[(q * 2) * 64 + lane]);
  float Z = 0.0f;
#pragma unroll
  for (int q = 0; q < R; ++q) Z = fmaf(part[(q * 2 + 1) * 64 + lane], __expf(part[(q * 2) * 64 + lane] - M), Z);
  return RootMZ{M, Z};
}
template <int R>
__device__ __forceinline__ float root_value(const float* part, int lane) { return root_mz<R>(part, lane).value(); }
// mean over the pixels of (1 - w): the overlap statistic of the glimpse
template <int D>
__device__ __forceinline__ float tile_overlap(const float* __restrict__ tile, int lane) {
  float s1 = 0.0f;
  for (int p = 0; p < D; ++p) s1 += 1.0f - tile[(p * 2 + 1) * 64 + lane];
  return s1 * (1.0f / D);
}

// ---- forward -------------------------------------------------------------------------------
// out[sample] = root log-density; ovl[sample] = mean over pixels of (1 - w) (optional).
template <int R, int S, int G, int K>
__global__ __launch_bounds__(128 * R) void objspn_fwd_k(
    const float* __restrict__ xw, const int* __restrict__ scope, const float* __restrict__ coef,
    const float* __restrict__ wsum, const float* __restrict__ wroot,
    float* __restrict__ out, float* __restrict__ ovl, int n_samples, int n_batches) {
  constexpr int D = 4 * S;
  __shared__ float xch[R * K * 64];        // side 1 -> side 0 of every replica (66 KB of LDS in all: two workgroups per CU)
  __shared__ float part[R * 2 * 64];
  __shared__ __attribute__((aligned(16))) float wtab[R * 2][G * G * K];
  const int lane = lane_id();
  const int wv = wave_id();
  const int r = wv >> 1, side = wv & 1;
  const float* W = wtab[wv];
  lds_table_load<G * G * K>(wtab[wv], wsum + (size_t)(r * 2 + side) * G * G * K, lane);
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    const float* tile = xw + (size_t)b * (D * 2 * 64);
    SideState<S, G, K> st;
    side_forward<S, G, K>(tile, lane, scope + (r * 4 + side * 2) * S, coef + (size_t)(r * 4 + side * 2) * S * G * 3, W, st);
    if (side == 1)
#pragma unroll
      for (int s = 0; s < K; ++s) xch[(r * K + s) * 64 + lane] = st.o[s];
    __syncthreads();
    if (side == 0) {
      float oB[K];
#pragma unroll
      for (int s = 0; s < K; ++s) oB[s] = xch[(r * K + s) * 64 + lane];
      const float mA = vmax<K>(st.o), mB = vmax<K>(oB);
      float EA[K], EB[K];
#pragma unroll
      for (int s = 0; s < K; ++s) {
        EA[s] = __expf(st.o[s] - mA);
        EB[s] = __expf(oB[s] - mB);
      }
      part[(r * 2) * 64 + lane] = mA + mB;
      part[(r * 2 + 1) * 64 + lane] = root_pair_sum<K>(EA, EB, wroot + r * K * K);
    } else if (wv == 1 && ovl != nullptr) {
      const int smp = b * 64 + lane;
      const float v = tile_overlap<D>(tile, lane);
      if (smp < n_samples) ovl[smp] = v;
    }
    __syncthreads();
    if (wv == 0) {
      const int smp = b * 64 + lane;
      const float ro = root_value<R>(part, lane);
      if (smp < n_samples) out[smp] = ro;
    }
    // next iteration overwrites xch only after its own leaf sweep + barrier; part is re-read
    // only by wave 0 before it reaches the next first barrier, so no extra barrier is needed
    // for xch, but part needs one:
    __syncthreads();
  }
}

// ---- MPE walk (Supair.spn_mpe, supair.py:382-424; RatSpn.reconstruct, rat_torch.py:359-372 with the node walks of
// :126-135, :177-183, :224-229).  Per sample: every sum node takes argmax_k(child_k + log w_k) -- taken here on the
// max-shifted linear products E1 E2 W, which order the same way -- then the walk from the root picks one replica, one sum
// node per side and one Gaussian per leaf; the output is those components' means on the leaf scopes, clamped to [0, 1].
// Ties resolve to the first child in the reference's concatenation order (np.argmax): replica-major, then j2*n1 + j1.
// mu [R*4][S][G] leaf means in the order of `scope`; out [n][4S]; pick [n][5] = (replica, comps of the 4 leaves) or null.
template <int R, int S, int G, int K>
__global__ __launch_bounds__(128 * R) void objspn_mpe_k(
    const float* __restrict__ xw, const int* __restrict__ scope, const float* __restrict__ coef,
    const float* __restrict__ wsum, const float* __restrict__ wroot, const float* __restrict__ mu,
    float* __restrict__ out, int* __restrict__ pick, int n_samples, int n_batches) {
  constexpr int D = 4 * S;
  __shared__ float xch[R * 2 * K * 64];
  __shared__ unsigned char best_pair[R * 2 * K * 64];   // per (replica, side) sum node: winning j2*G+j1
  __shared__ float root_val[R * 64];
  __shared__ int root_arg[R * 64];
  __shared__ int chosen[64];                            // r*K*K + j2*K + j1 of the root's winner
  __shared__ __attribute__((aligned(16))) float wtab[R * 2][G * G * K];
  const int lane = lane_id();
  const int wv = wave_id();
  const int r = wv >> 1, side = wv & 1;
  const float* W = wtab[wv];
  lds_table_load<G * G * K>(wtab[wv], wsum + (size_t)(r * 2 + side) * G * G * K, lane);
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    const float* tile = xw + (size_t)b * (D * 2 * 64);
    SideState<S, G, K> st;
    side_forward<S, G, K>(tile, lane, scope + (r * 4 + side * 2) * S, coef + (size_t)(r * 4 + side * 2) * S * G * 3, W, st);
    {
      float bv[K];
      int bk[K];
#pragma unroll
      for (int s = 0; s < K; ++s) {
        bv[s] = -1.0f;
        bk[s] = 0;
      }
#pragma unroll
      for (int j2 = 0; j2 < G; ++j2)
#pragma unroll
        for (int j1 = 0; j1 < G; ++j1) {
          const float e = st.E1[j1] * st.E2[j2];
#pragma unroll
          for (int s2 = 0; s2 < K / 2; ++s2) {
            const v2f w = lds_pair(W, (j2 * G + j1) * K + 2 * s2);
            const float t0 = e * w.x, t1 = e * w.y;
            if (t0 > bv[2 * s2]) { bv[2 * s2] = t0; bk[2 * s2] = j2 * G + j1; }
            if (t1 > bv[2 * s2 + 1]) { bv[2 * s2 + 1] = t1; bk[2 * s2 + 1] = j2 * G + j1; }
          }
        }
#pragma unroll
      for (int s = 0; s < K; ++s) {
        best_pair[((r * 2 + side) * K + s) * 64 + lane] = (unsigned char)bk[s];
        xch[((r * 2 + side) * K + s) * 64 + lane] = st.o[s];
      }
    }
    __syncthreads();
    if (side == 0) {
      float oB[K];
#pragma unroll
      for (int s = 0; s < K; ++s) oB[s] = xch[((r * 2 + 1) * K + s) * 64 + lane];
      const float mA = vmax<K>(st.o), mB = vmax<K>(oB);
      const float* wr = wroot + r * K * K;
      float bv = -1.0f;
      int bk = 0;
#pragma unroll
      for (int j2 = 0; j2 < K; ++j2) {
        const float eb = __expf(oB[j2] - mB);
#pragma unroll
        for (int j1 = 0; j1 < K; ++j1) {
          const float t = __expf(st.o[j1] - mA) * eb * wr[j2 * K + j1];
          if (t > bv) { bv = t; bk = j2 * K + j1; }
        }
      }
      root_val[r * 64 + lane] = mA + mB + __logf(bv);
      root_arg[r * 64 + lane] = bk;
    }
    __syncthreads();
    if (wv == 0) {
      float bv = root_val[lane];
      int br = 0;
#pragma unroll
      for (int q = 1; q < R; ++q) {
        const float v = root_val[q * 64 + lane];
        if (v > bv) { bv = v; br = q; }
      }
      chosen[lane] = br * K * K + root_arg[br * 64 + lane];
    }
    __syncthreads();
    const int smp = b * 64 + lane;
    const int ch = chosen[lane];
    const int rs = ch / (K * K), nB = (ch % (K * K)) / K, nA = ch % K;     // replica, side-B node (row), side-A node (col)
    const int pA = best_pair[((rs * 2) * K + nA) * 64 + lane], pB = best_pair[((rs * 2 + 1) * K + nB) * 64 + lane];
    const int comp[4] = {pA % G, pA / G, pB % G, pB / G};                    // leaf = side * 2 + (0: in1 = col, 1: in2 = row)
    if (smp < n_samples) {
      for (int q = wv; q < D; q += 2 * R) {
        const int L = q / S, i = q % S;
        const int g = L == 0 ? comp[0] : L == 1 ? comp[1] : L == 2 ? comp[2] : comp[3];
        const int cell = (rs * 4 + L) * S + i;
        out[(size_t)smp * D + scope[cell]] = fminf(fmaxf(mu[(size_t)cell * G + g], 0.0f), 1.0f);
      }
      if (wv == 0 && pick != nullptr) {
        int* pk = pick + (size_t)smp * 5;
        pk[0] = rs;
        pk[1] = comp[0], pk[2] = comp[1], pk[3] = comp[2], pk[4] = comp[3];
      }
    }
    __syncthreads();
  }
}

// ---- backward (main): recompute forward, back-propagate to the leaf outputs ---------------
// Writes   Dscr[batch][r][4][G][64]        dL/d leaf log-densities
//          Sscr[batch][r*2+side][K+2G][64] gamma[s] = g_s / acc_s, E1[G], E2[G]   (sum-weight grads)
//          Rscr[batch][r][1+2K][64]        rho_r, EA[K], EB[K]                     (root-weight grads)
// Everything below a sample's root is LINEAR in its upstream gradient go: rho, gamma and the leaf gradients carry one factor
// go each, E1 / E2 / EA / EB none.  side_backward is handed rho = go exp(mO + mP - root) and does the rest.
template <int R, int S, int G, int K>
__device__ __forceinline__ void side_backward(const SideState<S, G, K>& st, const float (&EO)[K], const float (&EP)[K], float rho,
                                              int b, int r, int side, int lane, const float* W, const LaneTable<K * K>& WR,
                                              float* __restrict__ Dscr, float* __restrict__ Sscr, float* __restrict__ Rscr) {
  // g[s] = dL/d o[s] of this side
  float g[K];
#pragma unroll
  for (int s = 0; s < K; ++s) g[s] = 0.0f;
  if (side == 0) {
#pragma unroll
    for (int j2 = 0; j2 < K; ++j2)
#pragma unroll
      for (int j1 = 0; j1 < K; ++j1) g[j1] = fmaf(EP[j2], WR.at(j2 * K + j1), g[j1]);
  } else {
#pragma unroll
    for (int j2 = 0; j2 < K; ++j2)
#pragma unroll
      for (int j1 = 0; j1 < K; ++j1) g[j2] = fmaf(EP[j1], WR.at(j2 * K + j1), g[j2]);
  }
  float gam[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    g[s] *= rho * EO[s];
    gam[s] = g[s] / st.acc[s];
  }
  float d1[G], d2[G];
#pragma unroll
  for (int j = 0; j < G; ++j) d1[j] = d2[j] = 0.0f;
  v2f gam2[K / 2];
#pragma unroll
  for (int s = 0; s < K / 2; ++s) gam2[s] = v2f{gam[2 * s], gam[2 * s + 1]};
#pragma unroll
  for (int j2 = 0; j2 < G; ++j2) {
#pragma unroll
    for (int j1 = 0; j1 < G; ++j1) {
      v2f tp = gam2[0] * lds_pair(W, (j2 * G + j1) * K);
#pragma unroll
      for (int s = 1; s < K / 2; ++s) tp = pk_fma(gam2[s], lds_pair(W, (j2 * G + j1) * K + 2 * s), tp);
      const float t = tp.x + tp.y;
      d1[j1] = fmaf(st.E2[j2], t, d1[j1]);
      d2[j2] = fmaf(st.E1[j1], t, d2[j2]);
    }
  }
  float* Dp = Dscr + ((size_t)(b * R + r) * 4 + side * 2) * G * 64;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    Dp[j * 64 + lane] = d1[j] * st.E1[j];
    Dp[(G + j) * 64 + lane] = d2[j] * st.E2[j];
  }
  float* Sp = Sscr + (size_t)(b * R * 2 + r * 2 + side) * (K + 2 * G) * 64;
#pragma unroll
  for (int s = 0; s < K; ++s) Sp[s * 64 + lane] = gam[s];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    Sp[(K + j) * 64 + lane] = st.E1[j];
    Sp[(K + G + j) * 64 + lane] = st.E2[j];
  }
  if (side == 0) {
    float* Rp = Rscr + (size_t)(b * R + r) * (1 + 2 * K) * 64;
    Rp[lane] = rho;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      Rp[(1 + s) * 64 + lane] = EO[s];
      Rp[(1 + K + s) * 64 + lane] = EP[s];
    }
  }
}

// dout[sample] = dL/d root, out[sample] = root value saved by the forward.
template <int R, int S, int G, int K>
__global__ __launch_bounds__(128 * R) void objspn_bwd_k(
    const float* __restrict__ xw, const int* __restrict__ scope, const float* __restrict__ coef,
    const float* __restrict__ wsum, const float* __restrict__ wroot,
    const float* __restrict__ out, const float* __restrict__ dout,
    float* __restrict__ Dscr, float* __restrict__ Sscr, float* __restrict__ Rscr, int n_samples, int n_batches) {
  constexpr int D = 4 * S;
  __shared__ float xch[R * 2 * K * 64];
  __shared__ float part[R * 2 * 64];
  const int lane = lane_id();
  const int wv = wave_id();
  const int r = wv >> 1, side = wv & 1;
  __shared__ __attribute__((aligned(16))) float wtab[R * 2][G * G * K];
  const float* W = wtab[wv];
  lds_table_load<G * G * K>(wtab[wv], wsum + (size_t)(r * 2 + side) * G * G * K, lane);
  LaneTable<K * K> WR;
  WR.load(wroot + r * K * K, lane);
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    const float* tile = xw + (size_t)b * (D * 2 * 64);
    SideState<S, G, K> st;
    side_forward<S, G, K>(tile, lane, scope + (r * 4 + side * 2) * S, coef + (size_t)(r * 4 + side * 2) * S * G * 3, W, st);
#pragma unroll
    for (int s = 0; s < K; ++s) xch[((r * 2 + side) * K + s) * 64 + lane] = st.o[s];
    __syncthreads();
    float oP[K];
#pragma unroll
    for (int s = 0; s < K; ++s) oP[s] = xch[((r * 2 + (1 - side)) * K + s) * 64 + lane];
    const float mO = vmax<K>(st.o), mP = vmax<K>(oP);
    float EO[K], EP[K];   // own side, partner side
#pragma unroll
    for (int s = 0; s < K; ++s) {
      EO[s] = __expf(st.o[s] - mO);
      EP[s] = __expf(oP[s] - mP);
    }
    const int smp = b * 64 + lane;
    const bool live = smp < n_samples;
    // the root's (M, Z) once more (as objspn_fwd_unit_k forms them) instead of the forward's saved value `out`: see RootMZ
    if (side == 0) {
      part[(r * 2) * 64 + lane] = mO + mP;
      part[(r * 2 + 1) * 64 + lane] = root_pair_sum<K>(EO, EP, wroot + r * K * K);
    }
    __syncthreads();
    const float rho = live ? dout[smp] * root_mz<R>(part, lane).share(mO + mP) : 0.0f;
    side_backward<R, S, G, K>(st, EO, EP, rho, b, r, side, lane, W, WR, Dscr, Sscr, Rscr);
    __syncthreads();   // xch / part reuse
  }
}

// ---- forward + backward at UNIT upstream gradient in one pass (the training path of the scene likelihood) ----------------
// The root value is formed inside the workgroup, so the backward of a sample can follow its forward while the leaf
// exponentials and the sum-node accumulators are still in registers: no forward-state dump (1 920 B per glimpse written and
// read back), no second sweep over the tile, one launch less on the step's critical path.  What the upstream gradient will
// be is not known yet -- it does not have to be: Dscr / gamma / rho are written for go = 1 and their consumers
// (scene_pixtile_bwd_k, objspn_tablegrad*_k) multiply by go[sample] while they stage them.  `out` is computed by the same
// code as objspn_fwd_k's: bit-identical between the two.
template <int R, int S, int G, int K>
__global__ __launch_bounds__(128 * R) void objspn_fwd_unit_k(
    const float* __restrict__ xw, const int* __restrict__ scope, const float* __restrict__ coef,
    const float* __restrict__ wsum, const float* __restrict__ wroot,
    float* __restrict__ out, float* __restrict__ ovl,
    float* __restrict__ Dscr, float* __restrict__ Sscr, float* __restrict__ Rscr, int n_samples, int n_batches) {
  constexpr int D = 4 * S;
  __shared__ float xch[R * 2 * K * 64];
  __shared__ float part[R * 2 * 64];
  __shared__ __attribute__((aligned(16))) float wtab[R * 2][G * G * K];
  const int lane = lane_id();
  const int wv = wave_id();
  const int r = wv >> 1, side = wv & 1;
  const float* W = wtab[wv];
  lds_table_load<G * G * K>(wtab[wv], wsum + (size_t)(r * 2 + side) * G * G * K, lane);
  LaneTable<K * K> WR;
  WR.load(wroot + r * K * K, lane);
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    const float* tile = xw + (size_t)b * (D * 2 * 64);
    SideState<S, G, K> st;
    side_forward<S, G, K>(tile, lane, scope + (r * 4 + side * 2) * S, coef + (size_t)(r * 4 + side * 2) * S * G * 3, W, st);
#pragma unroll
    for (int s = 0; s < K; ++s) xch[((r * 2 + side) * K + s) * 64 + lane] = st.o[s];
    __syncthreads();
    float oP[K];
#pragma unroll
    for (int s = 0; s < K; ++s) oP[s] = xch[((r * 2 + (1 - side)) * K + s) * 64 + lane];
    const float mO = vmax<K>(st.o), mP = vmax<K>(oP);
    float EO[K], EP[K];   // own side, partner side
#pragma unroll
    for (int s = 0; s < K; ++s) {
      EO[s] = __expf(st.o[s] - mO);
      EP[s] = __expf(oP[s] - mP);
    }
    const int smp = b * 64 + lane;
    const bool live = smp < n_samples;
    if (side == 0) {
      part[(r * 2) * 64 + lane] = mO + mP;
      part[(r * 2 + 1) * 64 + lane] = root_pair_sum<K>(EO, EP, wroot + r * K * K);
    } else if (wv == 1 && ovl != nullptr) {
      const float v = tile_overlap<D>(tile, lane);
      if (live) ovl[smp] = v;
    }
    __syncthreads();      // (also orders this batch's xch reads before the next batch's xch writes)
    const RootMZ rt = root_mz<R>(part, lane);         // every wave for itself: the same arithmetic as wave 0's, no third barrier
    if (wv == 0 && live) out[smp] = rt.value();
    const float rho = live ? rt.share(mO + mP) : 0.0f;
    side_backward<R, S, G, K>(st, EO, EP, rho, b, r, side, lane, W, WR, Dscr, Sscr, Rscr);
    // part: rewritten only after the next batch's first barrier, which every wave reaches after it has read part here
  }
}

// ---- backward (pixels): dL/dx and dL/dw of every tile pixel --------------------------------
// thread = (sample lane, pixel slot); the batch's leaf-gradient tile is staged in LDS.
template <int R, int S, int G, int SLOTS>
__global__ __launch_bounds__(64 * SLOTS) void objspn_pix_k(
    const float* __restrict__ xw, const float* __restrict__ Dscr, const int* __restrict__ leaf_slot,
    const float* __restrict__ coef, float* __restrict__ dxw, int n_batches) {
  constexpr int D = 4 * S;
  constexpr int DT = R * 4 * G * 64;
  __shared__ float dl[DT];
  const int lane = lane_id();
  const int slot = wave_id();
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    const float4* src = reinterpret_cast<const float4*>(Dscr + (size_t)b * DT);
    float4* dst = reinterpret_cast<float4*>(dl);
    for (int i = threadIdx.x; i < DT / 4; i += 64 * SLOTS) dst[i] = src[i];
    __syncthreads();
    const float* tile = xw + (size_t)b * (D * 2 * 64);
    float* otile = dxw + (size_t)b * (D * 2 * 64);
    for (int p = slot; p < D; p += SLOTS) {
      const float x = tile[(p * 2) * 64 + lane];
      const float w = tile[(p * 2 + 1) * 64 + lane];
      // sum_g d_g (a_g x^2 + b_g x + c_g) = x^2 A + x B + C with A = sum d_g a_g, ...: 3 FMAs per (replica, gaussian)
      float A = 0.0f, Bq = 0.0f, C = 0.0f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ls = leaf_slot[r * D + p];          // L*S + i (wave-uniform)
        const int L = ls / S;
        const float* cf = coef + (size_t)(r * 4 * S + ls) * G * 3;
        const float* dlp = dl + ((r * 4 + L) * G) * 64 + lane;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float d = dlp[g * 64];
          A = fmaf(d, cf[g * 3], A);
          Bq = fmaf(d, cf[g * 3 + 1], Bq);
          C = fmaf(d, cf[g * 3 + 2], C);
        }
      }
      otile[(p * 2) * 64 + lane] = fmaf(x + x, A, Bq) * w;
      otile[(p * 2 + 1) * 64 + lane] = fmaf(x, fmaf(x, A, Bq), C);
    }
    __syncthreads();
  }
}



// The scratch of objspn_fwd_unit_k is written for an upstream gradient of 1: the factor go[sample] that its three linear
// pieces (leaf gradients, gamma, rho) still lack is applied here.  For the leaf-coefficient products it goes onto the OTHER
// operand: all three features (w x^2, w x, w) are linear in the pixel's w, so the w rows of the staged tile are multiplied
// by go[sample] once per batch; gamma and rho take it as they are fetched.  scale == nullptr: the scratch already carries
// it (objspn_bwd_k).  `scale` holds a multiple of 64 entries, zeros beyond the last sample.
__device__ __forceinline__ float4 tg_scale4(const float* __restrict__ scale, int first) {
  return scale == nullptr ? float4{1.0f, 1.0f, 1.0f, 1.0f} : *reinterpret_cast<const float4*>(scale + first);
}
// stage the batch's glimpse tile as [2 D rows][LD]; NT = threads of the workgroup (a multiple of 32: a thread keeps its
// four samples and its row parity over the whole copy)
template <int D, int LD, int NT>
__device__ __forceinline__ void tg_stage_tile(float* tileP, const float* __restrict__ xw_b, const float* __restrict__ scale,
                                              int first_sample) {
  static_assert(NT % 32 == 0, "row parity and sample slot per thread");
  const float4* src = reinterpret_cast<const float4*>(xw_b);
  const bool wrow = ((threadIdx.x >> 4) & 1) != 0;
  const float4 sc = wrow ? tg_scale4(scale, first_sample + 4 * (threadIdx.x & 15)) : float4{1.0f, 1.0f, 1.0f, 1.0f};
  for (int i = threadIdx.x; i < 2 * D * 16; i += NT) {
    float4 v = src[i];
    v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
    *reinterpret_cast<float4*>(tileP + (i >> 4) * LD + 4 * (i & 15)) = v;
  }
}

// ---- backward (all table gradients) on the matrix cores ------------------------------------------------------------
// The three table gradients are sums over the samples of outer products,
//     d coef[r][L][i][g][c] = sum_s  f_c(x, w)[pixel(r, L, i)][s] * dl[r][L][g][s]          f = (w x^2, w x, w)
//     d wsum[node][j2 G + j1][k] = sum_s  E1[j1][s] E2[j2][s] * gamma[k][s]
//     d wroot[r][j2 K + j1]      = sum_s  rho[s] EB[j2][s] * EA[j1][s]
// i.e. GEMMs with the sample index as K: M = 75 / 100 / 10 rows, N = 10 columns, K = 64 per batch -- what the round-1 kernels
// (objspn_coefgrad_k / objspn_wgrad_k, removed) did with one thread per output row on the
// VALU, which is also what every other kernel of this phase of the step is bound by.  Here they go through
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, so the numerics stay those of an fp32 sum; the matrix pipe is otherwise idle
// in the backward): a wave = one (replica, side) = its two leaves, its sum node and (side 0) the replica's root, 18 output
// tiles of 16 x 16 held in 72 accumulator registers across ALL batches of the workgroup's chunk.
// Operands are rows of 64 samples stored sample-contiguous ([row][64]); lane (row = lane & 15, kk = lane >> 4) takes the 16
// samples 16 kk .. 16 kk + 15 of its row as four float4 and feeds sample 16 kk + s in k-slot kk of MFMA s (A and B use the
// same assignment; a sum over samples does not care).  Rows beyond M and columns beyond N compute garbage that is never
// stored (rows / columns of an MFMA result are independent).  The glimpse tile (read by all six replicas) and the sum-node
// scratch are staged in LDS with rows padded to 68 floats (2-way instead of 8-way bank conflicts on the 16-byte reads).
// Fixed summation order: bitwise reproducible.
template <int R, int S, int G, int K>
__global__ __launch_bounds__(128 * R) void objspn_tablegrad_k(
    const float* __restrict__ xw, const float* __restrict__ Dscr, const float* __restrict__ Sscr, const float* __restrict__ Rscr,
    const int* __restrict__ scope, float* __restrict__ part_c, float* __restrict__ part_w, float* __restrict__ part_r,
    int n_batches, int n_chunks, const float* __restrict__ scale) {
  constexpr int D = 4 * S, LD = 68, NS = K + 2 * G;
  constexpr int TC = (3 * S + 15) / 16, TW = (G * G + 15) / 16;       // 5 coefficient tiles per leaf, 7 sum-weight tiles
  static_assert(G <= 16 && K <= 16, "one column tile");
  typedef __attribute__((ext_vector_type(4))) float f4;
  extern __shared__ __attribute__((aligned(16))) float tg_lds[];
  float* tileP = tg_lds;                         // [2 D][LD]: row 2 p = x of pixel p, 2 p + 1 = w
  float* nodeS = tileP + 2 * D * LD;             // [2 R][NS][LD]: gamma[K] | E1[G] | E2[G] of every sum node
  const int lane = lane_id(), wv = wave_id();
  const int r = wv >> 1, side = wv & 1, node = wv;
  const int rr = lane & 15, kk = lane >> 4;
  const int c = blockIdx.x;
  // per-lane row descriptors (constant over the batches)
  int c_row[2][TC], c_feat[2][TC];               // tileP row of the pixel's x, feature index 0..2
#pragma unroll
  for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      const int row = min(16 * t + rr, 3 * S - 1);
      c_feat[l2][t] = row / S;
      c_row[l2][t] = 2 * scope[(r * 4 + side * 2 + l2) * S + row % S];
    }
  int w_e1[TW], w_e2[TW];                        // nodeS rows of E1[j1], E2[j2] of sum-weight row j2 G + j1
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const int row = min(16 * t + rr, G * G - 1);
    w_e1[t] = K + row % G;
    w_e2[t] = K + G + row / G;
  }
  f4 acc_c[2][TC], acc_w[TW], acc_r;
#pragma unroll
  for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
    for (int t = 0; t < TC; ++t) acc_c[l2][t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < TW; ++t) acc_w[t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
  acc_r = f4{0.0f, 0.0f, 0.0f, 0.0f};

  for (int b = c; b < n_batches; b += n_chunks) {
    {   // stage the tile (all waves) and the wave's own sum-node scratch
      tg_stage_tile<D, LD, 128 * R>(tileP, xw + (size_t)b * (D * 2 * 64), scale, b * 64);
      const float4* ss = reinterpret_cast<const float4*>(Sscr + (size_t)(b * R * 2 + node) * NS * 64);
      float* dst = nodeS + node * NS * LD;
      for (int i = lane; i < NS * 16; i += 64) *reinterpret_cast<float4*>(dst + (i >> 4) * LD + 4 * (i & 15)) = ss[i];
    }
    __syncthreads();
    // ---- leaf coefficients: A = features of the leaf's pixels, B = leaf gradients
#pragma unroll
    for (int l2 = 0; l2 < 2; ++l2) {
      const int L = side * 2 + l2;
      const float4* dp = reinterpret_cast<const float4*>(Dscr + (((size_t)(b * R + r) * 4 + L) * G + min(rr, G - 1)) * 64 + 16 * kk);
      const float4 d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
      const float dv[16] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x, d3.y, d3.z, d3.w};
#pragma unroll
      for (int t = 0; t < TC; ++t) {
        const float* xr = tileP + c_row[l2][t] * LD + 16 * kk;
        const int feat = c_feat[l2][t];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const float4 x4 = *reinterpret_cast<const float4*>(xr + 4 * h);
          const float4 w4 = *reinterpret_cast<const float4*>(xr + LD + 4 * h);
          const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float wx = ws[e] * xs[e];
            const float fv = feat == 0 ? wx * xs[e] : (feat == 1 ? wx : ws[e]);
            acc_c[l2][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fv, dv[4 * h + e], acc_c[l2][t], 0, 0, 0);
          }
        }
      }
    }
    // ---- sum-node weights: A = E1[j1] E2[j2], B = gamma
    {
      const float* nb = nodeS + node * NS * LD + 16 * kk;
      const float* gp = nb + min(rr, K - 1) * LD;
      float gv[16];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float4 g4 = *reinterpret_cast<const float4*>(gp + 4 * h);
        const float4 s4 = tg_scale4(scale, b * 64 + 16 * kk + 4 * h);
        gv[4 * h] = g4.x * s4.x; gv[4 * h + 1] = g4.y * s4.y; gv[4 * h + 2] = g4.z * s4.z; gv[4 * h + 3] = g4.w * s4.w;
      }
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const float* e1 = nb + w_e1[t] * LD;
        const float* e2 = nb + w_e2[t] * LD;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const float4 a4 = *reinterpret_cast<const float4*>(e1 + 4 * h);
          const float4 b4 = *reinterpret_cast<const float4*>(e2 + 4 * h);
          acc_w[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x * b4.x, gv[4 * h], acc_w[t], 0, 0, 0);
          acc_w[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y * b4.y, gv[4 * h + 1], acc_w[t], 0, 0, 0);
          acc_w[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z * b4.z, gv[4 * h + 2], acc_w[t], 0, 0, 0);
          acc_w[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w * b4.w, gv[4 * h + 3], acc_w[t], 0, 0, 0);
        }
      }
    }
    // ---- root weights of the replica (side-0 wave): A = rho EB[j2], B = EA[j1]
    if (side == 0) {
      const float* rp = Rscr + (size_t)(b * R + r) * (1 + 2 * K) * 64 + 16 * kk;
      const float4* rho = reinterpret_cast<const float4*>(rp);
      const float4* ea = reinterpret_cast<const float4*>(rp + (1 + min(rr, K - 1)) * 64);
      const float4* eb = reinterpret_cast<const float4*>(rp + (1 + K + min(rr, K - 1)) * 64);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float4 q4 = rho[h], a4 = ea[h], b4 = eb[h];
        const float4 s4 = tg_scale4(scale, b * 64 + 16 * kk + 4 * h);
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.x * s4.x * b4.x, a4.x, acc_r, 0, 0, 0);
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.y * s4.y * b4.y, a4.y, acc_r, 0, 0, 0);
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.z * s4.z * b4.z, a4.z, acc_r, 0, 0, 0);
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.w * s4.w * b4.w, a4.w, acc_r, 0, 0, 0);
      }
    }
    __syncthreads();      // the LDS images are restaged for the next batch
  }
  // ---- partial sums of this chunk: result element (row 4 (lane >> 4) + e, column lane & 15) in register e
  const int col = lane & 15;
#pragma unroll
  for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * t + 4 * kk + e;
        if (row < 3 * S && col < G) {
          const int feat = row / S, i = row % S;
          part_c[((size_t)(c * R + r) * D + (side * 2 + l2) * S + i) * G * 3 + col * 3 + feat] = acc_c[l2][t][e];
        }
      }
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 16 * t + 4 * kk + e;
      if (row < G * G && col < K) part_w[((size_t)(c * R * 2 + node) * G * G + row) * K + col] = acc_w[t][e];
    }
  if (side == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 4 * kk + e;
      if (row < K && col < K) part_r[((size_t)(c * R + r) * K + row) * K + col] = acc_r[e];
    }
  }
}

// The same three contractions shaped to run UNDERNEATH the recursion's backward (dyn_loop_bwd_small_k: one 4-wave workgroup per
// CU for ~0.47 ms, 360 of a SIMD's 512 registers, bound by dependent-issue latency while the rest of the chip idles).  A
// 12-wave workgroup of objspn_tablegrad_k (3 waves x 158 registers per SIMD) cannot be resident next to it, so held back
// until then it simply ran afterwards; this variant is 4 waves -- ONE per SIMD -- well inside the 152 registers that are left:
// workgroup = one replica (grid.y = R), wave = (side, half); half 0 owns the side's first leaf and sum-weight tiles 0..3,
// half 1 the second leaf, tiles 4..6 and (side 0) the root: 36 accumulator registers per wave.  With one wave per SIMD
// nothing hides an MFMA's latency but the wave's own independent accumulators, so the products go tile-innermost (5 / 4
// independent MFMAs back to back).  Every workgroup stages the whole glimpse tile for its one replica: six times the tile
// reads of the wide kernel, out of the MALL, at a time when HBM is nearly idle.  Same fixed summation order per output as
// objspn_tablegrad_k over the same chunk count -> bitwise reproducible (the chunking differs between the two kernels).
template <int R, int S, int G, int K>
__global__ __launch_bounds__(256, 3) void objspn_tablegrad_under_k(      // <= 168 registers: the recursion's backward (N <= 4) holds at most 306 of a SIMD's 512
   
    const float* __restrict__ xw, const float* __restrict__ Dscr, const float* __restrict__ Sscr, const float* __restrict__ Rscr,
    const int* __restrict__ scope, float* __restrict__ part_c, float* __restrict__ part_w, float* __restrict__ part_r,
    int n_batches, int n_chunks, const float* __restrict__ scale) {
  constexpr int D = 4 * S, LD = 68, NS = K + 2 * G;
  constexpr int TC = (3 * S + 15) / 16, TW = (G * G + 15) / 16, TWH = (TW + 1) / 2;
  static_assert(G <= 16 && K <= 16, "one column tile");
  typedef __attribute__((ext_vector_type(4))) float f4;
  extern __shared__ __attribute__((aligned(16))) float tg_lds[];
  float* tileP = tg_lds;                         // [2 D][LD]: row 2 p = x of pixel p, 2 p + 1 = w -- ALL the LDS there is next
                                                 // to the recursion's 107 KB: the sum-node rows are read from global memory
  float* scl = tileP + 2 * D * LD;               // [64] dL/d root of the batch's samples
  const int lane = lane_id(), wv = wave_id();
  // workgroup -> (chunk c, replica r).  The R workgroups of a chunk stage the SAME glimpse tiles: when the chunk count is a multiple
  // of 8 they are placed on one XCD (workgroup ids congruent mod 8 share an L2) and start together, so five of the six tile
  // reads of a batch are L2 hits instead of HBM reads (round 2: 501 MB per launch, the tile six times over)
  int c, r;
  if ((n_chunks & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    c = xcd * (n_chunks >> 3) + j / R;
    r = j % R;
  } else {
    c = blockIdx.x / R;
    r = blockIdx.x % R;
  }
  const int side = wv & 1, half = wv >> 1, node = 2 * r + side, L = side * 2 + half;
  const int rr = lane & 15, kk = lane >> 4;
  int c_row[TC], c_feat[TC];
#pragma unroll
  for (int t = 0; t < TC; ++t) {
    const int row = min(16 * t + rr, 3 * S - 1);
    c_feat[t] = row / S;
    c_row[t] = 2 * scope[(r * 4 + L) * S + row % S];
  }
  int w_e1[TWH], w_e2[TWH];
#pragma unroll
  for (int i = 0; i < TWH; ++i) {
    const int row = min(16 * (half * TWH + i) + rr, G * G - 1);
    w_e1[i] = K + row % G;
    w_e2[i] = K + G + row / G;
  }
  f4 acc_c[TC], acc_w[TWH], acc_r = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < TC; ++t) acc_c[t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < TWH; ++i) acc_w[i] = f4{0.0f, 0.0f, 0.0f, 0.0f};

  for (int b = c; b < n_batches; b += n_chunks) {
    tg_stage_tile<D, LD, 256>(tileP, xw + (size_t)b * (D * 2 * 64), scale, b * 64);
    // the batch's 64 scale values go through LDS as well: gamma and rho take them from there (as global loads inside this
    // one-wave-per-SIMD chain they stretched the kernel 514 -> 570 us)
    if (threadIdx.x < 16) *reinterpret_cast<float4*>(scl + 4 * threadIdx.x) = tg_scale4(scale, b * 64 + 4 * threadIdx.x);
    __syncthreads();
    // ---- leaf coefficients of leaf L: A = features of its pixels, B = its gradients
    {
      const float4* dp = reinterpret_cast<const float4*>(Dscr + (((size_t)(b * R + r) * 4 + L) * G + min(rr, G - 1)) * 64 + 16 * kk);
      const float4 d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
      const float dv[16] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x, d3.y, d3.z, d3.w};
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        float fv[TC][4];
#pragma unroll
        for (int t = 0; t < TC; ++t) {
          const float* xr = tileP + c_row[t] * LD + 16 * kk + 4 * h;
          const float4 x4 = *reinterpret_cast<const float4*>(xr);
          const float4 w4 = *reinterpret_cast<const float4*>(xr + LD);
          const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float wx = ws[e] * xs[e];
            fv[t][e] = c_feat[t] == 0 ? wx * xs[e] : (c_feat[t] == 1 ? wx : ws[e]);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < TC; ++t) acc_c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[t][e], dv[4 * h + e], acc_c[t], 0, 0, 0);
      }
    }
    // ---- this half's sum-node weight tiles: A = E1[j1] E2[j2], B = gamma
    {
      const float* nb = Sscr + (size_t)(b * R * 2 + node) * NS * 64 + 16 * kk;
      const float* gp = nb + min(rr, K - 1) * 64;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float4 g4 = *reinterpret_cast<const float4*>(gp + 4 * h);
        const float4 s4 = *reinterpret_cast<const float4*>(scl + 16 * kk + 4 * h);
        const float gv[4] = {g4.x * s4.x, g4.y * s4.y, g4.z * s4.z, g4.w * s4.w};
        float pr[TWH][4];
#pragma unroll
        for (int i = 0; i < TWH; ++i) {
          const float4 a4 = *reinterpret_cast<const float4*>(nb + w_e1[i] * 64 + 4 * h);
          const float4 b4 = *reinterpret_cast<const float4*>(nb + w_e2[i] * 64 + 4 * h);
          pr[i][0] = a4.x * b4.x; pr[i][1] = a4.y * b4.y; pr[i][2] = a4.z * b4.z; pr[i][3] = a4.w * b4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TWH; ++i)      // (tile 7 of half 1 does not exist: its products are computed and never stored)
            acc_w[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[i][e], gv[e], acc_w[i], 0, 0, 0);
      }
    }
    // ---- root weights of the replica (side 0, half 1): A = rho EB[j2], B = EA[j1]
    if (side == 0 && half == 1) {
      const float* rp = Rscr + (size_t)(b * R + r) * (1 + 2 * K) * 64 + 16 * kk;
      const float4* rho = reinterpret_cast<const float4*>(rp);
      const float4* ea = reinterpret_cast<const float4*>(rp + (1 + min(rr, K - 1)) * 64);
      const float4* eb = reinterpret_cast<const float4*>(rp + (1 + K + min(rr, K - 1)) * 64);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float4 q4 = rho[h], a4 = ea[h], b4 = eb[h];
        const float4 s4 = *reinterpret_cast<const float4*>(scl + 16 * kk + 4 * h);
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.x * s4.x * b4.x, a4.x, acc_r, 0, 0, 0);
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.y * s4.y * b4.y, a4.y, acc_r, 0, 0, 0);
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.z * s4.z * b4.z, a4.z, acc_r, 0, 0, 0);
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.w * s4.w * b4.w, a4.w, acc_r, 0, 0, 0);
      }
    }
    __syncthreads();      // the LDS images are restaged for the next batch
  }
  const int col = lane & 15;
#pragma unroll
  for (int t = 0; t < TC; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 16 * t + 4 * kk + e;
      if (row < 3 * S && col < G) {
        const int feat = row / S, i = row % S;
        part_c[((size_t)(c * R + r) * D + L * S + i) * G * 3 + col * 3 + feat] = acc_c[t][e];
      }
    }
#pragma unroll
  for (int i = 0; i < TWH; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 16 * (half * TWH + i) + 4 * kk + e;
      if (row < G * G && col < K) part_w[((size_t)(c * R * 2 + node) * G * G + row) * K + col] = acc_w[i][e];
    }
  if (side == 0 && half == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 4 * kk + e;
      if (row < K && col < K) part_r[((size_t)(c * R + r) * K + row) * K + col] = acc_r[e];
    }
  }
}

// ---- tile staging for the stand-alone RatSpn.forward(inputs, marginalized) operator --------
// inputs/marg are (n, D) row-major; w = 1 - clamp(marg, 0, 1) (rat_torch.py:104-106).
__global__ void objspn_tile_from_arrays_k(const float* __restrict__ inputs, const float* __restrict__ marg,
                                          float* __restrict__ xw, int n_samples, int n_batches, int D) {
  extern __shared__ float lds[];   // [64][D+1] x then w
  const int stride = D + 1;
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    for (int i = threadIdx.x; i < 64 * D; i += blockDim.x) {
      const int smp = i / D, p = i % D;
      const size_t gi = (size_t)(b * 64 + smp) * D + p;
      const bool live = (b * 64 + smp) < n_samples;
      const float x = live ? inputs[gi] : 0.0f;
      float w = 0.0f;
      if (live) w = (marg != nullptr) ? 1.0f - fminf(fmaxf(marg[gi], 0.0f), 1.0f) : 1.0f;
      lds[smp * stride + p] = x;
      lds[64 * stride + smp * stride + p] = w;
    }
    __syncthreads();
    float* tile = xw + (size_t)b * D * 2 * 64;
    for (int i = threadIdx.x; i < 64 * D * 2; i += blockDim.x) {
      const int lane = i & 63, pc = i >> 6;   // pc = p*2 + c
      tile[i] = lds[(pc & 1) * 64 * stride + lane * stride + (pc >> 1)];
    }
    __syncthreads();
  }
}

// d_inputs = dxw.x ; d_marg = -dxw.w where 0 <= marg <= 1 (clamp passes its boundary, as ATen does)
__global__ void objspn_tile_to_arrays_k(const float* __restrict__ dxw, const float* __restrict__ marg,
                                        float* __restrict__ d_inputs, float* __restrict__ d_marg,
                                        int n_samples, int n_batches, int D) {
  extern __shared__ float lds[];
  const int stride = D + 1;
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    const float* tile = dxw + (size_t)b * D * 2 * 64;
    for (int i = threadIdx.x; i < 64 * D * 2; i += blockDim.x) {
      const int lane = i & 63, pc = i >> 6;
      lds[(pc & 1) * 64 * stride + lane * stride + (pc >> 1)] = tile[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * D; i += blockDim.x) {
      const int smp = i / D, p = i % D;
      if ((b * 64 + smp) >= n_samples) continue;
      const size_t gi = (size_t)(b * 64 + smp) * D + p;
      if (d_inputs != nullptr) d_inputs[gi] = lds[smp * stride + p];
      if (d_marg != nullptr) {
        const float m = marg[gi];
        d_marg[gi] = (m >= 0.0f && m <= 1.0f) ? -lds[64 * stride + smp * stride + p] : 0.0f;
      }
    }
    __syncthreads();
  }
}

// out[j] = sum_c part[c][j]  (fixed order -> bitwise reproducible).
// 256 threads = 32 elements x 8 chunk slices: slice q adds chunks q, q+8, ...; the 8 slice sums are then
// added in slice order.  (One thread per element walking all chunks serially was latency-bound.)
__global__ __launch_bounds__(256) void reduce_chunks_k(const float* __restrict__ part, float* __restrict__ out, int n, int n_chunks, int accumulate,
                                                       float* __restrict__ out2 = nullptr) {      // out2: a second destination of the same sums
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + e;
  float s = 0.0f;
  if (j < n)
    for (int c = q; c < n_chunks; c += 8) s += part[(size_t)c * n + j];
  red[q][e] = s;
  __syncthreads();
  if (q == 0 && j < n) {
    float t = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][e];
    out[j] = accumulate ? out[j] + t : t;
    if (out2 != nullptr) out2[j] = accumulate ? out2[j] + t : t;
  }
}

// Three such reductions as ONE launch (the object SPN's coefficient, sum-weight and root-weight partials share their chunk count): a
// launch costs 5-8 us whatever it does, and these sit in a row on the parameter stream.  Block ranges [0, nb0), [nb0, nb0 + nb1), ...
// take the three arrays; per element the same slices in the same order as reduce_chunks_k: identical sums.
struct Reduce3 {
  const float* part[3];
  float* out[3];
  int n[3];
};
__global__ __launch_bounds__(256) void reduce_chunks3_k(Reduce3 a, int n_chunks) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
  int blk = blockIdx.x, seg = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int nbk = (a.n[k] + 31) / 32;
    if (seg == k && blk >= nbk) {
      blk -= nbk;
      seg = k + 1;
    }
  }
  const float* __restrict__ part = seg == 0 ? a.part[0] : (seg == 1 ? a.part[1] : a.part[2]);
  float* __restrict__ out = seg == 0 ? a.out[0] : (seg == 1 ? a.out[1] : a.out[2]);
  const int n = seg == 0 ? a.n[0] : (seg == 1 ? a.n[1] : a.n[2]);
  const int j = blk * 32 + e;
  float s = 0.0f;
  if (j < n)
    for (int c = q; c < n_chunks; c += 8) s += part[(size_t)c * n + j];
  red[q][e] = s;
  __syncthreads();
  if (q == 0 && j < n) {
    float t = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][e];
    out[j] = t;
  }
}

}  // namespace stove

// =============================================================================================
// host-side launchers (C ABI lives in capi.hip)
// =============================================================================================
namespace stove {

static inline int grid_for(int n_items, int cap) { return n_items < cap ? (n_items > 0 ? n_items : 1) : cap; }

int objspn_tile_from_arrays(const float* inputs, const float* marg, float* xw, int n, hipStream_t st) {
  const int nb = (n + 63) / 64;
  if (nb == 0) return 0;
  const int D = 100;
  STOVE_LAUNCH(objspn_tile_from_arrays_k, dim3(grid_for(nb, 2048)), dim3(256), 2 * 64 * (D + 1) * sizeof(float), st,
                     inputs, marg, xw, n, nb, D);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int objspn_tile_to_arrays(const float* dxw, const float* marg, float* d_inputs, float* d_marg, int n, hipStream_t st) {
  const int nb = (n + 63) / 64;
  if (nb == 0) return 0;
  const int D = 100;
  STOVE_LAUNCH(objspn_tile_to_arrays_k, dim3(grid_for(nb, 2048)), dim3(256), 2 * 64 * (D + 1) * sizeof(float), st,
                     dxw, marg, d_inputs, d_marg, n, nb, D);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int objspn_forward(const float* xw, const int* scope, const float* coef, const float* wsum, const float* wroot,
                   float* out, float* ovl, int n, hipStream_t st) {
  const int nb = (n + 63) / 64;
  if (nb == 0) return 0;
  STOVE_LAUNCH((objspn_fwd_k<6, 25, 10, 10>), dim3(grid_for(nb, 4096)), dim3(768), 0, st,
                     xw, scope, coef, wsum, wroot, out, ovl, n, nb);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int objspn_mpe(const float* xw, const int* scope, const float* coef, const float* wsum, const float* wroot, const float* mu,
               float* out, int* pick, int n, hipStream_t st) {
  const int nb = (n + 63) / 64;
  if (nb == 0) return 0;
  STOVE_LAUNCH((objspn_mpe_k<6, 25, 10, 10>), dim3(grid_for(nb, 4096)), dim3(768), 0, st,
                     xw, scope, coef, wsum, wroot, mu, out, pick, n, nb);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// workspace layout (floats) for the backward, per batch of 64 samples
constexpr size_t kObjD = 6 * 4 * 10 * 64, kObjS = 12 * 30 * 64, kObjR = 6 * 21 * 64, kObjX = 100 * 2 * 64;
constexpr int kObjChunks = 256;        // one workgroup of the table-gradient kernel per CU
constexpr size_t kObjCoefN = 6 * 100 * 10 * 3, kObjWN = 12 * 100 * 10, kObjRootN = 6 * 100;

// per-sample scratch of the backward (leaf gradients, gamma | E1 | E2, rho | EA | EB) and the chunk partials of the table gradients
size_t objspn_scratch_floats(int n) { return (size_t)((n + 63) / 64) * (kObjD + kObjS + kObjR); }
size_t objspn_partial_floats() { return (size_t)kObjChunks * (kObjCoefN + kObjWN + kObjRootN); }
size_t objspn_bwd_ws_floats(int n) { return objspn_scratch_floats(n) + objspn_partial_floats(); }

// forward that also leaves the backward's per-sample scratch behind, for an upstream gradient of 1 (objspn_fwd_unit_k)
int objspn_forward_unit(const float* xw, const int* scope, const float* coef, const float* wsum, const float* wroot,
                        float* out, float* ovl, float* scratch, int n, hipStream_t st) {
  const int nb = (n + 63) / 64;
  if (nb == 0) return 0;
  float* Dscr = scratch;
  float* Sscr = Dscr + (size_t)nb * kObjD;
  float* Rscr = Sscr + (size_t)nb * kObjS;
  STOVE_LAUNCH((objspn_fwd_unit_k<6, 25, 10, 10>), dim3(grid_for(nb, 4096)), dim3(768), 0, st,
                     xw, scope, coef, wsum, wroot, out, ovl, Dscr, Sscr, Rscr, n, nb);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// dxw: [nb][100][2][64] out.  g_coef/g_wsum/g_wroot: gradients w.r.t. the baked tables (overwritten).
// The backward in two parts: `objspn_backward_data` (leaf gradients + dL/d tile, on `st`) and `objspn_backward_params`
// (table gradients from the per-sample scratch; they only feed the optimiser, so a caller may put them on
// another stream, ordered after the data part, where they overlap with whatever runs on `st` next).
int objspn_backward_data(const float* xw, const int* scope, const int* leaf_slot, const float* coef, const float* wsum,
                         const float* wroot, const float* out, const float* dout, float* dxw, float* scratch, int n, hipStream_t st) {
  const int nb = (n + 63) / 64;
  if (nb == 0) return 0;
  float* Dscr = scratch;
  float* Sscr = Dscr + (size_t)nb * kObjD;
  float* Rscr = Sscr + (size_t)nb * kObjS;
  STOVE_LAUNCH((objspn_bwd_k<6, 25, 10, 10>), dim3(grid_for(nb, 4096)), dim3(768), 0, st,
                     xw, scope, coef, wsum, wroot, out, dout, Dscr, Sscr, Rscr, n, nb);
  STOVE_LAUNCH_CHECK();
  if (dxw == nullptr) return 0;
  STOVE_LAUNCH((objspn_pix_k<6, 25, 10, 8>), dim3(grid_for(nb, 4096)), dim3(512), 0, st,
                     xw, Dscr, leaf_slot, coef, dxw, nb);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// scratch: objspn_scratch_floats(n) from objspn_backward_data (scale == nullptr) or from objspn_forward_unit (scale = the
// upstream gradient per sample: a multiple of 64 floats, zeros beyond sample n); partial: objspn_partial_floats()
int objspn_backward_params(const float* xw, const int* scope, float* g_coef, float* g_wsum, float* g_wroot, const float* scratch,
                           float* partial, const float* scale, int n, hipStream_t st, bool under = false) {
  const int nb = (n + 63) / 64;
  if (nb == 0) {
    hipMemsetAsync(g_coef, 0, kObjCoefN * 4, st);
    hipMemsetAsync(g_wsum, 0, kObjWN * 4, st);
    hipMemsetAsync(g_wroot, 0, kObjRootN * 4, st);
    return 0;
  }
  const float* Dscr = scratch;
  const float* Sscr = Dscr + (size_t)nb * kObjD;
  const float* Rscr = Sscr + (size_t)nb * kObjS;
  float* pc = partial;
  float* pw = pc + (size_t)kObjChunks * kObjCoefN;
  float* pr = pw + (size_t)kObjChunks * kObjWN;
  int chunks = nb < kObjChunks ? nb : kObjChunks;
  if (under) {
    // one resident round next to the recursion's workgroups: 6 workgroups (replicas) per chunk, one per CU
    constexpr int kTgLdsU = (2 * 100 * 68 + 64) * (int)sizeof(float);      // + the recursion's 106 880 B + 512 B < 160 KB
    if (chunks > kObjChunks / 6) chunks = (kObjChunks / 6) & ~7;      // 40: a multiple of 8 (XCD-aware placement), 240 workgroups
    int rc = (int)hipFuncSetAttribute((const void*)objspn_tablegrad_under_k<6, 25, 10, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, kTgLdsU);
    if (rc) return rc;
    STOVE_LAUNCH((objspn_tablegrad_under_k<6, 25, 10, 10>), dim3(chunks * 6), dim3(256), kTgLdsU, st, xw, Dscr, Sscr, Rscr, scope, pc, pw, pr, nb, chunks, scale);
    STOVE_LAUNCH_CHECK();
  } else {
    constexpr int kTgLds = (2 * 100 * 68 + 12 * 30 * 68) * (int)sizeof(float);
    int rc = (int)hipFuncSetAttribute((const void*)objspn_tablegrad_k<6, 25, 10, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, kTgLds);
    if (rc) return rc;
    STOVE_LAUNCH((objspn_tablegrad_k<6, 25, 10, 10>), dim3(chunks), dim3(768), kTgLds, st, xw, Dscr, Sscr, Rscr, scope, pc, pw, pr, nb, chunks, scale);
    STOVE_LAUNCH_CHECK();
  }
  Reduce3 r3;
  r3.part[0] = pc; r3.part[1] = pw; r3.part[2] = pr;
  r3.out[0] = g_coef; r3.out[1] = g_wsum; r3.out[2] = g_wroot;
  r3.n[0] = (int)kObjCoefN; r3.n[1] = (int)kObjWN; r3.n[2] = (int)kObjRootN;
  STOVE_LAUNCH(reduce_chunks3_k, dim3((kObjCoefN + 31) / 32 + (kObjWN + 31) / 32 + (kObjRootN + 31) / 32), dim3(256), 0, st, r3, chunks);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ws: objspn_bwd_ws_floats(n) = [ per-sample scratch | chunk partials ]
int objspn_backward(const float* xw, const int* scope, const int* leaf_slot, const float* coef, const float* wsum,
                    const float* wroot, const float* out, const float* dout, float* dxw,
                    float* g_coef, float* g_wsum, float* g_wroot, float* ws, int n, hipStream_t st) {
  int rc = objspn_backward_data(xw, scope, leaf_slot, coef, wsum, wroot, out, dout, dxw, ws, n, st);
  if (rc) return rc;
  return objspn_backward_params(xw, scope, g_coef, g_wsum, g_wroot, ws, ws + objspn_scratch_floats(n), nullptr, n, st);
}

}  // namespace stove
