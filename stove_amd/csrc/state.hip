// SuPAIR state pipeline and ELBO assembly around the fused kernels, gfx950.
//
// Between the recognition LSTM and the inference recursion, and between the recursion / scene likelihood and
// the scalar ELBO, the reference runs ~80 tiny ATen launches forward and ~100 backward on (n, T, o, <=18) tensors
// (Supair.constrain_zp supair.py:112-149, Stove.fix_supair stove.py:516-563, v_from_state / std_from_pos
// stove.py:172-198, the index gathers of the matchers, sy_from_quotient, Normal.log_prob sums and means
// stove.py:716-760).  All of it is elementwise or a short stencil in t, so it becomes 5 launches forward and
// 4 backward here.  Everything is deterministic (gather-form backward, fixed-order reductions).
#include "common.h"

namespace stove {

struct ZpConst {
  float span[8], low[8];      // mean/std of [sx, sy/sx, x, y]: low + span * sigmoid(code)
};

// codes (M, 8) -> zc (M, 8) constrained [mean 4 | std 4], pos (M, 2) = positions, the matchers' features
__global__ void zp_constrain_k(const float* __restrict__ codes, ZpConst kc, float* __restrict__ zc, float* __restrict__ pos, int M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * 8) return;
  const int d = i & 7;
  const float v = fmaf(kc.span[d], sigmoidf_(codes[i]), kc.low[d]);
  zc[i] = v;
  if (d == 2 || d == 3) pos[(i >> 3) * 2 + d - 2] = v;
}

__device__ __forceinline__ void load8(const float* __restrict__ p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// matched + smoothed state of slot k at time t (fix_supair): returns the 2-bit hit mask
__device__ __forceinline__ int fixed_state(const float* __restrict__ zc, const long long* __restrict__ idx, int b, int t, int k, int T,
                                           int o, int fix, float* out) {
  const size_t row = (size_t)b * T + t;
  float cur[8];
  load8(zc + (row * o + idx[row * o + k]) * 8, cur);
  int hit = 0;
  if (fix && t >= 1 && t <= T - 2) {
    float pv[8], nx[8];
    load8(zc + ((row - 1) * o + idx[(row - 1) * o + k]) * 8, pv);
    load8(zc + ((row + 1) * o + idx[(row + 1) * o + k]) * 8, nx);
#pragma unroll
    for (int a = 0; a < 2; ++a)
      if (fabsf(cur[a] - pv[a]) > 0.095f && fabsf(nx[a] - cur[a]) > 0.095f) hit |= 1 << a;
#pragma unroll
    for (int d = 0; d < 8; ++d)
      if (hit & (1 << (d & 1))) cur[d] = 0.5f * (pv[d] + nx[d]);
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) out[d] = cur[d];
  return hit;
}

// thread (b, t, k): zfix (n,T,o,8), hits (n,T,o) u8, and the recursion's inputs
//   zl / sl (n,Ts,o,6) = z_sup_full / z_sup_std_full [:, skip:],  init6 (n,o,6) = z_sup_full[:, skip-1]
__global__ void supair_state_fwd_k(const float* __restrict__ zc, const long long* __restrict__ idx, float* __restrict__ zfix,
                                   unsigned char* __restrict__ hits, float* __restrict__ zl, float* __restrict__ sl,
                                   float* __restrict__ init6, int n, int T, int o, int skip, int fix, int init_ld,
                                   const float* __restrict__ lat_noise, int lat_dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * T * o) return;
  const int k = i % o, t = (i / o) % T, b = i / (o * T);
  float cur[8], pv[8];
  const int hit = fixed_state(zc, idx, b, t, k, T, o, fix, cur);
  hits[i] = (unsigned char)hit;
#pragma unroll
  for (int d = 0; d < 8; ++d) zfix[(size_t)i * 8 + d] = cur[d];
  if (t < skip - 1 || t < 1) return;
  fixed_state(zc, idx, b, t - 1, k, T, o, fix, pv);
  float z6[6], s6[6];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    z6[d] = cur[d];
    s6[d] = cur[4 + d];
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    z6[4 + a] = cur[2 + a] - pv[2 + a];
    s6[4 + a] = sqrtf(cur[6 + a] * cur[6 + a] + pv[6 + a] * pv[6 + a]);
  }
  if (t == skip - 1) {
#pragma unroll
    for (int d = 0; d < 6; ++d) init6[((size_t)b * o + k) * init_ld + d] = z6[d];
    // the rest of the recursion's initial state: the unstructured latent = 0.01 x the caller's standard-normal draw (stove.py:663-668),
    // written here instead of a scale and a concatenation launch behind this kernel
    if (lat_noise != nullptr)
      for (int d = 0; d < lat_dim; ++d) init6[((size_t)b * o + k) * init_ld + 6 + d] = 0.01f * lat_noise[((size_t)b * o + k) * lat_dim + d];
  } else {
    const size_t r = (((size_t)b * (T - skip) + (t - skip)) * o + k) * 6;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      zl[r + d] = z6[d];
      sl[r + d] = s6[d];
    }
  }
}

// gradient of the six-vectors of time t (0 where that time feeds nothing)
__device__ __forceinline__ void g6_at(const float* __restrict__ g_zl, const float* __restrict__ g_sl, const float* __restrict__ g_init6,
                                      int b, int t, int k, int T, int o, int skip, float* gz, float* gs, int init_ld) {
#pragma unroll
  for (int d = 0; d < 6; ++d) gz[d] = gs[d] = 0.0f;
  if (t < 1 || t > T - 1) return;
  if (t == skip - 1) {
    if (g_init6 != nullptr)
#pragma unroll
      for (int d = 0; d < 6; ++d) gz[d] = g_init6[((size_t)b * o + k) * init_ld + d];
  } else if (t >= skip) {
    const size_t r = (((size_t)b * (T - skip) + (t - skip)) * o + k) * 6;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      if (g_zl != nullptr) gz[d] = g_zl[r + d];
      if (g_sl != nullptr) gs[d] = g_sl[r + d];
    }
  }
}

// gradient w.r.t. the smoothed state zfix[b, t, k, :] = element i of the (n, T, o) grid -> g[8]
__device__ __forceinline__ void supair_gfix(const float* __restrict__ zfix, const float* __restrict__ g_zfix, const float* __restrict__ g_zl,
                                            const float* __restrict__ g_sl, const float* __restrict__ g_init6, size_t i, int T, int o, int skip,
                                            int init_ld, float (&g)[8]) {
  const int k = (int)(i % o), t = (int)((i / o) % T), b = (int)(i / ((size_t)o * T));
#pragma unroll
  for (int d = 0; d < 8; ++d) g[d] = g_zfix != nullptr ? g_zfix[i * 8 + d] : 0.0f;
  float gz[6], gs[6], gzn[6], gsn[6];
  g6_at(g_zl, g_sl, g_init6, b, t, k, T, o, skip, gz, gs, init_ld);
  g6_at(g_zl, g_sl, g_init6, b, t + 1, k, T, o, skip, gzn, gsn, init_ld);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    g[d] += gz[d];
    g[4 + d] += gs[d];
  }
  float cur[8];
  load8(zfix + i * 8, cur);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    g[2 + a] += gz[4 + a] - gzn[4 + a];                  // v[t] = x[t] - x[t-1],  v[t+1] = x[t+1] - x[t]
    if (gs[4 + a] != 0.0f) {                             // vstd[t] = sqrt(s[t]^2 + s[t-1]^2)
      const float p = zfix[(i - o) * 8 + 6 + a];
      g[6 + a] += gs[4 + a] * cur[6 + a] / sqrtf(cur[6 + a] * cur[6 + a] + p * p);
    }
    if (gsn[4 + a] != 0.0f) {                            // vstd[t+1] = sqrt(s[t+1]^2 + s[t]^2)
      const float q = zfix[(i + o) * 8 + 6 + a];
      g[6 + a] += gsn[4 + a] * cur[6 + a] / sqrtf(q * q + cur[6 + a] * cur[6 + a]);
    }
  }
}

// The backward in ONE launch (round 6), thread (b, t, j) with j the PRE-matching object index: undo the smoothing stencil and the
// gather (as a gather over the slots k with idx[k] == j: no atomics, also right for the non-permutation 'volatile' mode), then the
// sigmoid constraint -> g_codes (M, 8).  The gradient of the smoothed state at (t, k) -- and, where fix_supair fired, at its time
// neighbours -- is formed on the fly by supair_gfix (until round 5 a first launch wrote it for all (t, k) and a second one read it
// back: two sub-10-us kernels on the step's serial chain); the same expressions in the same order: identical gradients.
__global__ void supair_state_bwd_k(const float* __restrict__ zc, const long long* __restrict__ idx, const unsigned char* __restrict__ hits,
                                   const float* __restrict__ zfix, const float* __restrict__ g_zfix, const float* __restrict__ g_zl,
                                   const float* __restrict__ g_sl, const float* __restrict__ g_init6, ZpConst kc, float* __restrict__ g_codes,
                                   int n, int T, int o, int skip, int init_ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * T * o) return;
  const int j = i % o, t = (i / o) % T, b = i / (o * T);
  const size_t row = (size_t)b * T + t;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < o; ++k) {
    if ((int)idx[row * o + k] != j) continue;
    const size_t e = row * o + k;
    const int h0 = hits[e];
    const int hp = (t >= 1) ? hits[e - o] : 0, hn = (t <= T - 2) ? hits[e + o] : 0;
    float g0[8], gp[8], gn[8];
    supair_gfix(zfix, g_zfix, g_zl, g_sl, g_init6, e, T, o, skip, init_ld, g0);
    if (hp) supair_gfix(zfix, g_zfix, g_zl, g_sl, g_init6, e - o, T, o, skip, init_ld, gp);
    if (hn) supair_gfix(zfix, g_zfix, g_zl, g_sl, g_init6, e + o, T, o, skip, init_ld, gn);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int bit = 1 << (d & 1);
      float v = (h0 & bit) ? 0.0f : g0[d];
      if (hp & bit) v += 0.5f * gp[d];
      if (hn & bit) v += 0.5f * gn[d];
      acc[d] += v;
    }
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const float s = (zc[(size_t)i * 8 + d] - kc.low[d]) / kc.span[d];          // the sigmoid value
    g_codes[(size_t)i * 8 + d] = acc[d] * kc.span[d] * s * (1.0f - s);
  }
}

// ---- z for the scene likelihood: frames 1..skip-1 use the SuPAIR means, frames skip..T-1 the sampled states;
// [sx, sy/sx, x, y] -> [sx, sy, x, y]   (stove.py:731-736 + Supair.sy_from_quotient)
__global__ void zall_fwd_k(const float* __restrict__ zfix, const float* __restrict__ zs, float* __restrict__ zall, int n, int T, int o, int skip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * (T - 1) * o) return;
  const int k = i % o, j = (i / o) % (T - 1), b = i / (o * (T - 1));
  const float* src = (j < skip - 1) ? zfix + (((size_t)b * T + 1 + j) * o + k) * 8 : zs + (((size_t)b * (T - skip) + (j - (skip - 1))) * o + k) * 18;
  const float sx = src[0];
  zall[(size_t)i * 4] = sx;
  zall[(size_t)i * 4 + 1] = sx * src[1];
  zall[(size_t)i * 4 + 2] = src[2];
  zall[(size_t)i * 4 + 3] = src[3];
}
// writes EVERY element of g_zfix (n,T,o,8) and g_zs (n,Ts,o,18): no memset needed
// dz_in (NULL = none): what the OTHER consumers of zs contribute to its gradient (the ELBO's log q and transition terms), same
// layout as g_zs -- added here instead of by a separate elementwise launch behind this one
__global__ void zall_bwd_k(const float* __restrict__ zfix, const float* __restrict__ zs, const float* __restrict__ g_zall,
                           const float* __restrict__ dz_in, float* __restrict__ g_zfix, float* __restrict__ g_zs, int n, int T, int o, int skip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nfix = n * T * o, nzs = n * (T - skip) * o;
  if (i < nfix) {
    const int k = i % o, t = (i / o) % T, b = i / (o * T);
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t >= 1 && t < skip) {
      const size_t a = (((size_t)b * (T - 1) + (t - 1)) * o + k) * 4;
      const float* src = zfix + (size_t)i * 8;
      g[0] = g_zall[a] + g_zall[a + 1] * src[1];
      g[1] = g_zall[a + 1] * src[0];
      g[2] = g_zall[a + 2];
      g[3] = g_zall[a + 3];
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) g_zfix[(size_t)i * 8 + d] = g[d];
  } else if (i < nfix + nzs) {
    const int q = i - nfix;
    const int k = q % o, ts = (q / o) % (T - skip), b = q / (o * (T - skip));
    const size_t a = (((size_t)b * (T - 1) + (skip - 1 + ts)) * o + k) * 4;
    const float* src = zs + (size_t)q * 18;
    float* g = g_zs + (size_t)q * 18;
    float v[4];
    v[0] = g_zall[a] + g_zall[a + 1] * src[1];
    v[1] = g_zall[a + 1] * src[0];
    v[2] = g_zall[a + 2];
    v[3] = g_zall[a + 3];
    if (dz_in == nullptr) {
#pragma unroll
      for (int d = 0; d < 18; ++d) g[d] = d < 4 ? v[d] : 0.0f;
    } else {
      const float* gi = dz_in + (size_t)q * 18;
#pragma unroll
      for (int d = 0; d < 18; ++d) g[d] = d < 4 ? v[d] + gi[d] : gi[d];
    }
  }
}

// ---- reparameterisation noise (stove.py:667, 679, 146/167: latent_prior.rsample, z_std_prior.rsample, one draw per step): all of a
// step's standard-normal draws as ONE counter-based launch.  Philox-4x32-10 keyed by state[0] (the seed), counter = (state[1] = the
// call number, thread index): thread i makes elements 4 i .. 4 i + 3 (two Box-Muller pairs on v_log / v_sin / v_cos), so a draw
// depends on (seed, call, element) only -- independent of the launch geometry, bitwise reproducible.  The call counter lives in device
// memory and is advanced by noise_tick_k behind the draw: a captured launch replays with fresh noise, and no host-side generator state
// (torch's graph-safe Philox costs two fill launches in front of every replay, on the step's serial chain) is involved.
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  c1 = (uint32_t)p1;
  c3 = (uint32_t)p0;
  c0 = n0;
  c2 = n2;
}
__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }      // (0, 1)
__global__ __launch_bounds__(256) void noise_normal_k(float* __restrict__ out, size_t n, const unsigned long long* __restrict__ state) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * i >= n) return;
  const unsigned long long seed = state[0], call = state[1];
  uint32_t c0 = (uint32_t)call, c1 = (uint32_t)(call >> 32), c2 = (uint32_t)i, c3 = (uint32_t)(i >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  // Box-Muller: radius from v_log_f32 (log2), angle in revolutions straight into v_sin_f32 / v_cos_f32
  const float r0 = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(c0))), r1 = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(c2)));
  const float a0 = u01(c1), a1 = u01(c3);
  const float v[4] = {r0 * __builtin_amdgcn_cosf(a0), r0 * __builtin_amdgcn_sinf(a0), r1 * __builtin_amdgcn_cosf(a1), r1 * __builtin_amdgcn_sinf(a1)};
  if (4 * i + 3 < n) {
    *reinterpret_cast<float4*>(out + 4 * i) = float4{v[0], v[1], v[2], v[3]};
  } else {
    for (size_t e = 4 * i; e < n; ++e) out[e] = v[e - 4 * i];
  }
}
__global__ void noise_tick_k(unsigned long long* __restrict__ state) { state[1] += 1ull; }

// ---- ELBO (stove.py:738-748): per sequence b
//   part[b] = { sum_t (trans - logq + lik[b, skip-1+t]),  sum_{j<skip-1} lik[b, j],  sum_t trans,  sum_t logq }
// with trans = sum_{k,d<16} log N(z[2+d]; zdyn[d], tstd[d]), logq = sum_{k,q<18} log N(z[q]; mean[q], std[q])
struct TransStd {
  float s[16];
};
__global__ __launch_bounds__(256) void elbo_part_k(const float* __restrict__ zs, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                   const float* __restrict__ zdyn, const float* __restrict__ lik, TransStd ts_,
                                                   float* __restrict__ part, int T, int o, int skip) {
  const int b = blockIdx.x, Ts = T - skip, rows = Ts * o;
  float tr = 0.0f, lq = 0.0f, lk = 0.0f, ls = 0.0f;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    const size_t e = (size_t)b * rows + r;
    const float* z = zs + e * 18;
    const float* m = mean + e * 18;
    const float* s = stdv + e * 18;
    const float* zd = zdyn + e * 16;
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const float u = (z[q] - m[q]) / s[q];
      lq += -0.5f * u * u - logf(s[q]) - 0.5f * kLog2Pi;
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      const float u = (z[2 + d] - zd[d]) / ts_.s[d];
      tr += -0.5f * u * u - logf(ts_.s[d]) - 0.5f * kLog2Pi;
    }
  }
  for (int j = threadIdx.x; j < T - 1; j += blockDim.x) {
    const float v = lik[(size_t)b * (T - 1) + j];
    if (j < skip - 1) ls += v;
    else lk += v;
  }
  __shared__ float red[4][4];
  float v4[4] = {tr - lq + lk, ls, tr, lq};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float w = wave_sum(v4[c]);
    if (lane_id() == 0) red[c][wave_id()] = w;
  }
  __syncthreads();
  if (threadIdx.x < 4) part[b * 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
// out = { average ELBO, mean trans_lik, mean log_q }: one block, fixed order
__global__ __launch_bounds__(256) void elbo_final_k(const float* __restrict__ part, float* __restrict__ out, int n, int T, int skip) {
  float a[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < n; b += blockDim.x)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] += part[b * 4 + c];
  __shared__ float red[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float w = wave_sum(a[c]);
    if (lane_id() == 0) red[c][wave_id()] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
    const float nt = (float)n * (float)(T - skip);
    out[0] = s[0] / nt + (skip > 1 ? s[1] / ((float)n * (float)(skip - 1)) : 0.0f);
    out[1] = s[2] / nt;
    out[2] = s[3] / nt;
  }
}
// g = dL/d(average ELBO) (device scalar): element-parallel over the (b, ts, k) rows, then the lik entries
__global__ __launch_bounds__(256) void elbo_bwd_k(const float* __restrict__ zs, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                  const float* __restrict__ zdyn, TransStd ts_, const float* __restrict__ gout, float* __restrict__ g_zs,
                                                  float* __restrict__ g_mean, float* __restrict__ g_std, float* __restrict__ g_zdyn, float* __restrict__ g_lik,
                                                  int n, int T, int o, int skip) {
  // one ELEMENT (row, q) per thread: every load and store of a wave is one contiguous 256-byte piece (a thread per row of 18 walked
  // them with a 72-byte stride: 16.5 us -> 9 us; the same change made the block-per-sequence reduction of elbo_part_k slower)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t rows = (size_t)n * (T - skip) * o, ne = rows * 18;
  const float g = gout[0];
  const float c1 = g / ((float)n * (float)(T - skip));
  if (i < ne) {
    const size_t r = i / 18;
    const int q = (int)(i - r * 18);
    const float z = zs[i];
    const float is = 1.0f / stdv[i], u = (z - mean[i]) * is;
    float gz = c1 * u * is;                                    // -logq: +c1 (z - m) / s^2
    g_mean[i] = -c1 * u * is;
    g_std[i] = -c1 * (u * u - 1.0f) * is;
    if (q >= 2) {
      const float it = 1.0f / ts_.s[q - 2], v = (z - zdyn[r * 16 + q - 2]) * it;
      gz -= c1 * v * it;
      g_zdyn[r * 16 + q - 2] = c1 * v * it;
    }
    g_zs[i] = gz;
  } else if (i < ne + (size_t)n * (T - 1)) {
    const size_t e = i - ne;
    const int j = (int)(e % (T - 1));
    g_lik[e] = (j < skip - 1) ? g / ((float)n * (float)(skip - 1)) : c1;
  }
}

}  // namespace stove
