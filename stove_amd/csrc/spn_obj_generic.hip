// Object RAT-SPN operator for ANY glimpse size and vector widths (round 4).  The reference builds the object SPN over
// c x patch_width x patch_height dimensions with six two-level random binary splits, obj_spn_num_gauss Gaussians per leaf and
// obj_spn_num_sums sum nodes per inner region (probabilistic_models.py:8-22; config.py:99-100, 119-120); the tuned kernels of
// spn_obj.hip are instantiated for its defaults (100 dimensions, 25-pixel leaves, 10 / 10).  These are the same RatSpn.forward /
// backward (rat_torch.py:83-109 leaves, :147-163 products, :202-222 sums, :354-357) with every size at run time:
//   replica r = one child of the root sum = product of two sum vectors (S nodes each), each over the product (G x G) of two
//   Gaussian leaves whose scopes split the replica's half of the D dimensions.
// Correctness first (the operator behind Supair.likelihood for other glimpse sizes / vector widths): one workgroup per sample
// for the forward and the data gradients, sample-contraction loops with one owner thread per table entry for the table
// gradients (fixed summation order, no atomics).
//
// Tables (made by RatSpn.tables, stove_amd/spn/rat_torch.py):
//   lscope [R*4][Lmax] int32   pixels of leaf (r, l), l = 2 j + side, padded with -1
//   slot   [R][D]      int32   l * Lmax + i of pixel p in replica r
//   coef   [R*4][Lmax][G][3]   (a, b, c): leaf log-density = sum_p w_p (a x^2 + b x + c)
//   wsum   [R*2][G*G][S]       softmaxed over G*G; product node i = g1 * G + g0 (input 0 <-> g0), rat_torch.py:147-163
//   wroot  [R][S*S]            softmaxed over all R*S*S; product node k1 * S + k0
#include "common.h"

namespace stove {

constexpr int kOaThreads = 256;
constexpr int kOaMaxG = 16, kOaMaxS = 16, kOaMaxR = 8, kOaMaxD = 1024;

struct ObjAnyShape {
  int R, G, S, D, Lmax;
  __host__ __device__ int n_ell() const { return R * 4 * G; }
  __host__ __device__ int n_s() const { return R * 2 * S; }
};

// saved per sample: [ ell (R*4*G) | s (R*2*S) | out ]
__host__ __device__ inline size_t oa_saved_stride(const ObjAnyShape& sh) { return (size_t)sh.n_ell() + sh.n_s() + 1; }

__device__ __forceinline__ float oa_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// out[e] = sum_c part[c * stride + e], c in order: the partial rows of the table-gradient kernel
__global__ __launch_bounds__(256) void reduce_rows_strided_k(const float* __restrict__ part, float* __restrict__ out, int n, int rows, int stride) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.0f;
  for (int c = 0; c < rows; ++c) s += part[(size_t)c * stride + e];
  out[e] = s;
}
__device__ __forceinline__ float oa_block_max(float v, float* red) {
  v = oa_wave_max(v);
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  float m = red[0];
  for (int i = 1; i < kOaThreads / 64; ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  return m;
}
__device__ __forceinline__ float oa_block_sum(float v, float* red) {
  v = wave_sum(v);
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  float s = red[0];
  for (int i = 1; i < kOaThreads / 64; ++i) s += red[i];
  __syncthreads();
  return s;
}

// ---- forward: one workgroup per sample ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kOaThreads) void objany_fwd_k(const float* __restrict__ inputs, const float* __restrict__ marg,
                                                           const int* __restrict__ lscope, const float* __restrict__ coef,
                                                           const float* __restrict__ wsum, const float* __restrict__ wroot,
                                                           float* __restrict__ saved, float* __restrict__ out, int n, ObjAnyShape sh) {
  extern __shared__ float oa_lds[];
  float* xs = oa_lds;                  // [D]
  float* ws = xs + sh.D;               // [D]
  float* ell = ws + sh.D;              // [R*4*G]
  float* sv = ell + sh.n_ell();        // [R*2*S]
  float* red = sv + sh.n_s();          // [4]
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int p = tid; p < sh.D; p += kOaThreads) {
    xs[p] = inputs[(size_t)b * sh.D + p];
    ws[p] = marg != nullptr ? 1.0f - fminf(fmaxf(marg[(size_t)b * sh.D + p], 0.0f), 1.0f) : 1.0f;
  }
  __syncthreads();
  for (int t = tid; t < sh.n_ell(); t += kOaThreads) {
    const int rl = t / sh.G, g = t % sh.G;
    float acc = 0.0f;
    for (int i = 0; i < sh.Lmax; ++i) {
      const int p = lscope[rl * sh.Lmax + i];
      if (p < 0) break;
      const float* c = coef + (((size_t)rl * sh.Lmax + i) * sh.G + g) * 3;
      const float x = xs[p];
      acc = fmaf(ws[p], fmaf(x, fmaf(c[0], x, c[1]), c[2]), acc);
    }
    ell[t] = acc;
  }
  __syncthreads();
  for (int t = tid; t < sh.n_s(); t += kOaThreads) {
    const int rj = t / sh.S, k = t % sh.S;
    const float* e0 = ell + (rj * 2) * sh.G;
    const float* e1 = e0 + sh.G;
    float m0 = e0[0], m1 = e1[0];
    for (int g = 1; g < sh.G; ++g) {
      m0 = fmaxf(m0, e0[g]);
      m1 = fmaxf(m1, e1[g]);
    }
    const float* w = wsum + (size_t)rj * sh.G * sh.G * sh.S + k;
    float acc = 0.0f;
    for (int g1 = 0; g1 < sh.G; ++g1) {
      const float a1 = __expf(e1[g1] - m1);
      float row = 0.0f;
      for (int g0 = 0; g0 < sh.G; ++g0) row = fmaf(w[(size_t)(g1 * sh.G + g0) * sh.S], __expf(e0[g0] - m0), row);
      acc = fmaf(a1, row, acc);
    }
    sv[t] = __logf(acc) + m0 + m1;
  }
  __syncthreads();
  // root: max over all products, then the weighted sum
  const int SS = sh.S * sh.S, NT = sh.R * SS;
  float m = -3.0e38f;
  for (int t = tid; t < NT; t += kOaThreads) {
    const int r = t / SS, k1 = (t % SS) / sh.S, k0 = t % sh.S;
    m = fmaxf(m, sv[(r * 2) * sh.S + k0] + sv[(r * 2 + 1) * sh.S + k1]);
  }
  m = oa_block_max(m, red);
  float acc = 0.0f;
  for (int t = tid; t < NT; t += kOaThreads) {
    const int r = t / SS, k1 = (t % SS) / sh.S, k0 = t % sh.S;
    acc = fmaf(wroot[t], __expf(sv[(r * 2) * sh.S + k0] + sv[(r * 2 + 1) * sh.S + k1] - m), acc);
  }
  acc = oa_block_sum(acc, red);
  const float o = __logf(acc) + m;
  float* sp = saved + (size_t)b * oa_saved_stride(sh);
  for (int t = tid; t < sh.n_ell(); t += kOaThreads) sp[t] = ell[t];
  for (int t = tid; t < sh.n_s(); t += kOaThreads) sp[sh.n_ell() + t] = sv[t];
  if (tid == 0) {
    sp[sh.n_ell() + sh.n_s()] = o;
    out[b] = o;
  }
}

// ---- backward, data side: one workgroup per sample -> ds, dell (kept for the table gradients), d_inputs, d_marg --------------------
// grads per sample: [ dell (R*4*G) | ds (R*2*S) ]
__global__ __launch_bounds__(kOaThreads) void objany_bwd_k(const float* __restrict__ inputs, const float* __restrict__ marg,
                                                           const int* __restrict__ slot, const float* __restrict__ coef,
                                                           const float* __restrict__ wsum, const float* __restrict__ wroot,
                                                           const float* __restrict__ saved, const float* __restrict__ dout,
                                                           float* __restrict__ grads, float* __restrict__ d_inputs,
                                                           float* __restrict__ d_marg, int n, ObjAnyShape sh) {
  extern __shared__ float oa_lds[];
  float* ell = oa_lds;                 // [R*4*G]
  float* sv = ell + sh.n_ell();        // [R*2*S]
  float* ds = sv + sh.n_s();           // [R*2*S]
  float* dell = ds + sh.n_s();         // [R*4*G]
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* sp = saved + (size_t)b * oa_saved_stride(sh);
  for (int t = tid; t < sh.n_ell(); t += kOaThreads) ell[t] = sp[t];
  for (int t = tid; t < sh.n_s(); t += kOaThreads) sv[t] = sp[sh.n_ell() + t];
  const float o = sp[sh.n_ell() + sh.n_s()], go = dout[b];
  __syncthreads();
  const int SS = sh.S * sh.S;
  for (int t = tid; t < sh.n_s(); t += kOaThreads) {
    const int rj = t / sh.S, k = t % sh.S, r = rj >> 1, j = rj & 1;
    const float* other = sv + (r * 2 + (j ^ 1)) * sh.S;
    const float mine = sv[t];
    float acc = 0.0f;
    for (int q = 0; q < sh.S; ++q) {
      const int k0 = j == 0 ? k : q, k1 = j == 0 ? q : k;
      acc = fmaf(wroot[r * SS + k1 * sh.S + k0], __expf(mine + other[q] - o), acc);
    }
    ds[t] = go * acc;
  }
  __syncthreads();
  for (int t = tid; t < sh.n_ell(); t += kOaThreads) {
    const int rl = t / sh.G, g = t % sh.G, rj = rl >> 1, side = rl & 1;
    const float* eo = ell + (rj * 2 + (side ^ 1)) * sh.G;
    const float mine = ell[t];
    const float* w = wsum + (size_t)rj * sh.G * sh.G * sh.S;
    float acc = 0.0f;
    for (int k = 0; k < sh.S; ++k) {
      const float base = mine - sv[rj * sh.S + k];
      float row = 0.0f;
      for (int q = 0; q < sh.G; ++q) {
        const int g0 = side == 0 ? g : q, g1 = side == 0 ? q : g;
        row = fmaf(w[(size_t)(g1 * sh.G + g0) * sh.S + k], __expf(base + eo[q]), row);
      }
      acc = fmaf(ds[rj * sh.S + k], row, acc);
    }
    dell[t] = acc;
  }
  __syncthreads();
  float* gp = grads + (size_t)b * (sh.n_ell() + sh.n_s());
  for (int t = tid; t < sh.n_ell(); t += kOaThreads) gp[t] = dell[t];
  for (int t = tid; t < sh.n_s(); t += kOaThreads) gp[sh.n_ell() + t] = ds[t];
  if (d_inputs == nullptr && d_marg == nullptr) return;
  for (int p = tid; p < sh.D; p += kOaThreads) {
    const float x = inputs[(size_t)b * sh.D + p];
    float mraw = 0.0f, w = 1.0f;
    if (marg != nullptr) {
      mraw = marg[(size_t)b * sh.D + p];
      w = 1.0f - fminf(fmaxf(mraw, 0.0f), 1.0f);
    }
    float dw = 0.0f, dx = 0.0f;
    for (int r = 0; r < sh.R; ++r) {
      const int sl = slot[r * sh.D + p];               // l * Lmax + i
      const int l = sl / sh.Lmax;
      const float* c = coef + ((size_t)(r * 4) * sh.Lmax + sl) * sh.G * 3;
      const float* de = dell + (r * 4 + l) * sh.G;
      for (int g = 0; g < sh.G; ++g) {
        dw = fmaf(de[g], fmaf(x, fmaf(c[g * 3], x, c[g * 3 + 1]), c[g * 3 + 2]), dw);
        dx = fmaf(de[g], fmaf(2.0f * c[g * 3], x, c[g * 3 + 1]), dx);
      }
    }
    if (d_marg != nullptr) d_marg[(size_t)b * sh.D + p] = (mraw >= 0.0f && mraw <= 1.0f) ? -dw : 0.0f;
    if (d_inputs != nullptr) d_inputs[(size_t)b * sh.D + p] = dx * w;
  }
}

// ---- backward, table side: workgroup c walks the samples c, c + chunks, ... in groups staged through LDS; every table entry has
// one owner thread, which sums its contributions of a group in sample order and adds them to the workgroup's partial row ----------------
constexpr int kOaGroup = 16;
__host__ __device__ inline size_t oa_n_coef(const ObjAnyShape& sh) { return (size_t)sh.R * 4 * sh.Lmax * sh.G * 3; }
__host__ __device__ inline size_t oa_n_wsum(const ObjAnyShape& sh) { return (size_t)sh.R * 2 * sh.G * sh.G * sh.S; }
__host__ __device__ inline size_t oa_n_wroot(const ObjAnyShape& sh) { return (size_t)sh.R * sh.S * sh.S; }
__host__ __device__ inline size_t oa_n_tab(const ObjAnyShape& sh) { return oa_n_coef(sh) + oa_n_wsum(sh) + oa_n_wroot(sh); }

__global__ __launch_bounds__(kOaThreads) void objany_tablegrad_k(const float* __restrict__ inputs, const float* __restrict__ marg,
                                                                 const int* __restrict__ lscope, const float* __restrict__ saved,
                                                                 const float* __restrict__ grads, const float* __restrict__ dout,
                                                                 float* __restrict__ part, int n, int chunks, ObjAnyShape sh) {
  extern __shared__ float oa_lds[];
  const int NE = sh.n_ell(), NS = sh.n_s();
  const int per = 2 * NE + 2 * NS + 2 + 2 * sh.D;           // ell, dell, s, ds, out, dout, x, w of one sample
  const int tid = threadIdx.x, c = blockIdx.x;
  float* mine = part + (size_t)c * oa_n_tab(sh);
  for (size_t e = tid; e < oa_n_tab(sh); e += kOaThreads) mine[e] = 0.0f;
  const size_t n_coef = oa_n_coef(sh), n_wsum = oa_n_wsum(sh), n_wroot = oa_n_wroot(sh);
  const int SS = sh.S * sh.S, GG = sh.G * sh.G;
  for (int first = c; first < n; first += chunks * kOaGroup) {
    int cnt = 0;
    for (int q = 0; q < kOaGroup; ++q)
      if (first + q * chunks < n) cnt = q + 1;
    __syncthreads();
    for (int q = 0; q < cnt; ++q) {
      const int b = first + q * chunks;
      float* d = oa_lds + (size_t)q * per;
      const float* sp = saved + (size_t)b * oa_saved_stride(sh);
      const float* gp = grads + (size_t)b * (NE + NS);
      for (int t = tid; t < NE; t += kOaThreads) {
        d[t] = sp[t];
        d[NE + t] = gp[t];
      }
      for (int t = tid; t < NS; t += kOaThreads) {
        d[2 * NE + t] = sp[NE + t];
        d[2 * NE + NS + t] = gp[NE + t];
      }
      if (tid == 0) {
        d[2 * NE + 2 * NS] = sp[NE + NS];
        d[2 * NE + 2 * NS + 1] = dout[b];
      }
      float* xw = d + 2 * NE + 2 * NS + 2;
      for (int p = tid; p < sh.D; p += kOaThreads) {
        xw[p] = inputs[(size_t)b * sh.D + p];
        xw[sh.D + p] = marg != nullptr ? 1.0f - fminf(fmaxf(marg[(size_t)b * sh.D + p], 0.0f), 1.0f) : 1.0f;
      }
    }
    __syncthreads();
    // leaf coefficients: entry (rl, i, g, e) <- sum_b dell[rl][g] w[p] (x^2, x, 1)[e]
    for (size_t e = tid; e < n_coef; e += kOaThreads) {
      const int f = (int)(e % 3), g = (int)((e / 3) % sh.G), i = (int)((e / 3 / sh.G) % sh.Lmax), rl = (int)(e / 3 / sh.G / sh.Lmax);
      const int p = lscope[rl * sh.Lmax + i];
      if (p < 0) continue;
      float acc = 0.0f;
      for (int q = 0; q < cnt; ++q) {
        const float* d = oa_lds + (size_t)q * per;
        const float* xw = d + 2 * NE + 2 * NS + 2;
        const float x = xw[p], w = xw[sh.D + p];
        acc = fmaf(d[NE + rl * sh.G + g], w * (f == 0 ? x * x : (f == 1 ? x : 1.0f)), acc);
      }
      mine[e] += acc;
    }
    // sum weights (linear domain): entry (rj, i = g1 G + g0, k) <- sum_b ds[rj][k] exp(ell0[g0] + ell1[g1] - s[rj][k])
    for (size_t e = tid; e < n_wsum; e += kOaThreads) {
      const int k = (int)(e % sh.S), i = (int)((e / sh.S) % GG), rj = (int)(e / sh.S / GG);
      const int g0 = i % sh.G, g1 = i / sh.G;
      float acc = 0.0f;
      for (int q = 0; q < cnt; ++q) {
        const float* d = oa_lds + (size_t)q * per;
        acc = fmaf(d[2 * NE + NS + rj * sh.S + k], __expf(d[(rj * 2) * sh.G + g0] + d[(rj * 2 + 1) * sh.G + g1] - d[2 * NE + rj * sh.S + k]), acc);
      }
      mine[n_coef + e] += acc;
    }
    // root weights: entry (r, k1 S + k0) <- sum_b dout exp(s0[k0] + s1[k1] - out)
    for (size_t e = tid; e < n_wroot; e += kOaThreads) {
      const int r = (int)(e / SS), k1 = (int)((e % SS) / sh.S), k0 = (int)(e % sh.S);
      float acc = 0.0f;
      for (int q = 0; q < cnt; ++q) {
        const float* d = oa_lds + (size_t)q * per;
        acc = fmaf(d[2 * NE + 2 * NS + 1], __expf(d[2 * NE + (r * 2) * sh.S + k0] + d[2 * NE + (r * 2 + 1) * sh.S + k1] - d[2 * NE + 2 * NS]), acc);
      }
      mine[n_coef + n_wsum + e] += acc;
    }
  }
}

// ---- the reference's fixed-Gaussian debug models (probabilistic_models.py:42-90, config.debug_bg_model / debug_obj_spn): every
// pixel scored under Normal(mean, scale), weighted by (1 - marg) as it comes (no clamp there), summed per row.  One wave per row.
__global__ __launch_bounds__(256) void gauss_ll_fwd_k(const float* __restrict__ x, const float* __restrict__ marg, float* __restrict__ out,
                                                      int n, int d, float mean, float scale) {
  const int row = blockIdx.x * (blockDim.x >> 6) + wave_id();
  if (row >= n) return;
  const float i2 = 0.5f / (scale * scale), c0 = -__logf(scale) - 0.5f * kLog2Pi;
  float acc = 0.0f;
  for (int p = lane_id(); p < d; p += 64) {
    const float u = x[(size_t)row * d + p] - mean;
    acc = fmaf(1.0f - marg[(size_t)row * d + p], fmaf(-u * u, i2, c0), acc);
  }
  acc = wave_sum(acc);
  if (lane_id() == 0) out[row] = acc;
}
__global__ __launch_bounds__(256) void gauss_ll_bwd_k(const float* __restrict__ x, const float* __restrict__ marg, const float* __restrict__ dout,
                                                      float* __restrict__ dx, float* __restrict__ dm, int n, int d, float mean, float scale) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * d) return;
  const float g = dout[t / d], u = x[t] - mean;
  const float i2 = 0.5f / (scale * scale), c0 = -__logf(scale) - 0.5f * kLog2Pi;
  if (dx != nullptr) dx[t] = g * (1.0f - marg[t]) * (-2.0f * i2 * u);
  if (dm != nullptr) dm[t] = -g * fmaf(-u * u, i2, c0);
}

static inline bool oa_shape_ok(const ObjAnyShape& sh) {
  if (!(sh.R >= 1 && sh.R <= kOaMaxR && sh.G >= 1 && sh.G <= kOaMaxG && sh.S >= 1 && sh.S <= kOaMaxS && sh.D >= 4 && sh.D <= kOaMaxD &&
        sh.Lmax >= 1 && sh.Lmax <= sh.D))
    return false;
  // the table-gradient kernel stages kOaGroup samples in LDS: a shape whose backward cannot run is refused by the forward already
  // (before round 5 the forward ran and the first backward failed with hipErrorInvalidValue after objany_bwd_k had been launched)
  return sizeof(float) * (size_t)kOaGroup * (2 * sh.n_ell() + 2 * sh.n_s() + 2 + 2 * (size_t)sh.D) <= 160u * 1024u;
}
static inline int oa_chunks(int n) {
  const int c = (n + kOaGroup - 1) / kOaGroup;
  return c < 1 ? 1 : (c > 256 ? 256 : c);
}
size_t objany_saved_floats(int n, const ObjAnyShape& sh) { return (size_t)n * oa_saved_stride(sh); }
// ws of the backward: [ grads (n x (NE + NS)) | partial rows (chunks x n_tab) ]
size_t objany_bwd_ws_floats(int n, const ObjAnyShape& sh) {
  return (size_t)n * (sh.n_ell() + sh.n_s()) + (size_t)oa_chunks(n) * oa_n_tab(sh);
}

int objany_forward(const float* inputs, const float* marg, const int* lscope, const float* coef, const float* wsum, const float* wroot,
                   float* saved, float* out, int n, const ObjAnyShape& sh, hipStream_t st) {
  if (n == 0) return 0;
  if (!oa_shape_ok(sh)) return (int)hipErrorInvalidValue;
  const size_t lds = sizeof(float) * (2 * (size_t)sh.D + sh.n_ell() + sh.n_s() + 8);
  STOVE_LAUNCH(objany_fwd_k, dim3(n), dim3(kOaThreads), lds, st, inputs, marg, lscope, coef, wsum, wroot, saved, out, n, sh);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// g_coef [R*4][Lmax][G][3], g_wsum [R*2][G*G][S], g_wroot [R][S*S]: overwritten (padded leaf rows: zero)
int objany_backward(const float* inputs, const float* marg, const int* lscope, const int* slot, const float* coef, const float* wsum,
                    const float* wroot, const float* saved, const float* dout, float* d_inputs, float* d_marg, float* g_coef,
                    float* g_wsum, float* g_wroot, float* ws, int n, const ObjAnyShape& sh, hipStream_t st) {
  if (!oa_shape_ok(sh)) return (int)hipErrorInvalidValue;
  if (n == 0) {
    hipMemsetAsync(g_coef, 0, sizeof(float) * oa_n_coef(sh), st);
    hipMemsetAsync(g_wsum, 0, sizeof(float) * oa_n_wsum(sh), st);
    hipMemsetAsync(g_wroot, 0, sizeof(float) * oa_n_wroot(sh), st);
    return 0;
  }
  float* grads = ws;
  float* part = ws + (size_t)n * (sh.n_ell() + sh.n_s());
  const size_t lds_b = sizeof(float) * (2 * (size_t)sh.n_ell() + 2 * sh.n_s() + 8);
  STOVE_LAUNCH(objany_bwd_k, dim3(n), dim3(kOaThreads), lds_b, st, inputs, marg, slot, coef, wsum, wroot, saved, dout, grads, d_inputs, d_marg, n, sh);
  STOVE_LAUNCH_CHECK();
  const int chunks = oa_chunks(n);
  const size_t lds_t = sizeof(float) * (size_t)kOaGroup * (2 * sh.n_ell() + 2 * sh.n_s() + 2 + 2 * (size_t)sh.D);
  if (lds_t > 160 * 1024) return (int)hipErrorInvalidValue;
  int rc = (int)hipFuncSetAttribute((const void*)objany_tablegrad_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
  if (rc) return rc;
  STOVE_LAUNCH(objany_tablegrad_k, dim3(chunks), dim3(kOaThreads), lds_t, st, inputs, marg, lscope, saved, (const float*)grads, dout, part, n, chunks, sh);
  STOVE_LAUNCH_CHECK();
  const size_t nc = oa_n_coef(sh), nw = oa_n_wsum(sh), nr = oa_n_wroot(sh), nt = oa_n_tab(sh);
  // the partial rows hold [coef | wsum | wroot] back to back with row stride nt: three strided reductions
  STOVE_LAUNCH(reduce_rows_strided_k, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, (const float*)part, g_coef, (int)nc, chunks, (int)nt);
  STOVE_LAUNCH(reduce_rows_strided_k, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, (const float*)part + nc, g_wsum, (int)nw, chunks, (int)nt);
  STOVE_LAUNCH(reduce_rows_strided_k, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, (const float*)part + nc + nw, g_wroot, (int)nr, chunks, (int)nt);
  STOVE_LAUNCH_CHECK();
  return 0;
}

}  // namespace stove
