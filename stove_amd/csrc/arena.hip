// Flat parameter arena support: bake the SPN tables and the GNN parameter image straight from the
// flat fp32 parameter buffer, and push table / image gradients straight back into the flat gradient
// buffer (which is also the data-parallel all-reduce bucket).
//
// With the reference-layout parameters (162 separate tensors: vector_list.L.i.means ..., dyn.*.weight)
// as views into one arena, this replaces ~250 small ATen launches per training step (stack / index /
// sigmoid / softmax and their backward, the gradient clones of AccumulateGrad) by four launches.
#include "common.h"

namespace stove {

// plan of the SPN part (all int32, device): offsets are float offsets into the arena
//   obj_mu[24], obj_rho[24] (kernel leaf order), obj_sum[12] (kernel order), obj_root
//   bg_mu[6], bg_rho[6] (reference leaf order), bg_root, bg_gidx[3*1024] = leaf*512 + row
struct SpnArenaPlan {
  const int* obj_mu;
  const int* obj_rho;
  const int* obj_sum;
  const int* bg_mu;
  const int* bg_rho;
  const int* bg_gidx;
  int obj_root, bg_root;
  float obj_vmin, obj_vmax, bg_vmin, bg_vmax;
};

__device__ __forceinline__ void leaf_coef(float mu, float rho, float vmin, float vmax, float* abc) {
  const float v = vmin + (vmax - vmin) * (1.0f / (1.0f + expf(-rho)));
  const float iv = 1.0f / v;
  abc[0] = -0.5f * iv;
  abc[1] = mu * iv;
  abc[2] = -0.5f * mu * mu * iv - 0.5f * logf(6.283185307179586f * v);
}
__device__ __forceinline__ void leaf_coef_bwd(float mu, float rho, float vmin, float vmax, const float* g, float* dmu, float* drho) {
  const float s = 1.0f / (1.0f + expf(-rho));
  const float v = vmin + (vmax - vmin) * s;
  const float iv = 1.0f / v;
  *dmu = g[1] * iv - g[2] * mu * iv;
  const float dv = g[0] * 0.5f * iv * iv - g[1] * mu * iv * iv + g[2] * (0.5f * mu * mu * iv * iv - 0.5f * iv);
  *drho = dv * (vmax - vmin) * s * (1.0f - s);
}

constexpr int kAObjLeaves = 24, kAObjS = 25, kAObjG = 10, kAObjSums = 12, kAObjK = 10;
constexpr int kABgR = 3, kABgPix = 1024, kABgG = 6;
constexpr int kAObjCoef = kAObjLeaves * kAObjS * kAObjG;    // 6000 (x3)
constexpr int kABgCoef = kABgR * kABgPix * kABgG;           // 18432 (x3)
constexpr int kASoftBlocks = (kAObjSums * kAObjK + 2 + 3) / 4;

// column softmax over n rows with row stride ld: one 64-lane wave per column
__device__ __forceinline__ void softmax_col(const float* __restrict__ p, float* __restrict__ w, int n, int ld) {
  const int lane = lane_id();
  float m = -3.0e38f;
  for (int k = lane; k < n; k += 64) m = fmaxf(m, p[k * ld]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.0f;
  for (int k = lane; k < n; k += 64) s += expf(p[k * ld] - m);
  s = wave_sum(s);
  const float inv = 1.0f / s;
  for (int k = lane; k < n; k += 64) w[k * ld] = expf(p[k * ld] - m) * inv;
}
// dp_k += w_k (g_k - sum_j w_j g_j)
__device__ __forceinline__ void softmax_col_bwd(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ dp, int n, int ld) {
  const int lane = lane_id();
  float m = -3.0e38f;
  for (int k = lane; k < n; k += 64) m = fmaxf(m, p[k * ld]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.0f, d = 0.0f;
  for (int k = lane; k < n; k += 64) {
    const float e = expf(p[k * ld] - m);
    s += e;
    d = fmaf(e, g[k * ld], d);
  }
  s = wave_sum(s);
  d = wave_sum(d);
  const float inv = 1.0f / s, dot = d * inv;
  for (int k = lane; k < n; k += 64) dp[k * ld] += expf(p[k * ld] - m) * inv * (g[k * ld] - dot);
}

// grid: ceil((6000 + 18432) / 256) element blocks, then 31 blocks of 4 waves, one wave per softmax column
// (120 sum-node columns + the two roots)
__global__ __launch_bounds__(256) void spn_bake_k(const float* __restrict__ arena, SpnArenaPlan pl, float* __restrict__ obj_coef,
                                                  float* __restrict__ obj_wsum, float* __restrict__ obj_wroot,
                                                  float* __restrict__ bg_coef, float* __restrict__ bg_wroot, int n_elem_blocks) {
  if ((int)blockIdx.x < n_elem_blocks) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < kAObjCoef) {
      const int q = t / (kAObjS * kAObjG), rem = t % (kAObjS * kAObjG);
      leaf_coef(arena[pl.obj_mu[q] + rem], arena[pl.obj_rho[q] + rem], pl.obj_vmin, pl.obj_vmax, obj_coef + (size_t)t * 3);
    } else if (t < kAObjCoef + kABgCoef) {
      const int u = t - kAObjCoef;
      const int rp = u / kABgG, g = u % kABgG;
      const int gi = pl.bg_gidx[rp];
      const int leaf = gi >> 9, row = gi & 511;
      leaf_coef(arena[pl.bg_mu[leaf] + row * kABgG + g], arena[pl.bg_rho[leaf] + row * kABgG + g], pl.bg_vmin, pl.bg_vmax,
                bg_coef + (size_t)u * 3);
    }
    return;
  }
  const int col = ((int)blockIdx.x - n_elem_blocks) * 4 + wave_id();
  if (col < kAObjSums * kAObjK + 2) {
    if (col < kAObjSums * kAObjK) {
      const int node = col / kAObjK, s = col % kAObjK;
      softmax_col(arena + pl.obj_sum[node] + s, obj_wsum + (size_t)node * 100 * kAObjK + s, 100, kAObjK);
    } else if (col == kAObjSums * kAObjK) {
      softmax_col(arena + pl.obj_root, obj_wroot, 600, 1);
    } else {
      softmax_col(arena + pl.bg_root, bg_wroot, kABgR * kABgG * kABgG, 1);
    }
  }
}

__global__ __launch_bounds__(256) void spn_bake_bwd_k(const float* __restrict__ arena, SpnArenaPlan pl,
                                                      const float* __restrict__ g_obj_coef, const float* __restrict__ g_obj_wsum,
                                                      const float* __restrict__ g_obj_wroot, const float* __restrict__ g_bg_coef,
                                                      const float* __restrict__ g_bg_wroot, float* __restrict__ garena,
                                                      int n_elem_blocks) {
  if ((int)blockIdx.x < n_elem_blocks) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < kAObjCoef) {
      const int q = t / (kAObjS * kAObjG), rem = t % (kAObjS * kAObjG);
      float dmu, drho;
      leaf_coef_bwd(arena[pl.obj_mu[q] + rem], arena[pl.obj_rho[q] + rem], pl.obj_vmin, pl.obj_vmax, g_obj_coef + (size_t)t * 3, &dmu, &drho);
      garena[pl.obj_mu[q] + rem] += dmu;
      garena[pl.obj_rho[q] + rem] += drho;
    } else if (t < kAObjCoef + kABgCoef) {
      const int u = t - kAObjCoef;
      const int rp = u / kABgG, g = u % kABgG;
      const int gi = pl.bg_gidx[rp];
      const int leaf = gi >> 9, row = gi & 511;
      const int om = pl.bg_mu[leaf] + row * kABgG + g, orh = pl.bg_rho[leaf] + row * kABgG + g;
      float dmu, drho;
      leaf_coef_bwd(arena[om], arena[orh], pl.bg_vmin, pl.bg_vmax, g_bg_coef + (size_t)u * 3, &dmu, &drho);
      garena[om] += dmu;
      garena[orh] += drho;
    }
    return;
  }
  const int col = ((int)blockIdx.x - n_elem_blocks) * 4 + wave_id();
  if (col < kAObjSums * kAObjK + 2) {
    if (col < kAObjSums * kAObjK) {
      const int node = col / kAObjK, s = col % kAObjK;
      softmax_col_bwd(arena + pl.obj_sum[node] + s, g_obj_wsum + (size_t)node * 100 * kAObjK + s, garena + pl.obj_sum[node] + s, 100, kAObjK);
    } else if (col == kAObjSums * kAObjK) {
      softmax_col_bwd(arena + pl.obj_root, g_obj_wroot, garena + pl.obj_root, 600, 1);
    } else {
      softmax_col_bwd(arena + pl.bg_root, g_bg_wroot, garena + pl.bg_root, kABgR * kABgG * kABgG, 1);
    }
  }
}

// image[i] = src[i] >= 0 ? arena[src[i]] : 0      /      garena[src[i]] += gimage[i]
__global__ void arena_gather_k(const float* __restrict__ arena, const int* __restrict__ src, float* __restrict__ image, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) image[i] = src[i] >= 0 ? arena[src[i]] : 0.0f;
}
__global__ void arena_scatter_add_k(const float* __restrict__ gimage, const int* __restrict__ src, float* __restrict__ garena, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && src[i] >= 0) garena[src[i]] += gimage[i];
}

}  // namespace stove

namespace stove {

// Adam / AMSGrad step of torch.optim.Adam (weight_decay = 0) over the flat arena, with clip_grad_norm_ folded in:
//   g' = g * min(1, max_norm / (norm + 1e-6))        (norm: device scalar, null = no clipping)
//   m = b1 m + (1 - b1) g' ; v = b2 v + (1 - b2) g'^2 ; vmax = max(vmax, v)   (vmax null = plain Adam)
//   p -= (lr / bc1) * m / (sqrt(vmax or v) / sqrt(bc2) + eps)
// ---- Adam / AMSGrad over the flat arena with torch.optim.Adam's per-parameter semantics.
// torch keeps one step count per parameter TENSOR and skips tensors whose .grad is None (frozen ones, and ones that took no
// part in the backward pass).  In the arena every gradient is a slice of one buffer, so "no gradient" shows up as a slice
// that is exactly zero.  A SEGMENT = one parameter tensor (every tensor starts on a float4 boundary):
//   grad_scan_k   flags[seg] = 1 for every segment with a non-zero gradient element; part[block] = the block's share of
//                 sum g^2 (fixed grid-stride assignment, fixed reduction order: bitwise reproducible);
//   flat_adam_k   segments that are trainable AND flagged take one step with THEIR step count (bias corrections from
//                 steps[seg] + 1, in double like torch's host arithmetic); all others are not touched at all;
//   adam_tick_k   steps[seg] += 1 for the segments that stepped, flags cleared for the next call.
// hyper = [lr, beta1, beta2, eps, max_norm] is read from device memory when hyper_dev != NULL (captured hipGraphs: kernel
// arguments are frozen at capture, the learning-rate schedule is not), else from the by-value copy.
struct AdamHyper {
  float lr, b1, b2, eps, max_norm;
};
constexpr int ADAM_SCAN_BLOCKS = 256;

__global__ __launch_bounds__(256) void grad_scan_k(const float* __restrict__ g, const int* __restrict__ seg_of4, int* __restrict__ flags,
                                                   float* __restrict__ part, int n4) {
  float acc = 0.0f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += ADAM_SCAN_BLOCKS * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    if (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f) flags[seg_of4[i]] = 1;      // every writer stores the same value
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void flat_adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ vmax, const int* __restrict__ seg_of4,
                                                   const unsigned char* __restrict__ trainable, const float* __restrict__ steps,
                                                   const int* __restrict__ flags, const float* __restrict__ part, float* __restrict__ norm_out,
                                                   const float* __restrict__ hyper_dev, AdamHyper k, int clip, int n4) {
  if (hyper_dev != nullptr) {
    k.lr = hyper_dev[0]; k.b1 = hyper_dev[1]; k.b2 = hyper_dev[2]; k.eps = hyper_dev[3]; k.max_norm = hyper_dev[4];
  }
  __shared__ float s_norm;
  if (threadIdx.x < 64) {                   // total gradient norm: the 256 partials in a fixed order, by every block alike
    float t = (part[threadIdx.x] + part[threadIdx.x + 64]) + (part[threadIdx.x + 128] + part[threadIdx.x + 192]);
    t = wave_sum(t);
    if (threadIdx.x == 0) {
      s_norm = sqrtf(t);
      if (blockIdx.x == 0 && norm_out != nullptr) norm_out[0] = s_norm;
    }
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int seg = seg_of4[i];
  if (trainable[seg] == 0 || flags[seg] == 0) return;
  const float coef = clip ? fminf(1.0f, k.max_norm / (s_norm + 1e-6f)) : 1.0f;
  const double t = (double)steps[seg] + 1.0;
  const float bc1 = (float)(1.0 - pow((double)k.b1, t));
  const float sqrt_bc2 = (float)sqrt(1.0 - pow((double)k.b2, t));
  const float4 g4 = reinterpret_cast<const float4*>(g)[i];
  float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
  float4 x4 = vmax != nullptr ? reinterpret_cast<float4*>(vmax)[i] : v4;
  const float step = k.lr / bc1;
#define STOVE_ADAM1(c)                                           \
  {                                                              \
    const float gg = g4.c * coef;                                \
    m4.c = m4.c + (1.0f - k.b1) * (gg - m4.c);                   \
    v4.c = k.b2 * v4.c + (1.0f - k.b2) * gg * gg;                \
    x4.c = vmax != nullptr ? fmaxf(x4.c, v4.c) : v4.c;           \
    p4.c -= step * m4.c / (sqrtf(x4.c) / sqrt_bc2 + k.eps);      \
  }
  STOVE_ADAM1(x) STOVE_ADAM1(y) STOVE_ADAM1(z) STOVE_ADAM1(w)
#undef STOVE_ADAM1
  reinterpret_cast<float4*>(p)[i] = p4;
  reinterpret_cast<float4*>(m)[i] = m4;
  reinterpret_cast<float4*>(v)[i] = v4;
  if (vmax != nullptr) reinterpret_cast<float4*>(vmax)[i] = x4;
}

// sticky: a segment that has once received a gradient keeps its flag (torch.optim.Adam on `.grad` tensors that zero_grad() filled
// with zeros instead of dropping: the momentum keeps moving the parameter -- the torch 1.0.1 behaviour the reference pins)
__global__ void adam_tick_k(const unsigned char* __restrict__ trainable, float* __restrict__ steps, int* __restrict__ flags, int nseg, int sticky) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  if (trainable[s] != 0 && flags[s] != 0) steps[s] += 1.0f;
  if (!sticky) flags[s] = 0;
}

// out[j] = sum_c part[c][j], fixed order; n4 = n / 4 (split-K partials of the batched weight-gradient GEMMs)
// bias4 / cols4: a row bias (cols4 float4 per output row) added to the sum; add4: an element-wise term (same shape as out, another
// tensor) -- the epilogue terms of a split-K GEMM
__global__ void sum_chunks4_k(const float* __restrict__ part, float* __restrict__ out, int n4, int chunks, int accumulate = 0,
                              const float* __restrict__ bias4 = nullptr, int cols4 = 0, const float* __restrict__ add4 = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = reinterpret_cast<const float4*>(part)[i];
  if (add4 != nullptr) {
    const float4 b = reinterpret_cast<const float4*>(add4)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  if (bias4 != nullptr) {
    const float4 b = reinterpret_cast<const float4*>(bias4)[i % cols4];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  if (accumulate) {           // out += sum (a gradient view that may already hold a contribution)
    const float4 o = reinterpret_cast<const float4*>(out)[i];
    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
  }
  for (int c = 1; c < chunks; ++c) {
    const float4 b = reinterpret_cast<const float4*>(part)[(size_t)c * n4 + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  reinterpret_cast<float4*>(out)[i] = a;
}

// The same for many chunks (split-K with 16+ slices): 256 threads = 64 float4 elements x 4 chunk slices, slice q adds chunks
// q, q + 4, ..; the four slice sums are added in slice order (one thread walking 32 slices of a 1 MB product serially was
// latency-bound: 50 us for 33 MB).
__global__ __launch_bounds__(256) void sum_chunks4_par_k(const float* __restrict__ part, float* __restrict__ out, int n4, int chunks, int accumulate) {
  __shared__ float4 red[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  float4 a = {0.0f, 0.0f, 0.0f, 0.0f};
  if (i < n4)
    for (int c = q; c < chunks; c += 4) {
      const float4 b = reinterpret_cast<const float4*>(part)[(size_t)c * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  red[q][e] = a;
  __syncthreads();
  if (q == 0 && i < n4) {
    float4 t = red[0][e];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      t.x += red[k][e].x; t.y += red[k][e].y; t.z += red[k][e].z; t.w += red[k][e].w;
    }
    if (accumulate) {
      const float4 o = reinterpret_cast<const float4*>(out)[i];
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    reinterpret_cast<float4*>(out)[i] = t;
  }
}

// out (M, N) = a^T b over the rows of a (rows, M) and b (rows, N) with M * N <= 256: weight gradient of a narrow linear layer
// (the action embedding 9 -> 12, the reward heads) over ~25 000 rows.  As a library GEMM this is a 16 x 16 x 256 tile on one
// or sixteen workgroups (83 us); here one thread per output walks a chunk of rows (both operands' rows are L1 lines shared by
// the whole block), stage 2 is reduce_chunks_k.
__global__ __launch_bounds__(256) void small_tn_part_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part,
                                                        int rows, int M, int N, int rows_per_chunk) {
  const int t = threadIdx.x;
  if (t >= M * N) return;
  const int m = t / N, n = t % N;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float s0 = 0.0f, s1 = 0.0f;
  int r = r0;
  for (; r + 1 < r1; r += 2) {
    s0 = fmaf(a[(size_t)r * M + m], b[(size_t)r * N + n], s0);
    s1 = fmaf(a[(size_t)(r + 1) * M + m], b[(size_t)(r + 1) * N + n], s1);
  }
  if (r < r1) s0 = fmaf(a[(size_t)r * M + m], b[(size_t)r * N + n], s0);
  part[(size_t)blockIdx.x * M * N + t] = s0 + s1;
}

// Column sums of a row-major (rows, cols) matrix, stage 1: part[chunk][col] = sum of the chunk's rows (coalesced
// float4 row reads); stage 2 is sum_chunks4_k.  (ATen's reduce_kernel runs this 105 MB reduction at 0.46 TB/s.)
__global__ __launch_bounds__(256) void colsum_part_k(const float* __restrict__ a, float* __restrict__ part, int rows, int cols4, int rows_per_chunk) {
  const int c4 = blockIdx.y * 256 + threadIdx.x;
  if (c4 >= cols4) return;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
  int r = r0;
  for (; r + 1 < r1; r += 2) {
    const float4 u = reinterpret_cast<const float4*>(a)[(size_t)r * cols4 + c4];
    const float4 v = reinterpret_cast<const float4*>(a)[(size_t)(r + 1) * cols4 + c4];
    s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
  }
  if (r < r1) {
    const float4 u = reinterpret_cast<const float4*>(a)[(size_t)r * cols4 + c4];
    s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
  }
  s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
  reinterpret_cast<float4*>(part)[(size_t)blockIdx.x * cols4 + c4] = s0;
}

// The same for narrow matrices (cols <= 64, any cols): thread = (row slot, column), row slots reduced through LDS.
__global__ __launch_bounds__(256) void colsum_narrow_part_k(const float* __restrict__ a, float* __restrict__ part, int rows, int cols, int cpad,
                                                            int rows_per_chunk) {
  __shared__ float red[256];
  const int c = threadIdx.x % cpad, rr = threadIdx.x / cpad, nslot = 256 / cpad;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float s = 0.0f;
  if (c < cols)
    for (int r = r0 + rr; r < r1; r += nslot) s += a[(size_t)r * cols + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rr == 0 && c < cols) {
    float t = 0.0f;
    for (int q = 0; q < nslot; ++q) t += red[q * cpad + c];
    part[(size_t)blockIdx.x * cols + c] = t;
  }
}

// bw_transform (reference utils.py:8-13): (N, C, P) one-ball-per-channel frames -> (N, P) = clamp(sum_c, 0, 1); P % 4 == 0
__global__ void bw_transform_k(const float* __restrict__ x, float* __restrict__ out, int N, int C, int P4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * P4) return;
  const int n = i / P4, p = i % P4;
  float4 s = reinterpret_cast<const float4*>(x)[((size_t)n * C) * P4 + p];
  for (int c = 1; c < C; ++c) {
    const float4 v = reinterpret_cast<const float4*>(x)[((size_t)n * C + c) * P4 + p];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  s.x = fminf(fmaxf(s.x, 0.0f), 1.0f);
  s.y = fminf(fmaxf(s.y, 0.0f), 1.0f);
  s.z = fminf(fmaxf(s.z, 0.0f), 1.0f);
  s.w = fminf(fmaxf(s.w, 0.0f), 1.0f);
  reinterpret_cast<float4*>(out)[i] = s;
}

// the same from an 8-bit frame store (load_data.DeviceClipLoader, frame_store='u8'): x (N, C, P) uint8 holding round(255 v);
// out = clamp(sum_c x_c / 255, 0, 1) -- the division per channel, summed in channel order, as the reference's
// bw_transform computes it on the float frames u8 / 255 (utils.py:10-15).  One uchar4 (4 pixels) per channel and thread.
__global__ void bw_transform_u8_k(const unsigned char* __restrict__ x, float* __restrict__ out, int N, int C, int P4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * P4) return;
  const int n = i / P4, p = i % P4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c < C; ++c) {
    const uchar4 v = reinterpret_cast<const uchar4*>(x)[((size_t)n * C + c) * P4 + p];
    const float4 f = make_float4((float)v.x / 255.0f, (float)v.y / 255.0f, (float)v.z / 255.0f, (float)v.w / 255.0f);
    if (c == 0) s = f;
    else { s.x += f.x; s.y += f.y; s.z += f.z; s.w += f.w; }
  }
  s.x = fminf(fmaxf(s.x, 0.0f), 1.0f);
  s.y = fminf(fmaxf(s.y, 0.0f), 1.0f);
  s.z = fminf(fmaxf(s.z, 0.0f), 1.0f);
  s.w = fminf(fmaxf(s.w, 0.0f), 1.0f);
  reinterpret_cast<float4*>(out)[i] = s;
}

}  // namespace stove
