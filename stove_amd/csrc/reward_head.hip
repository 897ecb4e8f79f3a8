// Reward head of the action-conditioned dynamics model (reference dynamics.py:254-263, 62-70):
//     reward = sigmoid( head1( sum_objects head0(dynamic_pred) ) ),
//     head0 = Linear(32,32) - ReLU - Linear(32,32),   head1 = Linear(32,16) - ReLU - Linear(16,8) - ReLU - Linear(8,1)
// as ONE kernel each way instead of ~10 / ~25 library launches on (B (T-2) N, 32) rows (the action-conditioned step was the last part
// of SURVEY 8 row A8 still running on library GEMMs).  One wave per (sequence, step) item, lane l < 32 = element l of a 32-wide vector; a
// layer is the input vector through a per-wave LDS scratch, read back as broadcasts, against the lane's weight column / row in LDS.
// Parameters come packed as the layers' own tensors one after the other (nn.Linear layout, weight (out, in) row-major):
//   [W0a 1024 | b0a 32 | W0b 1024 | b0b 32 | W1a 512 | b1a 16 | W1b 128 | b1b 8 | W1c 8 | b1c 1]  = kRhParams floats,
// the gradient comes back in the same layout (per-wave partial sums reduced in a fixed order: bitwise reproducible).
#include "common.h"

namespace stove {

constexpr int RH_W0A = 0, RH_B0A = 1024, RH_W0B = 1056, RH_B0B = 2080, RH_W1A = 2112, RH_B1A = 2624, RH_W1B = 2640, RH_B1B = 2768, RH_W1C = 2776,
              RH_B1C = 2784, kRhParams = 2785;
constexpr int kRhWaves = 4;

// y[l] = sum_k Wt[k * OUT + l] x[k]  (lane l < OUT), x broadcast from the wave's LDS scratch
template <int K, int OUT>
__device__ __forceinline__ float rh_dot_t(const float* Wt, const float* xs, int l) {
  float a = 0.0f, b = 0.0f;
#pragma unroll
  for (int k = 0; k < K; k += 4) {
    const float4 x4 = *reinterpret_cast<const float4*>(xs + k);
    const int ll = l < OUT ? l : 0;
    a = fmaf(Wt[(k + 0) * OUT + ll], x4.x, a);
    b = fmaf(Wt[(k + 1) * OUT + ll], x4.y, b);
    a = fmaf(Wt[(k + 2) * OUT + ll], x4.z, a);
    b = fmaf(Wt[(k + 3) * OUT + ll], x4.w, b);
  }
  return a + b;
}
// dx[k] = sum_l W[l * K + k] dy[l]  (lane k < K), dy broadcast from the scratch; W row-major (OUT, K)
template <int K, int OUT>
__device__ __forceinline__ float rh_dot_n(const float* W, const float* dys, int k) {
  float a = 0.0f, b = 0.0f;
  const int kk = k < K ? k : 0;
#pragma unroll
  for (int l = 0; l < OUT; l += 4) {
    const float4 d4 = *reinterpret_cast<const float4*>(dys + l);
    a = fmaf(W[(l + 0) * K + kk], d4.x, a);
    b = fmaf(W[(l + 1) * K + kk], d4.y, b);
    a = fmaf(W[(l + 2) * K + kk], d4.z, a);
    b = fmaf(W[(l + 3) * K + kk], d4.w, b);
  }
  return a + b;
}

// pred (items, N, 32) -> reward (items); saved for the backward: H0 (items, N, 32) = relu(head0.0), Q (items, 32), A1 (items, 16), A2 (items, 8)
__global__ __launch_bounds__(64 * kRhWaves) void reward_head_fwd_k(const float* __restrict__ pred, const float* __restrict__ P, float* __restrict__ reward,
                                                                     float* __restrict__ H0, float* __restrict__ Q, float* __restrict__ A1,
                                                                     float* __restrict__ A2, int items, int N) {
  __shared__ __attribute__((aligned(16))) float Wt0a[1024], Wt0b[1024], Wt1a[512], Wt1b[128], V[32 + 32 + 16 + 8 + 8 + 4], xs[kRhWaves][32];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l = lane & 31;
  for (int i = tid; i < 1024; i += blockDim.x) {
    const int o = i >> 5, k = i & 31;                    // W (out o, in k) -> Wt[k][o]
    Wt0a[k * 32 + o] = P[RH_W0A + i];
    Wt0b[k * 32 + o] = P[RH_W0B + i];
  }
  for (int i = tid; i < 512; i += blockDim.x) Wt1a[(i & 31) * 16 + (i >> 5)] = P[RH_W1A + i];      // (16, 32) -> [k][16]
  for (int i = tid; i < 128; i += blockDim.x) Wt1b[(i & 15) * 8 + (i >> 4)] = P[RH_W1B + i];       // (8, 16)  -> [k][8]
  float* b0a = V; float* b0b = V + 32; float* b1a = V + 64; float* b1b = V + 80; float* w1c = V + 88; float* b1c = V + 96;
  for (int i = tid; i < 32; i += blockDim.x) { b0a[i] = P[RH_B0A + i]; b0b[i] = P[RH_B0B + i]; }
  for (int i = tid; i < 16; i += blockDim.x) b1a[i] = P[RH_B1A + i];
  for (int i = tid; i < 8; i += blockDim.x) { b1b[i] = P[RH_B1B + i]; w1c[i] = P[RH_W1C + i]; }
  if (tid == 0) b1c[0] = P[RH_B1C];
  __syncthreads();
  float* x = xs[wv];
  const bool act = lane < 32;
  for (int it = blockIdx.x * kRhWaves + wv; it < items; it += gridDim.x * kRhWaves) {
    float q = 0.0f;
    for (int o = 0; o < N; ++o) {
      const size_t row = ((size_t)it * N + o) * 32;
      if (act) x[l] = pred[row + l];
      float h = b0a[l] + rh_dot_t<32, 32>(Wt0a, x, l);
      h = fmaxf(h, 0.0f);
      if (act) {
        H0[row + l] = h;
        x[l] = h;
      }
      q += b0b[l] + rh_dot_t<32, 32>(Wt0b, x, l);
    }
    if (act) {
      Q[(size_t)it * 32 + l] = q;
      x[l] = q;
    }
    float a = fmaxf(b1a[l & 15] + rh_dot_t<32, 16>(Wt1a, x, l), 0.0f);
    if (lane < 16) {
      A1[(size_t)it * 16 + l] = a;
      x[l] = a;
    }
    float c = fmaxf(b1b[l & 7] + rh_dot_t<16, 8>(Wt1b, x, l), 0.0f);
    if (lane < 8) {
      A2[(size_t)it * 8 + l] = c;
      x[l] = c;
    }
    if (lane == 0) {
      float z = b1c[0];
#pragma unroll
      for (int k = 0; k < 8; ++k) z = fmaf(w1c[k], x[k], z);
      reward[it] = 1.0f / (1.0f + expf(-z));
    }
  }
}

// d_reward (items) -> d_pred (items, N, 32) and per-wave partial parameter gradients part[(block * kRhWaves + wave)][kRhParams]
__global__ __launch_bounds__(64 * kRhWaves) void reward_head_bwd_k(const float* __restrict__ pred, const float* __restrict__ P,
                                                                     const float* __restrict__ reward, const float* __restrict__ H0,
                                                                     const float* __restrict__ Q, const float* __restrict__ A1,
                                                                     const float* __restrict__ A2, const float* __restrict__ d_reward,
                                                                     float* __restrict__ d_pred, float* __restrict__ part, int items, int N) {
  __shared__ __attribute__((aligned(16))) float W0a[1024], W0b[1024], W1a[512], W1b[128], w1c[8], ds[kRhWaves][32], xs[kRhWaves][32];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l = lane & 31;
  for (int i = tid; i < 1024; i += blockDim.x) { W0a[i] = P[RH_W0A + i]; W0b[i] = P[RH_W0B + i]; }
  for (int i = tid; i < 512; i += blockDim.x) W1a[i] = P[RH_W1A + i];
  for (int i = tid; i < 128; i += blockDim.x) W1b[i] = P[RH_W1B + i];
  for (int i = tid; i < 8; i += blockDim.x) w1c[i] = P[RH_W1C + i];
  __syncthreads();
  float* d = ds[wv];
  float* x = xs[wv];
  const bool act = lane < 32;
  // lane l's rows of the weight gradients + its bias gradients
  float gW0a[32], gW0b[32], gW1a[32], gW1b[16];
  float gb0a = 0.0f, gb0b = 0.0f, gb1a = 0.0f, gb1b = 0.0f, gw1c = 0.0f, gb1c = 0.0f;
#pragma unroll
  for (int k = 0; k < 32; ++k) gW0a[k] = gW0b[k] = gW1a[k] = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) gW1b[k] = 0.0f;
  for (int it = blockIdx.x * kRhWaves + wv; it < items; it += gridDim.x * kRhWaves) {
    const float r = reward[it];
    const float dz = d_reward[it] * r * (1.0f - r);
    // head1.4: z = w1c . c + b1c
    const float c = lane < 8 ? A2[(size_t)it * 8 + l] : 0.0f;
    gw1c = fmaf(dz, c, gw1c);
    gb1c += dz;
    const float dc = (lane < 8 && c > 0.0f) ? dz * w1c[l & 7] : 0.0f;
    // head1.2: c = relu(W1b a + b1b), (8, 16)
    const float a = lane < 16 ? A1[(size_t)it * 16 + l] : 0.0f;
    if (act) {
      d[l] = dc;            // lanes 8..31 write 0
      x[l] = a;             // lanes 16..31 write 0
    }
    gb1b += dc;
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      const float4 x4 = *reinterpret_cast<const float4*>(x + k);
      gW1b[k] = fmaf(dc, x4.x, gW1b[k]); gW1b[k + 1] = fmaf(dc, x4.y, gW1b[k + 1]);
      gW1b[k + 2] = fmaf(dc, x4.z, gW1b[k + 2]); gW1b[k + 3] = fmaf(dc, x4.w, gW1b[k + 3]);
    }
    float da = rh_dot_n<16, 8>(W1b, d, l);
    da = (lane < 16 && a > 0.0f) ? da : 0.0f;
    // head1.0: a = relu(W1a q + b1a), (16, 32)
    const float q = act ? Q[(size_t)it * 32 + l] : 0.0f;
    if (act) {
      d[l] = da;            // lanes 16..31 write 0
      x[l] = q;
    }
    gb1a += da;
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      const float4 x4 = *reinterpret_cast<const float4*>(x + k);
      gW1a[k] = fmaf(da, x4.x, gW1a[k]); gW1a[k + 1] = fmaf(da, x4.y, gW1a[k + 1]);
      gW1a[k + 2] = fmaf(da, x4.z, gW1a[k + 2]); gW1a[k + 3] = fmaf(da, x4.w, gW1a[k + 3]);
    }
    const float dq = rh_dot_n<32, 16>(W1a, d, l);       // = dL/d y_o for every object o (q = sum_o y_o)
    // head0.2: y_o = W0b h_o + b0b;  dL/dh (before the ReLU mask) is the same for all objects
    if (act) d[l] = dq;
    const float dh_all = rh_dot_n<32, 32>(W0b, d, l);
    for (int o = 0; o < N; ++o) {
      const size_t row = ((size_t)it * N + o) * 32;
      const float h = act ? H0[row + l] : 0.0f;
      const float xin = act ? pred[row + l] : 0.0f;
      if (act) x[l] = h;
      gb0b += dq;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        const float4 x4 = *reinterpret_cast<const float4*>(x + k);
        gW0b[k] = fmaf(dq, x4.x, gW0b[k]); gW0b[k + 1] = fmaf(dq, x4.y, gW0b[k + 1]);
        gW0b[k + 2] = fmaf(dq, x4.z, gW0b[k + 2]); gW0b[k + 3] = fmaf(dq, x4.w, gW0b[k + 3]);
      }
      const float dh = h > 0.0f ? dh_all : 0.0f;
      // head0.0: h_o = relu(W0a x_o + b0a)
      if (act) {
        x[l] = xin;
        d[l] = dh;
      }
      gb0a += dh;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        const float4 x4 = *reinterpret_cast<const float4*>(x + k);
        gW0a[k] = fmaf(dh, x4.x, gW0a[k]); gW0a[k + 1] = fmaf(dh, x4.y, gW0a[k + 1]);
        gW0a[k + 2] = fmaf(dh, x4.z, gW0a[k + 2]); gW0a[k + 3] = fmaf(dh, x4.w, gW0a[k + 3]);
      }
      const float dx = rh_dot_n<32, 32>(W0a, d, l);
      if (act) {
        d_pred[row + l] = dx;
        d[l] = dq;           // the scratch holds dq again for the next object's ... (only dh_all needed it; kept consistent)
      }
    }
  }
  // this wave's partial gradient image
  float* out = part + (size_t)(blockIdx.x * kRhWaves + wv) * kRhParams;
  if (act) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      out[RH_W0A + l * 32 + k] = gW0a[k];
      out[RH_W0B + l * 32 + k] = gW0b[k];
    }
    out[RH_B0A + l] = gb0a;
    out[RH_B0B + l] = gb0b;
    if (l < 16) {
#pragma unroll
      for (int k = 0; k < 32; ++k) out[RH_W1A + l * 32 + k] = gW1a[k];
      out[RH_B1A + l] = gb1a;
    }
    if (l < 8) {
#pragma unroll
      for (int k = 0; k < 16; ++k) out[RH_W1B + l * 16 + k] = gW1b[k];
      out[RH_B1B + l] = gb1b;
      out[RH_W1C + l] = gw1c;
    }
    if (l == 0) out[RH_B1C] = gb1c;
  }
}

// y (rows, OUT) = x (rows, IN) W^T + b for a narrow layer (the action embedding Linear(9, 4 N), dynamics.py:238-244), one thread per output;
// wt != 0: W is given as (IN, OUT) -- the same kernel then forms dx = dy W of the layer's backward.  b may be NULL.
__global__ void small_linear_k(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ y,
                               int rows, int IN, int OUT, int wt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * OUT) return;
  const size_t r = i / OUT;
  const int o = (int)(i % OUT);
  float acc = b != nullptr ? b[o] : 0.0f;
  const float* xr = x + r * IN;
  for (int k = 0; k < IN; ++k) acc = fmaf(xr[k], wt ? W[(size_t)k * OUT + o] : W[(size_t)o * IN + k], acc);
  y[i] = acc;
}

}  // namespace stove
