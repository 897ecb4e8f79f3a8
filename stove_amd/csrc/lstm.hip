// LSTM cell of the SuPAIR recognition network, fused gate math (gfx950).
//
// The reference runs nn.LSTM(1024 -> 256) for num_obj steps on the SAME flattened frame
// (model/video_prediction/encoder.py:43-51).  The two GEMMs per step stay on rocBLAS (plain
// library GEMMs); everything between them -- bias-free gate sum, 3 sigmoids, 2 tanh, cell and
// hidden update, and in the backward the gate gradients plus the running sum of the input-side
// gate gradient over the steps -- is one elementwise pass each way instead of ~10 / ~25 ATen
// launches over 26-105 MB tensors.  Gate order i, f, g, o as in torch.nn.LSTM.
#include "common.h"

namespace stove {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float sig_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// FAST: sigmoid / tanh on v_exp_f32 / v_rcp_f32 (absolute error ~1e-7) instead of an IEEE division and ocml's tanhf (~100
// instructions per call): the backward cell is then bound by its HBM streams alone
template <bool FAST>
__device__ __forceinline__ float sigm(float x) {
  if (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
  return sig_(x);
}
template <bool FAST>
__device__ __forceinline__ float tanh_(float x) {
  if (FAST) {
    const float t = __builtin_amdgcn_exp2f(-2.8853900817779268f * fabsf(x));      // e^(-2|x|) in (0, 1]
    return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), x);
  }
  return tanhf(x);
}

// gx (n,4H) input-side pre-activations (with both biases), gh (n,4H) recurrent pre-activations or null (h = 0),
// c_prev (n,H) or null (zero) -> c, h (n,H)
template <bool FAST>
__global__ void lstm_cell_fwd_k(const float* __restrict__ gx, const float* __restrict__ gh, const float* __restrict__ c_prev,
                                float* __restrict__ c, float* __restrict__ h, int n, int H) {
  const int q = H / 4;
  const size_t total = (size_t)n * q;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t row = t / q;
    const int j = (int)(t % q) * 4;
    const float* g0 = gx + row * 4 * H + j;
    float4 gi = ld4(g0), gf = ld4(g0 + H), gg = ld4(g0 + 2 * H), go = ld4(g0 + 3 * H);
    if (gh != nullptr) {
      const float* r0 = gh + row * 4 * H + j;
      const float4 a = ld4(r0), b = ld4(r0 + H), cc = ld4(r0 + 2 * H), d = ld4(r0 + 3 * H);
      gi.x += a.x; gi.y += a.y; gi.z += a.z; gi.w += a.w;
      gf.x += b.x; gf.y += b.y; gf.z += b.z; gf.w += b.w;
      gg.x += cc.x; gg.y += cc.y; gg.z += cc.z; gg.w += cc.w;
      go.x += d.x; go.y += d.y; go.z += d.z; go.w += d.w;
    }
    float4 cp = {0.0f, 0.0f, 0.0f, 0.0f};
    if (c_prev != nullptr) cp = ld4(c_prev + row * H + j);
    float4 cn, hn;
    cn.x = sigm<FAST>(gf.x) * cp.x + sigm<FAST>(gi.x) * tanh_<FAST>(gg.x);
    cn.y = sigm<FAST>(gf.y) * cp.y + sigm<FAST>(gi.y) * tanh_<FAST>(gg.y);
    cn.z = sigm<FAST>(gf.z) * cp.z + sigm<FAST>(gi.z) * tanh_<FAST>(gg.z);
    cn.w = sigm<FAST>(gf.w) * cp.w + sigm<FAST>(gi.w) * tanh_<FAST>(gg.w);
    hn.x = sigm<FAST>(go.x) * tanh_<FAST>(cn.x);
    hn.y = sigm<FAST>(go.y) * tanh_<FAST>(cn.y);
    hn.z = sigm<FAST>(go.z) * tanh_<FAST>(cn.z);
    hn.w = sigm<FAST>(go.w) * tanh_<FAST>(cn.w);
    st4(c + row * H + j, cn);
    st4(h + row * H + j, hn);
  }
}

// dh (n,H): gradient of this step's hidden state (output grad + recurrent grad); dc_in (n,H) or null:
// gradient flowing into this step's cell from the next step.
// -> dg (n,4H) gate pre-activation gradients (or null: not stored), dc_out (n,H) gradient of the previous cell,
//    dgx_sum (n,4H) or null: this step's dg + the dg of `n_more` other steps, dg_more[m][n][4H] -- the input projection is
//    shared by all steps, so its gradient is the sum over the steps; it is formed ONCE, by the last backward step (step 0),
//    from the stored gate gradients of the others, instead of a read-modify-write of a running sum in every step.
//    more_stride: floats between dg_more[m] and dg_more[m + 1] (n * 4H, or more when the launch covers a row range of the batch).
template <bool FAST>
__global__ void lstm_cell_bwd_k(const float* __restrict__ gx, const float* __restrict__ gh, const float* __restrict__ c_prev,
                                const float* __restrict__ c, const float* __restrict__ dh, const float* __restrict__ dc_in,
                                float* __restrict__ dg, float* __restrict__ dc_out, float* __restrict__ dgx_sum,
                                const float* __restrict__ dg_more, int n_more, size_t more_stride, int n, int H) {
  const int q = H / 4;
  const size_t total = (size_t)n * q;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t row = t / q;
    const int j = (int)(t % q) * 4;
    const size_t go_ = row * 4 * H + j, ho = row * H + j;
    float pre[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 v = ld4(gx + go_ + k * H);
      if (gh != nullptr) {
        const float4 r = ld4(gh + go_ + k * H);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      pre[k][0] = v.x; pre[k][1] = v.y; pre[k][2] = v.z; pre[k][3] = v.w;
    }
    float cp[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dci[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (c_prev != nullptr) { const float4 v = ld4(c_prev + ho); cp[0] = v.x; cp[1] = v.y; cp[2] = v.z; cp[3] = v.w; }
    if (dc_in != nullptr) { const float4 v = ld4(dc_in + ho); dci[0] = v.x; dci[1] = v.y; dci[2] = v.z; dci[3] = v.w; }
    const float4 cv = ld4(c + ho), dhv = ld4(dh + ho);
    const float cc[4] = {cv.x, cv.y, cv.z, cv.w}, dhh[4] = {dhv.x, dhv.y, dhv.z, dhv.w};
    float out[4][4], dcp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float i = sigm<FAST>(pre[0][e]), f = sigm<FAST>(pre[1][e]), g = tanh_<FAST>(pre[2][e]), o = sigm<FAST>(pre[3][e]);
      const float tc = tanh_<FAST>(cc[e]);
      const float dc = dci[e] + dhh[e] * o * (1.0f - tc * tc);
      out[0][e] = dc * g * i * (1.0f - i);
      out[1][e] = dc * cp[e] * f * (1.0f - f);
      out[2][e] = dc * i * (1.0f - g * g);
      out[3][e] = dhh[e] * tc * o * (1.0f - o);
      dcp[e] = dc * f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = {out[k][0], out[k][1], out[k][2], out[k][3]};
      if (dg != nullptr) st4(dg + go_ + k * H, v);
      if (dgx_sum != nullptr) {
        float4 a = v;
        for (int m = 0; m < n_more; ++m) {
          const float4 p = ld4(dg_more + (size_t)m * more_stride + go_ + k * H);
          a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        }
        st4(dgx_sum + go_ + k * H, a);
      }
    }
    st4(dc_out + ho, float4{dcp[0], dcp[1], dcp[2], dcp[3]});
  }
}

// ---- output head of the recognition network: zps = fc2(sigmoid(fc1(h)))  (encoder.py:53-56) ---------------------------------
// fc1 (256 -> 50) stays a library GEMM; everything behind it is narrow (50 -> 8) and HBM-bound: as library calls the
// backward is a (rows x 8)(8 x 50) GEMM on 16x16 tiles, an elementwise sigmoid', a split-K GEMM for the 8 x 50 weight
// gradient and two column sums (~130 us for 76 800 rows); here it is one pass over h1 (15 MB) each way.
// One wave owns a tile of 64 rows: the tile is staged in the wave's LDS slice (row stride odd -> conflict-free both by
// row and by column), lane = row for the products along the features, lane = feature for the sums over rows.
constexpr int kHeadOut = 8;      // max outputs (2 * z_size)
constexpr int kHeadHid = 64;     // max hidden width

__device__ __forceinline__ int head_stride(int H1) { return H1 | 1; }
constexpr int kHeadV4 = 64 * kHeadHid / 4 / 64;      // float4 loads per lane that cover a 64 x kHeadHid tile

// A tile = 64 consecutive rows = 64 * H1 consecutive floats (16-byte aligned since 64 * H1 * 4 is): float4 loads, scattered
// into the LDS tile [row][stride].  The loads go out in batches of kHeadBatch before their first use -- with one tile per
// wave a "load, use, load, use" loop was pure memory latency -- and the batches bound the registers, so the kernels fit next
// to the register-heavy MFMA kernels they overlap with.  SIG: sigmoid on the way, result also stored to `dst`.
constexpr int kHeadBatch = 4;
template <bool SIG>
__device__ __forceinline__ void head_stream(const float* __restrict__ src, float* __restrict__ dst, float* tile, int n_el, int H1,
                                            int stride, int lane) {
  int row = (lane * 4) / H1, col = (lane * 4) % H1;
  const int dq = 256 / H1, dr = 256 % H1;
  for (int i0 = 0; i0 < kHeadV4 && i0 * 256 < n_el; i0 += kHeadBatch) {
    float4 q[kHeadBatch];
#pragma unroll
    for (int u = 0; u < kHeadBatch; ++u) {
      const int e = ((i0 + u) * 64 + lane) * 4;
      q[u] = float4{0.0f, 0.0f, 0.0f, 0.0f};
      if (e + 3 < n_el) {
        q[u] = ld4(src + e);
      } else if (e < n_el) {
        q[u].x = src[e];
        if (e + 1 < n_el) q[u].y = src[e + 1];
        if (e + 2 < n_el) q[u].z = src[e + 2];
      }
    }
#pragma unroll
    for (int u = 0; u < kHeadBatch; ++u) {
      const int e = ((i0 + u) * 64 + lane) * 4;
      float vv[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
      if (SIG) {
#pragma unroll
        for (int k = 0; k < 4; ++k) vv[k] = sig_(vv[k]);
        if (e + 3 < n_el) {
          st4(dst + e, float4{vv[0], vv[1], vv[2], vv[3]});
        } else if (e < n_el) {
          dst[e] = vv[0];
          if (e + 1 < n_el) dst[e + 1] = vv[1];
          if (e + 2 < n_el) dst[e + 2] = vv[2];
        }
      }
      int r = row, c = col;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (e + k < n_el) tile[r * stride + c] = vv[k];
        if (++c == H1) { c = 0; ++r; }
      }
      row += dq;
      col += dr;
      if (col >= H1) { col -= H1; ++row; }
    }
  }
}
// W2 (OUT, H1) -> LDS w2t[j][kHeadOut] (zero padded): two broadcast float4 reads per feature
__device__ __forceinline__ void head_stage_w2(const float* __restrict__ W2, float* w2t, int H1, int OUT) {
  for (int i = threadIdx.x; i < H1 * kHeadOut; i += blockDim.x) {
    const int j = i / kHeadOut, k = i % kHeadOut;
    w2t[i] = k < OUT ? W2[k * H1 + j] : 0.0f;
  }
  __syncthreads();
}

// a1 (rows, H1) = fc1 pre-activations -> h1 = sigmoid(a1) (rows, H1), codes (rows, OUT) = h1 W2^T + b2;  W2 (OUT, H1)
__global__ __launch_bounds__(256) void head_fwd_k(const float* __restrict__ a1, const float* __restrict__ W2, const float* __restrict__ b2,
                           float* __restrict__ h1, float* __restrict__ codes, int rows, int H1, int OUT) {
  extern __shared__ __attribute__((aligned(16))) float head_lds[];
  const int lane = lane_id(), wv = wave_id(), nw = blockDim.x >> 6;
  const int stride = head_stride(H1);
  float* w2t = head_lds;
  float* tile = head_lds + kHeadHid * kHeadOut + (size_t)wv * 64 * stride;
  head_stage_w2(W2, w2t, H1, OUT);
  const int n_tiles = (rows + 63) / 64;
  for (int t = blockIdx.x * nw + wv; t < n_tiles; t += gridDim.x * nw) {
    const size_t r0 = (size_t)t * 64;
    const int live = rows - (int)r0 < 64 ? rows - (int)r0 : 64;
    const int n_el = live * H1;
    head_stream<true>(a1 + r0 * H1, h1 + r0 * H1, tile, n_el, H1, stride, lane);
    if (lane < live) {
      float acc[kHeadOut];
#pragma unroll
      for (int k = 0; k < kHeadOut; ++k) acc[k] = k < OUT ? b2[k] : 0.0f;
      const float* hr = tile + lane * stride;
      for (int j = 0; j < H1; ++j) {
        const float h = hr[j];
        const float4 wa = ld4(w2t + j * kHeadOut), wb = ld4(w2t + j * kHeadOut + 4);
        acc[0] = fmaf(h, wa.x, acc[0]); acc[1] = fmaf(h, wa.y, acc[1]); acc[2] = fmaf(h, wa.z, acc[2]); acc[3] = fmaf(h, wa.w, acc[3]);
        acc[4] = fmaf(h, wb.x, acc[4]); acc[5] = fmaf(h, wb.y, acc[5]); acc[6] = fmaf(h, wb.z, acc[6]); acc[7] = fmaf(h, wb.w, acc[7]);
      }
      float* o = codes + (r0 + lane) * OUT;
      if (OUT == kHeadOut) {
        st4(o, float4{acc[0], acc[1], acc[2], acc[3]});
        st4(o + 4, float4{acc[4], acc[5], acc[6], acc[7]});
      } else {
#pragma unroll
        for (int k = 0; k < kHeadOut; ++k)
          if (k < OUT) o[k] = acc[k];
      }
    }
  }
}

// dcodes (rows, OUT), h1 (rows, H1) -> d_a1 (rows, H1) = (dcodes W2) * h1 (1 - h1); per-block partial sums
// part[block][OUT*H1 | H1 | OUT] = (dW2 = dcodes^T h1, db1 = colsum d_a1, db2 = colsum dcodes), waves added in order.
__global__ __launch_bounds__(256) void head_bwd_k(const float* __restrict__ dcodes, const float* __restrict__ h1, const float* __restrict__ W2,
                           float* __restrict__ d_a1, float* __restrict__ part, int rows, int H1, int OUT) {
  extern __shared__ __attribute__((aligned(16))) float head_lds[];
  const int lane = lane_id(), wv = wave_id(), nw = blockDim.x >> 6;
  const int stride = head_stride(H1);
  const int slice = 64 * stride + 64 * kHeadOut;
  float* w2t = head_lds;
  float* tile = head_lds + kHeadHid * kHeadOut + (size_t)wv * slice;
  float* gt = tile + 64 * stride;                 // [64][kHeadOut]
  head_stage_w2(W2, w2t, H1, OUT);
  float aw[kHeadOut], ab1 = 0.0f, ab2 = 0.0f;
#pragma unroll
  for (int k = 0; k < kHeadOut; ++k) aw[k] = 0.0f;
  const int n_tiles = (rows + 63) / 64;
  for (int t = blockIdx.x * nw + wv; t < n_tiles; t += gridDim.x * nw) {
    const size_t r0 = (size_t)t * 64;
    const int live = rows - (int)r0 < 64 ? rows - (int)r0 : 64;
    const int n_el = live * H1;
    head_stream<false>(h1 + r0 * H1, nullptr, tile, n_el, H1, stride, lane);
    float g[kHeadOut];
    if (OUT == kHeadOut && lane < live) {
      const float4 ga = ld4(dcodes + (r0 + lane) * OUT), gb = ld4(dcodes + (r0 + lane) * OUT + 4);
      g[0] = ga.x, g[1] = ga.y, g[2] = ga.z, g[3] = ga.w, g[4] = gb.x, g[5] = gb.y, g[6] = gb.z, g[7] = gb.w;
    } else {
#pragma unroll
      for (int k = 0; k < kHeadOut; ++k) g[k] = (lane < live && k < OUT) ? dcodes[(r0 + lane) * OUT + k] : 0.0f;
    }
    st4(gt + lane * kHeadOut, float4{g[0], g[1], g[2], g[3]});
    st4(gt + lane * kHeadOut + 4, float4{g[4], g[5], g[6], g[7]});
    // sums over the tile's rows, lane = feature (h tile still intact)
    if (lane < H1) {
      for (int r = 0; r < live; ++r) {
        const float hv = tile[r * stride + lane];
        const float4 ga = ld4(gt + r * kHeadOut), gb = ld4(gt + r * kHeadOut + 4);
        aw[0] = fmaf(ga.x, hv, aw[0]); aw[1] = fmaf(ga.y, hv, aw[1]); aw[2] = fmaf(ga.z, hv, aw[2]); aw[3] = fmaf(ga.w, hv, aw[3]);
        aw[4] = fmaf(gb.x, hv, aw[4]); aw[5] = fmaf(gb.y, hv, aw[5]); aw[6] = fmaf(gb.z, hv, aw[6]); aw[7] = fmaf(gb.w, hv, aw[7]);
      }
    }
    if (lane < OUT)
      for (int r = 0; r < live; ++r) ab2 += gt[r * kHeadOut + lane];
    // lane = row: d_a1 over the h tile
    if (lane < live) {
      float* hr = tile + lane * stride;
      for (int j = 0; j < H1; ++j) {
        const float4 wa = ld4(w2t + j * kHeadOut), wb = ld4(w2t + j * kHeadOut + 4);
        float tsum = g[0] * wa.x;
        tsum = fmaf(g[1], wa.y, tsum); tsum = fmaf(g[2], wa.z, tsum); tsum = fmaf(g[3], wa.w, tsum);
        tsum = fmaf(g[4], wb.x, tsum); tsum = fmaf(g[5], wb.y, tsum); tsum = fmaf(g[6], wb.z, tsum); tsum = fmaf(g[7], wb.w, tsum);
        const float h = hr[j];
        hr[j] = tsum * h * (1.0f - h);
      }
    }
    if (lane < H1)
      for (int r = 0; r < live; ++r) ab1 += tile[r * stride + lane];
    // back out in the global (row-major) order
    {
      int row = (lane * 4) / H1, col = (lane * 4) % H1;
      const int dq = 256 / H1, dr = 256 % H1;
      float* dst = d_a1 + r0 * H1;
#pragma unroll 4
      for (int i = 0; i < kHeadV4; ++i) {
        const int e = (i * 64 + lane) * 4;
        if (e >= n_el) break;
        float vv[4];
        int r = row, c = col;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          vv[k] = (e + k < n_el) ? tile[r * stride + c] : 0.0f;
          if (++c == H1) { c = 0; ++r; }
        }
        if (e + 3 < n_el) {
          st4(dst + e, float4{vv[0], vv[1], vv[2], vv[3]});
        } else {
          dst[e] = vv[0];
          if (e + 1 < n_el) dst[e + 1] = vv[1];
          if (e + 2 < n_el) dst[e + 2] = vv[2];
        }
        row += dq;
        col += dr;
        if (col >= H1) { col -= H1; ++row; }
      }
    }
  }
  // block partials: waves in fixed order through their own LDS slices
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kHeadOut; ++k) tile[k * 64 + lane] = aw[k];
  tile[kHeadOut * 64 + lane] = ab1;
  tile[(kHeadOut + 1) * 64 + lane] = ab2;
  __syncthreads();
  if (wv == 0) {
    const int n_out = OUT * H1 + H1 + OUT;
    float* o = part + (size_t)blockIdx.x * n_out;
    for (int q = 0; q < kHeadOut + 2; ++q) {
      float s = 0.0f;
      for (int w = 0; w < nw; ++w) s += head_lds[kHeadHid * kHeadOut + (size_t)w * slice + q * 64 + lane];
      if (q < kHeadOut) {
        if (q < OUT && lane < H1) o[q * H1 + lane] = s;
      } else if (q == kHeadOut) {
        if (lane < H1) o[OUT * H1 + lane] = s;
      } else if (lane < OUT) {
        o[OUT * H1 + H1 + lane] = s;
      }
    }
  }
}

}  // namespace stove
