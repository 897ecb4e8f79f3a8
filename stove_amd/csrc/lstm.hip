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

}  // namespace stove
