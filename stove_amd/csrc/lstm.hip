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

// gx (n,4H) input-side pre-activations (with both biases), gh (n,4H) recurrent pre-activations or null (h = 0),
// c_prev (n,H) or null (zero) -> c, h (n,H)
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
    cn.x = sig_(gf.x) * cp.x + sig_(gi.x) * tanhf(gg.x);
    cn.y = sig_(gf.y) * cp.y + sig_(gi.y) * tanhf(gg.y);
    cn.z = sig_(gf.z) * cp.z + sig_(gi.z) * tanhf(gg.z);
    cn.w = sig_(gf.w) * cp.w + sig_(gi.w) * tanhf(gg.w);
    hn.x = sig_(go.x) * tanhf(cn.x);
    hn.y = sig_(go.y) * tanhf(cn.y);
    hn.z = sig_(go.z) * tanhf(cn.z);
    hn.w = sig_(go.w) * tanhf(cn.w);
    st4(c + row * H + j, cn);
    st4(h + row * H + j, hn);
  }
}

// dh (n,H): gradient of this step's hidden state (output grad + recurrent grad); dc_in (n,H) or null:
// gradient flowing into this step's cell from the next step.
// -> dg (n,4H) gate pre-activation gradients, dc_out (n,H) gradient of the previous cell,
//    dgx_acc (n,4H): += dg (running sum over the steps; `first` overwrites instead).
__global__ void lstm_cell_bwd_k(const float* __restrict__ gx, const float* __restrict__ gh, const float* __restrict__ c_prev,
                                const float* __restrict__ c, const float* __restrict__ dh, const float* __restrict__ dc_in,
                                float* __restrict__ dg, float* __restrict__ dc_out, float* __restrict__ dgx_acc,
                                int first, int n, int H) {
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
      const float i = sig_(pre[0][e]), f = sig_(pre[1][e]), g = tanhf(pre[2][e]), o = sig_(pre[3][e]);
      const float tc = tanhf(cc[e]);
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
      st4(dg + go_ + k * H, v);
      float4 a = v;
      if (!first) {
        const float4 p = ld4(dgx_acc + go_ + k * H);
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
      }
      st4(dgx_acc + go_ + k * H, a);
    }
    st4(dc_out + ho, float4{dcp[0], dcp[1], dcp[2], dcp[3]});
  }
}

}  // namespace stove
