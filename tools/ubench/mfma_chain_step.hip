// How fast could the T-serial inference recursion be as ONE wave per sequence with every dense layer a chain of split-bf16 MFMAs?
//
// Today (gnn_small.hip) a step is ~9 500 cycles forward: four waves per sequence, node layers as LDS-broadcast + v_pk_fma chains (300-480
// cycles each), edge layers as fp32 v_mfma_f32_16x16x4 chains (32 cycles per MFMA, 16 per 32 x 32 layer), two workgroup barriers.
// Alternative measured here (dataflow and instruction mix of the real step, synthetic weights -- NOT a product kernel):
//   * one wave owns a sequence; node quantities are the columns 0..N-1 of a 16-column MFMA tile, edge quantities the columns 0..N(N-1)-1;
//   * a 32-wide activation is 8 registers per lane (feature 16 t + 4 g + e in register [t][e] of lane (column, g)): the accumulators of
//     one layer ARE the B operand's eight k-slots of the next (after the hi / lo split), so layers chain with no data movement;
//   * a 32 -> 32 layer = 2 output tiles x 3 v_mfma_f32_16x16x32_bf16 (hi hi, hi lo, lo hi) = 6 MFMAs of 16 cycles;
//   * node -> edge gather and edge -> node aggregation are ds_bpermute's; no LDS staging of activations, no barrier.
// Precision of the split-bf16 products in the recursion: tools/bf16x3_recursion_probe.py (ELBO 6e-7 relative against fp64).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain_step mfma_chain_step.hip ; run: ./mfma_chain_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
struct Opnd { bf16x8 hi, lo; };            // one k-step (32 k) of a B operand: 8 k-slots per lane
// the two output tiles of a 32-wide activation -> the next layer's k-step operand (hi / lo split)
__device__ __forceinline__ Opnd to_operand(const f32x4 a, const f32x4 b) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pack2(v[2 * i], v[2 * i + 1]);
    const float r0 = v[2 * i] - __uint_as_float(h[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u);
    l[i] = pack2(r0, r1);
  }
  Opnd o;
  o.hi = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
  o.lo = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
  return o;
}
// weights of one layer in LDS: [out tile][k-step][hi | lo][64 lanes] x 16 B
template <int KS, int OT>
__device__ __forceinline__ void layer(const char* W, const Opnd (&x)[KS], f32x4 (&acc)[OT], int lane) {
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(W + (((t * KS + s) * 2 + 0) * 64 + lane) * 16);
      const bf16x8 wl = *reinterpret_cast<const bf16x8*>(W + (((t * KS + s) * 2 + 1) * 64 + lane) * 16);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, x[s].hi, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, x[s].lo, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, x[s].hi, acc[t], 0, 0, 0);
    }
}
constexpr int layer_bytes(int ks, int ot) { return ot * ks * 2 * 64 * 16; }
__device__ __forceinline__ f32x4 lrelu(f32x4 v) { f32x4 r; for (int e = 0; e < 4; ++e) r[e] = fmaxf(v[e], 0.01f * v[e]); return r; }
__device__ __forceinline__ f32x4 tanh4(f32x4 v) { f32x4 r; for (int e = 0; e < 4; ++e) r[e] = tanhf(v[e]); return r; }
__device__ __forceinline__ f32x4 perm4(f32x4 v, int src_lane) {
  f32x4 r;
  for (int e = 0; e < 4; ++e) r[e] = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(v[e])));
  return r;
}

// layer offsets in the LDS weight image
constexpr int O_ENC = 0, O_S0 = O_ENC + layer_bytes(1, 2), O_S1 = O_S0 + layer_bytes(1, 2), O_R0 = O_S1 + layer_bytes(1, 2),
              O_A0 = O_R0 + layer_bytes(2, 4), O_R1 = O_A0 + layer_bytes(2, 4), O_A1 = O_R1 + layer_bytes(2, 2), O_R2 = O_A1 + layer_bytes(2, 2),
              O_F0 = O_R2 + layer_bytes(1, 2), O_F1 = O_F0 + layer_bytes(1, 2), O_F2 = O_F1 + layer_bytes(1, 2), O_O0 = O_F2 + layer_bytes(1, 2),
              O_O1 = O_O0 + layer_bytes(2, 2), O_END = O_O1 + layer_bytes(1, 2);

template <int N, bool SAVE>
__global__ __launch_bounds__(64) void chain_step_k(const float* __restrict__ wsrc, const float* __restrict__ eps, const float* __restrict__ zsup,
                                                   float* __restrict__ zout, float* __restrict__ act, int Ts) {
  extern __shared__ __attribute__((aligned(16))) char W0[];
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4, b = blockIdx.x;
  for (int i = lane; i < O_END / 16; i += 64) reinterpret_cast<float4*>(W0)[i] = reinterpret_cast<const float4*>(wsrc)[i % 4096];
  __syncthreads();
  constexpr int E = N * (N - 1);
  // edge column q -> nodes (i, j)
  const int q = c < E ? c : 0;
  const int ei = q / (N - 1), ejj = q % (N - 1), ej = ejj + (ejj >= ei ? 1 : 0);
  const int lane_i = ei + 16 * g, lane_j = ej + 16 * g;
  f32x4 s_in[2] = {f32x4{0.1f * c, 0.2f, 0.3f, 0.1f * g}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
  const f32x4 bias = {0.01f, -0.02f, 0.03f, 0.0f};
  for (int ts = 0; ts < Ts; ++ts) {
    // the weights are re-read from LDS every step (laundered base: hoisted out of the loop they would need 1 400 registers per lane)
    int woff = 0;
    asm volatile("" : "+s"(woff));
    const char* W = W0 + woff;
    const size_t o = ((size_t)b * Ts + ts) * 16 + c;
    // this step's inputs (issued now, used in the epilogue)
    const float4 ep = *reinterpret_cast<const float4*>(eps + o * 16 + 4 * g);
    const float4 zs = *reinterpret_cast<const float4*>(zsup + o * 16 + 4 * g);
    // encoder (positions pass through raw)
    Opnd x1[1] = {to_operand(s_in[0], s_in[1])};
    f32x4 S[2] = {bias, bias};
    layer<1, 2>(W + O_ENC, x1, S, lane);
    if (g == 0) { S[0][0] = s_in[0][0]; S[0][1] = s_in[0][1]; }
    const Opnd xS[1] = {to_operand(S[0], S[1])};
    // self-dynamics
    f32x4 h[2] = {bias, bias};
    layer<1, 2>(W + O_S0, xS, h, lane);
    h[0] = lrelu(h[0]); h[1] = lrelu(h[1]);
    const Opnd xh[1] = {to_operand(h[0], h[1])};
    f32x4 sd[2] = {bias, bias};
    layer<1, 2>(W + O_S1, xh, sd, lane);
    sd[0] += h[0]; sd[1] += h[1];
    // node -> edge gather of S_i, S_j (and the squared distance of the two positions)
    const f32x4 si0 = perm4(S[0], lane_i), si1 = perm4(S[1], lane_i), sj0 = perm4(S[0], lane_j), sj1 = perm4(S[1], lane_j);
    float dx = si0[0] - sj0[0], dy = si0[1] - sj0[1];
    float d = dx * dx + dy * dy;
    d = __int_as_float(__builtin_amdgcn_ds_bpermute(c * 4, __float_as_int(d)));      // from the column's g = 0 lane
    const Opnd xe[2] = {to_operand(si0, si1), to_operand(sj0, sj1)};
    // relation chain
    f32x4 r1[4], a1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { r1[t] = bias * d; a1[t] = bias * d; }
    layer<2, 4>(W + O_R0, xe, r1, lane);
    layer<2, 4>(W + O_A0, xe, a1, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) { r1[t] = lrelu(r1[t]); a1[t] = lrelu(a1[t]); }
    const Opnd xr1[2] = {to_operand(r1[0], r1[1]), to_operand(r1[2], r1[3])}, xa1[2] = {to_operand(a1[0], a1[1]), to_operand(a1[2], a1[3])};
    f32x4 r2[2] = {bias, bias}, a2[2] = {bias, bias};
    layer<2, 2>(W + O_R1, xr1, r2, lane);
    layer<2, 2>(W + O_A1, xa1, a2, lane);
    r2[0] = lrelu(r2[0]); r2[1] = lrelu(r2[1]); a2[0] = lrelu(a2[0]); a2[1] = lrelu(a2[1]);
    const Opnd xr2[1] = {to_operand(r2[0], r2[1])};
    f32x4 r3[2] = {bias, bias};
    layer<1, 2>(W + O_R2, xr2, r3, lane);
    r3[0] += r2[0]; r3[1] += r2[1];
    // attention: exp(w . A2 + b): per-lane partial dot + reduction over the four lane groups
    float att = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) att += a2[0][e] * 0.01f * (e + 1) + a2[1][e] * 0.02f;
    att += __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) * 4, __float_as_int(att)));
    att += __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 32) & 63) * 4, __float_as_int(att)));
    att = __expf(att * 0.01f);
    r3[0] *= att; r3[1] *= att;
    // edge -> node aggregation: node i sums its N - 1 outgoing edge columns i (N-1) .. i (N-1) + N-2
    f32x4 ag[2] = {sd[0], sd[1]};
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
      const int src = ((c < N ? c : 0) * (N - 1) + k) + 16 * g;
      ag[0] += perm4(r3[0], src); ag[1] += perm4(r3[1], src);
    }
    // affector
    const Opnd xp[1] = {to_operand(ag[0], ag[1])};
    f32x4 f1[2] = {bias, bias};
    layer<1, 2>(W + O_F0, xp, f1, lane);
    f1[0] = tanh4(f1[0]); f1[1] = tanh4(f1[1]);
    const Opnd xf1[1] = {to_operand(f1[0], f1[1])};
    f32x4 f2[2] = {bias, bias};
    layer<1, 2>(W + O_F1, xf1, f2, lane);
    f2[0] = tanh4(f2[0]) + f1[0]; f2[1] = tanh4(f2[1]) + f1[1];
    const Opnd xf2[1] = {to_operand(f2[0], f2[1])};
    f32x4 f3[2] = {bias, bias};
    layer<1, 2>(W + O_F2, xf2, f3, lane);
    // output
    const Opnd xo[2] = {to_operand(f3[0], f3[1]), xS[0]};
    f32x4 o1[2] = {bias, bias};
    layer<2, 2>(W + O_O0, xo, o1, lane);
    o1[0] = tanh4(o1[0]); o1[1] = tanh4(o1[1]);
    const Opnd xo1[1] = {to_operand(o1[0], o1[1])};
    f32x4 res[2] = {bias, bias};
    layer<1, 2>(W + O_O1, xo1, res, lane);
    res[0] += o1[0]; res[1] += o1[1];
    // epilogue: means / stds, product of Gaussians with the SuPAIR state, sample (lane holds dims 4 g .. 4 g + 3 and their std halves)
    f32x4 zn;
    const float epv[4] = {ep.x, ep.y, ep.z, ep.w}, zsv[4] = {zs.x, zs.y, zs.z, zs.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m = 2.0f / (1.0f + __expf(-res[0][e])) - 1.0f;
      const float sdv = 0.04f / (1.0f + __expf(-res[1][e]));
      const float zd = m + (g == 0 && e < 2 ? s_in[0][e] : 0.0f);
      const float ss = 0.05f + 0.01f * zsv[e], D = sdv * sdv + ss * ss;
      const float mu = g == 0 ? (ss * ss * zd + sdv * sdv * zsv[e]) / D : zd;
      const float sg = g == 0 ? sdv * ss * rsqrtf(D) : sdv;
      zn[e] = fmaf(sg, epv[e], mu);
    }
    if (c < N) {
      *reinterpret_cast<float4*>(zout + o * 16 + 4 * g) = float4{zn[0], zn[1], zn[2], zn[3]};
      if (SAVE) {          // the saved activations of the backward: ~9 KB per step
        float* a = act + o * 9 * 32 + 4 * g;
        const f32x4* sv[9] = {&S[0], &h[0], &ag[0], &f1[0], &f2[0], &f3[0], &o1[0], &res[0], &sd[0]};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          *reinterpret_cast<f32x4*>(a + k * 32) = sv[k][0];
          *reinterpret_cast<f32x4*>(a + k * 32 + 16) = sv[k][1];
        }
      }
    }
    if (SAVE && c < E) {
      float* a = act + ((size_t)gridDim.x * Ts * 16) * 9 * 32 + o * 6 * 32 + 4 * g;
      const f32x4* sv[6] = {&r1[0], &r1[2], &a1[0], &a1[2], &r2[0], &a2[0]};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        *reinterpret_cast<f32x4*>(a + k * 32) = sv[k][0];
        *reinterpret_cast<f32x4*>(a + k * 32 + 16) = sv[k][1];
      }
    }
    s_in[0] = zn;
    s_in[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
}

int main() {
  const int B = 256, Ts = 98;
  float *w, *eps, *zsup, *zout, *act;
  CK(hipMalloc(&w, 4096 * 16));
  CK(hipMalloc(&eps, (size_t)B * Ts * 16 * 16 * 4));
  CK(hipMalloc(&zsup, (size_t)B * Ts * 16 * 16 * 4));
  CK(hipMalloc(&zout, (size_t)B * Ts * 16 * 16 * 4));
  CK(hipMalloc(&act, (size_t)B * Ts * 16 * 15 * 32 * 4));
  std::vector<float> hw(4096 * 4);
  for (size_t i = 0; i < hw.size(); ++i) {        // small bf16 weights (two per float)
    const unsigned short v = 0x3c00 + (i * 37 % 251);
    unsigned u = ((unsigned)v << 16) | v;
    hw[i] = *reinterpret_cast<float*>(&u);
  }
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(eps, 0, (size_t)B * Ts * 16 * 16 * 4));
  CK(hipMemset(zsup, 0, (size_t)B * Ts * 16 * 16 * 4));
  printf("LDS weight image %d bytes\n", O_END);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int save = 0; save < 2; ++save) {
    auto kern = save ? chain_step_k<3, true> : chain_step_k<3, false>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, O_END));
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(kern, dim3(B), dim3(64), O_END, 0, w, eps, zsup, zout, act, Ts);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("one wave per sequence, split-bf16 MFMA chains, N = 3, save = %d: %.1f us per launch of %d steps = %.2f us per step\n", save, best * 1e3, Ts, best * 1e3 / Ts);
  }
  return 0;
}
