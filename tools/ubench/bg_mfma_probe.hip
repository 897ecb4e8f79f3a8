// Probe: where does bgspn_mfma_fwd_k spend its time?  Variants of the kernel with one ingredient removed.
#include "../../stove_amd/csrc/common.h"
namespace stove {
constexpr int kBgPix = 1024, kBgSide = 32;
constexpr int kBgDenseF = 3 * 3 * 64 * 4 * 16 * 4;
typedef float bgf4 __attribute__((ext_vector_type(4)));
struct BoxGeom { float inv_sx, inv_sy, off_x, off_y; };
__device__ __forceinline__ BoxGeom box_geom(const float* z) { BoxGeom g; g.inv_sx = 1.0f / z[0]; g.inv_sy = 1.0f / z[1]; g.off_x = -z[2] / z[0]; g.off_y = -z[3] / z[1]; return g; }
__device__ __forceinline__ float inv_coord(float inv_s, float off, int idx) {
  const float u = (2.0f * idx + 1.0f) * (1.0f / kBgSide) - 1.0f;
  const float gq = fmaf(inv_s, u, off);
  return ((gq + 1.0f) * kBgSide - 1.0f) * 0.5f;
}
// NOBJ > 0: compile-time object count (the per-object loop unrolls and the whole pixel block becomes one basic block,
// so the scheduler overlaps one tile's mask / feature VALU work with the other tile's MFMAs); NOBJ = 0: runtime n_obj.
template <int TPW, int NOBJ, int MODE>
__global__ __launch_bounds__(256) void probe_k(const float* __restrict__ frames, const float* __restrict__ z, int n_obj_rt,
                                                       const float* __restrict__ Cf, float* __restrict__ ell, int F) {
  const int n_obj = NOBJ > 0 ? NOBJ : n_obj_rt;
  extern __shared__ __attribute__((aligned(16))) float bg_lds[];
  const int wv = wave_id(), lane = lane_id(), i = lane & 15, kq = lane >> 4;
  const int waves = blockDim.x >> 6;
  const int f0 = (blockIdx.x * waves + wv) * (16 * TPW);
  // [frame][object][cx 32 | cy 32]; the frame stride is padded by 4 floats so that the 16 frames of a lane group hit
  // 16 different 4-bank groups (stride n_obj * 64 alone is a multiple of the 64 banks: a 16-way conflict)
  const int fstride = n_obj * 64 + 4;
  float* tab = bg_lds + (size_t)wv * ((TPW * 16) * fstride + TPW * 16 * n_obj * 4);
  float* geo = tab + (TPW * 16) * fstride;                           // [frame][object][inv_sx, inv_sy, off_x, off_y]
  // stage the box geometry of the wave's frames with one parallel pass of loads (walking z inside the table loop
  // below made every iteration wait on a dependent global load: ~100 us per launch, more than the GEMM itself)
  for (int q = lane; q < TPW * 16 * n_obj; q += 64) {
    const int fr = q / n_obj;
    float4 g4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (f0 + fr < F) {
      const BoxGeom bg = box_geom(z + ((size_t)f0 * n_obj + q) * 4);
      g4 = float4{bg.inv_sx, bg.inv_sy, bg.off_x, bg.off_y};
    }
    *reinterpret_cast<float4*>(geo + q * 4) = g4;
  }
  for (int idx = lane; idx < TPW * 16 * n_obj * 64; idx += 64) {
    const int c = idx & 63, q = idx >> 6, k = q % n_obj, fr = q / n_obj;
    float v = 0.0f;
    if (f0 + fr < F) {
      const float4 g4 = *reinterpret_cast<const float4*>(geo + q * 4);
      float dq;
      v = (c < 32) ? cover(inv_coord(g4.x, g4.z, c), kBgSide, &dq) : cover(inv_coord(g4.y, g4.w, c - 32), kBgSide, &dq);
    }
    tab[fr * fstride + k * 64 + c] = v;
  }
  bgf4 acc[TPW][3];
#pragma unroll
  for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[tl][t] = bgf4{0.0f, 0.0f, 0.0f, 0.0f};
  const float4* Cq = reinterpret_cast<const float4*>(Cf) + kq * 16 + i;      // + ((feat*3 + t)*64 + kb) * 64
  float4 bq[9], bn[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) bq[q] = Cq[(size_t)(q * 64 + ((blockIdx.x * 4) & 63)) * 64];
  // frame pixels are fetched a whole GROUP of 4 pixel blocks ahead (~4 us of MFMA work: HBM latency under these 64-byte
  // strided reads is several us), the weight fragments one block ahead (L2)
  // Every workgroup walks the SAME 590 KB weight image; started in lockstep they would all hit the same L2 channel
  // at the same time.  Each workgroup therefore starts at its own pixel block (kofs) and wraps around.
  const int kofs = (blockIdx.x * 4) & 63;
  constexpr int GRP = 4;
  float4 xc[GRP][TPW], xn[GRP][TPW];
  const float* fptr[TPW];
#pragma unroll
  for (int tl = 0; tl < TPW; ++tl) {
    const int f = f0 + tl * 16 + i;
    fptr[tl] = frames + (size_t)(f < F ? f : 0) * kBgPix + 4 * kq;
#pragma unroll
    for (int u = 0; u < GRP; ++u) xc[u][tl] = *reinterpret_cast<const float4*>(fptr[tl] + 16 * ((kofs + u) & 63));
  }
  for (int kg = 0; kg < (MODE == 4 ? 0 : 64); kg += GRP) {
    if (kg + GRP < 64) {
#pragma unroll
      for (int u = 0; u < GRP; ++u)
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) xn[u][tl] = *reinterpret_cast<const float4*>(fptr[tl] + 16 * ((kofs + kg + GRP + u) & 63));
    }
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      const int kb = (kofs + kg + u) & 63;
      if (kg + u + 1 < 64) {
#pragma unroll
        for (int q = 0; q < 9; ++q) bn[q] = (MODE == 3) ? bq[q] : Cq[(size_t)(q * 64 + ((kb + 1) & 63)) * 64];
      }
      const int row = kb >> 1, c0 = 16 * (kb & 1) + 4 * kq;
#pragma unroll
      for (int tl = 0; tl < TPW; ++tl) {
        const int fr = tl * 16 + i, f = f0 + fr;
        const float4 x = (MODE == 2) ? float4{0.5f, 0.25f, 0.125f, 0.75f} : xc[u][tl];
        float4 run = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < (MODE == 2 ? 0 : n_obj); ++k) {
          const float* tk = tab + fr * fstride + k * 64;
          const float cy = tk[32 + row];
          const float4 cx = *reinterpret_cast<const float4*>(tk + c0);
          run.x = fmaf(cx.x, cy, run.x);
          run.y = fmaf(cx.y, cy, run.y);
          run.z = fmaf(cx.z, cy, run.z);
          run.w = fmaf(cx.w, cy, run.w);
        }
        const bool live = f < F;
        float w[4] = {1.0f - fminf(run.x, 1.0f), 1.0f - fminf(run.y, 1.0f), 1.0f - fminf(run.z, 1.0f), 1.0f - fminf(run.w, 1.0f)};
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float wm = live ? w[m] : 0.0f;
          const float wx = wm * xs[m], wxx = wx * xs[m];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            if (MODE == 1) {
              acc[tl][t][0] += wxx * bq[0 * 3 + t][m] + wx * bq[1 * 3 + t][m] + wm * bq[2 * 3 + t][m];
            } else {
              acc[tl][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxx, bq[0 * 3 + t][m], acc[tl][t], 0, 0, 0);
              acc[tl][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wx, bq[1 * 3 + t][m], acc[tl][t], 0, 0, 0);
              acc[tl][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm, bq[2 * 3 + t][m], acc[tl][t], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) bq[q] = bn[q];
    }
#pragma unroll
    for (int u = 0; u < GRP; ++u)
#pragma unroll
      for (int tl = 0; tl < TPW; ++tl) xc[u][tl] = xn[u][tl];
  }
#pragma unroll
  for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int f = f0 + tl * 16 + 4 * kq + reg, col = 16 * t + i;
        if (f < F && col < 36) ell[(size_t)f * 36 + col] = acc[tl][t][reg];
      }
}

}  // namespace stove
#include <stdio.h>
using namespace stove;
template <int MODE> float run(const float* frames, const float* z, const float* Cf, float* ell, int F) {
  const int waves = 4, TPW = 2, n_obj = 3;
  const size_t lds = (size_t)waves * TPW * 16 * (n_obj * 64 + 4 + n_obj * 4) * sizeof(float);
  hipFuncSetAttribute((const void*)probe_k<2, 3, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int per_block = waves * TPW * 16;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((probe_k<2, 3, MODE>), dim3((F + per_block - 1) / per_block), dim3(256), lds, 0, frames, z, n_obj, Cf, ell, F);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best;
}
int main() {
  const int F = 25344;
  float *frames, *z, *Cf, *ell;
  hipMalloc(&frames, (size_t)F * 1024 * 4); hipMalloc(&z, (size_t)F * 3 * 4 * 4); hipMalloc(&Cf, kBgDenseF * 4); hipMalloc(&ell, (size_t)F * 36 * 4);
  hipMemset(frames, 0, (size_t)F * 1024 * 4); hipMemset(Cf, 0, kBgDenseF * 4);
  float* hz = (float*)malloc((size_t)F * 12 * 4);
  for (int i = 0; i < F * 3; ++i) { hz[i * 4] = 0.2f; hz[i * 4 + 1] = 0.2f; hz[i * 4 + 2] = 0.1f * (i % 7) - 0.3f; hz[i * 4 + 3] = 0.05f * (i % 11) - 0.2f; }
  hipMemcpy(z, hz, (size_t)F * 12 * 4, hipMemcpyHostToDevice);
  printf("full            %.3f ms\n", run<0>(frames, z, Cf, ell, F));
  printf("no MFMA         %.3f ms\n", run<1>(frames, z, Cf, ell, F));
  printf("no frames/mask  %.3f ms\n", run<2>(frames, z, Cf, ell, F));
  printf("no B loads      %.3f ms\n", run<3>(frames, z, Cf, ell, F));
  printf("tables only     %.3f ms\n", run<4>(frames, z, Cf, ell, F));
  return 0;
}
