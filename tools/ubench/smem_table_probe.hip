// Probe: can wave-uniform weight tables feed the VALU as scalar operands (s_load -> SGPR -> v_fmac) at full rate?  Each wave runs the
// object SPN's sum layer shape (100 products x 10 sum nodes = 1000 fmac on 1000 table floats, plus a 750-fmac leaf-like sweep on 1500
// floats) once; what varies is how many DISTINCT 10 KB tables the workgroups resident on a CU read (scalar-cache working set).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o smem_table_probe smem_table_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef LEAF_UNROLL
#define LEAF_UNROLL 2
#endif
#ifndef SUM_UNROLL
#define SUM_UNROLL 1
#endif
constexpr int kTab = 2500;            // floats per table: 1500 leaf coefficients + 1000 sum weights
__device__ __forceinline__ float e2_at(const float (&e2)[10], int j) {   // register array, uniform runtime index: a select chain
  float v = e2[0];
#pragma unroll
  for (int q = 1; q < 10; ++q) v = (j == q) ? e2[q] : v;
  return v;
}
template <int MODE>                   // 0: every block the same table; 1: table = block % 12; 2: table-major (block / blocks_per_table)
__global__ __launch_bounds__(256) void probe_k(const float* __restrict__ tabs, const float* __restrict__ x, float* __restrict__ out, int blocks_per_table) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int table = MODE == 0 ? 0 : (MODE == 1 ? (int)(blockIdx.x % 12) : (int)(blockIdx.x / blocks_per_table));
  const float* __restrict__ T = tabs + (size_t)table * kTab;
  const float* xb = x + ((size_t)blockIdx.x * 4 + wv) * 64 * 8;
  float e1[10], e2[10];
  {
    float acc[10];
#pragma unroll
    for (int g = 0; g < 10; ++g) acc[g] = 0.0f;
    float xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = xb[i * 64 + lane];
#pragma unroll LEAF_UNROLL
    for (int i = 0; i < 50; ++i) {
      const float w = xv[0] + i, wx = w * xv[1], wxx = wx * xv[1];
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        const float* c3 = T + (i * 10 + g) * 3;
        acc[g] = fmaf(wxx, c3[0], acc[g]);
        acc[g] = fmaf(wx, c3[1], acc[g]);
        acc[g] = fmaf(w, c3[2], acc[g]);
      }
      if (i == 24) {   // (uniform branch)
#pragma unroll
        for (int g = 0; g < 10; ++g) { e1[g] = acc[g]; acc[g] = 0.0f; }
      }
    }
#pragma unroll
    for (int g = 0; g < 10; ++g) e2[g] = acc[g];
  }
  float o[10];
#pragma unroll
  for (int s = 0; s < 10; ++s) o[s] = 0.0f;
  const float* W = T + 1500;
#pragma unroll SUM_UNROLL
  for (int j2 = 0; j2 < 10; ++j2)
  { const float e2j = e2_at(e2, j2);
#pragma unroll
    for (int j1 = 0; j1 < 10; ++j1) {
      const float t = e1[j1] * e2j;
#pragma unroll
      for (int s = 0; s < 10; ++s) o[s] = fmaf(t, W[(j2 * 10 + j1) * 10 + s], o[s]);
    } }
  float r = 0.0f;
#pragma unroll
  for (int s = 0; s < 10; ++s) r += o[s];
  out[((size_t)blockIdx.x * 4 + wv) * 64 + lane] = r;
}
template <int MODE>
static float run(const float* tabs, const float* x, float* out, int nblk) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe_k<MODE>, dim3(nblk), dim3(256), 0, 0, tabs, x, out, nblk / 12);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe_k<MODE>, dim3(nblk), dim3(256), 0, 0, tabs, x, out, nblk / 12);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / 20;
}
int main() {
  const int nblk = 12 * 297;               // 14 256 wave-batches, four per workgroup: the object SPN's forward at 76 032 glimpses
  std::vector<float> ht(12 * kTab), hx((size_t)nblk * 4 * 64 * 8);
  for (size_t i = 0; i < ht.size(); ++i) ht[i] = 0.001f * (float)((i * 7919) % 997) - 0.5f;
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0.001f * (float)((i * 104729) % 991);
  float *tabs, *x, *out;
  hipMalloc(&tabs, ht.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&out, (size_t)nblk * 256 * 4);
  hipMemcpy(tabs, ht.data(), ht.size() * 4, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  // floor: 14 256 wave-batches x ~2 750 VALU x 2 cycles / 1024 SIMDs
  printf("one table for every block        %7.1f us\n", run<0>(tabs, x, out, nblk));
  printf("table = block %% 12 (interleaved)  %7.1f us\n", run<1>(tabs, x, out, nblk));
  printf("table-major (297 blocks each)    %7.1f us\n", run<2>(tabs, x, out, nblk));
  printf("VALU floor at 2 cycles per fmac: %7.1f us (2.1 GHz)\n", 14256.0 * 2750 * 2 / 1024 / 2100.0);
  return 0;
}
