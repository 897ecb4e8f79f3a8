// v_permlane32_swap (gfx950) through the builtin: what the two results hold.  hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane32_swap.hip -o /tmp/pl && /tmp/pl
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
  const float v = (float)threadIdx.x;
  const unsigned a = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);          // one value twice: one register twice
  o[threadIdx.x] = __builtin_bit_cast(float, r[0]);
  o[64 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
  unsigned b;
  asm volatile("v_mov_b32 %0, %1" : "=v"(b) : "v"(a));
  const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);          // two registers
  o[128 + threadIdx.x] = __builtin_bit_cast(float, q[0]);
  o[192 + threadIdx.x] = __builtin_bit_cast(float, q[1]);
  float lo = v, hi;                                                              // the instruction itself, as inline asm
  asm volatile("v_mov_b32 %0, %1" : "=v"(hi) : "v"(lo));
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
  o[256 + threadIdx.x] = lo;
  o[320 + threadIdx.x] = hi;
  float a16 = v, b16;                                                            // v_permlane16_swap
  asm volatile("v_mov_b32 %0, %1" : "=v"(b16) : "v"(a16));
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a16), "+v"(b16));
  o[384 + threadIdx.x] = a16;
  o[448 + threadIdx.x] = b16;
}
int main() {
  float* d; hipMalloc(&d, 512 * 4);
  k<<<1, 64>>>(d);
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[0], h[31], h[32], h[63]);
  printf("r1: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[64], h[95], h[96], h[127]);
  printf("two registers:\nq0: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[128], h[159], h[160], h[191]);
  printf("q1: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[192], h[223], h[224], h[255]);
  printf("inline asm:\nlo: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[256], h[287], h[288], h[319]);
  printf("hi: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[320], h[351], h[352], h[383]);
  printf("v_permlane16_swap:\na: lane0 %g lane16 %g lane32 %g lane48 %g\n", h[384], h[400], h[416], h[432]);
  printf("b: lane0 %g lane16 %g lane32 %g lane48 %g\n", h[448], h[464], h[480], h[496]);
  return 0;
}
