// Sixteen-bit hi / lo pieces of fp32 operands and the MFMA that multiplies them (shared by the recognition network's GEMMs,
// csrc/gemm_bf16.hip, the fused head, csrc/head_fused.hip, and the edge chains of the small-graph recursion, csrc/gnn_small.hip).
#pragma once
#include "common.h"

namespace stove {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4_;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  const f16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// hi / lo 16-bit pieces of 4 floats: hi[2], lo[2] dwords.  F16 = false: bf16 pieces (8 + 8 significant bits, fp32's exponent range:
// any operand, the gradients of the backward included).  F16 = true (round 5): IEEE half pieces, 11 + 11 significant bits -- the
// two pieces carry 2^-22 of the value instead of 2^-18, i.e. the three-MFMA product is as good as an fp32 one -- for operands
// inside half's range: the FORWARD products of the recognition network (frames in [0, 1], hidden states in (-1, 1), weights),
// whose error the head's weights amplify into the codes (tests: the 'stress' weight regime).  Same MFMA rate, same images.
// Range of the half pieces: hi is a normal half for 2^-14 <= |x| < 65504; lo, at most 2^-11 |x|, is a normal half only for |x| >= 1/4 and
// a subnormal below (granularity 2^-24): an element keeps max(2^-22 |x|, 2^-25).  Operands whose scale is known to be small -- the
// weights of the forward products -- are therefore cut as 2^kGemmHalfShift x (exact; the product's epilogue shifts back): every
// |w| >= 2^-10 keeps its 22 bits, |w| < 256 stays in range.
constexpr int kGemmHalfShift = 8;
template <int NSPLIT, bool F16 = false>
__device__ __forceinline__ void split4(const float4 v, u32x2& hi, u32x2& lo) {
  if constexpr (F16) {
    // hi = v_cvt_pk_f16_f32 (round to nearest even), residual x - (float)hi as ONE v_fma_mix_f32 per element (the half operand is
    // converted inside the instruction: fma(hi_half, -1, x), exact), lo = v_cvt_pk_f16_f32 of the residuals: 8 instructions per
    // float4 (the bf16 form below takes 12).  Inline asm: written in C++ with the residuals taken from the packed words, hipcc 7.2
    // -O3 subtracts the halves of hi.x from BOTH pairs (a miscompile seen in round 5), and element-wise conversions cost 16.
    hi.x = pack_f16(v.x, v.y);
    hi.y = pack_f16(v.z, v.w);
    if (NSPLIT == 2) {
      float rx, ry, rz, rw;
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(hi.x), "v"(v.x));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(hi.x), "v"(v.y));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rz) : "v"(hi.y), "v"(v.z));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rw) : "v"(hi.y), "v"(v.w));
      lo.x = pack_f16(rx, ry);
      lo.y = pack_f16(rz, rw);
    }
  } else {
    hi.x = pack_bf16(v.x, v.y);
    hi.y = pack_bf16(v.z, v.w);
    if (NSPLIT == 2) {
      const float rx = v.x - __uint_as_float(hi.x << 16), ry = v.y - __uint_as_float(hi.x & 0xffff0000u);
      const float rz = v.z - __uint_as_float(hi.y << 16), rw = v.w - __uint_as_float(hi.y & 0xffff0000u);
      lo.x = pack_bf16(rx, ry);
      lo.y = pack_bf16(rz, rw);
    }
  }
}
// one 16 x 16 x 32 MFMA on two fragments of 8 sixteen-bit pieces (held as bf16x8 whatever they are)
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

}  // namespace stove
