// Output head of the recognition network on the matrix cores, one kernel each way (gfx950):
//   codes = fc2(sigmoid(fc1(h)))      h (rows, 256) LSTM outputs, fc1 256 -> HID (50), fc2 HID -> 8     (reference encoder.py:53-56)
// Forward : h is read once (79 MB at 76 800 rows), h1 = sigmoid(fc1) and the codes are written; fc1 never leaves the registers.
// Backward: dcodes, h1, h -> gh = d_a1 W1 (rows, 256) and the parameter gradients; d_a1 never leaves the registers.
// As separate launches (fc1 GEMM, head_fwd_k; head_bwd_k, two GEMMs over 50-wide operands, split-K sums) the head took
// 64 us forward and 237 us backward per step at 76 800 rows; 50-wide matrices are what no tile shape of a general GEMM likes.
//
// All products run as v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulate: no split, no rounding beyond fp32) in their
// TRANSPOSED form, so that the accumulator tile of one product is the B operand of the next without data movement:
// with lane = (c = lane % 16, g = lane / 16) the instruction takes A[i = c][k = g], B[k = g][j = c] and returns D[i = 4g + r][j = c]
// in register r; a D tile whose rows are features and whose columns are batch rows is, register by register, a B operand with
// k = feature 4g + r.  Batch rows therefore always sit on c, features on (g, r).
#include "common.h"

namespace stove {

constexpr int kEhH = 256;       // LSTM width (K of fc1)
constexpr int kEhLd = 264;      // LDS row stride of the W1 image: 264 / 4 = 66 = 2 (mod 16) -> the ds_read_b128 of 16 rows x 4 k-groups
                                // (row c at k = 16 q + 4 g: 16-byte slot 2 c + g + 4 q) are conflict-free in every lane group
constexpr int kEhOut = 8;
constexpr int kEhHid = 64;      // max hidden width (4 feature tiles)

__device__ __forceinline__ f32x4 eh_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float eh_comp(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
// The LSTM hands over its outputs step-major (row = step * n + frame); the codes are wanted frame-major (frame * steps + step,
// reference encoder.py:57: (n, num_obj, 8)).  pn = n > 0: the codes / their gradients live at the transposed row.
__device__ __forceinline__ size_t eh_orow(int row, int pn, int pk) { return pn > 0 ? (size_t)(row % pn) * pk + row / pn : (size_t)row; }

// ---- forward ----------------------------------------------------------------------------------------------------------------------
// One wave = one tile of 16 batch rows x all HID features.  a1^T (feature x row) = W1 (feature x 256) h^T (256 x row):
// A = W1 from the LDS image (one b128 per feature tile and 4 k-steps), B = h straight from global memory -- lane (c, g) owns
// the 16 float4 h[row c][16 q + 4 g], each a whole k-step quadruple (k = 16 q + 4 g + j for step 4 q + j): one load instruction
// covers 64 contiguous bytes of each of the 16 rows (with k = 64 g + 4 q it touched 64 different lines for 16 B each), and
// every byte of h crosses the memory system once.
// SPLIT (round 4; round 5: the pieces are IEEE halves, v_mfma_f32_16x16x32_f16 -- gemm_bf16.hip split4<.., true>): fc1 -- 97 % of the
// kernel's flops -- as split 16-bit MFMAs (hi / lo pieces, three per product,
// fp32 accumulate: the recognition network's GEMMs, gemm_bf16.hip) instead of v_mfma_f32_16x16x4_f32: 96 MFMAs of 16 cycles per 16-row
// tile instead of 256 of 32.  The accumulator layout D[feature 4 g + r][row c] is the same for both instructions, so everything behind
// fc1 is unchanged.  W1 sits in LDS as two bf16 images (rows of 528 bytes: conflict-free b128 fragment reads), h is split in registers.
constexpr int kEhLdB = 528;
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void enc_head_fwd_k(const float* __restrict__ h, const float* __restrict__ W1, const float* __restrict__ b1,
                                                       const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ h1,
                                                       float* __restrict__ codes, int rows, int HID, int pn) {
  extern __shared__ __attribute__((aligned(16))) float eh_lds[];      // W1 image [64][kEhLd], rows >= HID zero
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  const int c = lane & 15, g = lane >> 4;
  // W1 -> LDS with all of a thread's loads in flight before the first store (a load-store loop pays one memory latency per
  // trip); the constants and the first tile's rows are requested in the same breath, so the kernel starts after ONE latency
  constexpr int NP = kEhHid * (kEhH / 4) / 256;
  float4 wst[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int i = tid + 256 * u, n = i / (kEhH / 4), k4 = i % (kEhH / 4);
    wst[u] = ld4(W1 + (size_t)(n < HID ? n : 0) * kEhH + 4 * k4);
  }
  // fc2 as A operand: W2[o = c][feature 16 jt + 4 g + s]; fc1 bias per accumulator register
  float a2[4][4], bias1[4][4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // every load unconditional on a clamped address, then a select: a load inside a divergent branch is waited for before
      // the next one is issued, and 32 (forward) to 90 (backward) of them in a row cost tens of microseconds
      const int n = 16 * jt + 4 * g + s;
      const bool ok = n < HID;
      const float w2v = W2[(c < kEhOut ? c : 0) * HID + (ok ? n : 0)], b1v = b1[ok ? n : 0];
      a2[jt][s] = (c < kEhOut && ok) ? w2v : 0.0f;
      bias1[jt][s] = ok ? b1v : 0.0f;
    }
  const float4 bias2 = ld4(b2 + 4 * (g & 1));
  const int n_tiles = (rows + 15) / 16, nw = blockDim.x >> 6;
  const bool even = (HID & 1) == 0;
  int t = blockIdx.x * nw + wv;
  float4 hn[16];                 // the next tile's rows, fetched while the current tile is in the matrix pipe
  auto fetch = [&](int tt) {
    const int row = tt * 16 + c;
    if (SPLIT) {          // lane (c, g): the 8 consecutive k = 32 q + 8 g .. + 7 of row c per k-block (128 contiguous bytes per row and load pair)
      const float* hp = h + (size_t)(row < rows ? row : rows - 1) * kEhH + 8 * g;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        hn[2 * q] = ld4(hp + 32 * q);
        hn[2 * q + 1] = ld4(hp + 32 * q + 4);
      }
    } else {
      const float* hp = h + (size_t)(row < rows ? row : rows - 1) * kEhH + 4 * g;
#pragma unroll
      for (int q = 0; q < 16; ++q) hn[q] = ld4(hp + 16 * q);
    }
  };
  if (t < n_tiles) fetch(t);
  char* img_hi = reinterpret_cast<char*>(eh_lds);
  char* img_lo = img_hi + kEhHid * kEhLdB;
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int i = tid + 256 * u, n = i / (kEhH / 4), k4 = i % (kEhH / 4);
    const float4 wv4 = n < HID ? wst[u] : float4{0.0f, 0.0f, 0.0f, 0.0f};
    if (SPLIT) {
      u32x2 hi, lo;
      // half pieces (round 5: 2^-22 of the value instead of bf16's 2^-18) of 2^8 W1 -- shifted so that the lo piece of a small weight
      // stays out of half's subnormals (gemm_bf16.hip, kGemmHalfShift); the sigmoid's argument takes the shift out again
      constexpr float ws = (float)(1 << kGemmHalfShift);
      split4<2, true>(float4{wv4.x * ws, wv4.y * ws, wv4.z * ws, wv4.w * ws}, hi, lo);
      *reinterpret_cast<u32x2*>(img_hi + n * kEhLdB + 8 * k4) = hi;
      *reinterpret_cast<u32x2*>(img_lo + n * kEhLdB + 8 * k4) = lo;
    } else {
      st4(eh_lds + n * kEhLd + 4 * k4, wv4);
    }
  }
  __syncthreads();
  const float* wrow = eh_lds + c * kEhLd + 4 * g;
  for (; t < n_tiles; t += gridDim.x * nw) {
    const int row = t * 16 + c;
    const bool live = row < rows;
    float4 hb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) hb[q] = hn[q];
    if (t + gridDim.x * nw < n_tiles) fetch(t + gridDim.x * nw);
    f32x4 acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (SPLIT) {
      const char* fr = img_hi + c * kEhLdB + 16 * g;          // fragment of feature tile jt, k-block q: + 16 jt rows, + 64 q bytes
      bf16x8 ahi[4], alo[4], nhi[4], nlo[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        ahi[jt] = *reinterpret_cast<const bf16x8*>(fr + 16 * jt * kEhLdB);
        alo[jt] = *reinterpret_cast<const bf16x8*>(fr + kEhHid * kEhLdB + 16 * jt * kEhLdB);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q + 1 < 8) {
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) {
            nhi[jt] = *reinterpret_cast<const bf16x8*>(fr + 16 * jt * kEhLdB + 64 * (q + 1));
            nlo[jt] = *reinterpret_cast<const bf16x8*>(fr + kEhHid * kEhLdB + 16 * jt * kEhLdB + 64 * (q + 1));
          }
        }
        u32x2 h0, l0, h1_, l1;
        split4<2, true>(hb[2 * q], h0, l0);
        split4<2, true>(hb[2 * q + 1], h1_, l1);
        const bf16x8 bhi = __builtin_bit_cast(bf16x8, u32x4{h0.x, h0.y, h1_.x, h1_.y});
        const bf16x8 blo = __builtin_bit_cast(bf16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[jt] = mfma16<true>(alo[jt], bhi, acc[jt]);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[jt] = mfma16<true>(ahi[jt], blo, acc[jt]);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[jt] = mfma16<true>(ahi[jt], bhi, acc[jt]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          ahi[jt] = nhi[jt];
          alo[jt] = nlo[jt];
        }
      }
    } else {
      // W1 fragments one quadruple of k-steps ahead of the MFMAs that use them (the scheduling barrier keeps the compiler from
      // hoisting all 64 LDS reads to the top: 256 registers)
      float4 a[4], an[4];
  #pragma unroll
      for (int jt = 0; jt < 4; ++jt) a[jt] = ld4(wrow + 16 * jt * kEhLd);
  #pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (q + 1 < 16) {
  #pragma unroll
          for (int jt = 0; jt < 4; ++jt) an[jt] = ld4(wrow + 16 * jt * kEhLd + 16 * (q + 1));
        }
  #pragma unroll
        for (int j = 0; j < 4; ++j)
  #pragma unroll
          for (int jt = 0; jt < 4; ++jt) acc[jt] = eh_mfma(eh_comp(a[jt], j), eh_comp(hb[q], j), acc[jt]);
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int jt = 0; jt < 4; ++jt) a[jt] = an[jt];
      }
    }
    // h1 = sigmoid(a1 + b1): this lane holds h1[row c][16 jt + 4 g + r]; features >= HID are exact zeros (fc2 operand)
    float hv[4][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[jt][r] = (16 * jt + 4 * g + r < HID) ? sig_(acc[jt][r] * (SPLIT ? 1.0f / (float)(1 << kGemmHalfShift) : 1.0f) + bias1[jt][r]) : 0.0f;
    if (live) {
      float* o = h1 + (size_t)row * HID + 4 * g;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int n = 16 * jt + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          if (even && n + r + 1 < HID) {
            *reinterpret_cast<float2*>(o + 16 * jt + r) = float2{hv[jt][r], hv[jt][r + 1]};       // (row HID + n) even: 8-byte aligned
          } else {
            if (n + r < HID) o[16 * jt + r] = hv[jt][r];
            if (n + r + 1 < HID) o[16 * jt + r + 1] = hv[jt][r + 1];
          }
        }
      }
    }
    // codes^T (output x row) = W2 (output x feature) h1^T: the h1 registers are the B operand as they are
    f32x4 cd = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < 4; ++s) cd = eh_mfma(a2[jt][s], hv[jt][s], cd);
    if (live && g < 2) st4(codes + eh_orow(row, pn, pn > 0 ? rows / pn : 0) * kEhOut + 4 * g, float4{cd[0] + bias2.x, cd[1] + bias2.y, cd[2] + bias2.z, cd[3] + bias2.w});
  }
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------
// Workgroup = (row group, role).  Roles 0..3 own a quarter of the 256 LSTM columns, m = 64 w + 16 mt + .., their four waves
// take different 16-row tiles of the group:
//   1. d_h1^T (feature x row) = W2^T (feature x 8) dcodes^T (8 x row)                       8 MFMAs
//   2. d_a1 = d_h1 h1 (1 - h1), h1 loaded in the same (feature, row) register layout
//   3. gh^T (m x row) = W1^T (m x feature) d_a1^T: d_a1 registers as B, W1 fragments from an LDS image in operand order   4 x KS MFMAs
//   4. gw1 (feature x m) += d_a1^T (feature x row) h (row x m): d_a1 turned into an A operand through a wave-private LDS slice,
//      h read once by the quarter that owns its columns; 64 accumulator registers per wave over the whole launch      64 MFMAs
// Role 4 repeats 1 and 2 (cheap) for the bias sums gb1, gb2 and forms gw2 (8 x feature) = dcodes^T h1: kept out of the
// column roles, whose register budget goes to the 64 accumulators and to the next tile's operands (fetched a tile ahead).
// The four waves add their accumulators through LDS in wave order; part[group] = [gw1 (HID x 256) | gw2 (8 x HID) | gb1 | gb2]
// is summed over the groups by reduce_chunks_k (fixed order: bit-reproducible).
__host__ __device__ inline int eh_part_floats(int HID) { return (HID * kEhH + kEhOut * HID + HID + kEhOut + 3) / 4 * 4; }
constexpr int kEhTLd = 17;      // d_a1 slice [feature][row]: odd stride -> the A-operand reads (feature c, row g + 4 s) are conflict-free
constexpr int kEhBwdLds = 4 * 4 * 64 * 4 + 4 * kEhHid * kEhTLd + 256;      // floats: W1 fragment image | 4 slices (the reductions alias both)
// (Round 4 also wrote the column roles' products as split-bf16 MFMAs: 101 -> 72 us in the three-object step with the step unchanged, and
// 274 -> 523 us at six objects next to gnn_dw_small_k, step + 55 us -- docs/history/r04.md; removed in round 5.)

struct EhTileIn {      // what a column role fetches for one tile
  float gv[2];         // dcodes[row c][g + 4 s]
  float2 h1v[8];       // h1[row c][16 jt + 4 g + (0,1 | 2,3)]
  float hb[4][4];      // h[row g + 4 s][m0 + 16 mt + c]
};
__device__ __forceinline__ void eh_fetch(EhTileIn& x, const float* __restrict__ dcodes, const float* __restrict__ h1, const float* __restrict__ h,
                                         int t, int rows, int HID, int m0, int c, int g, bool even, bool with_h, int pn) {
  const int R0 = t * 16, row = R0 + c;
  const int rowc = row < rows ? row : rows - 1;
  const size_t orow = eh_orow(rowc, pn, pn > 0 ? rows / pn : 0);
#pragma unroll
  for (int s = 0; s < 2; ++s) x.gv[s] = dcodes[orow * kEhOut + g + 4 * s];
  const float* hp = h1 + (size_t)rowc * HID;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const int n = 16 * jt + 4 * g + r;
      if (even) {             // wave-uniform; (row HID + n) even: 8-byte aligned; an out-of-range pair reads feature pair 0
        x.h1v[jt * 2 + r / 2] = *reinterpret_cast<const float2*>(hp + (n + 1 < HID ? n : 0));
      } else {
        x.h1v[jt * 2 + r / 2] = float2{hp[n < HID ? n : 0], hp[n + 1 < HID ? n + 1 : 0]};
      }
    }
  if (with_h) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int rr = R0 + g + 4 * s;
      const float* q = h + (size_t)(rr < rows ? rr : rows - 1) * kEhH + m0 + c;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) x.hb[mt][s] = q[16 * mt];
    }
  }
}
// products 1 and 2: d_a1[feature 16 jt + 4 g + r][row c] in da[jt][r], zero for rows / features outside the problem
__device__ __forceinline__ void eh_da(const EhTileIn& x, const float (&a1)[4][2], bool live, int g, int HID, float (&gv)[2], float (&da)[4][4]) {
  gv[0] = live ? x.gv[0] : 0.0f;
  gv[1] = live ? x.gv[1] : 0.0f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    f32x4 d = eh_mfma(a1[jt][0], gv[0], f32x4{0.0f, 0.0f, 0.0f, 0.0f});
    d = eh_mfma(a1[jt][1], gv[1], d);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float hv = (r & 1) ? x.h1v[jt * 2 + r / 2].y : x.h1v[jt * 2 + r / 2].x;
      da[jt][r] = (16 * jt + 4 * g + r < HID) ? d[r] * hv * (1.0f - hv) : 0.0f;      // dcodes = 0 on dead rows makes those columns zero
    }
  }
}

template <int KS>               // k-steps of product 3 = feature quadruples (jt, s) with 16 jt + s < HID, in (jt, s) order
__global__ __launch_bounds__(256, 2) void enc_head_bwd_k(const float* __restrict__ dcodes, const float* __restrict__ h1, const float* __restrict__ h,
                                                          const float* __restrict__ W1, const float* __restrict__ W2, float* __restrict__ gh,
                                                          float* __restrict__ part, int rows, int HID, int n_groups, int pn) {
  extern __shared__ __attribute__((aligned(16))) float eh_lds[];
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  const int c = lane & 15, g = lane >> 4;
  const int w = blockIdx.x % 5, grp = blockIdx.x / 5;
  const int n_tiles = (rows + 15) / 16;
  const bool even = (HID & 1) == 0;
  float* P = part + (size_t)grp * eh_part_floats(HID);
  float a1[4][2];               // product 1: W2[o = g + 4 s][feature c + 16 jt]
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bool ok = c + 16 * jt < HID;
      const float v = W2[(g + 4 * s) * HID + (ok ? c + 16 * jt : 0)];
      a1[jt][s] = ok ? v : 0.0f;
    }
  if (w == 4) {
    // ---- role 4: gw2, gb1, gb2 -----------------------------------------------------------------------------------------------------
    f32x4 accw2[4];
    float accb1[4][4], accb2[2] = {0.0f, 0.0f};
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      accw2[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int r = 0; r < 4; ++r) accb1[jt][r] = 0.0f;
    }
    for (int t = grp * 4 + wv; t < n_tiles; t += n_groups * 4) {
      const int R0 = t * 16;
      EhTileIn x;
      eh_fetch(x, dcodes, h1, h, t, rows, HID, 0, c, g, even, false, pn);
      // gw2: A = dcodes[row g + 4 s][o = c], B = h1[row g + 4 s][feature c + 16 jt]
      float av[4], bv[4][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int rr = R0 + g + 4 * s, rc = rr < rows ? rr : rows - 1;
        av[s] = dcodes[eh_orow(rc, pn, pn > 0 ? rows / pn : 0) * kEhOut + (c & (kEhOut - 1))];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) bv[jt][s] = h1[(size_t)rc * HID + (c + 16 * jt < HID ? c + 16 * jt : 0)];
      }
      float gv[2], da[4][4];
      eh_da(x, a1, R0 + c < rows, g, HID, gv, da);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) accb1[jt][r] += da[jt][r];
      accb2[0] += gv[0];
      accb2[1] += gv[1];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = R0 + g + 4 * s < rows;
        const float a = (ok && c < kEhOut) ? av[s] : 0.0f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) accw2[jt] = eh_mfma(a, (ok && c + 16 * jt < HID) ? bv[jt][s] : 0.0f, accw2[jt]);
      }
    }
    float* S = eh_lds + wv * (34 * 64);         // [16 gb1 partials | 2 gb2 partials | 16 gw2] x 64 lanes
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[(jt * 4 + r) * 64 + lane] = accb1[jt][r];
        S[(18 + jt * 4 + r) * 64 + lane] = accw2[jt][r];
      }
    S[16 * 64 + lane] = accb2[0];
    S[17 * 64 + lane] = accb2[1];
    __syncthreads();
    float* Pw2 = P + (size_t)HID * kEhH;
    float* Pb1 = Pw2 + kEhOut * HID;
    float* Pb2 = Pb1 + HID;
    auto four = [&](int idx) {           // element idx of the per-wave image, waves in order
      return ((eh_lds[idx] + eh_lds[34 * 64 + idx]) + eh_lds[2 * 34 * 64 + idx]) + eh_lds[3 * 34 * 64 + idx];
    };
    for (int e = tid; e < kEhOut * HID; e += blockDim.x) {          // gw2[o = 4 g + r][feature 16 jt + c]
      const int o = e / HID, n = e % HID;
      Pw2[e] = four((18 + (n >> 4) * 4 + (o & 3)) * 64 + 16 * (o >> 2) + (n & 15));
    }
    if (tid < HID) {                                                  // gb1[feature 16 jt + 4 g + r] = sum over the 16 rows c
      const int n = tid, base = ((n >> 4) * 4 + (n & 3)) * 64 + 16 * ((n >> 2) & 3);
      float s = 0.0f;
      for (int cc = 0; cc < 16; ++cc) s += four(base + cc);
      Pb1[n] = s;
    }
    if (tid < kEhOut) {                                               // gb2[o = g + 4 s]
      const int o = tid, base = (16 + (o >> 2)) * 64 + 16 * (o & 3);
      float s = 0.0f;
      for (int cc = 0; cc < 16; ++cc) s += four(base + cc);
      Pb2[o] = s;
    }
    for (int e = HID * kEhH + kEhOut * HID + HID + kEhOut + tid; e < eh_part_floats(HID); e += blockDim.x) P[e] = 0.0f;      // padding
    return;
  }
  // ---- roles 0..3: gh and gw1 of the column quarter -------------------------------------------------------------------------------
  const int m0 = 64 * w;
  float* A3 = eh_lds;                                   // [mt][kq][lane][4]: W1[feature 16 kq + 4 g + j][m0 + 16 mt + c], zero beyond HID
  float* T = eh_lds + 4 * 4 * 64 * 4 + wv * (kEhHid * kEhTLd);
  {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = tid + 256 * u, j = i & 3, ln = (i >> 2) & 63, kq = (i >> 8) & 3, mt = i >> 10;
      const int n = 16 * kq + 4 * (ln >> 4) + j;
      v[u] = W1[(size_t)(n < HID ? n : 0) * kEhH + m0 + 16 * mt + (ln & 15)];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = tid + 256 * u, j = i & 3, ln = (i >> 2) & 63, kq = (i >> 8) & 3;
      A3[i] = (16 * kq + 4 * (ln >> 4) + j < HID) ? v[u] : 0.0f;
    }
  }
  f32x4 accw[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) accw[nt][mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  __syncthreads();
  for (int t = grp * 4 + wv; t < n_tiles; t += n_groups * 4) {
    const int R0 = t * 16, row = R0 + c;
    const bool live = row < rows;
    // no tile-ahead fetch: its 34 registers would put the kernel above the 192 that gnn_dw_small_k (the recursion's
    // weight-gradient pass on the side stream, 320 registers per lane, memory-bound) leaves on a SIMD -- sharing the SIMDs
    // with it is worth more than hiding this kernel's own load latency
    EhTileIn x;
    eh_fetch(x, dcodes, h1, h, t, rows, HID, m0, c, g, even, true, pn);
    float gv[2], da[4][4];
    eh_da(x, a1, live, g, HID, gv, da);
    // 3. gh
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int kq = 0; kq < (KS + 3) / 4; ++kq) {
        const float4 a = ld4(A3 + ((mt * 4 + kq) * 64 + lane) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * kq + j < KS) o = eh_mfma(eh_comp(a, j), da[kq][j], o);
      }
      if (live) st4(gh + (size_t)row * kEhH + m0 + 16 * mt + 4 * g, float4{o[0], o[1], o[2], o[3]});
    }
    // 4. d_a1 as A operand (feature c + 16 nt, row g + 4 s) through the wave's slice
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(16 * jt + 4 * g + r) * kEhTLd + c] = da[jt][r];
    __builtin_amdgcn_wave_barrier();
    float a4[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int s = 0; s < 4; ++s) a4[nt][s] = T[(c + 16 * nt) * kEhTLd + g + 4 * s];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = R0 + g + 4 * s < rows;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float b = ok ? x.hb[mt][s] : 0.0f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) accw[nt][mt] = eh_mfma(a4[nt][s], b, accw[nt][mt]);
      }
    }
  }
  // ---- the four waves' sums, in wave order, two feature-tile pairs at a time (32 KB of LDS) -----------------------------------------
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    float* R = eh_lds + wv * (32 * 64);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) R[((nt * 4 + mt) * 4 + r) * 64 + lane] = accw[2 * half + nt][mt][r];
    __syncthreads();
    for (int e = tid; e < 32 * 64; e += blockDim.x) {
      const int reg = e >> 6, ln = e & 63;
      const float v = ((eh_lds[e] + eh_lds[2048 + e]) + eh_lds[2 * 2048 + e]) + eh_lds[3 * 2048 + e];
      const int nt = 2 * half + (reg >> 4), mt = (reg >> 2) & 3, r = reg & 3;
      const int n = 16 * nt + 4 * (ln >> 4) + r, m = m0 + 16 * mt + (ln & 15);
      if (n < HID) P[(size_t)n * kEhH + m] = v;
    }
  }
}

// part[group][gw1 | gw2 | gb1 | gb2] summed over the groups in group order (256 threads = 32 elements x 8 group slices, as
// reduce_chunks_k) and handed to the four gradient tensors; accumulate: added to what they hold.
__global__ __launch_bounds__(256) void enc_head_reduce_k(const float* __restrict__ part, float* __restrict__ gW1, float* __restrict__ gW2,
                                                          float* __restrict__ gb1, float* __restrict__ gb2, int HID, int n_groups, int accumulate) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + e;
  const int P = eh_part_floats(HID), n = HID * kEhH + kEhOut * HID + HID + kEhOut;
  float s = 0.0f;
  if (j < n)
    for (int c = q; c < n_groups; c += 8) s += part[(size_t)c * P + j];
  red[q][e] = s;
  __syncthreads();
  if (q == 0 && j < n) {
    float t = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][e];
    const int o1 = HID * kEhH, o2 = o1 + kEhOut * HID, o3 = o2 + HID;
    float* dst = j < o1 ? gW1 + j : j < o2 ? gW2 + (j - o1) : j < o3 ? gb1 + (j - o2) : gb2 + (j - o3);
    *dst = accumulate ? *dst + t : t;
  }
}

}  // namespace stove
