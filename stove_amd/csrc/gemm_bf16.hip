// fp32 GEMMs of the recognition network (reference encoder.py:43-57: torch.nn.LSTM + fc1) on the bf16 matrix cores.
//
// The five dense products of RnnStates -- x W_ih^T, h W_hh^T and, in the backward, dg W_hh, dg^T h, dgx^T x -- are fp32 GEMMs
// in the reference.  gfx950 runs f32-input MFMA at 1/16 of the bf16 rate, so the fp32 operands are split ON THE FLY into
// bf16 pieces while they are staged into LDS,
//      x = hi + lo + O(2^-16 |x|),   hi = bf16(x),  lo = bf16(x - hi),
// and every product is three bf16 MFMAs with fp32 accumulation:  a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi   (NSPLIT = 2; the
// dropped lo*lo term and the residuals are ~2^-16 relative per product, random in sign, against 2^-8 for plain bf16 operands).
// NSPLIT = 1 is the plain bf16-operand GEMM (config.encoder_bf16, BASELINE.json configs[1] "bf16").  Inputs, outputs and the
// accumulators stay fp32; nothing is pre-converted in HBM.
//
//   C[m][n] = sum_k a(m,k) b(n,k) (+ bias[n]) (+ D[m][n]),   a(m,k) = A[m lda + k]  or  A[k lda + m] (A_KMAJOR),  b likewise.
//
// Workgroup = 512 threads = 8 waves (4 along M x 2 along N), tile 256 x 128 x 32, wave tile 64 x 64 = 4 x 4 MFMA tiles of
// v_mfma_f32_16x16x32_bf16 (64 accumulator registers).  Per k-step every thread loads 6 float4 of the NEXT tile from global
// memory (issued before the MFMAs of the current tile), converts them after the MFMAs and writes the bf16 images of the other LDS
// buffer: one barrier per k-step.
// LDS images (bf16):
//   * operand stored K-contiguous ([row][k]): rows of 32 k = 64 B, written as 8-byte pieces, read as the 16-byte MFMA fragment
//     (ds_read_b128) at byte (16 g) ^ (((row >> 3) & 1) << 5) of the row -- conflict-free for the b128 lane groups;
//   * operand stored K-major ([k][row]): image [k][rows + 16] (row stride = 32 B mod 256 B), written as it arrives (4 rows of one
//     k = 8 bytes), read TRANSPOSED with ds_read_b64_tr_b16 (two per fragment), conflict-free as well.
//   The 32 k of a step are assigned to the MFMA k-slots as  slot 8 g + j  <->  k = 4 g + j (j < 4), 16 + 4 g + (j - 4) (j >= 4)
//   for BOTH operands (a sum over k does not care), which is what makes the transposed reads of one half-wave touch 8
//   consecutive k-rows.
// The MFMA is issued as D^T = B-frag x A-frag, so a lane ends up with 4 consecutive n of one row m: float4 stores.
// Workgroup -> tile map: XCD-aware (ids congruent mod 8 share an L2): every XCD walks a contiguous range of tiles, n fastest,
// so the workgroups that re-read one 256-row A panel run on the same L2.
#include "common.h"

namespace stove {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4_;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kGemmBM = 256, kGemmBN = 128, kGemmBK = 32, kGemmThreads = 512;
constexpr int kGemmAStrideKM = (kGemmBM + 16) * 2;        // bytes per k-row of a K-major A image
constexpr int kGemmBStrideKM = (kGemmBN + 16) * 2;
constexpr int kGemmAPart = kGemmBK * kGemmAStrideKM;      // 17 408 B >= 256 rows x 64 B
constexpr int kGemmBPart = kGemmBK * kGemmBStrideKM;      //  9 216 B >= 128 rows x 64 B

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// hi / lo bf16 pieces of 4 floats: hi[2], lo[2] dwords
template <int NSPLIT>
__device__ __forceinline__ void split4(const float4 v, u32x2& hi, u32x2& lo) {
  hi.x = pack_bf16(v.x, v.y);
  hi.y = pack_bf16(v.z, v.w);
  if (NSPLIT == 2) {
    const float rx = v.x - __uint_as_float(hi.x << 16), ry = v.y - __uint_as_float(hi.x & 0xffff0000u);
    const float rz = v.z - __uint_as_float(hi.y << 16), rw = v.w - __uint_as_float(hi.y & 0xffff0000u);
    lo.x = pack_bf16(rx, ry);
    lo.y = pack_bf16(rz, rw);
  }
}

// one operand tile (ROWS x 32) of k-step `k0`: NLD float4 per thread
template <int ROWS, bool KMAJOR>
struct TileLoad {
  static constexpr int NLD = ROWS * kGemmBK / 4 / kGemmThreads;
  float4 v[NLD];
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int n_rows, int k0, int k_end, int tid) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + kGemmThreads * j;
      int row, k;
      if (KMAJOR) {
        k = k0 + f / (ROWS / 4);
        row = row0 + 4 * (f % (ROWS / 4));
      } else {
        row = row0 + (f >> 3);
        k = k0 + 4 * (f & 7);
      }
      const bool ok = row < n_rows && k < k_end;
      const float* src = KMAJOR ? P + (size_t)k * ld + row : P + (size_t)row * ld + k;
      v[j] = ok ? *reinterpret_cast<const float4*>(src) : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  }
  template <int NSPLIT>
  __device__ __forceinline__ void store(char* hi_img, char* lo_img, int tid) const {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + kGemmThreads * j;
      int off;
      if (KMAJOR) {
        off = (f / (ROWS / 4)) * ((ROWS + 16) * 2) + (f % (ROWS / 4)) * 8;
      } else {
        const int row = f >> 3, kq = f & 7;
        off = row * 64 + ((((kq & 3) << 4) | ((kq >> 2) << 3)) ^ (((row >> 3) & 1) << 5));
      }
      u32x2 hi, lo;
      split4<NSPLIT>(v[j], hi, lo);
      *reinterpret_cast<u32x2*>(hi_img + off) = hi;
      if (NSPLIT == 2) *reinterpret_cast<u32x2*>(lo_img + off) = lo;
    }
  }
};

// MFMA fragment of the 16 rows starting at `r0` (tile-local) from an image
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ bf16x8 read_frag(const char* img, int r0, int lane) {
  if (KMAJOR) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const char* a = img + (4 * g + q) * ((ROWS + 16) * 2) + (r0 + 4 * p) * 2;
    const short4_ lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_*)(a));
    const short4_ hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_*)(a + 16 * ((ROWS + 16) * 2)));
    typedef __attribute__((ext_vector_type(8))) short short8_;
    const short8_ s = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    return __builtin_bit_cast(bf16x8, s);
  } else {
    const int row = r0 + (lane & 15);
    const int off = row * 64 + (((lane >> 4) << 4) ^ (((row >> 3) & 1) << 5));
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + off));
  }
}

template <bool A_KMAJOR, bool B_KMAJOR, int NSPLIT>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_bf16_k(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                               const float* __restrict__ add, float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                               int tiles_m, int tiles_n, int splitk, int k_per_slice) {
  extern __shared__ __attribute__((aligned(16))) char gemm_lds[];
  constexpr int STAGE = NSPLIT * (kGemmAPart + kGemmBPart);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  // XCD-aware tile id
  const int n_wg = tiles_m * tiles_n * splitk;
  int id = blockIdx.x;
  if ((n_wg & 7) == 0) id = (id & 7) * (n_wg >> 3) + (id >> 3);
  const int tn = id % tiles_n, tm = (id / tiles_n) % tiles_m, z = id / (tiles_n * tiles_m);
  const int m0 = tm * kGemmBM, n0 = tn * kGemmBN;
  const int k_begin = z * k_per_slice;
  const int k_end = k_begin + k_per_slice < K ? k_begin + k_per_slice : K;
  const int nt = (k_end - k_begin + kGemmBK - 1) / kGemmBK;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  TileLoad<kGemmBM, A_KMAJOR> la;
  TileLoad<kGemmBN, B_KMAJOR> lb;
  auto img = [&](int stage, int which) -> char* {      // which: 0 A_hi, 1 A_lo, 2 B_hi, 3 B_lo
    char* s = gemm_lds + stage * STAGE;
    if (NSPLIT == 2) return s + (which < 2 ? which * kGemmAPart : 2 * kGemmAPart + (which - 2) * kGemmBPart);
    return s + (which < 2 ? 0 : kGemmAPart);
  };
  if (nt > 0) {
    la.load(A, lda, m0, M, k_begin, k_end, tid);
    lb.load(B, ldb, n0, N, k_begin, k_end, tid);
    la.template store<NSPLIT>(img(0, 0), img(0, 1), tid);
    lb.template store<NSPLIT>(img(0, 2), img(0, 3), tid);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < nt;
    if (more) {
      la.load(A, lda, m0, M, k_begin + (t + 1) * kGemmBK, k_end, tid);
      lb.load(B, ldb, n0, N, k_begin + (t + 1) * kGemmBK, k_end, tid);
    }
    bf16x8 ah[4], bh[4], al[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = read_frag<kGemmBM, A_KMAJOR>(img(cur, 0), wm * 64 + i * 16, lane);
      bh[i] = read_frag<kGemmBN, B_KMAJOR>(img(cur, 2), wn * 64 + i * 16, lane);
      if (NSPLIT == 2) {
        al[i] = read_frag<kGemmBM, A_KMAJOR>(img(cur, 1), wm * 64 + i * 16, lane);
        bl[i] = read_frag<kGemmBN, B_KMAJOR>(img(cur, 3), wn * 64 + i * 16, lane);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (NSPLIT == 2) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
      }
    if (more) {
      la.template store<NSPLIT>(img(cur ^ 1, 0), img(cur ^ 1, 1), tid);
      lb.template store<NSPLIT>(img(cur ^ 1, 2), img(cur ^ 1, 3), tid);
    }
    __syncthreads();
  }
  // epilogue: lane holds C[m][n .. n+3], m = tile row (lane & 15), n = 4 (lane >> 4) + reg
  float* out = C + (splitk > 1 ? (size_t)z * M * ldc : 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      if (m < M && n < N) {
        f32x4 v = acc[i][j];
        if (bias != nullptr && splitk == 1) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        }
        if (add != nullptr && splitk == 1) {
          const float4 d4 = *reinterpret_cast<const float4*>(add + (size_t)m * ldc + n);
          v.x += d4.x; v.y += d4.y; v.z += d4.z; v.w += d4.w;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)m * ldc + n) = v;
      }
    }
  }
}

// LDS bytes of the kernel
constexpr int gemm_lds_bytes(int nsplit) { return 2 * nsplit * (kGemmAPart + kGemmBPart); }

template <bool AK, bool BK_, int NS>
static int gemm_launch(const float* A, const float* B, const float* bias, const float* add, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       int splitk, hipStream_t st) {
  const int tiles_m = (M + kGemmBM - 1) / kGemmBM, tiles_n = (N + kGemmBN - 1) / kGemmBN;
  int kper = ((K + splitk - 1) / splitk + kGemmBK - 1) / kGemmBK * kGemmBK;
  const void* fn = (const void*)gemm_bf16_k<AK, BK_, NS>;
  int rc = (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(NS));
  if (rc) return rc;
  STOVE_LAUNCH((gemm_bf16_k<AK, BK_, NS>), dim3(tiles_m * tiles_n * splitk), dim3(kGemmThreads), gemm_lds_bytes(NS), st, A, B, bias, add, C,
               M, N, K, lda, ldb, ldc, tiles_m, tiles_n, splitk, kper);
  STOVE_LAUNCH_CHECK();
  return 0;
}

}  // namespace stove
