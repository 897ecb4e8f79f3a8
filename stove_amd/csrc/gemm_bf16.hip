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
#include "split16.h"
#include <type_traits>

namespace stove {

constexpr int kGemmBK = 32;
// bytes of one bf16 image of a ROWS x 32 operand tile: K-major images are [32][ROWS + 16], K-contiguous ones [ROWS][32]
constexpr int gemm_part_bytes(int rows) { return kGemmBK * (rows + 16) * 2; }

// Where one thread's NLD float4 pieces of an operand tile (ROWS x 32) come from: a running pointer per piece, advanced by one
// k-step per tile, plus the piece's k offset inside the tile and whether its rows exist (edge tiles).
template <int ROWS, bool KMAJOR, int THREADS, bool SCALAR = false>
struct TileSrc {
  static constexpr int NLD = ROWS * kGemmBK / 4 / THREADS;
  // piece j of a thread is float4 number f = tid + THREADS j of the tile.  K-contiguous: row (f >> 3), k 4 (f & 7); K-major: k
  // f / (ROWS / 4), rows 4 (f % (ROWS / 4)).  THREADS is a multiple of 8 and of ROWS / 4, so the pieces of a thread are an
  // arithmetic sequence in memory: ONE running pointer and a stride instead of NLD pointers (registers: the 256 x 256 kernel).
  static_assert(THREADS % 8 == 0 && THREADS % (ROWS / 4) == 0, "pieces of a thread must be equally spaced");
  static constexpr int KJ = KMAJOR ? THREADS / (ROWS / 4) : 0;       // k offset between consecutive pieces
  const float* p0;
  const float* safe;
  size_t jstride;
  int kloc0;
  unsigned rowmask;
  unsigned rowsleft;     // scalar mode, K-major: 3 bits per piece = how many of its 4 rows exist
  // SCALAR: the operand is not float4-addressable (odd leading dimension / extent, e.g. fc1's 50 columns): every element is
  // loaded and predicated on its own (a compile-time variant, so that the vector path keeps its branch-free loads)
  size_t step;
  __device__ __forceinline__ int kloc(int j) const { return kloc0 + KJ * j; }
  __device__ __forceinline__ const float* p(int j) const { return p0 + jstride * j; }
  __device__ __forceinline__ void init(const float* __restrict__ P, int ld, int row0, int n_rows, int k_begin, int tid) {
    rowmask = 0u;
    rowsleft = 0u;
    safe = P;
    step = KMAJOR ? (size_t)kGemmBK * ld : (size_t)kGemmBK;
    jstride = KMAJOR ? (size_t)KJ * ld : (size_t)(THREADS / 8) * ld;
    if (KMAJOR) {
      kloc0 = tid / (ROWS / 4);
      p0 = P + (size_t)(k_begin + kloc0) * ld + row0 + 4 * (tid % (ROWS / 4));
    } else {
      kloc0 = 4 * (tid & 7);
      p0 = P + (size_t)(row0 + (tid >> 3)) * ld + k_begin + kloc0;
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + THREADS * j;
      const int row = KMAJOR ? row0 + 4 * (f % (ROWS / 4)) : row0 + (f >> 3);
      rowmask |= row < n_rows ? (1u << j) : 0u;
      const int left = n_rows - row;
      rowsleft |= (unsigned)(left < 0 ? 0 : (left > 4 ? 4 : left)) << (3 * j);
    }
  }
};

// one staged operand tile: NLD float4 per thread
template <int ROWS, bool KMAJOR, int THREADS, bool SCALAR = false>
struct TileLoad {
  static constexpr int NLD = ROWS * kGemmBK / 4 / THREADS;
  float4 v[NLD];
  unsigned okmask;      // bit j: piece j is inside the operand (the zeroing of edge pieces is deferred to store_piece: a
                        // select right behind the load would make the wave wait for it at once)
  // FULLT variants (interior workgroups: every row of the tile exists and K is a whole number of k-steps): no predicates, no
  // masks -- a third of the staging instructions of the masked form.  `advance`: whether another k-tile follows (the source
  // stays on the last tile otherwise: loads past the end re-read it, the image they are written to is never read).
  __device__ __forceinline__ void load_full(TileSrc<ROWS, KMAJOR, THREADS, SCALAR>& src, bool advance) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) v[j] = *reinterpret_cast<const float4*>(src.p(j));
    src.p0 += advance ? src.step : 0;
  }
  __device__ __forceinline__ void load_piece_full(int j, const TileSrc<ROWS, KMAJOR, THREADS, SCALAR>& src) {
    v[j] = *reinterpret_cast<const float4*>(src.p(j));
  }
  // loads the tile whose first k is `k0` and advances the source by one k-step
  __device__ __forceinline__ void load(TileSrc<ROWS, KMAJOR, THREADS, SCALAR>& src, int k0, int k_end) {
    okmask = 0u;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      // branch-free edge predication: out-of-range pieces read the operand's first float4 (always valid) and are zeroed later
      const bool ok = ((src.rowmask >> j) & 1u) && (k0 + src.kloc(j) < k_end);
      if constexpr (!SCALAR) {
        v[j] = *reinterpret_cast<const float4*>(ok ? src.p(j) : src.safe);
        okmask |= ok ? (1u << j) : 0u;
      } else {
        // element-wise: along k (K-contiguous: the K tail) or along the rows (K-major: the row tail)
        const int n_ok = !ok ? 0 : (KMAJOR ? (int)((src.rowsleft >> (3 * j)) & 7u) : (k_end - (k0 + src.kloc(j)) > 4 ? 4 : k_end - (k0 + src.kloc(j))));
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = q < n_ok ? src.p(j)[q] : 0.0f;
        v[j] = float4{e[0], e[1], e[2], e[3]};
        okmask |= 1u << j;
      }
    }
    src.p0 += src.step;
  }
  // one piece of the tile whose first k is `k0`; the caller advances the source (src.p0 += src.step) after the last piece
  __device__ __forceinline__ void load_piece(int j, const TileSrc<ROWS, KMAJOR, THREADS, SCALAR>& src, int k0, int k_end) {
    static_assert(!SCALAR, "vector path only");
    const bool ok = ((src.rowmask >> j) & 1u) && (k0 + src.kloc(j) < k_end);
    v[j] = *reinterpret_cast<const float4*>(ok ? src.p(j) : src.safe);
    okmask = (okmask & ~(1u << j)) | (ok ? (1u << j) : 0u);
  }
  // SH: the operand is multiplied by 2^SH before it is cut into pieces (exact; the epilogue takes it out again) -- see kGemmHalfShift
  template <int NSPLIT, bool FULLT = false, bool F16 = false, int SH = 0>
  __device__ __forceinline__ void store(char* hi_img, char* lo_img, int tid) const {
#pragma unroll
    for (int j = 0; j < NLD; ++j) store_piece<NSPLIT, FULLT, F16, SH>(j, hi_img, lo_img, tid);
  }
  template <int NSPLIT, bool FULLT = false, bool F16 = false, int SH = 0>
  __device__ __forceinline__ void store_piece(int j, char* hi_img, char* lo_img, int tid) const {
    {
      const int f = tid + THREADS * j;
      int off;
      if (KMAJOR) {
        off = (f / (ROWS / 4)) * ((ROWS + 16) * 2) + (f % (ROWS / 4)) * 8;
      } else {
        const int row = f >> 3, kq = f & 7;
        off = row * 64 + ((((kq & 3) << 4) | ((kq >> 2) << 3)) ^ (((row >> 3) & 1) << 5));
      }
      u32x2 hi, lo;
      float4 x = (FULLT || ((okmask >> j) & 1u)) ? v[j] : float4{0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (SH != 0) {
        constexpr float s = (float)(1 << SH);
        x.x *= s; x.y *= s; x.z *= s; x.w *= s;
      }
      split4<NSPLIT, F16>(x, hi, lo);
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

// DEBUG (tools only): 1 = no loads in the loop, 2 = no MFMAs.  A_SCALAR: A is read element-wise (see TileSrc); bit 2 of `scalar_bits`
// (run time): C / bias / add are written and read element-wise.
// WTM x WTN: the wave tile.  64 x 64: eight waves on 256 x 128 or four on 128 x 128, 16 accumulator tiles per wave.  Larger
// wave tiles (256 x 256 workgroup tile: 64 x 128 with eight waves, 128 x 128 with four): every fragment read from LDS feeds more
// MFMAs and a workgroup moves 2/3 of the operand bytes per flop through L2.
// (Measured and removed, round 4: the two waves of a SIMD half a k-step apart -- one in its MFMAs while the other stages, a barrier
// between the halves -- 169 vs 164 us warm, 178 vs 186 cold on x W_ih^T: level.  DESIGN.md section 7.)
template <bool A_KMAJOR, bool B_KMAJOR, int NSPLIT, int BM, int BN, int DEBUG = 0, bool A_SCALAR = false, int WTM = 64, int WTN = 64, bool F16 = false>
__global__ __launch_bounds__(BM * BN / (WTM * WTN) * 64) void gemm_bf16_k(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                               const float* __restrict__ add, float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                               int tiles_m, int tiles_n, int splitk, int k_per_slice, int scalar_bits) {
  extern __shared__ __attribute__((aligned(16))) char gemm_lds[];
  constexpr int kGemmBM = BM, kGemmBN = BN, THREADS = BM * BN / (WTM * WTN) * 64;      // one wave per WTM x WTN of the tile
  constexpr int TM = WTM / 16, TN = WTN / 16;                                            // MFMA tiles along the sides of the wave tile
  constexpr bool SQ64 = WTM == 64 && WTN == 64;
  // Half pieces: B (the weights of the forward products) is cut as 2^8 B.  The lo piece of x is below 2^-11 |x| and lies in half's
  // subnormals -- absolute error 2^-25 instead of a relative 2^-22 -- once |x| < 1/4, so a weight matrix of scale 0.003 would fall
  // back to bf16-piece accuracy (measured 6e-6 vs 4e-7 of the largest entry); shifted, every |w| >= 2^-10 keeps its 22 bits and
  // |w| < 256 stays inside half's range.  A (frames in [0, 1], hidden states in (-1, 1)) is taken as it is.
  constexpr int BSH = F16 ? kGemmHalfShift : 0;
  constexpr int kGemmAPart = gemm_part_bytes(BM), kGemmBPart = gemm_part_bytes(BN);
  constexpr int STAGE = NSPLIT * (kGemmAPart + kGemmBPart);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / (BN / WTN), wn = wv % (BN / WTN);
  // XCD-aware tile id
  const int n_wg = tiles_m * tiles_n * splitk;
  int id = blockIdx.x;
  if ((n_wg & 7) == 0) id = (id & 7) * (n_wg >> 3) + (id >> 3);
  const int tn = id % tiles_n, tm = (id / tiles_n) % tiles_m, z = id / (tiles_n * tiles_m);
  const int m0 = tm * kGemmBM, n0 = tn * kGemmBN;
  const int k_begin = z * k_per_slice;
  const int k_end = k_begin + k_per_slice < K ? k_begin + k_per_slice : K;
  const int nt = (k_end - k_begin + kGemmBK - 1) / kGemmBK;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  auto img = [&](int stage, int which) -> char* {      // which: 0 A_hi, 1 A_lo, 2 B_hi, 3 B_lo
    char* s = gemm_lds + stage * STAGE;
    if (NSPLIT == 2) return s + (which < 2 ? which * kGemmAPart : 2 * kGemmAPart + (which - 2) * kGemmBPart);
    return s + (which < 2 ? 0 : kGemmAPart);
  };
  // Two register sets of staged fp32 pieces: tile t+1 sits in one while tile t+2 is being loaded into the other.  The loads of
  // tile t+2 are issued at the top of step t and consumed in step t+1 (a whole k-step of MFMAs later); the pieces of tile t+1
  // are converted and written into the other LDS buffer BETWEEN the MFMA groups of step t, so the matrix pipe never waits for
  // the staging of the same wave.  One barrier per k-step.
  TileLoad<kGemmBM, A_KMAJOR, THREADS, A_SCALAR> la0, la1;
  TileLoad<kGemmBN, B_KMAJOR, THREADS> lb0, lb1;
  TileSrc<kGemmBM, A_KMAJOR, THREADS, A_SCALAR> sa;
  TileSrc<kGemmBN, B_KMAJOR, THREADS> sb;
  sa.init(A, lda, m0, M, k_begin, tid);
  sb.init(B, ldb, n0, N, k_begin, tid);
  // Interior workgroups (the whole tile inside the matrices, K a whole number of k-steps, vector path) run the loop without the
  // edge predicates; the choice is uniform over the workgroup.
  const bool full_tile = !A_SCALAR && m0 + kGemmBM <= M && n0 + kGemmBN <= N && (k_end - k_begin) % kGemmBK == 0 && nt > 0;
  auto run = [&](auto full_tag) {
    constexpr bool F = decltype(full_tag)::value;
    if (nt > 0) {
      if constexpr (F) {
        la0.load_full(sa, nt > 1);
        lb0.load_full(sb, nt > 1);
        if constexpr (SQ64) {
          la1.load_full(sa, nt > 2);
          lb1.load_full(sb, nt > 2);
        }
      } else {
        la0.load(sa, k_begin, k_end);
        lb0.load(sb, k_begin, k_end);
        if constexpr (SQ64) {
          la1.load(sa, k_begin + kGemmBK, k_end);
          lb1.load(sb, k_begin + kGemmBK, k_end);
        }
      }
      la0.template store<NSPLIT, F, F16>(img(0, 0), img(0, 1), tid);
      lb0.template store<NSPLIT, F, F16, BSH>(img(0, 2), img(0, 3), tid);
      if constexpr (!SQ64) {          // one register set: tile 1 follows tile 0 through it
        if constexpr (F) {
          la0.load_full(sa, nt > 2);
          lb0.load_full(sb, nt > 2);
        } else {
          la0.load(sa, k_begin + kGemmBK, k_end);
          lb0.load(sb, k_begin + kGemmBK, k_end);
        }
      }
    }
    __syncthreads();
    auto kstep = [&](int t, auto& a_next, auto& b_next, auto& a_far, auto& b_far) {
      const int cur = t & 1;
      // no branches in here: past the last tile the loads are predicated off (they read the operand's first float4) and the
      // stores write a zero tile nobody reads.  With a branch around either, hipcc has to assume the worse path and waits for the
      // loads it has just issued (s_waitcnt vmcnt counts in issue order).
      if (DEBUG != 1 && SQ64) {
        if constexpr (F) {
          a_far.load_full(sa, t + 3 < nt);
          b_far.load_full(sb, t + 3 < nt);
        } else {
          a_far.load(sa, k_begin + (t + 2) * kGemmBK, k_end);
          b_far.load(sb, k_begin + (t + 2) * kGemmBK, k_end);
        }
      }
      if constexpr (SQ64) {
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
        for (int i = 0; i < 4; ++i) {
    #pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (DEBUG == 2) {
              acc[i][j].x += (float)ah[i][0] * (float)bh[j][0] + (NSPLIT == 2 ? (float)al[i][0] * (float)bl[j][0] : 0.0f);
              continue;
            }
            if (NSPLIT == 2) {
              acc[i][j] = mfma16<F16>(bl[j], ah[i], acc[i][j]);
              acc[i][j] = mfma16<F16>(bh[j], al[i], acc[i][j]);
            }
            acc[i][j] = mfma16<F16>(bh[j], ah[i], acc[i][j]);
          }
          // a quarter of the next tile's staging behind every quarter of the MFMAs
          constexpr int NA = decltype(la0)::NLD, NB = decltype(lb0)::NLD;
    #pragma unroll
          for (int q = 0; q < NA; ++q)
            if (q % 4 == i) a_next.template store_piece<NSPLIT, F, F16>(q, img(cur ^ 1, 0), img(cur ^ 1, 1), tid);
    #pragma unroll
          for (int q = 0; q < NB; ++q)
            if ((q + 2) % 4 == i) b_next.template store_piece<NSPLIT, F, F16, BSH>(q, img(cur ^ 1, 2), img(cur ^ 1, 3), tid);
        }
      } else {
        // Large wave tile: the A fragments of the step stay in registers, the B fragments come one column tile at a time; the MFMAs
        // of a column tile are issued term by term (b_lo a_hi over all rows, then b_hi a_lo, then b_hi a_hi), so that consecutive
        // MFMAs never share an accumulator.  One register set of staged fp32 pieces: behind the first TN / 2 column tiles the
        // pieces of k-tile t + 1 (requested half a step ago) are converted and written to the other LDS buffer, behind the last
        // TN / 2 the pieces of k-tile t + 2 are requested into the registers just freed.
        constexpr int NA = decltype(la0)::NLD, NB = decltype(lb0)::NLD, H = TN / 2;
        static_assert(NA % H == 0 && NB % H == 0, "pieces per half step");
        const int k_far = k_begin + (t + 2) * kGemmBK;
        bf16x8 ah[TM], al[TM];
  #pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = read_frag<kGemmBM, A_KMAJOR>(img(cur, 0), wm * WTM + i * 16, lane);
          if (NSPLIT == 2) al[i] = read_frag<kGemmBM, A_KMAJOR>(img(cur, 1), wm * WTM + i * 16, lane);
        }
  #pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bf16x8 bh = read_frag<kGemmBN, B_KMAJOR>(img(cur, 2), wn * WTN + j * 16, lane);
          bf16x8 bl;
          if (NSPLIT == 2) bl = read_frag<kGemmBN, B_KMAJOR>(img(cur, 3), wn * WTN + j * 16, lane);
          if (DEBUG == 2) {
  #pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j].x += (float)ah[i][0] * (float)bh[0] + (NSPLIT == 2 ? (float)al[i][0] * (float)bl[0] : 0.0f);
          } else {
            if (NSPLIT == 2) {
  #pragma unroll
              for (int i = 0; i < TM; ++i) acc[i][j] = mfma16<F16>(bl, ah[i], acc[i][j]);
  #pragma unroll
              for (int i = 0; i < TM; ++i) acc[i][j] = mfma16<F16>(bh, al[i], acc[i][j]);
            }
  #pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = mfma16<F16>(bh, ah[i], acc[i][j]);
          }
          if (j < H && DEBUG == 3) {
            // (measurement variant: the staged registers are only kept alive, not converted or written)
  #pragma unroll
            for (int q = 0; q < NA / H; ++q) asm volatile("" ::"v"(a_next.v[j * (NA / H) + q].x));
  #pragma unroll
            for (int q = 0; q < NB / H; ++q) asm volatile("" ::"v"(b_next.v[j * (NB / H) + q].x));
          } else if (j < H) {
  #pragma unroll
            for (int q = 0; q < NA / H; ++q) a_next.template store_piece<NSPLIT, F, F16>(j * (NA / H) + q, img(cur ^ 1, 0), img(cur ^ 1, 1), tid);
  #pragma unroll
            for (int q = 0; q < NB / H; ++q) b_next.template store_piece<NSPLIT, F, F16, BSH>(j * (NB / H) + q, img(cur ^ 1, 2), img(cur ^ 1, 3), tid);
          } else if (DEBUG != 1) {
  #pragma unroll
            for (int q = 0; q < NA / H; ++q) {
              if constexpr (F) a_next.load_piece_full((j - H) * (NA / H) + q, sa);
              else a_next.load_piece((j - H) * (NA / H) + q, sa, k_far, k_end);
            }
  #pragma unroll
            for (int q = 0; q < NB / H; ++q) {
              if constexpr (F) b_next.load_piece_full((j - H) * (NB / H) + q, sb);
              else b_next.load_piece((j - H) * (NB / H) + q, sb, k_far, k_end);
            }
          }
        }
        // (full tiles: the source stays on the last k-tile once it is reached -- at this point it stands on tile t + 2)
        const bool adv = !F || t + 3 < nt;
        sa.p0 += adv ? sa.step : 0;
        sb.p0 += adv ? sb.step : 0;
      }
      __syncthreads();
    };
    if constexpr (SQ64) {
      for (int t = 0; t < nt; t += 2) {
        kstep(t, la1, lb1, la0, lb0);
        if (t + 1 < nt) kstep(t + 1, la0, lb0, la1, lb1);
      }
    } else {
      for (int t = 0; t < nt; ++t) kstep(t, la0, lb0, la0, lb0);
    }
  };
  if (full_tile) run(std::integral_constant<bool, true>{});
  else run(std::integral_constant<bool, false>{});
  // epilogue: lane holds C[m][n .. n+3], m = tile row (lane & 15), n = 4 (lane >> 4) + reg
  float* out = C + (splitk > 1 ? (size_t)z * M * ldc : 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 16 + 4 * (lane >> 4);
      if (m < M && n < N) {
        f32x4 v = acc[i][j];
        if constexpr (BSH != 0) v *= 1.0f / (float)(1 << BSH);
        if (!(scalar_bits & 4)) {
          if (bias != nullptr && splitk == 1) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          }
          if (add != nullptr && splitk == 1) {
            const float4 d4 = *reinterpret_cast<const float4*>(add + (size_t)m * ldc + n);
            v.x += d4.x; v.y += d4.y; v.z += d4.z; v.w += d4.w;
          }
          *reinterpret_cast<f32x4*>(out + (size_t)m * ldc + n) = v;
        } else {        // N or ldc not a multiple of 4: element-wise epilogue
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e < N) {
              float t = v[e];
              if (bias != nullptr && splitk == 1) t += bias[n + e];
              if (add != nullptr && splitk == 1) t += add[(size_t)m * ldc + n + e];
              out[(size_t)m * ldc + n + e] = t;
            }
          }
        }
      }
    }
  }
}

// LDS bytes of the kernel
constexpr int gemm_lds_bytes(int nsplit, int bm, int bn) { return 2 * nsplit * (gemm_part_bytes(bm) + gemm_part_bytes(bn)); }

template <bool AK, bool BK_, int NS, int BM = 256, int BN = 128, int DEBUG = 0, bool A_SCALAR = false, int WTM = 64, int WTN = 64, bool F16 = false>
static int gemm_launch(const float* A, const float* B, const float* bias, const float* add, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       int splitk, hipStream_t st, int scalar_bits = 0) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  int kper = ((K + splitk - 1) / splitk + kGemmBK - 1) / kGemmBK * kGemmBK;
  const void* fn = (const void*)gemm_bf16_k<AK, BK_, NS, BM, BN, DEBUG, A_SCALAR, WTM, WTN, F16>;
  int rc = (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(NS, BM, BN));
  if (rc) return rc;
  STOVE_LAUNCH((gemm_bf16_k<AK, BK_, NS, BM, BN, DEBUG, A_SCALAR, WTM, WTN, F16>), dim3(tiles_m * tiles_n * splitk), dim3(BM * BN / (WTM * WTN) * 64), gemm_lds_bytes(NS, BM, BN), st, A, B, bias, add, C,
               M, N, K, lda, ldb, ldc, tiles_m, tiles_n, splitk, kper, scalar_bits);
  STOVE_LAUNCH_CHECK();
  return 0;
}

}  // namespace stove
