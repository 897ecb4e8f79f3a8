// Background SPN leaf layer on the matrix cores (scene mode), gfx950.
//
// The leaf log-densities of a frame are 36 masked sums over its 1024 pixels,
//     ell[f][(r, side, g)] = sum_{p : side_r(p) = side} w_p (a_{r,p,g} x_p^2 + b_{r,p,g} x_p + c_{r,p,g}),
// i.e. one GEMM   ell (F x 36) = Phi (F x 3072) . C (3072 x 36)   with the per-frame features
// Phi = [w x^2 | w x | w] and a fixed coefficient matrix C (zero where the pixel is on the other side of the
// replica's split).  The lane-per-pixel kernel of spn_bg.hip spends its time in 36 cross-lane reductions per
// frame (144 DPP adds per lane and frame); as a GEMM on v_mfma_f32_16x16x4_f32 (exact fp32) the reductions
// are the matrix cores' contraction and the kernel is bounded by 2 304 MFMAs per 16 frames.
// The occlusion weight w comes from the closed-form separable box coverage (cover_x(col) cover_y(row), see
// spn_bg.hip): the 64 coverage values per (frame, object) are tabulated once per tile in LDS, so a pixel's
// weight costs n_obj multiply-adds instead of 2 n_obj coverage evaluations.
#include "common.h"

namespace stove {

constexpr int kBgNC = 48;                                   // 36 leaf outputs padded to three 16-column tiles
constexpr int kBgDenseF = 3 * 3 * 64 * 4 * 16 * 4;          // forward image floats: [feat][tile][kb][kq][j][m]
typedef float bgf4 __attribute__((ext_vector_type(4)));

// Cf[feat][t][kb][kq][j][m] = C[(pixel 16 kb + 4 kq + m, feat)][col 16 t + j]: the B fragment of MFMA m of pixel
// block kb is one coalesced float4 per lane (lane = (j, kq)); the K order inside a block is the same permutation
// on both operands (slot kq of MFMA m <-> pixel 4 kq + m).
__global__ void bg_dense_fwd_k(const int* __restrict__ side, const float* __restrict__ coef, float* __restrict__ Cf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kBgDenseF) return;
  const int m = idx & 3, j = (idx >> 2) & 15, kq = (idx >> 6) & 3, kb = (idx >> 8) & 63, t = (idx >> 14) % 3, feat = idx / (3 << 14);
  const int p = 16 * kb + 4 * kq + m, col = 16 * t + j;
  float v = 0.0f;
  if (col < 36) {
    const int r = col / 12, sd = (col / 6) & 1, g = col % 6;
    if ((side[r * kBgPix + p] != 0) == (sd != 0)) v = coef[((size_t)(r * kBgPix + p) * 6 + g) * 3 + feat];
  }
  Cf[idx] = v;
}

// One wave = TPW tiles of 16 frames; per 16-pixel block: 9 B fragments (3 features x 3 column tiles) shared by the
// wave's tiles, 36 MFMAs per tile.  ell (F, 36).  Dynamic LDS: waves * TPW * 16 * n_obj * 64 floats of coverage tables.
// NOBJ > 0: compile-time object count (the per-object loop unrolls and the whole pixel block becomes one basic block,
// so the scheduler overlaps one tile's mask / feature VALU work with the other tile's MFMAs); NOBJ = 0: runtime n_obj.
template <int TPW, int NOBJ>
__global__ __launch_bounds__(256) void bgspn_mfma_fwd_k(const float* __restrict__ frames, const float* __restrict__ z, int n_obj_rt,
                                                       const float* __restrict__ Cf, float* __restrict__ ell, int F, FrameMap fm) {
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
#pragma unroll 4
  for (int idx = lane; idx < TPW * 16 * n_obj * 64; idx += 64) {      // unrolled: four independent coverage chains in flight
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
    fptr[tl] = frames + fm.row(f < F ? f : 0) * kBgPix + 4 * kq;
#pragma unroll
    for (int u = 0; u < GRP; ++u) xc[u][tl] = *reinterpret_cast<const float4*>(fptr[tl] + 16 * ((kofs + u) & 63));
  }
  // a lane only ever touches columns 4 kq .. 4 kq + 3 of the two 16-column halves of a row: its x-coverage values are
  // loop invariants (kept in registers when the object count is a compile-time constant)
  constexpr int NK = NOBJ > 0 ? NOBJ : 1;
  float4 cxr[TPW][2][NK];
  if (NOBJ > 0) {
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
      for (int hp = 0; hp < 2; ++hp)
#pragma unroll
        for (int k = 0; k < NK; ++k) cxr[tl][hp][k] = *reinterpret_cast<const float4*>(tab + (tl * 16 + i) * fstride + k * 64 + 16 * hp + 4 * kq);
  }
  for (int kg = 0; kg < 64; kg += GRP) {
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
        for (int q = 0; q < 9; ++q) bn[q] = Cq[(size_t)(q * 64 + ((kb + 1) & 63)) * 64];
      }
      const int row = kb >> 1, c0 = 16 * (kb & 1) + 4 * kq;
#pragma unroll
      for (int tl = 0; tl < TPW; ++tl) {
        const int fr = tl * 16 + i, f = f0 + fr;
        const float4 x = xc[u][tl];
        float4 run = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < n_obj; ++k) {
          const float* tk = tab + fr * fstride + k * 64;
          const float cy = tk[32 + row];
          const float4 cx = (NOBJ > 0) ? cxr[tl][u & 1][k < NK ? k : 0] : *reinterpret_cast<const float4*>(tk + c0);
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
            acc[tl][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxx, bq[0 * 3 + t][m], acc[tl][t], 0, 0, 0);
            acc[tl][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wx, bq[1 * 3 + t][m], acc[tl][t], 0, 0, 0);
            acc[tl][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm, bq[2 * 3 + t][m], acc[tl][t], 0, 0, 0);
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
