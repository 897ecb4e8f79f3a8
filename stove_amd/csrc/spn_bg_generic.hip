// Background RAT-SPN operator for ANY frame size (round 4).  The reference builds the background SPN over c x w x h dimensions
// (probabilistic_models.py:25-39: three one-level random binary splits, six Gaussians per leaf) and its stock gravity /
// multibilliards data are 50 x 50 (envs.py:771-773, 841-844); the tuned kernels of spn_bg.hip / spn_bg_mfma.hip are laid out for
// 32 x 32 = 1024 pixels (two 512-lane halves, 16-pixel MFMA blocks, 32-entry coverage tables).  These are the same RatSpn.forward /
// backward (rat_torch.py:83-109, 147-163, 202-222, 354-357) with the pixel count at run time: lane = pixel, ceil(n_pix / 512) "halves",
// coefficients of the lane's pixel in registers for the whole walk over the frames, lanes past the last pixel contribute zeros.
// Correctness first (the operator behind Supair.likelihood for frame sizes other than 32 x 32); the root / root-gradient
// kernels are those of spn_bg.hip, which take the number of halves as an argument.
#include "common.h"

namespace stove {

// ---- the background's marginalisation mask of a scene and its backward, any frame size (Supair.masks_from_z, supair.py:304-356) -----
// mask[f][Y][X] = min(1, sum_k cover_x,k(X) cover_y,k(Y)): a pasted unit box is separable and the sequential clamps collapse (scene.hip).
// cover(q) is the bilinear sample of a ones image with zero padding at pixel coordinate q = (X - cx) / sx_k + cx - sxa x_k / sx_k.
__device__ __forceinline__ float cover_n(float q, float n, float* dq) {
  const float a = q + 1.0f, b = n - q;
  const float m = fminf(a, b);
  *dq = (m > 0.0f && m < 1.0f) ? (a < b ? 1.0f : -1.0f) : 0.0f;
  return fminf(fmaxf(m, 0.0f), 1.0f);
}
// Scene mode of the two kernels below: the mask is not read from memory but formed per pixel from per-frame coverage TABLES -- a pasted box
// is separable, so a frame needs (W + H) coverage values per object instead of two evaluations per (pixel, object) -- which the
// workgroup fills into LDS one frame ahead (two buffers; the loop's barrier publishes them).  Frames up to kBgTabMax pixels a side.
constexpr int kBgTabMax = 128, kBgTabObj = 8;
struct SceneBoxes {
  const float* z;      // (n_frames * n_obj, 4) = [sx, sy, x, y], or null: mask from `marg`
  int n_obj;
  SceneGeom gm;
};
__device__ __forceinline__ void bg_fill_tables(const SceneBoxes& sb, int f, float (*tx)[kBgTabMax], float (*ty)[kBgTabMax],
                                               float (*dtx)[kBgTabMax], float (*dty)[kBgTabMax]) {
  const int W = sb.gm.W, H = sb.gm.H, per = W + H;
  for (int i = threadIdx.x; i < sb.n_obj * per; i += blockDim.x) {
    const int k = i / per, c = i - k * per;
    const float4 zk = *reinterpret_cast<const float4*>(sb.z + ((size_t)f * sb.n_obj + k) * 4);
    float d;
    if (c < W) {
      const float isx = 1.0f / zk.x;
      tx[k][c] = cover_n(fmaf(isx, (float)c - sb.gm.cx, fmaf(-sb.gm.sxa * zk.z, isx, sb.gm.cx)), (float)W, &d);
      if (dtx != nullptr) dtx[k][c] = d;
    } else {
      const float isy = 1.0f / zk.y;
      ty[k][c - W] = cover_n(fmaf(isy, (float)(c - W) - sb.gm.cy, fmaf(-sb.gm.sya * zk.w, isy, sb.gm.cy)), (float)H, &d);
      if (dty != nullptr) dty[k][c - W] = d;
    }
  }
}

template <int R, int G>
__global__ __launch_bounds__(kBgThreads) void bgspn_fwd_any_k(const float* __restrict__ inputs, const float* __restrict__ marg,
                                                              const int* __restrict__ side, const float* __restrict__ coef,
                                                              float* __restrict__ ell_part, int n_frames, int n_pix, int halves, FrameMap fm,
                                                              SceneBoxes sb) {
  constexpr int NO = R * 2 * G;
  constexpr int NW = kBgThreads / 64;
  __shared__ float part[2][NW * 4][NO];     // one partial per 16-lane row of every wave
  __shared__ float tbx[2][kBgTabObj][kBgTabMax], tby[2][kBgTabObj][kBgTabMax];
  const bool scene = sb.z != nullptr;
  const int half = blockIdx.x % halves;
  const int p = half * kBgThreads + threadIdx.x;
  const bool live = p < n_pix;
  const int pc = live ? p : n_pix - 1;
  const int lane = lane_id(), wv = wave_id();
  float cf[R][G][3];
  bool sd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    sd[r] = side[(size_t)r * n_pix + pc] != 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) cf[r][g][k] = coef[(((size_t)r * n_pix + pc) * G + g) * 3 + k];
  }
  int it = 0;
  const int fstep = gridDim.x / halves;
  const int X = scene ? pc % sb.gm.W : 0, Y = scene ? pc / sb.gm.W : 0;
  if (scene && (int)(blockIdx.x / halves) < n_frames) {
    bg_fill_tables(sb, blockIdx.x / halves, tbx[0], tby[0], nullptr, nullptr);
    __syncthreads();
  }
  for (int f = blockIdx.x / halves; f < n_frames; f += fstep, ++it) {
    const float x = inputs[fm.row(f) * n_pix + pc];
    float w = (marg != nullptr) ? 1.0f - fminf(fmaxf(marg[(size_t)f * n_pix + pc], 0.0f), 1.0f) : 1.0f;
    if (scene) {
      float run = 0.0f;
      for (int k = 0; k < sb.n_obj; ++k) run = fmaf(tbx[it & 1][k][X], tby[it & 1][k][Y], run);
      w = 1.0f - fminf(run, 1.0f);
      if (f + fstep < n_frames) bg_fill_tables(sb, f + fstep, tbx[(it + 1) & 1], tby[(it + 1) & 1], nullptr, nullptr);
    }
    if (!live) w = 0.0f;
    const float wx = w * x, wxx = wx * x;
    float* pp = part[it & 1][wv * 4 + (lane >> 4)];
    const bool row_last = (lane & 15) == 15;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float val = fmaf(wxx, cf[r][g][0], fmaf(wx, cf[r][g][1], w * cf[r][g][2]));
        const float v1 = sd[r] ? val : 0.0f;
        const float s0 = row_sum_lane15(val - v1);
        const float s1 = row_sum_lane15(v1);
        if (row_last) {
          pp[(r * 2) * G + g] = s0;
          pp[(r * 2 + 1) * G + g] = s1;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < NO) {
      float s = 0.0f;
#pragma unroll
      for (int q = 0; q < NW * 4; ++q) s += part[it & 1][q][threadIdx.x];
      ell_part[((size_t)f * halves + half) * NO + threadIdx.x] = s;
    }
  }
}

// dell[frame][(r*2+side)*G+g] = dL/d leaf (bgspn_root_bwd_k) -> d_inputs / d_marg [frame][pixel] (either may be null) and the
// per-block coefficient gradients gcoef_part[block][r][lane][g][3] (summed over the block's frames in frame order)
template <int R, int G>
__global__ __launch_bounds__(kBgThreads) void bgspn_bwd_any_k(const float* __restrict__ inputs, const float* __restrict__ marg,
                                                              const int* __restrict__ side, const float* __restrict__ coef,
                                                              const float* __restrict__ dell, float* __restrict__ d_inputs,
                                                              float* __restrict__ d_marg, float* __restrict__ gcoef_part, int n_frames,
                                                              int n_pix, int halves, FrameMap fm, SceneBoxes sb) {
  constexpr int NO = R * 2 * G;
  __shared__ float tbx[2][kBgTabObj][kBgTabMax], tby[2][kBgTabObj][kBgTabMax];
  const bool scene = sb.z != nullptr;
  const int half = blockIdx.x % halves;
  const int p = half * kBgThreads + threadIdx.x;
  const bool live = p < n_pix;
  const int pc = live ? p : n_pix - 1;
  float cf[R][G][3], gc[R][G][3];
  int doff[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    doff[r] = (r * 2 + (side[(size_t)r * n_pix + pc] != 0 ? 1 : 0)) * G;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        cf[r][g][k] = coef[(((size_t)r * n_pix + pc) * G + g) * 3 + k];
        gc[r][g][k] = 0.0f;
      }
  }
  const int fstep = gridDim.x / halves;
  const int X = scene ? pc % sb.gm.W : 0, Y = scene ? pc / sb.gm.W : 0;
  if (scene && (int)(blockIdx.x / halves) < n_frames) {
    bg_fill_tables(sb, blockIdx.x / halves, tbx[0], tby[0], nullptr, nullptr);
    __syncthreads();
  }
  int it = 0;
  for (int f = blockIdx.x / halves; f < n_frames; f += fstep, ++it) {
    const float x = inputs[fm.row(f) * n_pix + pc];
    float mraw = 0.0f, w = 1.0f;
    if (marg != nullptr) {
      mraw = marg[(size_t)f * n_pix + pc];
      w = 1.0f - fminf(fmaxf(mraw, 0.0f), 1.0f);
    }
    if (scene) {
      // mraw = the unclamped sum of the boxes: the gradient passes while it is <= 1 (min(1, .)); never negative
      for (int k = 0; k < sb.n_obj; ++k) mraw = fmaf(tbx[it & 1][k][X], tby[it & 1][k][Y], mraw);
      w = 1.0f - fminf(mraw, 1.0f);
      if (f + fstep < n_frames) bg_fill_tables(sb, f + fstep, tbx[(it + 1) & 1], tby[(it + 1) & 1], nullptr, nullptr);
    }
    if (!live) w = 0.0f;
    const float wx = w * x, wxx = wx * x, x2 = x * x;
    const float* dl = dell + (size_t)f * NO;
    float dwr[R], dx = 0.0f;            // one partial sum per replica (short dependent chains), replicas added in order
#pragma unroll
    for (int r = 0; r < R; ++r) {
      dwr[r] = 0.0f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float d = dl[doff[r] + g];
        dwr[r] = fmaf(d, fmaf(cf[r][g][0], x2, fmaf(cf[r][g][1], x, cf[r][g][2])), dwr[r]);
        dx = fmaf(d, fmaf(cf[r][g][0], x + x, cf[r][g][1]), dx);
        gc[r][g][0] = fmaf(d, wxx, gc[r][g][0]);
        gc[r][g][1] = fmaf(d, wx, gc[r][g][1]);
        gc[r][g][2] = fmaf(d, w, gc[r][g][2]);
      }
    }
    float dw = dwr[0];
#pragma unroll
    for (int r = 1; r < R; ++r) dw += dwr[r];
    if (live) {
      if (d_marg != nullptr) d_marg[(size_t)f * n_pix + p] = (mraw >= 0.0f && mraw <= 1.0f) ? -dw : 0.0f;      // boundaries pass, as ATen's clamp
      if (d_inputs != nullptr) d_inputs[(size_t)f * n_pix + p] = dx * w;
    }
    if (scene) __syncthreads();       // the next frame's tables are complete; this frame's may be overwritten
  }
  float* o = gcoef_part + ((size_t)blockIdx.x * R * kBgThreads) * G * 3;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) o[((size_t)(r * kBgThreads + threadIdx.x) * G + g) * 3 + k] = gc[r][g][k];
}

// g_coef[r][p][g][3] = sum over the blocks that own pixel-half(p) of gcoef_part, in block order
template <int R, int G>
__global__ __launch_bounds__(256) void bgspn_coef_reduce_any_k(const float* __restrict__ gcoef_part, float* __restrict__ g_coef, int n_blocks,
                                                                int n_pix, int halves) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
  const size_t j = (size_t)blockIdx.x * 32 + el;               // over R * n_pix * G * 3
  float s = 0.0f;
  const bool live = j < (size_t)R * n_pix * G * 3;
  if (live) {
    const int e = (int)(j % (G * 3));
    const int p = (int)((j / (G * 3)) % n_pix);
    const int r = (int)(j / ((size_t)G * 3 * n_pix));
    const int half = p / kBgThreads, pl = p % kBgThreads;
    for (int b = half + q * halves; b < n_blocks; b += 8 * halves)
      s += gcoef_part[(((size_t)b * R + r) * kBgThreads + pl) * G * 3 + e];
  }
  red[q][el] = s;
  __syncthreads();
  if (q == 0 && live) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][el];
    g_coef[j] = t;
  }
}

__global__ __launch_bounds__(256) void bg_mask_any_k(const float* __restrict__ z, float* __restrict__ mask, int n_frames, int n_obj, SceneGeom gm) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n_pix = gm.W * gm.H;
  if (t >= (size_t)n_frames * n_pix) return;
  const int f = (int)(t / n_pix), p = (int)(t % n_pix);
  const float X = (float)(p % gm.W) - gm.cx, Y = (float)(p / gm.W) - gm.cy;
  float run = 0.0f;
  for (int k = 0; k < n_obj; ++k) {
    const float4 zk = *reinterpret_cast<const float4*>(z + ((size_t)f * n_obj + k) * 4);
    const float isx = 1.0f / zk.x, isy = 1.0f / zk.y;
    float d;
    const float cxv = cover_n(fmaf(isx, X, fmaf(-gm.sxa * zk.z, isx, gm.cx)), (float)gm.W, &d);
    const float cyv = cover_n(fmaf(isy, Y, fmaf(-gm.sya * zk.w, isy, gm.cy)), (float)gm.H, &d);
    run += cxv * cyv;
  }
  mask[t] = fminf(run, 1.0f);
}
// d_mask[f][p] (= dL/d mask, from the background SPN's backward) -> dz_bg[f][k][4] = dL/d(sx, sy, x, y) of every pasted box.
// One workgroup per frame, thread = pixels p, p + 256, ...; the 4 n_obj sums are reduced over the workgroup in a fixed order.
template <int NMAX>
__global__ __launch_bounds__(256) void bg_mask_bwd_any_k(const float* __restrict__ z, const float* __restrict__ d_mask, float* __restrict__ dz_bg,
                                                         int n_frames, int n_obj, SceneGeom gm) {
  __shared__ float red[4][NMAX * 4];
  __shared__ float tx[kBgTabObj][kBgTabMax], ty[kBgTabObj][kBgTabMax], dtx[kBgTabObj][kBgTabMax], dty[kBgTabObj][kBgTabMax];
  const int f = blockIdx.x;
  const int n_pix = gm.W * gm.H;
  const bool tabs = gm.W <= kBgTabMax && gm.H <= kBgTabMax;      // workgroup-uniform: coverage from per-frame tables (see bg_fill_tables)
  if (tabs) {
    SceneBoxes sb;
    sb.z = z; sb.n_obj = n_obj; sb.gm = gm;
    bg_fill_tables(sb, f, tx, ty, dtx, dty);
    __syncthreads();
  }
  float isx[NMAX], isy[NMAX], ox[NMAX], oy[NMAX], zx[NMAX], zy[NMAX], acc[NMAX][4];
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    const float4 zk = *reinterpret_cast<const float4*>(z + ((size_t)f * n_obj + (k < n_obj ? k : 0)) * 4);
    isx[k] = 1.0f / zk.x; isy[k] = 1.0f / zk.y; zx[k] = zk.z; zy[k] = zk.w;
    ox[k] = fmaf(-gm.sxa * zk.z, isx[k], gm.cx);
    oy[k] = fmaf(-gm.sya * zk.w, isy[k], gm.cy);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[k][e] = 0.0f;
  }
  for (int p = threadIdx.x; p < n_pix; p += 256) {
    const int Xi = p % gm.W, Yi = p / gm.W;
    const float X = (float)Xi - gm.cx, Y = (float)Yi - gm.cy;
    const float u = fmaf(gm.fax, (float)Xi, gm.fbx), v = fmaf(gm.fay, (float)Yi, gm.fby);
    float cxv[NMAX], cyv[NMAX], dcx[NMAX], dcy[NMAX], run = 0.0f;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if (tabs) {
        const int kk = k < n_obj ? k : 0;
        cxv[k] = tx[kk][Xi]; dcx[k] = dtx[kk][Xi]; cyv[k] = ty[kk][Yi]; dcy[k] = dty[kk][Yi];
      } else {
        cxv[k] = cover_n(fmaf(isx[k], X, ox[k]), (float)gm.W, &dcx[k]);
        cyv[k] = cover_n(fmaf(isy[k], Y, oy[k]), (float)gm.H, &dcy[k]);
      }
      if (k < n_obj) run += cxv[k] * cyv[k];
    }
    const float d = run <= 1.0f ? d_mask[(size_t)f * n_pix + p] : 0.0f;        // min(1, .): the gradient passes while the sum is not clamped
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if (k < n_obj) {
        // q_x = (X - cx) / sx + cx - sxa x / sx:  dq/d(1/sx) = sxa (u - x),  dq/dx = -sxa / sx,  d(1/sx)/dsx = -1/sx^2
        const float gx = d * cyv[k] * dcx[k], gy = d * cxv[k] * dcy[k];
        acc[k][0] = fmaf(gx, -gm.sxa * (u - zx[k]) * isx[k] * isx[k], acc[k][0]);
        acc[k][1] = fmaf(gy, -gm.sya * (v - zy[k]) * isy[k] * isy[k], acc[k][1]);
        acc[k][2] = fmaf(gx, -gm.sxa * isx[k], acc[k][2]);
        acc[k][3] = fmaf(gy, -gm.sya * isy[k], acc[k][3]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NMAX; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = wave_sum(acc[k][e]);
      if (lane_id() == 0) red[wave_id()][k * 4 + e] = s;
    }
  __syncthreads();
  if (threadIdx.x < n_obj * 4) dz_bg[(size_t)f * n_obj * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static inline int bg_halves_any(int n_pix) { return (n_pix + kBgThreads - 1) / kBgThreads; }
static inline int bg_grid_any(int n_frames, int n_pix) {
  int g = n_frames < 128 ? n_frames : 128;
  if (g < 1) g = 1;
  return g * bg_halves_any(n_pix);
}
size_t bgspn_any_saved_floats(int n_frames, int n_pix) { return (size_t)n_frames * bg_halves_any(n_pix) * kBgNO; }
size_t bgspn_any_bwd_ws_floats(int n_frames, int n_pix) {
  return (size_t)n_frames * (kBgNO + kBgR * (1 + 2 * kBgG)) + (size_t)bg_grid_any(n_frames, n_pix) * kBgR * kBgThreads * kBgG * 3 +
         (size_t)kBgRootChunks * kBgR * kBgG * kBgG;
}

int bgspn_any_forward(const float* inputs, const float* marg, const int* side, const float* coef, const float* wroot, float* ell_part,
                      float* out, int n_frames, int n_pix, hipStream_t st, FrameMap fm = FrameMap{0, 0}, SceneBoxes sb = SceneBoxes{nullptr, 0, SceneGeom{}}) {
  if (n_frames == 0) return 0;
  if (n_pix < 1) return (int)hipErrorInvalidValue;
  const int halves = bg_halves_any(n_pix);
  STOVE_LAUNCH((bgspn_fwd_any_k<kBgR, kBgG>), dim3(bg_grid_any(n_frames, n_pix)), dim3(kBgThreads), 0, st, inputs, marg, side, coef, ell_part,
               n_frames, n_pix, halves, fm, sb);
  STOVE_LAUNCH_CHECK();
  STOVE_LAUNCH((bgspn_root_fwd_k<kBgR, kBgG>), dim3((n_frames + 255) / 256), dim3(256), 0, st, ell_part, wroot, out, n_frames, halves);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// g_coef [R][n_pix][G][3], g_wroot [R*G*G] overwritten
int bgspn_any_backward(const float* inputs, const float* marg, const int* side, const float* coef, const float* wroot, const float* ell_part,
                       const float* out, const float* dout, float* d_inputs, float* d_marg, float* g_coef, float* g_wroot, float* ws,
                       int n_frames, int n_pix, hipStream_t st, FrameMap fm = FrameMap{0, 0}, SceneBoxes sb = SceneBoxes{nullptr, 0, SceneGeom{}}) {
  if (n_pix < 1) return (int)hipErrorInvalidValue;
  if (n_frames == 0) {
    hipMemsetAsync(g_coef, 0, sizeof(float) * kBgR * (size_t)n_pix * kBgG * 3, st);
    hipMemsetAsync(g_wroot, 0, sizeof(float) * kBgR * kBgG * kBgG, st);
    return 0;
  }
  const int halves = bg_halves_any(n_pix), grid = bg_grid_any(n_frames, n_pix);
  float* dell = ws;
  float* rsc = dell + (size_t)n_frames * kBgNO;
  float* gpart = rsc + (size_t)n_frames * kBgR * (1 + 2 * kBgG);
  float* rpart = gpart + (size_t)grid * kBgR * kBgThreads * kBgG * 3;
  STOVE_LAUNCH((bgspn_root_bwd_k<kBgR, kBgG>), dim3((n_frames + 255) / 256), dim3(256), 0, st, ell_part, wroot, out, dout, dell, rsc, n_frames, halves);
  STOVE_LAUNCH_CHECK();
  STOVE_LAUNCH((bgspn_bwd_any_k<kBgR, kBgG>), dim3(grid), dim3(kBgThreads), 0, st, inputs, marg, side, coef, (const float*)dell, d_inputs, d_marg,
               gpart, n_frames, n_pix, halves, fm, sb);
  STOVE_LAUNCH_CHECK();
  const size_t nc = (size_t)kBgR * n_pix * kBgG * 3;
  STOVE_LAUNCH((bgspn_coef_reduce_any_k<kBgR, kBgG>), dim3((unsigned)((nc + 31) / 32)), dim3(256), 0, st, (const float*)gpart, g_coef, grid, n_pix, halves);
  STOVE_LAUNCH_CHECK();
  const int chunks = n_frames < kBgRootChunks ? n_frames : kBgRootChunks;
  STOVE_LAUNCH((bgspn_rootgrad_k<kBgR, kBgG>), dim3(chunks), dim3(128), 0, st, (const float*)rsc, rpart, n_frames, chunks);
  STOVE_LAUNCH_CHECK();
  STOVE_LAUNCH(reduce_chunks_k, dim3((kBgR * kBgG * kBgG + 31) / 32), dim3(256), 0, st, (const float*)rpart, g_wroot, kBgR * kBgG * kBgG, chunks, 0);
  STOVE_LAUNCH_CHECK();
  return 0;
}

}  // namespace stove
