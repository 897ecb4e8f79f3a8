// Background RAT-SPN (depth-1 random region graph over the 32x32 frame) -- gfx950 kernels.
//
// Replaces RatSpn.forward for the background SPN (model/spn/rat_torch.py:83-109, 147-163,
// 202-222) as called from Supair.likelihood (model/video_prediction/supair.py:61-67), and, in
// "scene" mode, the background half of Supair.masks_from_z (supair.py:304-344): the pasted
// unit boxes are separable (coverage_x * coverage_y of the inverse affine grid), so the final
// mask is a closed form of z and never touches memory.
//
// Mapping: one lane = one frame pixel (coalesced frame reads), the pixel's R*G leaf
// coefficients live in registers for the whole persistent loop over frames; per-frame leaf
// sums are reduced with DPP wave reductions + a small LDS stage.  A frame is split over
// 1024/512 = 2 workgroups ("halves"); the tiny root kernels add the two partial leaf vectors.
#include "common.h"

namespace stove {

constexpr int kBgPix = 1024;   // 32 x 32 x 1
constexpr int kBgSide = 32;
constexpr int kBgThreads = 512;
constexpr int kBgHalves = kBgPix / kBgThreads;

struct BoxGeom {
  float inv_sx, inv_sy, off_x, off_y;   // invert_z, supair.py:233-237
};
__device__ __forceinline__ BoxGeom box_geom(const float* z) {
  BoxGeom g;
  g.inv_sx = 1.0f / z[0];
  g.inv_sy = 1.0f / z[1];
  g.off_x = -z[2] / z[0];
  g.off_y = -z[3] / z[1];
  return g;
}
// pixel-space coordinate of frame column/row `idx` under the inverse transform
__device__ __forceinline__ float inv_coord(float inv_s, float off, int idx) {
  const float u = (2.0f * idx + 1.0f) * (1.0f / kBgSide) - 1.0f;
  const float gq = fmaf(inv_s, u, off);
  return ((gq + 1.0f) * kBgSide - 1.0f) * 0.5f;
}

// ---- forward: partial leaf log-densities of every frame ------------------------------------
// ell_part[frame][half][(r*2+side)*G+g]
// SCENE: w = 1 - min(1, sum_k box_k) from z[frame][N][4];  else w = 1 - clamp(marg) (or 1).
template <int R, int G, bool SCENE>
__global__ __launch_bounds__(kBgThreads) void bgspn_fwd_k(
    const float* __restrict__ frames, const float* __restrict__ marg, const float* __restrict__ z, int n_obj,
    const int* __restrict__ side, const float* __restrict__ coef, float* __restrict__ ell_part, int n_frames, FrameMap fm) {
  constexpr int NO = R * 2 * G;
  constexpr int NW = kBgThreads / 64;
  __shared__ float part[2][NW * 4][NO];     // one partial per 16-lane row of every wave
  const int half = blockIdx.x % kBgHalves;
  const int p = half * kBgThreads + threadIdx.x;
  const int lane = lane_id(), wv = wave_id();
  float cf[R][G][3];
  bool sd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    sd[r] = side[r * kBgPix + p] != 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) cf[r][g][k] = coef[((size_t)(r * kBgPix + p) * G + g) * 3 + k];
  }
  const int col = p % kBgSide, row = p / kBgSide;
  int it = 0;
  for (int f = blockIdx.x / kBgHalves; f < n_frames; f += gridDim.x / kBgHalves, ++it) {
    const float x = frames[fm.row(f) * kBgPix + p];
    float w;
    if (SCENE) {
      float run = 0.0f;
      for (int k = 0; k < n_obj; ++k) {
        const BoxGeom bg = box_geom(z + ((size_t)f * n_obj + k) * 4);
        float dq;
        const float box = cover(inv_coord(bg.inv_sx, bg.off_x, col), kBgSide, &dq) *
                          cover(inv_coord(bg.inv_sy, bg.off_y, row), kBgSide, &dq);
        run = fminf(run + box, 1.0f);
      }
      w = 1.0f - run;
    } else {
      w = (marg != nullptr) ? 1.0f - fminf(fmaxf(marg[(size_t)f * kBgPix + p], 0.0f), 1.0f) : 1.0f;
    }
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
      ell_part[((size_t)f * kBgHalves + half) * NO + threadIdx.x] = s;
    }
  }
}

// ---- root: product (G x G per replica) + root sum over R*G*G, one thread per frame ----------
template <int R, int G>
__global__ void bgspn_root_fwd_k(const float* __restrict__ ell_part, const float* __restrict__ wroot,
                                 float* __restrict__ out, int n_frames, int halves) {
  constexpr int NO = R * 2 * G;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  float M[R], Sr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float e1[G], e2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float a = 0.0f, b = 0.0f;
      for (int h = 0; h < halves; ++h) {
        a += ell_part[((size_t)f * halves + h) * NO + (r * 2) * G + g];
        b += ell_part[((size_t)f * halves + h) * NO + (r * 2 + 1) * G + g];
      }
      e1[g] = a;
      e2[g] = b;
    }
    float m1 = e1[0], m2 = e2[0];
#pragma unroll
    for (int g = 1; g < G; ++g) {
      m1 = fmaxf(m1, e1[g]);
      m2 = fmaxf(m2, e2[g]);
    }
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      e1[g] = __expf(e1[g] - m1);
      e2[g] = __expf(e2[g] - m2);
    }
#pragma unroll
    for (int j2 = 0; j2 < G; ++j2)
#pragma unroll
      for (int j1 = 0; j1 < G; ++j1) s = fmaf(e1[j1] * e2[j2], wroot[r * G * G + j2 * G + j1], s);
    M[r] = m1 + m2;
    Sr[r] = s;
  }
  float mm = M[0];
#pragma unroll
  for (int r = 1; r < R; ++r) mm = fmaxf(mm, M[r]);
  float Z = 0.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) Z = fmaf(Sr[r], __expf(M[r] - mm), Z);
  out[f] = mm + __logf(Z);
}

// dell[frame][(r*2+side)*G+g] = dL/d leaf ; rsc[frame][r][1+2G] = rho_r, E1[G], E2[G]
template <int R, int G>
__global__ void bgspn_root_bwd_k(const float* __restrict__ ell_part, const float* __restrict__ wroot,
                                 const float* __restrict__ out, const float* __restrict__ dout,
                                 float* __restrict__ dell, float* __restrict__ rsc, int n_frames, int halves) {
  constexpr int NO = R * 2 * G;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  const float go = dout[f], ro = out[f];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float e1[G], e2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float a = 0.0f, b = 0.0f;
      for (int h = 0; h < halves; ++h) {
        a += ell_part[((size_t)f * halves + h) * NO + (r * 2) * G + g];
        b += ell_part[((size_t)f * halves + h) * NO + (r * 2 + 1) * G + g];
      }
      e1[g] = a;
      e2[g] = b;
    }
    float m1 = e1[0], m2 = e2[0];
#pragma unroll
    for (int g = 1; g < G; ++g) {
      m1 = fmaxf(m1, e1[g]);
      m2 = fmaxf(m2, e2[g]);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      e1[g] = __expf(e1[g] - m1);
      e2[g] = __expf(e2[g] - m2);
    }
    const float rho = go * __expf(m1 + m2 - ro);
    float d1[G], d2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) d1[g] = d2[g] = 0.0f;
#pragma unroll
    for (int j2 = 0; j2 < G; ++j2)
#pragma unroll
      for (int j1 = 0; j1 < G; ++j1) {
        const float wk = wroot[r * G * G + j2 * G + j1];
        d1[j1] = fmaf(e2[j2], wk, d1[j1]);
        d2[j2] = fmaf(e1[j1], wk, d2[j2]);
      }
    float* rp = rsc + ((size_t)f * R + r) * (1 + 2 * G);
    rp[0] = rho;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      dell[(size_t)f * NO + (r * 2) * G + g] = rho * e1[g] * d1[g];
      dell[(size_t)f * NO + (r * 2 + 1) * G + g] = rho * e2[g] * d2[g];
      rp[1 + g] = e1[g];
      rp[1 + G + g] = e2[g];
    }
  }
}

// root weight grads (linear domain): part[c][r*G*G + j2*G + j1] = sum_frames rho_r E1[j1] E2[j2]
template <int R, int G>
__global__ void bgspn_rootgrad_k(const float* __restrict__ rsc, float* __restrict__ part, int n_frames, int n_chunks) {
  const int k = threadIdx.x;
  if (k >= R * G * G) return;
  const int c = blockIdx.x;
  const int r = k / (G * G), j2 = (k / G) % G, j1 = k % G;
  float acc = 0.0f;
#pragma unroll 4
  for (int f = c; f < n_frames; f += n_chunks) {
    const float* rp = rsc + ((size_t)f * R + r) * (1 + 2 * G);
    acc = fmaf(rp[0] * rp[1 + j1], rp[1 + G + j2], acc);
  }
  part[(size_t)c * R * G * G + k] = acc;
}

// ---- per (frame, object) box coverage tables for the scene backward -----------------------------------------
// T[(f * n_obj + k) * kBgTab + ..] = [cover_x(col) 32 | d cover_x 32 | cover_y(row) 32 | d cover_y 32 | 1/sx, 1/sy, x, y]
// Every pixel lane of bgspn_bwd_k needs these for all objects of every frame; evaluated in place they cost four IEEE
// divisions and two coverage evaluations per (pixel, frame, object) -- 40 % of that kernel's instructions.
constexpr int kBgTab = 132;
__global__ void bg_cover_tables_k(const float* __restrict__ z, float* __restrict__ T, int n_pairs) {
  const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= n_pairs) return;
  const int c = threadIdx.x & 63;
  const float* zk = z + (size_t)q * 4;
  const BoxGeom bg = box_geom(zk);
  float dq;
  const float v = (c < 32) ? cover(inv_coord(bg.inv_sx, bg.off_x, c), kBgSide, &dq) : cover(inv_coord(bg.inv_sy, bg.off_y, c - 32), kBgSide, &dq);
  float* t = T + (size_t)q * kBgTab;
  const int o = (c < 32) ? c : 64 + (c - 32);
  t[o] = v;
  t[o + 32] = dq;
  if (c < 4) t[128 + c] = (c == 0) ? bg.inv_sx : (c == 1) ? bg.inv_sy : zk[c];
}

// ---- backward main: per-pixel dL/dw (-> marg or z) and leaf coefficient grads ----------------
// SCENE: dz_part[frame][half][n_obj][4] (dsx, dsy, dx, dy of the pasted boxes)
// else : d_marg[frame][p] (and d_inputs if non-null)
// gcoef_part[block][r][p_local][g][3]
template <int R, int G, bool SCENE, int NMAX, bool EXACT = false>
__global__ __launch_bounds__(kBgThreads) void bgspn_bwd_k(
    const float* __restrict__ frames, const float* __restrict__ marg, const float* __restrict__ z, int n_obj,
    const int* __restrict__ side, const float* __restrict__ coef, const float* __restrict__ dell,
    float* __restrict__ d_inputs, float* __restrict__ d_marg, float* __restrict__ dz_part,
    float* __restrict__ gcoef_part, int n_frames, const float* __restrict__ T, FrameMap fm) {
  if (EXACT) n_obj = NMAX;      // the object count at compile time: the per-object predicates around the table loads go, the loads batch
  constexpr int NO = R * 2 * G;
  constexpr int NW = kBgThreads / 64;
  // SCENE: d box (= -dL/dw where no clamp fired) of the last NW frames, [slot][pixel of this half]; every NW frames the
  // block turns them into dz, one wave per frame (see flush below)
  __shared__ __attribute__((aligned(16))) float dbx[SCENE ? NW : 1][SCENE ? kBgThreads : 1];
  const int half = blockIdx.x % kBgHalves;
  const int p = half * kBgThreads + threadIdx.x;
  const int lane = lane_id(), wv = wave_id();
  float cf[R][G][3], gc[R][G][3];
  bool sd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    sd[r] = side[r * kBgPix + p] != 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        cf[r][g][k] = coef[((size_t)(r * kBgPix + p) * G + g) * 3 + k];
        gc[r][g][k] = 0.0f;
      }
  }
  const int col = p % kBgSide, row = p / kBgSide;
  int it = 0;
  // the pixel and the coverage-table entries of the NEXT frame are fetched while the current one is processed: with
  // ~100 frames per block and a dependent global load at the top of every iteration, the load latency was the kernel
  const int fstep = gridDim.x / kBgHalves;
  const int f0 = blockIdx.x / kBgHalves;
  float xn = 0.0f, fxn[NMAX], fyn[NMAX];
  // this lane's leaf gradients of the next frame (per replica the side its pixel belongs to): R * G vector loads that were
  // issued at the top of the iteration that uses them -- the pass over the frame waited for them
  float den[R][G];
  const float* de_lane[R];
#pragma unroll
  for (int r = 0; r < R; ++r) de_lane[r] = dell + (sd[r] ? (r * 2 + 1) * G : (r * 2) * G);
  auto prefetch = [&](int f) {
    xn = frames[fm.row(f) * kBgPix + p];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int g = 0; g < G; ++g) den[r][g] = de_lane[r][(size_t)f * NO + g];
    if (SCENE) {
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if (k < n_obj) {
          const float* tk = T + ((size_t)f * n_obj + k) * kBgTab;
          fxn[k] = tk[col];
          fyn[k] = tk[64 + row];
        }
      }
    }
  };
  // dz of the buffered frames.  The d box image of a frame enters the four gradients of object k only through three sums
  // per image row,  Sx = sum_c d dcover_x(c),  Sxu = sum_c d dcover_x(c) u(c),  Sy = sum_c d cover_x(c)  (the row factors
  // cover_y, dcover_y, v come out of the column sum), so one wave takes one frame: lane = (row, quarter of the columns),
  // 8 columns each, then ONE wave reduction per gradient -- instead of twelve 16-lane reductions, an LDS stage and a
  // workgroup barrier per frame in the pixel-parallel layout above.
  auto flush = [&](int n_buf, int f_first) {
    __syncthreads();
    if (wv < n_buf) {
      const int f = f_first + wv * fstep;
      const int r16 = lane >> 2, q = lane & 3;
      const int rw = half * (kBgThreads / kBgSide) + r16;
      const float4 da = *reinterpret_cast<const float4*>(&dbx[wv][r16 * kBgSide + q * 8]);
      const float4 db = *reinterpret_cast<const float4*>(&dbx[wv][r16 * kBgSide + q * 8 + 4]);
      const float d[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
      const float vrow = (2.0f * rw + 1.0f) * (1.0f / kBgSide) - 1.0f;
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if (k < n_obj) {
          const float* tk = T + ((size_t)f * n_obj + k) * kBgTab;
          const float4 ca = *reinterpret_cast<const float4*>(tk + q * 8), cb = *reinterpret_cast<const float4*>(tk + q * 8 + 4);
          const float4 ga = *reinterpret_cast<const float4*>(tk + 32 + q * 8), gb = *reinterpret_cast<const float4*>(tk + 32 + q * 8 + 4);
          const float cx[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
          const float dcx[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
          const float fyk = tk[64 + rw], dfyk = tk[96 + rw];
          const float isx = tk[128], isy = tk[129], zx = tk[130], zy = tk[131];
          float Sx = 0.0f, Sxu = 0.0f, Sy = 0.0f;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float u = (2.0f * (q * 8 + c) + 1.0f) * (1.0f / kBgSide) - 1.0f;
            const float t = d[c] * dcx[c];
            Sx += t;
            Sxu = fmaf(t, u, Sxu);
            Sy = fmaf(d[c], cx[c], Sy);
          }
          // q = ((u - x)/sx + 1) * 16 - 0.5  ->  d q / d(1/sx) etc. carry the factor 0.5 * kBgSide
          const float hx = -(0.5f * kBgSide) * fyk * isx, hy = -(0.5f * kBgSide) * dfyk * isy * Sy;
          const float g_sx = wave_sum_lane63(hx * isx * (Sxu - zx * Sx));
          const float g_sy = wave_sum_lane63(hy * isy * (vrow - zy));
          const float g_x = wave_sum_lane63(hx * Sx);
          const float g_y = wave_sum_lane63(hy);
          if (lane == 63)
            *reinterpret_cast<float4*>(dz_part + (((size_t)f * kBgHalves + half) * n_obj + k) * 4) = float4{g_sx, g_sy, g_x, g_y};
        }
      }
    }
    __syncthreads();
  };
  if (f0 < n_frames) prefetch(f0);
  int f_first = f0;
  for (int f = f0; f < n_frames; f += fstep, ++it) {
    const float x = xn;
    float w, mraw = 0.0f;
    // per-object box factors for the scene backward
    float fx[NMAX], fy[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      fx[k] = fxn[k];
      fy[k] = fyn[k];
    }
    float dcur[R][G];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int g = 0; g < G; ++g) dcur[r][g] = den[r][g];
    if (f + fstep < n_frames) prefetch(f + fstep);
    bool pass = true;
    if (SCENE) {
      float run = 0.0f;
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if (k < n_obj) {
          run += fx[k] * fy[k];
          if (run > 1.0f) {
            run = 1.0f;
            pass = false;
          }
        }
      }
      w = 1.0f - run;
    } else {
      if (marg != nullptr) {
        mraw = marg[(size_t)f * kBgPix + p];
        w = 1.0f - fminf(fmaxf(mraw, 0.0f), 1.0f);
      } else {
        w = 1.0f;
      }
    }
    const float wx = w * x, wxx = wx * x, x2 = x * x;
    float dwr[R], dx = 0.0f;      // one partial sum per replica: R short dependent FMA chains instead of one of R * G links
#pragma unroll                    // (0.30 -> 0.23 ms: at two waves per SIMD the serial chain was what each frame waited for)
    for (int r = 0; r < R; ++r) {
      dwr[r] = 0.0f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float d = dcur[r][g];
        dwr[r] = fmaf(d, fmaf(cf[r][g][0], x2, fmaf(cf[r][g][1], x, cf[r][g][2])), dwr[r]);
        if (!SCENE) dx = fmaf(d, fmaf(cf[r][g][0], x + x, cf[r][g][1]), dx);
        gc[r][g][0] = fmaf(d, wxx, gc[r][g][0]);
        gc[r][g][1] = fmaf(d, wx, gc[r][g][1]);
        gc[r][g][2] = fmaf(d, w, gc[r][g][2]);
      }
    }
    float dw = dwr[0];
#pragma unroll
    for (int r = 1; r < R; ++r) dw += dwr[r];
    if (SCENE) {
      // w = 1 - min(1, sum box): d box_k = -dw when no clamp fired
      const int slot = it % NW;
      if (slot == 0) f_first = f;
      dbx[slot][threadIdx.x] = pass ? -dw : 0.0f;
      if (slot == NW - 1) flush(NW, f_first);
    } else {
      if (d_marg != nullptr) d_marg[(size_t)f * kBgPix + p] = (mraw >= 0.0f && mraw <= 1.0f) ? -dw : 0.0f;
      if (d_inputs != nullptr) d_inputs[(size_t)f * kBgPix + p] = dx * w;
    }
  }
  if (SCENE && (it % NW) != 0) flush(it % NW, f_first);
  float* o = gcoef_part + ((size_t)blockIdx.x * R * kBgThreads) * G * 3;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) o[((size_t)(r * kBgThreads + threadIdx.x) * G + g) * 3 + k] = gc[r][g][k];
}

// g_coef[r][p][g][3] = sum over the blocks that own pixel-half(p) of gcoef_part (fixed order).
// 256 threads = 32 elements x 8 slices of the partial blocks, as reduce_chunks_k.
template <int R, int G>
__global__ __launch_bounds__(256) void bgspn_coef_reduce_k(const float* __restrict__ gcoef_part, float* __restrict__ g_coef, int n_blocks) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + el;               // over R * 1024 * G * 3
  float s = 0.0f;
  const bool live = j < R * kBgPix * G * 3;
  if (live) {
    const int e = j % (G * 3);
    const int p = (j / (G * 3)) % kBgPix;
    const int r = j / (G * 3 * kBgPix);
    const int half = p / kBgThreads, pl = p % kBgThreads;
    for (int b = half + q * kBgHalves; b < n_blocks; b += 8 * kBgHalves)
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

}  // namespace stove
#include "spn_bg_mfma.hip"
namespace stove {

// =============================================================================================
constexpr int kBgR = 3, kBgG = 6, kBgNO = kBgR * 2 * kBgG;
constexpr int kBgRootChunks = 256;

static inline int bg_grid(int n_frames) {
  int g = n_frames < 256 ? n_frames : 256;
  if (g < 1) g = 1;
  return g * kBgHalves;
}

// [ell: n * halves * 36][dense coefficient image of the MFMA path (scene mode)]
size_t bgspn_fwd_ws_floats(int n_frames) { return (size_t)n_frames * kBgHalves * kBgNO + kBgDenseF; }
static inline float* bg_dense_of(float* ell_part, int n_frames) { return ell_part + (size_t)n_frames * kBgHalves * kBgNO; }

// ell_part must stay alive until the backward (it is the saved activation).
int bgspn_forward(const float* frames, const float* marg, const float* z, int n_obj, const int* side, const float* coef,
                  const float* wroot, float* ell_part, float* out, int n_frames, hipStream_t st, FrameMap fm = FrameMap{0, 0}) {
  if (n_frames == 0) return 0;
  const int grid = bg_grid(n_frames);
  int halves = kBgHalves;
  if (z != nullptr && n_obj >= 1 && n_obj <= 8) {
    // scene mode: leaf layer as a GEMM on the matrix cores (spn_bg_mfma.hip); ell is (n, 36), one "half"
    float* Cf = bg_dense_of(ell_part, n_frames);
    STOVE_LAUNCH(bg_dense_fwd_k, dim3((kBgDenseF + 255) / 256), dim3(256), 0, st, side, coef, Cf);
    STOVE_LAUNCH_CHECK();
    // one 16-frame tile per wave: at two tiles a launch over 25 344 frames was 198 workgroups of 4 waves -- three quarters of
    // the CUs with one wave per SIMD; 396 workgroups pay the coefficient stream twice (L2) and win 12 % (129 -> 113 us)
    constexpr int TPW = 1;
    const int waves = n_obj <= 4 ? 4 : 2;                                    // coverage tables: waves * 32 * n_obj * 64 floats of LDS
    const size_t lds = (size_t)waves * TPW * 16 * (n_obj * 64 + 4 + n_obj * 4) * sizeof(float);
    const int per_block = waves * TPW * 16;
    const dim3 grid_m((n_frames + per_block - 1) / per_block), block_m(waves * 64);
#define STOVE_BG_FWD(NOBJ)                                                                                                        \
  {                                                                                                                               \
    int rc = (int)hipFuncSetAttribute((const void*)bgspn_mfma_fwd_k<TPW, NOBJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    if (rc) return rc;                                                                                                            \
    STOVE_LAUNCH((bgspn_mfma_fwd_k<TPW, NOBJ>), grid_m, block_m, lds, st, frames, z, n_obj, (const float*)Cf, ell_part, n_frames, fm);  \
  }
    if (n_obj == 3) STOVE_BG_FWD(3)
    else if (n_obj == 6) STOVE_BG_FWD(6)
    else if (n_obj == 2) STOVE_BG_FWD(2)
    else if (n_obj == 4) STOVE_BG_FWD(4)
    else STOVE_BG_FWD(0)
#undef STOVE_BG_FWD
    STOVE_LAUNCH_CHECK();
    halves = 1;
  } else if (z != nullptr) {
    STOVE_LAUNCH((bgspn_fwd_k<kBgR, kBgG, true>), dim3(grid), dim3(kBgThreads), 0, st, frames, marg, z, n_obj, side, coef, ell_part, n_frames, fm);
    STOVE_LAUNCH_CHECK();
  } else {
    STOVE_LAUNCH((bgspn_fwd_k<kBgR, kBgG, false>), dim3(grid), dim3(kBgThreads), 0, st, frames, marg, z, n_obj, side, coef, ell_part, n_frames, fm);
    STOVE_LAUNCH_CHECK();
  }
  STOVE_LAUNCH((bgspn_root_fwd_k<kBgR, kBgG>), dim3((n_frames + 255) / 256), dim3(256), 0, st, ell_part, wroot, out, n_frames, halves);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// the backward's per-(frame, object) coverage tables depend on z only: the scene forward makes them on its background stream,
// where that chain has slack, instead of the backward making them at the head of its longest chain
size_t bg_cover_floats(int n_frames, int n_obj) { return (size_t)n_frames * n_obj * kBgTab; }
int bg_cover_tables(const float* z, float* T, int n_frames, int n_obj, hipStream_t st) {
  const int pairs = n_frames * n_obj;
  if (pairs == 0) return 0;
  STOVE_LAUNCH(bg_cover_tables_k, dim3((pairs + 3) / 4), dim3(256), 0, st, z, T, pairs);
  STOVE_LAUNCH_CHECK();
  return 0;
}

size_t bgspn_bwd_ws_floats(int n_frames, int n_obj = 0) {
  const size_t grid = bg_grid(n_frames);
  return (size_t)n_frames * (kBgNO + kBgR * (1 + 2 * kBgG)) + (size_t)n_frames * kBgHalves * 8 * 4 +
         grid * kBgR * kBgThreads * kBgG * 3 + (size_t)kBgRootChunks * kBgR * kBgG * kBgG + (size_t)n_frames * n_obj * kBgTab;
}

template <int NMAX>
static int bg_bwd_launch(bool scene, int grid, hipStream_t st, const float* frames, const float* marg, const float* z,
                         int n_obj, const int* side, const float* coef, const float* dell, float* d_inputs,
                         float* d_marg, float* dz_part, float* gpart, int n_frames, const float* T, FrameMap fm) {
  if (scene)
    if (n_obj == NMAX)
      STOVE_LAUNCH((bgspn_bwd_k<kBgR, kBgG, true, NMAX, true>), dim3(grid), dim3(kBgThreads), 0, st, frames, marg, z, n_obj, side, coef, dell, d_inputs, d_marg, dz_part, gpart, n_frames, T, fm);
    else
      STOVE_LAUNCH((bgspn_bwd_k<kBgR, kBgG, true, NMAX, false>), dim3(grid), dim3(kBgThreads), 0, st, frames, marg, z, n_obj, side, coef, dell, d_inputs, d_marg, dz_part, gpart, n_frames, T, fm);
  else
    STOVE_LAUNCH((bgspn_bwd_k<kBgR, kBgG, false, 1>), dim3(grid), dim3(kBgThreads), 0, st, frames, marg, z, n_obj, side, coef, dell, d_inputs, d_marg, dz_part, gpart, n_frames, T, fm);
  STOVE_LAUNCH_CHECK();
  return 0;
}

__global__ void bg_dz_halves_k(const float* __restrict__ dz_part, float* __restrict__ dz, int n_frames, int no4) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_frames * no4) return;
  const int f = j / no4, e = j % no4;
  float s = 0.0f;
  for (int h = 0; h < kBgHalves; ++h) s += dz_part[((size_t)f * kBgHalves + h) * no4 + e];
  dz[j] = s;
}

// dz (SCENE): [n_frames][n_obj][4] overwritten.  g_coef [R][1024][G][3], g_wroot [R*G*G] overwritten.
int bgspn_backward(const float* frames, const float* marg, const float* z, int n_obj, const int* side, const float* coef,
                   const float* wroot, const float* ell_part, const float* out, const float* dout,
                   float* d_inputs, float* d_marg, float* dz, float* g_coef, float* g_wroot, float* ws,
                   int n_frames, hipStream_t st, hipStream_t st_par = nullptr, FrameMap fm = FrameMap{0, 0}, const float* T_pre = nullptr) {
  if (st_par == nullptr) st_par = st;          // stream of the parameter-gradient reductions (see objspn_backward)
  if (n_frames == 0) {
    hipMemsetAsync(g_coef, 0, sizeof(float) * kBgR * kBgPix * kBgG * 3, st);
    hipMemsetAsync(g_wroot, 0, sizeof(float) * kBgR * kBgG * kBgG, st);
    return 0;
  }
  if (n_obj > 8) return (int)hipErrorInvalidValue;
  const int grid = bg_grid(n_frames);
  float* dell = ws;
  float* rsc = dell + (size_t)n_frames * kBgNO;
  float* dz_part = rsc + (size_t)n_frames * kBgR * (1 + 2 * kBgG);
  float* gpart = dz_part + (size_t)n_frames * kBgHalves * 8 * 4;
  float* rpart = gpart + (size_t)grid * kBgR * kBgThreads * kBgG * 3;
  float* T = rpart + (size_t)kBgRootChunks * kBgR * kBgG * kBgG;            // scene mode only (ws sized with n_obj)
  const int halves = (z != nullptr && n_obj >= 1 && n_obj <= 8) ? 1 : kBgHalves;      // as written by bgspn_forward
  STOVE_LAUNCH((bgspn_root_bwd_k<kBgR, kBgG>), dim3((n_frames + 255) / 256), dim3(256), 0, st, ell_part, wroot, out, dout, dell, rsc, n_frames, halves);
  STOVE_LAUNCH_CHECK();
  const bool scene = z != nullptr;
  if (scene && T_pre != nullptr) {
    T = const_cast<float*>(T_pre);          // the coverage tables of this z, made by the forward (bg_cover_tables)
  } else if (scene) {
    const int pairs = n_frames * n_obj;
    STOVE_LAUNCH(bg_cover_tables_k, dim3((pairs + 3) / 4), dim3(256), 0, st, z, T, pairs);
    STOVE_LAUNCH_CHECK();
  }
  int rc;
  if (!scene || n_obj <= 3)
    rc = bg_bwd_launch<3>(scene, grid, st, frames, marg, z, n_obj, side, coef, dell, d_inputs, d_marg, dz_part, gpart, n_frames, T, fm);
  else if (n_obj <= 6)
    rc = bg_bwd_launch<6>(scene, grid, st, frames, marg, z, n_obj, side, coef, dell, d_inputs, d_marg, dz_part, gpart, n_frames, T, fm);
  else
    rc = bg_bwd_launch<8>(scene, grid, st, frames, marg, z, n_obj, side, coef, dell, d_inputs, d_marg, dz_part, gpart, n_frames, T, fm);
  if (rc) return rc;
  STOVE_TRY(stream_after(st_par, st));         // gcoef_part of bgspn_bwd_k, rsc of bgspn_root_bwd_k
  if (scene) {
    const int n = n_frames * n_obj * 4;
    STOVE_LAUNCH(bg_dz_halves_k, dim3((n + 255) / 256), dim3(256), 0, st, dz_part, dz, n_frames, n_obj * 4);
    STOVE_LAUNCH_CHECK();
  }
  const int nc = kBgR * kBgPix * kBgG * 3;
  STOVE_LAUNCH((bgspn_coef_reduce_k<kBgR, kBgG>), dim3((nc + 31) / 32), dim3(256), 0, st_par, gpart, g_coef, grid);
  STOVE_LAUNCH_CHECK();
  const int chunks = n_frames < kBgRootChunks ? n_frames : kBgRootChunks;
  STOVE_LAUNCH((bgspn_rootgrad_k<kBgR, kBgG>), dim3(chunks), dim3(128), 0, st_par, rsc, rpart, n_frames, chunks);
  STOVE_LAUNCH_CHECK();
  STOVE_LAUNCH(reduce_chunks_k, dim3((kBgR * kBgG * kBgG + 31) / 32), dim3(256), 0, st_par, rpart, g_wroot, kBgR * kBgG * kBgG, chunks, 0);
  STOVE_LAUNCH_CHECK();
  return 0;
}

}  // namespace stove
