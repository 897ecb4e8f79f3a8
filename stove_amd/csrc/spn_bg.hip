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
// The replica's share of the root, rho_r = go exp(M_r - mm) / Z, is formed from the SAME rounded M_r = m1 + m2, mm = max_r M_r and
// Z = sum_r S_r exp(M_r - mm) as the forward's value mm + log Z (M_r - mm is an exact subtraction) -- not as exp(M_r - out): a
// frame's log-density is ~ -10^4 in a trained model, `out` and M_r carry 10^-3 of absolute rounding each, and exp() of their
// difference put a common relative error of that size on every gradient of the frame (round 5, the 'stress' weight regime:
// 4.7e-4 on the background SPN's parameter gradients against 2e-5 for the reference's own float32 run).  `out` stays in the
// signature (the saved value of the forward) and is no longer read.
template <int R, int G>
__global__ void bgspn_root_bwd_k(const float* __restrict__ ell_part, const float* __restrict__ wroot,
                                 const float* __restrict__ out, const float* __restrict__ dout,
                                 float* __restrict__ dell, float* __restrict__ rsc, int n_frames, int halves) {
  constexpr int NO = R * 2 * G;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  const float go = dout[f];
  float M[R], Sr[R], E1[R][G], E2[R][G];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float a = 0.0f, b = 0.0f;
      for (int h = 0; h < halves; ++h) {
        a += ell_part[((size_t)f * halves + h) * NO + (r * 2) * G + g];
        b += ell_part[((size_t)f * halves + h) * NO + (r * 2 + 1) * G + g];
      }
      E1[r][g] = a;
      E2[r][g] = b;
    }
    float m1 = E1[r][0], m2 = E2[r][0];
#pragma unroll
    for (int g = 1; g < G; ++g) {
      m1 = fmaxf(m1, E1[r][g]);
      m2 = fmaxf(m2, E2[r][g]);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      E1[r][g] = __expf(E1[r][g] - m1);
      E2[r][g] = __expf(E2[r][g] - m2);
    }
    float s = 0.0f;
#pragma unroll
    for (int j2 = 0; j2 < G; ++j2)
#pragma unroll
      for (int j1 = 0; j1 < G; ++j1) s = fmaf(E1[r][j1] * E2[r][j2], wroot[r * G * G + j2 * G + j1], s);
    M[r] = m1 + m2;
    Sr[r] = s;
  }
  float mm = M[0];
#pragma unroll
  for (int r = 1; r < R; ++r) mm = fmaxf(mm, M[r]);
  float Z = 0.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) Z = fmaf(Sr[r], __expf(M[r] - mm), Z);
  const float goz = go / Z;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float rho = goz * __expf(M[r] - mm);
    float d1[G], d2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) d1[g] = d2[g] = 0.0f;
#pragma unroll
    for (int j2 = 0; j2 < G; ++j2)
#pragma unroll
      for (int j1 = 0; j1 < G; ++j1) {
        const float wk = wroot[r * G * G + j2 * G + j1];
        d1[j1] = fmaf(E2[r][j2], wk, d1[j1]);
        d2[j2] = fmaf(E1[r][j1], wk, d2[j2]);
      }
    float* rp = rsc + ((size_t)f * R + r) * (1 + 2 * G);
    rp[0] = rho;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      dell[(size_t)f * NO + (r * 2) * G + g] = rho * E1[r][g] * d1[g];
      dell[(size_t)f * NO + (r * 2 + 1) * G + g] = rho * E2[r][g] * d2[g];
      rp[1 + g] = E1[r][g];
      rp[1 + G + g] = E2[r][g];
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
//
// A workgroup walks its frames in GROUPS of NW = 8.  What every pixel lane of the workgroup needs of a frame beyond its own
// pixel -- the 36 leaf gradients and, in scene mode, the objects' coverage rows -- is the same few hundred floats for all 512
// lanes, so it goes through LDS: while group g is processed, the group g + 1 rows arrive as ONE float4 per thread (three for
// eight objects) and are parked in the other half of a double buffer at the group boundary, where the dz flush synchronises the
// workgroup anyway.  The lanes then read them with LDS latency.  (Before: 9 + 6 global loads per lane and frame, issued one
// frame ahead; at two waves per SIMD an iteration lasted what those loads took, ~2 600 cycles for ~100 instructions.)  The lane's
// own pixels of the next group are fetched a whole group ahead as well.
template <int R, int G, bool SCENE, int NMAX, bool EXACT = false>
__global__ __launch_bounds__(kBgThreads) void bgspn_bwd_k(
    const float* __restrict__ frames, const float* __restrict__ marg, const float* __restrict__ z, int n_obj,
    const int* __restrict__ side, const float* __restrict__ coef, const float* __restrict__ dell,
    float* __restrict__ d_inputs, float* __restrict__ d_marg, float* __restrict__ dz_part,
    float* __restrict__ gcoef_part, int n_frames, const float* __restrict__ T, FrameMap fm) {
  if (EXACT) n_obj = NMAX;      // the object count at compile time: the per-object predicates go, the table reads batch
  constexpr int NO = R * 2 * G;
  constexpr int NW = kBgThreads / 64;
  static_assert(NO % 4 == 0 && G % 2 == 0, "leaf-gradient rows are staged as float4, Gaussians are processed in pairs");
  constexpr int kDl4 = NW * (NO / 4);                                    // float4 items of a group: leaf-gradient rows ...
  // ... and the objects' coverage tables: whole rows of bg_cover_tables_k (132 floats; the flush then reads LDS as well) while
  // two buffers of them fit the 64 KB of static LDS, else just cover_x | cover_y (64 floats) and the flush reads global memory
  constexpr bool FULLTAB = NMAX <= 3;
  constexpr int TBW = FULLTAB ? kBgTab : 64, TB4 = TBW / 4, TBY = FULLTAB ? 64 : 32;
  constexpr int kTb4 = SCENE ? NW * NMAX * TB4 : 0;
  constexpr int NSTG = (kDl4 + kTb4 + kBgThreads - 1) / kBgThreads;
  // SCENE: d box (= -dL/dw where no clamp fired) of the group's frames, [slot][pixel of this half]; at the end of the group the
  // block turns them into dz, one wave per frame (see flush below)
  __shared__ __attribute__((aligned(16))) float dbx[SCENE ? NW : 1][SCENE ? kBgThreads : 1];
  __shared__ __attribute__((aligned(16))) float dl[2][NW][NO];
  __shared__ __attribute__((aligned(16))) float tb[2][SCENE ? NW : 1][SCENE ? NMAX : 1][TBW];
  const int half = blockIdx.x % kBgHalves;
  const int p = half * kBgThreads + threadIdx.x;
  const int lane = lane_id(), wv = wave_id();
  // Gaussians in pairs (g, g + 1): the inner products below are v_pk_fma_f32 -- two fp32 FMAs per lane and instruction --
  // with the per-pixel factors (x, x^2, w x^2, w x, w) as splat operands
  typedef float v2f __attribute__((ext_vector_type(2)));
  constexpr int GP = G / 2;
  v2f cf[R][GP][3], gc[R][GP][3];
  int doff[R];                  // where this lane's side of replica r starts in a leaf-gradient row
#pragma unroll
  for (int r = 0; r < R; ++r) {
    doff[r] = (r * 2 + (side[r * kBgPix + p] != 0 ? 1 : 0)) * G;
#pragma unroll
    for (int j = 0; j < GP; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        cf[r][j][k] = v2f{coef[((size_t)(r * kBgPix + p) * G + 2 * j) * 3 + k], coef[((size_t)(r * kBgPix + p) * G + 2 * j + 1) * 3 + k]};
        gc[r][j][k] = v2f{0.0f, 0.0f};
      }
  }
  const int col = p % kBgSide, row = p / kBgSide;
  const int fstep = gridDim.x / kBgHalves;
  const int f0 = blockIdx.x / kBgHalves;
  const int n_it = f0 < n_frames ? (n_frames - f0 + fstep - 1) / fstep : 0;      // frames of this block: f0 + it * fstep

  // ---- this thread's staging items (constant over the groups)
  const float* s_src[NSTG];     // source of frame 0
  int s_fstride[NSTG];          // floats per frame in the source; 0 = no item
  int s_slot[NSTG];
  float* s_dst[NSTG];           // destination in buffer 0
  int s_bstride[NSTG];          // floats between the two buffers
  const int n_items = kDl4 + (SCENE ? NW * n_obj * TB4 : 0);
#pragma unroll
  for (int u = 0; u < NSTG; ++u) {
    const int i = threadIdx.x + u * kBgThreads;
    s_src[u] = dell; s_fstride[u] = 0; s_slot[u] = 0; s_dst[u] = &dl[0][0][0]; s_bstride[u] = 0;
    if (i < kDl4) {
      const int j = i / (NO / 4), q = i % (NO / 4);
      s_src[u] = dell + 4 * q; s_fstride[u] = NO; s_slot[u] = j; s_dst[u] = &dl[0][j][4 * q]; s_bstride[u] = NW * NO;
    } else if (SCENE && i < n_items) {
      const int i2 = i - kDl4;
      const int j = i2 / (n_obj * TB4), rem = i2 % (n_obj * TB4), k = rem / TB4, q = rem % TB4;
      s_src[u] = T + k * kBgTab + (FULLTAB ? 4 * q : (q < 8 ? 4 * q : 64 + 4 * (q - 8)));
      s_fstride[u] = n_obj * kBgTab; s_slot[u] = j; s_dst[u] = &tb[0][j][k][4 * q]; s_bstride[u] = NW * NMAX * TBW;
    }
  }
  float4 stg[NSTG];
  float xcur[NW], xnext[NW];
  auto group_load = [&](int g, float* xs) {       // global loads of group g: staging items + this lane's pixels
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
      const int it = g * NW + s_slot[u];
      stg[u] = float4{0.0f, 0.0f, 0.0f, 0.0f};
      if (s_fstride[u] != 0 && it < n_it) stg[u] = *reinterpret_cast<const float4*>(s_src[u] + (size_t)(f0 + it * fstep) * s_fstride[u]);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int it = g * NW + j;
      xs[j] = 0.0f;
      if (it < n_it) xs[j] = frames[fm.row(f0 + it * fstep) * kBgPix + p];
    }
  };
  auto group_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NSTG; ++u)
      if (s_fstride[u] != 0) *reinterpret_cast<float4*>(s_dst[u] + buf * s_bstride[u]) = stg[u];
  };
  // dz of the group's frames.  The d box image of a frame enters the four gradients of object k only through three sums
  // per image row,  Sx = sum_c d dcover_x(c),  Sxu = sum_c d dcover_x(c) u(c),  Sy = sum_c d cover_x(c)  (the row factors
  // cover_y, dcover_y, v come out of the column sum), so one wave takes one frame: lane = (row, quarter of the columns),
  // 8 columns each, then ONE wave reduction per gradient -- instead of twelve 16-lane reductions, an LDS stage and a
  // workgroup barrier per frame in the pixel-parallel layout above.
  auto flush = [&](int n_buf, int f_first, int buf) {
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
          const float* tk = FULLTAB ? &tb[buf][wv][k][0] : T + ((size_t)f * n_obj + k) * kBgTab;
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

  if (n_it > 0) {
    group_load(0, xcur);
    group_store(0);
  }
  __syncthreads();
  for (int g = 0; g * NW < n_it; ++g) {
    const int buf = g & 1;
    const bool more = (g + 1) * NW < n_it;
    if (more) group_load(g + 1, xnext);
#pragma unroll
    for (int slot = 0; slot < NW; ++slot) {
      const int it = g * NW + slot;
      if (it < n_it) {            // block-uniform
        const int f = f0 + it * fstep;
        const float x = xcur[slot];
        float w, mraw = 0.0f;
        v2f dcur[R][GP];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < GP; ++j) dcur[r][j] = *reinterpret_cast<const v2f*>(&dl[buf][slot][doff[r] + 2 * j]);
        bool pass = true;
        if (SCENE) {
          float run = 0.0f;
#pragma unroll
          for (int k = 0; k < NMAX; ++k) {
            if (k < n_obj) {
              run += tb[buf][slot][k][col] * tb[buf][slot][k][TBY + row];
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
        const v2f X{x, x}, X2{x2, x2}, XX{x + x, x + x}, W{w, w}, WX{wx, wx}, WXX{wxx, wxx};
        v2f dwr[R], dxr{0.0f, 0.0f};  // one partial sum per replica: R short dependent FMA chains instead of one of R * G links
#pragma unroll
        for (int r = 0; r < R; ++r) {
          dwr[r] = v2f{0.0f, 0.0f};
#pragma unroll
          for (int j = 0; j < GP; ++j) {
            const v2f d = dcur[r][j];
            dwr[r] = __builtin_elementwise_fma(d, __builtin_elementwise_fma(cf[r][j][0], X2, __builtin_elementwise_fma(cf[r][j][1], X, cf[r][j][2])), dwr[r]);
            if (!SCENE) dxr = __builtin_elementwise_fma(d, __builtin_elementwise_fma(cf[r][j][0], XX, cf[r][j][1]), dxr);
            gc[r][j][0] = __builtin_elementwise_fma(d, WXX, gc[r][j][0]);
            gc[r][j][1] = __builtin_elementwise_fma(d, WX, gc[r][j][1]);
            gc[r][j][2] = __builtin_elementwise_fma(d, W, gc[r][j][2]);
          }
        }
        v2f dw2 = dwr[0];
#pragma unroll
        for (int r = 1; r < R; ++r) dw2 += dwr[r];
        const float dw = dw2.x + dw2.y, dx = dxr.x + dxr.y;
        if (SCENE) {
          // w = 1 - min(1, sum box): d box_k = -dw when no clamp fired
          dbx[slot][threadIdx.x] = pass ? -dw : 0.0f;
        } else {
          if (d_marg != nullptr) d_marg[(size_t)f * kBgPix + p] = (mraw >= 0.0f && mraw <= 1.0f) ? -dw : 0.0f;
          if (d_inputs != nullptr) d_inputs[(size_t)f * kBgPix + p] = dx * w;
        }
      }
    }
    // the next group's rows go into the other buffer (last read a group ago, two barriers back); the barriers of the flush
    // (or the one below) publish them
    if (more) group_store(buf ^ 1);
    if (SCENE) {
      const int left = n_it - g * NW;
      flush(left < NW ? left : NW, f0 + g * NW * fstep, buf);
    } else {
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) xcur[j] = xnext[j];
  }
  float* o = gcoef_part + ((size_t)blockIdx.x * R * kBgThreads) * G * 3;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < GP; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        o[((size_t)(r * kBgThreads + threadIdx.x) * G + 2 * j) * 3 + k] = gc[r][j][k].x;
        o[((size_t)(r * kBgThreads + threadIdx.x) * G + 2 * j + 1) * 3 + k] = gc[r][j][k].y;
      }
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
                  const float* wroot, float* ell_part, float* out, int n_frames, hipStream_t st, FrameMap fm = FrameMap{0, 0},
                  const float* dense = nullptr) {      // dense: the coefficient image of the scene-mode GEMM if the caller has it
  if (n_frames == 0) return 0;
  const int grid = bg_grid(n_frames);
  int halves = kBgHalves;
  if (z != nullptr && n_obj >= 1 && n_obj <= 8) {
    // scene mode: leaf layer as a GEMM on the matrix cores (spn_bg_mfma.hip); ell is (n, 36), one "half"
    const float* Cf = dense;
    if (Cf == nullptr) {
      float* own = bg_dense_of(ell_part, n_frames);
      STOVE_LAUNCH(bg_dense_fwd_k, dim3((kBgDenseF + 255) / 256), dim3(256), 0, st, side, coef, own);
      STOVE_LAUNCH_CHECK();
      Cf = own;
    }
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
    STOVE_LAUNCH((bgspn_mfma_fwd_k<TPW, NOBJ>), grid_m, block_m, lds, st, frames, z, n_obj, Cf, ell_part, n_frames, fm);  \
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

// the per-half partial images of dz inside bgspn_backward's workspace: [n_frames][kBgHalves][n_obj * 4]
static inline float* bgspn_dz_parts(float* ws, int n_frames) {
  return ws + (size_t)n_frames * kBgNO + (size_t)n_frames * kBgR * (1 + 2 * kBgG);
}
// dz (SCENE): [n_frames][n_obj][4] overwritten, or null: the caller sums the partial images (bgspn_dz_parts) itself.
// g_coef [R][1024][G][3], g_wroot [R*G*G] overwritten.
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
  // the backward holds one 8-wave workgroup per CU (212 registers): 128 frame slices x 2 halves = one resident round (the
  // workspace is sized for bg_grid, which is never smaller)
  const int grid = n_frames < 128 ? bg_grid(n_frames) : 128 * kBgHalves;
  float* dell = ws;
  float* rsc = dell + (size_t)n_frames * kBgNO;
  float* dz_part = bgspn_dz_parts(ws, n_frames);
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
  if (scene && dz != nullptr) {
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
