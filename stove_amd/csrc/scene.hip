// SuPAIR scene side of the likelihood: glimpses, occlusion masks, assembly (gfx950).
//
// Replaces, fused with the SPN sweeps, the reference's
//   Supair.patches_from_z  (model/video_prediction/supair.py:241-276)  F.affine_grid + F.grid_sample
//   Supair.masks_from_z    (supair.py:278-356)                        N sequential sample/paste rounds
//   Supair.likelihood      (supair.py:79-94)                          scaling, overlap prior, sum
// Semantics are those of torch >= 1.3 (align_corners=False), which is what the runnable
// reference computes.
//
// The sequential mask recursion is removed algebraically: a pasted unit box is separable,
// box_k[Y][X] = cover_x(X) * cover_y(Y), and clamp(clamp(a+b)+c) = min(1, a+b+c) for
// non-negative terms, so the mask an object sees at a bilinear tap is a closed form of the
// earlier objects' z.  Every (patch, pixel) is therefore independent: one lane per patch,
// pixels looped, results written coalesced into the [batch][pixel][x|w][64] tile the
// object-SPN kernels consume.
#include "common.h"

namespace stove {

constexpr int kImg = 32;     // frame side
constexpr int kPatch = 10;   // glimpse side
constexpr int kPD = kPatch * kPatch;

struct PatchPix {
  Tap1 tx, ty;
  float u, v;
};
__device__ __forceinline__ PatchPix patch_pix(const float* zk, int p) {
  PatchPix q;
  const int i = p / kPatch, j = p % kPatch;
  q.u = (2.0f * j + 1.0f) * (1.0f / kPatch) - 1.0f;
  q.v = (2.0f * i + 1.0f) * (1.0f / kPatch) - 1.0f;
  const float gx = fmaf(zk[0], q.u, zk[2]);
  const float gy = fmaf(zk[1], q.v, zk[3]);
  q.tx = make_tap(((gx + 1.0f) * kImg - 1.0f) * 0.5f, kImg);
  q.ty = make_tap(((gy + 1.0f) * kImg - 1.0f) * 0.5f, kImg);
  return q;
}

__device__ __forceinline__ float inv_pix(float inv_s, float off, int idx) {
  const float u = (2.0f * idx + 1.0f) * (1.0f / kImg) - 1.0f;
  return ((fmaf(inv_s, u, off) + 1.0f) * kImg - 1.0f) * 0.5f;
}
// the same from the geometry struct
__device__ __forceinline__ PatchPix patch_pix_g(const float* zk, int p, const SceneGeom& gm) {
  PatchPix q;
  const int i = p / kPatch, j = p % kPatch;
  q.u = fmaf(gm.pa, (float)j, gm.pb);
  q.v = fmaf(gm.pa, (float)i, gm.pb);
  const float gx = fmaf(zk[0], q.u, zk[2]);
  const float gy = fmaf(zk[1], q.v, zk[3]);
  q.tx = make_tap(fmaf(gm.sxa, gx, gm.cx), gm.W);
  q.ty = make_tap(fmaf(gm.sya, gy, gm.cy), gm.H);
  return q;
}
__device__ __forceinline__ float inv_pix_x(float inv_s, float off, int idx, const SceneGeom& gm) {
  return fmaf(gm.sxa, fmaf(inv_s, fmaf(gm.fax, (float)idx, gm.fbx), off), gm.cx);
}
__device__ __forceinline__ float inv_pix_y(float inv_s, float off, int idx, const SceneGeom& gm) {
  return fmaf(gm.sya, fmaf(inv_s, fmaf(gm.fay, (float)idx, gm.fby), off), gm.cy);
}

// ---- tile forward: thread = (patch lane, pixel) -------------------------------------------
// frames [n_frames][1024], z [n_frames*n_obj][4] = [sx, sy, x, y]; xw [n_batches][100][2][64]
template <int NMAX, bool ANY = false>
__global__ __launch_bounds__(256) void scene_tile_fwd_k(const float* __restrict__ frames, const float* __restrict__ z,
                                                        float* __restrict__ xw, int n_obj, int n_patches, int n_batches, FrameMap fm,
                                                        SceneGeom gm = SceneGeom{}) {
  const int IW = ANY ? gm.W : kImg, IH = ANY ? gm.H : kImg;
  const int lane = lane_id();
  // workgroup = one batch of 64 glimpses, its 4 waves share the 100 pixels: the ~22 frames of a batch are then gathered by ONE
  // workgroup (one XCD's L2).  With (batch, pixel) items dealt round-robin over the whole grid every frame was pulled into
  // several L2s: 363 MB of HBM fetches per launch for 104 MB of frames (rocprofv3 FETCH_SIZE).
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    const int patch = b * 64 + lane;
    const bool live = patch < n_patches;
    const int pc = live ? patch : n_patches - 1;
    const int f = pc / n_obj, k = pc % n_obj;
    // per-glimpse constants of the 25 pixels this wave takes: the glimpse's transform and, for every EARLIER object of the
    // frame (the occluders), its inverse transform -- two IEEE divisions each, which the per-(batch, pixel) item loop redid
    // for every pixel
    const float* zf = z + (size_t)f * n_obj * 4;
    const float4 z4 = *reinterpret_cast<const float4*>(zf + k * 4);
    const float zk[4] = {z4.x, z4.y, z4.z, z4.w};
    constexpr int NOCC = NMAX > 1 ? NMAX - 1 : 1;      // the last object of a frame occludes nobody
    float isx[NOCC], isy[NOCC], ox[NOCC], oy[NOCC];
#pragma unroll
    for (int j = 0; j < NOCC; ++j) {
      const float4 zj = *reinterpret_cast<const float4*>(zf + (j < n_obj ? j : 0) * 4);
      isx[j] = 1.0f / zj.x;
      isy[j] = 1.0f / zj.y;
      ox[j] = -zj.z * isx[j];
      oy[j] = -zj.w * isy[j];
    }
    const float* img = frames + fm.row(f) * (size_t)(IW * IH);
    for (int p = wave_id(); p < kPD; p += 4) {
      const PatchPix q = ANY ? patch_pix_g(zk, p, gm) : patch_pix(zk, p);
      // the four taps: unconditional loads from clamped coordinates, the in-bounds flags folded into the weights
      float tap[4], wt[4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int iy = min(max(q.ty.i0 + a, 0), IH - 1), ix = min(max(q.tx.i0 + c, 0), IW - 1);
          tap[a * 2 + c] = img[iy * IW + ix];
          wt[a * 2 + c] = (a ? q.ty.in1 : q.ty.in0) * (c ? q.tx.in1 : q.tx.in0) * (a ? q.ty.t : 1.0f - q.ty.t) * (c ? q.tx.t : 1.0f - q.tx.t);
        }
      // earlier objects' coverage at the two tap columns / rows; run[a][c] = min(1, sum of their boxes) at tap (a, c)
      float run[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < NOCC; ++j) {
        if (j < k) {
          float d, cx0, cx1, cy0, cy1;
          if (ANY) {
            cx0 = cover(inv_pix_x(isx[j], ox[j], q.tx.i0, gm), IW, &d), cx1 = cover(inv_pix_x(isx[j], ox[j], q.tx.i0 + 1, gm), IW, &d);
            cy0 = cover(inv_pix_y(isy[j], oy[j], q.ty.i0, gm), IH, &d), cy1 = cover(inv_pix_y(isy[j], oy[j], q.ty.i0 + 1, gm), IH, &d);
          } else {
            cx0 = cover(inv_pix(isx[j], ox[j], q.tx.i0), kImg, &d), cx1 = cover(inv_pix(isx[j], ox[j], q.tx.i0 + 1), kImg, &d);
            cy0 = cover(inv_pix(isy[j], oy[j], q.ty.i0), kImg, &d), cy1 = cover(inv_pix(isy[j], oy[j], q.ty.i0 + 1), kImg, &d);
          }
          run[0] = fminf(run[0] + cx0 * cy0, 1.0f);
          run[1] = fminf(run[1] + cx1 * cy0, 1.0f);
          run[2] = fminf(run[2] + cx0 * cy1, 1.0f);
          run[3] = fminf(run[3] + cx1 * cy1, 1.0f);
        }
      }
      float xv = 0.0f, seen = 0.0f;
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        xv = fmaf(wt[t4], tap[t4], xv);
        seen = fmaf(wt[t4], 1.0f - run[t4], seen);
      }
      const float mg = 1.0f - seen;                               // supair.py:331
      const float wv = 1.0f - fminf(fmaxf(mg, 0.0f), 1.0f);       // rat_torch.py:104-106
      float* t = xw + ((size_t)b * kPD + p) * 2 * 64;
      t[lane] = live ? xv : 0.0f;
      t[64 + lane] = live ? wv : 0.0f;
    }
  }
}

// ---- the same with lane = PIXEL and the tile transposed through LDS (round 6) ------------------------------------------------
// scene_tile_fwd_k has lane = glimpse: each of a wave's 400 tap loads touches up to 64 cache lines of ~22 different frames, and the
// kernel sat in the texture path's issue stall for 56 % of its cycles (profiles/r06_pmc_sq_scene.json) -- 77 us per launch in front of
// the object SPN on the step's critical path.  Here a wave takes glimpses one at a time with its lanes on the glimpse's pixels (two
// passes: 64 + 36): the four taps of neighbouring pixels fall into the same few rows of ONE frame, so a tap load touches a handful of
// lines; the glimpse's transform and its occluders' inverse transforms are wave-uniform.  The [pixel][x | w][64 glimpses] tile the
// SPN kernels read is assembled in LDS (row stride 129 floats: the lanes of a pass write 64 different banks twice) and written out
// with coalesced 256-B rows.  Arithmetic, operand order and outputs are those of scene_tile_fwd_k: bit-identical (tests).
constexpr int kTileTWaves = 8;
constexpr int kTileTStride = 2 * 64 + 1;
template <int NMAX, bool ANY = false>
__global__ __launch_bounds__(64 * kTileTWaves) void scene_tile_fwd_t_k(const float* __restrict__ frames, const float* __restrict__ z,
                                                                       float* __restrict__ xw, int n_obj, int n_patches, int n_batches, FrameMap fm,
                                                                       SceneGeom gm = SceneGeom{}) {
  __shared__ float tl[kPD * kTileTStride];
  const int IW = ANY ? gm.W : kImg, IH = ANY ? gm.H : kImg;
  const int lane = lane_id(), wv = wave_id();
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    for (int g = wv; g < 64; g += kTileTWaves) {
      const int patch = b * 64 + g;
      const bool live = patch < n_patches;                    // wave-uniform
      const int pc = live ? patch : n_patches - 1;
      const int f = pc / n_obj, k = pc % n_obj;
      const float* zf = z + (size_t)f * n_obj * 4;
      const float4 z4 = *reinterpret_cast<const float4*>(zf + k * 4);
      const float zk[4] = {z4.x, z4.y, z4.z, z4.w};
      constexpr int NOCC = NMAX > 1 ? NMAX - 1 : 1;
      float isx[NOCC], isy[NOCC], ox[NOCC], oy[NOCC];
#pragma unroll
      for (int j = 0; j < NOCC; ++j) {
        const float4 zj = *reinterpret_cast<const float4*>(zf + (j < n_obj ? j : 0) * 4);
        isx[j] = 1.0f / zj.x;
        isy[j] = 1.0f / zj.y;
        ox[j] = -zj.z * isx[j];
        oy[j] = -zj.w * isy[j];
      }
      const float* img = frames + fm.row(f) * (size_t)(IW * IH);
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int p = pass * 64 + lane;
        if (p < kPD) {
          const PatchPix q = ANY ? patch_pix_g(zk, p, gm) : patch_pix(zk, p);
          float tap[4], wt[4];
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int iy = min(max(q.ty.i0 + a, 0), IH - 1), ix = min(max(q.tx.i0 + c, 0), IW - 1);
              tap[a * 2 + c] = img[iy * IW + ix];
              wt[a * 2 + c] = (a ? q.ty.in1 : q.ty.in0) * (c ? q.tx.in1 : q.tx.in0) * (a ? q.ty.t : 1.0f - q.ty.t) * (c ? q.tx.t : 1.0f - q.tx.t);
            }
          float run[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int j = 0; j < NOCC; ++j) {
            if (j < k) {
              float d, cx0, cx1, cy0, cy1;
              if (ANY) {
                cx0 = cover(inv_pix_x(isx[j], ox[j], q.tx.i0, gm), IW, &d), cx1 = cover(inv_pix_x(isx[j], ox[j], q.tx.i0 + 1, gm), IW, &d);
                cy0 = cover(inv_pix_y(isy[j], oy[j], q.ty.i0, gm), IH, &d), cy1 = cover(inv_pix_y(isy[j], oy[j], q.ty.i0 + 1, gm), IH, &d);
              } else {
                cx0 = cover(inv_pix(isx[j], ox[j], q.tx.i0), kImg, &d), cx1 = cover(inv_pix(isx[j], ox[j], q.tx.i0 + 1), kImg, &d);
                cy0 = cover(inv_pix(isy[j], oy[j], q.ty.i0), kImg, &d), cy1 = cover(inv_pix(isy[j], oy[j], q.ty.i0 + 1), kImg, &d);
              }
              run[0] = fminf(run[0] + cx0 * cy0, 1.0f);
              run[1] = fminf(run[1] + cx1 * cy0, 1.0f);
              run[2] = fminf(run[2] + cx0 * cy1, 1.0f);
              run[3] = fminf(run[3] + cx1 * cy1, 1.0f);
            }
          }
          float xv = 0.0f, seen = 0.0f;
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) {
            xv = fmaf(wt[t4], tap[t4], xv);
            seen = fmaf(wt[t4], 1.0f - run[t4], seen);
          }
          const float mg = 1.0f - seen;                               // supair.py:331
          const float wvv = 1.0f - fminf(fmaxf(mg, 0.0f), 1.0f);      // rat_torch.py:104-106
          tl[p * kTileTStride + g] = live ? xv : 0.0f;
          tl[p * kTileTStride + 64 + g] = live ? wvv : 0.0f;
        }
      }
    }
    __syncthreads();
    float* t = xw + (size_t)b * kPD * 2 * 64;
    for (int i = threadIdx.x; i < kPD * 2 * 64; i += 64 * kTileTWaves) t[i] = tl[(i >> 7) * kTileTStride + (i & 127)];
    __syncthreads();                                          // the next batch's glimpses write the buffer again
  }
}

// (the tile backward lives in scene_fused.hip: scene_pixtile_bwd_k)

// ---- assemble: log p(x, z) per frame (supair.py:79-94) -------------------------------------
// parts[frame][3] = (bg, patches, overlap prior)
__global__ void scene_assemble_fwd_k(const float* __restrict__ bg_ll, const float* __restrict__ obj_ll,
                                     const float* __restrict__ ovl, const float* __restrict__ z,
                                     float* __restrict__ ll, float* __restrict__ parts,
                                     int n_obj, int n_frames, float beta, float log_beta) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  float pl = 0.0f, ol = 0.0f;
  for (int k = 0; k < n_obj; ++k) {
    const size_t i = (size_t)f * n_obj + k;
    pl += obj_ll[i] * z[i * 4] * z[i * 4 + 1];
    ol += log_beta - beta * ovl[i];
  }
  const float b = bg_ll[f];
  ll[f] = (b + pl) + ol;
  if (parts != nullptr) {
    parts[f * 3] = b;
    parts[f * 3 + 1] = pl;
    parts[f * 3 + 2] = ol;
  }
}

// per patch: d obj_ll, d overlap from d ll
__global__ void scene_assemble_bwd_k(const float* __restrict__ dll, const float* __restrict__ z,
                                     float* __restrict__ d_obj, float* __restrict__ d_ovl,
                                     int n_obj, int n_patches, float beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_patches) {          // d_obj is read per batch of 64 patches as the scale of the unit-gradient scratch: zeros for the tail
    if (i < ((n_patches + 63) & ~63)) d_obj[i] = 0.0f;
    return;
  }
  const float g = dll[i / n_obj];
  d_obj[i] = g * z[(size_t)i * 4] * z[(size_t)i * 4 + 1];
  d_ovl[i] = -beta * g;
}

// dz[frame*n_obj + j][4] = bg part + sum_{k>=j} dzc[frame*n_obj+k][j] + direct scale terms
// bg_parts == 1: dz_bg [patch][4] is the background's whole share; > 1: the per-frame partial images the background SPN's backward
// leaves ([frame][bg_parts][n_obj * 4], one per half-frame workgroup), summed here in part order -- the sum was a launch of its own
// (bg_dz_halves_k) on the step's critical path until round 5.
template <int NMAX>
__global__ void scene_finalize_bwd_k(const float* __restrict__ dll, const float* __restrict__ z,
                                     const float* __restrict__ obj_ll, const float* __restrict__ dz_bg,
                                     const float* __restrict__ dzc, float* __restrict__ dz, int n_obj, int n_patches, int bg_parts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_patches) return;
  const int f = i / n_obj, j = i % n_obj;
  float s[4];
  if (bg_parts == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = dz_bg[(size_t)i * 4 + e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = 0.0f;
    for (int h = 0; h < bg_parts; ++h) {
      const float4 p = *reinterpret_cast<const float4*>(dz_bg + (((size_t)f * bg_parts + h) * n_obj + j) * 4);
      s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
    }
  }
  for (int k = j; k < n_obj; ++k) {
    const float* c = dzc + (((size_t)f * n_obj + k) * NMAX + j) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += c[e];
  }
  const float g = dll[f] * obj_ll[i];
  s[0] = fmaf(g, z[(size_t)i * 4 + 1], s[0]);
  s[1] = fmaf(g, z[(size_t)i * 4], s[1]);
#pragma unroll
  for (int e = 0; e < 4; ++e) dz[(size_t)i * 4 + e] = s[e];
}

// ---- object appearance embedding (Stove.object_embedding, stove.py:565-590): mean colour of every object's glimpse of
// the colour frame.  The 100 bilinear samples of patches_from_z are taken on the fly, so the
// (frames x objects x channels x 32 x 32) expansion and the glimpse tensor of the PyTorch path never exist.
// x_color [n_frames][C][1024], z [n_frames*n_obj][4] = [sx, sy, x, y] -> emb [n_frames*n_obj][C]
// One wave per glimpse: lane = pixel (two passes cover the 100), the four taps of every channel as unconditional loads from
// clamped coordinates with the in-bounds flag folded into the weight, then one wave reduction per channel.  (The first
// version ran one THREAD per (glimpse, channel) through 400 taps, each load inside its own in-bounds branch: 246 us.)
__global__ __launch_bounds__(256) void glimpse_mean_k(const float* __restrict__ x_color, const float* __restrict__ z, float* __restrict__ emb,
                                                      int n_patches, int n_obj, int C) {
  const int patch = blockIdx.x * (blockDim.x >> 6) + wave_id();
  if (patch >= n_patches) return;           // wave-uniform
  const int lane = lane_id();
  const int f = patch / n_obj;
  const float zk[4] = {z[(size_t)patch * 4], z[(size_t)patch * 4 + 1], z[(size_t)patch * 4 + 2], z[(size_t)patch * 4 + 3]};
  const float* img = x_color + (size_t)f * C * kImg * kImg;
  constexpr int CMAX = 4;
  float acc[CMAX] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int p = lane + 64 * pass;
    const bool live = p < kPD;
    const PatchPix q = patch_pix(zk, live ? p : 0);
    int off[4];
    float wt[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int iy = min(max(q.ty.i0 + a, 0), kImg - 1), ix = min(max(q.tx.i0 + b, 0), kImg - 1);
        off[a * 2 + b] = iy * kImg + ix;
        const float inb = (a ? q.ty.in1 : q.ty.in0) * (b ? q.tx.in1 : q.tx.in0);
        wt[a * 2 + b] = live ? inb * (a ? q.ty.t : 1.0f - q.ty.t) * (b ? q.tx.t : 1.0f - q.tx.t) : 0.0f;
      }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < C) {
        const float* ic = img + (size_t)c * kImg * kImg;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[c] = fmaf(wt[k], ic[off[k]], acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    if (c < C) {
      const float sum = wave_sum(acc[c]);
      if (lane == c) emb[(size_t)patch * C + c] = sum * (1.0f / kPD);
    }
  }
}

// ---- frame rendering (Supair.reconstruct_from_z, supair.py:484-498): out = clamp(bg + sum_k paste(patch_k; z_k), 0, 1) with
// paste = grid_sample of the 10x10 patch through the inverse transform [[1/sx, 0, -x/sx], [0, 1/sy, -y/sy]] onto the frame.
// thread = one pixel of one frame.  patches [.][100]: row (f / frames_per_patch) * n_obj + k, or the single row 0 shared by all
// objects when frames_per_patch == 0 (max-activation rendering).  bg [1024], z [n_frames*n_obj][4], out [n_frames][1024].
__global__ __launch_bounds__(256) void render_frames_k(const float* __restrict__ bg, const float* __restrict__ patches,
                                                       int frames_per_patch, const float* __restrict__ z, float* __restrict__ out,
                                                       int n_frames, int n_obj) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n_frames * kImg * kImg) return;
  const int f = (int)(t / (kImg * kImg)), pix = (int)(t % (kImg * kImg));
  const int Y = pix / kImg, X = pix % kImg;
  const float u = (2.0f * X + 1.0f) * (1.0f / kImg) - 1.0f, v = (2.0f * Y + 1.0f) * (1.0f / kImg) - 1.0f;
  float acc = bg[pix];
  for (int k = 0; k < n_obj; ++k) {
    const float* zk = z + ((size_t)f * n_obj + k) * 4;
    const float* pt = patches + (frames_per_patch > 0 ? ((size_t)(f / frames_per_patch) * n_obj + k) * kPD : 0);
    const float gx = (1.0f / zk[0]) * u + (-zk[2] / zk[0]);
    const float gy = (1.0f / zk[1]) * v + (-zk[3] / zk[1]);
    const Tap1 tx = make_tap(((gx + 1.0f) * kPatch - 1.0f) * 0.5f, kPatch);
    const Tap1 ty = make_tap(((gy + 1.0f) * kPatch - 1.0f) * 0.5f, kPatch);
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float inb = (a ? ty.in1 : ty.in0) * (b ? tx.in1 : tx.in0);
        if (inb != 0.0f) s = fmaf((a ? ty.t : 1.0f - ty.t) * (b ? tx.t : 1.0f - tx.t), pt[(ty.i0 + a) * kPatch + tx.i0 + b], s);
      }
    acc += s;
  }
  out[t] = fminf(fmaxf(acc, 0.0f), 1.0f);
}

}  // namespace stove
