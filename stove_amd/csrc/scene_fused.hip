// Fused backward of the glimpse tile: object-SPN pixel gradients + spatial-transformer / occlusion-mask backward in ONE pass.
//
// Replaces, inside stove_scene_bwd, the pair
//     objspn_pix_k      dL/d(x, w) of every tile pixel from the leaf gradients   (wrote a 51 KB dxw tile per 64 glimpses)
//     scene_tile_bwd_k  dL/d(x, w) -> dL/dz of the glimpse's own object and of its occluders   (re-read that tile)
// i.e. the backward of Supair.patches_from_z / masks_from_z (reference supair.py:241-356) chained to the backward of
// GaussVector.forward (rat_torch.py:83-109).  A pixel's (dL/dx, dL/dw) is formed in registers and consumed at once by the
// transformer backward of the same pixel: the dxw tile never exists.  thread = (glimpse lane, pixel slot); the batch's leaf
// gradients (61 KB) are staged in LDS and the same LDS is reused for the cross-slot reduction.
//
// Arithmetic, per (glimpse, pixel), restated branch-free:
//   * occluder coverage of a pasted unit box sampled at frame column X:  cover(q) = clamp(min(q + 1, 32 - q), 0, 1) with
//     q = (X - 15.5) / sx_j + 15.5 - 16 x_j / sx_j  (the inverse affine grid of supair.py:233-237 in pixel units) -- the
//     closed form of grid_sample on a ones image with zero padding; its derivative is +-1 on the two ramps;
//   * objects j >= k do not occlude glimpse k (supair.py:304-356 pastes in order): they get a coverage of 0 through their
//     constants instead of a branch;
//   * sequential clamping of the mask, clamp(clamp(a + b) + c) = min(1, a + b + c): the gradient passes iff the sum is <= 1;
//   * the occluder gradients are accumulated as raw sums over the pixels and scaled by 1/sx, 1/sx^2 once at the end.
#include "common.h"

namespace stove {

// coverage of a ones image of kImg samples (bilinear, zero padding) at pixel coordinate q, and d/dq
__device__ __forceinline__ float cover_cf(float q, float* dq, float n = (float)kImg) {
  const float a = q + 1.0f, b = n - q;
  const float m = fminf(a, b);
  *dq = (m > 0.0f && m < 1.0f) ? (a < b ? 1.0f : -1.0f) : 0.0f;
  return fminf(fmaxf(m, 0.0f), 1.0f);
}

template <int R, int S, int G, int NMAX, int SLOTS, bool ANY = false>
__global__ __launch_bounds__(64 * SLOTS) void scene_pixtile_bwd_k(
    const float* __restrict__ frames, const float* __restrict__ z, const float* __restrict__ xw, const float* __restrict__ Dscr,
    const int* __restrict__ leaf_slot, const float* __restrict__ coef, const float* __restrict__ d_ovl, float* __restrict__ dzc,
    int n_obj, int n_patches, int n_batches, FrameMap fm, const float* __restrict__ scale, SceneGeom gm = SceneGeom{}) {
  // ANY: frame size and sampling convention from `gm` (scene.hip SceneGeom); else the 32 x 32 / align_corners=False constants
  const int IW = ANY ? gm.W : kImg, IH = ANY ? gm.H : kImg;
  const float SXA = ANY ? gm.sxa : 0.5f * kImg, SYA = ANY ? gm.sya : 0.5f * kImg;          // d pixel / d normalised coordinate
  const float CXc = ANY ? gm.cx : 15.5f, CYc = ANY ? gm.cy : 15.5f;                        // pixel coordinate of the frame centre
  constexpr int D = 4 * S;
  constexpr int DT = R * 4 * G * 64;
  constexpr int RED = SLOTS * NMAX * 4 * 64;
  constexpr int CFT = D * R * 32;            // per (pixel, replica): the leaf's G x 3 coefficients (30 floats), leaf index, pad
  // occluders of object k are the objects j < k of its frame: the last object occludes nobody, so NMAX - 1 occluder slots
  // cover every case (a third of the per-pixel coverage work at three objects)
  constexpr int NOCC = NMAX > 1 ? NMAX - 1 : 1;
  extern __shared__ __attribute__((aligned(16))) float pt_lds[];          // [CFT] coefficient rows by pixel | max(DT, RED): leaf gradients, then the slot reduction
  float* cft = pt_lds;
  float* dl = pt_lds + CFT;
  const int lane = lane_id(), slot = wave_id();
  // The coefficients a pixel needs (one leaf per replica: 6 x 30 floats) are wave-uniform.  As scalar loads they came out as
  // ~30 dependent "s_load; s_waitcnt lgkmcnt(0)" round trips per pixel (the leaf index feeds the row address, and lgkmcnt also
  // counts the LDS reads of the leaf gradients): ~5 000 cycles per pixel, 20 % VALU issue.  The whole table is 77 KB: it is
  // staged in LDS once per workgroup (persistent over its batches) and read back as broadcast float4.
  for (int i = threadIdx.x; i < D * R; i += 64 * SLOTS) {
    const int p = i / R, r = i % R;
    const int ls = leaf_slot[r * D + p];
    const float* cf = coef + (size_t)(r * 4 * S + ls) * G * 3;
    float* dst = cft + i * 32;
#pragma unroll
    for (int e = 0; e < G * 3; ++e) dst[e] = cf[e];
    dst[30] = __int_as_float(ls / S);
    dst[31] = 0.0f;
  }
  for (int b = blockIdx.x; b < n_batches; b += gridDim.x) {
    {
      const float4* src = reinterpret_cast<const float4*>(Dscr + (size_t)b * DT);
      float4* dst = reinterpret_cast<float4*>(dl);
      if (scale == nullptr) {
        for (int i = threadIdx.x; i < DT / 4; i += 64 * SLOTS) dst[i] = src[i];
      } else {
        // leaf gradients written for an upstream gradient of 1 (objspn_fwd_unit_k): times dL/d root of the sample, here.
        // 64 * SLOTS is a multiple of 16, so a thread keeps its four samples over the whole copy
        const float4 sc = *reinterpret_cast<const float4*>(scale + b * 64 + ((4 * threadIdx.x) & 63));      // a multiple of 64 entries, zeros beyond the last patch
        for (int i = threadIdx.x; i < DT / 4; i += 64 * SLOTS) {
          float4 v = src[i];
          v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
          dst[i] = v;
        }
      }
    }
    const int patch = b * 64 + lane;
    const bool live = patch < n_patches;
    const int f = live ? patch / n_obj : 0, k = live ? patch % n_obj : 0;
    const float* zf = z + (size_t)f * n_obj * 4;
    const float zk[4] = {zf[k * 4], zf[k * 4 + 1], zf[k * 4 + 2], zf[k * 4 + 3]};
    const float* img = frames + fm.row(f) * (size_t)(IW * IH);
    const float govl = live ? d_ovl[patch] * (-1.0f / kPD) : 0.0f;   // d overlap / d seen = -1/100
    // occluders j < k: q(X) = isx (X - 15.5) + cxo;  j >= k: coverage 0 everywhere
    float isx[NOCC], isy[NOCC], cxo[NOCC], cyo[NOCC], xj[NOCC], yj[NOCC];
#pragma unroll
    for (int j = 0; j < NOCC; ++j) {
      const bool occ = live && j < k;
      const float sx = occ ? zf[j * 4] : 1.0f, sy = occ ? zf[j * 4 + 1] : 1.0f;
      xj[j] = occ ? zf[j * 4 + 2] : 0.0f;
      yj[j] = occ ? zf[j * 4 + 3] : 0.0f;
      isx[j] = occ ? 1.0f / sx : 0.0f;
      isy[j] = occ ? 1.0f / sy : 0.0f;
      cxo[j] = occ ? (ANY ? fmaf(-SXA * xj[j], isx[j], CXc) : fmaf(-16.0f * xj[j], isx[j], 15.5f)) : -100.0f;
      cyo[j] = occ ? (ANY ? fmaf(-SYA * yj[j], isy[j], CYc) : fmaf(-16.0f * yj[j], isy[j], 15.5f)) : -100.0f;
    }
    float own[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float sx0[NOCC], sx2[NOCC], sy1[NOCC], sy3[NOCC];      // raw occluder sums: dqx (uu - x_j), dqx, dqy (vv - y_j), dqy
#pragma unroll
    for (int j = 0; j < NOCC; ++j) sx0[j] = sx2[j] = sy1[j] = sy3[j] = 0.0f;
    __syncthreads();
    const float* tile = xw + (size_t)b * (D * 2 * 64);
    for (int p = slot; p < D; p += SLOTS) {
      // ---- dL/dx, dL/dw of the pixel (GaussVector backward in expanded form: x^2 A + x B + C)
      const float x = tile[(p * 2) * 64 + lane];
      const float w = tile[(p * 2 + 1) * 64 + lane];
      // per-replica partial sums: 3 R independent FMA chains of length G instead of 3 chains of length R G
      float Ar[R], Br[R], Cr[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4* row = reinterpret_cast<const float4*>(cft + (p * R + r) * 32);      // broadcast reads
        float cf[32];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float4 v = row[e];
          cf[4 * e] = v.x; cf[4 * e + 1] = v.y; cf[4 * e + 2] = v.z; cf[4 * e + 3] = v.w;
        }
        const int L = __float_as_int(cf[30]);
        const float* dlp = dl + ((r * 4 + L) * G) * 64 + lane;
        float a = 0.0f, bq = 0.0f, c = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float d = dlp[g * 64];
          a = fmaf(d, cf[g * 3], a);
          bq = fmaf(d, cf[g * 3 + 1], bq);
          c = fmaf(d, cf[g * 3 + 2], c);
        }
        Ar[r] = a; Br[r] = bq; Cr[r] = c;
      }
      float A = 0.0f, Bq = 0.0f, C = 0.0f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        A += Ar[r]; Bq += Br[r]; C += Cr[r];
      }
      const float gX = fmaf(x + x, A, Bq) * w;
      const float gW = fmaf(x, fmaf(x, A, Bq), C);
      // ---- transformer / mask backward of the pixel
      const PatchPix q = ANY ? patch_pix_g(zk, p, gm) : patch_pix(zk, p);
      const int c0 = min(max(q.tx.i0, 0), IW - 1), c1 = min(max(q.tx.i0 + 1, 0), IW - 1);
      const int r0 = min(max(q.ty.i0, 0), IH - 1), r1 = min(max(q.ty.i0 + 1, 0), IH - 1);
      const float inb00 = q.ty.in0 * q.tx.in0, inb01 = q.ty.in0 * q.tx.in1, inb10 = q.ty.in1 * q.tx.in0, inb11 = q.ty.in1 * q.tx.in1;
      const float im00 = img[r0 * IW + c0] * inb00, im01 = img[r0 * IW + c1] * inb01;
      const float im10 = img[r1 * IW + c0] * inb10, im11 = img[r1 * IW + c1] * inb11;
      float cx[NOCC][2], cy[NOCC][2], dcx[NOCC][2], dcy[NOCC][2];
      float s00 = 0.0f, s01 = 0.0f, s10 = 0.0f, s11 = 0.0f;     // mask sum at tap (row a, column c): s_ac
      const float fx = (float)q.tx.i0 - CXc, fy = (float)q.ty.i0 - CYc;
      const float nW = (float)IW, nH = (float)IH;
#pragma unroll
      for (int j = 0; j < NOCC; ++j) {
        const float qx = fmaf(isx[j], fx, cxo[j]), qy = fmaf(isy[j], fy, cyo[j]);
        cx[j][0] = cover_cf(qx, &dcx[j][0], nW);
        cx[j][1] = cover_cf(qx + isx[j], &dcx[j][1], nW);
        cy[j][0] = cover_cf(qy, &dcy[j][0], nH);
        cy[j][1] = cover_cf(qy + isy[j], &dcy[j][1], nH);
        s00 = fmaf(cx[j][0], cy[j][0], s00);
        s01 = fmaf(cx[j][1], cy[j][0], s01);
        s10 = fmaf(cx[j][0], cy[j][1], s10);
        s11 = fmaf(cx[j][1], cy[j][1], s11);
      }
      // visible fraction at the taps (0 outside the frame) and whether the clamp lets the gradient through
      const float vis00 = (1.0f - fminf(s00, 1.0f)) * inb00, vis01 = (1.0f - fminf(s01, 1.0f)) * inb01;
      const float vis10 = (1.0f - fminf(s10, 1.0f)) * inb10, vis11 = (1.0f - fminf(s11, 1.0f)) * inb11;
      const float wy0 = 1.0f - q.ty.t, wy1 = q.ty.t, wx0 = 1.0f - q.tx.t, wx1 = q.tx.t;
      const float seen = wy0 * (wx0 * vis00 + wx1 * vis01) + wy1 * (wx0 * vis10 + wx1 * vis11);
      const float mg = 1.0f - seen;
      // w = 1 - clamp(1 - seen): dw/dseen = 1 inside the clamp range (boundaries pass, as ATen)
      const float dseen = ((mg >= 0.0f && mg <= 1.0f) ? gW : 0.0f) + govl;
      // own object: through the sample location
      const float dpx = gX * (wy0 * (im01 - im00) + wy1 * (im11 - im10)) + dseen * (wy0 * (vis01 - vis00) + wy1 * (vis11 - vis10));
      const float dpy = gX * (wx0 * (im10 - im00) + wx1 * (im11 - im01)) + dseen * (wx0 * (vis10 - vis00) + wx1 * (vis11 - vis01));
      const float dgx = dpx * SXA, dgy = dpy * SYA;
      own[0] = fmaf(dgx, q.u, own[0]);
      own[1] = fmaf(dgy, q.v, own[1]);
      own[2] += dgx;
      own[3] += dgy;
      // occluders: through the mask value at each tap; dbox_ac = -dseen * wt_ac where the tap is inside and unclamped
      const float nb00 = (inb00 != 0.0f && s00 <= 1.0f) ? -dseen * wy0 * wx0 : 0.0f;
      const float nb01 = (inb01 != 0.0f && s01 <= 1.0f) ? -dseen * wy0 * wx1 : 0.0f;
      const float nb10 = (inb10 != 0.0f && s10 <= 1.0f) ? -dseen * wy1 * wx0 : 0.0f;
      const float nb11 = (inb11 != 0.0f && s11 <= 1.0f) ? -dseen * wy1 * wx1 : 0.0f;
      const float uu0 = ANY ? fmaf(gm.fax, (float)q.tx.i0, gm.fbx) : (2.0f * q.tx.i0 + 1.0f) * (1.0f / kImg) - 1.0f;
      const float uu1 = uu0 + (ANY ? gm.fax : 2.0f / kImg);
      const float vv0 = ANY ? fmaf(gm.fay, (float)q.ty.i0, gm.fby) : (2.0f * q.ty.i0 + 1.0f) * (1.0f / kImg) - 1.0f;
      const float vv1 = vv0 + (ANY ? gm.fay : 2.0f / kImg);
#pragma unroll
      for (int j = 0; j < NOCC; ++j) {
        const float gx0 = fmaf(nb00, cy[j][0], nb10 * cy[j][1]) * dcx[j][0];     // column c = 0: sum over the two rows
        const float gx1 = fmaf(nb01, cy[j][0], nb11 * cy[j][1]) * dcx[j][1];
        const float gy0 = fmaf(nb00, cx[j][0], nb01 * cx[j][1]) * dcy[j][0];     // row a = 0: sum over the two columns
        const float gy1 = fmaf(nb10, cx[j][0], nb11 * cx[j][1]) * dcy[j][1];
        sx0[j] = fmaf(gx0, uu0 - xj[j], fmaf(gx1, uu1 - xj[j], sx0[j]));
        sx2[j] += gx0 + gx1;
        sy1[j] = fmaf(gy0, vv0 - yj[j], fmaf(gy1, vv1 - yj[j], sy1[j]));
        sy3[j] += gy0 + gy1;
      }
    }
    __syncthreads();            // every slot is done with the leaf gradients: the LDS becomes red[slot][NMAX * 4][64]
    float* red = dl;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
      const bool mine = (j == k);
      // d q / d(1/s) etc.: q = (X - 15.5)/s + 15.5 - 16 x/s  =>  dL/ds = -16 (uu - x)/s^2 dL/dq,  dL/dx = -16/s dL/dq
      const float h = -SXA, hy = -SYA;
      const int jo = j < NOCC ? j : 0;                      // the last object is nobody's occluder: its sums are zeros
      const float sc = j < NOCC ? 1.0f : 0.0f;
      const float a0 = sc * h * isx[jo] * isx[jo] * sx0[jo], a1 = sc * hy * isy[jo] * isy[jo] * sy1[jo], a2 = sc * h * isx[jo] * sx2[jo], a3 = sc * hy * isy[jo] * sy3[jo];
      red[(slot * NMAX * 4 + j * 4 + 0) * 64 + lane] = mine ? own[0] : a0;
      red[(slot * NMAX * 4 + j * 4 + 1) * 64 + lane] = mine ? own[1] : a1;
      red[(slot * NMAX * 4 + j * 4 + 2) * 64 + lane] = mine ? own[2] : a2;
      red[(slot * NMAX * 4 + j * 4 + 3) * 64 + lane] = mine ? own[3] : a3;
    }
    __syncthreads();
    // sum over the slots in slot order (fixed order: bitwise reproducible); waves share the NMAX * 4 outputs
    for (int o = slot; o < NMAX * 4; o += SLOTS) {
      const int j = o >> 2, e = o & 3;
      if (live && j < n_obj) {
        float s = red[o * 64 + lane];
#pragma unroll
        for (int t = 1; t < SLOTS; ++t) s += red[(t * NMAX * 4 + o) * 64 + lane];
        dzc[((size_t)patch * NMAX + j) * 4 + e] = s;
      }
    }
    __syncthreads();
  }
}

template <int NMAX, bool ANY = false>
static int scene_pixtile_bwd(const float* frames, const float* z, const float* xw, const float* Dscr, const int* leaf_slot,
                             const float* coef, const float* d_ovl, float* dzc, int n_obj, int np, hipStream_t st, FrameMap fm,
                             const float* scale, SceneGeom gm = SceneGeom{}) {      // scale: see the kernel's staging loop; nullptr = Dscr carries it
  // 16 waves (4 per SIMD) when the cross-slot reduction buffer allows: the per-pixel code is a chain of dependent instructions
  // (~9 cycles per instruction at 2 waves per SIMD), more resident waves hide it
  constexpr int SLOTS = NMAX <= 3 ? 16 : 8;
  constexpr int DT = 6 * 4 * 10 * 64, RED = SLOTS * NMAX * 4 * 64, CFT = 100 * 6 * 32;
  constexpr int LDS = (CFT + (DT > RED ? DT : RED)) * (int)sizeof(float);
  const int nb = (np + 63) / 64;
  if (nb == 0) return 0;
  int rc = (int)hipFuncSetAttribute((const void*)scene_pixtile_bwd_k<6, 25, 10, NMAX, SLOTS, ANY>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (rc) return rc;
  STOVE_LAUNCH((scene_pixtile_bwd_k<6, 25, 10, NMAX, SLOTS, ANY>), dim3(nb < 256 ? nb : 256), dim3(64 * SLOTS), LDS, st, frames, z, xw, Dscr,      // persistent: one workgroup per CU
               leaf_slot, coef, d_ovl, dzc, n_obj, np, nb, fm, scale, gm);
  STOVE_LAUNCH_CHECK();
  return 0;
}

}  // namespace stove
