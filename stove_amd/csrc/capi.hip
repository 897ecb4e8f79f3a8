// C ABI of libstove_hip.so (see include/stove_hip.h).  Single translation unit: the kernel
// files are included here so every launch sees its kernel without relocatable device code.
#include "../../include/stove_hip.h"
#include <algorithm>

#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <atomic>
#include "validate.h"
// argument checks shared with the sanitizer-built host driver (csrc/validate.h, tests/abi/validate_driver.cpp)
static_assert(stove_validate::kStoveInvalidValue == (int)hipErrorInvalidValue, "validate.h returns hipErrorInvalidValue");
#define STOVE_VALIDATE(expr)                      \
  do {                                            \
    const int v__ = stove_validate::expr;         \
    if (v__) return v__;                          \
  } while (0)
#include "spn_obj.hip"
#include "spn_obj_generic.hip"
#include "spn_bg.hip"
#include "spn_bg_generic.hip"
#include "scene.hip"
#include "scene_fused.hip"
#include "gnn.hip"
#include "match.hip"
#include "gnn_small.hip"
#include "gnn_small_bwd.hip"
#include "lstm.hip"
#include "arena.hip"
#include "state.hip"
#include "gemm_bf16.hip"
#include "head_fused.hip"
#include "reward_head.hip"

namespace stove {

__global__ void wave_sum_test_k(const float* __restrict__ in, float* __restrict__ out) {
  const int w = blockIdx.x * (blockDim.x >> 6) + wave_id();
  const float v = in[(size_t)w * 64 + lane_id()];
  const float s = wave_sum(v);
  out[(size_t)w * 64 + lane_id()] = s;
}

// tile [nb][100][2][64] -> patches (n,100), keep (n,100)
__global__ void tile_unpack_k(const float* __restrict__ tile, float* __restrict__ patches, float* __restrict__ keep,
                              int n, int n_batches) {
  const int total = n_batches * kPD * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, p = (i >> 6) % kPD, b = i / (64 * kPD);
    const int smp = b * 64 + lane;
    if (smp >= n) continue;
    const float* t = tile + ((size_t)b * kPD + p) * 2 * 64;
    if (patches) patches[(size_t)smp * kPD + p] = t[lane];
    if (keep) keep[(size_t)smp * kPD + p] = t[64 + lane];
  }
}

// which glimpse-tile kernel the scene forward runs: 1 (default) = lane per PIXEL with the tile transposed through LDS up to three
// objects, lane per glimpse beyond (28 % more VALU instructions in the transposing kernel -- 36 of 64 lanes idle in its second pass --
// which the VALU-bound six-object scene phase pays for: 5.45 against 5.42 ms per step); 0 = lane per glimpse always; 2 = lane per
// pixel always (the tests compare the two bit for bit at every object count)
static std::atomic<int> g_tile_lds{1};
static inline bool tile_transposed(int n_obj) {
  const int m = g_tile_lds.load();
  return m == 2 || (m == 1 && n_obj <= 3);
}
template <int NMAX>
static int scene_tile_fwd(const float* frames, const float* z, float* xw, int n_obj, int np, hipStream_t st, FrameMap fm) {
  const int nb = (np + 63) / 64;
  const int grid = nb < 8192 ? nb : 8192;          // one workgroup per batch of 64 glimpses
  if (tile_transposed(n_obj)) {          // lane = pixel, tile transposed through LDS (scene_tile_fwd_t_k)
    STOVE_LAUNCH((scene_tile_fwd_t_k<NMAX>), dim3(grid), dim3(64 * kTileTWaves), 0, st, frames, z, xw, n_obj, np, nb, fm, SceneGeom{});
    STOVE_LAUNCH_CHECK();
    return 0;
  }
  STOVE_LAUNCH((scene_tile_fwd_k<NMAX>), dim3(grid), dim3(256), 0, st, frames, z, xw, n_obj, np, nb, fm, SceneGeom{});
  STOVE_LAUNCH_CHECK();
  return 0;
}
// the same kernels with the frame size / sampling convention at run time (stove_scene_fwd_any)
template <int NMAX>
static int scene_tile_fwd_g(const float* frames, const float* z, float* xw, int n_obj, int np, hipStream_t st, FrameMap fm, SceneGeom gm) {
  const int nb = (np + 63) / 64;
  const int grid = nb < 8192 ? nb : 8192;
  if (tile_transposed(n_obj)) {
    STOVE_LAUNCH((scene_tile_fwd_t_k<NMAX, true>), dim3(grid), dim3(64 * kTileTWaves), 0, st, frames, z, xw, n_obj, np, nb, fm, gm);
    STOVE_LAUNCH_CHECK();
    return 0;
  }
  STOVE_LAUNCH((scene_tile_fwd_k<NMAX, true>), dim3(grid), dim3(256), 0, st, frames, z, xw, n_obj, np, nb, fm, gm);
  STOVE_LAUNCH_CHECK();
  return 0;
}
// SceneGeom of a W x H frame under either sampling convention of affine_grid / grid_sample (common.h)
static SceneGeom scene_geom(int W, int H, int align_corners) {
  SceneGeom g;
  g.W = W; g.H = H;
  g.cx = 0.5f * (W - 1); g.cy = 0.5f * (H - 1);
  if (align_corners) {
    g.pa = 2.0f / (kPatch - 1); g.pb = -1.0f;
    g.sxa = 0.5f * (W - 1); g.sya = 0.5f * (H - 1);
    g.fax = 2.0f / (W - 1); g.fbx = -1.0f; g.fay = 2.0f / (H - 1); g.fby = -1.0f;
  } else {
    g.pa = 2.0f / kPatch; g.pb = 1.0f / kPatch - 1.0f;
    g.sxa = 0.5f * W; g.sya = 0.5f * H;
    g.fax = 2.0f / W; g.fbx = 1.0f / W - 1.0f; g.fay = 2.0f / H; g.fby = 1.0f / H - 1.0f;
  }
  return g;
}

static int scene_tile_fwd_any(const float* frames, const float* z, float* xw, int n_obj, int np, hipStream_t st, FrameMap fm = FrameMap{0, 0}) {
  if (n_obj <= 3) return scene_tile_fwd<3>(frames, z, xw, n_obj, np, st, fm);
  if (n_obj <= 6) return scene_tile_fwd<6>(frames, z, xw, n_obj, np, st, fm);
  if (n_obj <= 8) return scene_tile_fwd<8>(frames, z, xw, n_obj, np, st, fm);
  return (int)hipErrorInvalidValue;
}

static inline int nmax_of(int n_obj) { return n_obj <= 3 ? 3 : (n_obj <= 6 ? 6 : 8); }

// Tail of the scene backward: dL/d tile + transformer / mask backward in one pass over the leaf gradients `Dscr`
// (scene_pixtile_bwd_k), then the per-object sums with the background chain's dz_bg.
template <int NMAX>
static int scene_bwd_tail(const float* frames, const float* z, const float* xw, const float* Dscr, const int* leaf_slot,
                          const float* coef, const float* d_ovl, float* dzc, const float* dll, const float* obj_ll,
                          const float* dz_bg, float* dz, int n_obj, int np, hipStream_t st, hipStream_t bg_stream, FrameMap fm,
                          const float* d_obj, int bg_parts) {      // Dscr is at unit upstream gradient: times d_obj[patch] as it is staged
  int rc = scene_pixtile_bwd<NMAX>(frames, z, xw, Dscr, leaf_slot, coef, d_ovl, dzc, n_obj, np, st, fm, d_obj);
  if (rc) return rc;
  STOVE_TRY(stream_after(st, bg_stream));       // join: only the last kernel needs the background chain's dz_bg
  STOVE_LAUNCH((scene_finalize_bwd_k<NMAX>), dim3((np + 255) / 256), dim3(256), 0, st, dll, z, obj_ll, dz_bg, dzc, dz, n_obj, np, bg_parts);
  STOVE_LAUNCH_CHECK();
  return 0;
}

__global__ void fill_words_k(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

static inline size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

}  // namespace stove

using namespace stove;

extern "C" {

// 2: the GNN parameter image carries the LDS-order weight sections ([W | W^T | vectors | W packed | W^T packed],
//    stove_gnn_param_floats() floats); the measurement switches are explicit setters (stove_set_overlap,
//    ...) instead of environment reads; cross-capture events are owned by the caller
//    (stove_event_list_*).
// 3: stove_profile_report lines carry a fourth column, the time the kernel's launches cover.
// 4 (round 5): stove_gemm_bf16 takes nsplit = 3 (half pieces); the A/B entry points whose losing side is recorded are gone
//    (stove_set_tablegrad_placement, stove_lstm_cell_bwd_rows).
int stove_abi_version(void) { return 5; }

const char* stove_error_string(int code) { return hipGetErrorString((hipError_t)code); }

int stove_selftest_wave_sum(const float* in, float* out, int n_waves, void* stream) {
  if (n_waves % 4) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(wave_sum_test_k, dim3(n_waves / 4), dim3(256), 0, (hipStream_t)stream, in, out);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- object SPN operator
size_t stove_objspn_tile_floats(int n) { return (size_t)((n + 63) / 64) * kObjX; }

int stove_objspn_fwd(const StoveSpnTables* t, const float* inputs, const float* marg, float* xw, float* out, int n,
                     void* stream) {
  STOVE_VALIDATE(objspn_fwd(t, inputs, xw, out, n));
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int rc = objspn_tile_from_arrays(inputs, marg, xw, n, st);
  if (rc) return rc;
  return objspn_forward(xw, t->obj_scope, t->obj_coef, t->obj_wsum, t->obj_wroot, out, nullptr, n, st);
}

size_t stove_objspn_bwd_ws_bytes(int n) { return (objspn_bwd_ws_floats(n) + stove_objspn_tile_floats(n)) * sizeof(float); }

int stove_objspn_bwd(const StoveSpnTables* t, const float* marg, const float* xw, const float* out, const float* dout,
                     float* d_inputs, float* d_marg, StoveSpnTableGrads* g, void* ws, int n, void* stream) {
  STOVE_VALIDATE(objspn_bwd(t, marg, xw, out, dout, d_marg, g, ws, n));
  hipStream_t st = (hipStream_t)stream;
  float* dxw = (float*)ws;
  float* rest = dxw + stove_objspn_tile_floats(n);
  int rc = objspn_backward(xw, t->obj_scope, t->obj_leaf_slot, t->obj_coef, t->obj_wsum, t->obj_wroot, out, dout, dxw,
                           g->obj_coef, g->obj_wsum, g->obj_wroot, rest, n, st);
  if (rc) return rc;
  if (d_inputs == nullptr && d_marg == nullptr) return 0;
  if (d_marg != nullptr && marg == nullptr) return (int)hipErrorInvalidValue;
  return objspn_tile_to_arrays(dxw, marg, d_inputs, d_marg, n, st);
}

// ---------------------------------------------------------------- background SPN operator
size_t stove_bgspn_saved_floats(int n) { return bgspn_fwd_ws_floats(n); }

int stove_bgspn_fwd(const StoveSpnTables* t, const float* inputs, const float* marg, float* ell, float* out, int n,
                    void* stream) {
  STOVE_VALIDATE(bgspn_fwd(t, inputs, ell, out, n, 1024));
  if (n == 0) return 0;
  return bgspn_forward(inputs, marg, nullptr, 0, t->bg_side, t->bg_coef, t->bg_wroot, ell, out, n, (hipStream_t)stream);
}

size_t stove_bgspn_bwd_ws_bytes(int n) { return bgspn_bwd_ws_floats(n) * sizeof(float); }

int stove_bgspn_bwd(const StoveSpnTables* t, const float* inputs, const float* marg, const float* ell, const float* out,
                    const float* dout, float* d_inputs, float* d_marg, StoveSpnTableGrads* g, void* ws, int n,
                    void* stream) {
  STOVE_VALIDATE(bgspn_bwd(t, inputs, marg, ell, out, dout, d_marg, g, ws, n, 1024));
  return bgspn_backward(inputs, marg, nullptr, 0, t->bg_side, t->bg_coef, t->bg_wroot, ell, out, dout, d_inputs, d_marg,
                        nullptr, g->bg_coef, g->bg_wroot, (float*)ws, n, (hipStream_t)stream);
}

// the same operator for any number of input dimensions (frame sizes other than 32 x 32): csrc/spn_bg_generic.hip
size_t stove_bgspn_saved_floats_d(int n, int n_pix) { return bgspn_any_saved_floats(n, n_pix); }
int stove_bgspn_fwd_d(const StoveSpnTables* t, const float* inputs, const float* marg, float* ell, float* out, int n, int n_pix, void* stream) {
  STOVE_VALIDATE(bgspn_fwd(t, inputs, ell, out, n, n_pix));
  if (n == 0) return 0;
  return bgspn_any_forward(inputs, marg, t->bg_side, t->bg_coef, t->bg_wroot, ell, out, n, n_pix, (hipStream_t)stream);
}
size_t stove_bgspn_bwd_ws_bytes_d(int n, int n_pix) { return bgspn_any_bwd_ws_floats(n, n_pix) * sizeof(float); }
int stove_bgspn_bwd_d(const StoveSpnTables* t, const float* inputs, const float* marg, const float* ell, const float* out, const float* dout,
                      float* d_inputs, float* d_marg, StoveSpnTableGrads* g, void* ws, int n, int n_pix, void* stream) {
  STOVE_VALIDATE(bgspn_bwd(t, inputs, marg, ell, out, dout, d_marg, g, ws, n, n_pix));
  return bgspn_any_backward(inputs, marg, t->bg_side, t->bg_coef, t->bg_wroot, ell, out, dout, d_inputs, d_marg, g->bg_coef, g->bg_wroot,
                            (float*)ws, n, n_pix, (hipStream_t)stream);
}

// ---------------------------------------------------------------- fused scene likelihood
// saved = [ xw tile | obj_ll (np) | ovl (np) | bg_out (nf) | bg_ell | box coverage tables | object-SPN backward scratch at unit gradient ]
// (the last section only exists / is only written when the forward is asked for it: with_grad)
struct SceneSaved {
  size_t xw, obj_ll, ovl, bg_out, bg_ell, cover, obj_scratch, total;
};
static SceneSaved scene_saved_layout(int nf, int n_obj, bool with_grad = true) {
  const size_t np = (size_t)nf * n_obj;
  SceneSaved s;
  s.xw = 0;
  s.obj_ll = align64(stove_objspn_tile_floats((int)np));
  s.ovl = s.obj_ll + align64(np);
  s.bg_out = s.ovl + align64(np);
  s.bg_ell = s.bg_out + align64(nf);
  s.cover = s.bg_ell + align64(bgspn_fwd_ws_floats(nf));
  s.obj_scratch = s.cover + align64(bg_cover_floats(nf, n_obj));
  s.total = s.obj_scratch + (with_grad ? align64(objspn_scratch_floats((int)np)) : 0);
  return s;
}

size_t stove_scene_saved_floats(int n_frames, int n_obj) { return scene_saved_layout(n_frames, n_obj).total; }
size_t stove_scene_fwd_floats(int n_frames, int n_obj, int with_grad) { return scene_saved_layout(n_frames, n_obj, with_grad != 0).total; }

// Internal fork stream (one per device, created on first use): the background-SPN chain of a scene call runs on it next
// to the object-SPN chain -- they are independent until the assemble / tail kernels -- and is joined back before the call
// returns, so callers see plain single-stream semantics.  STOVE_NO_OVERLAP=1 keeps everything on the caller's stream.
static hipStream_t g_fork_user[16] = {nullptr};
static bool g_fork_user_set[16] = {false};
static std::mutex g_fork_mu;

static std::atomic<int> g_overlap{1};            // stove_set_overlap

static hipStream_t scene_fork_stream(hipStream_t st) {
  static hipStream_t side[16] = {nullptr};
  if (!g_overlap.load(std::memory_order_relaxed)) return st;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return st;
  std::lock_guard<std::mutex> lock(g_fork_mu);    // backward is called from the autograd thread, forward from the main one
  if (g_fork_user_set[dev]) return g_fork_user[dev] != nullptr ? g_fork_user[dev] : st;      // the caller's stream (NULL: no fork at all)
  if (side[dev] == nullptr && hipStreamCreateWithFlags(&side[dev], hipStreamNonBlocking) != hipSuccess) return st;
  return side[dev];
}

// Measurement switch (explicit state instead of an environment read; default 1).
// overlap 0: the scene calls run their background-SPN chain on the call's stream (no internal fork at all).
int stove_set_tile_lds(int mode) {
  if (mode < 0 || mode > 2) return (int)hipErrorInvalidValue;
  g_tile_lds.store(mode);
  return 0;
}

int stove_set_overlap(int on) {
  g_overlap.store(on != 0, std::memory_order_relaxed);
  return 0;
}
// The caller owns the fork stream of `device` from now on (the scene calls run their background-SPN chain on it, forked from and joined
// into the call's stream): stream != NULL -- use this one; NULL -- no internal fork, everything on the call's stream.
// restore_default != 0: back to the library-owned stream that is created on first use.
int stove_set_fork_stream(int device, void* stream, int restore_default) {
  if (device < 0 || device >= 16) return (int)hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_fork_mu);
  g_fork_user_set[device] = restore_default == 0;
  g_fork_user[device] = restore_default == 0 ? (hipStream_t)stream : nullptr;
  return 0;
}

static int frame_map(int n_frames, int seq_frames, int seq_stride, FrameMap* fm) {
  *fm = FrameMap{0, 0};
  if (seq_frames == 0) return 0;                                   // dense
  if (seq_frames < 0 || seq_stride < seq_frames || n_frames % seq_frames != 0) return (int)hipErrorInvalidValue;
  if (seq_stride != seq_frames) *fm = FrameMap{seq_frames, seq_stride};
  return 0;
}

int stove_scene_fwd(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                    int seq_stride, float overlap_beta, float* ll, float* parts, float* saved, void* stream) {
  return stove_scene_fwd_from(t, frames, z, n_frames, n_obj, seq_frames, seq_stride, overlap_beta, ll, parts, saved, stream, stream, 1);
}

// fork_from: the stream the internal background-SPN chain is ordered behind (its inputs -- frames, z, tables -- must be ready there).
// Normally `stream` itself.  A caller that runs the call on a stream which is itself a fork of a stream under hipGraph capture passes
// that capture's ORIGIN stream: the HIP 7.0 runtime cannot end a capture in which two forked streams wait on each other
// (fork from X, join into X, with X not the origin: hip::Stream::EndCapture recurses forever; tools/ubench/graph_ext2.hip).
// with_grad != 0: the object SPN runs forward + backward at unit upstream gradient in one pass (objspn_fwd_unit_k) and leaves
// the per-glimpse backward scratch in `saved` (stove_scene_fwd_floats(.., 1) = stove_scene_saved_floats floats): what
// stove_scene_bwd* needs.  0: likelihood only, `saved` of stove_scene_fwd_floats(.., 0) floats, no backward from it.
int stove_scene_fwd_from(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                         int seq_stride, float overlap_beta, float* ll, float* parts, float* saved, void* stream, void* fork_from,
                         int with_grad) {
  STOVE_VALIDATE(scene_fwd(t, frames, z, n_frames, n_obj, seq_frames, seq_stride, ll, saved));
  hipStream_t st = (hipStream_t)stream;
  hipStream_t root = fork_from != nullptr ? (hipStream_t)fork_from : st;
  if (n_frames == 0) return 0;
  FrameMap fm;
  if (frame_map(n_frames, seq_frames, seq_stride, &fm)) return (int)hipErrorInvalidValue;
  const SceneSaved L = scene_saved_layout(n_frames, n_obj);
  const int np = n_frames * n_obj;
  hipStream_t sb = scene_fork_stream(st);       // background chain (MFMA-bound) next to the object chain (VALU-bound)
  STOVE_TRY(stream_after(sb, root));            // fork: inputs are ready in `root` order
  JoinGuard jb(st, sb);                         // joined on every exit path
  int rc = scene_tile_fwd_any(frames, z, saved + L.xw, n_obj, np, st, fm);
  if (rc) return rc;
  if (with_grad)
    rc = objspn_forward_unit(saved + L.xw, t->obj_scope, t->obj_coef, t->obj_wsum, t->obj_wroot, saved + L.obj_ll, saved + L.ovl,
                             saved + L.obj_scratch, np, st);
  else
    rc = objspn_forward(saved + L.xw, t->obj_scope, t->obj_coef, t->obj_wsum, t->obj_wroot, saved + L.obj_ll, saved + L.ovl, np, st);
  if (rc) return rc;
  rc = bgspn_forward(frames, nullptr, z, n_obj, t->bg_side, t->bg_coef, t->bg_wroot, saved + L.bg_ell, saved + L.bg_out, n_frames, sb, fm, t->bg_dense);
  if (rc) return rc;
  rc = bg_cover_tables(z, saved + L.cover, n_frames, n_obj, sb);       // for the backward of this z
  if (rc) return rc;
  STOVE_TRY(jb.join());
  STOVE_LAUNCH(scene_assemble_fwd_k, dim3((n_frames + 255) / 256), dim3(256), 0, st, saved + L.bg_out, saved + L.obj_ll,
                     saved + L.ovl, z, ll, parts, n_obj, n_frames, overlap_beta, logf(overlap_beta));
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ws = [ d_obj (np) | d_ovl (np) | dzc (np*NMAX*4) | dz_bg (np*4) | obj table-gradient chunk partials | bg ws ]
struct SceneWs {
  size_t d_obj, d_ovl, dzc, dz_bg, obj, bg, total;
};
static SceneWs scene_ws_layout(int nf, int n_obj) {
  const size_t np = (size_t)nf * n_obj;
  SceneWs s;
  s.d_obj = 0;
  s.d_ovl = s.d_obj + align64(np);
  s.dzc = s.d_ovl + align64(np);
  s.dz_bg = s.dzc + align64(np * nmax_of(n_obj) * 4);
  s.obj = s.dz_bg + align64(np * 4);
  s.bg = s.obj + align64(objspn_partial_floats());
  s.total = s.bg + align64(bgspn_bwd_ws_floats(nf, n_obj));
  return s;
}

size_t stove_scene_bwd_ws_bytes(int n_frames, int n_obj) { return scene_ws_layout(n_frames, n_obj).total * sizeof(float); }

constexpr int kLateTableGradGlimpses = 16384;
// Where the object-SPN table gradients run when the caller gives a parameter stream: with up to four objects they are held back
// until dz is out and then run underneath the recursion's backward as objspn_tablegrad_under_k (one wave per SIMD in the registers
// and LDS that kernel leaves free).  Measured in round 2 (B = 256): the SPN backward phase 600 -> 495 us without them, the
// recursion's backward 470 -> 545 us with them on its SIMDs, the step 3.273 -> 3.254 ms.  (The round-1 placement -- right behind
// their producer, next to pix / bgspn_bwd -- was a switch until round 5.)

size_t stove_bg_dense_floats(void) { return (size_t)kBgDenseF; }
int stove_bg_dense(const int32_t* bg_side, const float* bg_coef, float* dense, void* stream) {
  STOVE_LAUNCH(bg_dense_fwd_k, dim3((kBgDenseF + 255) / 256), dim3(256), 0, (hipStream_t)stream, bg_side, bg_coef, dense);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_scene_bwd(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                    int seq_stride, float overlap_beta, const float* saved, const float* dll, float* dz, StoveSpnTableGrads* g,
                    void* ws_, void* stream) {
  return stove_scene_bwd_overlap(t, frames, z, n_frames, n_obj, seq_frames, seq_stride, overlap_beta, saved, dll, dz, g, ws_, stream, stream);
}

int stove_scene_bwd_overlap(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                            int seq_stride, float overlap_beta, const float* saved, const float* dll, float* dz,
                            StoveSpnTableGrads* g, void* ws_, void* stream, void* param_stream) {
  return stove_scene_bwd_from(t, frames, z, n_frames, n_obj, seq_frames, seq_stride, overlap_beta, saved, dll, dz, g, ws_, stream, param_stream, stream);
}

// fork_from: see stove_scene_fwd_from (the background chain's inputs -- frames, z, saved, dll -- must be ready in its order)
int stove_scene_bwd_from(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                         int seq_stride, float overlap_beta, const float* saved, const float* dll, float* dz,
                         StoveSpnTableGrads* g, void* ws_, void* stream, void* param_stream, void* fork_from) {
  STOVE_VALIDATE(scene_bwd(t, frames, z, n_frames, n_obj, seq_frames, seq_stride, saved, dll, dz, g, ws_));
  hipStream_t st = (hipStream_t)stream;
  hipStream_t root = fork_from != nullptr ? (hipStream_t)fork_from : st;
  hipStream_t sp = param_stream != nullptr ? (hipStream_t)param_stream : st;
  if (n_frames == 0) return 0;
  FrameMap fm;
  if (frame_map(n_frames, seq_frames, seq_stride, &fm)) return (int)hipErrorInvalidValue;
  float* ws = (float*)ws_;
  const SceneSaved L = scene_saved_layout(n_frames, n_obj);
  const SceneWs W = scene_ws_layout(n_frames, n_obj);
  const int np = n_frames * n_obj;
  STOVE_LAUNCH(scene_assemble_bwd_k, dim3((np + 255) / 256), dim3(256), 0, st, dll, z, ws + W.d_obj, ws + W.d_ovl, n_obj, np, overlap_beta);
  STOVE_LAUNCH_CHECK();
  hipStream_t sb = scene_fork_stream(st);       // background chain next to the object chain, joined before the tail
  STOVE_TRY(stream_after(sb, root));
  JoinGuard jb(st, sb);                         // error paths: the tail below is what joins `sb` normally
  JoinGuard jp(st, sp);                         // error paths only: the caller joins the parameter stream after a clean return
  // The object SPN's backward already ran, at unit upstream gradient, inside the forward (objspn_fwd_unit_k): what is left is to
  // apply d_obj[patch] where its scratch is consumed -- the tail below and the table gradients.
  int rc = 0;
  // The object-SPN table gradients go to the parameter stream once dz is out (underneath what the caller enqueues next, the
  // recursion's backward) ...
  // ... for up to four objects: that recursion (dyn_loop_bwd_small_k<4, ..>: 253 + 58 registers, one wave per SIMD on every CU at
  // B = 256) is the latency-bound kernel with room beside it.  The five / six-object instantiation holds 256 + 192 of a SIMD lane's
  // 512 registers (hipcc -Rpass-analysis=kernel-resource-usage, round 5), so nothing wider than 64 registers can run beside it --
  // the 150-register table-gradient kernel would queue up behind it instead of under it -- and the seven / eight-object recursion
  // (gnn.hip) fills the matrix pipe itself (stretched 2.49 -> 3.20 ms by a co-runner, round 2)
  // ... and for a batch large enough that the recursion's backward is long (98 steps): at the reference's default training shape
  // (256 clips x 8 frames: 6 912 glimpses, a 35 us recursion) the held-back table gradients would be the head of a 260 us chain of
  // small launches on the parameter stream that outlasts the main stream by 110 us -- there they start at once, beside the data half
  const bool late = sp != st && n_obj <= 4 && np >= kLateTableGradGlimpses;
  if (!late) {
    STOVE_TRY(stream_after(sp, st));
    rc = objspn_backward_params(saved + L.xw, t->obj_scope, g->obj_coef, g->obj_wsum, g->obj_wroot, saved + L.obj_scratch, ws + W.obj,
                                ws + W.d_obj, np, sp);
    if (rc) return rc;
  }
  // (dz = null: the per-half partial images of dz_bg stay where bgspn_bwd_k wrote them; the tail's last kernel sums them)
  rc = bgspn_backward(frames, nullptr, z, n_obj, t->bg_side, t->bg_coef, t->bg_wroot, saved + L.bg_ell, saved + L.bg_out, dll,
                      nullptr, nullptr, nullptr, g->bg_coef, g->bg_wroot, ws + W.bg, n_frames, sb, sp == st ? sb : sp, fm, saved + L.cover);
  if (rc) return rc;
  const float* dz_bg_parts = bgspn_dz_parts(ws + W.bg, n_frames);
  // the tail joins `sb` before its last kernel (dz_bg; without a parameter stream also the bg table grads)
  if (n_obj <= 3)
    rc = scene_bwd_tail<3>(frames, z, saved + L.xw, saved + L.obj_scratch, t->obj_leaf_slot, t->obj_coef, ws + W.d_ovl, ws + W.dzc, dll, saved + L.obj_ll,
                           dz_bg_parts, dz, n_obj, np, st, sb, fm, ws + W.d_obj, kBgHalves);
  else if (n_obj <= 6)
    rc = scene_bwd_tail<6>(frames, z, saved + L.xw, saved + L.obj_scratch, t->obj_leaf_slot, t->obj_coef, ws + W.d_ovl, ws + W.dzc, dll, saved + L.obj_ll,
                           dz_bg_parts, dz, n_obj, np, st, sb, fm, ws + W.d_obj, kBgHalves);
  else if (n_obj <= 8)
    rc = scene_bwd_tail<8>(frames, z, saved + L.xw, saved + L.obj_scratch, t->obj_leaf_slot, t->obj_coef, ws + W.d_ovl, ws + W.dzc, dll, saved + L.obj_ll,
                           dz_bg_parts, dz, n_obj, np, st, sb, fm, ws + W.d_obj, kBgHalves);
  else
    rc = (int)hipErrorInvalidValue;
  if (rc) return rc;
  jb.dismiss();                                 // scene_bwd_tail joined `sb`
  if (late) {
    STOVE_TRY(stream_after(sp, st));
    rc = objspn_backward_params(saved + L.xw, t->obj_scope, g->obj_coef, g->obj_wsum, g->obj_wroot, saved + L.obj_scratch, ws + W.obj,
                                ws + W.d_obj, np, sp, true);
    if (rc) return rc;
  }
  jp.dismiss();
  return 0;
}

// ---------------------------------------------------------------- object RAT-SPN operator, any glimpse size / vector widths
static ObjAnyShape obj_any_shape(int R, int G, int S, int D, int Lmax) {
  ObjAnyShape sh;
  sh.R = R; sh.G = G; sh.S = S; sh.D = D; sh.Lmax = Lmax;
  return sh;
}
size_t stove_objspn_saved_floats_any(int n, int R, int G, int S, int D, int Lmax) { return objany_saved_floats(n, obj_any_shape(R, G, S, D, Lmax)); }
size_t stove_objspn_bwd_ws_bytes_any(int n, int R, int G, int S, int D, int Lmax) { return objany_bwd_ws_floats(n, obj_any_shape(R, G, S, D, Lmax)) * sizeof(float); }
int stove_objspn_fwd_any(const float* inputs, const float* marg, const int* lscope, const float* coef, const float* wsum, const float* wroot,
                         float* saved, float* out, int n, int R, int G, int S, int D, int Lmax, void* stream) {
  return objany_forward(inputs, marg, lscope, coef, wsum, wroot, saved, out, n, obj_any_shape(R, G, S, D, Lmax), (hipStream_t)stream);
}
int stove_objspn_bwd_any(const float* inputs, const float* marg, const int* lscope, const int* slot, const float* coef, const float* wsum,
                         const float* wroot, const float* saved, const float* dout, float* d_inputs, float* d_marg, float* g_coef,
                         float* g_wsum, float* g_wroot, void* ws, int n, int R, int G, int S, int D, int Lmax, void* stream) {
  return objany_backward(inputs, marg, lscope, slot, coef, wsum, wroot, saved, dout, d_inputs, d_marg, g_coef, g_wsum, g_wroot, (float*)ws, n,
                         obj_any_shape(R, G, S, D, Lmax), (hipStream_t)stream);
}

int stove_gauss_ll_fwd(const float* x, const float* marg, float* out, int n, int d, float mean, float scale, void* stream) {
  if (n == 0) return 0;
  if (d < 1 || !(scale > 0.0f) || x == nullptr || marg == nullptr || out == nullptr) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(gauss_ll_fwd_k, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, marg, out, n, d, mean, scale);
  STOVE_LAUNCH_CHECK();
  return 0;
}
int stove_gauss_ll_bwd(const float* x, const float* marg, const float* dout, float* dx, float* dm, int n, int d, float mean, float scale,
                       void* stream) {
  if (n == 0) return 0;
  if (d < 1 || !(scale > 0.0f) || x == nullptr || marg == nullptr || dout == nullptr) return (int)hipErrorInvalidValue;
  const size_t tot = (size_t)n * d;
  STOVE_LAUNCH(gauss_ll_bwd_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, marg, dout, dx, dm, n, d, mean, scale);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- fused scene likelihood, any frame size / sampling convention
// The object side is the 32 x 32 path's own kernels with the geometry at run time (scene_tile_fwd_k / scene_pixtile_bwd_k <.., ANY>,
// objspn_fwd_unit_k, the table-gradient kernels); the background side is the mask in closed form (bg_mask_any_k), the general-size
// operator of spn_bg_generic.hip, and the mask's backward to z (bg_mask_bwd_any_k).
// saved = [ xw tile | obj_ll | ovl | bg_out | bg_ell (n x halves x 36) | mask (n x n_pix) | object-SPN scratch at unit gradient ]
struct SceneSavedAny {
  size_t xw, obj_ll, ovl, bg_out, bg_ell, mask, obj_scratch, total;
};
// frames up to kBgTabMax a side: the background kernels form the mask themselves from per-frame coverage tables (spn_bg_generic.hip), no
// mask image is kept
static bool scene_any_inline(int W, int H) { return W <= kBgTabMax && H <= kBgTabMax; }
static SceneSavedAny scene_saved_layout_any(int nf, int n_obj, int n_pix, bool with_grad, bool inline_mask = false) {
  const size_t np = (size_t)nf * n_obj;
  SceneSavedAny s;
  s.xw = 0;
  s.obj_ll = align64(stove_objspn_tile_floats((int)np));
  s.ovl = s.obj_ll + align64(np);
  s.bg_out = s.ovl + align64(np);
  s.bg_ell = s.bg_out + align64(nf);
  s.mask = s.bg_ell + align64(bgspn_any_saved_floats(nf, n_pix));
  s.obj_scratch = s.mask + (inline_mask ? 0 : align64((size_t)nf * n_pix));
  s.total = s.obj_scratch + (with_grad ? align64(objspn_scratch_floats((int)np)) : 0);
  return s;
}
struct SceneWsAny {
  size_t d_obj, d_ovl, dzc, dz_bg, obj, d_mask, bg, total;
};
static SceneWsAny scene_ws_layout_any(int nf, int n_obj, int n_pix) {
  const size_t np = (size_t)nf * n_obj;
  SceneWsAny s;
  s.d_obj = 0;
  s.d_ovl = s.d_obj + align64(np);
  s.dzc = s.d_ovl + align64(np);
  s.dz_bg = s.dzc + align64(np * nmax_of(n_obj) * 4);
  s.obj = s.dz_bg + align64(np * 4);
  s.d_mask = s.obj + align64(objspn_partial_floats());
  s.bg = s.d_mask + align64((size_t)nf * n_pix);
  s.total = s.bg + align64(bgspn_any_bwd_ws_floats(nf, n_pix));
  return s;
}
size_t stove_scene_saved_floats_any(int n_frames, int n_obj, int n_pix, int with_grad) { return scene_saved_layout_any(n_frames, n_obj, n_pix, with_grad != 0).total; }
size_t stove_scene_bwd_ws_bytes_any(int n_frames, int n_obj, int n_pix) { return scene_ws_layout_any(n_frames, n_obj, n_pix).total * sizeof(float); }

int stove_scene_fwd_any(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames, int seq_stride,
                        int W, int H, int align_corners, float overlap_beta, float* ll, float* parts, float* saved, void* stream, int with_grad) {
  STOVE_VALIDATE(scene_fwd(t, frames, z, n_frames, n_obj, seq_frames, seq_stride, ll, saved));
  hipStream_t st = (hipStream_t)stream;
  if (n_frames == 0) return 0;
  if (W < 2 || H < 2 || n_obj < 1 || n_obj > 8) return (int)hipErrorInvalidValue;
  FrameMap fm;
  if (frame_map(n_frames, seq_frames, seq_stride, &fm)) return (int)hipErrorInvalidValue;
  const int n_pix = W * H, np = n_frames * n_obj;
  const SceneGeom gm = scene_geom(W, H, align_corners);
  const bool inl = scene_any_inline(W, H);
  const SceneSavedAny L = scene_saved_layout_any(n_frames, n_obj, n_pix, true, inl);
  SceneBoxes boxes;
  boxes.z = inl ? z : nullptr; boxes.n_obj = n_obj; boxes.gm = gm;
  hipStream_t sb = scene_fork_stream(st);
  STOVE_TRY(stream_after(sb, st));
  JoinGuard jb(st, sb);
  int rc = n_obj <= 3 ? scene_tile_fwd_g<3>(frames, z, saved + L.xw, n_obj, np, st, fm, gm)
                      : (n_obj <= 6 ? scene_tile_fwd_g<6>(frames, z, saved + L.xw, n_obj, np, st, fm, gm) : scene_tile_fwd_g<8>(frames, z, saved + L.xw, n_obj, np, st, fm, gm));
  if (rc) return rc;
  if (with_grad)
    rc = objspn_forward_unit(saved + L.xw, t->obj_scope, t->obj_coef, t->obj_wsum, t->obj_wroot, saved + L.obj_ll, saved + L.ovl,
                             saved + L.obj_scratch, np, st);
  else
    rc = objspn_forward(saved + L.xw, t->obj_scope, t->obj_coef, t->obj_wsum, t->obj_wroot, saved + L.obj_ll, saved + L.ovl, np, st);
  if (rc) return rc;
  if (!inl) {
    const size_t tot = (size_t)n_frames * n_pix;
    STOVE_LAUNCH(bg_mask_any_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, sb, z, saved + L.mask, n_frames, n_obj, gm);
    STOVE_LAUNCH_CHECK();
  }
  rc = bgspn_any_forward(frames, inl ? nullptr : saved + L.mask, t->bg_side, t->bg_coef, t->bg_wroot, saved + L.bg_ell, saved + L.bg_out, n_frames,
                         n_pix, sb, fm, boxes);
  if (rc) return rc;
  STOVE_TRY(jb.join());
  STOVE_LAUNCH(scene_assemble_fwd_k, dim3((n_frames + 255) / 256), dim3(256), 0, st, saved + L.bg_out, saved + L.obj_ll,
               saved + L.ovl, z, ll, parts, n_obj, n_frames, overlap_beta, logf(overlap_beta));
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_scene_bwd_any(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames, int seq_stride,
                        int W, int H, int align_corners, float overlap_beta, const float* saved, const float* dll, float* dz,
                        StoveSpnTableGrads* g, void* ws_, void* stream, void* param_stream) {
  STOVE_VALIDATE(scene_bwd(t, frames, z, n_frames, n_obj, seq_frames, seq_stride, saved, dll, dz, g, ws_));
  hipStream_t st = (hipStream_t)stream;
  hipStream_t sp = param_stream != nullptr ? (hipStream_t)param_stream : st;
  if (n_frames == 0) return 0;
  if (W < 2 || H < 2 || n_obj < 1 || n_obj > 8) return (int)hipErrorInvalidValue;
  FrameMap fm;
  if (frame_map(n_frames, seq_frames, seq_stride, &fm)) return (int)hipErrorInvalidValue;
  float* ws = (float*)ws_;
  const int n_pix = W * H, np = n_frames * n_obj;
  const SceneGeom gm = scene_geom(W, H, align_corners);
  const bool inl = scene_any_inline(W, H);
  const SceneSavedAny L = scene_saved_layout_any(n_frames, n_obj, n_pix, true, inl);
  const SceneWsAny Wl = scene_ws_layout_any(n_frames, n_obj, n_pix);
  SceneBoxes boxes;
  boxes.z = inl ? z : nullptr; boxes.n_obj = n_obj; boxes.gm = gm;
  STOVE_LAUNCH(scene_assemble_bwd_k, dim3((np + 255) / 256), dim3(256), 0, st, dll, z, ws + Wl.d_obj, ws + Wl.d_ovl, n_obj, np, overlap_beta);
  STOVE_LAUNCH_CHECK();
  hipStream_t sb = scene_fork_stream(st);
  STOVE_TRY(stream_after(sb, st));
  JoinGuard jb(st, sb);
  JoinGuard jp(st, sp);
  // background chain: operator backward (d mask, table gradients), then the mask's backward to z
  int rc = bgspn_any_backward(frames, inl ? nullptr : saved + L.mask, t->bg_side, t->bg_coef, t->bg_wroot, saved + L.bg_ell, saved + L.bg_out, dll,
                              nullptr, ws + Wl.d_mask, g->bg_coef, g->bg_wroot, ws + Wl.bg, n_frames, n_pix, sb, fm, boxes);
  if (rc) return rc;
  if (n_obj <= 3) STOVE_LAUNCH((bg_mask_bwd_any_k<3>), dim3(n_frames), dim3(256), 0, sb, z, (const float*)(ws + Wl.d_mask), ws + Wl.dz_bg, n_frames, n_obj, gm);
  else STOVE_LAUNCH((bg_mask_bwd_any_k<8>), dim3(n_frames), dim3(256), 0, sb, z, (const float*)(ws + Wl.d_mask), ws + Wl.dz_bg, n_frames, n_obj, gm);
  STOVE_LAUNCH_CHECK();
  // object chain: pixel / transformer backward from the unit-gradient scratch, then the per-object sums with dz_bg
  const float* d_obj = ws + Wl.d_obj;
  if (n_obj <= 3)
    rc = scene_pixtile_bwd<3, true>(frames, z, saved + L.xw, saved + L.obj_scratch, t->obj_leaf_slot, t->obj_coef, ws + Wl.d_ovl, ws + Wl.dzc, n_obj, np, st, fm, d_obj, gm);
  else if (n_obj <= 6)
    rc = scene_pixtile_bwd<6, true>(frames, z, saved + L.xw, saved + L.obj_scratch, t->obj_leaf_slot, t->obj_coef, ws + Wl.d_ovl, ws + Wl.dzc, n_obj, np, st, fm, d_obj, gm);
  else
    rc = scene_pixtile_bwd<8, true>(frames, z, saved + L.xw, saved + L.obj_scratch, t->obj_leaf_slot, t->obj_coef, ws + Wl.d_ovl, ws + Wl.dzc, n_obj, np, st, fm, d_obj, gm);
  if (rc) return rc;
  STOVE_TRY(jb.join());
  if (n_obj <= 3) STOVE_LAUNCH((scene_finalize_bwd_k<3>), dim3((np + 255) / 256), dim3(256), 0, st, dll, z, saved + L.obj_ll, ws + Wl.dz_bg, ws + Wl.dzc, dz, n_obj, np, 1);
  else if (n_obj <= 6) STOVE_LAUNCH((scene_finalize_bwd_k<6>), dim3((np + 255) / 256), dim3(256), 0, st, dll, z, saved + L.obj_ll, ws + Wl.dz_bg, ws + Wl.dzc, dz, n_obj, np, 1);
  else STOVE_LAUNCH((scene_finalize_bwd_k<8>), dim3((np + 255) / 256), dim3(256), 0, st, dll, z, saved + L.obj_ll, ws + Wl.dz_bg, ws + Wl.dzc, dz, n_obj, np, 1);
  STOVE_LAUNCH_CHECK();
  // table gradients of the object SPN on the parameter stream (the background's are complete in `st` order: ordered into it as well)
  STOVE_TRY(stream_after(sp, st));
  rc = objspn_backward_params(saved + L.xw, t->obj_scope, g->obj_coef, g->obj_wsum, g->obj_wroot, saved + L.obj_scratch, ws + Wl.obj, ws + Wl.d_obj, np, sp);
  if (rc) return rc;
  jp.dismiss();
  return 0;
}

int stove_glimpse_mean(const float* x_color, const float* z, float* emb, int n_frames, int n_obj, int channels, void* stream) {
  const long long total = (long long)n_frames * n_obj;
  if (total == 0) return 0;
  if (total > 0x7fffffffLL || channels < 1 || channels > 4) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(glimpse_mean_k, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x_color, z, emb,
               n_frames * n_obj, n_obj, channels);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_objspn_mpe(const StoveSpnTables* t, const float* leaf_means, const float* inputs, float* xw, float* out,
                     int32_t* pick, int n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = objspn_tile_from_arrays(inputs, nullptr, xw, n, st);
  if (rc) return rc;
  return objspn_mpe(xw, t->obj_scope, t->obj_coef, t->obj_wsum, t->obj_wroot, leaf_means, out, pick, n, st);
}

int stove_render_frames(const float* bg, const float* patches, int frames_per_patch, const float* z, float* out, int n_frames,
                        int n_obj, void* stream) {
  const long long total = (long long)n_frames * 1024;
  if (total == 0) return 0;
  if (frames_per_patch < 0 || n_obj < 0 || (total + 255) / 256 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(render_frames_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bg, patches,
               frames_per_patch, z, out, n_frames, n_obj);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_scene_glimpses(const float* frames, const float* z, int n_frames, int n_obj, float* tile, float* patches,
                         float* keep, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int np = n_frames * n_obj;
  if (np == 0) return 0;
  int rc = scene_tile_fwd_any(frames, z, tile, n_obj, np, st);
  if (rc) return rc;
  const int nb = (np + 63) / 64;
  STOVE_LAUNCH(tile_unpack_k, dim3(nb < 2048 ? nb * 25 : 2048 * 25), dim3(256), 0, st, tile, patches, keep, np, nb);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- GNN dynamics core
static int gnn_lds_attr(const void* fn) {
  return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGnnLdsFloats * sizeof(float)));
}

size_t stove_gnn_param_floats(void) { return kGnnParams; }
size_t stove_gnn_grad_floats(void) { return kGnnGrads; }
int stove_gnn_blocks(int B, int N) {
  const int g = gnn_group_for(B, N);
  return (B + g - 1) / g;
}

int stove_gnn_fwd(const float* s_in, const float* params, float* result, float* pred, int B, int N, int sin_dim,
                  int lim_enc, int elu, void* stream) {
  STOVE_VALIDATE(gnn_fwd(s_in, params, result, B, N, sin_dim));
  if (B == 0) return 0;
  int rc = gnn_lds_attr((const void*)gnn_step_fwd_k);
  if (rc) return rc;
  STOVE_LAUNCH(gnn_step_fwd_k, dim3(stove_gnn_blocks(B, N)), dim3(256), kGnnLdsFloats * sizeof(float), (hipStream_t)stream,
                     s_in, params, result, pred, B, N, gnn_group_for(B, N), sin_dim, lim_enc, elu);
  STOVE_LAUNCH_CHECK();
  return 0;
}

size_t stove_gnn_bwd_ws_bytes(int B, int N) { return (size_t)stove_gnn_blocks(B, N) * kGnnGrads * sizeof(float); }

int stove_gnn_bwd(const float* s_in, const float* params, const float* d_result, const float* d_pred, float* d_s_in,
                  float* g_params, void* ws, int B, int N, int sin_dim, int lim_enc, int elu, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (g_params == nullptr) return (int)hipErrorInvalidValue;
  STOVE_VALIDATE(gnn_bwd(s_in, params, d_result, d_s_in, g_params, ws, B, N, sin_dim));
  if (B == 0) {
    hipMemsetAsync(g_params, 0, kGnnGrads * sizeof(float), st);
    return 0;
  }
  if (N < 1 || N > 8 || sin_dim < 16 || sin_dim > 32) return (int)hipErrorInvalidValue;
  int rc = gnn_lds_attr((const void*)gnn_step_bwd_k);
  if (rc) return rc;
  const int nb = stove_gnn_blocks(B, N);
  STOVE_LAUNCH(gnn_step_bwd_k, dim3(nb), dim3(256), kGnnLdsFloats * sizeof(float), st, s_in, params, d_result, d_pred,
                     d_s_in, (float*)ws, B, N, gnn_group_for(B, N), sin_dim, lim_enc, elu, (long long*)nullptr);
  STOVE_LAUNCH_CHECK();
  STOVE_LAUNCH(reduce_chunks_k, dim3((kGnnGrads + 31) / 32), dim3(256), 0, st, (const float*)ws, g_params, kGnnGrads, nb, 0);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// debug: one fwd+bwd step with per-stage cycle stamps of block 0 (stamps: 64 int64 on the device)
int stove_gnn_debug_stamps(const float* s_in, const float* params, const float* d_result, float* d_s_in, void* ws,
                           long long* stamps, int B, int N, int sin_dim, int lim_enc, int elu, void* stream) {
  int rc = gnn_lds_attr((const void*)gnn_step_bwd_k);
  if (rc) return rc;
  STOVE_LAUNCH(gnn_step_bwd_k, dim3(stove_gnn_blocks(B, N)), dim3(256), kGnnLdsFloats * sizeof(float), (hipStream_t)stream, s_in, params,
               d_result, (const float*)nullptr, d_s_in, (float*)ws, B, N, gnn_group_for(B, N), sin_dim, lim_enc, elu, stamps);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// debug: device buffer ([4 waves][16] int64) that the small-graph time loops stamp their phases into (last step,
// workgroup 0); null = off.  Set by the measurement tools only.
static long long* g_sm_stamps = nullptr;
void stove_debug_set_stamps(long long* device_buffer) { g_sm_stamps = device_buffer; }

// The small-graph recursion (gnn_small*.hip): kernels built for up to four objects (one node row per wave) and for up to six (two).
static bool small_graph(int N) { return N >= 2 && N <= 6; }

size_t stove_dynloop_act_floats(int B, int Ts, int N) {
  const int g = gnn_group_for(B, N);
  const size_t blockwise = (size_t)stove_gnn_blocks(B, N) * Ts * gnn_act_floats(N, g);
  if (small_graph(N)) {
    const size_t streams = (size_t)B * sm_act2_floats(N, Ts);
    return streams > blockwise ? streams : blockwise;
  }
  return blockwise;
}

// the recursion over the steps [ts0, ts1) (the whole range from the entry points; the kernels keep the piece interface of the
// round-3 pipelining experiment, docs/experiments/r06_pipeline_pieces_removed.patch)
static int dynloop_fwd_range(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                            const float* params, float* z, float* zdyn, float* zdstd, float* mean, float* std_, float* pred, float* act,
                            int B, int Ts, int N, int sin_dim, int lim_enc, int elu, float pos_var, float vel_std, float lat_std,
                            int ts0, int ts1, void* stream);
static int dynloop_bwd_range(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                            const float* params, const float* z, const float* act, const float* dz, const float* dzdyn,
                            const float* dmean, const float* dstd, const float* dpred, float* dz1, float* dzsup, float* dzsstd,
                            float* dextra, float* g_params, void* ws, int B, int Ts, int N, int sin_dim, int lim_enc, int elu,
                            float pos_var, float vel_std, float lat_std, int ts0, int ts1, float* carry, void* stream, void* param_stream);
int stove_dynloop_fwd(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                      const float* params, float* z, float* zdyn, float* zdstd, float* mean, float* std_, float* pred, float* act,
                      int B, int Ts, int N, int sin_dim, int lim_enc, int elu, float pos_var, float vel_std, float lat_std,
                      void* stream) {
  return dynloop_fwd_range(z1, zsup, zsstd, eps, extra, params, z, zdyn, zdstd, mean, std_, pred, act, B, Ts, N, sin_dim, lim_enc, elu,
                                 pos_var, vel_std, lat_std, 0, Ts, stream);
}


static int dynloop_fwd_range(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                            const float* params, float* z, float* zdyn, float* zdstd, float* mean, float* std_, float* pred, float* act,
                            int B, int Ts, int N, int sin_dim, int lim_enc, int elu, float pos_var, float vel_std, float lat_std,
                            int ts0, int ts1, void* stream) {
  STOVE_VALIDATE(dynloop_fwd(z1, zsup, zsstd, eps, extra, params, z, zdyn, zdstd, mean, std_, B, Ts, N, sin_dim));
  if (B == 0 || Ts == 0) return 0;
  if (ts0 < 0 || ts1 > Ts || ts0 >= ts1) return (int)hipErrorInvalidValue;
  if ((ts0 != 0 || ts1 != Ts) && !small_graph(N)) return (int)hipErrorInvalidValue;      // pieces: small-graph kernels only
  if (N < 1 || N > 8 || sin_dim < 16 || sin_dim > 32 || (sin_dim > 16 && extra == nullptr)) return (int)hipErrorInvalidValue;
  LoopConst kc{pos_var, vel_std, lat_std};
  if (small_graph(N)) {      // small graphs: (half-)wave-per-node-row formulation, two barriers per step (gnn_small.hip)
#define STOVE_LOOP_LAUNCH_N(SAVE_, NMX_, ELU_, NT_)                                                                             \
  do {                                                                                                                          \
    int rc = (int)hipFuncSetAttribute((const void*)dyn_loop_fwd_small_k<SAVE_, NMX_, ELU_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(SmShape<NMX_>::kLdsFloats * sizeof(float)));                                        \
    if (rc) return rc;                                                                                                          \
    STOVE_LAUNCH((dyn_loop_fwd_small_k<SAVE_, NMX_, ELU_, NT_>), dim3(B), dim3(64 * kSmWaves), SmShape<NMX_>::kLdsFloats * sizeof(float), \
                 (hipStream_t)stream, z1, zsup, zsstd, eps, extra, params, z, zdyn, zdstd, mean, std_, pred, act, B, Ts, N, sin_dim, lim_enc, \
                 elu, kc, g_sm_stamps, ts0, ts1);                                                                               \
  } while (0)
#define STOVE_LOOP_LAUNCH_E(SAVE_, ELU_)                          \
  do {                                                            \
    if (N == 3) STOVE_LOOP_LAUNCH_N(SAVE_, 4, ELU_, 3);           \
    else if (N <= 4) STOVE_LOOP_LAUNCH_N(SAVE_, 4, ELU_, 0);      \
    else if (N == 6) STOVE_LOOP_LAUNCH_N(SAVE_, 6, ELU_, 6);      \
    else STOVE_LOOP_LAUNCH_N(SAVE_, 6, ELU_, 0);                  \
  } while (0)
#define STOVE_LOOP_LAUNCH(SAVE_)                      \
  do {                                                \
    if (elu) STOVE_LOOP_LAUNCH_E(SAVE_, true);        \
    else STOVE_LOOP_LAUNCH_E(SAVE_, false);           \
  } while (0)
    if (g_sm_stamps != nullptr && act != nullptr && !elu && N == 3) {      // tools/loop_stamps.py
      int rc = (int)hipFuncSetAttribute((const void*)dyn_loop_fwd_small_k<2, 4, false, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(SmShape<4>::kLdsFloats * sizeof(float)));
      if (rc) return rc;
      STOVE_LAUNCH((dyn_loop_fwd_small_k<2, 4, false, 3, true>), dim3(B), dim3(64 * kSmWaves), SmShape<4>::kLdsFloats * sizeof(float), (hipStream_t)stream,
                   z1, zsup, zsstd, eps, extra, params, z, zdyn, zdstd, mean, std_, pred, act, B, Ts, N, sin_dim, lim_enc, elu, kc, g_sm_stamps,
                   ts0, ts1);
    } else if (act != nullptr) {
      STOVE_LOOP_LAUNCH(2);
    } else {
      STOVE_LOOP_LAUNCH(0);
    }
#undef STOVE_LOOP_LAUNCH
#undef STOVE_LOOP_LAUNCH_E
#undef STOVE_LOOP_LAUNCH_N
    STOVE_LAUNCH_CHECK();
    return 0;
  }
  const int G = gnn_group_for(B, N);
  int rc = 0;
#define STOVE_MLOOP_LAUNCH(ELU_, N6_)                                                                                                  \
  do {                                                                                                                                 \
    rc = gnn_lds_attr((const void*)dyn_loop_fwd_k<ELU_, N6_>);                                                                         \
    if (rc) return rc;                                                                                                                 \
    STOVE_LAUNCH((dyn_loop_fwd_k<ELU_, N6_>), dim3(stove_gnn_blocks(B, N)), dim3(256), kGnnLdsFloats * sizeof(float), (hipStream_t)stream, \
                 z1, zsup, zsstd, eps, extra, params, z, zdyn, zdstd, mean, std_, pred, act, B, Ts, N, G, sin_dim, lim_enc, elu, kc);  \
  } while (0)
  const bool n6 = N == 6 && G == 1;
  if (elu && n6) STOVE_MLOOP_LAUNCH(true, true);
  else if (elu) STOVE_MLOOP_LAUNCH(true, false);
  else if (n6) STOVE_MLOOP_LAUNCH(false, true);
  else STOVE_MLOOP_LAUNCH(false, false);
#undef STOVE_MLOOP_LAUNCH
  STOVE_LAUNCH_CHECK();
  return 0;
}

size_t stove_dynloop_bwd_ws_bytes(int B, int N) { return stove_gnn_bwd_ws_bytes(B, N); }

static bool small_bwd_path(int N, const float* act) { return small_graph(N) && act != nullptr; }

// workspace of stove_dynloop_bwd for Ts steps: per-workgroup partial weight gradients, plus (small-graph path) the dY streams
size_t stove_dynloop_bwd_ws_bytes_ts(int B, int Ts, int N) {
  size_t f = stove_gnn_bwd_ws_bytes(B, N) / sizeof(float);
  if (small_graph(N)) f = (size_t)B * kGnnGrads + (size_t)B * sm_dy_floats(N, Ts);
  return f * sizeof(float);
}

int stove_dynloop_bwd(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                      const float* params, const float* z, const float* act, const float* dz, const float* dzdyn, const float* dmean,
                      const float* dstd, const float* dpred, float* dz1, float* dzsup, float* dzsstd, float* dextra,
                      float* g_params, void* ws, int B, int Ts, int N, int sin_dim, int lim_enc, int elu, float pos_var,
                      float vel_std, float lat_std, void* stream) {
  return stove_dynloop_bwd_overlap(z1, zsup, zsstd, eps, extra, params, z, act, dz, dzdyn, dmean, dstd, dpred, dz1, dzsup, dzsstd,
                                   dextra, g_params, ws, B, Ts, N, sin_dim, lim_enc, elu, pos_var, vel_std, lat_std, stream, stream);
}

int stove_dynloop_bwd_overlap(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                              const float* params, const float* z, const float* act, const float* dz, const float* dzdyn,
                              const float* dmean, const float* dstd, const float* dpred, float* dz1, float* dzsup, float* dzsstd,
                              float* dextra, float* g_params, void* ws, int B, int Ts, int N, int sin_dim, int lim_enc, int elu,
                              float pos_var, float vel_std, float lat_std, void* stream, void* param_stream) {
  return dynloop_bwd_range(z1, zsup, zsstd, eps, extra, params, z, act, dz, dzdyn, dmean, dstd, dpred, dz1, dzsup, dzsstd, dextra,
                                 g_params, ws, B, Ts, N, sin_dim, lim_enc, elu, pos_var, vel_std, lat_std, 0, Ts, nullptr, stream, param_stream);
}

static int dynloop_bwd_range(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                            const float* params, const float* z, const float* act, const float* dz, const float* dzdyn,
                            const float* dmean, const float* dstd, const float* dpred, float* dz1, float* dzsup, float* dzsstd,
                            float* dextra, float* g_params, void* ws, int B, int Ts, int N, int sin_dim, int lim_enc, int elu,
                            float pos_var, float vel_std, float lat_std, int ts0, int ts1, float* carry, void* stream, void* param_stream) {
  STOVE_VALIDATE(dynloop_bwd(z1, zsup, zsstd, eps, extra, params, z, dz1, dzsup, dzsstd, dextra, g_params, ws, B, Ts, N, sin_dim));
  hipStream_t st = (hipStream_t)stream;
  if (ts0 < 0 || ts1 > Ts || ts0 >= ts1) return (int)hipErrorInvalidValue;
  const bool whole = ts0 == 0 && ts1 == Ts;
  if (!whole && (!small_bwd_path(N, act) || carry == nullptr)) return (int)hipErrorInvalidValue;
  hipStream_t sp = param_stream != nullptr ? (hipStream_t)param_stream : st;
  if (B == 0 || Ts == 0) return (int)hipErrorInvalidValue;
  if (N < 1 || N > 8 || sin_dim < 16 || sin_dim > 32 || (sin_dim > 16 && (extra == nullptr || dextra == nullptr)))
    return (int)hipErrorInvalidValue;
  LoopConst kc{pos_var, vel_std, lat_std};
  if (small_bwd_path(N, act)) {
    // small graphs: T-serial data-gradient chain (gnn_small_bwd.hip), then the weight gradients as a throughput pass
    float* gpart = (float*)ws;
    float* dy = gpart + (size_t)B * kGnnGrads;
    int rc = 0;
#define STOVE_LOOPB_LAUNCH_H(NMX_, ELU_, NT_, HD_)                                                                                    \
  do {                                                                                                                                \
    rc = (int)hipFuncSetAttribute((const void*)dyn_loop_bwd_small_k<NMX_, ELU_, NT_, HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)(smb_lds_floats<NMX_>() * sizeof(float)));                                                     \
    if (rc) return rc;                                                                                                                \
    STOVE_LAUNCH((dyn_loop_bwd_small_k<NMX_, ELU_, NT_, HD_>), dim3(B), dim3(64 * kSmWaves), smb_lds_floats<NMX_>() * sizeof(float), st, zsup, \
                 zsstd, eps, params, const_cast<float*>(act), dz, dzdyn, dmean, dstd, dpred, dz1, dzsup, dzsstd, dextra, dy, B, Ts, N, \
                 sin_dim, lim_enc, elu, kc, g_sm_stamps, ts0, ts1, carry);                                                            \
  } while (0)
    const bool head = dz != nullptr && dzdyn != nullptr && dmean != nullptr && dstd != nullptr && dpred == nullptr && sin_dim == 16 && lim_enc == 2;
#define STOVE_LOOPB_LAUNCH(ELU_)                                          \
  do {                                                                    \
    if (N == 3 && head) STOVE_LOOPB_LAUNCH_H(4, ELU_, 3, true);           \
    else if (N == 3) STOVE_LOOPB_LAUNCH_H(4, ELU_, 3, false);             \
    else if (N <= 4) STOVE_LOOPB_LAUNCH_H(4, ELU_, 0, false);             \
    else if (N == 6 && head) STOVE_LOOPB_LAUNCH_H(6, ELU_, 6, true);      \
    else STOVE_LOOPB_LAUNCH_H(6, ELU_, 0, false);                         \
  } while (0)
    if (g_sm_stamps != nullptr && !elu && N == 3 && head) {       // tools/loop_stamps.py
      rc = (int)hipFuncSetAttribute((const void*)dyn_loop_bwd_small_k<4, false, 3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(smb_lds_floats<4>() * sizeof(float)));
      if (rc) return rc;
      STOVE_LAUNCH((dyn_loop_bwd_small_k<4, false, 3, true, true>), dim3(B), dim3(64 * kSmWaves), smb_lds_floats<4>() * sizeof(float), st, zsup, zsstd,
                   eps, params, const_cast<float*>(act), dz, dzdyn, dmean, dstd, dpred, dz1, dzsup, dzsstd, dextra, dy, B, Ts, N, sin_dim,
                   lim_enc, elu, kc, g_sm_stamps, ts0, ts1, carry);
    } else if (elu) STOVE_LOOPB_LAUNCH(true);
    else STOVE_LOOPB_LAUNCH(false);
#undef STOVE_LOOPB_LAUNCH
#undef STOVE_LOOPB_LAUNCH_H
    STOVE_LAUNCH_CHECK();
    if (ts0 > 0) return 0;           // the weight-gradient pass contracts the streams of ALL steps: behind the piece that ends the backward
    rc = (int)hipFuncSetAttribute((const void*)gnn_dw_small_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kDwLdsFloats * sizeof(float)));
    if (rc) return rc;
    STOVE_TRY(stream_after(sp, st));        // the weight-gradient pass reads the dY streams; it only feeds the optimiser (second stream)
    STOVE_LAUNCH(gnn_dw_small_k, dim3(B), dim3(256), kDwLdsFloats * sizeof(float), sp, act, (const float*)dy, gpart, B, Ts, N);
    STOVE_LAUNCH_CHECK();
    STOVE_LAUNCH(reduce_chunks_k, dim3((kGnnGrads + 31) / 32), dim3(256), 0, sp, (const float*)gpart, g_params, kGnnGrads, B, 0);
    STOVE_LAUNCH_CHECK();
    return 0;
  }
  const int nb = stove_gnn_blocks(B, N), G = gnn_group_for(B, N);
  int rc = 0;
#define STOVE_MLOOPB_LAUNCH(ELU_, N6_)                                                                                                 \
  do {                                                                                                                                 \
    rc = gnn_lds_attr((const void*)dyn_loop_bwd_k<ELU_, N6_>);                                                                         \
    if (rc) return rc;                                                                                                                 \
    STOVE_LAUNCH((dyn_loop_bwd_k<ELU_, N6_>), dim3(nb), dim3(256), kGnnLdsFloats * sizeof(float), st, z1, zsup, zsstd, eps, extra,     \
                 params, z, act, dz, dzdyn, dmean, dstd, dpred, dz1, dzsup, dzsstd, dextra, (float*)ws, B, Ts, N, G, sin_dim, lim_enc, \
                 elu, kc);                                                                                                             \
  } while (0)
  const bool n6 = N == 6 && G == 1;
  if (elu && n6) STOVE_MLOOPB_LAUNCH(true, true);
  else if (elu) STOVE_MLOOPB_LAUNCH(true, false);
  else if (n6) STOVE_MLOOPB_LAUNCH(false, true);
  else STOVE_MLOOPB_LAUNCH(false, false);
#undef STOVE_MLOOPB_LAUNCH
  STOVE_LAUNCH_CHECK();
  STOVE_TRY(stream_after(sp, st));
  STOVE_LAUNCH(reduce_chunks_k, dim3((kGnnGrads + 31) / 32), dim3(256), 0, sp, (const float*)ws, g_params, kGnnGrads, nb, 0);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_rollout_fwd(const float* z_last, const float* extra, const float* params, float* z_pred, float* zstd, float* pred,
                      int B, int num, int A, int N, int sin_dim, int lim_enc, int elu, float pos_var, float vel_std,
                      float lat_std, void* stream) {
  STOVE_VALIDATE(rollout_fwd(z_last, extra, params, z_pred, B, num, A, N, sin_dim));
  if (B == 0 || num == 0) return 0;
  if (N < 1 || N > 8 || sin_dim < 16 || sin_dim > 32 || (sin_dim > 16 && (extra == nullptr || A < 1))) return (int)hipErrorInvalidValue;
  LoopConst kc{pos_var, vel_std, lat_std};
  if (N >= 2 && N <= 6) {
    int rc;
#define STOVE_ROLL_LAUNCH_N(NMX_, ELU_, NT_)                                                                                         \
  do {                                                                                                                               \
    rc = (int)hipFuncSetAttribute((const void*)rollout_fwd_small_k<NMX_, ELU_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                  (int)(SmShape<NMX_>::kLdsFloats * sizeof(float)));                                                 \
    if (rc) return rc;                                                                                                               \
    STOVE_LAUNCH((rollout_fwd_small_k<NMX_, ELU_, NT_>), dim3(B), dim3(64 * kSmWaves), SmShape<NMX_>::kLdsFloats * sizeof(float), (hipStream_t)stream, \
                 z_last, extra, params, z_pred, zstd, pred, B, num, A < 1 ? 1 : A, N, sin_dim, lim_enc, elu, kc);                    \
  } while (0)
#define STOVE_ROLL_LAUNCH(ELU_)                         \
  do {                                                  \
    if (N == 3) STOVE_ROLL_LAUNCH_N(4, ELU_, 3);        \
    else if (N <= 4) STOVE_ROLL_LAUNCH_N(4, ELU_, 0);   \
    else STOVE_ROLL_LAUNCH_N(6, ELU_, 0);               \
  } while (0)
    if (elu) STOVE_ROLL_LAUNCH(true);
    else STOVE_ROLL_LAUNCH(false);
#undef STOVE_ROLL_LAUNCH
#undef STOVE_ROLL_LAUNCH_N
    STOVE_LAUNCH_CHECK();
    return 0;
  }
  int rc = gnn_lds_attr((const void*)rollout_fwd_k);
  if (rc) return rc;
  STOVE_LAUNCH(rollout_fwd_k, dim3(stove_gnn_blocks(B, N)), dim3(256), kGnnLdsFloats * sizeof(float), (hipStream_t)stream,
                     z_last, extra, params, z_pred, zstd, pred, B, num, A < 1 ? 1 : A, N, gnn_group_for(B, N), sin_dim, lim_enc, elu, kc);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- flat parameter arena
static SpnArenaPlan arena_plan(const StoveSpnArenaPlan* p) {
  SpnArenaPlan q;
  q.obj_mu = p->obj_mu; q.obj_rho = p->obj_rho; q.obj_sum = p->obj_sum;
  q.bg_mu = p->bg_mu; q.bg_rho = p->bg_rho; q.bg_gidx = p->bg_gidx;
  q.obj_root = p->obj_root; q.bg_root = p->bg_root;
  q.obj_vmin = p->obj_vmin; q.obj_vmax = p->obj_vmax; q.bg_vmin = p->bg_vmin; q.bg_vmax = p->bg_vmax;
  return q;
}

int stove_spn_bake(const float* arena, const StoveSpnArenaPlan* plan, float* obj_coef, float* obj_wsum, float* obj_wroot,
                   float* bg_coef, float* bg_wroot, void* stream) {
  const int nb = (kAObjCoef + kABgCoef + 255) / 256;
  STOVE_LAUNCH(spn_bake_k, dim3(nb + kASoftBlocks), dim3(256), 0, (hipStream_t)stream, arena, arena_plan(plan), obj_coef, obj_wsum, obj_wroot,
               bg_coef, bg_wroot, nb);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_spn_bake_bwd(const float* arena, const StoveSpnArenaPlan* plan, const StoveSpnTableGrads* g, float* grad_arena, void* stream) {
  const int nb = (kAObjCoef + kABgCoef + 255) / 256;
  STOVE_LAUNCH(spn_bake_bwd_k, dim3(nb + kASoftBlocks), dim3(256), 0, (hipStream_t)stream, arena, arena_plan(plan), (const float*)g->obj_coef,
               (const float*)g->obj_wsum, (const float*)g->obj_wroot, (const float*)g->bg_coef, (const float*)g->bg_wroot, grad_arena, nb);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_arena_gather(const float* arena, const int32_t* src, float* image, int n, void* stream) {
  if (n == 0) return 0;
  STOVE_LAUNCH(arena_gather_k, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, arena, src, image, n);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_arena_scatter_add(const float* gimage, const int32_t* src, float* grad_arena, int n, void* stream) {
  if (n == 0) return 0;
  STOVE_LAUNCH(arena_scatter_add_k, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gimage, src, grad_arena, n);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_sum_chunks(const float* parts, float* out, size_t n, int chunks, void* stream) {
  if (n == 0) return 0;
  if (n % 4 != 0 || chunks < 1) return (int)hipErrorInvalidValue;
  const int n4 = (int)(n / 4);
  STOVE_LAUNCH(sum_chunks4_k, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, parts, out, n4, chunks, 0);
  STOVE_LAUNCH_CHECK();
  return 0;
}

size_t stove_gemm_bf16_ws_floats(int M, int N, int splitk) { return splitk > 1 ? (size_t)splitk * M * N : 0; }

int stove_gemm_bf16(const float* A, const float* B, const float* bias, const float* add, float* C, int M, int N, int K, int lda,
                    int ldb, int ldc, int a_kmajor, int b_kmajor, int nsplit, int splitk, int tile, float* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  STOVE_VALIDATE(gemm(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, nsplit, splitk, ws));
  if (M == 0 || N == 0) return 0;
  if (K <= 0 || splitk < 1 || nsplit < 1 || nsplit > 3 || tile < 0 || (tile > 3 && (tile < 11 || tile > 15))) return (int)hipErrorInvalidValue;
  // tile 11 / 12: measurement variants of the 256 x 128 NT kernel (tools/gemm_bf16_bench.py): no loads in the loop / no MFMAs
  if (tile == 11 && !a_kmajor && !b_kmajor && nsplit == 2) return gemm_launch<false, false, 2, 256, 128, 1>(A, B, bias, add, C, M, N, K, lda, ldb, ldc, 1, st);
  if (tile == 12 && !a_kmajor && !b_kmajor && nsplit == 2) return gemm_launch<false, false, 2, 256, 128, 2>(A, B, bias, add, C, M, N, K, lda, ldb, ldc, 1, st);
  // 13 / 14: the same two variants of the 256 x 256 tile
  if (tile == 13 && !a_kmajor && !b_kmajor && nsplit == 2) return gemm_launch<false, false, 2, 256, 256, 1, false, 64, 128>(A, B, bias, add, C, M, N, K, lda, ldb, ldc, 1, st);
  if (tile == 14 && !a_kmajor && !b_kmajor && nsplit == 2) return gemm_launch<false, false, 2, 256, 256, 2, false, 64, 128>(A, B, bias, add, C, M, N, K, lda, ldb, ldc, 1, st);
  if (tile == 15 && !a_kmajor && !b_kmajor && nsplit == 2) return gemm_launch<false, false, 2, 256, 256, 3, false, 64, 128>(A, B, bias, add, C, M, N, K, lda, ldb, ldc, 1, st);
  if (tile > 3) tile = 1;
  // float4 granularity along the contiguous dimension of an operand / of C, or the element-wise slow path for it
  // (fc1 of the recognition network has 50 columns)
  auto odd = [](const void* p, int ld, int extent) { return ((uintptr_t)p & 15) != 0 || (ld & 3) != 0 || (extent & 3) != 0; };
  const int scalar_bits = (odd(A, lda, a_kmajor ? M : K) ? 1 : 0) | (odd(B, ldb, b_kmajor ? N : K) ? 2 : 0) |
                          ((odd(C, ldc, N) || (bias != nullptr && ((uintptr_t)bias & 15)) || (add != nullptr && ((uintptr_t)add & 15))) ? 4 : 0);
  // split-K: no epilogue terms, except add == C (accumulate into C, e.g. a gradient view), applied by the slice sum
  const bool acc_c = splitk > 1 && add != nullptr && add == C;
  if (acc_c) add = nullptr;
  // ... or a dense (M x N, float4-addressable) add term of fewer than 16 slices: added by the slice sum, like the bias
  const float* sum_add = (splitk > 1 && splitk < 16 && add != nullptr && (((uintptr_t)add & 15) == 0)) ? add : nullptr;
  if (sum_add != nullptr) add = nullptr;
  const float* sum_bias = splitk > 1 ? bias : nullptr;      // split-K: the bias is added by the slice sum (fewer than 16 slices)
  if (splitk > 1 && splitk < 16) bias = nullptr;
  if (splitk > 1 && (ws == nullptr || bias != nullptr || add != nullptr || ldc != N || (scalar_bits & 4))) return (int)hipErrorInvalidValue;
  if (scalar_bits & 2) return (int)hipErrorInvalidValue;           // B has to be float4-addressable
  float* out = splitk > 1 ? ws : C;
  const int ldo = splitk > 1 ? N : ldc;
  if (tile == 0) tile = 1;
  int rc;
  if (nsplit == 3) {
    // round 5: hi / lo pieces in IEEE half instead of bf16 (11 + 11 significant bits: the product is as good as an fp32 one) for
    // operands inside half's range -- the forward products x W_ih^T, h W_hh^T of the recognition network.  K-contiguous,
    // float4-addressable operands.
    if (a_kmajor || b_kmajor || scalar_bits) return (int)hipErrorInvalidValue;
    rc = tile == 3 ? gemm_launch<false, false, 2, 256, 256, 0, false, 64, 128, true>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits)
       : tile == 1 ? gemm_launch<false, false, 2, 256, 128, 0, false, 64, 64, true>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits)
                   : gemm_launch<false, false, 2, 128, 128, 0, false, 64, 64, true>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits);
  } else if (scalar_bits & 1) {        // element-wise A: the two layouts fc1's backward needs (d_a1 W1 and d_a1^T h), 128 x 128 tile
    if (!b_kmajor) return (int)hipErrorInvalidValue;
    if (a_kmajor) rc = nsplit == 2 ? gemm_launch<true, true, 2, 128, 128, 0, true>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits)
                                   : gemm_launch<true, true, 1, 128, 128, 0, true>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits);
    else rc = nsplit == 2 ? gemm_launch<false, true, 2, 128, 128, 0, true>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits)
                          : gemm_launch<false, true, 1, 128, 128, 0, true>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits);
  } else {
#define STOVE_GEMM_TILE(AK, BK_, NS)                                                                                \
  rc = tile == 3 ? gemm_launch<AK, BK_, NS, 256, 256, 0, false, 64, 128>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits)           \
     : tile == 1 ? gemm_launch<AK, BK_, NS, 256, 128>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits)           \
                 : gemm_launch<AK, BK_, NS, 128, 128>(A, B, bias, add, out, M, N, K, lda, ldb, ldo, splitk, st, scalar_bits)
#define STOVE_GEMM_CASE(AK, BK_)             \
  do {                                       \
    if (nsplit == 2) {                       \
      STOVE_GEMM_TILE(AK, BK_, 2);           \
    } else {                                 \
      STOVE_GEMM_TILE(AK, BK_, 1);           \
    }                                        \
  } while (0)
  if (!a_kmajor && !b_kmajor) STOVE_GEMM_CASE(false, false);
  else if (!a_kmajor && b_kmajor) STOVE_GEMM_CASE(false, true);
  else if (a_kmajor && b_kmajor) STOVE_GEMM_CASE(true, true);
  else STOVE_GEMM_CASE(true, false);
  }
#undef STOVE_GEMM_CASE
#undef STOVE_GEMM_TILE
  if (rc) return rc;
  if (splitk > 1) {
    const int n4 = (int)((size_t)M * N / 4);
    if (splitk >= 16)
      STOVE_LAUNCH(sum_chunks4_par_k, dim3((n4 + 63) / 64), dim3(256), 0, st, (const float*)ws, C, n4, splitk, acc_c ? 1 : 0);
    else
      STOVE_LAUNCH(sum_chunks4_k, dim3((n4 + 255) / 256), dim3(256), 0, st, (const float*)ws, C, n4, splitk, acc_c ? 1 : 0, sum_bias, N / 4, sum_add);
    STOVE_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------- reward head of the action-conditioned model
size_t stove_reward_head_param_floats(void) { return (size_t)kRhParams; }
size_t stove_reward_head_saved_floats(int items, int n_obj) { return (size_t)items * ((size_t)n_obj * 32 + 32 + 16 + 8); }
static int reward_head_blocks(int items) {
  const int b = (items + kRhWaves - 1) / kRhWaves;
  return b < 1 ? 1 : (b > 256 ? 256 : b);
}
size_t stove_reward_head_bwd_ws_floats(int items) { return (size_t)reward_head_blocks(items) * kRhWaves * kRhParams; }

int stove_reward_head_fwd(const float* pred, const float* params, float* reward, float* saved, int items, int n_obj, void* stream) {
  if (items == 0) return 0;
  if (n_obj < 1 || pred == nullptr || params == nullptr || reward == nullptr || saved == nullptr) return (int)hipErrorInvalidValue;
  float* H0 = saved;
  float* Q = H0 + (size_t)items * n_obj * 32;
  float* A1 = Q + (size_t)items * 32;
  float* A2 = A1 + (size_t)items * 16;
  STOVE_LAUNCH(reward_head_fwd_k, dim3(reward_head_blocks(items)), dim3(64 * kRhWaves), 0, (hipStream_t)stream, pred, params, reward, H0, Q, A1, A2,
               items, n_obj);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_reward_head_bwd(const float* pred, const float* params, const float* reward, const float* saved, const float* d_reward, float* d_pred,
                          float* g_params, float* ws, int items, int n_obj, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (items == 0) {
    (void)hipMemsetAsync(g_params, 0, kRhParams * sizeof(float), st);
    return 0;
  }
  if (n_obj < 1 || ws == nullptr || d_pred == nullptr || g_params == nullptr) return (int)hipErrorInvalidValue;
  const float* H0 = saved;
  const float* Q = H0 + (size_t)items * n_obj * 32;
  const float* A1 = Q + (size_t)items * 32;
  const float* A2 = A1 + (size_t)items * 16;
  const int blocks = reward_head_blocks(items);
  STOVE_LAUNCH(reward_head_bwd_k, dim3(blocks), dim3(64 * kRhWaves), 0, st, pred, params, reward, H0, Q, A1, A2, d_reward, d_pred, ws, items, n_obj);
  STOVE_LAUNCH_CHECK();
  STOVE_LAUNCH(reduce_chunks_k, dim3((kRhParams + 31) / 32), dim3(256), 0, st, (const float*)ws, g_params, kRhParams, blocks * kRhWaves, 0);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_small_linear(const float* x, const float* W, const float* b, float* y, int rows, int in_dim, int out_dim, int w_transposed, void* stream) {
  if (rows == 0) return 0;
  if (in_dim < 1 || out_dim < 1 || in_dim > 64 || out_dim > 64) return (int)hipErrorInvalidValue;
  const size_t total = (size_t)rows * out_dim;
  if ((total + 255) / 256 > 0x7fffffffULL) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(small_linear_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, W, b, y, rows, in_dim, out_dim, w_transposed);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- stream ordering / graph replay helpers
int stove_stream_after(void* to, void* from) { return (int)stream_after((hipStream_t)to, (hipStream_t)from); }

// Events that order two DIFFERENT captures against each other (stream_after) must live as long as the graphs holding their
// record / wait nodes.  A caller that captures opens a list first: every such event created while it is open is appended, and
// the caller destroys the list together with its graphs (stove_amd/graphed.py).  With no list open they are never destroyed.
void* stove_event_list_begin(void) {
  std::lock_guard<std::mutex> g(event_list_mu());
  auto* v = new std::vector<hipEvent_t>();
  event_list_current() = v;
  return (void*)v;
}
int stove_event_list_end(void* list) {
  std::lock_guard<std::mutex> g(event_list_mu());
  if (event_list_current() == (std::vector<hipEvent_t>*)list) event_list_current() = nullptr;
  return list == nullptr ? 0 : (int)((std::vector<hipEvent_t>*)list)->size();
}
int stove_event_list_destroy(void* list) {
  if (list == nullptr) return 0;
  std::lock_guard<std::mutex> g(event_list_mu());
  auto* v = (std::vector<hipEvent_t>*)list;
  if (event_list_current() == v) event_list_current() = nullptr;
  hipError_t e = hipSuccess;
  for (hipEvent_t ev : *v) {
    const hipError_t d = hipEventDestroy(ev);
    if (e == hipSuccess) e = d;
  }
  delete v;
  return (int)e;
}

// test / debugging utility: n 32-bit words at p <- value (tests poison a captured step's memory pool between replays)
int stove_fill_words(void* p, uint32_t value, size_t n_words, void* stream) {
  if (n_words == 0) return 0;
  const size_t blocks = (n_words + 256 * 8 - 1) / (256 * 8);
  STOVE_LAUNCH(fill_words_k, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, (uint32_t*)p, value, n_words);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_capture_begin(void* stream) { return (int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed); }

int stove_capture_end(void* stream, void** graph_out, int* n_nodes) {
  hipGraph_t g = nullptr;
  if (graph_out == nullptr) return (int)hipErrorInvalidValue;
  *graph_out = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess) return (int)e;
  size_t n = 0;
  e = hipGraphGetNodes(g, nullptr, &n);
  if (n_nodes != nullptr) *n_nodes = (int)n;
  *graph_out = (void*)g;
  return (int)e;
}

int stove_graph_instantiate(void* graph, void** exec_out) {
  if (exec_out == nullptr) return (int)hipErrorInvalidValue;
  *exec_out = nullptr;
  if (graph == nullptr) return 0;
  hipGraph_t g = (hipGraph_t)graph;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(g, nullptr, &n);
  if (e == hipSuccess && n > 0) {
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e == hipSuccess) *exec_out = (void*)x;
  }
  const hipError_t d = hipGraphDestroy(g);
  return (int)(e != hipSuccess ? e : d);
}

int stove_graph_launch(void* exec, void* stream) { return (int)hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream); }

int stove_graph_destroy(void* exec) { return exec == nullptr ? 0 : (int)hipGraphExecDestroy((hipGraphExec_t)exec); }

int stove_bw_transform(const float* x, float* out, int n_frames, int channels, int pixels, void* stream) {
  if (n_frames == 0) return 0;
  if (pixels % 4 != 0 || channels < 1) return (int)hipErrorInvalidValue;
  const long long total = (long long)n_frames * (pixels / 4);
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(bw_transform_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, n_frames, channels, pixels / 4);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_bw_transform_u8(const unsigned char* x, float* out, int n_frames, int channels, int pixels, void* stream) {
  if (n_frames == 0) return 0;
  if (pixels % 4 != 0 || channels < 1) return (int)hipErrorInvalidValue;
  const long long total = (long long)n_frames * (pixels / 4);
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(bw_transform_u8_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, n_frames, channels, pixels / 4);
  STOVE_LAUNCH_CHECK();
  return 0;
}

size_t stove_colsum_ws_floats(int rows, int cols) {
  const int chunks = rows < 512 ? (rows < 1 ? 1 : rows) : 512;
  return (size_t)(chunks + 16) * cols;
}

// one level: `rows` rows -> `used` partial rows (returned), each the sum of `per` consecutive rows
static int colsum_level(const float* a, float* part, int rows, int cols, int max_chunks, hipStream_t st) {
  const int chunks = rows < max_chunks ? rows : max_chunks;
  const int per = (rows + chunks - 1) / chunks;
  const int used = (rows + per - 1) / per;
  if (cols <= 64) {
    int cpad = 1;
    while (cpad < cols) cpad <<= 1;
    STOVE_LAUNCH(colsum_narrow_part_k, dim3(used), dim3(256), 0, st, a, part, rows, cols, cpad, per);
  } else {
    STOVE_LAUNCH(colsum_part_k, dim3(used, (cols / 4 + 255) / 256), dim3(256), 0, st, a, part, rows, cols / 4, per);
  }
  return used;
}

int stove_colsum(const float* a, float* out, float* ws, int rows, int cols, void* stream) { return stove_colsum2(a, out, nullptr, 0, ws, rows, cols, stream); }

int stove_colsum2(const float* a, float* out, float* out2, int accumulate, float* ws, int rows, int cols, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (cols <= 0 || (cols % 4 != 0 && cols > 64)) return (int)hipErrorInvalidValue;
  if (rows == 0) {
    if (!accumulate) {
      hipMemsetAsync(out, 0, sizeof(float) * cols, st);
      if (out2 != nullptr) hipMemsetAsync(out2, 0, sizeof(float) * cols, st);
    }
    return 0;
  }
  // 512 row chunks keep the first pass at HBM speed; a second pass folds them to <= 16 so that the final fixed-order
  // sum is short (a single 512-way pass over 1024 columns has only 32 workgroups and is latency-bound: 30 us)
  int used = colsum_level(a, ws, rows, cols, 512, st);
  STOVE_LAUNCH_CHECK();
  const float* src = ws;
  if (used > 16) {
    float* ws2 = ws + (size_t)(rows < 512 ? rows : 512) * cols;
    used = colsum_level(ws, ws2, used, cols, 16, st);
    STOVE_LAUNCH_CHECK();
    src = ws2;
  }
  STOVE_LAUNCH(reduce_chunks_k, dim3((cols + 31) / 32), dim3(256), 0, st, src, out, cols, used, accumulate, out2);      // both outputs in one launch
  STOVE_LAUNCH_CHECK();
  return 0;
}

static int small_tn_chunks(int rows) { return rows < 64 ? 1 : (rows / 64 < 512 ? rows / 64 : 512); }
size_t stove_small_tn_ws_floats(int rows, int M, int N) { return (size_t)small_tn_chunks(rows) * M * N; }

int stove_small_tn(const float* a, const float* b, float* out, float* ws, int rows, int M, int N, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (M < 1 || N < 1 || M * N > 256) return (int)hipErrorInvalidValue;
  if (rows == 0) {
    hipMemsetAsync(out, 0, sizeof(float) * M * N, st);
    return 0;
  }
  const int chunks = small_tn_chunks(rows), per = (rows + chunks - 1) / chunks;
  STOVE_LAUNCH(small_tn_part_k, dim3(chunks), dim3(256), 0, st, a, b, ws, rows, M, N, per);
  STOVE_LAUNCH_CHECK();
  STOVE_LAUNCH(reduce_chunks_k, dim3((M * N + 31) / 32), dim3(256), 0, st, (const float*)ws, out, M * N, chunks, 0);
  STOVE_LAUNCH_CHECK();
  return 0;
}

size_t stove_flat_adam_ws_bytes(int nseg) { return (size_t)nseg * sizeof(int) + ADAM_SCAN_BLOCKS * sizeof(float); }

int stove_flat_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, size_t numel,
                    const int* seg_of4, const unsigned char* seg_trainable, float* seg_steps, int nseg, void* ws, float* grad_norm_out,
                    const float* hyper_dev, float lr, float beta1, float beta2, float eps, float max_norm, int clip, void* stream) {
  if (numel == 0 || nseg == 0) return 0;
  if (numel % 4 != 0 || ws == nullptr || seg_of4 == nullptr || seg_trainable == nullptr || seg_steps == nullptr)
    return (int)hipErrorInvalidValue;
  AdamHyper k;
  k.lr = lr; k.b1 = beta1; k.b2 = beta2; k.eps = eps; k.max_norm = max_norm;
  const int n4 = (int)(numel / 4);
  hipStream_t st = (hipStream_t)stream;
  float* part = reinterpret_cast<float*>(ws);
  int* flags = reinterpret_cast<int*>(part + ADAM_SCAN_BLOCKS);          // cleared by the caller once, by adam_tick_k ever after
  STOVE_LAUNCH(grad_scan_k, dim3(ADAM_SCAN_BLOCKS), dim3(256), 0, st, grads, seg_of4, flags, part, n4);
  STOVE_LAUNCH(flat_adam_k, dim3((n4 + 255) / 256), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq, seg_of4,
               seg_trainable, (const float*)seg_steps, (const int*)flags, (const float*)part, grad_norm_out, hyper_dev, k, clip & 1, n4);
  STOVE_LAUNCH(adam_tick_k, dim3((nseg + 255) / 256), dim3(256), 0, st, seg_trainable, seg_steps, flags, nseg, (clip >> 1) & 1);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- SuPAIR state pipeline / ELBO assembly
static ZpConst zp_const(const float* span_low) {
  ZpConst k;
  for (int d = 0; d < 8; ++d) {
    k.span[d] = span_low[d];
    k.low[d] = span_low[8 + d];
  }
  return k;
}

int stove_supair_state_fwd(const float* codes, const float* span_low, float* zc, float* pos, long long* idx, float* zfix,
                           unsigned char* hits, float* zl, float* sl, float* init6, int n, int T, int o, int skip, int fix,
                           int mode, void* stream) {
  return stove_supair_state_fwd2(codes, span_low, zc, pos, idx, zfix, hits, zl, sl, init6, 6, nullptr, 0, n, T, o, skip, fix, mode, stream);
}

int stove_supair_state_fwd2(const float* codes, const float* span_low, float* zc, float* pos, long long* idx, float* zfix,
                            unsigned char* hits, float* zl, float* sl, float* init, int init_ld, const float* lat_noise, int lat_dim,
                            int n, int T, int o, int skip, int fix, int mode, void* stream) {
  if (init_ld < 6 + (lat_noise != nullptr ? lat_dim : 0) || lat_dim < 0) return (int)hipErrorInvalidValue;
  float* init6 = init;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return 0;
  if (T < 2 || skip < 1 || skip >= T || o < 1 || o > kMatchN) return (int)hipErrorInvalidValue;
  const int M = n * T * o;
  if (codes != nullptr) {        // codes == NULL: zc (constrained states) and idx (a matching) are the caller's; only the last stage runs
    STOVE_LAUNCH(zp_constrain_k, dim3((M * 8 + 255) / 256), dim3(256), 0, st, codes, zp_const(span_low), zc, pos, M);
    STOVE_LAUNCH_CHECK();
    int rc = stove_match_objects(pos, idx, nullptr, n, T, o, 2, mode, stream);
    if (rc) return rc;
  }
  STOVE_LAUNCH(supair_state_fwd_k, dim3((M + 255) / 256), dim3(256), 0, st, (const float*)zc, (const long long*)idx, zfix, hits, zl, sl,
               init6, n, T, o, skip, fix, init_ld, lat_noise, lat_dim);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_supair_state_bwd(const float* zc, const long long* idx, const unsigned char* hits, const float* zfix, const float* g_zfix,
                           const float* g_zl, const float* g_sl, const float* g_init6, const float* span_low, float* gfix_ws,
                           float* g_codes, int n, int T, int o, int skip, void* stream) {
  return stove_supair_state_bwd2(zc, idx, hits, zfix, g_zfix, g_zl, g_sl, g_init6, 6, span_low, gfix_ws, g_codes, n, T, o, skip, stream);
}

int stove_supair_state_bwd2(const float* zc, const long long* idx, const unsigned char* hits, const float* zfix, const float* g_zfix,
                            const float* g_zl, const float* g_sl, const float* g_init6, int init_ld, const float* span_low, float* gfix_ws,
                            float* g_codes, int n, int T, int o, int skip, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return 0;
  const int M = n * T * o;
  (void)gfix_ws;      // (until round 5 the intermediate of a two-launch form; the argument stays for the ABI)
  STOVE_LAUNCH(supair_state_bwd_k, dim3((M + 255) / 256), dim3(256), 0, st, zc, idx, hits, zfix, g_zfix, g_zl, g_sl, g_init6, zp_const(span_low),
               g_codes, n, T, o, skip, init_ld);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_noise_normal(float* out, size_t n, unsigned long long* state, void* stream) {
  if (n == 0) return 0;
  if (((uintptr_t)out & 15) != 0 || state == nullptr) return (int)hipErrorInvalidValue;
  const size_t threads = (n + 3) / 4;
  if ((threads + 255) / 256 > 0x7fffffffULL) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(noise_normal_k, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, (const unsigned long long*)state);
  STOVE_LAUNCH(noise_tick_k, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_zall_fwd(const float* zfix, const float* zs, float* zall, int n, int T, int o, int skip, void* stream) {
  if (n == 0) return 0;
  const int M = n * (T - 1) * o;
  STOVE_LAUNCH(zall_fwd_k, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, zfix, zs, zall, n, T, o, skip);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_zall_bwd(const float* zfix, const float* zs, const float* g_zall, const float* dz_in, float* g_zfix, float* g_zs, int n, int T, int o,
                   int skip, void* stream) {
  if (n == 0) return 0;
  const int M = n * T * o + n * (T - skip) * o;
  STOVE_LAUNCH(zall_bwd_k, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, zfix, zs, g_zall, dz_in, g_zfix, g_zs, n, T, o, skip);
  STOVE_LAUNCH_CHECK();
  return 0;
}

static TransStd trans_std(const float* s16) {
  TransStd t;
  for (int d = 0; d < 16; ++d) t.s[d] = s16[d];
  return t;
}

int stove_elbo_fwd(const float* zs, const float* mean, const float* std_, const float* zdyn, const float* lik, const float* trans_std16,
                   float* part_ws, float* out3, int n, int T, int o, int skip, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n == 0 || skip < 1 || skip >= T) return (int)hipErrorInvalidValue;
  STOVE_LAUNCH(elbo_part_k, dim3(n), dim3(256), 0, st, zs, mean, std_, zdyn, lik, trans_std(trans_std16), part_ws, T, o, skip);
  STOVE_LAUNCH_CHECK();
  STOVE_LAUNCH(elbo_final_k, dim3(1), dim3(256), 0, st, (const float*)part_ws, out3, n, T, skip);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_elbo_bwd(const float* zs, const float* mean, const float* std_, const float* zdyn, const float* trans_std16, const float* g_out,
                   float* g_zs, float* g_mean, float* g_std, float* g_zdyn, float* g_lik, int n, int T, int o, int skip, void* stream) {
  if (n == 0) return 0;
  const size_t M = (size_t)n * (T - skip) * o * 18 + (size_t)n * (T - 1);
  STOVE_LAUNCH(elbo_bwd_k, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, zs, mean, std_, zdyn, trans_std(trans_std16), g_out,
               g_zs, g_mean, g_std, g_zdyn, g_lik, n, T, o, skip);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- LSTM cell (recognition network)
// ---------------------------------------------------------------- recognition-network head
// ---- fused output head (csrc/head_fused.hip): H = 256, HID <= 64, OUT = 8 ------------------------------------------------------
static int enc_head_groups(int rows) {
  const int tiles = (rows + 15) / 16;
  int g = (tiles + 3) / 4;
  return g < 1 ? 1 : (g > 102 ? 102 : g);       // 102 groups x 5 roles = 510 workgroups: all resident at 2 per CU (a second round of
                                                // workgroups would start when the first ends and double the kernel's time)
}

int stove_enc_head_fwd(const float* h, const float* W1, const float* b1, const float* W2, const float* b2, float* h1, float* codes, int rows,
                       int H, int HID, int OUT, int frames, int fc1_split, void* stream) {
  if (H != kEhH || HID < 1 || HID > kEhHid || OUT != kEhOut || frames < 0 || (frames > 0 && rows % frames != 0)) return (int)hipErrorInvalidValue;
  if (rows == 0) return 0;
  static_assert(2 * kEhHid * kEhLdB == sizeof(float) * kEhHid * kEhLd, "the two bf16 images of W1 take the fp32 image's place");
  const size_t lds = sizeof(float) * kEhHid * kEhLd;
  const int tiles = (rows + 15) / 16;
  const int grid = (tiles + 3) / 4 < 512 ? (tiles + 3) / 4 : 512;
  int rc;
  if (fc1_split) {       // fc1 as three half-piece MFMAs per product (the default); else on v_mfma_f32_16x16x4_f32 (encoder_gemm = 'fp32')
    rc = (int)hipFuncSetAttribute((const void*)enc_head_fwd_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    STOVE_LAUNCH((enc_head_fwd_k<true>), dim3(grid), dim3(256), lds, (hipStream_t)stream, h, W1, b1, W2, b2, h1, codes, rows, HID, frames);
  } else {
    rc = (int)hipFuncSetAttribute((const void*)enc_head_fwd_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    STOVE_LAUNCH((enc_head_fwd_k<false>), dim3(grid), dim3(256), lds, (hipStream_t)stream, h, W1, b1, W2, b2, h1, codes, rows, HID, frames);
  }
  STOVE_LAUNCH_CHECK();
  return 0;
}

size_t stove_enc_head_bwd_ws_floats(int rows, int HID) { return (size_t)enc_head_groups(rows) * eh_part_floats(HID); }

int stove_enc_head_bwd(const float* dcodes, const float* h1, const float* h, const float* W1, const float* W2, float* gh, float* gW1,
                       float* gb1, float* gW2, float* gb2, int accumulate, float* ws, int rows, int H, int HID, int OUT, int frames, void* stream) {
  if (H != kEhH || HID < 1 || HID > kEhHid || OUT != kEhOut || frames < 0 || (frames > 0 && rows % frames != 0)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int P = eh_part_floats(HID);
  if (rows == 0) {
    if (!accumulate) {
      hipMemsetAsync(gW1, 0, sizeof(float) * HID * kEhH, st);
      hipMemsetAsync(gW2, 0, sizeof(float) * kEhOut * HID, st);
      hipMemsetAsync(gb1, 0, sizeof(float) * HID, st);
      hipMemsetAsync(gb2, 0, sizeof(float) * kEhOut, st);
    }
    return 0;
  }
  const int groups = enc_head_groups(rows);
  const size_t lds = sizeof(float) * kEhBwdLds;
  int rc;
  if (HID <= 50) {
    rc = (int)hipFuncSetAttribute((const void*)enc_head_bwd_k<14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    STOVE_LAUNCH(enc_head_bwd_k<14>, dim3(groups * 5), dim3(256), lds, st, dcodes, h1, h, W1, W2, gh, ws, rows, HID, groups, frames);
  } else {
    rc = (int)hipFuncSetAttribute((const void*)enc_head_bwd_k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    STOVE_LAUNCH(enc_head_bwd_k<16>, dim3(groups * 5), dim3(256), lds, st, dcodes, h1, h, W1, W2, gh, ws, rows, HID, groups, frames);
  }
  STOVE_LAUNCH_CHECK();
  (void)P;
  STOVE_LAUNCH(enc_head_reduce_k, dim3((eh_part_floats(HID) + 31) / 32), dim3(256), 0, st, (const float*)ws, gW1, gW2, gb1, gb2, HID, groups, accumulate);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_lstm_cell_fwd(const float* gx, const float* gh, const float* c_prev, float* c, float* h, int n, int H, int fast, void* stream) {
  if (n == 0) return 0;
  if (H % 4) return (int)hipErrorInvalidValue;
  const size_t total = (size_t)n * (H / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (fast) STOVE_LAUNCH(lstm_cell_fwd_k<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gx, gh, c_prev, c, h, n, H);
  else STOVE_LAUNCH(lstm_cell_fwd_k<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gx, gh, c_prev, c, h, n, H);
  STOVE_LAUNCH_CHECK();
  return 0;
}

int stove_lstm_cell_bwd(const float* gx, const float* gh, const float* c_prev, const float* c, const float* dh,
                        const float* dc_in, float* dg, float* dc_out, float* dgx_sum, const float* dg_more, int n_more, int n, int H,
                        int fast, void* stream) {
  if (n == 0) return 0;
  if (H % 4 || n_more < 0 || (n_more > 0 && dg_more == nullptr)) return (int)hipErrorInvalidValue;
  const size_t more_stride = (size_t)n * 4 * H;
  const size_t total = (size_t)n * (H / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (fast) STOVE_LAUNCH(lstm_cell_bwd_k<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gx, gh, c_prev, c, dh, dc_in, dg, dc_out, dgx_sum, dg_more,
                         n_more, more_stride, n, H);
  else STOVE_LAUNCH(lstm_cell_bwd_k<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gx, gh, c_prev, c, dh, dc_in, dg, dc_out, dgx_sum, dg_more,
                    n_more, more_stride, n, H);
  STOVE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- profiling hooks
void stove_profile_enable(int on) {
  std::lock_guard<std::mutex> g(prof_mu());
  prof_on() = on != 0;
}

// Synchronises every recorded event pair, aggregates by kernel name and writes lines
// "name\ttotal_ms\tcount\twall_ms\n" (wall_ms: the union of the launches' time spans) into buf (at most cap bytes); returns the
// number of bytes needed.
size_t stove_profile_report(char* buf, size_t cap) {
  std::lock_guard<std::mutex> g(prof_mu());
  struct Agg {
    std::string name;
    double total = 0.0;
    long count = 0;
    std::vector<std::pair<double, double>> spans;      // [start, end) in ms since the first recorded launch
  };
  std::vector<Agg> agg;
  auto& recs = prof_recs();
  for (auto& r : recs) (void)hipEventSynchronize(r.b);
  for (auto& r : recs) {
    float ms = 0.0f, t0 = 0.0f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    (void)hipEventElapsedTime(&t0, recs.front().a, r.a);
    Agg* e = nullptr;
    for (auto& x : agg)
      if (x.name == r.name) {
        e = &x;
        break;
      }
    if (e == nullptr) {
      agg.push_back(Agg());
      e = &agg.back();
      e->name = r.name;
    }
    e->total += ms;
    e->count += 1;
    e->spans.push_back({(double)t0, (double)t0 + ms});
  }
  for (auto& r : recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  recs.clear();
  std::string out;
  for (auto& e : agg) {
    // launches of one kernel on two streams overlap in time (the recognition network's row chunks): the time they COVER
    std::sort(e.spans.begin(), e.spans.end());
    double wall = 0.0, lo = 0.0, hi = -1.0;
    for (auto& sp : e.spans) {
      if (hi < lo || sp.first > hi) {
        if (hi >= lo) wall += hi - lo;
        lo = sp.first;
        hi = sp.second;
      } else if (sp.second > hi) {
        hi = sp.second;
      }
    }
    if (hi >= lo) wall += hi - lo;
    out += e.name + "\t" + std::to_string(e.total) + "\t" + std::to_string(e.count) + "\t" + std::to_string(wall) + "\n";
  }
  if (buf != nullptr && cap > 0) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}

// ---------------------------------------------------------------- temporal object matching
int stove_match_objects(const float* feat, long long* idx, float* perm, int B, int T, int N, int F, int mode, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (N < 1 || N > kMatchN || F < 1 || F > kMatchF || mode < 0 || mode > 3) return (int)hipErrorInvalidValue;
  const bool serial = mode == 3;       // '3_only' through the frame-by-frame walk (the check of match3_table_k in the tests)
  if (serial) mode = 0;
  const size_t lds = (size_t)T * N * (F + 1) * sizeof(float);
  if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  // the lane-parallel kernel wins where the serial walk of lane 0 has real work per frame (the greedy matcher, more than
  // three objects: 580 -> 170 us for six); for three objects and the nearest-slot rules the serial walk is shorter
  if (!serial && mode == 0 && N == 3 && F == 2)
    STOVE_LAUNCH((match3_table_k<2>), dim3(B), dim3(64), lds, st, feat, idx, B, T);
  else if (!serial && mode == 0 && N == 3 && F == 5)
    STOVE_LAUNCH((match3_table_k<5>), dim3(B), dim3(64), lds, st, feat, idx, B, T);
  else if (mode == 1 && perm == nullptr && (size_t)T * N * 5 + T + 8 <= 64 * 1024) {
    // greedy: per-frame assignments in parallel, then the composition walk (match.hip); scratch = the tail bytes of idx itself
    STOVE_LAUNCH(match_greedy_frames_k, dim3((B * T + 255) / 256), dim3(256), 0, st, feat, idx, B, T, N, F);
    STOVE_LAUNCH(match_greedy_compose_k, dim3(B), dim3(64), (size_t)((T * N + T + 3) & ~3) + (size_t)T * N * sizeof(int), st, feat, idx, B, T, N, F);
  } else if (mode == 1 || N > 3)
    STOVE_LAUNCH(match_objects_par_k, dim3(B), dim3(64), lds, st, feat, idx, B, T, N, F, mode);
  else if (N == 3 && F == 2)
    STOVE_LAUNCH((match_objects_k<3, 2>), dim3(B), dim3(64), lds, st, feat, idx, perm, B, T, N, F, mode);
  else if (N == 3 && F == 5)
    STOVE_LAUNCH((match_objects_k<3, 5>), dim3(B), dim3(64), lds, st, feat, idx, perm, B, T, N, F, mode);
  else
    STOVE_LAUNCH((match_objects_k<0, 0>), dim3(B), dim3(64), lds, st, feat, idx, perm, B, T, N, F, mode);
  STOVE_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
