// Argument validation of the C ABI's entry points (include/stove_hip.h): plain host C++, no HIP types, so that the same code is
// compiled twice -- into libstove_hip.so (every entry point below calls its check first and returns the code) and, with
// -fsanitize=address,undefined, into the host-only driver tests/abi/validate_driver.cpp that the CPU test suite runs (SURVEY section 5,
// "Race detection / sanitizers").  A check reads its pointer ARGUMENTS (NULL, alignment) and the host-side table structs; it never
// dereferences device memory.  Return: 0 or kStoveInvalidValue (== hipErrorInvalidValue, asserted in capi.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/stove_hip.h"

namespace stove_validate {

constexpr int kStoveInvalidValue = 1;
constexpr int kMaxObjects = 8;        // N <= 8 everywhere (csrc/gnn.hip, scene kernels); the small-graph recursion kernels take 2..6

inline bool null_any() { return false; }
template <typename T, typename... Rest>
inline bool null_any(const T* p, const Rest*... rest) { return p == nullptr || null_any(rest...); }
inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

inline int obj_tables(const StoveSpnTables* t) {
  if (t == nullptr || null_any(t->obj_scope, t->obj_coef, t->obj_wsum, t->obj_wroot)) return kStoveInvalidValue;
  return 0;
}
inline int bg_tables(const StoveSpnTables* t) {
  if (t == nullptr || null_any(t->bg_side, t->bg_coef, t->bg_wroot)) return kStoveInvalidValue;
  return 0;
}
inline int table_grads(const StoveSpnTableGrads* g, bool obj, bool bg) {
  if (g == nullptr) return kStoveInvalidValue;
  if (obj && null_any(g->obj_coef, g->obj_wsum, g->obj_wroot)) return kStoveInvalidValue;
  if (bg && null_any(g->bg_coef, g->bg_wroot)) return kStoveInvalidValue;
  return 0;
}

// ---- RatSpn operator (stove_objspn_*, stove_bgspn_*); n == 0 is a valid empty call (the forward returns at once)
inline int objspn_fwd(const StoveSpnTables* t, const float* inputs, const float* xw, const float* out, int n) {
  if (n < 0) return kStoveInvalidValue;
  if (n == 0) return 0;
  if (obj_tables(t) || null_any(inputs, xw, out)) return kStoveInvalidValue;
  return 0;
}
inline int objspn_bwd(const StoveSpnTables* t, const float* marg, const float* xw, const float* out, const float* dout,
                      const float* d_marg, const StoveSpnTableGrads* g, const void* ws, int n) {
  if (n < 0) return kStoveInvalidValue;
  // (n == 0 is a valid call that overwrites the table gradients with zeros: everything but the per-sample arrays is still needed)
  if (obj_tables(t) || t->obj_leaf_slot == nullptr || ws == nullptr || table_grads(g, true, false)) return kStoveInvalidValue;
  if (n == 0) return 0;
  if (null_any(xw, out, dout)) return kStoveInvalidValue;
  if (d_marg != nullptr && marg == nullptr) return kStoveInvalidValue;
  return 0;
}
inline int bgspn_fwd(const StoveSpnTables* t, const float* inputs, const float* ell, const float* out, int n, int n_pix) {
  if (n < 0 || n_pix < 1) return kStoveInvalidValue;
  if (n == 0) return 0;
  if (bg_tables(t) || null_any(inputs, ell, out)) return kStoveInvalidValue;
  return 0;
}
inline int bgspn_bwd(const StoveSpnTables* t, const float* inputs, const float* marg, const float* ell, const float* out, const float* dout,
                     const float* d_marg, const StoveSpnTableGrads* g, const void* ws, int n, int n_pix) {
  if (n < 0 || n_pix < 1) return kStoveInvalidValue;
  if (bg_tables(t) || ws == nullptr || table_grads(g, false, true)) return kStoveInvalidValue;
  if (n == 0) return 0;
  if (null_any(inputs, ell, out, dout)) return kStoveInvalidValue;
  if (d_marg != nullptr && marg == nullptr) return kStoveInvalidValue;
  return 0;
}

// ---- Supair.likelihood (stove_scene_*): frame map as stove_hip.h states it (0, 0 = dense)
inline int frame_map(int n_frames, int seq_frames, int seq_stride) {
  if (seq_frames == 0) return 0;
  if (seq_frames < 0 || seq_stride < seq_frames || n_frames % seq_frames != 0) return kStoveInvalidValue;
  return 0;
}
inline int scene_fwd(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames, int seq_stride,
                     const float* ll, const float* saved) {
  if (n_frames < 0 || n_obj < 1 || n_obj > kMaxObjects || frame_map(n_frames, seq_frames, seq_stride)) return kStoveInvalidValue;
  if (n_frames == 0) return 0;
  if (obj_tables(t) || bg_tables(t) || null_any(frames, z, ll, saved)) return kStoveInvalidValue;
  return 0;
}
inline int scene_bwd(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames, int seq_stride,
                     const float* saved, const float* dll, const float* dz, const StoveSpnTableGrads* g, const void* ws) {
  if (n_frames < 0 || n_obj < 1 || n_obj > kMaxObjects || frame_map(n_frames, seq_frames, seq_stride)) return kStoveInvalidValue;
  if (n_frames == 0) return 0;
  if (obj_tables(t) || t->obj_leaf_slot == nullptr || bg_tables(t) || null_any(frames, z, saved, dll, dz) || ws == nullptr ||
      table_grads(g, true, true))
    return kStoveInvalidValue;
  return 0;
}

// ---- Dynamics.forward (stove_gnn_*), the recursion (stove_dynloop_*), Stove.rollout
inline int gnn_shape(int B, int N, int sin_dim) {
  if (B < 0 || N < 1 || N > kMaxObjects || sin_dim < 16 || sin_dim > 32) return kStoveInvalidValue;
  return 0;
}
inline int gnn_fwd(const float* s_in, const float* params, const float* result, int B, int N, int sin_dim) {
  if (gnn_shape(B, N, sin_dim)) return kStoveInvalidValue;
  if (B == 0) return 0;
  return null_any(s_in, params, result) ? kStoveInvalidValue : 0;
}
inline int gnn_bwd(const float* s_in, const float* params, const float* d_result, const float* d_s_in, const float* g_params, const void* ws,
                   int B, int N, int sin_dim) {
  if (gnn_shape(B, N, sin_dim)) return kStoveInvalidValue;
  if (B == 0) return 0;
  return (null_any(s_in, params, d_result, d_s_in, g_params) || ws == nullptr) ? kStoveInvalidValue : 0;
}
inline int dynloop_fwd(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra, const float* params,
                       const float* z, const float* zdyn, const float* zdstd, const float* mean, const float* std_, int B, int Ts, int N,
                       int sin_dim) {
  if (gnn_shape(B, N, sin_dim) || Ts < 0) return kStoveInvalidValue;
  if (B == 0 || Ts == 0) return 0;
  if (null_any(z1, zsup, zsstd, eps, params, z, zdyn, zdstd, mean, std_)) return kStoveInvalidValue;
  if (sin_dim > 16 && extra == nullptr) return kStoveInvalidValue;
  return 0;
}
inline int dynloop_bwd(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra, const float* params,
                       const float* z, const float* dz1, const float* dzsup, const float* dzsstd, const float* dextra, const float* g_params,
                       const void* ws, int B, int Ts, int N, int sin_dim) {
  if (gnn_shape(B, N, sin_dim) || B == 0 || Ts <= 0) return kStoveInvalidValue;       // (an empty backward has nothing to overwrite g_params with)
  if (null_any(z1, zsup, zsstd, eps, params, z, dz1, dzsup, dzsstd, g_params) || ws == nullptr) return kStoveInvalidValue;
  if (sin_dim > 16 && (extra == nullptr || dextra == nullptr)) return kStoveInvalidValue;
  return 0;
}
inline int rollout_fwd(const float* z_last, const float* extra, const float* params, const float* z_pred, int B, int num, int A, int N,
                       int sin_dim) {
  if (gnn_shape(B, N, sin_dim) || num < 0) return kStoveInvalidValue;
  if (B == 0 || num == 0) return 0;
  if (null_any(z_last, params, z_pred)) return kStoveInvalidValue;
  if (sin_dim > 16 && (extra == nullptr || A < 1)) return kStoveInvalidValue;
  return 0;
}

// ---- stove_gemm_bf16: C (M x N) = A (M x K) B^T (N x K) [+ bias + add]; leading dimensions cover their rows, B float4-addressable
inline int gemm(const float* A, const float* B, const float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kmajor, int b_kmajor,
                int nsplit, int splitk, const float* ws) {
  if (M < 0 || N < 0) return kStoveInvalidValue;
  if (M == 0 || N == 0) return 0;
  if (K <= 0 || splitk < 1 || nsplit < 1 || nsplit > 3) return kStoveInvalidValue;
  if (null_any(A, B, C)) return kStoveInvalidValue;
  if (lda < (a_kmajor ? M : K) || ldb < (b_kmajor ? N : K) || ldc < N) return kStoveInvalidValue;
  if (misaligned16(B) || (ldb & 3) != 0 || ((b_kmajor ? N : K) & 3) != 0) return kStoveInvalidValue;
  if (splitk > 1 && (ws == nullptr || ldc != N)) return kStoveInvalidValue;
  return 0;
}

}  // namespace stove_validate
