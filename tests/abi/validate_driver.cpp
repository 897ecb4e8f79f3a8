// Host-only driver of the C ABI's argument-validation layer (stove_amd/csrc/validate.h -- the very header libstove_hip.so's entry
// points call first), built with -fsanitize=address,undefined by tests/test_host_cpu.py and run in the CPU suite: every case is
// (what, expected code), one line per failure, exit status = number of failures.  "Device" pointers are fake, suitably aligned
// addresses: the layer must never dereference them (a dereference is an ASan report and a non-zero exit).
#include <cstdio>
#include <cstdint>

#include "../../stove_amd/csrc/validate.h"

using namespace stove_validate;

static int failures = 0;
static void expect(const char* what, int got, int want) {
  if ((got != 0) != (want != 0) || (want != 0 && got != kStoveInvalidValue)) {
    std::printf("FAIL %s: got %d, want %d\n", what, got, want);
    ++failures;
  }
}
template <typename T>
static T* dev(uintptr_t k) { return reinterpret_cast<T*>(uintptr_t(0x7000000000ull) + k * 4096); }   // never mapped

int main() {
  const float* f = dev<const float>(1);
  float* o = dev<float>(2);
  void* ws = dev<void>(3);
  StoveSpnTables t{};
  t.obj_scope = dev<const int32_t>(10); t.obj_leaf_slot = dev<const int32_t>(11); t.obj_coef = dev<const float>(12);
  t.obj_wsum = dev<const float>(13); t.obj_wroot = dev<const float>(14); t.bg_side = dev<const int32_t>(15);
  t.bg_coef = dev<const float>(16); t.bg_wroot = dev<const float>(17); t.bg_dense = nullptr;
  StoveSpnTableGrads g{dev<float>(20), dev<float>(21), dev<float>(22), dev<float>(23), dev<float>(24)};
  StoveSpnTables t_no_obj = t; t_no_obj.obj_coef = nullptr;
  StoveSpnTables t_no_bg = t; t_no_bg.bg_wroot = nullptr;
  StoveSpnTableGrads g_hole = g; g_hole.obj_wsum = nullptr;

  // ---- RatSpn operator
  expect("objspn_fwd ok", objspn_fwd(&t, f, o, o, 64), 0);
  expect("objspn_fwd n = 0 with nothing else", objspn_fwd(nullptr, nullptr, nullptr, nullptr, 0), 0);
  expect("objspn_fwd NULL tables", objspn_fwd(nullptr, f, o, o, 64), 1);
  expect("objspn_fwd table with a NULL member", objspn_fwd(&t_no_obj, f, o, o, 64), 1);
  expect("objspn_fwd n < 0", objspn_fwd(&t, f, o, o, -1), 1);
  expect("objspn_fwd NULL out", objspn_fwd(&t, f, o, nullptr, 64), 1);
  expect("objspn_bwd ok", objspn_bwd(&t, f, f, f, f, o, &g, ws, 8), 0);
  expect("objspn_bwd d_marg without marg", objspn_bwd(&t, nullptr, f, f, f, o, &g, ws, 8), 1);
  expect("objspn_bwd NULL ws", objspn_bwd(&t, f, f, f, f, o, &g, nullptr, 8), 1);
  expect("objspn_bwd n = 0 still needs tables and gradient buffers (it zero-fills them)", objspn_bwd(nullptr, f, f, f, f, o, &g, ws, 0), 1);
  expect("objspn_bwd n = 0", objspn_bwd(&t, nullptr, nullptr, nullptr, nullptr, nullptr, &g, ws, 0), 0);
  expect("objspn_bwd gradient struct with a NULL member", objspn_bwd(&t, f, f, f, f, o, &g_hole, ws, 8), 1);
  expect("bgspn_fwd ok", bgspn_fwd(&t, f, o, o, 3, 1024), 0);
  expect("bgspn_fwd NULL bg table member", bgspn_fwd(&t_no_bg, f, o, o, 3, 1024), 1);
  expect("bgspn_fwd n_pix = 0", bgspn_fwd(&t, f, o, o, 3, 0), 1);
  expect("bgspn_bwd NULL grads", bgspn_bwd(&t, f, f, f, f, f, nullptr, nullptr, ws, 3, 1024), 1);
  // ---- scene likelihood
  expect("scene_fwd ok, dense", scene_fwd(&t, f, f, 25344, 3, 0, 0, o, o), 0);
  expect("scene_fwd ok, strided clips", scene_fwd(&t, f, f, 99 * 4, 3, 99, 100, o, o), 0);
  expect("scene_fwd n_frames = 0", scene_fwd(nullptr, nullptr, nullptr, 0, 3, 0, 0, nullptr, nullptr), 0);
  expect("scene_fwd n_obj = 0", scene_fwd(&t, f, f, 8, 0, 0, 0, o, o), 1);
  expect("scene_fwd n_obj = 9", scene_fwd(&t, f, f, 8, 9, 0, 0, o, o), 1);
  expect("scene_fwd stride shorter than the slice", scene_fwd(&t, f, f, 8, 3, 4, 3, o, o), 1);
  expect("scene_fwd frames not a whole number of slices", scene_fwd(&t, f, f, 9, 3, 4, 5, o, o), 1);
  expect("scene_fwd NULL tables", scene_fwd(nullptr, f, f, 8, 3, 0, 0, o, o), 1);
  expect("scene_fwd NULL saved", scene_fwd(&t, f, f, 8, 3, 0, 0, o, nullptr), 1);
  expect("scene_bwd ok", scene_bwd(&t, f, f, 8, 3, 0, 0, f, f, o, &g, ws), 0);
  expect("scene_bwd NULL ws", scene_bwd(&t, f, f, 8, 3, 0, 0, f, f, o, &g, nullptr), 1);
  expect("scene_bwd NULL grads", scene_bwd(&t, f, f, 8, 3, 0, 0, f, f, o, nullptr, ws), 1);
  expect("scene_bwd negative frames", scene_bwd(&t, f, f, -8, 3, 0, 0, f, f, o, &g, ws), 1);
  // ---- GNN step, recursion, rollout
  expect("gnn_fwd ok", gnn_fwd(f, f, o, 256, 3, 16), 0);
  expect("gnn_fwd B = 0", gnn_fwd(nullptr, nullptr, nullptr, 0, 3, 16), 0);
  expect("gnn_fwd N = 0", gnn_fwd(f, f, o, 256, 0, 16), 1);
  expect("gnn_fwd N = 9", gnn_fwd(f, f, o, 256, 9, 16), 1);
  expect("gnn_fwd sin_dim = 15", gnn_fwd(f, f, o, 256, 3, 15), 1);
  expect("gnn_fwd sin_dim = 33", gnn_fwd(f, f, o, 256, 3, 33), 1);
  expect("gnn_fwd NULL params", gnn_fwd(f, nullptr, o, 256, 3, 16), 1);
  expect("gnn_bwd NULL ws", gnn_bwd(f, f, f, o, o, nullptr, 4, 3, 16), 1);
  expect("dynloop_fwd ok", dynloop_fwd(f, f, f, f, nullptr, f, o, o, o, o, o, 256, 98, 3, 16), 0);
  expect("dynloop_fwd action inputs without `extra`", dynloop_fwd(f, f, f, f, nullptr, f, o, o, o, o, o, 256, 98, 3, 23), 1);
  expect("dynloop_fwd Ts < 0", dynloop_fwd(f, f, f, f, nullptr, f, o, o, o, o, o, 256, -1, 3, 16), 1);
  expect("dynloop_fwd NULL eps", dynloop_fwd(f, f, f, nullptr, nullptr, f, o, o, o, o, o, 256, 98, 3, 16), 1);
  expect("dynloop_bwd ok", dynloop_bwd(f, f, f, f, nullptr, f, f, o, o, o, nullptr, o, ws, 256, 98, 3, 16), 0);
  expect("dynloop_bwd B = 0", dynloop_bwd(f, f, f, f, nullptr, f, f, o, o, o, nullptr, o, ws, 0, 98, 3, 16), 1);
  expect("dynloop_bwd NULL ws", dynloop_bwd(f, f, f, f, nullptr, f, f, o, o, o, nullptr, o, nullptr, 256, 98, 3, 16), 1);
  expect("dynloop_bwd extra without dextra", dynloop_bwd(f, f, f, f, f, f, f, o, o, o, nullptr, o, ws, 256, 98, 3, 23), 1);
  expect("rollout ok", rollout_fwd(f, nullptr, f, o, 256, 92, 0, 3, 16), 0);
  expect("rollout actions without A", rollout_fwd(f, f, f, o, 256, 92, 0, 3, 20), 1);
  expect("rollout num < 0", rollout_fwd(f, nullptr, f, o, 256, -1, 0, 3, 16), 1);
  // ---- GEMM
  expect("gemm ok", gemm(f, f, o, 25600, 1024, 1024, 1024, 1024, 1024, 0, 0, 3, 1, nullptr), 0);
  expect("gemm M = 0", gemm(nullptr, nullptr, nullptr, 0, 1024, 1024, 1024, 1024, 1024, 0, 0, 3, 1, nullptr), 0);
  expect("gemm K = 0", gemm(f, f, o, 8, 8, 0, 8, 8, 8, 0, 0, 2, 1, nullptr), 1);
  expect("gemm lda shorter than a row", gemm(f, f, o, 8, 8, 64, 32, 64, 8, 0, 0, 2, 1, nullptr), 1);
  expect("gemm ldb not a multiple of 4", gemm(f, f, o, 8, 8, 64, 64, 66, 8, 0, 0, 2, 1, nullptr), 1);
  expect("gemm B not 16-byte aligned", gemm(f, reinterpret_cast<const float*>(reinterpret_cast<uintptr_t>(f) + 4), o, 8, 8, 64, 64, 64, 8, 0, 0, 2, 1, nullptr), 1);
  expect("gemm split-K without a workspace", gemm(f, f, o, 8, 8, 64, 64, 64, 8, 0, 0, 2, 4, nullptr), 1);
  expect("gemm nsplit = 4", gemm(f, f, o, 8, 8, 64, 64, 64, 8, 0, 0, 4, 1, nullptr), 1);
  std::printf("%d failure(s)\n", failures);
  return failures;
}
