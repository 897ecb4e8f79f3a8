// Shared device helpers for the STOVE gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <vector>

namespace stove {

constexpr int kWave = 64;
constexpr float kLog2Pi = 1.8378770664093453f;

// Wave-uniform value of the wave index inside the block (lets hipcc emit s_load for table reads).
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Sum over the 64 lanes of a wave with DPP moves (VALU only, no LDS crossbar traffic).
// The result is valid in lane 63; wave_sum() broadcasts it.
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, false));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, false));  // row_shr:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1,3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2,3
  return v;
}
// Sum within each 16-lane row only (4 fused DPP adds); the row totals end up in lanes 15, 31, 47, 63.
// Cheaper than a full wave sum when the four row totals can be combined later (e.g. by an LDS stage).
__device__ __forceinline__ float row_sum_lane15(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, false));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = wave_sum_lane63(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Lane exchanges over 16 and 32 lanes with gfx950's v_permlane16_swap / v_permlane32_swap (one VALU instruction each on two
// copies of the value; a ds_bpermute / __shfl_xor is an LDS round trip, ~100+ cycles on a one-wave-per-SIMD serial chain).
// As inline asm: hipcc 7.2 hands back the first result of __builtin_amdgcn_permlane32_swap for both members of its pair
// (tools/ubench/permlane32_swap.hip).  Every lane of the wave has to be active.  The copy, the wait states and the swap are ONE
// asm block: a VGPR written by a VALU instruction needs two wait states before v_permlane*_swap reads it (the compiler emits
// `s_nop 1` for its builtin), and the hazard recognizer does not look inside inline asm -- the v_mov is one wait state for the
// value's producer, the s_nop covers the copy.
//   v_permlane32_swap a, b: lanes 32..63 of a <-> lanes 0..31 of b      -> a = [lower | lower], b = [upper | upper]
//   v_permlane16_swap a, b: odd 16-lane rows of a <-> even rows of b    -> a = [r0 r0 r2 r2],   b = [r1 r1 r3 r3]
__device__ __forceinline__ float sum_xor32(float v) {          // v + (value of lane ^ 32)
  float a = v, b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
  return a + b;
}
__device__ __forceinline__ float sum_xor16(float v) {          // v + (value of lane ^ 16)
  float a = v, b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
  return a + b;
}
__device__ __forceinline__ float max_xor32(float v) {          // max(v, value of lane ^ 32)
  float a = v, b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ float max_xor16(float v) {          // max(v, value of lane ^ 16)
  float a = v, b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ float from_xor16(float v) {         // value of lane ^ 16
  float a = v, b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
  return (threadIdx.x & 16) ? a : b;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Packed fp32 pair: v_pk_fma_f32 does two fp32 FMAs per lane in one VALU slot (full rate on CDNA3/4) when its operands sit in
// natural register pairs.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// One-dimensional bilinear tap pair of torch's grid_sample (align_corners=False, zero padding):
// pixel coordinate q in an axis of `n` samples -> floor index i0, weights (1-t, t), in-bounds flags.
struct Tap1 {
  int i0;
  float t;
  float in0, in1;   // 1.0 / 0.0
};
__device__ __forceinline__ Tap1 make_tap(float q, int n) {
  Tap1 r;
  const float f = floorf(q);
  r.i0 = (int)f;
  r.t = q - f;
  r.in0 = (r.i0 >= 0 && r.i0 < n) ? 1.0f : 0.0f;
  r.in1 = (r.i0 + 1 >= 0 && r.i0 + 1 < n) ? 1.0f : 0.0f;
  return r;
}
// Where frame f of a (sequences x frames) batch lives when the caller hands over a time-slice of a longer clip without
// copying it (Stove.forward scores frames 1..T-1 of x (n, T, 1024), reference stove.py:731-736: x[:, 1:]): the slice has
// `seq_frames` frames per sequence, consecutive sequences are `seq_stride` frames apart.  seq_frames = 0: dense.
struct FrameMap {
  int seq_frames, seq_stride;
  __host__ __device__ __forceinline__ size_t row(int f) const {
    return seq_frames > 0 ? (size_t)(f / seq_frames) * seq_stride + (f % seq_frames) : (size_t)f;
  }
};

// Geometry of the spatial transformer for frames other than 32 x 32 and for the torch-1.0.1 sampling convention
// (align_corners=True), csrc/capi.hip scene_geom().  A normalised coordinate g maps to the pixel coordinate sxa g + cx (columns;
// sya g + cy for rows) with cx = (W - 1) / 2 in both conventions and sxa = W / 2 (align_corners False) or (W - 1) / 2 (True);
// glimpse pixel j sits at the normalised coordinate pa j + pb ((2j + 1)/10 - 1, or -1 + 2j/9), frame column X at fax X + fbx.
// The kernels that take it are templates on ANY: false = the 32 x 32 / align_corners=False constants at compile time (the
// expressions below are then exactly the ones the tuned path was validated with), true = everything from this struct.
struct SceneGeom {
  int W, H;
  float pa, pb;
  float sxa, sya, cx, cy;
  float fax, fbx, fay, fby;
};
// Coverage of a ones-image sampled at q (zero padding): value and d/dq.
__device__ __forceinline__ float cover(float q, int n, float* dq) {
  const Tap1 t = make_tap(q, n);
  *dq = t.in1 - t.in0;
  return (1.0f - t.t) * t.in0 + t.t * t.in1;
}

}  // namespace stove

// make stream `to` wait for everything enqueued on `from` so far (no-op when they are the same stream).  A failed fork or
// join would turn into an unordered race on saved tensors / workspaces, so every caller propagates the error code.
namespace stove {
// When the caller replays the step as hipGraphs it may capture `to` and `from` in two DIFFERENT captures (stove_amd/graphed.py:
// the long parameter-gradient chain of the backward pass is its own graph, launched on its own stream, because the runtime's
// scheduler of ONE multi-branch graph serialised it behind the main chain).  An event recorded in one capture cannot be
// waited for in another, so the dependency becomes a pair of explicit graph nodes on a persistent event: an event-record node
// behind `from`'s current capture dependencies, an event-wait node in front of whatever `to` captures next.  The graph that
// records must be launched before the graph that waits (each replay re-records the event before the wait is enqueued).
struct CaptureInfo {
  bool active;
  unsigned long long id;
  hipGraph_t graph;
  const hipGraphNode_t* deps;
  size_t ndeps;
};
inline hipError_t capture_info(hipStream_t s, CaptureInfo* ci) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  ci->id = 0; ci->graph = nullptr; ci->deps = nullptr; ci->ndeps = 0;
  const hipError_t e = hipStreamGetCaptureInfo_v2(s, &st, &ci->id, &ci->graph, &ci->deps, &ci->ndeps);
  ci->active = e == hipSuccess && st == hipStreamCaptureStatusActive;
  return e;
}
inline hipError_t capture_add_event_node(hipStream_t s, const CaptureInfo& ci, hipEvent_t ev, bool record) {
  hipGraphNode_t n;
  hipError_t e = record ? hipGraphAddEventRecordNode(&n, ci.graph, ci.deps, ci.ndeps, ev)
                        : hipGraphAddEventWaitNode(&n, ci.graph, ci.deps, ci.ndeps, ev);
  if (e != hipSuccess) return e;
  return hipStreamUpdateCaptureDependencies(s, &n, 1, hipStreamSetCaptureDependencies);
}
// the open event list of the capturing caller (stove_event_list_begin): cross-capture events are appended to it
inline std::mutex& event_list_mu() {
  static std::mutex m;
  return m;
}
inline std::vector<hipEvent_t>*& event_list_current() {
  static std::vector<hipEvent_t>* cur = nullptr;
  return cur;
}
inline hipError_t stream_after(hipStream_t to, hipStream_t from) {
  if (to == from) return hipSuccess;
  CaptureInfo cf, ct;
  if (capture_info(from, &cf) == hipSuccess && capture_info(to, &ct) == hipSuccess && cf.active && ct.active && cf.id != ct.id) {
    hipEvent_t ev;                               // lives as long as the graphs that hold it: owned by the caller's event list
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    {
      std::lock_guard<std::mutex> g(event_list_mu());
      if (event_list_current() != nullptr) event_list_current()->push_back(ev);      // else: leaked (no owner announced)
    }
    e = capture_add_event_node(from, cf, ev, true);
    if (e != hipSuccess) return e;               // the record node stays in `from`'s graph; the event dies with the list
    return capture_add_event_node(to, ct, ev, false);
  }
  hipEvent_t ev;
  hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) return e;
  e = hipEventRecord(ev, from);
  if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
  const hipError_t d = hipEventDestroy(ev);          // released by the runtime once the wait has been satisfied
  return e != hipSuccess ? e : d;
}
// A forked stream must be joined on EVERY exit path (an unjoined stream inside a hipGraph capture invalidates the capture;
// outside one it leaves work racing with whatever the caller enqueues next): joins in the destructor unless join() ran.
struct JoinGuard {
  hipStream_t to, from;
  bool done;
  JoinGuard(hipStream_t to_, hipStream_t from_) : to(to_), from(from_), done(to_ == from_) {}
  hipError_t join() {
    if (done) return hipSuccess;
    done = true;
    return stream_after(to, from);
  }
  void dismiss() { done = true; }
  ~JoinGuard() {
    if (!done) (void)stream_after(to, from);
  }
};
}  // namespace stove
#define STOVE_TRY(expr)                               \
  do {                                                \
    hipError_t e__ = (expr);                          \
    if (e__ != hipSuccess) return (int)e__;           \
  } while (0)

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py `roofline`) ----
// Off by default (no overhead, no global state touched).  When enabled, every kernel launch is
// bracketed by two events; stove_profile_report() synchronises them and aggregates by kernel.
#include <mutex>
#include <string>
#include <vector>
namespace stove {
struct ProfRec {
  const char* name;
  hipEvent_t a, b;
};
inline bool& prof_on() {
  static bool on = false;
  return on;
}
inline std::vector<ProfRec>& prof_recs() {
  static std::vector<ProfRec> v;
  return v;
}
inline std::mutex& prof_mu() {
  static std::mutex m;
  return m;
}
struct ProfScope {
  const char* name;
  hipStream_t st;
  hipEvent_t a, b;
  bool live;
  ProfScope(const char* n, hipStream_t s) : name(n), st(s), live(prof_on()) {
    if (live) {
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipEventRecord(a, st);
    }
  }
  ~ProfScope() {
    if (live) {
      (void)hipEventRecord(b, st);
      std::lock_guard<std::mutex> g(prof_mu());
      prof_recs().push_back({name, a, b});
    }
  }
};
}  // namespace stove
#define STOVE_LAUNCH(kern, grid, block, lds, st, ...)                \
  do {                                                               \
    stove::ProfScope ps__(#kern, st);                                \
    hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);     \
  } while (0)

#define STOVE_LAUNCH_CHECK()                          \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return (int)e__;           \
  } while (0)
