// How does hipGraph (ROCm 7.2, gfx950) map the branches of a captured multi-stream DAG onto hardware queues?
// Each kernel is ONE workgroup spinning for a fixed time, so two kernels overlap iff they sit on different queues.
// Build: hipcc --offload-arch=gfx950 -O2 -o graph_sched graph_sched.hip ; run: ./graph_sched
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && ticks < 0) *sink = 1;
}
static const long long MS = 100000;   // wall_clock64 runs at 100 MHz

static hipStream_t M, S, F;
static void K(hipStream_t st, double ms) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, (long long)(ms * MS), (int*)nullptr); }
// fork: `to` waits for what `from` has so far.  cont: put an empty continuation node on `from` BEFORE `to` gets its first node,
// so that `from`'s own successor is the first child of the fork point.
static int fork_(hipStream_t to, hipStream_t from, bool cont) {
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventRecord(ev, from));
  if (cont) {
    hipStreamCaptureStatus stt; unsigned long long id; hipGraph_t g; const hipGraphNode_t* deps; size_t nd;
    CK(hipStreamGetCaptureInfo_v2(from, &stt, &id, &g, &deps, &nd));
    if (stt == hipStreamCaptureStatusActive) {
      hipGraphNode_t n; CK(hipGraphAddEmptyNode(&n, g, deps, nd));
      CK(hipStreamUpdateCaptureDependencies(from, &n, 1, hipStreamSetCaptureDependencies));
    }
  }
  CK(hipStreamWaitEvent(to, ev, 0));
  CK(hipEventDestroy(ev));
  return 0;
}
static int join_(hipStream_t to, hipStream_t from) { return fork_(to, from, false); }

typedef int (*Pattern)(bool cont);
static int p1(bool c) { K(M, .05); if (fork_(S, M, c)) return 1; K(S, 1); K(M, 1); if (join_(M, S)) return 1; K(M, .05); return 0; }
static int p2(bool c) {   // two fork/join episodes on the same side stream
  K(M, .05); if (fork_(S, M, c)) return 1; K(S, 1); K(M, 1); if (join_(M, S)) return 1; K(M, .05);
  if (fork_(S, M, c)) return 1; K(S, 1); K(M, 1); if (join_(M, S)) return 1; K(M, .05); return 0; }
static int p3(bool c) {   // side episode + a different fork stream episode, then side again under a long main kernel
  K(M, .05); if (fork_(S, M, c)) return 1; K(S, .5); K(M, .5); if (join_(M, S)) return 1;
  if (fork_(F, M, c)) return 1; K(F, .5); K(M, .5); if (join_(M, F)) return 1;
  if (fork_(S, M, c)) return 1; K(S, .3); K(S, .3); K(S, .3); K(M, 1); if (join_(M, S)) return 1; K(M, .05); return 0; }
static int p4(bool c) {   // side forked early, joined only at the very end, main running a chain meanwhile; a fork-stream episode in between
  K(M, .05); if (fork_(S, M, c)) return 1; K(S, 1.5);
  K(M, .5); if (fork_(F, M, c)) return 1; K(F, .5); K(M, .5); if (join_(M, F)) return 1; K(M, .5);
  if (join_(M, S)) return 1; K(M, .05); return 0; }
static int p5(bool c) {   // side and fork active at the same time (three-way overlap)
  K(M, .05); if (fork_(S, M, c)) return 1; if (fork_(F, M, c)) return 1; K(S, 1); K(F, 1); K(M, 1);
  if (join_(M, S)) return 1; if (join_(M, F)) return 1; K(M, .05); return 0; }

static int run(const char* name, Pattern p, double ideal, double serial) {
  for (int cont = 0; cont < 2; ++cont) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(M, hipStreamCaptureModeRelaxed));
    if (p(cont)) return 1;
    CK(hipStreamEndCapture(M, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int i = 0; i < 4; ++i) {
      CK(hipEventRecord(a, M)); CK(hipGraphLaunch(ge, M)); CK(hipEventRecord(b, M)); CK(hipStreamSynchronize(M));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    // the same pattern launched eagerly on the three streams
    float eb = 1e9;
    for (int i = 0; i < 3; ++i) {
      CK(hipEventRecord(a, M)); if (p(false)) return 1; CK(hipEventRecord(b, M)); CK(hipStreamSynchronize(M));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < eb) eb = ms;
    }
    printf("%-4s cont=%d  graph %.2f ms   eager %.2f ms   (ideal %.2f, serial %.2f)\n", name, cont, best, eb, ideal, serial);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
int main() {
  CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&F, hipStreamNonBlocking));
  K(M, .01); K(S, .01); K(F, .01); CK(hipDeviceSynchronize());
  if (run("p1", p1, 1.1, 2.1)) return 1;
  if (run("p2", p2, 2.15, 4.15)) return 1;
  if (run("p3", p3, 2.05, 3.95)) return 1;
  if (run("p4", p4, 1.6, 3.6)) return 1;
  if (run("p5", p5, 1.1, 3.1)) return 1;
  return 0;
}
