// Can two hipGraphs captured on two streams be ordered against each other with EXTERNAL event nodes
// (hipEventRecordWithFlags(.., hipEventRecordExternal) inside one capture, hipStreamWaitEvent(.., hipEventWaitExternal) inside the other)?
// G1 (stream M): a(0.3ms) ; record e1 ; b(1ms) ; record e2 ; c(0.3ms)
// G2 (stream S): wait e1 ; x(1ms) ; wait e2 ; y(0.2ms)
// correct + overlapped: x starts after a ends, overlaps b; y starts after b ends.  Kernels log [start, end] ticks.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long ticks, long long* log, int slot) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) { log[2 * slot] = t0; log[2 * slot + 1] = wall_clock64(); }
}
static const long long MS = 100000;
static int add_node(hipStream_t st, hipEvent_t ev, bool record) {
  hipStreamCaptureStatus stt; unsigned long long id; hipGraph_t g; const hipGraphNode_t* deps; size_t nd;
  CK(hipStreamGetCaptureInfo_v2(st, &stt, &id, &g, &deps, &nd));
  hipGraphNode_t n;
  if (record) CK(hipGraphAddEventRecordNode(&n, g, deps, nd, ev));
  else CK(hipGraphAddEventWaitNode(&n, g, deps, nd, ev));
  CK(hipStreamUpdateCaptureDependencies(st, &n, 1, hipStreamSetCaptureDependencies));
  return 0;
}
int main(int argc, char** argv) {
  const bool manual = argc > 1;
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t M, S;
  CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  long long* log; CK(hipMalloc(&log, 16 * sizeof(long long))); CK(hipMemset(log, 0, 16 * sizeof(long long)));
  hipEvent_t e1, e2, done; CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  hipGraph_t g1, g2; hipGraphExec_t x1, x2;
  CK(hipStreamBeginCapture(M, hipStreamCaptureModeRelaxed));
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, M, (long long)(0.3 * MS), log, 0);
  if (manual) { if (add_node(M, e1, true)) return 1; } else CK(hipEventRecordWithFlags(e1, M, hipEventRecordExternal)); printf("rec1 ok\n");
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, M, (long long)(1.0 * MS), log, 1);
  if (manual) { if (add_node(M, e2, true)) return 1; } else CK(hipEventRecordWithFlags(e2, M, hipEventRecordExternal));
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, M, (long long)(0.3 * MS), log, 2);
  CK(hipStreamEndCapture(M, &g1));
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeRelaxed));
  if (manual) { if (add_node(S, e1, false)) return 1; } else CK(hipStreamWaitEvent(S, e1, hipEventWaitExternal)); printf("wait1 ok\n");
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, S, (long long)(1.0 * MS), log, 3);
  if (manual) { if (add_node(S, e2, false)) return 1; } else CK(hipStreamWaitEvent(S, e2, hipEventWaitExternal));
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, S, (long long)(0.2 * MS), log, 4);
  CK(hipStreamEndCapture(S, &g2));
  size_t n1 = 0, n2 = 0; CK(hipGraphGetNodes(g1, nullptr, &n1)); CK(hipGraphGetNodes(g2, nullptr, &n2));
  printf("g1 nodes %zu, g2 nodes %zu\n", n1, n2);
  CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
  for (int it = 0; it < 4; ++it) {
    CK(hipGraphLaunch(x1, M));
    CK(hipGraphLaunch(x2, S));
    CK(hipEventRecord(done, S)); CK(hipStreamWaitEvent(M, done, 0));
    CK(hipStreamSynchronize(M)); CK(hipStreamSynchronize(S));
    long long h[10]; CK(hipMemcpy(h, log, sizeof(h), hipMemcpyDeviceToHost));
    const long long t0 = h[0];
    const char* nm[5] = {"a", "b", "c", "x", "y"};
    printf("iter %d:", it);
    for (int k = 0; k < 5; ++k) printf("  %s[%.2f..%.2f]", nm[k], (h[2 * k] - t0) / 1e5, (h[2 * k + 1] - t0) / 1e5);
    const bool ok = h[6] >= h[1] && h[8] >= h[3] && h[8] >= h[7];
    const bool ovl = h[6] < h[3];
    printf("  order %s overlap %s\n", ok ? "OK" : "VIOLATED", ovl ? "yes" : "no");
  }
  // back-to-back replays without host sync in between (the host runs ahead): ordering must hold per replay
  for (int it = 0; it < 3; ++it) { CK(hipGraphLaunch(x1, M)); CK(hipGraphLaunch(x2, S)); CK(hipEventRecord(done, S)); CK(hipStreamWaitEvent(M, done, 0)); }
  CK(hipDeviceSynchronize());
  long long h[10]; CK(hipMemcpy(h, log, sizeof(h), hipMemcpyDeviceToHost));
  printf("after 3 back-to-back: x after a: %d, y after b: %d\n", h[6] >= h[1], h[8] >= h[3]);
  return 0;
}
