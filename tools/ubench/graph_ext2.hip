// Which capture patterns with manual event nodes on FORKED streams survive hipStreamEndCapture?  (debugging helper)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(int* p) { if (p) atomicAdd(p, 1); }
static int after(hipStream_t to, hipStream_t from) {           // the library's stream_after
  hipStreamCaptureStatus sf, st; unsigned long long idf = 0, idt = 0; hipGraph_t gf = nullptr, gt = nullptr; const hipGraphNode_t *df = nullptr, *dt = nullptr; size_t nf = 0, nt = 0;
  CK(hipStreamGetCaptureInfo_v2(from, &sf, &idf, &gf, &df, &nf));
  CK(hipStreamGetCaptureInfo_v2(to, &st, &idt, &gt, &dt, &nt));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (sf == hipStreamCaptureStatusActive && st == hipStreamCaptureStatusActive && idf != idt) {
    hipGraphNode_t n; CK(hipGraphAddEventRecordNode(&n, gf, df, nf, ev)); CK(hipStreamUpdateCaptureDependencies(from, &n, 1, hipStreamSetCaptureDependencies));
    hipGraphNode_t m; CK(hipGraphAddEventWaitNode(&m, gt, dt, nt, ev)); CK(hipStreamUpdateCaptureDependencies(to, &m, 1, hipStreamSetCaptureDependencies));
    return 0;
  }
  CK(hipEventRecord(ev, from)); CK(hipStreamWaitEvent(to, ev, 0)); CK(hipEventDestroy(ev));
  return 0;
}
int main(int argc, char** argv) {
  const int v = argc > 1 ? atoi(argv[1]) : 0;
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t M, P, S, F;
  CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&P, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&F, hipStreamNonBlocking));
  int* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
#define L(s) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d)
  hipGraph_t g1, g2;
  CK(hipStreamBeginCapture(M, hipStreamCaptureModeGlobal));
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeRelaxed));
  L(M);
  if (v == 0) {            // manual node on the origin stream
    if (after(S, M)) return 1; L(S); L(M);
  } else if (v == 1) {     // manual node on a forked stream, joined afterwards
    if (after(P, M)) return 1; L(P); if (after(S, P)) return 1; L(S); L(P); if (after(M, P)) return 1; L(M);
  } else if (v == 2) {     // forked stream re-forked before it rejoined
    if (after(P, M)) return 1; L(P); L(M); if (after(P, M)) return 1; L(P); if (after(S, P)) return 1; L(S); if (after(M, P)) return 1; L(M);
  } else if (v == 3) {     // the manual node is the LAST thing on the forked stream before the join
    if (after(P, M)) return 1; L(P); if (after(S, P)) return 1; L(S); if (after(M, P)) return 1; L(M);
  } else if (v == 4) {     // fork F from P (nested fork), manual node on F
    if (after(P, M)) return 1; L(P); if (after(F, P)) return 1; L(F); if (after(S, F)) return 1; L(S); L(F); if (after(P, F)) return 1; L(P); if (after(M, P)) return 1; L(M);
  } else if (v == 5) {     // nested fork, manual node last on F
    if (after(P, M)) return 1; L(P); if (after(F, P)) return 1; L(F); if (after(S, F)) return 1; L(S); if (after(P, F)) return 1; L(P); if (after(M, P)) return 1; L(M);
  } else if (v == 6) {     // F pre-forked from the ORIGIN, then used as a branch of P (waits P's event, P waits F's event)
    if (after(F, M)) return 1; if (after(P, M)) return 1; L(P); if (after(F, P)) return 1; L(F); L(P); if (after(P, F)) return 1; L(P); if (after(M, P)) return 1; L(M);
  } else if (v == 7) {     // nested fork F from P, but F joins the ORIGIN directly (P and F both wait-ed by M)
    if (after(P, M)) return 1; L(P); if (after(F, P)) return 1; L(F); L(P); if (after(M, F)) return 1; if (after(M, P)) return 1; L(M);
  }
  printf("v%d: captured\n", v);
  CK(hipStreamEndCapture(S, &g2)); printf("side ended\n");
  CK(hipStreamEndCapture(M, &g1)); printf("main ended\n");
  hipGraphExec_t x1, x2; CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
  CK(hipGraphLaunch(x1, M)); CK(hipGraphLaunch(x2, S)); CK(hipDeviceSynchronize());
  int h = 0; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost)); printf("v%d ok, kernels ran: %d\n", v, h);
  return 0;
}
