// Debug helper (not part of the product library): dump a captured hipGraph_t -- node types, kernel names, edges -- as text.
// Build: hipcc --offload-arch=gfx950 -O2 -fPIC -shared -o libgraphdump.so graphdump.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
extern "C" int graph_dump(void* graph, const char* path) {
  hipGraph_t g = (hipGraph_t)graph;
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return 1;
  std::vector<hipGraphNode_t> nodes(n);
  if (hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return 2;
  std::map<hipGraphNode_t, int> id;
  for (size_t i = 0; i < n; ++i) id[nodes[i]] = (int)i;
  FILE* f = fopen(path, "w");
  if (!f) return 3;
  fprintf(f, "nodes %zu\n", n);
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    hipGraphNodeGetType(nodes[i], &t);
    std::string name = "?";
    if (t == hipGraphNodeTypeKernel) {
      hipKernelNodeParams p;
      memset(&p, 0, sizeof(p));
      if (hipGraphKernelNodeGetParams(nodes[i], &p) == hipSuccess && p.func) {
        const char* nm = hipKernelNameRefByPtr(p.func, nullptr);
        name = nm ? nm : "kernel?";
        char buf[64];
        snprintf(buf, sizeof(buf), " grid=%u block=%u", p.gridDim.x * p.gridDim.y * p.gridDim.z, p.blockDim.x);
        name += buf;
      }
    } else if (t == hipGraphNodeTypeMemcpy) name = "memcpy";
    else if (t == hipGraphNodeTypeMemset) name = "memset";
    else if (t == hipGraphNodeTypeEmpty) name = "empty";
    else if (t == hipGraphNodeTypeEventRecord) name = "event_record";
    else if (t == hipGraphNodeTypeWaitEvent) name = "event_wait";
    else { char b[32]; snprintf(b, sizeof(b), "type%d", (int)t); name = b; }
    size_t nd = 0;
    hipGraphNodeGetDependencies(nodes[i], nullptr, &nd);
    std::vector<hipGraphNode_t> deps(nd);
    if (nd) hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd);
    size_t nc = 0;
    hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nc);
    std::vector<hipGraphNode_t> ch(nc);
    if (nc) hipGraphNodeGetDependentNodes(nodes[i], ch.data(), &nc);
    fprintf(f, "%zu\t%s\tdeps:", i, name.c_str());
    for (auto d : deps) fprintf(f, " %d", id[d]);
    fprintf(f, "\tchildren:");
    for (auto c : ch) fprintf(f, " %d", id[c]);
    fprintf(f, "\n");
  }
  fclose(f);
  return 0;
}

// the graph a capturing stream is building (NULL if it is not capturing): lets a DAG be dumped BEFORE hipStreamEndCapture
extern "C" void* graph_of_stream(void* stream) {
  hipStreamCaptureStatus st; unsigned long long id; hipGraph_t g = nullptr; const hipGraphNode_t* deps; size_t nd;
  if (hipStreamGetCaptureInfo_v2((hipStream_t)stream, &st, &id, &g, &deps, &nd) != hipSuccess || st != hipStreamCaptureStatusActive) return nullptr;
  return (void*)g;
}
