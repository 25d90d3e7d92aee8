// What does a dependent kernel launch cost on the device?  N empty kernels in a row (a) inside ONE hipGraph, (b) as N graphs of
// one kernel, (c) as plain stream launches; grids of 1 block and of 256 blocks x 512 threads with 132 KB of LDS (the fused forward's).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/launch_gap_probe tools/probes/launch_gap_probe.hip && tools/probes/bin/launch_gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_empty(int* p) {
  extern __shared__ float sm[];
  if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) p[0] = (int)sm[0];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  CK(hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  const int N = 200;
  struct Shape { dim3 g, b; size_t lds; const char* name; } shapes[] = {{dim3(1), dim3(64), 0, "1 block x 64"},
      {dim3(256), dim3(512), 132 * 1024, "256 blocks x 512, 132 KB LDS"}, {dim3(1024), dim3(256), 0, "1024 blocks x 256"}};
  for (auto& s : shapes) {
    // (a) one graph of N kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, s.g, s.b, s.lds, st, (int*)nullptr);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    // (b) a graph of ONE kernel, launched N times
    hipGraph_t g1; hipGraphExec_t ge1;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_empty, s.g, s.b, s.lds, st, (int*)nullptr);
    CK(hipStreamEndCapture(st, &g1)); CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
    double best[3] = {1e9, 1e9, 1e9};
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipStreamSynchronize(st));
      double t0 = now(); CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); double t1 = now();
      best[0] = std::min(best[0], (t1 - t0) / N * 1e6);
      t0 = now(); for (int i = 0; i < N; ++i) CK(hipGraphLaunch(ge1, st)); CK(hipStreamSynchronize(st)); t1 = now();
      best[1] = std::min(best[1], (t1 - t0) / N * 1e6);
      t0 = now(); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, s.g, s.b, s.lds, st, (int*)nullptr); CK(hipStreamSynchronize(st)); t1 = now();
      best[2] = std::min(best[2], (t1 - t0) / N * 1e6);
    }
    printf("%-30s: %6.2f us per kernel inside one graph, %6.2f as one-kernel graphs, %6.2f as stream launches (wall / N, host included)\n", s.name,
           best[0], best[1], best[2]);
  }
  return 0;
}
