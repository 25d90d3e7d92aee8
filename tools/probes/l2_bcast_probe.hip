// How fast can all 256 CUs stream the SAME weight set out of L2 (the access pattern of k_mlp_fwd_fused)?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/l2_bcast_probe.hip -o l2_bcast_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// each block: 8 waves; wave w streams tiles {2w, 2w+1} of a (16 tiles x K8 x 64 lanes) float4 table, `reps` times
template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_stream(const float4* __restrict__ tab, int K8, int reps, int nets, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int net = nets == 2 ? (blockIdx.x & 1) : 0;
  const float4* base = tab + (long long)net * 16 * K8 * 64;
  const float4* w0 = base + (long long)(2 * wave) * K8 * 64 + lane;
  const float4* w1 = base + (long long)(2 * wave + 1) * K8 * 64 + lane;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < reps; ++r) {
    for (int k8 = 0; k8 < K8; k8 += INFLIGHT) {
      float4 a[INFLIGHT], b[INFLIGHT];
#pragma unroll
      for (int s = 0; s < INFLIGHT; ++s) { a[s] = w0[(k8 + s) * 64]; b[s] = w1[(k8 + s) * 64]; }
#pragma unroll
      for (int s = 0; s < INFLIGHT; ++s) { acc.x += a[s].x + b[s].x; acc.y += a[s].y + b[s].y; acc.z += a[s].z + b[s].z; acc.w += a[s].w + b[s].w; }
    }
  }
  if (acc.x == 12345.678f) out[0] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  const int K8 = 64;
  const size_t n4 = (size_t)2 * 16 * K8 * 64;   // two nets x 1 MiB
  float4* tab;
  hipMalloc(&tab, n4 * sizeof(float4));
  hipMemset(tab, 0, n4 * sizeof(float4));
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nets = 1; nets <= 2; ++nets)
    for (int blocks : {256, 512}) {
      const int reps = 8;
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_stream<8>, dim3(blocks), dim3(512), 0, 0, tab, K8, reps, nets, out);
      hipEventRecord(e0, 0);
      const int iters = 20;
      for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_stream<8>, dim3(blocks), dim3(512), 0, 0, tab, K8, reps, nets, out);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)blocks * reps * 16 * K8 * 64 * 16;
      printf("nets %d blocks %d: %.1f us/launch, %.2f TB/s L2->CU (%.1f B/clk/CU at 2.1 GHz)\n", nets, blocks, ms * 1e3 / iters,
             bytes / (ms * 1e-3 / iters) * 1e-12, bytes / (ms * 1e-3 / iters) / 256 / 2.1e9);
    }
  return 0;
}
