// Pure v_mfma_f32_32x32x2_f32 issue-rate probe: waves/SIMD x accumulators, reports TF/s and in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* stamps) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = threadIdx.x * 0.001f + 0.5f, y = blockIdx.x * 0.0001f + 0.25f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int blocks, int iters, float* out, unsigned long long* st) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<NACC><<<blocks, 256>>>(out, iters, st);
  hipEventRecord(a, 0);
  k<NACC><<<blocks, 256>>>(out, iters, st);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  double flops = (double)blocks * 4 * iters * 16 * NACC * 2.0 * 32 * 32 * 2;
  printf("blocks=%5d (%.1f waves/SIMD) acc=%d: %8.1f us  %7.1f TF/s  clock %.2f GHz  cycles/MFMA/wave %.1f\n", blocks, blocks / 256.0,
         NACC, ms * 1e3, flops / ms / 1e9, cyc / rt * 0.1, cyc / blocks / (iters * 16.0 * NACC));
}

int main() {
  float* out; unsigned long long* st;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&st, 4096 * 16);
  for (int blocks : {256, 512, 1024}) {
    run<1>(blocks, 64, out, st);
    run<2>(blocks, 64, out, st);
    run<4>(blocks, 64, out, st);
  }
  run<4>(512, 512, out, st);
  run<4>(512, 4096, out, st);
  return 0;
}
