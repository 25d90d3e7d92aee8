// Does the fp32 matrix pipe hold a higher clock on one MFMA shape than on the other?  (MI355X_MICROARCH.md, DVFS give-back item 7:
// for bf16 the 16x16x32 shape delivered 1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP, on random data only.)
// Same work per wave either way -- a 64 x 64 output tile, 8 reduction steps per loop trip, both operands re-read from LDS with
// ds_read_b128 every trip (random data) -- as 16 x v_mfma_f32_32x32x2_f32 (4 accumulators of 16 registers) or as
// 32 x v_mfma_f32_16x16x4_f32 (16 accumulators of 4 registers); 8 waves per CU (two per SIMD), one block per CU, launches of the
// length of the learner's fused forward (~120 us).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_shape_probe.hip -o tools/probes/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* out, int iters, unsigned long long* stamps) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = src[(blockIdx.x * 8192 + i) & 0xFFFFF];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4* l4 = reinterpret_cast<const float4*>(lds);
  float s = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 32) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int base = ((it * 4 + wave) * 64 + lane) & 2047;
      float4 a[2], b[2];
      a[0] = l4[base & 2047]; a[1] = l4[(base + 256) & 2047]; b[0] = l4[(base + 512) & 2047]; b[1] = l4[(base + 768) & 2047];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(t == 0 ? b[j].x : t == 1 ? b[j].y : t == 2 ? b[j].z : b[j].w,
                                                             t == 0 ? a[i].x : t == 1 ? a[i].y : t == 2 ? a[i].z : a[i].w, acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int base = ((it * 4 + wave) * 64 + lane) & 2047;
      float4 a[2], b[2];   // 8 floats per lane per operand: rows 16 i.. (i = 0..3) x two 4-deep reduction steps
      a[0] = l4[base & 2047]; a[1] = l4[(base + 256) & 2047]; b[0] = l4[(base + 512) & 2047]; b[1] = l4[(base + 768) & 2047];
      const float av[8] = {a[0].x, a[0].y, a[0].z, a[0].w, a[1].x, a[1].y, a[1].z, a[1].w};
      const float bv[8] = {b[0].x, b[0].y, b[0].z, b[0].w, b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[4 * ks + j], av[4 * ks + i], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
double run(const float* src, float* out, unsigned long long* st, int iters, bool print) {
  const int blocks = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<SHAPE><<<blocks, 512>>>(src, out, iters, st);
  hipEventRecord(a, 0);
  for (int r = 0; r < 10; ++r) k<SHAPE><<<blocks, 512>>>(src, out, iters, st);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  ms /= 10;
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  const double flop = (double)blocks * 8 * iters * 2.0 * 64 * 64 * 8;
  if (print) printf("  %dx%d: iters %5d  %7.1f us/launch  %6.1f TFLOP/s  in-kernel clock %.2f GHz  cycles/trip/wave %.0f\n", SHAPE, SHAPE, iters,
                    ms * 1e3, flop / ms / 1e9, cyc / rt * 0.1, cyc / blocks / iters);
  return ms;
}

int main() {
  float *src, *out; unsigned long long* st;
  const size_t n = 1 << 20;
  std::vector<float> h(n);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 4096 * 16);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int round = 0; round < 3; ++round) {
    printf("round %d\n", round);
    for (int iters : {128, 256, 1024}) {
      run<32>(src, out, st, iters, true);
      run<16>(src, out, st, iters, true);
    }
  }
  return 0;
}
