// Can the fp32 vector pipe add throughput beside the fp32 matrix pipe?  One block = MW waves issuing independent
// v_mfma_f32_32x32x2_f32 + VW waves issuing independent v_pk_fma_f32, operands in registers (no memory traffic at all), one
// block per CU.  Reports the MFMA rate alone, the packed-FMA rate alone, and both together, with the in-kernel clock.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_probe.hip -o tools/probes/bin/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

// mode bit 0: MFMA waves work, bit 1: VALU waves work
template <int MW, int VW>
__global__ __launch_bounds__(64 * (MW + VW)) void k(float* out, int iters, int mode, unsigned long long* stamps) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (wave < MW) {
    if (mode & 1) {
      f32x16 acc[4];
      for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
      float x = threadIdx.x * 0.001f + 0.5f, y = blockIdx.x * 0.0001f + 0.25f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
      }
      for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    }
  } else if (mode & 2) {
    f2 acc[16];
    for (int a = 0; a < 16; ++a) acc[a] = f2{0.f, 0.f};
    f2 x = f2{threadIdx.x * 0.001f + 0.5f, threadIdx.x * 0.002f + 0.25f}, y = f2{blockIdx.x * 0.0001f + 0.25f, 0.75f};
    asm volatile("" : "+v"(x), "+v"(y));
    // per MFMA-wave iteration (8 x 4 MFMAs x 64 cycles = 2048 cycles) a VALU wave that kept its SIMD's vector pipe busy would
    // issue 512 packed FMAs; give it exactly that many so both roles finish together when nothing is in the way
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 32; ++u)
#pragma unroll
        for (int a = 0; a < 16; ++a) acc[a] = __builtin_elementwise_fma(x, y, acc[a]);
    }
    for (int a = 0; a < 16; ++a) s += acc[a][0] + acc[a][1];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 * (MW + VW) + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MW, int VW>
void run(int mode, int iters, float* out, unsigned long long* st) {
  const int blocks = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MW, VW><<<blocks, 64 * (MW + VW)>>>(out, iters, mode, st);
  hipEventRecord(a, 0);
  k<MW, VW><<<blocks, 64 * (MW + VW)>>>(out, iters, mode, st);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  const double mf = (mode & 1) ? (double)blocks * MW * iters * 32 * 2.0 * 32 * 32 * 2 : 0.0;
  const double vf = (mode & 2) ? (double)blocks * VW * iters * 512 * 64 * 2 * 2.0 : 0.0;
  printf("MFMA waves/SIMD %.1f  VALU waves/SIMD %.1f  mode %d: %8.1f us   MFMA %6.1f TF/s + packed FMA %6.1f TF/s = %6.1f   clock %.2f GHz\n",
         MW / 4.0, VW / 4.0, mode, ms * 1e3, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9, cyc / rt * 0.1);
}

int main() {
  float* out; unsigned long long* st;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 4096 * 16);
  const int iters = 2048;
  for (int it : {32, 64, 128, 512}) {   // short launches: 8 MFMA waves per CU, iters x 2048 cycles each
    printf("iters %d: ", it);
    run<8, 4>(1, it, out, st);
  }
  for (int mode : {1, 2, 3}) run<4, 4>(mode, iters, out, st);
  for (int mode : {1, 2, 3}) run<8, 4>(mode, iters, out, st);
  for (int mode : {1, 2, 3}) run<8, 8>(mode, iters, out, st);
  run<8, 4>(1, iters, out, st);
  for (int it : {32, 64, 128, 512}) {   // the same short launches right after 7 ms of sustained MFMA load
    printf("warm, iters %d: ", it);
    run<8, 4>(1, it, out, st);
  }
  return 0;
}
