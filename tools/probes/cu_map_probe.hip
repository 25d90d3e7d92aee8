// Which blocks of a 512-block, two-per-CU launch share a CU?  (HW_ID / XCC_ID of every block, in dispatch order.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/cu_map_probe tools/probes/cu_map_probe.hip && tools/probes/bin/cu_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, long long* t) {
  extern __shared__ float sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = wall_clock64();
  sm[threadIdx.x] = (float)hw;
  __syncthreads();
  // stay resident long enough for the whole grid to be placed
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; t[blockIdx.x] = t0; }
}
int main() {
  const int n = 512;
  unsigned* d; long long* dt;
  hipMalloc(&d, n * 8); hipMalloc(&dt, n * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 70 * 1024, 0, d, dt);
  std::vector<unsigned> h(2 * n); std::vector<long long> ht(n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < n; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned key = (xcc << 16) | (hw & 0xff00);   // cu_id[11:8], sh_id[12], se_id[15:13]
    cu[key].push_back(b);
  }
  printf("%zu distinct CUs for %d blocks\n", cu.size(), n);
  std::map<int, int> delta;
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ < 12) { printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
  }
  for (auto& kv : delta) printf("pairs with block distance %d: %d\n", kv.first, kv.second);
  long long tmin = ht[0]; for (auto x : ht) tmin = x < tmin ? x : tmin;
  printf("start time (100 MHz ticks) of blocks 0, 8, 255, 256, 264, 511 after the first: %lld %lld %lld %lld %lld %lld\n", ht[0] - tmin, ht[8] - tmin,
         ht[255] - tmin, ht[256] - tmin, ht[264] - tmin, ht[511] - tmin);
  return 0;
}
