// Times k_mlp_fwd_fused alone on the critic (2 nets) and actor (1 net) shapes of cfg #2, for kernel tuning.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off [-DPQLK_FP_CLK] tools/probes/fused_probe.hip -o fused_probe
#if defined(PQLK_FP_CLK)
long long* g_fp_clk = nullptr;
#endif
#include "../../pql_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" int64_t pqlk_ld(int64_t cols) { return pqlk_round_up(cols < 1 ? 1 : cols, 32); }

static float* dalloc(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
  float* d;
  if (hipMalloc(&d, n * 4) != hipSuccess) { printf("alloc fail\n"); exit(1); }
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8192;
  struct Case { const char* name; int nets, in, stash; } cases[] = {{"critic 104->512->512->256 x2 stash", 2, 104, 1},
                                                                     {"target 104->512->512->256 x2", 2, 104, 0},
                                                                     {"actor   88->512->512->256 x1", 1, 88, 0}};
#if defined(PQLK_FP_CLK)
  hipMalloc(&g_fp_clk, 128 * sizeof(long long));
#endif
  for (auto& c : cases) {
    PqlMlpDesc d = {};
    d.n_layers = 4; d.n_nets = c.nets;
    d.dims[0] = c.in; d.dims[1] = 512; d.dims[2] = 512; d.dims[3] = 256; d.dims[4] = 1;
    const int64_t ldx = pqlk_ld(c.in);
    float* params = dalloc((size_t)pqlk_mlp_param_floats(&d), 0.05f);
    float* packed = dalloc((size_t)pqlk_mlp_packed_floats(&d), 0.05f);
    float* x = dalloc((size_t)B * ldx, 1.f);
    float* acts = dalloc((size_t)pqlk_mlp_acts_floats(&d, B), 0.f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_fused_hidden(&d, params, packed, x, ldx, B, acts, c.stash, 0);
    const int iters = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_fused_hidden(&d, params, packed, x, ldx, B, acts, c.stash, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double flops = 2.0 * B * c.nets * ((double)ldx * 512 + 512.0 * 512 + 512.0 * 256);
    printf("%-40s %8.1f us  %6.1f TFLOP/s\n", c.name, us, flops / us * 1e-6);
#if defined(PQLK_FP_CLK)
    long long hc[128];
    hipMemcpy(hc, g_fp_clk, sizeof(hc), hipMemcpyDeviceToHost);
    printf("   block 8 wave 0: staging %lld ticks\n", hc[1] - hc[0]);
    for (int l = 0; l < 3; ++l) {
      const long long* t = hc + 8 + 8 * l;
      printf("   layer %d: main %lld | prefetch %lld | barrier A %lld | epilogue %lld | barrier B %lld   (start +%lld)\n", l, t[1] - t[0],
             0LL, t[2] - t[1], t[3] - t[2], t[4] - t[3], t[0] - hc[0]);
    }
#endif
    hipFree(params); hipFree(packed); hipFree(x); hipFree(acts);
  }
  return 0;
}
