// Times pqlk_replay_gather_fused in isolation: random vs sequential indices, batch sizes, cfg #2 / #5 record shapes.
#include "../../pql_amd/csrc/replay.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
  struct Cfg { int O, A; int64_t cap; int64_t B; };
  for (Cfg c : {Cfg{88, 16, 1000000, 8192}, Cfg{88, 16, 1000000, 32768}, Cfg{108, 21, 5000000, 32768}, Cfg{211, 20, 2000000, 8192}}) {
    RecLayout L = rec_layout(c.O, c.A);
    float* rec; hipMalloc(&rec, (size_t)c.cap * L.ld * 4); hipMemset(rec, 0, (size_t)c.cap * L.ld * 4);
    const int ld_sa = (int)pqlk_ld(c.O + c.A), ld_o = (int)pqlk_ld(c.O);
    float *x_sa, *xn_sa, *xn_o, *rew, *done, *mean, *var;
    hipMalloc(&x_sa, c.B * ld_sa * 4); hipMalloc(&xn_sa, c.B * ld_sa * 4); hipMalloc(&xn_o, c.B * ld_o * 4);
    hipMalloc(&rew, c.B * 4); hipMalloc(&done, c.B * 4); hipMalloc(&mean, 4096); hipMalloc(&var, 4096);
    hipMemset(mean, 0, 4096);
    std::vector<float> ones(1024, 1.f); hipMemcpy(var, ones.data(), 4096, hipMemcpyHostToDevice);
    const int NI = 24;
    std::vector<int64_t> hi((size_t)NI * c.B), hs((size_t)NI * c.B);
    srand(7);
    for (size_t i = 0; i < hi.size(); ++i) { hi[i] = ((int64_t)rand() * 2147483648LL + rand()) % c.cap; hs[i] = (int64_t)(i % c.cap); }
    int64_t *di, *ds; hipMalloc(&di, hi.size() * 8); hipMalloc(&ds, hs.size() * 8);
    hipMemcpy(di, hi.data(), hi.size() * 8, hipMemcpyHostToDevice); hipMemcpy(ds, hs.data(), hs.size() * 8, hipMemcpyHostToDevice);
    PqlReplayDesc d = {rec, c.cap, c.O, c.A, L.ld, 0};
    for (int mode = 0; mode < 3; ++mode) {
      const int64_t* ix = mode == 1 ? ds : di;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      auto run = [&](int i) {
        pqlk_replay_gather_fused(&d, ix + (size_t)i * c.B, c.B, mode == 2 ? nullptr : mean, mode == 2 ? nullptr : var, 1e-4f, 1, x_sa, ld_sa, xn_sa, xn_o,
                                 ld_o, rew, done, nullptr);
      };
      for (int i = 0; i < 4; ++i) run(i);
      hipEventRecord(a, 0);
      for (int i = 4; i < NI; ++i) run(i);
      hipEventRecord(b, 0); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double us = ms * 1e3 / (NI - 4);
      const double alg = (double)c.B * ((2 * c.O + c.A) * 4 + 4 + 1 + 8 + (2 * c.O + c.A) * 4 + 8);
      printf("O=%d A=%d cap=%lld B=%lld %-10s %7.2f us  %7.1f GB/s algorithmic (rec %d B)\n", c.O, c.A, (long long)c.cap, (long long)c.B,
             mode == 0 ? "random" : (mode == 1 ? "sequential" : "rand-nonorm"), us, alg / us / 1e3, L.ld * 4);
    }
    hipFree(rec); hipFree(x_sa); hipFree(xn_sa); hipFree(xn_o); hipFree(di); hipFree(ds);
  }
  return 0;
}
