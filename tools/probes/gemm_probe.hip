// Times the k_gemm shapes of one V-learner / P-learner step in isolation (HIP events), for kernel tuning.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/probes/gemm_probe.hip -o tools/probes/gemm_probe
#include "../../pql_amd/csrc/gemm.hip"
#include <cstdio>
extern "C" int64_t pqlk_ld(int64_t cols) { return pqlk_round_up(cols < 1 ? 1 : cols, 32); }
#include <cstdlib>
#include <vector>

static float* dalloc(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
  float* d;
  if (hipMalloc(&d, n * 4) != hipSuccess) { printf("alloc fail\n"); exit(1); }
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

template <typename F>
static float time_us(F f, int iters = 30) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8192;
  struct Sh { const char* name; int mode, M, N, K, groups, epi, splits; };
  // FWD: C[M=B, N=out] over K=in ; DX: C[M=B, N=in] over K=out ; DW: C[M=out, N=in] over K=B
  std::vector<Sh> shapes = {
      {"fwd  128->512 x2 elu", MODE_FWD, B, 512, 128, 2, EPI_ELU, 1},
      {"fwd  128->512 x2 none", MODE_FWD, B, 512, 128, 2, EPI_NONE, 1},
      {"fwd  512->512 x2 elu", MODE_FWD, B, 512, 512, 2, EPI_ELU, 1},
      {"fwd  512->512 x2 none", MODE_FWD, B, 512, 512, 2, EPI_NONE, 1},
      {"fwd  512->256 x2 elu", MODE_FWD, B, 256, 512, 2, EPI_ELU, 1},
      {"fwd  256->1   x2 none", MODE_FWD, B, 1, 256, 2, EPI_NONE, 1},
      {"fwd  96->512  x1 elu", MODE_FWD, B, 512, 96, 1, EPI_ELU, 1},
      {"fwd  512->512 x1 elu", MODE_FWD, B, 512, 512, 1, EPI_ELU, 1},
      {"fwd  512->256 x1 elu", MODE_FWD, B, 256, 512, 1, EPI_ELU, 1},
      {"fwd  256->16  x1 tanh", MODE_FWD, B, 16, 256, 1, EPI_TANH, 1},
      {"dx   512<-512 x2 delu", MODE_DX, B, 512, 512, 2, EPI_DELU, 1},
      {"dx   512<-256 x2 delu", MODE_DX, B, 512, 256, 2, EPI_DELU, 1},
      {"dx   256<-1   x2 delu", MODE_DX, B, 256, 1, 2, EPI_DELU, 1},
      {"dw   512x512  x2 s16", MODE_DW, 512, 512, B, 2, EPI_NONE, 16},
      {"dw   256x512  x2 s16", MODE_DW, 256, 512, B, 2, EPI_NONE, 16},
      {"dw   512x128  x2 s16", MODE_DW, 512, 128, B, 2, EPI_NONE, 16},
      {"dw   1x256    x2 s16", MODE_DW, 1, 256, B, 2, EPI_NONE, 16},
  };
  const size_t big = (size_t)2 * B * 512 + 4096;
  float* A = dalloc(big, 1.f);
  float* Bm = dalloc(big, 0.05f);
  float* Cm = dalloc((size_t)17 * 2 * 512 * 512 + big, 0.f);
  float* aux = dalloc(big, 1.f);
  float* bias = dalloc(4096, 0.1f);
  double tot_us = 0, tot_fl = 0;
  for (auto& s : shapes) {
    GemmP p = {};
    p.A = A; p.B = Bm; p.C = Cm; p.bias = bias; p.aux = aux; p.epi = s.epi;
    p.groups = s.groups; p.M = s.M; p.N = s.N; p.K = s.K;
    double flops;
    int gz = s.groups;
    if (s.mode == MODE_FWD) {
      const int ldk = (int)pqlk_ld(s.K), ldn = (int)pqlk_ld(s.N);
      p.K = ldk; p.lda = ldk; p.ldb = ldk; p.ldc = ldn; p.ncols_store = ldn;
      p.sA = 0; p.sB = (long long)s.N * ldk; p.sC = (long long)B * ldn; p.sBias = 1024;
      flops = 2.0 * s.M * s.N * s.K * s.groups;
    } else if (s.mode == MODE_DX) {
      const int ldk = (int)pqlk_ld(s.K), ldn = (int)pqlk_ld(s.N);
      p.lda = ldk; p.ldb = ldn; p.ldc = ldn; p.ldaux = ldn; p.ncols_store = ldn;
      p.sA = (long long)B * ldk; p.sB = (long long)s.K * ldn; p.sC = (long long)B * ldn; p.sAux = (long long)B * ldn;
      flops = 2.0 * s.M * s.N * s.K * s.groups;
    } else {
      const int ldm = (int)pqlk_ld(s.M), ldn = (int)pqlk_ld(s.N);
      p.lda = ldm; p.ldb = ldn; p.ldc = ldn; p.N = ldn; p.ncols_store = ldm;
      p.sA = (long long)B * ldm; p.sB = (long long)B * ldn; p.sC = 512 * 512; p.sBias = 512 * 512;
      p.dbias = Cm + 16 * 2 * 512 * 512;
      p.splits = s.splits; p.rows_per_split = (int)pqlk_round_up((B + s.splits - 1) / s.splits, KT_MAX);
      p.sSplit = 2 * 512 * 512 + 2048;
      gz = s.groups * s.splits;
      flops = 2.0 * s.M * s.N * (double)s.K * s.groups;
    }
    int rc = 0;
    float us = time_us([&] {
      if (s.mode == MODE_FWD) rc = launch_auto<MODE_FWD>(p, gz, 0);
      else if (s.mode == MODE_DX) rc = launch_auto<MODE_DX>(p, gz, 0);
      else rc = launch_auto<MODE_DW>(p, gz, 0);
    });
    if (rc) printf("rc=%d\n", rc);
    printf("%-24s %8.1f us  %7.2f GFLOP  %6.1f TF/s\n", s.name, us, flops / 1e9, flops / us / 1e6);
    tot_us += us; tot_fl += flops;
  }
  printf("TOTAL %.1f us %.2f GFLOP %.1f TF/s\n", tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6);
  {  // fused hidden-layer forward vs the per-layer path, twin critic 128->512->512->256->1 and actor 96->..->16
    for (int cfg = 0; cfg < 2; ++cfg) {
      PqlMlpDesc d = {};
      d.n_layers = 4; d.n_nets = cfg == 0 ? 2 : 1;
      const int dm0[5] = {104, 512, 512, 256, 1}, dm1[5] = {88, 512, 512, 256, 16};
      for (int i = 0; i < 5; ++i) d.dims[i] = cfg == 0 ? dm0[i] : dm1[i];
      float* params = dalloc(pqlk_mlp_param_floats(&d), 0.05f);
      float* packed = dalloc(pqlk_mlp_packed_floats(&d), 0.f);
      float* acts = dalloc(pqlk_mlp_acts_floats(&d, B), 0.f);
      float* x = dalloc((size_t)B * 128, 1.f);
      pqlk_mlp_pack(&d, params, packed, 0);
      float u0 = time_us([&] { pqlk_mlp_forward(&d, params, nullptr, 1, x, 128, B, 0, nullptr, 0, 0, acts, nullptr, 0, 0); });
      float u1 = time_us([&] { pqlk_mlp_forward(&d, params, packed, 1, x, 128, B, 0, nullptr, 0, 0, acts, nullptr, 0, 0); });
      float u2 = time_us([&] { pqlk_mlp_forward(&d, params, packed, 0, x, 128, B, 0, nullptr, 0, 0, acts, nullptr, 0, 0); });
      float u3 = time_us([&] { pqlk_mlp_pack(&d, params, packed, 0); });
      printf("mlp fwd nets=%d: per-layer %.1f us | fused+stash %.1f us | fused no-stash %.1f us | pack %.1f us\n", d.n_nets, u0, u1, u2, u3);
    }
  }
  {  // skinny last-layer kernels
    struct Sk { const char* name; int kind, N, K, groups; };
    for (Sk k : {Sk{"skinny fwd N=1 x2", 0, 1, 256, 2}, Sk{"skinny dx  N=1 x2", 1, 1, 256, 2}, Sk{"skinny dx  N=16 x1", 1, 16, 256, 1},
                 Sk{"skinny dw  N=1 x2", 2, 1, 256, 2}, Sk{"skinny dw  N=16 x1", 2, 16, 256, 1}}) {
      SkinnyP q = {};
      q.X = A; q.ldx = k.K; q.sX = (long long)B * k.K; q.W = Bm; q.ldk = k.K; q.sW = 32 * 1024; q.bias = bias; q.sBias = 1024;
      q.C = Cm; q.ldc = 32; q.sC = (long long)B * (k.kind == 1 ? k.K : 32); q.dY = aux; q.ldy = 32; q.sY = (long long)B * 32;
      q.M = B; q.N = k.N; q.K = k.K; q.epi = k.kind == 1 ? SK_EPI_DELU : SK_EPI_NONE;
      q.dW = Cm; q.dB = Cm + 16 * 2 * 512 * 512; q.sSplit = 2 * 512 * 512 + 2048; q.splits = 16;
      q.rows_per_split = (int)pqlk_round_up((B + 15) / 16, 32);
      float us = time_us([&] {
        if (k.kind == 0) launch_skinny_fwd(q, k.groups, 0);
        else if (k.kind == 1) launch_skinny_dx(q, k.groups, 0);
        else launch_skinny_dw(q, k.groups, 0);
      });
      printf("%-22s %8.1f us\n", k.name, us);
    }
  }
  {  // two streams
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    float* C2b = dalloc((size_t)2 * B * 512, 0.f);
    auto mk = [&](float* Cout, int K) {
      GemmP p = {};
      const int ldk = (int)pqlk_ld(K);
      p.A = A; p.B = Bm; p.C = Cout; p.bias = bias; p.aux = aux; p.epi = EPI_ELU; p.groups = 2; p.M = B; p.N = 512; p.K = ldk;
      p.lda = ldk; p.ldb = ldk; p.ldc = 512; p.ncols_store = 512; p.sA = 0; p.sB = (long long)512 * ldk; p.sC = (long long)B * 512; p.sBias = 1024;
      return p;
    };
    for (int K : {128, 512}) {
      GemmP pa = mk(Cm, K), pb = mk(C2b, K);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int n = 20;
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(e0, sa);
        for (int i = 0; i < 2 * n; ++i) launch_auto<MODE_FWD>(pa, 2, sa);
        hipEventRecord(e1, sa); hipEventSynchronize(e1);
        float ms_seq; hipEventElapsedTime(&ms_seq, e0, e1);
        hipDeviceSynchronize();
        hipEvent_t fb; hipEventCreate(&fb);
        hipEventRecord(e0, sa);
        hipStreamWaitEvent(sb, e0, 0);
        for (int i = 0; i < n; ++i) { launch_auto<MODE_FWD>(pa, 2, sa); launch_auto<MODE_FWD>(pb, 2, sb); }
        hipEventRecord(fb, sb); hipStreamWaitEvent(sa, fb, 0);
        hipEventRecord(e1, sa); hipEventSynchronize(e1);
        float ms_par; hipEventElapsedTime(&ms_par, e0, e1);
        if (rep) printf("K=%d fwd x2: %d launches back-to-back %.1f us/launch ; on two streams %.1f us/launch\n", K, 2 * n, ms_seq * 1e3 / (2 * n), ms_par * 1e3 / (2 * n));
      }
    }
  }
  return 0;
}
// (appended) two-stream concurrency check: are two independent GEMM chains faster side by side than back to back?
