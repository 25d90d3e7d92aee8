// What can the memory system do for the replay gather's access pattern?  cfg #2 x 8: 65 536 samples per launch out of a 1 M x 896-B
// record ring (784 B used: 49 x 16-B chunks, one wave per record), written into two (rows, 128)-float tiles (416 + 352 B per row)
// plus two scalars.  Four kernels with the product kernel's wave / trip structure (R records in flight per wave, WPC waves per CU,
// grid-stride trips) and none of its arithmetic:
//   read   -- the random record reads only (values xor-folded, stored only if impossible)
//   write  -- the tile stores only (constants)
//   copy   -- both (the gather without normalisation)
//   seq    -- copy with records taken in ring order instead of at random (what the randomness costs)
// each timed back to back (50 launches in one hipGraph) and ISOLATED (16 x [384-MB eviction write, launch] minus 16 x [eviction]).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/gather_limit_probe tools/probes/gather_limit_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int REC_LD = 224, NCHUNK = 49, LD_SA = 128, O = 88, A = 16;   // floats

template <int MODE, int R>   // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ rec, const long long* __restrict__ idx, long long b,
                                               float* __restrict__ x_sa, float* __restrict__ xn_sa, float* __restrict__ rew,
                                               float* __restrict__ done, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long nwaves = (long long)gridDim.x * 4;
  const int c = lane * 4;
  float* dst = nullptr;
  bool rd = false;
  if (lane < 22) dst = x_sa + c;
  else if (lane < 44) dst = xn_sa + (c - 88);
  else if (lane < 48) dst = x_sa + O + (c - 176);
  else if (lane == 48) rd = true;
  float4 fold = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long r0 = wave * R; r0 < b; r0 += nwaves * R) {
    float4 v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      v[i] = make_float4(1.f, 2.f, 3.f, 4.f);
      if (MODE != 1) {
        const long long src = r0 + i < b ? idx[r0 + i] : 0;
        if (lane < NCHUNK) v[i] = reinterpret_cast<const float4*>(rec + src * REC_LD)[lane];
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const long long r = r0 + i;
      if (r >= b) break;
      if (MODE == 0) { fold.x += v[i].x; fold.y += v[i].y; fold.z += v[i].z; fold.w += v[i].w; continue; }
      if (dst) *reinterpret_cast<float4*>(dst + r * LD_SA) = v[i];
      if (rd) { rew[r] = v[i].x; done[r] = v[i].y; }
    }
  }
  if (MODE == 0 && fold.x + fold.y + fold.z + fold.w == 1.2345e30f) sink[threadIdx.x] = fold.x;
}

__global__ void k_fill(float* p, long long n, float v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (long long)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(p)[i] = make_float4(v, v, v, v);
}

static float replay_ms(hipGraphExec_t ge, hipStream_t st) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, st));
  std::vector<float> t;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return t[2];
}

int main() {
  const long long cap = 1000000, B = 65536;
  const int SETS = 4, NISO = 16, NB2B = 48;
  hipStream_t st; CK(hipStreamCreate(&st));
  float *rec, *evict, *sink; long long *idx, *seq;
  float *xs[SETS], *xn[SETS], *rw[SETS], *dn[SETS];
  CK(hipMalloc(&rec, cap * REC_LD * 4)); CK(hipMalloc(&evict, 384ll << 20)); CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&idx, (NB2B + 2) * B * 8)); CK(hipMalloc(&seq, B * 8));
  for (int s = 0; s < SETS; ++s) { CK(hipMalloc(&xs[s], B * LD_SA * 4)); CK(hipMalloc(&xn[s], B * LD_SA * 4)); CK(hipMalloc(&rw[s], B * 4)); CK(hipMalloc(&dn[s], B * 4)); }
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, rec, cap * REC_LD, 0.5f);
  std::vector<long long> h((NB2B + 2) * B), hs(B);
  unsigned long long x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (long long)(x % cap); }
  for (long long i = 0; i < B; ++i) hs[i] = i;
  CK(hipMemcpy(idx, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(seq, hs.data(), B * 8, hipMemcpyHostToDevice));
  CK(hipStreamSynchronize(st));
  const double alg = B * 1557.0, rd_b = B * (896.0 + 8), wr_b = B * 776.0;
  printf("cfg2x8: %lld rows, algorithmic %.1f MB; record lines read %.1f MB, tile bytes written %.1f MB\n", B, alg / 1e6, rd_b / 1e6, wr_b / 1e6);

  auto launch = [&](int mode, int R, int wpc, const long long* ix, int s) {
    long long fb = (B + 4 * R - 1) / (4 * R);
    if (fb > 64ll * wpc) fb = 64ll * wpc;
    dim3 g((unsigned)fb), t(256);
#define GO(M, RR) hipLaunchKernelGGL((k_probe<M, RR>), g, t, 0, st, rec, ix, B, xs[s], xn[s], rw[s], dn[s], sink)
#define GOR(M) do { if (R == 2) GO(M, 2); else if (R == 4) GO(M, 4); else GO(M, 8); } while (0)
    if (mode == 0) GOR(0); else if (mode == 1) GOR(1); else GOR(2);
  };
  auto capture = [&](auto body) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    body();
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    return ge;
  };
  auto ge_evict = capture([&] { for (int i = 0; i < NISO; ++i) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, evict, (384ll << 20) / 4, 0.f); });
  const char* names[4] = {"read ", "write", "copy ", "seq  "};
  for (int kind = 0; kind < 4; ++kind)
    for (int R : {2, 4, 8})
      for (int wpc : {8, 12, 16, 24, 32}) {
        const int mode = kind == 3 ? 2 : kind;
        auto ge_b = capture([&] { for (int i = 0; i < NB2B; ++i) launch(mode, R, wpc, kind == 3 ? seq : idx + (long long)(i + 2) * B, 0); });
        auto ge_i = capture([&] {
          for (int i = 0; i < NISO; ++i) {
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, evict, (384ll << 20) / 4, 0.f);
            launch(mode, R, wpc, kind == 3 ? seq : idx + (long long)(i + 2) * B, i % SETS);
          }
        });
        const float b2b = replay_ms(ge_b, st) / NB2B * 1e3f;
        float ti[3], te[3];
        for (int k = 0; k < 3; ++k) { ti[k] = replay_ms(ge_i, st); te[k] = replay_ms(ge_evict, st); }
        std::sort(ti, ti + 3); std::sort(te, te + 3);
        const float iso = (ti[1] - te[1]) / NISO * 1e3f;
        const double bytes = kind == 0 ? rd_b : kind == 1 ? wr_b : rd_b + wr_b;
        printf("%s R=%d wpc=%2d: b2b %6.2f us (%5.2f TB/s moved)   iso %6.2f us (%5.2f TB/s moved)\n", names[kind], R, wpc, b2b, bytes / b2b / 1e6,
               iso, bytes / iso / 1e6);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge_b)); CK(hipGraphExecDestroy(ge_i));
      }
  return 0;
}
