// Are fp32 '/', sqrtf, expm1f-free ops correctly rounded on gfx950 with our flags?  Compares device vs host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* a, const float* b, float* d, float* s, float* n, int N, float eps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  d[i] = a[i] / b[i];
  s[i] = sqrtf(b[i]);
  n[i] = (a[i] - 0.25f) / sqrtf(b[i] + eps);
}
int main() {
  const int N = 1 << 20;
  std::vector<float> a(N), b(N), d(N), s(N), n(N);
  srand(1);
  for (int i = 0; i < N; ++i) { a[i] = (rand() / (float)RAND_MAX) * 60.f - 30.f; b[i] = (rand() / (float)RAND_MAX) * 4.f + 0.01f; }
  float *da, *db, *dd, *ds, *dn;
  hipMalloc(&da, N * 4); hipMalloc(&db, N * 4); hipMalloc(&dd, N * 4); hipMalloc(&ds, N * 4); hipMalloc(&dn, N * 4);
  hipMemcpy(da, a.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), N * 4, hipMemcpyHostToDevice);
  k<<<N / 256, 256>>>(da, db, dd, ds, dn, N, 1e-4f);
  hipMemcpy(d.data(), dd, N * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, N * 4, hipMemcpyDeviceToHost);
  hipMemcpy(n.data(), dn, N * 4, hipMemcpyDeviceToHost);
  int bd = 0, bs = 0, bn = 0;
  for (int i = 0; i < N; ++i) {
    volatile float hd = a[i] / b[i]; volatile float hs = sqrtf(b[i]);
    volatile float t = b[i] + 1e-4f; volatile float hs2 = sqrtf(t); volatile float u = a[i] - 0.25f; volatile float hn = u / hs2;
    bd += hd != d[i]; bs += hs != s[i]; bn += hn != n[i];
  }
  printf("mismatch div=%d sqrt=%d norm=%d of %d\n", bd, bs, bn, N);
  return 0;
}
