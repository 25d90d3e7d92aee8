// BatchNorm1d + ELU for the CrossQ critic: `Linear -> BatchNorm1d -> ELU` blocks of
// create_simple_mlp(use_batchnorm=True) (pql/models/mlp.py:15-24) as used by DoubleQBatchNorm (:224-241) and
// AgentCrossQ.update_critic / update_actor (pql/algo/crossQ.py:144-166).  The Linear parts run on the existing fp32-MFMA
// GEMMs (a one-layer PqlMlpDesc); batch statistics come from pqlk_batch_moments; what is new is the normalise+activate
// pass, its backward and the column sums the backward needs.  All HBM-bound passes over (M, cols) fp32.
//
// torch semantics followed (ATen native batch_norm, training): invstd = 1 / sqrt(biased_var + eps),
// y = x * (gamma * invstd) + (beta - mean * gamma * invstd); running_mean/var <- (1 - momentum) * running + momentum *
// (mean, UNBIASED var).  Eval: the running statistics replace the batch ones.
#include "pqlk_common.h"

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1f(x); }   // nn.ELU(alpha=1): libm expm1 like torch CPU

__global__ __launch_bounds__(256) void k_bn_elu_fwd(const float* __restrict__ z, int64_t ld, int64_t m, int cols,
                                                    const float* __restrict__ mean, const float* __restrict__ var_unb,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                    int training, float momentum, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var, float* __restrict__ y) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  if (c >= cols) return;
  float mu, var;
  if (training) {
    mu = mean[c];
    var = var_unb[c] * ((float)(m - 1) / (float)m);   // biased variance normalises
  } else {
    mu = running_mean[c];
    var = running_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float w = gamma[c] * invstd, b = beta[c] - mu * w;
  const int64_t rows_per = (m + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(m, r0 + rows_per);
  for (int64_t r = r0 + rl; r < r1; r += 4) y[r * ld + c] = elu_f(z[r * ld + c] * w + b);
  if (training && blockIdx.y == 0 && rl == 0 && running_mean) {   // one writer per column; readers of running_* are the eval path only
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var_unb[c];
  }
}

// column sums of g = dy * ELU'(pre) and g * xhat over a chunk of rows -> part[(chunk * 2 + {0,1}) * cols + c]
__global__ __launch_bounds__(256) void k_bn_bwd_sums(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                                     int64_t ld, int64_t m, int cols, const float* __restrict__ mean,
                                                     const float* __restrict__ var_unb, float eps, float* __restrict__ part) {
  __shared__ float sh[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < cols) {
    const float mu = mean[c];
    const float invstd = 1.0f / sqrtf(var_unb[c] * ((float)(m - 1) / (float)m) + eps);
    const int64_t rows_per = (m + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(m, r0 + rows_per);
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float yy = y[r * ld + c];
      const float g = dy[r * ld + c] * (yy > 0.f ? 1.f : yy + 1.f);
      s1 += g;
      s2 += g * ((z[r * ld + c] - mu) * invstd);
    }
  }
  sh[0][rl][cl] = s1; sh[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < cols) {
    part[((int64_t)blockIdx.y * 2 + 0) * cols + c] = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
    part[((int64_t)blockIdx.y * 2 + 1) * cols + c] = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
  }
}

// fold the chunk partials in order -> dbeta (= s1), dgamma (= s2); then dz = gamma * invstd * (g - s1/m - xhat * s2/m)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                                      int64_t ld, int64_t m, int cols, const float* __restrict__ mean,
                                                      const float* __restrict__ var_unb, const float* __restrict__ gamma, float eps,
                                                      const float* __restrict__ part, int chunks, float* __restrict__ dz,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  if (c >= cols) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < chunks; ++k) { s1 += part[((int64_t)k * 2 + 0) * cols + c]; s2 += part[((int64_t)k * 2 + 1) * cols + c]; }
  if (blockIdx.y == 0 && rl == 0) {
    if (dbeta) dbeta[c] = s1;
    if (dgamma) dgamma[c] = s2;
  }
  const float mu = mean[c];
  const float invstd = 1.0f / sqrtf(var_unb[c] * ((float)(m - 1) / (float)m) + eps);
  const float k0 = gamma[c] * invstd, a1 = s1 / (float)m, a2 = s2 / (float)m;
  const int64_t rows_per = (m + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(m, r0 + rows_per);
  for (int64_t r = r0 + rl; r < r1; r += 4) {
    const float yy = y[r * ld + c];
    const float g = dy[r * ld + c] * (yy > 0.f ? 1.f : yy + 1.f);
    dz[r * ld + c] = k0 * (g - a1 - ((z[r * ld + c] - mu) * invstd) * a2);
  }
}

#define BN_CHUNKS 64

extern "C" int pqlk_bn_elu_forward(const float* z, int64_t ld, int64_t m, int32_t cols, const float* mean, const float* var_unbiased,
                                   const float* gamma, const float* beta, float eps, int32_t training, float momentum,
                                   float* running_mean, float* running_var, float* y, pqlk_stream_t stream) {
  PQLK_REQUIRE(z && gamma && beta && y, PQLK_E_NULL);
  PQLK_REQUIRE(training ? (mean && var_unbiased) : (running_mean && running_var), PQLK_E_NULL);
  PQLK_REQUIRE(m > 1 && cols > 0 && ld >= cols, PQLK_E_SHAPE);
  hipLaunchKernelGGL(k_bn_elu_fwd, dim3((unsigned)((cols + 63) / 64), BN_CHUNKS), dim3(256), 0, pqlk_s(stream), z, ld, m, (int)cols, mean,
                     var_unbiased, gamma, beta, eps, (int)training, momentum, running_mean, running_var, y);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

extern "C" int pqlk_bn_elu_backward(const float* dy, const float* y, const float* z, int64_t ld, int64_t m, int32_t cols,
                                    const float* mean, const float* var_unbiased, const float* gamma, float eps, float* dz,
                                    float* dgamma, float* dbeta, float* scratch, pqlk_stream_t stream) {
  PQLK_REQUIRE(dy && y && z && mean && var_unbiased && gamma && dz && scratch, PQLK_E_NULL);
  PQLK_REQUIRE(m > 1 && cols > 0 && ld >= cols, PQLK_E_SHAPE);
  const dim3 grid((unsigned)((cols + 63) / 64), BN_CHUNKS);
  hipLaunchKernelGGL(k_bn_bwd_sums, grid, dim3(256), 0, pqlk_s(stream), dy, y, z, ld, m, (int)cols, mean, var_unbiased, eps, scratch);
  PQLK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_bwd_apply, grid, dim3(256), 0, pqlk_s(stream), dy, y, z, ld, m, (int)cols, mean, var_unbiased, gamma, eps, scratch,
                     BN_CHUNKS, dz, dgamma, dbeta);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}
