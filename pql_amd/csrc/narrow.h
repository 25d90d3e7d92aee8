// fp32-MFMA kernels for GEMMs with a NARROW output (<= 64 columns) and a long row axis: the actor's action head
// (256 -> 16..21, or 2A for SAC), the C51 logits head (-> 51).  k_gemm's 64x64 block tile wastes 3/4 of its MFMAs on a
// 16-wide head, launches only M/64 = 128 blocks at batch 8192 and pays a full LDS pipeline for 16 KB of weights.  Here
// one wave owns a 32-row tile and NT 32-column output tiles; BOTH operands come straight from global memory as the
// k-contiguous float4 quads the MFMA fragments want (x row r / weight row n, reduction index 8 k8 + 4 h + t): the weight
// block is a few KB and lives in L1/L2, no LDS, no barrier.  Same k-order and operand roles as k_gemm, so results are
// bitwise equal to the generic path.
#pragma once
#include "pqlk_common.h"

template <int NT, int EPI>
__global__ __launch_bounds__(64) void k_fwd_narrow(GemmP p) {
  typedef float acc_t __attribute__((ext_vector_type(16)));
  constexpr int D = 4;   // ring depth (reduction steps of 8)
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int g = blockIdx.y;
  const int m0 = blockIdx.x * 32;
  const int row = min(m0 + r, p.M - 1);   // clamped for the loads; stores are guarded
  const float4* xp = reinterpret_cast<const float4*>(p.A + (long long)g * p.sA + (long long)row * p.lda) + h;
  const float4* wp[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)   // weight rows past N are clamped: their output columns are never stored
    wp[j] = reinterpret_cast<const float4*>(p.B + (long long)g * p.sB + (long long)min(32 * j + r, p.N - 1) * p.ldb) + h;
  acc_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const int K8 = p.K >> 3;   // K is a multiple of 32
  float4 xq[D], wq[D][NT];
#pragma unroll
  for (int s = 0; s < D; ++s) {
    xq[s] = xp[2 * s];
#pragma unroll
    for (int j = 0; j < NT; ++j) wq[s][j] = wp[j][2 * s];
  }
  for (int k8 = 0; k8 < K8; k8 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const float av[4] = {xq[s].x, xq[s].y, xq[s].z, xq[s].w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = t == 0 ? wq[s][j].x : t == 1 ? wq[s][j].y : t == 2 ? wq[s][j].z : wq[s][j].w;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av[t], acc[j], 0, 0, 0);
        }
      const int kn = min(k8 + s + D, K8 - 1);   // clamped, unconditional refill (counted waits)
      xq[s] = xp[2 * kn];
#pragma unroll
      for (int j = 0; j < NT; ++j) wq[s][j] = wp[j][2 * kn];
    }
  }
  // epilogue: lane (r, h) owns row m0 + r, columns 32 j + 8 q + 4 h + {0..3}
  const int orow = m0 + r;
  if (orow >= p.M) return;
  float* C = p.C + (long long)g * p.sC;
  const float* bias = p.bias ? p.bias + (long long)g * p.sBias : nullptr;
  const float* aux = p.aux ? p.aux + (long long)g * p.sAux : nullptr;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = 32 * j + 8 * q + 4 * h + u;
        if (c >= p.ncols_store) continue;
        float x = 0.f;   // pad column
        if (c < p.N) {
          x = acc[j][4 * q + u] + (bias ? bias[c] : 0.f);
          if (EPI == EPI_TANH) x = tanhf(x);
          else if (EPI == EPI_TANH_NOISE) {
            x = tanhf(x);
            float nz = p.noise_std * aux[(long long)orow * p.N + c];
            nz = fminf(fmaxf(nz, -p.noise_clip), p.noise_clip);
            x = fminf(fmaxf(x + nz, -1.f), 1.f);
          }
          if (p.C2 && g == 0) p.C2[(long long)orow * p.ldc2 + c] = x;
        }
        C[(long long)orow * p.ldc + c] = x;
      }
}

static bool narrow_fwd_ok(const GemmP& p) { return p.N <= 64 && p.K >= 32 && (p.K & 31) == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0; }

template <int EPI>
static int launch_fwd_narrow_e(const GemmP& p, int groups, hipStream_t st) {
  const dim3 grid((unsigned)((p.M + 31) / 32), (unsigned)groups), block(64);
  if (p.N <= 32) hipLaunchKernelGGL((k_fwd_narrow<1, EPI>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((k_fwd_narrow<2, EPI>), grid, block, 0, st, p);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

static int launch_fwd_narrow(const GemmP& p, int groups, hipStream_t st) {
  switch (p.epi) {
    case EPI_NONE: return launch_fwd_narrow_e<EPI_NONE>(p, groups, st);
    case EPI_TANH: return launch_fwd_narrow_e<EPI_TANH>(p, groups, st);
    case EPI_TANH_NOISE: return launch_fwd_narrow_e<EPI_TANH_NOISE>(p, groups, st);
    default: return PQLK_E_UNSUPPORTED;
  }
}
