// Loss kernels: TD target + twin MSE, C51 softmax/projection/BCE, DPG actor loss.
// Reference: pql/algo/pql_v_learner.py:80-108, pql/utils/distl_util.py:4-20, pql/algo/pql_p_learner.py:55-57,
// pql/models/mlp.py:256-263.  Each kernel writes the gradient w.r.t. the critic's last-layer output
// (padded (2,B,ld) layout expected by pqlk_mlp_backward) and deterministic per-block loss partials.
#include "pqlk_common.h"

#define LOSS_MAX_BLOCKS 1024

// sum `n` partials in fixed order with one block -> out[0] = scale * sum  (strided per-thread sums -> xor-shuffle tree per wave
// -> the four waves in order: the order in which k_adamw folds the same partials when the loss rides in its launch)
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ part, int n, float scale,
                                                      float* __restrict__ out, const int32_t* __restrict__ slot_dev,
                                                      int ring_len) {
  __shared__ float shw[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[slot_dev ? (slot_dev[0] % ring_len) : 0] = ((shw[0] + shw[1]) + (shw[2] + shw[3])) * scale;
}

__device__ __forceinline__ float block_sum_256(float v) {
  __shared__ float shw[4];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = v;
  __syncthreads();
  return (shw[0] + shw[1]) + (shw[2] + shw[3]);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_td_mse(const float* __restrict__ q, const float* __restrict__ qt, int64_t ld,
                                                const float* __restrict__ rew, const float* __restrict__ done,
                                                float gamma_n, int64_t b, float* __restrict__ dy,
                                                float* __restrict__ part) {
  const float two_over_b = 2.0f / (float)b;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < b; i += (int64_t)gridDim.x * 256) {
    const float t1 = qt[i * ld], t2 = qt[(b + i) * ld];
    const float tq = fminf(t1, t2);
    const float y = rew[i] + ((1.f - done[i]) * gamma_n) * tq;  // r + (1-d)*gamma^n*minQ'  (:105)
    const float d1 = q[i * ld] - y, d2 = q[(b + i) * ld] - y;
    acc += d1 * d1 + d2 * d2;
    dy[i * ld] = two_over_b * d1;
    dy[(b + i) * ld] = two_over_b * d2;
  }
  const float s = block_sum_256(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

extern "C" int pqlk_td_mse_loss(const float* q, const float* qt, int64_t ld, const float* rew, const float* done,
                                float gamma_n, int64_t b, float* dy, float* loss_out, const int32_t* slot_dev,
                                int32_t ring_len, float* scratch, pqlk_stream_t stream) {
  PQLK_REQUIRE(q && qt && rew && done && dy && scratch, PQLK_E_NULL);
  PQLK_REQUIRE(!slot_dev || ring_len > 0, PQLK_E_SHAPE);
  PQLK_REQUIRE(b > 0, PQLK_E_SHAPE);
  PQLK_REQUIRE(ld >= 32 && ld % 32 == 0, PQLK_E_ALIGN);
  int blocks = (int)((b + 255) / 256);
  if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
  hipLaunchKernelGGL(k_td_mse, dim3(blocks), dim3(256), 0, pqlk_s(stream), q, qt, ld, rew, done, gamma_n, b, dy, scratch);
  PQLK_LAUNCH_CHECK();
  if (!loss_out) return PQLK_OK;   // partials stay in scratch[0, pqlk_loss_parts(b, 1)): folded by pqlk_adamw_polyak_fused
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, pqlk_s(stream), scratch, blocks, 1.0f / (float)b, loss_out,
                     slot_dev, (int)ring_len);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

// ------------------------------------------------------------------------------------------------
// C51.  One wave per batch row, lane k <-> atom k (K <= 64).
__device__ __forceinline__ float wave_softmax(float x, bool valid) {
  const float m = wave_max(valid ? x : -INFINITY);
  const float e = valid ? expf(x - m) : 0.f;
  const float s = wave_sum(e);
  return e / s;
}

// projection of one row's pmf p (lane k) -> projected pmf (lane j).  Deposits are accumulated per bin
// in atom order, all lower-neighbour deposits first, then all upper ones: the order of the reference's
// two sequential index_add_ passes (distl_util.py:18-19), so the result is deterministic.
template <int KMAX>
__device__ __forceinline__ float project_row(float p, float zk, float r, float d, float gamma_n, float v_min, float v_max,
                                             float dz, int K, int lane) {
  float tz = r + ((1.f - d) * gamma_n) * zk;
  tz = fminf(fmaxf(tz, v_min), v_max);
  const float bpos = (tz - v_min) / dz;
  int lo = (int)floorf(bpos), up = (int)ceilf(bpos);
  if (up > 0 && lo == up) lo -= 1;
  if (lo < K - 1 && lo == up) up += 1;
  const float w_lo = p * ((float)up - bpos);
  const float w_up = p * (bpos - (float)lo);
  float out = 0.f;
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) {
    if (kk < K) {
      const int l2 = __shfl(lo, kk, 64);
      const float w2 = __shfl(w_lo, kk, 64);
      if (l2 == lane) out += w2;
    }
  }
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) {
    if (kk < K) {
      const int u2 = __shfl(up, kk, 64);
      const float w2 = __shfl(w_up, kk, 64);
      if (u2 == lane) out += w2;
    }
  }
  return out;
}

__global__ __launch_bounds__(256) void k_c51_project(const float* __restrict__ p, const float* __restrict__ rew,
                                                     const float* __restrict__ done, const float* __restrict__ support,
                                                     float gamma_n, float v_min, float v_max, float dz, int K, int64_t b,
                                                     float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const bool valid = lane < K;
  const float zk = valid ? support[lane] : 0.f;
  for (int64_t i = wave; i < b; i += nw) {
    const float pv = valid ? p[i * K + lane] : 0.f;
    const float o = project_row<64>(pv, zk, rew[i], done[i], gamma_n, v_min, v_max, dz, K, lane);
    if (valid) out[i * K + lane] = o;
  }
}

extern "C" int pqlk_c51_project(const float* p, const float* rew, const float* done, const float* support, float gamma_n,
                                float v_min, float v_max, int32_t k, int64_t b, float* out, pqlk_stream_t stream) {
  PQLK_REQUIRE(p && rew && done && support && out, PQLK_E_NULL);
  PQLK_REQUIRE(b > 0 && k >= 2, PQLK_E_SHAPE);
  PQLK_REQUIRE(k <= 64, PQLK_E_UNSUPPORTED);
  const float dz = (float)(((double)v_max - (double)v_min) / (double)(k - 1));
  int blocks = (int)((b + 3) / 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_c51_project, dim3(blocks), dim3(256), 0, pqlk_s(stream), p, rew, done, support, gamma_n, v_min,
                     v_max, dz, (int)k, b, out);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

__global__ __launch_bounds__(256) void k_c51_bce(const float* __restrict__ logits, const float* __restrict__ logits_t,
                                                 int64_t ld, int K, const float* __restrict__ rew,
                                                 const float* __restrict__ done, const float* __restrict__ support,
                                                 float gamma_n, float v_min, float v_max, float dz, int64_t b,
                                                 float* __restrict__ dy, float* __restrict__ proj_out,
                                                 float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const bool valid = lane < K;
  const float zk = valid ? support[lane] : 0.f;
  const float inv_numel = 1.0f / ((float)b * (float)K);
  float acc = 0.f;
  for (int64_t i = wave; i < b; i += nw) {
    const float r = rew[i], d = done[i];
    // target pmf: min of the two projected target distributions (pql_v_learner.py:83-102)
    const float pt1 = wave_softmax(valid ? logits_t[i * ld + lane] : 0.f, valid);
    const float pt2 = wave_softmax(valid ? logits_t[(b + i) * ld + lane] : 0.f, valid);
    const float pr1 = project_row<64>(pt1, zk, r, d, gamma_n, v_min, v_max, dz, K, lane);
    const float pr2 = project_row<64>(pt2, zk, r, d, gamma_n, v_min, v_max, dz, K, lane);
    const float t = fminf(pr1, pr2);
    if (proj_out && valid) proj_out[i * K + lane] = t;
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      const int64_t row = (int64_t)net * b + i;
      const float pc = wave_softmax(valid ? logits[row * ld + lane] : 0.f, valid);
      float le = 0.f, gp = 0.f;
      if (valid) {
        // F.binary_cross_entropy: (t-1)*max(log(1-p),-100) - t*max(log(p),-100); grad (p-t)/max((1-p)p,1e-12)
        le = (t - 1.f) * fmaxf(logf(1.f - pc), -100.f) - t * fmaxf(logf(pc), -100.f);
        gp = (pc - t) / fmaxf((1.f - pc) * pc, 1e-12f) * inv_numel;
      }
      acc += le;
      const float dot = wave_sum(gp * pc);
      const float dl = valid ? (gp - dot) * pc : 0.f;  // softmax backward
      if (lane < ld) dy[row * ld + lane] = dl;
      for (int c = 64 + lane; c < ld; c += 64) dy[row * ld + c] = 0.f;
    }
  }
  const float s = block_sum_256(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

extern "C" int pqlk_c51_bce_loss(const float* logits, const float* logits_t, int64_t ld, int32_t k, const float* rew,
                                 const float* done, const float* support, float gamma_n, float v_min, float v_max,
                                 int64_t b, float* dy, float* loss_out, const int32_t* slot_dev, int32_t ring_len,
                                 float* proj_out, float* scratch, pqlk_stream_t stream) {
  PQLK_REQUIRE(logits && logits_t && rew && done && support && dy && scratch, PQLK_E_NULL);
  PQLK_REQUIRE(!slot_dev || ring_len > 0, PQLK_E_SHAPE);
  PQLK_REQUIRE(b > 0 && k >= 2, PQLK_E_SHAPE);
  PQLK_REQUIRE(k <= 64, PQLK_E_UNSUPPORTED);
  PQLK_REQUIRE(ld % 32 == 0 && ld >= k, PQLK_E_ALIGN);
  const float dz = (float)(((double)v_max - (double)v_min) / (double)(k - 1));
  int blocks = (int)((b + 3) / 4);
  if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
  hipLaunchKernelGGL(k_c51_bce, dim3(blocks), dim3(256), 0, pqlk_s(stream), logits, logits_t, ld, (int)k, rew, done,
                     support, gamma_n, v_min, v_max, dz, b, dy, proj_out, scratch);
  PQLK_LAUNCH_CHECK();
  if (!loss_out) return PQLK_OK;   // partials stay in scratch[0, pqlk_loss_parts(b, k))
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, pqlk_s(stream), scratch, blocks,
                     1.0f / ((float)b * (float)k), loss_out, slot_dev, (int)ring_len);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

// ------------------------------------------------------------------------------------------------
// DPG: L = -mean(min(Q1, Q2)); gradient w.r.t. the critic outputs (ties split evenly like torch.min).
__global__ __launch_bounds__(256) void k_dpg_scalar(const float* __restrict__ q, int64_t ld, int64_t b,
                                                    float* __restrict__ dy, float* __restrict__ part,
                                                    uint8_t* __restrict__ owner) {
  const float g = -1.0f / (float)b;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < b; i += (int64_t)gridDim.x * 256) {
    const float a = q[i * ld], c = q[(b + i) * ld];
    acc += fminf(a, c);
    const float g1 = a < c ? g : (a == c ? 0.5f * g : 0.f);
    const float g2 = c < a ? g : (a == c ? 0.5f * g : 0.f);
    dy[i * ld] = g1;
    dy[(b + i) * ld] = g2;
    if (owner) owner[i] = (uint8_t)((a <= c ? 1 : 0) | (c <= a ? 2 : 0));   // which net(s) the gradient reaches (minnet.h)
  }
  const float s = block_sum_256(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_dpg_dist(const float* __restrict__ logits, int64_t ld, int K,
                                                  const float* __restrict__ support, int64_t b, float* __restrict__ dy,
                                                  float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const bool valid = lane < K;
  const float zk = valid ? support[lane] : 0.f;
  const float g = -1.0f / (float)b;
  float acc = 0.f;
  for (int64_t i = wave; i < b; i += nw) {
    const float p1 = wave_softmax(valid ? logits[i * ld + lane] : 0.f, valid);
    const float p2 = wave_softmax(valid ? logits[(b + i) * ld + lane] : 0.f, valid);
    const float q1 = wave_sum(p1 * zk), q2 = wave_sum(p2 * zk);  // E[z] (mlp.py:258-259)
    if (lane == 0) acc += fminf(q1, q2);
    const float g1 = q1 < q2 ? g : (q1 == q2 ? 0.5f * g : 0.f);
    const float g2 = q2 < q1 ? g : (q1 == q2 ? 0.5f * g : 0.f);
    // dQ/dlogit_k = p_k (z_k - Q)
    const float d1 = valid ? g1 * p1 * (zk - q1) : 0.f;
    const float d2 = valid ? g2 * p2 * (zk - q2) : 0.f;
    if (lane < ld) {
      dy[i * ld + lane] = d1;
      dy[(b + i) * ld + lane] = d2;
    }
    for (int c = 64 + lane; c < ld; c += 64) {
      dy[i * ld + c] = 0.f;
      dy[(b + i) * ld + c] = 0.f;
    }
  }
  const float s = block_sum_256(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

static int dpg_loss_impl(const float* q, int64_t ld, int32_t k, const float* support, int64_t b, float* dy,
                         float* loss_out, const int32_t* slot_dev, int32_t ring_len, float* scratch, uint8_t* owner,
                         pqlk_stream_t stream) {
  PQLK_REQUIRE(q && dy && scratch, PQLK_E_NULL);
  PQLK_REQUIRE(!slot_dev || ring_len > 0, PQLK_E_SHAPE);
  PQLK_REQUIRE(b > 0 && k >= 1, PQLK_E_SHAPE);
  PQLK_REQUIRE(k <= 64, PQLK_E_UNSUPPORTED);
  PQLK_REQUIRE(ld % 32 == 0 && ld >= k, PQLK_E_ALIGN);
  int blocks;
  if (k == 1) {
    blocks = (int)((b + 255) / 256);
    if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
    hipLaunchKernelGGL(k_dpg_scalar, dim3(blocks), dim3(256), 0, pqlk_s(stream), q, ld, b, dy, scratch, owner);
  } else {
    PQLK_REQUIRE(support, PQLK_E_NULL);
    blocks = (int)((b + 3) / 4);
    if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
    hipLaunchKernelGGL(k_dpg_dist, dim3(blocks), dim3(256), 0, pqlk_s(stream), q, ld, (int)k, support, b, dy, scratch);
  }
  PQLK_LAUNCH_CHECK();
  if (!loss_out) return PQLK_OK;   // partials stay in scratch[0, pqlk_loss_parts(b, k))
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, pqlk_s(stream), scratch, blocks, -1.0f / (float)b, loss_out,
                     slot_dev, (int)ring_len);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

extern "C" int pqlk_dpg_loss(const float* q, int64_t ld, int32_t k, const float* support, int64_t b, float* dy,
                             float* loss_out, const int32_t* slot_dev, int32_t ring_len, float* scratch,
                             pqlk_stream_t stream) {
  return dpg_loss_impl(q, ld, k, support, b, dy, loss_out, slot_dev, ring_len, scratch, nullptr, stream);
}

// Same; for scalar heads (k == 1) additionally owner[m] = bit 0: the gradient of min(Q1, Q2) reaches net 0, bit 1: net 1
// (both on an exact tie) -- the input of pqlk_dpg_critic_backward's partition.  owner is not written when k > 1.
extern "C" int pqlk_dpg_loss_owner(const float* q, int64_t ld, int32_t k, const float* support, int64_t b, float* dy,
                                   float* loss_out, const int32_t* slot_dev, int32_t ring_len, float* scratch,
                                   uint8_t* owner, pqlk_stream_t stream) {
  return dpg_loss_impl(q, ld, k, support, b, dy, loss_out, slot_dev, ring_len, scratch, owner, stream);
}

// number of per-block loss partials the loss entry points leave in `scratch` (k = 1: scalar heads; k > 1: one wave per row)
extern "C" int32_t pqlk_loss_parts(int64_t b, int32_t k) {
  int64_t blocks = k <= 1 ? (b + 255) / 256 : (b + 3) / 4;
  if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
  return (int32_t)blocks;
}
