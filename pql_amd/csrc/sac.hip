// SAC on the PQL kernels: the squashed-Gaussian policy head and the temperature terms.
//
// Reference arithmetic: TanhDiagGaussianMLPPolicy.get_actions_logprob (pql/models/mlp.py:144-174), SquashedNormal /
// TanhTransform (pql/utils/torch_util.py:15-65) over torch.distributions.Normal (rsample = loc + eps * scale;
// log_prob = -((v - loc)^2) / (2 scale^2) - log(scale) - log(sqrt(2 pi))), AgentSAC.update_critic / update_actor
// (pql/algo/sac.py:32-42,138-156).  All of it is per-row elementwise work over (B, A <= 64): HBM-bound, a few KB/row.
// G = the power of two >= A lanes share a row, so the row sums are butterfly shuffles inside a wave.
#include "pqlk_common.h"

__device__ __forceinline__ float group_sum(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus(beta=1, threshold=20)

#define SG_LOG_STD_MIN (-5.f)
#define SG_LOG_STD_MAX (5.f)

__global__ __launch_bounds__(256) void k_sg_head_fwd(const float* __restrict__ y, int64_t ld_y, const float* __restrict__ eps,
                                                     int64_t b, int A, int G, float* __restrict__ act, int64_t ld_act,
                                                     float* __restrict__ logp) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / G;
  const int j = (int)(t % G);
  const bool ok = row < b && j < A;
  float lp = 0.f;
  if (ok) {
    const float mu = y[row * ld_y + j];
    if (!eps) {   // get_actions(sample=False): dist.mean = tanh(loc)
      act[row * ld_act + j] = tanhf(mu);
    } else {
      const float ls = fminf(fmaxf(y[row * ld_y + A + j], SG_LOG_STD_MIN), SG_LOG_STD_MAX);
      const float sd = expf(ls);
      const float u = mu + eps[row * A + j] * sd;
      const float a = tanhf(u);
      act[row * ld_act + j] = a;
      const float d = u - mu;
      const float base = -(d * d) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274f;
      const float jac = 2.f * (0.69314718055994531f - u - softplus_t(-2.f * u));   // TanhTransform.log_abs_det_jacobian(u, a)
      lp = (0.f - jac) + base;
    }
  }
  if (eps && logp) {
    lp = group_sum(lp, G);
    if (row < b && j == 0) logp[row] = lp;
  }
}

// d loss / d [mu | log_std] from d loss / d a and d loss / d logp (= glp, one scalar for the whole batch):
//   u = mu + e s, a = tanh u:  d logp_j / d mu = 2 a,  d logp_j / d s = 2 a e - 1/s  (the Normal terms cancel analytically),
//   d s / d log_std = s inside the clamp (torch's clamp backward passes the bounds themselves)
__global__ __launch_bounds__(256) void k_sg_head_bwd(const float* __restrict__ y, int64_t ld_y, const float* __restrict__ eps,
                                                     const float* __restrict__ act, int64_t ld_act, const float* __restrict__ da,
                                                     int64_t ld_da, const float* __restrict__ log_alpha, float glp_scale, int64_t b,
                                                     int A, float* __restrict__ dy) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / ld_y;
  const int c = (int)(t % ld_y);
  if (row >= b) return;
  float out = 0.f;   // pad columns of dy are rewritten as zero
  if (c < 2 * A) {
    const int j = c < A ? c : c - A;
    const float glp = glp_scale * (log_alpha ? expf(log_alpha[0]) : 1.f);
    const float a = act[row * ld_act + j];
    const float gu = da[row * ld_da + j] * (1.f - a * a);
    if (c < A) {
      out = gu + glp * (2.f * a);
    } else {
      const float raw = y[row * ld_y + A + j];
      if (raw >= SG_LOG_STD_MIN && raw <= SG_LOG_STD_MAX) {
        const float sd = expf(raw), e = eps[row * A + j];
        out = gu * e * sd + glp * (2.f * a * e * sd - 1.f);
      }
    }
  }
  dy[row * ld_y + c] = out;
}

// target-critic outputs, column 0 of each net: q -= exp(log_alpha) * logp   (min(q1 - c, q2 - c) = min(q1, q2) - c)
__global__ __launch_bounds__(256) void k_sac_entropy_shift(float* __restrict__ qt, int64_t ld, int64_t net_stride, int n_nets,
                                                           const float* __restrict__ logp, const float* __restrict__ log_alpha,
                                                           int64_t b) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= b) return;
  const float c = expf(log_alpha[0]) * logp[r];
  for (int n = 0; n < n_nets; ++n) qt[n * net_stride + r * ld] -= c;
}

// one block: m = mean(logp) in fixed order;  grad_log_alpha = alpha * (-m - target_entropy)  [= d/d log_alpha of
// mean(exp(log_alpha) * (-logp - target_entropy))],  alpha_loss = the same value,  actor_loss_ring[slot] += alpha * m
__global__ __launch_bounds__(1024) void k_sac_alpha_terms(const float* __restrict__ logp, int64_t b, const float* __restrict__ log_alpha,
                                                          float target_entropy, float* __restrict__ grad_out,
                                                          float* __restrict__ alpha_loss_out, float* __restrict__ actor_loss_ring,
                                                          const int32_t* __restrict__ slot_dev, int ring_len) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < b; i += 1024) s += logp[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += sh[w];
    const float m = tot / (float)b;
    const float alpha = expf(log_alpha[0]);
    const float g = alpha * (-m - target_entropy);
    if (grad_out) grad_out[0] = g;
    if (alpha_loss_out) alpha_loss_out[0] = g;
    if (actor_loss_ring) actor_loss_ring[slot_dev ? (slot_dev[0] % ring_len) : 0] += alpha * m;
  }
}

static int group_of(int A) {
  int G = 1;
  while (G < A) G <<= 1;
  return G;
}

extern "C" int pqlk_sg_head_forward(const float* y, int64_t ld_y, const float* eps, int64_t b, int32_t act_dim, float* act,
                                    int64_t ld_act, float* logp, pqlk_stream_t stream) {
  PQLK_REQUIRE(y && act, PQLK_E_NULL);
  PQLK_REQUIRE(b > 0 && act_dim > 0 && act_dim <= 64 && ld_y >= 2 * act_dim && ld_act >= act_dim, PQLK_E_SHAPE);
  const int G = group_of(act_dim);
  const int64_t blocks = (b * G + 255) / 256;
  hipLaunchKernelGGL(k_sg_head_fwd, dim3((unsigned)blocks), dim3(256), 0, pqlk_s(stream), y, ld_y, eps, b, (int)act_dim, G, act,
                     ld_act, logp);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

extern "C" int pqlk_sg_head_backward(const float* y, int64_t ld_y, const float* eps, const float* act, int64_t ld_act, const float* da,
                                     int64_t ld_da, const float* log_alpha, float glp_scale, int64_t b, int32_t act_dim, float* dy,
                                     pqlk_stream_t stream) {
  PQLK_REQUIRE(y && eps && act && da && dy, PQLK_E_NULL);
  PQLK_REQUIRE(b > 0 && act_dim > 0 && act_dim <= 64 && ld_y >= 2 * act_dim && ld_act >= act_dim && ld_da >= act_dim, PQLK_E_SHAPE);
  const int64_t blocks = (b * ld_y + 255) / 256;
  hipLaunchKernelGGL(k_sg_head_bwd, dim3((unsigned)blocks), dim3(256), 0, pqlk_s(stream), y, ld_y, eps, act, ld_act, da, ld_da, log_alpha,
                     glp_scale, b, (int)act_dim, dy);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

extern "C" int pqlk_sac_entropy_shift(float* qt, int64_t ld, int64_t net_stride, int32_t n_nets, const float* logp,
                                      const float* log_alpha, int64_t b, pqlk_stream_t stream) {
  PQLK_REQUIRE(qt && logp && log_alpha, PQLK_E_NULL);
  PQLK_REQUIRE(b > 0 && n_nets > 0 && ld > 0, PQLK_E_SHAPE);
  hipLaunchKernelGGL(k_sac_entropy_shift, dim3((unsigned)((b + 255) / 256)), dim3(256), 0, pqlk_s(stream), qt, ld, net_stride,
                     (int)n_nets, logp, log_alpha, b);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

extern "C" int pqlk_sac_alpha_terms(const float* logp, int64_t b, const float* log_alpha, float target_entropy, float* grad_out,
                                    float* alpha_loss_out, float* actor_loss_ring, const int32_t* slot_dev, int32_t ring_len,
                                    pqlk_stream_t stream) {
  PQLK_REQUIRE(logp && log_alpha, PQLK_E_NULL);
  PQLK_REQUIRE(b > 0 && (!actor_loss_ring || ring_len > 0), PQLK_E_SHAPE);
  hipLaunchKernelGGL(k_sac_alpha_terms, dim3(1), dim3(1024), 0, pqlk_s(stream), logp, b, log_alpha, target_entropy, grad_out,
                     alpha_loss_out, actor_loss_ring, slot_dev, (int)ring_len);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}
