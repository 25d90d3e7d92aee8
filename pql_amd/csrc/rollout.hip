// Per-env-step bookkeeping of the rollout in ONE launch.
//
// Reference: PQLActor.explore_env / update_tracker (pql/algo/pql_actor.py:104-114, :129-135) and handle_timeout
// (pql/utils/common.py:195-202).  Per env step the reference (and round 1 of this build) issues ~25 small ATen
// launches: five trajectory-slab writes, the episode return / length accumulators, `torch.where(done)` + a host
// round trip for the two moving windows, the masked resets and the time-limit mask.  Here:
//   blocks >= 1 : copy obs / action / next_obs into column t of the (N, T, .) slabs (16-B lanes), reward and done' into theirs,
//                 done' = done * !truncated                                                 (handle_timeout)
//   block 0     : cur_return += reward, cur_length += 1; the values of the finished envs are appended to the two windows
//                 in env order, exactly as `deque.extend` would (only the LAST `win_len` of them when more finish in one
//                 step), the finished envs' accumulators reset, the window pointer advanced.  An ordered compaction over N
//                 flags = a block-wide exclusive scan in chunks of 1024; deterministic, no atomics, no host sync.
#include "pqlk_common.h"

struct RolloutP {
  int64_t n; int O, A, T, t;
  const float* obs; const float* act; const float* nobs; const float* rew; const uint8_t* done; const uint8_t* trunc;
  float* s_obs; float* s_act; float* s_rew; float* s_nobs; float* s_done;   // (N, T, O), (N, T, A), (N, T, 1), (N, T, O), (N, T, 1)
  float* cur_ret; float* cur_len;                                           // (N)
  float* win_ret; float* win_len; int64_t* ptr_ret; int64_t* ptr_len; int win;   // windows of `win` floats + their write pointers
};

__device__ __forceinline__ void copy_rows(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int w, int T, int t,
                                          int64_t tid, int64_t nthreads) {
  if ((w & 3) == 0) {   // 16-B lanes
    const int w4 = w >> 2;
    for (int64_t i = tid; i < n * w4; i += nthreads) {
      const int64_t e = i / w4; const int c = (int)(i - e * w4);
      reinterpret_cast<float4*>(dst + (e * T + t) * w)[c] = reinterpret_cast<const float4*>(src + e * w)[c];
    }
  } else {
    for (int64_t i = tid; i < n * w; i += nthreads) {
      const int64_t e = i / w; const int c = (int)(i - e * w);
      dst[(e * T + t) * w + c] = src[i];
    }
  }
}

__global__ __launch_bounds__(1024) void k_rollout_step(RolloutP p) {
  if (blockIdx.x > 0) {
    const int64_t nthreads = (int64_t)(gridDim.x - 1) * 1024, tid = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x;
    copy_rows(p.obs, p.s_obs, p.n, p.O, p.T, p.t, tid, nthreads);
    copy_rows(p.nobs, p.s_nobs, p.n, p.O, p.T, p.t, tid, nthreads);
    copy_rows(p.act, p.s_act, p.n, p.A, p.T, p.t, tid, nthreads);
    for (int64_t e = tid; e < p.n; e += nthreads) {
      p.s_rew[e * p.T + p.t] = p.rew[e];
      const bool d = p.done[e] != 0 && !(p.trunc && p.trunc[e] != 0);
      p.s_done[e * p.T + p.t] = d ? 1.f : 0.f;
    }
    return;
  }
  // ---- block 0: accumulators + ordered append of the finished episodes
  __shared__ int wave_tot[16];
  __shared__ int s_total, s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // pass 1: how many episodes finished (the appended run keeps only its last `win` entries)
  int cnt = 0;
  for (int64_t e = threadIdx.x; e < p.n; e += 1024) cnt += p.done[e] != 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) wave_tot[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < 16; ++w) s += wave_tot[w];
    s_total = s; s_base = 0;
  }
  __syncthreads();
  const int total = s_total;
  const int64_t ptr_r = p.ptr_ret[0], ptr_l = p.ptr_len[0];
  // pass 2: chunks of 1024 envs in order; rank = finished envs before this one
  for (int64_t e0 = 0; e0 < p.n; e0 += 1024) {
    const int64_t e = e0 + threadIdx.x;
    const bool in = e < p.n;
    const bool fin = in && p.done[e] != 0;
    float r = 0.f, l = 0.f;
    if (in) { r = p.cur_ret[e] + p.rew[e]; l = p.cur_len[e] + 1.f; }
    const unsigned long long ball = __ballot(fin);
    const int before_in_wave = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(ball);
    __syncthreads();
    int before = s_base + before_in_wave;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    if (fin && before >= total - p.win) {
      p.win_ret[(ptr_r + before) % p.win] = r;
      p.win_len[(ptr_l + before) % p.win] = l;
    }
    if (in) { p.cur_ret[e] = fin ? 0.f : r; p.cur_len[e] = fin ? 0.f : l; }
    __syncthreads();
    if (threadIdx.x == 0) {
      int s = 0;
      for (int w = 0; w < 16; ++w) s += wave_tot[w];
      s_base += s;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { p.ptr_ret[0] = (ptr_r + total) % p.win; p.ptr_len[0] = (ptr_l + total) % p.win; }
}

extern "C" int pqlk_rollout_step(int64_t n, int32_t obs_dim, int32_t act_dim, int32_t horizon, int32_t t, const float* obs,
                                 const float* action, const float* next_obs, const float* reward, const uint8_t* done,
                                 const uint8_t* truncated, float* slab_obs, float* slab_act, float* slab_rew, float* slab_nobs,
                                 float* slab_done, float* cur_return, float* cur_length, float* win_return, float* win_length,
                                 int64_t* win_return_ptr, int64_t* win_length_ptr, int32_t win_len, pqlk_stream_t stream) {
  PQLK_REQUIRE(obs && action && next_obs && reward && done && slab_obs && slab_act && slab_rew && slab_nobs && slab_done &&
               cur_return && cur_length && win_return && win_length && win_return_ptr && win_length_ptr, PQLK_E_NULL);
  PQLK_REQUIRE(n > 0 && obs_dim > 0 && act_dim > 0 && horizon > 0 && t >= 0 && t < horizon && win_len > 0, PQLK_E_SHAPE);
  if ((obs_dim & 3) == 0)
    PQLK_REQUIRE(pqlk_aligned16(obs) && pqlk_aligned16(next_obs) && pqlk_aligned16(slab_obs) && pqlk_aligned16(slab_nobs), PQLK_E_ALIGN);
  if ((act_dim & 3) == 0) PQLK_REQUIRE(pqlk_aligned16(action) && pqlk_aligned16(slab_act), PQLK_E_ALIGN);
  RolloutP p;
  p.n = n; p.O = obs_dim; p.A = act_dim; p.T = horizon; p.t = t;
  p.obs = obs; p.act = action; p.nobs = next_obs; p.rew = reward; p.done = done; p.trunc = truncated;
  p.s_obs = slab_obs; p.s_act = slab_act; p.s_rew = slab_rew; p.s_nobs = slab_nobs; p.s_done = slab_done;
  p.cur_ret = cur_return; p.cur_len = cur_length; p.win_ret = win_return; p.win_len = win_length; p.ptr_ret = win_return_ptr; p.ptr_len = win_length_ptr; p.win = win_len;
  int64_t copy_blocks = (n * (2 * (int64_t)obs_dim + act_dim) / 4 + 1023) / 1024;
  if (copy_blocks < 1) copy_blocks = 1;
  if (copy_blocks > 512) copy_blocks = 512;
  hipLaunchKernelGGL(k_rollout_step, dim3((unsigned)(1 + copy_blocks)), dim3(1024), 0, pqlk_s(stream), p);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}


// ------------------------------------------------------------------------------------------------
// The rest of the per-env-step ATen micro-kernels of the rollout, one launch each (were 12 + 4 + 4 launches):
//   RunningMeanStd.update_from_moments (pql/utils/torch_util.py:91-103), op for op in fp32 with the python scalars rounded to
//   fp32 where torch rounds them; RunningMeanStd.normalize (:83-85); add_normal_noise / add_mixed_normal_noise (noise.py:19-41).
__global__ __launch_bounds__(256) void k_rms_merge(const float* __restrict__ mean, const float* __restrict__ var,
                                                   const float* __restrict__ bm, const float* __restrict__ bv, float count, float bcount,
                                                   float tot, int cols, float* __restrict__ mean_out, float* __restrict__ var_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cols) return;
  const float delta = bm[i] - mean[i];
  const float m2 = (var[i] * count + bv[i] * bcount) + (((delta * delta) * count) * bcount) / tot;
  mean_out[i] = mean[i] + (delta * bcount) / tot;
  var_out[i] = m2 / tot;
}

extern "C" int pqlk_rms_merge(const float* mean, const float* var, const float* batch_mean, const float* batch_var, float count,
                              float batch_count, float total, int32_t cols, float* mean_out, float* var_out, pqlk_stream_t stream) {
  PQLK_REQUIRE(mean && var && batch_mean && batch_var && mean_out && var_out, PQLK_E_NULL);
  PQLK_REQUIRE(cols > 0 && total > 0.f, PQLK_E_SHAPE);
  hipLaunchKernelGGL(k_rms_merge, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, pqlk_s(stream), mean, var, batch_mean, batch_var,
                     count, batch_count, total, cols, mean_out, var_out);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

__global__ __launch_bounds__(256) void k_rms_normalize(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ var, float eps, long long total, int cols,
                                                       float* __restrict__ out, long long ld_out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / cols;
    const int c = (int)(i - row * cols);
    out[row * ld_out + c] = (x[i] - mean[c]) / sqrtf(var[c] + eps);
  }
}

extern "C" int pqlk_rms_normalize(const float* x, int64_t rows, int32_t cols, const float* mean, const float* var, float eps, float* out,
                                  int64_t ld_out, pqlk_stream_t stream) {
  PQLK_REQUIRE(x && mean && var && out, PQLK_E_NULL);
  PQLK_REQUIRE(rows > 0 && cols > 0 && ld_out >= cols, PQLK_E_SHAPE);
  const long long total = rows * (long long)cols;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_rms_normalize, dim3((unsigned)blocks), dim3(256), 0, pqlk_s(stream), x, mean, var, eps, total, cols, out,
                     (long long)ld_out);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

// out = clamp(act + draw * sigma, lo, hi); sigma = std_rows[row] (mixed noise: one sigma per env) or std_scalar
__global__ __launch_bounds__(256) void k_action_noise(const float* __restrict__ act, const float* __restrict__ draw,
                                                      const float* __restrict__ std_rows, float std_scalar, long long total, int cols,
                                                      float lo, float hi, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float sigma = std_rows ? std_rows[i / cols] : std_scalar;
    const float noise = draw[i] * sigma;
    out[i] = fminf(fmaxf(act[i] + noise, lo), hi);
  }
}

extern "C" int pqlk_action_noise(const float* act, const float* draw, const float* std_rows, float std_scalar, int64_t rows, int32_t cols,
                                 float lo, float hi, float* out, pqlk_stream_t stream) {
  PQLK_REQUIRE(act && draw && out, PQLK_E_NULL);
  PQLK_REQUIRE(rows > 0 && cols > 0 && lo <= hi, PQLK_E_SHAPE);
  const long long total = rows * (long long)cols;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_action_noise, dim3((unsigned)blocks), dim3(256), 0, pqlk_s(stream), act, draw, std_rows, std_scalar, total, cols, lo,
                     hi, out);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}
