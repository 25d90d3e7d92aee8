// n-step transition assembler.  Reference: pql/replay/nstep_replay.py:29-92 (five torch.cat FIFO
// shifts + compute_nstep_return per env-step).  Here: a circular per-env window (slot = step % n),
// one wave per env, one launch per env-step; the step being pushed is consumed from the inputs, older
// steps from the window, so there is no read-after-write through memory inside a launch.
#include "pqlk_common.h"

#define PQLK_MAX_NSTEP 16

struct GammaPow {
  float g[PQLK_MAX_NSTEP];
};

// slabs are (N, T, .) contiguous; this launch handles time index t (global step s = count + t).
__global__ __launch_bounds__(256) void k_nstep_step(float* __restrict__ window, RecLayout L, int64_t N, int n, int64_t s,
                                                    int64_t T, int64_t t, const float* __restrict__ obs,
                                                    const float* __restrict__ act, const float* __restrict__ rew,
                                                    const float* __restrict__ nobs, const float* __restrict__ done,
                                                    GammaPow gp, int emit, int64_t out_row0, float* __restrict__ o_obs,
                                                    float* __restrict__ o_act, float* __restrict__ o_rew,
                                                    float* __restrict__ o_nobs, float* __restrict__ o_done) {
  const int lane = threadIdx.x & 63;
  const int64_t env = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (env >= N) return;
  const int slot_new = (int)(s % n);
  const float* in_obs = obs + (env * T + t) * L.O;
  const float* in_act = act + (env * T + t) * L.A;
  const float* in_nobs = nobs + (env * T + t) * L.O;
  const float r_new = rew[env * T + t];
  const float d_new = done[env * T + t];
  float* wenv = window + env * (int64_t)n * L.ld;

  if (emit) {
    // window order: j = 0 oldest (step s-n+1) .. n-1 newest (this step)
    float rj[PQLK_MAX_NSTEP], dj[PQLK_MAX_NSTEP];
#pragma unroll
    for (int j = 0; j < PQLK_MAX_NSTEP; ++j) {
      if (j < n - 1) {
        const int slot = (int)((s - (n - 1) + j) % n);
        const float* rec = wenv + (int64_t)slot * L.ld;
        rj[j] = rec[L.off_rd];
        dj[j] = rec[L.off_rd + 1];
      } else if (j == n - 1) {
        rj[j] = r_new;
        dj[j] = d_new;
      } else {
        rj[j] = 0.f;
        dj[j] = 0.f;
      }
    }
    // any / first-max (nstep_replay.py:77-79: where(done), argmax(dim=1) = first maximal entry)
    bool any = false;
    float mx = dj[0];
    int first = 0;
#pragma unroll
    for (int j = 0; j < PQLK_MAX_NSTEP; ++j) {
      if (j < n) {
        any = any || (dj[j] != 0.f);
        if (dj[j] > mx) {
          mx = dj[j];
          first = j;
        }
      }
    }
    // R: four interleaved partial sums, then ((A0+A1)+A2)+A3 -- torch's row-sum order for n <= 5
    float lanes4[4] = {0.f, 0.f, 0.f, 0.f};
    bool used4[4] = {false, false, false, false};
#pragma unroll
    for (int j = 0; j < PQLK_MAX_NSTEP; ++j) {
      if (j < n) {
        const float keep = (!any || j <= first) ? 1.f : 0.f;
        const float term = __fmul_rn(__fmul_rn(rj[j], gp.g[j]), keep);  // no fma contraction: (r*g)*mask
        const int a = j & 3;
        lanes4[a] = used4[a] ? __fadd_rn(lanes4[a], term) : term;
        used4[a] = true;
      }
    }
    float R = lanes4[0];
#pragma unroll
    for (int a = 1; a < 4; ++a)
      if (used4[a]) R = __fadd_rn(R, lanes4[a]);

    const int64_t row = out_row0 + env;
    // obs / action of the oldest step
    if (n == 1) {
      for (int c = lane; c < L.O; c += 64) o_obs[row * L.O + c] = in_obs[c];
      for (int c = lane; c < L.A; c += 64) o_act[row * L.A + c] = in_act[c];
    } else {
      const float* rec0 = wenv + (int64_t)((s - (n - 1)) % n) * L.ld;
      for (int c = lane; c < L.O; c += 64) o_obs[row * L.O + c] = rec0[c];
      for (int c = lane; c < L.A; c += 64) o_act[row * L.A + c] = rec0[L.off_act + c];
    }
    // next_obs at the first done, else of the newest step
    const int sel = any ? first : n - 1;
    if (sel == n - 1) {
      for (int c = lane; c < L.O; c += 64) o_nobs[row * L.O + c] = in_nobs[c];
    } else {
      const float* recs = wenv + (int64_t)((s - (n - 1) + sel) % n) * L.ld;
      for (int c = lane; c < L.O; c += 64) o_nobs[row * L.O + c] = recs[L.off_nobs + c];
    }
    if (lane == 0) {
      o_rew[row] = R;
      o_done[row] = any ? 1.f : d_new;  // done[:, -1] with any -> True (:81-82)
    }
  }

  // push this step into its slot (overwrites step s-n, which is no longer needed: its last use was
  // as the oldest entry of the window emitted at step s-1)
  float* recn = wenv + (int64_t)slot_new * L.ld;
  for (int c = lane; c < L.ld; c += 64) {
    float v = 0.f;
    if (c < L.O) v = in_obs[c];
    else if (c >= L.off_nobs && c < L.off_nobs + L.O) v = in_nobs[c - L.off_nobs];
    else if (c >= L.off_act && c < L.off_act + L.A) v = in_act[c - L.off_act];
    else if (c == L.off_rd) v = r_new;
    else if (c == L.off_rd + 1) v = d_new;
    recn[c] = v;
  }
}

extern "C" int pqlk_nstep_push_emit(float* window, int64_t num_envs, int32_t nstep, int32_t obs_dim, int32_t act_dim,
                                    int64_t count, int64_t t_steps, const float* obs, const float* act, const float* rew,
                                    const float* next_obs, const float* done, const float* gamma_pow, float* o_obs,
                                    float* o_act, float* o_rew, float* o_next_obs, float* o_done, int64_t* rows_out,
                                    pqlk_stream_t stream) {
  PQLK_REQUIRE(window && obs && act && rew && next_obs && done && gamma_pow, PQLK_E_NULL);
  PQLK_REQUIRE(num_envs > 0 && obs_dim > 0 && act_dim > 0 && t_steps >= 0 && count >= 0, PQLK_E_SHAPE);
  PQLK_REQUIRE(nstep >= 1 && nstep <= PQLK_MAX_NSTEP, PQLK_E_UNSUPPORTED);
  RecLayout L = rec_layout(obs_dim, act_dim);
  GammaPow gp;
  for (int j = 0; j < PQLK_MAX_NSTEP; ++j) gp.g[j] = j < nstep ? gamma_pow[j] : 0.f;
  // first emitting step: s + 1 >= nstep
  int64_t first_emit_t = nstep - 1 - count;
  if (first_emit_t < 0) first_emit_t = 0;
  int64_t emitted = t_steps > first_emit_t ? (t_steps - first_emit_t) * num_envs : 0;
  if (emitted > 0) PQLK_REQUIRE(o_obs && o_act && o_rew && o_next_obs && o_done, PQLK_E_NULL);
  const unsigned blocks = (unsigned)((num_envs + 3) / 4);
  for (int64_t t = 0; t < t_steps; ++t) {
    const int64_t s = count + t;
    const int emit = (t >= first_emit_t) ? 1 : 0;
    const int64_t row0 = emit ? (t - first_emit_t) * num_envs : 0;
    hipLaunchKernelGGL(k_nstep_step, dim3(blocks), dim3(256), 0, pqlk_s(stream), window, L, num_envs, (int)nstep, s,
                       t_steps, t, obs, act, rew, next_obs, done, gp, emit, row0, o_obs, o_act, o_rew, o_next_obs,
                       o_done);
    PQLK_LAUNCH_CHECK();
  }
  if (rows_out) *rows_out = emitted;
  return PQLK_OK;
}
