// Synthetic vectorised environment step in ONE launch.  This is the Isaac-Gym stand-in BASELINE.json asks for (not a
// reference component): a counter-based generator, every output a pure function of (seed, global env id, step, column),
// so data-parallel shards reproduce slices of the global env.  Same arithmetic as the torch-op definition in
// pql_amd/envs/synthetic.py (which stays the CPU / test form); the torch version costs ~150 tiny launches per step.
#include "pqlk_common.h"

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x = (x ^ (x >> 16)) * 0x7FEB352Du;
  x = (x ^ (x >> 15)) * 0x846CA68Bu;
  return x ^ (x >> 16);
}

__device__ __forceinline__ float uni(uint32_t env, uint32_t seed, uint32_t t, uint32_t stream, uint32_t col) {
  const uint32_t key = hash32(env * 0x9E3779B1u + seed * 0x85EBCA77u + t * 0xC2B2AE3Du + stream * 0x27D4EB2Fu);
  const uint32_t h = hash32(key * 0x165667B1u + col * 0x9E3779B1u + 0x5BD1E995u);
  return ((float)h + 0.5f) * (1.0f / 4294967296.0f);
}

__device__ __forceinline__ float gauss(uint32_t env, uint32_t seed, uint32_t t, uint32_t stream, uint32_t col) {
  const float u1 = uni(env, seed, t, 2 * stream, col), u2 = uni(env, seed, t, 2 * stream + 1, col);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__global__ __launch_bounds__(256) void k_synth_env_step(int64_t n, int obs_dim, int act_dim, uint32_t seed, uint32_t env0, uint32_t t,
                                                        float p_done, const float* __restrict__ action,
                                                        float* __restrict__ next_obs, float* __restrict__ reward,
                                                        uint8_t* __restrict__ done) {
  const int64_t total = n * obs_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i / obs_dim;
    const int c = (int)(i - e * obs_dim);
    next_obs[i] = gauss(env0 + (uint32_t)e, seed, t, 1, (uint32_t)c);
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    float a2 = 0.f;
    if (action)
      for (int j = 0; j < act_dim; ++j) { const float a = action[e * act_dim + j]; a2 += a * a; }
    const uint32_t ge = env0 + (uint32_t)e;
    reward[e] = gauss(ge, seed, t, 2, 0) - 0.1f * (a2 / (float)act_dim);
    done[e] = uni(ge, seed, t, 7, 0) < p_done ? 1 : 0;
  }
}

extern "C" int pqlk_synth_env_step(int64_t n, int32_t obs_dim, int32_t act_dim, uint32_t seed, uint32_t env_offset, uint32_t t,
                                   float p_done, const float* action, float* next_obs, float* reward, uint8_t* done,
                                   pqlk_stream_t stream) {
  PQLK_REQUIRE(next_obs && reward && done, PQLK_E_NULL);
  PQLK_REQUIRE(n > 0 && obs_dim > 0 && act_dim > 0, PQLK_E_SHAPE);
  int64_t blocks = (n * obs_dim + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_synth_env_step, dim3((unsigned)blocks), dim3(256), 0, pqlk_s(stream), n, (int)obs_dim, (int)act_dim, seed,
                     env_offset, t, p_done, action, next_obs, reward, done);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}
