// Replay ring kernels: insert, plain gather, fused gather+normalise+concat.
// Reference behaviour: pql/replay/simple_replay.py:40-104, pql/algo/pql_p_learner.py:49-50,66-85,
// pql/utils/common.py:139-145.  HBM-bound byte movement: one wave per record, 16-B lane accesses,
// records 128-B aligned so a random sample touches the minimum number of HBM lines.
#include "pqlk_common.h"

extern "C" int64_t pqlk_ld(int64_t cols) { return pqlk_round_up(cols < 1 ? 1 : cols, 32); }

extern "C" int64_t pqlk_replay_rec_ld(int32_t obs_dim, int32_t act_dim) {
  if (obs_dim <= 0) return 0;
  return rec_layout(obs_dim, act_dim).ld;
}

// ------------------------------------------------------------------------------------------------
// insert: row r of the five source arrays -> record (dst_start + r).  One wave per record.
__global__ __launch_bounds__(256) void k_replay_insert(float* __restrict__ records, RecLayout L, int64_t dst_start,
                                                       int64_t m, const float* __restrict__ obs, int64_t ld_obs,
                                                       const float* __restrict__ act, int64_t ld_act,
                                                       const float* __restrict__ rew, int64_t ld_rew,
                                                       const float* __restrict__ nobs, int64_t ld_nobs,
                                                       const float* __restrict__ done, int64_t ld_done) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < m; r += nwaves) {
    float* rec = records + (dst_start + r) * L.ld;
    for (int c = lane; c < L.ld; c += 64) {
      float v = 0.f;
      if (c < L.O) {
        v = obs[r * ld_obs + c];
      } else if (L.A >= 0) {
        if (c >= L.off_nobs && c < L.off_nobs + L.O) v = nobs[r * ld_nobs + (c - L.off_nobs)];
        else if (c >= L.off_act && c < L.off_act + L.A) v = act[r * ld_act + (c - L.off_act)];
        else if (c == L.off_rd) v = rew[r * ld_rew];
        else if (c == L.off_rd + 1) v = (done[r * ld_done] != 0.f) ? 1.f : 0.f;  // .bool() then .float()
      }
      rec[c] = v;
    }
  }
}

extern "C" int pqlk_replay_insert(const PqlReplayDesc* ring, int64_t dst_start, int64_t m, const float* obs,
                                  int64_t ld_obs, const float* act, int64_t ld_act, const float* rew, int64_t ld_rew,
                                  const float* next_obs, int64_t ld_nobs, const float* done, int64_t ld_done,
                                  pqlk_stream_t stream) {
  PQLK_REQUIRE(ring && ring->records && obs, PQLK_E_NULL);
  PQLK_REQUIRE(ring->obs_dim > 0 && ring->capacity > 0 && m >= 0, PQLK_E_SHAPE);
  RecLayout L = rec_layout(ring->obs_dim, ring->act_dim);
  PQLK_REQUIRE(ring->rec_ld == L.ld, PQLK_E_SHAPE);
  PQLK_REQUIRE(dst_start >= 0 && dst_start + m <= ring->capacity, PQLK_E_RANGE);
  if (L.A >= 0) PQLK_REQUIRE(act && rew && next_obs && done, PQLK_E_NULL);
  PQLK_REQUIRE(ld_obs >= L.O, PQLK_E_SHAPE);
  if (m == 0) return PQLK_OK;
  int64_t blocks = (m + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_replay_insert, dim3((unsigned)blocks), dim3(256), 0, pqlk_s(stream), ring->records, L, dst_start, m,
                     obs, ld_obs, act, ld_act, rew, ld_rew, next_obs, ld_nobs, done, ld_done);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

// ------------------------------------------------------------------------------------------------
// plain gather: five contiguous outputs, byte-exact copies (done already 0.0/1.0).
__global__ __launch_bounds__(256) void k_replay_gather(const float* __restrict__ records, RecLayout L, int64_t capacity,
                                                       const int64_t* __restrict__ idx, int64_t b,
                                                       float* __restrict__ o_obs, float* __restrict__ o_act,
                                                       float* __restrict__ o_rew, float* __restrict__ o_nobs,
                                                       float* __restrict__ o_done) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < b; r += nwaves) {
    int64_t src = idx[r];
    if (src < 0 || src >= capacity) src = 0;  // never fault on a bad index; host validates in debug paths
    const float4* rec4 = reinterpret_cast<const float4*>(records + src * L.ld);
    const int nchunk = L.used >> 2;
    for (int q = lane; q < nchunk; q += 64) {
      float4 v = rec4[q];
      const float e[4] = {v.x, v.y, v.z, v.w};
      const int c = q << 2;
      if (c < L.o4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < L.O) o_obs[r * L.O + c + j] = e[j];
      } else if (L.A >= 0) {
        if (c < L.off_act) {
          const int cc = c - L.off_nobs;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (cc + j < L.O) o_nobs[r * L.O + cc + j] = e[j];
        } else if (c < L.off_rd) {
          const int cc = c - L.off_act;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (cc + j < L.A) o_act[r * L.A + cc + j] = e[j];
        } else {
          o_rew[r] = e[0];
          o_done[r] = e[1];
        }
      }
    }
  }
}

extern "C" int pqlk_replay_gather(const PqlReplayDesc* ring, const int64_t* idx, int64_t b, float* obs, float* act,
                                  float* rew, float* next_obs, float* done, pqlk_stream_t stream) {
  PQLK_REQUIRE(ring && ring->records && idx && obs, PQLK_E_NULL);
  PQLK_REQUIRE(ring->obs_dim > 0 && ring->capacity > 0 && b >= 0, PQLK_E_SHAPE);
  RecLayout L = rec_layout(ring->obs_dim, ring->act_dim);
  PQLK_REQUIRE(ring->rec_ld == L.ld, PQLK_E_SHAPE);
  if (L.A >= 0) PQLK_REQUIRE(act && rew && next_obs && done, PQLK_E_NULL);
  if (b == 0) return PQLK_OK;
  int64_t blocks = (b + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_replay_gather, dim3((unsigned)blocks), dim3(256), 0, pqlk_s(stream), ring->records, L,
                     ring->capacity, idx, b, obs, act, rew, next_obs, done);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

// ------------------------------------------------------------------------------------------------
// fused gather: sample + normalise (+-5 clamp) + concat into the padded GEMM input tiles.
// IEEE division on purpose.  A reciprocal per column + two fma corrections (Markstein) also gives the correctly rounded
// quotient and was tried (round 2): with the range / exceptional-significand guards it needs it is ~15 instructions against
// the hardware sequence's ~11, and measured 0.9 us SLOWER per 32768-row launch.
__device__ __forceinline__ float norm1(float x, float mean, float sd, int clamp5) {
  float y = (x - mean) / sd;  // IEEE division: bit-identical to (x-mean)/sqrt(var+eps) evaluated by torch
  if (clamp5) y = fminf(fmaxf(y, -5.f), 5.f);
  return y;
}

#define GATHER_MAX_OBS 1024

// store up to 4 consecutive columns [col, col+4) of a row, clipped to `limit` columns; one 16-B store when the
// destination is 16-B aligned and unclipped, scalar stores otherwise
__device__ __forceinline__ void store4(float* __restrict__ row, int col, int limit, const float e[4], bool aligned) {
  if (aligned && col + 4 <= limit) {
    *reinterpret_cast<float4*>(row + col) = make_float4(e[0], e[1], e[2], e[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (col + j < limit) row[col + j] = e[j];
  }
}

// R rows per wave per trip with all their record loads in flight at once (memory-level parallelism for the random
// HBM reads); CH = 16-B chunks per lane per record (1 covers records up to 1 KiB, e.g. cfg #2's 784 used bytes).
template <bool HAS_NORM, int R, int CH>
__global__ __launch_bounds__(256) void k_replay_gather_fused(const float* __restrict__ records, RecLayout L,
                                                             int64_t capacity, const int64_t* __restrict__ idx, int64_t b,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             float eps, int clamp5, float* __restrict__ x_sa, int64_t ld_sa,
                                                             float* __restrict__ xn_sa, float* __restrict__ xn_obs,
                                                             int64_t ld_o, float* __restrict__ o_rew,
                                                             float* __restrict__ o_done, int write_pads) {
  // per-column mean and sd = sqrt(var + eps) (correctly rounded sqrtf) staged once per block: the per-element work
  // is then one subtract and one IEEE divide instead of a divide AND a square root
  __shared__ float s_mean[HAS_NORM ? GATHER_MAX_OBS : 1];
  __shared__ float s_sd[HAS_NORM ? GATHER_MAX_OBS : 1];
  if (HAS_NORM) {
    for (int c = threadIdx.x; c < L.O; c += 256) {
      s_mean[c] = mean[c];
      s_sd[c] = sqrtf(var[c] + eps);
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nchunk = L.used >> 2;
  const int A = L.A < 0 ? 0 : L.A;
  const int sa_cols = L.O + A;
  const bool act_aligned = (L.O & 3) == 0;
  for (int64_t r0 = wave * R; r0 < b; r0 += nwaves * R) {
    float4 v[R][CH];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i;
      int64_t src = r < b ? idx[r] : 0;
      if (src < 0 || src >= capacity) src = 0;  // never fault on a bad index
      const float4* rec4 = reinterpret_cast<const float4*>(records + src * L.ld);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = lane + 64 * c;
        v[i][c] = q < nchunk ? rec4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i;
      if (r >= b) break;
      float* xs = x_sa ? x_sa + r * ld_sa : nullptr;
      float* xns = xn_sa ? xn_sa + r * ld_sa : nullptr;
      float* xno = xn_obs ? xn_obs + r * ld_o : nullptr;
#pragma unroll
      for (int cch = 0; cch < CH; ++cch) {
        const int q = lane + 64 * cch;
        if (q >= nchunk) continue;
        float e[4] = {v[i][cch].x, v[i][cch].y, v[i][cch].z, v[i][cch].w};
        const int c = q << 2;
        if (c < L.o4) {  // obs
          if (HAS_NORM) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (c + j < L.O) e[j] = norm1(e[j], s_mean[c + j], s_sd[c + j], clamp5);
          }
          if (xs) store4(xs, c, L.O, e, true);
          if (L.A < 0 && xno) store4(xno, c, L.O, e, true);  // obs-only ring: the sample IS the learner's obs batch
        } else if (L.A >= 0) {
          if (c < L.off_act) {  // next_obs
            const int cc = c - L.off_nobs;
            if (HAS_NORM) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (cc + j < L.O) e[j] = norm1(e[j], s_mean[cc + j], s_sd[cc + j], clamp5);
            }
            if (xns) store4(xns, cc, L.O, e, true);
            if (xno) store4(xno, cc, L.O, e, true);
          } else if (c < L.off_rd) {  // action -> columns O.. of the critic input
            const int cc = c - L.off_act;
            if (xs) store4(xs + L.O, cc, L.A, e, act_aligned);
          } else {
            if (o_rew) o_rew[r] = e[0];
            if (o_done) o_done[r] = e[1];
          }
        }
      }
      if (L.A < 0 && xs) {
        for (int c = L.O + lane; c < sa_cols; c += 64) xs[c] = 0.f;
      }
      if (!write_pads) continue;
      // zero the pad columns so the GEMM K-loop can run over the padded width unchecked
      for (int c = sa_cols + lane; c < ld_sa; c += 64) {
        if (xs) xs[c] = 0.f;
        if (xns) xns[c] = 0.f;
      }
      if (xno)
        for (int c = L.O + lane; c < ld_o; c += 64) xno[c] = 0.f;
    }
  }
}

// (x - mean) / sd with the division as ONE double-precision product rounded once to fp32 -- bit for bit the IEEE quotient the
// reference's `(x - mean) / sqrt(var + eps)` rounds to (common.py:139-145), in 3 instructions per element instead of the 12 of the
// fp32 division sequence (v_div_scale x2, v_rcp, 6 FMAs / multiplies, v_div_fmas, v_div_fixup): the gather kernels spent ~100 VALU
// instructions per record on four divisions.  rd = 1 / (double)sd is formed once per lane (correctly rounded: relative error
// <= 2^-53), the product adds <= 2^-53, so the double result is within 2^-51.9 (relative) of a / sd.  The exact quotient of two
// fp32 numbers is never that close to a boundary between two fp32 results: a boundary m has <= 25 significant bits, a - m sd != 0
// is a multiple of ulp(m) ulp(sd), which puts |a / sd - m| / |a / sd| above 2^-49 (normal and denormal results alike; the overflow
// threshold is one more such boundary).  Signed zeros, infinities and NaN come out as IEEE division gives them (a * (1 / inf) =
// a * 0).  tests: the gather tests are bit-exact against the oracle's fp32 division; test_gather_division_is_the_ieee_quotient.
__device__ __forceinline__ float norm_div(float a, double rd) { return (float)((double)a * rd); }

// Fast path for transition rings whose record fits one 16-B chunk per lane (<= 1 KiB used, e.g. cfg #2 and #5): the field
// decode, destination and normalisation constants of a lane depend only on its chunk index, so they are computed ONCE
// per kernel and the per-row work is load -> (sub, IEEE div, clamp) x4 -> one or two 16-B stores.  The generic kernel
// above re-decodes the field per row and is instruction-issue bound (~450 instructions per row).
// (launch-shape overrides for tools/bench_gather.py arrive in the call's flag word: no state between calls)

template <bool HAS_NORM, int R>
__global__ __launch_bounds__(256) void k_replay_gather_fast(const float* __restrict__ records, RecLayout L, int64_t capacity,
                                                            const int64_t* __restrict__ idx, int64_t b,
                                                            const float* __restrict__ mean, const float* __restrict__ var,
                                                            float eps, int clamp5, float* __restrict__ x_sa, int64_t ld_sa,
                                                            float* __restrict__ xn_sa, float* __restrict__ xn_obs, int64_t ld_o,
                                                            float* __restrict__ o_rew, float* __restrict__ o_done, int write_pads,
                                                            int nt_loads, int halves) {
  typedef float f4n __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  // halves == 2: records of 1-2 KiB (cfg #4: 1792 B) -- the odd waves of a block take the second KiB of the rows the even waves
  // take the first of, so a lane still owns ONE 16-B chunk and its plan
  const int half = halves == 2 ? (threadIdx.x >> 6) & 1 : 0;
  const int cl = lane + 64 * half;   // this lane's chunk of the record
  const int c = cl << 2;
  const int nchunk = L.used >> 2;
  // per-lane plan
  float* dstA = nullptr;
  float* dstB = nullptr;
  int64_t ldA = 0, ldB = 0;
  int ncol = -1;    // column of the normalisation constants, -1 = copy
  int nvalid = 4;   // logical elements in this chunk (< 4 only in a field's last chunk when O or A is not a multiple of 4)
  bool vecA = true, is_rd = false;
  if (cl < nchunk) {
    if (c < L.o4) {
      dstA = x_sa ? x_sa + c : nullptr; ldA = ld_sa; ncol = c; nvalid = min(4, L.O - c);
    } else if (c < L.off_act) {
      const int cc = c - L.off_nobs;
      dstA = xn_sa ? xn_sa + cc : nullptr; ldA = ld_sa;
      dstB = xn_obs ? xn_obs + cc : nullptr; ldB = ld_o;
      ncol = cc; nvalid = min(4, L.O - cc);
    } else if (c < L.off_rd) {
      const int cc = c - L.off_act;
      dstA = x_sa ? x_sa + L.O + cc : nullptr; ldA = ld_sa; nvalid = min(4, L.A - cc);
      vecA = (L.O & 3) == 0;   // action columns start at O: 16-B aligned only then
    } else {
      is_rd = true;
    }
  }
  vecA = vecA && nvalid == 4;
  float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
  double rd[4] = {1.0, 1.0, 1.0, 1.0};   // 1 / sd (norm_div)
  const bool do_norm = HAS_NORM && ncol >= 0;
  if (do_norm) {   // scalar loads: O need not be a multiple of 4; invalid tail elements keep (0, 1)
    float mm[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nvalid) { mm[j] = mean[ncol + j]; ss[j] = sqrtf(var[ncol + j] + eps); rd[j] = 1.0 / (double)ss[j]; }
    m4 = make_float4(mm[0], mm[1], mm[2], mm[3]);
  }
  // pad columns [O+A, ld_sa) of x_sa / xn_sa and [O, ld_o) of xn_obs: lanes take one 16-B zero store each
  // (scalar, <= 31 per matrix: ld - cols < 32 + 3; pads are a few dozen bytes per row)
  const int sa_cols = L.O + L.A;
  const int npad_sa = (int)(ld_sa - sa_cols), npad_o = xn_obs ? (int)(ld_o - L.O) : 0;
  const bool pad_vec = (sa_cols & 3) == 0 && (L.O & 3) == 0;

  const int hs = halves - 1;   // 0 or 1: a shift, not a 64-bit division (which cost the 8192-row launch 0.3 us)
  // (readfirstlane: the wave's row numbers are provably uniform, so the index loads below are scalar loads -- their own counter,
  //  nothing to do with the in-order vector-memory queue the record loads and the tile stores share)
  const int64_t wave = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) >> hs;
  const int64_t nwaves = ((int64_t)gridDim.x * 4) >> hs;
  // The sample indices run ONE trip ahead of the records they address: idx -> record -> store is a dependent chain per trip, and
  // a wave of the K-batch launch makes five or six trips (65 536 rows over 1536 blocks): the index latency of every trip but the
  // first now passes under the previous trip's record loads and stores.
  int64_t nsrc[R];
#pragma unroll
  for (int i = 0; i < R; ++i) nsrc[i] = wave * R + i < b ? idx[wave * R + i] : 0;
  for (int64_t r0 = wave * R; r0 < b; r0 += nwaves * R) {
    float4 v[R];
    int64_t srcs[R];
#pragma unroll
    for (int i = 0; i < R; ++i) srcs[i] = nsrc[i];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t rn = r0 + nwaves * R + i;
      nsrc[i] = rn < b ? idx[rn] : 0;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int64_t src = srcs[i];
      if (src < 0 || src >= capacity) src = 0;
      if (cl < nchunk) {
        if (nt_loads & 1) {   // records are read once per sample: keep them out of the caches the output tiles will be read from
          const f4n t = __builtin_nontemporal_load(reinterpret_cast<const f4n*>(records + src * L.ld) + cl);
          v[i] = make_float4(t[0], t[1], t[2], t[3]);
        } else {
          v[i] = reinterpret_cast<const float4*>(records + src * L.ld)[cl];
        }
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i;
      if (r >= b) break;
      float4 x = v[i];
      if (do_norm) {
        x.x = norm_div(x.x - m4.x, rd[0]); x.y = norm_div(x.y - m4.y, rd[1]);
        x.z = norm_div(x.z - m4.z, rd[2]); x.w = norm_div(x.w - m4.w, rd[3]);
        if (clamp5) {
          x.x = fminf(fmaxf(x.x, -5.f), 5.f); x.y = fminf(fmaxf(x.y, -5.f), 5.f);
          x.z = fminf(fmaxf(x.z, -5.f), 5.f); x.w = fminf(fmaxf(x.w, -5.f), 5.f);
        }
      }
      const float e[4] = {x.x, x.y, x.z, x.w};
      if (dstA) {
        if (vecA) {
          if (nt_loads & 2) __builtin_nontemporal_store(f4n{x.x, x.y, x.z, x.w}, reinterpret_cast<f4n*>(dstA + r * ldA));
          else *reinterpret_cast<float4*>(dstA + r * ldA) = x;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < nvalid) dstA[r * ldA + j] = e[j];
        }
      }
      if (dstB) {
        if (nvalid == 4) *reinterpret_cast<float4*>(dstB + r * ldB) = x;
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < nvalid) dstB[r * ldB + j] = e[j];
        }
      }
      if (is_rd) {
        if (o_rew) o_rew[r] = x.x;
        if (o_done) o_done[r] = x.y;
      }
      if (!write_pads || half) continue;
      if (pad_vec) {   // pads start on a 16-B boundary: one 16-B zero store per lane
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < (npad_sa >> 2)) {
          if (x_sa) *reinterpret_cast<float4*>(x_sa + r * ld_sa + sa_cols + 4 * lane) = z4;
          if (xn_sa) *reinterpret_cast<float4*>(xn_sa + r * ld_sa + sa_cols + 4 * lane) = z4;
        }
        if (lane < (npad_o >> 2)) *reinterpret_cast<float4*>(xn_obs + r * ld_o + L.O + 4 * lane) = z4;
      } else {
        if (lane < npad_sa) {
          if (x_sa) x_sa[r * ld_sa + sa_cols + lane] = 0.f;
          if (xn_sa) xn_sa[r * ld_sa + sa_cols + lane] = 0.f;
        }
        if (lane < npad_o) xn_obs[r * ld_o + L.O + lane] = 0.f;
      }
    }
  }
}

// Fast path for OBS-ONLY rings (the P-learner's replay, pql_p_learner.py:32-37,49-50).  A record is O floats -- 384 B at cfg #2,
// 22 of a wave's 64 lanes at one 16-B chunk per lane -- so with one record per wave instruction (the generic kernel above, which
// also re-decodes the field per row: 18 us per 32 768 rows, 0.16 of the HBM roof) two thirds of every load and store instruction
// are idle lanes.  Here a record takes P = 2^lgp lanes (the power of two >= its chunk count) and G = 64 / P records share each
// instruction; a lane's chunk, destinations and normalisation constants are fixed for the whole kernel (as in
// k_replay_gather_fast), a wave keeps R x G records in flight, and the sample indices run one trip ahead of the records.
template <bool HAS_NORM, int R>
__global__ __launch_bounds__(256) void k_replay_gather_obs(const float* __restrict__ records, RecLayout L, int64_t capacity,
                                                           const int64_t* __restrict__ idx, int64_t b,
                                                           const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                           int clamp5, float* __restrict__ x_sa, int64_t ld_sa,
                                                           float* __restrict__ x_obs, int64_t ld_o, int write_pads, int lgp) {
  const int lane = threadIdx.x & 63;
  const int P = 1 << lgp, G = 64 >> lgp;
  const int grp = lane >> lgp, cl = lane & (P - 1);   // record of the instruction, chunk of the record
  const int c = cl << 2;
  const int nchunk = L.used >> 2;
  const bool active = cl < nchunk;
  const int nvalid = active ? min(4, L.O - c) : 0;   // < 4 only in the last chunk when O is not a multiple of 4
  float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
  double rd[4] = {1.0, 1.0, 1.0, 1.0};   // 1 / sd (norm_div)
  if (HAS_NORM && active) {
    float mm[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nvalid) { mm[j] = mean[c + j]; ss[j] = sqrtf(var[c + j] + eps); rd[j] = 1.0 / (double)ss[j]; }
    m4 = make_float4(mm[0], mm[1], mm[2], mm[3]);
  }
  const int64_t rows_trip = (int64_t)R * G;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  // Every index and record load below is UNCONDITIONAL straight-line code (rows past b re-read index b - 1, idle lanes re-read
  // chunk 0 of their record: same cache line, nothing stored).  With the loads under `r < b` / `active` branches the compiler put a
  // full `s_waitcnt vmcnt(0)` between the R record loads of a trip -- three dependent memory round trips where one was meant
  // (tools/kernel_resources.py full_drains: 7 -> see the test).
  const int clq = active ? cl : 0;
  int64_t nsrc[R];
#pragma unroll
  for (int i = 0; i < R; ++i) nsrc[i] = idx[min(wave * rows_trip + i * G + grp, b - 1)];
  for (int64_t r0 = wave * rows_trip; r0 < b; r0 += nwaves * rows_trip) {
    float4 v[R];
    int64_t srcs[R];
#pragma unroll
    for (int i = 0; i < R; ++i) srcs[i] = nsrc[i];
#pragma unroll
    for (int i = 0; i < R; ++i) nsrc[i] = idx[min(r0 + nwaves * rows_trip + i * G + grp, b - 1)];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int64_t src = srcs[i];
      src = (src < 0 || src >= capacity) ? 0 : src;   // never fault on a bad index
      v[i] = reinterpret_cast<const float4*>(records + src * L.ld)[clq];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i * G + grp;
      if (r >= b) continue;
      float4 x = v[i];
      if (HAS_NORM) {
        x.x = norm_div(x.x - m4.x, rd[0]); x.y = norm_div(x.y - m4.y, rd[1]);
        x.z = norm_div(x.z - m4.z, rd[2]); x.w = norm_div(x.w - m4.w, rd[3]);
        if (clamp5) {
          x.x = fminf(fmaxf(x.x, -5.f), 5.f); x.y = fminf(fmaxf(x.y, -5.f), 5.f);
          x.z = fminf(fmaxf(x.z, -5.f), 5.f); x.w = fminf(fmaxf(x.w, -5.f), 5.f);
        }
      }
      if (nvalid == 4) {
        if (x_sa) *reinterpret_cast<float4*>(x_sa + r * ld_sa + c) = x;
        if (x_obs) *reinterpret_cast<float4*>(x_obs + r * ld_o + c) = x;
      } else {
        const float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < nvalid) {
            if (x_sa) x_sa[r * ld_sa + c + j] = e[j];
            if (x_obs) x_obs[r * ld_o + c + j] = e[j];
          }
      }
      if (!write_pads) continue;   // (the learners keep the pads zero themselves: PQLK_GATHER_PADS_ZERO)
      if (x_sa)
        for (int64_t k = L.O + cl; k < ld_sa; k += P) x_sa[r * ld_sa + k] = 0.f;
      if (x_obs)
        for (int64_t k = L.O + cl; k < ld_o; k += P) x_obs[r * ld_o + k] = 0.f;
    }
  }
}

template <bool HAS_NORM>
static void launch_gather_fused(int nchunk, unsigned blocks, hipStream_t st, const float* records, RecLayout L, int64_t capacity,
                                const int64_t* idx, int64_t b, const float* mean, const float* var, float eps, int clamp5,
                                float* x_sa, int64_t ld_sa, float* xn_sa, float* xn_obs, int64_t ld_o, float* rew, float* done,
                                int write_pads) {
#define PQLK_GF(R, CH)                                                                                                   \
  hipLaunchKernelGGL((k_replay_gather_fused<HAS_NORM, R, CH>), dim3(blocks), dim3(256), 0, st, records, L, capacity, idx, b, \
                     mean, var, eps, clamp5, x_sa, ld_sa, xn_sa, xn_obs, ld_o, rew, done, write_pads)
  if (nchunk <= 64) PQLK_GF(4, 1);
  else if (nchunk <= 128) PQLK_GF(2, 2);
  else if (nchunk <= 256) PQLK_GF(1, 4);
  else PQLK_GF(1, 16);
#undef PQLK_GF
}

extern "C" int pqlk_replay_gather_fused(const PqlReplayDesc* ring, const int64_t* idx, int64_t b, const float* mean,
                                        const float* var, float eps, int flags, float* x_sa, int64_t ld_sa,
                                        float* xn_sa, float* xn_obs, int64_t ld_o, float* rew, float* done,
                                        pqlk_stream_t stream) {
  const int clamp5 = flags & PQLK_GATHER_CLAMP5;
  const int write_pads = (flags & PQLK_GATHER_PADS_ZERO) ? 0 : 1;
  const int tune_R = (flags >> 8) & 15, tune_wpc = (flags >> 12) & 63;
  int nt_loads = ((flags & PQLK_GATHER_NT_LOADS) ? 1 : 0) | ((flags & PQLK_GATHER_NT_STORES) ? 2 : 0);
  PQLK_REQUIRE(ring && ring->records && idx, PQLK_E_NULL);
  PQLK_REQUIRE(ring->obs_dim > 0 && ring->capacity > 0 && b >= 0, PQLK_E_SHAPE);
  RecLayout L = rec_layout(ring->obs_dim, ring->act_dim);
  PQLK_REQUIRE(ring->rec_ld == L.ld, PQLK_E_SHAPE);
  PQLK_REQUIRE((mean == nullptr) == (var == nullptr), PQLK_E_NULL);
  if (mean) PQLK_REQUIRE(L.O <= GATHER_MAX_OBS, PQLK_E_UNSUPPORTED);
  if (x_sa || xn_sa) PQLK_REQUIRE(ld_sa % 32 == 0 && ld_sa >= L.O + (L.A < 0 ? 0 : L.A), PQLK_E_ALIGN);
  if (xn_obs) PQLK_REQUIRE(ld_o % 32 == 0 && ld_o >= L.O, PQLK_E_ALIGN);
  if (L.A < 0) PQLK_REQUIRE(xn_sa == nullptr, PQLK_E_UNSUPPORTED);
  if (b == 0) return PQLK_OK;
  PQLK_REQUIRE((L.used >> 2) <= 1024, PQLK_E_UNSUPPORTED);
  const int nchunk = L.used >> 2;
  const int rows_per_wave = nchunk <= 64 ? 4 : (nchunk <= 128 ? 2 : 1);
  int64_t blocks = (b + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
  if (blocks > 2048) blocks = 2048;
  // aligned transition-ring shape -> lean kernel (pads: at most 64 16-B chunks in total, one lane each)
  // (the pad-column bound only matters when this call has to zero the pads: one lane per pad chunk)
  const bool fast = L.A >= 0 && nchunk <= 128 && (!write_pads || ((ld_sa - L.O - L.A) <= 64 && (!xn_obs || (ld_o - L.O) <= 64))) &&
                    pqlk_aligned16(x_sa) && pqlk_aligned16(xn_sa) && pqlk_aligned16(xn_obs);
  if (fast) {
    // rows in flight per wave: 2 up to 16 Ki rows (more waves, shorter dependent idx -> row chain: 9.3 vs 10.5 us at 8192),
    // 4 beyond (same time at 32 Ki rows, fewer blocks)
    // 2 rows in flight per wave, at most 12 (rounds 1-2: 24) resident waves per CU: beyond that the grid-stride loop takes
    // further trips.  Measured at cfg 5 (32768 rows x 1 KiB, pads not re-zeroed; tools/bench_gather.py): 24 waves/CU x 2 rows
    // 18.5 us, 16 x 2 19.5, 12 x 2 21.4, 32 x 2 22.3, 16 x 4 20.3, 32 x 4 (one trip per wave, the round-1 shape) 23.0, 8 rows
    // in flight 30; at cfg 2 (8192 rows, one trip whatever the cap) 2 rows per wave 7.1 us vs 8.1 (4) and 9.0 (1).
    // Non-temporal record loads: no gain (18.9 vs 18.5).  Requesting trip k+1's records before trip k is normalised and stored
    // (software pipeline): WORSE, 21.5 us -- on gfx9 stores share the in-order vmcnt queue with loads, so the wait for the
    // prefetched records also waits for the previous trip's store acknowledgements.  With the ring small enough to sit in the
    // Infinity Cache (51 MB) the same launch takes 14.9 us: the 5-GB ring's random 1-KiB reads cost ~4 us on top of that, and
    // ring size hardly matters beyond the cache (0.25 GB 18.0 us, 5 GB 19.3, 20 GB 20.1: not a TLB effect).  A loader / consumer
    // split (one wave per workgroup issuing LDS-DMA record loads into a 16-slot LDS ring behind a counted vmcnt, three waves
    // normalising and storing out of the ring, so that no wave ever mixes loads and stores) was built and measured too:
    // 20.2 us at best -- no better than this kernel, i.e. the limit is the memory system's rate for this mix (about 4 TB/s of
    // bytes moved), not the way one wave's loads and stores queue.
    // Round 3 (the K-batch launches: 65 536 rows at cfg 2, five or more trips per wave): with the sample indices loaded one trip
    // ahead, as SCALAR loads (k_replay_gather_fast), the optimum moved to 12 waves per CU -- 25.1 us at cfg 2 x 8 against 26.5
    // at 24 and 31.5 at 8 (before: 35.3 at 12, 27.0 at 24); cfg 4 x 8 54.9-57.0 at 12-32; cfg 5 x 8 flat, 145-148 us at 8-32.
    int R = 2;
    if (tune_R == 1 || tune_R == 2 || tune_R == 4 || tune_R == 8) R = tune_R;
    const int halves = nchunk > 64 ? 2 : 1;   // 1-2 KiB records: two waves per row
    const int rows_blk = 4 / halves * R;
    int64_t fb = (b + rows_blk - 1) / rows_blk;
    const int wpc = tune_wpc ? tune_wpc : 12;
    if (fb > 256 * (int64_t)wpc / 4) fb = 256 * (int64_t)wpc / 4;
    // Launches whose record reads alone exceed what the 256-MB Infinity Cache can hold beside the output tiles (cfg #5 x 8: 262 144
    // rows x 1 KiB) take the records with non-temporal loads: they are read once and would only push the tiles out.  Round 4,
    // tools/bench_gather.py cfg5x8: 150 -> 98 us back to back, 129 -> 123 us one launch at a time; cfg #2 x 8 (59 MB of records):
    // no difference either way, left alone.
    if (!tune_R && !tune_wpc && b * (int64_t)L.ld * 4 > ((int64_t)192 << 20)) nt_loads |= 1;
    const dim3 g((unsigned)fb), t(256);
#define PQLK_GATHER_FAST(NORM, RR) \
    hipLaunchKernelGGL((k_replay_gather_fast<NORM, RR>), g, t, 0, pqlk_s(stream), ring->records, L, ring->capacity, idx, b, mean, var, \
                       eps, clamp5, x_sa, ld_sa, xn_sa, xn_obs, ld_o, rew, done, write_pads, nt_loads, halves)
#define PQLK_GATHER_FAST_R(NORM) \
    do { if (R == 1) PQLK_GATHER_FAST(NORM, 1); else if (R == 2) PQLK_GATHER_FAST(NORM, 2); else if (R == 4) PQLK_GATHER_FAST(NORM, 4); \
         else PQLK_GATHER_FAST(NORM, 8); } while (0)
    if (mean) PQLK_GATHER_FAST_R(true); else PQLK_GATHER_FAST_R(false);
#undef PQLK_GATHER_FAST_R
#undef PQLK_GATHER_FAST
    PQLK_LAUNCH_CHECK();
    return PQLK_OK;
  }
  // obs-only ring whose record fits one 16-B chunk per lane (O <= 256): several records per wave instruction
  if (L.A < 0 && nchunk <= 64 && pqlk_aligned16(x_sa) && pqlk_aligned16(xn_obs) && (x_sa || xn_obs)) {
    int lgp = 0;
    while ((1 << lgp) < nchunk) ++lgp;
    const int G = 64 >> lgp;
    // rows in flight per wave: R x G, R chosen so that the launch still has ~16 waves per CU to issue from (32 768 rows of
    // cfg #2, G = 2: R = 4, 4096 waves, one trip); waves per CU capped at 16 (grid-stride trips beyond)
    const int64_t groups = (b + G - 1) / G;
    int R = groups >= 4 * 4096 ? 4 : (groups >= 2 * 4096 ? 2 : 1);
    if (tune_R == 1 || tune_R == 2 || tune_R == 4) R = tune_R;
    const int wpc = tune_wpc ? tune_wpc : 16;
    int64_t fb = (groups + 4 * R - 1) / (4 * R);
    if (fb > 256 * (int64_t)wpc / 4) fb = 256 * (int64_t)wpc / 4;
    const dim3 g((unsigned)fb), t(256);
#define PQLK_GATHER_OBS(NORM, RR) \
    hipLaunchKernelGGL((k_replay_gather_obs<NORM, RR>), g, t, 0, pqlk_s(stream), ring->records, L, ring->capacity, idx, b, mean, var, eps, \
                       clamp5, x_sa, ld_sa, xn_obs, ld_o, write_pads, lgp)
#define PQLK_GATHER_OBS_R(NORM) \
    do { if (R == 1) PQLK_GATHER_OBS(NORM, 1); else if (R == 2) PQLK_GATHER_OBS(NORM, 2); else PQLK_GATHER_OBS(NORM, 4); } while (0)
    if (mean) PQLK_GATHER_OBS_R(true); else PQLK_GATHER_OBS_R(false);
#undef PQLK_GATHER_OBS_R
#undef PQLK_GATHER_OBS
    PQLK_LAUNCH_CHECK();
    return PQLK_OK;
  }
  if (mean)
    launch_gather_fused<true>(nchunk, (unsigned)blocks, pqlk_s(stream), ring->records, L, ring->capacity, idx, b, mean, var, eps,
                              clamp5, x_sa, ld_sa, xn_sa, xn_obs, ld_o, rew, done, write_pads);
  else
    launch_gather_fused<false>(nchunk, (unsigned)blocks, pqlk_s(stream), ring->records, L, ring->capacity, idx, b, mean, var, eps,
                               clamp5, x_sa, ld_sa, xn_sa, xn_obs, ld_o, rew, done, write_pads);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}
