// fp32-MFMA kernels for GEMMs with a NARROW output (<= 64 columns) and a long row axis: the actor's action head
// (256 -> 16..21, or 2A for SAC), the C51 logits head (-> 51).  k_gemm's 64x64 block tile wastes 3/4 of its MFMAs on a
// 16-wide head, launches only M/64 = 128 blocks at batch 8192 and pays a full LDS pipeline for 16 KB of weights.  Here
// one wave owns a 32-row tile and NT 32-column output tiles; BOTH operands come straight from global memory as the
// k-contiguous float4 quads the MFMA fragments want (x row r / weight row n, reduction index 8 k8 + 4 h + t): the weight
// block is a few KB and lives in L1/L2, no LDS, no barrier.
#pragma once
#include "pqlk_common.h"

typedef float narrow_acc_t __attribute__((ext_vector_type(16)));

// element e of the NA partial accumulators of one tile, summed pairwise
template <int NA>
__device__ __forceinline__ float fold_acc(const narrow_acc_t (&a)[NA], int e) {
  if (NA == 4) return (a[0][e] + a[1][e]) + (a[2 % NA][e] + a[3 % NA][e]);
  return a[0][e] + a[1 % NA][e];
}

template <int NT, int EPI, int D>
__global__ __launch_bounds__(64) void k_fwd_narrow(GemmP p) {
  typedef narrow_acc_t acc_t;
  // D = ring depth in reduction steps of 8 (K8 is a multiple of D: straight-line refills, counted waits).  One wave per SIMD
  // and nothing else to hide latency behind: the loop is bound by (memory latency) / D per step, so the launcher picks the
  // deepest ring that divides K8 and fits the registers.
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int g = blockIdx.y;
  const int m0 = blockIdx.x * 32;
  const int row = min(m0 + r, p.M - 1);   // clamped for the loads; stores are guarded
  const float4* xp = reinterpret_cast<const float4*>(p.A + (long long)g * p.sA + (long long)row * p.lda) + h;
  const float4* wp[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)   // weight rows past N are clamped: their output columns are never stored
    wp[j] = reinterpret_cast<const float4*>(p.B + (long long)g * p.sB + (long long)min(32 * j + r, p.N - 1) * p.ldb) + h;
  // A wave is alone on its SIMD here, and a dependent v_mfma_f32_32x32x2 chain issues only every ~125 cycles (measured:
  // 128 chained MFMAs = 7.7 us) against 64 for independent ones: each output tile therefore accumulates into NA
  // accumulators by reduction-step parity (step s -> s mod NA), summed pairwise at the end.  Deterministic; the
  // summation order differs from k_gemm's single chain by reassociation only.
  constexpr int NA = NT == 1 ? 4 : 2;
  acc_t acc[NT][NA];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][a][e] = 0.f;
  const int K8 = p.K >> 3;   // K is a multiple of 32
  float4 xq[D], wq[D][NT];
#pragma unroll
  for (int s = 0; s < D; ++s) {
    const int ks = min(s, K8 - 1);
    xq[s] = xp[2 * ks];
#pragma unroll
    for (int j = 0; j < NT; ++j) wq[s][j] = wp[j][2 * ks];
  }
  __builtin_amdgcn_sched_barrier(0);
  for (int k8 = 0; k8 < K8; k8 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const float av[4] = {xq[s].x, xq[s].y, xq[s].z, xq[s].w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = t == 0 ? wq[s][j].x : t == 1 ? wq[s][j].y : t == 2 ? wq[s][j].z : wq[s][j].w;
          acc[j][s % NA] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av[t], acc[j][s % NA], 0, 0, 0);
        }
      const int kn = min(k8 + s + D, K8 - 1);   // clamped, unconditional refill
      xq[s] = xp[2 * kn];
#pragma unroll
      for (int j = 0; j < NT; ++j) wq[s][j] = wp[j][2 * kn];
      __builtin_amdgcn_sched_barrier(0);   // keep the refill HERE: hipcc otherwise sinks every load to just before its use
    }
  }
  // epilogue: lane (r, h) owns row m0 + r, columns 32 j + 8 q + 4 h + {0..3}
  const int orow = m0 + r;
  if (orow >= p.M) return;
  float* C = p.C + (long long)g * p.sC;
  const float* bias = p.bias ? p.bias + (long long)g * p.sBias : nullptr;
  const float* aux = p.aux ? p.aux + (long long)g * p.sAux : nullptr;
  float* C2 = (p.C2 && g == 0) ? p.C2 : nullptr;
  // 16-B path: every quad lies inside [0, N) or inside the pad, and all four streams are 16-B aligned.  All loads of the
  // epilogue are issued before any of its math (a wave alone on its SIMD has nothing else to hide their latency behind).
  auto al16 = [](const void* q) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0; };
  const bool vec = (p.N & 3) == 0 && (p.ncols_store & 3) == 0 && bias && al16(bias) && al16(C) && (!aux || al16(aux)) &&
                   (!C2 || (al16(C2) && (p.ldc2 & 3) == 0));
  if (vec) {
    float4 b4[NT][4], n4[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = min(32 * j + 8 * q + 4 * h, p.N - 4);
        b4[j][q] = *reinterpret_cast<const float4*>(bias + c);
        n4[j][q] = (EPI == EPI_TANH_NOISE) ? *reinterpret_cast<const float4*>(aux + (long long)orow * p.N + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * j + 8 * q + 4 * h;
        if (c >= p.ncols_store) continue;
        float v[4] = {0.f, 0.f, 0.f, 0.f};   // pad quad
        if (c < p.N) {
          const float bb[4] = {b4[j][q].x, b4[j][q].y, b4[j][q].z, b4[j][q].w};
          const float nn[4] = {n4[j][q].x, n4[j][q].y, n4[j][q].z, n4[j][q].w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = 4 * q + u;
            const float dot = fold_acc<NA>(acc[j], e);
            float x = dot + bb[u];
            if (EPI == EPI_TANH) x = tanhf(x);
            else if (EPI == EPI_TANH_NOISE) {
              x = tanhf(x);
              const float nz = fminf(fmaxf(p.noise_std * nn[u], -p.noise_clip), p.noise_clip);
              x = fminf(fmaxf(x + nz, -1.f), 1.f);
            }
            v[u] = x;
          }
          if (C2) *reinterpret_cast<float4*>(C2 + (long long)orow * p.ldc2 + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
        *reinterpret_cast<float4*>(C + (long long)orow * p.ldc + c) = make_float4(v[0], v[1], v[2], v[3]);
      }
    return;
  }
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
          const int e = 4 * q + u;
          const float dot = fold_acc<NA>(acc[j], e);
          x = dot + (bias ? bias[c] : 0.f);
          if (EPI == EPI_TANH) x = tanhf(x);
          else if (EPI == EPI_TANH_NOISE) {
            x = tanhf(x);
            float nz = p.noise_std * aux[(long long)orow * p.N + c];
            nz = fminf(fmaxf(nz, -p.noise_clip), p.noise_clip);
            x = fminf(fmaxf(x + nz, -1.f), 1.f);
          }
          if (C2) C2[(long long)orow * p.ldc2 + c] = x;
        }
        C[(long long)orow * p.ldc + c] = x;
      }
}

static bool narrow_fwd_ok(const GemmP& p) { return p.N <= 64 && p.K >= 32 && (p.K & 31) == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0; }

template <int EPI>
static int launch_fwd_narrow_e(const GemmP& p, int groups, hipStream_t st) {
  const dim3 grid((unsigned)((p.M + 31) / 32), (unsigned)groups), block(64);
  const int K8 = p.K >> 3;   // multiple of 4
  if (p.N <= 32) {
    if (K8 % 16 == 0) hipLaunchKernelGGL((k_fwd_narrow<1, EPI, 16>), grid, block, 0, st, p);
    else if (K8 % 8 == 0) hipLaunchKernelGGL((k_fwd_narrow<1, EPI, 8>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((k_fwd_narrow<1, EPI, 4>), grid, block, 0, st, p);
  } else {
    if (K8 % 8 == 0) hipLaunchKernelGGL((k_fwd_narrow<2, EPI, 8>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((k_fwd_narrow<2, EPI, 4>), grid, block, 0, st, p);
  }
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

// ------------------------------------------------------------------------------------------------
// Input-gradient SLICE of the first layer: dX[m, col0 + c] * (1 - a[m, c]^2), c < ncol <= 32, summed over the nets --
// the DPG chain from the critic into the actor's tanh (pql_p_learner.py:55-58).  Output is 16-21 columns wide against a
// reduction of n_nets x 512, so k_gemm's 64x64 tile leaves 3/4 of its MFMAs and half of the CUs idle (55 us at batch
// 8192).  Here a block owns one 32-row tile and its 4 waves each take a quarter of the (net, k) reduction range; the
// four partial tiles are summed through LDS in wave order (deterministic).  Weight-side fragments need W[k][col0 + c]
// with k fastest, i.e. the transpose of the row-major slice: each wave stages ITS quarter transposed into LDS once
// (16-B global loads when col0 is 16-B aligned), batch-side fragments (dY rows) come straight from global.

// Round 4 (QPW > 0): the ACTOR's head backward rides in the same launch.  The slice's output row -- dL/d(pre-tanh action) of one
// sample, <= 16 floats -- is all the actor's last layer needs: dX_h = (dz W_h) * ELU'(H_h), dW_h += dz^T H_h, db_h += dz, a row-local
// chain that k_skinny_bwd<16> ran as a launch of its own after the (B, 16) matrix had gone through HBM (17.0 + 13.9 us at batch
// 8192).  The tile's 32 dz rows stay in LDS; wave w takes the column quads [w QPW, (w + 1) QPW) of the K_h = 16 QPW inputs, its
// lanes = QPW quads x (64 / QPW) row groups of QPW / 2 rows; the row groups are folded by a fixed xor-shuffle tree and every tile
// leaves ONE partial (N x K_h weights, then 32 bias floats: k_skinny_bwd's format) that k_reduce_slabs folds in tile order.
// Ties (min(Q1, Q2) attained by both nets: the sample sits in both runs, tie0[] of k_dpg_minnet_head) are finished in the run-1
// tile -- the lane that owns the row adds net 0's contribution from the sample's run-0 row of dZ with a plain dot product (an
// exact tie of two fp32 network outputs is a once-in-a-million-samples event) -- and skipped in the run-0 tile: every batch row is
// written exactly once, no atomics, no zero-fill.
struct SliceHeadX {
  const int* tie0;
  const float* Hh; int ldh;        // (B, ldh) the actor's last hidden activations
  const float* Wh; int N, Kh;      // (N, Kh) head weights
  float* dXh;                      // (B, Kh) dL/dZ of the actor's last hidden layer
  float* part; long long part_floats;
};

// Profiling hook (tools/probes/slice_phases.sh): -DPQLK_SLICE_SKIP=n builds a library whose k_dx_slice returns after phase n
// (1 head-operand requests, 2 weight staging, 3 MFMA loop, 4 reduction + dz tile); the difference of two builds' kernel durations is
// a phase's cost.  `keep` makes what the phase loaded or computed observable so that it is not optimised away.  Never defined
// in the product build.
#ifdef PQLK_SLICE_SKIP
#define PQLK_SLICE_PHASE_END(n, keep) do { if (PQLK_SLICE_SKIP == (n)) { if ((keep) == 1.2345e30f) p.C[0] = 1.f; return; } } while (0)
#else
#define PQLK_SLICE_PHASE_END(n, keep) do { } while (0)
#endif

template <int D, int QPW = 0>   // ring depth; K8 % D == 0
// (waves_per_eu 2: two blocks per CU -- the 66-KB weight stage allows exactly two -- so at most 256 registers per lane, AGPRs included)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_dx_slice(GemmP p, SliceHeadX x) {
  typedef float acc_t __attribute__((ext_vector_type(16)));
  extern __shared__ __attribute__((aligned(16))) float dxs_lds[];   // 4 waves x 32 columns x (kq + 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 32;
  // compact rows (GemmP::perm): this 32-row tile belongs to ONE net -- reduction over that net's K only (host: groups = 1);
  // outputs go to batch row perm[i] by atomic add (a tie of the two Q heads puts a sample in both nets' runs: two addends,
  // and a + b = b + a, so the result does not depend on which arrives first); tiles past the rows in use exit at once
  int cnet = 0;
  if (p.perm) {
    if (m0 >= p.mn[3]) return;
    cnet = m0 >= p.mn[2] ? 1 : 0;
  }
  const float* Bw = p.B + (long long)cnet * p.sB;      // (groups = 1 in compact mode, so the g * sB terms below vanish)
  // QPW > 0: what the head phase at the END of the kernel needs from memory is requested HERE, ahead of the weight staging and the
  // MFMA loop, so that its latency (batch row -> activation row of the actor's last hidden layer, two dependent loads) is not the
  // kernel's tail: lane (quad qg, row group rg) of wave w takes the column quad w QPW + qg of tile rows rg RPL .. rg RPL + RPL - 1
  constexpr int HQ = QPW > 0 ? QPW : 16, HRPL = 32 / (64 / HQ);
  float4 hxv[QPW > 0 ? HRPL : 1];
  int hrow[QPW > 0 ? HRPL : 1];
  float haux[8];      // wave 0: tanh outputs of this lane's (row, column) slots
  int horow = -1, ht0 = -1;
  if (QPW > 0) {
    const int hq = wave * HQ + lane % HQ, hrg = lane / HQ;
#pragma unroll
    for (int j = 0; j < HRPL; ++j) {
      const int rr = m0 + hrg * HRPL + j;
      int m = p.perm[rr];
      if (x.tie0[rr] == -2) m = -1;   // the run-0 copy of a tie sample: its run-1 tile writes the row
      hrow[j] = m;
    }
#pragma unroll
    for (int j = 0; j < HRPL; ++j)
      hxv[j] = hrow[j] >= 0 ? *reinterpret_cast<const float4*>(x.Hh + (long long)hrow[j] * x.ldh + 4 * hq) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave == 0) {
      horow = p.perm[m0 + r];
      ht0 = x.tie0[m0 + r];
      if (ht0 == -2) horow = -1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 8 * (e >> 2) + 4 * h + (e & 3);
        haux[e] = (horow >= 0 && c < p.ncol) ? p.aux[(long long)horow * p.ldaux + c] : 0.f;
      }
    }
  }
  PQLK_SLICE_PHASE_END(1, hxv[0].x + haux[0] + (float)hrow[0]);
  const int ktot = p.groups * p.K;          // p.K = hidden width (multiple of 32)
  const int kq = ktot >> 2;                 // reduction elements of this wave; multiple of 8
  const int kbeg = wave * kq;
  const int ldw = kq + 4;                   // LDS row stride: conflict-free b128 reads
  float* wt = dxs_lds + wave * 32 * ldw;    // wt[c][k - kbeg]
  {  // stage W[k][col0 .. col0 + 32) transposed; columns past ncol are zero
    const bool vec = (p.col0 & 3) == 0 && (p.ldb & 3) == 0;
    if (vec) {
      // kq * 8 float4 (8 per reduction row) over 64 lanes, 8 loads in flight per lane: a load -> LDS-store loop with one
      // load in flight costs a full L2 latency per iteration (this staging alone was ~20 us that way)
      for (int i0 = lane; i0 < kq * 8; i0 += 64 * 8) {
        float4 v[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int i = i0 + 64 * b;
          const int kk = i >> 3, c4 = i & 7;
          const int kg = kbeg + min(kk, kq - 1), g = kg / p.K, k = kg - g * p.K;
          v[b] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i < kq * 8 && 4 * c4 < p.ncol)
            v[b] = *reinterpret_cast<const float4*>(Bw + (long long)g * p.sB + (long long)k * p.ldb + p.col0 + 4 * c4);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int i = i0 + 64 * b;
          if (i < kq * 8) {
            const int kk = i >> 3, c4 = i & 7;
            const float e[4] = {v[b].x, v[b].y, v[b].z, v[b].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) wt[(4 * c4 + u) * ldw + kk] = (4 * c4 + u < p.ncol) ? e[u] : 0.f;
          }
        }
      }
    } else {
      for (int i = lane; i < kq * 32; i += 64) {
        const int kk = i >> 5, c = i & 31;
        const int kg = kbeg + kk, g = kg / p.K, k = kg - g * p.K;
        wt[c * ldw + kk] = c < p.ncol ? Bw[(long long)g * p.sB + (long long)k * p.ldb + p.col0 + c] : 0.f;
      }
    }
  }
  PQLK_SLICE_PHASE_END(2, wt[lane] + hxv[0].x + haux[0]);
  // the wave reads only what it staged itself: no block barrier needed before the loop (LDS ops of a wave are in order)
  const int row = min(m0 + r, p.M - 1);
  const int g0 = kbeg / p.K, koff = kbeg - g0 * p.K;   // a wave's quarter lies inside one net when n_nets divides 4
  const float4* yp = reinterpret_cast<const float4*>(p.A + (long long)g0 * p.sA + (long long)row * p.lda + koff) + h;
  acc_t acc4[4];   // by reduction-step parity: four independent MFMA chains (see k_fwd_narrow)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc4[a][e] = 0.f;
  const int K8 = kq >> 3;
  float4 yq[D];   // one wave per SIMD: latency-bound, so a deep ring (see k_fwd_narrow)
#pragma unroll
  for (int s = 0; s < D; ++s) yq[s] = yp[2 * min(s, K8 - 1)];
  __builtin_amdgcn_sched_barrier(0);
  for (int k8 = 0; k8 < K8; k8 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const float4 w4 = *reinterpret_cast<const float4*>(&wt[r * ldw + 8 * (k8 + s) + 4 * h]);
      acc_t& acc = acc4[s & 3];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, yq[s].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, yq[s].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, yq[s].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, yq[s].w, acc, 0, 0, 0);
      yq[s] = yp[2 * min(k8 + s + D, K8 - 1)];
      __builtin_amdgcn_sched_barrier(0);   // keep the refill here (see k_fwd_narrow)
    }
  }
  PQLK_SLICE_PHASE_END(3, acc4[0][0] + acc4[1][1] + acc4[2][2] + acc4[3][3] + hxv[0].x + haux[0]);
  // partial tiles -> LDS (reusing the staging area after everyone is done with it), summed in wave order by wave 0
  __syncthreads();
  float* red = dxs_lds;   // [wave][reg][lane]
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = (acc4[0][e] + acc4[1][e]) + (acc4[2][e] + acc4[3][e]);
  __syncthreads();
  if (QPW == 0) {
    if (wave != 0) return;
    int orow = m0 + r;
    if (orow >= p.M) return;
    if (p.perm) {
      orow = p.perm[orow];
      if (orow < 0) return;   // pad row of the compact layout
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float s = ((red[e * 64 + lane] + red[(16 + e) * 64 + lane]) + red[(32 + e) * 64 + lane]) + red[(48 + e) * 64 + lane];
      const int c = 8 * (e >> 2) + 4 * h + (e & 3);
      if (c < p.ncol) {
        const float a = p.aux[(long long)orow * p.ldaux + c];
        if (p.perm) atomicAdd(&p.C[(long long)orow * p.ldc + c], s * (1.f - a * a));
        else p.C[(long long)orow * p.ldc + c] = s * (1.f - a * a);
      }
    }
    return;
  }
  // ---- QPW > 0: compact rows only (host), ncol <= 16.  dz tile -> LDS, then the actor head's backward
  float* dzt = dxs_lds + 4096;                                  // [32][20] behind the 16-KB reduction buffer
  if (wave == 0) {
    const int orow = horow, t0 = ht0;     // (tile rows < rows_cap: host M = rows_cap, a multiple of 32)
    float sv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sv[e] = ((red[e * 64 + lane] + red[(16 + e) * 64 + lane]) + red[(32 + e) * 64 + lane]) + red[(48 + e) * 64 + lane];
    if (orow >= 0 && t0 >= 0) {   // tie: + net 0's contribution, from the sample's run-0 row of dZ (cnet == 1 here)
      float s0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* ar = p.A + (long long)t0 * p.lda;
      for (int k = 0; k < p.K; ++k) {
        const float av = ar[k];
        const float* wr = p.B + (long long)k * p.ldb + p.col0;   // net 0's weights
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = 8 * (e >> 2) + 4 * h + (e & 3);
          if (c < p.ncol) s0[e] += av * wr[c];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sv[e] += s0[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 8 * (e >> 2) + 4 * h + (e & 3);
      float dz = 0.f;
      if (orow >= 0 && c < p.ncol) {
        dz = sv[e] * (1.f - haux[e] * haux[e]);
        p.C[(long long)orow * p.ldc + c] = dz;
      }
      dzt[r * 20 + c] = dz;
    }
  }
  __syncthreads();
  PQLK_SLICE_PHASE_END(4, dzt[tid & 511] + hxv[0].x);
  if (QPW > 0) {
    constexpr int Q = HQ, RPL = HRPL;
    const int qg = lane % Q, rg = lane / Q;
    const int q = wave * Q + qg;                 // this lane's column quad of the K_h = 16 Q inputs
    float* out = x.part + (long long)blockIdx.x * x.part_floats;
    // the N <= 16 outputs in two passes of 8 (weights + dW accumulators of 8 outputs: 64 registers instead of 128, two blocks per CU)
    float4 asum[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) asum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma nounroll   // (unrolled, both passes' weight loads are hoisted to the top: 316 VGPRs)
    for (int n0 = 0; n0 < 16; n0 += 8) {
      if (n0 < x.N) {   // block-uniform
        float4 wq[8], acc[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          wq[n] = n0 + n < x.N ? *reinterpret_cast<const float4*>(x.Wh + (long long)(n0 + n) * x.Kh + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
          acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          const float* dr = dzt + (rg * RPL + j) * 20 + n0;
          const float4 d0 = *reinterpret_cast<const float4*>(dr), d1 = *reinterpret_cast<const float4*>(dr + 4);
          const float dn[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};   // (zero beyond N and for rows with nothing to do)
          const float4 xx = hxv[j];
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            asum[j].x += dn[n] * wq[n].x; asum[j].y += dn[n] * wq[n].y; asum[j].z += dn[n] * wq[n].z; asum[j].w += dn[n] * wq[n].w;
            acc[n].x += dn[n] * xx.x; acc[n].y += dn[n] * xx.y; acc[n].z += dn[n] * xx.z; acc[n].w += dn[n] * xx.w;
          }
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          if (n0 + n < x.N) {
            float4 v = acc[n];
#pragma unroll
            for (int o = Q; o < 64; o <<= 1) {   // fold the row groups: fixed xor tree
              v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
            }
            if (rg == 0) *reinterpret_cast<float4*>(out + (long long)(n0 + n) * x.Kh + 4 * q) = v;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      float4 a = asum[j];
      const float4 xx = hxv[j];
      a.x = xx.x > 0.f ? a.x : a.x * (xx.x + 1.f);   // ELU'(h) = h + 1 for h <= 0
      a.y = xx.y > 0.f ? a.y : a.y * (xx.y + 1.f);
      a.z = xx.z > 0.f ? a.z : a.z * (xx.z + 1.f);
      a.w = xx.w > 0.f ? a.w : a.w * (xx.w + 1.f);
      if (hrow[j] >= 0) *reinterpret_cast<float4*>(x.dXh + (long long)hrow[j] * x.Kh + 4 * q) = a;
    }
    if (wave == 0 && lane < 32) {   // db (rows ascending) + the zero pad of the bias block
      float db = 0.f;
      if (lane < x.N)
        for (int rr = 0; rr < 32; ++rr) db += dzt[rr * 20 + lane];
      out[(long long)x.N * x.Kh + lane] = db;
    }
  }
}

static size_t dx_slice_lds(const GemmP& p) {
  const size_t stage = (size_t)4 * 32 * (p.groups * p.K / 4 + 4) * sizeof(float);
  return stage > 16384 ? stage : 16384;   // the staging area doubles as the 4 x 16 x 64-float reduction buffer
}

static bool dx_slice_ok(const GemmP& p) {
  return p.epi == EPI_DTANH_SLICE && p.zsum && p.ncol <= 32 && (p.K & 31) == 0 && (4 % p.groups) == 0 && (p.lda & 3) == 0 &&
         dx_slice_lds(p) <= 160 * 1024;
}

// the actor head rides along when its input width is 16 QPW floats with QPW in {8, 16} (128 / 256: both BASELINE hidden shapes)
static bool dx_slice_head_ok(int n_out, int k_h) { return n_out >= 1 && n_out <= 16 && (k_h == 128 || k_h == 256); }

static int launch_dx_slice(const GemmP& p, hipStream_t st, const SliceHeadX* head = nullptr) {
  size_t shmem = dx_slice_lds(p);
  if (head && shmem < 16384 + 4096) shmem = 16384 + 4096;   // reduction buffer + dz tile + row table
  static PqlkPerDeviceOnce attr_once;
  if (int rc = attr_once.run([&] {
        for (const void* k : {reinterpret_cast<const void*>(&k_dx_slice<16>), reinterpret_cast<const void*>(&k_dx_slice<4>),
                              reinterpret_cast<const void*>(&k_dx_slice<1>), reinterpret_cast<const void*>(&k_dx_slice<16, 16>),
                              reinterpret_cast<const void*>(&k_dx_slice<16, 8>), reinterpret_cast<const void*>(&k_dx_slice<4, 16>),
                              reinterpret_cast<const void*>(&k_dx_slice<4, 8>)}) {
          hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          if (e != hipSuccess) return -(int)e;
        }
        return 0;
      }))
    return rc;
  const int K8 = p.groups * p.K / 32;   // reduction steps of 8 per wave
  const dim3 grid((unsigned)((p.M + 31) / 32)), block(256);
  if (head) {   // (host: compact mode, ncol <= 16, K8 % 4 == 0 -- hidden widths are multiples of 128 there)
    if (K8 % 4 != 0 || !p.perm || p.ncol > 16 || !dx_slice_head_ok(head->N, head->Kh)) return PQLK_E_UNSUPPORTED;
    const bool wide = head->Kh == 256;
    if (K8 % 16 == 0) {
      if (wide) hipLaunchKernelGGL((k_dx_slice<16, 16>), grid, block, shmem, st, p, *head);
      else hipLaunchKernelGGL((k_dx_slice<16, 8>), grid, block, shmem, st, p, *head);
    } else {
      if (wide) hipLaunchKernelGGL((k_dx_slice<4, 16>), grid, block, shmem, st, p, *head);
      else hipLaunchKernelGGL((k_dx_slice<4, 8>), grid, block, shmem, st, p, *head);
    }
    PQLK_LAUNCH_CHECK();
    return PQLK_OK;
  }
  const SliceHeadX none = {};
  if (K8 % 16 == 0) hipLaunchKernelGGL(k_dx_slice<16>, grid, block, shmem, st, p, none);
  else if (K8 % 4 == 0) hipLaunchKernelGGL(k_dx_slice<4>, grid, block, shmem, st, p, none);
  else hipLaunchKernelGGL(k_dx_slice<1>, grid, block, shmem, st, p, none);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}
