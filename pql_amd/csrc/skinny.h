// VALU kernels for the skinny last layer of an MLP (out features <= 16: the critic's scalar Q head, the
// actor's 16-21 -> here <=16 action head).  Reference ops: the final nn.Linear of create_simple_mlp
// (pql/models/mlp.py:15-24) and its autograd.  A 32x32 MFMA tile would waste >= 50 % (N=16) to 97 % (N=1) of its
// columns and a whole launch of prologue/epilogue; these are HBM-bound streaming passes over the (B, K) hidden
// activations instead: one wave per batch row, 16-B lane accesses, the (N, K) weight block staged in LDS.
#pragma once
#include "pqlk_common.h"

#define SKINNY_MAX_N 16
#define SKINNY_MAX_K 1024

struct SkinnyP {
  const float* X;     // (groups, M, ldx) hidden activations (input of the layer); sX = 0 when shared
  const float* W;     // (N, ldk) per group
  const float* bias;  // (ld(N)) per group
  float* C;           // fwd: (groups, M, ldc) output;  dx: (groups, M, ldk) dH
  const float* dY;    // dx/dw: (groups, M, ldy)
  const float* draw;  // fwd TANH_NOISE: (M, N) contiguous
  float* C2;          // fwd: optional second destination (group 0)
  float* dW;          // dw: slab base of W block (per group/split)
  float* dB;          // dw: slab base of bias block
  int M, N, K;        // rows, out features, padded in features (multiple of 32)
  int ldx, ldk, ldc, ldy, ldc2;
  long long sX, sW, sBias, sC, sY, sSplit;
  int epi;            // fwd: EPI_NONE / EPI_TANH / EPI_TANH_NOISE ; dx: EPI_DELU / EPI_NONE
  int splits, rows_per_split;
  float noise_std, noise_clip;
  // k_skinny_bwd<1, CH, true>: the scalar twin-Q head forms dL/dQ itself (TD target + MSE, pql_v_learner.py:104-108)
  // instead of reading it from dY: td_q / td_qt = (2, M, ldy) online / target head outputs (column 0), td_rew / td_done = (M),
  // td_part[block][group] = sum of (Q - y)^2 over the block's rows.
  const float* td_q; const float* td_qt; const float* td_rew; const float* td_done;
  float td_gamma_n, td_two_over_b;
  float* td_part;
};

enum { SK_EPI_NONE = 0, SK_EPI_TANH = 2, SK_EPI_TANH_NOISE = 3, SK_EPI_DELU = 4 };

// ---------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void k_skinny_fwd(SkinnyP p) {
  extern __shared__ __attribute__((aligned(16))) float w_lds[];  // (N, K)
  const int g = blockIdx.y;
  const float* W = p.W + (long long)g * p.sW;
  const int kq = p.K >> 2;  // float4 chunks per row
  for (int i = threadIdx.x; i < p.N * kq; i += 256) {
    const int n = i / kq, q = i % kq;
    reinterpret_cast<float4*>(w_lds)[i] = *reinterpret_cast<const float4*>(W + (long long)n * p.ldk + 4 * q);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float* X = p.X + (long long)g * p.sX;
  float* C = p.C + (long long)g * p.sC;
  const float bv = lane < p.N ? p.bias[(long long)g * p.sBias + lane] : 0.f;
  const int nwaves = gridDim.x * 4;
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < p.M; m += nwaves) {
    float4 h[SKINNY_MAX_K / 256];
#pragma unroll
    for (int c = 0; c < SKINNY_MAX_K / 256; ++c) {
      const int q = lane + 64 * c;
      h[c] = q < kq ? *reinterpret_cast<const float4*>(X + (long long)m * p.ldx + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float mine = 0.f;
    for (int n = 0; n < p.N; ++n) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < SKINNY_MAX_K / 256; ++c) {
        const int q = lane + 64 * c;
        if (q < kq) {
          const float4 w = reinterpret_cast<const float4*>(w_lds)[n * kq + q];
          s += h[c].x * w.x + h[c].y * w.y + h[c].z * w.z + h[c].w * w.w;
        }
      }
      s = wave_sum(s);
      if (lane == n) mine = s;
    }
    if (lane < p.ldc) {
      float v = 0.f;
      if (lane < p.N) {
        v = mine + bv;
        if (p.epi == SK_EPI_TANH) v = tanhf(v);
        else if (p.epi == SK_EPI_TANH_NOISE) {
          v = tanhf(v);
          float nz = p.noise_std * p.draw[(long long)m * p.N + lane];
          nz = fminf(fmaxf(nz, -p.noise_clip), p.noise_clip);
          v = fminf(fmaxf(v + nz, -1.f), 1.f);
        }
        if (p.C2 && g == 0) p.C2[(long long)m * p.ldc2 + lane] = v;
      }
      C[(long long)m * p.ldc + lane] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------- dX (+ ELU')
__global__ __launch_bounds__(256) void k_skinny_dx(SkinnyP p) {
  extern __shared__ __attribute__((aligned(16))) float w_lds[];  // (N, K)
  const int g = blockIdx.y;
  const float* W = p.W + (long long)g * p.sW;
  const int kq = p.K >> 2;
  for (int i = threadIdx.x; i < p.N * kq; i += 256) {
    const int n = i / kq, q = i % kq;
    reinterpret_cast<float4*>(w_lds)[i] = *reinterpret_cast<const float4*>(W + (long long)n * p.ldk + 4 * q);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float* X = p.X + (long long)g * p.sX;
  const float* dY = p.dY + (long long)g * p.sY;
  float* C = p.C + (long long)g * p.sC;
  const int nwaves = gridDim.x * 4;
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < p.M; m += nwaves) {
    const float dyl = lane < p.N ? dY[(long long)m * p.ldy + lane] : 0.f;
    float dn[SKINNY_MAX_N];  // dY row broadcast to every lane (all lanes active here)
#pragma unroll
    for (int n = 0; n < SKINNY_MAX_N; ++n) dn[n] = __shfl(dyl, n, 64);
#pragma unroll
    for (int c = 0; c < SKINNY_MAX_K / 256; ++c) {
      const int q = lane + 64 * c;
      if (q < kq) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < SKINNY_MAX_N; ++n) {
          if (n < p.N) {
            const float4 w = reinterpret_cast<const float4*>(w_lds)[n * kq + q];
            a.x += dn[n] * w.x; a.y += dn[n] * w.y; a.z += dn[n] * w.z; a.w += dn[n] * w.w;
          }
        }
        if (p.epi == SK_EPI_DELU) {
          const float4 hv = *reinterpret_cast<const float4*>(X + (long long)m * p.ldx + 4 * q);
          a.x = hv.x > 0.f ? a.x : a.x * (hv.x + 1.f);
          a.y = hv.y > 0.f ? a.y : a.y * (hv.y + 1.f);
          a.z = hv.z > 0.f ? a.z : a.z * (hv.z + 1.f);
          a.w = hv.w > 0.f ? a.w : a.w * (hv.w + 1.f);
        }
        *reinterpret_cast<float4*>(C + (long long)m * p.ldk + 4 * q) = a;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- dW, db (split-batch slabs)
// grid = (ceil(K / (4 CLS)), splits, groups); block = CLS column-lanes (one float4 = 4 columns each) x RLS = 256 / CLS
// row-lanes.  Each thread streams its rows with 4 independent 16-B loads in flight; the row-lanes are folded through LDS
// in fixed order.  CLS = 16 (64 columns per block) when that already gives every CU a block; CLS = 4 (16 columns per
// block, 4x the blocks) otherwise -- the 256 -> 16 action head at 16 splits is 64 blocks at CLS = 16 (47 us).
template <int CLS>
__global__ __launch_bounds__(256) void k_skinny_dw(SkinnyP p) {
  constexpr int RLS = 256 / CLS;
  constexpr int ROWS = RLS * 8 <= 256 ? RLS * 8 : 256;   // rows per chunk: 8 per thread (CLS = 16) or 4 (CLS = 4)
  constexpr int CSH = CLS == 16 ? 4 : 2;                 // log2(CLS)
  static_assert(CLS == 16 || CLS == 4, "column-lane count");
  __shared__ float dy_lds[ROWS][SKINNY_MAX_N + 1];
  __shared__ __attribute__((aligned(16))) float red[8][RLS][CLS][4];   // 32 KB
  __shared__ float db_part[16][16];
  const int g = blockIdx.z, split = blockIdx.y;
  const int cl = threadIdx.x & (CLS - 1), rl = threadIdx.x >> CSH;
  const int col = blockIdx.x * (4 * CLS) + 4 * cl;
  const float* X = p.X + (long long)g * p.sX;
  const float* dY = p.dY + (long long)g * p.sY;
  const int m_beg = split * p.rows_per_split;
  const int m_end = min(p.M, m_beg + p.rows_per_split);
  float4 acc[SKINNY_MAX_N];
#pragma unroll
  for (int n = 0; n < SKINNY_MAX_N; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  float dbacc = 0.f;   // block column 0 only: lanes tid < N each own one bias column over this block's rows
  const bool col_ok = col < p.K;
  const bool dy_vec = (p.N & 3) == 0 && (p.ldy & 3) == 0;
  for (int m0 = m_beg; m0 < m_end; m0 += ROWS) {
    // wide heads (dy_vec): the first batch of X rows is requested BEFORE dY is staged, so the two round trips overlap
    // instead of adding up (16.5 -> 11.5 us at N = 16; the scalar head N = 1 measured 0.8 us slower that way)
    float4 hv[4];
    if (dy_vec) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = rl + RLS * v;
        hv[v] = (col_ok && m0 + r < m_end) ? *reinterpret_cast<const float4*>(X + (long long)(m0 + r) * p.ldx + col)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __syncthreads();
    if (dy_vec) {   // 16-B loads, all of a thread's quads requested before the first LDS store
      const int qn = p.N >> 2;
      constexpr int QMAX = ROWS * (SKINNY_MAX_N / 4) / 256;
      float4 q[QMAX];
#pragma unroll
      for (int t = 0; t < QMAX; ++t) {
        const int i = threadIdx.x + 256 * t, r = i / qn, c4 = i - r * qn;
        q[t] = (i < ROWS * qn && m0 + r < m_end) ? *reinterpret_cast<const float4*>(dY + (long long)(m0 + r) * p.ldy + 4 * c4)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int t = 0; t < QMAX; ++t) {
        const int i = threadIdx.x + 256 * t, r = i / qn, c4 = i - r * qn;
        if (i < ROWS * qn) {
          dy_lds[r][4 * c4] = q[t].x; dy_lds[r][4 * c4 + 1] = q[t].y; dy_lds[r][4 * c4 + 2] = q[t].z; dy_lds[r][4 * c4 + 3] = q[t].w;
        }
      }
    } else {
      for (int i = threadIdx.x; i < ROWS * p.N; i += 256) {
        const int r = i / p.N, n = i % p.N;
        dy_lds[r][n] = (m0 + r < m_end) ? dY[(long long)(m0 + r) * p.ldy + n] : 0.f;
      }
    }
    __syncthreads();
    // rows rl, rl + RLS, ... of this chunk, loads issued 4 at a time
#pragma unroll
    for (int u = 0; u < ROWS / RLS; u += 4) {
      if (u > 0 || !dy_vec) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = rl + RLS * (u + v);
          hv[v] = (col_ok && m0 + r < m_end) ? *reinterpret_cast<const float4*>(X + (long long)(m0 + r) * p.ldx + col)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = rl + RLS * (u + v);
#pragma unroll
        for (int n = 0; n < SKINNY_MAX_N; ++n) {
          if (n < p.N) {
            const float d = dy_lds[r][n];
            acc[n].x += d * hv[v].x; acc[n].y += d * hv[v].y; acc[n].z += d * hv[v].z; acc[n].w += d * hv[v].w;
          }
        }
      }
    }
    if (blockIdx.x == 0) {   // bias gradient (block-uniform branch): 16 threads per output sum every 16th row, then a fixed fold
      const int bn = threadIdx.x & 15, bpart = threadIdx.x >> 4;
      float sb = 0.f;
      if (bn < p.N)
        for (int r = bpart; r < ROWS; r += 16) sb += dy_lds[r][bn];
      db_part[bpart][bn] = sb;
      __syncthreads();
      if ((int)threadIdx.x < p.N) {
#pragma unroll
        for (int k = 0; k < 16; ++k) dbacc += db_part[k][threadIdx.x];
      }
    }
  }
  float* dW = p.dW + (long long)g * p.sW + (long long)split * p.sSplit;
  // Fold the RLS row-lanes: eight outputs n at a time go to LDS, then ALL threads reduce -- P = 256 / (8 CLS) threads per
  // (n, column-lane) output, each summing a contiguous run of row-lanes, combined by a fixed shuffle tree.  (One
  // thread per column-lane walking all RLS partials for each n in turn was 1024 dependent LDS reads: 30 of the 47 us.)
  // Compile-time n everywhere: a runtime index would push acc[] to scratch memory (7x slower).
  constexpr int P = 256 / (8 * CLS);
  const int out = threadIdx.x / P, part = threadIdx.x % P;
  const int onn = out / CLS, oc = out % CLS;
  const int ocol = blockIdx.x * (4 * CLS) + 4 * oc;
#pragma unroll
  for (int half = 0; half < SKINNY_MAX_N / 8; ++half) {
    if (8 * half < p.N) {  // block-uniform
      __syncthreads();
#pragma unroll
      for (int nn = 0; nn < 8; ++nn)
        if (8 * half + nn < p.N) *reinterpret_cast<float4*>(&red[nn][rl][cl][0]) = acc[8 * half + nn];
      __syncthreads();
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (8 * half + onn < p.N) {
#pragma unroll 8
        for (int k = part * (RLS / P); k < (part + 1) * (RLS / P); ++k) {
          const float4 t = *reinterpret_cast<float4*>(&red[onn][k][oc][0]);
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
      }
#pragma unroll
      for (int o = 1; o < P; o <<= 1) {   // the P partial sums sit in adjacent lanes
        s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
      }
      if (part == 0 && 8 * half + onn < p.N && ocol < p.K) *reinterpret_cast<float4*>(dW + (long long)(8 * half + onn) * p.ldk + ocol) = s;
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < p.ldc) {  // ldc = pqlk_ld(N) <= 32: zero the bias pad
    float* dB = p.dB + (long long)g * p.sBias + (long long)split * p.sSplit;
    dB[threadIdx.x] = (int)threadIdx.x < p.N ? dbacc : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------- dX + dW + db in ONE pass
// The head's dX and dW both stream the same (M, K) activations and the same (M, N) dY: one kernel reads them once.
//   dX[m, :] = (sum_n dY[m, n] W[n, :]) * ELU'(X[m, :])        (written as it is formed)
//   dW[n, :] = sum_m dY[m, n] X[m, :],  db[n] = sum_m dY[m, n]   (per-block partial, folded later in block order)
// One wave per batch row (64 lanes x 16 B cover K = 256 floats; CH = ceil(K / 256) chunks per lane), 4 rows in flight
// per wave, SKB_ROWS rows per block so that B = 8192 x 2 nets gives 256 blocks.  Every block leaves its partial
// (N x ldk weights, then ld(N) biases: the arena layout of the layer) in `part[block][group]`; the slab-reduction kernel
// folds the blocks in index order (deterministic, no atomics).  NB = compile-time bound on N (register arrays).
#define SKB_ROWS 64   // rows per block when that still gives every CU a block; halved (down to 16) otherwise
template <int NB, int CH, bool TD = false>
__global__ __launch_bounds__(256) void k_skinny_bwd(SkinnyP p, float* __restrict__ part, long long part_floats, int rows_per_block) {
  static_assert(!TD || NB == 1, "the TD-fused head is the scalar Q head");
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];   // W (N, K) | red[3] (N, K) | dbred[4][16]
  const int g = blockIdx.y;
  const int kq = p.K >> 2;
  const int NK = p.N * p.K;
  float* w_lds = sk_lds;
  float* red = sk_lds + NK;
  float* dbred = red + 3 * NK;
  const float* W = p.W + (long long)g * p.sW;
  for (int i = threadIdx.x; i < p.N * kq; i += 256) {
    const int n = i / kq, q = i % kq;
    reinterpret_cast<float4*>(w_lds)[i] = *reinterpret_cast<const float4*>(W + (long long)n * p.ldk + 4 * q);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* X = p.X + (long long)g * p.sX;
  const float* dY = p.dY + (long long)g * p.sY;
  float* C = p.C + (long long)g * p.sC;
  float4 acc[NB][CH];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[n][c] = make_float4(0.f, 0.f, 0.f, 0.f);
  float dbacc = 0.f;   // lane n < N: sum of dY[:, n] over this wave's rows
  float tdacc = 0.f;   // TD: sum of (Q - y)^2 over this wave's rows (same value in every lane)
  const int m_blk = blockIdx.x * rows_per_block;
  // rows m_blk + wave + 4 j, j = 0 .. rows_per_block / 4 - 1, taken RIF at a time: all RIF x CH activation loads and the RIF dY
  // loads of a group are requested before the first is used
  // rows in flight per wave (all their loads requested before the first is used): 8 for the wide heads, whose per-row
  // arithmetic (N x 8 FMAs + N LDS reads) leaves few waves resident (actor head: P free-running +0.7 %); 4 for the scalar head
  constexpr int RIF = (CH == 1 && NB >= 8) ? 8 : 4;
  for (int j0 = 0; j0 < rows_per_block / 4; j0 += RIF) {
    float4 xv[RIF][CH];
    float dyl[RIF];
    float tdq[TD ? RIF : 1], tdr[TD ? RIF : 1], tdd[TD ? RIF : 1], tdt[TD ? RIF : 1];
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      const int m = m_blk + wave + 4 * (j0 + u);
      const bool ok = m < p.M && j0 + u < rows_per_block / 4;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = lane + 64 * c;
        xv[u][c] = (ok && q < kq) ? *reinterpret_cast<const float4*>(X + (long long)m * p.ldx + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if constexpr (TD) {   // wave-uniform addresses (one request each), no branch around the loads: a row past the end re-reads row 0
        const long long mc = ok ? m : 0;
        const float t1 = p.td_qt[mc * p.ldy], t2 = p.td_qt[((long long)p.M + mc) * p.ldy];
        const float rw = p.td_rew[mc], dn_ = p.td_done[mc], qv = p.td_q[((long long)g * p.M + mc) * p.ldy];
        tdq[u] = qv; tdr[u] = rw; tdd[u] = dn_; tdt[u] = fminf(t1, t2);
        dyl[u] = ok ? 1.f : 0.f;   // (finished below, once every load of the group has been requested)
      } else {
        dyl[u] = (ok && lane < p.N) ? dY[(long long)m * p.ldy + lane] : 0.f;
      }
    }
    if constexpr (TD) {
#pragma unroll
      for (int u = 0; u < RIF; ++u) {
        const float y = tdr[u] + ((1.f - tdd[u]) * p.td_gamma_n) * tdt[u];   // r + (1-d) gamma^n min Q'   (pql_v_learner.py:105)
        const float dq = dyl[u] != 0.f ? tdq[u] - y : 0.f;
        tdacc += dq * dq;
        dyl[u] = lane == 0 ? p.td_two_over_b * dq : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      const int m = m_blk + wave + 4 * (j0 + u);
      if (j0 + u >= rows_per_block / 4) break;
      float dn[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) dn[n] = __shfl(dyl[u], n, 64);
      dbacc += dyl[u];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = lane + 64 * c;
        if (q < kq) {
          float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 x = xv[u][c];
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            if (n < p.N) {
              const float4 w = reinterpret_cast<const float4*>(w_lds)[n * kq + q];
              a.x += dn[n] * w.x; a.y += dn[n] * w.y; a.z += dn[n] * w.z; a.w += dn[n] * w.w;
              acc[n][c].x += dn[n] * x.x; acc[n][c].y += dn[n] * x.y; acc[n][c].z += dn[n] * x.z; acc[n][c].w += dn[n] * x.w;
            }
          }
          if (p.epi == SK_EPI_DELU) {
            a.x = x.x > 0.f ? a.x : a.x * (x.x + 1.f);
            a.y = x.y > 0.f ? a.y : a.y * (x.y + 1.f);
            a.z = x.z > 0.f ? a.z : a.z * (x.z + 1.f);
            a.w = x.w > 0.f ? a.w : a.w * (x.w + 1.f);
          }
          if (m < p.M) *reinterpret_cast<float4*>(C + (long long)m * p.ldk + 4 * q) = a;
        }
      }
    }
  }
  // fold the four waves in wave order: waves 1..3 park their sums in LDS, wave 0 adds them 0 + 1 + 2 + 3
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (n < p.N)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          if (q < kq) reinterpret_cast<float4*>(red + (wave - 1) * NK)[n * kq + q] = acc[n][c];
        }
  }
  if (lane < 16) dbred[wave * 16 + lane] = dbacc;
  if (TD && lane == 16) dbred[64 + wave] = tdacc;
  __syncthreads();
  if (wave == 0) {
    float* out = part + ((long long)blockIdx.x * gridDim.y + g) * part_floats;
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (n < p.N)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          if (q < kq) {
            float4 s = acc[n][c];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
              const float4 t = reinterpret_cast<const float4*>(red + w * NK)[n * kq + q];
              s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            *reinterpret_cast<float4*>(out + (long long)n * p.ldk + 4 * q) = s;
          }
        }
    if (lane < p.ldc)   // ldc = pqlk_ld(N) <= 32: bias block with zero pad
      out[(long long)p.N * p.ldk + lane] = lane < p.N ? ((dbred[lane] + dbred[16 + lane]) + dbred[32 + lane]) + dbred[48 + lane] : 0.f;
    if (TD && lane == 0) p.td_part[(long long)blockIdx.x * gridDim.y + g] = ((dbred[64] + dbred[65]) + dbred[66]) + dbred[67];
  }
}

// fused backward of the head is used when the per-lane accumulators (N x CH float4) stay within 16 registers quads
static inline int skinny_bwd_ch(int k_padded) { const int c = (k_padded + 255) / 256; return c <= 2 ? c : 4; }   // template CH
static inline int skinny_bwd_nb(int n_out) { return n_out == 1 ? 1 : (n_out <= 4 ? 4 : (n_out <= 8 ? 8 : 16)); }       // template NB
static inline bool skinny_bwd_fused_ok(int n_out, int k_padded) {
  return n_out <= SKINNY_MAX_N && k_padded <= SKINNY_MAX_K && skinny_bwd_nb(n_out) * skinny_bwd_ch(k_padded) <= 16;
}
// rows per block: 64, halved while the grid has fewer than two blocks per CU (twin critic at batch 8192: 32 rows -> 512 blocks)
static inline int skinny_bwd_rows(int64_t m, int groups) {
  int r = SKB_ROWS;
  while (r > 16 && ((m + r - 1) / r) * groups < 512) r >>= 1;   // two blocks per CU (one: V step +1.4 us, P step +1.3; round 3)
  return r;
}
static inline int skinny_bwd_blocks(int64_t m, int groups) { const int r = skinny_bwd_rows(m, groups); return (int)((m + r - 1) / r); }

template <int NB, int CH, bool TD = false>
static int launch_skinny_bwd_t(const SkinnyP& p, int groups, float* part, long long part_floats, hipStream_t st) {
  const size_t sh = ((size_t)4 * p.N * p.K + 72) * sizeof(float);
  static PqlkPerDeviceOnce attr_once;
  if (int rc = attr_once.run([&] {
        return -(int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_skinny_bwd<NB, CH, TD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (4 * 4096 + 72) * (int)sizeof(float));   // N x K <= 16 x 256 floats
      }))
    return rc;
  hipLaunchKernelGGL((k_skinny_bwd<NB, CH, TD>), dim3(skinny_bwd_blocks(p.M, groups), groups), dim3(256), sh, st, p, part, part_floats,
                     skinny_bwd_rows(p.M, groups));
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

static int launch_skinny_bwd(const SkinnyP& p, int groups, float* part, long long part_floats, hipStream_t st) {
  const int ch = skinny_bwd_ch(p.K);
  if (p.td_q) {   // TD-fused scalar head (the caller checked N == 1, two groups)
    if (ch == 1) return launch_skinny_bwd_t<1, 1, true>(p, groups, part, part_floats, st);
    if (ch == 2) return launch_skinny_bwd_t<1, 2, true>(p, groups, part, part_floats, st);
    return launch_skinny_bwd_t<1, 4, true>(p, groups, part, part_floats, st);
  }
  if (p.N == 1) {
    if (ch == 1) return launch_skinny_bwd_t<1, 1>(p, groups, part, part_floats, st);
    if (ch == 2) return launch_skinny_bwd_t<1, 2>(p, groups, part, part_floats, st);
    return launch_skinny_bwd_t<1, 4>(p, groups, part, part_floats, st);
  }
  if (p.N <= 4) {
    if (ch == 1) return launch_skinny_bwd_t<4, 1>(p, groups, part, part_floats, st);
    if (ch == 2) return launch_skinny_bwd_t<4, 2>(p, groups, part, part_floats, st);
    return launch_skinny_bwd_t<4, 4>(p, groups, part, part_floats, st);
  }
  if (p.N <= 8 && ch <= 2) return ch == 1 ? launch_skinny_bwd_t<8, 1>(p, groups, part, part_floats, st)
                                          : launch_skinny_bwd_t<8, 2>(p, groups, part, part_floats, st);
  return launch_skinny_bwd_t<16, 1>(p, groups, part, part_floats, st);   // N <= 16, K <= 256 (skinny_bwd_fused_ok)
}

// forward: one wave_sum per output, so only worth it for a handful of outputs (the Q head); backward variants
// stream and stay ahead of a 64x64 MFMA tile up to 16 outputs.
static inline bool skinny_fwd_ok(int n_out, int k_padded) { return n_out <= 4 && k_padded <= SKINNY_MAX_K; }
static inline bool skinny_bwd_ok(int n_out, int k_padded) { return n_out <= SKINNY_MAX_N && k_padded <= SKINNY_MAX_K; }

static int launch_skinny_fwd(const SkinnyP& p, int groups, hipStream_t st) {
  int blocks = (p.M + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  const size_t sh = (size_t)p.N * p.K * sizeof(float);
  static PqlkPerDeviceOnce attr_once;
  if (int rc = attr_once.run([&] {
        return -(int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_skinny_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         SKINNY_MAX_N * SKINNY_MAX_K * 4);
      }))
    return rc;
  hipLaunchKernelGGL(k_skinny_fwd, dim3(blocks, groups), dim3(256), sh, st, p);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

static int launch_skinny_dx(const SkinnyP& p, int groups, hipStream_t st) {
  int blocks = (p.M + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  const size_t sh = (size_t)p.N * p.K * sizeof(float);
  static PqlkPerDeviceOnce attr_once;
  if (int rc = attr_once.run([&] {
        return -(int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_skinny_dx), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         SKINNY_MAX_N * SKINNY_MAX_K * 4);
      }))
    return rc;
  hipLaunchKernelGGL(k_skinny_dx, dim3(blocks, groups), dim3(256), sh, st, p);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

static int launch_skinny_dw(const SkinnyP& p, int groups, hipStream_t st) {
  const int wide = ((p.K + 63) / 64) * p.splits * groups;
  if (wide >= 256) hipLaunchKernelGGL(k_skinny_dw<16>, dim3((p.K + 63) / 64, p.splits, groups), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(k_skinny_dw<4>, dim3((p.K + 15) / 16, p.splits, groups), dim3(256), 0, st, p);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}
