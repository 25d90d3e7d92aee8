// "Min-net" compaction of the DPG backward through the frozen twin critic.
//
// Reference: pql/algo/pql_p_learner.py:55-58 -- actor_loss = -critic.get_q_min(obs, actor(obs)).mean(), i.e.
// min(Q1, Q2) per sample (pql/models/mlp.py:201-203).  Its gradient reaches only the net that attained the minimum
// (both, halved, on an exact tie), so in the dense backward half of the (net, sample) rows of every dZ are exactly zero and
// the dX GEMMs multiply them through three layers.  Here the samples are PARTITIONED by the net that owns them first:
//   compact rows [0, c0)            samples with Q1 <= Q2 (ties included), in batch order          -> net 0's weights
//   compact rows [base1, base1+c1)  samples with Q2 <= Q1 (ties included), base1 = c0 rounded up to 128 -> net 1's weights
// so every 128-row GEMM tile belongs to ONE net and the chain runs over ~B (+ ties + padding) rows instead of 2 B: half
// the MFMA work of the critic backward.  Rows are computed independently of each other, so each sample's dZ is bit for bit
// what the dense chain computes for it; the final scatter adds one or (tie) two contributions per action gradient.
// Used when the critic has scalar Q heads (the distributional head's minimum is over expectations: dense path).
#pragma once
#include "pqlk_common.h"

#define MN_TILE 128

// One block: stable partition of the batch by the owning net.  owner[m] (bit 0: net 0, bit 1: net 1; from
// pqlk_dpg_loss_owner) when given -- B contiguous bytes -- else derived from q = (2, B, ldq) head outputs, column 0 (a
// 128-B-strided read per sample: 40 us for 8192 samples from one block, against 4 us with the byte array).
// mn = {c0, c1, base1, rows in use}.
__device__ __forceinline__ int minnet_owner(const uint8_t* __restrict__ owner, const float* __restrict__ q, int64_t ldq, int64_t b,
                                            int64_t m) {
  if (owner) return owner[m];
  const float a = q[m * ldq], c = q[(b + m) * ldq];
  return (a <= c ? 1 : 0) | (c <= a ? 2 : 0);
}

__global__ __launch_bounds__(1024) void k_minnet_partition(const uint8_t* __restrict__ owner, const float* __restrict__ q, int64_t ldq,
                                                           int64_t b, int* __restrict__ perm, int64_t perm_len, int* __restrict__ mn) {
  // thread t owns the S consecutive samples [t S, (t+1) S): local counts -> wave scan -> wave totals through LDS -> ordered
  // positions.  Three barriers in all (a chunk-by-chunk ballot scan needed three PER 1024 samples: 13 us at 8192).
  __shared__ int w0[16], w1[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // S = samples per thread, a multiple of 8 so that a thread's owner bytes are whole 8-byte words, all loaded up front (one
  // byte load at a time, twice over, was 16 dependent-latency round trips: 11 of the kernel's 13 us)
  constexpr int MAXW = 16;   // up to 128 samples per thread: B <= 131072
  const int64_t S = ((b + 1023) / 1024 + 7) / 8 * 8;
  const int64_t m_lo = (int64_t)threadIdx.x * S, m_hi = m_lo + S < b ? m_lo + S : b;
  const int nw = (int)(S / 8);
  unsigned long long wv[MAXW];
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    wv[w] = 0ull;
    if (w < nw && m_lo + 8 * w < b) {
      if (owner && m_lo + 8 * w + 8 <= b) wv[w] = *reinterpret_cast<const unsigned long long*>(owner + m_lo + 8 * w);
      else
        for (int j = 0; j < 8; ++j)
          if (m_lo + 8 * w + j < b) wv[w] |= (unsigned long long)minnet_owner(owner, q, ldq, b, m_lo + 8 * w + j) << (8 * j);
    }
  }
  int n0 = 0, n1 = 0;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    n0 += __popcll(wv[w] & 0x0101010101010101ull);
    n1 += __popcll(wv[w] & 0x0202020202020202ull);
  }
  int i0 = n0, i1 = n1;   // inclusive scan over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t0 = __shfl_up(i0, o, 64), t1 = __shfl_up(i1, o, 64);
    if (lane >= o) { i0 += t0; i1 += t1; }
  }
  if (lane == 63) { w0[wave] = i0; w1[wave] = i1; }
  __syncthreads();
  int e0 = i0 - n0, e1 = i1 - n1, c0 = 0, c1 = 0;   // exclusive prefix of this thread; totals
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (w < wave) { e0 += w0[w]; e1 += w1[w]; }
    c0 += w0[w]; c1 += w1[w];
  }
  const int base1 = (c0 + MN_TILE - 1) / MN_TILE * MN_TILE;
  const int used = base1 + (c1 + MN_TILE - 1) / MN_TILE * MN_TILE;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < nw) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int o = (int)(wv[w] >> (8 * j)) & 3;
        const int64_t m = m_lo + 8 * w + j;
        if (m < m_hi) {
          if (o & 1) perm[e0++] = (int)m;
          if (o & 2) perm[base1 + e1++] = (int)m;
        }
      }
    }
  }
  for (int64_t i = c0 + threadIdx.x; i < base1; i += 1024) perm[i] = -1;                              // pad rows of the two runs
  for (int64_t i = base1 + c1 + threadIdx.x; i < used && i < perm_len; i += 1024) perm[i] = -1;
  if (threadIdx.x == 0) { mn[0] = c0; mn[1] = c1; mn[2] = base1; mn[3] = used; }
}

// Head backward (dX only, <= 16 outputs) straight into compact rows:
//   dZ[i, :] = (sum_n dY[net, m, n] W_net[n, :]) * ELU'(H[net, m, :]),  m = perm[i], net = run of row i;  pad rows -> 0.
// One wave per compact row, both nets' head weights in LDS.
struct MinnetHeadP {
  const float* H; long long sH; int ldh;        // (2, B, ldh) last hidden activations
  const float* W; long long sW; int ldk;        // (N, ldk) per net
  const float* dY; long long sY; int ldy;       // (2, B, ldy)
  float* C;                                     // (rows, ldk) compact output
  const int* perm; const int* mn;
  int N, K; long long rows_cap;
  float* zero_out; long long zero_floats;       // the matrix the slice kernel adds into: zeroed here, by the whole grid
};

__global__ __launch_bounds__(256) void k_minnet_head_dx(MinnetHeadP p) {
  extern __shared__ __attribute__((aligned(16))) float mn_w[];   // (2, N, K)
  const int kq = p.K >> 2;
  for (int i = threadIdx.x; i < 2 * p.N * kq; i += 256) {
    const int g = i / (p.N * kq), rem = i - g * p.N * kq, n = rem / kq, q = rem % kq;
    reinterpret_cast<float4*>(mn_w)[i] = *reinterpret_cast<const float4*>(p.W + (long long)g * p.sW + (long long)n * p.ldk + 4 * q);
  }
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (p.zero_floats >> 2); i += (long long)gridDim.x * 256)
    reinterpret_cast<float4*>(p.zero_out)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int lane = threadIdx.x & 63;
  const int used = p.mn[3], base1 = p.mn[2];
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < used; i += nwaves) {
    const int m = p.perm[i];
    const int g = i >= base1 ? 1 : 0;
    float dyl = 0.f;
    if (m >= 0 && lane < p.N) dyl = p.dY[(long long)g * p.sY + (long long)m * p.ldy + lane];
    float dn[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) dn[n] = __shfl(dyl, n, 64);
    for (int q = lane; q < kq; q += 64) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m >= 0) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          if (n < p.N) {
            const float4 w = reinterpret_cast<const float4*>(mn_w)[(g * p.N + n) * kq + q];
            a.x += dn[n] * w.x; a.y += dn[n] * w.y; a.z += dn[n] * w.z; a.w += dn[n] * w.w;
          }
        }
        const float4 hv = *reinterpret_cast<const float4*>(p.H + (long long)g * p.sH + (long long)m * p.ldh + 4 * q);
        a.x = hv.x > 0.f ? a.x : a.x * (hv.x + 1.f);
        a.y = hv.y > 0.f ? a.y : a.y * (hv.y + 1.f);
        a.z = hv.z > 0.f ? a.z : a.z * (hv.z + 1.f);
        a.w = hv.w > 0.f ? a.w : a.w * (hv.w + 1.f);
      }
      *reinterpret_cast<float4*>(p.C + i * p.ldk + 4 * q) = a;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 4: DPG loss + partition + compact head in ONE launch (k_dpg_scalar 4.8 us + k_minnet_partition 8.8 + k_minnet_head_dx 9.3
// at batch 8192 before).  The twin critic's fused forward leaves its head outputs COMPACT as well, qc = (2, B) floats (one 64-KB
// coalesced read instead of 2 B lines 128 B apart), so every block can afford to redo the whole partition scan itself -- 1024
// threads x S consecutive samples each, the scan of k_minnet_partition -- and then owns the compact rows [blockIdx R, (blockIdx+1) R):
// it keeps only the perm entries that land there (LDS), writes them out, and runs the head's backward for them
//   dZ[i, :] = g_i w_net * ELU'(H[net, perm[i], :]),   g_i = -1/B (-0.5/B on an exact tie: both nets own the sample)
// i.e. the values k_dpg_scalar wrote into dY and k_minnet_head_dx read back.  Block 0 also leaves the loss partials (sums of
// min(Q1, Q2) over 32 runs of consecutive samples; folded with scale -1/B by the optimiser launch) and mn.
// tie0[i]: -1 = ordinary row; for a tie sample its run-1 row holds the index of its run-0 row (>= 0) and its run-0 row holds -2:
// k_dx_slice_head forms a tie sample's action gradient in ONE place (the run-1 tile) and skips the other.
#define DPG_LOSS_PARTS 32

struct DpgHeadP {
  const float* qc; long long B;
  const float* H; long long sH; int ldh;     // (2, B, ldh) last hidden activations of the critic
  const float* W; long long sW;              // head weight row (K floats) of net g at W + g sW
  float* C; int K;                           // (rows, K) compact dZ of the last hidden layer
  int* perm; int* tie0; long long perm_len; int* mn;
  float* loss_part;
  float gb;                                  // -1 / B
  int rows_cap_blk;                          // LDS room: compact rows per block
};

__global__ __launch_bounds__(1024) void k_dpg_minnet_head(DpgHeadP p) {
  __shared__ int w0[16], w1[16];
  extern __shared__ __attribute__((aligned(16))) int dh_lds[];   // perm_l[R] | tie_l[R] | w (2, K) floats
  int* perm_l = dh_lds;
  int* tie_l = dh_lds + p.rows_cap_blk;
  float* w_l = reinterpret_cast<float*>(dh_lds + 2 * p.rows_cap_blk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long b = p.B;
  for (int i = tid; i < 2 * p.rows_cap_blk; i += 1024) dh_lds[i] = -1;
  for (int i = tid; i < 2 * (p.K >> 2); i += 1024) {
    const int g = i / (p.K >> 2), q = i - g * (p.K >> 2);
    reinterpret_cast<float4*>(w_l)[i] = *reinterpret_cast<const float4*>(p.W + (long long)g * p.sW + 4 * q);
  }
  constexpr int MAXW = 16;   // up to 128 samples per thread: B <= 131072
  const long long S = ((b + 1023) / 1024 + 7) / 8 * 8;
  const long long m_lo = (long long)tid * S, m_hi = m_lo + S < b ? m_lo + S : b;
  const int nw = (int)(S / 8);
  unsigned long long wv[MAXW];
  float lacc = 0.f;
  const bool vec = (b & 3) == 0;   // qc + b is 16-B aligned then
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    wv[w] = 0ull;
    const long long m0 = m_lo + 8 * w;
    if (w < nw && m0 < b) {
      float a[8], c[8];
      if (vec && m0 + 8 <= b) {
        const float4 a0 = *reinterpret_cast<const float4*>(p.qc + m0), a1 = *reinterpret_cast<const float4*>(p.qc + m0 + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(p.qc + b + m0), c1 = *reinterpret_cast<const float4*>(p.qc + b + m0 + 4);
        a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
        c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = m0 + j < b;
          a[j] = ok ? p.qc[m0 + j] : 0.f; c[j] = ok ? p.qc[b + m0 + j] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (m0 + j < b) {
          lacc += fminf(a[j], c[j]);
          wv[w] |= (unsigned long long)((a[j] <= c[j] ? 1 : 0) | (c[j] <= a[j] ? 2 : 0)) << (8 * j);
        }
    }
  }
  if (blockIdx.x == 0) {   // loss partials: DPG_LOSS_PARTS = 32 half-wave sums, samples ascending
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) lacc += __shfl_xor(lacc, o, 64);
    if ((lane & 31) == 0) p.loss_part[2 * wave + (lane >> 5)] = lacc;
  }
  int n0 = 0, n1 = 0;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    n0 += __popcll(wv[w] & 0x0101010101010101ull);
    n1 += __popcll(wv[w] & 0x0202020202020202ull);
  }
  int i0 = n0, i1 = n1;   // inclusive scan over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t0 = __shfl_up(i0, o, 64), t1 = __shfl_up(i1, o, 64);
    if (lane >= o) { i0 += t0; i1 += t1; }
  }
  if (lane == 63) { w0[wave] = i0; w1[wave] = i1; }
  __syncthreads();   // (also: the -1 fill and the head weights are in LDS)
  int e0 = i0 - n0, e1 = i1 - n1, c0 = 0, c1 = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (w < wave) { e0 += w0[w]; e1 += w1[w]; }
    c0 += w0[w]; c1 += w1[w];
  }
  const int base1 = (c0 + MN_TILE - 1) / MN_TILE * MN_TILE;
  const int used = base1 + (c1 + MN_TILE - 1) / MN_TILE * MN_TILE;
  const int R = (used + (int)gridDim.x - 1) / (int)gridDim.x;   // <= rows_cap_blk (host: rows_cap / grid, rounded up)
  const int lo = blockIdx.x * R, hi = min(lo + R, used);
  e1 += base1;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < nw) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int o = (int)(wv[w] >> (8 * j)) & 3;
        const long long m = m_lo + 8 * w + j;
        if (m < m_hi) {
          const int p0 = e0, p1 = e1, tie = o == 3 ? 1 : 0;
          if (o & 1) { if (p0 >= lo && p0 < hi) { perm_l[p0 - lo] = (int)m * 2 + tie; if (tie) tie_l[p0 - lo] = -2; } ++e0; }
          if (o & 2) { if (p1 >= lo && p1 < hi) { perm_l[p1 - lo] = (int)m * 2 + tie; if (tie) tie_l[p1 - lo] = p0; } ++e1; }
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < hi - lo; i += 1024) {
    if (lo + i < p.perm_len) {
      const int v = perm_l[i];
      p.perm[lo + i] = v < 0 ? -1 : (v >> 1);
      p.tie0[lo + i] = tie_l[i];
    }
  }
  if (blockIdx.x == 0 && tid == 0) { p.mn[0] = c0; p.mn[1] = c1; p.mn[2] = base1; p.mn[3] = used; }
  // head backward of the compact rows [lo, hi): one wave per row, four rows of a wave in flight (their activation loads are all
  // requested before the first is used: row after row, each trip paid a full memory latency -- 33 rows per block at batch 8192 were
  // three dependent trips, half of the kernel's 12.5 us)
  const int kq = p.K >> 2;
  constexpr int RIF = 4;
  for (int i0 = lo + wave; i0 < hi; i0 += 16 * RIF) {
    for (int q = lane; q < kq; q += 64) {
      int vv[RIF];
      float4 hv[RIF];
#pragma unroll
      for (int u = 0; u < RIF; ++u) {
        const int i = i0 + 16 * u;
        vv[u] = i < hi ? perm_l[i - lo] : -1;
        hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vv[u] >= 0) hv[u] = *reinterpret_cast<const float4*>(p.H + (long long)(i >= base1 ? 1 : 0) * p.sH + (long long)(vv[u] >> 1) * p.ldh + 4 * q);
      }
#pragma unroll
      for (int u = 0; u < RIF; ++u) {
        const int i = i0 + 16 * u;
        if (i >= hi) break;
        const int g = i >= base1 ? 1 : 0;
        const float dn = (vv[u] & 1) ? 0.5f * p.gb : p.gb;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vv[u] >= 0) {
          const float4 w = reinterpret_cast<const float4*>(w_l)[g * kq + q];
          a = make_float4(dn * w.x, dn * w.y, dn * w.z, dn * w.w);
          a.x = hv[u].x > 0.f ? a.x : a.x * (hv[u].x + 1.f);
          a.y = hv[u].y > 0.f ? a.y : a.y * (hv[u].y + 1.f);
          a.z = hv[u].z > 0.f ? a.z : a.z * (hv[u].z + 1.f);
          a.w = hv[u].w > 0.f ? a.w : a.w * (hv[u].w + 1.f);
        }
        *reinterpret_cast<float4*>(p.C + (long long)i * p.K + 4 * q) = a;
      }
    }
  }
}
