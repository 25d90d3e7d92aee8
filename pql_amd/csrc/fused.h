// Fused hidden-layer forward of an MLP: Linear -> ELU -> Linear -> ELU ... for one 32-row batch tile per block, all
// hidden layers in ONE launch (reference: the nn.Sequential of pql/models/mlp.py:15-24 evaluated layer by layer).
//
// Why: at batch 8192 every per-layer GEMM launch pays a fixed ~15 us (launch gap, first-tile latency, and the 33 MB
// activation write that all blocks drain at once before the next launch can read it back) on top of 15-60 us of MFMA
// time.  Here the activations of a 32-row tile never leave the CU: they ping-pong between two LDS buffers
// ((32 x (width+4)) fp32 each, 132 KB for 512-wide layers), the only HBM traffic is the optional stash write (needed
// by backward) and it is asynchronous.  Weights are streamed straight from L2 into MFMA operand registers: with one
// 32-row tile per block a weight element is used exactly once per block, so an LDS stage would buy nothing; instead the
// weights are kept in a second, fragment-ordered copy (`pqlk_mlp_pack`) in which the 64 lanes of a wave read one
// contiguous KiB per instruction:
//     packed[layer][tile t = n/32][k8 = k/8][lane = (r, h)][j]  =  W[32 t + r][8 k8 + 4 h + j]
// i.e. exactly the B-fragment quad of the k-ordering used by k_gemm (kk = 8 k8 + 4 h + j), so fused and unfused
// paths accumulate every output element in the same order.
// One wave per SIMD (4 waves / block, one block per CU at 512-wide layers), each wave owning TPW 32-column tiles and a
// 4-deep register ring of weight quads (3 k8-groups = ~3k cycles of MFMA work in flight ahead of use).
#pragma once
#include "pqlk_common.h"

typedef float f32x16f __attribute__((ext_vector_type(16)));

struct FusedP {
  const float* X;       // (B, ldx) input shared by all nets
  const float* params;  // parameter arena (biases are read from here)
  const float* packed;  // fragment-ordered hidden-layer weights
  float* acts;          // activation stash (layout of pqlk_mlp_act_offset)
  int B, ldx, n_hidden, stash_all, buf_ld, n_nets;
  int dims[PQLK_MAX_LAYERS + 1];  // in (logical), h1, h2, ...
  long long net_stride, packed_net_stride;
  long long b_off[PQLK_MAX_LAYERS], p_off[PQLK_MAX_LAYERS], a_off[PQLK_MAX_LAYERS];
};

__device__ __forceinline__ float fused_elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }

// in_off / out_off: float offsets of the two activation buffers inside the dynamic LDS array.  They are passed as
// OFFSETS, not pointers: a runtime-selected pointer loses its address space, the reads become flat_load and every
// wait degrades to `vmcnt(0) lgkmcnt(0)`, which drains the weight prefetch ring on each k-step.
extern __shared__ __attribute__((aligned(16))) float fsm[];

template <int TPW, int NW>
__device__ __forceinline__ void fused_layer(int in_off, int out_off, int buf_ld, int K,
                                            int N, const float* __restrict__ packed_l, const float* __restrict__ bias_l,
                                            float* __restrict__ gout, int g_ld, int row0, int B, int wave, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int K8 = K >> 3;
  const int ntiles = N >> 5;
  for (int tbase = wave * TPW; tbase < ntiles; tbase += NW * TPW) {
    f32x16f acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const float4* wp[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) wp[j] = reinterpret_cast<const float4*>(packed_l) + (long long)(tbase + j) * K8 * 64 + lane;
    float4 bq[4][TPW];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < TPW; ++j) bq[s][j] = wp[j][s * 64];   // K8 >= 4
    __builtin_amdgcn_sched_barrier(0);
    for (int k8 = 0; k8 < K8; k8 += 4) {  // K is a multiple of 32 -> K8 a multiple of 4
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(&fsm[in_off + r * buf_ld + 8 * (k8 + s) + 4 * h]);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const float bv[4] = {bq[s][j].x, bq[s][j].y, bq[s][j].z, bq[s][j].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t], av[t], acc[j], 0, 0, 0);
        }
        // unconditional refill (index clamped at the tail): straight-line code lets the compiler keep the other
        // three ring stages in flight behind a counted s_waitcnt vmcnt(N); a branch here degrades every wait to vmcnt(0)
        const int kn = min(k8 + s + 4, K8 - 1);
#pragma unroll
        for (int j = 0; j < TPW; ++j) bq[s][j] = wp[j][kn * 64];
        __builtin_amdgcn_sched_barrier(0);  // keep the refill HERE: hipcc otherwise sinks the loads to just before use
      }
    }
    // epilogue: lane (r, h) owns row r, columns 32*tile + 8q + 4h + {0..3} (transposed-tile accumulator layout)
    const bool row_ok = row0 + r < B;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * (tbase + j) + 8 * q + 4 * h;
        const float4 b4 = *reinterpret_cast<const float4*>(bias_l + col);
        float4 v;
        v.x = fused_elu(acc[j][4 * q] + b4.x);
        v.y = fused_elu(acc[j][4 * q + 1] + b4.y);
        v.z = fused_elu(acc[j][4 * q + 2] + b4.z);
        v.w = fused_elu(acc[j][4 * q + 3] + b4.w);
        *reinterpret_cast<float4*>(&fsm[out_off + r * buf_ld + col]) = v;
        if (gout && row_ok) *reinterpret_cast<float4*>(gout + (long long)(row0 + r) * g_ld + col) = v;
      }
    }
  }
}

#ifndef PQLK_FUSED_WAVES
#define PQLK_FUSED_WAVES 8   // waves per block: 8 = two per SIMD, so one wave's LDS / L2 waits hide behind the other's MFMAs
#endif
__global__ __launch_bounds__(64 * PQLK_FUSED_WAVES) void k_mlp_fwd_fused(FusedP p) {
  constexpr int NW = PQLK_FUSED_WAVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware block -> (net, row tile) map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share
  // an L2), and a twin critic's fragment-ordered weights are 2 x 1.8 MB against a 4 MB L2: with both nets on every
  // XCD the weight stream thrashes L2 and falls back to the Infinity Cache.  Even XCD groups take net 0, odd ones
  // net 1, so each L2 keeps ONE net's weights resident.  (Speed only: any placement computes the same result.)
  int net, tile;
  const int tiles = (p.B + 31) >> 5;
  if (p.n_nets == 2 && (tiles & 3) == 0) {
    const int b = blockIdx.x, g = b & 7, i = b >> 3;
    net = g & 1;
    tile = i * 4 + (g >> 1);
  } else {
    net = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  const int row0 = tile * 32;
  const int boff[2] = {0, 32 * p.buf_ld};
  {  // stage the input tile (pad columns of X are zero by contract; rows past B are zero-filled)
    const int k0 = (p.dims[0] + 31) & ~31;
    const int cpr = k0 >> 2;
    for (int i = tid; i < 32 * cpr; i += 64 * NW) {
      const int row = i / cpr, c4 = i % cpr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + row < p.B) v = *reinterpret_cast<const float4*>(p.X + (long long)(row0 + row) * p.ldx + 4 * c4);
      // columns past the logical input width are forced to zero here, so X may be a wider matrix whose extra columns
      // hold something else (the target actor reads its observations straight out of the critic's [obs | action] tile)
      const int c = 4 * c4, w = p.dims[0];
      if (c + 3 >= w) {
        if (c >= w) v.x = 0.f;
        if (c + 1 >= w) v.y = 0.f;
        if (c + 2 >= w) v.z = 0.f;
        v.w = 0.f;
      }
      *reinterpret_cast<float4*>(&fsm[row * p.buf_ld + 4 * c4]) = v;
    }
  }
  __syncthreads();
  for (int l = 0; l < p.n_hidden; ++l) {
    const int K = (p.dims[l] + 31) & ~31, N = p.dims[l + 1];
    const float* packed_l = p.packed + (long long)net * p.packed_net_stride + p.p_off[l];
    const float* bias_l = p.params + (long long)net * p.net_stride + p.b_off[l];
    float* gout = (p.stash_all || l == p.n_hidden - 1) ? p.acts + p.a_off[l] + (long long)net * p.B * N : nullptr;
    const int ntiles = N >> 5;
    const int in = boff[l & 1], out = boff[(l & 1) ^ 1];
    if (ntiles % (4 * NW) == 0) fused_layer<4, NW>(in, out, p.buf_ld, K, N, packed_l, bias_l, gout, N, row0, p.B, wave, lane);
    else if (ntiles % (2 * NW) == 0) fused_layer<2, NW>(in, out, p.buf_ld, K, N, packed_l, bias_l, gout, N, row0, p.B, wave, lane);
    else fused_layer<1, NW>(in, out, p.buf_ld, K, N, packed_l, bias_l, gout, N, row0, p.B, wave, lane);
    __syncthreads();
  }
}

// arena -> fragment-ordered copy of the hidden layers' weights (one thread per element; 1-3 M elements)
__global__ __launch_bounds__(256) void k_mlp_pack(const float* __restrict__ params, float* __restrict__ packed, long long w_off,
                                                  long long p_off, int N, int K, int ldk, long long net_stride,
                                                  long long packed_net_stride) {
  const int net = blockIdx.y;
  const float* W = params + (long long)net * net_stride + w_off;
  float* dst = packed + (long long)net * packed_net_stride + p_off;
  const long long total = (long long)N * K;
  const int K8 = K >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long long g = i >> 8;  // tile * K8 + k8
    const int k8 = (int)(g % K8), tile = (int)(g / K8);
    const int n = 32 * tile + (lane & 31), k = 8 * k8 + 4 * (lane >> 5) + j;
    dst[i] = W[(long long)n * ldk + k];
  }
}
