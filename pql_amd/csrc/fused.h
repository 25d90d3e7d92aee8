// Fused hidden-layer forward of an MLP: Linear -> ELU -> Linear -> ELU ... for one tile of 32 or 64 batch rows per block,
// all hidden layers in ONE launch (reference: the nn.Sequential of pql/models/mlp.py:15-24 evaluated layer by layer).
//
// Why: at batch 8192 every per-layer GEMM launch pays a fixed ~10-15 us (launch gap, first-tile latency, and the 33 MB
// activation write that all blocks drain at once before the next launch can read it back) on top of 15-60 us of MFMA
// time.  Here the activations of a row tile never leave the CU: they live in ONE LDS buffer ((32 R) x (width + 4) fp32),
// the only HBM traffic is the optional stash write (needed by backward).  Weights are streamed straight from L2 into
// MFMA operand registers: a weight element is used once per row tile, so an LDS stage would buy nothing; instead the
// weights are kept in a second, fragment-ordered copy (`pqlk_mlp_pack`) in which the 64 lanes of a wave read one
// contiguous KiB per instruction:
//     packed[layer][tile t = n/32][k8 = k/8][lane = (r, h)][j]  =  W[32 t + r][8 k8 + 4 h + j]
// i.e. exactly the B-fragment quad of the k-ordering used by k_gemm (kk = 8 k8 + 4 h + j), so fused and unfused
// paths accumulate every output element in the same order and stay bitwise equal.
//
// Block = 8 waves (two per SIMD, so one wave's waits hide behind the other's MFMAs).  Wave w owns output tiles
// [w TPW, (w+1) TPW) of a layer (TPW = 1, 2 or 4 by layer width) for all R row tiles, and a D-deep register ring of
// weight quads per tile.  A layer's outputs stay in the MFMA accumulators until every wave has finished reading the
// layer's input (barrier A), then overwrite it (barrier B).  What the measurements behind this shape said:
//  * R = 2 (64 rows, 132 KB of LDS at 512-wide layers): every weight quad feeds two row tiles, halving the VMEM
//    instructions per MFMA.  Each global_load_dwordx4 costs MFMA issue slots wherever it is placed and whichever cache
//    serves it (loads the MFMAs do not depend on, or that always hit the per-CU L1, cost the same ~15 % at R = 1; a
//    deeper ring buys nothing), so fewer loads per MFMA is the lever: 155 -> 138 us on the twin critic of cfg #2.
//  * The next layer's first ring stages are fetched BEFORE the epilogue, and the first ring group of a layer is peeled
//    so that the previous layer's stash stores, which sit behind those loads in the in-order vmcnt queue, are counted
//    (vmcnt(stores + ring - 1)) instead of drained.  The last group of a layer does not refill.
//  * Biases are copied to LDS once; the epilogue must not wait on global loads queued behind the ring.
#pragma once
#include "pqlk_common.h"

typedef float f32x16f __attribute__((ext_vector_type(16)));

struct FusedP {
  const float* X;       // (B, ldx) input shared by all nets
  const float* X2;      // optional second source: input columns [x2_col0, dims[0]) come from X2 (B, ldx2), column c from X2[:, c - x2_col0]
  int ldx2, x2_col0;    //   (x2_col0 a multiple of 4; torch.cat((obs, action), dim=1) of mlp.py:197 without either a copy or a shared tile)
  const float* params;  // parameter arena (biases are read from here)
  const float* packed;  // fragment-ordered hidden-layer weights
  float* acts;          // activation stash (layout of pqlk_mlp_act_offset)
  int B, ldx, n_hidden, stash_all, buf_ld, n_nets;
  int dims[PQLK_MAX_LAYERS + 1];  // in (logical), h1, h2, ...
  long long net_stride, packed_net_stride;
  long long b_off[PQLK_MAX_LAYERS], p_off[PQLK_MAX_LAYERS], a_off[PQLK_MAX_LAYERS];
  // optional fused output layer (<= 32 outputs): weights read row-major from the arena, no packed copy
  int head_n, head_epi, head_ld, ld_out2;      // head_n = 0: the caller launches the last layer itself
  long long head_w_off, head_b_off, head_a_off;
  const float* draw;                           // (B, head_n) standard-normal draw for HEAD_TANH_NOISE
  float* out2;                                 // optional second destination of the output (net 0), row stride ld_out2
  float* qc;                                   // optional COMPACT copy of a scalar head's output: qc[net * B + row] (k_dpg_minnet_head's input)
  float noise_std, noise_clip;
  // optional TD head (td_dz != NULL; scalar twin-Q head, head_n == 1, two nets): the block forms the TD error of its rows from the
  // Q it has just computed and leaves the head's whole backward -- dL/dZ of the last hidden layer, the head's dW / db partial and
  // the loss partial -- while that layer's activations are still in LDS (fused_head_td below)
  const float* td_qt;        // target critic's head output, (2, B, head_ld)
  const float* td_rew;       // (B)
  const float* td_done;      // (B)
  float td_gamma_n, td_two_over_b;
  float* td_dz;              // (n_nets, B, dims[n_hidden]) dL/dZ of the last hidden layer
  float* td_head_part;       // [row tile][net][td_part_floats]: W row (dims[n_hidden] floats), then ld(1) = 32 bias floats
  long long td_part_floats;
  float* td_loss_part;       // [row tile][net] sum of (Q - y)^2 over the tile's rows
};


__device__ __forceinline__ float fused_elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }

// All LDS addressing is by OFFSET into this array (a runtime-selected pointer loses its address space: the reads become
// flat_load and every wait degrades to `vmcnt(0) lgkmcnt(0)`, which drains the weight ring on each step).
extern __shared__ __attribute__((aligned(16))) float fsm[];

constexpr int FUSED_NW = 8;   // waves per block
#ifndef PQLK_FUSED_DEEP
#define PQLK_FUSED_DEEP 4     // ring depth in reduction steps of 8 (8 and 16 measured no faster)
#endif

// one ring group: up to D reduction steps of 8, each consuming ring slot s and (REFILL) refilling it D steps ahead
template <int R, int TPW, int TM, int D, bool REFILL>
__device__ __forceinline__ void fused_group(f32x16f (&acc)[R][TPW], float4 (&bq)[D][TM], const float4* const (&wp)[TPW],
                                            const float4* __restrict__ lds4, int abase, int buf_ld4, int k8, int K8, int steps,
                                            float4 (&an)[R]) {
#pragma unroll
  for (int s = 0; s < D; ++s) {
    if (s < steps) {
      // The activation fragments run ONE reduction step ahead of the MFMAs that consume them (`an`, carried from step to step,
      // clamped at the layer's end).  Read at the top of their own step, both waves of a SIMD finish a step's 16 MFMAs at about
      // the same time and then both sit out the LDS latency with the matrix pipe idle: -3 % on every fused forward (round 3,
      // tools/kbench.py: 124.5 -> 121.2 us target critic, 134.4 -> 130.1 stashing critic, 70.7 -> 68.6 actor).
      float4 a[R];
      const int kq = min(k8 + s + 1, K8 - 1);
#pragma unroll
      for (int i = 0; i < R; ++i) { a[i] = an[i]; an[i] = lds4[abase + 32 * i * buf_ld4 + 2 * kq]; }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads HERE, ahead of this step's MFMAs (hipcc sinks them to their use)
      // unconditional, clamped refill: straight-line code lets the compiler keep the other ring stages in flight behind
      // a counted s_waitcnt vmcnt(N); a branch here degrades every wait to vmcnt(0)
      const int kn = min(k8 + s + D, K8 - 1);
      // MFMA order t-major over the R x TPW accumulators, so consecutive MFMAs are independent
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const float bv = t == 0 ? bq[s][j].x : t == 1 ? bq[s][j].y : t == 2 ? bq[s][j].z : bq[s][j].w;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const float av = t == 0 ? a[i].x : t == 1 ? a[i].y : t == 2 ? a[i].z : a[i].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[i][j], 0, 0, 0);
          }
          if (REFILL && t == 3) {
            bq[s][j] = wp[j][kn * 64];
            __builtin_amdgcn_sched_barrier(0);   // keep the refill HERE: hipcc otherwise sinks the loads to just before use
          }
        }
      }
    }
  }
}

// ring <- the first D steps of (layer, wave)'s tiles; indices clamped so that idle waves / short layers stay in bounds
template <int TM, int D>
__device__ __forceinline__ void fused_ring_fill(float4 (&bq)[D][TM], const float4* __restrict__ packed_l, int K8, int K8s, int ntiles, int tpw,
                                                int wave, int lane) {
  const float4* w[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) w[j] = packed_l + (long long)min(wave * tpw + j, ntiles - 1) * K8s * 64 + lane;
#pragma unroll
  for (int s = 0; s < D; ++s) {
    const int ks = min(s, K8 - 1) * 64;
#pragma unroll
    for (int j = 0; j < TM; ++j) bq[s][j] = w[j][ks];
  }
}

template <int R, int TPW, int TM, int D>
__device__ __forceinline__ void fused_layer(float4 (&bq)[D][TM], int buf_ld4, int K8, int K8s, int ntiles, const float4* __restrict__ packed_l,
                                            int bias_lds4, float* __restrict__ gout, int g_ld, int row0, int B, int wave, int lane,
                                            const float4* __restrict__ packed_n, int K8n, int ntiles_n, int tpw_n,
                                            float* __restrict__ gprev, int nprev4) {
  const int r = lane & 31, h = lane >> 5;
  const int t0 = wave * TPW;
  const bool active = t0 < ntiles;   // wave-uniform
  const float4* lds4 = reinterpret_cast<const float4*>(fsm);
  // (a stagger of the SIMD partners -- waves 4-7 entering each layer 256 / 512 / 1024 cycles behind waves 0-3 -- measured
  //  +0.5 / 0 / -0.3 % on the V step, inside the noise: not used)
  // Deferred stash: this layer's INPUT buffer is the previous layer's output.  Writing it to HBM from here, spread over
  // the main loop (thread tid takes float4 elements tid, tid + 512, ... of the (32 R) x nprev tile: whole 1 KiB lines per
  // wave instruction), replaces the epilogue's burst of 32-B-per-row stores that every block issues at the same moment
  // and that is store-issue bound (~10 B/clk/CU: 12-18 k cycles per 512-wide layer at R = 2).  Full tiles only.
  float4* gprev4 = reinterpret_cast<float4*>(gprev) + (long long)row0 * nprev4;
  const int srows = gprev ? 32 * R : 0;
  const int tid = wave * 64 + lane;
  int srow = tid / nprev4, sc4 = tid % nprev4;                               // element tid of the row-major tile ...
  const int sdq = (64 * FUSED_NW) / nprev4, sdm = (64 * FUSED_NW) % nprev4;   // ... and the stride of 512 elements
  f32x16f acc[R][TPW];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if (active) {
    const float4* wp[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) wp[j] = packed_l + (long long)min(t0 + j, ntiles - 1) * K8s * 64 + lane;
    const int abase = r * buf_ld4 + h;
    // K8 reduction steps of 8 = G whole ring groups + a remainder of 0..D-1 steps (ring slots 0.. hold those); K8s = the steps per
    // tile in the packed copy (the padded width).  The first layer runs ceil(in / 8) steps, not ld(in) / 8: the critic's 104 inputs
    // are 13 steps, not 16 -- the three steps of zero padding were 19 % of that layer's MFMAs (adding 0 x 0 changes no bit).
    const int K8full = K8 - (K8 % D);
    const int G = K8full / D;
    const bool rem = K8full < K8;
    float4 an[R];
#pragma unroll
    for (int i = 0; i < R; ++i) an[i] = lds4[abase + 32 * i * buf_ld4];
    if (G > 1) fused_group<R, TPW, TM, D, true>(acc, bq, wp, lds4, abase, buf_ld4, 0, K8, D, an);   // peeled (see header)
    for (int g = 1; g < G - 1; ++g) {
      if (srow < srows) {   // deferred stash of the previous layer (see above): one coalesced 16-B store per group
        gprev4[(long long)srow * nprev4 + sc4] = lds4[srow * buf_ld4 + sc4];
        srow += sdq; sc4 += sdm;
        if (sc4 >= nprev4) { sc4 -= nprev4; ++srow; }
      }
      fused_group<R, TPW, TM, D, true>(acc, bq, wp, lds4, abase, buf_ld4, g * D, K8, D, an);
    }
    if (G >= 1) {
      if (rem) fused_group<R, TPW, TM, D, true>(acc, bq, wp, lds4, abase, buf_ld4, (G - 1) * D, K8, D, an);
      else fused_group<R, TPW, TM, D, false>(acc, bq, wp, lds4, abase, buf_ld4, (G - 1) * D, K8, D, an);
    }
    if (rem) fused_group<R, TPW, TM, D, false>(acc, bq, wp, lds4, abase, buf_ld4, K8full, K8, K8 - K8full, an);
  }
  while (srow < srows) {   // what the main loop did not cover (idle waves, short reductions)
    gprev4[(long long)srow * nprev4 + sc4] = lds4[srow * buf_ld4 + sc4];
    srow += sdq; sc4 += sdm;
    if (sc4 >= nprev4) { sc4 -= nprev4; ++srow; }
  }
  if (packed_n) fused_ring_fill<TM, D>(bq, packed_n, K8n, K8n, ntiles_n, tpw_n, wave, lane);   // ahead of this layer's epilogue
  __syncthreads();   // A: every wave is done reading this layer's input
  if (active) {
    // lane (r, h) owns row r of each row tile, columns 32*tile + 8q + 4h + {0..3} (transposed-tile accumulator layout)
    float4* out4 = reinterpret_cast<float4*>(fsm);
    const bool full = row0 + 32 * R <= B;   // block-uniform: no per-row guard on the stash stores
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if (t0 + j < ntiles) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c4 = 8 * (t0 + j) + 2 * q + h;   // float4 column
          const float4 b4 = lds4[bias_lds4 + c4];
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const int row = row0 + 32 * i + r;
            float4 v;
            v.x = fused_elu(acc[i][j][4 * q] + b4.x);
            v.y = fused_elu(acc[i][j][4 * q + 1] + b4.y);
            v.z = fused_elu(acc[i][j][4 * q + 2] + b4.z);
            v.w = fused_elu(acc[i][j][4 * q + 3] + b4.w);
            out4[(32 * i + r) * buf_ld4 + c4] = v;
            if (gout && (full || row < B)) *reinterpret_cast<float4*>(gout + (long long)row * g_ld + 4 * c4) = v;
          }
        }
      }
    }
  }
  __syncthreads();   // B: the next layer's input is in place
}

// Output layer fused behind the hidden stack: Linear(K_h -> N <= 32) + {none, tanh, tanh + clipped target-policy noise}
// (+ second store), reference mlp.py:15-24,179 and pql_v_learner.py:62-71.  The 32 R x N outputs of a block are a
// sliver of MFMA work (K_h / 8 x 4 x R instructions), so the reduction range is SPLIT over the 8 waves, the partial tiles
// go through LDS (over the activation buffer, once every wave has read it) and are summed in wave order by all threads
// together with bias and activation.  That replaces a separate launch per forward (k_fwd_narrow / k_skinny_fwd: 6-10 us
// each); the summation order differs from theirs by reassociation only.
enum { HEAD_NONE = PQLK_ACT_NONE, HEAD_TANH = PQLK_ACT_TANH, HEAD_TANH_NOISE = PQLK_ACT_TANH_NOISE };

template <int R, bool TDH = true>
__device__ __forceinline__ void fused_head(const FusedP& p, const float* __restrict__ params, float* __restrict__ acts, int net, int row0,
                                           int buf_ld4, int wave, int lane) {
  constexpr int NW = FUSED_NW;
  const int r = lane & 31, h = lane >> 5;
  const int Kh = p.dims[p.n_hidden], K8 = Kh >> 3, N = p.head_n;
  const float* W = params + (long long)net * p.net_stride + p.head_w_off;     // (N, Kh) row-major: Kh % 32 == 0 -> ld = Kh
  const float4* wp = reinterpret_cast<const float4*>(W + (long long)min(r, N - 1) * Kh) + h;   // rows past N: clamped, never stored
  const float4* lds4 = reinterpret_cast<const float4*>(fsm);
  const int per = (K8 + NW - 1) / NW, k0 = wave * per, k1 = min(K8, k0 + per);
  f32x16f acc[R];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int k8 = k0; k8 < k1; ++k8) {
    const float4 w4 = wp[2 * k8];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const float4 a4 = lds4[(32 * i + r) * buf_ld4 + 2 * k8 + h];
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, a4.x, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, a4.y, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, a4.z, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, a4.w, acc[i], 0, 0, 0);
    }
  }
  // The partial tiles (NW x R x 16 x 64 floats) go over the activation buffer once every wave has read it -- except with the TD
  // head, which still needs the activations: there they go into the 256 columns to the right of them (float x of the partial
  // area sits in row x / 256; the host checked buf_ld - Kh >= 256, and 32 R rows x 256 = the NW x R x 1024 floats needed).
  const bool td = TDH && p.td_dz != nullptr;
  // TD inputs of the rows whose Q this thread will hold (head_ld = 32 columns per row -> thread tid owns column tid % 32 of rows
  // (tid + 512 k) / 32): requested here, ahead of the partial-tile round trip, so that their latency is not the kernel's tail
  constexpr int TDI = 2 * R;
  float tdt[TDI], tdr[TDI], tdd[TDI];
  if (td && ((wave * 64 + lane) & 31) == 0) {
#pragma unroll
    for (int k = 0; k < TDI; ++k) {
      const long long m = min(row0 + ((wave * 64 + lane) >> 5) + 16 * k, p.B - 1);
      tdt[k] = fminf(p.td_qt[m * p.head_ld], p.td_qt[((long long)p.B + m) * p.head_ld]);
      tdr[k] = p.td_rew[m]; tdd[k] = p.td_done[m];
    }
  }
  auto part_at = [&](int x) { return td ? (x >> 8) * p.buf_ld + Kh + (x & 255) : x; };
  __syncthreads();   // every wave is done reading the activations
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) fsm[part_at(((wave * R + i) * 16 + e) * 64 + lane)] = acc[i][e];
  __syncthreads();
  const float* bias = params + (long long)net * p.net_stride + p.head_b_off;
  float* out = acts + p.head_a_off + (long long)net * p.B * p.head_ld;
  const int tdbase = 32 * R * p.buf_ld;   // TD: [0, 32 R) 2/B (Q - y) per row, [32 R, 64 R) (Q - y)^2; over the bias table (no longer read)
  int ok = 0;   // trip count (TD: indexes the inputs requested above)
  for (int o = wave * 64 + lane; o < 32 * R * p.head_ld; o += 64 * NW, ++ok) {
    const int row = o / p.head_ld, c = o - row * p.head_ld;
    if (row0 + row >= p.B) {
      if (td && c == 0) { fsm[tdbase + row] = 0.f; fsm[tdbase + 32 * R + row] = 0.f; }
      continue;
    }
    float x = 0.f;   // pad column
    if (c < N) {
      const int i = row >> 5, ln = (row & 31) + 32 * ((c >> 2) & 1), e = 4 * (c >> 3) + (c & 3);   // accumulator slot of (row, c)
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += fsm[part_at(((w * R + i) * 16 + e) * 64 + ln)];
      x = s + bias[c];
      if (p.qc) p.qc[(long long)net * p.B + row0 + row] = x;   // (head_n == 1: c == 0)
      if (td) {   // N == 1: this thread holds Q(row) of this block's net.  y = r + (1-d) gamma^n min Q'  (pql_v_learner.py:104-108)
        float tq = tdt[0], tr = tdr[0], tn = tdd[0];
#pragma unroll
        for (int k = 1; k < TDI; ++k)
          if (ok == k) { tq = tdt[k]; tr = tdr[k]; tn = tdd[k]; }   // (compile-time indices: a runtime one would push the arrays to scratch)
        const float y = tr + ((1.f - tn) * p.td_gamma_n) * tq;
        const float dq = x - y;
        fsm[tdbase + row] = p.td_two_over_b * dq;
        fsm[tdbase + 32 * R + row] = dq * dq;
      }
      if (p.head_epi == HEAD_TANH) x = tanhf(x);
      else if (p.head_epi == HEAD_TANH_NOISE) {
        x = tanhf(x);
        float nz = p.noise_std * p.draw[(long long)(row0 + row) * N + c];
        nz = fminf(fmaxf(nz, -p.noise_clip), p.noise_clip);
        x = fminf(fmaxf(x + nz, -1.f), 1.f);
      }
      if (p.out2 && net == 0) p.out2[(long long)(row0 + row) * p.ld_out2 + c] = x;
    }
    out[(long long)(row0 + row) * p.head_ld + c] = x;
  }
  if (!td) return;
  // ---- the head's backward for this block's rows, off the activations still in LDS (what k_skinny_bwd<1, CH, true> does in a
  // launch of its own after re-reading them from HBM): dZ = dQ w * ELU'(h), bit for bit its values; dW / db / loss partials in a
  // fixed order (rows ascending inside a wave, then the waves 0..7), folded later by k_reduce_slabs / k_adamw like k_skinny_bwd's.
  __syncthreads();   // dQ of every row is in LDS; nobody reads the partial tiles any more
  const int kq = Kh >> 2;
  const int tile = row0 / (32 * R);
  float* dz = p.td_dz + ((long long)net * p.B + row0) * Kh;
  // wave w takes rows [w RW, (w + 1) RW), lane = column quad (chunks of 64 quads): ONE read of h serves dZ (a 1-KiB row segment per
  // store instruction) and the dW partial, which the wave sums over its rows in ascending order, parks in the free columns of
  // row w, and wave 0 adds over the eight waves in wave order
  constexpr int RW = 32 * R / NW;
  float* hp = p.td_head_part + ((long long)tile * p.n_nets + net) * p.td_part_floats;
  for (int q0 = 0; q0 < kq; q0 += 64) {
    const int q = q0 + lane;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < kq) {
      const float4 w4 = *reinterpret_cast<const float4*>(W + 4 * q);
#pragma unroll
      for (int u = 0; u < RW; ++u) {
        const int row = wave * RW + u;
        const float dn = fsm[tdbase + row];
        const float4 h4 = lds4[row * buf_ld4 + q];
        s.x += dn * h4.x; s.y += dn * h4.y; s.z += dn * h4.z; s.w += dn * h4.w;
        float4 a = make_float4(dn * w4.x, dn * w4.y, dn * w4.z, dn * w4.w);
        a.x = h4.x > 0.f ? a.x : a.x * (h4.x + 1.f);   // ELU'(x) = elu(x) + 1 for x <= 0
        a.y = h4.y > 0.f ? a.y : a.y * (h4.y + 1.f);
        a.z = h4.z > 0.f ? a.z : a.z * (h4.z + 1.f);
        a.w = h4.w > 0.f ? a.w : a.w * (h4.w + 1.f);
        if (row0 + row < p.B) *reinterpret_cast<float4*>(dz + (long long)row * Kh + 4 * q) = a;
      }
    }
    __syncthreads();   // (the previous chunk's sums have been folded)
    if (q < kq) *reinterpret_cast<float4*>(&fsm[wave * p.buf_ld + Kh + 4 * lane]) = s;
    __syncthreads();
    if (wave == 0 && q < kq) {
      float4 t = *reinterpret_cast<const float4*>(&fsm[Kh + 4 * lane]);
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(&fsm[w * p.buf_ld + Kh + 4 * lane]);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      *reinterpret_cast<float4*>(hp + 4 * q) = t;
    }
  }
  if (wave == 1) {   // db (+ the zero pad of the bias block) and the loss partial: lane = row, then the fixed xor-shuffle tree
    float db = lane < 32 * R ? fsm[tdbase + lane] : 0.f, ls = lane < 32 * R ? fsm[tdbase + 32 * R + lane] : 0.f;
    db = wave_sum(db); ls = wave_sum(ls);   // (one lane walking the 32 R rows serially was 128 dependent LDS reads at the block's tail: +4 us)
    if (lane < 32) hp[Kh + lane] = lane == 0 ? db : 0.f;
    if (lane == 0) p.td_loss_part[(long long)tile * p.n_nets + net] = ls;
  }
}

// R row tiles of 32 per block; TM = widest per-wave tile count any layer needs (2: widths <= 512, 4: <= 1024).
// OUT_ONLY: the forward-only instantiation behind PQLK_STASH_OUTPUT_ONLY (no stash, no TD head: the launch that serves K learner
// steps at once, e.g. the target policy's K x B rows) -- a kernel of its own also in the profiler's per-kernel tables.
template <int R, int TM, bool OUT_ONLY = false>
__global__ __launch_bounds__(64 * FUSED_NW) void k_mlp_fwd_fused(FusedP p) {
  // four tiles per wave: 16 MFMAs per step already cover the L2 latency with a 2-deep ring (and 4 x 4 quads would spill)
  constexpr int NW = FUSED_NW, D = TM >= 4 ? 2 : PQLK_FUSED_DEEP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware block -> (net, row tile) map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share
  // an L2), and a twin critic's fragment-ordered weights are 2 x 1.8 MB against a 4 MB L2: with both nets on every
  // XCD the weight stream thrashes L2 and falls back to the Infinity Cache.  Even XCD groups take net 0, odd ones
  // net 1, so each L2 keeps ONE net's weights resident.  (Speed only: any placement computes the same result.)
  int net, tile;
  const int tiles = (p.B + 32 * R - 1) / (32 * R);
  if (p.n_nets == 2 && (tiles & 3) == 0) {
    const int b = blockIdx.x, g = b & 7, i = b >> 3;
    net = g & 1;
    tile = i * 4 + (g >> 1);
  } else {
    net = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  const float* __restrict__ pX = p.X;
  const float* __restrict__ pparams = p.params;
  const float* __restrict__ ppacked = p.packed;
  float* __restrict__ pacts = p.acts;
  const int pstash = OUT_ONLY ? 0 : p.stash_all;
  const int row0 = tile * 32 * R;
  const bool full_tile = row0 + 32 * R <= p.B;
  const int buf_ld4 = p.buf_ld >> 2;
  const float4* packed_net = reinterpret_cast<const float4*>(ppacked + (long long)net * p.packed_net_stride);
  auto tpw_of = [](int ntiles) { return ntiles > 2 * NW ? 4 : ntiles > NW ? 2 : 1; };
  float4 bq[D][TM];
  // ring of layer 0, issued before the input tile is staged
  fused_ring_fill<TM, D>(bq, packed_net + (p.p_off[0] >> 2), (p.dims[0] + 7) >> 3, ((p.dims[0] + 31) & ~31) >> 3, p.dims[1] >> 5,
                         tpw_of(p.dims[1] >> 5), wave, lane);
  float4* out4 = reinterpret_cast<float4*>(fsm);
  const int bias4 = 32 * R * buf_ld4;   // float4 offset of the bias table: layer l at bias4 + l * (buf_ld4 - 1)
  // Staging: every global load of the prologue (bias rows, then the input tile four quads at a time) is issued before
  // the LDS stores that consume it.  One load -> one store per iteration exposes a full memory latency each time, and
  // every block of the grid sits in this prologue at the same moment.
  {
    float4 bv[PQLK_MAX_LAYERS];   // hidden widths <= 1024 -> at most one float4 of each layer's bias per thread
#pragma unroll
    for (int l = 0; l < PQLK_MAX_LAYERS; ++l) {
      bv[l] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (l < p.n_hidden && tid < (p.dims[l + 1] >> 2))
        bv[l] = reinterpret_cast<const float4*>(pparams + (long long)net * p.net_stride + p.b_off[l])[tid];
    }
    const int k0 = (p.dims[0] + 31) & ~31, cpr = k0 >> 2, w = p.dims[0], total = 32 * R * cpr;
    // columns past the logical input width are forced to zero, so X may be a wider matrix whose extra columns hold
    // something else (the target actor reads its observations straight out of the critic's [obs | action] tile);
    // rows past B are zero-filled
    for (int i0 = tid; i0 < total; i0 += 4 * 64 * NW) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 64 * NW;
        const int row = i / cpr, c4 = i - row * cpr;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total && row0 + row < p.B) {
          const int c = 4 * c4;
          if (p.X2 && c >= p.x2_col0) {
            if (c - p.x2_col0 < p.ldx2) v[u] = *reinterpret_cast<const float4*>(p.X2 + (long long)(row0 + row) * p.ldx2 + (c - p.x2_col0));
          } else if (c < p.ldx) {
            v[u] = *reinterpret_cast<const float4*>(pX + (long long)(row0 + row) * p.ldx + c);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 64 * NW;
        if (i < total) {
          const int row = i / cpr, c4 = i - row * cpr, c = 4 * c4;
          float4 x = v[u];
          if (c + 3 >= w) {
            if (c >= w) x.x = 0.f;
            if (c + 1 >= w) x.y = 0.f;
            if (c + 2 >= w) x.z = 0.f;
            x.w = 0.f;
          }
          out4[row * buf_ld4 + c4] = x;
        }
      }
    }
#pragma unroll
    for (int l = 0; l < PQLK_MAX_LAYERS; ++l)
      if (l < p.n_hidden && tid < (p.dims[l + 1] >> 2)) out4[bias4 + l * (buf_ld4 - 1) + tid] = bv[l];
  }
  __syncthreads();
  for (int l = 0; l < p.n_hidden; ++l) {
    const int K8s = ((p.dims[l] + 31) & ~31) >> 3, K8 = (p.dims[l] + 7) >> 3, N = p.dims[l + 1], ntiles = N >> 5;
    const float4* packed_l = packed_net + (p.p_off[l] >> 2);
    const int bias_l = bias4 + l * (buf_ld4 - 1);
    const bool last = l + 1 == p.n_hidden;
    // layer l's output goes to HBM either from its own epilogue (last hidden layer, ragged tiles) or from the next
    // layer's main loop (deferred: full tiles of a stashing forward)
    const bool defer = pstash && full_tile;
    // (a forward-only call whose output layer is fused needs no HBM copy of the last hidden layer at all)
    const bool keep_last = last && (pstash || p.head_n == 0) && !p.td_dz;   // (the TD head consumes the last hidden layer in LDS: backward never reads it)
    float* gout = (keep_last || (pstash && !full_tile)) ? pacts + p.a_off[l] + (long long)net * p.B * N : nullptr;
    float* gprev = (defer && l > 0) ? pacts + p.a_off[l - 1] + (long long)net * p.B * p.dims[l] : nullptr;
    const int nprev4 = p.dims[l] >> 2;
    const float4* packed_n = last ? nullptr : packed_net + (p.p_off[l + 1] >> 2);
    const int K8n = last ? 1 : N >> 3, ntiles_n = last ? 1 : p.dims[l + 2] >> 5;
    const int tpw = tpw_of(ntiles), tpw_n = tpw_of(ntiles_n);
    // The wide instantiation compiles THREE layer bodies (4, 2, 1 tiles per wave); hipcc hoists the loop-invariant part of all
    // their epilogue / stash addresses above the layer loop and, at 256 registers, spills it (55 VGPRs, round 3).  An opaque
    // per-layer copy of the lane id keeps those few integer ops inside the layer they belong to.
    int lane_l = lane;
    if (TM >= 4) asm volatile("" : "+v"(lane_l));
    if (TM >= 4 && tpw == 4)
      fused_layer<R, (TM >= 4 ? 4 : 1), TM, D>(bq, buf_ld4, K8, K8s, ntiles, packed_l, bias_l, gout, N, row0, p.B, wave, lane_l, packed_n, K8n,
                                               ntiles_n, tpw_n, gprev, nprev4);
    else if (tpw == 2)
      fused_layer<R, 2, TM, D>(bq, buf_ld4, K8, K8s, ntiles, packed_l, bias_l, gout, N, row0, p.B, wave, lane_l, packed_n, K8n, ntiles_n,
                               tpw_n, gprev, nprev4);
    else
      fused_layer<R, 1, TM, D>(bq, buf_ld4, K8, K8s, ntiles, packed_l, bias_l, gout, N, row0, p.B, wave, lane_l, packed_n, K8n, ntiles_n,
                               tpw_n, gprev, nprev4);
  }
  if (p.head_n > 0) fused_head<R, !OUT_ONLY>(p, pparams, pacts, net, row0, buf_ld4, wave, lane);   // the LDS buffer holds the last hidden layer's output
}

// arena -> fragment-ordered copy of the hidden layers' weights (one thread per element; 1-3 M elements); ONE launch for all
// hidden layers: grid.z = layer (a re-pack follows every weight hand-off, three per rollout iteration)
struct PackP {
  int n_layers;
  long long w_off[PQLK_MAX_LAYERS], p_off[PQLK_MAX_LAYERS];
  int N[PQLK_MAX_LAYERS], K[PQLK_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void k_mlp_pack(const float* __restrict__ params, float* __restrict__ packed, PackP pp,
                                                  long long net_stride, long long packed_net_stride) {
  const int net = blockIdx.y, layer = blockIdx.z;
  const int N = pp.N[layer], K = pp.K[layer], ldk = K;
  const float* W = params + (long long)net * net_stride + pp.w_off[layer];
  float* dst = packed + (long long)net * packed_net_stride + pp.p_off[layer];
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
