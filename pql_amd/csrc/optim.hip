// Optimiser over flat arenas: global-norm clip + AdamW (+ Polyak), Polyak alone, batch moments.
// Reference: torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW(defaults) as called at
// pql/algo/pql_v_learner.py:124-133 / pql_p_learner.py:87-96; soft_update pql/utils/torch_util.py:9-12;
// RunningMeanStd.update pql/utils/torch_util.py:77-81.
// HBM-bound: 20 B read + 16 B written per parameter (+4 B for the norm pass); the reference issues
// ~100 small launches for the same work.
#include "pqlk_common.h"

#define OPT_MAX_BLOCKS 1024

__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, int64_t n, float* __restrict__ part,
                                               int32_t* __restrict__ step_dev) {
  __shared__ float shw[4];
  const int64_t n4 = n >> 2;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0) {  // tail (n is a multiple of 32 for MLP arenas; kept for generality)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (shw[0] + shw[1]) + (shw[2] + shw[3]);
    if (blockIdx.x == 0 && step_dev) step_dev[0] += 1;  // t for the update kernel that follows on the stream
  }
}

// Optional fused re-pack: while the optimiser has each new weight (and Polyak target) in a register it also writes
// it to the fragment-ordered copy used by the fused forward kernel (layout: fused.h), so no separate pack launches.
#define PACK_MAX 16
struct PackSpec {
  int n;                      // number of (net, hidden layer) weight blocks; 0 = no packing
  long long w_off[PACK_MAX];  // arena offset of the block
  long long w_end[PACK_MAX];  // w_off + N * K
  long long p_off[PACK_MAX];  // offset of the block in the packed buffer
  int K[PACK_MAX];            // padded in-features (row length of the block)
  float* packed_p;
  float* packed_t;
};

__device__ __forceinline__ long long packed_index(const PackSpec& ps, long long i) {
#pragma unroll 1
  for (int e = 0; e < ps.n; ++e) {
    if (i >= ps.w_off[e] && i < ps.w_end[e]) {
      const long long rel = i - ps.w_off[e];
      const int K = ps.K[e];
      const int n = (int)(rel / K), k = (int)(rel - (long long)n * K);
      const int tile = n >> 5, r = n & 31, k8 = k >> 3, h = (k >> 2) & 1, j = k & 3;
      return ps.p_off[e] + (((long long)tile * (K >> 3) + k8) * 64 + h * 32 + r) * 4 + j;
    }
  }
  return -1;
}

// Which 16-B quad of the arena thread number `q` (linear over the arena's quads) works on, and where that quad lives in the
// fragment-ordered copy (-1: nowhere).  Outside the packed weight blocks: quad q itself.  Inside one (N x K, both multiples of 32)
// the block's quads are dealt out in tiles of 32 rows x 32 columns, 256 consecutive thread numbers per tile, thread t of a tile
// taking (row t / 8, quad t % 8): eight lanes still cover one 128-B line of every arena stream, and the eight ROWS a wave holds for
// a given (k8, h) are eight CONSECUTIVE 16-B slots of the packed copy (slot = (tile, k8, h, r)), so the re-pack stores go out as
// whole 128-B runs.  With the identity map the same stores were isolated 16-B writes 512 B apart: 3.4 of the launch's ~10 us
// (tools/kbench.py opt / opt_nopack, round 3).  A bijection on each block's quads, elementwise arithmetic: same bits.
__device__ __forceinline__ long long adam_quad(const PackSpec& ps, long long q, long long& packed) {
  packed = -1;
#pragma unroll 1
  for (int e = 0; e < ps.n; ++e) {
    const long long q0 = ps.w_off[e] >> 2, q1 = ps.w_end[e] >> 2;
    if (q >= q0 && q < q1) {
      const long long rel = q - q0;
      const int K = ps.K[e], kb = K >> 5;
      const int jb = (int)(rel >> 8), t = (int)(rel & 255);
      const int tr = jb / kb, cb = jb - tr * kb, r = t >> 3, cq = t & 7;
      packed = ps.p_off[e] + (((long long)tr * (K >> 3) + 4 * cb + (cq >> 1)) * 64 + (cq & 1) * 32 + r) * 4;
      return q0 + (long long)(32 * tr + r) * (K >> 2) + 8 * cb + cq;
    }
  }
  return q;
}

struct AdamC {
  float lr_wd_decay;  // 1 - lr*wd
  float w1;           // 1 - b1      (lerp weight)
  float b2, one_m_b2;
  float eps, tau, one_m_tau;
  float max_norm, grad_scale;
  double lr, b1d, b2d;
};

// A pending loss fold carried by the optimiser launch: block 0 sums the loss kernel's per-block partials exactly like
// k_sum_partials (loss.hip) would -- same per-thread order, same tree -- and writes ring slot (t - 1) % ring_len, t = the
// step counter AFTER this step's increment (the slot the stand-alone fold would have used before it).
struct LossFold {
  const float* part; int n; float scale; float* ring; int ring_len;
};

__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, float* __restrict__ target, int64_t n,
                                               const float* __restrict__ part, int nparts, AdamC c,
                                               const int32_t* __restrict__ step_dev, float* __restrict__ gnorm_out,
                                               PackSpec ps, LossFold lf) {
  __shared__ float shn[4], shl[4];
  __shared__ float s_step_size, s_bc2_sqrt;
  // The operands of this thread's first quad are requested BEFORE the norm / step-size preamble (most blocks only ever
  // process one quad per thread, and a block otherwise has nothing in flight meanwhile).
  typedef float f4 __attribute__((ext_vector_type(4)));
  auto al16 = [](const void* q) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0; };
  const bool vec = al16(p) && al16(g) && al16(m) && al16(v) && (!target || al16(target));
  const int64_t n4 = vec ? (n >> 2) : 0;
  const int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  f4 g4n = f4{0.f, 0.f, 0.f, 0.f}, p4n = g4n, m4n = g4n, v4n = g4n, t4n = g4n;
  long long pkn = -1;                                         // packed-copy offset of the quad held in *4n
  int64_t qan = q0 < n4 ? adam_quad(ps, q0, pkn) : 0;         // ... and its place in the arena
  if (q0 < n4) {
    g4n = reinterpret_cast<const f4*>(g)[qan];
    p4n = reinterpret_cast<f4*>(p)[qan]; m4n = reinterpret_cast<f4*>(m)[qan]; v4n = reinterpret_cast<f4*>(v)[qan];
    if (target) t4n = reinterpret_cast<f4*>(target)[qan];
  }
  // Preamble, ONE barrier: every block re-reduces the (<= 2048) squared-norm partials in the same fixed order (strided
  // per-thread sums -> xor-shuffle tree per wave -> the four waves in order) -> identical clip factor everywhere; the two
  // double-precision pow() of the bias corrections run meanwhile on one lane of wave 1, and block 0 folds the pending loss
  // partials the same way.  (Round 2's form -- an 8-level LDS tree with a barrier per level, then the pow() chain on thread
  // 0, then another barrier -- kept every block idle for ~5 us of a 14-us launch.)
  const bool fold = lf.part && blockIdx.x == 0;   // block-uniform
  if (threadIdx.x == 64) {
    const int t = step_dev[0];
    const double bc1 = 1.0 - pow(c.b1d, (double)t);
    const double bc2 = 1.0 - pow(c.b2d, (double)t);
    s_step_size = (float)(c.lr / bc1);
    s_bc2_sqrt = (float)sqrt(bc2);
  }
  float s = 0.f, ls = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  if (fold)
    for (int i = threadIdx.x; i < lf.n; i += 256) ls += lf.part[i];
  s = wave_sum(s);
  if (fold) ls = wave_sum(ls);
  if ((threadIdx.x & 63) == 0) { shn[threadIdx.x >> 6] = s; shl[threadIdx.x >> 6] = ls; }
  __syncthreads();
  const float total = sqrtf((shn[0] + shn[1]) + (shn[2] + shn[3])) * c.grad_scale;  // norm of the scaled gradient
  float clipc = 1.f;
  if (c.max_norm > 0.f) clipc = fminf(c.max_norm / (total + 1e-6f), 1.f);  // clip_grad_norm_
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (gnorm_out) gnorm_out[0] = total;
    if (fold) lf.ring[(step_dev[0] - 1) % lf.ring_len] = ((shl[0] + shl[1]) + (shl[2] + shl[3])) * lf.scale;
  }
  const float coef = clipc * c.grad_scale, step_size = s_step_size, bc2_sqrt = s_bc2_sqrt;
  // one parameter: the arithmetic of torch's AdamW foreach kernels, op for op (per element, so the 16-B path below
  // produces the same bits as the scalar one)
  auto upd = [&](float gi, float& pi, float& mi, float& vi, float& ti) {
    gi = gi * coef;
    pi = pi * c.lr_wd_decay;                     // p.mul_(1 - lr*wd)
    mi = mi + c.w1 * (gi - mi);                  // m.lerp_(g, 1-b1)
    vi = vi * c.b2 + (c.one_m_b2 * gi) * gi;     // v.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(vi) / bc2_sqrt + c.eps;
    pi = pi + (-step_size) * (mi / denom);       // p.addcdiv_(m, denom, -lr/bc1)
    ti = target ? pi * c.tau + ti * c.one_m_tau : 0.f;   // soft_update
  };
  // 16-B path: four consecutive parameters per thread.  Arena blocks start on multiples of 32 floats and rows are
  // multiples of 32 long, so an aligned quad never straddles a packed block and maps to ONE 16-B quad of the
  // fragment-ordered copy (j = k & 3 runs over the quad): the re-pack is a 16-B store too.  Which quad a thread takes:
  // adam_quad above.
  for (int64_t q4 = q0; q4 < n4; q4 += (int64_t)gridDim.x * 256) {
    const int64_t qa = qan;
    const long long pk = pkn;
    const f4 g4 = g4n;
    f4 p4 = p4n, m4 = m4n, v4 = v4n, t4 = t4n;
    const int64_t qn = q4 + (int64_t)gridDim.x * 256;   // the next quad of this thread, requested before this one is stored
    if (qn < n4) {
      qan = adam_quad(ps, qn, pkn);
      g4n = reinterpret_cast<const f4*>(g)[qan];
      p4n = reinterpret_cast<f4*>(p)[qan]; m4n = reinterpret_cast<f4*>(m)[qan]; v4n = reinterpret_cast<f4*>(v)[qan];
      if (target) t4n = reinterpret_cast<f4*>(target)[qan];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float pi = p4[u], mi = m4[u], vi = v4[u], ti = t4[u];
      upd(g4[u], pi, mi, vi, ti);
      p4[u] = pi; m4[u] = mi; v4[u] = vi; t4[u] = ti;
    }
    reinterpret_cast<f4*>(p)[qa] = p4;
    reinterpret_cast<f4*>(m)[qa] = m4;
    reinterpret_cast<f4*>(v)[qa] = v4;
    if (target) reinterpret_cast<f4*>(target)[qa] = t4;
    if (pk >= 0) {
      *reinterpret_cast<f4*>(ps.packed_p + pk) = p4;
      if (target && ps.packed_t) *reinterpret_cast<f4*>(ps.packed_t + pk) = t4;
    }
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {   // tail / unaligned
    float pi = p[i], mi = m[i], vi = v[i], ti = target ? target[i] : 0.f;
    upd(g[i], pi, mi, vi, ti);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (target) target[i] = ti;
    if (ps.n > 0) {
      const long long q = packed_index(ps, i);
      if (q >= 0) {
        ps.packed_p[q] = pi;
        if (target && ps.packed_t) ps.packed_t[q] = ti;
      }
    }
  }
}

static int adamw_impl(float* p, float* g, float* m, float* v, float* target, int64_t n, float grad_scale, float max_norm, float lr,
                      float b1, float b2, float eps, float wd, float tau, int32_t* step_dev, float* gnorm_out, float* scratch,
                      const PackSpec& ps, pqlk_stream_t stream, int prenorm = 0, LossFold lf = LossFold{}) {
  PQLK_REQUIRE(p && g && m && v && step_dev && scratch, PQLK_E_NULL);
  PQLK_REQUIRE(n > 0, PQLK_E_SHAPE);
  PQLK_REQUIRE(pqlk_aligned16(g), PQLK_E_ALIGN);
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > OPT_MAX_BLOCKS) blocks = OPT_MAX_BLOCKS;
  if (prenorm > 0) {   // `scratch` already holds `prenorm` partials and the step counter is bumped (pqlk_mlp_backward_norm)
    PQLK_REQUIRE(prenorm <= 2048, PQLK_E_SHAPE);
    blocks = prenorm;
  } else {
    hipLaunchKernelGGL(k_sumsq, dim3(blocks), dim3(256), 0, pqlk_s(stream), g, n, scratch, step_dev);
    PQLK_LAUNCH_CHECK();
  }
  AdamC c;
  // scalar constants are formed in double (python floats in torch) and rounded once to fp32
  c.lr_wd_decay = (float)(1.0 - (double)lr * (double)wd);
  c.w1 = (float)(1.0 - (double)b1);
  c.b2 = b2;
  c.one_m_b2 = (float)(1.0 - (double)b2);
  c.eps = eps;
  c.tau = tau;
  c.one_m_tau = (float)(1.0 - (double)tau);
  c.max_norm = max_norm;
  c.grad_scale = grad_scale;
  c.lr = (double)lr;
  c.b1d = (double)b1;
  c.b2d = (double)b2;
  int blocks2 = (int)(((n + 3) / 4 + 255) / 256);   // one 16-B quad per thread
  if (blocks2 > 2048) blocks2 = 2048;
  hipLaunchKernelGGL(k_adamw, dim3(blocks2), dim3(256), 0, pqlk_s(stream), p, g, m, v, target, n, scratch, blocks, c, step_dev,
                     gnorm_out, ps, lf);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

extern "C" int pqlk_clip_adamw_polyak(float* p, float* g, float* m, float* v, float* target, int64_t n, float grad_scale,
                                      float max_norm, float lr, float b1, float b2, float eps, float wd, float tau,
                                      int32_t* step_dev, float* gnorm_out, float* scratch, pqlk_stream_t stream) {
  PackSpec ps = {};
  return adamw_impl(p, g, m, v, target, n, grad_scale, max_norm, lr, b1, b2, eps, wd, tau, step_dev, gnorm_out, scratch, ps, stream);
}

// Same, for the arena of an MLP described by `d`, also refreshing the fragment-ordered copies of the hidden-layer
// weights (pqlk_mlp_pack layout): packed_p for the parameters, packed_t (may be NULL) for the Polyak target.
static int build_pack_spec(const PqlMlpDesc* d, float* packed_p, float* packed_t, PackSpec& ps) {
  PQLK_REQUIRE(d && packed_p, PQLK_E_NULL);
  PQLK_REQUIRE(d->n_layers >= 2 && d->n_layers <= PQLK_MAX_LAYERS && d->n_nets >= 1 && d->n_nets <= 2, PQLK_E_SHAPE);
  PQLK_REQUIRE((d->n_layers - 1) * d->n_nets <= PACK_MAX, PQLK_E_UNSUPPORTED);
  PQLK_REQUIRE(pqlk_mlp_packed_floats(d) > 0, PQLK_E_UNSUPPORTED);
  const int64_t net_stride = pqlk_mlp_net_stride(d);
  const int64_t packed_stride = pqlk_mlp_packed_floats(d) / d->n_nets;
  ps = PackSpec{};
  ps.packed_p = packed_p;
  ps.packed_t = packed_t;
  for (int net = 0; net < d->n_nets; ++net) {
    int64_t p_off = 0;
    for (int l = 0; l + 1 < d->n_layers; ++l) {
      int64_t w_off, b_off;
      pqlk_mlp_layer_offsets(d, l, &w_off, &b_off);
      const int64_t K = pqlk_ld(d->dims[l]), N = d->dims[l + 1];
      const int e = ps.n++;
      ps.w_off[e] = net * net_stride + w_off;
      ps.w_end[e] = ps.w_off[e] + N * K;
      ps.p_off[e] = net * packed_stride + p_off;
      ps.K[e] = (int)K;
      p_off += N * K;
    }
  }
  return PQLK_OK;
}

extern "C" int pqlk_clip_adamw_polyak_pack(const PqlMlpDesc* d, float* p, float* g, float* m, float* v, float* target,
                                           float* packed_p, float* packed_t, float grad_scale, float max_norm, float lr, float b1,
                                           float b2, float eps, float wd, float tau, int32_t* step_dev, float* gnorm_out,
                                           float* scratch, pqlk_stream_t stream) {
  PackSpec ps;
  const int rc = build_pack_spec(d, packed_p, packed_t, ps);
  if (rc) return rc;
  return adamw_impl(p, g, m, v, target, pqlk_mlp_param_floats(d), grad_scale, max_norm, lr, b1, b2, eps, wd, tau, step_dev,
                    gnorm_out, scratch, ps, stream);
}

// The optimiser launch of a fused learner step.  packed_p may be NULL (no fragment-ordered copies to refresh).
//   prenorm > 0  : `scratch` already holds that many squared-norm partials of the gradient and `step_dev` is already
//                  incremented -- both left by pqlk_mlp_backward_norm -- so the k_sumsq launch is skipped;
//   loss_part    : optional per-block loss partials (pqlk_*_loss called with loss_out = NULL): block 0 folds loss_parts of
//                  them, times loss_scale, into loss_ring[(t - 1) % ring_len] instead of a separate one-block launch.
extern "C" int pqlk_adamw_polyak_fused(const PqlMlpDesc* d, float* p, float* g, float* m, float* v, float* target,
                                       float* packed_p, float* packed_t, float grad_scale, float max_norm, float lr, float b1,
                                       float b2, float eps, float wd, float tau, int32_t* step_dev, float* gnorm_out,
                                       float* scratch, int32_t prenorm, const float* loss_part, int32_t loss_parts,
                                       float loss_scale, float* loss_ring, int32_t ring_len, pqlk_stream_t stream) {
  PQLK_REQUIRE(d, PQLK_E_NULL);
  PackSpec ps = {};
  if (packed_p) {
    const int rc = build_pack_spec(d, packed_p, packed_t, ps);
    if (rc) return rc;
  }
  LossFold lf = {};
  if (loss_part) {
    PQLK_REQUIRE(loss_ring && ring_len > 0 && loss_parts > 0, PQLK_E_SHAPE);
    lf.part = loss_part; lf.n = loss_parts; lf.scale = loss_scale; lf.ring = loss_ring; lf.ring_len = ring_len;
  }
  return adamw_impl(p, g, m, v, target, pqlk_mlp_param_floats(d), grad_scale, max_norm, lr, b1, b2, eps, wd, tau, step_dev,
                    gnorm_out, scratch, ps, stream, prenorm, lf);
}

__global__ __launch_bounds__(256) void k_polyak(float* __restrict__ target, const float* __restrict__ cur, int64_t n,
                                                float tau, float one_m_tau) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    target[i] = cur[i] * tau + target[i] * one_m_tau;
}

extern "C" int pqlk_polyak(float* target, const float* cur, int64_t n, float tau, pqlk_stream_t stream) {
  PQLK_REQUIRE(target && cur, PQLK_E_NULL);
  PQLK_REQUIRE(n > 0, PQLK_E_SHAPE);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_polyak, dim3(blocks), dim3(256), 0, pqlk_s(stream), target, cur, n, tau,
                     (float)(1.0 - (double)tau));
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

// ------------------------------------------------------------------------------------------------
// batch mean / unbiased variance per column, two stages so the whole chip streams the (N, cols) block:
//   stage 1: grid (col tiles of 32, row chunks): per chunk mean and M2 = sum (x - chunk_mean)^2 (two passes over
//            rows the block just touched, L2-resident);  stage 2: Chan merge of the chunk moments in fixed order.
#define MOM_CHUNKS 64
__global__ __launch_bounds__(256) void k_moments_stage1(const float* __restrict__ x, int64_t ldx, int64_t n, int cols,
                                                        int64_t rows_per_chunk, float* __restrict__ part) {
  __shared__ float sh[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cx;
  const int64_t r0 = blockIdx.y * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < n ? r0 + rows_per_chunk : n;
  const float cnt = (float)(r1 > r0 ? r1 - r0 : 0);
  float s = 0.f;
  if (col < cols)
    for (int64_t r = r0 + ry; r < r1; r += 8) s += x[r * ldx + col];
  sh[ry][cx] = s;
  __syncthreads();
  float mean = 0.f;
  for (int k = 0; k < 8; ++k) mean += sh[k][cx];
  mean = cnt > 0.f ? mean / cnt : 0.f;
  __syncthreads();
  float q = 0.f;
  if (col < cols)
    for (int64_t r = r0 + ry; r < r1; r += 8) {
      const float d = x[r * ldx + col] - mean;
      q += d * d;
    }
  sh[ry][cx] = q;
  __syncthreads();
  if (ry == 0 && col < cols) {
    float t = 0.f;
    for (int k = 0; k < 8; ++k) t += sh[k][cx];
    float* o = part + ((int64_t)blockIdx.y * cols + col) * 3;
    o[0] = cnt; o[1] = mean; o[2] = t;
  }
}

// One WAVE per column, lane = chunk: every partial is fetched by its own lane (one round trip), then the 64 chunk moments are merged
// pairwise (Chan) by a fixed xor-shuffle tree -- at every level both partners evaluate the SAME expression (lower lane = a, upper
// lane = b), so the result does not depend on which lane one reads.  (Rounds 1-3 let one thread per column walk the 64 partials
// serially, two IEEE divisions per partial in a dependent chain: 20 us for a 4096 x 88 batch, the slowest launch of the rollout.)
__global__ __launch_bounds__(256) void k_moments_stage2(const float* __restrict__ part, int chunks, int cols, int64_t n,
                                                        float* __restrict__ mean_out, float* __restrict__ var_out) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= cols) return;   // wave-uniform
  float cnt = 0.f, mean = 0.f, m2 = 0.f;
  if (lane < chunks) {
    const float* o = part + ((int64_t)lane * cols + col) * 3;
    cnt = o[0]; mean = o[1]; m2 = o[2];
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float c2 = __shfl_xor(cnt, o, 64), e2 = __shfl_xor(mean, o, 64), q2 = __shfl_xor(m2, o, 64);
    const bool upper = (lane & o) != 0;
    const float ca = upper ? c2 : cnt, ea = upper ? e2 : mean, qa = upper ? q2 : m2;
    const float cb = upper ? cnt : c2, eb = upper ? mean : e2, qb = upper ? m2 : q2;
    const float tot = ca + cb;
    if (tot > 0.f) {
      const float delta = eb - ea;
      mean = ea + delta * cb / tot;
      m2 = (qa + qb) + delta * delta * ca * cb / tot;
    } else {
      mean = 0.f; m2 = 0.f;
    }
    cnt = tot;
  }
  if (lane == 0) {
    mean_out[col] = mean;
    var_out[col] = m2 / (float)(n - 1);
  }
}

extern "C" int pqlk_batch_moments(const float* x, int64_t ldx, int64_t n, int32_t cols, float* mean_out, float* var_out,
                                  float* scratch, pqlk_stream_t stream) {
  PQLK_REQUIRE(x && mean_out && var_out && scratch, PQLK_E_NULL);
  PQLK_REQUIRE(n >= 2 && cols > 0 && ldx >= cols, PQLK_E_SHAPE);
  int chunks = (int)((n + 63) / 64);
  if (chunks > MOM_CHUNKS) chunks = MOM_CHUNKS;
  const int64_t rows_per_chunk = (n + chunks - 1) / chunks;
  hipLaunchKernelGGL(k_moments_stage1, dim3((cols + 31) / 32, chunks), dim3(256), 0, pqlk_s(stream), x, ldx, n, (int)cols,
                     rows_per_chunk, scratch);
  PQLK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_moments_stage2, dim3((cols + 3) / 4), dim3(256), 0, pqlk_s(stream), scratch, chunks, (int)cols, n,
                     mean_out, var_out);
  PQLK_LAUNCH_CHECK();
  return PQLK_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int pqlk_version(void) { return PQLK_VERSION; }

extern "C" const char* pqlk_strerror(int rc) {
  if (rc == 0) return "ok";
  if (rc < 0) return hipGetErrorString((hipError_t)(-rc));
  switch (rc) {
    case PQLK_E_NULL: return "pqlk: required pointer is NULL";
    case PQLK_E_SHAPE: return "pqlk: bad or inconsistent dimension";
    case PQLK_E_RANGE: return "pqlk: index or offset out of range";
    case PQLK_E_ALIGN: return "pqlk: leading dimension not a multiple of 32 or pointer not 16-byte aligned";
    case PQLK_E_UNSUPPORTED: return "pqlk: unsupported configuration";
    case PQLK_E_WORKSPACE: return "pqlk: workspace too small";
    default: return "pqlk: unknown error";
  }
}
