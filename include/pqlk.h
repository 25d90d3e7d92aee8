/*
 * pqlk.h -- C ABI of libpqlk.so, the MI355X (gfx950) kernel library for the
 * Parallel-Q-Learning learner hot path.
 *
 * The reference (supersglzc/pql) is pure Python on PyTorch: it has no FFI.  Its
 * plugin boundary is (i) class-name lookup of models/algos, (ii) the Python
 * signatures of the replay/learner objects and (iii) the Hydra key set
 * (SURVEY.md section 8b).  pql_amd/ keeps (i)-(iii) and calls the entry points below
 * through ctypes; every entry point names the reference code it replaces
 * (path:line under the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / HIP types in signatures
 *     (pqlk_stream_t is a hipStream_t passed as void*; NULL = default stream).
 *   - All pointers are DEVICE pointers on the current HIP device unless a
 *     parameter is documented as host.  The caller owns every buffer; the
 *     library never allocates, frees or retains a pointer past the call.
 *   - Every call is asynchronous on the given stream and performs no host
 *     synchronisation, so calls can be captured into a hipGraph.
 *   - Return value: 0 = OK; >0 = PQLK_E_* argument error; <0 = -(hipError_t).
 *     pqlk_strerror() describes either.
 *   - Matrices are row-major fp32 with a leading dimension ("ld", in floats)
 *     that is a multiple of 32 (pqlk_ld()); pad columns must be zero on input
 *     and are written as zero on output.
 */
#ifndef PQLK_H
#define PQLK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pqlk_stream_t;

#define PQLK_VERSION 100 /* 0.1.0 */

enum {
  PQLK_OK = 0,
  PQLK_E_NULL = 1,     /* required pointer is NULL */
  PQLK_E_SHAPE = 2,    /* dimension <= 0 or inconsistent */
  PQLK_E_RANGE = 3,    /* index / pointer argument out of range */
  PQLK_E_ALIGN = 4,    /* leading dimension not a multiple of 32 / pointer not 16-B aligned */
  PQLK_E_UNSUPPORTED = 5,
  PQLK_E_WORKSPACE = 6 /* workspace too small */
};

int pqlk_version(void);
const char* pqlk_strerror(int rc);

/* Leading dimension used for a matrix with `cols` logical columns: round up to 32 floats (128 B). */
int64_t pqlk_ld(int64_t cols);

/* ------------------------------------------------------------------------------------------------
 * Replay ring  (reference: pql/replay/simple_replay.py:4-104, pql/algo/pql_p_learner.py:32-37,49-50,66-85)
 *
 * Storage is one array of fixed-stride records instead of the reference's five SoA tensors:
 *     record = [ obs(O) pad4 | next_obs(O) pad4 | action(A) pad4 | reward, done(0.0/1.0), 0, 0 | zero pad ]
 * (each field starts on a 16-B boundary so records move with dwordx4 accesses for any O, A).
 * rec_ld = pqlk_replay_rec_ld(O, A) floats (multiple of 32 => every record starts on a 128-B line, so
 * a random sample touches ceil(rec_bytes/128) HBM lines instead of ~10 for the SoA layout).
 * The obs-only ring of the P-learner is the same thing with A = -1 (record = obs, rec_ld = pqlk_ld(O)).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float* records;    /* (capacity, rec_ld) */
  int64_t capacity;  /* rows */
  int32_t obs_dim;   /* O */
  int32_t act_dim;   /* A, or -1 for an obs-only ring */
  int32_t rec_ld;    /* floats per record */
  int32_t reserved;
} PqlReplayDesc;

int64_t pqlk_replay_rec_ld(int32_t obs_dim, int32_t act_dim);

/* Ring insert of m rows at row `next_p` (host-side pointer law stays in Python, it is integer
 * bookkeeping: simple_replay.py:52-83).  Row r of the source goes to (dst_start + r) for r in
 * [0, m); the caller issues one call per contiguous segment (head, then wrapped tail).
 * done is canonicalised to 0.0/1.0 (reference: `.bool()` on ingest :51, `.float()` on sample :103).
 * act/rew/next_obs/done are ignored (may be NULL) for an obs-only ring.
 * src_ld_* are the source row strides in floats (contiguous tensors: O, A, 1, O, 1). */
int pqlk_replay_insert(const PqlReplayDesc* ring, int64_t dst_start, int64_t m,
                       const float* obs, int64_t ld_obs, const float* act, int64_t ld_act,
                       const float* rew, int64_t ld_rew, const float* next_obs, int64_t ld_nobs,
                       const float* done, int64_t ld_done, pqlk_stream_t stream);

/* Plain sample gather = ReplayBuffer.sample_batch with the index vector supplied
 * (simple_replay.py:98-104): five contiguous outputs (B,O),(B,A),(B,1),(B,O),(B,1) fp32. */
int pqlk_replay_gather(const PqlReplayDesc* ring, const int64_t* idx, int64_t b,
                       float* obs, float* act, float* rew, float* next_obs, float* done,
                       pqlk_stream_t stream);

/* Fused learner gather: sample + normalize (pql/utils/common.py:139-145:
 * clamp((x-mean)/sqrt(var+eps), -5, 5); mean == NULL => identity; clamp5 = 0 => no clamp, the DDPG
 * flavour ddpg.py:124-126) + torch.cat((obs, action)) (pql/models/mlp.py:197) in one pass.
 *   x_sa   (B, ld_sa)  : [ norm(obs) | action | 0 ]          critic input            (may be NULL)
 *   xn_sa  (B, ld_sa)  : [ norm(next_obs) | untouched | 0 ]  target-critic input; the action columns
 *                        are filled later by the target actor                        (may be NULL)
 *   xn_obs (B, ld_o)   : [ norm(next_obs) | 0 ]              target-actor input      (may be NULL)
 *   rew, done (B)                                                                    (may be NULL)
 * For an obs-only ring only x_sa (obs columns + zeroed action columns) and/or xn_obs (= norm(obs)
 * with ld_o) are written (pql_p_learner.py:49-52).
 * clamp5 is a flag word: bit 0 = apply the +-5 clamp; bit 1 (PQLK_GATHER_PADS_ZERO) = the caller guarantees that the pad
 * columns [O+A, ld_sa) / [O, ld_o) of the destinations are already zero (allocated zeroed, written by nothing else), so
 * the kernel does not re-zero them on every call (11 % fewer bytes stored at cfg 5).  Launch-shape overrides for
 * tools/bench_gather.py ride in the same word, so the library keeps no tuning state between calls: bit 2 = non-temporal
 * record loads, bits 8-11 = rows in flight per wave (1, 2, 4, 8; 0 = automatic), bits 12-17 = resident waves per CU
 * (0 = automatic). */
#define PQLK_GATHER_CLAMP5 1
#define PQLK_GATHER_PADS_ZERO 2
#define PQLK_GATHER_NT_LOADS 4
#define PQLK_GATHER_NT_STORES 8   /* experiment: non-temporal stores of the main 16-B tile writes */
#define PQLK_GATHER_ROWS_IN_FLIGHT(r) (((r) & 15) << 8)
#define PQLK_GATHER_WAVES_PER_CU(w) (((w) & 63) << 12)
int pqlk_replay_gather_fused(const PqlReplayDesc* ring, const int64_t* idx, int64_t b,
                             const float* mean, const float* var, float eps, int clamp5,
                             float* x_sa, int64_t ld_sa, float* xn_sa, float* xn_obs, int64_t ld_o,
                             float* rew, float* done, pqlk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * n-step assembler  (reference: pql/replay/nstep_replay.py:6-92)
 *
 * Circular per-env window instead of the reference's five torch.cat FIFO shifts per step.
 * window: (N, nstep, win_ld) fp32, win_ld = pqlk_replay_rec_ld(O, A), same record layout as the ring.
 * `count` = number of steps pushed before this call (host integer; slot of step s is s % nstep).
 * Inputs are the (N, T, .) slabs of PQLActor.explore_env (pql_actor.py:89-93,117-121).
 * Emits, for every step s = count + t with s + 1 >= nstep, one row per env in TIME-MAJOR order
 * (row = (s - first_emit_step) * N + env; nstep_replay.py:65), to five contiguous outputs.
 * gamma_pow: nstep floats (host pointer), gamma^j computed in double, rounded to fp32 (:24).
 * Reward sum order is torch's row-sum order for n <= 5 (see oracle NStepRef._emit).
 * Returns the number of emitted rows through *rows_out (host pointer, may be NULL). */
int pqlk_nstep_push_emit(float* window, int64_t num_envs, int32_t nstep, int32_t obs_dim, int32_t act_dim,
                         int64_t count, int64_t t_steps,
                         const float* obs, const float* act, const float* rew, const float* next_obs,
                         const float* done, const float* gamma_pow /*host*/,
                         float* o_obs, float* o_act, float* o_rew, float* o_next_obs, float* o_done,
                         int64_t* rows_out /*host*/, pqlk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MLP family  (reference: pql/models/mlp.py:15-40 MLPNet, :177-179 TanhMLPPolicy, :186-203 DoubleQ,
 * :244-267 DistributionalDoubleQ; backward = torch autograd of the same)
 *
 * One descriptor covers `n_nets` structurally identical nets evaluated on a shared input
 * (1 = actor, 2 = twin critic).  Parameters live in ONE flat fp32 arena:
 *   for net in 0..n_nets-1: for layer in 0..n_layers-1:
 *       W  (out_l, pqlk_ld(in_l))  row-major, y = x W^T + b  (same (out,in) orientation as nn.Linear)
 *       b  (pqlk_ld(out_l))
 * Gradients, Adam moments and Polyak targets use the same layout, so the optimiser is one pass over
 * the arena.  Hidden activation is ELU(alpha=1).
 * ---------------------------------------------------------------------------------------------- */
#define PQLK_MAX_LAYERS 8

enum { PQLK_ACT_NONE = 0, PQLK_ACT_TANH = 1, PQLK_ACT_TANH_NOISE = 2 };

typedef struct {
  int32_t n_layers;                 /* number of Linear layers (hidden + 1) */
  int32_t n_nets;                   /* 1 or 2 */
  int32_t dims[PQLK_MAX_LAYERS + 1]; /* in, h1, ..., out */
} PqlMlpDesc;

int64_t pqlk_mlp_param_floats(const PqlMlpDesc* d);                       /* arena size, all nets */
int64_t pqlk_mlp_net_stride(const PqlMlpDesc* d);                         /* floats between nets */
int pqlk_mlp_layer_offsets(const PqlMlpDesc* d, int32_t layer, int64_t* w_off, int64_t* b_off); /* within a net */
int64_t pqlk_mlp_acts_floats(const PqlMlpDesc* d, int64_t b);             /* activation stash, all nets */
int pqlk_mlp_act_offset(const PqlMlpDesc* d, int64_t b, int32_t net, int32_t layer, int64_t* off, int64_t* ld);
int64_t pqlk_mlp_bwd_ws_floats(const PqlMlpDesc* d, int64_t b, int32_t splits); /* backward workspace */

/* Fragment-ordered copy of the hidden layers' weights used by the fused forward path:
 *   packed[net][layer < L-1][tile n/32][k/8][lane (r,h)][j] = W[32 tile + r][8 (k/8) + 4 h + j]
 * pqlk_mlp_packed_floats() is 0 when the descriptor cannot take the fused path (a hidden width not a multiple of
 * 32, or wider than 636 so that two 32-row LDS activation buffers exceed 160 KB).  Call pqlk_mlp_pack() whenever
 * the arena changes (after the optimiser / Polyak step); it is one tiny launch per hidden layer. */
int64_t pqlk_mlp_packed_floats(const PqlMlpDesc* d);
int pqlk_mlp_pack(const PqlMlpDesc* d, const float* params, float* packed, pqlk_stream_t stream);

/* Forward.  x: (B, ldx) with ldx >= pqlk_ld(dims[0]).  acts: stash of every layer's post-activation
 * output (layer l of net n at pqlk_mlp_act_offset).  The last layer's output gets `out_act`:
 *   NONE       : logits / Q values
 *   TANH       : TanhMLPPolicy (mlp.py:179)
 *   TANH_NOISE : tanh, then target-policy smoothing a' = clamp(a + clamp(noise_std*draw, +-noise_clip), +-1)
 *                (pql/utils/noise.py:19-27 via pql_v_learner.py:63-71); draw: (B, dims[L]) contiguous N(0,1).
 * If out2 != NULL (n_nets must be 1) the final output is ALSO written to out2 with row stride ld_out2
 * (used to drop the actor's action into the action columns of a critic input: torch.cat for free).
 * packed != NULL (from pqlk_mlp_pack of the SAME params) selects the fused path: all hidden layers in one launch with
 * the activations of a 32- or 64-row tile resident in LDS; hidden activations are identical to the per-layer path (same
 * accumulation order).  An output layer of at most 32 columns runs inside the same launch with its reduction split over
 * the block's waves (a reassociation of the same products: ~1e-7 relative to the per-layer result).
 * stash_all = 0 on that path writes only what a forward-only caller needs: the output block, plus the last hidden layer
 * when the output layer is NOT fused; the other activation blocks are left untouched.  Backward needs stash_all = 1.  On the fused path columns [dims[0], ldx) of x are IGNORED (masked to zero while the
 * tile is staged), so x may alias a wider matrix; the per-layer path requires them to be zero.
 * stash_all = PQLK_STASH_OUTPUT_ONLY (2): forward-only call where `acts` IS the output block, (n_nets, B, pqlk_ld(out)) floats, and
 * nothing else is written -- the target policy's actions for the next K learner steps come from ONE launch over K x B rows
 * (the policy is fixed between two hand-offs, pql_v_learner.py:62-71,117-122) without a K x B activation stash.  Fused hidden stack
 * + fused output layer only (PQLK_E_UNSUPPORTED otherwise). */
#define PQLK_STASH_OUTPUT_ONLY 2
int pqlk_mlp_forward(const PqlMlpDesc* d, const float* params, const float* packed, int32_t stash_all,
                     const float* x, int64_t ldx, int64_t b,
                     int32_t out_act, const float* draw, float noise_std, float noise_clip,
                     float* acts, float* out2, int64_t ld_out2, pqlk_stream_t stream);

/* Backward.  dy: (n_nets, B, pqlk_ld(out)) gradient w.r.t. the last layer's PRE-activation output
 * (loss kernels below produce exactly that).  grads (arena layout) is overwritten with the full
 * parameter gradient when != NULL (deterministic split-batch partial sums through `ws`).
 * dx (B, ld_dx) receives the input gradient summed over nets when != NULL; if dx_tanh_of != NULL the
 * columns [dx_col0, dx_col0 + dx_cols) are multiplied by (1 - a^2) with a = dx_tanh_of (B, ld_tanh)
 * and ONLY those columns are written, compacted to column 0 of dx (DPG chain through the actor's
 * tanh, pql_p_learner.py:55-58). */
int pqlk_mlp_backward(const PqlMlpDesc* d, const float* params, const float* x, int64_t ldx, int64_t b,
                      const float* acts, const float* dy, float* grads, int32_t splits,
                      float* dx, int64_t ld_dx, int32_t dx_col0, int32_t dx_cols,
                      const float* dx_tanh_of, int64_t ld_tanh,
                      float* ws, int64_t ws_floats, pqlk_stream_t stream);

/* Same backward, with clip_grad_norm_'s first half folded into its last pass (pql_v_learner.py:128): the kernel that sums
 * the split slabs into `grads` also leaves per-block partials of sum(g^2) in sumsq_part[0, pqlk_mlp_norm_parts(d)) (room for
 * 2048 floats) and increments the optimiser's device step counter.  Follow with pqlk_adamw_polyak_fused(prenorm =
 * pqlk_mlp_norm_parts(d)).  Same gradient as pqlk_mlp_backward; the norm's partial sums are grouped differently from
 * pqlk_clip_adamw_polyak*'s own pass (last-bit differences in the clip factor).  Not for data parallel, where the gradient
 * all-reduce sits between backward and the norm. */
int pqlk_mlp_backward_norm(const PqlMlpDesc* d, const float* params, const float* x, int64_t ldx, int64_t b,
                           const float* acts, const float* dy, float* grads, int32_t splits,
                           float* dx, int64_t ld_dx, int32_t dx_col0, int32_t dx_cols,
                           const float* dx_tanh_of, int64_t ld_tanh,
                           float* ws, int64_t ws_floats, float* sumsq_part, int32_t* step_dev, pqlk_stream_t stream);
int32_t pqlk_mlp_norm_parts(const PqlMlpDesc* d);

/* Backward of a twin critic with scalar Q heads whose head pass forms dL/dQ itself: TD target y = r + (1-d) gamma^n
 * min(Q1', Q2') and the twin MSE loss of pql_v_learner.py:104-108 (replaces pqlk_td_mse_loss + pqlk_mlp_backward[_norm]: one
 * launch less, dL/dQ never goes through memory).  acts / acts_target: activation stashes of the online and the target
 * forward (only the target's head output is read).  loss_part receives pqlk_td_head_loss_parts(d, b) partial sums of
 * (Q - y)^2 over both nets: fold with scale 1 / b (pqlk_adamw_polyak_fused's loss_part).  sumsq_part / step_dev: both
 * non-NULL = as pqlk_mlp_backward_norm, both NULL = as pqlk_mlp_backward.  PQLK_E_UNSUPPORTED unless n_nets == 2, one
 * output, and the last hidden width is <= 1024. */
int pqlk_mlp_backward_td(const PqlMlpDesc* d, const float* params, const float* x, int64_t ldx, int64_t b,
                         const float* acts, const float* acts_target, const float* rew, const float* done, float gamma_n,
                         float* loss_part, float* grads, int32_t splits, float* ws, int64_t ws_floats,
                         float* sumsq_part, int32_t* step_dev, pqlk_stream_t stream);
int32_t pqlk_td_head_loss_parts(const PqlMlpDesc* d, int64_t b);   /* 0: this layout cannot take pqlk_mlp_backward_td */

/* The same scalar-head step with the head's backward inside the critic's FORWARD launch: pqlk_mlp_forward_td is
 * pqlk_mlp_forward(packed, stash_all = 1, PQLK_ACT_NONE) whose blocks, having formed Q for their rows, also form the TD error
 * (pql_v_learner.py:104-108), dL/dZ of the last hidden layer, the head's dW / db partials and the loss partials while that
 * layer's activations are still in LDS -- no head-backward launch, no second read of them (and no stash of them: that block of
 * `acts` is left unwritten).  bwd_ws / splits: the workspace and split count the backward will be called with.  Follow with
 * pqlk_mlp_backward_td_tail (same arguments as pqlk_mlp_backward_td minus the TD inputs): layers n_layers-2 .. 0 and the slab
 * reduction.  loss_part receives pqlk_td_forward_loss_parts(d, b) partials (0: this layout / batch cannot take the path: no
 * fused hidden stack, not a twin scalar head, or no room beside the last hidden layer in LDS). */
int32_t pqlk_td_forward_loss_parts(const PqlMlpDesc* d, int64_t b);
int pqlk_mlp_forward_td(const PqlMlpDesc* d, const float* params, const float* packed, const float* x, int64_t ldx, int64_t b,
                        float* acts, const float* acts_target, const float* rew, const float* done, float gamma_n,
                        float* loss_part, float* bwd_ws, int64_t bwd_ws_floats, int32_t splits, pqlk_stream_t stream);
int pqlk_mlp_backward_td_tail(const PqlMlpDesc* d, const float* params, const float* x, int64_t ldx, int64_t b,
                              const float* acts, float* grads, int32_t splits, float* ws, int64_t ws_floats,
                              float* sumsq_part, int32_t* step_dev, pqlk_stream_t stream);

/* Data-parallel buckets (the gradient all-reduce "overlapped with the last dW GEMMs"; the reference has no
 * counterpart, its learners are single-GPU -- pql_v_learner.py:110-113 is `backward(); step()`).  Layers layer_hi >= l >=
 * layer_lo of the backward above (dW_l, db_l, dX_l) and then the split-slab reduction of exactly those layers into `grads`:
 * what this call leaves in grads is final, so its all-reduce can be issued while the calls for the layers below still run.
 * Calls walk the layers downwards (first: layer_hi = n_layers - 1, last: layer_lo = 0) over the SAME workspace; their union
 * leaves in grads bit for bit what one pqlk_mlp_backward / pqlk_mlp_backward_td call leaves.  dy is read by the call holding
 * the last layer; give acts_target / rew / done / loss_part (all four or none) and that call forms the TD error in the head
 * pass as pqlk_mlp_backward_td does (dy is then ignored). */
int pqlk_mlp_backward_layers(const PqlMlpDesc* d, const float* params, const float* x, int64_t ldx, int64_t b,
                             const float* acts, const float* dy, const float* acts_target, const float* rew,
                             const float* done, float gamma_n, float* loss_part, float* grads, int32_t splits,
                             float* ws, int64_t ws_floats, int32_t layer_hi, int32_t layer_lo, pqlk_stream_t stream);

/* The learners' random draws with torch's own numbers (csrc/philox.hip; reference draws: pql/replay/simple_replay.py:87,
 * pql/utils/noise.py:20-21, pql/algo/pql_p_learner.py:49).  `chunks` consecutive learner steps in ONE launch: per step n_idx
 * int64 indices uniform on [0, range) into idx[chunk][n_idx], then n_normal standard normals into normal[chunk][n_normal]
 * (either part may be absent: n = 0 / NULL).  Chunk c holds exactly what
 *     torch.randint(range, (n_idx,), generator=g);  torch.empty(n_normal).normal_(generator=g)
 * return on a device generator g with this seed and Philox offset
 *     base_offset + (step - base_step + c) * (pqlk_philox_increment(n_idx) + pqlk_philox_increment(n_normal)),
 * step = step_dev[0] (a device step counter; NULL: base_step).  range < 2^28 (torch draws 64-bit values above).  contract:
 * Box-Muller's affine map of the angle as one fma (how torch's build of rocRAND compiles it) or as mul + add; which one
 * matches is checked against torch on the device by pql_amd/utils/rng.py, which falls back to ATen launches otherwise. */
int pqlk_philox_draws(uint64_t seed, int64_t base_offset, int32_t base_step, const int32_t* step_dev, int64_t range,
                      int64_t* idx, int64_t n_idx, float* normal, int64_t n_normal, int32_t chunks, int32_t contract,
                      pqlk_stream_t stream);
/* Philox offset increment of one torch draw of `numel` elements (4 values per counter block, at most 2048 x 256 threads). */
int64_t pqlk_philox_increment(int64_t numel);

/* DPG backward through the frozen twin critic (pql_p_learner.py:55-58): the input gradient of pqlk_mlp_backward's
 * (grads = NULL, dx + dx_tanh_of) form.  With scalar Q heads the gradient of min(Q1, Q2) reaches one net per sample, so the
 * samples are first partitioned by owning net and the dX chain runs over compact rows -- half the MFMA work, each sample's
 * values bit for bit those of the dense chain; other critics take the dense chain.  dx (B, ld_dx) is fully overwritten
 * (columns >= dx_cols with zero).  ws >= pqlk_dpg_backward_ws_floats(d, b) floats. */
int64_t pqlk_dpg_backward_ws_floats(const PqlMlpDesc* d, int64_t b);
int pqlk_dpg_critic_backward(const PqlMlpDesc* d, const float* params, const float* x, int64_t ldx, int64_t b,
                             const float* acts, const float* dy, float* dx, int64_t ld_dx, int32_t dx_col0, int32_t dx_cols,
                             const float* dx_tanh_of, int64_t ld_tanh, const uint8_t* owner /* (B) from pqlk_dpg_loss_owner, or NULL */,
                             float* ws, int64_t ws_floats, pqlk_stream_t stream);

/* The P-learner's whole backward through the frozen critic in four launches (round 4; pql_p_learner.py:54-58: actor(obs) ->
 * -critic.get_q_min(obs, a).mean() -> .backward()).  Twin scalar-head critics whose forward is fused (hidden stack + head):
 *   pqlk_mlp_forward_qc        the critic's forward (as pqlk_mlp_forward, out_act none) that ALSO leaves the head outputs compact,
 *                              qc (2, B) = Q1 | Q2 (DoubleQ.get_q1_q2, mlp.py:197-199).  With x2 != NULL the input is
 *                              [ x[:, :x2_col0] | x2[:, :dims[0] - x2_col0] ] -- `torch.cat((state, action), dim=1)` (mlp.py:197) read
 *                              from where the two halves already lie (the actor's input tile, the actor's output block): the
 *                              P-learner's gather then writes the observations once instead of twice.  x2_col0 % 4 == 0.
 *   pqlk_dpg_backward_fused    (1) DPG loss partials, partition of the batch by the net that attained min(Q1, Q2) and the head's dX
 *                              over compact rows in ONE launch (replaces pqlk_dpg_loss_owner + two launches of
 *                              pqlk_dpg_critic_backward), (2) the compact dX GEMMs, (3) the action slice through tanh' TOGETHER
 *                              WITH the actor's last-layer backward (dX, dW / db partials per 32-row tile: replaces the actor
 *                              backward's head launch).  loss_part[pqlk_dpg_fused_loss_parts()] = partial sums of min(Q1, Q2):
 *                              fold with scale -1 / b.  dz_a (B, ld_dz): columns [0, A) of every row are written (A = the actor's
 *                              output width = the critic input's action columns [dx_col0, dx_col0 + A)), pad columns untouched.
 *   pqlk_mlp_backward_tail     the rest of the actor's backward (layers below the head, slab reduction incl. the head fold,
 *                              optional squared-norm partials as pqlk_mlp_backward_norm) over the SAME actor workspace;
 *                              head_parts = pqlk_dpg_fused_head_parts(b), head_parts_dev = critic_ws +
 *                              pqlk_dpg_fused_mn_offset(critic, b) floats (a device int[4]; only the partial tiles in use exist).
 * pqlk_dpg_fused_ok = 1 when this critic / actor pair can take the path (else use pqlk_dpg_loss_owner +
 * pqlk_dpg_critic_backward + pqlk_mlp_backward); every entry returns PQLK_E_UNSUPPORTED otherwise.  pqlk_dpg_backward_fused does not
 * read x (the critic's input): it may be NULL.
 * critic_ws >= pqlk_dpg_backward_ws_floats(critic, b), actor_ws >= pqlk_mlp_bwd_ws_floats(actor, b, splits) floats. */
int32_t pqlk_dpg_fused_ok(const PqlMlpDesc* critic, const PqlMlpDesc* actor, int64_t b);
int32_t pqlk_dpg_fused_loss_parts(void);
int32_t pqlk_dpg_fused_head_parts(int64_t b);
int64_t pqlk_dpg_fused_mn_offset(const PqlMlpDesc* critic, int64_t b);
int pqlk_mlp_forward_qc(const PqlMlpDesc* d, const float* params, const float* packed, int32_t stash_all, const float* x,
                        int64_t ldx, const float* x2 /* or NULL */, int64_t ldx2, int32_t x2_col0, int64_t b, float* acts, float* qc,
                        pqlk_stream_t stream);
int pqlk_dpg_backward_fused(const PqlMlpDesc* critic, const float* params, const float* x, int64_t ldx, int64_t b, const float* acts,
                            const float* qc, float* dz_a, int64_t ld_dz, int32_t dx_col0, const float* a_out, int64_t ld_tanh,
                            float* loss_part, float* critic_ws, int64_t critic_ws_floats, const PqlMlpDesc* actor,
                            const float* actor_params, const float* actor_acts, float* actor_ws, int64_t actor_ws_floats,
                            int32_t actor_splits, pqlk_stream_t stream);
int pqlk_mlp_backward_tail(const PqlMlpDesc* d, const float* params, const float* x, int64_t ldx, int64_t b, const float* acts,
                           float* grads, int32_t splits, float* ws, int64_t ws_floats, float* sumsq_part, int32_t* step_dev,
                           int32_t head_parts, const int32_t* head_parts_dev, pqlk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Losses.  Each writes dy (2, B, ld) for pqlk_mlp_backward and one scalar loss (device), via per-block
 * partials in `scratch` (>= 1024 floats) reduced in fixed order.  The scalar lands in
 * loss_out[slot_dev[0] % ring_len] when slot_dev != NULL (a device int32 counter, normally the optimiser's
 * step counter: gives a graph-replay-safe ring of the last `ring_len` losses, replacing the reference's
 * per-step `.item()` host sync, pql_v_learner.py:111), else in loss_out[0].  loss_out = NULL leaves the per-block
 * partials in scratch[0, pqlk_loss_parts(b, k)) for pqlk_adamw_polyak_fused to fold (one launch less).
 * ---------------------------------------------------------------------------------------------- */

/* TD target + twin MSE (pql_v_learner.py:104-108): y = r + (1-d) * gamma_n * min(qt1, qt2);
 * loss = mean((q1-y)^2) + mean((q2-y)^2).  q, qt: (2, B, ld) with the value in column 0. */
int pqlk_td_mse_loss(const float* q, const float* qt, int64_t ld, const float* rew, const float* done,
                     float gamma_n, int64_t b, float* dy, float* loss_out, const int32_t* slot_dev, int32_t ring_len,
                     float* scratch, pqlk_stream_t stream);

/* C51: softmax of target logits, categorical projection x2 (pql/utils/distl_util.py:4-20), elementwise
 * min (pql_v_learner.py:83-102), softmax of current logits, twin BCE (mean over B*K, log clamped at
 * -100 like torch) and its gradient w.r.t. the current logits.  logits: (2, B, ld), K <= 64 atoms;
 * support = DistributionalDoubleQ.z_atoms (mlp.py:253), passed in so its fp32 values are torch.linspace's.
 * If proj_out != NULL the (B, K) target pmf is also stored (tests). */
int pqlk_c51_bce_loss(const float* logits, const float* logits_t, int64_t ld, int32_t k,
                      const float* rew, const float* done, const float* support /*(K) z atoms*/, float gamma_n,
                      float v_min, float v_max, int64_t b, float* dy, float* loss_out, const int32_t* slot_dev,
                      int32_t ring_len, float* proj_out, float* scratch, pqlk_stream_t stream);

/* Stand-alone projection = projection() of distl_util.py:4-20 on a given pmf (B, K) contiguous. */
int pqlk_c51_project(const float* p, const float* rew, const float* done, const float* support, float gamma_n,
                     float v_min, float v_max, int32_t k, int64_t b, float* out, pqlk_stream_t stream);

/* DPG actor loss (pql_p_learner.py:56-57): L = -mean(min(Q1,Q2)); dy = dL/dQ (ties split evenly, as
 * torch.min's backward).  k == 1: q (2,B,ld) scalar heads.  k > 1: logits of the distributional critic,
 * Q_i = sum softmax(logits_i) * z (mlp.py:256-260), dy = gradient w.r.t. the logits. */
int pqlk_dpg_loss_owner(const float* q, int64_t ld, int32_t k, const float* support, int64_t b, float* dy, float* loss_out,
                        const int32_t* slot_dev, int32_t ring_len, float* scratch,
                        uint8_t* owner /* (B), k == 1: bit 0 / bit 1 = the gradient of min(Q1, Q2) reaches net 0 / net 1 */,
                        pqlk_stream_t stream);
int pqlk_dpg_loss(const float* q, int64_t ld, int32_t k, const float* support /*(K), NULL when k == 1*/, int64_t b,
                  float* dy, float* loss_out, const int32_t* slot_dev, int32_t ring_len, float* scratch,
                  pqlk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser: clip_grad_norm_ + AdamW (+ Polyak) over flat arenas
 * (pql_v_learner.py:124-133, pql_p_learner.py:87-96, pql/utils/torch_util.py:9-12).
 *   g *= grad_scale   (1/world_size after a data-parallel sum all-reduce; 1.0 otherwise)
 *   total = ||g||_2 ; g *= min(1, max_norm / (total + 1e-6))   (max_norm <= 0 disables clipping)
 *   p *= 1 - lr*wd ; m += (g-m)(1-b1) ; v = b2 v + (1-b2) g^2
 *   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 *   target = p*tau + target*(1-tau)                              (target may be NULL)
 * step_dev: device int32 counter t, incremented by the call (graph-replay safe).
 * scratch: >= 2048 floats.  gnorm_out (device, may be NULL) receives the pre-clip norm.
 * ---------------------------------------------------------------------------------------------- */
int pqlk_clip_adamw_polyak(float* p, float* g, float* m, float* v, float* target, int64_t n, float grad_scale,
                           float max_norm, float lr, float b1, float b2, float eps, float wd, float tau,
                           int32_t* step_dev, float* gnorm_out, float* scratch, pqlk_stream_t stream);

/* Same update for the arena of the MLP `d`, which ALSO refreshes the fragment-ordered weight copies of the fused
 * forward path while each new value is in a register (packed_p for the parameters, packed_t -- may be NULL -- for the
 * Polyak target), replacing the separate pqlk_mlp_pack launches. */
int pqlk_clip_adamw_polyak_pack(const PqlMlpDesc* d, float* p, float* g, float* m, float* v, float* target,
                                float* packed_p, float* packed_t, float grad_scale, float max_norm, float lr, float b1,
                                float b2, float eps, float wd, float tau, int32_t* step_dev, float* gnorm_out,
                                float* scratch, pqlk_stream_t stream);

/* The optimiser launch of a fused learner step (pql_v_learner.py:124-133 + :109-111, pql_p_learner.py:87-96): the same
 * update as pqlk_clip_adamw_polyak_pack (packed_p may be NULL: nothing to re-pack), plus
 *   prenorm > 0  : `scratch` already holds that many squared-norm partials and step_dev is already incremented (both left
 *                  by pqlk_mlp_backward_norm; prenorm = pqlk_mlp_norm_parts(d)), so no separate norm launch;
 *   loss_part    : per-block loss partials left in a loss kernel's scratch (pqlk_td_mse_loss / pqlk_c51_bce_loss /
 *                  pqlk_dpg_loss called with loss_out = NULL; count = pqlk_loss_parts(b, k)): block 0 folds them, times
 *                  loss_scale (1/B, 1/(B K), -1/B), into loss_ring[(t - 1) % ring_len], t = the incremented step --
 *                  the slot and the bits the stand-alone fold writes.  NULL = no fold. */
int pqlk_adamw_polyak_fused(const PqlMlpDesc* d, float* p, float* g, float* m, float* v, float* target,
                            float* packed_p, float* packed_t, float grad_scale, float max_norm, float lr, float b1,
                            float b2, float eps, float wd, float tau, int32_t* step_dev, float* gnorm_out,
                            float* scratch, int32_t prenorm, const float* loss_part, int32_t loss_parts,
                            float loss_scale, float* loss_ring, int32_t ring_len, pqlk_stream_t stream);
/* Number of per-block partials a loss entry point leaves in `scratch` (k = atoms; 1 for scalar heads). */
int32_t pqlk_loss_parts(int64_t b, int32_t k);

/* soft_update alone (torch_util.py:9-12): target = cur*tau + target*(1-tau). */
int pqlk_polyak(float* target, const float* cur, int64_t n, float tau, pqlk_stream_t stream);

/* RunningMeanStd.update batch moments (torch_util.py:77-81): mean and UNBIASED variance over rows of
 * x (N, ldx) for `cols` columns -> mean_out, var_out (cols).  The Chan merge (:87-103) stays on host.
 * scratch: >= 64 * cols * 3 floats (per-chunk count / mean / M2). */
int pqlk_batch_moments(const float* x, int64_t ldx, int64_t n, int32_t cols, float* mean_out, float* var_out,
                       float* scratch, pqlk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SAC on the same kernels (reference pql/algo/sac.py:138-156, pql/models/mlp.py:144-174,
 * pql/utils/torch_util.py:15-65).
 *
 * Squashed-Gaussian policy head.  y (B, ld_y) = [ mu(A) | log_std(A) | pad ] is the policy MLP's output
 * (pqlk_mlp_forward with PQLK_ACT_NONE), eps (B, A) contiguous the standard-normal draw of Normal.rsample:
 *     std = exp(clamp(log_std, -5, 5));  u = mu + eps * std;  a = tanh(u)
 *     logp = sum_j [ -((u - mu)^2) / (2 std^2) - log(std) - log(sqrt(2 pi)) - 2 (log 2 - u - softplus(-2u)) ]
 * act (B, ld_act) receives a in columns [0, A) (pass a pointer into the critic's [obs | action] tile); logp (B).
 * eps == NULL => the deterministic action tanh(mu) (get_actions(sample=False)); logp is not written.  A <= 64. */
int pqlk_sg_head_forward(const float* y, int64_t ld_y, const float* eps, int64_t b, int32_t act_dim,
                         float* act, int64_t ld_act, float* logp, pqlk_stream_t stream);

/* Backward of the head: dy (B, ld_y) <- d loss / d [mu | log_std] (pad columns written as zero) given
 * da (B, ld_da) = d loss / d a and d loss / d logp = glp_scale * exp(*log_alpha) for every row
 * (log_alpha: device scalar, NULL => factor 1).  dy feeds pqlk_mlp_backward of the policy MLP. */
int pqlk_sg_head_backward(const float* y, int64_t ld_y, const float* eps, const float* act, int64_t ld_act,
                          const float* da, int64_t ld_da, const float* log_alpha, float glp_scale,
                          int64_t b, int32_t act_dim, float* dy, pqlk_stream_t stream);

/* Entropy term of the SAC target (sac.py:140-142): column 0 of each target-critic net,
 * qt[n * net_stride + r * ld] -= exp(*log_alpha) * logp[r]; pqlk_td_mse_loss then forms
 * r + (1 - d) gamma^n (min(q1, q2) - alpha logp). */
int pqlk_sac_entropy_shift(float* qt, int64_t ld, int64_t net_stride, int32_t n_nets, const float* logp,
                           const float* log_alpha, int64_t b, pqlk_stream_t stream);

/* Temperature terms (sac.py:146-156), one launch: with m = mean(logp) and alpha = exp(*log_alpha),
 *   grad_out[0] = alpha_loss_out[0] = alpha * (-m - target_entropy)   (the alpha loss and its gradient w.r.t. log_alpha
 *                                                                      have the same value; either may be NULL)
 *   actor_loss_ring[*slot_dev % ring_len] += alpha * m                (completes mean(alpha logp - Q) after pqlk_dpg_loss) */
int pqlk_sac_alpha_terms(const float* logp, int64_t b, const float* log_alpha, float target_entropy,
                         float* grad_out, float* alpha_loss_out, float* actor_loss_ring,
                         const int32_t* slot_dev, int32_t ring_len, pqlk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm1d + ELU blocks of the CrossQ critic (reference pql/models/mlp.py:15-24,224-241,
 * pql/algo/crossQ.py:144-166).  The Linear layers run as one-layer PqlMlpDesc calls of pqlk_mlp_forward / _backward; the
 * batch statistics come from pqlk_batch_moments (mean, UNBIASED variance over the m rows).
 *
 * Forward: y = ELU(z * w + b), w = gamma * invstd, b = beta - mean * w, invstd = 1/sqrt(var + eps) (ATen's order).
 * training != 0: mean / var = the batch statistics (biased variance = var_unbiased * (m-1)/m) and, when running_mean is
 * not NULL, running_{mean,var} <- (1 - momentum) * running + momentum * (mean, var_unbiased) (torch: momentum 0.1,
 * eps 1e-5).  training == 0: the running statistics normalise and nothing is updated.  z, y: (m, ld), cols <= ld. */
int pqlk_bn_elu_forward(const float* z, int64_t ld, int64_t m, int32_t cols, const float* mean, const float* var_unbiased,
                        const float* gamma, const float* beta, float eps, int32_t training, float momentum,
                        float* running_mean, float* running_var, float* y, pqlk_stream_t stream);

/* Backward of the training-mode block: dy = d loss / d y (post-ELU), y and z as stored by the forward ->
 * dz = gamma * invstd * (g - mean_rows(g) - xhat * mean_rows(g * xhat)), g = dy * ELU'(pre) with ELU' taken from y,
 * dgamma = sum_rows(g * xhat), dbeta = sum_rows(g) (either may be NULL).  dz may alias dy.
 * scratch: >= 128 * cols floats (64 row chunks x 2 column sums, folded in chunk order: deterministic). */
int pqlk_bn_elu_backward(const float* dy, const float* y, const float* z, int64_t ld, int64_t m, int32_t cols,
                         const float* mean, const float* var_unbiased, const float* gamma, float eps, float* dz,
                         float* dgamma, float* dbeta, float* scratch, pqlk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Synthetic vectorised environment step (the Isaac-Gym stand-in of BASELINE.json; not a reference component).
 * Counter-based: outputs depend only on (seed, env_offset + env, t, column), so shards of the env axis reproduce
 * slices of the global env.  next_obs ~ N(0,1) (N, obs_dim); reward = N(0,1) - 0.1 mean(action^2) (N);
 * done ~ Bernoulli(p_done) as bytes (N).  Same arithmetic as pql_amd/envs/synthetic.py's torch-op definition. */
int pqlk_synth_env_step(int64_t n, int32_t obs_dim, int32_t act_dim, uint32_t seed, uint32_t env_offset, uint32_t t,
                        float p_done, const float* action, float* next_obs, float* reward, uint8_t* done,
                        pqlk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-env-step bookkeeping of the rollout in one launch (pql_actor.py:104-114 slab writes, :129-135 update_tracker,
 * common.py:195-202 handle_timeout): column t of the (N, horizon, .) trajectory slabs <- (obs, action, reward, next_obs,
 * done * !truncated); cur_return += reward, cur_length += 1; the finished envs' values are appended IN ENV ORDER to the two
 * moving windows of win_len floats (the last win_len of them when more finish in one step, like deque.extend) behind the
 * windows' device write pointers, and their accumulators reset.  done / truncated are bytes (truncated may be NULL). */
int pqlk_rollout_step(int64_t n, int32_t obs_dim, int32_t act_dim, int32_t horizon, int32_t t, const float* obs,
                      const float* action, const float* next_obs, const float* reward, const uint8_t* done,
                      const uint8_t* truncated, float* slab_obs, float* slab_act, float* slab_rew, float* slab_nobs,
                      float* slab_done, float* cur_return, float* cur_length, float* win_return, float* win_length,
                      int64_t* win_return_ptr, int64_t* win_length_ptr, int32_t win_len, pqlk_stream_t stream);

/* The remaining per-env-step elementwise work of the rollout, one launch each (the reference issues 12 + 4 + 4 ATen launches):
 * RunningMeanStd.update_from_moments (pql/utils/torch_util.py:91-103): Chan merge of one batch's moments into the running
 * ones, op for op in fp32 (count / batch_count / total = the python scalars rounded to fp32 as torch rounds them), written to
 * mean_out / var_out (may alias mean / var); RunningMeanStd.normalize (:83-85): (x - mean) / sqrt(var + eps) of a contiguous
 * (rows, cols) matrix into rows of stride ld_out (columns past cols untouched), IEEE division, no clamp; add_normal_noise / add_mixed_normal_noise (pql/utils/noise.py:19-41):
 * out = clamp(act + draw * sigma, lo, hi) with sigma = std_rows[row] (one per env) when std_rows != NULL, else std_scalar. */
int pqlk_rms_merge(const float* mean, const float* var, const float* batch_mean, const float* batch_var, float count,
                   float batch_count, float total, int32_t cols, float* mean_out, float* var_out, pqlk_stream_t stream);
int pqlk_rms_normalize(const float* x, int64_t rows, int32_t cols, const float* mean, const float* var, float eps, float* out,
                       int64_t ld_out, pqlk_stream_t stream);
int pqlk_action_noise(const float* act, const float* draw, const float* std_rows, float std_scalar, int64_t rows, int32_t cols,
                      float lo, float hi, float* out, pqlk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PQLK_H */
