/*
 * rl4co_amd.h — C-ABI of the MI355X-native autoregressive rollout engine.
 *
 * This is the drop-in boundary for ONE hot path of ai4co/rl4co: the batched
 * TSP/CVRP environment step + AttentionModel decode loop + tour-length reward.
 * Every entry point below names the reference function(s) it replaces
 * (paths relative to the reference checkout, `rl4co/...:line`).
 *
 * Conventions
 *   - all pointers are DEVICE pointers into caller-owned buffers (the Python
 *     boundary hands in torch tensors' data_ptr()); the library allocates
 *     nothing persistent and keeps no threads;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns an int status (RL4CO_OK = 0) and never throws;
 *   - the reference raises Python `assert`s from inside the loop (one host
 *     sync each). Here those conditions are recorded as sticky bits
 *     OR-ed into the caller's `int32_t* err` word, which the host checks ONCE
 *     per rollout and converts back into the reference's assertion messages;
 *   - masks are uint8 with 1 = feasible (torch.bool storage), indices int64,
 *     payload float32 unless a dtype tag says otherwise.
 */
#ifndef RL4CO_AMD_H
#define RL4CO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ABI version ----------------------------------------------------------
 * Bumped whenever an exported signature or struct layout changes. A binding compares rl4co_abi_version() (what the
 * loaded library was built with) against the RL4CO_ABI_VERSION of the header it was written for and refuses to run on a
 * mismatch: a changed argument list under an unchanged symbol name still links (r05: rl4co_attn_bwd_* gained `out` in
 * second position — a caller built for the old list would pass dout as out and lse as dout).
 *   6: rl4co_attn_bwd_{bf16,f16} take the forward's `out`; rl4co_abi_version() itself.
 *   7: rl4co_am_decode_args / rl4co_am_teacher_args end in the context tables' dtype and strides (ctx_dtype ...).
 *   8: the 16 rl4co_<op>_bf16 / rl4co_<op>_f16 pairs are ONE rl4co_<op>(int dtype, ...) each. */
#define RL4CO_ABI_VERSION 12

/* ---- status codes ------------------------------------------------------ */
#define RL4CO_OK 0
#define RL4CO_ERR_ARG 1     /* bad argument (null pointer, size out of range)  */
#define RL4CO_ERR_HIP 2     /* HIP runtime error, see rl4co_last_error()       */
#define RL4CO_ERR_UNSUPPORTED 3

/* ---- sticky error bits written to `err` (device int32) ------------------ */
#define RL4CO_EBIT_NAN_LOGIT 1   /* nn/attention.py:295-296 "Logits contain NaNs"        */
#define RL4CO_EBIT_INFEASIBLE 2  /* utils/decoding.py:393,409 "infeasible action selected" */
#define RL4CO_EBIT_INVALID_TOUR 4 /* tsp/env.py:161, cvrp/env.py:157-163 "Invalid tour"   */
#define RL4CO_EBIT_CAPACITY 8    /* cvrp/env.py:175-177 "Used more than capacity"         */
#define RL4CO_EBIT_MAX_STEPS 16  /* constructive/base.py:236-238 max_steps exceeded       */
#define RL4CO_EBIT_NEG_INF_LOGP 32 /* utils/decoding.py:56 "Logprobs should not be -inf"  */
#define RL4CO_EBIT_DUPLICATES 64  /* op/env.py:178-181 "Duplicates"                        */
#define RL4CO_EBIT_MAX_LENGTH 128 /* op/env.py:192-194 "Max length exceeded"               */
#define RL4CO_EBIT_PRIZE 256      /* pctsp/env.py:192-201 "Total prize does not satisfy min total prize" */
#define RL4CO_EBIT_NOT_ALL_NODES 512   /* pdp/env.py:208-213 "Not visiting all nodes"                  */
#define RL4CO_EBIT_DEPOT_MIDDLE 1024   /* pdp/env.py:216-218 "Going back to depot in the middle ..."   */
#define RL4CO_EBIT_NO_PICKUP 2048      /* pdp/env.py:220-223 "Deliverying without pick-up"             */
#define RL4CO_EBIT_TW_NEGATIVE 4096    /* cvrptw/env.py:153 "Time windows must be non-negative."       */
#define RL4CO_EBIT_TW_RETURN 8192      /* cvrptw/env.py:154-157 "vehicle cannot perform service and get back to depot in time." */
#define RL4CO_EBIT_TW_DURATION 16384   /* cvrptw/env.py:158 "Service durations must be non-negative."  */
#define RL4CO_EBIT_TW_EMPTY 32768      /* cvrptw/env.py:159-161 "there are unfeasible time windows"    */
#define RL4CO_EBIT_TW_DEADLINE 65536   /* cvrptw/env.py:176-179 "vehicle cannot start service before deadline" */

/* ---- enums --------------------------------------------------------------- */
#define RL4CO_ENV_TSP 0
#define RL4CO_ENV_CVRP 1
#define RL4CO_ENV_OP 2 /* orienteering problem: STREAM / LDS / WIDE decode variants */
#define RL4CO_ENV_PCTSP 3 /* prize-collecting TSP (same row): STREAM / LDS / WIDE decode variants */
#define RL4CO_ENV_PDP 4 /* pickup and delivery (same row): STREAM / LDS / WIDE decode variants */
#define RL4CO_ENV_CVRPTW 5 /* CVRP with time windows (same row): STREAM / LDS / WIDE decode variants */

#define RL4CO_DECODE_GREEDY 0   /* utils/decoding.py:387-397 */
#define RL4CO_DECODE_SAMPLE 1   /* utils/decoding.py:399-413 */
#define RL4CO_DECODE_EVALUATE 2 /* utils/decoding.py:448-461 */

#define RL4CO_DT_F32 0
#define RL4CO_DT_BF16 1
#define RL4CO_DT_F16 2 /* IEEE half planes: the reference's default "16-mixed" regime (utils/trainer.py:57); served by the
                          fused encoder, the streaming and the multistart (MS) decode kernels and both teacher-forced
                          backward variants */

/* kernel variants of rl4co_am_decode (same results up to the documented summation tree) */
#define RL4CO_VARIANT_AUTO 0
#define RL4CO_VARIANT_STREAM 1 /* planes streamed from HBM every step, 1 wave per trajectory     */
#define RL4CO_VARIANT_LDS 2    /* bf16 planes loaded into LDS once per rollout, 4 waves / trajectory */
#define RL4CO_VARIANT_WIDE 3   /* bf16 planes streamed every step, 4 waves / trajectory (few trajectories, large N) */
#define RL4CO_VARIANT_MS 4     /* multistart on the matrix cores: one workgroup per (instance, column tile of 16
                                  starts), glimpse planes in LDS, logit-key tile in registers, two workgroups per
                                  CU; bf16 query / glimpse (pinned to the rounding-model oracle, 5e-3); every
                                  environment; auto where measured faster: TSP / PDP / PCTSP from 8 starts per
                                  instance, CVRP from 16 */

#define RL4CO_EMBED_DIM 128 /* the engine is specialised for the AM default d=128, H=8 */
#define RL4CO_NUM_HEADS 8

/* Library identification. */
const char* rl4co_version(void);
/* RL4CO_ABI_VERSION of the header the loaded library was compiled against (see above). */
int rl4co_abi_version(void);
/* Last HIP error string seen by the calling thread ("" if none). */
const char* rl4co_last_error(void);

/* --------------------------------------------------------------------------
 * a1  gather_by_index(src, idx, dim=1)            rl4co/utils/ops.py:54-66
 * out[b,k,:] = src[b, idx[b,k], :]   src [B,N,D] f32, idx [B,K] i64, out [B,K,D]
 * -------------------------------------------------------------------------- */
int rl4co_gather_by_index_f32(const float* src, const int64_t* idx, int B, int N, int D,
                              int K, float* out, int32_t* err, void* stream);

/* --------------------------------------------------------------------------
 * a2/a3  get_tour_length / TSPEnv._get_reward / CVRPEnv._get_reward
 *        rl4co/utils/ops.py:77-90, envs/routing/tsp/env.py:150-156,
 *        envs/routing/cvrp/env.py:138-147
 * out[b] = (negate ? -1 : 1) * sum_t || p[t+1 mod n] - p[t] ||_2  over the closed
 * tour p = (prepend_depot ? locs[b,0] : -) ++ locs[b, actions[b,:]]  (n = T + prepend).
 * Bit-exact restatement of ATen's CPU arithmetic: per segment
 * sqrt(fma(dy,dy,fl(dx*dx))), row sum in the 8-lane x 4-ILP cascade order of
 * aten/src/ATen/native/cpu/SumKernel.cpp.
 * locs [B_locs,N,2] f32 with trajectory b reading instance b % B_locs
 * (multistart: s-major batchify, ops.py:10-28); actions [B,T] i64.
 * -------------------------------------------------------------------------- */
int rl4co_tour_length_f32(const float* locs, const int64_t* actions, int B, int B_locs, int N,
                          int T, int prepend_depot, int negate, float* out, void* stream);
/* Same arithmetic with the horizon read ON THE DEVICE: actions is a [B,row_stride] buffer whose first
 * T = t_add + *steps_dev columns (clamped to row_stride) are the tour — steps_dev is word [0] of the decode launch's
 * steps_summary. Lets the reward of a data-dependent horizon (CVRP) go out before the rollout's single read-back and
 * inside a captured HIP graph; the row-sum association follows the real T exactly as above. */
int rl4co_tour_length_dyn_f32(const float* locs, const int64_t* actions, int B, int B_locs, int N, int row_stride,
                              const int32_t* steps_dev, int t_add, int prepend_depot, int negate, float* out,
                              void* stream);

/* --------------------------------------------------------------------------
 * a4  check_solution_validity   tsp/env.py:158-164, cvrp/env.py:149-177
 * TSP : every row of actions [B,T==N] is a permutation of 0..N-1.
 * CVRP: customers 1..N-1 appear exactly once, everything else is 0, and the
 *       running load (reset at the depot, clamped at 0) never exceeds
 *       vehicle_capacity + 1e-5.  demand [B_inst,N-1], capacity [B_inst].
 * Violations set RL4CO_EBIT_INVALID_TOUR / RL4CO_EBIT_CAPACITY in *err.
 * -------------------------------------------------------------------------- */
int rl4co_tsp_check_solution(const int64_t* actions, int B, int N, int T, int32_t* err,
                             void* stream);
int rl4co_cvrp_check_solution(const int64_t* actions, const float* demand,
                              const float* vehicle_capacity, int B, int B_inst, int N, int T,
                              int32_t* err, void* stream);

/* --------------------------------------------------------------------------
 * a6  TSPEnv._step              envs/routing/tsp/env.py:60-86
 * first = (i==0) ? action : first ; cur = action ; mask[action] = 0 ; i += 1 ;
 * done = !any(mask).   action/first/cur/i int64 [B], mask/done uint8.
 * (a5 TSPEnv._reset is pure allocation and lives in the host mirror.)
 * -------------------------------------------------------------------------- */
int rl4co_tsp_step(const int64_t* action, uint8_t* action_mask, int64_t* first_node,
                   int64_t* current_node, int64_t* step_i, uint8_t* done, int B, int N,
                   int32_t* err, void* stream);

/* --------------------------------------------------------------------------
 * a8+a9  CVRPEnv._step + get_action_mask   envs/routing/cvrp/env.py:66-96,126-136
 * used = (used + demand[clamp(a-1,0,n-1)]) * (a != 0); visited[a] = 1; cur = a;
 * done = (sum visited == N); mask_loc = visited[1:] | (demand + used > cap + 1e-5);
 * mask[0] = !((cur==0) & any(!mask_loc)); mask[1:] = !mask_loc.
 * demand [B_inst,N-1] (trajectory b reads row b % B_inst), used/cap f32 [B].
 * Passing action == NULL only recomputes the mask (a9 / a7's reset mask).
 * -------------------------------------------------------------------------- */
int rl4co_cvrp_step(const int64_t* action, const float* demand, float* used_capacity,
                    const float* vehicle_capacity, uint8_t* visited, int64_t* current_node,
                    uint8_t* action_mask, uint8_t* done, int B, int B_inst, int N,
                    int32_t* err, void* stream);

/* --------------------------------------------------------------------------
 * N4  OPEnv (orienteering)      envs/routing/op/env.py:67-194
 * reset: max_length_table[b,j] = (max_length[b] - |loc_0 - loc_j|) - 1e-6  (op/env.py:118-122)
 * step : tour += |loc_a - loc_cur| ; visited[a] = 1 ; done = (a == 0) & (i > 0) ; i += 1 ; cur = a ;
 *        mask[j] = !(visited[j] | visited[0] | tour + |loc_j - loc_cur| > max_length_table[j]) ; mask[0] = 1
 *        (action == NULL: mask only). Distances are sqrt(fma(dy,dy,dx*dx)) like the tour length.
 * reward: out[b] = sum_t prize[b % B_values, actions[b,t]] in ATen's inner-dim sum order (op/env.py:156-166)
 * check : no customer twice (RL4CO_EBIT_DUPLICATES), closed tour length <= max_length + 1e-5 measured
 *         at every node as the reference does (RL4CO_EBIT_MAX_LENGTH)   (op/env.py:168-194)
 * locs [B_inst,N,2], max_length_table / prize [B_inst,N] (trajectory b reads row b % B_inst).
 * -------------------------------------------------------------------------- */
int rl4co_op_max_length(const float* locs, const float* max_length, int B, int N, float* table, void* stream);
int rl4co_op_step(const int64_t* action, const float* locs, const float* max_length_table, float* tour_length,
                  uint8_t* visited, int64_t* current_node, int64_t* step_i, uint8_t* action_mask, uint8_t* done,
                  int B, int B_inst, int N, int32_t* err, void* stream);
int rl4co_gather_sum_f32(const float* values, const int64_t* actions, int B, int B_values, int N, int T, float* out,
                         void* stream);
int rl4co_op_check_solution(const int64_t* actions, const float* locs, const float* max_length_table, int B, int B_inst,
                            int N, int T, int32_t* err, void* stream);

/* --------------------------------------------------------------------------
 * N4  PCTSPEnv (prize-collecting TSP)   envs/routing/pctsp/env.py:62-201
 * step : prize += real_prize[a] ; visited[a] = 1 ; done = (i > 0) & (a == 0) ; i += 1 ; cur = a ;
 *        mask[j>=1] = !(visited[j] | visited[0]) ; mask[0] = !((prize < 1.0) & (some customer unvisited))
 *        (action == NULL: mask only). real_prize [B_inst,N] with 0 for the depot.
 * reward (host composition, fp32): gather_sum(penalty, actions) - (tour_length(depot ++ tour) +
 *        gather_sum(penalty, 1..N-1)) — three ATen-order sums, pctsp/env.py:150-173.
 * check : no customer twice (RL4CO_EBIT_DUPLICATES); collected prize >= 1 - 1e-5 or every customer
 *         visited (RL4CO_EBIT_PRIZE)   pctsp/env.py:175-201. Trailing depot zeros are neutral.
 * -------------------------------------------------------------------------- */
int rl4co_pctsp_step(const int64_t* action, const float* real_prize, float* cur_total_prize, uint8_t* visited,
                     int64_t* current_node, int64_t* step_i, uint8_t* action_mask, uint8_t* done, int B, int B_inst,
                     int N, int32_t* err, void* stream);
/* prize_sum [B] = rl4co_gather_sum_f32(real_prize, actions) (the reference's summation order) */
int rl4co_pctsp_check_solution(const int64_t* actions, const float* prize_sum, int B, int N, int T, int32_t* err,
                               void* stream);

/* --------------------------------------------------------------------------
 * N4  PDPEnv (pickup and delivery)   envs/routing/pdp/env.py:64-99,204-223
 * Nodes: 0 = depot, 1..n/2 pickups, n/2+1..n their deliveries (n = N - 1 even).
 * step : available[a] = 0 ; to_deliver[(a + n/2) % (n + 1)] = 1 ; mask = available & to_deliver ;
 *        done = no node available ; i += 1 ; cur = a   (action == NULL: mask only).
 * check (actions [B,T]; a depot visit is prepended unless force_start_at_depot): every node exactly once
 *        (RL4CO_EBIT_NOT_ALL_NODES), no depot visit strictly inside the tour (RL4CO_EBIT_DEPOT_MIDDLE),
 *        every pickup before its delivery (RL4CO_EBIT_NO_PICKUP).
 * reward: rl4co_tour_length_f32(prepend_depot = 1, negate = 1)   pdp/env.py:191-202
 * -------------------------------------------------------------------------- */
int rl4co_pdp_step(const int64_t* action, uint8_t* available, uint8_t* to_deliver, int64_t* current_node, int64_t* step_i,
                   uint8_t* action_mask, uint8_t* done, int B, int N, int32_t* err, void* stream);
int rl4co_pdp_check_solution(const int64_t* actions, int B, int N, int T, int force_start_at_depot, int32_t* err,
                             void* stream);

/* --------------------------------------------------------------------------
 * T transitions of given trajectories in ONE launch, with what the decoder saw before each of them
 * (the `evaluate` decoding strategy's state sequence: utils/decoding.py:448-461, models/common/constructive/base.py:226-263
 * with `actions` given; rl/ppo/ppo.py:128-170 re-evaluates stored actions the same way). Equivalent to, for t in [0, T):
 *   masks[:, t] = action_mask ; prev[:, t] = current_node ;
 *   TSP: first[:, t] = first_node ; use_placeholder[:, t] = (i < 1)                     (env_embeddings/context.py:86-103)
 *   CVRP / CVRPTW / OP / PCTSP: rem[:, t] = rem_base - scalar (PCTSP: clamped at 0)     (context.py:105-213)
 *   CVRPTW: now[:, t] = current_time ;
 *   rl4co_<env>_step(actions[:, t], state ...)
 * on the same state arrays (updated in place; the step semantics are those entry points', literally the same device code).
 * `scalar` = used_capacity (CVRP, CVRPTW), tour_length (OP), cur_total_prize (PCTSP); `visited` = `available` for PDP;
 * `demand` = real_prize [B_inst, N] for PCTSP; `max_length` = the OP entry-limit table of rl4co_op_max_length.
 * -------------------------------------------------------------------------- */
typedef struct rl4co_env_replay_args {
  int32_t env;    /* RL4CO_ENV_* */
  int32_t B;      /* trajectories (rows of the state) */
  int32_t B_inst; /* instances (rows of the instance data); B % B_inst == 0, row b reads instance b % B_inst */
  int32_t N;      /* nodes */
  int32_t T;      /* steps to replay */
  int32_t reserved0;
  const int64_t* actions; /* [B, T] */
  /* state, updated in place */
  uint8_t* action_mask;  /* [B, N] */
  int64_t* current_node; /* [B] */
  uint8_t* done;         /* [B] */
  int64_t* first_node;   /* [B]    TSP */
  int64_t* step_i;       /* [B]    TSP, OP, PCTSP, PDP */
  uint8_t* visited;      /* [B, N] CVRP, CVRPTW, OP, PCTSP; PDP: available */
  uint8_t* to_deliver;   /* [B, N] PDP */
  float* scalar;         /* [B]    see above */
  float* current_time;   /* [B]    CVRPTW */
  /* instance data */
  const float* vehicle_capacity; /* [B]             CVRP, CVRPTW */
  const float* demand;           /* [B_inst, N - 1] CVRP, CVRPTW; PCTSP: real prize [B_inst, N] */
  const float* locs;             /* [B_inst, N, 2]  OP, CVRPTW */
  const float* max_length;       /* [B_inst, N]     OP */
  const float* time_windows;     /* [B_inst, N, 2]  CVRPTW */
  const float* durations;        /* [B_inst, N]     CVRPTW */
  const float* rem_base;         /* [B] minuend of the context scalar: capacity / max_length[:, 0] / prize_required */
  /* outputs */
  uint8_t* masks;           /* [B, T, N] */
  int64_t* prev;            /* [B, T] */
  int64_t* first;           /* [B, T] TSP */
  uint8_t* use_placeholder; /* [B, T] TSP */
  float* rem;               /* [B, T] CVRP, CVRPTW, OP, PCTSP */
  float* now;               /* [B, T] CVRPTW */
  int32_t* err;             /* sticky bits (RL4CO_EBIT_INFEASIBLE: an action out of range), may be NULL */
  uint32_t* mask_bits;      /* [B, T, mask_words] optional: `masks` as bits (bit j of word j / 32 = node j feasible; the
                               padding bits are 0) — rl4co_cross_attn_*'s mask */
  int32_t mask_words;       /* >= ceil(N / 32) */
  int32_t reserved1;
} rl4co_env_replay_args;

int rl4co_env_replay(const rl4co_env_replay_args* args, void* stream);

/* --------------------------------------------------------------------------
 * N4  CVRPTWEnv (CVRP + time windows)   envs/routing/cvrptw/env.py:83-190
 * step : d = |loc_a - loc_cur| ; time = (a != 0) * (max(time + d, tw_start[a]) + duration[a]) ; then the CVRP
 *        transition (a8) ; mask = cvrp_mask & (time + |loc_j - loc_a| <= tw_end[j])   (action == NULL: mask only).
 *        Distances are sqrt(fma(dy, dy, dx * dx)) like the tour-length kernel.
 * check: the instance-data assertions (RL4CO_EBIT_TW_NEGATIVE / _RETURN / _DURATION / _EMPTY; `max_time` is
 *        tw_end of instance 0's depot, as in the reference) and the deadline replay
 *        t = max(trunc(t + d), tw_start) <= tw_end ; t += duration ; t = 0 at the depot (RL4CO_EBIT_TW_DEADLINE).
 *        The CVRP part of the check is rl4co_cvrp_check_solution. Trailing depot zeros are neutral.
 * -------------------------------------------------------------------------- */
int rl4co_cvrptw_step(const int64_t* action, const float* demand, const float* locs, const float* time_windows,
                      const float* durations, float* used_capacity, const float* vehicle_capacity, float* current_time,
                      uint8_t* visited, int64_t* current_node, uint8_t* action_mask, uint8_t* done, int B, int B_inst, int N,
                      int32_t* err, void* stream);
int rl4co_cvrptw_check_solution(const int64_t* actions, const float* locs, const float* time_windows,
                                const float* durations, int B, int B_inst, int N, int T, int32_t* err, void* stream);

/* --------------------------------------------------------------------------
 * a13-a21  AttentionModel decode: one step, or the whole autoregressive loop.
 *
 * Replaces, per step: TSPContext/VRPContext (env_embeddings/context.py:105-149),
 * AttentionModelDecoder._compute_q/_compute_kvl/forward (zoo/am/decoder.py:128-193),
 * PointerAttention.forward (nn/attention.py:274-320), process_logits
 * (utils/decoding.py:138-188), Greedy/Sampling/Evaluate (decoding.py:344-461),
 * TSPEnv._step / CVRPEnv._step (+ get_action_mask); and, with max_steps > 1, the
 * `while not td["done"].all()` loop of ConstructivePolicy.forward
 * (models/common/constructive/base.py:226-238).
 *
 * The cache is the *folded* form of PrecomputedCache (zoo/am/decoder.py:21-40,
 * 201-228) — algebraically identical, built once per rollout by dense GEMMs:
 *   glimpse_key = h Wk^T, glimpse_val = h Wv^T                     [B_inst,N,128]
 *   logit_key   = h (W_out^T Wl)^T   (project_out folded into the logit key)
 *   ctx_first   = h W_ctx[:, :128]^T, ctx_cur = h W_ctx[:,128:256]^T (TSP)
 *   ctx_cur     = h W_ctx[:, :128]^T, w_cap = W_ctx[:,128]           (CVRP)
 *   q_bias      = project_fixed_context(mean_j h_j) or NULL (POMO)  [B_inst,128]
 *   q_step0     = W_ctx W_placeholder (TSP, step i==0)               [128]
 * so that one decode step is a pure stream over the three [N,128] planes.
 * Trajectory r (0..B-1) uses instance r % B_inst (s-major multistart layout).
 * -------------------------------------------------------------------------- */
typedef struct rl4co_am_decode_args {
  /* problem */
  int32_t env;         /* RL4CO_ENV_*                                            */
  int32_t B;           /* trajectories = instances x starts                      */
  int32_t B_inst;      /* instances owning cache rows                            */
  int32_t N;           /* nodes incl. depot                                      */
  /* decode configuration */
  int32_t mode;        /* RL4CO_DECODE_*                                         */
  int32_t max_steps;   /* 1 = single step; >= horizon = full rollout             */
  int32_t mask_inner;  /* PointerAttention(mask_inner=True) default 1            */
  int32_t mask_logits; /* process_logits(mask_logits=True) default 1             */
  float tanh_clipping; /* 10.0 default (am/policy.py:74); 0 disables             */
  float temperature;   /* 1.0 default                                            */
  /* folded cache */
  int32_t cache_dtype; /* RL4CO_DT_F32 | RL4CO_DT_BF16 for the three planes      */
  int32_t variant;     /* RL4CO_VARIANT_*: 0 = let the library choose              */
  const void* glimpse_key;
  const void* glimpse_val;
  const void* logit_key;
  int64_t kvl_row_stride;   /* elements between node rows   (128 planar, 384 interleaved) */
  int64_t kvl_batch_stride; /* elements between instances                                  */
  const void* ctx_first;    /* [B_inst,N,128] TSP only; element type ctx_dtype (fp32 unless set) */
  const void* ctx_cur;      /* [B_inst,N,128]                                              */
  const float* q_bias;      /* [B_inst,128] or NULL                                        */
  const float* q_step0;     /* [128] TSP only                                              */
  const float* w_cap;       /* [128] CVRP only                                             */
  /* "unfolded" parity mode (unfold = 1; TSP / CVRP, streaming variant): the three batch-shared matrices are
   * applied PER STEP in the reference's association instead of being folded into per-node rows —
   * q = project_context([h_first ; h_cur] | W_placeholder | [h_cur ; cap - used]) + graph context
   * (env_embeddings/context.py:61-74,120-134), glimpse = project_out(heads) (nn/attention.py:287), logits =
   * glimpse . K_l with logit_key the RAW third chunk of project_node_embeddings (zoo/am/decoder.py:201-228);
   * each GEMV is one fma chain over the input dims in ascending order, starting from 0. ctx_first / ctx_cur /
   * q_step0 / w_cap are not read. Measures how many greedy near-tie flips the fold itself causes. */
  int32_t unfold;
  int32_t ctx_width;          /* input width of project_context: 256 TSP, 129 CVRP              */
  const float* node_embed;    /* [B_inst,N,128] encoder output h                               */
  const float* w_ctx_t;       /* [ctx_width,128] project_context.weight transposed             */
  const float* w_out_t;       /* [128,128] project_out.weight transposed                       */
  const float* w_placeholder; /* [256] TSP first-step context (context.py:120-128)             */
  /* environment state, read at entry and written back at exit */
  uint8_t* action_mask;     /* [B,N] 1 = feasible                                          */
  int64_t* first_node;      /* [B] TSP                                                     */
  int64_t* current_node;    /* [B]                                                         */
  int64_t* step_i;          /* [B] TSP                                                     */
  uint8_t* done;            /* [B]                                                         */
  const float* demand;      /* [B_inst,N-1] CVRP                                           */
  float* used_capacity;     /* [B] CVRP                                                    */
  const float* vehicle_capacity; /* [B] CVRP                                               */
  uint8_t* visited;         /* [B,N] CVRP, OP                                              */
  /* OP (orienteering, envs/routing/op/env.py): used_capacity carries the tour length so far, step_i
   * the step counter; the depot row of max_length plays the vehicle capacity in the context scalar */
  const float* locs;        /* [B_inst,N,2] OP                                             */
  const float* max_length;  /* [B_inst,N] OP: longest tour with which node j may be entered */
  /* PCTSP (envs/routing/pctsp/env.py): demand = real prize WITH the depot column [B_inst,N],
   * used_capacity = prize collected so far, vehicle_capacity = prize_required [B], step_i, visited */
  /* PDP (envs/routing/pdp/env.py): visited carries the `available` flags (1 = not yet visited), to_deliver
   * the precedence flags; step_i; the context is the current node's row alone (no scalar, w_cap NULL) */
  uint8_t* to_deliver;      /* [B,N] PDP                                                   */
  /* CVRPTW (envs/routing/cvrptw/env.py): the CVRP fields plus locs, time_windows, durations, current_time; the
   * context carries two scalars, q = w_time * time + (w_cap * (cap - used) + ctx_cur[cur]) (context.py:152-166) */
  const float* time_windows; /* [B_inst,N,2] (start, end), fp32                             */
  const float* durations;    /* [B_inst,N] service times                                    */
  float* current_time;       /* [B]                                                         */
  const float* w_time;       /* [128] = W_ctx[:, 129]                                       */
  /* decoding inputs */
  const float* exp_noise;   /* [max_steps,B,N] Exp(1) draws (parity mode) or NULL          */
  uint64_t philox_seed;     /* in-kernel Exp(1) noise when exp_noise == NULL               */
  uint64_t philox_offset;
  const uint64_t* philox_seed_dev; /* optional device word XOR-ed into philox_seed at kernel entry: a captured HIP
                                    * graph bakes the argument block, the word lets every replay draw fresh noise */
  const int64_t* forced_actions; /* [B,out_stride] for RL4CO_DECODE_EVALUATE               */
  /* outputs */
  int32_t t0;               /* column of the first step in actions/logps                   */
  int32_t out_stride;       /* row stride (Tmax) of actions/logps/forced_actions           */
  int64_t* actions;         /* [B,out_stride]                                              */
  float* logps;             /* [B,out_stride] log p(a_t)                                   */
  float* all_logps;         /* [B,out_stride,N] or NULL (store_all_logp / entropy)         */
  float* entropy;           /* [B] accumulated -sum p log p, or NULL                       */
  int32_t* n_steps;         /* [B] steps actually taken by each trajectory, or NULL        */
  int32_t* steps_summary;   /* [4], 8-byte aligned, or NULL (zero-initialised by the caller): [0] = max and [1] += sum
                             * over the trajectories of the steps taken — what the host needs of n_steps without a
                             * reduction launch; [2..3] = ONE 64-bit counter (little endian: low word, high word) += cache
                             * rows the launch streamed per plane from HBM (feasible rows of every step; 0 for the
                             * variants whose planes are LDS-resident): the measured numerator of the decode kernel's
                             * roofline in bench.py. 64 bits: 409 600 trajectories x 5050 rows already reach 2.07e9.
                             * A misaligned pointer is refused (RL4CO_STATUS_INVALID_ARGUMENT)                       */
  int32_t* err;             /* sticky error bits                                           */
  /* (r06) the folded context tables in the element type of the planes: ctx_dtype = RL4CO_DT_BF16 / _F16 (0 = RL4CO_DT_F32,
   * the default) with their own strides in ELEMENTS (0 = dense: row 128, instance N * 128) — e.g. columns 3 and 4 of the
   * ONE [B_inst * N, 5 * 128] 16-bit matrix a fused cache-fold GEMM writes (row stride 640): no fp32 copies of the tables.
   * The rows are widened to fp32 on load; everything downstream is the fp32 arithmetic of the fp32 tables. Multistart
   * variant (am_decode_ms.hip); the other variants require RL4CO_DT_F32. */
  int32_t ctx_dtype;
  int32_t reserved0;
  int64_t ctx_row_stride;
  int64_t ctx_batch_stride;
} rl4co_am_decode_args;

int rl4co_am_decode(const rl4co_am_decode_args* args, void* stream);

/* Bytes of LDS one trajectory needs for N nodes (for occupancy planning / tests). */
int rl4co_am_decode_lds_bytes(int N, int env);

/* Number of row groups G the kernel that will serve `args` splits a trajectory's cache rows
 * into (row j -> group j % G). G fixes the fp32 summation tree of the glimpse (am_decode.hip
 * header), so the specified-order oracle asks for it instead of guessing. -1 on bad args. */
int rl4co_am_decode_row_groups(const rl4co_am_decode_args* args); /* 0: variant without such a contract (MS) */
/* The variant (RL4CO_VARIANT_STREAM / _LDS) rl4co_am_decode will run for `args`. */
int rl4co_am_decode_variant(const rl4co_am_decode_args* args);

/* --------------------------------------------------------------------------
 * a11-a13  fused encoder + decoder-cache fold on the matrix cores (inference rollouts).
 *
 * Replaces, per instance and in one launch: TSPInitEmbedding / VRPInitEmbedding
 * (models/nn/env_embeddings/init.py:55-68,115-136), GraphAttentionNetwork with L x
 * [x + MHA(x) -> Norm -> x + MLP(x) -> Norm] (models/nn/graph/attnnet.py:16-106,
 * nn/attention.py:110-134, nn/ops.py:30-54, nn/mlp.py:52-61; 8 heads, d = 128, FFN 512) and
 * AttentionModelDecoder._precompute_cache (zoo/am/decoder.py:201-228) in the folded form of
 * rl4co_am_decode_args. 16-bit MFMA inputs (act_dtype: bf16 or fp16), fp32 accumulation, 16-bit
 * residual stream (the reference's mixed-precision regimes, utils/trainer.py:57). Normalisation: norm = 0 is
 * batch norm in EVAL mode, passed as a per-channel (scale, shift) pair folded from the
 * running statistics; norm = 1 is instance norm (POMO), scale/shift = gamma/beta; norm = 2 is the
 * reference's "layer" normalisation (nn/ops.py:48-51: ONE mean and ONE unbiased variance over all N x 128
 * values of an instance, eps 1e-5, no affine) — scale is ignored, the 16-bit kernel reads the bias of the
 * GEMM in front of the norm (bo / b2) from the shift slot and adds it before the statistics (the fp32
 * kernel adds bo / b2 itself).
 * Train-mode batch statistics couple instances and stay on the torch path (which also
 * provides autograd). N <= rl4co_am_encoder_max_nodes().
 *
 * Weight matrices are nn.Linear weights [out,in] re-packed once per weight update into MFMA
 * fragment order: [out/32 tiles][in/16 ksteps][64 lanes][8] elements of act_dtype with lane = 32*hi + row,
 * element s = W[32*tile + row][16*kstep + 8*hi + s]   (rl4co_amd/encoder.py: pack_weight).
 * -------------------------------------------------------------------------- */
typedef struct rl4co_am_encoder_args {
  int32_t env;         /* RL4CO_ENV_TSP | RL4CO_ENV_CVRP (every depot env with a 3- or 4-feature customer embedding) | RL4CO_ENV_PDP */
  int32_t B;           /* instances                                                */
  int32_t N;           /* nodes incl. depot                                        */
  int32_t num_layers;  /* 3 (AM) / 6 (POMO)                                        */
  int32_t norm;        /* 0 = per-channel affine (batch norm, eval), 1 = instance, 2 = layer (nn/ops.py:48-51: one mean / one
                          unbiased variance per instance, no affine) — every entry point serves all three, the token-tile
                          ones through split half-layer kernels + a norm-apply kernel. What n*_scale / n*_shift hold:
                            0: scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale
                               (16-bit kernels: + the preceding GEMM's bias * scale; fp32 kernels add bo / b2 themselves)
                            1: gamma, beta (statistics per instance and channel over the nodes)
                            2: 16-bit kernels: scale unused, shift = the bias of the GEMM in front of the norm (out_proj /
                               the MLP's second linear: it does not cancel under a whole-instance mean and is added before
                               the statistics); fp32 kernels: unused (they add bo / b2 themselves) */
  int32_t cache_dtype; /* dtype of the three kvl planes written: RL4CO_DT_F32 or act_dtype */
  int32_t act_dtype;   /* 16-bit element type of the MFMA operands, the packed weights and the LDS residual stream:
                          RL4CO_DT_BF16 (autocast bfloat16) or RL4CO_DT_F16 (autocast float16, the reference's
                          default "16-mixed", utils/trainer.py:57: v_mfma_f32_32x32x16_f16, softmax always
                          max-subtracted — fp16 lacks the range of the bounded-score shortcut) */
  int32_t ctx_dtype;   /* (r06; was reserved0 = 0) element type of ctx_first / ctx_cur: RL4CO_DT_F32 (0, the default) or —
                          rl4co_am_encoder only, 16-bit planes — act_dtype: the tables written as dense 16-bit [B,N,128] rows
                          for rl4co_am_decode_args.ctx_dtype; every other entry point requires 0 */
  const float* locs;   /* [B,N,2] (CVRP: depot first, cvrp/env.py:108)             */
  const float* demand; /* [B,N-1] CVRP demand / OP prize / PCTSP expected prize     */
  const float* feature4; /* [B,N-1] PCTSP penalty / CVRPTW tw start (w_init is then [128,4] / [128,6]) or NULL */
  const float* feature5; /* [B,N-1] CVRPTW tw end, or NULL                          */
  const float* feature6; /* [B,N-1] CVRPTW service time, or NULL (feature4..6 are set together) */
  const float* w_init; /* [128,2] TSP / [128,3] CVRP customers / [128,4] PCTSP / [128,6] CVRPTW */
  const float* b_init; /* [128]                                                    */
  const float* w_depot; /* [128,2] depot environments                              */
  const float* b_depot; /* [128]                                                   */
  const float* w_extra; /* [128,2] RL4CO_ENV_PDP: the delivery embedding; w_init is then the pickup embedding */
  const float* b_extra; /* [128]     [128,4] over (x, y, x', y' of the paired delivery), init.py:335-360      */
  const void* wqkv_packed; /* [L] x packed [384,128]                               */
  const float* bqkv;       /* [L,384]                                              */
  const void* wo_packed;   /* [L] x packed [128,128]                               */
  const float* bo;         /* [L,128]                                              */
  const float* n1_scale;   /* [L,128]                                              */
  const float* n1_shift;   /* [L,128]                                              */
  const void* w1_packed;   /* [L] x packed [512,128]                               */
  const float* b1;         /* [L,512]                                              */
  const void* w2_packed;   /* [L] x packed [128,512]                               */
  const float* b2;         /* [L,128]                                              */
  const float* n2_scale;   /* [L,128]                                              */
  const float* n2_shift;   /* [L,128]                                              */
  const void* wfold_packed; /* 5 (TSP) / 4 (CVRP) x packed [128,128]: Wk, Wv, W_out^T Wl, W_ctx blocks */
  const float* w_fixed;     /* [128,128] project_fixed_context or NULL              */
  void* kvl;                /* planes 0..2 of the folded cache                      */
  int64_t kvl_plane_stride; /* elements between planes                              */
  int64_t kvl_batch_stride; /* elements between instances                           */
  void* ctx_first;          /* [B,N,128] TSP; fp32 unless ctx_dtype says otherwise   */
  void* ctx_cur;            /* [B,N,128]                                            */
  float* q_bias;            /* [B,128] or NULL                                      */
  float* hidden;            /* [B,N,128] final node embeddings, or NULL             */
} rl4co_am_encoder_args;

int rl4co_am_encoder(const rl4co_am_encoder_args* args, void* stream);
int rl4co_am_encoder_max_nodes(void);

/* a12 (training): the INPUT gradient of SkipConnection(MLP 128 -> 512 -> 128, ReLU) in one launch
 *   rl4co/models/nn/mlp.py:52-61 under rl4co/models/nn/ops.py:9-15 (autograd's backward of x + lin2(relu(lin1(x))))
 *   dh[M,512] = (dy . W2) * [h > 0]  (written once: rl4co_wgrad_* reads it for dW1 / db1)      dx[M,128] = dh . W1 + dy
 * dy [M,128], h [M,512] (the forward's hidden activations), dh, dx: 16-bit elements of `dtype` (RL4CO_DT_BF16 / _F16), fp32
 * accumulation on the matrix cores; w2t_packed = W2^T ([512,128]) and w1t_packed = W1^T ([128,512]) in the encoder's
 * fragment order (rl4co_am_encoder_args: "[out/32 tiles][in/16 ksteps][64 lanes][8]"). The 512-wide gradient goes
 * through LDS in four chunks: 1.05 GB of traffic per layer at M = 409 600 where two GEMM launches moved 1.58. */
int rl4co_mlp_input_grad(const void* dy, const void* h, int64_t M, const void* w2t_packed, const void* w1t_packed, int dtype,
                         void* dh, void* dx, void* stream);

/* The TRAINING forward of an instance-norm encoder stack (POMO: zoo/pomo/model.py:59-63; nn/graph/attnnet.py:16-106) in ONE
 * launch — the fused kernel's layer body fed with the init embedding, one workgroup per instance, N <= 128 — writing, per
 * layer, every tensor the backward kernels (rl4co_linear_* input gradients, rl4co_wgrad_*, rl4co_attn_bwd_*,
 * rl4co_skip_inorm_bwd_*) read: the per-op forward it replaces took seven launches and 27 [B N,128]-sized passes per
 * layer. Of `args` only B, N, num_layers, norm (must be 1), act_dtype, the packed weights (W_q UNSCALED: the scores are
 * scaled in the kernel, the saved q is the q the products used), bqkv, b1 and n*_scale / n*_shift (gamma, beta) are read. */
typedef struct rl4co_am_train_save {
  const void* x0; /* [B,N,128] act_dtype: the encoder's input (init embedding)                     */
  void* out;      /* [L,B,N,128] layer outputs (out[l-1] is layer l's input; out[L-1] the result)   */
  void* qkv;      /* [L,B,N,384] q | k | v                                                          */
  void* att;      /* [L,B,N,128] attention output before out_proj                                   */
  void* y1;       /* [L,B,N,128] x + attention branch (pre-norm; the branch's bias cancels in the norm) */
  void* x1;       /* [L,B,N,128] norm1 output                                                       */
  void* h;        /* [L,B,N,512] relu(x1 W1^T + b1)                                                 */
  void* y2;       /* [L,B,N,128] x1 + MLP branch (pre-norm)                                         */
  float* lse;     /* [L,B,8,N] log-sum-exp of the scaled scores, log2 domain (rl4co_attn_fwd's)     */
  float* stats;   /* [L,4,B,128] mean1, rstd1, mean2, rstd2                                         */
} rl4co_am_train_save;
int rl4co_am_encoder_train_fwd(const rl4co_am_encoder_args* args, const rl4co_am_train_save* save, void* stream);

/* The 16-bit encoder + cache fold for graphs of ANY size (csrc/am_encoder.hip: token-tile kernels; BASELINE configs[4],
 * CVRP-500): the fused kernel's layer algebra, GEMM routine and packed weights over tiles of 128 nodes — init embedding,
 * per layer [Q / K / V projection (+ per-head score bounds), rl4co_attn_flash_pre_*, ONE kernel for out-proj + norm + MLP +
 * norm], fold, graph context. Same argument struct and packing as rl4co_am_encoder; norm 0 (batch norm, eval) runs
 * the three-launch layer, norm 1 / 2 (instance / layer: statistics over ALL tiles of an instance) two half-layer kernels
 * that stop before their norm, write the pre-norm sums and per-tile (mean, centred sum of squares), and a norm-apply
 * kernel combining the tiles' pairs (Chan's update) — the workspace then also holds those sums and statistics. B <= 65535.
 * `workspace`: at least rl4co_am_encoder_tokens16_workspace(B, N) bytes of device memory, 16-byte aligned. */
int rl4co_am_encoder_tokens16(const rl4co_am_encoder_args* args, void* workspace, int64_t workspace_bytes, void* stream);
int64_t rl4co_am_encoder_tokens16_workspace(int B, int N);

/* The same fused encoder + cache fold in EXACT fp32 (csrc/am_encoder_f32.hip, v_mfma_f32_16x16x4_f32: f32 operands,
 * f32 accumulate, a k-ordered fmaf chain) — the encoder of the bit-identical configuration (no autocast), replacing the
 * same reference functions as rl4co_am_encoder with arithmetic in ATen's own order where that is knowable: GEMM then
 * + bias, x + branch, norm as x * alpha + beta, softmax as exp(s - max) / sum. Same argument struct, read this way:
 *   act_dtype   RL4CO_DT_F32; cache_dtype any of F32 / BF16 / F16 (planes rounded once on the way out)
 *   w*_packed   fp32, [out/16 tiles][in/16 chunks][64 lanes][4]: lane = 16 g + c, element s = W[16 tile + c][16 chunk + 4 g + s]
 *               (rl4co_amd/encoder.py: pack_weight_f32); the query rows of Wqkv and bqkv carry 1 / sqrt(16) (exact)
 *   bo, b2      ADDED by the kernel (the 16-bit kernel has them folded into the norm's shift)
 *   n*_scale / n*_shift   norm = 0: alpha = weight / sqrt(running_var + eps), beta = bias - running_mean * alpha;
 *               norm = 1: gamma, beta (alpha / beta built per instance from two-pass statistics)
 *   wfold_packed  3 + (ctx_first != NULL) + (ctx_cur != NULL) blocks of [128,128] in that order; with both context
 *               tables NULL the three blocks are the raw rows of project_node_embeddings (the reference's own
 *               association of the decoder, cache.py fold=False) and `hidden` carries the node embeddings. */
int rl4co_am_encoder_f32(const rl4co_am_encoder_args* args, void* stream);

/* The exact-fp32 encoder + cache fold for graphs of ANY size (csrc/am_tokens_f32.hip; BASELINE configs[4], CVRP-500): the
 * same arithmetic as rl4co_am_encoder_f32 (same GEMM routine and summation order) as launches over tiles of 128 nodes —
 * init embedding, per layer [Q/K/V projection, attention with keys / values streamed and an online softmax, out-proj +
 * norm + MLP + norm], fold, graph context. Same argument struct and packing as rl4co_am_encoder_f32; norm 0 .. 2 as for
 * rl4co_am_encoder_tokens16 (instance / layer norm: half-layer kernels + norm-apply over the tiles' combined statistics).
 * B <= 65535. `workspace`: at least rl4co_am_encoder_tokens_f32_workspace(B, N) bytes of device memory, 16-byte aligned,
 * contents irrelevant on entry (the [B N, 128] fp32 activation buffers, the per-head transposed values, the per-tile
 * statistics). */
int rl4co_am_encoder_tokens_f32(const rl4co_am_encoder_args* args, void* workspace, int64_t workspace_bytes, void* stream);
int64_t rl4co_am_encoder_tokens_f32_workspace(int B, int N);

/* The init embeddings alone: out[B,N,128] in the activations' type (act_dtype BF16 / F16: ..._init_embeds16; F32:
 * ..._init_embeds_f32) from the same argument struct — only env, B, N, act_dtype, the feature pointers and the init
 * embedding's weights are read. Replaces `init_embeds` of AttentionModelPolicy.forward(return_init_embeds=True)
 * (rl4co/models/zoo/am/encoder.py:84-103 returns them next to the final embeddings; env_embeddings/init.py). It is the
 * token path's first launch, i.e. the routine the fused kernels run internally. */
int rl4co_am_encoder_init_embeds16(const rl4co_am_encoder_args* args, void* out, void* stream);
int rl4co_am_encoder_init_embeds_f32(const rl4co_am_encoder_args* args, float* out, void* stream);

/* fp32 side of the cache fold from the final node embeddings of ANY encoder (zoo/am/decoder.py:201-228, cache.py):
 * out[i][B,N,128] = h . W_i^T for nblocks <= 5 blocks of [128,128] packed as for rl4co_am_encoder_f32 (the context
 * tables), and q_bias[B,128] = w_fixed . mean_j h[b,j] (w_fixed plain [128,128] fp32; both NULL: skipped). h: [B,N,128]
 * rows of h_dtype (RL4CO_DT_F32 / _BF16 / _F16, widened on load); fp32 MFMA, fp32 outputs. */
int rl4co_am_fold_tables_f32(const void* h, int h_dtype, int B, int N, const float* w_packed, int nblocks, float* const* out,
                             const float* w_fixed, float* q_bias, void* stream);

/* --------------------------------------------------------------------------
 * N1  teacher-forced log-likelihood, backward pass.
 *
 * Replaces the autograd graph the reference builds through its T-step decode loop
 * (rl/reinforce/reinforce.py:99-102 differentiates out["log_likelihood"];
 * models/common/constructive/base.py:226-263; decode_type="evaluate",
 * utils/decoding.py:448-461): given the actions of B trajectories and the upstream
 * gradient g[B,T] of their per-step log-probs, recompute every step from the reset state and
 * return dL/d(folded cache) for L = sum g * log p. Outputs are fp32; d_ctx_first, d_q_step0
 * and d_w_cap are accumulated atomically and must be zero-initialised by the caller, the
 * others are written. One workgroup per instance, its B / B_inst trajectories (s-major rows)
 * replayed in turn. N <= rl4co_am_teacher_max_nodes(). Every RL4CO_ENV_* on both variants (RL4CO_TEACHER_MMA: bf16
 * planes, 16-step blocks on the matrix cores; RL4CO_TEACHER_REPLAY: fp32 arithmetic, step by step).
 * -------------------------------------------------------------------------- */
#define RL4CO_TEACHER_AUTO 0   /* MMA when the planes are bf16 and T fits its step tables, else REPLAY */
#define RL4CO_TEACHER_REPLAY 1 /* am_teacher.hip: fp32 step-by-step replay, planes in registers        */
#define RL4CO_TEACHER_MMA 2    /* am_teacher_mma.hip: 16-step blocks on v_mfma_f32_16x16x16_bf16       */
typedef struct rl4co_am_teacher_args {
  int32_t env;
  int32_t B;            /* trajectories                                             */
  int32_t B_inst;       /* instances                                                */
  int32_t N;
  int32_t T;            /* columns of actions / grad_logp / logp_out                */
  int32_t t0;           /* 1: column 0 is the imposed multistart node (log-prob 0)  */
  int32_t mask_inner;
  int32_t mask_logits;
  float tanh_clipping;
  float temperature;
  int32_t cache_dtype;  /* dtype of the three planes                                */
  int32_t variant;      /* RL4CO_TEACHER_AUTO / _REPLAY / _MMA                      */
  const void* glimpse_key;
  const void* glimpse_val;
  const void* logit_key;
  int64_t kvl_row_stride;
  int64_t kvl_batch_stride;
  const void* ctx_first;         /* element type ctx_dtype (fp32 unless set), strides below */
  const void* ctx_cur;
  const float* q_bias;
  const float* q_step0;
  const float* w_cap;
  const int64_t* actions;        /* [B,T]                                           */
  const float* demand;           /* [B_inst,N-1] CVRP                               */
  const float* vehicle_capacity; /* [B_inst] CVRP                                   */
  const float* locs;             /* [B_inst,N,2] OP, CVRPTW                         */
  const float* max_length;       /* [B_inst,N] OP entry-limit table                 */
  const float* time_windows;     /* [B_inst,N,2] CVRPTW (start, end), fp32          */
  const float* durations;        /* [B_inst,N] CVRPTW service times                 */
  const float* w_time;           /* [128] CVRPTW: W_ctx[:, 129]                     */
  const float* grad_logp;        /* [B,T]                                           */
  float* d_kvl;                  /* [3,B_inst,N,128]                                */
  float* d_ctx_first;            /* [B_inst,N,128] TSP, zero-initialised            */
  float* d_ctx_cur;              /* [B_inst,N,128]                                  */
  float* d_q_bias;               /* [B_inst,128] or NULL                            */
  float* d_q_step0;              /* [128] TSP, zero-initialised                     */
  float* d_w_cap;                /* [128] CVRP, zero-initialised                    */
  float* d_w_time;               /* [128] CVRPTW, zero-initialised                  */
  float* logp_out;               /* [B,T] forward values, or NULL                   */
  int32_t* err;
  /* MMA variant: the three plane gradients as bf16 rows with the caller's strides instead of fp32 d_kvl (which may
   * then be NULL) — element [p][b][n][c] at d_planes_bf16 + p * plane_stride + b * batch_stride + n * row_stride + c.
   * Lets the caller place them next to each other as the columns of ONE [B_inst * N, 5 * 128] gradient matrix, the
   * operand of the fold GEMMs' backward (rl4co_linear / rl4co_wgrad), without an fp32 round trip. */
  void* d_planes_bf16;
  int64_t d_planes_row_stride;
  int64_t d_planes_batch_stride;
  int64_t d_planes_plane_stride;
  /* (r06, MMA variant) context tables in the planes' element type, as in rl4co_am_decode_args: ctx_dtype (0 = fp32),
   * strides in elements (0 = dense). d_ctx_in_planes = 1 (needs d_planes_bf16): the context-table gradients leave the
   * kernel converted to the planes' type as planes 3 (ctx_first, TSP) and 4 (ctx_cur; plane 3 in the depot environments)
   * of the same strided gradient matrix — d_ctx_first / d_ctx_cur remain the kernel's fp32 accumulation scratch
   * (d_ctx_first zero-initialised by the caller) and hold nothing the caller needs afterwards. */
  int32_t ctx_dtype;
  int32_t d_ctx_in_planes;
  int64_t ctx_row_stride;
  int64_t ctx_batch_stride;
} rl4co_am_teacher_args;

int rl4co_am_teacher_backward(const rl4co_am_teacher_args* args, void* stream);
int rl4co_am_teacher_max_nodes(void);
/* The variant rl4co_am_teacher_backward would run for these arguments (1 or 2), -1 if invalid. */
int rl4co_am_teacher_variant(const rl4co_am_teacher_args* args);

/* --------------------------------------------------------------------------
 * Element type of the 16-bit training-encoder / attention entry points (r06: ONE entry point per operation; until r05
 * every one of them existed twice, rl4co_<op>_bf16 and rl4co_<op>_f16). `dtype` = RL4CO_DT_BF16 (torch.autocast(bfloat16))
 * or RL4CO_DT_F16 (torch.autocast(float16): the reference's DEFAULT precision, Lightning's "16-mixed",
 * rl4co/utils/trainer.py:57); any other value is RL4CO_ERR_ARG. Every `const void*` / `void*` activation, weight or
 * gradient operand holds elements of that type, fp32 operands stay fp32; conversions to half round to nearest even and
 * overflow to infinity (what GradScaler's inf check expects). One source per kernel, compiled for both element types
 * (csrc/elem16.h); csrc/entry16.hip dispatches. The comments below say "bf16" where they mean "the 16-bit type".
 * -------------------------------------------------------------------------- */

/* --------------------------------------------------------------------------
 * a11 (training)  init embedding  env_embeddings/init.py:55-68,115-136
 * out[m,:] = W[128,F] . feats[m,:F] + b  (F <= 6: x, y (, demand, ...)); fp32 in, bf16 out [M,128].
 * -------------------------------------------------------------------------- */
int rl4co_init_embed(int dtype, const float* feats, const float* w, const float* b, int64_t M, int F, void* out, void* stream);
/* backward of the above w.r.t. W and b: partial[g, c, 0..F-1] = sum over the rows of block g of dout[m,c] * feats[m,f],
 * partial[g, c, F] = sum of dout[m,c]; dout bf16 [M,128], partial fp32 [blocks,128,F+1] (summed over g by the caller in a
 * fixed order: deterministic). Returns the number of blocks the launch uses through *blocks_out when partial == NULL. */
int rl4co_init_embed_wgrad(int dtype, const void* dout, const float* feats, int64_t M, int F, float* partial, int* blocks_out,
                                void* stream);

/* --------------------------------------------------------------------------
 * a12 (training)  SkipConnection + Normalization("instance")
 *   rl4co/models/nn/ops.py:9-15,30-54 ; nn/graph/attnnet.py:16-54 ; zoo/pomo/model.py:59-63
 * forward : y = x + s ; out = (y - mean_n y) * rsqrt(var_n y + eps) * gamma + beta, statistics per
 *           instance and channel over the N nodes (biased variance). bf16 activations [B,N,128],
 *           fp32 arithmetic; y, mean[B,128], rstd[B,128] are kept for the backward pass.
 * backward: dy (the gradient of BOTH skip inputs) from dout; dgamma / dbeta [B,128]: the per-instance
 *           contributions (every element written; the caller sums over B). N <= rl4co_skip_inorm_max_nodes().
 * -------------------------------------------------------------------------- */
int rl4co_skip_inorm_fwd(int dtype, const void* x, const void* s, const float* gamma, const float* beta, float eps,
                              int B, int N, void* y, void* out, float* mean, float* rstd, void* stream);
int rl4co_skip_inorm_bwd(int dtype, const void* dout, const void* y, const float* gamma, const float* mean,
                              const float* rstd, int B, int N, void* dy, float* dgamma, float* dbeta,
                              void* stream);
int rl4co_skip_inorm_max_nodes(void);
/* rl4co_skip_inorm_fwd / _bwd and rl4co_skip_lnorm_fwd / _bwd serve N <= rl4co_skip_inorm_wide_max_nodes() (r06): beyond
 * rl4co_skip_inorm_max_nodes() the instance's rows are re-read per pass instead of waiting in registers; same arithmetic,
 * same accumulation order. */
int rl4co_skip_inorm_wide_max_nodes(void);

/* --------------------------------------------------------------------------
 * a12 (training)  SkipConnection + Normalization("layer")
 *   rl4co/models/nn/ops.py:9-15,48-51
 * forward : y = x + s ; out = (y - mean y) / sqrt(var y + eps) with ONE mean and ONE unbiased variance (divisor M - 1)
 *           over all M = N x 128 values of the instance, no affine. 16-bit activations [B,N,128], fp32 arithmetic;
 *           y and stats[B,2] = (mean, 1 / sqrt(var + eps)) are kept for the backward pass.
 * backward: dy (the gradient of BOTH skip inputs) = r (dout - sum(dout) / M - xh sum(dout xh) / (M - 1)), xh = (y - mean) r.
 *           N <= rl4co_skip_inorm_max_nodes().
 * -------------------------------------------------------------------------- */
int rl4co_skip_lnorm_fwd(int dtype, const void* x, const void* s, float eps, int B, int N, void* y, void* out, float* stats,
                              void* stream);
int rl4co_skip_lnorm_bwd(int dtype, const void* dout, const void* y, const float* stats, int B, int N, void* dy, void* stream);

/* --------------------------------------------------------------------------
 * a12 (training)  SkipConnection + Normalization("batch") — the AttentionModel default
 *   rl4co/models/nn/ops.py:9-15,30-54 ; zoo/am/policy.py:50-122
 * BatchNorm1d over the M = B x nodes rows of bf16 [M,128] activations, batch statistics:
 *   stats : y = x + s (s may be NULL: y is not written, statistics of x);
 *           sums[0][c] += sum_m y, sums[1][c] += sum_m y^2      (fp32 [2,128], zero-initialised)
 *   apply : out = (y - mean) * rstd * gamma + beta               (mean / rstd from the host)
 *   bwd   : sums[0][c] += sum dout, sums[1][c] += sum dout * xh  (zero-initialised), then
 *           dy = rstd gamma (dout - sums[0]/M - xh sums[1]/M); sums[1] = dgamma, sums[0] = dbeta.
 * -------------------------------------------------------------------------- */
int rl4co_skip_bnorm_stats(int dtype, const void* x, const void* s, int64_t M, void* y, float* sums, void* stream);
int rl4co_bnorm_apply(int dtype, const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           int64_t M, void* out, void* stream);
/* out = bn_eval(x + skip) in one pass (SkipConnection + Normalization("batch") in eval mode, nn/ops.py:9-54): the
 * token-parallel inference encoder for graphs beyond rl4co_am_encoder_max_nodes(); x, skip, out bf16 [M,128]. */
int rl4co_skip_bnorm_eval(int dtype, const void* x, const void* skip, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, int64_t M, void* out, void* stream);
int rl4co_bnorm_bwd(int dtype, const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma,
                         int64_t M, float* sums, void* dy, void* stream);

/* --------------------------------------------------------------------------
 * a12 (training)  nn.Linear over the token rows: Wqkv, out_proj, MLP
 *   rl4co/models/nn/attention.py:64-134 ; nn/mlp.py:52-61
 * out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]); bf16 A, W, out, fp32 bias and accumulate.
 * epilogue: relu != 0 -> max(.,0); mask != NULL -> (mask[m,n] > 0 ? . : 0) (ReLU backward, bf16 mask);
 * residual != NULL -> . + residual[m,n] (bf16 [M,N]: the skip connection's gradient joins the branch's input
 * gradient here, nn/ops.py:9-15 backward, instead of in one more pass over both). mask and residual exclude each other.
 * With W = weight^T (contiguous) it is the input gradient dX = dY . weight.
 * N and K multiples of 128; bias, mask and residual may be NULL.
 * -------------------------------------------------------------------------- */
int rl4co_linear(int dtype, const void* a, const void* w, const float* bias, const void* mask, const void* residual, int64_t M,
                      int N, int K, int relu, void* out, void* stream);
/* Weight gradient of the same layers: partial[c][N,K] = dY[rows of chunk c]^T . X[rows of chunk c]
 * (bf16 dY [M,N], X [M,K]; fp32 partial [chunks,N,K], every element written); dW = sum over c.
 * partial_bias [chunks,N] (or NULL) receives the column sums of dY per chunk: db = sum over c.
 * The rows are split into `chunks` equal ranges (a multiple of 32 rows each).
 * chunk_stride: elements between consecutive chunks of BOTH partial arrays; 0 = packed (N*K and N). With
 * partial_bias = partial + N*K and chunk_stride = N*K + N one reduction over the chunk axis yields dW and db. */
int rl4co_wgrad(int dtype, const void* dy, const void* x, int64_t M, int N, int K, int chunks, float* partial,
                     float* partial_bias, int64_t chunk_stride, void* stream);

/* --------------------------------------------------------------------------
 * a12 (training)  encoder self-attention on the packed projection output
 *   rl4co/models/nn/attention.py:110-134 (rearrange to heads + scaled_dot_product_attention)
 * qkv [B,N,384] bf16 = per node (q | k | v), each 8 heads x 16 dims. forward: out [B,N,128] bf16
 * (heads concatenated) and lse [B,8,N] fp32 (log2-domain log-sum-exp of the scaled scores);
 * backward: dqkv [B,N,384] bf16 from dout [B,N,128] and the forward's own out (the softmax
 * backward's row term sum_keys P dP is taken as sum_d dout out). rl4co_attn_bwd: N <= rl4co_attn_max_nodes() (one
 * workgroup holds an instance's keys); rl4co_attn_fwd: N <= rl4co_attn_wide_max_nodes() (beyond rl4co_attn_max_nodes()
 * the keys stream through LDS with an online softmax, r06).
 * rl4co_attn_bwd_wide (r06): the backward for any N <= rl4co_attn_wide_max_nodes(): one workgroup per (instance, half of
 * the heads, chunk of 128 keys); d k / d v of a chunk are complete in its workgroup, the chunks' shares of d q go through
 * `dq_partial` — a caller-provided fp32 workspace [ceil(N / 128), B, N, 128] — and are summed in chunk order.
 * -------------------------------------------------------------------------- */
int rl4co_attn_fwd(int dtype, const void* qkv, int B, int N, void* out, float* lse, void* stream);
int rl4co_attn_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, int B, int N,
                        void* dqkv, void* stream);
int rl4co_attn_bwd_wide(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, int B, int N,
                        void* dqkv, float* dq_partial, void* stream);
int rl4co_attn_max_nodes(void);
int rl4co_attn_wide_max_nodes(void);

/* --------------------------------------------------------------------------
 * (r06) The decoder's masked glimpse attention over ALL steps of given trajectories at once, forward and backward —
 * the inner multi-head attention of PointerAttention (rl4co/models/nn/attention.py:255-296 with the action mask,
 * models/zoo/am/decoder.py:150-190) as the dense re-evaluation of known actions uses it (`evaluate` decoding with
 * autograd: utils/decoding.py:448-461; rl/ppo/ppo.py:128-170): T step queries per trajectory against the N node keys of
 * its instance. 16-bit operands (dtype argument), fp32 softmax; heads = softmax_keys(q k^T / 4, masked) v per head.
 *   q    [B, T] rows of 128 (8 heads x 16), `q_stride` elements apart
 *   kv   [B_inst, N] rows of k 128 | v 128, `kv_stride` elements apart; trajectory b reads instance b % B_inst
 *   mask [B, T, mask_words] words of 32 keys, bit j = node j feasible at that step (rl4co_env_replay's mask_bits);
 *        mask_words % 4 == 0; NULL = every key. Every query needs at least one feasible key.
 *   forward : out [B, T, 128], lse [B, 8, T] (log2 domain)
 *   backward: dq [B, T, 128]; dkv [B_inst, N, dk 128 | dv 128] summed over steps and starts; `dq_partial` = fp32
 *             workspace [rl4co_cross_attn_chunks(N), B, T, 128]
 * -------------------------------------------------------------------------- */
typedef struct rl4co_cross_attn_args {
  int32_t B, B_inst, T, N;
  const void* q;
  int64_t q_stride;
  const void* kv;
  int64_t kv_stride;
  const uint32_t* mask;
  int32_t mask_words;
  int32_t reserved0;
  void* out;
  float* lse;
  const void* dout; /* backward only from here */
  void* dq;
  void* dkv;
  float* dq_partial;
} rl4co_cross_attn_args;
int rl4co_cross_attn_fwd(int dtype, const rl4co_cross_attn_args* args, void* stream);
int rl4co_cross_attn_bwd(int dtype, const rl4co_cross_attn_args* args, void* stream);
int rl4co_cross_attn_chunks(int N);

/* --------------------------------------------------------------------------
 * (r06) log p(a_t | s_t) of GIVEN actions from the pointer's raw logits (glimpse . logit_key, before the 1 / sqrt(128)),
 * every step of every trajectory as one row — the tail of the dense re-evaluation:
 *   u = raw / sqrt(128); z = tanh_clipping * tanh(u) (0: z = u); z = -inf where infeasible; z /= temperature;
 *   logp = log_softmax(z)[action]        rl4co/models/nn/attention.py:291-293, utils/decoding.py:169-188, :381
 * raw [rows, N] fp32; mask_bits [rows, mask_words] (bit j = node j feasible; NULL = all, i.e. mask_logits = False);
 * actions [rows]; forward -> logp [rows], lse [rows]; backward: d_raw [rows, N] from grad_logp [rows] and the saved lse.
 * err: RL4CO_EBIT_NAN_LOGIT / RL4CO_EBIT_INFEASIBLE (an action out of range), may be NULL.
 * -------------------------------------------------------------------------- */
int rl4co_logit_logp_fwd(const float* raw, const uint32_t* mask_bits, int mask_words, const int64_t* actions, int64_t rows, int N,
                         float tanh_clipping, float temperature, float* logp, float* lse, int32_t* err, void* stream);
int rl4co_logit_logp_bwd(const float* raw, const uint32_t* mask_bits, int mask_words, const int64_t* actions, const float* lse,
                         const float* grad_logp, int64_t rows, int N, float tanh_clipping, float temperature, float* d_raw,
                         void* stream);

/* --------------------------------------------------------------------------
 * a12 (inference, large graphs)  MultiHeadAttention.forward   rl4co/models/nn/attention.py:110-134
 * out[b, i, 16 h ..] = softmax_j(q_h[i] . k_h[j] / 4) v_h[j] on the packed projection output qkv [B,N,384]
 * (q | k | v, 8 heads x 16), bf16 -> out [B,N,128] bf16, any N: keys / values stream through LDS in blocks of
 * 64 nodes with a running (max, sum, output) per query — the N x N score matrix is never materialised
 * (csrc/am_attn_flash.hip). Serves the encoder beyond rl4co_am_encoder_max_nodes() (BASELINE configs[4]).
 * -------------------------------------------------------------------------- */
int rl4co_attn_flash(int dtype, const void* qkv, int B, int N, void* out, void* stream);
/* The same with q already in the exp2 domain (1/4 log2 e folded into the packed W_q: rl4co_am_encoder_tokens16) and,
 * bound != NULL, per (instance, head) the maxima over the nodes of |q_h|^2 and |k_h|^2 ([B,8,2] fp32): heads whose
 * product stays below 48^2 take the max-free softmax path (bf16 only; every |score| is then below 48 by Cauchy-Schwarz). */
int rl4co_attn_flash_pre(int dtype, const void* qkv, const float* bound, int B, int N, void* out, void* stream);

/* --------------------------------------------------------------------------
 * a19  select_start_nodes        rl4co/utils/ops.py:128-161
 * out[s*B + b] = s % num_loc (+1 for depot environments), s-major.
 * -------------------------------------------------------------------------- */
int rl4co_select_start_nodes(int64_t* out, int B, int num_starts, int num_loc, int has_depot,
                             void* stream);

/* --------------------------------------------------------------------------
 * N3  state augmentation                      rl4co/data/transforms.py:16-87, 105-151
 * Both kernels read the B instances ONCE and write the aug-major layout [A*B, N, 2] (row a*B + b) that the
 * multistart rollout consumes — what batchify (utils/ops.py:10-30) followed by the transform produces in the reference.
 *
 * dihedral8: the 8 symmetries of the unit square in the reference's order (transforms.py:27-37):
 *   (x,y) (1-x,y) (x,1-y) (1-x,1-y) (y,x) (1-y,x) (y,1-x) (1-y,1-x); xy [B,N,2] f32 -> out [8*B,N,2]. Bit-exact.
 * symmetric: rotation by phi_r about (offset, offset) and an axis swap where phi_r > 2 pi (transforms.py:49-69), one
 *   angle per OUTPUT row r. The caller passes cos(phi), sin(phi) [A*B] f32 and the swap flags [A*B] u8 (the angles are
 *   host-side RNG state: transforms.py:81 draws them from torch's global generator); the kernel does the fp32
 *   arithmetic in the reference's order, x' = cos*x - sin*y, y' = sin*x + cos*y (no fused multiply-add).
 * -------------------------------------------------------------------------- */
int rl4co_augment_dihedral8_f32(const float* xy, int B, int N, float* out, void* stream);
int rl4co_augment_symmetric_f32(const float* xy, const float* cos_phi, const float* sin_phi, const uint8_t* swap_axes,
                                int B, int A, int N, float offset, float* out, void* stream);

/* --------------------------------------------------------------------------
 * N3  POMO evaluation epilogue                rl4co/models/zoo/pomo/model.py:112-140
 *                                             (unbatchify / gather_by_index: rl4co/utils/ops.py:33-66)
 * reward [S*A*B] f32 and actions [S*A*B, T] i64 of a multistart rollout over an augmented batch, rows ordered
 * (s * A + a) * B + b. One launch computes
 *   max_reward[b,a]      = max_s reward            best_start[b,a] = its first arg-max (torch.max's tie rule)
 *   max_aug_reward[b]    = max_a max_reward[b,a]   best_aug[b]     = its first arg-max
 *   best_ms_actions[b,a] = actions of (best_start[b,a], a, b)          [B,A,T]
 *   best_aug_actions[b]  = actions of (best_start[b,a*], a* = best_aug[b], b)   [B,T]
 * Any output pointer may be NULL; actions may be NULL when no action output is asked for. A = 1 and / or S = 1 are
 * the degenerate cases of the same reduction.
 * -------------------------------------------------------------------------- */
int rl4co_pomo_best(const float* reward, const int64_t* actions, int A, int S, int B, int T, float* max_reward,
                    int64_t* best_start, float* max_aug_reward, int64_t* best_aug, int64_t* best_ms_actions,
                    int64_t* best_aug_actions, void* stream);

/* --------------------------------------------------------------------------
 * N2 / a10  instance generation on the device
 *        rl4co/envs/common/utils.py:34-62 (get_sampler -> Uniform), tsp/generator.py:49-58,
 *        cvrp/generator.py:114-140
 * out[i] = low + (high - low) * u_i, u_i = k_i / 2^24 from Philox4x32-10 keyed by `seed` (counter = element block,
 * stream_id) — U(low, high) like the reference's sampler, drawn straight into HBM in one launch (the reference draws on
 * the host and uploads). mode 1 = CVRP demands: (trunc(v) + 1) / capacity, i.e. `(U(min-1, max-1).int() + 1) / cap`.
 * A different generator than torch's: the CPU stream of the reference is reproduced by the host mirror
 * (envs.*Generator(device="cpu")), not by this kernel; this one is pinned to oracle_uniform_f32 bit for bit.
 * -------------------------------------------------------------------------- */
int rl4co_uniform_f32(float* out, int64_t n, float low, float high, uint64_t seed, uint32_t stream_id, int mode,
                      float capacity, void* stream);

/* --------------------------------------------------------------------------
 * Micro-benchmark helper: HBM read stream (float4 grid-stride sum), used by
 * bench.py to report the achievable-copy ceiling next to the 8 TB/s spec.
 * -------------------------------------------------------------------------- */
int rl4co_hbm_read_probe(const void* src, int64_t bytes, float* sink, void* stream);

/* --------------------------------------------------------------------------
 * Deterministic-math probe: y[i] = f(x[i]) on the device, f = 0 exp / 1 log / 2 tanh of
 * csrc/rl4co_math.h (the fp32 polynomials behind torch.tanh / log_softmax / exp in
 * utils/decoding.py:169-188). tests/test_math.py checks device == host bit for bit and
 * both against float64 libm.
 * -------------------------------------------------------------------------- */
int rl4co_math_probe_f32(int fn, const float* x, int64_t n, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RL4CO_AMD_H */
