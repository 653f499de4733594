/*
 * rollout_ref.c — ORACLE (test infrastructure, NOT product): plain-C CPU restatement of the
 * hot path with a SPECIFIED fp32 operation order.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file. Nothing under rl4co_amd/ links or loads it.
 *
 * Two jobs:
 *  1. get_tour_length / get_reward (rl4co/utils/ops.py:77-90, envs/routing/tsp/env.py:150-156,
 *     envs/routing/cvrp/env.py:138-147) restated to be BIT-IDENTICAL to the reference's ATen CPU
 *     arithmetic: per segment sqrt(fma(dy,dy,fl(dx*dx))) (ATen vector_norm on a size-2 dim), row
 *     sum in the 8-lane x 4-ILP cascade order of ATen's vectorized_inner_sum/multi_row_sum
 *     (SURVEY.md §8a-a2). tests/test_oracle_cpu.py pins this against torch itself.
 *  2. the AttentionModel decode loop (models/common/constructive/base.py:226-238 and callees —
 *     env_embeddings/context.py:105-149, zoo/am/decoder.py:128-193, nn/attention.py:274-320,
 *     utils/decoding.py:138-188,344-461, tsp/env.py:60-86, cvrp/env.py:66-96,126-136) in the
 *     specified operation order documented at the top of rl4co_amd/csrc/am_decode.hip (including
 *     the per-step list of nodes actually read: masked nodes contribute an exact 0). No fixed
 *     order can be bitwise equal to ATen's opaque SDPA/GEMM kernels, so the
 *     chain of evidence is: reference source == torch restatement (bit-exact, oracle/gen_golden.py)
 *     ~= this file (same actions except fp32 near-ties, log-likelihood <= 1e-5; tests/
 *     test_oracle_cpu.py) == HIP kernel (bit-exact, tests/test_gpu_*.py).
 *
 * The deterministic exp/log/tanh/Philox come from the product header rl4co_math.h (header-only,
 * plain C): oracle and kernel must round identically, so they share the definition.
 * Build: gcc -O2 -ffp-contract=off -mfma -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rl4co_amd.h"
#include "../rl4co_amd/csrc/rl4co_math.h"

#define D RL4CO_EMBED_DIM
#define H RL4CO_NUM_HEADS
#define DH (D / H)

/* ------------------------------------------------------------------------------------------ */
/* tour length                                                                                  */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  const float* locs;      /* [N,2] */
  const int64_t* actions; /* [T] */
  int prepend, n;
  const float* values;    /* non-NULL: element t is values[actions[t]] (OP reward: gathered prizes) */
} tour_view;

static inline void tv_point(const tour_view* tv, int t, float* x, float* y) {
  int node;
  if (tv->prepend) node = (t == 0) ? 0 : (int)tv->actions[t - 1];
  else node = (int)tv->actions[t];
  *x = tv->locs[2 * node];
  *y = tv->locs[2 * node + 1];
}

static inline float tv_seg(const tour_view* tv, int t) {
  if (tv->values) return tv->values[tv->actions[t]];
  float x0, y0, x1, y1;
  tv_point(tv, t, &x0, &y0);
  tv_point(tv, t + 1 == tv->n ? 0 : t + 1, &x1, &y1); /* torch.roll(-1) */
  const float dx = x1 - x0, dy = y1 - y0;
  return sqrtf(fmaf(dy, dy, dx * dx));
}

/* ATen multi_row_sum<float, 4> for ONE of the 8 vector lanes (element 8*v + lane of the row). */
static float lane_row_sum(const tour_view* tv, int lane8, int nvec) {
  enum { ILP = 4, LEVELS = 4 };
  const int size = nvec / ILP;
  int ceil_log2 = 0;
  while ((1LL << ceil_log2) < size) ++ceil_log2;
  int level_power = ceil_log2 / LEVELS;
  if (level_power < 4) level_power = 4;
  const int level_step = 1 << level_power;
  const int level_mask = level_step - 1;
  float acc[LEVELS][ILP];
  memset(acc, 0, sizeof(acc));
  int i = 0;
  for (; i + level_step <= size;) {
    for (int j = 0; j < level_step; ++j, ++i)
      for (int k = 0; k < ILP; ++k) acc[0][k] = acc[0][k] + tv_seg(tv, 8 * (i * ILP + k) + lane8);
    for (int j = 1; j < LEVELS; ++j) {
      for (int k = 0; k < ILP; ++k) {
        acc[j][k] = acc[j][k] + acc[j - 1][k];
        acc[j - 1][k] = 0.0f;
      }
      const int mask = level_mask << (j * level_power);
      if ((i & mask) != 0) break;
    }
  }
  for (; i < size; ++i)
    for (int k = 0; k < ILP; ++k) acc[0][k] = acc[0][k] + tv_seg(tv, 8 * (i * ILP + k) + lane8);
  for (int j = 1; j < LEVELS; ++j)
    for (int k = 0; k < ILP; ++k) acc[0][k] = acc[0][k] + acc[j][k];
  for (int v = size * ILP; v < nvec; ++v) acc[0][0] = acc[0][0] + tv_seg(tv, 8 * v + lane8);
  for (int k = 1; k < ILP; ++k) acc[0][0] = acc[0][0] + acc[0][k];
  return acc[0][0];
}

static int row_sums(const float* locs, const float* gather_values, const int64_t* actions, int B, int B_locs, int N, int T,
                    int prepend_depot, int negate, float* out) {
  for (int b = 0; b < B; ++b) {
    tour_view tv = {locs + (int64_t)(b % B_locs) * N * 2, actions + (int64_t)b * T, prepend_depot ? 1 : 0,
                    T + (prepend_depot ? 1 : 0), NULL};
    if (gather_values) { /* out[b] = sum_t values[b % B_locs][actions[b][t]]  (op/env.py:156-166) */
      tv.locs = NULL;
      tv.values = gather_values + (int64_t)(b % B_locs) * N;
      tv.prepend = 0;
      tv.n = T;
    }
    const int n = tv.n, nvec = n / 8;
    float fin = 0.0f;
    if (n < 8) {
      /* ATen takes scalar_inner_sum below one vector: row_sum<float> with 4 ILP partials,
       * leftovers into partial 0, then partials folded in order */
      float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      const int g = n / 4;
      for (int i = 0; i < g; ++i)
        for (int k = 0; k < 4; ++k) p[k] = p[k] + tv_seg(&tv, 4 * i + k);
      for (int k = 4 * g; k < n; ++k) p[0] = p[0] + tv_seg(&tv, k);
      for (int k = 1; k < 4; ++k) p[0] = p[0] + p[k];
      out[b] = negate ? -p[0] : p[0];
      continue;
    }
    for (int k = nvec * 8; k < n; ++k) fin = fin + tv_seg(&tv, k); /* scalar tail first */
    for (int l = 0; l < 8; ++l) fin = fin + lane_row_sum(&tv, l, nvec);
    out[b] = negate ? -fin : fin;
  }
  return 0;
}

int oracle_tour_length_f32(const float* locs, const int64_t* actions, int B, int B_locs, int N, int T,
                           int prepend_depot, int negate, float* out) {
  return row_sums(locs, NULL, actions, B, B_locs, N, T, prepend_depot, negate, out);
}

/* OPEnv._get_reward (op/env.py:156-166): prize.gather(1, actions).sum(-1), ATen's inner-dim sum order */
int oracle_gather_sum_f32(const float* values, const int64_t* actions, int B, int B_values, int N, int T, float* out) {
  return row_sums(NULL, values, actions, B, B_values, N, T, 0, 0, out);
}

/* ------------------------------------------------------------------------------------------ */
/* environment transitions (bit/byte work + one fp32 add, mul, compare)                         */
/* ------------------------------------------------------------------------------------------ */

int oracle_tsp_step(const int64_t* action, uint8_t* mask, int64_t* first, int64_t* cur, int64_t* step_i,
                    uint8_t* done, int B, int N) {
  for (int b = 0; b < B; ++b) {
    const int64_t a = action[b];
    if (a < 0 || a >= N) return 1;
    if (step_i[b] == 0) first[b] = a; /* tsp/env.py:63 */
    cur[b] = a;
    mask[(int64_t)b * N + a] = 0;
    step_i[b] += 1;
    int any = 0;
    for (int j = 0; j < N; ++j) any |= mask[(int64_t)b * N + j];
    done[b] = any ? 0 : 1;
  }
  return 0;
}

static void cvrp_mask_row(const float* dem, float used, float cap, const uint8_t* vis, int64_t cur, uint8_t* mk,
                          int N) {
  const float thr = cap + 1e-5f; /* cvrp/env.py:128 */
  int any_feasible = 0;
  for (int j = 1; j < N; ++j) {
    const int masked = (vis[j] != 0) || (dem[j - 1] + used > thr);
    mk[j] = masked ? 0 : 1;
    any_feasible |= !masked;
  }
  mk[0] = ((cur == 0) && any_feasible) ? 0 : 1; /* cvrp/env.py:134-135 */
}

int oracle_cvrp_step(const int64_t* action, const float* demand, float* used, const float* cap, uint8_t* visited,
                     int64_t* cur, uint8_t* mask, uint8_t* done, int B, int B_inst, int N) {
  for (int b = 0; b < B; ++b) {
    const float* dem = demand + (int64_t)(b % B_inst) * (N - 1);
    uint8_t* vis = visited + (int64_t)b * N;
    if (action) {
      const int64_t a = action[b];
      if (a < 0 || a >= N) return 1;
      int64_t di = a - 1;
      if (di < 0) di = 0;
      if (di > N - 2) di = N - 2;
      used[b] = (used[b] + dem[di]) * (a != 0 ? 1.0f : 0.0f); /* cvrp/env.py:76 */
      cur[b] = a;
      vis[a] = 1;
      int all = 1;
      for (int j = 0; j < N; ++j) all &= vis[j] != 0;
      done[b] = all ? 1 : 0;
    }
    cvrp_mask_row(dem, used[b], cap[b], vis, cur[b], mask + (int64_t)b * N, N);
  }
  return 0;
}

/* ---- orienteering problem (envs/routing/op/env.py) ------------------------------------------- */
static inline float op_dist(const float* locs, int i, int j) { /* (locs[j] - locs[i]).norm(p=2, dim=-1) */
  const float dx = locs[2 * j] - locs[2 * i], dy = locs[2 * j + 1] - locs[2 * i + 1];
  return sqrtf(fmaf(dy, dy, dx * dx));
}

/* op/env.py:118-122: max_length[b][j] = (max_length[b] - |depot - loc_j|) - 1e-6 */
int oracle_op_max_length(const float* locs, const float* max_length, int B, int N, float* out) {
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < N; ++j) out[(int64_t)b * N + j] = (max_length[b] - op_dist(locs + (int64_t)b * N * 2, j, 0)) - 1e-6f;
  return 0;
}

/* op/env.py:137-154 */
static void op_mask_row(const float* locs, const float* maxlen, float tour, const uint8_t* vis, int cur, uint8_t* mk, int N) {
  for (int j = 0; j < N; ++j) {
    const int exceeds = tour + op_dist(locs, cur, j) > maxlen[j];
    mk[j] = (vis[j] || vis[0] || exceeds) ? 0 : 1;
  }
  mk[0] = 1; /* the depot can always be visited */
}

/* op/env.py:67-98 (action == NULL: mask only, as at reset) */
int oracle_op_step(const int64_t* action, const float* locs, const float* maxlen, float* tour_length, uint8_t* visited,
                   int64_t* cur, int64_t* step_i, uint8_t* mask, uint8_t* done, int B, int B_inst, int N) {
  for (int b = 0; b < B; ++b) {
    const float* lc = locs + (int64_t)(b % B_inst) * N * 2;
    const float* ml = maxlen + (int64_t)(b % B_inst) * N;
    uint8_t* vis = visited + (int64_t)b * N;
    if (action) {
      const int64_t a = action[b];
      if (a < 0 || a >= N) return 1;
      tour_length[b] = tour_length[b] + op_dist(lc, (int)cur[b], (int)a);
      vis[a] = 1;
      done[b] = (a == 0 && step_i[b] > 0) ? 1 : 0;
      step_i[b] += 1;
      cur[b] = a;
    }
    op_mask_row(lc, ml, tour_length[b], vis, (int)cur[b], mask + (int64_t)b * N, N);
  }
  return 0;
}

/* ---- CVRP with time windows (envs/routing/cvrptw/env.py:83-113) -------------------------------- */
static float cvrptw_time(const float* locs, const float* tw, const float* dur, float time, int cur, int a) {
  const float d = op_dist(locs, cur, a); /* td["distances"][a]: measured from the node the vehicle stands on */
  const float served = fmaxf(time + d, tw[2 * a]) + dur[a];
  return (a != 0 ? 1.0f : 0.0f) * served; /* cvrptw/env.py:108-110 */
}

static void cvrptw_mask_row(const float* dem, float used, float cap, const uint8_t* vis, int cur, const float* locs,
                            const float* tw, float time, uint8_t* mk, int N) {
  cvrp_mask_row(dem, used, cap, vis, cur, mk, N);
  for (int j = 0; j < N; ++j) /* cvrptw/env.py:91-95 */
    if (!(time + op_dist(locs, cur, j) <= tw[2 * j + 1])) mk[j] = 0;
}

int oracle_cvrptw_step(const int64_t* action, const float* demand, const float* locs, const float* time_windows,
                       const float* durations, float* used, const float* cap, float* time, uint8_t* visited, int64_t* cur,
                       uint8_t* mask, uint8_t* done, int B, int B_inst, int N) {
  for (int b = 0; b < B; ++b) {
    const int ib = b % B_inst;
    const float* dem = demand + (int64_t)ib * (N - 1);
    const float* lc = locs + (int64_t)ib * N * 2;
    const float* tw = time_windows + (int64_t)ib * N * 2;
    const float* du = durations + (int64_t)ib * N;
    uint8_t* vis = visited + (int64_t)b * N;
    if (action) {
      const int64_t a = action[b];
      if (a < 0 || a >= N) return 1;
      time[b] = cvrptw_time(lc, tw, du, time[b], (int)cur[b], (int)a);
      int64_t di = a - 1;
      if (di < 0) di = 0;
      if (di > N - 2) di = N - 2;
      used[b] = (used[b] + dem[di]) * (a != 0 ? 1.0f : 0.0f);
      cur[b] = a;
      vis[a] = 1;
      int all = 1;
      for (int j = 0; j < N; ++j) all &= vis[j] != 0;
      done[b] = all ? 1 : 0;
    }
    cvrptw_mask_row(dem, used[b], cap[b], vis, (int)cur[b], lc, tw, time[b], mask + (int64_t)b * N, N);
  }
  return 0;
}

/* ---- prize-collecting TSP (envs/routing/pctsp/env.py) ------------------------------------------ */
static void pctsp_mask_row(float prize, const uint8_t* vis, uint8_t* mk, int N) { /* pctsp/env.py:141-148 */
  int unvisited = 0;
  for (int j = 1; j < N; ++j) {
    mk[j] = (vis[j] || vis[0]) ? 0 : 1;
    unvisited |= !vis[j];
  }
  mk[0] = ((prize < 1.0f) && unvisited) ? 0 : 1;
}

int oracle_pctsp_step(const int64_t* action, const float* real_prize, float* prize, uint8_t* visited, int64_t* cur,
                      int64_t* step_i, uint8_t* mask, uint8_t* done, int B, int B_inst, int N) {
  for (int b = 0; b < B; ++b) {
    const float* rp = real_prize + (int64_t)(b % B_inst) * N;
    uint8_t* vis = visited + (int64_t)b * N;
    if (action) {
      const int64_t a = action[b];
      if (a < 0 || a >= N) return 1;
      prize[b] = prize[b] + rp[a];
      vis[a] = 1;
      done[b] = (step_i[b] > 0 && a == 0) ? 1 : 0;
      step_i[b] += 1;
      cur[b] = a;
    }
    pctsp_mask_row(prize[b], vis, mask + (int64_t)b * N, N);
  }
  return 0;
}

/* ---- pickup and delivery (envs/routing/pdp/env.py:64-99) ---------------------------------------- */
static void pdp_transition(int a, uint8_t* avail, uint8_t* tod, uint8_t* mk, int N) {
  const int n = N - 1;
  avail[a] = 0;
  tod[(a + n / 2) % (n + 1)] = 1;
  for (int j = 0; j < N; ++j) mk[j] = (avail[j] && tod[j]) ? 1 : 0;
}

int oracle_pdp_step(const int64_t* action, uint8_t* available, uint8_t* to_deliver, int64_t* cur, int64_t* step_i,
                    uint8_t* mask, uint8_t* done, int B, int N) {
  for (int b = 0; b < B; ++b) {
    uint8_t* av = available + (int64_t)b * N;
    uint8_t* td = to_deliver + (int64_t)b * N;
    uint8_t* mk = mask + (int64_t)b * N;
    if (action) {
      const int64_t a = action[b];
      if (a < 0 || a >= N) return 1;
      pdp_transition((int)a, av, td, mk, N);
      int left = 0;
      for (int j = 0; j < N; ++j) left |= av[j];
      done[b] = left ? 0 : 1;
      step_i[b] += 1;
      cur[b] = a;
    } else {
      for (int j = 0; j < N; ++j) mk[j] = (av[j] && td[j]) ? 1 : 0;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* AttentionModel decode loop in the specified operation order                                  */
/* ------------------------------------------------------------------------------------------ */

/* pairwise butterfly sum == the value every lane holds after xor-shuffle adds over n lanes */
static float tree_sum(const float* x, int n) {
  if (n == 1) return x[0];
  return tree_sum(x, n / 2) + tree_sum(x + n / 2, n / 2);
}

/* IEEE binary16 -> binary32, exact (what v_cvt_f32_f16 does), subnormals and inf / nan included */
static inline float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  if (exp == 0) {
    if (man == 0) return rl4co_bits_to_float(sign);
    int e = -1;
    do { man <<= 1; ++e; } while (!(man & 0x400u));
    return rl4co_bits_to_float(sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13));
  }
  if (exp == 31) return rl4co_bits_to_float(sign | 0x7f800000u | (man << 13));
  return rl4co_bits_to_float(sign | ((exp + 112u) << 23) | (man << 13));
}

static inline float cache_at(const void* base, int dtype, int64_t idx) {
  if (dtype == RL4CO_DT_BF16) return rl4co_bits_to_float((uint32_t)((const uint16_t*)base)[idx] << 16);
  if (dtype == RL4CO_DT_F16) return half_bits_to_float(((const uint16_t*)base)[idx]);
  return ((const float*)base)[idx];
}

/* Same argument block as rl4co_am_decode, with HOST pointers. `row_groups` = the G reported by
 * rl4co_am_decode_row_groups(cache_dtype). Returns 0, or 1 on a bad argument. */
int oracle_am_decode(const rl4co_am_decode_args* a, int row_groups) {
  const int N = a->N, G = row_groups;
  const int EPL = a->cache_dtype != RL4CO_DT_F32 ? 8 : 4;
  const int LPR = D / EPL, LPH = DH / EPL;
  if (G < 1 || G > 64 || (G & (G - 1))) return 1;
  float* sc = (float*)malloc(sizeof(float) * (size_t)N * H);
  float* lg = (float*)malloc(sizeof(float) * (size_t)N);
  uint8_t* mk = (uint8_t*)malloc((size_t)N);
  uint8_t* vis = (uint8_t*)malloc((size_t)N);
  uint8_t* tod = (uint8_t*)malloc((size_t)N); /* PDP: to_deliver (vis holds `available`) */
  float* og = (float*)malloc(sizeof(float) * (size_t)G * D);
  float* lgp = (float*)malloc(sizeof(float) * (size_t)G * H);
  int* fl = (int*)malloc(sizeof(int) * (size_t)N);
  const float sqrt_d = 11.3137084989847604f;
  const float neg_inf = -INFINITY;
  const int single = a->max_steps == 1;
  uint32_t errbits_all = 0;

  for (int r = 0; r < a->B; ++r) {
    const int cb = r % a->B_inst;
    const int64_t cbase = (int64_t)cb * a->kvl_batch_stride;
    const float* ctxc = a->unfold ? NULL : (const float*)a->ctx_cur + (int64_t)cb * N * D;
    const float* ctxf = (a->env == RL4CO_ENV_TSP && !a->unfold) ? (const float*)a->ctx_first + (int64_t)cb * N * D : NULL;
    uint8_t* gmask = a->action_mask + (int64_t)r * N;
    memcpy(mk, gmask, (size_t)N);
    if (a->env != RL4CO_ENV_TSP) memcpy(vis, a->visited + (int64_t)r * N, (size_t)N);
    if (a->env == RL4CO_ENV_PDP) memcpy(tod, a->to_deliver + (int64_t)r * N, (size_t)N);
    const int scalar_ctx = a->env == RL4CO_ENV_CVRP || a->env == RL4CO_ENV_OP || a->env == RL4CO_ENV_PCTSP ||
                           a->env == RL4CO_ENV_CVRPTW;
    const int tw_env = a->env == RL4CO_ENV_CVRPTW;
    const float* twlocs = tw_env ? a->locs + (int64_t)cb * N * 2 : NULL;
    const float* tw = tw_env ? a->time_windows + (int64_t)cb * N * 2 : NULL;
    const float* dur = tw_env ? a->durations + (int64_t)cb * N : NULL;
    float now = tw_env ? a->current_time[r] : 0.0f;
    int cur = (int)a->current_node[r];
    int first = a->env == RL4CO_ENV_TSP ? (int)a->first_node[r] : 0;
    long long step_i = (a->env != RL4CO_ENV_CVRP && a->env != RL4CO_ENV_CVRPTW) ? a->step_i[r] : 0;
    /* OP: `used` is the tour length so far and `cap` the longest tour that may still reach the depot
     * directly, max_length[.., 0]; the context scalar is cap - used in both environments */
    float used = scalar_ctx ? a->used_capacity[r] : 0.0f;
    const float* oplocs = a->env == RL4CO_ENV_OP ? a->locs + (int64_t)cb * N * 2 : NULL;
    const float* opmax = a->env == RL4CO_ENV_OP ? a->max_length + (int64_t)cb * N : NULL;
    const float cap = (a->env == RL4CO_ENV_CVRP || a->env == RL4CO_ENV_PCTSP || a->env == RL4CO_ENV_CVRPTW) ? a->vehicle_capacity[r]
                      : (a->env == RL4CO_ENV_OP ? opmax[0] : 0.0f); /* PCTSP: prize_required */
    const float* rprize = a->env == RL4CO_ENV_PCTSP ? a->demand + (int64_t)cb * N : NULL;
    const float* dem = (a->env == RL4CO_ENV_CVRP || a->env == RL4CO_ENV_CVRPTW) ? a->demand + (int64_t)cb * (N - 1) : NULL;
    int done = a->done[r] != 0;
    uint32_t errbits = 0;
    float ent_acc = 0.0f;
    int t = 0;
    int rows_read = 0;

    for (; t < a->max_steps && (!done || single); ++t) {
      /* query */
      float q[D];
      float cv[2 * D]; /* unfolded mode: the context vector the reference feeds project_context */
      if (a->unfold) {
        const float* hrow = a->node_embed + (int64_t)cb * N * D;
        if (a->env == RL4CO_ENV_TSP) {
          for (int d = 0; d < D; ++d) {
            cv[d] = step_i < 1 ? a->w_placeholder[d] : hrow[(int64_t)first * D + d];
            cv[D + d] = step_i < 1 ? a->w_placeholder[D + d] : hrow[(int64_t)cur * D + d];
          }
        } else {
          for (int d = 0; d < D; ++d) cv[d] = hrow[(int64_t)cur * D + d];
          cv[D] = cap - used;
        }
      }
      for (int d = 0; d < D; ++d) {
        const float qb = a->q_bias ? a->q_bias[(int64_t)cb * D + d] : 0.0f;
        float v;
        if (a->unfold) { /* one fma chain over the context dims, ascending, from 0; then + graph context */
          float acc = 0.0f;
          for (int k = 0; k < a->ctx_width; ++k) acc = fmaf(a->w_ctx_t[(int64_t)k * D + d], cv[k], acc);
          v = acc + qb;
        } else if (a->env == RL4CO_ENV_TSP) {
          if (step_i < 1) v = a->q_step0[d] + qb;
          else v = (ctxf[(int64_t)first * D + d] + ctxc[(int64_t)cur * D + d]) + qb;
        } else if (a->env == RL4CO_ENV_PDP) {
          v = ctxc[(int64_t)cur * D + d] + qb; /* context.py:232-243: the current node alone */
        } else {
          float rem = cap - used;
          if (a->env == RL4CO_ENV_PCTSP && !(rem > 0.0f)) rem = 0.0f; /* clamp(min=0), context.py:195 */
          v = fmaf(a->w_cap[d], rem, ctxc[(int64_t)cur * D + d]);
          if (tw_env) v = fmaf(a->w_time[d], now, v); /* context.py:152-166: second scalar, the current time */
          v = v + qb;
        }
        q[d] = v * 0.25f;
      }
      /* the nodes this step reads: feasible ones ascending (or all, if a mask flag is off) */
      int F = 0;
      for (int j = 0; j < N; ++j)
        if (!(a->mask_inner && a->mask_logits) || mk[j]) fl[F++] = j;
      rows_read += F;
      /* pass 1: scores, indexed by list position c */
      float m[H];
      for (int h = 0; h < H; ++h) m[h] = neg_inf;
      for (int c = 0; c < F; ++c) {
        const int j = fl[c];
        const int feas = !a->mask_inner || mk[j] != 0;
        for (int h = 0; h < H; ++h) {
          float part[4];
          for (int cc = 0; cc < LPH; ++cc) {
            float acc = 0.0f;
            for (int e = 0; e < EPL; ++e) {
              const int d = h * DH + cc * EPL + e;
              acc = fmaf(q[d], cache_at(a->glimpse_key, a->cache_dtype, cbase + (int64_t)j * a->kvl_row_stride + d),
                         acc);
            }
            part[cc] = acc;
          }
          const float s = feas ? tree_sum(part, LPH) : neg_inf;
          sc[c * H + h] = s;
          m[h] = fmaxf(m[h], s);
        }
      }
      /* pass 2: softmax-weighted values, per row group (c % G) then tree */
      for (int i = 0; i < G * D; ++i) og[i] = 0.0f;
      for (int i = 0; i < G * H; ++i) lgp[i] = 0.0f;
      for (int c = 0; c < F; ++c) {
        const int j = fl[c];
        const int g = c % G;
        for (int h = 0; h < H; ++h) {
          const float p = rl4co_expf(sc[c * H + h] - m[h]);
          lgp[g * H + h] = lgp[g * H + h] + p;
          for (int e = 0; e < DH; ++e) {
            const int d = h * DH + e;
            og[g * D + d] =
                fmaf(p, cache_at(a->glimpse_val, a->cache_dtype, cbase + (int64_t)j * a->kvl_row_stride + d),
                     og[g * D + d]);
          }
        }
      }
      float o[D];
      for (int h = 0; h < H; ++h) {
        float tmp[64];
        for (int g = 0; g < G; ++g) tmp[g] = lgp[g * H + h];
        const float rl = 1.0f / tree_sum(tmp, G); /* one division per head; heads = o * (1/l) */
        for (int e = 0; e < DH; ++e) {
          const int d = h * DH + e;
          for (int g = 0; g < G; ++g) tmp[g] = og[g * D + d];
          o[d] = tree_sum(tmp, G) * rl;
        }
      }
      if (a->unfold) { /* glimpse = project_out(heads), attention.py:287 */
        float gl[D];
        for (int d = 0; d < D; ++d) {
          float acc = 0.0f;
          for (int k = 0; k < D; ++k) acc = fmaf(a->w_out_t[(int64_t)k * D + d], o[k], acc);
          gl[d] = acc;
        }
        memcpy(o, gl, sizeof(gl));
      }
      /* pass 3: logits of the listed nodes */
      int nan_seen = 0;
      float zmax = neg_inf;
      for (int c = 0; c < F; ++c) {
        const int j = fl[c];
        float part[32];
        for (int cc = 0; cc < LPR; ++cc) {
          float acc = 0.0f;
          for (int e = 0; e < EPL; ++e) {
            const int d = cc * EPL + e;
            acc = fmaf(o[d], cache_at(a->logit_key, a->cache_dtype, cbase + (int64_t)j * a->kvl_row_stride + d), acc);
          }
          part[cc] = acc;
        }
        float z = tree_sum(part, LPR) / sqrt_d;
        if (z != z) nan_seen = 1;
        if (a->tanh_clipping > 0.0f) z = rl4co_tanhf(z) * a->tanh_clipping;
        if (a->mask_logits && mk[j] == 0) z = neg_inf;
        if (a->temperature != 1.0f) z = z / a->temperature;
        lg[c] = z;
        zmax = fmaxf(zmax, z);
      }
      if (nan_seen) errbits |= RL4CO_EBIT_NAN_LOGIT;
      /* log_softmax: 64 position-strided partial sums, then tree */
      float part64[64];
      for (int k = 0; k < 64; ++k) {
        float s = 0.0f;
        for (int c = k; c < F; c += 64) s = s + rl4co_expf(lg[c] - zmax);
        part64[k] = s;
      }
      const float lse = rl4co_logf(tree_sum(part64, 64));
      /* selection */
      const int64_t tcol = (int64_t)a->t0 + t;
      float best = neg_inf;
      int bc = -1;
      for (int k = 0; k < 64; ++k) part64[k] = 0.0f;
      float* alp = a->all_logps ? a->all_logps + ((int64_t)r * a->out_stride + tcol) * N : NULL;
      if (alp)
        for (int j = 0; j < N; ++j) alp[j] = neg_inf;
      for (int c = 0; c < F; ++c) {
        const int j = fl[c];
        const float lp = (lg[c] - zmax) - lse;
        lg[c] = lp;
        float key = lp;
        if (a->mode == RL4CO_DECODE_SAMPLE) {
          const float nz = a->exp_noise ? a->exp_noise[((int64_t)t * a->B + r) * N + j]
                                        : rl4co_exp1_noise(a->philox_seed ^ (a->philox_seed_dev ? *a->philox_seed_dev : 0ull), a->philox_offset + (uint64_t)tcol,
                                                           (uint32_t)r, (uint32_t)j);
          key = rl4co_expf(lp) / nz;
        }
        if (bc < 0 || key > best) { /* ascending c + strict '>' == lowest index among maxima */
          best = key;
          bc = c;
        }
        if (a->entropy && lp > neg_inf) part64[c % 64] = fmaf(rl4co_expf(lp), lp, part64[c % 64]);
        if (alp) alp[j] = lp;
      }
      if (a->entropy) ent_acc = ent_acc - tree_sum(part64, 64);
      int bi;
      float logp;
      if (a->mode == RL4CO_DECODE_EVALUATE) {
        bi = (int)a->forced_actions[(int64_t)r * a->out_stride + tcol];
        if (bi < 0 || bi >= N) {
          errbits |= RL4CO_EBIT_INFEASIBLE;
          bi = 0;
        }
        logp = neg_inf; /* a node outside the list is masked out */
        for (int c = 0; c < F; ++c)
          if (fl[c] == bi) logp = lg[c];
      } else {
        bi = bc >= 0 ? fl[bc] : 0;
        logp = bc >= 0 ? lg[bc] : neg_inf;
      }
      if (mk[bi] == 0) errbits |= RL4CO_EBIT_INFEASIBLE;
      if (!(logp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;
      a->actions[(int64_t)r * a->out_stride + tcol] = bi;
      a->logps[(int64_t)r * a->out_stride + tcol] = logp;
      /* environment transition */
      if (a->env == RL4CO_ENV_TSP) {
        if (step_i == 0) first = bi;
        cur = bi;
        mk[bi] = 0;
        step_i += 1;
        int any = 0;
        for (int j = 0; j < N; ++j) any |= mk[j];
        done = !any;
      } else if (a->env == RL4CO_ENV_CVRPTW) {
        now = cvrptw_time(twlocs, tw, dur, now, cur, bi);
        int di = bi - 1;
        if (di < 0) di = 0;
        if (di > N - 2) di = N - 2;
        used = (used + dem[di]) * (bi != 0 ? 1.0f : 0.0f);
        cur = bi;
        vis[bi] = 1;
        int all = 1;
        for (int j = 0; j < N; ++j) all &= vis[j] != 0;
        done = all;
        cvrptw_mask_row(dem, used, cap, vis, cur, twlocs, tw, now, mk, N);
      } else if (a->env == RL4CO_ENV_PDP) {
        pdp_transition(bi, vis, tod, mk, N);    /* pdp/env.py:64-80 */
        int left = 0;
        for (int j = 0; j < N; ++j) left |= vis[j];
        done = !left;                           /* pdp/env.py:83 */
        step_i += 1;
        cur = bi;
      } else if (a->env == RL4CO_ENV_PCTSP) {
        used = used + rprize[bi];               /* pctsp/env.py:66 */
        vis[bi] = 1;
        done = (step_i > 0) && (bi == 0);       /* pctsp/env.py:73 */
        step_i += 1;
        cur = bi;
        pctsp_mask_row(used, vis, mk, N);
      } else if (a->env == RL4CO_ENV_OP) {
        used = used + op_dist(oplocs, cur, bi); /* op/env.py:71-73 */
        vis[bi] = 1;
        done = (bi == 0) && (step_i > 0);       /* op/env.py:84 */
        step_i += 1;
        cur = bi;
        op_mask_row(oplocs, opmax, used, vis, cur, mk, N);
      } else {
        int di = bi - 1;
        if (di < 0) di = 0;
        if (di > N - 2) di = N - 2;
        used = (used + dem[di]) * (bi != 0 ? 1.0f : 0.0f);
        cur = bi;
        vis[bi] = 1;
        int all = 1;
        for (int j = 0; j < N; ++j) all &= vis[j] != 0;
        done = all;
        cvrp_mask_row(dem, used, cap, vis, cur, mk, N);
      }
    }
    if (!single && !done && t >= a->max_steps) errbits |= RL4CO_EBIT_MAX_STEPS;
    memcpy(gmask, mk, (size_t)N);
    if (a->env != RL4CO_ENV_TSP) memcpy(a->visited + (int64_t)r * N, vis, (size_t)N);
    if (a->env == RL4CO_ENV_PDP) memcpy(a->to_deliver + (int64_t)r * N, tod, (size_t)N);
    a->current_node[r] = cur;
    a->done[r] = done ? 1 : 0;
    if (a->env == RL4CO_ENV_TSP) a->first_node[r] = first;
    if (a->env != RL4CO_ENV_CVRP && !tw_env) a->step_i[r] = step_i;
    if (tw_env) a->current_time[r] = now;
    if (scalar_ctx) a->used_capacity[r] = used;
    if (a->n_steps) a->n_steps[r] = t;
    if (a->steps_summary) {
      if (t > a->steps_summary[0]) a->steps_summary[0] = t;
      a->steps_summary[1] += t;
      { uint64_t rows; memcpy(&rows, a->steps_summary + 2, 8); rows += (uint64_t)rows_read; memcpy(a->steps_summary + 2, &rows, 8); }
    }
    if (a->entropy) a->entropy[r] += ent_acc;
    errbits_all |= errbits;
  }
  if (a->err) *a->err |= (int32_t)errbits_all;
  free(sc); free(lg); free(mk); free(vis); free(tod); free(og); free(lgp); free(fl);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Rounding-model oracle of the multistart MFMA decode variant (csrc/am_decode_ms.hip)          */
/* ------------------------------------------------------------------------------------------ */
/* The MS kernel (RL4CO_VARIANT_MS, every POMO rollout) feeds the matrix cores bf16 operands: the query (scaled by
 * 1/4 * log2 e), the softmax numerators and the glimpse are ROUNDED TO bf16 (round-to-nearest-even) before their
 * products; accumulation, softmax sums, clipping and log-softmax are fp32, with the hardware's exp2 / log / rcp.
 * Those transcendental instructions are not correctly rounded and the MFMA's internal summation order is not
 * architecturally specified, so no CPU program can promise the kernel's last bit. What CAN be specified are the
 * rounding points above: this function restates the step with exactly those bf16 roundings and plain fp32 arithmetic
 * (libm exp2f / expf / logf, ascending-k fma chains). The kernel must follow it to fp32-rounding-noise level —
 * tests/test_gpu_decode_ms.py: identical actions except at near-ties, per-step log-probabilities on the
 * common prefix (tolerance 5e-3; measured max 2.1e-3, mean < 1e-6, 100 % identical trajectories) — ten times tighter than the previous comparison against the fp32-query
 * streaming kernel (0.05), which mixed the model's bf16 error into the tolerance. TSP / CVRP, bf16 planes. */
static inline float bf16_round(float x) {
  uint32_t u = rl4co_float_to_bits(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return rl4co_bits_to_float(u & 0xffff0000u);
}

/* fp32 -> binary16 -> fp32, round to nearest even, overflow to infinity (v_cvt_f16_f32): the fp16 build of the MS kernel
 * (csrc/am_decode_ms_f16.hip) rounds at the same three points with this instead of bf16_round */
static inline float half_round(float x) {
  const uint32_t u = rl4co_float_to_bits(x), sign = u & 0x80000000u, mag = u & 0x7fffffffu;
  if (mag > 0x7f800000u) return x;                                  /* NaN */
  if (mag >= 0x477ff000u) return rl4co_bits_to_float(sign | 0x7f800000u); /* >= 65520: rounds to infinity */
  if (mag < 0x38800000u) {                                          /* below 2^-14: subnormal half, spacing 2^-24 */
    const float ax = rl4co_bits_to_float(mag);
    const float r = (ax + 12582912.0f * 5.9604644775390625e-8f) - 12582912.0f * 5.9604644775390625e-8f; /* 1.5 * 2^23 * 2^-24 */
    return rl4co_bits_to_float(sign | rl4co_float_to_bits(r));
  }
  uint32_t v = mag + 0xfffu + ((mag >> 13) & 1u);                    /* 13 mantissa bits dropped */
  return rl4co_bits_to_float(sign | (v & 0xffffe000u));
}
static inline float round16(float x, int dtype) { return dtype == RL4CO_DT_F16 ? half_round(x) : bf16_round(x); }

int oracle_am_decode_ms(const rl4co_am_decode_args* a) {
  const int N = a->N;
  if ((a->cache_dtype != RL4CO_DT_BF16 && a->cache_dtype != RL4CO_DT_F16) || N > 128) return 1;
  const int dt16 = a->cache_dtype;
  const int env = a->env;
  const int tsp = env == RL4CO_ENV_TSP;
  const int cvrp_like = env == RL4CO_ENV_CVRP || env == RL4CO_ENV_CVRPTW;
  const int tw_env = env == RL4CO_ENV_CVRPTW;
  const int scalar_ctx = !tsp && env != RL4CO_ENV_PDP;
  const int has_step_i = !cvrp_like;
  const float neg_inf = -INFINITY;
  const float log2e = 1.44269504088896341f, sqrt_d = 11.3137084989847604f;
  const int single = a->max_steps == 1;
  const float inv_temp = 1.0f / a->temperature;
  const float clip_over_temp = a->tanh_clipping * inv_temp;
  uint32_t errbits_all = 0;
  float* sc = (float*)malloc(sizeof(float) * (size_t)N * H);
  float* z = (float*)malloc(sizeof(float) * (size_t)N);
  uint8_t* mk = (uint8_t*)malloc((size_t)N);
  uint8_t* vis = (uint8_t*)malloc((size_t)N);
  uint8_t* tod = (uint8_t*)malloc((size_t)N);
  for (int r = 0; r < a->B; ++r) {
    const int cb = r % a->B_inst;
    const int64_t cbase = (int64_t)cb * a->kvl_batch_stride;
    const float* ctxc = (const float*)a->ctx_cur + (int64_t)cb * N * D;
    const float* ctxf = tsp ? (const float*)a->ctx_first + (int64_t)cb * N * D : NULL;
    uint8_t* gmask = a->action_mask + (int64_t)r * N;
    memcpy(mk, gmask, (size_t)N);
    if (!tsp) memcpy(vis, a->visited + (int64_t)r * N, (size_t)N);
    if (env == RL4CO_ENV_PDP) memcpy(tod, a->to_deliver + (int64_t)r * N, (size_t)N);
    int cur = (int)a->current_node[r];
    int first = tsp ? (int)a->first_node[r] : 0;
    long long step_i = has_step_i ? a->step_i[r] : 0;
    float used = scalar_ctx ? a->used_capacity[r] : 0.0f;
    const float* oplocs = env == RL4CO_ENV_OP ? a->locs + (int64_t)cb * N * 2 : NULL;
    const float* opmax = env == RL4CO_ENV_OP ? a->max_length + (int64_t)cb * N : NULL;
    const float* twlocs = tw_env ? a->locs + (int64_t)cb * N * 2 : NULL;
    const float* tw = tw_env ? a->time_windows + (int64_t)cb * N * 2 : NULL;
    const float* dur = tw_env ? a->durations + (int64_t)cb * N : NULL;
    float now = tw_env ? a->current_time[r] : 0.0f;
    const float cap = (cvrp_like || env == RL4CO_ENV_PCTSP) ? a->vehicle_capacity[r] : (env == RL4CO_ENV_OP ? opmax[0] : 0.0f);
    const float* rprize = env == RL4CO_ENV_PCTSP ? a->demand + (int64_t)cb * N : NULL;
    const float* dem = cvrp_like ? a->demand + (int64_t)cb * (N - 1) : NULL;
    int done = a->done[r] != 0;
    uint32_t errbits = 0;
    int t = 0;
    for (; t < a->max_steps && (!done || single); ++t) {
      const int64_t tcol = (int64_t)a->t0 + t;
      float q[D];
      for (int d = 0; d < D; ++d) {
        const float qb = a->q_bias ? a->q_bias[(int64_t)cb * D + d] : 0.0f;
        float v;
        if (tsp) {
          v = step_i < 1 ? a->q_step0[d] + qb : (ctxf[(int64_t)first * D + d] + ctxc[(int64_t)cur * D + d]) + qb;
        } else if (env == RL4CO_ENV_PDP) {
          v = ctxc[(int64_t)cur * D + d] + qb;
        } else {
          float rem = cap - used;
          if (env == RL4CO_ENV_PCTSP && !(rem > 0.0f)) rem = 0.0f;
          v = fmaf(a->w_cap[d], rem, ctxc[(int64_t)cur * D + d]);
          if (tw_env) v = fmaf(a->w_time[d], now, v);
          v = v + qb;
        }
        q[d] = round16(v * (0.25f * log2e), dt16); /* rounding point 1: the MFMA B operand */
      }
      float heads[D];
      for (int h = 0; h < H; ++h) {
        float m = neg_inf;
        for (int j = 0; j < N; ++j) {
          float acc = 0.0f;
          for (int e = 0; e < DH; ++e)
            acc = fmaf(cache_at(a->glimpse_key, dt16, cbase + (int64_t)j * a->kvl_row_stride + h * DH + e), q[h * DH + e], acc);
          const int feas = !a->mask_inner || mk[j] != 0;
          sc[j * H + h] = feas ? acc : neg_inf;
          m = fmaxf(m, sc[j * H + h]);
        }
        if (!(m > neg_inf)) m = 0.0f;
        float l = 0.0f, o[DH];
        for (int e = 0; e < DH; ++e) o[e] = 0.0f;
        for (int j = 0; j < N; ++j) {
          const float p = exp2f(sc[j * H + h] - m);
          l += p;
          const float pb = round16(p, dt16); /* rounding point 2: softmax numerators as MFMA operand; l sums the fp32 p */
          for (int e = 0; e < DH; ++e)
            o[e] = fmaf(cache_at(a->glimpse_val, dt16, cbase + (int64_t)j * a->kvl_row_stride + h * DH + e), pb, o[e]);
        }
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        for (int e = 0; e < DH; ++e) heads[h * DH + e] = round16(o[e] * inv, dt16); /* rounding point 3: the glimpse */
      }
      float zmax = neg_inf;
      int nan_seen = 0;
      for (int j = 0; j < N; ++j) {
        float acc = 0.0f;
        for (int d = 0; d < D; ++d)
          acc = fmaf(cache_at(a->logit_key, dt16, cbase + (int64_t)j * a->kvl_row_stride + d), heads[d], acc);
        const float uu = acc * (1.0f / sqrt_d);
        if (uu != uu) nan_seen = 1;
        float zz;
        if (a->tanh_clipping > 0.0f) {
          const float ex = expf(-2.0f * fabsf(uu));
          zz = copysignf((1.0f - ex) / (1.0f + ex), uu) * clip_over_temp;
        } else {
          zz = uu * inv_temp;
        }
        if (a->mask_logits && mk[j] == 0) zz = neg_inf;
        z[j] = zz;
        zmax = fmaxf(zmax, zz);
      }
      if (nan_seen) errbits |= RL4CO_EBIT_NAN_LOGIT;
      float se = 0.0f;
      for (int j = 0; j < N; ++j) se += expf(z[j] - zmax);
      const float lse = zmax + logf(se);
      int bi = -1;
      float best = neg_inf;
      for (int j = 0; j < N; ++j) {
        if (!(z[j] > neg_inf)) continue;
        float key = z[j];
        if (a->mode == RL4CO_DECODE_SAMPLE) {
          const float nz = a->exp_noise ? a->exp_noise[((int64_t)t * a->B + r) * N + j]
                                        : rl4co_exp1_noise(a->philox_seed ^ (a->philox_seed_dev ? *a->philox_seed_dev : 0ull), a->philox_offset + (uint64_t)tcol, (uint32_t)r, (uint32_t)j);
          key = z[j] - logf(nz); /* argmax(p / Exp(1)) == argmax(z - log noise) */
        }
        if (bi < 0 || key > best) {
          best = key;
          bi = j;
        }
      }
      float logp;
      if (a->mode == RL4CO_DECODE_EVALUATE) {
        bi = (int)a->forced_actions[(int64_t)r * a->out_stride + tcol];
        if (bi < 0 || bi >= N) {
          errbits |= RL4CO_EBIT_INFEASIBLE;
          bi = 0;
        }
        logp = z[bi] - lse;
      } else {
        if (bi < 0) bi = 0;
        logp = z[bi] - lse;
      }
      if (mk[bi] == 0) errbits |= RL4CO_EBIT_INFEASIBLE;
      if (!(logp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;
      a->actions[(int64_t)r * a->out_stride + tcol] = bi;
      a->logps[(int64_t)r * a->out_stride + tcol] = logp;
      if (tsp) {
        if (step_i == 0) first = bi;
        cur = bi;
        mk[bi] = 0;
        step_i += 1;
        int any = 0;
        for (int j = 0; j < N; ++j) any |= mk[j];
        done = !any;
      } else if (env == RL4CO_ENV_PDP) {
        pdp_transition(bi, vis, tod, mk, N);
        int left = 0;
        for (int j = 0; j < N; ++j) left |= vis[j];
        done = !left;
        step_i += 1;
        cur = bi;
      } else if (env == RL4CO_ENV_PCTSP) {
        used = used + rprize[bi];
        vis[bi] = 1;
        done = (step_i > 0) && (bi == 0);
        step_i += 1;
        cur = bi;
        pctsp_mask_row(used, vis, mk, N);
      } else if (env == RL4CO_ENV_OP) {
        used = used + op_dist(oplocs, cur, bi);
        vis[bi] = 1;
        done = (bi == 0) && (step_i > 0);
        step_i += 1;
        cur = bi;
        op_mask_row(oplocs, opmax, used, vis, cur, mk, N);
      } else {
        if (tw_env) now = cvrptw_time(twlocs, tw, dur, now, cur, bi);
        int di = bi - 1;
        if (di < 0) di = 0;
        if (di > N - 2) di = N - 2;
        used = (used + dem[di]) * (bi != 0 ? 1.0f : 0.0f);
        cur = bi;
        vis[bi] = 1;
        int all = 1;
        for (int j = 0; j < N; ++j) all &= vis[j] != 0;
        done = all;
        if (tw_env) cvrptw_mask_row(dem, used, cap, vis, cur, twlocs, tw, now, mk, N);
        else cvrp_mask_row(dem, used, cap, vis, cur, mk, N);
      }
    }
    if (!single && !done && t >= a->max_steps) errbits |= RL4CO_EBIT_MAX_STEPS;
    memcpy(gmask, mk, (size_t)N);
    if (!tsp) memcpy(a->visited + (int64_t)r * N, vis, (size_t)N);
    if (env == RL4CO_ENV_PDP) memcpy(a->to_deliver + (int64_t)r * N, tod, (size_t)N);
    a->current_node[r] = cur;
    a->done[r] = done ? 1 : 0;
    if (tsp) a->first_node[r] = first;
    if (has_step_i) a->step_i[r] = step_i;
    if (scalar_ctx) a->used_capacity[r] = used;
    if (tw_env) a->current_time[r] = now;
    if (a->n_steps) a->n_steps[r] = t;
    errbits_all |= errbits;
  }
  if (a->err) *a->err |= (int32_t)errbits_all;
  free(sc); free(z); free(mk); free(vis); free(tod);
  return 0;
}

/* deterministic math exposed for tests/test_math.py */
float oracle_expf(float x) { return rl4co_expf(x); }
float oracle_logf(float x) { return rl4co_logf(x); }
float oracle_tanhf(float x) { return rl4co_tanhf(x); }
float oracle_exp1_noise(uint64_t seed, uint64_t step, uint32_t traj, uint32_t node) {
  return rl4co_exp1_noise(seed, step, traj, node);
}

/* array form (fn: 0 exp, 1 log, 2 tanh) so tests/test_math.py can sweep millions of arguments */
int oracle_math_array(int fn, const float* x, int64_t n, float* y) {
  for (int64_t i = 0; i < n; ++i) y[i] = fn == 0 ? rl4co_expf(x[i]) : (fn == 1 ? rl4co_logf(x[i]) : rl4co_tanhf(x[i]));
  return 0;
}

/* host restatement of rl4co_uniform_f32 (csrc/api.hip): same Philox block -> same words, same arithmetic */
int oracle_uniform_f32(float* out, int64_t n, float low, float high, uint64_t seed, uint32_t stream_id, int mode, float capacity) {
  for (int64_t b = 0; 4 * b < n; ++b) {
    uint32_t c[4] = {(uint32_t)b, (uint32_t)((uint64_t)b >> 32), stream_id, 0x52344347u};
    rl4co_philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int i = 0; i < 4 && 4 * b + i < n; ++i) {
      const float u = (float)(c[i] >> 8) * 5.9604644775390625e-8f;
      float x = fmaf(high - low, u, low);
      if (mode == 1) x = (truncf(x) + 1.0f) / capacity;
      out[4 * b + i] = x;
    }
  }
  return 0;
}

/* ---- N3: state augmentation and the POMO evaluation epilogue (host restatement of csrc/augment.hip) --------------
 * follows rl4co/data/transforms.py:16-69 (dihedral8 order; symmetric_transform's x' = cos x - sin y, y' = sin x + cos y,
 * swap where phi > 2 pi, + offset) and rl4co/models/zoo/pomo/model.py:112-140 over utils/ops.py:33-66 (unbatchify to
 * [B, A, S], max over S then over A with torch.max's first-index tie rule, gather_by_index of the action rows).
 * Compiled with -ffp-contract=off like the kernels: separate multiplies, one subtraction / addition. */
int oracle_augment_dihedral8_f32(const float* xy, int B, int N, float* out) {
  const int64_t pairs = (int64_t)B * N;
  for (int64_t i = 0; i < pairs; ++i) {
    const float x = xy[2 * i], y = xy[2 * i + 1], mx = 1.0f - x, my = 1.0f - y;
    const float t[8][2] = {{x, y}, {mx, y}, {x, my}, {mx, my}, {y, x}, {my, x}, {y, mx}, {my, mx}};
    for (int a = 0; a < 8; ++a) {
      out[2 * (i + a * pairs)] = t[a][0];
      out[2 * (i + a * pairs) + 1] = t[a][1];
    }
  }
  return 0;
}

int oracle_augment_symmetric_f32(const float* xy, const float* cos_phi, const float* sin_phi, const uint8_t* swap_axes, int B,
                                 int A, int N, float offset, float* out) {
  for (int64_t r = 0; r < (int64_t)A * B; ++r) {
    const float c = cos_phi[r], s = sin_phi[r];
    const float* src = xy + (r % B) * (int64_t)N * 2;
    float* dst = out + r * (int64_t)N * 2;
    for (int j = 0; j < N; ++j) {
      const float x = src[2 * j] - offset, y = src[2 * j + 1] - offset;
      const float xp = c * x - s * y;
      const float yp = s * x + c * y;
      dst[2 * j] = (swap_axes[r] ? yp : xp) + offset;
      dst[2 * j + 1] = (swap_axes[r] ? xp : yp) + offset;
    }
  }
  return 0;
}

int oracle_pomo_best(const float* reward, const int64_t* actions, int A, int S, int B, int T, float* max_reward,
                     int64_t* best_start, float* max_aug_reward, int64_t* best_aug, int64_t* best_ms_actions,
                     int64_t* best_aug_actions) {
  for (int b = 0; b < B; ++b) {
    float bv = 0.0f;
    int ba = -1, bs_of_ba = 0;
    for (int a = 0; a < A; ++a) {
      float best = reward[(int64_t)a * B + b];
      int bi = 0;
      for (int s = 1; s < S; ++s) {
        const float v = reward[((int64_t)s * A + a) * B + b];
        if (v > best || (v != v && best == best)) { best = v; bi = s; }  /* torch.max: a NaN is maximal, the first one wins */
      }
      if (max_reward) max_reward[(int64_t)b * A + a] = best;
      if (best_start) best_start[(int64_t)b * A + a] = bi;
      if (actions && best_ms_actions)
        memcpy(best_ms_actions + ((int64_t)b * A + a) * T, actions + (((int64_t)bi * A + a) * B + b) * T, (size_t)T * 8);
      if (ba < 0 || best > bv || (best != best && bv == bv)) { bv = best; ba = a; bs_of_ba = bi; }
    }
    if (max_aug_reward) max_aug_reward[b] = bv;
    if (best_aug) best_aug[b] = ba;
    if (actions && best_aug_actions)
      memcpy(best_aug_actions + (int64_t)b * T, actions + (((int64_t)bs_of_ba * A + ba) * B + b) * T, (size_t)T * 8);
  }
  return 0;
}

/* binary16 -> binary32 over an array (tests pin the conversion to torch's for all 65 536 bit patterns) */
int oracle_half_to_float_array(const uint16_t* h, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = half_bits_to_float(h[i]);
  return 0;
}

/* fp32 -> binary16 rounding over an array (tests pin it to torch's conversion) */
int oracle_half_round_array(const float* x, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = half_round(x[i]);
  return 0;
}
