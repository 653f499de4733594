// am_decode.hip — fused AttentionModel decode step / persistent rollout for gfx950.
//
// One launch runs the whole autoregressive loop of every trajectory: per step it streams the
// trajectory's three folded cache planes (glimpse key, glimpse value, logit key; [N,128] each),
// keeps the feasibility mask / per-head scores / logits in LDS, reduces with DPP / permlane
// butterflies, selects the action (greedy, sampled or forced), updates the TSP / CVRP state and
// goes on to the next step — no host round trip, no inter-workgroup traffic (instances are
// independent), no weights (they were folded into the cache once per rollout, rl4co_amd/cache.py).
//
// Reference semantics restated (file:line in the reference checkout):
//   context / query      env_embeddings/context.py:105-149, zoo/am/decoder.py:128-140
//   pointer attention    nn/attention.py:274-320 (8 heads x 16, mask_inner, /sqrt(128))
//   logits -> logprobs   utils/decoding.py:138-188 (tanh clip, mask, temperature, log_softmax)
//   selection            utils/decoding.py:387-413,448-461
//   env transition       envs/routing/tsp/env.py:60-86, envs/routing/cvrp/env.py:66-96,126-136
//   loop                 models/common/constructive/base.py:226-238
//
// Masked nodes are never read. The reference computes scores, values and logits for all N nodes
// and then overwrites the infeasible ones with -inf (attention.py:306-312, decoding.py:174-178):
// their softmax weight is exactly 0 and their log-prob exactly -inf. With mask_inner and
// mask_logits both on (the defaults) a step therefore only needs the cache rows of the F
// currently feasible nodes: each step compacts them (ascending node index) into a list and the
// three passes walk that list — on a TSP rollout F = N - t, i.e. half the bytes and half the
// arithmetic of the reference's formulation, with bit-identical semantics (a skipped term is an
// exact +0). If either flag is off the list simply holds all N nodes.
//
// Arithmetic order — the parity contract, mirrored value-for-value by
// oracle/rollout_ref.c (every op is an IEEE fp32 add/mul/fma/div, -ffp-contract=off):
//   list        = feasible nodes ascending (or all nodes), c = position in the list, F = length
//   EPL = elements per lane per 16-byte load (4 fp32 / 8 bf16); a cache row is covered by
//   LPR = 128/EPL lanes, a head by LPH = 16/EPL lanes; list entry c belongs to row group c % G
//   (G = rows handled concurrently: 64/LPR per wave x waves per trajectory).
//   tree(x_0..x_{n-1}) = pairwise butterfly sum: tree(lo half) + tree(hi half).
//   q[d]        = ((ctx_first[first][d] + ctx_cur[cur][d]) + q_bias[d]) * 0.25   (TSP, i > 0)
//   score(c,h)  = tree over the head's LPH chunks of [fma chain over the chunk's EPL dims,
//                 ascending, from 0]
//   softmax     = m = max_c score ; p_c = exp(score_c - m) ; per row group g:
//                 l_g = sum p_c, o_g[d] = fma(p_c, v_c[d], o_g[d]) over its entries in ascending c ;
//                 heads[d] = tree_g(o_g[d]) * (1 / tree_g(l_g))
//   logit(c)    = tree over the row's LPR chunks of [fma chain over EPL dims] ; / fl(sqrt(128))
//   log_softmax = zmax = max ; s_k = sum over c = k, k+64, ... of exp(z_c - zmax) (k < 64) ;
//                 lse = log(tree_k(s_k)) ; lp_c = (z_c - zmax) - lse
//   argmax      = maximum key, lowest index on ties (greedy: key = lp ; sampling:
//                 key = exp(lp) / noise[node])
#include <hip/hip_runtime.h>

#include "common.h"
#include "rl4co_math.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kH = RL4CO_NUM_HEADS;
constexpr int kDH = kD / kH;
constexpr int kUnroll = 4;  // wave-wide 1 KiB loads kept in flight per pass
constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kSqrtD = 11.3137084989847604f;  // fl32(sqrt(128)), attention.py:293

struct CacheF32 {
  using elem = float;
  using raw = float4;
  static constexpr int EPL = 4;
  __device__ static inline raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ static inline raw ld(const elem* p) { return *reinterpret_cast<const float4*>(p); }
  __device__ static inline void cvt(const raw& t, float (&v)[4]) {
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};

struct CacheBF16 {
  using elem = uint16_t;
  using raw = uint4;
  static constexpr int EPL = 8;
  __device__ static inline raw zero() { return make_uint4(0u, 0u, 0u, 0u); }
  __device__ static inline raw ld(const elem* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static inline void cvt(const raw& t, float (&v)[8]) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
};

// IEEE half planes (torch.float16: the reference's default "16-mixed" precision, utils/trainer.py:57): same 16-byte
// lanes as bf16, converted with v_cvt_f32_f16 — exact, like the bf16 shift
struct CacheF16 {
  using elem = uint16_t;
  using raw = uint4;
  static constexpr int EPL = 8;
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  __device__ static inline raw zero() { return make_uint4(0u, 0u, 0u, 0u); }
  __device__ static inline raw ld(const elem* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static inline void cvt(const raw& t, float (&v)[8]) {
    const half2_t a = __builtin_bit_cast(half2_t, t.x), b = __builtin_bit_cast(half2_t, t.y);
    const half2_t c = __builtin_bit_cast(half2_t, t.z), d = __builtin_bit_cast(half2_t, t.w);
    v[0] = (float)a[0]; v[1] = (float)a[1]; v[2] = (float)b[0]; v[3] = (float)b[1];
    v[4] = (float)c[0]; v[5] = (float)c[1]; v[6] = (float)d[0]; v[7] = (float)d[1];
  }
};

__host__ __device__ inline int lds_pad(int N) { return (N + 63) & ~63; }

// The folded context tables of one instance, EPL dims from e0 on, as fp32: dense fp32 [N,128] rows (the default), or — r06,
// ctx_dtype — rows in the planes' 16-bit element type with the caller's strides (e.g. column blocks of the one 16-bit
// matrix the fused cache fold writes), widened exactly on load: the arithmetic downstream is that of the fp32 tables, so
// the specified-order oracle, given the widened values, still equals the kernel bit for bit.
template <class C>
struct CtxTables {
  const char *cur, *first;
  int64_t rs;  // bytes between node rows
  bool half;
  __device__ inline CtxTables(const rl4co_am_decode_args& a, int cb, int N, int e0) {
    half = a.ctx_dtype != RL4CO_DT_F32;
    const int64_t esz = half ? 2 : 4;
    rs = (a.ctx_row_stride ? a.ctx_row_stride : (int64_t)kD) * esz;
    const int64_t off = ((int64_t)cb * (a.ctx_batch_stride ? a.ctx_batch_stride : (int64_t)N * kD) + e0) * esz;
    cur = a.ctx_cur ? static_cast<const char*>(a.ctx_cur) + off : nullptr;
    first = a.ctx_first ? static_cast<const char*>(a.ctx_first) + off : nullptr;
  }
  __device__ inline void load(const char* table, int64_t row, float (&v)[C::EPL]) const {
    const char* p = table + row * rs;
    if constexpr (sizeof(typename C::elem) == 2) {
      if (half) {
        C::cvt(C::ld(reinterpret_cast<const typename C::elem*>(p)), v);
        return;
      }
    }
#pragma unroll
    for (int e = 0; e < C::EPL; ++e) v[e] = reinterpret_cast<const float*>(p)[e];
  }
};

// blockIdx -> trajectory. Trajectories are stored s-major (row r = s * B_inst + instance, the
// reference's batchify layout) and the S trajectories of an instance stream the SAME cache planes.
// Workgroup b is dispatched to XCD b % 8 (observed placement, used for speed only), and each XCD
// has its own L2 — so consecutive workgroups of one XCD are given the S starts of one instance:
// they run concurrently on that XCD and all but the first touch of a plane row hit its L2.
__device__ inline int trajectory_of_block(int b, int B, int B_inst) {
  const int S = B / B_inst;
  if (S == 1 || (B_inst & 7) != 0) return b;
  const int xcd = b & 7, k = b >> 3;
  return (k % S) * B_inst + (k / S) * 8 + xcd;
}

// LDS ordering inside ONE wave (cross-lane hand-off through LDS, no other wave involved)
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct TrajState {
  int cur, first;
  long long step_i;
  float used;
  float time;  // CVRPTW: current time
  bool done;
  uint32_t errbits;
  float ent_acc;
};

// One wave: compact the nodes this step has to read (ascending) into fl[0..F).
__device__ inline int build_list(const rl4co_am_decode_args& a, const uint8_t* mk, uint16_t* fl, int N, int lane) {
  int F = 0;
  if (a.mask_inner && a.mask_logits) {
    for (int j0 = 0; j0 < N; j0 += 64) {
      const int j = j0 + lane;
      const bool f = j < N && mk[j] != 0;
      const unsigned long long bal = __ballot(f);
      if (f) fl[F + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)j;
      F += __popcll(bal);
    }
  } else {
    for (int j = lane; j < N; j += 64) fl[j] = (uint16_t)j;
    F = N;
  }
  wave_lds_sync();
  return F;
}

template <int ENV>
__device__ inline int commit_and_step(const rl4co_am_decode_args& a, TrajState& st, float* lg, const uint16_t* fl, int F,
                                      uint8_t* mk, uint8_t* vis, const float* dem, float cap, int r, int t, int N, int lane,
                                      const float* oplocs, const float* opmax, const float* twdur, int bc);

// One wave: raw logits lg[0..F) (list order) -> log-probs, selection, outputs, environment
// transition on the LDS-resident mask. Returns the action; st is updated (incl. done).
template <int ENV>
__device__ inline int finalize_and_step(const rl4co_am_decode_args& a, TrajState& st, float* lg, const uint16_t* fl,
                                        int F, uint8_t* mk, uint8_t* vis, const float* dem, float cap, int r, int t,
                                        int N, int lane, const float* oplocs = nullptr, const float* opmax = nullptr,
                                        const float* twdur = nullptr) {
  bool nan_seen = false;
  float zmax = kNegInf;
  for (int c = lane; c < F; c += 64) {
    float z = lg[c] / kSqrtD;
    if (z != z) nan_seen = true;  // attention.py:295-296
    if (a.tanh_clipping > 0.0f) z = rl4co_tanhf(z) * a.tanh_clipping;
    if (a.mask_logits && mk[fl[c]] == 0) z = kNegInf;
    if (a.temperature != 1.0f) z = z / a.temperature;  // x / 1.0f == x exactly
    lg[c] = z;
    zmax = fmaxf(zmax, z);
  }
  if (__any(nan_seen)) st.errbits |= RL4CO_EBIT_NAN_LOGIT;
  zmax = rl4co::bfly_max<1, 64>(zmax);
  float zsum = 0.0f;
  for (int c = lane; c < F; c += 64) zsum = zsum + rl4co_expf(lg[c] - zmax);
  zsum = rl4co::bfly_sum<1, 64>(zsum);
  const float lse = rl4co_logf(zsum);

  float best = kNegInf;
  int bc = 0x7fffffff;  // best list position
  float ent = 0.0f;
  const int64_t tcol = (int64_t)a.t0 + t;
  float* alp = a.all_logps ? a.all_logps + ((int64_t)r * a.out_stride + tcol) * N : nullptr;
  if (alp && F < N) {  // nodes outside the list have log-prob -inf
    for (int j = lane; j < N; j += 64)
      if (mk[j] == 0) alp[j] = kNegInf;
  }
  for (int c = lane; c < F; c += 64) {
    const int j = fl[c];
    const float lp = (lg[c] - zmax) - lse;
    lg[c] = lp;
    float key = lp;
    if (a.mode == RL4CO_DECODE_SAMPLE) {
      const float nz = a.exp_noise
                           ? a.exp_noise[((int64_t)t * a.B + r) * N + j]
                           : rl4co_exp1_noise(a.philox_seed ^ (a.philox_seed_dev ? *a.philox_seed_dev : 0ull),
                                              a.philox_offset + (uint64_t)tcol, (uint32_t)r,
                                              (uint32_t)j);
      key = rl4co_expf(lp) / nz;  // multinomial(p,1) == argmax(p / Exp(1))
    }
    if (bc == 0x7fffffff || key > best) {  // strict '>' keeps the lowest index on ties
      best = key;
      bc = c;
    }
    if (a.entropy && lp > kNegInf) ent = fmaf(rl4co_expf(lp), lp, ent);
    if (alp) alp[j] = lp;
  }
  rl4co::bfly_argmax(best, bc);
  if (a.entropy) st.ent_acc = st.ent_acc - rl4co::bfly_sum<1, 64>(ent);
  wave_lds_sync();
  return commit_and_step<ENV>(a, st, lg, fl, F, mk, vis, dem, cap, r, t, N, lane, oplocs, opmax, twdur, bc);
}

// One wave: lg[0..F) hold the step's log-probs (list order) and `bc` the selected list position. Evaluate-mode lookup,
// outputs, environment transition on the LDS-resident mask. Returns the action; st is updated (incl. done).
template <int ENV>
__device__ inline int commit_and_step(const rl4co_am_decode_args& a, TrajState& st, float* lg, const uint16_t* fl, int F,
                                      uint8_t* mk, uint8_t* vis, const float* dem, float cap, int r, int t, int N, int lane,
                                      const float* oplocs, const float* opmax, const float* twdur, int bc) {
  const int64_t tcol = (int64_t)a.t0 + t;
  int bi;
  float logp;
  if (a.mode == RL4CO_DECODE_EVALUATE) {
    bi = (int)a.forced_actions[(int64_t)r * a.out_stride + tcol];
    if (bi < 0 || bi >= N) {  // forced action out of range
      st.errbits |= RL4CO_EBIT_INFEASIBLE;
      bi = 0;
    }
    // position of the forced node in the list (absent = masked out = log-prob -inf)
    int pos = 0x7fffffff;
    for (int c = lane; c < F; c += 64)
      if (fl[c] == bi) pos = c;
    pos = -rl4co::bfly_i_max(-pos);
    logp = pos < F ? lg[pos] : kNegInf;
  } else {
    bi = (bc < F) ? (int)fl[bc] : 0;
    logp = (bc < F) ? lg[bc] : kNegInf;
  }
  if (mk[bi] == 0) st.errbits |= RL4CO_EBIT_INFEASIBLE;            // decoding.py:393,409
  if (!(logp > -1000.0f)) st.errbits |= RL4CO_EBIT_NEG_INF_LOGP;   // decoding.py:56
  if (lane == 0) {
    a.actions[(int64_t)r * a.out_stride + tcol] = bi;
    a.logps[(int64_t)r * a.out_stride + tcol] = logp;
  }
  wave_lds_sync();

  // ---- environment transition -------------------------------------------------------
  if (ENV == RL4CO_ENV_TSP) {
    if (st.step_i == 0) st.first = bi;  // tsp/env.py:63
    st.cur = bi;
    if (lane == 0) mk[bi] = 0;
    st.step_i += 1;
    wave_lds_sync();
    bool any_left = false;
    for (int j = lane; j < N; j += 64) any_left |= mk[j] != 0;
    st.done = !__any(any_left);  // tsp/env.py:71
  } else if (ENV == RL4CO_ENV_PDP) {
    // pickup and delivery (pdp/env.py:64-99); vis[j]: bit 0 = available, bit 1 = to_deliver
    const int n = N - 1;
    if (lane == 0) {
      vis[bi] &= (uint8_t)~1u;
      vis[(bi + n / 2) % (n + 1)] |= 2;
    }
    st.step_i += 1;
    st.cur = bi;
    wave_lds_sync();
    bool left = false;
    for (int j = lane; j < N; j += 64) {
      const uint8_t v = vis[j];
      mk[j] = (v == 3) ? 1 : 0;
      left |= (v & 1) != 0;
    }
    st.done = !__any(left);  // pdp/env.py:83
  } else if (ENV == RL4CO_ENV_PCTSP) {
    // prize-collecting TSP (pctsp/env.py:62-91,141-148); st.used is the prize collected so far,
    // dem the real prize per node (depot column 0)
    st.used = st.used + dem[bi];
    if (lane == 0) vis[bi] = 1;
    st.done = (bi == 0) && (st.step_i > 0);
    st.step_i += 1;
    st.cur = bi;
    wave_lds_sync();
    const bool depot_visited = vis[0] != 0;
    bool unvisited = false;
    for (int j = lane; j < N; j += 64) {
      if (j >= 1) {
        mk[j] = (vis[j] != 0 || depot_visited) ? 0 : 1;
        unvisited |= vis[j] == 0;
      }
    }
    unvisited = __any(unvisited);
    if (lane == 0) mk[0] = ((st.used < 1.0f) && unvisited) ? 0 : 1;
  } else if (ENV == RL4CO_ENV_OP) {
    // orienteering (op/env.py:67-98,137-154); st.used is the tour length, distances as in the
    // tour-length kernel: sqrt(fma(dy, dy, dx * dx))
    const float cx = oplocs[2 * st.cur], cy = oplocs[2 * st.cur + 1];
    const float bx = oplocs[2 * bi], by = oplocs[2 * bi + 1];
    {
      const float dx = bx - cx, dy = by - cy;
      st.used = st.used + sqrtf(fmaf(dy, dy, dx * dx));
    }
    if (lane == 0) vis[bi] = 1;
    st.done = (bi == 0) && (st.step_i > 0);
    st.step_i += 1;
    st.cur = bi;
    wave_lds_sync();
    const bool depot_visited = vis[0] != 0;
    for (int j = lane; j < N; j += 64) {
      const float dx = oplocs[2 * j] - bx, dy = oplocs[2 * j + 1] - by;
      const bool exceeds = st.used + sqrtf(fmaf(dy, dy, dx * dx)) > opmax[j];
      mk[j] = (j == 0 || !(vis[j] != 0 || depot_visited || exceeds)) ? 1 : 0;
    }
  } else {
    if (ENV == RL4CO_ENV_CVRPTW) {
      // cvrptw/env.py:97-113 (oplocs = coordinates, opmax = (start, end) windows, twdur = service times): the
      // clock advances by the distance from the node the vehicle stands on, waits for the window, serves
      const float dx = oplocs[2 * bi] - oplocs[2 * st.cur], dy = oplocs[2 * bi + 1] - oplocs[2 * st.cur + 1];
      const float served = fmaxf(st.time + sqrtf(fmaf(dy, dy, dx * dx)), opmax[2 * bi]) + twdur[bi];
      st.time = (bi != 0 ? 1.0f : 0.0f) * served;
    }
    const int di = min(max(bi - 1, 0), N - 2);                       // cvrp/env.py:71-73
    st.used = (st.used + dem[di]) * (bi != 0 ? 1.0f : 0.0f);         // cvrp/env.py:76
    st.cur = bi;
    if (lane == 0) vis[bi] = 1;
    wave_lds_sync();
    const float thr = cap + 1e-5f;  // cvrp/env.py:128
    bool any_feasible = false, all_visited = true;
    for (int j = lane; j < N; j += 64) {
      all_visited &= vis[j] != 0;
      if (j >= 1) {
        const bool masked = (vis[j] != 0) || (dem[j - 1] + st.used > thr);
        mk[j] = masked ? 0 : 1;
        any_feasible |= !masked;
      }
    }
    any_feasible = __any(any_feasible);
    st.done = __all(all_visited);  // cvrp/env.py:83
    if (lane == 0) mk[0] = ((st.cur == 0) && any_feasible) ? 0 : 1;  // cvrp/env.py:134-135
    if (ENV == RL4CO_ENV_CVRPTW) {  // cvrptw/env.py:91-95: only nodes whose window is still open on arrival
      wave_lds_sync();
      const float bx = oplocs[2 * bi], by = oplocs[2 * bi + 1];
      for (int j = lane; j < N; j += 64) {
        const float dx = oplocs[2 * j] - bx, dy = oplocs[2 * j + 1] - by;
        if (!(st.time + sqrtf(fmaf(dy, dy, dx * dx)) <= opmax[2 * j + 1])) mk[j] = 0;
      }
    }
  }
  wave_lds_sync();
  return bi;
}

// ================================================================================================
// Streaming kernel: ONE wavefront (= one 64-thread workgroup) per trajectory; with thousands of
// trajectories 16 independent waves per CU hide each other's latency chains and the kernel runs
// at the HBM / Infinity-Cache rate. kUnroll wave-wide 1 KiB loads in flight per pass.
// ================================================================================================
// UNFOLD (parity mode, TSP / CVRP): the context projection and project_out are per-step GEMVs in the reference's
// association — out[d] = fma chain over k ascending of Wt[k][d] * in[k], from 0 — with the input vector broadcast
// from LDS and the transposed weight rows read as 16-byte lanes (each lane produces the EPL dims it owns).
template <int EPL>
__device__ inline void gemv_t(float (&out)[EPL], const float* __restrict__ wt, const float* in_lds, int width, int e0) {
#pragma unroll
  for (int e = 0; e < EPL; ++e) out[e] = 0.0f;
  for (int k = 0; k < width; ++k) {
    const float c = in_lds[k];
#pragma unroll
    for (int e = 0; e < EPL; ++e) out[e] = fmaf(wt[(int64_t)k * kD + e0 + e], c, out[e]);
  }
}

template <class C, int ENV, bool UNFOLD = false>
__global__ void __launch_bounds__(64) am_decode_kernel(const rl4co_am_decode_args a) {
  constexpr int EPL = C::EPL;
  constexpr int LPR = kD / EPL;   // lanes per cache row
  constexpr int RPL = 64 / LPR;   // rows per wave-wide load (= row groups G)
  constexpr int LPH = kDH / EPL;  // lanes per head
  using elem = typename C::elem;
  using raw_t = typename C::raw;

  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int r = trajectory_of_block(blockIdx.x, a.B, a.B_inst);
  const int N = a.N;
  const int Np = lds_pad(N);
  float* sc = reinterpret_cast<float*>(smem);  // [Np*kH] per-head scores -> softmax numerators, (c*kH + h)
  float* lg = sc + Np * kH;                    // [Np] raw logits -> clipped logits -> log-probs (list order)
  float* mh = lg + Np;                         // [8] per-head score maxima
  uint16_t* fl = reinterpret_cast<uint16_t*>(mh + kH);  // [Np] nodes this step reads, ascending
  uint8_t* mk = reinterpret_cast<uint8_t*>(fl + Np);    // [Np] 1 = feasible
  uint8_t* vis = mk + Np;                               // [Np] CVRP visited flags

  const int cb = r % a.B_inst;  // instance whose cache this trajectory reads
  const int rg = lane / LPR;    // row group inside a wave-wide load
  const int li = lane % LPR;    // lane position inside the row
  const int hd = li / LPH;      // head this lane contributes to
  const int e0 = li * EPL;      // first embedding dim held by this lane

  const elem* Kg = static_cast<const elem*>(a.glimpse_key) + (int64_t)cb * a.kvl_batch_stride + e0;
  const elem* Vg = static_cast<const elem*>(a.glimpse_val) + (int64_t)cb * a.kvl_batch_stride + e0;
  const elem* Kl = static_cast<const elem*>(a.logit_key) + (int64_t)cb * a.kvl_batch_stride + e0;
  const int64_t rs = a.kvl_row_stride;
  const CtxTables<C> ctx(a, cb, N, e0);  // (UNFOLD: not read)
  const float* hrow = UNFOLD ? a.node_embed + (int64_t)cb * N * kD + e0 : nullptr;

  // ---- load the trajectory state ---------------------------------------------------
  uint8_t* gmask = a.action_mask + (int64_t)r * N;
  for (int j = lane; j < Np; j += 64) mk[j] = (j < N) ? gmask[j] : (uint8_t)0;
  constexpr bool kScalarCtx = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_OP || ENV == RL4CO_ENV_PCTSP || ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kCvrpLike = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_CVRPTW;
  if (ENV == RL4CO_ENV_PDP) {  // bit 0 = available, bit 1 = to_deliver
    const uint8_t* gv = a.visited + (int64_t)r * N;
    const uint8_t* gt = a.to_deliver + (int64_t)r * N;
    for (int j = lane; j < Np; j += 64) vis[j] = (j < N) ? (uint8_t)((gv[j] != 0 ? 1 : 0) | (gt[j] != 0 ? 2 : 0)) : (uint8_t)0;
  } else if (ENV != RL4CO_ENV_TSP) {
    const uint8_t* gv = a.visited + (int64_t)r * N;
    for (int j = lane; j < Np; j += 64) vis[j] = (j < N) ? gv[j] : (uint8_t)1;
  }
  TrajState st;
  st.cur = (int)a.current_node[r];
  st.first = (ENV == RL4CO_ENV_TSP) ? (int)a.first_node[r] : 0;
  st.step_i = !kCvrpLike ? a.step_i[r] : 0;
  st.time = (ENV == RL4CO_ENV_CVRPTW) ? a.current_time[r] : 0.0f;
  st.used = kScalarCtx ? a.used_capacity[r] : 0.0f;  // OP: tour length so far
  st.done = a.done[r] != 0;
  st.errbits = 0;
  st.ent_acc = 0.0f;
  const float* oplocs = (ENV == RL4CO_ENV_OP || ENV == RL4CO_ENV_CVRPTW) ? a.locs + (int64_t)cb * N * 2 : nullptr;
  const float* opmax = (ENV == RL4CO_ENV_OP)       ? a.max_length + (int64_t)cb * N
                       : (ENV == RL4CO_ENV_CVRPTW) ? a.time_windows + (int64_t)cb * N * 2  // (start, end) per node
                                                   : nullptr;
  const float* twdur = (ENV == RL4CO_ENV_CVRPTW) ? a.durations + (int64_t)cb * N : nullptr;
  // the context scalar is cap - used in both depot environments (context.py:147-149, 211-213):
  // OP: longest tour that may still end at the depot (its row of the table) minus the tour so far
  // PCTSP: prize still to collect, clamped at 0 (context.py:184-198)
  const float cap = (kCvrpLike || ENV == RL4CO_ENV_PCTSP) ? a.vehicle_capacity[r]
                                                                      : ((ENV == RL4CO_ENV_OP) ? opmax[0] : 0.0f);
  const float* dem = kCvrpLike                  ? a.demand + (int64_t)cb * (N - 1)
                     : (ENV == RL4CO_ENV_PCTSP) ? a.demand + (int64_t)cb * N
                                                : nullptr;
  wave_lds_sync();

  float qb[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) qb[e] = a.q_bias ? a.q_bias[(int64_t)cb * kD + e0 + e] : 0.0f;

  const bool single = a.max_steps == 1;
  int t = 0;
  int rows_read = 0;  // cache rows this trajectory streamed (x 3 planes): the launch's real HBM read volume

  for (; t < a.max_steps && (!st.done || single); ++t) {
    const int F = build_list(a, mk, fl, N, lane);
    rows_read += F;

    // ---- query: folded context projection + graph context (decoder.py:128-140) ------
    float q[EPL];
    if constexpr (UNFOLD) {
      // context vector into LDS (the score buffer is dead between steps), then q = W_ctx . ctx + graph context
      float* cv = sc;
      if (rg == 0) {
        if (ENV == RL4CO_ENV_TSP) {
          const bool ph = st.step_i < 1;  // context.py:120-128 placeholder
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            cv[e0 + e] = ph ? a.w_placeholder[e0 + e] : hrow[(int64_t)st.first * kD + e];
            cv[kD + e0 + e] = ph ? a.w_placeholder[kD + e0 + e] : hrow[(int64_t)st.cur * kD + e];
          }
        } else {  // context.py:70-74,147-149: [h_cur ; cap - used]
#pragma unroll
          for (int e = 0; e < EPL; ++e) cv[e0 + e] = hrow[(int64_t)st.cur * kD + e];
          if (li == 0) cv[kD] = cap - st.used;
        }
      }
      wave_lds_sync();
      gemv_t<EPL>(q, a.w_ctx_t, cv, a.ctx_width, e0);
#pragma unroll
      for (int e = 0; e < EPL; ++e) q[e] = q[e] + qb[e];
      wave_lds_sync();  // cv is overwritten by this step's scores
    } else if (ENV == RL4CO_ENV_TSP) {
      if (st.step_i < 1) {  // context.py:120 placeholder context
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = a.q_step0[e0 + e] + qb[e];
      } else {
        float cf[EPL], cc[EPL];
        ctx.load(ctx.first, st.first, cf);
        ctx.load(ctx.cur, st.cur, cc);
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = (cf[e] + cc[e]) + qb[e];
      }
    } else if (ENV == RL4CO_ENV_PDP) {  // context.py:232-243: the current node alone
      float cc[EPL];
      ctx.load(ctx.cur, st.cur, cc);
#pragma unroll
      for (int e = 0; e < EPL; ++e) q[e] = cc[e] + qb[e];
    } else {
      float rem = cap - st.used;  // context.py:147-149
      if (ENV == RL4CO_ENV_PCTSP && !(rem > 0.0f)) rem = 0.0f;
      float cc[EPL];
      ctx.load(ctx.cur, st.cur, cc);
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        float v = fmaf(a.w_cap[e0 + e], rem, cc[e]);
        if (ENV == RL4CO_ENV_CVRPTW) v = fmaf(a.w_time[e0 + e], st.time, v);  // context.py:152-166
        q[e] = v + qb[e];
      }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] = q[e] * 0.25f;  // 1/sqrt(16), exact

    // ---- pass 1: per-head scores over the glimpse keys of the listed nodes -------------
    float m = kNegInf;
    for (int c0 = 0; c0 < F; c0 += RPL * kUnroll) {
      raw_t rw[kUnroll];
      int jj[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int c = c0 + u * RPL + rg;
        jj[u] = (c < F) ? (int)fl[c] : -1;
        rw[u] = (jj[u] >= 0) ? C::ld(Kg + (int64_t)jj[u] * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int c = c0 + u * RPL + rg;
        const bool valid = jj[u] >= 0;
        float k[EPL];
        C::cvt(rw[u], k);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc = fmaf(q[e], k[e], acc);
        acc = rl4co::bfly_sum<1, LPH>(acc);
        const bool feas = valid && (!a.mask_inner || mk[valid ? jj[u] : 0] != 0);
        const float sv = feas ? acc : kNegInf;
        if (valid && (li % LPH) == 0) sc[c * kH + hd] = sv;
        m = fmaxf(m, sv);
      }
    }
    m = rl4co::bfly_max<LPR, 64>(m);
    if (rg == 0 && (li % LPH) == 0) mh[hd] = m;
    wave_lds_sync();
    // softmax numerators once per (list entry, head) — lane-strided over the F*8 scores, so
    // each exp is evaluated exactly once (index % 8 = head is fixed per lane: 64 % 8 == 0)
    {
      const float mm = mh[lane & (kH - 1)];
      for (int idx = lane; idx < F * kH; idx += 64) sc[idx] = rl4co_expf(sc[idx] - mm);
    }
    wave_lds_sync();

    // ---- pass 2: softmax-weighted value sum ------------------------------------------------
    float l = 0.0f;
    float o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.0f;
    for (int c0 = 0; c0 < F; c0 += RPL * kUnroll) {
      raw_t rw[kUnroll];
      bool ok[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int c = c0 + u * RPL + rg;
        ok[u] = c < F;
        rw[u] = ok[u] ? C::ld(Vg + (int64_t)fl[c] * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int c = c0 + u * RPL + rg;
        float v[EPL];
        C::cvt(rw[u], v);
        const float p = ok[u] ? sc[c * kH + hd] : 0.0f;
        l = l + p;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = fmaf(p, v[e], o[e]);
      }
    }
    l = 1.0f / rl4co::bfly_sum<LPR, 64>(l);  // one IEEE division per step; heads = o * (1/l)
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = rl4co::bfly_sum<LPR, 64>(o[e]) * l;
    if constexpr (UNFOLD) {  // glimpse = project_out(heads) (attention.py:287), heads broadcast through LDS
      wave_lds_sync();       // every lane is done with the softmax numerators in sc
      if (rg == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) sc[e0 + e] = o[e];
      }
      wave_lds_sync();
      gemv_t<EPL>(o, a.w_out_t, sc, kD, e0);
    }

    // ---- pass 3: pointer logits against the (project_out-folded) logit key -----------
    for (int c0 = 0; c0 < F; c0 += RPL * kUnroll) {
      raw_t rw[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int c = c0 + u * RPL + rg;
        rw[u] = (c < F) ? C::ld(Kl + (int64_t)fl[c] * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int c = c0 + u * RPL + rg;
        float k[EPL];
        C::cvt(rw[u], k);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc = fmaf(o[e], k[e], acc);
        acc = rl4co::bfly_sum<1, LPR>(acc);
        if (c < F && li == 0) lg[c] = acc;
      }
    }
    wave_lds_sync();

    finalize_and_step<ENV>(a, st, lg, fl, F, mk, vis, dem, cap, r, t, N, lane, oplocs, opmax, twdur);
  }
  if (!single && !st.done && t >= a.max_steps) st.errbits |= RL4CO_EBIT_MAX_STEPS;

  // ---- write the state back ------------------------------------------------------------
  for (int j = lane; j < N; j += 64) gmask[j] = mk[j];
  if (ENV == RL4CO_ENV_PDP) {
    uint8_t* gv = a.visited + (int64_t)r * N;
    uint8_t* gt = a.to_deliver + (int64_t)r * N;
    for (int j = lane; j < N; j += 64) {
      gv[j] = vis[j] & 1;
      gt[j] = (vis[j] >> 1) & 1;
    }
  } else if (ENV != RL4CO_ENV_TSP) {
    uint8_t* gv = a.visited + (int64_t)r * N;
    for (int j = lane; j < N; j += 64) gv[j] = vis[j];
  }
  if (lane == 0) {
    a.current_node[r] = st.cur;
    a.done[r] = st.done ? 1 : 0;
    if (ENV == RL4CO_ENV_TSP) a.first_node[r] = st.first;
    if (!kCvrpLike) a.step_i[r] = st.step_i;
    if (ENV == RL4CO_ENV_CVRPTW) a.current_time[r] = st.time;
    if (kScalarCtx) a.used_capacity[r] = st.used;
    if (a.n_steps) a.n_steps[r] = t;
    if (a.steps_summary) {
      atomicMax(a.steps_summary, t);
      atomicAdd(a.steps_summary + 1, t);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.steps_summary + 2), (unsigned long long)rows_read);
    }
    if (a.entropy) a.entropy[r] += st.ent_acc;
    if (st.errbits) atomicOr(a.err, (int)st.errbits);
  }
}

// ================================================================================================
// Four waves per trajectory (bf16 planes), G = 16 row groups: list entry c -> wave (c % 16) / 4,
// row group c % 4; per-group partial sums in ascending c, then the pairwise tree — butterfly
// inside the wave, ((w0 + w1) + (w2 + w3)) across waves through LDS.
//   RESIDENT = true : "LDS-resident" — the three planes are copied into LDS ONCE per rollout and
//                     every step reads them from there (needs them to fit half a CU's LDS).
//                     Wins when there are too few trajectories to fill the chip with one wave
//                     each: 0.65 vs 1.34 ms at 256 trajectories, 1.5 vs 1.8 ms at 1024 (TSP-100);
//                     with thousands of trajectories only two fit per CU and the latency chain
//                     of a step is exposed (5.8 vs 4.4 ms at 4096), so auto picks it for B <= 1024.
//   RESIDENT = false: "wide" streaming — same structure reading the planes from HBM/L2 every
//                     step; for few trajectories whose planes do not fit LDS (CVRP-500 x 1024:
//                     4096 waves instead of 1024, 120 -> 61 us per step).
// ================================================================================================
constexpr int kLdsWaves = 4;
constexpr int kLdsGroups = 16;

// The first half of finalize_and_step for the four-wave kernels, bit for bit: the ELEMENTWISE work of a step's N logits
// — clip, the softmax exponentials, log-probs, the sampling keys with their Philox draws: ~200 VALU operations per node —
// is dealt over all 256 threads instead of leaving three waves idle behind wave 0, while every order-dependent reduction
// (the lane-strided sum of the exponentials and its butterfly, the entropy) is still taken by wave 0 in finalize_and_step's
// own order from the staged terms; maxima and the (key, lowest position) arg-max are exact in any order. `xs`: 2 nw + 16
// floats of scratch (the score block, dead between B3 and the next step). Returns the selected list position (wave 0).
template <int ENV>
__device__ inline int wide_scores(const rl4co_am_decode_args& a, TrajState& st, float* lg, const uint16_t* fl, int F,
                                  const uint8_t* mk, float* xs, int nw, int r, int t, int N, int tid) {
  constexpr int T = 64 * kLdsWaves;
  const int w = tid >> 6, lane = tid & 63;
  float* ex = xs;             // [nw] softmax exponentials, then sampling keys
  float* xw = xs + 2 * nw;    // [4] wave maxima | [4] NaN flags | [1] log-sum-exp
  bool nan_seen = false;
  float zmax = kNegInf;
  for (int c = tid; c < F; c += T) {
    float z = lg[c] / kSqrtD;
    if (z != z) nan_seen = true;  // attention.py:295-296
    if (a.tanh_clipping > 0.0f) z = rl4co_tanhf(z) * a.tanh_clipping;
    if (a.mask_logits && mk[fl[c]] == 0) z = kNegInf;
    if (a.temperature != 1.0f) z = z / a.temperature;  // x / 1.0f == x exactly
    lg[c] = z;
    zmax = fmaxf(zmax, z);
  }
  zmax = rl4co::bfly_max<1, 64>(zmax);
  const bool wave_nan = __any(nan_seen);
  if (lane == 0) {
    xw[w] = zmax;
    xw[4 + w] = wave_nan ? 1.0f : 0.0f;
  }
  __syncthreads();
  zmax = fmaxf(fmaxf(xw[0], xw[1]), fmaxf(xw[2], xw[3]));
  if ((xw[4] + xw[5]) + (xw[6] + xw[7]) != 0.0f) st.errbits |= RL4CO_EBIT_NAN_LOGIT;
  for (int c = tid; c < F; c += T) ex[c] = rl4co_expf(lg[c] - zmax);
  __syncthreads();
  if (w == 0) {
    float zsum = 0.0f;
    for (int c = lane; c < F; c += 64) zsum = zsum + ex[c];
    zsum = rl4co::bfly_sum<1, 64>(zsum);
    if (lane == 0) xw[8] = rl4co_logf(zsum);
  }
  __syncthreads();
  const float lse = xw[8];
  const int64_t tcol = (int64_t)a.t0 + t;
  float* alp = a.all_logps ? a.all_logps + ((int64_t)r * a.out_stride + tcol) * N : nullptr;
  if (alp && F < N) {  // nodes outside the list have log-prob -inf
    for (int j = tid; j < N; j += T)
      if (mk[j] == 0) alp[j] = kNegInf;
  }
  for (int c = tid; c < F; c += T) {
    const int j = fl[c];
    const float lp = (lg[c] - zmax) - lse;
    lg[c] = lp;
    float key = lp;
    if (a.mode == RL4CO_DECODE_SAMPLE) {
      const float nz = a.exp_noise
                           ? a.exp_noise[((int64_t)t * a.B + r) * N + j]
                           : rl4co_exp1_noise(a.philox_seed ^ (a.philox_seed_dev ? *a.philox_seed_dev : 0ull),
                                              a.philox_offset + (uint64_t)tcol, (uint32_t)r,
                                              (uint32_t)j);
      key = rl4co_expf(lp) / nz;  // multinomial(p,1) == argmax(p / Exp(1))
    }
    ex[c] = key;
    if (alp) alp[j] = lp;
  }
  __syncthreads();
  int bc = 0x7fffffff;
  if (w == 0) {
    float best = kNegInf, ent = 0.0f;
    for (int c = lane; c < F; c += 64) {
      const float key = ex[c];
      if (bc == 0x7fffffff || key > best) {  // strict '>' keeps the lowest index on ties
        best = key;
        bc = c;
      }
      if (a.entropy) {
        const float lp = lg[c];
        if (lp > kNegInf) ent = fmaf(rl4co_expf(lp), lp, ent);
      }
    }
    rl4co::bfly_argmax(best, bc);
    if (a.entropy) st.ent_acc = st.ent_acc - rl4co::bfly_sum<1, 64>(ent);
    wave_lds_sync();
  }
  return bc;
}

__host__ __device__ inline int lds_variant_sc_rows(int N) { return N < 80 ? 80 : N; }
__host__ __device__ inline int wide_scratch_bytes(int N) {
  const int nw = (N + 3) & ~3;
  return lds_variant_sc_rows(N) * kH * 4 + nw * 4 + kLdsWaves * kH * 4 + 32 + 2 * nw + 2 * nw;
}
__host__ __device__ inline int lds_variant_bytes(int N) { return 3 * N * kD * 2 + wide_scratch_bytes(N); }

// Four workgroups per CU (<= 128 registers): CVRP-500 x 1024 is 1024 workgroups = exactly four per CU, ONE round. The half
// (fp16) builds took 130 - 138 registers under a bound of two — three per CU, so a quarter of the trajectories waited for
// a second round: 25.3 ms per C5 step against 19.9 with bf16 planes (r04).
template <int ENV, bool RESIDENT, class C = CacheBF16>  // C: CacheBF16 or CacheF16 (same 16-byte lanes, different convert)
__global__ void __launch_bounds__(64 * kLdsWaves, 4) am_decode_wide_kernel(const rl4co_am_decode_args a) {
  constexpr int EPL = 8, LPR = 16, LPH = 2;
  constexpr int U = 4;  // list entries per wave handled per unrolled block
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63;
  const int r = trajectory_of_block(blockIdx.x, a.B, a.B_inst);
  const int N = a.N;
  const int nw = (N + 3) & ~3;
  uint16_t* planes = reinterpret_cast<uint16_t*>(smem);             // [3][N][128] bf16 (RESIDENT only)
  float* sc = reinterpret_cast<float*>(planes + (RESIDENT ? 3 * N * kD : 0));  // [max(N,80)][8]; later o/l partials
  float* lg = sc + lds_variant_sc_rows(N) * kH;                     // [nw] logits -> log-probs (list order)
  float* mpart = lg + nw;                                           // [4][8] per-wave head maxima
  int* shi = reinterpret_cast<int*>(mpart + kLdsWaves * kH);        // [8] broadcast: action, done, F
  uint16_t* fl = reinterpret_cast<uint16_t*>(shi + 8);              // [nw] nodes this step reads
  uint8_t* mk = reinterpret_cast<uint8_t*>(fl + nw);                // [nw] 1 = feasible
  uint8_t* vis = mk + nw;                                           // [nw] CVRP visited

  const int cb = r % a.B_inst;
  const int rg = lane / LPR, li = lane % LPR, hd = li / LPH, e0 = li * EPL;
  const uint16_t *Kg, *Vg, *Kl;
  int64_t rs;
  if constexpr (RESIDENT) {
    Kg = planes + e0;
    Vg = planes + N * kD + e0;
    Kl = planes + 2 * N * kD + e0;
    rs = kD;
  } else {
    Kg = static_cast<const uint16_t*>(a.glimpse_key) + (int64_t)cb * a.kvl_batch_stride + e0;
    Vg = static_cast<const uint16_t*>(a.glimpse_val) + (int64_t)cb * a.kvl_batch_stride + e0;
    Kl = static_cast<const uint16_t*>(a.logit_key) + (int64_t)cb * a.kvl_batch_stride + e0;
    rs = a.kvl_row_stride;
  }
  const CtxTables<C> ctx(a, cb, N, e0);
  // o/l partial slots inside this wave's own (dead after pass 2) score chunks
  auto opart = [&](int wv, int d) -> float* { return sc + ((16 * (d >> 5) + 4 * wv) * kH) + (d & 31); };
  auto lpart = [&](int wv, int h) -> float* { return sc + ((16 * 4 + 4 * wv) * kH) + h; };

  // ---- planes HBM -> LDS, once per rollout (16-byte coalesced) --------------------------------
  if constexpr (RESIDENT) {
    const uint16_t* src[3] = {static_cast<const uint16_t*>(a.glimpse_key), static_cast<const uint16_t*>(a.glimpse_val),
                              static_cast<const uint16_t*>(a.logit_key)};
    const int chunks = N * (kD / 8);  // 16-byte chunks per plane
    for (int p = 0; p < 3; ++p) {
      const uint16_t* g = src[p] + (int64_t)cb * a.kvl_batch_stride;
      for (int c = tid; c < chunks; c += 64 * kLdsWaves) {
        const int row = c >> 4, col = (c & 15) * 8;
        *reinterpret_cast<uint4*>(planes + (p * N + row) * kD + col) =
            *reinterpret_cast<const uint4*>(g + (int64_t)row * a.kvl_row_stride + col);
      }
    }
  }
  // ---- trajectory state: the same six environments as the streaming kernel (wave 0 owns the transition, the other
  // waves receive the scalars it changes through LDS after every step) ----------------------------------------------
  constexpr bool kScalarCtx = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_OP || ENV == RL4CO_ENV_PCTSP || ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kCvrpLike = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_CVRPTW;
  uint8_t* gmask = a.action_mask + (int64_t)r * N;
  for (int j = tid; j < nw; j += 64 * kLdsWaves) mk[j] = (j < N) ? gmask[j] : (uint8_t)0;
  if (ENV == RL4CO_ENV_PDP) {  // bit 0 = available, bit 1 = to_deliver
    const uint8_t* gv = a.visited + (int64_t)r * N;
    const uint8_t* gt = a.to_deliver + (int64_t)r * N;
    for (int j = tid; j < nw; j += 64 * kLdsWaves)
      vis[j] = (j < N) ? (uint8_t)((gv[j] != 0 ? 1 : 0) | (gt[j] != 0 ? 2 : 0)) : (uint8_t)0;
  } else if (ENV != RL4CO_ENV_TSP) {
    const uint8_t* gv = a.visited + (int64_t)r * N;
    for (int j = tid; j < nw; j += 64 * kLdsWaves) vis[j] = (j < N) ? gv[j] : (uint8_t)1;
  }
  TrajState st;
  st.cur = (int)a.current_node[r];
  st.first = (ENV == RL4CO_ENV_TSP) ? (int)a.first_node[r] : 0;
  st.step_i = !kCvrpLike ? a.step_i[r] : 0;
  st.time = (ENV == RL4CO_ENV_CVRPTW) ? a.current_time[r] : 0.0f;
  st.used = kScalarCtx ? a.used_capacity[r] : 0.0f;  // OP: tour length so far; PCTSP: prize collected
  st.done = a.done[r] != 0;
  st.errbits = 0;
  st.ent_acc = 0.0f;
  const float* oplocs = (ENV == RL4CO_ENV_OP || ENV == RL4CO_ENV_CVRPTW) ? a.locs + (int64_t)cb * N * 2 : nullptr;
  const float* opmax = (ENV == RL4CO_ENV_OP)       ? a.max_length + (int64_t)cb * N
                       : (ENV == RL4CO_ENV_CVRPTW) ? a.time_windows + (int64_t)cb * N * 2
                                                   : nullptr;
  const float* twdur = (ENV == RL4CO_ENV_CVRPTW) ? a.durations + (int64_t)cb * N : nullptr;
  const float cap = (kCvrpLike || ENV == RL4CO_ENV_PCTSP) ? a.vehicle_capacity[r] : ((ENV == RL4CO_ENV_OP) ? opmax[0] : 0.0f);
  const float* dem = kCvrpLike                  ? a.demand + (int64_t)cb * (N - 1)
                     : (ENV == RL4CO_ENV_PCTSP) ? a.demand + (int64_t)cb * N
                                                : nullptr;
  __syncthreads();
  if (w == 0) {
    const int F0 = build_list(a, mk, fl, N, lane);
    if (lane == 0) shi[2] = F0;
  }
  __syncthreads();

  float qb[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) qb[e] = a.q_bias ? a.q_bias[(int64_t)cb * kD + e0 + e] : 0.0f;

  const bool single = a.max_steps == 1;
  int t = 0;
  int rows_read = 0;

  for (; t < a.max_steps && (!st.done || single); ++t) {
    const int F = shi[2];
    rows_read += F;
    const int iters = (F + kLdsGroups - 1) / kLdsGroups;
    // ---- query ---------------------------------------------------------------------------------
    float q[EPL];
    if (ENV == RL4CO_ENV_TSP) {
      if (st.step_i < 1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = a.q_step0[e0 + e] + qb[e];
      } else {
        float cf[EPL], cc[EPL];
        ctx.load(ctx.first, st.first, cf);
        ctx.load(ctx.cur, st.cur, cc);
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = (cf[e] + cc[e]) + qb[e];
      }
    } else if (ENV == RL4CO_ENV_PDP) {  // context.py:232-243: the current node alone
      float cc[EPL];
      ctx.load(ctx.cur, st.cur, cc);
#pragma unroll
      for (int e = 0; e < EPL; ++e) q[e] = cc[e] + qb[e];
    } else {
      float rem = cap - st.used;  // context.py:147-149
      if (ENV == RL4CO_ENV_PCTSP && !(rem > 0.0f)) rem = 0.0f;
      float cc[EPL];
      ctx.load(ctx.cur, st.cur, cc);
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        float v = fmaf(a.w_cap[e0 + e], rem, cc[e]);
        if (ENV == RL4CO_ENV_CVRPTW) v = fmaf(a.w_time[e0 + e], st.time, v);  // context.py:152-166
        q[e] = v + qb[e];
      }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] = q[e] * 0.25f;

    // ---- pass 1: scores of this wave's list entries ----------------------------------------------
    float m = kNegInf;
    for (int i0 = 0; i0 < iters; i0 += U) {
      uint4 rw[U];
      int jj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = kLdsGroups * (i0 + u) + 4 * w + rg;
        jj[u] = (c < F) ? (int)fl[c] : -1;
        rw[u] = (jj[u] >= 0) ? *reinterpret_cast<const uint4*>(Kg + (int64_t)jj[u] * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = kLdsGroups * (i0 + u) + 4 * w + rg;
        const bool valid = jj[u] >= 0;
        float k[EPL];
        C::cvt(rw[u], k);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc = fmaf(q[e], k[e], acc);
        acc = rl4co::bfly_sum<1, LPH>(acc);
        const bool feas = valid && (!a.mask_inner || mk[valid ? jj[u] : 0] != 0);
        const float sv = feas ? acc : kNegInf;
        if (valid && (li & 1) == 0) sc[c * kH + hd] = sv;
        m = fmaxf(m, sv);
      }
    }
    m = rl4co::bfly_max<LPR, 64>(m);
    if (rg == 0 && (li & 1) == 0) mpart[w * kH + hd] = m;
    __syncthreads();  // B1: all head maxima visible
    // softmax numerators of this wave's entries, each evaluated once (lane-strided over the
    // wave's 4-entry chunks: 32 consecutive floats per chunk, head = index % 8 fixed per lane)
    {
      const int hh = lane & (kH - 1);
      const float mm = fmaxf(fmaxf(mpart[hh], mpart[kH + hh]), fmaxf(mpart[2 * kH + hh], mpart[3 * kH + hh]));
      for (int tix = lane; tix < 32 * iters; tix += 64) {
        const int c0 = kLdsGroups * (tix >> 5) + 4 * w;  // first entry of the chunk
        if (c0 + ((tix & 31) >> 3) < F) {
          float* p = sc + c0 * kH + (tix & 31);
          *p = rl4co_expf(*p - mm);
        }
      }
    }
    wave_lds_sync();  // the numerators are consumed by this same wave only

    // ---- pass 2: softmax weights and weighted values ---------------------------------------------
    float l = 0.0f;
    float o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.0f;
    for (int i0 = 0; i0 < iters; i0 += U) {
      uint4 rw[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = kLdsGroups * (i0 + u) + 4 * w + rg;
        ok[u] = c < F;
        rw[u] = ok[u] ? *reinterpret_cast<const uint4*>(Vg + (int64_t)fl[c] * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = kLdsGroups * (i0 + u) + 4 * w + rg;
        float v[EPL];
        C::cvt(rw[u], v);
        const float p = ok[u] ? sc[c * kH + hd] : 0.0f;
        l = l + p;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = fmaf(p, v[e], o[e]);
      }
    }
    l = rl4co::bfly_sum<LPR, 64>(l);
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = rl4co::bfly_sum<LPR, 64>(o[e]);
    if (rg == 0) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) *opart(w, e0 + e) = o[e];
      if ((li & 1) == 0) *lpart(w, hd) = l;
    }
    __syncthreads();  // B2: partials of the four waves visible
    l = 1.0f / ((*lpart(0, hd) + *lpart(1, hd)) + (*lpart(2, hd) + *lpart(3, hd)));
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int d = e0 + e;
      o[e] = ((*opart(0, d) + *opart(1, d)) + (*opart(2, d) + *opart(3, d))) * l;
    }

    // ---- pass 3: logits of this wave's list entries ---------------------------------------------------
    for (int i0 = 0; i0 < iters; i0 += U) {
      uint4 rw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = kLdsGroups * (i0 + u) + 4 * w + rg;
        rw[u] = (c < F) ? *reinterpret_cast<const uint4*>(Kl + (int64_t)fl[c] * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = kLdsGroups * (i0 + u) + 4 * w + rg;
        float k[EPL];
        C::cvt(rw[u], k);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc = fmaf(o[e], k[e], acc);
        acc = rl4co::bfly_sum<1, LPR>(acc);
        if (c < F && li == 0) lg[c] = acc;
      }
    }
    __syncthreads();  // B3: all logits visible

    // ---- all waves: clip, exponentials, log-probs, sampling keys; wave 0: the reductions, then the environment
    // transition and the next step's list -----------------------------------------------------------------
    const int bc = wide_scores<ENV>(a, st, lg, fl, F, mk, sc, nw, r, t, N, tid);
    if (w == 0) {
      commit_and_step<ENV>(a, st, lg, fl, F, mk, vis, dem, cap, r, t, N, lane, oplocs, opmax, twdur, bc);
      const int Fn = build_list(a, mk, fl, N, lane);
      if (lane == 0) {  // the scalars the next query is built from, for the other three waves
        shi[0] = st.cur;
        shi[1] = st.done ? 1 : 0;
        shi[2] = Fn;
        shi[3] = st.first;
        shi[4] = (int)st.step_i;
        shi[5] = __float_as_int(st.used);
        shi[6] = __float_as_int(st.time);
      }
    }
    __syncthreads();  // B4: state scalars, mask and list visible to every wave
    if (w != 0) {
      st.cur = shi[0];
      st.done = shi[1] != 0;
      st.first = shi[3];
      st.step_i = shi[4];
      st.used = __int_as_float(shi[5]);
      st.time = __int_as_float(shi[6]);
    }
  }
  if (w == 0) {
    if (!single && !st.done && t >= a.max_steps) st.errbits |= RL4CO_EBIT_MAX_STEPS;
    for (int j = lane; j < N; j += 64) gmask[j] = mk[j];
    if (ENV == RL4CO_ENV_PDP) {
      uint8_t* gv = a.visited + (int64_t)r * N;
      uint8_t* gt = a.to_deliver + (int64_t)r * N;
      for (int j = lane; j < N; j += 64) {
        gv[j] = vis[j] & 1;
        gt[j] = (vis[j] >> 1) & 1;
      }
    } else if (ENV != RL4CO_ENV_TSP) {
      uint8_t* gv = a.visited + (int64_t)r * N;
      for (int j = lane; j < N; j += 64) gv[j] = vis[j];
    }
    if (lane == 0) {
      a.current_node[r] = st.cur;
      a.done[r] = st.done ? 1 : 0;
      if (ENV == RL4CO_ENV_TSP) a.first_node[r] = st.first;
      if (!kCvrpLike) a.step_i[r] = st.step_i;
      if (ENV == RL4CO_ENV_CVRPTW) a.current_time[r] = st.time;
      if (kScalarCtx) a.used_capacity[r] = st.used;
      if (a.n_steps) a.n_steps[r] = t;
      if (a.steps_summary) {
        atomicMax(a.steps_summary, t);
        atomicAdd(a.steps_summary + 1, t);
        atomicAdd(reinterpret_cast<unsigned long long*>(a.steps_summary + 2), (unsigned long long)(RESIDENT ? 0 : rows_read));  // resident planes are read once per rollout
      }
      if (a.entropy) a.entropy[r] += st.ent_acc;
      if (st.errbits) atomicOr(a.err, (int)st.errbits);
    }
  }
}

template <int ENV, bool RESIDENT, class C>
int launch_wide_c(const rl4co_am_decode_args& a, hipStream_t stream) {
  const int lds = RESIDENT ? lds_variant_bytes(a.N) : wide_scratch_bytes(a.N);
  if (lds > 64 * 1024) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_decode_wide_kernel<ENV, RESIDENT, C>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  hipLaunchKernelGGL((am_decode_wide_kernel<ENV, RESIDENT, C>), dim3(a.B), dim3(64 * kLdsWaves), lds, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
template <int ENV, bool RESIDENT>
int launch_wide(const rl4co_am_decode_args& a, hipStream_t stream) {
  return a.cache_dtype == RL4CO_DT_F16 ? launch_wide_c<ENV, RESIDENT, CacheF16>(a, stream)
                                       : launch_wide_c<ENV, RESIDENT, CacheBF16>(a, stream);
}

// Which kernel serves these arguments (rules from measurements on MI355X, see the kernel headers).
inline int resolve_variant(const rl4co_am_decode_args& a) {
  // the unfolded parity mode exists in the streaming kernel only
  if (a.unfold) return (a.variant == RL4CO_VARIANT_AUTO || a.variant == RL4CO_VARIANT_STREAM) ? RL4CO_VARIANT_STREAM : -1;
  const bool f16 = a.cache_dtype == RL4CO_DT_F16;  // fp16 planes: every variant the bf16 planes have
  const bool bf16 = a.cache_dtype == RL4CO_DT_BF16 || f16;  // (16-bit planes)
  // multistart on the matrix cores (am_decode_ms.hip): 16-bit planes, N <= 128, plain outputs; every environment
  const bool ms_ok = bf16 && a.N <= 128 && a.B_inst > 0 && a.all_logps == nullptr && a.entropy == nullptr;
  if (a.variant == RL4CO_VARIANT_MS) return ms_ok ? RL4CO_VARIANT_MS : -1;
  const bool fits = bf16 && lds_variant_bytes(a.N) <= 80 * 1024;
  const bool wide_ok = bf16 && wide_scratch_bytes(a.N) <= 64 * 1024;
  if (a.variant == RL4CO_VARIANT_STREAM) return RL4CO_VARIANT_STREAM;
  if (a.variant == RL4CO_VARIANT_LDS) return fits ? RL4CO_VARIANT_LDS : -1;
  if (a.variant == RL4CO_VARIANT_WIDE) return wide_ok ? RL4CO_VARIANT_WIDE : -1;
  if (a.max_steps < 4) return RL4CO_VARIANT_STREAM;
  // auto where it was measured faster than one wave per trajectory (N = 100, 4096 instances, sampling, r02; M trajectory-
  // steps/s MS vs STREAM): TSP 8 starts 667 vs 276; pickup-delivery 8 starts 633 vs 379; prize-collecting TSP 8 starts
  // 442 vs 178; CVRP 8 starts 254 vs 277 but 16 starts 463 vs 276 (a full 16-column tile). Orienteering (7-step
  // ragged tours under random weights: 84 vs 162) and CVRP with time windows (111 vs 278: the per-step mask over all
  // nodes with a square root each, replicated in every lane of the column) stay on STREAM unless MS is asked for
  // r03, with two instances per column tile at <= 8 starts (a launch then costs ~3.3 ms for 4096 instances whatever
  // the start count; tools/ms_bench.py, sampling, 4096 x S): TSP 3 starts 3.26 vs 5.11 ms (2: 3.24 vs 3.59), prize-
  // collecting TSP 3 starts 3.38 vs 5.09, pickup-delivery 3 starts 3.43 vs 3.64 (2: 3.42 vs 2.58 — STREAM), CVRP 8 starts
  // 13.6 vs 14.3 (4: 13.3 vs 7.8 — STREAM: every column of a tile runs to the tile's longest tour)
  const int ms_from = (a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_PDP || a.env == RL4CO_ENV_PCTSP) ? 3
                      : (a.env == RL4CO_ENV_CVRP ? 8 : 0);
  if (ms_ok && ms_from > 0 && a.B >= ms_from * a.B_inst) return RL4CO_VARIANT_MS;
  if (fits && a.B <= 1024) return RL4CO_VARIANT_LDS;
  // one wave per trajectory needs >= ~16 waves per CU to hide its latency chain: with fewer
  // trajectories than that, four waves per trajectory keep the memory pipes busier
  if (wide_ok && a.B <= 2048) return RL4CO_VARIANT_WIDE;
  return RL4CO_VARIANT_STREAM;
}

template <class C, int ENV, bool UNFOLD = false>
int launch(const rl4co_am_decode_args& a, hipStream_t stream) {
  const int lds = rl4co_am_decode_lds_bytes(a.N, ENV);
  if (lds > 64 * 1024) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_decode_kernel<C, ENV, UNFOLD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  hipLaunchKernelGGL((am_decode_kernel<C, ENV, UNFOLD>), dim3(a.B), dim3(64), lds, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

}  // namespace

extern "C" int rl4co_am_decode_lds_bytes(int N, int env) {
  (void)env;
  const int Np = lds_pad(N);
  return Np * kH * 4 + Np * 4 + kH * 4 + Np * 2 + Np + Np;
}

extern "C" int rl4co_am_decode_row_groups(const rl4co_am_decode_args* args) {
  if (args == nullptr) return -1;
  const int v = resolve_variant(*args);
  if (v < 0) return -1;
  if (v == RL4CO_VARIANT_MS) return 0;  // bf16 MFMA variant: tolerance-tested, no specified-order oracle
  if (v == RL4CO_VARIANT_LDS || v == RL4CO_VARIANT_WIDE) return kLdsGroups;
  return args->cache_dtype != RL4CO_DT_F32 ? 64 / (kD / CacheBF16::EPL) : 64 / (kD / CacheF32::EPL);
}

extern "C" int rl4co_am_decode_variant(const rl4co_am_decode_args* args) {
  return args == nullptr ? -1 : resolve_variant(*args);
}

extern "C" int rl4co_am_decode(const rl4co_am_decode_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_decode_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_OP ||
                a.env == RL4CO_ENV_PCTSP || a.env == RL4CO_ENV_PDP || a.env == RL4CO_ENV_CVRPTW);
  RL4CO_REQUIRE(a.B > 0 && a.B_inst > 0 && a.B % a.B_inst == 0);
  RL4CO_REQUIRE(a.N >= 2 && a.N <= 4096);
  RL4CO_REQUIRE(a.max_steps >= 1);
  RL4CO_REQUIRE(a.mode >= RL4CO_DECODE_GREEDY && a.mode <= RL4CO_DECODE_EVALUATE);
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == RL4CO_DT_BF16 || a.cache_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(a.cache_dtype != RL4CO_DT_F16 || !a.unfold);
  RL4CO_REQUIRE(a.glimpse_key && a.glimpse_val && a.logit_key && (a.ctx_cur || a.unfold));
  RL4CO_REQUIRE(a.unfold == 0 || a.unfold == 1);
  // (r06) context tables in the planes' 16-bit type / with their own strides: 16-byte rows for the 8-dim lanes
  RL4CO_REQUIRE(a.ctx_dtype == RL4CO_DT_F32 || (a.ctx_dtype == a.cache_dtype && !a.unfold));
  RL4CO_REQUIRE(a.ctx_row_stride == 0 || (a.ctx_row_stride >= kD && a.ctx_row_stride % 8 == 0));
  RL4CO_REQUIRE(a.ctx_batch_stride == 0 || (a.ctx_batch_stride >= (int64_t)a.N * kD && a.ctx_batch_stride % 8 == 0));
  RL4CO_REQUIRE(a.ctx_dtype == RL4CO_DT_F32 || ((reinterpret_cast<uintptr_t>(a.ctx_cur) | reinterpret_cast<uintptr_t>(a.ctx_first)) & 15) == 0);
  if (a.unfold) {
    RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP);
    RL4CO_REQUIRE(a.node_embed && a.w_ctx_t && a.w_out_t);
    RL4CO_REQUIRE(a.ctx_width == (a.env == RL4CO_ENV_TSP ? 2 * kD : kD + 1));
    RL4CO_REQUIRE(a.env != RL4CO_ENV_TSP || a.w_placeholder);
  }
  RL4CO_REQUIRE(a.kvl_row_stride >= kD && a.kvl_row_stride % 8 == 0);
  RL4CO_REQUIRE(a.kvl_batch_stride >= (int64_t)a.N * kD && a.kvl_batch_stride % 8 == 0);
  RL4CO_REQUIRE(a.action_mask && a.current_node && a.done && a.actions && a.logps && a.err);
  RL4CO_REQUIRE(a.out_stride >= 1 && a.t0 >= 0 && (int64_t)a.t0 + a.max_steps <= a.out_stride);
  RL4CO_REQUIRE(a.temperature > 0.0f);
  RL4CO_REQUIRE((reinterpret_cast<uintptr_t>(a.steps_summary) & 7) == 0);  // [2..3] is one 64-bit counter
  RL4CO_REQUIRE(a.mode != RL4CO_DECODE_EVALUATE || a.forced_actions != nullptr);
  if (a.env == RL4CO_ENV_TSP) {
    RL4CO_REQUIRE(((a.ctx_first && a.q_step0) || a.unfold) && a.first_node && a.step_i);
  } else if (a.env == RL4CO_ENV_CVRP) {
    RL4CO_REQUIRE((a.w_cap || a.unfold) && a.demand && a.used_capacity && a.vehicle_capacity && a.visited);
  } else if (a.env == RL4CO_ENV_CVRPTW) {
    RL4CO_REQUIRE(a.w_cap && a.w_time && a.demand && a.used_capacity && a.vehicle_capacity && a.visited);
    RL4CO_REQUIRE(a.locs && a.time_windows && a.durations && a.current_time);
  } else if (a.env == RL4CO_ENV_PDP) {
    RL4CO_REQUIRE(a.visited && a.to_deliver && a.step_i && (a.N - 1) % 2 == 0);
  } else if (a.env == RL4CO_ENV_PCTSP) {
    RL4CO_REQUIRE(a.w_cap && a.demand && a.used_capacity && a.vehicle_capacity && a.step_i && a.visited);
  } else {
    RL4CO_REQUIRE(a.w_cap && a.locs && a.max_length && a.used_capacity && a.step_i && a.visited);
  }
  RL4CO_REQUIRE(rl4co_am_decode_lds_bytes(a.N, a.env) <= 160 * 1024);
  RL4CO_REQUIRE(a.variant >= RL4CO_VARIANT_AUTO && a.variant <= RL4CO_VARIANT_MS);
  const int variant = resolve_variant(a);
  RL4CO_REQUIRE(variant >= 0);  // explicit variant requested that cannot serve these planes
  hipStream_t s = rl4co::as_stream(stream);
  if (a.unfold) {
    if (a.cache_dtype == RL4CO_DT_F32)
      return a.env == RL4CO_ENV_TSP ? launch<CacheF32, RL4CO_ENV_TSP, true>(a, s) : launch<CacheF32, RL4CO_ENV_CVRP, true>(a, s);
    return a.env == RL4CO_ENV_TSP ? launch<CacheBF16, RL4CO_ENV_TSP, true>(a, s) : launch<CacheBF16, RL4CO_ENV_CVRP, true>(a, s);
  }
  if (variant == RL4CO_VARIANT_MS && a.cache_dtype == RL4CO_DT_F16) return rl4co::launch_decode_ms_f16(a, s);
  if (variant == RL4CO_VARIANT_MS) return rl4co::launch_decode_ms(a, s);
  if (variant == RL4CO_VARIANT_LDS || variant == RL4CO_VARIANT_WIDE) {
    const bool res = variant == RL4CO_VARIANT_LDS;
    switch (a.env) {
      case RL4CO_ENV_TSP: return res ? launch_wide<RL4CO_ENV_TSP, true>(a, s) : launch_wide<RL4CO_ENV_TSP, false>(a, s);
      case RL4CO_ENV_CVRP: return res ? launch_wide<RL4CO_ENV_CVRP, true>(a, s) : launch_wide<RL4CO_ENV_CVRP, false>(a, s);
      case RL4CO_ENV_OP: return res ? launch_wide<RL4CO_ENV_OP, true>(a, s) : launch_wide<RL4CO_ENV_OP, false>(a, s);
      case RL4CO_ENV_PCTSP: return res ? launch_wide<RL4CO_ENV_PCTSP, true>(a, s) : launch_wide<RL4CO_ENV_PCTSP, false>(a, s);
      case RL4CO_ENV_PDP: return res ? launch_wide<RL4CO_ENV_PDP, true>(a, s) : launch_wide<RL4CO_ENV_PDP, false>(a, s);
      default: return res ? launch_wide<RL4CO_ENV_CVRPTW, true>(a, s) : launch_wide<RL4CO_ENV_CVRPTW, false>(a, s);
    }
  }
  if (a.cache_dtype == RL4CO_DT_F16) {
    switch (a.env) {
      case RL4CO_ENV_TSP: return launch<CacheF16, RL4CO_ENV_TSP>(a, s);
      case RL4CO_ENV_CVRP: return launch<CacheF16, RL4CO_ENV_CVRP>(a, s);
      case RL4CO_ENV_OP: return launch<CacheF16, RL4CO_ENV_OP>(a, s);
      case RL4CO_ENV_PCTSP: return launch<CacheF16, RL4CO_ENV_PCTSP>(a, s);
      case RL4CO_ENV_PDP: return launch<CacheF16, RL4CO_ENV_PDP>(a, s);
      default: return launch<CacheF16, RL4CO_ENV_CVRPTW>(a, s);
    }
  }
  if (a.env == RL4CO_ENV_CVRPTW)
    return a.cache_dtype == RL4CO_DT_F32 ? launch<CacheF32, RL4CO_ENV_CVRPTW>(a, s)
                                         : launch<CacheBF16, RL4CO_ENV_CVRPTW>(a, s);
  if (a.env == RL4CO_ENV_PDP)
    return a.cache_dtype == RL4CO_DT_F32 ? launch<CacheF32, RL4CO_ENV_PDP>(a, s) : launch<CacheBF16, RL4CO_ENV_PDP>(a, s);
  if (a.env == RL4CO_ENV_PCTSP)
    return a.cache_dtype == RL4CO_DT_F32 ? launch<CacheF32, RL4CO_ENV_PCTSP>(a, s)
                                         : launch<CacheBF16, RL4CO_ENV_PCTSP>(a, s);
  if (a.env == RL4CO_ENV_OP)
    return a.cache_dtype == RL4CO_DT_F32 ? launch<CacheF32, RL4CO_ENV_OP>(a, s) : launch<CacheBF16, RL4CO_ENV_OP>(a, s);
  if (a.cache_dtype == RL4CO_DT_F32) {
    return a.env == RL4CO_ENV_TSP ? launch<CacheF32, RL4CO_ENV_TSP>(a, s)
                                  : launch<CacheF32, RL4CO_ENV_CVRP>(a, s);
  }
  return a.env == RL4CO_ENV_TSP ? launch<CacheBF16, RL4CO_ENV_TSP>(a, s)
                                : launch<CacheBF16, RL4CO_ENV_CVRP>(a, s);
}
