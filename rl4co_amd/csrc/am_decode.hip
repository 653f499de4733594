// am_decode.hip — fused AttentionModel decode step / persistent rollout for gfx950.
//
// One wavefront (= one 64-thread workgroup) owns one trajectory for the whole
// autoregressive loop. Per step it streams the trajectory's three folded cache
// planes (glimpse key, glimpse value, logit key; [N,128] each) from HBM with
// 16-byte coalesced loads (kUnroll wave-wide loads in flight per pass), keeps the
// feasibility mask / per-head scores / logits in LDS, reduces with wave
// butterflies, selects the action (greedy, sampled or forced), updates the TSP /
// CVRP state in registers+LDS and goes on to the next step — no host round trip,
// no inter-workgroup traffic (instances are independent), no weights (they were
// folded into the cache once per rollout, rl4co_amd/cache.py).
//
// Reference semantics restated (file:line in the reference checkout):
//   context / query      env_embeddings/context.py:105-149, zoo/am/decoder.py:128-140
//   pointer attention    nn/attention.py:274-320 (8 heads x 16, mask_inner, /sqrt(128))
//   logits -> logprobs   utils/decoding.py:138-188 (tanh clip, mask, temperature, log_softmax)
//   selection            utils/decoding.py:387-413,448-461
//   env transition       envs/routing/tsp/env.py:60-86, envs/routing/cvrp/env.py:66-96,126-136
//   loop                 models/common/constructive/base.py:226-238
//
// Arithmetic order — the parity contract, mirrored value-for-value by
// oracle/rollout_ref.c (every op is an IEEE fp32 add/mul/fma/div, -ffp-contract=off):
//   EPL = elements per lane per 16-byte load (4 fp32 / 8 bf16); a cache row is
//   covered by LPR = 128/EPL lanes, a wave-wide load covers G = 64/LPR rows ("row
//   groups": row j belongs to group j % G), a head by LPH = 16/EPL lanes.
//   tree(x_0..x_{n-1}) = pairwise butterfly sum: tree(lo half) + tree(hi half).
//   q[d]        = ((ctx_first[first][d] + ctx_cur[cur][d]) + q_bias[d]) * 0.25   (TSP, i > 0)
//   score(j,h)  = tree over the head's LPH chunks of [fma chain over the chunk's EPL dims,
//                 ascending, from 0]
//   softmax     = m = max_j score ; p_j = exp(score_j - m) ; per row group g:
//                 l_g = sum_j p_j, o_g[d] = fma(p_j, v_j[d], o_g[d]) over its rows in
//                 ascending j ; l = tree_g(l_g), o[d] = tree_g(o_g[d]) ; heads[d] = o[d] * (1 / l)
//   logit(j)    = tree over the row's LPR chunks of [fma chain over EPL dims] ; / fl(sqrt(128))
//   log_softmax = zmax = max ; s_k = sum over j = k, k+64, ... of exp(z_j - zmax) (k < 64) ;
//                 lse = log(tree_k(s_k)) ; lp_j = (z_j - zmax) - lse
//   argmax      = maximum key, lowest index on ties (greedy: key = lp ; sampling:
//                 key = exp(lp) / noise)
#include <hip/hip_runtime.h>

#include "common.h"
#include "rl4co_math.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kH = RL4CO_NUM_HEADS;
constexpr int kDH = kD / kH;
constexpr int kUnroll = 4;  // wave-wide 1 KiB loads kept in flight per pass
constexpr float kNegInf = -__builtin_huge_valf();

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

__host__ __device__ inline int lds_pad(int N) { return (N + 63) & ~63; }

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

template <class C, int ENV>
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
  float* sc = reinterpret_cast<float*>(smem);  // [Np*kH] per-head scores, (j*kH + h)
  float* lg = sc + Np * kH;                    // [Np] raw logits -> clipped logits -> log-probs
  uint8_t* mk = reinterpret_cast<uint8_t*>(lg + Np);  // [Np] 1 = feasible
  uint8_t* vis = mk + Np;                             // [Np] CVRP visited flags
  float* mh = reinterpret_cast<float*>(vis + Np);     // [8] per-head score maxima

  const int cb = r % a.B_inst;  // instance whose cache this trajectory reads
  const int rg = lane / LPR;    // row group inside a wave-wide load
  const int li = lane % LPR;    // lane position inside the row
  const int hd = li / LPH;      // head this lane contributes to
  const int e0 = li * EPL;      // first embedding dim held by this lane

  const elem* Kg = static_cast<const elem*>(a.glimpse_key) + (int64_t)cb * a.kvl_batch_stride + e0;
  const elem* Vg = static_cast<const elem*>(a.glimpse_val) + (int64_t)cb * a.kvl_batch_stride + e0;
  const elem* Kl = static_cast<const elem*>(a.logit_key) + (int64_t)cb * a.kvl_batch_stride + e0;
  const int64_t rs = a.kvl_row_stride;
  const float* ctxc = a.ctx_cur + (int64_t)cb * N * kD + e0;
  const float* ctxf = (ENV == RL4CO_ENV_TSP) ? a.ctx_first + (int64_t)cb * N * kD + e0 : nullptr;

  // ---- load the trajectory state ---------------------------------------------------
  uint8_t* gmask = a.action_mask + (int64_t)r * N;
  for (int j = lane; j < Np; j += 64) mk[j] = (j < N) ? gmask[j] : (uint8_t)0;
  if (ENV == RL4CO_ENV_CVRP) {
    const uint8_t* gv = a.visited + (int64_t)r * N;
    for (int j = lane; j < Np; j += 64) vis[j] = (j < N) ? gv[j] : (uint8_t)1;
  }
  int cur = (int)a.current_node[r];
  int first = (ENV == RL4CO_ENV_TSP) ? (int)a.first_node[r] : 0;
  long long step_i = (ENV == RL4CO_ENV_TSP) ? a.step_i[r] : 0;
  float used = (ENV == RL4CO_ENV_CVRP) ? a.used_capacity[r] : 0.0f;
  const float cap = (ENV == RL4CO_ENV_CVRP) ? a.vehicle_capacity[r] : 0.0f;
  const float* dem = (ENV == RL4CO_ENV_CVRP) ? a.demand + (int64_t)cb * (N - 1) : nullptr;
  bool done = a.done[r] != 0;
  __syncthreads();

  float qb[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) qb[e] = a.q_bias ? a.q_bias[(int64_t)cb * kD + e0 + e] : 0.0f;

  const float sqrt_d = 11.3137084989847604f;  // fl32(sqrt(128)), attention.py:293
  const bool single = a.max_steps == 1;
  uint32_t errbits = 0;
  float ent_acc = 0.0f;
  int t = 0;

  for (; t < a.max_steps && (!done || single); ++t) {
    // ---- query: folded context projection + graph context (decoder.py:128-140) ------
    float q[EPL];
    if (ENV == RL4CO_ENV_TSP) {
      if (step_i < 1) {  // context.py:120 placeholder context
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = a.q_step0[e0 + e] + qb[e];
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
          q[e] = (ctxf[(int64_t)first * kD + e] + ctxc[(int64_t)cur * kD + e]) + qb[e];
      }
    } else {
      const float rem = cap - used;  // context.py:147-149
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        q[e] = fmaf(a.w_cap[e0 + e], rem, ctxc[(int64_t)cur * kD + e]) + qb[e];
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] = q[e] * 0.25f;  // 1/sqrt(16), exact

    // ---- pass 1: per-head scores over the glimpse keys -------------------------------
    float m = kNegInf;
    for (int j0 = 0; j0 < N; j0 += RPL * kUnroll) {
      raw_t rw[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int j = j0 + u * RPL + rg;
        rw[u] = (j < N) ? C::ld(Kg + (int64_t)j * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int j = j0 + u * RPL + rg;
        const bool valid = j < N;
        float k[EPL];
        C::cvt(rw[u], k);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc = fmaf(q[e], k[e], acc);
        acc = rl4co::bfly_sum<1, LPH>(acc);
        const bool feas = valid && (!a.mask_inner || mk[valid ? j : 0] != 0);
        const float sv = feas ? acc : kNegInf;
        if (valid && (li % LPH) == 0) sc[j * kH + hd] = sv;
        m = fmaxf(m, sv);
      }
    }
    m = rl4co::bfly_max<LPR, 64>(m);
    if (rg == 0 && (li % LPH) == 0) mh[hd] = m;
    __syncthreads();
    // softmax numerators once per (node, head) — lane-strided over the N*8 scores, so each
    // exp is evaluated exactly once instead of once per lane of the head (index % 8 = head is
    // fixed per lane because 64 % 8 == 0)
    {
      const float mm = mh[lane & (kH - 1)];
      for (int idx = lane; idx < N * kH; idx += 64) sc[idx] = rl4co_expf(sc[idx] - mm);
    }
    __syncthreads();

    // ---- pass 2: softmax weights and weighted value sum ------------------------------
    float l = 0.0f;
    float o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.0f;
    for (int j0 = 0; j0 < N; j0 += RPL * kUnroll) {
      raw_t rw[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int j = j0 + u * RPL + rg;
        rw[u] = (j < N) ? C::ld(Vg + (int64_t)j * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int j = j0 + u * RPL + rg;
        const bool valid = j < N;
        float v[EPL];
        C::cvt(rw[u], v);
        const float p = valid ? sc[j * kH + hd] : 0.0f;
        l = l + p;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = fmaf(p, v[e], o[e]);
      }
    }
    l = 1.0f / rl4co::bfly_sum<LPR, 64>(l);  // one IEEE division per step; heads = o * (1/l)
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = rl4co::bfly_sum<LPR, 64>(o[e]) * l;

    // ---- pass 3: pointer logits against the (project_out-folded) logit key -----------
    for (int j0 = 0; j0 < N; j0 += RPL * kUnroll) {
      raw_t rw[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int j = j0 + u * RPL + rg;
        rw[u] = (j < N) ? C::ld(Kl + (int64_t)j * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int j = j0 + u * RPL + rg;
        float k[EPL];
        C::cvt(rw[u], k);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc = fmaf(o[e], k[e], acc);
        acc = rl4co::bfly_sum<1, LPR>(acc);
        if (j < N && li == 0) lg[j] = acc;
      }
    }
    __syncthreads();

    // ---- logits -> clipped / masked / tempered (decoding.py:169-185), lane-strided ----
    bool nan_seen = false;
    float zmax = kNegInf;
    for (int j = lane; j < N; j += 64) {
      float z = lg[j] / sqrt_d;
      if (z != z) nan_seen = true;  // attention.py:295-296
      if (a.tanh_clipping > 0.0f) z = rl4co_tanhf(z) * a.tanh_clipping;
      if (a.mask_logits && mk[j] == 0) z = kNegInf;
      if (a.temperature != 1.0f) z = z / a.temperature;  // x / 1.0f == x exactly
      lg[j] = z;
      zmax = fmaxf(zmax, z);
    }
    if (__any(nan_seen)) errbits |= RL4CO_EBIT_NAN_LOGIT;

    // ---- log_softmax over the N logits (decoding.py:188) -----------------------------
    zmax = rl4co::bfly_max<1, 64>(zmax);
    float zsum = 0.0f;
    for (int j = lane; j < N; j += 64) zsum = zsum + rl4co_expf(lg[j] - zmax);
    zsum = rl4co::bfly_sum<1, 64>(zsum);
    const float lse = rl4co_logf(zsum);

    // ---- selection ---------------------------------------------------------------------
    float best = kNegInf;
    int bi = 0x7fffffff;
    float ent = 0.0f;
    const int64_t tcol = (int64_t)a.t0 + t;
    for (int j = lane; j < N; j += 64) {
      const float lp = (lg[j] - zmax) - lse;
      lg[j] = lp;
      float key = lp;
      if (a.mode == RL4CO_DECODE_SAMPLE) {
        const float nz = a.exp_noise
                             ? a.exp_noise[((int64_t)t * a.B + r) * N + j]
                             : rl4co_exp1_noise(a.philox_seed, a.philox_offset + (uint64_t)tcol,
                                                (uint32_t)r, (uint32_t)j);
        key = rl4co_expf(lp) / nz;  // multinomial(p,1) == argmax(p / Exp(1))
      }
      if (bi == 0x7fffffff || key > best) {  // strict '>' keeps the lowest index on ties
        best = key;
        bi = j;
      }
      if (a.entropy && lp > kNegInf) ent = fmaf(rl4co_expf(lp), lp, ent);
      if (a.all_logps) a.all_logps[((int64_t)r * a.out_stride + tcol) * N + j] = lp;
    }
    rl4co::bfly_argmax(best, bi);
    if (a.entropy) ent_acc = ent_acc - rl4co::bfly_sum<1, 64>(ent);
    if (a.mode == RL4CO_DECODE_EVALUATE) bi = (int)a.forced_actions[(int64_t)r * a.out_stride + tcol];
    if (bi < 0 || bi >= N) {  // forced action out of range
      errbits |= RL4CO_EBIT_INFEASIBLE;
      bi = 0;
    }
    __syncthreads();
    const float logp = lg[bi];
    if (mk[bi] == 0) errbits |= RL4CO_EBIT_INFEASIBLE;  // decoding.py:393,409
    if (!(logp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;  // decoding.py:56
    if (lane == 0) {
      a.actions[(int64_t)r * a.out_stride + tcol] = bi;
      a.logps[(int64_t)r * a.out_stride + tcol] = logp;
    }
    __syncthreads();

    // ---- environment transition -------------------------------------------------------
    if (ENV == RL4CO_ENV_TSP) {
      if (step_i == 0) first = bi;  // tsp/env.py:63
      cur = bi;
      if (lane == 0) mk[bi] = 0;
      step_i += 1;
      __syncthreads();
      bool any_left = false;
      for (int j = lane; j < N; j += 64) any_left |= mk[j] != 0;
      done = !__any(any_left);  // tsp/env.py:71
    } else {
      const int di = min(max(bi - 1, 0), N - 2);               // cvrp/env.py:71-73
      used = (used + dem[di]) * (bi != 0 ? 1.0f : 0.0f);       // cvrp/env.py:76
      cur = bi;
      if (lane == 0) vis[bi] = 1;
      __syncthreads();
      const float thr = cap + 1e-5f;  // cvrp/env.py:128
      bool any_feasible = false, all_visited = true;
      for (int j = lane; j < N; j += 64) {
        all_visited &= vis[j] != 0;
        if (j >= 1) {
          const bool masked = (vis[j] != 0) || (dem[j - 1] + used > thr);
          mk[j] = masked ? 0 : 1;
          any_feasible |= !masked;
        }
      }
      any_feasible = __any(any_feasible);
      done = __all(all_visited);  // cvrp/env.py:83
      if (lane == 0) mk[0] = ((cur == 0) && any_feasible) ? 0 : 1;  // cvrp/env.py:134-135
      __syncthreads();
    }
  }
  if (!single && !done && t >= a.max_steps) errbits |= RL4CO_EBIT_MAX_STEPS;

  // ---- write the state back ------------------------------------------------------------
  for (int j = lane; j < N; j += 64) gmask[j] = mk[j];
  if (ENV == RL4CO_ENV_CVRP) {
    uint8_t* gv = a.visited + (int64_t)r * N;
    for (int j = lane; j < N; j += 64) gv[j] = vis[j];
  }
  if (lane == 0) {
    a.current_node[r] = cur;
    a.done[r] = done ? 1 : 0;
    if (ENV == RL4CO_ENV_TSP) {
      a.first_node[r] = first;
      a.step_i[r] = step_i;
    } else {
      a.used_capacity[r] = used;
    }
    if (a.n_steps) a.n_steps[r] = t;
    if (a.entropy) a.entropy[r] += ent_acc;
    if (errbits) atomicOr(a.err, (int)errbits);
  }
}

// ================================================================================================
// LDS-resident variant (bf16 planes, N small enough that one trajectory's three planes fit in
// half a CU's LDS): the planes are read from HBM ONCE per rollout instead of once per decode
// step and every step streams them from LDS. 4 waves per trajectory, 2 trajectories per CU.
//
// Same arithmetic contract as the streaming kernel with G = 16 row groups (row j -> wave
// (j % 16) / 4, row group j % 4): per-group partial sums in ascending j, then the pairwise tree
// — butterfly inside the wave, ((w0 + w1) + (w2 + w3)) across waves through LDS.
// ================================================================================================
constexpr int kLdsWaves = 4;
constexpr int kLdsGroups = 16;

__host__ __device__ inline int lds_variant_sc_rows(int N) { return N < 80 ? 80 : N; }
__host__ __device__ inline int wide_scratch_bytes(int N) {
  const int nw = (N + 3) & ~3;
  return lds_variant_sc_rows(N) * kH * 4 + nw * 4 + kLdsWaves * kH * 4 + 32 + 2 * nw;
}
__host__ __device__ inline int lds_variant_bytes(int N) { return 3 * N * kD * 2 + wide_scratch_bytes(N); }

__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// RESIDENT = true : planes copied into LDS once per rollout (ITERS = rows per wave per pass, unrolled)
// RESIDENT = false: "wide" streaming — the same 4-wave structure reading the planes from HBM/L2
//                   every step; for rollouts with too few trajectories to fill the chip with one
//                   wave each and planes too large for LDS (CVRP-500 x 1024: 4096 waves instead
//                   of 1024). ITERS = 0: runtime row count, 4 loads in flight per wave and pass.
template <int ENV, int ITERS, bool RESIDENT>
__global__ void __launch_bounds__(64 * kLdsWaves, 2) am_decode_lds_kernel(const rl4co_am_decode_args a) {
  using C = CacheBF16;
  constexpr int U = ITERS > 0 ? ITERS : 4;  // rows per wave handled per unrolled block
  constexpr int EPL = 8, LPR = 16, LPH = 2;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63;
  const int r = trajectory_of_block(blockIdx.x, a.B, a.B_inst);
  const int N = a.N;
  const int nw = (N + 3) & ~3;
  uint16_t* planes = reinterpret_cast<uint16_t*>(smem);             // [3][N][128] bf16 (RESIDENT only)
  float* sc = reinterpret_cast<float*>(planes + (RESIDENT ? 3 * N * kD : 0));  // [max(N,80)][8] scores; later o/l partials
  float* lg = sc + lds_variant_sc_rows(N) * kH;                     // [nw] logits -> log-probs
  float* mpart = lg + nw;                                           // [4][8] per-wave head maxima
  int* shi = reinterpret_cast<int*>(mpart + kLdsWaves * kH);        // [8] broadcast: action, done
  uint8_t* mk = reinterpret_cast<uint8_t*>(shi + 8);                // [nw] 1 = feasible
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
  const int iters = ITERS > 0 ? ITERS : (N + kLdsGroups - 1) / kLdsGroups;
  const float* ctxc = a.ctx_cur + (int64_t)cb * N * kD + e0;
  const float* ctxf = (ENV == RL4CO_ENV_TSP) ? a.ctx_first + (int64_t)cb * N * kD + e0 : nullptr;
  // o/l partial slots inside this wave's own (dead after pass 2) score rows
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
  uint8_t* gmask = a.action_mask + (int64_t)r * N;
  for (int j = tid; j < nw; j += 64 * kLdsWaves) mk[j] = (j < N) ? gmask[j] : (uint8_t)0;
  if (ENV == RL4CO_ENV_CVRP) {
    const uint8_t* gv = a.visited + (int64_t)r * N;
    for (int j = tid; j < nw; j += 64 * kLdsWaves) vis[j] = (j < N) ? gv[j] : (uint8_t)1;
  }
  int cur = (int)a.current_node[r];
  int first = (ENV == RL4CO_ENV_TSP) ? (int)a.first_node[r] : 0;
  long long step_i = (ENV == RL4CO_ENV_TSP) ? a.step_i[r] : 0;
  float used = (ENV == RL4CO_ENV_CVRP) ? a.used_capacity[r] : 0.0f;
  const float cap = (ENV == RL4CO_ENV_CVRP) ? a.vehicle_capacity[r] : 0.0f;
  const float* dem = (ENV == RL4CO_ENV_CVRP) ? a.demand + (int64_t)cb * (N - 1) : nullptr;
  bool done = a.done[r] != 0;
  __syncthreads();

  float qb[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) qb[e] = a.q_bias ? a.q_bias[(int64_t)cb * kD + e0 + e] : 0.0f;

  const float sqrt_d = 11.3137084989847604f;
  const bool single = a.max_steps == 1;
  uint32_t errbits = 0;
  float ent_acc = 0.0f;
  int t = 0;
  for (; t < a.max_steps && (!done || single); ++t) {
    // ---- query ---------------------------------------------------------------------------------
    float q[EPL];
    if (ENV == RL4CO_ENV_TSP) {
      if (step_i < 1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = a.q_step0[e0 + e] + qb[e];
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
          q[e] = (ctxf[(int64_t)first * kD + e] + ctxc[(int64_t)cur * kD + e]) + qb[e];
      }
    } else {
      const float rem = cap - used;
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        q[e] = fmaf(a.w_cap[e0 + e], rem, ctxc[(int64_t)cur * kD + e]) + qb[e];
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] = q[e] * 0.25f;

    // ---- pass 1: scores of this wave's rows ------------------------------------------------------
    float m = kNegInf;
    for (int i0 = 0; i0 < iters; i0 += U) {
      uint4 rw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = kLdsGroups * (i0 + u) + 4 * w + rg;
        rw[u] = (j < N) ? *reinterpret_cast<const uint4*>(Kg + (int64_t)j * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = kLdsGroups * (i0 + u) + 4 * w + rg;
        const bool valid = j < N;
        float k[EPL];
        C::cvt(rw[u], k);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc = fmaf(q[e], k[e], acc);
        acc = rl4co::bfly_sum<1, LPH>(acc);
        const bool feas = valid && (!a.mask_inner || mk[valid ? j : 0] != 0);
        const float sv = feas ? acc : kNegInf;
        if (valid && (li & 1) == 0) sc[j * kH + hd] = sv;
        m = fmaxf(m, sv);
      }
    }
    m = rl4co::bfly_max<LPR, 64>(m);
    if (rg == 0 && (li & 1) == 0) mpart[w * kH + hd] = m;
    __syncthreads();  // B1: all head maxima visible
    // softmax numerators of this wave's rows, each evaluated once (lane-strided over the wave's
    // 4-row chunks: 32 consecutive floats per chunk, head = index % 8 fixed per lane)
    {
      const int hh = lane & (kH - 1);
      const float mm = fmaxf(fmaxf(mpart[hh], mpart[kH + hh]), fmaxf(mpart[2 * kH + hh], mpart[3 * kH + hh]));
      for (int tix = lane; tix < 32 * iters; tix += 64) {
        const int row0 = kLdsGroups * (tix >> 5) + 4 * w;  // first row of the chunk
        if (row0 + ((tix & 31) >> 3) < N) {
          float* p = sc + row0 * kH + (tix & 31);
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
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = kLdsGroups * (i0 + u) + 4 * w + rg;
        rw[u] = (j < N) ? *reinterpret_cast<const uint4*>(Vg + (int64_t)j * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = kLdsGroups * (i0 + u) + 4 * w + rg;
        const bool valid = j < N;
        float v[EPL];
        C::cvt(rw[u], v);
        const float p = valid ? sc[j * kH + hd] : 0.0f;
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

    // ---- pass 3: logits of this wave's rows ---------------------------------------------------------
    for (int i0 = 0; i0 < iters; i0 += U) {
      uint4 rw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = kLdsGroups * (i0 + u) + 4 * w + rg;
        rw[u] = (j < N) ? *reinterpret_cast<const uint4*>(Kl + (int64_t)j * rs) : C::zero();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = kLdsGroups * (i0 + u) + 4 * w + rg;
        float k[EPL];
        C::cvt(rw[u], k);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc = fmaf(o[e], k[e], acc);
        acc = rl4co::bfly_sum<1, LPR>(acc);
        if (j < N && li == 0) lg[j] = acc;
      }
    }
    __syncthreads();  // B3: all logits visible to wave 0

    // ---- wave 0: log-softmax, selection, environment transition (same code as streaming) ----------
    if (w == 0) {
      bool nan_seen = false;
      float zmax = kNegInf;
      for (int j = lane; j < N; j += 64) {
        float z = lg[j] / sqrt_d;
        if (z != z) nan_seen = true;
        if (a.tanh_clipping > 0.0f) z = rl4co_tanhf(z) * a.tanh_clipping;
        if (a.mask_logits && mk[j] == 0) z = kNegInf;
        if (a.temperature != 1.0f) z = z / a.temperature;
        lg[j] = z;
        zmax = fmaxf(zmax, z);
      }
      if (__any(nan_seen)) errbits |= RL4CO_EBIT_NAN_LOGIT;
      zmax = rl4co::bfly_max<1, 64>(zmax);
      float zsum = 0.0f;
      for (int j = lane; j < N; j += 64) zsum = zsum + rl4co_expf(lg[j] - zmax);
      zsum = rl4co::bfly_sum<1, 64>(zsum);
      const float lse = rl4co_logf(zsum);
      float best = kNegInf;
      int bi = 0x7fffffff;
      float ent = 0.0f;
      const int64_t tcol = (int64_t)a.t0 + t;
      for (int j = lane; j < N; j += 64) {
        const float lp = (lg[j] - zmax) - lse;
        lg[j] = lp;
        float key = lp;
        if (a.mode == RL4CO_DECODE_SAMPLE) {
          const float nz = a.exp_noise
                               ? a.exp_noise[((int64_t)t * a.B + r) * N + j]
                               : rl4co_exp1_noise(a.philox_seed, a.philox_offset + (uint64_t)tcol,
                                                  (uint32_t)r, (uint32_t)j);
          key = rl4co_expf(lp) / nz;
        }
        if (bi == 0x7fffffff || key > best) {
          best = key;
          bi = j;
        }
        if (a.entropy && lp > kNegInf) ent = fmaf(rl4co_expf(lp), lp, ent);
        if (a.all_logps) a.all_logps[((int64_t)r * a.out_stride + tcol) * N + j] = lp;
      }
      rl4co::bfly_argmax(best, bi);
      if (a.entropy) ent_acc = ent_acc - rl4co::bfly_sum<1, 64>(ent);
      if (a.mode == RL4CO_DECODE_EVALUATE) bi = (int)a.forced_actions[(int64_t)r * a.out_stride + tcol];
      if (bi < 0 || bi >= N) {
        errbits |= RL4CO_EBIT_INFEASIBLE;
        bi = 0;
      }
      wave_lds_sync();
      const float logp = lg[bi];
      if (mk[bi] == 0) errbits |= RL4CO_EBIT_INFEASIBLE;
      if (!(logp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;
      if (lane == 0) {
        a.actions[(int64_t)r * a.out_stride + tcol] = bi;
        a.logps[(int64_t)r * a.out_stride + tcol] = logp;
      }
      wave_lds_sync();
      bool new_done;
      if (ENV == RL4CO_ENV_TSP) {
        if (lane == 0) mk[bi] = 0;
        wave_lds_sync();
        bool any_left = false;
        for (int j = lane; j < N; j += 64) any_left |= mk[j] != 0;
        new_done = !__any(any_left);
      } else {
        const int di = min(max(bi - 1, 0), N - 2);
        const float used_next = (used + dem[di]) * (bi != 0 ? 1.0f : 0.0f);
        if (lane == 0) vis[bi] = 1;
        wave_lds_sync();
        const float thr = cap + 1e-5f;
        bool any_feasible = false, all_visited = true;
        for (int j = lane; j < N; j += 64) {
          all_visited &= vis[j] != 0;
          if (j >= 1) {
            const bool masked = (vis[j] != 0) || (dem[j - 1] + used_next > thr);
            mk[j] = masked ? 0 : 1;
            any_feasible |= !masked;
          }
        }
        any_feasible = __any(any_feasible);
        new_done = __all(all_visited);
        if (lane == 0) mk[0] = ((bi == 0) && any_feasible) ? 0 : 1;
      }
      if (lane == 0) {
        shi[0] = bi;
        shi[1] = new_done ? 1 : 0;
      }
    }
    __syncthreads();  // B4: action, done flag and the updated mask visible to every wave
    {
      const int bi = shi[0];
      done = shi[1] != 0;
      if (ENV == RL4CO_ENV_TSP) {
        if (step_i == 0) first = bi;
        cur = bi;
        step_i += 1;
      } else {
        const int di = min(max(bi - 1, 0), N - 2);
        used = (used + dem[di]) * (bi != 0 ? 1.0f : 0.0f);
        cur = bi;
      }
    }
  }
  if (w == 0) {
    if (!single && !done && t >= a.max_steps) errbits |= RL4CO_EBIT_MAX_STEPS;
    for (int j = lane; j < N; j += 64) gmask[j] = mk[j];
    if (ENV == RL4CO_ENV_CVRP) {
      uint8_t* gv = a.visited + (int64_t)r * N;
      for (int j = lane; j < N; j += 64) gv[j] = vis[j];
    }
    if (lane == 0) {
      a.current_node[r] = cur;
      a.done[r] = done ? 1 : 0;
      if (ENV == RL4CO_ENV_TSP) {
        a.first_node[r] = first;
        a.step_i[r] = step_i;
      } else {
        a.used_capacity[r] = used;
      }
      if (a.n_steps) a.n_steps[r] = t;
      if (a.entropy) a.entropy[r] += ent_acc;
      if (errbits) atomicOr(a.err, (int)errbits);
    }
  }
}

template <int ENV, int ITERS, bool RESIDENT>
int launch_wide(const rl4co_am_decode_args& a, hipStream_t stream) {
  const int lds = RESIDENT ? lds_variant_bytes(a.N) : wide_scratch_bytes(a.N);
  if (lds > 64 * 1024) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_decode_lds_kernel<ENV, ITERS, RESIDENT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  hipLaunchKernelGGL((am_decode_lds_kernel<ENV, ITERS, RESIDENT>), dim3(a.B), dim3(64 * kLdsWaves), lds, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

template <int ENV>
int launch_lds(const rl4co_am_decode_args& a, hipStream_t stream) {
  switch ((a.N + kLdsGroups - 1) / kLdsGroups) {  // rows per wave per pass, fully unrolled
    case 1: return launch_wide<ENV, 1, true>(a, stream);
    case 2: return launch_wide<ENV, 2, true>(a, stream);
    case 3: return launch_wide<ENV, 3, true>(a, stream);
    case 4: return launch_wide<ENV, 4, true>(a, stream);
    case 5: return launch_wide<ENV, 5, true>(a, stream);
    case 6: return launch_wide<ENV, 6, true>(a, stream);
    case 7: return launch_wide<ENV, 7, true>(a, stream);
    default: return rl4co::record_arg_error("LDS-resident decode variant supports N <= 112");
  }
}

// Which kernel serves these arguments. Measured on MI355X (TSP-100, bf16): with thousands of
// trajectories the streaming kernel keeps 16 independent waves per CU in flight and runs at ~0.9
// of the HBM peak (4.4 ms per 4096 x 100 steps), while the LDS-resident kernel can only host two
// trajectories per CU and becomes latency-bound (6.1 ms). The resident kernel wins when there
// are too few trajectories to fill the chip with one wave each: it puts 4 waves on every
// trajectory and takes HBM out of the per-step critical path (measured: 0.70 vs 1.54 ms at B = 256,
// 1.54 vs 1.84 ms at B = 1024, 3.06 vs 2.12 ms at B = 2048). Auto picks it for B <= 1024.
inline int resolve_variant(const rl4co_am_decode_args& a) {
  const bool bf16 = a.cache_dtype == RL4CO_DT_BF16;
  const bool fits = bf16 && lds_variant_bytes(a.N) <= 80 * 1024 && (a.N + kLdsGroups - 1) / kLdsGroups <= 7;
  const bool wide_ok = bf16 && wide_scratch_bytes(a.N) <= 64 * 1024;
  if (a.variant == RL4CO_VARIANT_STREAM) return RL4CO_VARIANT_STREAM;
  if (a.variant == RL4CO_VARIANT_LDS) return fits ? RL4CO_VARIANT_LDS : -1;
  if (a.variant == RL4CO_VARIANT_WIDE) return wide_ok ? RL4CO_VARIANT_WIDE : -1;
  if (a.max_steps < 4) return RL4CO_VARIANT_STREAM;
  if (fits && a.B <= 1024) return RL4CO_VARIANT_LDS;
  // one wave per trajectory needs >= ~16 waves per CU to hide its latency chain: with fewer
  // trajectories than that, four waves per trajectory keep the memory pipes busier
  if (wide_ok && a.B <= 2048) return RL4CO_VARIANT_WIDE;
  return RL4CO_VARIANT_STREAM;
}

template <class C, int ENV>
int launch(const rl4co_am_decode_args& a, hipStream_t stream) {
  const int lds = rl4co_am_decode_lds_bytes(a.N, ENV);
  if (lds > 64 * 1024) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_decode_kernel<C, ENV>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  hipLaunchKernelGGL((am_decode_kernel<C, ENV>), dim3(a.B), dim3(64), lds, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

}  // namespace

extern "C" int rl4co_am_decode_lds_bytes(int N, int env) {
  (void)env;
  const int Np = lds_pad(N);
  return Np * kH * 4 + Np * 4 + Np + Np + kH * 4;
}

extern "C" int rl4co_am_decode_row_groups(const rl4co_am_decode_args* args) {
  if (args == nullptr) return -1;
  const int v = resolve_variant(*args);
  if (v < 0) return -1;
  if (v == RL4CO_VARIANT_LDS || v == RL4CO_VARIANT_WIDE) return kLdsGroups;
  return args->cache_dtype == RL4CO_DT_BF16 ? 64 / (kD / CacheBF16::EPL) : 64 / (kD / CacheF32::EPL);
}

extern "C" int rl4co_am_decode_variant(const rl4co_am_decode_args* args) {
  return args == nullptr ? -1 : resolve_variant(*args);
}

extern "C" int rl4co_am_decode(const rl4co_am_decode_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_decode_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP);
  RL4CO_REQUIRE(a.B > 0 && a.B_inst > 0 && a.B % a.B_inst == 0);
  RL4CO_REQUIRE(a.N >= 2 && a.N <= 4096);
  RL4CO_REQUIRE(a.max_steps >= 1);
  RL4CO_REQUIRE(a.mode >= RL4CO_DECODE_GREEDY && a.mode <= RL4CO_DECODE_EVALUATE);
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == RL4CO_DT_BF16);
  RL4CO_REQUIRE(a.glimpse_key && a.glimpse_val && a.logit_key && a.ctx_cur);
  RL4CO_REQUIRE(a.kvl_row_stride >= kD && a.kvl_row_stride % 8 == 0);
  RL4CO_REQUIRE(a.kvl_batch_stride >= (int64_t)a.N * kD && a.kvl_batch_stride % 8 == 0);
  RL4CO_REQUIRE(a.action_mask && a.current_node && a.done && a.actions && a.logps && a.err);
  RL4CO_REQUIRE(a.out_stride >= 1 && a.t0 >= 0 && (int64_t)a.t0 + a.max_steps <= a.out_stride);
  RL4CO_REQUIRE(a.temperature > 0.0f);
  RL4CO_REQUIRE(a.mode != RL4CO_DECODE_EVALUATE || a.forced_actions != nullptr);
  if (a.env == RL4CO_ENV_TSP) {
    RL4CO_REQUIRE(a.ctx_first && a.q_step0 && a.first_node && a.step_i);
  } else {
    RL4CO_REQUIRE(a.w_cap && a.demand && a.used_capacity && a.vehicle_capacity && a.visited);
  }
  RL4CO_REQUIRE(rl4co_am_decode_lds_bytes(a.N, a.env) <= 160 * 1024);
  RL4CO_REQUIRE(a.variant >= RL4CO_VARIANT_AUTO && a.variant <= RL4CO_VARIANT_WIDE);
  const int variant = resolve_variant(a);
  RL4CO_REQUIRE(variant >= 0);  // RL4CO_VARIANT_LDS requested but the planes do not fit / are not bf16
  hipStream_t s = rl4co::as_stream(stream);
  if (variant == RL4CO_VARIANT_LDS) {
    return a.env == RL4CO_ENV_TSP ? launch_lds<RL4CO_ENV_TSP>(a, s) : launch_lds<RL4CO_ENV_CVRP>(a, s);
  }
  if (variant == RL4CO_VARIANT_WIDE) {
    return a.env == RL4CO_ENV_TSP ? launch_wide<RL4CO_ENV_TSP, 0, false>(a, s)
                                  : launch_wide<RL4CO_ENV_CVRP, 0, false>(a, s);
  }
  if (a.cache_dtype == RL4CO_DT_F32) {
    return a.env == RL4CO_ENV_TSP ? launch<CacheF32, RL4CO_ENV_TSP>(a, s)
                                  : launch<CacheF32, RL4CO_ENV_CVRP>(a, s);
  }
  return a.env == RL4CO_ENV_TSP ? launch<CacheBF16, RL4CO_ENV_TSP>(a, s)
                                : launch<CacheBF16, RL4CO_ENV_CVRP>(a, s);
}
