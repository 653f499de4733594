// am_teacher.hip — teacher-forced log-likelihood of given trajectories, forward AND backward,
// for the REINFORCE / POMO training step.
//
// The reference differentiates log_likelihood through the autograd graph of its T-step Python
// decode loop (rl/reinforce/reinforce.py:99-102, models/common/constructive/base.py:226-263);
// with the actions known (decode_type="evaluate", utils/decoding.py:448-461, the pattern PPO
// already uses: rl/ppo/ppo.py:128-170) every step can be recomputed independently of sampling.
// This kernel replays the trajectories step by step and accumulates dL/d(cache) for
// L = sum_{r,t} g[r,t] * log p(a[r,t] | state), i.e. the gradient w.r.t. the folded decoder cache
// (three planes, context tables, graph context, placeholder query / capacity column). torch
// autograd carries it on through the fold GEMMs and the encoder.
//
// Environments: all six of the decode kernel (the transition of each is replayed step by step with the given action,
// exactly as am_decode.hip's finalize_and_step does; pickup-delivery's reset flavour is read off the first action).
//
// One 256-thread workgroup per INSTANCE; its S trajectories (multistart) are replayed one after
// the other. Every lane owns a fixed set of cache elements — rows j = 16 i + 4 w + rg (i < ROWS),
// dims 8 li .. 8 li + 7 — of all three planes: they are loaded into registers ONCE per instance
// and their gradients are accumulated in registers over all S x T steps (no atomics, no plane
// traffic inside the loop; N <= 16 * ROWS <= 128). Per step the four waves exchange only the
// 128-wide reductions (glimpse, d heads, d query) and a few scalars through LDS.
// fp32 arithmetic throughout; this is a floating-point kernel tested against torch autograd
// (tests/test_gpu_teacher.py, tolerance stated there), not part of the bit-exact decode contract.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kH = RL4CO_NUM_HEADS;
constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kSqrtD = 11.3137084989847604f;
constexpr int kW = 4;  // waves per workgroup

__device__ inline float load_plane(const void* base, int dtype, int64_t idx) {
  if (dtype == RL4CO_DT_BF16) return __uint_as_float((uint32_t)static_cast<const uint16_t*>(base)[idx] << 16);
  if (dtype == RL4CO_DT_F16) return (float)static_cast<const _Float16*>(base)[idx];
  return static_cast<const float*>(base)[idx];
}

// sum over the 4 row groups of a wave (lanes differing in bits 4,5), result on every lane
__device__ inline float rg_sum(float v) { return rl4co::bfly_sum<16, 64>(v); }
__device__ inline float rg_max(float v) { return rl4co::bfly_max<16, 64>(v); }

template <int ENV, int ROWS>
__global__ void __launch_bounds__(64 * kW) am_teacher_kernel(const rl4co_am_teacher_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63;
  const int rg = lane >> 4, li = lane & 15, hd = li >> 1, e0 = li * 8;
  const int inst = blockIdx.x;
  const int N = a.N, T = a.T, S = a.B / a.B_inst;
  const int nw = (N + 3) & ~3;

  float* mpart = reinterpret_cast<float*>(smem);   // [4][8]
  float* lpart = mpart + kW * kH;                   // [4][8]
  float* adap = lpart + kW * kH;                    // [4][8]
  float* lsep = adap + kW * kH;                     // [4][2] (max, sum)
  float* shf = lsep + kW * 2;                       // [8] misc floats
  int* shi = reinterpret_cast<int*>(shf + 8);       // [8] misc ints
  float* opart = reinterpret_cast<float*>(shi + 8); // [4][128]
  float* dhpart = opart + kW * kD;                  // [4][128]
  float* dqpart = dhpart + kW * kD;                 // [4][128]
  float* dctx = dqpart + kW * kD;                   // [N][128] d ctx_cur of this instance
  uint8_t* mk = reinterpret_cast<uint8_t*>(dctx + N * kD);  // [nw]
  uint8_t* vis = mk + nw;                                   // [nw] visited (PDP: NOT available)
  uint8_t* tod = vis + nw;                                  // [nw] PDP: to_deliver

  // ---- this lane's slice of the three planes, resident in registers -------------------------
  float kg[ROWS][8], vv[ROWS][8], kl[ROWS][8];
  float dkg[ROWS][8], dvv[ROWS][8], dkl[ROWS][8];
  const int64_t pbase = (int64_t)inst * a.kvl_batch_stride;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int j = 16 * i + 4 * w + rg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t idx = pbase + (int64_t)j * a.kvl_row_stride + e0 + e;
      const bool ok = j < N;
      kg[i][e] = ok ? load_plane(a.glimpse_key, a.cache_dtype, idx) : 0.0f;
      vv[i][e] = ok ? load_plane(a.glimpse_val, a.cache_dtype, idx) : 0.0f;
      kl[i][e] = ok ? load_plane(a.logit_key, a.cache_dtype, idx) : 0.0f;
      dkg[i][e] = 0.0f;
      dvv[i][e] = 0.0f;
      dkl[i][e] = 0.0f;
    }
  }
  for (int idx = tid; idx < N * kD; idx += 64 * kW) dctx[idx] = 0.0f;

  const float* ctxc = static_cast<const float*>(a.ctx_cur) + (int64_t)inst * N * kD + e0;
  const float* ctxf = (ENV == RL4CO_ENV_TSP) ? static_cast<const float*>(a.ctx_first) + (int64_t)inst * N * kD + e0 : nullptr;
  constexpr bool kCvrpLike = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kClock = ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kScalar = ENV != RL4CO_ENV_TSP && ENV != RL4CO_ENV_PDP;  // one context scalar, cap - used
  const float* dem = kCvrpLike ? a.demand + (int64_t)inst * (N - 1)
                               : (ENV == RL4CO_ENV_PCTSP ? a.demand + (int64_t)inst * N : nullptr);  // PCTSP: real prize [N]
  const float* locs = (ENV == RL4CO_ENV_OP || kClock) ? a.locs + (int64_t)inst * N * 2 : nullptr;
  const float* opmax = (ENV == RL4CO_ENV_OP) ? a.max_length + (int64_t)inst * N : nullptr;
  const float* twin = kClock ? a.time_windows + (int64_t)inst * N * 2 : nullptr;
  const float* dur = kClock ? a.durations + (int64_t)inst * N : nullptr;
  // cap - used: vehicle capacity (CVRP / CVRPTW), prize still required (PCTSP, clamped at 0), longest tour that may
  // still end at the depot minus the tour so far (OP) — env_embeddings/context.py:105-213
  const float cap = (kCvrpLike || ENV == RL4CO_ENV_PCTSP) ? a.vehicle_capacity[inst] : (ENV == RL4CO_ENV_OP ? opmax[0] : 0.0f);
  auto dist = [&](int i, int j) {  // (locs[j] - locs[i]).norm(p=2, dim=-1)
    const float dx = locs[2 * j] - locs[2 * i], dy = locs[2 * j + 1] - locs[2 * i + 1];
    return sqrtf(fmaf(dy, dy, dx * dx));
  };
  float qb[8], dqb[8], dqs0[8], dwc[8], dwt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    qb[e] = a.q_bias ? a.q_bias[(int64_t)inst * kD + e0 + e] : 0.0f;
    dqb[e] = 0.0f;
    dqs0[e] = 0.0f;
    dwc[e] = 0.0f;
    dwt[e] = 0.0f;
  }
  const float inv_temp = 1.0f / a.temperature;
  uint32_t errbits = 0;

  for (int s = 0; s < S; ++s) {
    const int r = s * a.B_inst + inst;
    const int64_t* act = a.actions + (int64_t)r * T;
    const float* gl = a.grad_logp + (int64_t)r * T;
    // ---- reset state (tsp/env.py:88-113, cvrp/env.py:98-136) ---------------------------------
    __syncthreads();
    // a trajectory that opens with the depot in pickup-delivery was reset with force_start_at_depot (pdp/env.py:118-126):
    // only the depot is open at column 0; otherwise the depot is never available and the pickups are open
    const bool pdp_depot_start = ENV == RL4CO_ENV_PDP && act[0] == 0;
    for (int j = tid; j < nw; j += 64 * kW) {
      uint8_t m_ = 0, v_ = 1, d_ = 0;
      if (j < N) {
        v_ = 0;
        if (ENV == RL4CO_ENV_TSP) m_ = 1;
        else if (kCvrpLike) m_ = (j == 0) ? 0 : 1;  // fresh CVRP: depot masked (every demand fits an empty vehicle)
        else if (ENV == RL4CO_ENV_OP) m_ = (j == 0) ? 1 : ((0.0f + dist(0, j) > opmax[j]) ? 0 : 1);  // op/env.py:137-154
        else if (ENV == RL4CO_ENV_PCTSP) m_ = (j == 0) ? 0 : 1;  // pctsp/env.py:141-148: no prize yet, customers left
        else {                                                   // PDP
          const int half = (N - 1) / 2;
          d_ = j <= half ? 1 : 0;
          v_ = (!pdp_depot_start && j == 0) ? 1 : 0;  // vis = NOT available
          m_ = pdp_depot_start ? (j == 0 ? 1 : 0) : ((j >= 1 && j <= half) ? 1 : 0);
        }
        if (kClock && !(0.0f + dist(0, j) <= twin[2 * j + 1])) m_ = 0;  // cvrptw/env.py:91-95 at the depot, time 0
      }
      mk[j] = m_;
      vis[j] = v_;
      tod[j] = d_;
    }
    int cur = 0, first = 0;
    long long step_i = 0;
    float used = 0.0f, now = 0.0f;
    bool done = false;
    float dqf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dqf[e] = 0.0f;
    __syncthreads();

    for (int t = 0; t < T && !done; ++t) {
      int at = (int)act[t];
      if (at < 0 || at >= N) {
        errbits |= RL4CO_EBIT_INFEASIBLE;
        at = 0;
      }
      const bool decoded = t >= a.t0;  // multistart: column 0 is imposed, log-prob 0 (decoding.py:306-326)
      if (decoded) {
        const float g = gl[t];
        // ---- query -----------------------------------------------------------------------------
        float q[8];
        float rem = cap - used;
        if (ENV == RL4CO_ENV_PCTSP && !(rem > 0.0f)) rem = 0.0f;  // clamp(min=0), context.py:195
        if (ENV == RL4CO_ENV_TSP) {
          if (step_i < 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = a.q_step0[e0 + e] + qb[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = (ctxf[(int64_t)first * kD + e] + ctxc[(int64_t)cur * kD + e]) + qb[e];
          }
        } else if (ENV == RL4CO_ENV_PDP) {
#pragma unroll
          for (int e = 0; e < 8; ++e) q[e] = ctxc[(int64_t)cur * kD + e] + qb[e];  // context.py:232-243
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = fmaf(a.w_cap[e0 + e], rem, ctxc[(int64_t)cur * kD + e]);
            if (kClock) v = fmaf(a.w_time[e0 + e], now, v);  // context.py:152-166
            q[e] = v + qb[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] *= 0.25f;

        // ---- forward: scores, softmax, glimpse --------------------------------------------------
        float sv[ROWS];
        bool feas[ROWS];
        float m = kNegInf;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const int j = 16 * i + 4 * w + rg;
          float acc = 0.0f;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = fmaf(q[e], kg[i][e], acc);
          acc += rl4co::bfly_f<1>(acc);
          feas[i] = j < N && mk[j < N ? j : 0] != 0;
          const bool in_glimpse = j < N && (!a.mask_inner || feas[i]);
          sv[i] = in_glimpse ? acc : kNegInf;
          m = fmaxf(m, sv[i]);
        }
        m = rg_max(m);
        if (rg == 0 && (li & 1) == 0) mpart[w * kH + hd] = m;
        __syncthreads();  // B1
        m = fmaxf(fmaxf(mpart[hd], mpart[kH + hd]), fmaxf(mpart[2 * kH + hd], mpart[3 * kH + hd]));
        float l = 0.0f, o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.0f;
        float p[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          p[i] = __expf(sv[i] - m);
          l += p[i];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = fmaf(p[i], vv[i][e], o[e]);
        }
        l = rg_sum(l);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rg_sum(o[e]);
        if (rg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) opart[w * kD + e0 + e] = o[e];
          if ((li & 1) == 0) lpart[w * kH + hd] = l;
        }
        __syncthreads();  // B2
        l = (lpart[hd] + lpart[kH + hd]) + (lpart[2 * kH + hd] + lpart[3 * kH + hd]);
        const float inv_l = 1.0f / l;
        float heads[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = e0 + e;
          heads[e] = ((opart[d] + opart[kD + d]) + (opart[2 * kD + d] + opart[3 * kD + d])) * inv_l;
        }
        // ---- forward: logits, clip, log-softmax -----------------------------------------------------
        float z[ROWS], dzdu[ROWS];
        float zm = kNegInf;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const int j = 16 * i + 4 * w + rg;
          float acc = 0.0f;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = fmaf(heads[e], kl[i][e], acc);
          acc = rl4co::bfly_sum<1, 16>(acc);
          const float u = acc / kSqrtD;
          if (u != u) errbits |= RL4CO_EBIT_NAN_LOGIT;
          float zz, dd;
          if (a.tanh_clipping > 0.0f) {
            const float ex = __expf(-2.0f * fabsf(u));
            const float th = copysignf((1.0f - ex) / (1.0f + ex), u);
            zz = th * a.tanh_clipping * inv_temp;
            dd = a.tanh_clipping * inv_temp * (1.0f - th * th);
          } else {
            zz = u * inv_temp;
            dd = inv_temp;
          }
          const bool in_logits = j < N && (!a.mask_logits || feas[i]);
          z[i] = in_logits ? zz : kNegInf;
          dzdu[i] = dd;
          zm = fmaxf(zm, z[i]);
        }
        zm = rg_max(zm);
        float se = 0.0f;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) se += __expf(z[i] - zm);  // exp(-inf - zm) = 0; zm = -inf only if no row
        se = rg_sum(se);
        if (lane == 0) {
          lsep[w * 2] = zm;
          lsep[w * 2 + 1] = (zm > kNegInf) ? se : 0.0f;
        }
        __syncthreads();  // B3
        const float zmax = fmaxf(fmaxf(lsep[0], lsep[2]), fmaxf(lsep[4], lsep[6]));
        float tot = 0.0f;
#pragma unroll
        for (int ww = 0; ww < kW; ++ww)
          tot += (lsep[ww * 2] > kNegInf) ? lsep[ww * 2 + 1] * __expf(lsep[ww * 2] - zmax) : 0.0f;
        const float lse = zmax + __logf(tot);
        // log p(a_t): written by the lane that owns row a_t
        float dd_row[ROWS];  // d L / d (heads . kl_row), i.e. through clip and 1/sqrt(d)
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const int j = 16 * i + 4 * w + rg;
          const float prob = __expf(z[i] - lse);  // 0 for masked rows
          const float dz = g * ((j == at ? 1.0f : 0.0f) - prob);
          dd_row[i] = (z[i] > kNegInf) ? dz * dzdu[i] / kSqrtD : 0.0f;
          if (j == at && li == 0) {
            const float lp = z[i] - lse;
            if (a.logp_out) a.logp_out[(int64_t)r * T + t] = lp;
            if (!(lp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;
          }
        }
        // ---- backward: logits -> heads, logit keys ------------------------------------------------
        float dh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dh[e] = 0.0f;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            dh[e] = fmaf(dd_row[i], kl[i][e], dh[e]);
            dkl[i][e] = fmaf(dd_row[i], heads[e], dkl[i][e]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) dh[e] = rg_sum(dh[e]);
        if (rg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dhpart[w * kD + e0 + e] = dh[e];
        }
        __syncthreads();  // B4
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = e0 + e;
          dh[e] = (dhpart[d] + dhpart[kD + d]) + (dhpart[2 * kD + d] + dhpart[3 * kD + d]);
        }
        // ---- backward: glimpse attention (softmax over nodes, per head) ---------------------------
        float av[ROWS], da[ROWS];
        float ada = 0.0f;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          av[i] = p[i] * inv_l;
          float acc = 0.0f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            acc = fmaf(dh[e], vv[i][e], acc);
            dvv[i][e] = fmaf(av[i], dh[e], dvv[i][e]);
          }
          acc += rl4co::bfly_f<1>(acc);  // the head's 16 dims live on a lane pair
          da[i] = acc;
          ada = fmaf(av[i], acc, ada);
        }
        ada = rg_sum(ada);
        if (rg == 0 && (li & 1) == 0) adap[w * kH + hd] = ada;
        __syncthreads();  // B5
        ada = (adap[hd] + adap[kH + hd]) + (adap[2 * kH + hd] + adap[3 * kH + hd]);
        float dq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dq[e] = 0.0f;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const float ds = av[i] * (da[i] - ada);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            dq[e] = fmaf(ds, kg[i][e], dq[e]);
            dkg[i][e] = fmaf(ds, q[e], dkg[i][e]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) dq[e] = rg_sum(dq[e]);
        if (rg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dqpart[w * kD + e0 + e] = dq[e];
        }
        __syncthreads();  // B6
        // ---- backward: query -> context rows, graph context, placeholder / capacity column ------------
        if (w == 0 && rg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int d = e0 + e;
            const float dqr = 0.25f * ((dqpart[d] + dqpart[kD + d]) + (dqpart[2 * kD + d] + dqpart[3 * kD + d]));
            dqb[e] += dqr;
            if (ENV == RL4CO_ENV_TSP) {
              if (step_i < 1) {
                dqs0[e] += dqr;
              } else {
                dqf[e] += dqr;
                dctx[cur * kD + d] += dqr;
              }
            } else {
              if (kScalar) dwc[e] = fmaf(dqr, rem, dwc[e]);
              if (kClock) dwt[e] = fmaf(dqr, now, dwt[e]);
              dctx[cur * kD + d] += dqr;
            }
          }
        }
        if (mk[at] == 0) errbits |= RL4CO_EBIT_INFEASIBLE;
      }
      // ---- environment transition with the given action (all waves keep the scalars) -------------
      __syncthreads();
      if (ENV == RL4CO_ENV_TSP) {
        if (step_i == 0) first = at;
        cur = at;
        step_i += 1;
        if (tid == 0) mk[at] = 0;
        __syncthreads();
        if (w == 0) {
          bool any_left = false;
          for (int j = lane; j < N; j += 64) any_left |= mk[j] != 0;
          any_left = __any(any_left);
          if (lane == 0) shi[0] = any_left ? 0 : 1;
        }
      } else if (ENV == RL4CO_ENV_PDP) {
        // pdp/env.py:64-83: the node leaves `available`, its delivery becomes deliverable; done when nothing is available
        const int n = N - 1;
        cur = at;
        step_i += 1;
        if (tid == 0) {
          vis[at] = 1;
          tod[(at + n / 2) % (n + 1)] = 1;
        }
        __syncthreads();
        if (w == 0) {
          bool left = false;
          for (int j = lane; j < N; j += 64) {
            const bool avail = vis[j] == 0;
            mk[j] = (avail && tod[j] != 0) ? 1 : 0;
            left |= avail;
          }
          left = __any(left);
          if (lane == 0) shi[0] = left ? 0 : 1;
        }
      } else if (ENV == RL4CO_ENV_PCTSP) {
        // pctsp/env.py:62-75, 141-148
        used = used + dem[at];
        const bool fin = (step_i > 0) && (at == 0);
        step_i += 1;
        cur = at;
        if (tid == 0) vis[at] = 1;
        __syncthreads();
        if (w == 0) {
          const bool closed = vis[0] != 0;
          bool unvisited = false;
          for (int j = lane; j < N; j += 64) {
            if (j >= 1) {
              mk[j] = (vis[j] != 0 || closed) ? 0 : 1;
              unvisited |= vis[j] == 0;
            }
          }
          unvisited = __any(unvisited);
          if (lane == 0) {
            mk[0] = ((used < 1.0f) && unvisited) ? 0 : 1;
            shi[0] = fin ? 1 : 0;
          }
        }
      } else if (ENV == RL4CO_ENV_OP) {
        // op/env.py:67-98, 137-154
        used = used + dist(cur, at);
        const bool fin = (at == 0) && (step_i > 0);
        step_i += 1;
        cur = at;
        if (tid == 0) vis[at] = 1;
        __syncthreads();
        if (w == 0) {
          const bool closed = vis[0] != 0;
          for (int j = lane; j < N; j += 64) {
            const bool exceeds = used + dist(cur, j) > opmax[j];
            mk[j] = (j == 0) ? 1 : ((vis[j] != 0 || closed || exceeds) ? 0 : 1);
          }
          if (lane == 0) shi[0] = fin ? 1 : 0;
        }
      } else {
        if (kClock) now = (at != 0 ? 1.0f : 0.0f) * (fmaxf(now + dist(cur, at), twin[2 * at]) + dur[at]);  // cvrptw/env.py:97-113
        const int di = min(max(at - 1, 0), N - 2);
        used = (used + dem[di]) * (at != 0 ? 1.0f : 0.0f);
        cur = at;
        if (tid == 0) vis[at] = 1;
        __syncthreads();
        if (w == 0) {
          const float thr = cap + 1e-5f;
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
          all_visited = __all(all_visited);
          if (lane == 0) {
            mk[0] = ((cur == 0) && any_feasible) ? 0 : 1;
            shi[0] = all_visited ? 1 : 0;
          }
          if (kClock) {  // cvrptw/env.py:91-95: only nodes whose window is still open on arrival (the depot too)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int j = lane; j < N; j += 64)
              if (!(now + dist(cur, j) <= twin[2 * j + 1])) mk[j] = 0;
          }
        }
      }
      __syncthreads();  // B7: mask, done flag
      done = shi[0] != 0;
    }
    // d ctx_first: one row per trajectory (tsp: every step after the first reads h[first])
    if (ENV == RL4CO_ENV_TSP && w == 0 && rg == 0) {
      float* row = a.d_ctx_first + ((int64_t)inst * N + first) * kD + e0;
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(row + e, dqf[e]);
    }
  }

  // ---- write the instance's gradients ----------------------------------------------------------
  __syncthreads();
  float* dk = a.d_kvl + (int64_t)inst * N * kD;
  const int64_t plane = (int64_t)a.B_inst * N * kD;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int j = 16 * i + 4 * w + rg;
    if (j < N) {
      float* r0 = dk + (int64_t)j * kD + e0;
      *reinterpret_cast<float4*>(r0) = make_float4(dkg[i][0], dkg[i][1], dkg[i][2], dkg[i][3]);
      *reinterpret_cast<float4*>(r0 + 4) = make_float4(dkg[i][4], dkg[i][5], dkg[i][6], dkg[i][7]);
      *reinterpret_cast<float4*>(r0 + plane) = make_float4(dvv[i][0], dvv[i][1], dvv[i][2], dvv[i][3]);
      *reinterpret_cast<float4*>(r0 + plane + 4) = make_float4(dvv[i][4], dvv[i][5], dvv[i][6], dvv[i][7]);
      *reinterpret_cast<float4*>(r0 + 2 * plane) = make_float4(dkl[i][0], dkl[i][1], dkl[i][2], dkl[i][3]);
      *reinterpret_cast<float4*>(r0 + 2 * plane + 4) = make_float4(dkl[i][4], dkl[i][5], dkl[i][6], dkl[i][7]);
    }
  }
  float* dcc = a.d_ctx_cur + (int64_t)inst * N * kD;
  for (int idx = tid; idx < N * kD; idx += 64 * kW) dcc[idx] = dctx[idx];
  if (w == 0 && rg == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (a.d_q_bias) a.d_q_bias[(int64_t)inst * kD + e0 + e] = dqb[e];
      if (ENV == RL4CO_ENV_TSP) atomicAdd(a.d_q_step0 + e0 + e, dqs0[e]);
      else if (kScalar) atomicAdd(a.d_w_cap + e0 + e, dwc[e]);
      if (kClock) atomicAdd(a.d_w_time + e0 + e, dwt[e]);
    }
  }
  if (errbits && lane == 0) atomicOr(a.err, (int)errbits);
}

template <int ENV, int ROWS>
int launch_teacher(const rl4co_am_teacher_args& a, hipStream_t stream) {
  const int nw = (a.N + 3) & ~3;
  const int lds = (3 * kW * kH + kW * 2 + 8 + 8) * 4 + 3 * kW * kD * 4 + a.N * kD * 4 + 3 * nw;
  if (lds > 64 * 1024) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_teacher_kernel<ENV, ROWS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  hipLaunchKernelGGL((am_teacher_kernel<ENV, ROWS>), dim3(a.B_inst), dim3(64 * kW), lds, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

template <int ENV>
int dispatch_rows(const rl4co_am_teacher_args& a, hipStream_t s) {
  const int rows = (a.N + 15) / 16;
  if (rows <= 2) return launch_teacher<ENV, 2>(a, s);
  if (rows <= 4) return launch_teacher<ENV, 4>(a, s);
  if (rows <= 7) return launch_teacher<ENV, 7>(a, s);
  return launch_teacher<ENV, 8>(a, s);
}

}  // namespace

extern "C" int rl4co_am_teacher_max_nodes(void) { return 128; }

static int validate_teacher(const rl4co_am_teacher_args& a) {
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_OP ||
                a.env == RL4CO_ENV_PCTSP || a.env == RL4CO_ENV_PDP || a.env == RL4CO_ENV_CVRPTW);
  RL4CO_REQUIRE(a.B > 0 && a.B_inst > 0 && a.B % a.B_inst == 0);
  RL4CO_REQUIRE(a.N >= 2 && a.N <= 128 && a.T >= 1 && a.t0 >= 0 && a.t0 <= 1);
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == RL4CO_DT_BF16 || a.cache_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(a.variant >= RL4CO_TEACHER_AUTO && a.variant <= RL4CO_TEACHER_MMA);
  RL4CO_REQUIRE(a.glimpse_key && a.glimpse_val && a.logit_key && a.ctx_cur && a.actions && a.grad_logp);
  RL4CO_REQUIRE(a.kvl_row_stride >= kD && a.kvl_batch_stride >= (int64_t)a.N * kD);
  RL4CO_REQUIRE(a.temperature > 0.0f);
  RL4CO_REQUIRE((a.d_kvl || a.d_planes_bf16) && a.d_ctx_cur && a.err);
  if (a.d_planes_bf16)  // 8-byte stores of four bf16 dims
    RL4CO_REQUIRE(a.d_planes_row_stride >= kD && a.d_planes_row_stride % 4 == 0 && a.d_planes_batch_stride % 4 == 0 &&
                  a.d_planes_plane_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.d_planes_bf16) & 7) == 0);
  if (a.env == RL4CO_ENV_TSP) {
    RL4CO_REQUIRE(a.ctx_first && a.q_step0 && a.d_ctx_first && a.d_q_step0);
  } else if (a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_PCTSP) {  // PCTSP: real prize [B_inst,N], prize_required
    RL4CO_REQUIRE(a.w_cap && a.demand && a.vehicle_capacity && a.d_w_cap);
  } else if (a.env == RL4CO_ENV_CVRPTW) {
    RL4CO_REQUIRE(a.w_cap && a.demand && a.vehicle_capacity && a.d_w_cap);
    RL4CO_REQUIRE(a.locs && a.time_windows && a.durations && a.w_time && a.d_w_time);
  } else if (a.env == RL4CO_ENV_OP) {
    RL4CO_REQUIRE(a.w_cap && a.locs && a.max_length && a.d_w_cap);
  } else {
    RL4CO_REQUIRE((a.N - 1) % 2 == 0);  // PDP: pickups 1..n/2, deliveries n/2+1..n; no instance data
  }
  RL4CO_REQUIRE(a.q_bias == nullptr || a.d_q_bias != nullptr);
  // (r06) context tables in the planes' 16-bit type with their own strides, gradients into planes 3 / 4: MMA variant only
  RL4CO_REQUIRE(a.ctx_dtype == RL4CO_DT_F32 || (a.ctx_dtype == a.cache_dtype && a.cache_dtype != RL4CO_DT_F32));
  RL4CO_REQUIRE(a.ctx_row_stride == 0 || (a.ctx_row_stride >= kD && a.ctx_row_stride % 4 == 0));
  RL4CO_REQUIRE(a.ctx_batch_stride == 0 || (a.ctx_batch_stride >= (int64_t)a.N * kD && a.ctx_batch_stride % 4 == 0));
  RL4CO_REQUIRE(a.d_ctx_in_planes == 0 || (a.d_ctx_in_planes == 1 && a.d_planes_bf16 != nullptr));
  return RL4CO_OK;
}

// MMA needs bf16 planes (16-byte aligned rows) and its step tables to hold every action column
static int resolve_teacher_variant(const rl4co_am_teacher_args& a) {
  const bool mma_ok = a.cache_dtype != RL4CO_DT_F32 && a.N <= rl4co::teacher_mma_max_nodes() &&
                      a.T <= rl4co::teacher_mma_max_steps() && a.kvl_row_stride % 8 == 0 && a.kvl_batch_stride % 8 == 0;
  if (a.ctx_dtype != RL4CO_DT_F32 || a.ctx_row_stride || a.ctx_batch_stride)  // 16-bit / strided context tables: MMA variant only
    return (mma_ok && a.variant != RL4CO_TEACHER_REPLAY) ? RL4CO_TEACHER_MMA : -1;
  if (a.d_planes_bf16 || !a.d_kvl)  // bf16 plane gradients come out of the MMA variant only
    return (mma_ok && a.variant != RL4CO_TEACHER_REPLAY) ? RL4CO_TEACHER_MMA : -1;
  if (a.variant == RL4CO_TEACHER_MMA) return mma_ok ? RL4CO_TEACHER_MMA : -1;
  if (a.variant == RL4CO_TEACHER_REPLAY) return RL4CO_TEACHER_REPLAY;
  return mma_ok ? RL4CO_TEACHER_MMA : RL4CO_TEACHER_REPLAY;
}

extern "C" int rl4co_am_teacher_variant(const rl4co_am_teacher_args* args) {
  if (args == nullptr || validate_teacher(*args) != RL4CO_OK) return -1;
  return resolve_teacher_variant(*args);
}

extern "C" int rl4co_am_teacher_backward(const rl4co_am_teacher_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_teacher_args& a = *args;
  const int st = validate_teacher(a);
  if (st != RL4CO_OK) return st;
  const int variant = resolve_teacher_variant(a);
  RL4CO_REQUIRE(variant > 0);  // RL4CO_TEACHER_MMA requested for planes / sizes it does not support
  hipStream_t s = rl4co::as_stream(stream);
  if (variant == RL4CO_TEACHER_MMA)
    return a.cache_dtype == RL4CO_DT_F16 ? rl4co::launch_teacher_mma_f16(a, s) : rl4co::launch_teacher_mma(a, s);
  switch (a.env) {
    case RL4CO_ENV_TSP: return dispatch_rows<RL4CO_ENV_TSP>(a, s);
    case RL4CO_ENV_CVRP: return dispatch_rows<RL4CO_ENV_CVRP>(a, s);
    case RL4CO_ENV_OP: return dispatch_rows<RL4CO_ENV_OP>(a, s);
    case RL4CO_ENV_PCTSP: return dispatch_rows<RL4CO_ENV_PCTSP>(a, s);
    case RL4CO_ENV_PDP: return dispatch_rows<RL4CO_ENV_PDP>(a, s);
    default: return dispatch_rows<RL4CO_ENV_CVRPTW>(a, s);
  }
}
