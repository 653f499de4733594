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
// One 256- or 512-thread workgroup per INSTANCE; its S trajectories (multistart) are replayed one after
// the other. Every lane owns a fixed set of cache elements — rows j = 4 NW i + 4 w + rg (i < ROWS),
// dims 8 li .. 8 li + 7 — of all three planes: they are loaded into registers ONCE per instance
// and their gradients are accumulated in registers over all S x T steps (no atomics, no plane
// traffic inside the loop; N <= 4 NW ROWS <= 112). Per step the waves meet only three times:
// each exchange ships wave-local partial sums built so that the missing global scalar (softmax
// max / log-sum-exp / sum a*da) can be applied AFTER the exchange (online-softmax style), and the
// next step's feasibility mask is prepared by an otherwise idle wave into a second buffer.
// fp32 arithmetic throughout; this is a floating-point kernel tested against torch autograd
// (tests/test_gpu_teacher.py, tolerance stated there), not part of the bit-exact decode contract.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kH = RL4CO_NUM_HEADS;
constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kSqrtD = 11.3137084989847604f;

__device__ inline float load_plane(const void* base, int dtype, int64_t idx) {
  if (dtype == RL4CO_DT_BF16) return __uint_as_float((uint32_t)static_cast<const uint16_t*>(base)[idx] << 16);
  return static_cast<const float*>(base)[idx];
}

// sum over the 4 row groups of a wave (lanes differing in bits 4,5), result on every lane
__device__ inline float rg_sum(float v) { return rl4co::bfly_sum<16, 64>(v); }
__device__ inline float rg_max(float v) { return rl4co::bfly_max<16, 64>(v); }

// NW waves per workgroup; row j is owned by wave (j / 4) % NW, row group j % 4, slot j / (4 NW)
template <int ENV, int ROWS, int NW>
__global__ void __launch_bounds__(64 * NW) am_teacher_kernel(const rl4co_am_teacher_args a) {
  constexpr int kW = NW;
  constexpr int kRowStep = 4 * NW;
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
  uint8_t* mk = reinterpret_cast<uint8_t*>(dctx + N * kD);  // [2][nw] double-buffered feasibility mask
  uint8_t* vis = mk + 2 * nw;                               // [nw]

  // ---- this lane's slice of the three planes, resident in registers -------------------------
  float kg[ROWS][8], vv[ROWS][8], kl[ROWS][8];
  float dkg[ROWS][8], dvv[ROWS][8], dkl[ROWS][8];
  const int64_t pbase = (int64_t)inst * a.kvl_batch_stride;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int j = kRowStep * i + 4 * w + rg;
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

  const float* ctxc = a.ctx_cur + (int64_t)inst * N * kD + e0;
  const float* ctxf = (ENV == RL4CO_ENV_TSP) ? a.ctx_first + (int64_t)inst * N * kD + e0 : nullptr;
  const float* dem = (ENV == RL4CO_ENV_CVRP) ? a.demand + (int64_t)inst * (N - 1) : nullptr;
  const float cap = (ENV == RL4CO_ENV_CVRP) ? a.vehicle_capacity[inst] : 0.0f;
  float qb[8], dqb[8], dqs0[8], dwc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    qb[e] = a.q_bias ? a.q_bias[(int64_t)inst * kD + e0 + e] : 0.0f;
    dqb[e] = 0.0f;
    dqs0[e] = 0.0f;
    dwc[e] = 0.0f;
  }
  const float inv_temp = 1.0f / a.temperature;
  uint32_t errbits = 0;

  for (int s = 0; s < S; ++s) {
    const int r = s * a.B_inst + inst;
    const int64_t* act = a.actions + (int64_t)r * T;
    const float* gl = a.grad_logp + (int64_t)r * T;
    // ---- reset state (tsp/env.py:88-113, cvrp/env.py:98-136) ---------------------------------
    __syncthreads();
    for (int j = tid; j < nw; j += 64 * kW) {
      mk[j] = (j < N) ? ((ENV == RL4CO_ENV_CVRP && j == 0) ? 0 : 1) : 0;  // fresh CVRP: depot masked
      vis[j] = (j < N) ? 0 : 1;
    }
    if (tid == 0) shi[0] = 0;
    int cur = 0, first = 0;
    long long step_i = 0;
    float used = 0.0f;
    bool done = false;
    float dqf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dqf[e] = 0.0f;
    __syncthreads();

    // Everything a step reads from global memory is known one step ahead (teacher forcing): the
    // action, its upstream gradient and the context row of the node just visited are requested
    // during the previous step, so no load latency sits on the step's critical path.
    long long at_next = act[0];
    float g_next = gl[0];
    float crow[8], frow[8], wcap[8];  // ctx_cur[cur], ctx_first[first] (TSP), w_cap (CVRP)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      crow[e] = (ENV == RL4CO_ENV_CVRP) ? ctxc[e] : 0.0f;  // CVRP starts at the depot (node 0)
      frow[e] = 0.0f;
      wcap[e] = (ENV == RL4CO_ENV_CVRP) ? a.w_cap[e0 + e] : 0.0f;
    }
    for (int t = 0; t < T && !done; ++t) {
      int at = (int)at_next;
      const float g = g_next;
      if (t + 1 < T) {
        at_next = act[t + 1];
        g_next = gl[t + 1];
      }
      if (at < 0 || at >= N) {
        errbits |= RL4CO_EBIT_INFEASIBLE;
        at = 0;
      }
      float crow_next[8];  // context row of the NEXT step (its current node is this step's action)
#pragma unroll
      for (int e = 0; e < 8; ++e) crow_next[e] = ctxc[(int64_t)at * kD + e];
      if (ENV == RL4CO_ENV_TSP && step_i == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) frow[e] = ctxf[(int64_t)at * kD + e];  // first node, fixed from now on
      }
      // masks are double-buffered: this step reads mkb[t & 1]; wave 3 prepares the mask and the
      // done flag of step t + 1 into the other buffer while the step is being computed (the
      // transition only needs the given action), so no barrier is spent on the environment.
      const uint8_t* mkc = mk + (t & 1) * nw;
      uint8_t* mkn = mk + ((t + 1) & 1) * nw;
      const float rem = cap - used;   // context of THIS step (before the transition)
      const int cur_t = cur, first_t = first;
      const long long step_t = step_i;
      if (mkc[at] == 0 && t >= a.t0) errbits |= RL4CO_EBIT_INFEASIBLE;
      // ---- environment transition with the given action (scalars on every wave) --------------
      if (ENV == RL4CO_ENV_TSP) {
        if (step_i == 0) first = at;
        cur = at;
        step_i += 1;
        if (w == kW - 1) {
          bool any_left = false;
          for (int j = lane; j < nw; j += 64) {
            const uint8_t v = (j == at) ? (uint8_t)0 : mkc[j];
            mkn[j] = v;
            any_left |= v != 0;
          }
          any_left = __any(any_left);
          if (lane == 0) shi[(t + 1) & 1] = any_left ? 0 : 1;
        }
      } else {
        const int di = min(max(at - 1, 0), N - 2);
        used = (used + dem[di]) * (at != 0 ? 1.0f : 0.0f);
        cur = at;
        if (w == kW - 1) {
          const float thr = cap + 1e-5f;
          bool any_feasible = false, all_visited = true;
          for (int j = lane; j < nw; j += 64) {
            const bool v = j < N && (vis[j] != 0 || j == at);
            if (j < N && j == at) vis[j] = 1;
            all_visited &= (j >= N) || v;
            if (j >= 1 && j < N) {
              const bool masked = v || (dem[j - 1] + used > thr);
              mkn[j] = masked ? 0 : 1;
              any_feasible |= !masked;
            } else if (j >= N) {
              mkn[j] = 0;
            }
          }
          any_feasible = __any(any_feasible);
          all_visited = __all(all_visited);
          if (lane == 0) {
            mkn[0] = ((at == 0) && any_feasible) ? 0 : 1;
            shi[(t + 1) & 1] = all_visited ? 1 : 0;
          }
        }
      }
      const bool decoded = t >= a.t0;  // multistart: column 0 is imposed, log-prob 0 (decoding.py:306-326)
      if (decoded) {
        // ---- query -----------------------------------------------------------------------------
        float q[8];
        if (ENV == RL4CO_ENV_TSP) {
          if (step_t < 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = a.q_step0[e0 + e] + qb[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = (frow[e] + crow[e]) + qb[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) q[e] = fmaf(wcap[e], rem, crow[e]) + qb[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] *= 0.25f;

        // ---- forward: scores, wave-local softmax, ONE exchange (online-softmax merge) -----------
        float sv[ROWS];
        bool feas[ROWS];
        float mw = kNegInf;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const int j = kRowStep * i + 4 * w + rg;
          float acc = 0.0f;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = fmaf(q[e], kg[i][e], acc);
          acc += rl4co::bfly_f<1>(acc);
          feas[i] = j < N && mkc[j < N ? j : 0] != 0;
          const bool in_glimpse = j < N && (!a.mask_inner || feas[i]);
          sv[i] = in_glimpse ? acc : kNegInf;
          mw = fmaxf(mw, sv[i]);
        }
        mw = rg_max(mw);
        const float mw_safe = (mw > kNegInf) ? mw : 0.0f;  // a wave whose rows are all masked
        float l = 0.0f, o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.0f;
        float p[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          p[i] = __expf(sv[i] - mw_safe);
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
          if ((li & 1) == 0) {
            lpart[w * kH + hd] = l;
            mpart[w * kH + hd] = mw;
          }
        }
        __syncthreads();  // exchange 1: (max, sum, weighted values) of the four waves
        float m = kNegInf;
#pragma unroll
        for (int ww = 0; ww < kW; ++ww) m = fmaxf(m, mpart[ww * kH + hd]);
        float scl[kW];
        l = 0.0f;
#pragma unroll
        for (int ww = 0; ww < kW; ++ww) {
          const float mm = mpart[ww * kH + hd];
          scl[ww] = (mm > kNegInf) ? __expf(mm - m) : 0.0f;
          l = fmaf(lpart[ww * kH + hd], scl[ww], l);
        }
        const float inv_l = 1.0f / l;
        float heads[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = e0 + e;
          float acc = 0.0f;
#pragma unroll
          for (int ww = 0; ww < kW; ++ww) acc = fmaf(opart[ww * kD + d], scl[ww], acc);
          heads[e] = acc * inv_l;
        }
        const float my_scale = scl[w] * inv_l;  // p -> attention probability for this wave's rows
        // ---- forward: logits, clip; backward terms that do not need lse yet -------------------------
        float z[ROWS], dzdu[ROWS];
        float zm = kNegInf;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const int j = kRowStep * i + 4 * w + rg;
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
        const float zm_safe = (zm > kNegInf) ? zm : 0.0f;
        // d heads = g/sqrt(d) * [ dzdu_a * kl_a  -  sum_i softmax_i * dzdu_i * kl_i ]; the softmax
        // needs lse, but exp(z_i - zm_w) does not: ship the wave-local sums with (zm_w, se_w)
        float se = 0.0f, wsum[8], hot[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          wsum[e] = 0.0f;
          hot[e] = 0.0f;
        }
        float ez[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const int j = kRowStep * i + 4 * w + rg;
          ez[i] = __expf(z[i] - zm_safe);  // 0 for masked rows
          se += ez[i];
          const float c1 = ez[i] * dzdu[i];
          const float c2 = (j == at && z[i] > kNegInf) ? dzdu[i] : 0.0f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            wsum[e] = fmaf(c1, kl[i][e], wsum[e]);
            hot[e] = fmaf(c2, kl[i][e], hot[e]);
          }
        }
        se = rg_sum(se);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          wsum[e] = rg_sum(wsum[e]);
          hot[e] = rg_sum(hot[e]);
        }
        if (rg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            dhpart[w * kD + e0 + e] = wsum[e];
            dqpart[w * kD + e0 + e] = hot[e];  // dqpart doubles as the one-hot term buffer here
          }
        }
        if (lane == 0) {
          lsep[w * 2] = zm;
          lsep[w * 2 + 1] = se;
        }
        __syncthreads();  // exchange 2: log-sum-exp pieces and the d-heads partial sums
        float zmax = kNegInf;
#pragma unroll
        for (int ww = 0; ww < kW; ++ww) zmax = fmaxf(zmax, lsep[ww * 2]);
        float zs[kW];
        float tot = 0.0f;
#pragma unroll
        for (int ww = 0; ww < kW; ++ww) {
          zs[ww] = (lsep[ww * 2] > kNegInf) ? __expf(lsep[ww * 2] - zmax) : 0.0f;
          tot = fmaf(lsep[ww * 2 + 1], zs[ww], tot);
        }
        const float lse = zmax + __logf(tot);
        const float inv_tot = 1.0f / tot;
        float dh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = e0 + e;
          float soft = 0.0f, oh = 0.0f;
#pragma unroll
          for (int ww = 0; ww < kW; ++ww) {
            soft = fmaf(dhpart[ww * kD + d], zs[ww], soft);
            oh += dqpart[ww * kD + d];
          }
          dh[e] = (g / kSqrtD) * (oh - soft * inv_tot);
        }
        // per-row d(heads . kl_row) now that lse is known; logit-key gradient; log p(a_t)
        const float my_soft = zs[w] * inv_tot;  // exp(z - zm_w) -> softmax probability for this wave's rows
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const int j = kRowStep * i + 4 * w + rg;
          const float prob = ez[i] * my_soft;
          const float dz = g * ((j == at ? 1.0f : 0.0f) - prob);
          const float ddr = (z[i] > kNegInf) ? dz * dzdu[i] / kSqrtD : 0.0f;
#pragma unroll
          for (int e = 0; e < 8; ++e) dkl[i][e] = fmaf(ddr, heads[e], dkl[i][e]);
          if (j == at && li == 0) {
            const float lp = z[i] - lse;
            if (a.logp_out) a.logp_out[(int64_t)r * T + t] = lp;
            if (!(lp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;
          }
        }
        // ---- backward: glimpse attention. ds_i = a_i (da_i - ada); d q = sum ds_i kg_i =
        //      sum a_i da_i kg_i - ada * sum a_i kg_i  -> both sums and ada in ONE exchange --------
        float ada = 0.0f, xs[8], ys[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xs[e] = 0.0f;
          ys[e] = 0.0f;
        }
        float av[ROWS], da[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          av[i] = p[i] * my_scale;
          float acc = 0.0f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            acc = fmaf(dh[e], vv[i][e], acc);
            dvv[i][e] = fmaf(av[i], dh[e], dvv[i][e]);
          }
          acc += rl4co::bfly_f<1>(acc);  // the head's 16 dims live on a lane pair
          da[i] = acc;
          const float ad = av[i] * acc;
          ada += ad;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xs[e] = fmaf(ad, kg[i][e], xs[e]);
            ys[e] = fmaf(av[i], kg[i][e], ys[e]);
          }
        }
        ada = rg_sum(ada);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xs[e] = rg_sum(xs[e]);
          ys[e] = rg_sum(ys[e]);
        }
        __syncthreads();  // exchange-2 buffers (dhpart, dqpart) fully consumed before they are reused
        if (rg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            dhpart[w * kD + e0 + e] = xs[e];
            opart[w * kD + e0 + e] = ys[e];
          }
          if ((li & 1) == 0) adap[w * kH + hd] = ada;
        }
        __syncthreads();  // exchange 3: attention-backward sums
        ada = 0.0f;
#pragma unroll
        for (int ww = 0; ww < kW; ++ww) ada += adap[ww * kH + hd];
        float dqs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = e0 + e;
          float X = 0.0f, Y = 0.0f;
#pragma unroll
          for (int ww = 0; ww < kW; ++ww) {
            X += dhpart[ww * kD + d];
            Y += opart[ww * kD + d];
          }
          dqs[e] = X - ada * Y;  // d L / d q_scaled
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          const float ds = av[i] * (da[i] - ada);
#pragma unroll
          for (int e = 0; e < 8; ++e) dkg[i][e] = fmaf(ds, q[e], dkg[i][e]);
        }
        // ---- backward: query -> context rows, graph context, placeholder / capacity column ------------
        if (w == 0 && rg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int d = e0 + e;
            const float dqr = 0.25f * dqs[e];
            dqb[e] += dqr;
            if (ENV == RL4CO_ENV_TSP) {
              if (step_t < 1) {
                dqs0[e] += dqr;
              } else {
                dqf[e] += dqr;
                dctx[cur_t * kD + d] += dqr;
              }
            } else {
              dwc[e] = fmaf(dqr, rem, dwc[e]);
              dctx[cur_t * kD + d] += dqr;
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) crow[e] = crow_next[e];
      __syncthreads();  // end of step: next mask / done flag visible, exchange buffers free
      done = shi[(t + 1) & 1] != 0;
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
    const int j = kRowStep * i + 4 * w + rg;
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
      else atomicAdd(a.d_w_cap + e0 + e, dwc[e]);
    }
  }
  if (errbits && lane == 0) atomicOr(a.err, (int)errbits);
}

template <int ENV, int ROWS, int NW>
int launch_teacher(const rl4co_am_teacher_args& a, hipStream_t stream) {
  const int nw = (a.N + 3) & ~3;
  const int lds = (3 * NW * kH + NW * 2 + 8 + 8) * 4 + 3 * NW * kD * 4 + a.N * kD * 4 + 3 * nw;
  if (lds > 64 * 1024) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_teacher_kernel<ENV, ROWS, NW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  hipLaunchKernelGGL((am_teacher_kernel<ENV, ROWS, NW>), dim3(a.B_inst), dim3(64 * NW), lds, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// Measured on MI355X: eight waves per workgroup (2 per SIMD, 256-register cap) spill the step's
// working set to scratch and run 5x slower than four waves with the full 512-register budget, so
// the row ownership stays at 4 waves: 2 / 4 / 7 rows per lane for N <= 32 / 64 / 112.
template <int ENV>
int dispatch_rows(const rl4co_am_teacher_args& a, hipStream_t s) {
  if (a.N <= 32) return launch_teacher<ENV, 2, 4>(a, s);
  if (a.N <= 64) return launch_teacher<ENV, 4, 4>(a, s);
  return launch_teacher<ENV, 7, 4>(a, s);
}

}  // namespace

extern "C" int rl4co_am_teacher_max_nodes(void) { return 112; }

extern "C" int rl4co_am_teacher_backward(const rl4co_am_teacher_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_teacher_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP);
  RL4CO_REQUIRE(a.B > 0 && a.B_inst > 0 && a.B % a.B_inst == 0);
  RL4CO_REQUIRE(a.N >= 2 && a.N <= 112 && a.T >= 1 && a.t0 >= 0 && a.t0 <= 1);
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == RL4CO_DT_BF16);
  RL4CO_REQUIRE(a.glimpse_key && a.glimpse_val && a.logit_key && a.ctx_cur && a.actions && a.grad_logp);
  RL4CO_REQUIRE(a.kvl_row_stride >= kD && a.kvl_batch_stride >= (int64_t)a.N * kD);
  RL4CO_REQUIRE(a.temperature > 0.0f);
  RL4CO_REQUIRE(a.d_kvl && a.d_ctx_cur && a.err);
  if (a.env == RL4CO_ENV_TSP) {
    RL4CO_REQUIRE(a.ctx_first && a.q_step0 && a.d_ctx_first && a.d_q_step0);
  } else {
    RL4CO_REQUIRE(a.w_cap && a.demand && a.vehicle_capacity && a.d_w_cap);
  }
  RL4CO_REQUIRE(a.q_bias == nullptr || a.d_q_bias != nullptr);
  hipStream_t s = rl4co::as_stream(stream);
  return a.env == RL4CO_ENV_TSP ? dispatch_rows<RL4CO_ENV_TSP>(a, s) : dispatch_rows<RL4CO_ENV_CVRP>(a, s);
}
