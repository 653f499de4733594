// am_decode_ms.hip — multistart (POMO) decode on the matrix cores.
//
// With multistart decoding (utils/decoding.py:282-330, zoo/pomo/model.py:88-143) the S
// trajectories of one instance read the SAME decoder cache and differ only in their query and
// mask: a decode step for all of them is three small GEMMs — scores = K_g . Q^T, glimpse =
// V^T . P^T, logits = K_l' . heads^T — with the trajectories as the N dimension. The streaming
// kernel (am_decode.hip) runs one wave per trajectory and is VALU-bound there (275 M
// trajectory-steps/s at TSP-100 x 4096 x 8); here one workgroup owns one instance, keeps its three
// bf16 planes resident in LDS for the whole rollout and advances up to 32 trajectories per
// v_mfma_f32_32x32x16_bf16 column tile.
//
// Layout (same accumulator-layout trick as am_encoder.hip): every product is computed "transposed"
// so that the MFMA column index — lane & 31 — is the TRAJECTORY: the softmax over keys, the
// log-softmax and the argmax over nodes are then in-lane reductions over accumulator registers
// plus one cross-half exchange, and each lane carries its trajectory's state (feasibility bit
// mask, current / first node, load) in registers. Wave w owns heads (2w, 2w+1) for the glimpse and
// key tile w for the logits; the waves meet three times per step (query rows, glimpse rows,
// log-sum-exp / argmax pieces).
//
// Numerics: bf16 MFMA inputs (planes, query, glimpse), fp32 accumulation and fp32 softmax / tanh /
// log-softmax — the reference's mixed-precision regime. This variant is NOT part of the bit-exact
// contract of am_decode.hip (a bf16 query cannot reproduce the fp32 specified order); it is tested
// by tolerance against the streaming kernel on the same planes (tests/test_gpu_decode_ms.py).
#include <hip/hip_runtime.h>

#include "common.h"
#include "rl4co_math.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kH = RL4CO_NUM_HEADS;
constexpr int kRS = kD + 8;  // LDS row stride in bf16 elements
constexpr int kThreads = 256;
constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kSqrtD = 11.3137084989847604f;
constexpr float kLog2e = 1.44269504088896341f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ inline int rowmap(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ inline f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}
__device__ inline bf16x8 frag_from_acc(const f32x16& c, int u) {
  bf16x8 f;
#pragma unroll
  for (int s = 0; s < 8; ++s) f[s] = (__bf16)c[8 * u + s];
  return f;
}
__device__ inline bf16x8 lds_frag(const __bf16* base, int row, int col) {
  return *reinterpret_cast<const bf16x8*>(base + row * kRS + col);
}
__device__ inline float bf16_bits_to_float(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

// 128-bit node sets as four words that are only ever indexed with compile-time constants: a
// runtime index would send the array to scratch memory (a global-memory round trip per access)
struct Bits128 {
  uint32_t w[4];
  __device__ inline uint32_t word(int k) const {  // k runtime, wave-uniform or not
    return k == 0 ? w[0] : (k == 1 ? w[1] : (k == 2 ? w[2] : w[3]));
  }
  __device__ inline bool test(int j) const { return (word(j >> 5) >> (j & 31)) & 1u; }
  __device__ inline void set(int j, bool on) {
    const uint32_t bit = 1u << (j & 31);
    const int k = j >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t m = (i == k) ? bit : 0u;
      w[i] = on ? (w[i] | m) : (w[i] & ~m);
    }
  }
};

struct Xchg {  // per (wave, trajectory) pieces of the log-softmax / selection over nodes
  float zmax, se, best_key, best_z;
  int best_idx;
  float forced_z;
};

template <int ENV>
__global__ void __launch_bounds__(kThreads) am_decode_ms_kernel(const rl4co_am_decode_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __bf16* kgs = reinterpret_cast<__bf16*>(smem);  // [128 keys][kRS]   glimpse keys, natural
  __bf16* vts = kgs + 128 * kRS;                  // [128 dims][kRS]   glimpse values TRANSPOSED, keys in
                                                  //                   accumulator order inside each 16-group
  __bf16* kls = vts + 128 * kRS;                  // [128 keys][kRS]   logit keys, natural
  __bf16* qs = kls + 128 * kRS;                   // [32 traj][kRS]    queries of this step
  __bf16* hs = qs + 32 * kRS;                     // [32 traj][kRS]    glimpses of this step
  Xchg* xs = reinterpret_cast<Xchg*>(hs + 32 * kRS);  // [4 waves][32 traj]
  float* dems = reinterpret_cast<float*>(xs + 4 * 32);  // [128] CVRP demands (index j-1 at j), 0 elsewhere

  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int inst = blockIdx.x;
  const int N = a.N;
  const int S = a.B / a.B_inst;

  // ---- planes HBM -> LDS, once per instance -----------------------------------------------------
  {
    const uint16_t* gk = static_cast<const uint16_t*>(a.glimpse_key) + (int64_t)inst * a.kvl_batch_stride;
    const uint16_t* gv = static_cast<const uint16_t*>(a.glimpse_val) + (int64_t)inst * a.kvl_batch_stride;
    const uint16_t* gl = static_cast<const uint16_t*>(a.logit_key) + (int64_t)inst * a.kvl_batch_stride;
    for (int c = tid; c < 128 * 16; c += kThreads) {  // 16-byte chunks: row = c / 16, col = (c % 16) * 8
      const int row = c >> 4, col = (c & 15) * 8;
      uint4 k4 = make_uint4(0, 0, 0, 0), l4 = k4, v4 = k4;
      if (row < N) {
        k4 = *reinterpret_cast<const uint4*>(gk + (int64_t)row * a.kvl_row_stride + col);
        l4 = *reinterpret_cast<const uint4*>(gl + (int64_t)row * a.kvl_row_stride + col);
        v4 = *reinterpret_cast<const uint4*>(gv + (int64_t)row * a.kvl_row_stride + col);
      }
      *reinterpret_cast<uint4*>(kgs + row * kRS + col) = k4;
      *reinterpret_cast<uint4*>(kls + row * kRS + col) = l4;
      // V^T with the key index stored at the position the accumulator layout expects:
      // position 16 g + 8 h + s  <-  key 16 g + (s & 3) + 8 (s >> 2) + 4 h
      const int g = row >> 4, kk = row & 15;  // kk = (s & 3) + 8 (s >> 2) + 4 h
      const int h = (kk >> 2) & 1, s = (kk & 3) + 4 * (kk >> 3);
      const int pos = 16 * g + 8 * h + s;
      const uint16_t* v16 = reinterpret_cast<const uint16_t*>(&v4);
#pragma unroll
      for (int e = 0; e < 8; ++e) reinterpret_cast<uint16_t*>(vts)[(col + e) * kRS + pos] = v16[e];
    }
    for (int j = tid; j < 128; j += kThreads)
      dems[j] = (ENV == RL4CO_ENV_CVRP && j >= 1 && j < N) ? a.demand[(int64_t)inst * (N - 1) + j - 1] : 0.0f;
  }
  const float cap = (ENV == RL4CO_ENV_CVRP) ? a.vehicle_capacity[inst] : 0.0f;
  const float thr = cap + 1e-5f;
  const float* ctxc = a.ctx_cur + (int64_t)inst * N * kD;
  const float* ctxf = (ENV == RL4CO_ENV_TSP) ? a.ctx_first + (int64_t)inst * N * kD : nullptr;
  const bool single = a.max_steps == 1;
  const float inv_temp = 1.0f / a.temperature;
  const float clip_over_temp = a.tanh_clipping * inv_temp;
  uint32_t errbits = 0;
  uint32_t nv[4];  // nodes that exist (j < N), per 32-node word
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
    nv[kt] = (N >= 32 * (kt + 1)) ? 0xffffffffu : (N > 32 * kt ? ((1u << (N - 32 * kt)) - 1u) : 0u);

  for (int s0 = 0; s0 < S; s0 += 32) {  // column tiles of 32 trajectories
    // ---- per-lane trajectory state (lane l31 <-> trajectory s0 + l31; replicated in both halves
    //      and in all four waves) and the same for the query-building threads (trajectory tid / 8)
    const int sl = s0 + l31;
    const bool lane_ok = sl < S;
    const int r = (lane_ok ? sl : s0) * a.B_inst + inst;
    Bits128 mw, vw;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      uint32_t m = 0, v = 0;
      const uint8_t* gm = a.action_mask + (int64_t)r * N + 32 * kt;
      const uint8_t* gv = (ENV == RL4CO_ENV_CVRP) ? a.visited + (int64_t)r * N + 32 * kt : nullptr;
      for (int b = 0; b < 32 && 32 * kt + b < N; ++b) {
        m |= (lane_ok && gm[b]) ? (1u << b) : 0u;
        if (ENV == RL4CO_ENV_CVRP) v |= (lane_ok && gv[b]) ? (1u << b) : 0u;
      }
      mw.w[kt] = m;
      vw.w[kt] = v;
    }
    int cur = (int)a.current_node[r];
    int first = (ENV == RL4CO_ENV_TSP) ? (int)a.first_node[r] : 0;
    long long step_i = (ENV == RL4CO_ENV_TSP) ? a.step_i[r] : 0;
    float used = (ENV == RL4CO_ENV_CVRP) ? a.used_capacity[r] : 0.0f;
    bool done = !lane_ok || a.done[r] != 0;
    float ent_acc = 0.0f;
    int nsteps = 0;
    // query builder: thread tid builds 16 dims of trajectory tid / 8, whose state lives on lane
    // tid / 8 of this (and every) wave
    const int qt = tid >> 3, qd0 = (tid & 7) * 16;
    __syncthreads();

    int t = 0;
    for (; t < a.max_steps; ++t) {
      if (!single && __all(done)) break;  // identical on every wave
      // ---- 1. query rows (folded context + graph context), pre-scaled by 1/sqrt(16) * log2(e) ------
      {
        const int q_cur = __shfl(cur, qt, 64), q_first = __shfl(first, qt, 64);
        const int q_step = __shfl((int)(step_i > 0 ? 1 : 0), qt, 64);
        const float q_used = __shfl(used, qt, 64);
        float qv[16];
        const float* qb = a.q_bias ? a.q_bias + (int64_t)inst * kD + qd0 : nullptr;
        if (ENV == RL4CO_ENV_TSP) {
          if (q_step < 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) qv[e] = a.q_step0[qd0 + e] + (qb ? qb[e] : 0.0f);
          } else {
            const float* f = ctxf + (int64_t)q_first * kD + qd0;
            const float* c = ctxc + (int64_t)q_cur * kD + qd0;
#pragma unroll
            for (int e = 0; e < 16; ++e) qv[e] = (f[e] + c[e]) + (qb ? qb[e] : 0.0f);
          }
        } else {
          const float rem = cap - q_used;
          const float* c = ctxc + (int64_t)q_cur * kD + qd0;
#pragma unroll
          for (int e = 0; e < 16; ++e) qv[e] = fmaf(a.w_cap[qd0 + e], rem, c[e]) + (qb ? qb[e] : 0.0f);
        }
        bf16x8 lo, up;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          lo[e] = (__bf16)(qv[e] * (0.25f * kLog2e));
          up[e] = (__bf16)(qv[8 + e] * (0.25f * kLog2e));
        }
        *reinterpret_cast<bf16x8*>(qs + qt * kRS + qd0) = lo;
        *reinterpret_cast<bf16x8*>(qs + qt * kRS + qd0 + 8) = up;
      }
      __syncthreads();  // B1

      // ---- 2. glimpse of head pair w for all trajectories -----------------------------------------------
      {
        f32x16 o = zero16();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int h = 2 * w + hh;
          const bf16x8 qf = lds_frag(qs, l31, 16 * h + 8 * hi);
          f32x16 sc[4];
          float m = kNegInf;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            sc[kt] = mfma(lds_frag(kgs, 32 * kt + l31, 16 * h + 8 * hi), qf, zero16());
            const uint32_t mbits = (a.mask_inner ? mw.w[kt] : nv[kt]) >> (4 * hi);
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const bool f = (mbits >> ((rr & 3) + 8 * (rr >> 2))) & 1u;
              sc[kt][rr] = f ? sc[kt][rr] : kNegInf;
              m = fmaxf(m, sc[kt][rr]);
            }
          }
          m = fmaxf(m, rl4co::bfly_f<32>(m));
          const float ms = (m > kNegInf) ? m : 0.0f;
          float l = 0.0f;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const float p = __builtin_amdgcn_exp2f(sc[kt][rr] - ms);
              sc[kt][rr] = p;
              l += p;
            }
          }
          l += rl4co::bfly_f<32>(l);
          f32x16 acc = zero16();
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            acc = mfma(lds_frag(vts, 32 * w + l31, 32 * kt + 8 * hi), frag_from_acc(sc[kt], 0), acc);
            acc = mfma(lds_frag(vts, 32 * w + l31, 32 * kt + 16 + 8 * hi), frag_from_acc(sc[kt], 1), acc);
          }
          const float inv = (l > 0.0f) ? __builtin_amdgcn_rcpf(l) : 0.0f;
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) o[8 * hh + rr] = acc[8 * hh + rr] * inv;
        }
        __bf16* hrow = hs + l31 * kRS + 32 * w;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          bf16x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (__bf16)o[4 * c + i];
          *reinterpret_cast<bf16x4*>(hrow + 8 * c + 4 * hi) = v;
        }
      }
      __syncthreads();  // B2

      // ---- 3. logits of key tile w for all trajectories, local log-softmax / selection pieces -----
      const int64_t tcol = (int64_t)a.t0 + t;
      {
        f32x16 u = zero16();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          u = mfma(lds_frag(kls, 32 * w + l31, 16 * ks + 8 * hi), lds_frag(hs, l31, 16 * ks + 8 * hi), u);
        const uint32_t lbits = (a.mask_logits ? mw.word(w) : (w == 0 ? nv[0] : (w == 1 ? nv[1] : (w == 2 ? nv[2] : nv[3])))) >> (4 * hi);
        const int forced = (a.mode == RL4CO_DECODE_EVALUATE && lane_ok)
                               ? (int)a.forced_actions[(int64_t)r * a.out_stride + tcol] : -1;
        float z[16];
        float zmax = kNegInf;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          // this variant is tolerance-tested, not bit-exact: hardware reciprocals instead of IEEE division
          const float uu = u[rr] * (1.0f / kSqrtD);
          if (uu != uu && lane_ok && !done) errbits |= RL4CO_EBIT_NAN_LOGIT;
          float zz = uu;
          if (a.tanh_clipping > 0.0f) {
            const float ex = __expf(-2.0f * fabsf(uu));
            zz = copysignf((1.0f - ex) * __builtin_amdgcn_rcpf(1.0f + ex), uu) * clip_over_temp;
          } else {
            zz = zz * inv_temp;
          }
          const bool f = (lbits >> ((rr & 3) + 8 * (rr >> 2))) & 1u;
          z[rr] = f ? zz : kNegInf;
          zmax = fmaxf(zmax, z[rr]);
        }
        zmax = fmaxf(zmax, rl4co::bfly_f<32>(zmax));
        const float zs = (zmax > kNegInf) ? zmax : 0.0f;
        float se = 0.0f, best = kNegInf, best_z = kNegInf, fz = kNegInf;
        int bi = 0x7fffffff;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {  // 4 consecutive nodes 32 w + 8 g4 + 4 hi + (0..3)
          const int node0 = 32 * w + 8 * g4 + 4 * hi;
          float nz[4] = {1.0f, 1.0f, 1.0f, 1.0f};
          if (a.mode == RL4CO_DECODE_SAMPLE) {
            if (a.exp_noise) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                nz[i] = (node0 + i < N && lane_ok) ? a.exp_noise[((int64_t)t * a.B + r) * N + node0 + i] : 1.0f;
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                nz[i] = rl4co_exp1_noise(a.philox_seed, a.philox_offset + (uint64_t)tcol, (uint32_t)r, (uint32_t)(node0 + i));
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rr = 4 * g4 + i, node = node0 + i;
            const float zz = z[rr];
            se += __expf(zz - zs);
            // multinomial(p,1) == argmax(p / Exp(1)) == argmax(z - log(noise)); greedy: noise = 1
            const float key = (a.mode == RL4CO_DECODE_SAMPLE) ? zz - __logf(nz[i]) : zz;
            if (zz > kNegInf && (key > best || (key == best && node < bi))) {
              best = key;
              best_z = zz;
              bi = node;
            }
            if (node == forced) fz = zz;
          }
        }
        se += rl4co::bfly_f<32>(se);
        {  // combine the two halves (they hold different nodes of the same trajectory)
          const float ob = rl4co::bfly_f<32>(best), oz = rl4co::bfly_f<32>(best_z);
          const int oi = rl4co::bfly_i<32>(bi);
          if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) {
            best = ob;
            best_z = oz;
            bi = oi;
          }
          fz = fmaxf(fz, rl4co::bfly_f<32>(fz));
        }
        if (hi == 0) {
          Xchg x;
          x.zmax = zmax;
          x.se = (zmax > kNegInf) ? se : 0.0f;
          x.best_key = best;
          x.best_z = best_z;
          x.best_idx = bi;
          x.forced_z = fz;
          xs[w * 32 + l31] = x;
        }
      }
      __syncthreads();  // B3

      // ---- 4. every wave / every query thread: finish the selection of its trajectory, transition -----
      auto select = [&](int traj, int& action, float& logp) {
        float zmax = kNegInf;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) zmax = fmaxf(zmax, xs[ww * 32 + traj].zmax);
        float tot = 0.0f, best = kNegInf, bz = kNegInf, fz = kNegInf;
        int bi = 0x7fffffff;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
          const Xchg x = xs[ww * 32 + traj];
          tot += (x.zmax > kNegInf) ? x.se * __expf(x.zmax - zmax) : 0.0f;
          if (x.best_idx != 0x7fffffff && (bi == 0x7fffffff || x.best_key > best || (x.best_key == best && x.best_idx < bi))) {
            best = x.best_key;
            bz = x.best_z;
            bi = x.best_idx;
          }
          fz = fmaxf(fz, x.forced_z);
        }
        const float lse = zmax + __logf(tot);
        if (a.mode == RL4CO_DECODE_EVALUATE) {
          action = -2;  // caller substitutes the forced action
          logp = fz - lse;
        } else {
          action = (bi == 0x7fffffff) ? 0 : bi;
          logp = bz - lse;
        }
      };
      {
        int act;
        float logp;
        select(l31, act, logp);
        if (a.mode == RL4CO_DECODE_EVALUATE) {
          act = lane_ok ? (int)a.forced_actions[(int64_t)r * a.out_stride + tcol] : 0;
          if (act < 0 || act >= N) {
            if (lane_ok && !done) errbits |= RL4CO_EBIT_INFEASIBLE;
            act = 0;
          }
        }
        if (!done) {
          if (!mw.test(act)) errbits |= RL4CO_EBIT_INFEASIBLE;
          if (!(logp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;
          if (w == 0 && hi == 0) {
            a.actions[(int64_t)r * a.out_stride + tcol] = act;
            a.logps[(int64_t)r * a.out_stride + tcol] = logp;
          }
          nsteps = t + 1;
          // environment transition on the lane-resident state
          if (ENV == RL4CO_ENV_TSP) {
            if (step_i == 0) first = act;
            cur = act;
            step_i += 1;
            mw.set(act, false);
            done = (mw.w[0] | mw.w[1] | mw.w[2] | mw.w[3]) == 0u;
          } else {
            const int di = min(max(act - 1, 0), N - 2);
            used = (used + dems[di + 1]) * (act != 0 ? 1.0f : 0.0f);
            cur = act;
            vw.set(act, true);
            bool all_visited = true, any_feasible = false;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
              uint32_t mbits = 0;
              for (int b = 0; b < 32 && 32 * kt + b < N; ++b) {
                const int j = 32 * kt + b;
                const bool v = (vw.w[kt] >> b) & 1u;
                all_visited &= v;
                if (j >= 1) {
                  const bool masked = v || (dems[j] + used > thr);
                  mbits |= masked ? 0u : (1u << b);
                  any_feasible |= !masked;
                }
              }
              mw.w[kt] = mbits;
            }
            if (!((cur == 0) && any_feasible)) mw.w[0] |= 1u;
            done = all_visited;
          }
        }
      }
      if (single) {
        ++t;
        break;
      }
    }

    // ---- write back the column tile's final state -------------------------------------------------------
    if (w == 0 && hi == 0 && lane_ok) {
      uint8_t* gm = a.action_mask + (int64_t)r * N;
      for (int j = 0; j < N; ++j) gm[j] = mw.test(j) ? 1 : 0;
      if (ENV == RL4CO_ENV_CVRP) {
        uint8_t* gv = a.visited + (int64_t)r * N;
        for (int j = 0; j < N; ++j) gv[j] = vw.test(j) ? 1 : 0;
        a.used_capacity[r] = used;
      } else {
        a.first_node[r] = first;
        a.step_i[r] = step_i;
      }
      a.current_node[r] = cur;
      a.done[r] = done ? 1 : 0;
      if (a.n_steps) a.n_steps[r] = nsteps;
      if (a.entropy) a.entropy[r] += ent_acc;
      if (!single && !done) errbits |= RL4CO_EBIT_MAX_STEPS;
    }
    __syncthreads();
  }
  if (errbits && (lane & 31) == 0) atomicOr(a.err, (int)errbits);
}

}  // namespace

extern "C" int rl4co_am_decode_ms_lds_bytes(void) {
  return (3 * 128 + 2 * 32) * kRS * 2 + 4 * 32 * (int)sizeof(Xchg) + 128 * 4;
}

namespace rl4co {
int launch_decode_ms(const rl4co_am_decode_args& a, hipStream_t stream) {
  const int lds = rl4co_am_decode_ms_lds_bytes();
  if (a.env == RL4CO_ENV_TSP) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_decode_ms_kernel<RL4CO_ENV_TSP>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((am_decode_ms_kernel<RL4CO_ENV_TSP>), dim3(a.B_inst), dim3(kThreads), lds, stream, a);
  } else {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_decode_ms_kernel<RL4CO_ENV_CVRP>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((am_decode_ms_kernel<RL4CO_ENV_CVRP>), dim3(a.B_inst), dim3(kThreads), lds, stream, a);
  }
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
}  // namespace rl4co
