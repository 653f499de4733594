// am_cross_attn.hip — the decoder's glimpse attention over ALL steps of given trajectories at once, forward and backward
// (r06): the masked multi-head attention of the dense re-evaluation (policy.evaluate_log_probs) — the training gradient
// beyond the teacher kernels' node limit, `evaluate` decoding with autograd, PPO's re-evaluation
// (rl/ppo/ppo.py:128-170; decoder: models/zoo/am/decoder.py:150-190, nn/attention.py:255-296 PointerAttention's
// inner multi-head attention with the action mask).
//
//   queries  q [B, T, 128]        one row per (trajectory, step): 8 heads x 16
//   keys     kv [B_inst, N, k 128 | v 128]   per INSTANCE: trajectory b reads instance b % B_inst (multistart rows are s-major)
//   mask     bits [B, T, W] (W words of 32 keys, W % 4 == 0): bit j set = node j feasible at that step (rl4co_env_replay)
//   forward  heads [B, T, 128] = softmax_keys(q k^T / 4 masked) v ; lse [B, 8, T] (log2 domain)
//   backward dq [B, T, 128] ; dkv [B_inst, N, dk 128 | dv 128] summed over the steps AND the starts of an instance
//
// Same machinery as the training self-attention beyond 128 nodes (am_attn_flash.hip forward with the log-sum-exp,
// am_train_attn.hip: attn_bwd_wide_kernel): the forward streams the keys through LDS in blocks of 64 with an online softmax,
// the backward gives a workgroup one chunk of 128 keys of (instance, head half) — d k / d v complete in registers over every
// query block of every start, the chunks' shares of d q as fp32 rows summed in chunk order. What is new is the operand
// addressing (T queries != N keys, shared keys) and the mask: one 8- or 16-byte load per query and key block, a select per score.
// bf16 / fp16 operands, fp32 accumulation and softmax; tolerance-tested against torch SDPA with the same mask.
#include <hip/hip_runtime.h>

#include "common.h"
#include "elem16.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kKB = 64;          // keys per LDS block (forward)
constexpr int kQT = 4;           // query tiles (of 16) per workgroup pass (forward)
constexpr int kKS = 2 * kD + 8;  // LDS row stride of a k | v row (bf16 elements)
constexpr int kOS = kD + 8;      // LDS row stride of the output staging rows
constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kScale = 0.25f * 1.44269504088896341f;  // 1/sqrt(16) in the exp2 domain

typedef elem_t bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ inline f32x4 mfma16(const bf16x4& a, const bf16x4& b, const f32x4& c) { return rl4co_e16::mfma_16x16x16(a, b, c); }
__device__ inline f32x4 zero4() { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
__device__ inline bf16x4 lds_b64(const elem_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ inline bf16x4 lds_tr(const elem_t* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ inline bf16x4 to_e4(const f32x4& v) { return rl4co_e16::cvt4(v[0], v[1], v[2], v[3]); }
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ inline float rg_sum(float v) { return rl4co::bfly_sum<16, 64>(v); }

// ---- forward: one workgroup = (trajectory, block of kQT * 16 steps); wave h owns head h -----------------------------------
__global__ void __launch_bounds__(kThreads, 4) cross_fwd_kernel(const rl4co_cross_attn_args a, int QB) {
  constexpr int kLds = kKB * kKS > kQT * 16 * kOS ? kKB * kKS : kQT * 16 * kOS;
  __shared__ __align__(16) elem_t kv[kLds];
  const int tid = threadIdx.x, h = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int traj = blockIdx.x / QB, qb = blockIdx.x % QB;
  const int T = a.T, N = a.N;
  const uint16_t* qbase = static_cast<const uint16_t*>(a.q) + (int64_t)traj * T * a.q_stride;
  const uint16_t* kvbase = static_cast<const uint16_t*>(a.kv) + (int64_t)(traj % a.B_inst) * N * a.kv_stride;
  const int q0 = qb * kQT * 16;
  const int nblocks = (N + kKB - 1) / kKB;
  const int W = a.mask_words;

  bf16x4 qf[kQT];
  float m[kQT], l[kQT];
  f32x4 o[kQT];
  const uint32_t* mbase = a.mask ? a.mask + (int64_t)traj * T * W : nullptr;  // (uniform; the rows are 32-bit offsets)
  int moff[kQT];
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    const int q = min(q0 + 16 * t + tl, T - 1);  // clamped: rows >= T are computed and dropped
    qf[t] = *reinterpret_cast<const bf16x4*>(qbase + (int64_t)q * a.q_stride + 16 * h + 4 * g);
    moff[t] = q * W;
    m[t] = kNegInf;
    l[t] = 0.0f;
    o[t] = zero4();
  }
  uint4 pre[4];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + j * kThreads;
      const int row = min(kb * kKB + (c >> 5), N - 1), col = (c & 31) * 8;
      pre[j] = *reinterpret_cast<const uint4*>(kvbase + (int64_t)row * a.kv_stride + col);
    }
  };
  auto commit = [&](int kb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + j * kThreads;
      const int r = c >> 5, col = (c & 31) * 8;
      const bool ok = kb * kKB + r < N;
      *reinterpret_cast<uint4*>(kv + r * kKS + col) = ok ? pre[j] : make_uint4(0, 0, 0, 0);
    }
  };
  const int nao = tl * kKS + 4 * g;
  const int tro = (4 * g + (tl >> 2)) * kKS + 4 * (tl & 3);
  fetch(0);
  for (int kb = 0; kb < nblocks; ++kb) {
    if (kb > 0) __syncthreads();
    commit(kb);
    __syncthreads();
    fetch(min(kb + 1, nblocks - 1));  // (the last block re-reads itself: no branch around the loads)
    bf16x4 kf[4], vf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kf[j] = lds_b64(kv + 16 * j * kKS + 16 * h + nao);
      vf[j] = lds_tr(kv + 16 * j * kKS + kD + 16 * h + tro);
    }
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
      // this query's feasibility bits of the block's 64 keys (two words); keys past the graph carry no bit
      uint2 mw = make_uint2(0xffffffffu, 0xffffffffu);
      if (mbase != nullptr) mw = *reinterpret_cast<const uint2*>(mbase + moff[t] + 2 * kb);
      f32x4 s[4];
      float bm = kNegInf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[j] = mfma16(kf[j], qf[t], zero4());
        const uint32_t bits = ((j & 2) ? mw.y : mw.x) >> ((j & 1) * 16 + 4 * g);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const bool ok = ((bits >> rr) & 1u) != 0 && kb * kKB + 16 * j + 4 * g + rr < N;
          const float v = ok ? s[j][rr] * kScale : kNegInf;
          s[j][rr] = v;
          bm = fmaxf(bm, v);
        }
      }
      bm = rl4co::bfly_max<16, 64>(bm);
      const float mn = fmaxf(m[t], bm);
      // a block (or every block so far) without a feasible key: the maximum stays -inf, nothing is added, nothing rescaled
      const float base = mn > kNegInf ? mn : 0.0f;
      const float alpha = __builtin_amdgcn_exp2f(m[t] - base);  // exp2(-inf) = 0 on the first feasible block
      m[t] = mn;
      float ls = 0.0f;
      f32x4 acc = {o[t][0] * alpha, o[t][1] * alpha, o[t][2] * alpha, o[t][3] * alpha};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          p[rr] = __builtin_amdgcn_exp2f(s[j][rr] - base);
          ls += p[rr];
        }
        acc = mfma16(vf[j], rl4co_e16::cvt4(p[0], p[1], p[2], p[3]), acc);
      }
      o[t] = acc;
      l[t] = fmaf(l[t], alpha, ls);
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    const float lt = rl4co::bfly_sum<16, 64>(l[t]);
    const float inv = __builtin_amdgcn_rcpf(lt);
    if (a.lse != nullptr && g == 0 && q0 + 16 * t + tl < T)
      a.lse[((int64_t)traj * kWaves + h) * T + q0 + 16 * t + tl] = m[t] + __builtin_amdgcn_logf(lt);
    *reinterpret_cast<bf16x4*>(kv + (16 * t + tl) * kOS + 16 * h + 4 * g) =
        rl4co_e16::cvt4(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv);
  }
  __syncthreads();
  const int rows = min(kQT * 16, T - q0);
  uint16_t* dst = static_cast<uint16_t*>(a.out) + ((int64_t)traj * T + q0) * kD;
  for (int c = tid; c < rows * 16; c += kThreads) {
    const int row = c >> 4, col = (c & 15) * 8;
    *reinterpret_cast<uint4*>(dst + (int64_t)row * kD + col) = *reinterpret_cast<const uint4*>(kv + row * kOS + col);
  }
}

// ---- backward: one workgroup = (instance, half of the heads, chunk of 128 keys); four waves = four heads ---------------------
constexpr int kBwdWaves = 4;
constexpr int kBwdThreads = 64 * kBwdWaves;
constexpr int kWT = 8;            // key tiles per chunk
constexpr int kKH = kD + 8;       // LDS row stride of k | v of four heads (64 + 64 columns)
constexpr int kSS = 16;           // a wave's dS block: [128 keys][16 queries]
constexpr int kQD = 32 + 8;       // a wave's [16 queries][16 d-out | 16 q columns] rows
constexpr int kDW = 2 * 64 + 8;   // output staging row: d k | d v of four heads

__global__ void __launch_bounds__(kBwdThreads, 3) cross_bwd_kernel(const rl4co_cross_attn_args a, int KC) {
  constexpr int NT = kWT;
  extern __shared__ __align__(16) unsigned char smem[];
  elem_t* kv = reinterpret_cast<elem_t*>(smem);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int hh = blockIdx.x & 1, kc = (blockIdx.x >> 1) % KC;
  const int inst = (blockIdx.x >> 1) / KC;
  const int h = 4 * hh + w;
  const int T = a.T, N = a.N, W = a.mask_words;
  const int S = a.B / a.B_inst;
  const int key0 = 16 * NT * kc;
  constexpr int kWaveStage = NT * 16 * kSS + 16 * kQD + 64;
  elem_t* dsb = kv + NT * 16 * kKH + w * kWaveStage;
  elem_t* qd = dsb + NT * 16 * kSS;
  float2* ld = reinterpret_cast<float2*>(qd + 16 * kQD);
  const uint16_t* kvbase = static_cast<const uint16_t*>(a.kv) + (int64_t)inst * N * a.kv_stride;
  {
    constexpr int total = NT * 16 * 16;
    for (int c0 = tid; c0 < total; c0 += 8 * kBwdThreads) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = min(c0 + j * kBwdThreads, total - 1);
        const int row = min(key0 + (c >> 4), N - 1), part = (c >> 3) & 1, ch = c & 7;
        v[j] = *reinterpret_cast<const uint4*>(kvbase + (int64_t)row * a.kv_stride + kD * part + 64 * hh + 8 * ch);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j * kBwdThreads;
        if (c < total) {
          const int row = c >> 4, col = (c & 15) * 8;
          *reinterpret_cast<uint4*>(kv + row * kKH + col) = key0 + row < N ? v[j] : make_uint4(0, 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  const int nao = tl * kKH + 4 * g;
  const int tro = (4 * g + (tl >> 2)) * kKH + 4 * (tl & 3), tro_q = (4 * g + (tl >> 2)) * kQD + 4 * (tl & 3),
            tro_s = (4 * g + (tl >> 2)) * kSS + 4 * (tl & 3);
  f32x4 dk[NT], dv[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    dk[jt] = zero4();
    dv[jt] = zero4();
  }
  const int QT = (T + 15) >> 4;
  const uint16_t* qp = static_cast<const uint16_t*>(a.q);
  const uint16_t* dop = static_cast<const uint16_t*>(a.dout);
  const uint16_t* op = static_cast<const uint16_t*>(a.out);
  // the query blocks of every start of this instance, one after the other: (start, block) = (it / QT, it % QT)
  auto rows_of = [&](int it, int64_t& row, int64_t& lrow_base) {
    const int s = it / QT, tb = it - s * QT;
    const int64_t traj = (int64_t)s * a.B_inst + inst;
    row = traj * T + min(16 * tb + tl, T - 1);
    lrow_base = (traj * kWaves + h) * T + min(16 * tb + tl, T - 1);
  };
  const int total_it = S * QT;
  int64_t r0, l0;
  rows_of(0, r0, l0);
  uint2 q_next = *reinterpret_cast<const uint2*>(qp + r0 * a.q_stride + 16 * h + 4 * g);
  uint2 do_next = *reinterpret_cast<const uint2*>(dop + r0 * kD + 16 * h + 4 * g);
  uint2 o_next = *reinterpret_cast<const uint2*>(op + r0 * kD + 16 * h + 4 * g);
  float L_next = a.lse[l0];
  for (int it = 0; it < total_it; ++it) {
    const int s = it / QT, tb = it - s * QT;
    const int64_t traj = (int64_t)s * a.B_inst + inst;
    const int t = 16 * tb + tl;
    const bool tv = t < T;
    const bf16x4 qf = __builtin_bit_cast(bf16x4, q_next);
    const bf16x4 dof = __builtin_bit_cast(bf16x4, tv ? do_next : make_uint2(0u, 0u));
    const float L = L_next;
    float dsum;  // -D = -sum_d dO O of this (query, head)
    {
      const uint2 ou = o_next, du = tv ? do_next : make_uint2(0u, 0u);
      dsum = rl4co_e16::lo(ou.x) * rl4co_e16::lo(du.x);
      dsum = fmaf(rl4co_e16::hi(ou.x), rl4co_e16::hi(du.x), dsum);
      dsum = fmaf(rl4co_e16::lo(ou.y), rl4co_e16::lo(du.y), dsum);
      dsum = fmaf(rl4co_e16::hi(ou.y), rl4co_e16::hi(du.y), dsum);
      dsum = -rg_sum(dsum);
    }
    // feasibility bits of this chunk's 128 keys for the lane's accumulator rows (queries 4 g .. 4 g + 3 of the block)
    uint4 mw[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      mw[rr] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
      if (a.mask != nullptr)
        mw[rr] = *reinterpret_cast<const uint4*>(a.mask + (traj * T + min(16 * tb + 4 * g + rr, T - 1)) * W + 4 * kc);
    }
    {
      int64_t rn, ln;
      rows_of(min(it + 1, total_it - 1), rn, ln);  // (the last block re-reads itself: no branch around the loads)
      q_next = *reinterpret_cast<const uint2*>(qp + rn * a.q_stride + 16 * h + 4 * g);
      do_next = *reinterpret_cast<const uint2*>(dop + rn * kD + 16 * h + 4 * g);
      o_next = *reinterpret_cast<const uint2*>(op + rn * kD + 16 * h + 4 * g);
      L_next = a.lse[ln];
    }
    *reinterpret_cast<bf16x4*>(qd + tl * kQD + 4 * g) = dof;
    *reinterpret_cast<bf16x4*>(qd + tl * kQD + 16 + 4 * g) = qf;
    if (g == 0) ld[tl] = make_float2(L, dsum);
    wave_lds_sync();
    const float4 s01 = *reinterpret_cast<const float4*>(ld + 4 * g), s23 = *reinterpret_cast<const float4*>(ld + 4 * g + 2);
    const float Lr[4] = {s01.x, s01.z, s23.x, s23.z};
    const f32x4 negD = {s01.y, s01.w, s23.y, s23.w};
    const bf16x4 dt = lds_tr(qd + tro_q);
    const bf16x4 qt = lds_tr(qd + 16 + tro_q);
    constexpr int kG = 2;
#pragma clang loop unroll(full)
    for (int j0 = 0; j0 < NT; j0 += kG) {
      bf16x4 kf[kG], vf[kG], pf[kG], dsf[kG];
      f32x4 sc[kG], dp[kG];
#pragma unroll
      for (int j = 0; j < kG; ++j) {
        kf[j] = lds_b64(kv + 16 * (j0 + j) * kKH + 16 * w + nao);
        vf[j] = lds_b64(kv + 16 * (j0 + j) * kKH + 64 + 16 * w + nao);
      }
#pragma unroll
      for (int j = 0; j < kG; ++j) sc[j] = mfma16(qf, kf[j], zero4());  // S[query 4 g + r][key tl]
#pragma unroll
      for (int j = 0; j < kG; ++j) dp[j] = mfma16(dof, vf[j], negD);    // dP - D
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kG; ++j) {
        const int jt = j0 + j;  // key 16 jt + tl of the chunk: word jt >> 1, bit (jt & 1) * 16 + tl
        const bool in_graph = key0 + 16 * jt + tl < N;
        float p4[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const uint32_t word = (jt >> 1) == 0 ? mw[rr].x : ((jt >> 1) == 1 ? mw[rr].y : ((jt >> 1) == 2 ? mw[rr].z : mw[rr].w));
          const bool ok = in_graph && ((word >> ((jt & 1) * 16 + tl)) & 1u) != 0;
          const float e = __builtin_amdgcn_exp2f(fmaf(sc[j][rr], kScale, -Lr[rr]));
          p4[rr] = ok ? e : 0.0f;  // masked keys and keys past the graph: P = 0, hence dS = 0
        }
        pf[j] = rl4co_e16::cvt4(p4[0], p4[1], p4[2], p4[3]);
        dsf[j] = rl4co_e16::cvt4(p4[0] * dp[j][0], p4[1] * dp[j][1], p4[2] * dp[j][2], p4[3] * dp[j][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kG; ++j) {
        dv[j0 + j] = mfma16(dt, pf[j], dv[j0 + j]);
        dk[j0 + j] = mfma16(qt, dsf[j], dk[j0 + j]);
        *reinterpret_cast<bf16x4*>(dsb + (16 * (j0 + j) + tl) * kSS + 4 * g) = dsf[j];
      }
    }
    wave_lds_sync();
    f32x4 dq = zero4();
#pragma clang loop unroll(full)
    for (int jt = 0; jt < NT; ++jt)
      dq = mfma16(lds_tr(kv + 16 * jt * kKH + 16 * w + tro), lds_tr(dsb + 16 * jt * kSS + tro_s), dq);
    if (tv) *reinterpret_cast<f32x4*>(a.dq_partial + (((int64_t)kc * a.B + traj) * T + t) * kD + 16 * h + 4 * g) = dq;
    wave_lds_sync();
  }
  __syncthreads();
  elem_t* os = reinterpret_cast<elem_t*>(smem);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) dk[jt][rr] *= 0.25f;
    elem_t* row = os + (16 * jt + tl) * kDW + 16 * w + 4 * g;
    *reinterpret_cast<bf16x4*>(row) = to_e4(dk[jt]);
    *reinterpret_cast<bf16x4*>(row + 64) = to_e4(dv[jt]);
  }
  __syncthreads();
  const int nk = min(16 * NT, N - key0);
  uint16_t* dkv = static_cast<uint16_t*>(a.dkv);
  for (int c = tid; c < nk * 16; c += kBwdThreads) {
    const int row = c >> 4, seg = (c >> 3) & 1, ch = c & 7;
    *reinterpret_cast<uint4*>(dkv + ((int64_t)inst * N + key0 + row) * 2 * kD + seg * kD + 64 * hh + 8 * ch) =
        *reinterpret_cast<const uint4*>(os + row * kDW + 64 * seg + 8 * ch);
  }
}

// d q = 1 / 4 of the sum of the key chunks' shares, in chunk order
__global__ void __launch_bounds__(256) cross_dq_reduce_kernel(const float* __restrict__ part, int KC, int64_t rows,
                                                              uint16_t* __restrict__ dq) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * 16) return;
  const int64_t row = idx >> 4;
  const int col = (int)(idx & 15) * 8;
  f32x4 a0 = zero4(), a1 = zero4();
  for (int c = 0; c < KC; ++c) {
    const float* p = part + ((int64_t)c * rows + row) * kD + col;
    a0 += *reinterpret_cast<const f32x4*>(p);
    a1 += *reinterpret_cast<const f32x4*>(p + 4);
  }
  uint16_t* dst = dq + row * kD + col;
  *reinterpret_cast<bf16x4*>(dst) = rl4co_e16::cvt4(0.25f * a0[0], 0.25f * a0[1], 0.25f * a0[2], 0.25f * a0[3]);
  *reinterpret_cast<bf16x4*>(dst + 4) = rl4co_e16::cvt4(0.25f * a1[0], 0.25f * a1[1], 0.25f * a1[2], 0.25f * a1[3]);
}

int validate(const rl4co_cross_attn_args& a) {
  RL4CO_REQUIRE(a.B > 0 && a.B_inst > 0 && a.B % a.B_inst == 0 && a.T >= 1 && a.N >= 1 && a.N <= 65536);
  RL4CO_REQUIRE(a.q && a.kv && a.out && a.lse);
  RL4CO_REQUIRE(a.q_stride >= kD && a.q_stride % 8 == 0 && a.kv_stride >= 2 * kD && a.kv_stride % 8 == 0);
  RL4CO_REQUIRE((reinterpret_cast<uintptr_t>(a.q) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.kv) & 15) == 0);
  if (a.mask != nullptr)  // 16-byte loads of a chunk's four words
    RL4CO_REQUIRE(a.mask_words % 4 == 0 && a.mask_words * 32 >= a.N && (reinterpret_cast<uintptr_t>(a.mask) & 15) == 0);
  return RL4CO_OK;
}

}  // namespace

#if !RL4CO_ELEM_F16
extern "C" int rl4co_cross_attn_chunks(int N) { return (N + 16 * kWT - 1) / (16 * kWT); }
#endif

extern "C" int RL4CO_ENTRY(rl4co_cross_attn_fwd)(const rl4co_cross_attn_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_cross_attn_args& a = *args;
  const int st = validate(a);
  if (st != RL4CO_OK) return st;
  const int QB = (a.T + kQT * 16 - 1) / (kQT * 16);
  RL4CO_REQUIRE((int64_t)a.B * QB < (1ll << 31));
  hipLaunchKernelGGL(cross_fwd_kernel, dim3(a.B * QB), dim3(kThreads), 0, rl4co::as_stream(stream), a, QB);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_cross_attn_bwd)(const rl4co_cross_attn_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_cross_attn_args& a = *args;
  const int st = validate(a);
  if (st != RL4CO_OK) return st;
  RL4CO_REQUIRE(a.dout && a.dq && a.dkv && a.dq_partial);
  const int KC = (a.N + 16 * kWT - 1) / (16 * kWT);
  RL4CO_REQUIRE((int64_t)a.B_inst * KC * 2 < (1ll << 31));
  hipStream_t s = rl4co::as_stream(stream);
  constexpr int work = (kWT * 16 * kKH + kBwdWaves * (kWT * 16 * kSS + 16 * kQD + 64)) * 2, stage = kWT * 16 * kDW * 2;
  constexpr int lds = work > stage ? work : stage;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cross_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(cross_bwd_kernel, dim3(2 * a.B_inst * KC), dim3(kBwdThreads), lds, s, a, KC);
  RL4CO_HIP_TRY(hipGetLastError());
  const int64_t rows = (int64_t)a.B * a.T;
  hipLaunchKernelGGL(cross_dq_reduce_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s, a.dq_partial, KC, rows,
                     static_cast<uint16_t*>(a.dq));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
