// am_train_attn.hip — encoder self-attention for training, forward and backward, straight on the
// packed projection output (nn/attention.py:64-134 MultiHeadAttention).
//
// The reference computes qkv = Wqkv(x), rearranges it to [3, B, heads, N, 16] and calls
// scaled_dot_product_attention. At N ~ 100 nodes and 16 dims per head the library flash kernels
// are launch- and layout-bound (forward 0.44 ms, backward 1.25 ms per layer at 4096 x 100, plus
// the cat that re-packs dq / dk / dv into d qkv). Here one 512-thread workgroup owns one instance:
// its [N, 384] qkv rows sit in LDS once, wave h owns head h, and everything runs on
// v_mfma_f32_16x16x16_bf16 in the layout am_teacher_mma.hip uses — the QUERY is the accumulator
// column, so the softmax over keys is in-lane plus two row-group exchanges and probabilities chain
// into the value product without a shuffle; products that contract over keys or over queries read
// the same LDS rows through ds_read_b64_tr_b16.
//
//   forward   S^T = K_h Q_h^T / 4 ; P = softmax_keys ; O_h^T = V_h^T P^T      -> out [B,N,128], L = log-sum-exp [B,8,N]
//   backward  P = exp(S - L) ; dP^T = V_h dO_h^T ; D = sum_keys P dP ; dS = P (dP - D)
//             dQ_h^T = K_h^T dS^T / 4 ; dK_h^T += Q_h^T dS / 4 ; dV_h^T += dO_h^T P   -> d qkv [B,N,384]
//
// bf16 operands, fp32 accumulation and softmax; tolerance-tested against torch SDPA.
#include <hip/hip_runtime.h>

#include "common.h"
#include "elem16.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kQS = 3 * kD + 8;  // LDS row stride of the qkv tile (bf16)
constexpr int kKV = 2 * kD + 8;  // LDS row stride of a k | v row (forward)
constexpr int kOS = kD + 8;      // LDS row stride of the staged output rows (forward)
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kScale = 0.25f * 1.44269504088896341f;  // 1/sqrt(16) in the exp2 domain

typedef elem_t bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ inline f32x4 mfma16(const bf16x4& a, const bf16x4& b, const f32x4& c) {
  return rl4co_e16::mfma_16x16x16(a, b, c);
}
__device__ inline f32x4 zero4() { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
__device__ inline bf16x4 lds_b64(const elem_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ inline bf16x4 lds_tr(const elem_t* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ inline bf16x4 to_bf16(const f32x4& v) { return rl4co_e16::cvt4(v[0], v[1], v[2], v[3]); }  // two pair conversions (elem16.h)
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ inline float rg_sum(float v) { return rl4co::bfly_sum<16, 64>(v); }
__device__ inline float rg_max(float v) { return rl4co::bfly_max<16, 64>(v); }

// rows [0, 16 NT) x `cols` bf16 columns of an instance -> LDS (rows >= N zero)
template <int NT>
__device__ inline void stage_rows(const uint16_t* __restrict__ src, int N, int cols, elem_t* dst, int stride, int tid) {
  const int cpr = cols / 8;  // 16-byte chunks per row
  const int total = NT * 16 * cpr;
  // eight chunks per thread in flight: load -> LDS store one chunk at a time is one exposed HBM round trip per chunk
  for (int c0 = tid; c0 < total; c0 += 8 * kThreads) {
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j * kThreads;
      const int row = min(c / cpr, N - 1), col = (c % cpr) * 8;  // clamped: rows >= N are zeroed below
      v[j] = *reinterpret_cast<const uint4*>(src + (int64_t)row * cols + col);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j * kThreads;
      if (c < total) {
        const int row = c / cpr, col = (c % cpr) * 8;
        *reinterpret_cast<uint4*>(dst + row * stride + col) = row < N ? v[j] : make_uint4(0, 0, 0, 0);
      }
    }
  }
}

// rows [0, 16 NT) x the k | v columns (128 .. 383 of the packed row) of an instance -> LDS (rows >= N zero)
template <int NT>
__device__ inline void stage_kv(const uint16_t* __restrict__ src, int N, elem_t* dst, int tid) {
  constexpr int cpr = 2 * kD / 8;  // 16-byte chunks per row
  constexpr int total = NT * 16 * cpr;
  for (int c0 = tid; c0 < total; c0 += 8 * kThreads) {
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = min(c0 + j * kThreads, total - 1);
      const int row = min(c / cpr, N - 1), col = (c % cpr) * 8;
      v[j] = *reinterpret_cast<const uint4*>(src + (int64_t)row * 3 * kD + kD + col);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j * kThreads;
      if (c < total) {
        const int row = c / cpr, col = (c % cpr) * 8;
        *reinterpret_cast<uint4*>(dst + row * kKV + col) = row < N ? v[j] : make_uint4(0, 0, 0, 0);
      }
    }
  }
}

// Forward. Only the keys and values sit in LDS ([16 NT][k 128 | v 128]: 59 KB at N <= 112, so TWO workgroups share a
// CU and one stages its instance while the other computes); a query block's fragment is 8 bytes per lane in exactly
// the B-operand layout and comes straight from global memory one block ahead; the finished output tiles wait in
// registers (4 per block) and leave through the dead k | v rows as contiguous 16-byte lanes.
template <int NT>
__global__ void __launch_bounds__(kThreads, 4) attn_fwd_kernel(const uint16_t* __restrict__ qkv, int N, uint16_t* __restrict__ out,
                                                               float* __restrict__ lse) {
  extern __shared__ __align__(16) unsigned char smem[];
  elem_t* kv = reinterpret_cast<elem_t*>(smem);  // [16 NT][kKV]: k | v of every node; afterwards the output rows [16 NT][kOS]
  const int tid = threadIdx.x, h = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int64_t inst = blockIdx.x;
  const uint16_t* base = qkv + inst * N * 3 * kD;
  const uint16_t* qrow = base + 16 * h + 4 * g;
  auto load_q = [&](int tb) { return *reinterpret_cast<const uint2*>(qrow + (int64_t)min(16 * tb + tl, N - 1) * 3 * kD); };
  uint2 q_next = load_q(0);
  stage_kv<NT>(base, N, kv, tid);
  __syncthreads();
  const int nao = tl * kKV + 4 * g;
  const int tro = (4 * g + (tl >> 2)) * kKV + 4 * (tl & 3);
  bf16x4 ov[NT];
#pragma clang loop unroll(full)
  for (int tb = 0; tb < NT; ++tb) {
    const int t = 16 * tb + tl;
    const bf16x4 qf = __builtin_bit_cast(bf16x4, q_next);
    q_next = load_q(min(tb + 1, NT - 1));
    f32x4 sc[NT];
    float m = kNegInf;
#pragma clang loop unroll(full)
    for (int jt = 0; jt < NT; ++jt) {
      sc[jt] = mfma16(lds_b64(kv + 16 * jt * kKV + 16 * h + nao), qf, zero4());
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        sc[jt][rr] = (16 * jt + 4 * g + rr < N) ? sc[jt][rr] * kScale : kNegInf;
        m = fmaxf(m, sc[jt][rr]);
      }
    }
    m = rg_max(m);
    float l = 0.0f;
    f32x4 o0 = zero4(), o1 = zero4();
#pragma clang loop unroll(full)
    for (int jt = 0; jt < NT; ++jt) {
      float p4[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        p4[rr] = __builtin_amdgcn_exp2f(sc[jt][rr] - m);
        l += p4[rr];
      }
      const bf16x4 pf = rl4co_e16::cvt4(p4[0], p4[1], p4[2], p4[3]);
      const bf16x4 vf = lds_tr(kv + 16 * jt * kKV + kD + 16 * h + tro);
      if (jt & 1) o1 = mfma16(vf, pf, o1);
      else o0 = mfma16(vf, pf, o0);
    }
    l = rg_sum(l);
    const float inv = __builtin_amdgcn_rcpf(l);
    f32x4 o;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) o[rr] = (o0[rr] + o1[rr]) * inv;
    ov[tb] = to_bf16(o);
    if (t < N && g == 0) lse[(inst * kWaves + h) * N + t] = m + __builtin_amdgcn_logf(l);  // log2 domain
  }
  // 8-byte stores from the accumulators would put 32 bytes per wave into each of 16 rows; staged over the dead keys,
  // the instance's [N, 128] output is one contiguous run of 16-byte lanes
  __syncthreads();
#pragma unroll
  for (int tb = 0; tb < NT; ++tb) *reinterpret_cast<bf16x4*>(kv + (16 * tb + tl) * kOS + 16 * h + 4 * g) = ov[tb];
  __syncthreads();
  for (int c = tid; c < N * 16; c += kThreads) {
    const int row = c >> 4, col = (c & 15) * 8;
    *reinterpret_cast<uint4*>(out + (inst * N + row) * kD + col) = *reinterpret_cast<const uint4*>(kv + row * kOS + col);
  }
}

// Backward. One 256-thread workgroup per (instance, half of the heads): wave w owns head 4 hh + w. A head only ever
// touches its own 16 columns of q, k, v and d out, so the halves share nothing; what the split buys is footprint —
// k | v of four heads (30 KB) + the four waves' staging blocks (20 KB) let THREE workgroups (12 waves) share a CU
// where the eight-head workgroup (150 KB) ran alone. Per query block the q / d out fragments (8 bytes per lane) and the
// row statistics come straight from global memory one block ahead. Orientation (r05): the score tile is computed with the
// QUERY on the accumulator rows, S[q][key] = Q K^T — then P and dS leave the softmax already in the B-operand layout of
// BOTH products that contract over queries (d V^T = dO^T P, d K^T = Q^T dS) and only dS goes through LDS, for the one
// product that contracts over keys (d Q^T = K^T dS^T): a [key][query] block written as 8-byte lanes and read back through
// ds_read_b64_tr_b16. (Until r05 the key sat on the rows: P AND dS each made the round trip, two more wave syncs per block.)
// d q waits in registers; at the end the whole LDS holds [N][d q | d k | d v] of the four heads and leaves as 16-byte
// lanes over 128-byte row segments.
constexpr int kBwdWaves = 4;
constexpr int kBwdThreads = 64 * kBwdWaves;
constexpr int kKH = kD + 8;       // LDS row stride of k | v of four heads (64 + 64 columns)
constexpr int kSS = 16;           // a wave's dS block: [16 NT keys][16 queries], dense (the transpose read's 8 rows x 32 bytes tile the 64 banks)
constexpr int kQD = 32 + 8;       // a wave's [16 queries][16 d-out | 16 q columns] rows
constexpr int kDS = 3 * 64 + 8;   // output staging row: d q | d k | d v of four heads

#ifndef RL4CO_ATTN_BWD_PROBE
#define RL4CO_ATTN_BWD_PROBE 0  // timing probes (tools/kernel_variant.sh): 1 = no query blocks (load + store phases only), 2 = no
#endif                          // exponentials, 3 = per-phase shader-clock sums in g_attn_bwd_clk (rl4co_attn_bwd_probe_read)
#if RL4CO_ATTN_BWD_PROBE >= 3
__device__ unsigned long long g_attn_bwd_clk[4];
#define RL4CO_ABP_MARK(i)                                                               \
  {                                                                                     \
    const unsigned long long now_ = __builtin_readcyclecounter();                       \
    if (threadIdx.x == 0) atomicAdd(&g_attn_bwd_clk[i], now_ - clk_);                    \
    clk_ = now_;                                                                        \
  }
#else
#define RL4CO_ABP_MARK(i)
#endif

template <int NT>
__global__ void __launch_bounds__(kBwdThreads, 3) attn_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                                  const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                  int N, uint16_t* __restrict__ dqkv) {
  extern __shared__ __align__(16) unsigned char smem[];
#if RL4CO_ATTN_BWD_PROBE >= 3
  unsigned long long clk_ = __builtin_readcyclecounter();
#endif
  elem_t* kv = reinterpret_cast<elem_t*>(smem);  // [16 NT][kKH]: k (4 heads) | v (4 heads)
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int64_t inst = blockIdx.x >> 1;
  const int hh = blockIdx.x & 1, h = 4 * hh + w;
  constexpr int kWaveStage = NT * 16 * kSS + 16 * kQD + 64;  // (elements) dS block | d-out, q rows | 16 x (L, -D) fp32
  elem_t* dsb = kv + NT * 16 * kKH + w * kWaveStage;
  elem_t* qd = dsb + NT * 16 * kSS;
  float2* ld = reinterpret_cast<float2*>(qd + 16 * kQD);
  const uint16_t* base = qkv + inst * N * 3 * kD;
  {  // k | v columns of this half: 16 chunks of 16 bytes per row, eight in flight per thread
    constexpr int total = NT * 16 * 16;
    for (int c0 = tid; c0 < total; c0 += 8 * kBwdThreads) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = min(c0 + j * kBwdThreads, total - 1);
        const int row = min(c >> 4, N - 1), part = (c >> 3) & 1, ch = c & 7;
        v[j] = *reinterpret_cast<const uint4*>(base + (int64_t)row * 3 * kD + kD * (1 + part) + 64 * hh + 8 * ch);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j * kBwdThreads;
        if (c < total) {
          const int row = c >> 4, col = (c & 15) * 8;
          *reinterpret_cast<uint4*>(kv + row * kKH + col) = row < N ? v[j] : make_uint4(0, 0, 0, 0);
        }
      }
    }
  }
  // (unconditional loads from clamped rows, masked where they are consumed: a load under a branch makes the compiler
  // wait for ALL outstanding loads at the join, i.e. for the prefetch it has just issued)
  const uint16_t* qrow = base + 16 * h + 4 * g;
  const uint16_t* dorow = dout + inst * N * kD + 16 * h + 4 * g;
  const uint16_t* orow = out + inst * N * kD + 16 * h + 4 * g;
  const float* lrow = lse + (inst * kWaves + h) * N;
  auto row_of = [&](int tb) { return (int64_t)min(16 * tb + tl, N - 1); };
  uint2 q_next = *reinterpret_cast<const uint2*>(qrow + row_of(0) * 3 * kD);
  uint2 do_next = *reinterpret_cast<const uint2*>(dorow + row_of(0) * kD);
  uint2 o_next = *reinterpret_cast<const uint2*>(orow + row_of(0) * kD);
  float L_next = lrow[row_of(0)];
  __syncthreads();
  const int nao = tl * kKH + 4 * g;
  const int tro = (4 * g + (tl >> 2)) * kKH + 4 * (tl & 3), tro_q = (4 * g + (tl >> 2)) * kQD + 4 * (tl & 3),
            tro_s = (4 * g + (tl >> 2)) * kSS + 4 * (tl & 3);
  f32x4 dk[NT], dv[NT];  // [d = 4 g + r of head h][key 16 jt + (lane & 15)]
  bf16x4 dqv[NT];        // d q of the block's queries, kept until the output staging
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    dk[jt] = zero4();
    dv[jt] = zero4();
  }
  RL4CO_ABP_MARK(0)
#pragma clang loop unroll(full)
  for (int tb = 0; tb < (RL4CO_ATTN_BWD_PROBE == 1 ? 0 : NT); ++tb) {
    const int t = 16 * tb + tl;
    const bool tv = t < N;
    const bf16x4 qf = __builtin_bit_cast(bf16x4, q_next);
    const bf16x4 dof = __builtin_bit_cast(bf16x4, tv ? do_next : make_uint2(0u, 0u));
    const float L = L_next;
    // D = sum_keys P dP = sum_d dO O of this (query, head): from the forward's saved output — 4 products per lane and
    // the row-group sum — instead of 4 NT products with P converted back from its 16-bit form (r05). It enters the
    // dP product as the accumulator's initial value, so that MFMA leaves dP - D and dS is one multiply per element:
    // the kernel is VALU-issue bound (profiles/r05_c4_train_pmc.json: 10 VALU per MFMA), 51 -> 20 VALU per key tile.
    float dsum;
    {
      const uint2 ou = o_next, du = tv ? do_next : make_uint2(0u, 0u);
      dsum = rl4co_e16::lo(ou.x) * rl4co_e16::lo(du.x);
      dsum = fmaf(rl4co_e16::hi(ou.x), rl4co_e16::hi(du.x), dsum);
      dsum = fmaf(rl4co_e16::lo(ou.y), rl4co_e16::lo(du.y), dsum);
      dsum = fmaf(rl4co_e16::hi(ou.y), rl4co_e16::hi(du.y), dsum);
      dsum = -rg_sum(dsum);
    }
    if (RL4CO_ATTN_BWD_PROBE != 4) {
      const int64_t rn = row_of(min(tb + 1, NT - 1));  // (the last block re-reads itself: no branch around the loads)
      q_next = *reinterpret_cast<const uint2*>(qrow + rn * 3 * kD);
      do_next = *reinterpret_cast<const uint2*>(dorow + rn * kD);
      o_next = *reinterpret_cast<const uint2*>(orow + rn * kD);
      L_next = lrow[rn];
    }
    *reinterpret_cast<bf16x4*>(qd + tl * kQD + 4 * g) = dof;
    *reinterpret_cast<bf16x4*>(qd + tl * kQD + 16 + 4 * g) = qf;
    if (g == 0) ld[tl] = make_float2(L, dsum);
    wave_lds_sync();
    // this lane's accumulator rows are queries 4 g .. 4 g + 3 of the block: their (L, -D)
    const float4 s01 = *reinterpret_cast<const float4*>(ld + 4 * g), s23 = *reinterpret_cast<const float4*>(ld + 4 * g + 2);
    const float Lr[4] = {s01.x, s01.z, s23.x, s23.z};
    const f32x4 negD = {s01.y, s01.w, s23.y, s23.w};
    const bf16x4 dt = lds_tr(qd + tro_q);       // dO_h^T[d][queries]
    const bf16x4 qt = lds_tr(qd + 16 + tro_q);  // Q_h^T[d][queries]
    // No mask on the queries past N: their d out is zero, so dP - D = 0 = dS and d v receives nothing from them, while P
    // itself stays finite (the clamped row's own scores and log-sum-exp). Keys past N: their k and v rows are zero, so
    // d q receives nothing from them and their own d k / d v rows are never stored — their P (and with it dS) is set to
    // zero by a select in the tiles that can hold them (N > 16 (smallest tile count dispatched to this NT) - 16).
    constexpr int kMinN = NT == 2 ? 1 : (NT == 4 ? 33 : (NT == 7 ? 65 : 113));
    // Key tiles in groups of four, every stage over the whole group before the next one: written tile by tile the
    // compiler keeps that order — LDS read, wait, score MFMA, 8 idle slots, exponentials, ... — and with three waves per
    // SIMD nothing covers the latencies (r05: the whole block was one dependent chain of ~ 4.6 K cycles).
    constexpr int kG = NT == 8 ? 2 : 4;  // (eight tiles: d k, d v and the waiting d q take 80 registers — groups of two fit the 168 of three waves per SIMD)
#pragma clang loop unroll(full)
    for (int j0 = 0; j0 < NT; j0 += kG) {
      bf16x4 kf[kG], vf[kG], pf[kG], dsf[kG];
      f32x4 sc[kG], dp[kG];
#pragma unroll
      for (int j = 0; j < kG; ++j)
        if (j0 + j < NT) {
          kf[j] = lds_b64(kv + 16 * (j0 + j) * kKH + 16 * w + nao);
          vf[j] = lds_b64(kv + 16 * (j0 + j) * kKH + 64 + 16 * w + nao);
        }
#pragma unroll
      for (int j = 0; j < kG; ++j)
        if (j0 + j < NT) sc[j] = mfma16(qf, kf[j], zero4());  // S[query 4 g + r][key tl]
#pragma unroll
      for (int j = 0; j < kG; ++j)
        if (j0 + j < NT) dp[j] = mfma16(dof, vf[j], negD);  // dP - D
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kG; ++j)
        if (j0 + j < NT) {
          float p4[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float e = fmaf(sc[j][rr], kScale, -Lr[rr]);
            p4[rr] = RL4CO_ATTN_BWD_PROBE == 2 ? e : __builtin_amdgcn_exp2f(e);
            // tiles that can hold keys past N: P = 0 there, hence dS = 0 (one select per element, tail tiles only). Relying on
            // the zeroed k rows alone left dS_pad = -D P_pad in the staged block: under fp16 loss scaling (large d out, L <= 0
            // so P_pad = 1) it can overflow where no valid key's dS does, and inf x 0 in the d q product is a NaN for every
            // query of the head
            if (16 * (j0 + j + 1) > kMinN) p4[rr] = (16 * (j0 + j) + tl < N) ? p4[rr] : 0.0f;
          }
          pf[j] = rl4co_e16::cvt4(p4[0], p4[1], p4[2], p4[3]);
          dsf[j] = rl4co_e16::cvt4(p4[0] * dp[j][0], p4[1] * dp[j][1], p4[2] * dp[j][2], p4[3] * dp[j][3]);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kG; ++j)
        if (j0 + j < NT) {
          dv[j0 + j] = mfma16(dt, pf[j], dv[j0 + j]);
          dk[j0 + j] = mfma16(qt, dsf[j], dk[j0 + j]);
          *reinterpret_cast<bf16x4*>(dsb + (16 * (j0 + j) + tl) * kSS + 4 * g) = dsf[j];  // [key][queries 4 g ..]
        }
    }
    wave_lds_sync();
    f32x4 dq = zero4();
#pragma clang loop unroll(full)
    for (int jt = 0; jt < NT; ++jt)
      dq = mfma16(lds_tr(kv + 16 * jt * kKH + 16 * w + tro), lds_tr(dsb + 16 * jt * kSS + tro_s), dq);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) dq[rr] *= 0.25f;
    dqv[tb] = to_bf16(dq);
    wave_lds_sync();  // the next query block rewrites this wave's staging block
  }
  RL4CO_ABP_MARK(1)
  __syncthreads();  // every wave is done with k | v: the whole LDS becomes [16 NT][d q | d k | d v] of the four heads
  RL4CO_ABP_MARK(2)
  elem_t* os = reinterpret_cast<elem_t*>(smem);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) dk[jt][rr] *= 0.25f;
    elem_t* row = os + (16 * jt + tl) * kDS + 16 * w + 4 * g;
    *reinterpret_cast<bf16x4*>(row) = dqv[jt];
    *reinterpret_cast<bf16x4*>(row + 64) = to_bf16(dk[jt]);
    *reinterpret_cast<bf16x4*>(row + 128) = to_bf16(dv[jt]);
  }
  __syncthreads();
  for (int c = tid; c < N * 24; c += kBwdThreads) {  // three 128-byte segments per row
    const int row = c / 24, seg = (c % 24) >> 3, ch = c & 7;
    *reinterpret_cast<uint4*>(dqkv + (inst * N + row) * 3 * kD + seg * kD + 64 * hh + 8 * ch) =
        *reinterpret_cast<const uint4*>(os + row * kDS + 64 * seg + 8 * ch);
  }
  RL4CO_ABP_MARK(3)
}

// Backward beyond one workgroup's nodes (N > 128; r06). The same workgroup as above — four waves = four heads of one half,
// eight key tiles — now owns ONE CHUNK of 128 keys of an instance: its k | v rows sit in LDS, their d k / d v accumulate in
// registers over ALL query blocks of the instance (the probabilities are rebuilt from the forward's log-sum-exp over all
// keys, so a chunk needs nothing from the others), and what the chunk contributes to d q — the one product that contracts
// over keys — leaves per query block as fp32 partial rows [chunk][B][N][128]; attn_dq_reduce_kernel sums the chunks in a
// fixed order, applies the 1 / 4 and writes the 16-bit d q columns. (Atomics instead of the partial rows: 2 x B N 128 of
// them into L2 at TSP-200 — as long as the kernel itself; a second, query-stationary kernel for d q: every score and
// every dP twice.)
constexpr int kWT = 8;           // key tiles per chunk
constexpr int kDW = 2 * 64 + 8;  // output staging row: d k | d v of four heads

__global__ void __launch_bounds__(kBwdThreads, 3) attn_bwd_wide_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                                       const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                       int B, int N, int KC, uint16_t* __restrict__ dqkv,
                                                                       float* __restrict__ dq_partial) {
  constexpr int NT = kWT;
  extern __shared__ __align__(16) unsigned char smem[];
  elem_t* kv = reinterpret_cast<elem_t*>(smem);  // [128][kKH]: k (4 heads) | v (4 heads) of this chunk's keys
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int hh = blockIdx.x & 1, kc = (blockIdx.x >> 1) % KC;  // the chunks and halves of ONE instance are neighbours: they
  const int64_t inst = (blockIdx.x >> 1) / KC;                 // re-read the same q / d out / out rows (L2)
  const int h = 4 * hh + w;
  const int key0 = 16 * NT * kc;
  constexpr int kWaveStage = NT * 16 * kSS + 16 * kQD + 64;
  elem_t* dsb = kv + NT * 16 * kKH + w * kWaveStage;
  elem_t* qd = dsb + NT * 16 * kSS;
  float2* ld = reinterpret_cast<float2*>(qd + 16 * kQD);
  const uint16_t* base = qkv + inst * N * 3 * kD;
  {
    constexpr int total = NT * 16 * 16;
    for (int c0 = tid; c0 < total; c0 += 8 * kBwdThreads) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = min(c0 + j * kBwdThreads, total - 1);
        const int row = min(key0 + (c >> 4), N - 1), part = (c >> 3) & 1, ch = c & 7;
        v[j] = *reinterpret_cast<const uint4*>(base + (int64_t)row * 3 * kD + kD * (1 + part) + 64 * hh + 8 * ch);
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
  const uint16_t* qrow = base + 16 * h + 4 * g;
  const uint16_t* dorow = dout + inst * N * kD + 16 * h + 4 * g;
  const uint16_t* orow = out + inst * N * kD + 16 * h + 4 * g;
  const float* lrow = lse + (inst * kWaves + h) * N;
  float* dqp = dq_partial + (((int64_t)kc * B + inst) * N) * kD + 16 * h + 4 * g;
  const int QT = (N + 15) >> 4;
  auto row_of = [&](int tb) { return (int64_t)min(16 * tb + tl, N - 1); };
  uint2 q_next = *reinterpret_cast<const uint2*>(qrow + row_of(0) * 3 * kD);
  uint2 do_next = *reinterpret_cast<const uint2*>(dorow + row_of(0) * kD);
  uint2 o_next = *reinterpret_cast<const uint2*>(orow + row_of(0) * kD);
  float L_next = lrow[row_of(0)];
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
  for (int tb = 0; tb < QT; ++tb) {
    const int t = 16 * tb + tl;
    const bool tv = t < N;
    const bf16x4 qf = __builtin_bit_cast(bf16x4, q_next);
    const bf16x4 dof = __builtin_bit_cast(bf16x4, tv ? do_next : make_uint2(0u, 0u));
    const float L = L_next;
    float dsum;  // -D = -sum_d dO O of this (query, head): see attn_bwd_kernel
    {
      const uint2 ou = o_next, du = tv ? do_next : make_uint2(0u, 0u);
      dsum = rl4co_e16::lo(ou.x) * rl4co_e16::lo(du.x);
      dsum = fmaf(rl4co_e16::hi(ou.x), rl4co_e16::hi(du.x), dsum);
      dsum = fmaf(rl4co_e16::lo(ou.y), rl4co_e16::lo(du.y), dsum);
      dsum = fmaf(rl4co_e16::hi(ou.y), rl4co_e16::hi(du.y), dsum);
      dsum = -rg_sum(dsum);
    }
    {
      const int64_t rn = row_of(min(tb + 1, QT - 1));  // (the last block re-reads itself: no branch around the loads)
      q_next = *reinterpret_cast<const uint2*>(qrow + rn * 3 * kD);
      do_next = *reinterpret_cast<const uint2*>(dorow + rn * kD);
      o_next = *reinterpret_cast<const uint2*>(orow + rn * kD);
      L_next = lrow[rn];
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
        float p4[4];
        const bool kvd = key0 + 16 * (j0 + j) + tl < N;  // keys past the graph: P = 0, hence dS = 0 (see attn_bwd_kernel)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float e = __builtin_amdgcn_exp2f(fmaf(sc[j][rr], kScale, -Lr[rr]));
          p4[rr] = kvd ? e : 0.0f;
        }
        pf[j] = rl4co_e16::cvt4(p4[0], p4[1], p4[2], p4[3]);
        dsf[j] = rl4co_e16::cvt4(p4[0] * dp[j][0], p4[1] * dp[j][1], p4[2] * dp[j][2], p4[3] * dp[j][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kG; ++j) {
        dv[j0 + j] = mfma16(dt, pf[j], dv[j0 + j]);
        dk[j0 + j] = mfma16(qt, dsf[j], dk[j0 + j]);
        *reinterpret_cast<bf16x4*>(dsb + (16 * (j0 + j) + tl) * kSS + 4 * g) = dsf[j];  // [key][queries 4 g ..]
      }
    }
    wave_lds_sync();
    f32x4 dq = zero4();
#pragma clang loop unroll(full)
    for (int jt = 0; jt < NT; ++jt)
      dq = mfma16(lds_tr(kv + 16 * jt * kKH + 16 * w + tro), lds_tr(dsb + 16 * jt * kSS + tro_s), dq);
    if (tv) *reinterpret_cast<f32x4*>(dqp + (int64_t)t * kD) = dq;  // this chunk's share, unscaled: [dims 4 g ..] of query t
    wave_lds_sync();  // the next query block rewrites this wave's staging block
  }
  __syncthreads();  // every wave is done with k | v: the LDS becomes [128][d k | d v] of the four heads
  elem_t* os = reinterpret_cast<elem_t*>(smem);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) dk[jt][rr] *= 0.25f;
    elem_t* row = os + (16 * jt + tl) * kDW + 16 * w + 4 * g;
    *reinterpret_cast<bf16x4*>(row) = to_bf16(dk[jt]);
    *reinterpret_cast<bf16x4*>(row + 64) = to_bf16(dv[jt]);
  }
  __syncthreads();
  const int nk = min(16 * NT, N - key0);
  for (int c = tid; c < nk * 16; c += kBwdThreads) {  // two 128-byte segments per row
    const int row = c >> 4, seg = (c >> 3) & 1, ch = c & 7;
    *reinterpret_cast<uint4*>(dqkv + (inst * N + key0 + row) * 3 * kD + (1 + seg) * kD + 64 * hh + 8 * ch) =
        *reinterpret_cast<const uint4*>(os + row * kDW + 64 * seg + 8 * ch);
  }
}

// d q = 1 / 4 of the sum of the key chunks' shares, in chunk order -> the q columns of d qkv
__global__ void __launch_bounds__(256) attn_dq_reduce_kernel(const float* __restrict__ part, int KC, int64_t rows,
                                                             uint16_t* __restrict__ dqkv) {
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
  const bf16x4 lo = rl4co_e16::cvt4(0.25f * a0[0], 0.25f * a0[1], 0.25f * a0[2], 0.25f * a0[3]);
  const bf16x4 hi = rl4co_e16::cvt4(0.25f * a1[0], 0.25f * a1[1], 0.25f * a1[2], 0.25f * a1[3]);
  uint16_t* dst = dqkv + row * 3 * kD + col;
  *reinterpret_cast<bf16x4*>(dst) = lo;
  *reinterpret_cast<bf16x4*>(dst + 4) = hi;
}

template <int NT>
int launch_fwd(const void* qkv, int B, int N, void* out, float* lse, hipStream_t s) {
  const int lds = NT * 16 * kKV * 2;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(attn_fwd_kernel<NT>, dim3(B), dim3(kThreads), lds, s, static_cast<const uint16_t*>(qkv), N,
                     static_cast<uint16_t*>(out), lse);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
template <int NT>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, void* dqkv, hipStream_t s) {
  constexpr int work = (NT * 16 * kKH + kBwdWaves * (NT * 16 * kSS + 16 * kQD + 64)) * 2, stage = NT * 16 * kDS * 2;
  constexpr int lds = work > stage ? work : stage;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(attn_bwd_kernel<NT>, dim3(2 * B), dim3(kBwdThreads), lds, s, static_cast<const uint16_t*>(qkv),
                     static_cast<const uint16_t*>(out), static_cast<const uint16_t*>(dout), lse, N, static_cast<uint16_t*>(dqkv));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

}  // namespace

extern "C" int RL4CO_ENTRY(rl4co_attn_flash_lse)(const void* qkv, int B, int N, void* out, float* lse, void* stream);  // am_attn_flash.hip

#if !RL4CO_ELEM_F16
extern "C" int rl4co_attn_max_nodes(void) { return 128; }
extern "C" int rl4co_attn_wide_max_nodes(void) { return 1024; }
#if RL4CO_ATTN_BWD_PROBE >= 3
extern "C" int rl4co_attn_bwd_probe_read(unsigned long long* out, int reset) {
  RL4CO_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_bwd_clk), sizeof(g_attn_bwd_clk)));
  if (reset) {
    const unsigned long long z[4] = {0, 0, 0, 0};
    RL4CO_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_bwd_clk), z, sizeof(z)));
  }
  return RL4CO_OK;
}
#endif
#endif

extern "C" int RL4CO_ENTRY(rl4co_attn_fwd)(const void* qkv, int B, int N, void* out, float* lse, void* stream) {
  RL4CO_REQUIRE(qkv && out && lse && B > 0 && N >= 1 && N <= 1024);
  if (N > 128) return RL4CO_IMPL(rl4co_attn_flash_lse)(qkv, B, N, out, lse, stream);  // keys streamed through LDS, online softmax
  hipStream_t s = rl4co::as_stream(stream);
  const int nt = (N + 15) >> 4;
  if (nt <= 2) return launch_fwd<2>(qkv, B, N, out, lse, s);
  if (nt <= 4) return launch_fwd<4>(qkv, B, N, out, lse, s);
  if (nt <= 7) return launch_fwd<7>(qkv, B, N, out, lse, s);
  return launch_fwd<8>(qkv, B, N, out, lse, s);
}

extern "C" int RL4CO_ENTRY(rl4co_attn_bwd)(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, void* dqkv,
                                           void* stream) {
  RL4CO_REQUIRE(qkv && out && dout && lse && dqkv && B > 0 && B < (1 << 30) && N >= 1 && N <= 128);
  hipStream_t s = rl4co::as_stream(stream);
  const int nt = (N + 15) >> 4;
  if (nt <= 2) return launch_bwd<2>(qkv, out, dout, lse, B, N, dqkv, s);
  if (nt <= 4) return launch_bwd<4>(qkv, out, dout, lse, B, N, dqkv, s);
  if (nt <= 7) return launch_bwd<7>(qkv, out, dout, lse, B, N, dqkv, s);
  return launch_bwd<8>(qkv, out, dout, lse, B, N, dqkv, s);
}

extern "C" int RL4CO_ENTRY(rl4co_attn_bwd_wide)(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N,
                                                void* dqkv, float* dq_partial, void* stream) {
  RL4CO_REQUIRE(qkv && out && dout && lse && dqkv && dq_partial && B > 0 && N >= 1 && N <= 1024);
  const int KC = (N + 16 * kWT - 1) / (16 * kWT);
  RL4CO_REQUIRE((int64_t)B * KC * 2 < (1ll << 31));
  hipStream_t s = rl4co::as_stream(stream);
  constexpr int work = (kWT * 16 * kKH + kBwdWaves * (kWT * 16 * kSS + 16 * kQD + 64)) * 2, stage = kWT * 16 * kDW * 2;
  constexpr int lds = work > stage ? work : stage;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(attn_bwd_wide_kernel, dim3(2 * B * KC), dim3(kBwdThreads), lds, s, static_cast<const uint16_t*>(qkv),
                     static_cast<const uint16_t*>(out), static_cast<const uint16_t*>(dout), lse, B, N, KC,
                     static_cast<uint16_t*>(dqkv), dq_partial);
  RL4CO_HIP_TRY(hipGetLastError());
  const int64_t rows = (int64_t)B * N;
  hipLaunchKernelGGL(attn_dq_reduce_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s, dq_partial, KC, rows,
                     static_cast<uint16_t*>(dqkv));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
