// am_attn_flash.hip — encoder self-attention for graphs too large for one workgroup's LDS (N > 128: BASELINE
// configs[4], CVRP-500), inference. Replaces the head rearrangement + scaled_dot_product_attention of
// nn/attention.py:110-134 (MultiHeadAttention.forward) on the packed projection output qkv [B, N, 384]
// (q | k | v, 8 heads x 16), bf16 -> out [B, N, 128] bf16.
//
// The N x N score matrix never exists: keys / values stream through LDS in blocks of 64 nodes and every query
// keeps a running (max, sum, output) triple — the online softmax of flash attention — in registers:
//   * one 512-thread workgroup = (instance, block of kQT * 16 queries); wave h owns head h, like am_train_attn.hip;
//   * v_mfma_f32_16x16x16_bf16 with the QUERY as accumulator column (lane & 15) and four consecutive KEYS (rows
//     4 g ..) per lane: S^T = K_h Q_h^T chains into O_h^T += V_h^T P^T without a shuffle (P in the accumulator
//     layout is the B operand; V^T through ds_read_b64_tr_b16), the 16 head dims are exactly one MFMA k-step;
//   * a query tile's state is 8 registers (Q fragment 2, max 1, partial sum 1, O^T 4); a wave carries kQT = 4 tiles
//     and each key block's K / V fragments (read from LDS once per block) serve all of them. Four, not eight: at 128
//     registers and 34 KB of LDS TWO workgroups share a CU and cover each other's barriers and key-block round trips
//     (eight tiles: 167 registers, one workgroup per CU, 840 us per layer at 1024 x 501; four: 731 us);
//   * the block maximum needs the four row groups of a query to agree (two permlane swaps); the softmax
//     denominator does not — every lane sums its own keys against the shared running maximum and the four
//     partial sums meet once, after the last block;
//   * the next key block's rows are fetched into registers while the current one is consumed (one LDS buffer,
//     two barriers per block); finished tiles leave through the same LDS buffer as contiguous 16-byte lanes.
// bf16 operands, fp32 accumulation / softmax in the exp2 domain; tolerance-tested against fp32 torch attention.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"
#include "elem16.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kKB = 64;              // keys per LDS block
constexpr int kQT = 4;               // query tiles (of 16) per workgroup pass
constexpr int kKS = 2 * kD + 8;      // LDS row stride of a k | v row (bf16 elements)
constexpr int kOS = kD + 8;          // LDS row stride of the output staging rows
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

// blockIdx -> (instance, query block). Workgroup b lands on XCD b % 8 (observed placement): the query blocks of ONE
// instance re-read the same k | v rows, so they are given consecutive slots of one XCD and share its L2.
__device__ inline void block_map(int b, int B, int QB, int& inst, int& qb) {
  if ((B & 7) != 0) {
    inst = b / QB;
    qb = b % QB;
    return;
  }
  const int xcd = b & 7, k = b >> 3;
  qb = k % QB;
  inst = (k / QB) * 8 + xcd;
}

// PRE (the token-tile encoder, am_encoder.hip: tok16_qkv_kernel): q already carries 1/4 log2 e (folded into the packed
// W_q), so the scores leave the product in the exp2 domain, and `bound` holds per (instance, head) the maxima over the
// nodes of |q_h|^2 and |k_h|^2: when their product stays below kFastBound^2 every |score| does (Cauchy-Schwarz) and the
// softmax needs neither a running maximum nor a rescale — numerators in [2^-48, 2^48], one exp2 per score and nothing
// else (the wide-exponent element type only: fp16 numerators would overflow).
constexpr float kFastBound = 48.0f;
template <bool PRE>
__global__ void __launch_bounds__(kThreads, 4) attn_flash_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ bound, int B,
                                                                 int N, int QB, uint16_t* __restrict__ out, float* __restrict__ lse) {
  constexpr int kLds = kKB * kKS > kQT * 16 * kOS ? kKB * kKS : kQT * 16 * kOS;
  __shared__ __align__(16) elem_t kv[kLds];  // one key block: [64][k 128 | v 128]; later the output staging rows
  const int tid = threadIdx.x, h = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  int inst, qb;
  block_map(blockIdx.x, B, QB, inst, qb);
  const uint16_t* base = qkv + (int64_t)inst * N * 3 * kD;
  const int q0 = qb * kQT * 16;  // first query of this workgroup
  const int nblocks = (N + kKB - 1) / kKB;

  // ---- this wave's query fragments: B operand, lane = query column, four consecutive dims of head h ---------
  bf16x4 qf[kQT];
  float m[kQT], l[kQT];
  f32x4 o[kQT];
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    const int q = min(q0 + 16 * t + tl, N - 1);  // clamped: rows >= N are computed and dropped
    qf[t] = *reinterpret_cast<const bf16x4*>(base + (int64_t)q * 3 * kD + 16 * h + 4 * g);
    m[t] = kNegInf;
    l[t] = 0.0f;
    o[t] = zero4();
  }

  // ---- key blocks: rows prefetched into registers one block ahead --------------------------------------------
  // a block is 64 rows x 512 bytes (k | v) = 2048 16-byte chunks, four per thread
  uint4 pre[4];
  // (the four row addresses are derived from an OPAQUE copy of the thread index inside every call: as loop-carried 64-bit
  // pointers they and the two LDS offsets below did not fit the 128 registers of four waves per SIMD — r05 ISA: 11 scratch
  // accesses in the key-block loop, every reload behind an s_waitcnt vmcnt(0) that also drained the NEXT block's prefetch)
  const uint16_t* kvbase = base + kD;
  auto fetch = [&](int kb) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = t_ + j * kThreads;
      const int row = min(kb * kKB + (c >> 5), N - 1), col = (c & 31) * 8;
      pre[j] = *reinterpret_cast<const uint4*>(kvbase + (uint32_t)(row * (3 * kD) + col));
    }
  };
  auto commit = [&](int kb) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = t_ + j * kThreads;
      const int r = c >> 5, col = (c & 31) * 8;
      const bool ok = kb * kKB + r < N;  // rows past the graph are zero (their scores are masked as well)
      *reinterpret_cast<uint4*>(kv + r * kKS + col) = ok ? pre[j] : make_uint4(0, 0, 0, 0);
    }
  };
  bool fast = false;
  if (PRE && !RL4CO_ELEM_F16 && bound != nullptr) {
    const float b2 = bound[((int64_t)inst * 8 + h) * 2] * bound[((int64_t)inst * 8 + h) * 2 + 1];
    fast = __builtin_amdgcn_readfirstlane(b2 <= kFastBound * kFastBound ? 1 : 0) != 0;
  }
  const int nao = tl * kKS + 4 * g;                          // natural operand read: row lane & 15, columns 4 g ..
  const int tro = (4 * g + (tl >> 2)) * kKS + 4 * (tl & 3);  // transpose read
  fetch(0);
  // Only the LAST key block can hold rows past the graph. `tail` as a run-time condition inside the element loops came out
  // as two selects per score in EVERY block (index compare + tail select: 68 v_cndmask, 16 compares and 16 index adds per
  // block and wave, r05 ISA — a third of the block's VALU work): the last block is peeled off the loop instead, the loop
  // body carries no masks at all.
  auto key_block = [&](int kb, auto tail_c, auto fast_c) {
    constexpr bool TAIL = decltype(tail_c)::value, FAST = decltype(fast_c)::value;
    if (kb > 0) __syncthreads();  // every wave is done reading the previous block
    commit(kb);
    __syncthreads();
    // Both paths DEFINE the four registers: with the plain "if (..) fetch(..)" the merge of fetched / kept values made the
    // compiler load into temporaries and copy them right behind the loads — an s_waitcnt vmcnt(2) that put the block's
    // global round trip back on the critical path; an unconditional fetch gets sunk to the top of the next iteration
    if (!TAIL) {
      fetch(kb + 1);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) pre[j] = make_uint4(0, 0, 0, 0);
    }
    bf16x4 kf[4], vf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kf[j] = lds_b64(kv + 16 * j * kKS + 16 * h + nao);
      vf[j] = lds_tr(kv + 16 * j * kKS + kD + 16 * h + tro);
    }
      if constexpr (FAST) {
        // bounded scores: p = exp2(s) tile by tile — no maximum, no subtraction, no rescale of the accumulators
#pragma unroll
        for (int t = 0; t < kQT; ++t) {
          float ls = 0.0f;
          f32x4 acc = o[t];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 sj = mfma16(kf[j], qf[t], zero4());
            float p[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              p[rr] = __builtin_amdgcn_exp2f(sj[rr]);
              if (TAIL) p[rr] = (kb * kKB + 16 * j + 4 * g + rr < N) ? p[rr] : 0.0f;
              ls += p[rr];
            }
            acc = mfma16(vf[j], rl4co_e16::cvt4(p[0], p[1], p[2], p[3]), acc);
          }
          o[t] = acc;
          l[t] += ls;
        }
        return;
      }
#pragma unroll
      for (int t = 0; t < kQT; ++t) {
        f32x4 s[4];
        float bm = kNegInf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] = mfma16(kf[j], qf[t], zero4());
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float v = PRE ? s[j][rr] : s[j][rr] * kScale;
            if (TAIL) v = (kb * kKB + 16 * j + 4 * g + rr < N) ? v : kNegInf;
            s[j][rr] = v;
            bm = fmaxf(bm, v);
          }
        }
        bm = rl4co::bfly_max<16, 64>(bm);  // the four row groups of a query agree on the block maximum
        const float mn = fmaxf(m[t], bm);   // finite: every block holds at least one real key
        const float alpha = __builtin_amdgcn_exp2f(m[t] - mn);  // exp2(-inf) = 0 on the first block
        m[t] = mn;
        float ls = 0.0f;
        f32x4 acc = {o[t][0] * alpha, o[t][1] * alpha, o[t][2] * alpha, o[t][3] * alpha};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float p[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            p[rr] = __builtin_amdgcn_exp2f(s[j][rr] - mn);
            ls += p[rr];
          }
          acc = mfma16(vf[j], rl4co_e16::cvt4(p[0], p[1], p[2], p[3]), acc);
        }
        o[t] = acc;
        l[t] = fmaf(l[t], alpha, ls);  // this lane's keys only; the row groups meet after the last block
      }
  };
  // (the softmax path is chosen ONCE, outside the loop: with both bodies inside it the loop did not fit its 128 registers)
  auto key_blocks = [&](auto fast_c) {
    for (int kb = 0; kb + 1 < nblocks; ++kb) key_block(kb, std::false_type{}, fast_c);
    key_block(nblocks - 1, std::true_type{}, fast_c);
  };
  if (fast) key_blocks(std::true_type{});
  else key_blocks(std::false_type{});
  __syncthreads();  // the key block is dead: its buffer stages the output rows [kQT * 16][128]

  // ---- normalise, stage, leave as contiguous 16-byte lanes ------------------------------------------------------
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    const float lt = rl4co::bfly_sum<16, 64>(l[t]);
    const float inv = __builtin_amdgcn_rcpf(lt);
    // training (rl4co_attn_fwd beyond one workgroup's nodes): the log-sum-exp of the scaled scores, log2 domain — what
    // am_train_attn.hip's backward kernels rebuild the probabilities from
    if (!PRE && lse != nullptr && g == 0 && q0 + 16 * t + tl < N)
      lse[((int64_t)inst * kWaves + h) * N + q0 + 16 * t + tl] = m[t] + __builtin_amdgcn_logf(lt);
    *reinterpret_cast<bf16x4*>(kv + (16 * t + tl) * kOS + 16 * h + 4 * g) =
        rl4co_e16::cvt4(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv);
  }
  __syncthreads();
  const int rows = min(kQT * 16, N - q0);
  uint16_t* dst = out + ((int64_t)inst * N + q0) * kD;
  for (int c = tid; c < rows * 16; c += kThreads) {
    const int row = c >> 4, col = (c & 15) * 8;
    *reinterpret_cast<uint4*>(dst + (int64_t)row * kD + col) = *reinterpret_cast<const uint4*>(kv + row * kOS + col);
  }
}

}  // namespace

extern "C" int RL4CO_ENTRY(rl4co_attn_flash)(const void* qkv, int B, int N, void* out, void* stream) {
  RL4CO_REQUIRE(qkv && out && B > 0 && N >= 1 && N <= 65536);
  const int QB = (N + kQT * 16 - 1) / (kQT * 16);
  RL4CO_REQUIRE((int64_t)B * QB < (1ll << 31));
  hipLaunchKernelGGL(attn_flash_kernel<false>, dim3(B * QB), dim3(kThreads), 0, rl4co::as_stream(stream),
                     static_cast<const uint16_t*>(qkv), static_cast<const float*>(nullptr), B, N, QB, static_cast<uint16_t*>(out),
                     static_cast<float*>(nullptr));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// the same launch with the log-sum-exp kept [B, 8, N]: rl4co_attn_fwd's path beyond rl4co_attn_max_nodes() (am_train_attn.hip)
extern "C" int RL4CO_ENTRY(rl4co_attn_flash_lse)(const void* qkv, int B, int N, void* out, float* lse, void* stream) {
  RL4CO_REQUIRE(qkv && out && lse && B > 0 && N >= 1 && N <= 65536);
  const int QB = (N + kQT * 16 - 1) / (kQT * 16);
  RL4CO_REQUIRE((int64_t)B * QB < (1ll << 31));
  hipLaunchKernelGGL(attn_flash_kernel<false>, dim3(B * QB), dim3(kThreads), 0, rl4co::as_stream(stream),
                     static_cast<const uint16_t*>(qkv), static_cast<const float*>(nullptr), B, N, QB, static_cast<uint16_t*>(out), lse);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_attn_flash_pre)(const void* qkv, const float* bound, int B, int N, void* out, void* stream) {
  RL4CO_REQUIRE(qkv && out && B > 0 && N >= 1 && N <= 65536);
  const int QB = (N + kQT * 16 - 1) / (kQT * 16);
  RL4CO_REQUIRE((int64_t)B * QB < (1ll << 31));
  hipLaunchKernelGGL(attn_flash_kernel<true>, dim3(B * QB), dim3(kThreads), 0, rl4co::as_stream(stream),
                     static_cast<const uint16_t*>(qkv), bound, B, N, QB, static_cast<uint16_t*>(out), static_cast<float*>(nullptr));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
