// am_encoder.hip — fused AttentionModel encoder + decoder-cache fold on the gfx950 matrix cores.
//
// Replaces, for inference rollouts (eval-mode normalisation), per instance and in ONE launch:
//   TSPInitEmbedding / VRPInitEmbedding     models/nn/env_embeddings/init.py:55-68,115-136
//   GraphAttentionNetwork (L layers of      models/nn/graph/attnnet.py:16-106
//     x + MHA(x) -> Norm -> x + MLP(x) -> Norm)   nn/attention.py:110-134, nn/ops.py:30-54, nn/mlp.py:52-61
//   AttentionModelDecoder._precompute_cache models/zoo/am/decoder.py:201-228 (folded form, cache.py)
// The reference runs ~45 ATen kernels that round-trip [B*N,128..512] activations through HBM
// (profiles/r01_run2: 5.2 ms at TSP-100 x 4096 in bf16). Here one 256-thread workgroup owns one
// instance: the residual stream lives in LDS (bf16, like torch autocast), every GEMM runs on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, weights stream from L2 in pre-packed fragment
// order, and the only HBM traffic is the coordinates in and the folded cache planes out.
//
// Work split (4 waves, N <= 128 tokens = TT tiles of 32):
//   * "transposed" GEMMs  Out^T[dim][token] = W[dim][k] . In^T[k][token]: A = weight fragment
//     (one 16-B global load per lane, reused over the TT token tiles), B = activation rows read
//     from LDS ([token][k], ds_read_b128). Wave w owns output-dim tile w (32 dims) of every GEMM,
//     so each weight fragment is fetched by exactly one wave of the workgroup.
//   * attention is wave-private: wave w computes Q, K (transposed form) and V (plain form) for
//     head pair (2w, 2w+1) of all tokens and keeps them in registers as MFMA fragments:
//       S^T[key][query] = K . Q^T     (A = K regs, B = Q regs; K-dim 16 = one head, no waste)
//       softmax over keys = in-lane over the accumulator registers + one cross-half exchange
//       O^T[dim][query]  = V^T . P^T  (A = V regs, B = P regs)
//     All four operands are accumulator-layout registers (lane = row/col index, registers = k),
//     so no LDS transpose is needed; the k-slot permutation of the layout is the same on both
//     operands of each product and therefore cancels.
//   * exchanges through LDS only where a GEMM needs all 128 input dims produced by other waves:
//     attention output -> out-proj, norm1 output -> FFN1, FFN hidden chunks -> FFN2.
//   * 70 KB of LDS and <= 256 registers per wave: two instances per CU, so one's softmax / norm /
//     conversion VALU phases overlap the other's MFMA phases.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kFF = 512;
constexpr int kRS = kD + 8;  // LDS row stride (bf16 elements): 272 B rows spread the banks
constexpr int kThreads = 256;

// The kernel is generic in the 16-bit ELEMENT type E of its MFMA operands / LDS residual stream:
//   __bf16    torch.autocast(bfloat16) ("bf16-mixed")                      v_mfma_f32_32x32x16_bf16
//   _Float16  torch.autocast(float16), the reference's DEFAULT "16-mixed"  v_mfma_f32_32x32x16_f16
//             (rl4co/utils/trainer.py:57)
// Same fragment layouts, same rate, fp32 accumulation either way; (E)float conversions are the hardware's
// round-to-nearest-even converts. What differs is the RANGE: see kFastBound.
template <typename E> using vec8 = E __attribute__((ext_vector_type(8)));
template <typename E> using vec4 = E __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline f32x16 mfma(const vec8<__bf16>& a, const vec8<__bf16>& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ inline f32x16 mfma(const vec8<_Float16>& a, const vec8<_Float16>& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// accumulator layout of the 32x32 MFMA: register r of lane (l31, hi) is row rowmap(r, hi), col l31
__device__ inline int rowmap(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// bias of the wave's 32-dim output tile in accumulator layout (transposed form: register r <-> dim rowmap(r, hi))
// `bias` points into LDS (the layer's bqkv | b1, staged one layer ahead): the tile is the C operand of the GEMM's first
// MFMA, so it must be there when the call starts — 16 global loads would put an L2 round trip in front of every call;
// registers 4 c .. 4 c + 3 are four consecutive dims: four broadcast 16-byte LDS reads
__device__ inline f32x16 bias_tile(const float* bias, int dim0, int hi) {
  f32x16 t;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(bias + dim0 + 8 * c + 4 * hi);
    t[4 * c] = v.x;
    t[4 * c + 1] = v.y;
    t[4 * c + 2] = v.z;
    t[4 * c + 3] = v.w;
  }
  return t;
}
constexpr int kBiasFloats = 3 * kD + kFF;  // per layer: bqkv [384] | b1 [512]
constexpr int kNormFloats = 4 * kD;         // per layer: n1 scale | n1 shift | n2 scale | n2 shift
// xs | ys | meanv | biases and norm constants of two layers (this one's and the next one's): 81 408 B, two workgroups per CU
constexpr int kEncLds = 2 * 128 * kRS * 2 + kD * 4 + 2 * kBiasFloats * 4 + 2 * kNormFloats * 4;
static_assert(2 * kEncLds <= 160 * 1024, "am_encoder_kernel is built for two workgroups per CU");

// max(x, 0) in ONE instruction: fmaxf under the kernels' IEEE mode is v_max_f32 x, x, x (quieting a signalling NaN) followed
// by v_max_f32 0, x — two VALU operations per hidden unit, 768 per wave and instance in the fused kernel (r05 ISA). The
// median of (x, 0, +inf) is the same value for every x that is not a NaN; `inf` arrives through an opaque copy (the
// compiler folds the literal form back into the max pair), and unlike an inline-asm v_max the instruction stays visible to
// the hazard recogniser (the operands are MFMA results: wait states are software-managed).
__device__ inline float opaque_inf() {
  float v = __builtin_huge_valf();
  asm volatile("" : "+s"(v));
  return v;
}
__device__ inline float relu(float x, float inf) { return __builtin_amdgcn_fmed3f(x, 0.0f, inf); }

__device__ inline f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

// registers 8u..8u+7 of an accumulator as one bf16 MFMA operand fragment
template <typename E>
__device__ inline vec8<E> frag_from_acc(const f32x16& c, int u) {
  vec8<E> f;
#pragma unroll
  for (int s = 0; s < 8; ++s) f[s] = (E)c[8 * u + s];
  return f;
}

// packed weight fragment: [tile][kstep][64 lanes][8] bf16
template <typename E>
__device__ inline vec8<E> load_w(const E* packed, int ksteps, int tile, int ks, int lane) {
  return *reinterpret_cast<const vec8<E>*>(packed + (((int64_t)tile * ksteps + ks) * 64 + lane) * 8);
}

// activation fragment from LDS rows [token][k]: lane reads 8 contiguous k of its token row
template <typename E>
__device__ inline vec8<E> load_x(const E* xs, int tt, int ks, int l31, int hi) {
  return *reinterpret_cast<const vec8<E>*>(xs + (32 * tt + l31) * kRS + 16 * ks + 8 * hi);
}

// The 8 weight fragments of one GEMM call (8 x 1 KiB per wave, streamed from L2).
template <typename E>
__device__ inline void load_wfrags(vec8<E> (&wf)[8], const E* packed, int ksteps_total, int tile, int k0, int lane) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) wf[ks] = load_w(packed, ksteps_total, tile, k0 + ks, lane);
}

// Out^T tile (32 dims x 32*TT tokens) += W . X^T over 8 ksteps with the weight fragments `wf`
// already requested by the caller. An L2 round trip is 2-4 K cycles and a call is only 32 MFMAs
// (1 K cycles), so fetching the fragments at the top of the call that uses them leaves the matrix
// pipe idle most of the time: instead every call hands over — as soon as fragment ks has fed its
// last MFMA, the same registers receive fragment ks of the NEXT GEMM (nxt_*; nullptr: none), whose
// latency then hides under the rest of this call and whatever epilogue separates the two. The LDS
// activation fragments are double-buffered one kstep ahead.
// INIT: the first k-step takes `cinit` as its C operand (an MFMA's C need not be its D) — the bias tile of the GEMM,
// shared by the TT token tiles, enters the accumulators for free instead of through 16 TT adds in the epilogue.
// NEXT: 1 = the hand-over always happens (`nxt_packed` is a valid fragment stream), 0 = never, -1 = decided at run time from
// `nxt_packed`. The run-time form puts a branch behind every k-step (eight basic blocks per call, r05 ISA: s_cbranch +
// s_and per k-step, the scheduler cannot move anything across them); where the last call of a chain has nothing to fetch the
// callers hand in a dummy stream instead (8 KB per wave from L2, unused).
template <int TT, bool W_IS_A = true, bool INIT = true, int NEXT = -1, typename E>
__device__ inline void gemm_t(f32x16 (&acc)[TT], vec8<E> (&wf)[8], const E* xs, int lane, const E* nxt_packed,
                              int nxt_ksteps_total, int nxt_tile, int nxt_k0, const f32x16& cinit) {
  const int l31 = lane & 31, hi = lane >> 5;
  // Every MFMA takes one 1 KiB activation fragment from LDS. With the fragments of k-step ks + 1 requested while k-step
  // ks computes (TT MFMAs = 128 cycles at TT = 4), the eight waves of a CU keep the LDS queue deep enough that the
  // data is NOT back in time: replacing these reads by loop-invariant ones cut the kernel from 1.12 to 0.63 ms
  // (tools/enc_probe.sh, r02) — the kernel was waiting on LDS latency, not on issue slots or the matrix pipe. The
  // fragments are therefore requested TWO k-steps ahead (three rotating register sets).
  vec8<E> x[3][TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    x[0][tt] = load_x(xs, tt, 0, l31, hi);
    x[1][tt] = load_x(xs, tt, 1, l31, hi);
  }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    if (ks + 2 < 8) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) x[(ks + 2) % 3][tt] = load_x(xs, tt, ks + 2, l31, hi);
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const f32x16& c = (INIT && ks == 0) ? cinit : acc[tt];
      acc[tt] = W_IS_A ? mfma(wf[ks], x[ks % 3][tt], c) : mfma(x[ks % 3][tt], wf[ks], c);
    }
    if (NEXT == 1 || (NEXT < 0 && nxt_packed)) wf[ks] = load_w(nxt_packed, nxt_ksteps_total, nxt_tile, nxt_k0 + ks, lane);
    // straight-line calls: keep the k-steps in this order (left alone, the scheduler pulls the LDS reads back to just in
    // front of their MFMA — less register pressure on paper, the read latency exposed on every product)
    if (NEXT >= 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// write an Out^T accumulator tile to LDS rows [token][dim]: 4 consecutive dims per 8-byte store.
// `nvalid` (fused kernel): token rows >= nvalid — the padding of the last tile, TSP-100: 28 of its 32 rows — are written as
// ZEROS. No valid row ever reads them (the GEMMs are row-wise, attention masks padding keys, the norm statistics count
// valid tokens only), but every GEMM drags them through the matrix cores as B operands, and the kernel runs against the
// package POWER limit (r05, tools/enc_power_probe.py: the same binary on zeroed data is 16 % faster, 2.38 instead of 2.05
// GHz): a zero operand row toggles nothing. Eight v_cndmask per store of the last tile.
template <int TT, typename E>
__device__ inline void store_t(E* ys, const f32x16 (&acc)[TT], int dim0, int lane, int nvalid = 32 * TT) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const bool live = tt + 1 < TT || 32 * tt + l31 < nvalid;  // (TT = ceil(N / 32): only the last tile holds padding)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      vec4<E> v;
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = (E)acc[tt][4 * c + s];
      uint2 u = __builtin_bit_cast(uint2, v);
      if (tt + 1 == TT) u = live ? u : make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(ys + (32 * tt + l31) * kRS + dim0 + 8 * c + 4 * hi) = u;
    }
  }
}

// Out^T accumulator tile -> global rows [token][dim] of `row_stride` elements (tokens < N): 8-byte stores, the two halves of
// a wave side by side (16 bytes per token row and 8-dim group; the rows' other pieces come from the other waves and meet
// in the XCD's write-back L2)
// Two 8-byte pieces per lane (dims 8 c + 4 hi .. of groups c = c0, c0 + 1) -> ONE 16-byte piece per lane: the lower half
// of the wave ends up with all eight dims of group c0, the upper half with those of group c0 + 1 (v_permlane32_swap on the
// two dwords of each piece: no LDS). Returns the 16 bytes to store at dims 8 (c0 + hi) .. 8 (c0 + hi) + 7.
template <typename E>
__device__ inline uint4 pair16(const vec4<E>& p0, const vec4<E>& p1) {
  const uint2 a = __builtin_bit_cast(uint2, p0), b = __builtin_bit_cast(uint2, p1);
  const auto x = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);  // [0]: {lower: a.lower, upper: b.lower}; [1]: {a.upper, b.upper}
  const auto y = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
  return make_uint4(x[0], y[0], x[1], y[1]);
}

// Out^T accumulator tile -> global rows [token][dim] of `row_stride` elements (tokens < N): 16-byte stores (pair16), a token
// row's 32 dims of this wave in two of them; the rows' other pieces come from the other waves and meet in the XCD's
// write-back L2 (8-byte stores straight from the accumulator layout cost twice as much per byte: tools/probes/train_fwd_probe.sh)
template <int TT, typename E>
__device__ inline void save_t(E* dst, int64_t row_stride, const f32x16 (&acc)[TT], int dim0, int N, int lane) {
  asm volatile("" : "+v"(lane));  // (row offsets derived per call: kept from the top of the layer they are spilled, and a scratch reload waits with vmcnt(0))
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    vec4<E> v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) v[c][s] = (E)acc[tt][4 * c + s];
    const uint4 lo = pair16<E>(v[0], v[1]), up = pair16<E>(v[2], v[3]);
    if (32 * tt + l31 < N) {
      E* row = dst + (int64_t)(32 * tt + l31) * row_stride + dim0 + 8 * hi;
      *reinterpret_cast<uint4*>(row) = lo;
      *reinterpret_cast<uint4*>(row + 16) = up;
    }
  }
}

template <typename E>
struct LayerPtrs {
  const E *wqkv, *wo, *w1, *w2;
  const float *bqkv, *b1, *n1a, *n1b, *n2a, *n2b;
};

// residual + bias + normalisation epilogue for the wave's 32-dim tile; result back into xs (bf16)
// LAYER (normalization="layer", nn/ops.py:48-51): ONE mean and ONE unbiased variance over all N x 128 values of the
// instance and no affine. The GEMM's bias does not cancel there: it arrives in `nb` and is added BEFORE the statistics;
// the four waves' partial sums meet in `red` (2 x 4 floats of LDS; two workgroup barriers inside).
template <int TT, bool LAYER = false, typename E>
__device__ inline void residual_norm(E* xs, f32x16 (&y)[TT], int dim0, const float* na,
                                     const float* nb, int norm, int N, int lane, float* red = nullptr, int w = 0) {
  const int l31 = lane & 31, hi = lane >> 5;
  float ga[16], be[16];  // (the GEMM's bias is folded into `nb` on the host, or cancels: instance norm)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = dim0 + rowmap(r, hi);
    ga[r] = na[d];
    be[r] = nb[d];
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const vec4<E> x = *reinterpret_cast<const vec4<E>*>(xs + (32 * tt + l31) * kRS + dim0 + 8 * c + 4 * hi);
#pragma unroll
      for (int s = 0; s < 4; ++s) y[tt][4 * c + s] = (float)x[s] + y[tt][4 * c + s];
    }
  }
  if constexpr (LAYER) {
    float s = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      float t16 = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        y[tt][r] += be[r];
        t16 += y[tt][r];
      }
      s += (32 * tt + l31 < N) ? t16 : 0.0f;
    }
    s = rl4co::bfly_sum<1, 64>(s);
    if (lane == 0) red[w] = s;
    rl4co::lds_barrier();
    const float cnt = (float)(N * kD);
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / cnt;
    float v = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      float t16 = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = y[tt][r] - mean;
        t16 = fmaf(d, d, t16);
      }
      v += (32 * tt + l31 < N) ? t16 : 0.0f;
    }
    v = rl4co::bfly_sum<1, 64>(v);
    if (lane == 0) red[4 + w] = v;
    rl4co::lds_barrier();
    const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (cnt - 1.0f) + 1e-5f);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[tt][r] = (y[tt][r] - mean) * rstd;
  } else if (norm == 1) {
    // instance norm (nn/ops.py:46-47, POMO): per (instance, channel) statistics over the N tokens
    const float inv_n = 1.0f / (float)N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = 0.0f;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) s += (32 * tt + l31 < N) ? y[tt][r] : 0.0f;
      s = rl4co::bfly_sum<1, 32>(s);
      const float mean = s * inv_n;
      float v = 0.0f;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float d = y[tt][r] - mean;
        v += (32 * tt + l31 < N) ? d * d : 0.0f;
      }
      v = rl4co::bfly_sum<1, 32>(v);
      const float rstd = rsqrtf(v * inv_n + 1e-5f);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) y[tt][r] = (y[tt][r] - mean) * rstd * ga[r] + be[r];
    }
  } else {
    // batch norm in eval mode = per-channel affine folded on the host (scale, shift)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[tt][r] = fmaf(y[tt][r], ga[r], be[r]);
  }
  store_t<TT>(xs, y, dim0, lane, N);
}

// What the TRAINING forward (am_encoder_kernel<.., TRAIN = true>, rl4co_am_encoder_train_fwd) keeps for the backward
// kernels of csrc/am_train_ops.hip / am_train_attn.hip, per layer l — the tensors rl4co_amd/train_ops.py's per-op path
// saves, in the layouts those kernels read (nn/graph/attnnet.py:16-55, instance norm: zoo/pomo/model.py:59-63).
template <typename E>
struct TrainSave {
  const E* x0;   // [B,N,128] the encoder's input (init embedding)
  E* out;        // [L,B,N,128] layer outputs (out[l - 1] is layer l's input)
  E* qkv;        // [L,B,N,384] q | k | v, unscaled
  E* att;        // [L,B,N,128] attention output (before out_proj)
  E* y1;         // [L,B,N,128] x + attention branch, pre-norm (the branch's constant bias cancels in the instance norm)
  E* x1;         // [L,B,N,128] norm1 output = the MLP block's input
  E* h;          // [L,B,N,512] relu(x1 W1^T + b1)
  E* y2;         // [L,B,N,128] x1 + MLP branch, pre-norm
  float* lse;    // [L,B,8,N] log-sum-exp of the scaled scores, log2 domain (am_train_attn.hip's convention)
  float* stats;  // [L,4,B,128] mean1, rstd1, mean2, rstd2
};
constexpr float kTrainScale = 0.25f * 1.44269504088896341f;  // 1/sqrt(16) in the exp2 domain, applied to the scores

// Training epilogue: y = x + branch rounded to the element type (the value the backward re-reads), instance statistics of
// the ROUNDED values (as csrc/am_train_ops.hip: skip_inorm_fwd_kernel), both saved; normalised output back into xs.
template <int TT, typename E>
__device__ inline void residual_norm_train(E* xs, f32x16 (&y)[TT], int dim0, const float* na, const float* nb, int N, int lane,
                                           E* y_out, float* mean_out, float* rstd_out) {
  asm volatile("" : "+v"(lane));  // (as in save_t)
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    vec4<E> yr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const vec4<E> x = *reinterpret_cast<const vec4<E>*>(xs + (32 * tt + l31) * kRS + dim0 + 8 * c + 4 * hi);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        yr[c][s] = (E)((float)x[s] + y[tt][4 * c + s]);
        y[tt][4 * c + s] = (float)yr[c][s];
      }
    }
    const uint4 lo = pair16<E>(yr[0], yr[1]), up = pair16<E>(yr[2], yr[3]);
    if (32 * tt + l31 < N) {
      E* row = y_out + (int64_t)(32 * tt + l31) * kD + dim0 + 8 * hi;
      *reinterpret_cast<uint4*>(row) = lo;
      *reinterpret_cast<uint4*>(row + 16) = up;
    }
  }
  const float inv_n = 1.0f / (float)N;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = dim0 + rowmap(r, hi);
    float s = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) s += (32 * tt + l31 < N) ? y[tt][r] : 0.0f;
    s = rl4co::bfly_sum<1, 32>(s);
    const float mean = s * inv_n;
    float v = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const float dd = y[tt][r] - mean;
      v += (32 * tt + l31 < N) ? dd * dd : 0.0f;
    }
    v = rl4co::bfly_sum<1, 32>(v);
    const float rstd = rsqrtf(v * inv_n + 1e-5f);
    if (l31 == 0) {
      mean_out[d] = mean;
      rstd_out[d] = rstd;
    }
    const float ga = na[d], be = nb[d];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) y[tt][r] = fmaf((y[tt][r] - mean) * rstd, ga, be);
  }
  store_t<TT>(xs, y, dim0, lane, N);
}

// rows 0 .. rows - 1 (1 <= rows <= 32 TT) of an LDS tile [.][kRS] -> global rows of `row_stride` elements, 16-byte lanes.
// A FIXED number of UNCONDITIONAL stores per thread (pieces past the last row repeat that row's: the same bytes to the same
// address): behind a loop with a run-time trip count the compiler cannot tell how many stores are younger than the weight
// fragment the next GEMM call waits for — loads and stores share vmcnt — and emits s_waitcnt vmcnt(0): every such call then
// started by draining all of these stores, an HBM write round trip on the workgroup's chain (r05 ISA).
template <int TT, typename E>
__device__ inline void rows_out(const E* xs, E* dst, int64_t row_stride, int rows, int tid) {
  asm volatile("" : "+v"(tid));  // the 2 TT piece addresses are derived HERE, per call (shared between the calls of a layer they are 16 - 32 more live registers)
#pragma unroll
  for (int j = 0; j < 2 * TT; ++j) {
    const int i = tid + kThreads * j, row = min(i >> 4, rows - 1), c16 = i & 15;
    *reinterpret_cast<uint4*>(dst + (int64_t)row * row_stride + 8 * c16) = *reinterpret_cast<const uint4*>(xs + row * kRS + 8 * c16);
    if (j & 1) __builtin_amdgcn_sched_barrier(0);  // two pieces in flight at a time (all of them: 32 registers the training forward does not have)
  }
}

// Init embedding of rows n0 .. n0 + rows_pad - 1 of instance b into xs (rows past N zeroed); contains one __syncthreads().
// The instance's coordinates (and demands ...) are staged in LDS (`lsh`, 6 N floats at most) first: read per token from
// global memory they are a chain of dependent L2 round trips (28 K cycles per instance, measured).
template <typename E>
__device__ inline void init_embed_rows16(const rl4co_am_encoder_args& a, int b, int n0, int rows_pad, E* xs, float* lsh, int tid) {
  const int N = a.N;
  const float* loc = a.locs + (int64_t)b * N * 2;
  const bool pdp = a.env == RL4CO_ENV_PDP;  // depot | pickups (x, y, x', y' of the delivery) | deliveries, init.py:335-360
  const bool cvrp = a.env == RL4CO_ENV_CVRP;
  const bool depot = cvrp || pdp;
  const int half = (N - 1) / 2;
  for (int i = tid; i < 2 * N; i += kThreads) lsh[i] = loc[i];
  const bool four = cvrp && a.feature4 != nullptr;  // PCTSP: (x, y, expected prize, penalty), init.py:283-312
  if (cvrp)
    for (int i = tid; i < N - 1; i += kThreads) lsh[2 * N + 1 + i] = a.demand[(int64_t)b * (N - 1) + i];
  const bool six = four && a.feature5 != nullptr && a.feature6 != nullptr;  // CVRPTW: + tw start, tw end, service time
  if (four)
    for (int i = tid; i < N - 1; i += kThreads) lsh[3 * N + 1 + i] = a.feature4[(int64_t)b * (N - 1) + i];
  if (six)
    for (int i = tid; i < N - 1; i += kThreads) {
      lsh[4 * N + 1 + i] = a.feature5[(int64_t)b * (N - 1) + i];
      lsh[5 * N + 1 + i] = a.feature6[(int64_t)b * (N - 1) + i];
    }
  // thread = four consecutive channels (tid & 31) x one of eight token groups: every token row of the residual stream
  // receives an 8-byte LDS store per thread (a thread per channel stored 2 bytes per token: 64 dependent
  // ds_write_b16 per thread, part of the 44 K cycles this phase took of an instance's 285 K)
  const int d0 = 4 * (tid & 31);
  const int ws = six ? 6 : ((four || pdp) ? 4 : (cvrp ? 3 : 2));  // row stride of w_init
  float wq[4][6], bq[4], dq[4][2], dbq[4], eq[4][2], ebq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int f = 0; f < 6; ++f) wq[c][f] = f < ws ? a.w_init[ws * (d0 + c) + f] : 0.0f;
    bq[c] = a.b_init[d0 + c];
    dq[c][0] = depot ? a.w_depot[2 * (d0 + c)] : 0.0f;
    dq[c][1] = depot ? a.w_depot[2 * (d0 + c) + 1] : 0.0f;
    dbq[c] = depot ? a.b_depot[d0 + c] : 0.0f;
    eq[c][0] = pdp ? a.w_extra[2 * (d0 + c)] : 0.0f;
    eq[c][1] = pdp ? a.w_extra[2 * (d0 + c) + 1] : 0.0f;
    ebq[c] = pdp ? a.b_extra[d0 + c] : 0.0f;
  }
  __syncthreads();
  for (int row = tid >> 5; row < rows_pad; row += kThreads / 32) {
    const int tok = n0 + row;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (tok < N) {
      const float x = lsh[2 * tok], y = lsh[2 * tok + 1];
      // feature vector of this token in the order its embedding's weight rows take them (unused slots: weight 0)
      float f2 = 0.0f, f3 = 0.0f, f4 = 0.0f, f5 = 0.0f;
      if (pdp && tok <= half) {
        f2 = lsh[2 * (tok + half)];
        f3 = lsh[2 * (tok + half) + 1];
      } else if (cvrp) {
        f2 = lsh[2 * N + tok];
        if (four) f3 = lsh[3 * N + tok];
        if (six) {
          f4 = lsh[4 * N + tok];
          f5 = lsh[5 * N + tok];
        }
      }
      const bool is_depot = depot && tok == 0, is_delivery = pdp && tok > half;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float r;
        if (is_depot) r = fmaf(dq[c][1], y, fmaf(dq[c][0], x, dbq[c]));
        else if (is_delivery) r = fmaf(eq[c][1], y, fmaf(eq[c][0], x, ebq[c]));
        else {
          r = fmaf(wq[c][1], y, fmaf(wq[c][0], x, bq[c]));  // same fma order as the per-feature chains before
          if (ws > 2) r = fmaf(wq[c][2], f2, r);
          if (ws > 3) r = fmaf(wq[c][3], f3, r);
          if (ws > 4) r = fmaf(wq[c][5], f5, fmaf(wq[c][4], f4, r));
        }
        v[c] = r;
      }
    }
    vec4<E> pk;
#pragma unroll
    for (int c = 0; c < 4; ++c) pk[c] = (E)v[c];
    *reinterpret_cast<vec4<E>*>(xs + row * kRS + d0) = pk;
  }
}


// Two workgroups per CU (70 KB LDS, <= 256 registers): while one instance sits in a VALU-heavy
// phase (softmax, norms, conversions) the other one's waves keep the matrix pipe busy.
// |score| bound below which exp2 needs no max subtraction: scores in [-48, 48] (log2 domain) keep every softmax
// numerator in [2^-48, 2^48] and a row sum below 2^55 — far inside fp32 / bf16 range, same relative precision
// fp16 has neither range (numerators would have to stay inside [2^-14, 2^16]): it always takes the exact path, whose
// numerators exp2(s - max) lie in (0, 1] — what underflows there is below 2^-24 of the row's largest term.
constexpr float kFastBound = 48.0f;
template <typename E> constexpr bool kWideRange = true;
template <> constexpr bool kWideRange<_Float16> = false;

// squared norms of the two heads' 16-dim slices of one token column, from a transposed-form accumulator tile
// (register r of lane (l31, hi) is dim rowmap(r, hi) of token l31: a head = 8 registers here + 8 in the other half)
__device__ inline void head_sqnorms(const f32x16& c, float& h0, float& h1) {
  float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    s0 = fmaf(c[r], c[r], s0);
    s1 = fmaf(c[8 + r], c[8 + r], s1);
  }
  h0 = s0 + rl4co::bfly_f<32>(s0);
  h1 = s1 + rl4co::bfly_f<32>(s1);
}
__device__ inline float wave_max32(float v) {  // maximum over the 32 token columns (both halves hold the same value)
  return rl4co::bfly_max<1, 32>(v);
}

// VR4: valid 4-register groups of the LAST key tile (keys 32 (TT-1) ..): ceil((N - 32 (TT-1)) / 8). Registers beyond
// them are padding keys in every lane — their exps, conversions and (from 8 registers up) the second value product
// are dropped at compile time (TSP-100: four valid keys in the fourth tile, 12 of 16 registers gone).
// TRAIN: the training forward of an instance-norm encoder (POMO) — the same layer body fed with the init embedding, scores
// scaled in the kernel (the saved q is the q the products used), every tensor of TrainSave written on the way, no fold.
// (Barriers are LDS-only — `s_waitcnt lgkmcnt(0); s_barrier` — not __syncthreads(): every hand-over between the waves is
// LDS data, and __syncthreads()'s vmcnt(0) would make each of them wait for the acknowledgement of the global stores issued
// before it — neutral for inference (0.965 vs 0.965 ms, tools/ab_encoder.sh), but the TRAIN variant writes 12 passes per layer.)
template <typename E, int TT, int VR4, bool TRAIN = false, bool LAYER = false>
__global__ void __launch_bounds__(kThreads, 2) am_encoder_kernel(const rl4co_am_encoder_args a, const TrainSave<E> ts) {
  using bf16x8 = vec8<E>;  // (historic names: the 16-bit operand fragments of whichever element type E is)
  using bf16x4 = vec4<E>;
  extern __shared__ __align__(16) unsigned char smem[];
  E* xs = reinterpret_cast<E*>(smem);  // residual stream [128][kRS]
  E* ys = xs + 128 * kRS;              // Q^T (wave-private columns) -> attention output -> FFN hidden chunk
  float* meanv = reinterpret_cast<float*>(ys + 128 * kRS);  // [128]
  float* bl2 = meanv + kD;                                   // [2][kBiasFloats] bqkv | b1 by layer parity (see bias_tile)
  float* nl2 = bl2 + 2 * kBiasFloats;                        // [2][kNormFloats] n1 scale | n1 shift | n2 scale | n2 shift
  // A layer's constants travel global -> registers -> LDS with ALL loads of the set out before the first store. As one loop
  // (what `for (i = tid; ..) bl[i] = src[i]` compiles to: global_load_dword; s_waitcnt vmcnt(0); ds_write_b32 per iteration,
  // r05 ISA) staging the 896 biases was four SERIALISED L2 round trips on every wave at the start of the kernel and at the
  // end of every layer, and the two norm epilogues fetched their constants from global memory right where they need them —
  // six exposed round trips per layer on a chain that is latency-bound (profiles/r05_encoder_variants.json); now one, in
  // front of the end-of-layer barrier. Both sets by layer parity: a fast wave stages layer l + 1 while a slow one still
  // reads layer l's second norm. (Requested a GEMM or a barrier EARLIER than they are stored the six values cost 19 - 30
  // spilled registers: the kernel sits at 254 of 256.)
  // (`t0`: the thread index AS LAUNDERED at the top of the layer — derived from threadIdx.x the four store addresses are loop
  // invariants, hoisted out of the layer loop, spilled, and reloaded one scratch round trip at a time right here)
  auto stage_layer = [&](int layer, int t0) {
    // every source is chosen per WAVE (a scalar select): wave w takes norm array w, waves 0 - 2 the 384 qkv biases, all four
    // the 512 MLP biases, two consecutive floats per lane. (Chosen per lane the four norm pointers become a vector load of
    // the pointer from the kernel-argument segment in front of the load of the value: two dependent round trips.)
    const int ws = __builtin_amdgcn_readfirstlane(t0 >> 6), l2 = 2 * (t0 & 63);
    const float* nsrc = ws < 2 ? (ws == 0 ? a.n1_scale : a.n1_shift) : (ws == 2 ? a.n2_scale : a.n2_shift);
    const float2 nv = *reinterpret_cast<const float2*>(nsrc + layer * kD + l2);
    const float2 b1v = *reinterpret_cast<const float2*>(a.b1 + layer * kFF + 2 * t0);
    float2 qv = make_float2(0.0f, 0.0f);
    if (ws < 3) qv = *reinterpret_cast<const float2*>(a.bqkv + layer * 3 * kD + 2 * t0);
    float* bdst = bl2 + (layer & 1) * kBiasFloats;
    if (ws < 3) *reinterpret_cast<float2*>(bdst + 2 * t0) = qv;
    *reinterpret_cast<float2*>(bdst + 3 * kD + 2 * t0) = b1v;
    *reinterpret_cast<float2*>(nl2 + (layer & 1) * kNormFloats + ws * kD + l2) = nv;
  };

  int tid = threadIdx.x;
  int w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;  // (not const: see the top of the layer loop)
  const int b = blockIdx.x;
  const int N = a.N;

  const E* wqkv_all = static_cast<const E*>(a.wqkv_packed);
  const E* wo_all = static_cast<const E*>(a.wo_packed);
  const E* w1_all = static_cast<const E*>(a.w1_packed);
  const E* w2_all = static_cast<const E*>(a.w2_packed);
  const E* wf_all = static_cast<const E*>(a.wfold_packed);
  // weight fragments of the NEXT GEMM, always one call ahead (gemm_t). The first set is requested before anything else:
  // the fixed cost of an instance (init embedding, first fragments, fold, stores) is 0.38 of the kernel's 1.1 ms
  // (`tools/enc_layers.py`: 0.62 / 0.85 / 1.11 / 1.78 ms at 1 / 2 / 3 / 6 layers), most of it exposed round trips
  bf16x8 wf[8];
  load_wfrags(wf, wqkv_all, 8, w, 0, lane);

  // ---- init embedding (K = 2 .. 6: plain VALU), padding rows zeroed ---------------------------
  stage_layer(0, tid);
  if constexpr (TRAIN) {
    const E* src = ts.x0 + (int64_t)b * N * kD;
    for (int i = tid; i < 32 * TT * 16; i += kThreads) {
      const int row = i >> 4, c16 = i & 15;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < N) v = *reinterpret_cast<const uint4*>(src + (int64_t)row * kD + 8 * c16);
      *reinterpret_cast<uint4*>(xs + row * kRS + 8 * c16) = v;
    }
  } else {
#ifdef RL4CO_ENC_SKIP_INIT  // timing probe only: a constant residual stream instead of the embedding
    for (int i = tid; i < 32 * TT * 16; i += kThreads) *reinterpret_cast<uint4*>(xs + (i >> 4) * kRS + 8 * (i & 15)) = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
#else
    init_embed_rows16<E>(a, b, 0, 32 * TT, xs, reinterpret_cast<float*>(ys), tid);  // (ys is free here: the features are staged in it)
#endif
  }
  rl4co::lds_barrier();

  // the weight hand-over of the GEMM calls is compiled in (straight-line calls): with the run-time form the number of loads
  // younger than a fragment depends on the path, and every call's first MFMA waits with vmcnt(0) — draining, in the training
  // forward, all the saves issued before it
  constexpr int kNx = 1;
  const float kInf = opaque_inf();
  for (int layer = 0; layer < a.num_layers; ++layer) {
    // The lane indices pass through an opaque copy once per layer, so every per-lane LDS / weight address below is
    // derived INSIDE the iteration, next to its use. Hoisted out of the loop as invariants they were ~70 registers
    // live across the whole kernel — the allocator spilled them (296 bytes of scratch per lane, 310 MB of scratch
    // writes per launch: the "1.5x write amplification" of the PMC pass) and reloaded them inside the GEMM loops
    asm volatile("" : "+v"(tid), "+v"(w), "+v"(lane));
    l31 = lane & 31;
    hi = lane >> 5;
    LayerPtrs<E> L;
    L.wqkv = wqkv_all + (int64_t)layer * 3 * kD * kD;
    L.wo = wo_all + (int64_t)layer * kD * kD;
    L.w1 = w1_all + (int64_t)layer * kFF * kD;
    L.w2 = w2_all + (int64_t)layer * kD * kFF;
    L.bqkv = bl2 + (layer & 1) * kBiasFloats;
    L.b1 = L.bqkv + 3 * kD;
    L.n1a = nl2 + (layer & 1) * kNormFloats;
    L.n1b = L.n1a + kD;
    L.n2a = L.n1a + 2 * kD;
    L.n2b = L.n1a + 3 * kD;
    // ---- Q, K (transposed form) and V (plain form) of head pair w, kept as fragments ---------
    // vfh[hh]: V^T fragments for head hh of the pair — the lanes holding the OTHER head's dims carry ones instead, so
    // the value product's idle output rows deliver the softmax denominator (sum of the bf16 numerators) for free
    bf16x8 kf[TT][2], vfh[2][TT][2];
    float qn2[2] = {0.0f, 0.0f}, kn2[2] = {0.0f, 0.0f};  // max over tokens of |q|^2, |k|^2 per head of the pair
    {
      f32x16 acc[TT];
      // 1/sqrt(16) and log2(e) are folded into the packed Wq and its bias on the host (encoder.py): the softmax below
      // is exp2(s - max) and the projection needs no epilogue arithmetic at all
      gemm_t<TT, true, true, kNx>(acc, wf, xs, lane, L.wqkv, 8, 4 + w, 0, bias_tile(L.bqkv, 32 * w, hi));
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        float h0, h1;
        head_sqnorms(acc[tt], h0, h1);
        qn2[0] = fmaxf(qn2[0], h0);
        qn2[1] = fmaxf(qn2[1], h1);
      }
      // Q^T parked in this wave's own 32 columns of ys (each lane re-reads only its own token
      // row, and later overwrites it with the attention output of that same row)
      store_t<TT>(ys, acc, 32 * w, lane, N);
      if constexpr (TRAIN) save_t<TT>(ts.qkv + ((int64_t)layer * a.B + b) * N * 3 * kD, 3 * kD, acc, 32 * w, N, lane);
      gemm_t<TT, true, true, kNx>(acc, wf, xs, lane, L.wqkv, 8, 8 + w, 0, bias_tile(L.bqkv + kD, 32 * w, hi));
      if constexpr (TRAIN) save_t<TT>(ts.qkv + ((int64_t)layer * a.B + b) * N * 3 * kD + kD, 3 * kD, acc, 32 * w, N, lane);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        if (tt + 1 == TT) {  // padding keys: zero operand rows (see store_t) — their scores are masked either way
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tt][r] = (32 * tt + l31 < N) ? acc[tt][r] : 0.0f;
        }
        kf[tt][0] = frag_from_acc<E>(acc[tt], 0);
        kf[tt][1] = frag_from_acc<E>(acc[tt], 1);
        float h0, h1;
        head_sqnorms(acc[tt], h0, h1);
        kn2[0] = fmaxf(kn2[0], h0);
        kn2[1] = fmaxf(kn2[1], h1);
      }
      // V = X . Wv^T: A = token rows from LDS, B = weight fragment -> C[row = token][col = dim]
      {
        const float bv = L.bqkv[2 * kD + 32 * w + l31];  // plain form: the bias belongs to the lane's dim column
        f32x16 bt;
#pragma unroll
        for (int r = 0; r < 16; ++r) bt[r] = bv;
        gemm_t<TT, false, true, 0>(acc, wf, xs, lane, static_cast<const E*>(nullptr), 0, 0, 0, bt);  // nothing in flight across the attention (register peak)
      }
      if constexpr (TRAIN) {
        // plain form: the lane owns ONE dim column and sixteen token rows per tile — 2-byte stores, 32 consecutive dims
        // (64 bytes) of a token row per half wave; the workgroup's four waves complete the row in L2
        E* vdst = ts.qkv + ((int64_t)layer * a.B + b) * N * 3 * kD + 2 * kD + 32 * w + l31;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tok = 32 * tt + rowmap(r, hi);
            if (tok < N) vdst[(int64_t)tok * 3 * kD] = (E)acc[tt][r];
          }
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bf16x8 v = frag_from_acc<E>(acc[tt], u);
          bf16x8 ones;
#pragma unroll
          for (int i = 0; i < 8; ++i) ones[i] = (E)1.0f;
          vfh[0][tt][u] = (l31 < 16) ? v : ones;   // lane = dim column: dims 0..15 are head 2w, 16..31 head 2w + 1
          vfh[1][tt][u] = (l31 < 16) ? ones : v;
        }
      }
    }
    // |score| <= max_i |q_i| max_j |k_j| per head (Cauchy-Schwarz; fp32 norms, the products see their bf16 roundings:
    // the bound keeps a wide margin): below kFastBound the softmax needs no running maximum
    bool fast_head[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const float b2 = wave_max32(qn2[hh]) * wave_max32(kn2[hh]) * (TRAIN ? kTrainScale * kTrainScale : 1.0f);
      fast_head[hh] = kWideRange<E> && __builtin_amdgcn_readfirstlane((b2 <= kFastBound * kFastBound) ? 1 : 0) != 0;
    }

    // ---- attention for heads 2w, 2w+1 over all queries, wave-private ---------------------------
    constexpr int kLastRegs = 4 * VR4;  // registers of the last key tile that can hold real keys
#pragma unroll
    for (int qt = 0; qt < TT; ++qt) {
      E* qrow = ys + (32 * qt + l31) * kRS + 32 * w;
      f32x16 o = zero16();
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        bf16x8 qf;
        {
          const bf16x4 lo = *reinterpret_cast<const bf16x4*>(qrow + 16 * hh + 4 * hi);
          const bf16x4 up = *reinterpret_cast<const bf16x4*>(qrow + 16 * hh + 8 + 4 * hi);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            qf[i] = lo[i];
            qf[4 + i] = up[i];
          }
        }
        f32x16 acc0, acc1;  // two accumulators: half the dependent-MFMA chain; each starts from a literal-zero C operand
        if (TT == 1) acc1 = zero16();
        float m_used = 0.0f;  // (TRAIN: the maximum the numerators were taken against, for the saved log-sum-exp)
        if (fast_head[hh]) {
          // bounded scores: p = exp2(s) tile by tile — no maximum, no subtraction, no score tile kept alive
#pragma unroll
          for (int kt = 0; kt < TT; ++kt) {
            const f32x16 sk = mfma(kf[kt][hh], qf, zero16());
            const int nreg = (kt == TT - 1) ? kLastRegs : 16;
            f32x16 p;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = 0.0f;
              if (r < nreg) {
                v = __builtin_amdgcn_exp2f(TRAIN ? sk[r] * kTrainScale : sk[r]);
                if (kt == TT - 1) v = (32 * kt + rowmap(r, hi) < N) ? v : 0.0f;  // padding keys inside the valid registers
              }
              p[r] = v;
            }
            if (kt & 1) {
              acc1 = mfma(vfh[hh][kt][0], frag_from_acc<E>(p, 0), kt == 1 ? zero16() : acc1);
              if (nreg > 8) acc1 = mfma(vfh[hh][kt][1], frag_from_acc<E>(p, 1), acc1);
            } else {
              acc0 = mfma(vfh[hh][kt][0], frag_from_acc<E>(p, 0), kt == 0 ? zero16() : acc0);
              if (nreg > 8) acc0 = mfma(vfh[hh][kt][1], frag_from_acc<E>(p, 1), acc0);
            }
          }
        } else {
          f32x16 s[TT];
          float m = -__builtin_huge_valf();
#pragma unroll
          for (int kt = 0; kt < TT; ++kt) {
            s[kt] = mfma(kf[kt][hh], qf, zero16());
            if constexpr (TRAIN) {
#pragma unroll
              for (int r = 0; r < 16; ++r) s[kt][r] *= kTrainScale;
            }
            if (kt == TT - 1) {  // TT = ceil(N/32): only the last key tile can hold padding keys
#pragma unroll
              for (int r = 0; r < 16; ++r)
                s[kt][r] = (r < kLastRegs && 32 * kt + rowmap(r, hi) < N) ? s[kt][r] : -__builtin_huge_valf();
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, s[kt][r]);
          }
          m = fmaxf(m, rl4co::bfly_f<32>(m));
          m_used = m;
#pragma unroll
          for (int kt = 0; kt < TT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - m);
            if (kt & 1) {
              acc1 = mfma(vfh[hh][kt][0], frag_from_acc<E>(s[kt], 0), kt == 1 ? zero16() : acc1);
              acc1 = mfma(vfh[hh][kt][1], frag_from_acc<E>(s[kt], 1), acc1);
            } else {
              acc0 = mfma(vfh[hh][kt][0], frag_from_acc<E>(s[kt], 0), kt == 0 ? zero16() : acc0);
              acc0 = mfma(vfh[hh][kt][1], frag_from_acc<E>(s[kt], 1), acc0);
            }
          }
        }
        // rows of the other head's dims carried ones: every one of them is the row sum of this query's numerators
        const float inv = 1.0f / (acc0[8 * (1 - hh)] + acc1[8 * (1 - hh)]);
        if constexpr (TRAIN) {
          if (hi == 0 && 32 * qt + l31 < N)
            ts.lse[(((int64_t)layer * a.B + b) * 8 + 2 * w + hh) * N + 32 * qt + l31] =
                m_used + __builtin_amdgcn_logf(acc0[8 * (1 - hh)] + acc1[8 * (1 - hh)]);  // log2 domain
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) o[8 * hh + r] = (acc0[8 * hh + r] + acc1[8 * hh + r]) * inv;
      }
      // attention output row [token][dims of head pair w] over the Q^T this lane just consumed
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (E)o[4 * c + i];
        uint2 u = __builtin_bit_cast(uint2, v);
        if (qt + 1 == TT) u = (32 * qt + l31 < N) ? u : make_uint2(0u, 0u);  // padding queries: zero rows for the out-proj (see store_t)
        *reinterpret_cast<uint2*>(qrow + 8 * c + 4 * hi) = u;
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // keep these loads out of the attention loop (its register peak)
    load_wfrags(wf, L.wo, 8, w, 0, lane);  // out-proj weights: in flight across the barrier
    rl4co::lds_barrier();
    if constexpr (TRAIN) rows_out<TT>(ys, ts.att + ((int64_t)layer * a.B + b) * N * kD, kD, N, tid);

    // ---- out-proj + residual + norm1 ---------------------------------------------------------------
    {
      f32x16 y[TT];
      // (out_proj's bias rides in the norm's shift — batch norm — or cancels in the per-channel mean — instance norm:
      // folded on the host, encoder.py)
      gemm_t<TT, true, true, kNx>(y, wf, ys, lane, L.w1, 8, w, 0, zero16());  // next: FFN1 chunk 0
      if constexpr (TRAIN) {
        float* st = ts.stats + ((int64_t)layer * 4 * a.B + b) * kD;
        residual_norm_train<TT>(xs, y, 32 * w, L.n1a, L.n1b, N, lane, ts.y1 + ((int64_t)layer * a.B + b) * N * kD, st, st + (int64_t)a.B * kD);
      } else {
        residual_norm<TT, LAYER>(xs, y, 32 * w, L.n1a, L.n1b, a.norm, N, lane, meanv, w);
      }
    }
    rl4co::lds_barrier();
    if constexpr (TRAIN) rows_out<TT>(xs, ts.x1 + ((int64_t)layer * a.B + b) * N * kD, kD, N, tid);

    // ---- FFN: hidden in 4 chunks of 128, FFN2 accumulates across chunks -----------------------------
    {
      f32x16 y2[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) y2[tt] = zero16();  // the chunk loop accumulates; the MLP's output bias: see out_proj
      for (int c = 0; c < 4; ++c) {
        f32x16 h1[TT];
        gemm_t<TT, true, true, kNx>(h1, wf, xs, lane, L.w2, 32, w, 8 * c, bias_tile(L.b1, 32 * (4 * c + w), hi));  // next: FFN2 of this chunk
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) h1[tt][r] = relu(h1[tt][r], kInf);
        if (c > 0) rl4co::lds_barrier();  // every wave is done reading the previous chunk
        store_t<TT>(ys, h1, 32 * w, lane, N);
        rl4co::lds_barrier();
        if constexpr (TRAIN) rows_out<TT>(ys, ts.h + ((int64_t)layer * a.B + b) * N * kFF + kD * c, kFF, N, tid);
        // next: FFN1 of the next chunk, then the next layer's Q projection, finally the first fold block
        const bool last_layer = layer + 1 == a.num_layers;
        // (the training forward has no fold: its last call fetches W1's first tile again, unused)
        const E* nxt = c < 3 ? L.w1 : (last_layer ? (TRAIN ? L.w1 : wf_all) : wqkv_all + (int64_t)(layer + 1) * 3 * kD * kD);
        gemm_t<TT, true, false, kNx>(y2, wf, ys, lane, nxt, 8, c < 3 ? 4 * (c + 1) + w : w, 0, y2[0]);
      }
      if constexpr (TRAIN) {
        float* st = ts.stats + (((int64_t)layer * 4 + 2) * a.B + b) * kD;
        residual_norm_train<TT>(xs, y2, 32 * w, L.n2a, L.n2b, N, lane, ts.y2 + ((int64_t)layer * a.B + b) * N * kD, st, st + (int64_t)a.B * kD);
      } else {
        residual_norm<TT, LAYER>(xs, y2, 32 * w, L.n2a, L.n2b, a.norm, N, lane, meanv, w);
      }
    }
    // every wave is past the last chunk's barriers, i.e. done with the previous layer's half of both sets
    if (layer + 1 < a.num_layers) stage_layer(layer + 1, tid);
    rl4co::lds_barrier();
    if constexpr (TRAIN) rows_out<TT>(xs, ts.out + ((int64_t)layer * a.B + b) * N * kD, kD, N, tid);
  }
  if constexpr (TRAIN) return;  // the training forward ends with the last layer's output (the cache fold has its own autograd node)

  // ---- optional: final node embeddings h (fp32) ---------------------------------------------------
  if (a.hidden) {
    float* hout = a.hidden + (int64_t)b * N * kD;
    for (int idx = tid; idx < N * kD; idx += kThreads) hout[idx] = (float)xs[(idx >> 7) * kRS + (idx & 127)];
  }

  // ---- fold: cache planes straight out of the accumulators ------------------------------------------
#ifdef RL4CO_ENC_SKIP_FOLD  // timing probe only
  const int nblocks = 0;
#else
  const int nblocks = (a.env == RL4CO_ENV_TSP) ? 5 : 4;
#endif
  for (int blk = 0; blk < nblocks; ++blk) {
    f32x16 acc[TT];
    gemm_t<TT, true, true, 1>(acc, wf, xs, lane, wf_all + (int64_t)(blk + 1 < nblocks ? blk + 1 : 0) * kD * kD, 8, w, 0, zero16());  // (last block: dummy)
    // The tile leaves through LDS (`ys` is free after the last layer): stored straight from the accumulators a lane
    // owns 8 / 16 bytes in each of 32 token rows; staged, a plane of an instance is ONE contiguous run of 16-byte lanes.
    // 16-bit planes carry the element type of the activations — and so do (r06, ctx_dtype) the context tables: the decode
    // kernels widen their rows on load (rl4co_am_decode_args.ctx_dtype), the reference's own query is 16-bit under autocast
    // (project_context is a Linear), and the fold is bound by its HBM writes: 179 -> 128 KB per instance at TSP-100
    if ((blk < 3 || a.ctx_dtype != RL4CO_DT_F32) && a.cache_dtype != RL4CO_DT_F32) {
      E* out;
      if (blk < 3) out = static_cast<E*>(a.kvl) + (int64_t)blk * a.kvl_plane_stride + (int64_t)b * a.kvl_batch_stride;
      else out = static_cast<E*>((a.env == RL4CO_ENV_TSP && blk == 3) ? a.ctx_first : a.ctx_cur) + (int64_t)b * N * kD;
      store_t<TT>(ys, acc, 32 * w, lane);
      rl4co::lds_barrier();
      rows_out<TT>(ys, out, kD, N, tid);  // (a fixed number of stores: the next block's first MFMA waits for its weights only)
      rl4co::lds_barrier();
    } else {
      float* out;
      if (blk < 3) {
        out = static_cast<float*>(a.kvl) + (int64_t)blk * a.kvl_plane_stride + (int64_t)b * a.kvl_batch_stride;
      } else if (a.env == RL4CO_ENV_TSP) {
        out = static_cast<float*>(blk == 3 ? a.ctx_first : a.ctx_cur) + (int64_t)b * N * kD;
      } else {
        out = static_cast<float*>(a.ctx_cur) + (int64_t)b * N * kD;
      }
      constexpr int kFS = kD + 4;  // fp32 staging row stride: 64 token rows fit the 34 KB of ys
      float* fs = reinterpret_cast<float*>(ys);
#pragma unroll
      for (int p = 0; p < (TT + 1) / 2; ++p) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const int tt = 2 * p + t2;
          if (tt < TT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float4 v = make_float4(acc[tt][4 * c], acc[tt][4 * c + 1], acc[tt][4 * c + 2], acc[tt][4 * c + 3]);
              *reinterpret_cast<float4*>(fs + (32 * t2 + l31) * kFS + 32 * w + 8 * c + 4 * hi) = v;
            }
          }
        }
        rl4co::lds_barrier();
        const int rows = min(64, N - 64 * p);
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // 64 rows x 32 pieces / 256 threads, unconditional (see rows_out)
          const int i = tid + kThreads * j, row = min(i >> 5, rows - 1), c4 = i & 31;
          *reinterpret_cast<float4*>(out + (int64_t)(64 * p + row) * kD + 4 * c4) =
              *reinterpret_cast<const float4*>(fs + row * kFS + 4 * c4);
        }
        rl4co::lds_barrier();
      }
    }
  }

  // ---- graph context: project_fixed_context(mean_j h_j)  (decoder.py:216-219) -----------------------
#ifdef RL4CO_ENC_SKIP_GCTX
  if (false) {
#else
  if (a.q_bias) {
#endif
    // wave w: output rows 32 w .. 32 w + 31; a row of W is ONE coalesced 512-byte load (two consecutive inputs per
    // lane) and a butterfly sum (a thread per row read its 128 inputs at a 512-byte lane stride: 64 cache lines per
    // load instruction). The rows are requested BEFORE the column means they will meet (registers are free here).
    float2 wv[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) wv[r] = *reinterpret_cast<const float2*>(a.w_fixed + (int64_t)(32 * w + r) * kD + 2 * lane);
    if (tid < kD) {
      float s = 0.0f;
      for (int tok = 0; tok < N; ++tok) s += (float)xs[tok * kRS + tid];
      meanv[tid] = s / (float)N;
    }
    rl4co::lds_barrier();
    const float2 mv = *reinterpret_cast<const float2*>(meanv + 2 * lane);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const float acc = rl4co::bfly_sum<1, 64>(fmaf(wv[r].y, mv.y, wv[r].x * mv.x));
      if (lane == r) a.q_bias[(int64_t)b * kD + 32 * w + r] = acc;
    }
  }
}

// ================================================================================================================
// Token-tile kernels: the same layer algebra for graphs beyond 128 nodes (BASELINE configs[4], CVRP-500), batch norm in
// eval mode. A workgroup owns 128 consecutive nodes of one instance (grid = tiles x instances, two workgroups per CU);
// the GEMM routine, fragment packing, bias / norm folding are the fused kernel's own, so the packed weights serve both.
// Per layer three launches instead of seven: Q / K / V projection -> attention (am_attn_flash.hip) -> ONE kernel for
// out-proj + norm + MLP + norm, whose 512-wide hidden goes through LDS in four chunks and never through HBM (the per-op
// path moved 22 [B N, 128]-sized 16-bit passes per layer, this one 7).
// ================================================================================================================
constexpr int kTokT = 4;  // token tiles of 32 per workgroup
constexpr int kTok = 32 * kTokT;

template <typename E>
__device__ inline void tok_load(E* xs, const E* src, int b, int n0, int N, int tid) {
  const E* base = src + ((int64_t)b * N + n0) * kD;
  const int valid = min(kTok, N - n0);
  for (int i = tid; i < kTok * 16; i += kThreads) {
    const int row = i >> 4, c16 = i & 15;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < valid) v = *reinterpret_cast<const uint4*>(base + (int64_t)row * kD + 8 * c16);
    *reinterpret_cast<uint4*>(xs + row * kRS + 8 * c16) = v;
  }
}
template <typename E>
__device__ inline void tok_store(const E* xs, E* dst, int64_t row_stride, int valid, int tid) {
  rows_out<kTokT>(xs, dst, row_stride, valid, tid);  // a FIXED number of unconditional stores (see rows_out: no vmcnt(0) behind it)
}

template <typename E>
__global__ void __launch_bounds__(kThreads, 2) tok16_init_embed_kernel(const rl4co_am_encoder_args a, E* x0) {
  extern __shared__ __align__(16) unsigned char smem[];
  E* xs = reinterpret_cast<E*>(smem);
  float* lsh = reinterpret_cast<float*>(xs + kTok * kRS);
  const int tid = threadIdx.x, b = blockIdx.y, n0 = kTok * blockIdx.x;
  init_embed_rows16<E>(a, b, n0, kTok, xs, lsh, tid);
  __syncthreads();
  tok_store(xs, x0 + ((int64_t)b * a.N + n0) * kD, kD, min(kTok, a.N - n0), tid);
}

// x -> packed qkv rows [B N, 384] (q | k | v; q carries 1/4 log2 e from the packed weights: the attention kernel's scores
// are in the exp2 domain as they leave the product) and, per (instance, head), the maxima over the nodes of |q_h|^2 and
// |k_h|^2 (fp32 accumulators, before rounding): |score| <= sqrt of their product, the bound the attention kernel's
// max-free softmax path is taken under.
template <typename E>
__global__ void __launch_bounds__(kThreads, 2) tok16_qkv_kernel(const E* __restrict__ x, int N, const E* __restrict__ wqkv,
                                                                const float* __restrict__ bqkv, E* __restrict__ qkv,
                                                                uint32_t* __restrict__ bound) {
  extern __shared__ __align__(16) unsigned char smem[];
  E* xs = reinterpret_cast<E*>(smem);
  E* ys = xs + kTok * kRS;
  float* bl = reinterpret_cast<float*>(ys + kTok * kRS);  // [384]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y, n0 = kTok * blockIdx.x, valid = min(kTok, N - n0);
  vec8<E> wf[8];
  load_wfrags(wf, wqkv, 8, w, 0, lane);
  tok_load(xs, x, b, n0, N, tid);
  for (int i = tid; i < 3 * kD; i += kThreads) bl[i] = bqkv[i];
  __syncthreads();
#pragma unroll 1
  for (int part = 0; part < 3; ++part) {
    f32x16 acc[kTokT];
    gemm_t<kTokT, true, true, 1>(acc, wf, xs, lane, wqkv, 8, part < 2 ? 4 * (part + 1) + w : w, 0,  // (last part: a dummy fetch)
                                 bias_tile(bl + kD * part, 32 * w, hi));
    if (part < 2) {
      float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
      for (int tt = 0; tt < kTokT; ++tt) {
        float h0, h1;
        head_sqnorms(acc[tt], h0, h1);
        const bool live = 32 * tt + l31 < valid;
        m0 = fmaxf(m0, live ? h0 : 0.0f);
        m1 = fmaxf(m1, live ? h1 : 0.0f);
      }
      m0 = wave_max32(m0);
      m1 = wave_max32(m1);
      if (lane == 0) {  // non-negative floats order like their bit patterns
        atomicMax(bound + ((int64_t)b * 8 + 2 * w) * 2 + part, __float_as_uint(m0));
        atomicMax(bound + ((int64_t)b * 8 + 2 * w + 1) * 2 + part, __float_as_uint(m1));
      }
    }
    if (part > 0) __syncthreads();  // the previous part has left the staging rows
    store_t<kTokT>(ys, acc, 32 * w, lane);
    __syncthreads();
    tok_store(ys, qkv + ((int64_t)b * N + n0) * 3 * kD + kD * part, 3 * kD, valid, tid);
  }
}

// Norm(x + out_proj(att)) -> Norm(. + MLP(.)) on one token tile: the second half of am_encoder_kernel's layer body
template <typename E>
__global__ void __launch_bounds__(kThreads, 2) tok16_mlp_kernel(const E* __restrict__ x, const E* __restrict__ att, int N,
                                                                const E* __restrict__ wo, const E* __restrict__ w1,
                                                                const E* __restrict__ w2, const float* __restrict__ b1,
                                                                const float* __restrict__ n1a, const float* __restrict__ n1b,
                                                                const float* __restrict__ n2a, const float* __restrict__ n2b,
                                                                E* __restrict__ xout) {
  extern __shared__ __align__(16) unsigned char smem[];
  E* xs = reinterpret_cast<E*>(smem);
  E* ys = xs + kTok * kRS;
  float* bl = reinterpret_cast<float*>(ys + kTok * kRS);  // b1 [512]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, hi = lane >> 5;
  const int b = blockIdx.y, n0 = kTok * blockIdx.x;
  vec8<E> wf[8];
  const float kInf = opaque_inf();
  load_wfrags(wf, wo, 8, w, 0, lane);
  tok_load(xs, x, b, n0, N, tid);
  tok_load(ys, att, b, n0, N, tid);
  for (int i = tid; i < kFF; i += kThreads) bl[i] = b1[i];
  __syncthreads();
  {
    f32x16 y[kTokT];
    gemm_t<kTokT, true, true, 1>(y, wf, ys, lane, w1, 8, w, 0, zero16());  // (out_proj's bias rides in the norm's shift: encoder.py)
    residual_norm<kTokT>(xs, y, 32 * w, n1a, n1b, 0, N, lane);
  }
  __syncthreads();
  {
    f32x16 y2[kTokT];
#pragma unroll
    for (int tt = 0; tt < kTokT; ++tt) y2[tt] = zero16();
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      f32x16 h1[kTokT];
      gemm_t<kTokT, true, true, 1>(h1, wf, xs, lane, w2, 32, w, 8 * c, bias_tile(bl, 32 * (4 * c + w), hi));
#pragma unroll
      for (int tt = 0; tt < kTokT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[tt][r] = relu(h1[tt][r], kInf);
      __syncthreads();  // every wave is done reading ys (the attention output / the previous chunk)
      store_t<kTokT>(ys, h1, 32 * w, lane);
      __syncthreads();
      gemm_t<kTokT, true, false, 1>(y2, wf, ys, lane, w1, 8, c < 3 ? 4 * (c + 1) + w : w, 0, y2[0]);  // (last chunk: a dummy fetch)
    }
    residual_norm<kTokT>(xs, y2, 32 * w, n2a, n2b, 0, N, lane);
  }
  __syncthreads();
  tok_store(xs, xout + ((int64_t)b * N + n0) * kD, kD, min(kTok, N - n0), tid);
}

// ---- token tiles under instance / layer norm (norm = 1 / 2) ---------------------------------------------------------
// The statistics span all tiles of an instance, so the two sub-blocks of a layer end BEFORE their norm: each half writes
// its pre-norm sums (rounded to the element type, like autocast's x + module(x)) and, per tile and channel, the mean and
// the centred sum of squares of the tile's valid tokens; `tok16_norm_apply_kernel` combines the tiles' pairs (Chan's
// update: deterministic, no atomics, no cancellation) into the instance's statistics and normalises the rows.
//   nn/ops.py:46-51 (instance: per channel over the nodes, biased variance; layer: one mean / one unbiased variance)
constexpr int kStatFloats = 2 * kD;  // per (instance, tile): mean [128] | M2 [128]

// x + y (+ the bias in front of a layer norm) for the wave's 32-dim tile, rounded, back into xs; tile statistics -> st
template <int TT, typename E>
__device__ inline void residual_stats(E* xs, f32x16 (&y)[TT], int dim0, const float* pre_bias, int valid, int lane, float* st) {
  const int l31 = lane & 31, hi = lane >> 5;
  float be[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) be[r] = pre_bias ? pre_bias[dim0 + rowmap(r, hi)] : 0.0f;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const vec4<E> x = *reinterpret_cast<const vec4<E>*>(xs + (32 * tt + l31) * kRS + dim0 + 8 * c + 4 * hi);
#pragma unroll
      for (int s = 0; s < 4; ++s) y[tt][4 * c + s] = (float)(E)(((float)x[s] + y[tt][4 * c + s]) + be[4 * c + s]);
    }
  }
  const float inv = 1.0f / (float)valid;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) s += (32 * tt + l31 < valid) ? y[tt][r] : 0.0f;
    s = rl4co::bfly_sum<1, 32>(s);
    const float mean = s * inv;
    float v = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const float d = y[tt][r] - mean;
      v += (32 * tt + l31 < valid) ? d * d : 0.0f;
    }
    v = rl4co::bfly_sum<1, 32>(v);
    if (l31 == 0) {
      st[dim0 + rowmap(r, hi)] = mean;
      st[kD + dim0 + rowmap(r, hi)] = v;
    }
  }
  store_t<TT>(xs, y, dim0, lane);
}

// pre-norm x + out_proj(att) of one token tile
template <typename E>
__global__ void __launch_bounds__(kThreads, 2) tok16_attn_half_kernel(const E* __restrict__ x, const E* __restrict__ att, int N,
                                                                      const E* __restrict__ wo, const float* __restrict__ pre_bias,
                                                                      E* __restrict__ ypre, float* __restrict__ stats) {
  extern __shared__ __align__(16) unsigned char smem[];
  E* xs = reinterpret_cast<E*>(smem);
  E* ys = xs + kTok * kRS;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y, n0 = kTok * blockIdx.x, valid = min(kTok, N - n0);
  vec8<E> wf[8];
  load_wfrags(wf, wo, 8, w, 0, lane);
  tok_load(xs, x, b, n0, N, tid);
  tok_load(ys, att, b, n0, N, tid);
  __syncthreads();
  f32x16 y[kTokT];
  gemm_t<kTokT, true, true, 0>(y, wf, ys, lane, static_cast<const E*>(nullptr), 0, 0, 0, zero16());
  residual_stats<kTokT>(xs, y, 32 * w, pre_bias, valid, lane, stats + ((int64_t)b * gridDim.x + blockIdx.x) * kStatFloats);
  __syncthreads();
  tok_store(xs, ypre + ((int64_t)b * N + n0) * kD, kD, valid, tid);
}

// pre-norm x + MLP(x) of one token tile (x: the normalised output of the attention half)
template <typename E>
__global__ void __launch_bounds__(kThreads, 2) tok16_ffn_half_kernel(const E* __restrict__ x, int N, const E* __restrict__ w1,
                                                                     const E* __restrict__ w2, const float* __restrict__ b1,
                                                                     const float* __restrict__ pre_bias, E* __restrict__ ypre,
                                                                     float* __restrict__ stats) {
  extern __shared__ __align__(16) unsigned char smem[];
  E* xs = reinterpret_cast<E*>(smem);
  E* ys = xs + kTok * kRS;
  float* bl = reinterpret_cast<float*>(ys + kTok * kRS);  // b1 [512]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, hi = lane >> 5;
  const int b = blockIdx.y, n0 = kTok * blockIdx.x, valid = min(kTok, N - n0);
  vec8<E> wf[8];
  const float kInf = opaque_inf();
  load_wfrags(wf, w1, 8, w, 0, lane);
  tok_load(xs, x, b, n0, N, tid);
  for (int i = tid; i < kFF; i += kThreads) bl[i] = b1[i];
  __syncthreads();
  f32x16 y2[kTokT];
#pragma unroll
  for (int tt = 0; tt < kTokT; ++tt) y2[tt] = zero16();
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    f32x16 h1[kTokT];
    gemm_t<kTokT, true, true, 1>(h1, wf, xs, lane, w2, 32, w, 8 * c, bias_tile(bl, 32 * (4 * c + w), hi));
#pragma unroll
    for (int tt = 0; tt < kTokT; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) h1[tt][r] = relu(h1[tt][r], kInf);
    if (c > 0) __syncthreads();  // every wave is done reading the previous chunk
    store_t<kTokT>(ys, h1, 32 * w, lane);
    __syncthreads();
    gemm_t<kTokT, true, false, 1>(y2, wf, ys, lane, w1, 8, c < 3 ? 4 * (c + 1) + w : w, 0, y2[0]);
  }
  residual_stats<kTokT>(xs, y2, 32 * w, pre_bias, valid, lane, stats + ((int64_t)b * gridDim.x + blockIdx.x) * kStatFloats);
  __syncthreads();
  tok_store(xs, ypre + ((int64_t)b * N + n0) * kD, kD, valid, tid);
}

__device__ inline float block_sum4(float v, float* red, int tid) {  // sum over the 256 threads (red: 4 floats of LDS)
  v = rl4co::bfly_sum<1, 64>(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// rows of one token tile normalised with the statistics of the WHOLE instance (all its tiles' pairs combined)
template <typename E>
__global__ void __launch_bounds__(kThreads) tok16_norm_apply_kernel(const E* __restrict__ ypre, const float* __restrict__ stats, int N,
                                                                    int kind, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, E* __restrict__ xout) {
  __shared__ float ab[2 * kD];
  __shared__ float red[4];
  const int tid = threadIdx.x, b = blockIdx.y, tiles = gridDim.x, n0 = kTok * blockIdx.x, valid = min(kTok, N - n0);
  const float* st = stats + (int64_t)b * tiles * kStatFloats;
  float mean = 0.0f, m2 = 0.0f;
  if (tid < kD) {
    float tot = 0.0f;
    for (int t = 0; t < tiles; ++t) tot += (float)min(kTok, N - kTok * t) * st[t * kStatFloats + tid];
    mean = tot / (float)N;
    for (int t = 0; t < tiles; ++t) {
      const float d = st[t * kStatFloats + tid] - mean;
      m2 += st[t * kStatFloats + kD + tid] + (float)min(kTok, N - kTok * t) * d * d;
    }
  }
  if (kind == 1) {
    if (tid < kD) {
      const float alpha = rsqrtf(m2 / (float)N + 1e-5f) * gamma[tid];
      ab[tid] = alpha;
      ab[kD + tid] = beta[tid] - mean * alpha;
    }
  } else {  // layer: every channel holds N values
    const float mean_all = block_sum4(tid < kD ? mean : 0.0f, red, tid) / (float)kD;
    const float d = mean - mean_all;
    const float m2_all = block_sum4(tid < kD ? m2 + (float)N * d * d : 0.0f, red, tid);
    const float rstd = rsqrtf(m2_all / ((float)N * (float)kD - 1.0f) + 1e-5f);
    if (tid < kD) {
      ab[tid] = rstd;
      ab[kD + tid] = -mean_all * rstd;
    }
  }
  __syncthreads();
  const E* src = ypre + ((int64_t)b * N + n0) * kD;
  E* dst = xout + ((int64_t)b * N + n0) * kD;
  for (int i = tid; i < valid * 16; i += kThreads) {
    const int c0 = 8 * (i & 15);
    vec8<E> v = *reinterpret_cast<const vec8<E>*>(src + (int64_t)(i >> 4) * kD + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (E)fmaf((float)v[e], ab[c0 + e], ab[kD + c0 + e]);
    *reinterpret_cast<vec8<E>*>(dst + (int64_t)(i >> 4) * kD + c0) = v;
  }
}

// ---- training: the input gradient of x + MLP(x) on one tile of 128 token rows (nn/mlp.py:52-61 under nn/ops.py:9-15) ----
//   dh = (dy . W2) * [h > 0]   [M,512]  (written ONCE: the two weight-gradient launches read it)
//   dx = dh . W1 + dy          [M,128]
// The per-op path ran these as two GEMM launches with dh written (420 MB at 4096 x 100 rows) and read back; here the
// 512-wide gradient goes through LDS in four chunks exactly as the hidden does in tok16_mlp_kernel: per layer 1.05 GB of
// traffic instead of 1.58. Weights: W2^T ([512,128]) and W1^T ([128,512]) in pack_weight's fragment order.
template <int ROWS, typename E>
__device__ inline void rows_load_strided(E* xs, const E* src, int64_t row_stride, int valid, int tid) {
  for (int i = tid; i < ROWS * 16; i += kThreads) {
    const int row = i >> 4, c16 = i & 15;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < valid) v = *reinterpret_cast<const uint4*>(src + (int64_t)row * row_stride + 8 * c16);
    *reinterpret_cast<uint4*>(xs + row * kRS + 8 * c16) = v;
  }
}

// TT token tiles of 32 rows per workgroup. The kernel is a chain of four (stage h chunk -> product -> mask -> store ->
// product) rounds per tile whose HBM round trips are exposed to the workgroup, so what hides them is the number of
// workgroups a CU holds: TT = 2 (64 rows: 35 KB of LDS, <= 168 registers) keeps three of them resident where TT = 4 keeps two.
template <typename E, int TT>
__global__ void __launch_bounds__(kThreads, TT <= 2 ? 3 : 2) tok16_mlp_bwd_kernel(const E* __restrict__ dy, const E* __restrict__ h, int M,
                                                                                  const E* __restrict__ w2t, const E* __restrict__ w1t,
                                                                                  E* __restrict__ dh, E* __restrict__ dx) {
  constexpr int kRows = 32 * TT;
  extern __shared__ __align__(16) unsigned char smem[];
  E* xs = reinterpret_cast<E*>(smem);  // dy tile
  E* ys = xs + kRows * kRS;            // h chunk (the ReLU mask), then the dh chunk
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int r0 = kRows * blockIdx.x, valid = min(kRows, M - r0);
  vec8<E> wf[8];
  load_wfrags(wf, w2t, 8, w, 0, lane);
  rows_load_strided<kRows>(xs, dy + (int64_t)r0 * kD, kD, valid, tid);
  f32x16 acc[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) acc[tt] = zero16();
  // the h chunk of round c + 1 travels in registers while round c computes (2 TT 16-byte pieces per thread)
  constexpr int kPieces = kRows * 16 / kThreads;
  constexpr bool kAhead = TT <= 2;
  uint4 hq[kAhead ? kPieces : 1];
  auto fetch = [&](int c) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));  // (row addresses per call: kept across the chunk loop two of them were spilled, and a scratch reload waits with vmcnt(0))
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const int i = t_ + kThreads * p, row = i >> 4, c16 = i & 15;
      const uint4 hv = *reinterpret_cast<const uint4*>(h + (int64_t)(r0 + min(row, valid - 1)) * kFF + kD * c + 8 * c16);  // (unconditional: a counted load)
      hq[kAhead ? p : 0] = row < valid ? hv : make_uint4(0, 0, 0, 0);
    }
  };
  if constexpr (kAhead) fetch(0);
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    if (c > 0) __syncthreads();  // every wave is done reading the previous dh chunk
    if constexpr (kAhead) {
#pragma unroll
      for (int p = 0; p < kPieces; ++p) {
        const int i = tid + kThreads * p;
        *reinterpret_cast<uint4*>(ys + (i >> 4) * kRS + 8 * (i & 15)) = hq[p];
      }
      if (c < 3) fetch(c + 1);
    } else {
      rows_load_strided<kRows>(ys, h + (int64_t)r0 * kFF + kD * c, kFF, valid, tid);
    }
    __syncthreads();
    f32x16 g[TT];
    gemm_t<TT, true, true, 1>(g, wf, xs, lane, w1t, 32, w, 8 * c, zero16());  // next: dx += dh chunk . W1 (k-steps 8 c ..)
    // the ReLU mask from this wave's own 32 hidden columns of the staged h chunk; dh goes back over them
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const vec4<E> hv = *reinterpret_cast<const vec4<E>*>(ys + (32 * tt + l31) * kRS + 32 * w + 8 * q + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[tt][4 * q + e] = ((float)hv[e] > 0.0f) ? g[tt][4 * q + e] : 0.0f;
      }
    }
    store_t<TT>(ys, g, 32 * w, lane);
    __syncthreads();
    rows_out<TT>(ys, dh + (int64_t)r0 * kFF + kD * c, kFF, valid, tid);  // (fixed-trip stores: the product below waits for its weights only)
    gemm_t<TT, true, false, 1>(acc, wf, ys, lane, w2t, 8, c < 3 ? 4 * (c + 1) + w : w, 0, acc[0]);  // (last chunk: a dummy fetch)
  }
  // + dy (the skip connection's gradient), this wave's 32 columns of the dy tile, then out through the same tile
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const vec4<E> x = *reinterpret_cast<const vec4<E>*>(xs + (32 * tt + l31) * kRS + 32 * w + 8 * q + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[tt][4 * q + e] += (float)x[e];
    }
  }
  store_t<TT>(xs, acc, 32 * w, lane);
  __syncthreads();
  rows_out<TT>(xs, dx + (int64_t)r0 * kD, kD, valid, tid);
}

// cache planes (16-bit or fp32) and fp32 context tables of one token tile, as the fused kernel's fold writes them
template <typename E>
__global__ void __launch_bounds__(kThreads, 2) tok16_fold_kernel(const E* __restrict__ x, const rl4co_am_encoder_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  E* xs = reinterpret_cast<E*>(smem);
  E* ys = xs + kTok * kRS;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y, n0 = kTok * blockIdx.x, N = a.N, valid = min(kTok, N - n0);
  const E* wf_all = static_cast<const E*>(a.wfold_packed);
  vec8<E> wf[8];
  load_wfrags(wf, wf_all, 8, w, 0, lane);
  tok_load(xs, x, b, n0, N, tid);
  __syncthreads();
  const int nblocks = (a.env == RL4CO_ENV_TSP) ? 5 : 4;
#pragma unroll 1
  for (int blk = 0; blk < nblocks; ++blk) {
    f32x16 acc[kTokT];
    gemm_t<kTokT, true, true, 1>(acc, wf, xs, lane, wf_all + (int64_t)(blk + 1 < nblocks ? blk + 1 : 0) * kD * kD, 8, w, 0, zero16());
    if (blk < 3 && a.cache_dtype != RL4CO_DT_F32) {
      store_t<kTokT>(ys, acc, 32 * w, lane);
      __syncthreads();
      tok_store(ys, static_cast<E*>(a.kvl) + (int64_t)blk * a.kvl_plane_stride + (int64_t)b * a.kvl_batch_stride + (int64_t)n0 * kD, kD, valid, tid);
      __syncthreads();
    } else {
      float* out;
      if (blk < 3) out = static_cast<float*>(a.kvl) + (int64_t)blk * a.kvl_plane_stride + (int64_t)b * a.kvl_batch_stride;
      else if (a.env == RL4CO_ENV_TSP) out = static_cast<float*>(blk == 3 ? a.ctx_first : a.ctx_cur) + (int64_t)b * N * kD;
      else out = static_cast<float*>(a.ctx_cur) + (int64_t)b * N * kD;
      out += (int64_t)n0 * kD;
      constexpr int kFS = kD + 4;  // fp32 staging row stride: 64 token rows fit the 34 KB of ys
      float* fs = reinterpret_cast<float*>(ys);
#pragma unroll
      for (int p = 0; p < kTokT / 2; ++p) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x16& t = acc[2 * p + t2];
            *reinterpret_cast<float4*>(fs + (32 * t2 + l31) * kFS + 32 * w + 8 * c + 4 * hi) = make_float4(t[4 * c], t[4 * c + 1], t[4 * c + 2], t[4 * c + 3]);
          }
        }
        __syncthreads();
        const int rows = min(64, valid - 64 * p);
        if (rows > 0) {  // (uniform; a fixed number of unconditional stores: see rows_out)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = tid + kThreads * j, row = min(i >> 5, rows - 1), c4 = i & 31;
            *reinterpret_cast<float4*>(out + (int64_t)(64 * p + row) * kD + 4 * c4) = *reinterpret_cast<const float4*>(fs + row * kFS + 4 * c4);
          }
        }
        __syncthreads();
      }
    }
  }
}

template <typename E>
__global__ void widen_kernel(const E* __restrict__ src, int64_t n8, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const vec8<E> v = *reinterpret_cast<const vec8<E>*>(src + 8 * i);
  float4 lo = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]), hi4 = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
  *reinterpret_cast<float4*>(dst + 8 * i) = lo;
  *reinterpret_cast<float4*>(dst + 8 * i + 4) = hi4;
}

template <typename E>
int launch_tokens16(const rl4co_am_encoder_args& a, void* workspace, hipStream_t s) {
  const int N = a.N;
  const int64_t mx = ((int64_t)a.B * N * kD + 63) / 64 * 64;
  E* x0 = static_cast<E*>(workspace);
  E* x1 = x0 + mx;
  E* att = x0 + 2 * mx;
  E* qkv = x0 + 3 * mx;  // [B N, 384]
  E* ypre = x0 + 6 * mx;                                        // pre-norm sums (instance / layer norm only)
  uint32_t* bound = reinterpret_cast<uint32_t*>(x0 + 7 * mx);  // [B][8 heads][q, k] fp32 bit patterns
  float* stats = reinterpret_cast<float*>(bound + (int64_t)a.B * 8 * 2);  // [B][tiles][mean 128 | M2 128]
  const dim3 grid((N + kTok - 1) / kTok, a.B), block(kThreads);
  const int lds_tile = kTok * kRS * (int)sizeof(E);
  const int lds_init = lds_tile + 6 * N * 4 + 64, lds_qkv = 2 * lds_tile + 3 * kD * 4, lds_mlp = 2 * lds_tile + kFF * 4;
  RL4CO_REQUIRE(lds_init <= 80 * 1024);
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_init_embed_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_init));
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_qkv_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_qkv));
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_mlp_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_mlp));
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_fold_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * lds_tile));
  if (a.norm != 0) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_attn_half_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * lds_tile));
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_ffn_half_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_mlp));
  }
  hipLaunchKernelGGL(tok16_init_embed_kernel<E>, grid, block, lds_init, s, a, x0);
  E *xin = x0, *xout = x1;
  const E* wqkv = static_cast<const E*>(a.wqkv_packed);
  const E* wo = static_cast<const E*>(a.wo_packed);
  const E* w1 = static_cast<const E*>(a.w1_packed);
  const E* w2 = static_cast<const E*>(a.w2_packed);
  const bool half = a.act_dtype == RL4CO_DT_F16;
  for (int layer = 0; layer < a.num_layers; ++layer) {
    RL4CO_HIP_TRY(hipMemsetAsync(bound, 0, (size_t)a.B * 8 * 2 * 4, s));
    hipLaunchKernelGGL(tok16_qkv_kernel<E>, grid, block, lds_qkv, s, xin, N, wqkv + (int64_t)layer * 3 * kD * kD, a.bqkv + layer * 3 * kD, qkv, bound);
    const int st = rl4co_attn_flash_pre(half ? RL4CO_DT_F16 : RL4CO_DT_BF16, qkv, reinterpret_cast<const float*>(bound), a.B, N, att, s);
    if (st != RL4CO_OK) return st;
    if (a.norm == 0) {
      hipLaunchKernelGGL(tok16_mlp_kernel<E>, grid, block, lds_mlp, s, xin, att, N, wo + (int64_t)layer * kD * kD, w1 + (int64_t)layer * kFF * kD,
                         w2 + (int64_t)layer * kD * kFF, a.b1 + layer * kFF, a.n1_scale + layer * kD, a.n1_shift + layer * kD,
                         a.n2_scale + layer * kD, a.n2_shift + layer * kD, xout);
    } else {
      // instance / layer norm: each half stops before its norm; the apply kernel sees the whole instance's statistics.
      // (layer norm: the bias of the GEMM in front of the norm arrives in the shift slot, encoder.py)
      const float* pb1 = a.norm == 2 ? a.n1_shift + layer * kD : nullptr;
      const float* pb2 = a.norm == 2 ? a.n2_shift + layer * kD : nullptr;
      hipLaunchKernelGGL(tok16_attn_half_kernel<E>, grid, block, 2 * lds_tile, s, xin, att, N, wo + (int64_t)layer * kD * kD, pb1, ypre, stats);
      hipLaunchKernelGGL(tok16_norm_apply_kernel<E>, grid, block, 0, s, ypre, stats, N, a.norm, a.n1_scale + layer * kD, a.n1_shift + layer * kD, xout);
      hipLaunchKernelGGL(tok16_ffn_half_kernel<E>, grid, block, lds_mlp, s, xout, N, w1 + (int64_t)layer * kFF * kD, w2 + (int64_t)layer * kD * kFF,
                         a.b1 + layer * kFF, pb2, ypre, stats);
      hipLaunchKernelGGL(tok16_norm_apply_kernel<E>, grid, block, 0, s, ypre, stats, N, a.norm, a.n2_scale + layer * kD, a.n2_shift + layer * kD, xout);
    }
    E* t = xin;
    xin = xout;
    xout = t;
  }
  hipLaunchKernelGGL(tok16_fold_kernel<E>, grid, block, 2 * lds_tile, s, xin, a);
  if (a.q_bias) {
    const int st = rl4co_am_fold_tables_f32(xin, a.act_dtype, a.B, N, nullptr, 0, nullptr, a.w_fixed, a.q_bias, s);
    if (st != RL4CO_OK) return st;
  }
  if (a.hidden) {  // fp32 copy of the final embeddings (return_hidden)
    const int64_t n8 = (int64_t)a.B * N * kD / 8;
    hipLaunchKernelGGL(widen_kernel<E>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, xin, n8, a.hidden);
  }
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

template <typename E, int TT, int VR4>
int launch_encoder(const rl4co_am_encoder_args& a, hipStream_t stream) {
  const int lds = kEncLds;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_encoder_kernel<E, TT, VR4, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((am_encoder_kernel<E, TT, VR4>), dim3(a.B), dim3(kThreads), lds, stream, a, TrainSave<E>{});
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

template <typename E, int TT>
int launch_encoder_layer(const rl4co_am_encoder_args& a, hipStream_t stream) {
  // whole-instance ("layer") statistics: its own instantiation — the batch / instance kernel's code stays as tuned — with
  // the generic key-tile masks (VR4 = 4 serves every N of the tile count)
  const int lds = kEncLds;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_encoder_kernel<E, TT, 4, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((am_encoder_kernel<E, TT, 4, false, true>), dim3(a.B), dim3(kThreads), lds, stream, a, TrainSave<E>{});
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

template <typename E, int TT>
int launch_encoder_tiles(const rl4co_am_encoder_args& a, hipStream_t stream) {
  if (a.norm == 2) return launch_encoder_layer<E, TT>(a, stream);
  const int vr4 = (a.N - 32 * (TT - 1) + 7) / 8;  // valid 4-register groups of the last key tile
  switch (vr4) {
    case 1: return launch_encoder<E, TT, 1>(a, stream);
    case 2: return launch_encoder<E, TT, 2>(a, stream);
    case 3: return launch_encoder<E, TT, 3>(a, stream);
    default: return launch_encoder<E, TT, 4>(a, stream);
  }
}

template <typename E>
int launch_encoder_elem(const rl4co_am_encoder_args& a, hipStream_t s) {
#ifdef RL4CO_ENC_PROBE  // tools/enc_variants.sh: one instantiation (TSP-100, batch norm) so that a probe build takes seconds
  RL4CO_REQUIRE((a.N + 31) / 32 == 4 && (a.N - 96 + 7) / 8 == 1 && a.norm != 2);
  return launch_encoder<E, 4, 1>(a, s);
#else
  switch ((a.N + 31) / 32) {
    case 1: return launch_encoder_tiles<E, 1>(a, s);
    case 2: return launch_encoder_tiles<E, 2>(a, s);
    case 3: return launch_encoder_tiles<E, 3>(a, s);
    default: return launch_encoder_tiles<E, 4>(a, s);
  }
#endif
}

template <typename E, int TT, int VR4>
int launch_train(const rl4co_am_encoder_args& a, const TrainSave<E>& ts, hipStream_t stream) {
  const int lds = kEncLds;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_encoder_kernel<E, TT, VR4, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((am_encoder_kernel<E, TT, VR4, true>), dim3(a.B), dim3(kThreads), lds, stream, a, ts);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
template <typename E, int TT>
int launch_train_tiles(const rl4co_am_encoder_args& a, const TrainSave<E>& ts, hipStream_t stream) {
  switch ((a.N - 32 * (TT - 1) + 7) / 8) {
    case 1: return launch_train<E, TT, 1>(a, ts, stream);
    case 2: return launch_train<E, TT, 2>(a, ts, stream);
    case 3: return launch_train<E, TT, 3>(a, ts, stream);
    default: return launch_train<E, TT, 4>(a, ts, stream);
  }
}
template <typename E>
int launch_train_elem(const rl4co_am_encoder_args& a, const rl4co_am_train_save& sv, hipStream_t s) {
  TrainSave<E> ts;
  ts.x0 = static_cast<const E*>(sv.x0);
  ts.out = static_cast<E*>(sv.out);
  ts.qkv = static_cast<E*>(sv.qkv);
  ts.att = static_cast<E*>(sv.att);
  ts.y1 = static_cast<E*>(sv.y1);
  ts.x1 = static_cast<E*>(sv.x1);
  ts.h = static_cast<E*>(sv.h);
  ts.y2 = static_cast<E*>(sv.y2);
  ts.lse = sv.lse;
  ts.stats = sv.stats;
#ifdef RL4CO_ENC_PROBE
  RL4CO_REQUIRE((a.N + 31) / 32 == 4 && (a.N - 96 + 7) / 8 == 1);
  return launch_train<E, 4, 1>(a, ts, s);
#else
  switch ((a.N + 31) / 32) {
    case 1: return launch_train_tiles<E, 1>(a, ts, s);
    case 2: return launch_train_tiles<E, 2>(a, ts, s);
    case 3: return launch_train_tiles<E, 3>(a, ts, s);
    default: return launch_train_tiles<E, 4>(a, ts, s);
  }
#endif
}

}  // namespace

extern "C" int rl4co_am_encoder_train_fwd(const rl4co_am_encoder_args* args, const rl4co_am_train_save* save, void* stream) {
  RL4CO_REQUIRE(args != nullptr && save != nullptr);
  const rl4co_am_encoder_args& a = *args;
  RL4CO_REQUIRE(a.B > 0 && a.N >= 2 && a.N <= 128);
  RL4CO_REQUIRE(a.num_layers >= 1 && a.norm == 1);  // instance norm: batch statistics couple the instances
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_BF16 || a.act_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(a.wqkv_packed && a.wo_packed && a.w1_packed && a.w2_packed);
  RL4CO_REQUIRE(a.bqkv && a.b1 && a.n1_scale && a.n1_shift && a.n2_scale && a.n2_shift);
  RL4CO_REQUIRE(save->x0 && save->out && save->qkv && save->att && save->y1 && save->x1 && save->h && save->y2 && save->lse && save->stats);
  hipStream_t s = rl4co::as_stream(stream);
#ifdef RL4CO_ENC_PROBE
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_BF16);
  return launch_train_elem<__bf16>(a, *save, s);
#else
  return a.act_dtype == RL4CO_DT_F16 ? launch_train_elem<_Float16>(a, *save, s) : launch_train_elem<__bf16>(a, *save, s);
#endif
}

template <typename E, int TT>
int launch_mlp_bwd(const void* dy, const void* h, int64_t M, const void* w2t, const void* w1t, void* dh, void* dx, hipStream_t s) {
  constexpr int rows = 32 * TT;
  const int lds = 2 * rows * kRS * (int)sizeof(E);
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_mlp_bwd_kernel<E, TT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((tok16_mlp_bwd_kernel<E, TT>), dim3((unsigned)((M + rows - 1) / rows)), dim3(kThreads), lds, s, static_cast<const E*>(dy),
                     static_cast<const E*>(h), (int)M, static_cast<const E*>(w2t), static_cast<const E*>(w1t), static_cast<E*>(dh),
                     static_cast<E*>(dx));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_mlp_input_grad(const void* dy, const void* h, int64_t M, const void* w2t_packed, const void* w1t_packed, int dtype,
                                    void* dh, void* dx, void* stream) {
  RL4CO_REQUIRE(dy && h && w2t_packed && w1t_packed && dh && dx);
  RL4CO_REQUIRE(M > 0 && M < (int64_t)1 << 31);
  RL4CO_REQUIRE(dtype == RL4CO_DT_BF16 || dtype == RL4CO_DT_F16);
  hipStream_t s = rl4co::as_stream(stream);
  static const int tiles = [] {  // (probe knob: RL4CO_MLP_BWD_TILES = 2 | 4 token tiles of 32 rows per workgroup)
    const char* e = getenv("RL4CO_MLP_BWD_TILES");
    return (e && e[0] == '4') ? 4 : ((e && e[0] == '1') ? 1 : 2);
  }();
  if (tiles == 4)
    return dtype == RL4CO_DT_F16 ? launch_mlp_bwd<_Float16, 4>(dy, h, M, w2t_packed, w1t_packed, dh, dx, s)
                                 : launch_mlp_bwd<__bf16, 4>(dy, h, M, w2t_packed, w1t_packed, dh, dx, s);
  if (tiles == 1)
    return dtype == RL4CO_DT_F16 ? launch_mlp_bwd<_Float16, 1>(dy, h, M, w2t_packed, w1t_packed, dh, dx, s)
                                 : launch_mlp_bwd<__bf16, 1>(dy, h, M, w2t_packed, w1t_packed, dh, dx, s);
  return dtype == RL4CO_DT_F16 ? launch_mlp_bwd<_Float16, 2>(dy, h, M, w2t_packed, w1t_packed, dh, dx, s)
                               : launch_mlp_bwd<__bf16, 2>(dy, h, M, w2t_packed, w1t_packed, dh, dx, s);
}

extern "C" int rl4co_am_encoder_max_nodes(void) { return 128; }

extern "C" int rl4co_am_encoder(const rl4co_am_encoder_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_encoder_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_PDP);
  RL4CO_REQUIRE(a.B > 0 && a.N >= 2 && a.N <= 128);
  RL4CO_REQUIRE(a.num_layers >= 1 && a.norm >= 0 && a.norm <= 2);  // batch (eval, folded affine) | instance | layer
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_BF16 || a.act_dtype == RL4CO_DT_F16);
  // planes: fp32, or the 16-bit type the activations are computed in
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == a.act_dtype);
  RL4CO_REQUIRE(a.locs && a.w_init && a.b_init);
  RL4CO_REQUIRE(a.env != RL4CO_ENV_CVRP || (a.demand && a.w_depot && a.b_depot));
  RL4CO_REQUIRE(a.env != RL4CO_ENV_PDP || (a.w_depot && a.b_depot && a.w_extra && a.b_extra && (a.N - 1) % 2 == 0));
  RL4CO_REQUIRE(a.wqkv_packed && a.wo_packed && a.w1_packed && a.w2_packed && a.wfold_packed);
  RL4CO_REQUIRE(a.bqkv && a.bo && a.b1 && a.b2 && a.n1_scale && a.n1_shift && a.n2_scale && a.n2_shift);
  RL4CO_REQUIRE(a.kvl && a.ctx_cur && (a.env != RL4CO_ENV_TSP || a.ctx_first));
  // (r06) context tables in the activations' 16-bit type beside 16-bit planes (16-byte rows for the row copies)
  RL4CO_REQUIRE(a.ctx_dtype == RL4CO_DT_F32 || (a.ctx_dtype == a.act_dtype && a.cache_dtype == a.act_dtype));
  RL4CO_REQUIRE(a.ctx_dtype == RL4CO_DT_F32 || ((reinterpret_cast<uintptr_t>(a.ctx_cur) | reinterpret_cast<uintptr_t>(a.ctx_first)) & 15) == 0);
  RL4CO_REQUIRE(a.q_bias == nullptr || a.w_fixed != nullptr);
  RL4CO_REQUIRE(a.kvl_batch_stride >= (int64_t)a.N * kD && a.kvl_plane_stride >= a.kvl_batch_stride);
  hipStream_t s = rl4co::as_stream(stream);
#ifdef RL4CO_ENC_PROBE
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_BF16);
  return launch_encoder_elem<__bf16>(a, s);
#else
  return a.act_dtype == RL4CO_DT_F16 ? launch_encoder_elem<_Float16>(a, s) : launch_encoder_elem<__bf16>(a, s);
#endif
}

extern "C" int64_t rl4co_am_encoder_tokens16_workspace(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const int64_t mx = ((int64_t)B * N * kD + 63) / 64 * 64;
  return 7 * mx * 2 + (int64_t)B * 8 * 2 * 4 + (int64_t)B * ((N + kTok - 1) / kTok) * kStatFloats * 4;
}

extern "C" int rl4co_am_encoder_tokens16(const rl4co_am_encoder_args* args, void* workspace, int64_t workspace_bytes, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_encoder_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_PDP);
  RL4CO_REQUIRE(a.B > 0 && a.N >= 2 && a.B <= 65535);
  RL4CO_REQUIRE(a.num_layers >= 1 && a.norm >= 0 && a.norm <= 2);  // instance / layer norm: split sub-blocks + apply kernel
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_BF16 || a.act_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == a.act_dtype);
  RL4CO_REQUIRE(a.locs && a.w_init && a.b_init);
  RL4CO_REQUIRE(a.env != RL4CO_ENV_CVRP || (a.demand && a.w_depot && a.b_depot));
  RL4CO_REQUIRE(a.env != RL4CO_ENV_PDP || (a.w_depot && a.b_depot && a.w_extra && a.b_extra && (a.N - 1) % 2 == 0));
  RL4CO_REQUIRE(a.wqkv_packed && a.wo_packed && a.w1_packed && a.w2_packed && a.wfold_packed);
  RL4CO_REQUIRE(a.bqkv && a.b1 && a.n1_scale && a.n1_shift && a.n2_scale && a.n2_shift);
  RL4CO_REQUIRE(a.kvl && a.ctx_cur && (a.env != RL4CO_ENV_TSP || a.ctx_first));
  RL4CO_REQUIRE(a.ctx_dtype == RL4CO_DT_F32);  // (the token path's context tables come from the fp32 fold kernel)
  RL4CO_REQUIRE(a.q_bias == nullptr || a.w_fixed != nullptr);
  RL4CO_REQUIRE(a.kvl_batch_stride >= (int64_t)a.N * kD && a.kvl_plane_stride >= a.kvl_batch_stride);
  RL4CO_REQUIRE(workspace != nullptr && workspace_bytes >= rl4co_am_encoder_tokens16_workspace(a.B, a.N));
  RL4CO_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  hipStream_t s = rl4co::as_stream(stream);
  return a.act_dtype == RL4CO_DT_F16 ? launch_tokens16<_Float16>(a, workspace, s) : launch_tokens16<__bf16>(a, workspace, s);
}

// The init embeddings alone (reference: `init_embeds` of AttentionModelPolicy.forward(return_init_embeds=True),
// zoo/am/encoder.py:84-103, env_embeddings/init.py): the token path's first launch, i.e. the same init_embed_rows16 the
// fused kernel runs — so what is handed back is what the encoder computed from, in the activations' type.
extern "C" int rl4co_am_encoder_init_embeds16(const rl4co_am_encoder_args* args, void* out, void* stream) {
  RL4CO_REQUIRE(args != nullptr && out != nullptr);
  const rl4co_am_encoder_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_PDP);
  RL4CO_REQUIRE(a.B > 0 && a.N >= 2 && a.B <= 65535);
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_BF16 || a.act_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(a.locs && a.w_init && a.b_init);
  RL4CO_REQUIRE(a.env != RL4CO_ENV_CVRP || (a.demand && a.w_depot && a.b_depot));
  RL4CO_REQUIRE(a.env != RL4CO_ENV_PDP || (a.w_depot && a.b_depot && a.w_extra && a.b_extra && (a.N - 1) % 2 == 0));
  hipStream_t s = rl4co::as_stream(stream);
  const dim3 grid((a.N + kTok - 1) / kTok, a.B), block(kThreads);
  const bool half = a.act_dtype == RL4CO_DT_F16;
  const int lds_init = kTok * kRS * 2 + 6 * a.N * 4 + 64;
  RL4CO_REQUIRE(lds_init <= 80 * 1024);
  if (half) {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_init_embed_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_init));
    hipLaunchKernelGGL(tok16_init_embed_kernel<_Float16>, grid, block, lds_init, s, a, static_cast<_Float16*>(out));
  } else {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tok16_init_embed_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_init));
    hipLaunchKernelGGL(tok16_init_embed_kernel<__bf16>, grid, block, lds_init, s, a, static_cast<__bf16*>(out));
  }
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
