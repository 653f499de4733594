// am_encoder_f32.hip — the fused AttentionModel encoder + decoder-cache fold in EXACT fp32 on the gfx950 matrix cores.
//
// The bit-identical configuration of the rollout (BASELINE north_star: "greedy tour lengths bit-identical to the
// reference") needs node embeddings whose round-off is of the reference's own fp32 class: this kernel is the fp32
// sibling of am_encoder.hip (same fusion: one workgroup per instance, residual stream in LDS, weights streamed from L2
// in pre-packed fragment order, only the coordinates in and the cache planes out through HBM) on
// v_mfma_f32_16x16x4_f32 — f32 operands, f32 accumulate, bit-for-bit a k-ordered fmaf chain
// (/opt/skills/guides/cdna_hip_programming.md §3), 157 TFLOP/s chip peak. It replaces, for inference rollouts:
//   TSPInitEmbedding / VRPInitEmbedding / ...  models/nn/env_embeddings/init.py:55-68,115-153,254-360
//   GraphAttentionNetwork (L x [x + MHA(x) -> Norm -> x + MLP(x) -> Norm])   models/nn/graph/attnnet.py:16-106
//   MultiHeadAttention                          models/nn/attention.py:64-134 (F.scaled_dot_product_attention, fp32)
//   Normalization (batch, eval / instance)      models/nn/ops.py:30-54
//   AttentionModelDecoder._precompute_cache     models/zoo/am/decoder.py:201-228 (folded form, rl4co_amd/cache.py)
//
// Work split — 8 waves, tokens in TT tiles of 16 (N <= 128), wave w owns output dims 16 w .. 16 w + 15 of every
// 128-wide GEMM, which is exactly attention head w:
//   * "transposed" GEMMs  Out^T[dim][token] = W[dim][k] . X^T[k][token]: A = weight fragment (one 16-byte load per lane
//     and 16-k chunk: W[16 tile + c][16 j + 4 g + s], lane = 16 g + c, reused over the TT token tiles), B = activation
//     row xs[token c][16 j + 4 g + s] (one ds_read_b128 per chunk); component s of both feeds MFMA step s, so the k-slot
//     permutation is the same on both operands and cancels.
//   * attention is wave-private and needs NO data movement at all: the accumulator layout of a 16x16 tile (lane (c, g),
//     register r: row 4 g + r, column c) IS the operand layout of the next product —
//       Q^T, K^T tiles (transposed form)   S^T[key][query] = K . Q^T : A = K^T registers, B = Q^T registers
//       V tiles (plain form, A = X rows)   O^T[dim][query] = V^T . P^T : A = V registers,  B = exp(S^T) registers
//     softmax over the keys of a query column: in-lane over registers and key tiles + two permlane swaps across g.
//   * exchanges through LDS where a GEMM needs all 128 input dims: attention output -> out-proj, FFN hidden chunk -> FFN2.
// Arithmetic mirrors ATen's CPU kernels where the order is knowable: GEMM then + bias, x + branch, batch / instance norm
// as x * alpha + beta with alpha = invstd * gamma, beta = bias - mean * alpha (native/cpu/batch_norm_kernel.cpp), two-pass
// instance statistics, softmax as exp(s - max) / sum with the 1 / sqrt(16) scale (a power of two: exact) folded into Wq.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kFF = 512;
constexpr int kRS = kD + 4;  // LDS row stride (floats): 528-byte rows — the 16 lanes of a ds_read_b128 group (one g) hit 16 distinct bank quads
constexpr int kThreads = 512;
constexpr int kBiasFloats = 3 * kD + kFF + kD + kD;  // per layer: bqkv [384] | b1 [512] | bo [128] | b2 [128]
constexpr float kLog2e = 1.4426950408889634f;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f32x4 mfma4(float a, float b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ inline f32x4 zero4() {
  f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
  return z;
}

// packed weight fragment: [tile of 16 output dims][16-k chunk][64 lanes][4] fp32 (rl4co_amd/encoder.py: pack_weight_f32)
__device__ inline f32x4 load_w(const float* packed, int chunks_total, int tile, int j, int lane) {
  return *reinterpret_cast<const f32x4*>(packed + (((int64_t)tile * chunks_total + j) * 64 + lane) * 4);
}
__device__ inline void load_wfrags(f32x4 (&wf)[8], const float* packed, int chunks_total, int tile, int j0, int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) wf[j] = load_w(packed, chunks_total, tile, j0 + j, lane);
}

// acc[tt] += (W tile) . X^T over K = 128 (8 chunks of 16, four MFMA steps each). W_IS_A: transposed form (rows = dims,
// columns = tokens); otherwise plain form (rows = tokens, columns = dims). The weight fragments `wf` were requested one
// call ahead; as soon as chunk j has fed its last MFMA its registers take chunk j of the NEXT GEMM (nxt; nullptr: none),
// whose L2 round trip hides under the rest of this call. A chunk is 4 TT MFMAs of 32 cycles; the activation fragments
// follow the same rule tile by tile (next use (TT - 1) MFMAs = ~200 cycles later; the SIMD's other wave covers the rest).
template <int TT, bool W_IS_A>
__device__ inline void gemm16(f32x4 (&acc)[TT], f32x4 (&wf)[8], const float* xs, int lane, const float* nxt, int nxt_chunks,
                              int nxt_tile, int nxt_j0) {
  const float* xrow = xs + (lane & 15) * kRS + 4 * (lane >> 4);
  f32x4 x[2][TT];  // activation fragments of chunk j + 1 requested while chunk j computes (4 TT MFMAs = ~900 cycles)
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) x[0][tt] = *reinterpret_cast<const f32x4*>(xrow + 16 * tt * kRS);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j + 1 < 8) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) x[(j + 1) & 1][tt] = *reinterpret_cast<const f32x4*>(xrow + 16 * tt * kRS + 16 * (j + 1));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        acc[tt] = W_IS_A ? mfma4(wf[j][s], x[j & 1][tt][s], acc[tt]) : mfma4(x[j & 1][tt][s], wf[j][s], acc[tt]);
    }
    if (nxt) wf[j] = load_w(nxt, nxt_chunks, nxt_tile, nxt_j0 + j, lane);
    __builtin_amdgcn_sched_barrier(0);  // chunk by chunk: hoisted, the LDS reads of all eight chunks would be live at once
  }
}

// transposed-form tile -> LDS rows [token][dim]: the lane's four registers are four consecutive dims of its token
template <int TT>
__device__ inline void store_t(float* ys, const f32x4 (&acc)[TT], int dim0, int lane) {
  float* row = ys + (lane & 15) * kRS + dim0 + 4 * (lane >> 4);
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) *reinterpret_cast<f32x4*>(row + 16 * tt * kRS) = acc[tt];
}

// y = Norm(x + (y + bias)) for the wave's 16-dim tile, back into xs. nn/ops.py:9-15 (skip), 30-54 (norm).
template <int TT>
__device__ inline void residual_norm(float* xs, f32x4 (&y)[TT], int dim0, const float* bias_lds, const float* na, const float* nb,
                                     int norm, int N, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_lds + dim0 + 4 * g);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(na + dim0 + 4 * g);
  const f32x4 be = *reinterpret_cast<const f32x4*>(nb + dim0 + 4 * g);
  float* row = xs + c * kRS + dim0 + 4 * g;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + 16 * tt * kRS);
    y[tt] = x + (y[tt] + bias);
  }
  f32x4 alpha, beta;
  if (norm == 1) {
    // instance norm: statistics per (instance, channel) over the N nodes, two passes as ATen's CPU kernel takes them
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = 0.0f;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) s += (16 * tt + c < N) ? y[tt][r] : 0.0f;
      s = rl4co::bfly_sum<1, 16>(s);
      const float mean = s / (float)N;
      float v = 0.0f;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float d = y[tt][r] - mean;
        v += (16 * tt + c < N) ? d * d : 0.0f;
      }
      v = rl4co::bfly_sum<1, 16>(v);
      const float invstd = 1.0f / sqrtf(v / (float)N + 1e-5f);
      alpha[r] = invstd * ga[r];
      beta[r] = be[r] - mean * alpha[r];
    }
  } else {
    alpha = ga;  // batch norm in eval mode: (alpha, beta) from the running statistics, built on the host the same way
    beta = be;
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    y[tt] = y[tt] * alpha + beta;  // (-ffp-contract=off: a multiply and an add, as the vectorised CPU kernel)
    *reinterpret_cast<f32x4*>(row + 16 * tt * kRS) = y[tt];
  }
}

template <int TT>
__global__ void __launch_bounds__(kThreads) am_encoder_f32_kernel(const rl4co_am_encoder_args a) {
  constexpr int kRows = 16 * TT;
  extern __shared__ __align__(16) unsigned char smem[];
  float* xs = reinterpret_cast<float*>(smem);  // residual stream [kRows][kRS]
  float* ys = xs + kRows * kRS;                // attention output -> FFN hidden chunk -> plane staging
  float* meanv = ys + kRows * kRS;             // [128]
  float* bl = meanv + kD;                      // [kBiasFloats] this layer's biases
  auto stage_biases = [&](int layer) {
    for (int i = threadIdx.x; i < kBiasFloats; i += kThreads) {
      float v;
      if (i < 3 * kD) v = a.bqkv[layer * 3 * kD + i];
      else if (i < 3 * kD + kFF) v = a.b1[layer * kFF + i - 3 * kD];
      else if (i < 4 * kD + kFF) v = a.bo[layer * kD + i - 3 * kD - kFF];
      else v = a.b2[layer * kD + i - 4 * kD - kFF];
      bl[i] = v;
    }
  };

  int tid = threadIdx.x;
  int w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;  // (not const: see the top of the layer loop)
  const int b = blockIdx.x;
  const int N = a.N;

  const float* wqkv_all = static_cast<const float*>(a.wqkv_packed);
  const float* wo_all = static_cast<const float*>(a.wo_packed);
  const float* w1_all = static_cast<const float*>(a.w1_packed);
  const float* w2_all = static_cast<const float*>(a.w2_packed);
  const float* wf_all = static_cast<const float*>(a.wfold_packed);
  f32x4 wf[8];  // weight fragments of the next GEMM, always one call ahead
  load_wfrags(wf, wqkv_all, 8, w, 0, lane);

  // ---- init embedding (K = 2 .. 6: plain VALU), padding rows zeroed ----------------------------------------------------
  {
    float* lsh = ys;  // [2 N] coordinates, then up to four [N] feature rows
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
    // thread = four consecutive channels (tid & 31) x one of sixteen token groups
    const int d0 = 4 * (tid & 31);
    const int ws = six ? 6 : ((four || pdp) ? 4 : (cvrp ? 3 : 2));  // row stride of w_init
    float wq[4][6], bq[4], dq[4][2], dbq[4], eq[4][2], ebq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int f = 0; f < 6; ++f) wq[k][f] = f < ws ? a.w_init[ws * (d0 + k) + f] : 0.0f;
      bq[k] = a.b_init[d0 + k];
      dq[k][0] = depot ? a.w_depot[2 * (d0 + k)] : 0.0f;
      dq[k][1] = depot ? a.w_depot[2 * (d0 + k) + 1] : 0.0f;
      dbq[k] = depot ? a.b_depot[d0 + k] : 0.0f;
      eq[k][0] = pdp ? a.w_extra[2 * (d0 + k)] : 0.0f;
      eq[k][1] = pdp ? a.w_extra[2 * (d0 + k) + 1] : 0.0f;
      ebq[k] = pdp ? a.b_extra[d0 + k] : 0.0f;
    }
    stage_biases(0);
    __syncthreads();
    for (int tok = tid >> 5; tok < kRows; tok += kThreads / 32) {
      f32x4 v = zero4();
      if (tok < N) {
        const float x = lsh[2 * tok], y = lsh[2 * tok + 1];
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
        for (int k = 0; k < 4; ++k) {
          float r;  // the Linear as a k-ordered fma chain over its 2 .. 6 inputs, bias last (GEMM then + bias)
          if (is_depot) r = fmaf(dq[k][1], y, dq[k][0] * x) + dbq[k];
          else if (is_delivery) r = fmaf(eq[k][1], y, eq[k][0] * x) + ebq[k];
          else {
            r = fmaf(wq[k][1], y, wq[k][0] * x);
            if (ws > 2) r = fmaf(wq[k][2], f2, r);
            if (ws > 3) r = fmaf(wq[k][3], f3, r);
            if (ws > 4) r = fmaf(wq[k][5], f5, fmaf(wq[k][4], f4, r));
            r += bq[k];
          }
          v[k] = r;
        }
      }
      *reinterpret_cast<f32x4*>(xs + tok * kRS + d0) = v;
    }
  }
  __syncthreads();

  for (int layer = 0; layer < a.num_layers; ++layer) {
    // The lane indices pass through an opaque copy once per layer, so every per-lane LDS / weight address below is derived
    // INSIDE the iteration, next to its use: hoisted out of the loop as invariants they were spilled (am_encoder.hip, r03)
    asm volatile("" : "+v"(tid), "+v"(w), "+v"(lane));
    c = lane & 15;
    g = lane >> 4;
    const float* Lqkv = wqkv_all + (int64_t)layer * 3 * kD * kD;
    const float* Lwo = wo_all + (int64_t)layer * kD * kD;
    const float* Lw1 = w1_all + (int64_t)layer * kFF * kD;
    const float* Lw2 = w2_all + (int64_t)layer * kD * kFF;

    // ---- Q^T, K^T (transposed form) and V (plain form) of head w: kept as registers, which ARE the operand fragments --
    f32x4 kf[TT], vf[TT];
    {
      f32x4 qf[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) qf[tt] = kf[tt] = vf[tt] = zero4();
      gemm16<TT, true>(qf, wf, xs, lane, Lqkv, 8, 8 + w, 0);  // (1 / sqrt(16) rides in the packed Wq and bq: exact)
      const f32x4 bq4 = *reinterpret_cast<const f32x4*>(bl + 16 * w + 4 * g);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) qf[tt] += bq4;
      // Q^T parked in this wave's own 16 columns of ys: each lane re-reads only what it stored (its token row, its four
      // dims) and later overwrites it with the attention output of that same row
      store_t<TT>(ys, qf, 16 * w, lane);
      gemm16<TT, true>(kf, wf, xs, lane, Lqkv, 8, 16 + w, 0);
      const f32x4 bk4 = *reinterpret_cast<const f32x4*>(bl + kD + 16 * w + 4 * g);
      gemm16<TT, false>(vf, wf, xs, lane, Lwo, 8, w, 0);      // out-proj weights: in flight across the attention
      const float bv = bl[2 * kD + 16 * w + c];               // plain form: the bias belongs to the lane's dim column
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        kf[tt] += bk4;
        vf[tt] += bv;
      }
    }

    // ---- attention of head w over all queries ------------------------------------------------------------------------
#pragma unroll
    for (int qt = 0; qt < TT; ++qt) {
      __builtin_amdgcn_sched_barrier(0);  // one query tile at a time: interleaved, two tiles' score registers would be live
      float* qrow = ys + (16 * qt + c) * kRS + 16 * w + 4 * g;
      const f32x4 q = *reinterpret_cast<const f32x4*>(qrow);
      f32x4 s[TT];
#pragma unroll
      for (int kt = 0; kt < TT; ++kt) s[kt] = zero4();
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int kt = 0; kt < TT; ++kt) s[kt] = mfma4(kf[kt][st], q[st], s[kt]);  // S^T[key][query]
      float m = -__builtin_huge_valf();
#pragma unroll
      for (int kt = 0; kt < TT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt == TT - 1) s[kt][r] = (16 * kt + 4 * g + r < N) ? s[kt][r] : -__builtin_huge_valf();  // padding keys
          m = fmaxf(m, s[kt][r]);
        }
      m = fmaxf(m, rl4co::bfly_f<16>(m));
      m = fmaxf(m, rl4co::bfly_f<32>(m));
      float l = 0.0f;
#pragma unroll
      for (int kt = 0; kt < TT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f((s[kt][r] - m) * kLog2e);
          s[kt][r] = p;
          l += p;
        }
      l += rl4co::bfly_f<16>(l);
      l += rl4co::bfly_f<32>(l);
      f32x4 o0 = zero4(), o1 = zero4();  // two accumulators: half the dependent-MFMA chain
#pragma unroll
      for (int kt = 0; kt < TT; ++kt)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          if (kt & 1) o1 = mfma4(vf[kt][st], s[kt][st], o1);  // O^T[dim][query]
          else o0 = mfma4(vf[kt][st], s[kt][st], o0);
        }
      const float inv = 1.0f / l;
      *reinterpret_cast<f32x4*>(qrow) = (o0 + o1) * inv;
    }
    __syncthreads();

    // ---- out-proj + residual + norm1 ---------------------------------------------------------------------------------
    {
      f32x4 y[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) y[tt] = zero4();
      gemm16<TT, true>(y, wf, ys, lane, Lw1, 8, w, 0);  // next: FFN1 chunk 0
      residual_norm<TT>(xs, y, 16 * w, bl + 3 * kD + kFF, a.n1_scale + layer * kD, a.n1_shift + layer * kD, a.norm, N, lane);
    }
    __syncthreads();

    // ---- FFN: hidden in 4 chunks of 128 (one 16-dim tile per wave), FFN2 accumulates across chunks ---------------------
    {
      f32x4 y2[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) y2[tt] = zero4();
#pragma unroll 1
      for (int ch = 0; ch < 4; ++ch) {
        f32x4 h1[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) h1[tt] = zero4();
        gemm16<TT, true>(h1, wf, xs, lane, Lw2, 32, w, 8 * ch);  // next: FFN2 of this chunk
        const f32x4 b14 = *reinterpret_cast<const f32x4*>(bl + 3 * kD + 128 * ch + 16 * w + 4 * g);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          h1[tt] += b14;
#pragma unroll
          for (int r = 0; r < 4; ++r) h1[tt][r] = fmaxf(h1[tt][r], 0.0f);
        }
        if (ch > 0) __syncthreads();  // every wave is done reading the previous chunk
        store_t<TT>(ys, h1, 16 * w, lane);
        __syncthreads();
        const bool last_layer = layer + 1 == a.num_layers;
        const float* nxt = ch < 3 ? Lw1 : (last_layer ? wf_all : wqkv_all + (int64_t)(layer + 1) * 3 * kD * kD);
        gemm16<TT, true>(y2, wf, ys, lane, nxt, 8, ch < 3 ? 8 * (ch + 1) + w : w, 0);
      }
      residual_norm<TT>(xs, y2, 16 * w, bl + 4 * kD + kFF, a.n2_scale + layer * kD, a.n2_shift + layer * kD, a.norm, N, lane);
    }
    __syncthreads();  // (also: every wave is done with this layer's biases and with ys)
    if (layer + 1 < a.num_layers) {
      stage_biases(layer + 1);
      __syncthreads();
    }
  }

  // ---- optional: final node embeddings h ---------------------------------------------------------------------------------
  if (a.hidden) {
    float* hout = a.hidden + (int64_t)b * N * kD;
    for (int i = tid; i < N * 32; i += kThreads)
      *reinterpret_cast<f32x4*>(hout + (int64_t)(i >> 5) * kD + 4 * (i & 31)) = *reinterpret_cast<const f32x4*>(xs + (i >> 5) * kRS + 4 * (i & 31));
  }

  // ---- fold: cache planes (and context tables) out of the accumulators, through LDS as contiguous rows ------------------
  const int nblocks = 3 + (a.ctx_first ? 1 : 0) + (a.ctx_cur ? 1 : 0);
#pragma unroll 1
  for (int blk = 0; blk < nblocks; ++blk) {
    asm volatile("" : "+v"(tid), "+v"(w), "+v"(lane));
    f32x4 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[tt] = zero4();
    gemm16<TT, true>(acc, wf, xs, lane, blk + 1 < nblocks ? wf_all + (int64_t)(blk + 1) * kD * kD : static_cast<const float*>(nullptr), 8, w, 0);
    store_t<TT>(ys, acc, 16 * w, lane);
    __syncthreads();
    if (blk < 3 && a.cache_dtype != RL4CO_DT_F32) {  // 16-bit planes from the fp32 encoder: rounded once, on the way out
      uint16_t* out = static_cast<uint16_t*>(a.kvl) + (int64_t)blk * a.kvl_plane_stride + (int64_t)b * a.kvl_batch_stride;
      const bool half = a.cache_dtype == RL4CO_DT_F16;
      for (int i = tid; i < N * 32; i += kThreads) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ys + (i >> 5) * kRS + 4 * (i & 31));
        uint16_t h[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (half) {
            const _Float16 e = (_Float16)v[k];
            h[k] = __builtin_bit_cast(uint16_t, e);
          } else {
            const __bf16 e = (__bf16)v[k];
            h[k] = __builtin_bit_cast(uint16_t, e);
          }
        }
        *reinterpret_cast<uint2*>(out + (int64_t)(i >> 5) * kD + 4 * (i & 31)) =
            make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
      }
    } else {
      float* out;
      if (blk < 3) out = static_cast<float*>(a.kvl) + (int64_t)blk * a.kvl_plane_stride + (int64_t)b * a.kvl_batch_stride;
      else if (blk == 3 && a.ctx_first) out = a.ctx_first + (int64_t)b * N * kD;
      else out = a.ctx_cur + (int64_t)b * N * kD;
      for (int i = tid; i < N * 32; i += kThreads)
        *reinterpret_cast<f32x4*>(out + (int64_t)(i >> 5) * kD + 4 * (i & 31)) = *reinterpret_cast<const f32x4*>(ys + (i >> 5) * kRS + 4 * (i & 31));
    }
    __syncthreads();
  }

  // ---- graph context: project_fixed_context(mean_j h_j)  (decoder.py:216-219) ---------------------------------------------
  if (a.q_bias) {
    if (tid < kD) {
      float s = 0.0f;
      for (int tok = 0; tok < N; ++tok) s += xs[tok * kRS + tid];
      meanv[tid] = s / (float)N;
    }
    __syncthreads();
    // wave w: output rows 16 w .. 16 w + 15; a row of W is ONE coalesced 512-byte load and a butterfly sum
    const float2 mv = *reinterpret_cast<const float2*>(meanv + 2 * lane);
    float2 wv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wv[r] = *reinterpret_cast<const float2*>(a.w_fixed + (int64_t)(16 * w + r) * kD + 2 * lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float acc = rl4co::bfly_sum<1, 64>(fmaf(wv[r].y, mv.y, wv[r].x * mv.x));
      if (lane == r) a.q_bias[(int64_t)b * kD + 16 * w + r] = acc;
    }
  }
}

template <int TT>
int launch_f32(const rl4co_am_encoder_args& a, hipStream_t stream) {
  const int lds = (2 * 16 * TT * kRS + kD + kBiasFloats) * 4;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_encoder_f32_kernel<TT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((am_encoder_f32_kernel<TT>), dim3(a.B), dim3(kThreads), lds, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

}  // namespace

extern "C" int rl4co_am_encoder_f32(const rl4co_am_encoder_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_encoder_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_PDP);
  RL4CO_REQUIRE(a.B > 0 && a.N >= 2 && a.N <= 128);
  RL4CO_REQUIRE(a.num_layers >= 1 && (a.norm == 0 || a.norm == 1));
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_F32);
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == RL4CO_DT_BF16 || a.cache_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(a.locs && a.w_init && a.b_init);
  RL4CO_REQUIRE(a.env != RL4CO_ENV_CVRP || (a.demand && a.w_depot && a.b_depot));
  RL4CO_REQUIRE(a.env != RL4CO_ENV_PDP || (a.w_depot && a.b_depot && a.w_extra && a.b_extra && (a.N - 1) % 2 == 0));
  RL4CO_REQUIRE(a.wqkv_packed && a.wo_packed && a.w1_packed && a.w2_packed && a.wfold_packed);
  RL4CO_REQUIRE(a.bqkv && a.bo && a.b1 && a.b2 && a.n1_scale && a.n1_shift && a.n2_scale && a.n2_shift);
  RL4CO_REQUIRE(a.kvl != nullptr);
  RL4CO_REQUIRE(a.ctx_first == nullptr || a.ctx_cur != nullptr);  // block order: planes, (ctx_first), (ctx_cur)
  RL4CO_REQUIRE(a.q_bias == nullptr || a.w_fixed != nullptr);
  RL4CO_REQUIRE(a.kvl_batch_stride >= (int64_t)a.N * kD && a.kvl_plane_stride >= a.kvl_batch_stride);
  hipStream_t s = rl4co::as_stream(stream);
  switch ((a.N + 15) / 16) {
    case 1: return launch_f32<1>(a, s);
    case 2: return launch_f32<2>(a, s);
    case 3: return launch_f32<3>(a, s);
    case 4: return launch_f32<4>(a, s);
    case 5: return launch_f32<5>(a, s);
    case 6: return launch_f32<6>(a, s);
    case 7: return launch_f32<7>(a, s);
    default: return launch_f32<8>(a, s);
  }
}
