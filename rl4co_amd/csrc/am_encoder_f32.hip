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
#include "enc_f32.h"

namespace {

using namespace rl4co_f32;

template <int TT, bool LAYER = false>
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
  stage_biases(0);
  init_embed_rows(a, b, 0, kRows, xs, ys, tid);  // (ys is free here: the features are staged in it)
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
      residual_norm<TT, LAYER>(xs, y, 16 * w, bl + 3 * kD + kFF, a.n1_scale + layer * kD, a.n1_shift + layer * kD, a.norm, N, lane, meanv, w);
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
      residual_norm<TT, LAYER>(xs, y2, 16 * w, bl + 4 * kD + kFF, a.n2_scale + layer * kD, a.n2_shift + layer * kD, a.norm, N, lane, meanv, w);
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
    void* out;
    int plane16 = 0;
    if (blk < 3) {
      plane16 = a.cache_dtype == RL4CO_DT_F32 ? 0 : a.cache_dtype;  // 16-bit planes from the fp32 encoder: rounded once, on the way out
      const int64_t off = (int64_t)blk * a.kvl_plane_stride + (int64_t)b * a.kvl_batch_stride;
      out = plane16 ? static_cast<void*>(static_cast<uint16_t*>(a.kvl) + off) : static_cast<void*>(static_cast<float*>(a.kvl) + off);
    } else {
      out = static_cast<float*>((blk == 3 && a.ctx_first) ? a.ctx_first : a.ctx_cur) + (int64_t)b * N * kD;
    }
    fold_block_out<TT>(acc, ys, w, lane, tid, N, out, plane16);
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
  if (a.norm == 2) {  // whole-instance statistics: its own instantiation, so the batch / instance kernel's code is untouched
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_encoder_f32_kernel<TT, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((am_encoder_f32_kernel<TT, true>), dim3(a.B), dim3(kThreads), lds, stream, a);
  } else {
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_encoder_f32_kernel<TT, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((am_encoder_f32_kernel<TT, false>), dim3(a.B), dim3(kThreads), lds, stream, a);
  }
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

}  // namespace

extern "C" int rl4co_am_encoder_f32(const rl4co_am_encoder_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_encoder_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_PDP);
  RL4CO_REQUIRE(a.B > 0 && a.N >= 2 && a.N <= 128);
  RL4CO_REQUIRE(a.num_layers >= 1 && a.norm >= 0 && a.norm <= 2);  // batch (eval) | instance | layer
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_F32);
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == RL4CO_DT_BF16 || a.cache_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(a.locs && a.w_init && a.b_init);
  RL4CO_REQUIRE(a.env != RL4CO_ENV_CVRP || (a.demand && a.w_depot && a.b_depot));
  RL4CO_REQUIRE(a.env != RL4CO_ENV_PDP || (a.w_depot && a.b_depot && a.w_extra && a.b_extra && (a.N - 1) % 2 == 0));
  RL4CO_REQUIRE(a.wqkv_packed && a.wo_packed && a.w1_packed && a.w2_packed && a.wfold_packed);
  RL4CO_REQUIRE(a.bqkv && a.bo && a.b1 && a.b2 && a.n1_scale && a.n1_shift && a.n2_scale && a.n2_shift);
  RL4CO_REQUIRE(a.kvl != nullptr);
  RL4CO_REQUIRE(a.ctx_first == nullptr || a.ctx_cur != nullptr);  // block order: planes, (ctx_first), (ctx_cur)
  RL4CO_REQUIRE(a.ctx_dtype == RL4CO_DT_F32);  // the exact kernels write fp32 context tables
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
