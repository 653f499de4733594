// enc_f32.h — device helpers shared by the exact-fp32 encoder kernels (am_encoder_f32.hip: one workgroup per instance,
// N <= 128; am_tokens_f32.hip: token tiles of any graph size) on v_mfma_f32_16x16x4_f32.
//
// Operand layouts of the 16x16x4 fp32 MFMA (lane = 16 g + c): A[row c][k-slot g], B[k-slot g][col c] — one fp32 register
// each — and C/D[row 4 g + r][col c] in four registers r. A lane's 16-byte fragment (four consecutive k at 4 g of a
// 16-k chunk) feeds four MFMA steps, component s of both operands in step s: the k-slot permutation is the same on both
// sides and cancels.
#ifndef RL4CO_ENC_F32_H
#define RL4CO_ENC_F32_H

#include <hip/hip_runtime.h>

#include "common.h"

namespace rl4co_f32 {

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

// ---- RL4CO_F32_SPLIT (r06 experiment): fp32 products as six bf16 products ------------------------------------------------
// An fp32 value splits EXACTLY into three bf16 pieces by truncation (v = p1 + p2 + p3 + r, |r| <= 2^-24 |v|: each piece is
// the top 8 significant bits of what the previous ones left, the subtractions are exact). a b = sum of the nine piece
// products; the six of weight >= 2^-16 reproduce it to ~2^-23 |a b| — the size of an fp32 product's own rounding — and every
// piece product is exact in the MFMA's fp32 accumulator. A lane's 16-byte fragment (four consecutive k at 4 g of a 16-k
// chunk) IS the operand layout of v_mfma_f32_16x16x16_bf16, so six of those (8 passes each) replace the chunk's four
// v_mfma_f32_16x16x4_f32 (32 passes each): 48 instead of 128 matrix cycles per chunk and token tile, for 18 VALU operations
// per fragment. NOT the k-ordered fmaf chain of the exact kernel any more: the summation order inside an MFMA is the
// hardware's — accepted only where the trained-weight parity holds.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
struct Split3 {
  bf16x4_t p1, p2, p3;
};
__device__ inline Split3 split3(const f32x4& v) {
  uint32_t u[4], r1[4], r2[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    u[s] = __float_as_uint(v[s]);
    const float a = v[s] - __uint_as_float(u[s] & 0xffff0000u);
    r1[s] = __float_as_uint(a);
    r2[s] = __float_as_uint(a - __uint_as_float(r1[s] & 0xffff0000u));
  }
  // the upper halves of two words side by side: bytes (x0.b2, x0.b3, x1.b2, x1.b3)
  auto hi2 = [](uint32_t x1, uint32_t x0) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); };
  Split3 o;
  o.p1 = __builtin_bit_cast(bf16x4_t, make_uint2(hi2(u[1], u[0]), hi2(u[3], u[2])));
  o.p2 = __builtin_bit_cast(bf16x4_t, make_uint2(hi2(r1[1], r1[0]), hi2(r1[3], r1[2])));
  o.p3 = __builtin_bit_cast(bf16x4_t, make_uint2(hi2(r2[1], r2[0]), hi2(r2[3], r2[2])));
  return o;
}
__device__ inline f32x4 mfma16b(const bf16x4_t& a, const bf16x4_t& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// c += a . b over the chunk's 16 k, smallest terms first
__device__ inline f32x4 mfma_split(const Split3& a, const Split3& b, f32x4 c) {
  c = mfma16b(a.p3, b.p1, c);
  c = mfma16b(a.p2, b.p2, c);
  c = mfma16b(a.p1, b.p3, c);
  c = mfma16b(a.p2, b.p1, c);
  c = mfma16b(a.p1, b.p2, c);
  return mfma16b(a.p1, b.p1, c);
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
#ifdef RL4CO_F32_SPLIT
    {
      const Split3 ws = split3(wf[j]);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const Split3 xs3 = split3(x[j & 1][tt]);
        acc[tt] = W_IS_A ? mfma_split(ws, xs3, acc[tt]) : mfma_split(xs3, ws, acc[tt]);
      }
    }
#else
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        acc[tt] = W_IS_A ? mfma4(wf[j][s], x[j & 1][tt][s], acc[tt]) : mfma4(x[j & 1][tt][s], wf[j][s], acc[tt]);
    }
#endif
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
// LAYER (normalization="layer", nn/ops.py:48-51): ONE mean and ONE unbiased variance over all N x 128 values of the
// instance, no affine — the waves' partial sums meet in `red` (2 x 8 floats of LDS; two workgroup barriers inside).
template <int TT, bool LAYER = false>
__device__ inline void residual_norm(float* xs, f32x4 (&y)[TT], int dim0, const float* bias_lds, const float* na, const float* nb,
                                     int norm, int N, int lane, float* red = nullptr, int w = 0) {
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
  if constexpr (LAYER) {
    float s = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const float t4 = (y[tt][0] + y[tt][1]) + (y[tt][2] + y[tt][3]);
      s += (16 * tt + c < N) ? t4 : 0.0f;
    }
    s = rl4co::bfly_sum<1, 64>(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    const float cnt = (float)(N * kD);
    const float mean = (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) / cnt;
    float v = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      float t4 = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = y[tt][r] - mean;
        t4 = fmaf(d, d, t4);
      }
      v += (16 * tt + c < N) ? t4 : 0.0f;
    }
    v = rl4co::bfly_sum<1, 64>(v);
    if (lane == 0) red[8 + w] = v;
    __syncthreads();
    const float var = (((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]))) / (cnt - 1.0f);
    const float invstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      alpha[r] = invstd;
      beta[r] = -mean * invstd;
    }
  } else if (norm == 1) {
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


// Init embedding of rows n0 .. n0 + rows_pad - 1 of instance b into xs (rows past N zeroed). The instance's coordinates and
// feature rows are staged in `lsh` first (6 N floats at most): read per token from global memory they are a chain of
// dependent L2 round trips. thread = four consecutive channels (tid & 31) x one of sixteen token groups. The Linear is a
// k-ordered fma chain over its 2 .. 6 inputs, bias last (GEMM then + bias). Contains one __syncthreads().
//   env_embeddings/init.py:55-68 (TSP), 115-136 (VRP), 139-153 (VRPTW), 254-280 (OP), 283-312 (PCTSP), 335-360 (PDP)
__device__ inline void init_embed_rows(const rl4co_am_encoder_args& a, int b, int n0, int rows_pad, float* xs, float* lsh, int tid) {
  const int N = a.N;
  const float* loc = a.locs + (int64_t)b * N * 2;
  const bool pdp = a.env == RL4CO_ENV_PDP;  // depot | pickups (x, y, x', y' of the delivery) | deliveries
  const bool cvrp = a.env == RL4CO_ENV_CVRP;
  const bool depot = cvrp || pdp;
  const int half = (N - 1) / 2;
  for (int i = tid; i < 2 * N; i += kThreads) lsh[i] = loc[i];
  const bool four = cvrp && a.feature4 != nullptr;  // PCTSP: (x, y, expected prize, penalty)
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
  __syncthreads();
  for (int row = tid >> 5; row < rows_pad; row += kThreads / 32) {
    const int tok = n0 + row;
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
        float r;
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
    *reinterpret_cast<f32x4*>(xs + row * kRS + d0) = v;
  }
}

// One fold block: (rows of xs) . W^T for the wave's 16 dims, staged in ys as [row][dim], then written as contiguous rows:
// `out_rows` rows of 128 fp32 (or, `plane16` != 0, of the 16-bit cache type: rounded once, on the way out).
template <int TT>
__device__ inline void fold_block_out(f32x4 (&acc)[TT], float* ys, int w, int lane, int tid, int out_rows, void* out, int plane16) {
  store_t<TT>(ys, acc, 16 * w, lane);
  __syncthreads();
  if (plane16) {
    uint16_t* o16 = static_cast<uint16_t*>(out);
    const bool half = plane16 == RL4CO_DT_F16;
    for (int i = tid; i < out_rows * 32; i += kThreads) {
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
      *reinterpret_cast<uint2*>(o16 + (int64_t)(i >> 5) * kD + 4 * (i & 31)) =
          make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    }
  } else {
    float* o32 = static_cast<float*>(out);
    for (int i = tid; i < out_rows * 32; i += kThreads)
      *reinterpret_cast<f32x4*>(o32 + (int64_t)(i >> 5) * kD + 4 * (i & 31)) = *reinterpret_cast<const f32x4*>(ys + (i >> 5) * kRS + 4 * (i & 31));
  }
  __syncthreads();
}

}  // namespace rl4co_f32

#endif
