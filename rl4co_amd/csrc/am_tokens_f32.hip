// am_tokens_f32.hip — the exact-fp32 encoder + cache fold for graphs of ANY size (BASELINE configs[4]: CVRP-500), on token tiles.
//
// am_encoder_f32.hip keeps an instance's residual stream in LDS (N <= 128). Beyond that the same layer algebra runs as a
// short sequence of launches over tiles of 128 consecutive nodes of one instance (grid = tiles x instances), fp32 end to
// end on v_mfma_f32_16x16x4_f32 with the same GEMM routine (enc_f32.h: gemm16 — identical summation order per output),
// so the bit-identical configuration never reaches a library GEMM / SDPA at any graph size:
//
//   tok_init_embed   features -> x0                                   env_embeddings/init.py:55-68,115-153,254-360
//   per layer
//     tok_qkv        x -> Q (x 1/4, exact), K token-major; V dim-major per head [B,8,16,NP]   nn/attention.py:110-118
//     tok_attn       softmax(Q K^T) V per head, keys / values streamed from L2 in tiles of 16 with an online softmax
//                    (8 waves = 8 heads, 128 queries per workgroup; no N x N matrix)          nn/attention.py:119-134
//     tok_mlp        Norm(x + out_proj(att)) -> Norm(. + MLP(.)), the 512-wide hidden through LDS in four chunks
//                                                                      nn/graph/attnnet.py:16-55, nn/ops.py:9-54, nn/mlp.py:52-61
//   tok_fold         cache planes / context tables from the final embeddings                  zoo/am/decoder.py:201-228 (cache.py)
//   graph_context    project_fixed_context(mean_j h_j)                                        zoo/am/decoder.py:216-219
// Normalisation: batch norm in eval mode (alpha, beta from the running statistics) inside tok_mlp; instance / layer norm
// couple all nodes of an instance: the layer's two halves then stop before their norm (tok_attn_half / tok_ffn_half write
// the pre-norm sums and per-tile statistics) and tok_norm_apply normalises with the combined statistics.
#include <hip/hip_runtime.h>

#include "common.h"
#include "enc_f32.h"

namespace {

using namespace rl4co_f32;

constexpr int kTT = 8;          // token tiles of 16 per workgroup: 128 nodes
constexpr int kTile = 16 * kTT;
constexpr int kLdsTile = kTile * kRS * 4;

struct Workspace {  // fp32 buffers carved out of the caller's workspace
  float *x0, *x1, *q, *k, *vt, *att, *ypre, *stats;
};
__host__ inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
__host__ inline int64_t np_of(int n) { return round_up(n, 16); }
__host__ inline int64_t workspace_floats(int B, int N) {
  const int64_t mx = round_up((int64_t)B * N * kD, 64);
  return 6 * mx + round_up((int64_t)B * 8 * 16 * np_of(N), 64) + round_up((int64_t)B * ((N + 127) / 128) * 256, 64);
}
__host__ inline Workspace carve(float* base, int B, int N) {
  const int64_t mx = round_up((int64_t)B * N * kD, 64);
  Workspace w;
  w.x0 = base;
  w.x1 = base + mx;
  w.q = base + 2 * mx;
  w.k = base + 3 * mx;
  w.att = base + 4 * mx;
  w.vt = base + 5 * mx;
  w.ypre = w.vt + round_up((int64_t)B * 8 * 16 * np_of(N), 64);  // pre-norm sums + tile statistics: instance / layer norm only
  w.stats = w.ypre + mx;
  return w;
}

// rows n0 .. n0 + 127 of instance b (128 fp32 each) -> LDS tile, rows past N zeroed
__device__ inline void load_tile(float* xs, const float* src, int b, int n0, int N, int tid) {
  const float* base = src + ((int64_t)b * N + n0) * kD;
  const int valid = min(kTile, N - n0);
  for (int i = tid; i < kTile * 32; i += kThreads) {
    const int row = i >> 5, c4 = i & 31;
    f32x4 v = zero4();
    if (row < valid) v = *reinterpret_cast<const f32x4*>(base + (int64_t)row * kD + 4 * c4);
    *reinterpret_cast<f32x4*>(xs + row * kRS + 4 * c4) = v;
  }
}
__device__ inline void store_tile(const float* xs, float* dst, int b, int n0, int N, int tid) {
  float* base = dst + ((int64_t)b * N + n0) * kD;
  const int valid = min(kTile, N - n0);
  for (int i = tid; i < valid * 32; i += kThreads)
    *reinterpret_cast<f32x4*>(base + (int64_t)(i >> 5) * kD + 4 * (i & 31)) = *reinterpret_cast<const f32x4*>(xs + (i >> 5) * kRS + 4 * (i & 31));
}

__global__ void __launch_bounds__(kThreads) tok_init_embed_kernel(const rl4co_am_encoder_args a, float* x0) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  float* lsh = xs + kTile * kRS;  // [6 N] staged features
  const int tid = threadIdx.x, b = blockIdx.y, n0 = kTile * blockIdx.x;
  init_embed_rows(a, b, n0, kTile, xs, lsh, tid);
  __syncthreads();
  store_tile(xs, x0, b, n0, a.N, tid);
}

// Q, K token-major [B N, 128]; V transposed per head: vt[b][head][dim 16][NP] (keys contiguous: the A operand of P . V)
__global__ void __launch_bounds__(kThreads) tok_qkv_kernel(const float* __restrict__ x, int N, int NP, const float* __restrict__ wqkv,
                                                           const float* __restrict__ bqkv, float* __restrict__ q,
                                                           float* __restrict__ k, float* __restrict__ vt) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, n0 = kTile * blockIdx.x;
  f32x4 wf[8];
  load_wfrags(wf, wqkv, 8, w, 0, lane);
  load_tile(xs, x, b, n0, N, tid);
  __syncthreads();
  const int valid = min(kTile, N - n0);
  f32x4 acc[kTT];
#pragma unroll 1
  for (int part = 0; part < 2; ++part) {  // Q, then K: transposed form, four consecutive dims of a token per lane
#pragma unroll
    for (int tt = 0; tt < kTT; ++tt) acc[tt] = zero4();
    gemm16<kTT, true>(acc, wf, xs, lane, wqkv, 8, 8 * (part + 1) + w, 0);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bqkv + kD * part + 16 * w + 4 * g);
    float* dst = (part == 0 ? q : k) + ((int64_t)b * N + n0) * kD + 16 * w + 4 * g;
#pragma unroll
    for (int tt = 0; tt < kTT; ++tt)
      if (16 * tt + c < valid) *reinterpret_cast<f32x4*>(dst + (int64_t)(16 * tt + c) * kD) = acc[tt] + bias;
  }
#pragma unroll
  for (int tt = 0; tt < kTT; ++tt) acc[tt] = zero4();
  gemm16<kTT, true>(acc, wf, xs, lane, static_cast<const float*>(nullptr), 0, 0, 0);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bqkv + 2 * kD + 16 * w + 4 * g);
  float* vrow = vt + (((int64_t)b * 8 + w) * 16 + 4 * g) * NP + n0;
#pragma unroll
  for (int tt = 0; tt < kTT; ++tt) {
    const int tok = 16 * tt + c;
    if (n0 + tok < NP) {  // the padding keys N .. NP - 1 are written as zeros: they enter P . V with weight 0
#pragma unroll
      for (int r = 0; r < 4; ++r) vrow[(int64_t)r * NP + tok] = tok < valid ? acc[tt][r] + bv[r] : 0.0f;
    }
  }
}

// softmax(Q K^T) V for head w (wave w) over 128 queries; keys / values of the instance stream from L2 in tiles of 16.
// S^T[key][query] = K . Q^T (A = K rows, B = Q rows: both plain 16-byte loads), online softmax per query column with ONE
// running maximum per column (the four row groups exchange theirs: the rescale factor must be uniform over a column's
// accumulator), O^T[dim][query] += V^T . P^T (A = vt rows, B = the exp'd score registers).
__global__ void __launch_bounds__(kThreads) tok_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ vt, int N, int NP, float* __restrict__ att) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, q0 = kTile * blockIdx.x;
  f32x4 qf[kTT], o[kTT];
  float m[kTT], l[kTT];
#pragma unroll
  for (int qt = 0; qt < kTT; ++qt) {
    const int tok = min(q0 + 16 * qt + c, N - 1);
    qf[qt] = *reinterpret_cast<const f32x4*>(q + ((int64_t)b * N + tok) * kD + 16 * w + 4 * g);
    o[qt] = zero4();
    m[qt] = -__builtin_huge_valf();
    l[qt] = 0.0f;
  }
  const float* kbase = k + (int64_t)b * N * kD + 16 * w + 4 * g;
  const float* vbase = vt + (((int64_t)b * 8 + w) * 16 + c) * NP + 4 * g;
  const int nkt = NP / 16;
  f32x4 kn = *reinterpret_cast<const f32x4*>(kbase + (int64_t)min(c, N - 1) * kD);
  f32x4 vn = *reinterpret_cast<const f32x4*>(vbase);
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt) {
    const f32x4 kc = kn, vc = vn;
    if (kt + 1 < nkt) {
      kn = *reinterpret_cast<const f32x4*>(kbase + (int64_t)min(16 * (kt + 1) + c, N - 1) * kD);
      vn = *reinterpret_cast<const f32x4*>(vbase + 16 * (kt + 1));
    }
    f32x4 s[kTT];
#pragma unroll
    for (int qt = 0; qt < kTT; ++qt) s[qt] = zero4();
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int qt = 0; qt < kTT; ++qt) s[qt] = mfma4(kc[st], qf[qt][st], s[qt]);
    const bool last = kt + 1 == nkt;
#pragma unroll
    for (int qt = 0; qt < kTT; ++qt) {
      if (last) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[qt][r] = (16 * kt + 4 * g + r < N) ? s[qt][r] : -__builtin_huge_valf();  // padding keys
      }
      float mt = fmaxf(fmaxf(s[qt][0], s[qt][1]), fmaxf(s[qt][2], s[qt][3]));
      mt = fmaxf(mt, rl4co::bfly_f<16>(mt));
      mt = fmaxf(mt, rl4co::bfly_f<32>(mt));
      const float mn = fmaxf(m[qt], mt);
      const float alpha = __builtin_amdgcn_exp2f((m[qt] - mn) * kLog2e);
      m[qt] = mn;
      float ps = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f((s[qt][r] - mn) * kLog2e);
        s[qt][r] = p;
        ps += p;
      }
      l[qt] = l[qt] * alpha + ps;
      o[qt] = o[qt] * alpha;
    }
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int qt = 0; qt < kTT; ++qt) o[qt] = mfma4(vc[st], s[qt][st], o[qt]);
  }
#pragma unroll
  for (int qt = 0; qt < kTT; ++qt) {
    float lt = l[qt];
    lt += rl4co::bfly_f<16>(lt);
    lt += rl4co::bfly_f<32>(lt);
    const float inv = 1.0f / lt;
    const int tok = q0 + 16 * qt + c;
    if (tok < N) *reinterpret_cast<f32x4*>(att + ((int64_t)b * N + tok) * kD + 16 * w + 4 * g) = o[qt] * inv;
  }
}

// Norm(x + out_proj(att)) -> Norm(. + MLP(.)) on one token tile: the second half of am_encoder_f32_kernel's layer body
__global__ void __launch_bounds__(kThreads) tok_mlp_kernel(const float* __restrict__ x, const float* __restrict__ att, int N,
                                                           const float* __restrict__ wo, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, const float* __restrict__ bo,
                                                           const float* __restrict__ b1, const float* __restrict__ b2,
                                                           const float* __restrict__ n1a, const float* __restrict__ n1b,
                                                           const float* __restrict__ n2a, const float* __restrict__ n2b,
                                                           float* __restrict__ xout) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  float* ys = xs + kTile * kRS;
  float* bl = ys + kTile * kRS;  // bo [128] | b1 [512] | b2 [128]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = (tid & 63) >> 4;
  const int b = blockIdx.y, n0 = kTile * blockIdx.x;
  f32x4 wf[8];
  load_wfrags(wf, wo, 8, w, 0, lane);
  load_tile(xs, x, b, n0, N, tid);
  load_tile(ys, att, b, n0, N, tid);
  for (int i = tid; i < 2 * kD + kFF; i += kThreads) bl[i] = i < kD ? bo[i] : (i < kD + kFF ? b1[i - kD] : b2[i - kD - kFF]);
  __syncthreads();
  {
    f32x4 y[kTT];
#pragma unroll
    for (int tt = 0; tt < kTT; ++tt) y[tt] = zero4();
    gemm16<kTT, true>(y, wf, ys, lane, w1, 8, w, 0);
    residual_norm<kTT>(xs, y, 16 * w, bl, n1a, n1b, 0, N, lane);
  }
  __syncthreads();
  {
    f32x4 y2[kTT];
#pragma unroll
    for (int tt = 0; tt < kTT; ++tt) y2[tt] = zero4();
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
      f32x4 h1[kTT];
#pragma unroll
      for (int tt = 0; tt < kTT; ++tt) h1[tt] = zero4();
      gemm16<kTT, true>(h1, wf, xs, lane, w2, 32, w, 8 * ch);
      const f32x4 b14 = *reinterpret_cast<const f32x4*>(bl + kD + 128 * ch + 16 * w + 4 * g);
#pragma unroll
      for (int tt = 0; tt < kTT; ++tt) {
        h1[tt] += b14;
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[tt][r] = fmaxf(h1[tt][r], 0.0f);
      }
      __syncthreads();  // every wave is done reading ys (the attention output / the previous chunk)
      store_t<kTT>(ys, h1, 16 * w, lane);
      __syncthreads();
      gemm16<kTT, true>(y2, wf, ys, lane, ch < 3 ? w1 : static_cast<const float*>(nullptr), 8, 8 * (ch + 1) + w, 0);
    }
    residual_norm<kTT>(xs, y2, 16 * w, bl + kD + kFF, n2a, n2b, 0, N, lane);
  }
  __syncthreads();
  store_tile(xs, xout, b, n0, N, tid);
}

// ---- token tiles under instance / layer norm (norm = 1 / 2): nn/ops.py:46-51 -------------------------------------------
// The statistics span all tiles of an instance, so each half of a layer ends BEFORE its norm: it writes the pre-norm sums
// and, per tile and channel, the mean and the centred sum of squares of the tile's valid tokens; tok_norm_apply_kernel
// combines the tiles' pairs (Chan's update: deterministic, no atomics, no cancellation) and normalises the rows.
constexpr int kStatFloats = 2 * kD;  // per (instance, tile): mean [128] | M2 [128]

// x + (y + bias) for the wave's 16-dim tile back into xs; tile statistics -> st
template <int TT>
__device__ inline void residual_stats(float* xs, f32x4 (&y)[TT], int dim0, const float* bias_lds, int valid, int lane, float* st) {
  const int c = lane & 15, g = lane >> 4;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_lds + dim0 + 4 * g);
  float* row = xs + c * kRS + dim0 + 4 * g;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + 16 * tt * kRS);
    y[tt] = x + (y[tt] + bias);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) s += (16 * tt + c < valid) ? y[tt][r] : 0.0f;
    s = rl4co::bfly_sum<1, 16>(s);
    const float mean = s / (float)valid;
    float v = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const float d = y[tt][r] - mean;
      v += (16 * tt + c < valid) ? d * d : 0.0f;
    }
    v = rl4co::bfly_sum<1, 16>(v);
    if (c == 0) {
      st[dim0 + 4 * g + r] = mean;
      st[kD + dim0 + 4 * g + r] = v;
    }
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) *reinterpret_cast<f32x4*>(row + 16 * tt * kRS) = y[tt];
}

// pre-norm x + out_proj(att) of one token tile
__global__ void __launch_bounds__(kThreads) tok_attn_half_kernel(const float* __restrict__ x, const float* __restrict__ att, int N,
                                                                 const float* __restrict__ wo, const float* __restrict__ bo,
                                                                 float* __restrict__ ypre, float* __restrict__ stats) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  float* ys = xs + kTile * kRS;
  float* bl = ys + kTile * kRS;  // bo [128]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y, n0 = kTile * blockIdx.x, valid = min(kTile, N - n0);
  f32x4 wf[8];
  load_wfrags(wf, wo, 8, w, 0, lane);
  load_tile(xs, x, b, n0, N, tid);
  load_tile(ys, att, b, n0, N, tid);
  for (int i = tid; i < kD; i += kThreads) bl[i] = bo[i];
  __syncthreads();
  f32x4 y[kTT];
#pragma unroll
  for (int tt = 0; tt < kTT; ++tt) y[tt] = zero4();
  gemm16<kTT, true>(y, wf, ys, lane, static_cast<const float*>(nullptr), 0, 0, 0);
  residual_stats<kTT>(xs, y, 16 * w, bl, valid, lane, stats + ((int64_t)b * gridDim.x + blockIdx.x) * kStatFloats);
  __syncthreads();
  store_tile(xs, ypre, b, n0, N, tid);
}

// pre-norm x + MLP(x) of one token tile (x: the normalised output of the attention half)
__global__ void __launch_bounds__(kThreads) tok_ffn_half_kernel(const float* __restrict__ x, int N, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, const float* __restrict__ b1,
                                                                const float* __restrict__ b2, float* __restrict__ ypre,
                                                                float* __restrict__ stats) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  float* ys = xs + kTile * kRS;
  float* bl = ys + kTile * kRS;  // b1 [512] | b2 [128]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = (tid & 63) >> 4;
  const int b = blockIdx.y, n0 = kTile * blockIdx.x, valid = min(kTile, N - n0);
  f32x4 wf[8];
  load_wfrags(wf, w1, 8, w, 0, lane);
  load_tile(xs, x, b, n0, N, tid);
  for (int i = tid; i < kFF + kD; i += kThreads) bl[i] = i < kFF ? b1[i] : b2[i - kFF];
  __syncthreads();
  f32x4 y2[kTT];
#pragma unroll
  for (int tt = 0; tt < kTT; ++tt) y2[tt] = zero4();
#pragma unroll 1
  for (int ch = 0; ch < 4; ++ch) {
    f32x4 h1[kTT];
#pragma unroll
    for (int tt = 0; tt < kTT; ++tt) h1[tt] = zero4();
    gemm16<kTT, true>(h1, wf, xs, lane, w2, 32, w, 8 * ch);
    const f32x4 b14 = *reinterpret_cast<const f32x4*>(bl + 128 * ch + 16 * w + 4 * g);
#pragma unroll
    for (int tt = 0; tt < kTT; ++tt) {
      h1[tt] += b14;
#pragma unroll
      for (int r = 0; r < 4; ++r) h1[tt][r] = fmaxf(h1[tt][r], 0.0f);
    }
    if (ch > 0) __syncthreads();  // every wave is done reading the previous chunk
    store_t<kTT>(ys, h1, 16 * w, lane);
    __syncthreads();
    gemm16<kTT, true>(y2, wf, ys, lane, ch < 3 ? w1 : static_cast<const float*>(nullptr), 8, 8 * (ch + 1) + w, 0);
  }
  residual_stats<kTT>(xs, y2, 16 * w, bl + kFF, valid, lane, stats + ((int64_t)b * gridDim.x + blockIdx.x) * kStatFloats);
  __syncthreads();
  store_tile(xs, ypre, b, n0, N, tid);
}

__device__ inline float block_sum8(float v, float* red, int tid) {  // sum over the 512 threads (red: 8 floats of LDS)
  v = rl4co::bfly_sum<1, 64>(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// rows of one token tile normalised with the statistics of the WHOLE instance (all its tiles' pairs combined)
__global__ void __launch_bounds__(kThreads) tok_norm_apply_kernel(const float* __restrict__ ypre, const float* __restrict__ stats, int N,
                                                                  int kind, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ xout) {
  __shared__ __align__(16) float ab[2 * kD];
  __shared__ float red[8];
  const int tid = threadIdx.x, b = blockIdx.y, tiles = gridDim.x, n0 = kTile * blockIdx.x, valid = min(kTile, N - n0);
  const float* st = stats + (int64_t)b * tiles * kStatFloats;
  float mean = 0.0f, m2 = 0.0f;
  if (tid < kD) {
    float tot = 0.0f;
    for (int t = 0; t < tiles; ++t) tot += (float)min(kTile, N - kTile * t) * st[t * kStatFloats + tid];
    mean = tot / (float)N;
    for (int t = 0; t < tiles; ++t) {
      const float d = st[t * kStatFloats + tid] - mean;
      m2 += st[t * kStatFloats + kD + tid] + (float)min(kTile, N - kTile * t) * (d * d);
    }
  }
  if (kind == 1) {
    if (tid < kD) {
      const float alpha = (1.0f / sqrtf(m2 / (float)N + 1e-5f)) * gamma[tid];
      ab[tid] = alpha;
      ab[kD + tid] = beta[tid] - mean * alpha;
    }
  } else {  // layer: every channel holds N values
    const float mean_all = block_sum8(tid < kD ? mean : 0.0f, red, tid) / (float)kD;
    const float d = mean - mean_all;
    const float m2_all = block_sum8(tid < kD ? m2 + (float)N * (d * d) : 0.0f, red, tid);
    const float invstd = 1.0f / sqrtf(m2_all / ((float)N * (float)kD - 1.0f) + 1e-5f);
    if (tid < kD) {
      ab[tid] = invstd;
      ab[kD + tid] = -mean_all * invstd;
    }
  }
  __syncthreads();
  const float* src = ypre + ((int64_t)b * N + n0) * kD;
  float* dst = xout + ((int64_t)b * N + n0) * kD;
  for (int i = tid; i < valid * 32; i += kThreads) {
    const int c0 = 4 * (i & 31);
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)(i >> 5) * kD + c0);
    const f32x4 al = *reinterpret_cast<const f32x4*>(ab + c0), be = *reinterpret_cast<const f32x4*>(ab + kD + c0);
    *reinterpret_cast<f32x4*>(dst + (int64_t)(i >> 5) * kD + c0) = v * al + be;
  }
}

struct FoldOut {
  void* ptr[5];
  int plane16[5];
  int64_t batch_stride[5];  // elements between instances
  int nblocks;
};

// 16-bit rows (the embeddings of the 16-bit token path) widened on the way into the fp32 LDS tile
__device__ inline void load_tile16(float* xs, const uint16_t* src, int half, int b, int n0, int N, int tid) {
  const uint16_t* base = src + ((int64_t)b * N + n0) * kD;
  const int valid = min(kTile, N - n0);
  for (int i = tid; i < kTile * 32; i += kThreads) {
    const int row = i >> 5, c4 = i & 31;
    f32x4 v = zero4();
    if (row < valid) {
      const uint2 r = *reinterpret_cast<const uint2*>(base + (int64_t)row * kD + 4 * c4);
      const uint16_t h[4] = {(uint16_t)(r.x & 0xffff), (uint16_t)(r.x >> 16), (uint16_t)(r.y & 0xffff), (uint16_t)(r.y >> 16)};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = half ? (float)__builtin_bit_cast(_Float16, h[e]) : __builtin_bit_cast(float, (uint32_t)h[e] << 16);
    }
    *reinterpret_cast<f32x4*>(xs + row * kRS + 4 * c4) = v;
  }
}

// x_dtype: RL4CO_DT_F32, or the 16-bit type of the rows (widened on load)
__global__ void __launch_bounds__(kThreads) tok_fold_kernel(const void* __restrict__ x, int x_dtype, int N, const float* __restrict__ wfold, const FoldOut fo) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  float* ys = xs + kTile * kRS;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y, n0 = kTile * blockIdx.x;
  f32x4 wf[8];
  load_wfrags(wf, wfold, 8, w, 0, lane);
  if (x_dtype == RL4CO_DT_F32) load_tile(xs, static_cast<const float*>(x), b, n0, N, tid);
  else load_tile16(xs, static_cast<const uint16_t*>(x), x_dtype == RL4CO_DT_F16, b, n0, N, tid);
  __syncthreads();
  const int valid = min(kTile, N - n0);
#pragma unroll 1
  for (int blk = 0; blk < fo.nblocks; ++blk) {
    f32x4 acc[kTT];
#pragma unroll
    for (int tt = 0; tt < kTT; ++tt) acc[tt] = zero4();
    gemm16<kTT, true>(acc, wf, xs, lane, blk + 1 < fo.nblocks ? wfold + (int64_t)(blk + 1) * kD * kD : static_cast<const float*>(nullptr), 8, w, 0);
    const int64_t off = (int64_t)b * fo.batch_stride[blk] + (int64_t)n0 * kD;
    void* out = fo.plane16[blk] ? static_cast<void*>(static_cast<uint16_t*>(fo.ptr[blk]) + off)
                                : static_cast<void*>(static_cast<float*>(fo.ptr[blk]) + off);
    fold_block_out<kTT>(acc, ys, w, lane, tid, valid, out, fo.plane16[blk]);
  }
}

// q_bias[b] = W_fixed . mean_j h[b, j]: 512 threads = 128 channels x 4 row classes, then wave w -> output rows 16 w ..
__global__ void __launch_bounds__(kThreads) graph_context_kernel(const void* __restrict__ h, int h_dtype, int N, const float* __restrict__ w_fixed,
                                                                 float* __restrict__ q_bias) {
  __shared__ float part[4][kD];
  __shared__ float meanv[kD];
  const int tid = threadIdx.x, b = blockIdx.x, d = tid & 127, cls = tid >> 7;
  // four independent partial sums per thread (rows cls, cls + 4, ... dealt round-robin): a single accumulator made the 125
  // row loads of a 501-node graph one dependent chain (87 us per launch at C5)
  float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (h_dtype == RL4CO_DT_F32) {
    const float* hb = static_cast<const float*>(h) + (int64_t)b * N * kD;
    for (int j0 = cls; j0 < N; j0 += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + 4 * u;
        s4[u] += j < N ? hb[(int64_t)j * kD + d] : 0.0f;
      }
    }
  } else {
    const uint16_t* hb = static_cast<const uint16_t*>(h) + (int64_t)b * N * kD;
    for (int j0 = cls; j0 < N; j0 += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + 4 * u;
        const uint16_t e = j < N ? hb[(int64_t)j * kD + d] : (uint16_t)0;
        s4[u] += h_dtype == RL4CO_DT_F16 ? (float)__builtin_bit_cast(_Float16, e) : __builtin_bit_cast(float, (uint32_t)e << 16);
      }
    }
  }
  part[cls][d] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  __syncthreads();
  if (tid < kD) meanv[tid] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) / (float)N;
  __syncthreads();
  const int w = tid >> 6, lane = tid & 63;
  const float2 mv = *reinterpret_cast<const float2*>(meanv + 2 * lane);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float2 wv = *reinterpret_cast<const float2*>(w_fixed + (int64_t)(16 * w + r) * kD + 2 * lane);
    const float acc = rl4co::bfly_sum<1, 64>(fmaf(wv.y, mv.y, wv.x * mv.x));
    if (lane == r) q_bias[(int64_t)b * kD + 16 * w + r] = acc;
  }
}

template <typename F>
int set_lds(F kernel, int bytes) {
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return RL4CO_OK;
}

}  // namespace

extern "C" int64_t rl4co_am_encoder_tokens_f32_workspace(int B, int N) {
  return (B > 0 && N > 0) ? workspace_floats(B, N) * 4 : 0;
}

extern "C" int rl4co_am_encoder_tokens_f32(const rl4co_am_encoder_args* args, void* workspace, int64_t workspace_bytes, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_am_encoder_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_PDP);
  RL4CO_REQUIRE(a.B > 0 && a.N >= 2 && a.B <= 65535);
  RL4CO_REQUIRE(a.num_layers >= 1 && a.norm >= 0 && a.norm <= 2);  // instance / layer norm: split sub-blocks + apply kernel
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_F32);
  RL4CO_REQUIRE(a.cache_dtype == RL4CO_DT_F32 || a.cache_dtype == RL4CO_DT_BF16 || a.cache_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(a.locs && a.w_init && a.b_init);
  RL4CO_REQUIRE(a.env != RL4CO_ENV_CVRP || (a.demand && a.w_depot && a.b_depot));
  RL4CO_REQUIRE(a.env != RL4CO_ENV_PDP || (a.w_depot && a.b_depot && a.w_extra && a.b_extra && (a.N - 1) % 2 == 0));
  RL4CO_REQUIRE(a.wqkv_packed && a.wo_packed && a.w1_packed && a.w2_packed && a.wfold_packed);
  RL4CO_REQUIRE(a.bqkv && a.bo && a.b1 && a.b2 && a.n1_scale && a.n1_shift && a.n2_scale && a.n2_shift);
  RL4CO_REQUIRE(a.kvl != nullptr);
  RL4CO_REQUIRE(a.ctx_first == nullptr || a.ctx_cur != nullptr);
  RL4CO_REQUIRE(a.ctx_dtype == RL4CO_DT_F32);
  RL4CO_REQUIRE(a.q_bias == nullptr || a.w_fixed != nullptr);
  RL4CO_REQUIRE(a.kvl_batch_stride >= (int64_t)a.N * kD && a.kvl_plane_stride >= a.kvl_batch_stride);
  RL4CO_REQUIRE(workspace != nullptr && workspace_bytes >= workspace_floats(a.B, a.N) * 4);
  RL4CO_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  hipStream_t s = rl4co::as_stream(stream);
  const int N = a.N, NP = (int)np_of(N);
  const Workspace ws = carve(static_cast<float*>(workspace), a.B, N);
  const dim3 grid((N + kTile - 1) / kTile, a.B), block(kThreads);
  const int lds_init = kLdsTile + 6 * N * 4 + 64;
  RL4CO_REQUIRE(lds_init <= 160 * 1024);
  const int lds_mlp = 2 * kLdsTile + (2 * kD + kFF) * 4;
  if (int e = set_lds(tok_init_embed_kernel, lds_init)) return e;
  if (int e = set_lds(tok_qkv_kernel, kLdsTile)) return e;
  if (int e = set_lds(tok_mlp_kernel, lds_mlp)) return e;
  if (int e = set_lds(tok_fold_kernel, 2 * kLdsTile)) return e;
  if (a.norm != 0) {
    if (int e = set_lds(tok_attn_half_kernel, 2 * kLdsTile + kD * 4)) return e;
    if (int e = set_lds(tok_ffn_half_kernel, 2 * kLdsTile + (kFF + kD) * 4)) return e;
  }

  hipLaunchKernelGGL(tok_init_embed_kernel, grid, block, lds_init, s, a, ws.x0);
  float *xin = ws.x0, *xout = ws.x1;
  const float* wqkv = static_cast<const float*>(a.wqkv_packed);
  const float* wo = static_cast<const float*>(a.wo_packed);
  const float* w1 = static_cast<const float*>(a.w1_packed);
  const float* w2 = static_cast<const float*>(a.w2_packed);
  for (int layer = 0; layer < a.num_layers; ++layer) {
    hipLaunchKernelGGL(tok_qkv_kernel, grid, block, kLdsTile, s, xin, N, NP, wqkv + (int64_t)layer * 3 * kD * kD, a.bqkv + layer * 3 * kD,
                       ws.q, ws.k, ws.vt);
    hipLaunchKernelGGL(tok_attn_kernel, grid, block, 0, s, ws.q, ws.k, ws.vt, N, NP, ws.att);
    if (a.norm == 0) {
      hipLaunchKernelGGL(tok_mlp_kernel, grid, block, lds_mlp, s, xin, ws.att, N, wo + (int64_t)layer * kD * kD, w1 + (int64_t)layer * kFF * kD,
                         w2 + (int64_t)layer * kD * kFF, a.bo + layer * kD, a.b1 + layer * kFF, a.b2 + layer * kD, a.n1_scale + layer * kD,
                         a.n1_shift + layer * kD, a.n2_scale + layer * kD, a.n2_shift + layer * kD, xout);
    } else {  // instance / layer norm: each half stops before its norm; the apply kernel sees the whole instance's statistics
      hipLaunchKernelGGL(tok_attn_half_kernel, grid, block, 2 * kLdsTile + kD * 4, s, xin, ws.att, N, wo + (int64_t)layer * kD * kD,
                         a.bo + layer * kD, ws.ypre, ws.stats);
      hipLaunchKernelGGL(tok_norm_apply_kernel, grid, block, 0, s, ws.ypre, ws.stats, N, a.norm, a.n1_scale + layer * kD,
                         a.n1_shift + layer * kD, xout);
      hipLaunchKernelGGL(tok_ffn_half_kernel, grid, block, 2 * kLdsTile + (kFF + kD) * 4, s, xout, N, w1 + (int64_t)layer * kFF * kD,
                         w2 + (int64_t)layer * kD * kFF, a.b1 + layer * kFF, a.b2 + layer * kD, ws.ypre, ws.stats);
      hipLaunchKernelGGL(tok_norm_apply_kernel, grid, block, 0, s, ws.ypre, ws.stats, N, a.norm, a.n2_scale + layer * kD,
                         a.n2_shift + layer * kD, xout);
    }
    float* t = xin;
    xin = xout;
    xout = t;
  }
  // xin: the final node embeddings
  FoldOut fo;
  fo.nblocks = 3 + (a.ctx_first ? 1 : 0) + (a.ctx_cur ? 1 : 0);
  for (int blk = 0; blk < 5; ++blk) {
    fo.ptr[blk] = nullptr;
    fo.plane16[blk] = 0;
    fo.batch_stride[blk] = (int64_t)N * kD;
  }
  const int p16 = a.cache_dtype == RL4CO_DT_F32 ? 0 : a.cache_dtype;
  for (int blk = 0; blk < 3; ++blk) {
    fo.ptr[blk] = p16 ? static_cast<void*>(static_cast<uint16_t*>(a.kvl) + (int64_t)blk * a.kvl_plane_stride)
                      : static_cast<void*>(static_cast<float*>(a.kvl) + (int64_t)blk * a.kvl_plane_stride);
    fo.plane16[blk] = p16;
    fo.batch_stride[blk] = a.kvl_batch_stride;
  }
  int nb = 3;
  if (a.ctx_first) fo.ptr[nb++] = a.ctx_first;
  if (a.ctx_cur) fo.ptr[nb++] = a.ctx_cur;
  hipLaunchKernelGGL(tok_fold_kernel, grid, block, 2 * kLdsTile, s, static_cast<const void*>(xin), (int)RL4CO_DT_F32, N,
                     static_cast<const float*>(a.wfold_packed), fo);
  if (a.q_bias) hipLaunchKernelGGL(graph_context_kernel, dim3(a.B), block, 0, s, static_cast<const void*>(xin), (int)RL4CO_DT_F32, N, a.w_fixed, a.q_bias);
  if (a.hidden)
    RL4CO_HIP_TRY(hipMemcpyAsync(a.hidden, xin, (size_t)a.B * N * kD * 4, hipMemcpyDeviceToDevice, s));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// fp32 tables from the final node embeddings of ANY encoder: out_i[B,N,128] = h . W_i^T for up to five packed [128,128]
// blocks (pack_weight_f32), fp32 accumulate on the fp32 MFMA, and q_bias = W_fixed . mean_j h_j. Replaces the library
// GEMMs of the cache fold's fp32 side (context tables, graph context: zoo/am/decoder.py:201-228, cache.py) on the 16-bit
// token path (h in bf16 / fp16, widened on load). h_dtype: dtype of h's rows; out[i] fp32; any of q_bias / w_fixed NULL: skipped.
extern "C" int rl4co_am_fold_tables_f32(const void* h, int h_dtype, int B, int N, const float* w_packed, int nblocks, float* const* out,
                                        const float* w_fixed, float* q_bias, void* stream) {
  RL4CO_REQUIRE(h != nullptr && B > 0 && N >= 1 && B <= 65535);
  RL4CO_REQUIRE(h_dtype == RL4CO_DT_F32 || h_dtype == RL4CO_DT_BF16 || h_dtype == RL4CO_DT_F16);
  RL4CO_REQUIRE(nblocks >= 0 && nblocks <= 5 && (nblocks == 0 || (w_packed != nullptr && out != nullptr)));
  RL4CO_REQUIRE((q_bias == nullptr) == (w_fixed == nullptr) || q_bias == nullptr);
  hipStream_t s = rl4co::as_stream(stream);
  const dim3 grid((N + kTile - 1) / kTile, B), block(kThreads);
  if (nblocks > 0) {
    FoldOut fo;
    fo.nblocks = nblocks;
    for (int blk = 0; blk < 5; ++blk) {
      fo.ptr[blk] = blk < nblocks ? out[blk] : nullptr;
      fo.plane16[blk] = 0;
      fo.batch_stride[blk] = (int64_t)N * kD;
      RL4CO_REQUIRE(blk >= nblocks || out[blk] != nullptr);
    }
    if (int e = set_lds(tok_fold_kernel, 2 * kLdsTile)) return e;
    hipLaunchKernelGGL(tok_fold_kernel, grid, block, 2 * kLdsTile, s, h, h_dtype, N, w_packed, fo);
  }
  if (q_bias && w_fixed) hipLaunchKernelGGL(graph_context_kernel, dim3(B), block, 0, s, h, h_dtype, N, w_fixed, q_bias);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// The init embeddings alone, fp32 (`init_embeds` of AttentionModelPolicy.forward(return_init_embeds=True) in the
// bit-identical configuration): the token path's first launch — the same init_embed_rows as the fused fp32 kernel.
extern "C" int rl4co_am_encoder_init_embeds_f32(const rl4co_am_encoder_args* args, float* out, void* stream) {
  RL4CO_REQUIRE(args != nullptr && out != nullptr);
  const rl4co_am_encoder_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_PDP);
  RL4CO_REQUIRE(a.B > 0 && a.N >= 2 && a.B <= 65535);
  RL4CO_REQUIRE(a.act_dtype == RL4CO_DT_F32);
  RL4CO_REQUIRE(a.locs && a.w_init && a.b_init);
  RL4CO_REQUIRE(a.env != RL4CO_ENV_CVRP || (a.demand && a.w_depot && a.b_depot));
  RL4CO_REQUIRE(a.env != RL4CO_ENV_PDP || (a.w_depot && a.b_depot && a.w_extra && a.b_extra && (a.N - 1) % 2 == 0));
  hipStream_t s = rl4co::as_stream(stream);
  const dim3 grid((a.N + kTile - 1) / kTile, a.B), block(kThreads);
  const int lds_init = kLdsTile + 6 * a.N * 4 + 64;
  RL4CO_REQUIRE(lds_init <= 160 * 1024);
  if (int e = set_lds(tok_init_embed_kernel, lds_init)) return e;
  hipLaunchKernelGGL(tok_init_embed_kernel, grid, block, lds_init, s, a, out);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
