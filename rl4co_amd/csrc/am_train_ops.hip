// am_train_ops.hip — training-time encoder pieces (training path of C4).
//
// The reference encoder layer is SkipConnection(MHA) -> Normalization -> SkipConnection(MLP) ->
// Normalization (nn/graph/attnnet.py:16-54, nn/ops.py:9-54). In training the norm cannot be folded
// away and its autograd graph (mean, var, rsqrt, scale, shift and their backward) is a dozen
// elementwise / reduction launches per layer over [B,N,128] activations; for POMO
// (normalization="instance", zoo/pomo/model.py:59-63) the statistics are per instance and channel
// over the N nodes, so one workgroup per instance can do skip + norm in one pass over HBM, forward
// and backward:
//
//   forward   y = x + s ;  mu, var over nodes (biased, two-pass from registers) ;
//             out = (y - mu) * rsqrt(var + eps) * gamma + beta            -> out, y (bf16), mu, rstd
//   backward  xh = (y - mu) rstd ; dxh = dout * gamma ;
//             dy = rstd * (dxh - mean_n(dxh) - xh * mean_n(dxh * xh))     -> dy (bf16, both skip inputs)
//             dgamma += sum_n dout * xh ; dbeta += sum_n dout               (fp32 atomics, 128 + 128 per instance)
//
// bf16 activations in HBM (the autocast regime the encoder GEMMs run in), fp32 arithmetic.
#include <hip/hip_runtime.h>

#include "common.h"
#include "elem16.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kThreads = 256;
constexpr int kMaxRows = 32;  // nodes per thread: N <= 128

// (historic names: the two packed 16-bit elements of a word, in whichever element type this unit is compiled for —
// elem16.h)
__device__ inline float bf16_lo(uint32_t v) { return rl4co_e16::lo(v); }
__device__ inline float bf16_hi(uint32_t v) { return rl4co_e16::hi(v); }
__device__ inline uint32_t pack_bf16(float a, float b) { return rl4co_e16::pack(a, b); }

// thread = channel pair (tid & 63) x node class (tid >> 6: nodes q, q + 4, ...); sums over the four
// node classes meet in LDS
__device__ inline void sum4(float (&v)[2], float* red, int tid) {
  __syncthreads();
  red[tid * 2] = v[0];
  red[tid * 2 + 1] = v[1];
  __syncthreads();
  const int cp = tid & 63;
#pragma unroll
  for (int e = 0; e < 2; ++e)
    v[e] = (red[cp * 2 + e] + red[(64 + cp) * 2 + e]) + (red[(128 + cp) * 2 + e] + red[(192 + cp) * 2 + e]);
}

__global__ void __launch_bounds__(kThreads) skip_inorm_fwd_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ s,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float eps, int N, uint32_t* __restrict__ y,
                                                                  uint32_t* __restrict__ out, float* __restrict__ mean,
                                                                  float* __restrict__ rstd) {
  __shared__ float red[kThreads * 2];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * N * (kD / 2);
  float v[kMaxRows][2];
  float sum[2] = {0.0f, 0.0f};
  // every row's loads are issued before the first use (rows past N re-read row N - 1 and are ignored): guarded
  // per row the compiler emitted one load -> wait -> use round trip after another, 28 serial HBM latencies
  uint32_t ra[kMaxRows], rb[kMaxRows];
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = min(q + 4 * i, N - 1);
    ra[i] = x[base + (int64_t)n * (kD / 2) + cp];
    rb[i] = s[base + (int64_t)n * (kD / 2) + cp];
  }
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    v[i][0] = 0.0f;
    v[i][1] = 0.0f;
    if (n < N) {
      const uint32_t a = ra[i], b = rb[i];
      // the skip sum is rounded to bf16 like the reference's x + module(x) under autocast, and it is
      // the value the backward pass re-reads
      const uint32_t ys = pack_bf16(bf16_lo(a) + bf16_lo(b), bf16_hi(a) + bf16_hi(b));
      y[base + (int64_t)n * (kD / 2) + cp] = ys;
      v[i][0] = bf16_lo(ys);
      v[i][1] = bf16_hi(ys);
      sum[0] += v[i][0];
      sum[1] += v[i][1];
    }
  }
  sum4(sum, red, tid);
  const float inv_n = 1.0f / (float)N;
  const float mu[2] = {sum[0] * inv_n, sum[1] * inv_n};
  float sq[2] = {0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    if (q + 4 * i < N) {
      const float d0 = v[i][0] - mu[0], d1 = v[i][1] - mu[1];
      sq[0] = fmaf(d0, d0, sq[0]);
      sq[1] = fmaf(d1, d1, sq[1]);
    }
  }
  sum4(sq, red, tid);
  const float rs[2] = {rsqrtf(sq[0] * inv_n + eps), rsqrtf(sq[1] * inv_n + eps)};
  const float g0 = gamma[2 * cp], g1 = gamma[2 * cp + 1], b0 = beta[2 * cp], b1 = beta[2 * cp + 1];
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    if (n < N)
      out[base + (int64_t)n * (kD / 2) + cp] =
          pack_bf16(fmaf((v[i][0] - mu[0]) * rs[0], g0, b0), fmaf((v[i][1] - mu[1]) * rs[1], g1, b1));
  }
  if (q == 0) {
    mean[(int64_t)blockIdx.x * kD + 2 * cp] = mu[0];
    mean[(int64_t)blockIdx.x * kD + 2 * cp + 1] = mu[1];
    rstd[(int64_t)blockIdx.x * kD + 2 * cp] = rs[0];
    rstd[(int64_t)blockIdx.x * kD + 2 * cp + 1] = rs[1];
  }
}

__global__ void __launch_bounds__(kThreads) skip_inorm_bwd_kernel(const uint32_t* __restrict__ dout, const uint32_t* __restrict__ y,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, int N, uint32_t* __restrict__ dy,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[kThreads * 2];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * N * (kD / 2);
  const float mu[2] = {mean[(int64_t)blockIdx.x * kD + 2 * cp], mean[(int64_t)blockIdx.x * kD + 2 * cp + 1]};
  const float rs[2] = {rstd[(int64_t)blockIdx.x * kD + 2 * cp], rstd[(int64_t)blockIdx.x * kD + 2 * cp + 1]};
  const float g[2] = {gamma[2 * cp], gamma[2 * cp + 1]};
  float xh[kMaxRows][2], dd[kMaxRows][2];
  float s_d[2] = {0.0f, 0.0f}, s_dx[2] = {0.0f, 0.0f};
  uint32_t ra[kMaxRows], rb[kMaxRows];  // all loads first (see the forward kernel)
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = min(q + 4 * i, N - 1);
    ra[i] = dout[base + (int64_t)n * (kD / 2) + cp];
    rb[i] = y[base + (int64_t)n * (kD / 2) + cp];
  }
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    xh[i][0] = xh[i][1] = dd[i][0] = dd[i][1] = 0.0f;
    if (n < N) {
      const uint32_t a = ra[i], b = rb[i];
      dd[i][0] = bf16_lo(a);
      dd[i][1] = bf16_hi(a);
      xh[i][0] = (bf16_lo(b) - mu[0]) * rs[0];
      xh[i][1] = (bf16_hi(b) - mu[1]) * rs[1];
      s_d[0] += dd[i][0];
      s_d[1] += dd[i][1];
      s_dx[0] = fmaf(dd[i][0], xh[i][0], s_dx[0]);
      s_dx[1] = fmaf(dd[i][1], xh[i][1], s_dx[1]);
    }
  }
  sum4(s_d, red, tid);   // sum_n dout           (= d beta of this instance)
  sum4(s_dx, red, tid);  // sum_n dout * xh      (= d gamma of this instance)
  const float inv_n = 1.0f / (float)N;
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    if (n < N) {
      // dxh = dout * gamma; mean_n(dxh) = gamma * s_d / N; mean_n(dxh * xh) = gamma * s_dx / N
      const float r0 = rs[0] * g[0] * (dd[i][0] - s_d[0] * inv_n - xh[i][0] * s_dx[0] * inv_n);
      const float r1 = rs[1] * g[1] * (dd[i][1] - s_d[1] * inv_n - xh[i][1] * s_dx[1] * inv_n);
      dy[base + (int64_t)n * (kD / 2) + cp] = pack_bf16(r0, r1);
    }
  }
  if (q == 0) {
    // per-instance partials (summed over the instances by the caller in a fixed order): 4096 workgroups adding
    // into the same 8 cache lines serialised in L2 and set this kernel's duration
    *reinterpret_cast<float2*>(dgamma + (int64_t)blockIdx.x * kD + 2 * cp) = make_float2(s_dx[0], s_dx[1]);
    *reinterpret_cast<float2*>(dbeta + (int64_t)blockIdx.x * kD + 2 * cp) = make_float2(s_d[0], s_d[1]);
  }
}

// ---- the same two kernels for graphs beyond 4 * kMaxRows nodes (r06). Nothing waits in registers: the instance's rows
// (256 bytes per node) are read again for each pass — from L2, the workgroup wrote or read them a moment ago. Thread
// layout, accumulation order and arithmetic are the register-resident kernels', so the results are theirs bit for bit.
constexpr int kWideRows = 8;  // rows in flight per thread and pass

template <typename F>
__device__ inline void wide_rows(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, int64_t base, int cp, int q, int N,
                                 F&& body) {
  for (int n0 = q; n0 < N; n0 += 4 * kWideRows) {
    uint32_t ra[kWideRows], rb[kWideRows];
#pragma unroll
    for (int i = 0; i < kWideRows; ++i) {  // all loads first; rows past N re-read the last row and are ignored
      const int n = min(n0 + 4 * i, N - 1);
      ra[i] = a[base + (int64_t)n * (kD / 2) + cp];
      rb[i] = b != nullptr ? b[base + (int64_t)n * (kD / 2) + cp] : 0u;
    }
#pragma unroll
    for (int i = 0; i < kWideRows; ++i)
      if (n0 + 4 * i < N) body(n0 + 4 * i, ra[i], rb[i]);
  }
}

__global__ void __launch_bounds__(kThreads) skip_inorm_fwd_wide_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ s,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                       float eps, int N, uint32_t* y, uint32_t* __restrict__ out,
                                                                       float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ float red[kThreads * 2];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * N * (kD / 2);
  float sum[2] = {0.0f, 0.0f};
  wide_rows(x, s, base, cp, q, N, [&](int n, uint32_t a, uint32_t b) {
    const uint32_t ys = pack_bf16(bf16_lo(a) + bf16_lo(b), bf16_hi(a) + bf16_hi(b));
    y[base + (int64_t)n * (kD / 2) + cp] = ys;
    sum[0] += bf16_lo(ys);
    sum[1] += bf16_hi(ys);
  });
  sum4(sum, red, tid);
  const float inv_n = 1.0f / (float)N;
  const float mu[2] = {sum[0] * inv_n, sum[1] * inv_n};
  float sq[2] = {0.0f, 0.0f};
  // (a thread re-reads the words it wrote itself: same address, program order)
  wide_rows(y, nullptr, base, cp, q, N, [&](int, uint32_t ys, uint32_t) {
    const float d0 = bf16_lo(ys) - mu[0], d1 = bf16_hi(ys) - mu[1];
    sq[0] = fmaf(d0, d0, sq[0]);
    sq[1] = fmaf(d1, d1, sq[1]);
  });
  sum4(sq, red, tid);
  const float rs[2] = {rsqrtf(sq[0] * inv_n + eps), rsqrtf(sq[1] * inv_n + eps)};
  const float g0 = gamma[2 * cp], g1 = gamma[2 * cp + 1], b0 = beta[2 * cp], b1 = beta[2 * cp + 1];
  wide_rows(y, nullptr, base, cp, q, N, [&](int n, uint32_t ys, uint32_t) {
    out[base + (int64_t)n * (kD / 2) + cp] =
        pack_bf16(fmaf((bf16_lo(ys) - mu[0]) * rs[0], g0, b0), fmaf((bf16_hi(ys) - mu[1]) * rs[1], g1, b1));
  });
  if (q == 0) {
    mean[(int64_t)blockIdx.x * kD + 2 * cp] = mu[0];
    mean[(int64_t)blockIdx.x * kD + 2 * cp + 1] = mu[1];
    rstd[(int64_t)blockIdx.x * kD + 2 * cp] = rs[0];
    rstd[(int64_t)blockIdx.x * kD + 2 * cp + 1] = rs[1];
  }
}

__global__ void __launch_bounds__(kThreads) skip_inorm_bwd_wide_kernel(const uint32_t* __restrict__ dout, const uint32_t* __restrict__ y,
                                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd, int N, uint32_t* __restrict__ dy,
                                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[kThreads * 2];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * N * (kD / 2);
  const float mu[2] = {mean[(int64_t)blockIdx.x * kD + 2 * cp], mean[(int64_t)blockIdx.x * kD + 2 * cp + 1]};
  const float rs[2] = {rstd[(int64_t)blockIdx.x * kD + 2 * cp], rstd[(int64_t)blockIdx.x * kD + 2 * cp + 1]};
  const float g[2] = {gamma[2 * cp], gamma[2 * cp + 1]};
  float s_d[2] = {0.0f, 0.0f}, s_dx[2] = {0.0f, 0.0f};
  wide_rows(dout, y, base, cp, q, N, [&](int, uint32_t a, uint32_t b) {
    const float d0 = bf16_lo(a), d1 = bf16_hi(a);
    s_d[0] += d0;
    s_d[1] += d1;
    s_dx[0] = fmaf(d0, (bf16_lo(b) - mu[0]) * rs[0], s_dx[0]);
    s_dx[1] = fmaf(d1, (bf16_hi(b) - mu[1]) * rs[1], s_dx[1]);
  });
  sum4(s_d, red, tid);
  sum4(s_dx, red, tid);
  const float inv_n = 1.0f / (float)N;
  wide_rows(dout, y, base, cp, q, N, [&](int n, uint32_t a, uint32_t b) {
    const float x0 = (bf16_lo(b) - mu[0]) * rs[0], x1 = (bf16_hi(b) - mu[1]) * rs[1];
    const float r0 = rs[0] * g[0] * (bf16_lo(a) - s_d[0] * inv_n - x0 * s_dx[0] * inv_n);
    const float r1 = rs[1] * g[1] * (bf16_hi(a) - s_d[1] * inv_n - x1 * s_dx[1] * inv_n);
    dy[base + (int64_t)n * (kD / 2) + cp] = pack_bf16(r0, r1);
  });
  if (q == 0) {
    *reinterpret_cast<float2*>(dgamma + (int64_t)blockIdx.x * kD + 2 * cp) = make_float2(s_dx[0], s_dx[1]);
    *reinterpret_cast<float2*>(dbeta + (int64_t)blockIdx.x * kD + 2 * cp) = make_float2(s_d[0], s_d[1]);
  }
}

// ---- normalization="layer" (nn/ops.py:48-51): (y - mean) / sqrt(var + 1e-5) with ONE mean and ONE UNBIASED variance over
// all M = N x 128 values of the instance, no affine. Same thread layout as the instance-norm kernels above; the
// statistics are workgroup-wide sums.
//   backward  xh = (y - mu) r ;  dy = r * (dout - sum(dout) / M - xh * sum(dout * xh) / (M - 1))
__device__ inline float block_sum(float v, float* red4, int tid) {
  v = rl4co::bfly_sum<1, 64>(v);
  __syncthreads();
  if ((tid & 63) == 0) red4[tid >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

__global__ void __launch_bounds__(kThreads) skip_lnorm_fwd_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ s,
                                                                  float eps, int N, uint32_t* __restrict__ y,
                                                                  uint32_t* __restrict__ out, float* __restrict__ stats) {
  __shared__ float red4[4];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * N * (kD / 2);
  float v[kMaxRows][2];
  uint32_t ra[kMaxRows], rb[kMaxRows];  // all loads first (see skip_inorm_fwd_kernel)
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = min(q + 4 * i, N - 1);
    ra[i] = x[base + (int64_t)n * (kD / 2) + cp];
    rb[i] = s[base + (int64_t)n * (kD / 2) + cp];
  }
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    v[i][0] = 0.0f;
    v[i][1] = 0.0f;
    if (n < N) {
      const uint32_t a = ra[i], b = rb[i];
      const uint32_t ys = pack_bf16(bf16_lo(a) + bf16_lo(b), bf16_hi(a) + bf16_hi(b));  // the skip sum, rounded like autocast's
      y[base + (int64_t)n * (kD / 2) + cp] = ys;
      v[i][0] = bf16_lo(ys);
      v[i][1] = bf16_hi(ys);
      sum += v[i][0] + v[i][1];
    }
  }
  const float cnt = (float)(N * kD);
  const float mu = block_sum(sum, red4, tid) / cnt;
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    if (q + 4 * i < N) {
      const float d0 = v[i][0] - mu, d1 = v[i][1] - mu;
      sq = fmaf(d0, d0, sq);
      sq = fmaf(d1, d1, sq);
    }
  }
  const float rs = rsqrtf(block_sum(sq, red4, tid) / (cnt - 1.0f) + eps);
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    if (n < N) out[base + (int64_t)n * (kD / 2) + cp] = pack_bf16((v[i][0] - mu) * rs, (v[i][1] - mu) * rs);
  }
  if (tid == 0) {
    stats[2 * (int64_t)blockIdx.x] = mu;
    stats[2 * (int64_t)blockIdx.x + 1] = rs;
  }
}

__global__ void __launch_bounds__(kThreads) skip_lnorm_bwd_kernel(const uint32_t* __restrict__ dout, const uint32_t* __restrict__ y,
                                                                  const float* __restrict__ stats, int N, uint32_t* __restrict__ dy) {
  __shared__ float red4[4];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * N * (kD / 2);
  const float mu = stats[2 * (int64_t)blockIdx.x], rs = stats[2 * (int64_t)blockIdx.x + 1];
  float xh[kMaxRows][2], dd[kMaxRows][2];
  uint32_t ra[kMaxRows], rb[kMaxRows];
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = min(q + 4 * i, N - 1);
    ra[i] = dout[base + (int64_t)n * (kD / 2) + cp];
    rb[i] = y[base + (int64_t)n * (kD / 2) + cp];
  }
  float s_d = 0.0f, s_dx = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    xh[i][0] = xh[i][1] = dd[i][0] = dd[i][1] = 0.0f;
    if (q + 4 * i < N) {
      const uint32_t a = ra[i], b = rb[i];
      dd[i][0] = bf16_lo(a);
      dd[i][1] = bf16_hi(a);
      xh[i][0] = (bf16_lo(b) - mu) * rs;
      xh[i][1] = (bf16_hi(b) - mu) * rs;
      s_d += dd[i][0] + dd[i][1];
      s_dx = fmaf(dd[i][0], xh[i][0], s_dx);
      s_dx = fmaf(dd[i][1], xh[i][1], s_dx);
    }
  }
  const float cnt = (float)(N * kD);
  const float m_d = block_sum(s_d, red4, tid) / cnt;
  const float m_dx = block_sum(s_dx, red4, tid) / (cnt - 1.0f);
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    if (n < N)
      dy[base + (int64_t)n * (kD / 2) + cp] =
          pack_bf16(rs * (dd[i][0] - m_d - xh[i][0] * m_dx), rs * (dd[i][1] - m_d - xh[i][1] * m_dx));
  }
}

// the layer formula beyond 4 * kMaxRows nodes (r06): rows re-read per pass, as the instance-norm kernels above
__global__ void __launch_bounds__(kThreads) skip_lnorm_fwd_wide_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ s,
                                                                       float eps, int N, uint32_t* y, uint32_t* __restrict__ out,
                                                                       float* __restrict__ stats) {
  __shared__ float red4[4];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * N * (kD / 2);
  float sum = 0.0f;
  wide_rows(x, s, base, cp, q, N, [&](int n, uint32_t a, uint32_t b) {
    const uint32_t ys = pack_bf16(bf16_lo(a) + bf16_lo(b), bf16_hi(a) + bf16_hi(b));
    y[base + (int64_t)n * (kD / 2) + cp] = ys;
    sum += bf16_lo(ys) + bf16_hi(ys);
  });
  const float cnt = (float)(N * kD);
  const float mu = block_sum(sum, red4, tid) / cnt;
  float sq = 0.0f;
  wide_rows(y, nullptr, base, cp, q, N, [&](int, uint32_t ys, uint32_t) {
    const float d0 = bf16_lo(ys) - mu, d1 = bf16_hi(ys) - mu;
    sq = fmaf(d0, d0, sq);
    sq = fmaf(d1, d1, sq);
  });
  const float rs = rsqrtf(block_sum(sq, red4, tid) / (cnt - 1.0f) + eps);
  wide_rows(y, nullptr, base, cp, q, N, [&](int n, uint32_t ys, uint32_t) {
    out[base + (int64_t)n * (kD / 2) + cp] = pack_bf16((bf16_lo(ys) - mu) * rs, (bf16_hi(ys) - mu) * rs);
  });
  if (tid == 0) {
    stats[2 * (int64_t)blockIdx.x] = mu;
    stats[2 * (int64_t)blockIdx.x + 1] = rs;
  }
}

__global__ void __launch_bounds__(kThreads) skip_lnorm_bwd_wide_kernel(const uint32_t* __restrict__ dout, const uint32_t* __restrict__ y,
                                                                       const float* __restrict__ stats, int N, uint32_t* __restrict__ dy) {
  __shared__ float red4[4];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * N * (kD / 2);
  const float mu = stats[2 * (int64_t)blockIdx.x], rs = stats[2 * (int64_t)blockIdx.x + 1];
  float s_d = 0.0f, s_dx = 0.0f;
  wide_rows(dout, y, base, cp, q, N, [&](int, uint32_t a, uint32_t b) {
    const float d0 = bf16_lo(a), d1 = bf16_hi(a);
    s_d += d0 + d1;
    s_dx = fmaf(d0, (bf16_lo(b) - mu) * rs, s_dx);
    s_dx = fmaf(d1, (bf16_hi(b) - mu) * rs, s_dx);
  });
  const float cnt = (float)(N * kD);
  const float m_d = block_sum(s_d, red4, tid) / cnt;
  const float m_dx = block_sum(s_dx, red4, tid) / (cnt - 1.0f);
  wide_rows(dout, y, base, cp, q, N, [&](int n, uint32_t a, uint32_t b) {
    const float x0 = (bf16_lo(b) - mu) * rs, x1 = (bf16_hi(b) - mu) * rs;
    dy[base + (int64_t)n * (kD / 2) + cp] = pack_bf16(rs * (bf16_lo(a) - m_d - x0 * m_dx), rs * (bf16_hi(a) - m_d - x1 * m_dx));
  });
}

}  // namespace

#if !RL4CO_ELEM_F16
extern "C" int rl4co_skip_inorm_max_nodes(void) { return 4 * kMaxRows; }  // the register-resident kernels (instance and layer norm)
extern "C" int rl4co_skip_inorm_wide_max_nodes(void) { return 1024; }      // rl4co_skip_inorm_* / rl4co_skip_lnorm_*: rows re-read per pass beyond that
#endif

extern "C" int RL4CO_ENTRY(rl4co_skip_lnorm_fwd)(const void* x, const void* s, float eps, int B, int N, void* y, void* out,
                                                 float* stats, void* stream) {
  RL4CO_REQUIRE(x && s && y && out && stats);
  RL4CO_REQUIRE(B > 0 && N >= 1 && N <= 1024 && eps > 0.0f);
  if (N > 4 * kMaxRows)
    hipLaunchKernelGGL(skip_lnorm_fwd_wide_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream), static_cast<const uint32_t*>(x),
                       static_cast<const uint32_t*>(s), eps, N, static_cast<uint32_t*>(y), static_cast<uint32_t*>(out), stats);
  else
    hipLaunchKernelGGL(skip_lnorm_fwd_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream), static_cast<const uint32_t*>(x),
                       static_cast<const uint32_t*>(s), eps, N, static_cast<uint32_t*>(y), static_cast<uint32_t*>(out), stats);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_skip_lnorm_bwd)(const void* dout, const void* y, const float* stats, int B, int N, void* dy,
                                                 void* stream) {
  RL4CO_REQUIRE(dout && y && stats && dy);
  RL4CO_REQUIRE(B > 0 && N >= 1 && N <= 1024);
  if (N > 4 * kMaxRows)
    hipLaunchKernelGGL(skip_lnorm_bwd_wide_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream),
                       static_cast<const uint32_t*>(dout), static_cast<const uint32_t*>(y), stats, N, static_cast<uint32_t*>(dy));
  else
    hipLaunchKernelGGL(skip_lnorm_bwd_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream),
                       static_cast<const uint32_t*>(dout), static_cast<const uint32_t*>(y), stats, N, static_cast<uint32_t*>(dy));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_skip_inorm_fwd)(const void* x, const void* s, const float* gamma, const float* beta, float eps, int B,
                                         int N, void* y, void* out, float* mean, float* rstd, void* stream) {
  RL4CO_REQUIRE(x && s && gamma && beta && y && out && mean && rstd);
  RL4CO_REQUIRE(B > 0 && N >= 1 && N <= 1024 && eps > 0.0f);
  if (N > 4 * kMaxRows)
    hipLaunchKernelGGL(skip_inorm_fwd_wide_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream),
                       static_cast<const uint32_t*>(x), static_cast<const uint32_t*>(s), gamma, beta, eps, N,
                       static_cast<uint32_t*>(y), static_cast<uint32_t*>(out), mean, rstd);
  else
    hipLaunchKernelGGL(skip_inorm_fwd_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream),
                       static_cast<const uint32_t*>(x), static_cast<const uint32_t*>(s), gamma, beta, eps, N,
                       static_cast<uint32_t*>(y), static_cast<uint32_t*>(out), mean, rstd);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_skip_inorm_bwd)(const void* dout, const void* y, const float* gamma, const float* mean,
                                         const float* rstd, int B, int N, void* dy, float* dgamma, float* dbeta, void* stream) {
  RL4CO_REQUIRE(dout && y && gamma && mean && rstd && dy && dgamma && dbeta);
  RL4CO_REQUIRE(B > 0 && N >= 1 && N <= 1024);
  if (N > 4 * kMaxRows)
    hipLaunchKernelGGL(skip_inorm_bwd_wide_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream),
                       static_cast<const uint32_t*>(dout), static_cast<const uint32_t*>(y), gamma, mean, rstd, N,
                       static_cast<uint32_t*>(dy), dgamma, dbeta);
  else
    hipLaunchKernelGGL(skip_inorm_bwd_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream),
                       static_cast<const uint32_t*>(dout), static_cast<const uint32_t*>(y), gamma, mean, rstd, N,
                       static_cast<uint32_t*>(dy), dgamma, dbeta);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// ------------------------------------------------------------------------------------------------
// Token-parallel linear layers of the training encoder on the matrix cores.
//
//   out[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )        bf16 in / out, fp32 accumulate
//
// is nn.Linear (Wqkv, out_proj, the MLP: nn/attention.py:64-134, nn/mlp.py:52-61) over the
// M = B x nodes token rows, and — called with the transposed weight — its input gradient
// dX = dY . W. The shapes are tall and skinny (M ~ 4e5, K and N in {128, 384, 512}): the work is
// HBM-bound (read A once, write out once), which the library GEMM misses by 10x at these shapes
// (1.0 ms for [409600,128] x [128,512], 0.5 TB/s). One workgroup owns a stripe of 128 token rows:
// in every shape used either K = 128 (the A chunk is loaded once and reused for all column tiles)
// or N = 128 (a single column tile), so A is never re-read. Products are computed transposed —
// v_mfma_f32_32x32x16_bf16 with the FEATURE as the accumulator row — so a lane holds four
// consecutive features of one token and the epilogue packs 8-byte LDS writes; the tile leaves
// through LDS as coalesced 16-byte rows. Epilogue: + bias, then optionally ReLU, or the ReLU
// backward mask (mask[m,n] > 0 ? v : 0) when the call computes d hidden = d out . W2.
namespace {

constexpr int kTM = 128, kTN = 128, kTK = 128;  // tile: token rows x features x contraction chunk
constexpr int kLS = kTK + 8;                    // LDS row stride (bf16)
constexpr int kGemmThreads = 256;
constexpr int kMaxLinearN = 1024;  // output features (the bias sits in LDS)

typedef elem_t bf16x8 __attribute__((ext_vector_type(8)));
typedef elem_t bf16x4g __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(kGemmThreads, 2) linear_bf16_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                                      const float* __restrict__ bias, const uint16_t* __restrict__ mask,
                                                                      const uint16_t* __restrict__ residual, int M, int N, int K, int relu,
                                                                      uint16_t* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem_g[];
  elem_t* as = reinterpret_cast<elem_t*>(smem_g);  // [128 tokens][kLS]
  elem_t* ws = as + kTM * kLS;                       // [128 features][kLS]
  float* bl = reinterpret_cast<float*>(ws + kTN * kLS);  // [N] bias: read from LDS in the epilogue (sixteen dependent
                                                         // L2 round trips per feature tile when read from global there)
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * kTM;
  const int nkc = K / kTK, nnt = N / kTN, steps = nnt * nkc;
  // a step = (feature tile nt, contraction chunk kc). The operands of step i + 1 are fetched into
  // registers while the MFMAs of step i run; the token chunk is re-fetched only when it changes.
  const int srow = tid >> 4, scol = (tid & 15) * 8;  // this thread stages rows srow + 16 j, 16 bytes at scol
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // a plain vector: an array of HIP's uint4 struct goes to scratch
  u32x4 pa[8], pw[8];
#define RL4CO_FETCH(STEP, WITH_A)                                                                              \
  {                                                                                                            \
    const int nt_ = (STEP) / nkc, kc_ = (STEP) % nkc;                                                          \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                            \
      const int64_t row_ = min(m0 + srow + 16 * j, (int64_t)M - 1); /* rows past M: computed, never stored */   \
      if (WITH_A) pa[j] = *reinterpret_cast<const u32x4*>(A + row_ * K + kc_ * kTK + scol);                    \
      pw[j] = *reinterpret_cast<const u32x4*>(W + (int64_t)(nt_ * kTN + srow + 16 * j) * K + kc_ * kTK + scol); \
    }                                                                                                          \
  }
#define RL4CO_COMMIT(WITH_A)                                                                        \
  {                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
      if (WITH_A) *reinterpret_cast<u32x4*>(as + (srow + 16 * j) * kLS + scol) = pa[j];             \
      *reinterpret_cast<u32x4*>(ws + (srow + 16 * j) * kLS + scol) = pw[j];                         \
    }                                                                                               \
  }
  RL4CO_FETCH(0, true)
  if (bias)
    for (int i = tid; i < N; i += kGemmThreads) bl[i] = bias[i];
  RL4CO_COMMIT(true)
  __syncthreads();
  f32x16 acc[4];  // [feature tile ct][feature 32 ct + rowmap(r, hi)], token 32 w + l31
  // the ReLU-backward mask rows OR the residual rows of this wave (never both: validated by the entry point)
  const uint16_t* aux = mask ? mask : residual;
  const int orow = lane >> 4, ocol = (lane & 15) * 8;  // output pass: four rows per pass, 16 bytes per lane
  u32x4 mks[8];
  for (int step = 0; step < steps; ++step) {
    const int nt = step / nkc, kc = step % nkc;
    const bool more = step + 1 < steps, next_a = nkc > 1;  // K = 128: the token chunk never changes
    if (more) RL4CO_FETCH(step + 1, next_a)
    if (kc == 0) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
    }
    if (aux && kc == nkc - 1) {  // the epilogue's mask / residual rows of this wave: in flight during the products
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8) {
        const int64_t row = min(m0 + 32 * w + 4 * p8 + orow, (int64_t)M - 1);
        mks[p8] = *reinterpret_cast<const u32x4*>(aux + row * N + nt * kTN + ocol);
      }
    }
#pragma unroll
    for (int ks = 0; ks < kTK / 16; ++ks) {
      const bf16x8 tok = *reinterpret_cast<const bf16x8*>(as + (32 * w + l31) * kLS + 16 * ks + 8 * hi);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const bf16x8 feat = *reinterpret_cast<const bf16x8*>(ws + (32 * ct + l31) * kLS + 16 * ks + 8 * hi);
        acc[ct] = rl4co_e16::mfma_32x32x16(feat, tok, acc[ct]);
      }
    }
    {
      // The prefetched operands are pinned in their registers HERE, before this step's stores are issued: vmcnt is
      // one in-order counter for loads and stores, so waiting for them where they are committed to LDS (after the
      // stores) meant waiting for every store of the step to be acknowledged by L2 first — the step's whole write
      // latency, exposed once per feature tile (wide-N shapes ran at 3 TB/s, the K-deep ones at 4.4)
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // (unconditional: under `if (more)` the compiler's scoreboard forgets it at the join)
        asm volatile("" ::"v"(pw[j]));
        asm volatile("" ::"v"(pa[j]));
        asm volatile("" ::"v"(mks[j]));
      }
    }
    if (kc == nkc - 1) {
      // epilogue: + bias, ReLU in the accumulators; the tile then leaves through LDS so that the global stores
      // are 16 bytes per lane over contiguous 256-byte row segments (stored straight from the accumulators a lane
      // owns 8 bytes in each of 32 different rows: the write path, not HBM, bounded the wide-N shapes at 2.4 TB/s).
      // The weight tile `ws` is dead once every wave has issued its products; each wave stages and drains only
      // its own 32 token rows, so the staging itself needs no workgroup barrier.
      rl4co::lds_barrier();
      elem_t* os = ws;  // [128 tokens][kLS]
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int fl = 32 * ct + 8 * q + 4 * hi;  // feature inside the tile
          float v[4];
          if (bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(bl + nt * kTN + fl);
            v[0] = acc[ct][4 * q] + b4.x;
            v[1] = acc[ct][4 * q + 1] + b4.y;
            v[2] = acc[ct][4 * q + 2] + b4.z;
            v[3] = acc[ct][4 * q + 3] + b4.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[ct][4 * q + i];
          }
          if (relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
          }
          bf16x4g o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (elem_t)v[i];
          *reinterpret_cast<bf16x4g*>(os + (32 * w + l31) * kLS + fl) = o;
        }
      }
      rl4co::lds_barrier_wave();
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8) {
        const int tr = 32 * w + 4 * p8 + orow;
        const int64_t row = m0 + tr;
        u32x4 val = *reinterpret_cast<const u32x4*>(os + tr * kLS + ocol);
        if (row < M) {
          if (mask) {  // ReLU backward: keep where the forward activation was positive (bf16 > 0)
            const u32x4 mk = mks[p8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t m_ = mk[i];
              const uint32_t keep_lo = ((m_ & 0x8000u) == 0 && (m_ & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
              const uint32_t keep_hi = ((m_ >> 31) == 0 && (m_ & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
              val[i] &= (keep_lo | keep_hi);
            }
          } else if (residual) {  // + the skip connection's gradient: the sum autograd would form in one more pass
            const u32x4 rs = mks[p8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float lo = rl4co_e16::lo(val[i]) + rl4co_e16::lo(rs[i]);
              const float hi = rl4co_e16::hi(val[i]) + rl4co_e16::hi(rs[i]);
              val[i] = rl4co_e16::pack(lo, hi);
            }
          }
          *reinterpret_cast<u32x4*>(out + row * N + nt * kTN + ocol) = val;
        }
      }
    }
    if (more) {
      rl4co::lds_barrier();  // every wave is done with this step's LDS operands (and with the staged output tile)
      RL4CO_COMMIT(next_a)
      rl4co::lds_barrier();
    }
  }
#undef RL4CO_FETCH
#undef RL4CO_COMMIT
}

// K = 128 without mask / residual rows (Wqkv, MLP up, the fold GEMM, out_proj both ways): the token tile's B-operand
// fragments live in REGISTERS for all feature tiles (32 per lane) instead of a second LDS tile — no token reads in the
// product loop (a fifth of its LDS traffic), and with one 35 KB tile per workgroup and <= 168 registers THREE workgroups
// share a CU. The step structure is the generic kernel's. (A variant carrying the mask / residual rows, 32 more
// registers and two workgroups per CU, measured slower than the generic kernel on d hidden 128 -> 512: 205 vs 187 us.)
__global__ void __launch_bounds__(kGemmThreads, 3) linear_k128_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                                      const float* __restrict__ bias, int M, int N, int relu,
                                                                      uint16_t* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem_g[];
  elem_t* ws = reinterpret_cast<elem_t*>(smem_g);        // [128][kLS]: the token tile once, then feature tiles / staged output
  float* bl = reinterpret_cast<float*>(ws + kTN * kLS);  // [N] bias
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * kTM;
  const int nnt = N / kTN;
  const int srow = tid >> 4, scol = (tid & 15) * 8;  // this thread stages rows srow + 16 j, 16 bytes at scol
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 pw[8];
  {
    u32x4 pa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t row = min(m0 + srow + 16 * j, (int64_t)M - 1);  // rows past M: computed, never stored
      pa[j] = *reinterpret_cast<const u32x4*>(A + row * kTK + scol);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) pw[j] = *reinterpret_cast<const u32x4*>(W + (int64_t)(srow + 16 * j) * kTK + scol);
    if (bias)
      for (int i = tid; i < N; i += kGemmThreads) bl[i] = bias[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(ws + (srow + 16 * j) * kLS + scol) = pa[j];
  }
  __syncthreads();
  bf16x8 tok[kTK / 16];
#pragma unroll
  for (int ks = 0; ks < kTK / 16; ++ks) tok[ks] = *reinterpret_cast<const bf16x8*>(ws + (32 * w + l31) * kLS + 16 * ks + 8 * hi);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(ws + (srow + 16 * j) * kLS + scol) = pw[j];
  __syncthreads();
  const int orow = lane >> 4, ocol = (lane & 15) * 8;  // output pass: four rows per pass, 16 bytes per lane
  for (int nt = 0; nt < nnt; ++nt) {
    const bool more = nt + 1 < nnt;
    {
      const int ntn = min(nt + 1, nnt - 1);  // (the last tile re-reads itself: no branch around the loads)
#pragma unroll
      for (int j = 0; j < 8; ++j) pw[j] = *reinterpret_cast<const u32x4*>(W + (int64_t)(ntn * kTN + srow + 16 * j) * kTK + scol);
    }
    f32x16 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < kTK / 16; ++ks) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const bf16x8 feat = *reinterpret_cast<const bf16x8*>(ws + (32 * ct + l31) * kLS + 16 * ks + 8 * hi);
        acc[ct] = rl4co_e16::mfma_32x32x16(feat, tok[ks], acc[ct]);
      }
    }
    // the prefetched tile pinned in its registers before this tile's stores are issued (see linear_bf16_kernel)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(pw[j]));
    rl4co::lds_barrier();  // every wave has issued its products: the feature tile is dead, the output is staged over it
    elem_t* os = ws;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int fl = 32 * ct + 8 * q + 4 * hi;  // feature inside the tile
        float v[4];
        if (bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(bl + nt * kTN + fl);
          v[0] = acc[ct][4 * q] + b4.x;
          v[1] = acc[ct][4 * q + 1] + b4.y;
          v[2] = acc[ct][4 * q + 2] + b4.z;
          v[3] = acc[ct][4 * q + 3] + b4.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[ct][4 * q + i];
        }
        if (relu) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
        }
        bf16x4g o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (elem_t)v[i];
        *reinterpret_cast<bf16x4g*>(os + (32 * w + l31) * kLS + fl) = o;
      }
    }
    rl4co::lds_barrier_wave();
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
      const int tr = 32 * w + 4 * p8 + orow;
      const int64_t row = m0 + tr;
      const u32x4 val = *reinterpret_cast<const u32x4*>(os + tr * kLS + ocol);
      if (row < M) *reinterpret_cast<u32x4*>(out + row * N + nt * kTN + ocol) = val;
    }
    if (more) {
      rl4co::lds_barrier();  // every wave has drained its staged rows
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(ws + (srow + 16 * j) * kLS + scol) = pw[j];
      rl4co::lds_barrier();
    }
  }
}

// K = 128 AND N = 128 (out_proj both ways — six input-gradient launches per REINFORCE step): persistent workgroups. One tile
// per workgroup made every tile pay its own load -> product -> store chain (13 us of a workgroup's life for 0.5 us of
// products; 54 us per launch against 26 at the HBM rate, r05). Here the WEIGHTS are the resident operand — wave w keeps the
// A fragments of its 32 output features in registers for the whole launch — and the workgroup walks token tiles: the next
// tile's rows travel in registers under the products of this one, the tile leaves through the same LDS rows it came in by.
__global__ void __launch_bounds__(kGemmThreads, 3) linear_k128_n128_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                                           const float* __restrict__ bias, int M, int relu,
                                                                           uint16_t* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem_g[];
  elem_t* xs = reinterpret_cast<elem_t*>(smem_g);  // [128 tokens][kLS]: the token tile, then the staged output tile
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  bf16x8 wf[kTK / 16];
#pragma unroll
  for (int ks = 0; ks < kTK / 16; ++ks)
    wf[ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const elem_t*>(W) + (int64_t)(32 * w + l31) * kTK + 16 * ks + 8 * hi);
  float4 bq[4];  // the bias of the four-feature groups this lane's accumulator rows hold
#pragma unroll
  for (int q = 0; q < 4; ++q)
    bq[q] = bias ? *reinterpret_cast<const float4*>(bias + 32 * w + 8 * q + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int ntiles = (M + kTM - 1) / kTM;
  const int srow = tid >> 4, scol = (tid & 15) * 8;  // this thread moves rows srow + 16 j, 16 bytes at scol
  u32x4 pa[8];
  auto fetch = [&](int tile) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t row = min((int64_t)tile * kTM + srow + 16 * j, (int64_t)M - 1);  // rows past M: computed, never stored
      pa[j] = *reinterpret_cast<const u32x4*>(A + row * kTK + scol);
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(xs + (srow + 16 * j) * kLS + scol) = pa[j];
    __syncthreads();
    fetch(min(tile + (int)gridDim.x, ntiles - 1));  // (the last round re-reads a tile it drops: no branch around the loads)
    f32x16 acc[4];  // [token tile tt][feature 32 w + rowmap(r, hi)], token 32 tt + l31
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < kTK / 16; ++ks) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const bf16x8 tok = *reinterpret_cast<const bf16x8*>(xs + (32 * tt + l31) * kLS + 16 * ks + 8 * hi);
        acc[tt] = rl4co_e16::mfma_32x32x16(wf[ks], tok, acc[tt]);
      }
    }
    // the prefetched rows pinned in their registers before this tile's stores are issued (see linear_bf16_kernel)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(pa[j]));
    __syncthreads();  // every wave has read the token tile: the output is staged over it
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4] = {acc[tt][4 * q] + bq[q].x, acc[tt][4 * q + 1] + bq[q].y, acc[tt][4 * q + 2] + bq[q].z, acc[tt][4 * q + 3] + bq[q].w};
        if (relu) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
        }
        *reinterpret_cast<bf16x4g*>(xs + (32 * tt + l31) * kLS + 32 * w + 8 * q + 4 * hi) = rl4co_e16::cvt4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t row = (int64_t)tile * kTM + srow + 16 * j;
      const u32x4 val = *reinterpret_cast<const u32x4*>(xs + (srow + 16 * j) * kLS + scol);
      if (row < M) *reinterpret_cast<u32x4*>(out + row * kTN + scol) = val;
    }
    __syncthreads();  // the staged tile is drained before the next one is committed over it
  }
}

int launch_k128(const void* a, const void* w, const float* bias, int64_t M, int N, int relu, void* out, void* stream) {
  if (N == kTN) {
    const int ntiles = (int)((M + kTM - 1) / kTM);
    hipLaunchKernelGGL(linear_k128_n128_kernel, dim3(ntiles < 768 ? ntiles : 768), dim3(kGemmThreads), kTM * kLS * 2, rl4co::as_stream(stream),
                       static_cast<const uint16_t*>(a), static_cast<const uint16_t*>(w), bias, (int)M, relu, static_cast<uint16_t*>(out));
    RL4CO_HIP_TRY(hipGetLastError());
    return RL4CO_OK;
  }
  const int lds = kTN * kLS * 2 + kMaxLinearN * 4;
  hipLaunchKernelGGL(linear_k128_kernel, dim3((int)((M + kTM - 1) / kTM)), dim3(kGemmThreads), lds, rl4co::as_stream(stream),
                     static_cast<const uint16_t*>(a), static_cast<const uint16_t*>(w), bias, (int)M, N, relu,
                     static_cast<uint16_t*>(out));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

}  // namespace

extern "C" int RL4CO_ENTRY(rl4co_linear)(const void* a, const void* w, const float* bias, const void* mask, const void* residual,
                                 int64_t M, int N, int K, int relu, void* out, void* stream) {
  RL4CO_REQUIRE(a && w && out);
  RL4CO_REQUIRE(M > 0 && M < (int64_t)1 << 31 && N > 0 && K > 0 && N % kTN == 0 && K % kTK == 0);
  RL4CO_REQUIRE(!(relu && mask) && !(mask && residual));
  RL4CO_REQUIRE(N <= kMaxLinearN);
  if (K == kTK && !mask && !residual) return launch_k128(a, w, bias, M, N, relu, out, stream);
  const int lds = (kTM + kTN) * kLS * 2 + kMaxLinearN * 4;
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int blocks = (int)((M + kTM - 1) / kTM);
  hipLaunchKernelGGL(linear_bf16_kernel, dim3(blocks), dim3(kGemmThreads), lds, rl4co::as_stream(stream),
                     static_cast<const uint16_t*>(a), static_cast<const uint16_t*>(w), bias,
                     static_cast<const uint16_t*>(mask), static_cast<const uint16_t*>(residual), (int)M, N, K, relu,
                     static_cast<uint16_t*>(out));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradients of those linear layers: dW[N,K] = dY[M,N]^T . X[M,K], the contraction running
// over the M ~ 4e5 token rows. The library GEMM takes 0.7 - 1.0 ms per call here (24 calls per
// REINFORCE step); the work is two streaming reads. Split over the token rows: workgroup
// (tile, chunk) accumulates one 128 x 128 tile of dW over its chunk of rows on
// v_mfma_f32_16x16x16_bf16 — both operands want the TOKEN on the contraction slots, so the
// [token][column] LDS tiles are read through ds_read_b64_tr_b16 — and writes an fp32 partial;
// the chunks are summed afterwards (deterministic, no atomics).
namespace {

constexpr int kWT = 64;        // token rows per LDS step: the step is one exposed HBM round trip, so the products per
                               // step must outlast it (at 32 rows two resident workgroups kept the matrix pipe ~40 % busy)
#ifndef RL4CO_WLS
#define RL4CO_WLS 144  // (probe knob: tools/kernel_variant.sh am_train_ops.hip <name> "-DRL4CO_WLS=136")
#endif
// LDS row stride (bf16). Both tiles are READ only through ds_read_b64_tr_b16: a 32-lane group takes eight rows x 32 bytes,
// and with 272-byte rows (68 dwords) consecutive rows overlap in four of their eight banks — SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 0.50 (profiles/r05_c4_train_pmc.json). 288-byte rows (72 dwords = 8 mod 64) put the eight rows on
// disjoint banks: -3 % on the wide shapes (r05: 123.6 -> 119.0 us at N = 512, K = 128).
constexpr int kWLS = RL4CO_WLS;

typedef elem_t bf16x4w __attribute__((ext_vector_type(4)));
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4w lds_s16x4w;

__device__ inline bf16x4w lds_tr_w(const elem_t* p) {
  const s16x4w v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w*)p);
  return __builtin_bit_cast(bf16x4w, v);
}

// (launch bounds WITH a minimum of two workgroups per CU, r05: left to 512 registers the compiler put the 72 accumulators in
// AGPRs with a different destination than source register for every product and moved all of them back through
// v_accvgpr_read / _write each step — 112 of the loop's 180 VALU instructions, none in the source)
__global__ void __launch_bounds__(256, 2) wgrad_bf16_kernel(const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X, int M, int N,
                                                         int K, int rows_per_chunk, int n_chunks, float* __restrict__ partial,
                                                         float* __restrict__ partial_bias, int64_t pstride, int64_t bstride) {
  __shared__ __align__(16) elem_t dyt[kWT * kWLS];  // [32 tokens][128 output features of this tile]
  __shared__ __align__(16) elem_t xt[kWT * kWLS];   // [32 tokens][128 input features of this tile]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int kt_n = K / 128;
  // workgroup -> (tile, row chunk). The tiles of ONE chunk re-read the same rows of the operand they share (X for
  // the wide-N layers, dY for the wide-K one): workgroup b lands on XCD b % 8, so with the chunk count a multiple of 8
  // the tiles of a chunk take consecutive slots of one XCD and meet in its L2 instead of each pulling the rows from HBM
  const int tiles = gridDim.x / n_chunks;
  int tile, chunk;
  if ((n_chunks & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tile = slot % tiles;
    chunk = xcd + 8 * (slot / tiles);
  } else {
    tile = blockIdx.x % tiles;
    chunk = blockIdx.x / tiles;
  }
  const int nt = tile / kt_n, kt = tile % kt_n;
  const int64_t m_begin = (int64_t)chunk * rows_per_chunk;
  const int64_t m_end = min((int64_t)M, m_begin + rows_per_chunk);
  const int tro = (4 * g + (tl >> 2)) * kWLS + 4 * (tl & 3);
  f32x4w acc[2][8];  // [feature block 32 w + 16 nb][input block 16 kb]
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) acc[nb][kb] = f32x4w{0.0f, 0.0f, 0.0f, 0.0f};
  // bias gradient = column sums of dY: one more MFMA per feature block against a ones operand
  // (tiles with kt == 0 only); every accumulator column then holds the same sum
  const bool with_bias = partial_bias != nullptr && kt == 0;
  f32x4w accb[2] = {f32x4w{0.0f, 0.0f, 0.0f, 0.0f}, f32x4w{0.0f, 0.0f, 0.0f, 0.0f}};
  typedef elem_t bf16x8o __attribute__((ext_vector_type(8)));
  const bf16x8o ones = {(elem_t)1.0f, (elem_t)1.0f, (elem_t)1.0f, (elem_t)1.0f, (elem_t)1.0f, (elem_t)1.0f, (elem_t)1.0f, (elem_t)1.0f};
  const int lrow = tid >> 3, lcol = (tid & 7) * 16;  // this thread stages 32 bytes of each tile row
  typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
  // the rows of step i + 1 are requested before the products of step i (registers), so a step no longer opens with
  // an exposed HBM round trip: with two workgroups per CU that latency set the kernel's pace (3.3 TB/s of reads)
  u32x4w a0[kWT / 32], a1[kWT / 32], b0[kWT / 32], b1[kWT / 32];  // a thread stages row lrow of every 32-row group
  auto fetch = [&](int64_t m0) {
#pragma unroll
    for (int rg = 0; rg < kWT / 32; ++rg) {
      const int64_t row = min(m0 + 32 * rg + lrow, m_end - 1);  // rows past the chunk: re-read, zeroed at the commit
      const uint16_t* pa = dY + row * N + nt * 128 + lcol;
      const uint16_t* pb = X + row * K + kt * 128 + lcol;
      a0[rg] = *reinterpret_cast<const u32x4w*>(pa);
      a1[rg] = *reinterpret_cast<const u32x4w*>(pa + 8);
      b0[rg] = *reinterpret_cast<const u32x4w*>(pb);
      b1[rg] = *reinterpret_cast<const u32x4w*>(pb + 8);
    }
  };
  if (m_begin < m_end) fetch(m_begin);
  for (int64_t m0 = m_begin; m0 < m_end; m0 += kWT) {
    __syncthreads();  // the previous step's fragments are consumed
#pragma unroll
    for (int rg = 0; rg < kWT / 32; ++rg) {
      const bool live = m0 + 32 * rg + lrow < m_end;
      const u32x4w z = {0u, 0u, 0u, 0u};
      const int r = 32 * rg + lrow;
      *reinterpret_cast<u32x4w*>(dyt + r * kWLS + lcol) = live ? a0[rg] : z;
      *reinterpret_cast<u32x4w*>(dyt + r * kWLS + lcol + 8) = live ? a1[rg] : z;
      *reinterpret_cast<u32x4w*>(xt + r * kWLS + lcol) = live ? b0[rg] : z;
      *reinterpret_cast<u32x4w*>(xt + r * kWLS + lcol + 8) = live ? b1[rg] : z;
    }
    __syncthreads();
    fetch(min(m0 + kWT, m_end - 1));  // (the last step re-reads a row it drops: no branch around the loads)
    // 32 tokens per product: v_mfma_f32_16x16x32_bf16 (gfx950) runs at twice the rate of the 16-deep instruction, and
    // the contraction runs over tokens, so ANY assignment of tokens to its 32 slots is right as long as both operands
    // use the same one — two 16-token transpose reads side by side (tokens 16 ts + 4 g .. and 16 (ts + 1) + 4 g ..)
    typedef elem_t bf16x8w __attribute__((ext_vector_type(8)));
    auto pair = [&](const elem_t* p) {
      const bf16x4w lo = lds_tr_w(p), hi = lds_tr_w(p + 16 * kWLS);
      return bf16x8w{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
#pragma unroll
    for (int ts = 0; ts < kWT / 16; ts += 2) {
      bf16x8w af[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) af[nb] = pair(dyt + 16 * ts * kWLS + 32 * w + 16 * nb + tro);
      if (with_bias) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) accb[nb] = rl4co_e16::mfma_16x16x32(af[nb], ones, accb[nb]);
      }
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const bf16x8w bf = pair(xt + 16 * ts * kWLS + 16 * kb + tro);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb][kb] = rl4co_e16::mfma_16x16x32(af[nb], bf, acc[nb][kb]);
      }
    }
  }
  float* out = partial + (int64_t)chunk * pstride;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(int64_t)(nt * 128 + 32 * w + 16 * nb + 4 * g + r) * K + kt * 128 + 16 * kb + tl] = acc[nb][kb][r];
  if (with_bias && tl == 0) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) partial_bias[(int64_t)chunk * bstride + nt * 128 + 32 * w + 16 * nb + 4 * g + r] = accb[nb][r];
  }
}

}  // namespace

extern "C" int RL4CO_ENTRY(rl4co_wgrad)(const void* dy, const void* x, int64_t M, int N, int K, int chunks, float* partial,
                                float* partial_bias, int64_t chunk_stride, void* stream) {
  RL4CO_REQUIRE(chunk_stride == 0 || chunk_stride >= (int64_t)N * K);
  const int64_t pstride = chunk_stride ? chunk_stride : (int64_t)N * K, bstride = chunk_stride ? chunk_stride : (int64_t)N;
  RL4CO_REQUIRE(dy && x && partial);
  RL4CO_REQUIRE(M > 0 && M < (int64_t)1 << 31 && N > 0 && K > 0 && N % 128 == 0 && K % 128 == 0 && chunks > 0 && chunks <= 65535);
  const int rows = (int)(((M + chunks - 1) / chunks + kWT - 1) / kWT * kWT);
  hipLaunchKernelGGL(wgrad_bf16_kernel, dim3((N / 128) * (K / 128) * chunks), dim3(256), 0, rl4co::as_stream(stream),
                     static_cast<const uint16_t*>(dy), static_cast<const uint16_t*>(x), (int)M, N, K, rows, chunks, partial,
                     partial_bias, pstride, bstride);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// ------------------------------------------------------------------------------------------------
// SkipConnection + Normalization("batch") in training (the AttentionModel default,
// nn/ops.py:30-54, zoo/am/policy.py:50-122): BatchNorm1d over the M = B x nodes token rows, i.e.
// statistics ACROSS instances — two passes over HBM instead of one:
//   stats    per-channel sum and sum of squared deviations of y = x + s (bf16-rounded, written)
//            by a shifted two-term formula: partial sums per workgroup -> fp32 atomics [2][128]
//   apply    out = (y - mean) rstd gamma + beta
//   backward reduce: sum dout, sum dout * xh  -> [2][128] ; apply: dy = rstd gamma (dout - m1 - xh m2)
// Running statistics are updated by the host wrapper exactly as nn.BatchNorm1d does.
namespace {

constexpr int kBnRows = 64;  // token rows per workgroup pass

// y = x + s (optional), per-channel partial sums of y and y^2 in fp32 -> atomics. The variance is
// formed as E[y^2] - mean^2 in fp32 from fp32 sums of bf16 values: with |mean| <~ sigma (post-skip
// activations of a normalised residual stream) the cancellation costs < 1e-6 relative.
__global__ void __launch_bounds__(256) bn_stats_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ s, int64_t M,
                                                       uint32_t* __restrict__ y, float* __restrict__ sums) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  for (int64_t r0 = (int64_t)blockIdx.x * kBnRows; r0 < M; r0 += (int64_t)gridDim.x * kBnRows) {
    // eight rows per thread in flight: loads first (rows past M re-read row M - 1 and are ignored) — a guard per
    // row makes the compiler wait for each load before issuing the next
    for (int i0 = 0; i0 < kBnRows / 4; i0 += 8) {
      uint32_t xv[8], sv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t r = min(r0 + q + 4 * (i0 + j), M - 1);
        xv[j] = x[r * 64 + cp];
        sv[j] = s ? s[r * 64 + cp] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t r = r0 + q + 4 * (i0 + j);
        if (r < M) {
          uint32_t ys = xv[j];
          if (s) {
            ys = pack_bf16(bf16_lo(xv[j]) + bf16_lo(sv[j]), bf16_hi(xv[j]) + bf16_hi(sv[j]));
            y[r * 64 + cp] = ys;
          }
          const float v0 = bf16_lo(ys), v1 = bf16_hi(ys);
          a0 += v0;
          a1 += v1;
          b0 = fmaf(v0, v0, b0);
          b1 = fmaf(v1, v1, b1);
        }
      }
    }
  }
  red[tid * 4] = a0;
  red[tid * 4 + 1] = a1;
  red[tid * 4 + 2] = b0;
  red[tid * 4 + 3] = b1;
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = (red[cp * 4 + e] + red[(64 + cp) * 4 + e]) + (red[(128 + cp) * 4 + e] + red[(192 + cp) * 4 + e]);
      // sums[0][c] = sum y, sums[1][c] = sum y^2, channels 2 cp + (e & 1)
      unsafeAtomicAdd(sums + (e >> 1) * kD + 2 * cp + (e & 1), v);
    }
  }
}

// out = (y - mean) * rstd * gamma + beta
__global__ void __launch_bounds__(256) bn_apply_kernel(const uint32_t* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int64_t M, uint32_t* __restrict__ out) {
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const float m0 = mean[2 * cp], m1 = mean[2 * cp + 1];
  const float k0 = rstd[2 * cp] * gamma[2 * cp], k1 = rstd[2 * cp + 1] * gamma[2 * cp + 1];
  const float b0 = beta[2 * cp], b1 = beta[2 * cp + 1];
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + q; r < M; r += 4 * stride) {  // four rows in flight per thread
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = y[min(r + j * stride, M - 1) * 64 + cp];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (r + j * stride < M)
        out[(r + j * stride) * 64 + cp] = pack_bf16(fmaf(bf16_lo(v[j]) - m0, k0, b0), fmaf(bf16_hi(v[j]) - m1, k1, b1));
  }
}

// out = ((x + s) - mean) * rstd * gamma + beta: the skip connection and eval-mode batch norm of an encoder layer in one
// pass over the token rows (inference on graphs beyond the fused encoder's 128 nodes); the sum stays in fp32
__global__ void __launch_bounds__(256) skip_bn_eval_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ sk,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int64_t M,
                                                           uint32_t* __restrict__ out) {
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const float m0 = mean[2 * cp], m1 = mean[2 * cp + 1];
  const float k0 = rstd[2 * cp] * gamma[2 * cp], k1 = rstd[2 * cp + 1] * gamma[2 * cp + 1];
  const float b0 = beta[2 * cp], b1 = beta[2 * cp + 1];
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + q; r < M; r += 4 * stride) {  // four rows in flight per thread
    uint32_t v[4], u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = min(r + j * stride, M - 1);
      v[j] = x[row * 64 + cp];
      u[j] = sk[row * 64 + cp];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (r + j * stride < M)
        out[(r + j * stride) * 64 + cp] = pack_bf16(fmaf((bf16_lo(v[j]) + bf16_lo(u[j])) - m0, k0, b0),
                                                    fmaf((bf16_hi(v[j]) + bf16_hi(u[j])) - m1, k1, b1));
  }
}

// sums[0][c] += sum dout ; sums[1][c] += sum dout * xh
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const uint32_t* __restrict__ dout, const uint32_t* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, int64_t M,
                                                            float* __restrict__ sums) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const float m0 = mean[2 * cp], m1 = mean[2 * cp + 1], r0s = rstd[2 * cp], r1s = rstd[2 * cp + 1];
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  for (int64_t r0 = (int64_t)blockIdx.x * kBnRows; r0 < M; r0 += (int64_t)gridDim.x * kBnRows) {
    for (int i0 = 0; i0 < kBnRows / 4; i0 += 8) {  // loads first, as in bn_stats_kernel
      uint32_t dv[8], yv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t r = min(r0 + q + 4 * (i0 + j), M - 1);
        dv[j] = dout[r * 64 + cp];
        yv[j] = y[r * 64 + cp];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (r0 + q + 4 * (i0 + j) < M) {
          const float d0 = bf16_lo(dv[j]), d1 = bf16_hi(dv[j]);
          a0 += d0;
          a1 += d1;
          b0 = fmaf(d0, (bf16_lo(yv[j]) - m0) * r0s, b0);
          b1 = fmaf(d1, (bf16_hi(yv[j]) - m1) * r1s, b1);
        }
      }
    }
  }
  red[tid * 4] = a0;
  red[tid * 4 + 1] = a1;
  red[tid * 4 + 2] = b0;
  red[tid * 4 + 3] = b1;
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = (red[cp * 4 + e] + red[(64 + cp) * 4 + e]) + (red[(128 + cp) * 4 + e] + red[(192 + cp) * 4 + e]);
      unsafeAtomicAdd(sums + (e >> 1) * kD + 2 * cp + (e & 1), v);
    }
  }
}

// dy = rstd * gamma * (dout - sums[0] / M - xh * sums[1] / M)
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const uint32_t* __restrict__ dout, const uint32_t* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums, int64_t M,
                                                           uint32_t* __restrict__ dy) {
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  const float inv_m = 1.0f / (float)M;
  const float m0 = mean[2 * cp], m1 = mean[2 * cp + 1], r0s = rstd[2 * cp], r1s = rstd[2 * cp + 1];
  const float k0 = r0s * gamma[2 * cp], k1 = r1s * gamma[2 * cp + 1];
  const float sd0 = sums[2 * cp] * inv_m, sd1 = sums[2 * cp + 1] * inv_m;
  const float sx0 = sums[kD + 2 * cp] * inv_m, sx1 = sums[kD + 2 * cp + 1] * inv_m;
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + q; r < M; r += 4 * stride) {  // four rows in flight per thread
    uint32_t dv[4], yv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t rr = min(r + j * stride, M - 1);
      dv[j] = dout[rr * 64 + cp];
      yv[j] = y[rr * 64 + cp];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (r + j * stride < M) {
        const float x0 = (bf16_lo(yv[j]) - m0) * r0s, x1 = (bf16_hi(yv[j]) - m1) * r1s;
        dy[(r + j * stride) * 64 + cp] =
            pack_bf16(k0 * (bf16_lo(dv[j]) - sd0 - x0 * sx0), k1 * (bf16_hi(dv[j]) - sd1 - x1 * sx1));
      }
    }
  }
}

}  // namespace

extern "C" int RL4CO_ENTRY(rl4co_skip_bnorm_stats)(const void* x, const void* s, int64_t M, void* y, float* sums, void* stream) {
  RL4CO_REQUIRE(x && sums && M > 0 && (s == nullptr || y != nullptr));
  const int blocks = (int)min((int64_t)2048, (M + kBnRows - 1) / kBnRows);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks), dim3(256), 0, rl4co::as_stream(stream), static_cast<const uint32_t*>(x),
                     static_cast<const uint32_t*>(s), M, static_cast<uint32_t*>(y), sums);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_bnorm_apply)(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                      int64_t M, void* out, void* stream) {
  RL4CO_REQUIRE(y && mean && rstd && gamma && beta && out && M > 0);
  const int blocks = (int)min((int64_t)8192, (M + 3) / 4);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, rl4co::as_stream(stream), static_cast<const uint32_t*>(y), mean,
                     rstd, gamma, beta, M, static_cast<uint32_t*>(out));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_skip_bnorm_eval)(const void* x, const void* skip, const float* mean, const float* rstd, const float* gamma,
                                          const float* beta, int64_t M, void* out, void* stream) {
  RL4CO_REQUIRE(x && skip && mean && rstd && gamma && beta && out && M > 0);
  const int blocks = (int)min((int64_t)8192, (M + 3) / 4);
  hipLaunchKernelGGL(skip_bn_eval_kernel, dim3(blocks), dim3(256), 0, rl4co::as_stream(stream), static_cast<const uint32_t*>(x),
                     static_cast<const uint32_t*>(skip), mean, rstd, gamma, beta, M, static_cast<uint32_t*>(out));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_bnorm_bwd)(const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma,
                                    int64_t M, float* sums, void* dy, void* stream) {
  RL4CO_REQUIRE(dout && y && mean && rstd && gamma && sums && dy && M > 0);
  hipStream_t st = rl4co::as_stream(stream);
  const int rb = (int)min((int64_t)2048, (M + kBnRows - 1) / kBnRows);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(rb), dim3(256), 0, st, static_cast<const uint32_t*>(dout),
                     static_cast<const uint32_t*>(y), mean, rstd, M, sums);
  const int ab = (int)min((int64_t)8192, (M + 3) / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ab), dim3(256), 0, st, static_cast<const uint32_t*>(dout),
                     static_cast<const uint32_t*>(y), mean, rstd, gamma, sums, M, static_cast<uint32_t*>(dy));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// ------------------------------------------------------------------------------------------------
// a11 init embedding in training: out[m,:] = W[128,F] . feats[m,:F] + b, F = 2 (x, y) or 3 (x, y,
// demand) (env_embeddings/init.py:55-68,115-136). A K = 2 "GEMM" costs the library 0.97 ms at
// 409 600 rows; it is 128 fused multiply-adds per row.
namespace {
__global__ void __launch_bounds__(256) init_embed_kernel(const float* __restrict__ feats, const float* __restrict__ W,
                                                         const float* __restrict__ b, int64_t M, int F, uint32_t* __restrict__ out) {
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  float w0[6], w1[6];
#pragma unroll
  for (int f = 0; f < 6; ++f) {
    w0[f] = f < F ? W[(2 * cp) * F + f] : 0.0f;
    w1[f] = f < F ? W[(2 * cp + 1) * F + f] : 0.0f;
  }
  const float b0 = b[2 * cp], b1 = b[2 * cp + 1];
  for (int64_t r = (int64_t)blockIdx.x * 4 + q; r < M; r += (int64_t)gridDim.x * 4) {
    float a0 = b0, a1 = b1;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      if (f < F) {
        const float x = feats[r * F + f];
        a0 = fmaf(w0[f], x, a0);
        a1 = fmaf(w1[f], x, a1);
      }
    }
    out[r * 64 + cp] = pack_bf16(a0, a1);
  }
}
}  // namespace

namespace {
constexpr int kInitWgradBlocks = 1024;
// Thread = channel pair (tid & 63) x row lane (tid >> 6): a wave reads one 256-byte row of dout per iteration,
// the row's F features are a broadcast load. Per-block partial sums, reduced by the caller in a fixed order.
__global__ void __launch_bounds__(256) init_embed_wgrad_kernel(const uint32_t* __restrict__ dout, const float* __restrict__ feats,
                                                               int64_t M, int F, float* __restrict__ partial) {
  const int tid = threadIdx.x, cp = tid & 63, q = tid >> 6;
  __shared__ float red[4][128][7];
  float a0[7], a1[7];
#pragma unroll
  for (int f = 0; f < 7; ++f) a0[f] = a1[f] = 0.0f;
  const int64_t per = (M + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = min(M, lo + per);
  for (int64_t r = lo + q; r < hi; r += 4) {
    const uint32_t pk = dout[r * 64 + cp];
    const float d0 = rl4co_e16::lo(pk), d1 = rl4co_e16::hi(pk);
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      if (f < F) {
        const float x = feats[r * F + f];
        a0[f] = fmaf(d0, x, a0[f]);
        a1[f] = fmaf(d1, x, a1[f]);
      }
    }
    a0[6] += d0;
    a1[6] += d1;
  }
#pragma unroll
  for (int f = 0; f < 7; ++f) {
    red[q][2 * cp][f] = a0[f];
    red[q][2 * cp + 1][f] = a1[f];
  }
  __syncthreads();
  float* out = partial + (int64_t)blockIdx.x * 128 * (F + 1);
  for (int i = tid; i < 128 * (F + 1); i += 256) {
    const int c = i / (F + 1), f = i % (F + 1);
    const int src = f < F ? f : 6;
    out[i] = ((red[0][c][src] + red[1][c][src]) + red[2][c][src]) + red[3][c][src];
  }
}
}  // namespace

extern "C" int RL4CO_ENTRY(rl4co_init_embed_wgrad)(const void* dout, const float* feats, int64_t M, int F, float* partial,
                                           int* blocks_out, void* stream) {
  RL4CO_REQUIRE(M > 0 && F >= 1 && F <= 6);
  const int blocks = (int)min((int64_t)kInitWgradBlocks, (M + 63) / 64);
  if (blocks_out) *blocks_out = blocks;
  if (partial == nullptr) return RL4CO_OK;  // size query
  RL4CO_REQUIRE(dout && feats);
  hipLaunchKernelGGL(init_embed_wgrad_kernel, dim3(blocks), dim3(256), 0, rl4co::as_stream(stream),
                     static_cast<const uint32_t*>(dout), feats, M, F, partial);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int RL4CO_ENTRY(rl4co_init_embed)(const float* feats, const float* w, const float* b, int64_t M, int F, void* out, void* stream) {
  RL4CO_REQUIRE(feats && w && b && out && M > 0 && F >= 1 && F <= 6);
  const int blocks = (int)min((int64_t)8192, (M + 3) / 4);
  hipLaunchKernelGGL(init_embed_kernel, dim3(blocks), dim3(256), 0, rl4co::as_stream(stream), feats, w, b, M, F,
                     static_cast<uint32_t*>(out));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
