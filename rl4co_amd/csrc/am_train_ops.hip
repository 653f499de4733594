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

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kThreads = 256;
constexpr int kMaxRows = 32;  // nodes per thread: N <= 128

__device__ inline float bf16_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ inline float bf16_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ inline uint32_t pack_bf16(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}

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
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    v[i][0] = 0.0f;
    v[i][1] = 0.0f;
    if (n < N) {
      const uint32_t a = x[base + (int64_t)n * (kD / 2) + cp], b = s[base + (int64_t)n * (kD / 2) + cp];
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
#pragma unroll
  for (int i = 0; i < kMaxRows; ++i) {
    const int n = q + 4 * i;
    xh[i][0] = xh[i][1] = dd[i][0] = dd[i][1] = 0.0f;
    if (n < N) {
      const uint32_t a = dout[base + (int64_t)n * (kD / 2) + cp], b = y[base + (int64_t)n * (kD / 2) + cp];
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
    unsafeAtomicAdd(dgamma + 2 * cp, s_dx[0]);
    unsafeAtomicAdd(dgamma + 2 * cp + 1, s_dx[1]);
    unsafeAtomicAdd(dbeta + 2 * cp, s_d[0]);
    unsafeAtomicAdd(dbeta + 2 * cp + 1, s_d[1]);
  }
}

}  // namespace

extern "C" int rl4co_skip_inorm_max_nodes(void) { return 4 * kMaxRows; }

extern "C" int rl4co_skip_inorm_fwd_bf16(const void* x, const void* s, const float* gamma, const float* beta, float eps, int B,
                                         int N, void* y, void* out, float* mean, float* rstd, void* stream) {
  RL4CO_REQUIRE(x && s && gamma && beta && y && out && mean && rstd);
  RL4CO_REQUIRE(B > 0 && N >= 1 && N <= 4 * kMaxRows && eps > 0.0f);
  hipLaunchKernelGGL(skip_inorm_fwd_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream),
                     static_cast<const uint32_t*>(x), static_cast<const uint32_t*>(s), gamma, beta, eps, N,
                     static_cast<uint32_t*>(y), static_cast<uint32_t*>(out), mean, rstd);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_skip_inorm_bwd_bf16(const void* dout, const void* y, const float* gamma, const float* mean,
                                         const float* rstd, int B, int N, void* dy, float* dgamma, float* dbeta, void* stream) {
  RL4CO_REQUIRE(dout && y && gamma && mean && rstd && dy && dgamma && dbeta);
  RL4CO_REQUIRE(B > 0 && N >= 1 && N <= 4 * kMaxRows);
  hipLaunchKernelGGL(skip_inorm_bwd_kernel, dim3(B), dim3(kThreads), 0, rl4co::as_stream(stream),
                     static_cast<const uint32_t*>(dout), static_cast<const uint32_t*>(y), gamma, mean, rstd, N,
                     static_cast<uint32_t*>(dy), dgamma, dbeta);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
