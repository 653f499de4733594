// am_logit_logp.hip — log p(a_t | s_t) of GIVEN actions from the pointer's raw logits, all steps of all trajectories at once,
// forward and backward (r06): the tail of the dense re-evaluation (policy.evaluate_log_probs) —
//   u = raw / sqrt(128) ; z = C tanh(u) (C = tanh_clipping, 0: z = u) ; z = -inf where infeasible ; z /= temperature ;
//   logp = log_softmax(z)[a]                         nn/attention.py:291-293, utils/decoding.py:169-188, :381
// as one pass over the [rows, N] fp32 logits instead of five elementwise / reduction launches that each write a tensor of
// that size for autograd (at 512 x 8 starts x 200 steps x 200 nodes: 655 MB apiece). The backward rebuilds the probabilities
// from the saved log-sum-exp:  d raw_j = g (1[j = a] - p_j) dz/du_j / sqrt(128), 0 where infeasible.
// One wave per row (a trajectory's step); fp32 throughout; tolerance-tested against the torch chain it replaces.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kInvSqrtD = 0.08838834764831845f;  // 1 / sqrt(128)
constexpr int kRowsPerBlock = 4;

struct Clip {
  float c, inv_temp;
  __device__ inline float z(float raw, float& dzdu) const {
    const float u = raw * kInvSqrtD;
    if (c > 0.0f) {
      const float ex = __expf(-2.0f * fabsf(u));
      const float th = copysignf((1.0f - ex) / (1.0f + ex), u);
      dzdu = c * inv_temp * (1.0f - th * th);
      return th * c * inv_temp;
    }
    dzdu = inv_temp;
    return u * inv_temp;
  }
};

__device__ inline bool feasible(const uint32_t* bits, int j) { return bits == nullptr || ((bits[j >> 5] >> (j & 31)) & 1u) != 0; }

__global__ void __launch_bounds__(64 * kRowsPerBlock) logit_logp_fwd_kernel(const float* __restrict__ raw, const uint32_t* __restrict__ bits,
                                                                            int words, const int64_t* __restrict__ actions, int64_t rows,
                                                                            int N, float c, float inv_temp, float* __restrict__ logp,
                                                                            float* __restrict__ lse, int32_t* __restrict__ err) {
  const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* row = raw + r * N;
  const uint32_t* brow = bits ? bits + r * words : nullptr;
  const Clip clip{c, inv_temp};
  float zmax = kNegInf;
  bool nan = false;
  for (int j = lane; j < N; j += 64) {
    float d;
    const float v = row[j];
    nan |= v != v;
    if (feasible(brow, j)) zmax = fmaxf(zmax, clip.z(v, d));
  }
  zmax = rl4co::bfly_max<1, 64>(zmax);
  float se = 0.0f;
  for (int j = lane; j < N; j += 64) {
    float d;
    if (feasible(brow, j)) se += __expf(clip.z(row[j], d) - zmax);
  }
  se = rl4co::bfly_sum<1, 64>(se);
  const float l = zmax + __logf(se);
  const bool any_nan = __any(nan);
  if (lane == 0) {
    int64_t a = actions[r];
    int bad = 0;
    if (a < 0 || a >= N) {
      a = 0;
      bad = RL4CO_EBIT_INFEASIBLE;
    }
    float d;
    const float za = feasible(brow, (int)a) ? clip.z(row[a], d) : kNegInf;
    logp[r] = za - l;
    lse[r] = l;
    if (any_nan) bad |= RL4CO_EBIT_NAN_LOGIT;  // nn/attention.py:295-296
    if (bad && err) atomicOr(err, bad);
  }
}

__global__ void __launch_bounds__(64 * kRowsPerBlock) logit_logp_bwd_kernel(const float* __restrict__ raw, const uint32_t* __restrict__ bits,
                                                                            int words, const int64_t* __restrict__ actions,
                                                                            const float* __restrict__ lse, const float* __restrict__ g,
                                                                            int64_t rows, int N, float c, float inv_temp,
                                                                            float* __restrict__ draw) {
  const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* row = raw + r * N;
  float* drow = draw + r * N;
  const uint32_t* brow = bits ? bits + r * words : nullptr;
  const Clip clip{c, inv_temp};
  const float l = lse[r], gr = g[r];
  int64_t a = actions[r];
  if (a < 0 || a >= N) a = 0;
  for (int j = lane; j < N; j += 64) {
    float d, out = 0.0f;
    if (feasible(brow, j)) {
      const float z = clip.z(row[j], d);
      out = gr * ((j == a ? 1.0f : 0.0f) - __expf(z - l)) * d * kInvSqrtD;
    }
    drow[j] = out;
  }
}

}  // namespace

extern "C" int rl4co_logit_logp_fwd(const float* raw, const uint32_t* mask_bits, int mask_words, const int64_t* actions, int64_t rows,
                                    int N, float tanh_clipping, float temperature, float* logp, float* lse, int32_t* err,
                                    void* stream) {
  RL4CO_REQUIRE(raw && actions && logp && lse && rows > 0 && N >= 1 && temperature > 0.0f && tanh_clipping >= 0.0f);
  RL4CO_REQUIRE(mask_bits == nullptr || mask_words * 32 >= N);
  RL4CO_REQUIRE((rows + kRowsPerBlock - 1) / kRowsPerBlock < (1ll << 31));
  hipLaunchKernelGGL(logit_logp_fwd_kernel, dim3((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)), dim3(64 * kRowsPerBlock), 0,
                     rl4co::as_stream(stream), raw, mask_bits, mask_words, actions, rows, N, tanh_clipping, 1.0f / temperature, logp,
                     lse, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_logit_logp_bwd(const float* raw, const uint32_t* mask_bits, int mask_words, const int64_t* actions, const float* lse,
                                    const float* grad_logp, int64_t rows, int N, float tanh_clipping, float temperature, float* d_raw,
                                    void* stream) {
  RL4CO_REQUIRE(raw && actions && lse && grad_logp && d_raw && rows > 0 && N >= 1 && temperature > 0.0f && tanh_clipping >= 0.0f);
  RL4CO_REQUIRE(mask_bits == nullptr || mask_words * 32 >= N);
  RL4CO_REQUIRE((rows + kRowsPerBlock - 1) / kRowsPerBlock < (1ll << 31));
  hipLaunchKernelGGL(logit_logp_bwd_kernel, dim3((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)), dim3(64 * kRowsPerBlock), 0,
                     rl4co::as_stream(stream), raw, mask_bits, mask_words, actions, lse, grad_logp, rows, N, tanh_clipping,
                     1.0f / temperature, d_raw);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
