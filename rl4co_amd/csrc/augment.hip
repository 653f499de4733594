// augment.hip — SURVEY.md §8f row N3: the POMO evaluation path around the rollout.
//
//   rl4co_augment_dihedral8_f32   data/transforms.py:16-46    the 8 symmetries of the unit square
//   rl4co_augment_symmetric_f32   data/transforms.py:49-87    rotation about the centre (+ axis swap), one angle per row
//   rl4co_pomo_best               zoo/pomo/model.py:112-140   best start per augmentation, best augmentation per
//                                 (utils/ops.py:33-66)        instance, and the two action gathers, in ONE launch
//
// All three are HBM-bound byte shuffling. The augmentation kernels read every instance ONCE and write the aug-major
// [A*B, N, 2] layout the multistart rollout consumes directly (the reference materialises the batchified copy first
// — A x the input — and then builds each transform with split / cat passes over it: 8 reads + 24 writes of the batch
// per feature for dihedral-8; here 1 read + 8 writes). fp32 arithmetic in the reference's operation order
// (1 - x; cos*x - sin*y with separate multiplies: the library is built with -ffp-contract=off), so the outputs are
// bit-identical to the reference's for the same inputs — tests/test_data_cpu.py holds the C restatement
// (oracle/rollout_ref.c) against the reference source, tests/test_gpu_data.py the kernels against both.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

// one thread = one coordinate pair of one instance; 8 coalesced 8-byte stores, one per augmentation block
__global__ void __launch_bounds__(256) dihedral8_kernel(const float2* __restrict__ xy, int64_t pairs, float2* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * blockDim.x) {
    const float2 p = xy[i];
    const float x = p.x, y = p.y, mx = 1.0f - p.x, my = 1.0f - p.y;
    // transforms.py:27-37, in the reference's order: (x,y) (1-x,y) (x,1-y) (1-x,1-y) (y,x) (1-y,x) (y,1-x) (1-y,1-x)
    out[i] = make_float2(x, y);
    out[i + pairs] = make_float2(mx, y);
    out[i + 2 * pairs] = make_float2(x, my);
    out[i + 3 * pairs] = make_float2(mx, my);
    out[i + 4 * pairs] = make_float2(y, x);
    out[i + 5 * pairs] = make_float2(my, x);
    out[i + 6 * pairs] = make_float2(y, mx);
    out[i + 7 * pairs] = make_float2(my, mx);
  }
}

// output row r = a * B + b reads instance b; (cos, sin, swap) are per OUTPUT row
__global__ void __launch_bounds__(256) symmetric_kernel(const float2* __restrict__ xy, const float* __restrict__ cs,
                                                        const float* __restrict__ sn, const uint8_t* __restrict__ swap_axes,
                                                        int B, int N, int64_t total, float offset, float2* __restrict__ out) {
  const int64_t per_block = (int64_t)B * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / N;
    const float2 p = xy[i % per_block];
    const float c = cs[r], s = sn[r];
    const float x = p.x - offset, y = p.y - offset;
    const float xp = c * x - s * y;  // transforms.py:60-61 (two products, one subtraction / addition: no fma)
    const float yp = s * x + c * y;
    out[i] = swap_axes[r] ? make_float2(yp + offset, xp + offset) : make_float2(xp + offset, yp + offset);
  }
}

// One 64-lane workgroup per instance. Rollout rows are start-major over augmentation-major over instances
// (the policy batchifies the AUGMENTED batch: row = (s * A + a) * B + b; utils/ops.py:10-30 applied twice).
// Phase 1: lane a scans its S starts (first maximum wins, torch.max's tie rule); phase 2: lane 0 scans the A
// augmentations; phase 3: the wave copies the selected action rows (coalesced 8-byte lanes).
__global__ void __launch_bounds__(64) pomo_best_kernel(const float* __restrict__ reward, const int64_t* __restrict__ actions, int A,
                                                       int S, int B, int T, float* __restrict__ max_reward,
                                                       int64_t* __restrict__ best_start, float* __restrict__ max_aug_reward,
                                                       int64_t* __restrict__ best_aug, int64_t* __restrict__ best_ms_actions,
                                                       int64_t* __restrict__ best_aug_actions) {
  extern __shared__ int lds_i[];
  int* sel = lds_i;                                  // [A] best start of every augmentation
  float* val = reinterpret_cast<float*>(lds_i + A);  // [A] its reward
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int a = lane; a < A; a += 64) {
    float best = reward[(int64_t)a * B + b];
    int bi = 0;
    for (int s = 1; s < S; ++s) {
      const float v = reward[((int64_t)s * A + a) * B + b];
      if (v > best || (v != v && best == best)) {  // torch.max: a NaN is maximal, the first one wins
        best = v;
        bi = s;
      }
    }
    sel[a] = bi;
    val[a] = best;
    if (max_reward) max_reward[(int64_t)b * A + a] = best;
    if (best_start) best_start[(int64_t)b * A + a] = bi;
  }
  __syncthreads();
  int ba = 0;
  float bv = val[0];
  for (int a = 1; a < A; ++a) {
    if (val[a] > bv || (val[a] != val[a] && bv == bv)) {
      bv = val[a];
      ba = a;
    }
  }
  if (lane == 0) {
    if (max_aug_reward) max_aug_reward[b] = bv;
    if (best_aug) best_aug[b] = ba;
  }
  if (actions == nullptr) return;
  if (best_ms_actions) {
    for (int a = 0; a < A; ++a) {
      const int64_t* src = actions + (((int64_t)sel[a] * A + a) * B + b) * T;
      int64_t* dst = best_ms_actions + ((int64_t)b * A + a) * T;
      for (int t = lane; t < T; t += 64) dst[t] = src[t];
    }
  }
  if (best_aug_actions) {
    const int64_t* src = actions + (((int64_t)sel[ba] * A + ba) * B + b) * T;
    int64_t* dst = best_aug_actions + (int64_t)b * T;
    for (int t = lane; t < T; t += 64) dst[t] = src[t];
  }
}

inline int grid_for(int64_t items) {
  int64_t blocks = (items + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

}  // namespace

extern "C" int rl4co_augment_dihedral8_f32(const float* xy, int B, int N, float* out, void* stream) {
  RL4CO_REQUIRE(xy && out && B > 0 && N > 0);
  RL4CO_REQUIRE(((reinterpret_cast<uintptr_t>(xy) | reinterpret_cast<uintptr_t>(out)) & 7) == 0);
  const int64_t pairs = (int64_t)B * N;
  hipLaunchKernelGGL(dihedral8_kernel, dim3(grid_for(pairs)), dim3(256), 0, rl4co::as_stream(stream),
                     reinterpret_cast<const float2*>(xy), pairs, reinterpret_cast<float2*>(out));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_augment_symmetric_f32(const float* xy, const float* cos_phi, const float* sin_phi, const uint8_t* swap_axes,
                                           int B, int A, int N, float offset, float* out, void* stream) {
  RL4CO_REQUIRE(xy && cos_phi && sin_phi && swap_axes && out && B > 0 && A > 0 && N > 0);
  RL4CO_REQUIRE(((reinterpret_cast<uintptr_t>(xy) | reinterpret_cast<uintptr_t>(out)) & 7) == 0);
  const int64_t total = (int64_t)A * B * N;
  hipLaunchKernelGGL(symmetric_kernel, dim3(grid_for(total)), dim3(256), 0, rl4co::as_stream(stream),
                     reinterpret_cast<const float2*>(xy), cos_phi, sin_phi, swap_axes, B, N, total, offset,
                     reinterpret_cast<float2*>(out));
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_pomo_best(const float* reward, const int64_t* actions, int A, int S, int B, int T, float* max_reward,
                               int64_t* best_start, float* max_aug_reward, int64_t* best_aug, int64_t* best_ms_actions,
                               int64_t* best_aug_actions, void* stream) {
  RL4CO_REQUIRE(reward && A > 0 && S > 0 && B > 0 && A <= 4096);
  RL4CO_REQUIRE(actions != nullptr || (best_ms_actions == nullptr && best_aug_actions == nullptr));
  RL4CO_REQUIRE(actions == nullptr || T > 0);
  hipLaunchKernelGGL(pomo_best_kernel, dim3(B), dim3(64), (size_t)A * 8, rl4co::as_stream(stream), reward, actions, A, S, B, T,
                     max_reward, best_start, max_aug_reward, best_aug, best_ms_actions, best_aug_actions);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
