// env_step.hip — stand-alone TSP / CVRP environment transitions (the RL4COEnvBase.step
// surface, used when a caller drives the loop step by step instead of through the
// fused rollout kernel in am_decode.hip, which contains the same transitions inline).
//
//   TSPEnv._step              envs/routing/tsp/env.py:60-86
//   CVRPEnv._step             envs/routing/cvrp/env.py:66-96
//   CVRPEnv.get_action_mask   envs/routing/cvrp/env.py:126-136
// Byte/index work + one fp32 add, one mul, one compare per node: bit-exact by
// construction. One wavefront per trajectory, mask rows read/written coalesced.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

__global__ void __launch_bounds__(64) tsp_step_kernel(const int64_t* __restrict__ action,
                                                      uint8_t* __restrict__ mask,
                                                      int64_t* __restrict__ first,
                                                      int64_t* __restrict__ cur,
                                                      int64_t* __restrict__ step_i,
                                                      uint8_t* __restrict__ done, int B, int N,
                                                      int32_t* err) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int64_t a = action[b];
  bool bad = false;
  if (a < 0 || a >= N) {
    bad = true;
    a = 0;
  }
  uint8_t* row = mask + (int64_t)b * N;
  bool any_left = false;
  for (int j = lane; j < N; j += 64) {
    uint8_t v = row[j];
    if (j == a) {
      v = 0;  // scatter(-1, action, 0)
      row[j] = 0;
    }
    any_left |= v != 0;
  }
  any_left = __any(any_left);
  if (lane == 0) {
    const int64_t i = step_i[b];
    if (i == 0) first[b] = a;  // td["i"].all() == 0 : rows are in lock-step
    cur[b] = a;
    step_i[b] = i + 1;
    done[b] = any_left ? 0 : 1;
    if (bad && err) atomicOr(err, RL4CO_EBIT_INFEASIBLE);
  }
}

__global__ void __launch_bounds__(64) cvrp_step_kernel(
    const int64_t* __restrict__ action, const float* __restrict__ demand,
    float* __restrict__ used_capacity, const float* __restrict__ vehicle_capacity,
    uint8_t* __restrict__ visited, int64_t* __restrict__ cur, uint8_t* __restrict__ mask,
    uint8_t* __restrict__ done, int B, int B_inst, int N, int32_t* err) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const float* dem = demand + (int64_t)(b % B_inst) * (N - 1);
  uint8_t* vis = visited + (int64_t)b * N;
  uint8_t* row = mask + (int64_t)b * N;
  float used = used_capacity[b];
  int64_t c = cur[b];
  bool bad = false;
  if (action != nullptr) {
    int64_t a = action[b];
    if (a < 0 || a >= N) {
      bad = true;
      a = 0;
    }
    int64_t di = a - 1;  // clamp(a-1, 0, n_loc-1)
    if (di < 0) di = 0;
    if (di > N - 2) di = N - 2;
    used = (used + dem[di]) * (a != 0 ? 1.0f : 0.0f);
    c = a;
  }
  const float thr = vehicle_capacity[b] + 1e-5f;
  bool any_feasible = false, all_visited = true;
  for (int j = lane; j < N; j += 64) {
    uint8_t v = vis[j];
    if (action != nullptr && j == c) {
      v = 1;  // scatter(-1, current_node, 1)
      vis[j] = 1;
    }
    all_visited &= v != 0;
    if (j >= 1) {
      const bool masked = (v != 0) || (dem[j - 1] + used > thr);
      row[j] = masked ? 0 : 1;
      any_feasible |= !masked;
    }
  }
  any_feasible = __any(any_feasible);
  all_visited = __all(all_visited);
  if (lane == 0) {
    row[0] = ((c == 0) && any_feasible) ? 0 : 1;
    if (action != nullptr) {
      used_capacity[b] = used;
      cur[b] = c;
      done[b] = all_visited ? 1 : 0;
    }
    if (bad && err) atomicOr(err, RL4CO_EBIT_INFEASIBLE);
  }
}

}  // namespace

extern "C" int rl4co_tsp_step(const int64_t* action, uint8_t* action_mask, int64_t* first_node,
                              int64_t* current_node, int64_t* step_i, uint8_t* done, int B, int N,
                              int32_t* err, void* stream) {
  RL4CO_REQUIRE(action && action_mask && first_node && current_node && step_i && done);
  RL4CO_REQUIRE(B > 0 && N > 0);
  hipLaunchKernelGGL(tsp_step_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), action,
                     action_mask, first_node, current_node, step_i, done, B, N, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_cvrp_step(const int64_t* action, const float* demand, float* used_capacity,
                               const float* vehicle_capacity, uint8_t* visited,
                               int64_t* current_node, uint8_t* action_mask, uint8_t* done, int B,
                               int B_inst, int N, int32_t* err, void* stream) {
  RL4CO_REQUIRE(demand && used_capacity && vehicle_capacity && visited && current_node &&
                action_mask);
  RL4CO_REQUIRE(action == nullptr || done != nullptr);
  RL4CO_REQUIRE(B > 0 && B_inst > 0 && B % B_inst == 0 && N > 1);
  hipLaunchKernelGGL(cvrp_step_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), action,
                     demand, used_capacity, vehicle_capacity, visited, current_node, action_mask,
                     done, B, B_inst, N, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
