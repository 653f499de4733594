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

__device__ inline void tsp_step_body(const int64_t* action,
                                                      uint8_t* mask,
                                                      int64_t* first,
                                                      int64_t* cur,
                                                      int64_t* step_i,
                                                      uint8_t* done, int B, int N,
                                                      int32_t* err, const int b, const int lane) {
  int64_t a = *action;
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

__global__ void __launch_bounds__(64) tsp_step_kernel(
    const int64_t* action,
    uint8_t* mask,
    int64_t* first,
    int64_t* cur,
    int64_t* step_i,
    uint8_t* done,
    int B,
    int N,
    int32_t* err) {
  tsp_step_body(action ? action + blockIdx.x : nullptr, mask, first, cur, step_i, done, B, N, err, (int)blockIdx.x, (int)threadIdx.x);
}

__device__ inline void cvrp_step_body(
    const int64_t* action, const float* demand,
    float* used_capacity, const float* vehicle_capacity,
    uint8_t* visited, int64_t* cur, uint8_t* mask,
    uint8_t* done, int B, int B_inst, int N, int32_t* err, const int b, const int lane) {
  const float* dem = demand + (int64_t)(b % B_inst) * (N - 1);
  uint8_t* vis = visited + (int64_t)b * N;
  uint8_t* row = mask + (int64_t)b * N;
  float used = used_capacity[b];
  int64_t c = cur[b];
  bool bad = false;
  if (action != nullptr) {
    int64_t a = *action;
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

__global__ void __launch_bounds__(64) cvrp_step_kernel(
    const int64_t* action,
    const float* demand,
    float* used_capacity,
    const float* vehicle_capacity,
    uint8_t* visited,
    int64_t* cur,
    uint8_t* mask,
    uint8_t* done,
    int B,
    int B_inst,
    int N,
    int32_t* err) {
  cvrp_step_body(action ? action + blockIdx.x : nullptr, demand, used_capacity, vehicle_capacity, visited, cur, mask, done, B, B_inst, N, err, (int)blockIdx.x, (int)threadIdx.x);
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

// ------------------------------------------------------------------------------------------------
// Orienteering problem (envs/routing/op/env.py). Distances are (a - b).norm(p=2, dim=-1) on a size-2
// dim, i.e. sqrt(fma(dy, dy, dx * dx)) — the arithmetic of the tour-length kernel.
namespace {

__device__ inline float op_dist(const float* locs, int i, int j) {
  const float dx = locs[2 * j] - locs[2 * i], dy = locs[2 * j + 1] - locs[2 * i + 1];
  return sqrtf(fmaf(dy, dy, dx * dx));
}

__global__ void op_max_length_kernel(const float* __restrict__ locs, const float* __restrict__ max_length, int B, int N,
                                     float* __restrict__ table) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * N) return;
  const int b = (int)(idx / N), j = (int)(idx % N);
  table[idx] = (max_length[b] - op_dist(locs + (int64_t)b * N * 2, j, 0)) - 1e-6f;  // op/env.py:118-122
}

__device__ inline void op_step_body(const int64_t* action, const float* locs,
                                                     const float* maxlen, float* tour_length,
                                                     uint8_t* visited, int64_t* cur,
                                                     int64_t* step_i, uint8_t* mask,
                                                     uint8_t* done, int B, int B_inst, int N, int32_t* err, const int b, const int lane) {
  const float* lc = locs + (int64_t)(b % B_inst) * N * 2;
  const float* ml = maxlen + (int64_t)(b % B_inst) * N;
  uint8_t* vis = visited + (int64_t)b * N;
  uint8_t* row = mask + (int64_t)b * N;
  float tour = tour_length[b];
  int c = (int)cur[b];
  bool bad = false;
  if (action != nullptr) {
    int64_t a = *action;
    if (a < 0 || a >= N) {
      bad = true;
      a = 0;
    }
    tour = tour + op_dist(lc, c, (int)a);  // op/env.py:71-73
    if (lane == 0) {
      const int64_t i = step_i[b];
      vis[a] = 1;                                // op/env.py:81
      done[b] = (a == 0 && i > 0) ? 1 : 0;       // op/env.py:84
      step_i[b] = i + 1;
      cur[b] = a;
      tour_length[b] = tour;
    }
    c = (int)a;
  }
  __syncthreads();  // vis[a] visible to the wave
  const bool depot_visited = vis[0] != 0;
  for (int j = lane; j < N; j += 64) {
    const bool exceeds = tour + op_dist(lc, c, j) > ml[j];  // op/env.py:142-146
    row[j] = (j == 0 || !(vis[j] != 0 || depot_visited || exceeds)) ? 1 : 0;
  }
  if (bad && lane == 0 && err) atomicOr(err, RL4CO_EBIT_INFEASIBLE);
}

__global__ void __launch_bounds__(64) op_step_kernel(
    const int64_t* action,
    const float* locs,
    const float* maxlen,
    float* tour_length,
    uint8_t* visited,
    int64_t* cur,
    int64_t* step_i,
    uint8_t* mask,
    uint8_t* done,
    int B,
    int B_inst,
    int N,
    int32_t* err) {
  op_step_body(action ? action + blockIdx.x : nullptr, locs, maxlen, tour_length, visited, cur, step_i, mask, done, B, B_inst, N, err, (int)blockIdx.x, (int)threadIdx.x);
}

// op/env.py:168-194 on the padded action buffer (trailing depot zeros are neutral): duplicates among
// the customers; closed tour length (ATen order is irrelevant for an inequality with a 1e-5 margin
// unless the tour sits within rounding of the limit: the kernel sums in visiting order) against
// max_length_table[j] + |loc_0 - loc_j| + 1e-6 + 1e-5 for every node j, as the reference's broadcast does.
__global__ void __launch_bounds__(64) op_check_kernel(const int64_t* __restrict__ actions, const float* __restrict__ locs,
                                                      const float* __restrict__ maxlen, int B_inst, int N, int T,
                                                      int32_t* __restrict__ err) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* act = actions + (int64_t)b * T;
  const float* lc = locs + (int64_t)(b % B_inst) * N * 2;
  const float* ml = maxlen + (int64_t)(b % B_inst) * N;
  __shared__ int seen[1024];
  bool bad = false;
  for (int j = lane; j < N; j += 64) seen[j] = 0;
  __syncthreads();
  for (int t = lane; t < T; t += 64) {
    const int64_t a = act[t];
    if (a < 0 || a >= N) bad = true;
    else if (a != 0 && atomicAdd(&seen[a], 1) != 0) bad = true;
  }
  float length = 0.0f;
  if (lane == 0) {
    for (int t = 0; t < T; ++t) {
      const int p0 = (int)act[t], p1 = (int)act[t + 1 == T ? 0 : t + 1];
      if (p0 >= 0 && p0 < N && p1 >= 0 && p1 < N) length += op_dist(lc, p0, p1);
    }
  }
  length = __shfl(length, 0, 64);
  bool over = false;
  for (int j = lane; j < N; j += 64) over |= !(length <= ((ml[j] + op_dist(lc, j, 0)) + 1e-6f) + 1e-5f);
  if (__any(bad) && lane == 0) atomicOr(err, RL4CO_EBIT_DUPLICATES);
  if (__any(over) && lane == 0) atomicOr(err, RL4CO_EBIT_MAX_LENGTH);
}

// ---- CVRP with time windows (envs/routing/cvrptw/env.py:83-113) -----------------------------------
__device__ inline void cvrptw_step_body(const int64_t* action, const float* demand,
                                                         const float* locs, const float* tw,
                                                         const float* dur, float* used_capacity,
                                                         const float* vehicle_capacity,
                                                         float* current_time, uint8_t* visited,
                                                         int64_t* cur, uint8_t* mask,
                                                         uint8_t* done, int B_inst, int N, int32_t* err, const int b, const int lane) {
  const int ib = b % B_inst;
  const float* dem = demand + (int64_t)ib * (N - 1);
  const float* lc = locs + (int64_t)ib * N * 2;
  const float* w = tw + (int64_t)ib * N * 2;
  const float* du = dur + (int64_t)ib * N;
  uint8_t* vis = visited + (int64_t)b * N;
  uint8_t* row = mask + (int64_t)b * N;
  float used = used_capacity[b], now = current_time[b];
  int c = (int)cur[b];
  bool bad = false;
  if (action != nullptr) {
    int64_t a = *action;
    if (a < 0 || a >= N) {
      bad = true;
      a = 0;
    }
    now = (a != 0 ? 1.0f : 0.0f) * (fmaxf(now + op_dist(lc, c, (int)a), w[2 * a]) + du[a]);  // cvrptw/env.py:108-110
    int64_t di = a - 1;
    if (di < 0) di = 0;
    if (di > N - 2) di = N - 2;
    used = (used + dem[di]) * (a != 0 ? 1.0f : 0.0f);
    c = (int)a;
  }
  const float thr = vehicle_capacity[b] + 1e-5f;
  bool any_feasible = false, all_visited = true;
  for (int j = lane; j < N; j += 64) {
    uint8_t v = vis[j];
    if (action != nullptr && j == c) {
      v = 1;
      vis[j] = 1;
    }
    all_visited &= v != 0;
    if (j >= 1) {
      const bool masked = (v != 0) || (dem[j - 1] + used > thr);
      any_feasible |= !masked;  // the depot rule looks at capacity and visits only (CVRPEnv.get_action_mask)
      row[j] = (!masked && (now + op_dist(lc, c, j) <= w[2 * j + 1])) ? 1 : 0;  // cvrptw/env.py:91-95
    }
  }
  any_feasible = __any(any_feasible);
  all_visited = __all(all_visited);
  if (lane == 0) {
    row[0] = (!((c == 0) && any_feasible) && (now + op_dist(lc, c, 0) <= w[1])) ? 1 : 0;
    if (action != nullptr) {
      used_capacity[b] = used;
      current_time[b] = now;
      cur[b] = c;
      done[b] = all_visited ? 1 : 0;
    }
    if (bad && err) atomicOr(err, RL4CO_EBIT_INFEASIBLE);
  }
}

__global__ void __launch_bounds__(64) cvrptw_step_kernel(
    const int64_t* action,
    const float* demand,
    const float* locs,
    const float* tw,
    const float* dur,
    float* used_capacity,
    const float* vehicle_capacity,
    float* current_time,
    uint8_t* visited,
    int64_t* cur,
    uint8_t* mask,
    uint8_t* done,
    int B_inst,
    int N,
    int32_t* err) {
  cvrptw_step_body(action ? action + blockIdx.x : nullptr, demand, locs, tw, dur, used_capacity, vehicle_capacity, current_time, visited, cur, mask, done, B_inst, N, err, (int)blockIdx.x, (int)threadIdx.x);
}

// cvrptw/env.py:146-190 without its CVRP part (rl4co_cvrp_check_solution): instance-data assertions and the
// deadline replay along the tour (trailing depot zeros are neutral: t stays 0 at the depot)
__global__ void __launch_bounds__(64) cvrptw_check_kernel(const int64_t* __restrict__ actions, const float* __restrict__ locs,
                                                          const float* __restrict__ tw, const float* __restrict__ dur,
                                                          int B_inst, int N, int T, int32_t* __restrict__ err) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int ib = b % B_inst;
  const float* lc = locs + (int64_t)ib * N * 2;
  const float* w = tw + (int64_t)ib * N * 2;
  const float* du = dur + (int64_t)ib * N;
  const float max_time = tw[1];  // time_windows[..., 0, 1][0]: the depot's closing time of the FIRST instance
  bool neg = false, ret = false, dneg = false, empty = false;
  for (int j = lane; j < N; j += 64) {
    neg |= !(w[2 * j] >= 0.0f) || !(w[2 * j + 1] >= 0.0f);
    ret |= !((w[2 * j] + op_dist(lc, j, 0)) + du[j] <= max_time);
    dneg |= !(du[j] >= 0.0f);
    empty |= !(w[2 * j] < w[2 * j + 1]);
  }
  int bits = 0;
  if (__any(neg)) bits |= RL4CO_EBIT_TW_NEGATIVE;
  if (__any(ret)) bits |= RL4CO_EBIT_TW_RETURN;
  if (__any(dneg)) bits |= RL4CO_EBIT_TW_DURATION;
  if (__any(empty)) bits |= RL4CO_EBIT_TW_EMPTY;
  if (lane == 0) {
    const int64_t* act = actions + (int64_t)b * T;
    float t = 0.0f;
    int c = 0;
    bool late = false;
    for (int i = 0; i < T; ++i) {
      int64_t nx = act[i];
      if (nx < 0 || nx >= N) nx = 0;  // out of range is the CVRP check's finding
      t = fmaxf(truncf(t + op_dist(lc, c, (int)nx)), w[2 * nx]);  // max((t + d).int(), tw_start)
      late |= !(t <= w[2 * nx + 1]);
      t = t + du[nx];
      c = (int)nx;
      if (c == 0) t = 0.0f;
    }
    if (late) bits |= RL4CO_EBIT_TW_DEADLINE;
    if (bits) atomicOr(err, bits);
  }
}

// ---- prize-collecting TSP (envs/routing/pctsp/env.py:62-91,141-148) ---------------------------
__device__ inline void pctsp_step_body(const int64_t* action,
                                                        const float* real_prize,
                                                        float* total_prize, uint8_t* visited,
                                                        int64_t* cur, int64_t* step_i,
                                                        uint8_t* mask, uint8_t* done, int B_inst,
                                                        int N, int32_t* err, const int b, const int lane) {
  const float* rp = real_prize + (int64_t)(b % B_inst) * N;
  uint8_t* vis = visited + (int64_t)b * N;
  uint8_t* row = mask + (int64_t)b * N;
  float prize = total_prize[b];
  bool bad = false;
  if (action != nullptr) {
    int64_t a = *action;
    if (a < 0 || a >= N) {
      bad = true;
      a = 0;
    }
    prize = prize + rp[a];  // pctsp/env.py:66
    if (lane == 0) {
      const int64_t i = step_i[b];
      vis[a] = 1;                           // pctsp/env.py:70
      done[b] = (i > 0 && a == 0) ? 1 : 0;  // pctsp/env.py:73
      step_i[b] = i + 1;
      cur[b] = a;
      total_prize[b] = prize;
    }
  }
  __syncthreads();
  const bool depot_visited = vis[0] != 0;
  bool unvisited = false;
  for (int j = lane; j < N; j += 64) {
    if (j >= 1) {
      row[j] = (vis[j] != 0 || depot_visited) ? 0 : 1;
      unvisited |= vis[j] == 0;
    }
  }
  unvisited = __any(unvisited);
  if (lane == 0) row[0] = ((prize < 1.0f) && unvisited) ? 0 : 1;  // pctsp/env.py:144-147
  if (bad && lane == 0 && err) atomicOr(err, RL4CO_EBIT_INFEASIBLE);
}

__global__ void __launch_bounds__(64) pctsp_step_kernel(
    const int64_t* action,
    const float* real_prize,
    float* total_prize,
    uint8_t* visited,
    int64_t* cur,
    int64_t* step_i,
    uint8_t* mask,
    uint8_t* done,
    int B_inst,
    int N,
    int32_t* err) {
  pctsp_step_body(action ? action + blockIdx.x : nullptr, real_prize, total_prize, visited, cur, step_i, mask, done, B_inst, N, err, (int)blockIdx.x, (int)threadIdx.x);
}

// pctsp/env.py:175-201 on the padded action buffer: duplicates among the customers; total prize
// (summed by rl4co_gather_sum_f32 in the reference's order) >= 1 - 1e-5 unless every customer is on the tour
__global__ void __launch_bounds__(64) pctsp_check_kernel(const int64_t* __restrict__ actions,
                                                         const float* __restrict__ prize_sum, int N, int T,
                                                         int32_t* __restrict__ err) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* act = actions + (int64_t)b * T;
  __shared__ int seen[1024];
  bool bad = false;
  for (int j = lane; j < N; j += 64) seen[j] = 0;
  __syncthreads();
  int customers = 0;
  for (int t = lane; t < T; t += 64) {
    const int64_t a = act[t];
    if (a < 0 || a >= N) bad = true;
    else if (a != 0) {
      customers += 1;
      if (atomicAdd(&seen[a], 1) != 0) bad = true;
    }
  }
  customers = rl4co::bfly_i_sum(customers);
  const bool short_prize = !(prize_sum[b] >= 1.0f - 1e-5f) && customers != N - 1;
  if (__any(bad) && lane == 0) atomicOr(err, RL4CO_EBIT_DUPLICATES);
  if (short_prize && lane == 0) atomicOr(err, RL4CO_EBIT_PRIZE);
}

// ---- pickup and delivery (envs/routing/pdp/env.py:64-99) --------------------------------------
__device__ inline void pdp_step_body(const int64_t* action, uint8_t* available,
                                                      uint8_t* to_deliver, int64_t* cur,
                                                      int64_t* step_i, uint8_t* mask,
                                                      uint8_t* done, int N, int32_t* err, const int b, const int lane) {
  uint8_t* av = available + (int64_t)b * N;
  uint8_t* td = to_deliver + (int64_t)b * N;
  uint8_t* row = mask + (int64_t)b * N;
  bool bad = false;
  if (action != nullptr) {
    int64_t a = *action;
    if (a < 0 || a >= N) {
      bad = true;
      a = 0;
    }
    if (lane == 0) {
      const int n = N - 1;
      av[a] = 0;                            // pdp/env.py:73
      td[((int)a + n / 2) % (n + 1)] = 1;   // pdp/env.py:70,75
      step_i[b] += 1;
      cur[b] = a;
    }
  }
  __syncthreads();
  bool left = false;
  for (int j = lane; j < N; j += 64) {
    const bool a_ = av[j] != 0;
    row[j] = (a_ && td[j] != 0) ? 1 : 0;    // pdp/env.py:79
    left |= a_;
  }
  left = __any(left);
  if (action != nullptr && lane == 0) done[b] = left ? 0 : 1;  // pdp/env.py:83
  if (bad && lane == 0 && err) atomicOr(err, RL4CO_EBIT_INFEASIBLE);
}

__global__ void __launch_bounds__(64) pdp_step_kernel(
    const int64_t* action,
    uint8_t* available,
    uint8_t* to_deliver,
    int64_t* cur,
    int64_t* step_i,
    uint8_t* mask,
    uint8_t* done,
    int N,
    int32_t* err) {
  pdp_step_body(action ? action + blockIdx.x : nullptr, available, to_deliver, cur, step_i, mask, done, N, err, (int)blockIdx.x, (int)threadIdx.x);
}

// pdp/env.py:204-223. The tour checked is 0 ++ actions unless force_start_at_depot (then actions itself).
__global__ void __launch_bounds__(64) pdp_check_kernel(const int64_t* __restrict__ actions, int N, int T, int force,
                                                       int32_t* __restrict__ err) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* act = actions + (int64_t)b * T;
  __shared__ int when[1024];  // position of node j in the checked tour, -1 = absent
  const int off = force ? 0 : 1;
  const int L = T + off;
  for (int j = lane; j < N; j += 64) when[j] = -1;
  __syncthreads();
  bool dup = L != N, mid = false;
  if (!force && lane == 0) when[0] = 0;
  __syncthreads();
  for (int t = lane; t < T; t += 64) {
    const int64_t a = act[t];
    if (a < 0 || a >= N) dup = true;
    else {
      if (atomicExch(&when[a], t + off) != -1) dup = true;
      const int p = t + off;  // position in the checked tour
      if (a == 0 && p >= 1 && p <= L - 2) mid = true;
    }
  }
  __syncthreads();
  bool order = false;
  const int half = (N - 1) / 2;
  for (int j = 1 + lane; j <= half; j += 64) {
    if (when[j] < 0 || when[j + half] < 0) dup = true;
    else if (!(when[j] < when[j + half])) order = true;
  }
  if (lane == 0 && when[0] < 0) dup = true;
  dup = __any(dup);
  if (dup && lane == 0) atomicOr(err, RL4CO_EBIT_NOT_ALL_NODES);
  // the reference asserts in sequence: later conditions are only reached when the earlier ones hold
  if (!dup && __any(mid) && lane == 0) atomicOr(err, RL4CO_EBIT_DEPOT_MIDDLE);
  if (!dup && __any(order) && lane == 0) atomicOr(err, RL4CO_EBIT_NO_PICKUP);
}

// ---- T transitions in ONE launch, with what the decoder saw before each of them ----------------------
// The dense re-evaluation of given trajectories (policy.evaluate_log_probs: the training gradient beyond the backward
// kernels' node limit, `evaluate` decoding with autograd, PPO) needs per step the mask, the context node(s) and the context
// scalar(s) — ppo.py:128-170, decoding.py:448-461. Stepping the state with the kernels above costs T launches of 64-thread
// workgroups per call (CVRP-500 x 64: 658 steps, 20 ms of launches for 2 ms of work); this is the same loop on the device:
// one wave per trajectory runs the SAME step bodies T times over the same state arrays and tabulates between them.
__global__ void __launch_bounds__(64) env_replay_kernel(const rl4co_env_replay_args a) {
  // The trajectory's state lives in LDS for the T steps (a step is a chain of dependent reads and writes of its mask /
  // visited rows and scalars: through L2 that chain was 2.5 - 4 us per step) and goes back to the caller's arrays at the end;
  // the step bodies are called on the LDS copies with row index 0 and this trajectory's instance data.
  extern __shared__ __align__(16) unsigned char rsm[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = a.N, T = a.T;
  const int Np = (N + 15) & ~15;
  uint8_t* lmask = rsm;
  uint8_t* lvis = lmask + Np;
  uint8_t* ltod = lvis + Np;
  int64_t* l64 = reinterpret_cast<int64_t*>(ltod + Np);  // current_node, first_node, step_i
  float* lf = reinterpret_cast<float*>(l64 + 3);         // scalar, current_time
  uint8_t* ldone = reinterpret_cast<uint8_t*>(lf + 2);
  float* linst = reinterpret_cast<float*>(ldone + 16);  // the instance's data: demand / prize [N], locs [2 N], max_length [N], windows [2 N], durations [N]
  const int ib = b % a.B_inst;
  for (int j = lane; j < N; j += 64) {
    lmask[j] = a.action_mask[(int64_t)b * N + j];
    lvis[j] = a.visited ? a.visited[(int64_t)b * N + j] : 0;
    ltod[j] = a.to_deliver ? a.to_deliver[(int64_t)b * N + j] : 0;
  }
  if (lane == 0) {
    l64[0] = a.current_node[b];
    l64[1] = a.first_node ? a.first_node[b] : 0;
    l64[2] = a.step_i ? a.step_i[b] : 0;
    lf[0] = a.scalar ? a.scalar[b] : 0.0f;
    lf[1] = a.current_time ? a.current_time[b] : 0.0f;
    ldone[0] = a.done[b];
  }
  // (every step reads them: from L2 that was one more round trip on the step's chain)
  const int nd = a.env == RL4CO_ENV_PCTSP ? N : N - 1;
  float* ldem = linst;
  float* llocs = ldem + N;
  float* lmaxlen = llocs + 2 * N;
  float* ltw = lmaxlen + N;
  float* ldur = ltw + 2 * N;
  for (int j = lane; j < 2 * N; j += 64) {
    if (a.demand && j < nd) ldem[j] = a.demand[(int64_t)ib * nd + j];
    if (a.locs) llocs[j] = a.locs[(int64_t)ib * N * 2 + j];
    if (a.max_length && j < N) lmaxlen[j] = a.max_length[(int64_t)ib * N + j];
    if (a.time_windows) ltw[j] = a.time_windows[(int64_t)ib * N * 2 + j];
    if (a.durations && j < N) ldur[j] = a.durations[(int64_t)ib * N + j];
  }
  __syncthreads();
  const float* dem = a.demand ? ldem : nullptr;
  const float* locs = a.locs ? llocs : nullptr;
  const float* maxlen = a.max_length ? lmaxlen : nullptr;
  const float* tw = a.time_windows ? ltw : nullptr;
  const float* dur = a.durations ? ldur : nullptr;
  const float* cap = a.vehicle_capacity ? a.vehicle_capacity + b : nullptr;
  const float base = a.rem_base ? a.rem_base[b] : 0.0f;
  for (int t = 0; t < T; ++t) {
    const int64_t o = (int64_t)b * T + t;
    uint8_t* out = a.masks + o * N;
    for (int j0 = 0; j0 < N; j0 += 64) {
      const int j = j0 + lane;
      const uint8_t v = j < N ? lmask[j] : 0;
      if (j < N) out[j] = v;
      if (a.mask_bits != nullptr) {  // the same row as bits
        const unsigned long long bal = __ballot(v != 0);
        uint32_t* wrow = a.mask_bits + o * a.mask_words + (j0 >> 5);
        if (lane == 0) wrow[0] = (uint32_t)bal;
        if (lane == 1 && (j0 >> 5) + 1 < a.mask_words) wrow[1] = (uint32_t)(bal >> 32);
      }
    }
    if (a.mask_bits != nullptr)  // words past the graph
      for (int wd = ((N + 63) >> 6) * 2 + lane; wd < a.mask_words; wd += 64) a.mask_bits[o * a.mask_words + wd] = 0u;
    if (lane == 0) {
      a.prev[o] = l64[0];
      if (a.env == RL4CO_ENV_TSP) {
        a.first[o] = l64[1];
        a.use_placeholder[o] = l64[2] < 1 ? 1 : 0;  // context.py:86-103: the placeholder until a node is chosen
      } else if (a.env != RL4CO_ENV_PDP) {
        float r = base - lf[0];                          // context.py:105-213: capacity / length / prize still to go
        if (a.env == RL4CO_ENV_PCTSP && r < 0.0f) r = 0.0f;  // clamp(min=0), context.py:195
        a.rem[o] = r;
        if (a.env == RL4CO_ENV_CVRPTW) a.now[o] = lf[1];
      }
    }
    __syncthreads();
    const int64_t* act = a.actions + o;
    switch (a.env) {
      case RL4CO_ENV_TSP:
        tsp_step_body(act, lmask, l64 + 1, l64, l64 + 2, ldone, 1, N, a.err, 0, lane);
        break;
      case RL4CO_ENV_CVRP:
        cvrp_step_body(act, dem, lf, cap, lvis, l64, lmask, ldone, 1, 1, N, a.err, 0, lane);
        break;
      case RL4CO_ENV_OP:
        op_step_body(act, locs, maxlen, lf, lvis, l64, l64 + 2, lmask, ldone, 1, 1, N, a.err, 0, lane);
        break;
      case RL4CO_ENV_CVRPTW:
        cvrptw_step_body(act, dem, locs, tw, dur, lf, cap, lf + 1, lvis, l64, lmask, ldone, 1, N, a.err, 0, lane);
        break;
      case RL4CO_ENV_PCTSP:
        pctsp_step_body(act, dem, lf, lvis, l64, l64 + 2, lmask, ldone, 1, N, a.err, 0, lane);
        break;
      default:
        pdp_step_body(act, lvis, ltod, l64, l64 + 2, lmask, ldone, N, a.err, 0, lane);
        break;
    }
    __syncthreads();  // the row and the scalars lane 0 wrote are this wave's next reads
  }
  // the state as T step calls leave it
  for (int j = lane; j < N; j += 64) {
    a.action_mask[(int64_t)b * N + j] = lmask[j];
    if (a.visited) a.visited[(int64_t)b * N + j] = lvis[j];
    if (a.to_deliver) a.to_deliver[(int64_t)b * N + j] = ltod[j];
  }
  if (lane == 0) {
    a.current_node[b] = l64[0];
    if (a.first_node) a.first_node[b] = l64[1];
    if (a.step_i) a.step_i[b] = l64[2];
    if (a.scalar) a.scalar[b] = lf[0];
    if (a.current_time) a.current_time[b] = lf[1];
    a.done[b] = ldone[0];
  }
}

}  // namespace

extern "C" int rl4co_env_replay(const rl4co_env_replay_args* args, void* stream) {
  RL4CO_REQUIRE(args != nullptr);
  const rl4co_env_replay_args& a = *args;
  RL4CO_REQUIRE(a.env == RL4CO_ENV_TSP || a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_OP || a.env == RL4CO_ENV_PCTSP ||
                a.env == RL4CO_ENV_PDP || a.env == RL4CO_ENV_CVRPTW);
  RL4CO_REQUIRE(a.B > 0 && a.B_inst > 0 && a.B % a.B_inst == 0 && a.N >= 2 && a.T >= 1);
  RL4CO_REQUIRE(a.actions && a.action_mask && a.current_node && a.done && a.masks && a.prev);
  if (a.env == RL4CO_ENV_TSP) {
    RL4CO_REQUIRE(a.first_node && a.step_i && a.first && a.use_placeholder);
  } else if (a.env == RL4CO_ENV_PDP) {
    RL4CO_REQUIRE(a.visited && a.to_deliver && a.step_i && a.N >= 3 && (a.N - 1) % 2 == 0);
  } else {
    RL4CO_REQUIRE(a.visited && a.scalar && a.rem_base && a.rem);
    if (a.env == RL4CO_ENV_CVRP || a.env == RL4CO_ENV_CVRPTW) RL4CO_REQUIRE(a.demand && a.vehicle_capacity);
    if (a.env == RL4CO_ENV_PCTSP) RL4CO_REQUIRE(a.demand && a.step_i);
    if (a.env == RL4CO_ENV_OP) RL4CO_REQUIRE(a.locs && a.max_length && a.step_i);
    if (a.env == RL4CO_ENV_CVRPTW) RL4CO_REQUIRE(a.locs && a.time_windows && a.durations && a.current_time && a.now);
  }
  RL4CO_REQUIRE(a.mask_bits == nullptr || a.mask_words * 32 >= a.N);
  const int lds = 3 * ((a.N + 15) & ~15) + 3 * 8 + 2 * 4 + 16 + 7 * a.N * 4;
  RL4CO_REQUIRE(lds <= 160 * 1024);  // (N <= 5 300 nodes)
  if (lds > 64 * 1024)
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(env_replay_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(env_replay_kernel, dim3(a.B), dim3(64), lds, rl4co::as_stream(stream), a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_cvrptw_step(const int64_t* action, const float* demand, const float* locs, const float* time_windows,
                                 const float* durations, float* used_capacity, const float* vehicle_capacity,
                                 float* current_time, uint8_t* visited, int64_t* current_node, uint8_t* action_mask,
                                 uint8_t* done, int B, int B_inst, int N, int32_t* err, void* stream) {
  RL4CO_REQUIRE(demand && locs && time_windows && durations && used_capacity && vehicle_capacity && current_time);
  RL4CO_REQUIRE(visited && current_node && action_mask && (action == nullptr || done != nullptr));
  RL4CO_REQUIRE(B > 0 && B_inst > 0 && B % B_inst == 0 && N > 1);
  hipLaunchKernelGGL(cvrptw_step_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), action, demand, locs, time_windows,
                     durations, used_capacity, vehicle_capacity, current_time, visited, current_node, action_mask, done,
                     B_inst, N, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_cvrptw_check_solution(const int64_t* actions, const float* locs, const float* time_windows,
                                           const float* durations, int B, int B_inst, int N, int T, int32_t* err,
                                           void* stream) {
  RL4CO_REQUIRE(actions && locs && time_windows && durations && err);
  RL4CO_REQUIRE(B > 0 && B_inst > 0 && B % B_inst == 0 && N > 1 && T >= 1);
  hipLaunchKernelGGL(cvrptw_check_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), actions, locs, time_windows,
                     durations, B_inst, N, T, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_pdp_step(const int64_t* action, uint8_t* available, uint8_t* to_deliver, int64_t* current_node,
                              int64_t* step_i, uint8_t* action_mask, uint8_t* done, int B, int N, int32_t* err, void* stream) {
  RL4CO_REQUIRE(available && to_deliver && current_node && step_i && action_mask && done);
  RL4CO_REQUIRE(B > 0 && N >= 3 && (N - 1) % 2 == 0);
  hipLaunchKernelGGL(pdp_step_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), action, available, to_deliver,
                     current_node, step_i, action_mask, done, N, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_pdp_check_solution(const int64_t* actions, int B, int N, int T, int force_start_at_depot, int32_t* err,
                                        void* stream) {
  RL4CO_REQUIRE(actions && err && B > 0 && N >= 3 && N <= 1024 && (N - 1) % 2 == 0 && T >= 1);
  hipLaunchKernelGGL(pdp_check_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), actions, N, T,
                     force_start_at_depot ? 1 : 0, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_pctsp_step(const int64_t* action, const float* real_prize, float* cur_total_prize, uint8_t* visited,
                                int64_t* current_node, int64_t* step_i, uint8_t* action_mask, uint8_t* done, int B,
                                int B_inst, int N, int32_t* err, void* stream) {
  RL4CO_REQUIRE(real_prize && cur_total_prize && visited && current_node && step_i && action_mask && done);
  RL4CO_REQUIRE(B > 0 && B_inst > 0 && B % B_inst == 0 && N >= 2);
  hipLaunchKernelGGL(pctsp_step_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), action, real_prize, cur_total_prize,
                     visited, current_node, step_i, action_mask, done, B_inst, N, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_pctsp_check_solution(const int64_t* actions, const float* prize_sum, int B, int N, int T, int32_t* err,
                                          void* stream) {
  RL4CO_REQUIRE(actions && prize_sum && err && B > 0 && N >= 2 && N <= 1024 && T >= 1);
  hipLaunchKernelGGL(pctsp_check_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), actions, prize_sum, N, T, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_op_max_length(const float* locs, const float* max_length, int B, int N, float* table, void* stream) {
  RL4CO_REQUIRE(locs && max_length && table && B > 0 && N >= 2);
  const int64_t total = (int64_t)B * N;
  hipLaunchKernelGGL(op_max_length_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, rl4co::as_stream(stream), locs,
                     max_length, B, N, table);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_op_step(const int64_t* action, const float* locs, const float* max_length_table, float* tour_length,
                             uint8_t* visited, int64_t* current_node, int64_t* step_i, uint8_t* action_mask, uint8_t* done,
                             int B, int B_inst, int N, int32_t* err, void* stream) {
  RL4CO_REQUIRE(locs && max_length_table && tour_length && visited && current_node && step_i && action_mask && done);
  RL4CO_REQUIRE(B > 0 && B_inst > 0 && B % B_inst == 0 && N >= 2);
  hipLaunchKernelGGL(op_step_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), action, locs, max_length_table, tour_length,
                     visited, current_node, step_i, action_mask, done, B, B_inst, N, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_op_check_solution(const int64_t* actions, const float* locs, const float* max_length_table, int B,
                                       int B_inst, int N, int T, int32_t* err, void* stream) {
  RL4CO_REQUIRE(actions && locs && max_length_table && err && B > 0 && B_inst > 0 && N >= 2 && N <= 1024 && T >= 1);
  hipLaunchKernelGGL(op_check_kernel, dim3(B), dim3(64), 0, rl4co::as_stream(stream), actions, locs, max_length_table, B_inst, N,
                     T, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

