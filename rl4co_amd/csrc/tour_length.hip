// tour_length.hip — reward kernels: gather_by_index, get_tour_length, check_solution_validity.
//
// get_tour_length (rl4co/utils/ops.py:82-90) is restated so that the result is
// BIT-IDENTICAL to the reference's ATen CPU arithmetic:
//   segment:  d = p[t+1] - p[t] ; len = sqrt(fma(d.y, d.y, fl(d.x*d.x)))
//             (ATen vector_norm over a size-2 last dim)
//   row sum:  aten/src/ATen/native/cpu/SumKernel.cpp vectorized_inner_sum with
//             Vec = 8 fp32 lanes, row_sum ILP = 4, multi_row_sum 4-level cascade.
// Eight GPU lanes play the eight SIMD lanes of one CPU vector register, so one
// wavefront reduces eight trajectories; the final lane-order scalar fold is done
// with shuffles in the same l = 0..7 order as the CPU's store-and-add loop.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

// Segment t of the closed tour over n points, for trajectory `b`.
struct TourView {
  const float* locs;       // [N,2] of this trajectory's instance
  const int64_t* actions;  // [T]
  int prepend;             // 1: point 0 is the depot locs[0]
  int n;                   // number of tour points = T + prepend
  const float* values;     // non-null: element t is values[actions[t]] (OP reward: gathered prizes)
  __device__ inline float2 point(int t) const {
    int node;
    if (prepend) node = (t == 0) ? 0 : (int)actions[t - 1];
    else node = (int)actions[t];
    return *reinterpret_cast<const float2*>(locs + 2 * (int64_t)node);
  }
  __device__ inline float seg(int t) const {
    if (values) return values[actions[t]];
    const float2 p0 = point(t);
    const float2 p1 = point(t + 1 == n ? 0 : t + 1);  // torch.roll(-1)
    const float dx = p1.x - p0.x;
    const float dy = p1.y - p0.y;
    return sqrtf(fmaf(dy, dy, dx * dx));
  }
};

// multi_row_sum<acc_t=float(lane of a Vec), nrows=4> restated for ONE vector lane.
// `vecs` = number of 8-wide vectors in the row; this lane reads element 8*v + lane.
__device__ float lane_row_sum(const TourView& tv, int lane8, int nvec) {
  constexpr int kIlp = 4;
  constexpr int kLevels = 4;
  const int size = nvec / kIlp;  // size_ilp
  // level_power = max(4, ceil_log2(size) / num_levels)
  int ceil_log2 = 0;
  while ((1LL << ceil_log2) < size) ++ceil_log2;
  int level_power = ceil_log2 / kLevels;
  if (level_power < 4) level_power = 4;
  const int level_step = 1 << level_power;
  const int level_mask = level_step - 1;
  float acc[kLevels][kIlp];
  for (int l = 0; l < kLevels; ++l)
    for (int k = 0; k < kIlp; ++k) acc[l][k] = 0.0f;
  int i = 0;
  for (; i + level_step <= size;) {
    for (int j = 0; j < level_step; ++j, ++i) {
      for (int k = 0; k < kIlp; ++k) acc[0][k] = acc[0][k] + tv.seg(8 * (i * kIlp + k) + lane8);
    }
    for (int j = 1; j < kLevels; ++j) {
      for (int k = 0; k < kIlp; ++k) {
        acc[j][k] = acc[j][k] + acc[j - 1][k];
        acc[j - 1][k] = 0.0f;
      }
      const int mask = level_mask << (j * level_power);
      if ((i & mask) != 0) break;
    }
  }
  for (; i < size; ++i) {
    for (int k = 0; k < kIlp; ++k) acc[0][k] = acc[0][k] + tv.seg(8 * (i * kIlp + k) + lane8);
  }
  for (int j = 1; j < kLevels; ++j)
    for (int k = 0; k < kIlp; ++k) acc[0][k] = acc[0][k] + acc[j][k];
  // row_sum tail: leftover vectors go to partial 0, then fold the ILP partials
  for (int v = size * kIlp; v < nvec; ++v) acc[0][0] = acc[0][0] + tv.seg(8 * v + lane8);
  for (int k = 1; k < kIlp; ++k) acc[0][0] = acc[0][0] + acc[0][k];
  return acc[0][0];
}

__global__ void __launch_bounds__(256) tour_length_kernel(const float* __restrict__ locs,
                                                          const int64_t* __restrict__ actions,
                                                          int B, int B_locs, int N, int T,
                                                          int prepend, int negate,
                                                          float* __restrict__ out,
                                                          const float* __restrict__ gather_values,
                                                          int row_stride, const int32_t* __restrict__ t_dev, int t_add) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 3;
  const int lane8 = gid & 7;
  const bool active = b < B;
  // device-side horizon (rl4co_tour_length_dyn_f32): the tour length of a rollout whose step count the HOST does not
  // know yet — T = t_add + *t_dev, what the decode launch left in its steps summary — so the reward can be issued
  // before the one read-back of the rollout (and inside a captured HIP graph). The association of the row sum depends
  // on T, so it has to be the real one: summing the padded buffer is not the reference's arithmetic.
  if (t_dev) T = min(row_stride, t_add + *t_dev);
  TourView tv;
  tv.prepend = prepend;
  tv.n = T + prepend;
  tv.locs = locs ? locs + (int64_t)(active ? b % B_locs : 0) * N * 2 : nullptr;
  tv.actions = actions + (int64_t)(active ? b : 0) * row_stride;
  // OPEnv._get_reward (op/env.py:156-166): the same inner-dim sum over prize.gather(1, actions)
  tv.values = gather_values ? gather_values + (int64_t)(active ? b % B_locs : 0) * N : nullptr;
  const int n = tv.n;
  const int nvec = n / 8;
  if (n < 8) {
    // below one 8-wide vector ATen takes scalar_inner_sum: row_sum<float> with 4 ILP
    // partials, leftovers into partial 0, partials folded in order
    if (active && lane8 == 0) {
      float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      const int g = n / 4;
      for (int i = 0; i < g; ++i)
        for (int k = 0; k < 4; ++k) p[k] = p[k] + tv.seg(4 * i + k);
      for (int k = 4 * g; k < n; ++k) p[0] = p[0] + tv.seg(k);
      for (int k = 1; k < 4; ++k) p[0] = p[0] + p[k];
      out[b] = negate ? -p[0] : p[0];
    }
    return;
  }
  float part = 0.0f;
  if (active) part = lane_row_sum(tv, lane8, nvec);
  // vectorized_inner_sum epilogue: scalar tail first, then the 8 lane partials in order.
  float fin = 0.0f;
  if (active && lane8 == 0) {
    for (int k = nvec * 8; k < n; ++k) fin = fin + tv.seg(k);
  }
  const int base = (threadIdx.x & 63) & ~7;
  for (int l = 0; l < 8; ++l) {
    const float pl = __shfl(part, base + l, 64);
    fin = fin + pl;
  }
  if (active && lane8 == 0) out[b] = negate ? -fin : fin;
}

__global__ void gather_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, int B,
                              int N, int D, int K, float* __restrict__ out, int32_t* err) {
  const int64_t total = (int64_t)B * K * D;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(g % D);
    const int64_t bk = g / D;
    const int b = (int)(bk / K);
    int64_t n = idx[bk];
    if (n < 0 || n >= N) {
      if (err) atomicOr(err, RL4CO_EBIT_INVALID_TOUR);
      n = 0;
    }
    out[g] = src[((int64_t)b * N + n) * D + d];
  }
}

// TSP: each row of actions must be a permutation of 0..N-1 (tsp/env.py:158-164).
// One wave per row; a per-wave LDS bitmap counts visits.
__global__ void __launch_bounds__(64) tsp_check_kernel(const int64_t* __restrict__ actions, int B,
                                                       int N, int T, int32_t* err) {
  extern __shared__ unsigned int seen[];  // ceil(N/32) words
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int words = (N + 31) / 32;
  for (int w = lane; w < words; w += 64) seen[w] = 0;
  __syncthreads();
  bool bad = (T != N);
  for (int t = lane; t < T; t += 64) {
    const int64_t a = actions[(int64_t)b * T + t];
    if (a < 0 || a >= N) {
      bad = true;
    } else {
      const unsigned int bit = 1u << (a & 31);
      const unsigned int old = atomicOr(&seen[a >> 5], bit);
      if (old & bit) bad = true;
    }
  }
  if (__any(bad) && lane == 0) atomicOr(err, RL4CO_EBIT_INVALID_TOUR);
}

// CVRP (cvrp/env.py:149-177): customers exactly once, the rest depot; running load
// (d = -capacity at the depot, clamped at 0) never above capacity + 1e-5.
__global__ void __launch_bounds__(64) cvrp_check_kernel(const int64_t* __restrict__ actions,
                                                        const float* __restrict__ demand,
                                                        const float* __restrict__ capacity, int B,
                                                        int B_inst, int N, int T, int32_t* err) {
  extern __shared__ unsigned int seen[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int words = (N + 31) / 32;
  for (int w = lane; w < words; w += 64) seen[w] = 0;
  __syncthreads();
  bool bad = false;
  int customers = 0;
  for (int t = lane; t < T; t += 64) {
    const int64_t a = actions[(int64_t)b * T + t];
    if (a < 0 || a >= N) {
      bad = true;
    } else if (a > 0) {
      const unsigned int bit = 1u << (a & 31);
      const unsigned int old = atomicOr(&seen[a >> 5], bit);
      if (old & bit) bad = true;
      ++customers;
    }
  }
  for (int s = 1; s < 64; s <<= 1) customers += __shfl_xor(customers, s, 64);
  if (customers != N - 1) bad = true;
  if (__any(bad) && lane == 0) atomicOr(err, RL4CO_EBIT_INVALID_TOUR);
  if (lane == 0) {
    const int inst = b % B_inst;
    const float cap = capacity[inst];
    const float thr = cap + 1e-5f;
    float used = 0.0f;
    bool over = false;
    for (int t = 0; t < T; ++t) {
      int64_t a = actions[(int64_t)b * T + t];
      if (a < 0 || a >= N) a = 0;
      const float d = (a == 0) ? -cap : demand[(int64_t)inst * (N - 1) + (a - 1)];
      used = used + d;
      if (used < 0.0f) used = 0.0f;
      if (!(used <= thr)) over = true;
    }
    if (over) atomicOr(err, RL4CO_EBIT_CAPACITY);
  }
}

__global__ void select_start_nodes_kernel(int64_t* out, int B, int S, int num_loc, int has_depot) {
  const int64_t total = (int64_t)B * S;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = g / B;  // arange(S).repeat_interleave(B)
    out[g] = s % num_loc + (has_depot ? 1 : 0);
  }
}

__global__ void __launch_bounds__(256) hbm_read_probe_kernel(const float4* __restrict__ src,
                                                             int64_t n16, float* sink) {
  float acc = 0.0f;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n16;
       g += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = src[g];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;  // never true in practice; keeps the loads alive
}

}  // namespace

extern "C" int rl4co_tour_length_f32(const float* locs, const int64_t* actions, int B, int B_locs,
                                     int N, int T, int prepend_depot, int negate, float* out,
                                     void* stream) {
  RL4CO_REQUIRE(locs && actions && out);
  RL4CO_REQUIRE(B > 0 && B_locs > 0 && B % B_locs == 0 && N > 0 && T > 0);
  const int threads = 256;
  const int64_t total = (int64_t)B * 8;
  const int blocks = (int)((total + threads - 1) / threads);
  hipLaunchKernelGGL(tour_length_kernel, dim3(blocks), dim3(threads), 0, rl4co::as_stream(stream),
                     locs, actions, B, B_locs, N, T, prepend_depot ? 1 : 0, negate ? 1 : 0, out,
                     static_cast<const float*>(nullptr), T, static_cast<const int32_t*>(nullptr), 0);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_tour_length_dyn_f32(const float* locs, const int64_t* actions, int B, int B_locs, int N, int row_stride,
                                         const int32_t* steps_dev, int t_add, int prepend_depot, int negate, float* out,
                                         void* stream) {
  RL4CO_REQUIRE(locs && actions && out && steps_dev);
  RL4CO_REQUIRE(B > 0 && B_locs > 0 && B % B_locs == 0 && N > 0 && row_stride > 0 && t_add >= 0);
  const int threads = 256;
  const int64_t total = (int64_t)B * 8;
  const int blocks = (int)((total + threads - 1) / threads);
  hipLaunchKernelGGL(tour_length_kernel, dim3(blocks), dim3(threads), 0, rl4co::as_stream(stream), locs, actions, B, B_locs, N,
                     row_stride, prepend_depot ? 1 : 0, negate ? 1 : 0, out, static_cast<const float*>(nullptr), row_stride,
                     steps_dev, t_add);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_gather_sum_f32(const float* values, const int64_t* actions, int B, int B_values, int N, int T, float* out,
                                    void* stream) {
  RL4CO_REQUIRE(values && actions && out);
  RL4CO_REQUIRE(B > 0 && B_values > 0 && B % B_values == 0 && N > 0 && T > 0);
  const int threads = 256;
  const int64_t total = (int64_t)B * 8;
  const int blocks = (int)((total + threads - 1) / threads);
  hipLaunchKernelGGL(tour_length_kernel, dim3(blocks), dim3(threads), 0, rl4co::as_stream(stream),
                     static_cast<const float*>(nullptr), actions, B, B_values, N, T, 0, 0, out, values, T,
                     static_cast<const int32_t*>(nullptr), 0);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_gather_by_index_f32(const float* src, const int64_t* idx, int B, int N, int D,
                                         int K, float* out, int32_t* err, void* stream) {
  RL4CO_REQUIRE(src && idx && out);
  RL4CO_REQUIRE(B > 0 && N > 0 && D > 0 && K > 0);
  const int64_t total = (int64_t)B * K * D;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, rl4co::as_stream(stream), src, idx,
                     B, N, D, K, out, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_tsp_check_solution(const int64_t* actions, int B, int N, int T, int32_t* err,
                                        void* stream) {
  RL4CO_REQUIRE(actions && err && B > 0 && N > 0 && T > 0);
  const int lds = ((N + 31) / 32) * 4;
  hipLaunchKernelGGL(tsp_check_kernel, dim3(B), dim3(64), lds, rl4co::as_stream(stream), actions, B,
                     N, T, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_cvrp_check_solution(const int64_t* actions, const float* demand,
                                         const float* vehicle_capacity, int B, int B_inst, int N,
                                         int T, int32_t* err, void* stream) {
  RL4CO_REQUIRE(actions && demand && vehicle_capacity && err);
  RL4CO_REQUIRE(B > 0 && B_inst > 0 && B % B_inst == 0 && N > 1 && T > 0);
  const int lds = ((N + 31) / 32) * 4;
  hipLaunchKernelGGL(cvrp_check_kernel, dim3(B), dim3(64), lds, rl4co::as_stream(stream), actions,
                     demand, vehicle_capacity, B, B_inst, N, T, err);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_select_start_nodes(int64_t* out, int B, int num_starts, int num_loc,
                                        int has_depot, void* stream) {
  RL4CO_REQUIRE(out && B > 0 && num_starts > 0 && num_loc > 0);
  const int64_t total = (int64_t)B * num_starts;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(select_start_nodes_kernel, dim3(blocks), dim3(256), 0,
                     rl4co::as_stream(stream), out, B, num_starts, num_loc, has_depot);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

extern "C" int rl4co_hbm_read_probe(const void* src, int64_t bytes, float* sink, void* stream) {
  RL4CO_REQUIRE(src && sink && bytes >= 16);
  hipLaunchKernelGGL(hbm_read_probe_kernel, dim3(256 * 8), dim3(256), 0, rl4co::as_stream(stream),
                     static_cast<const float4*>(src), bytes / 16, sink);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
