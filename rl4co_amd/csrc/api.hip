// api.hip — library identification and error reporting for the C-ABI.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "common.h"
#include "rl4co_math.h"

namespace {
thread_local char g_last_error[512] = "";
}

namespace rl4co {

int record_hip_error(hipError_t e, const char* what) {
  std::snprintf(g_last_error, sizeof(g_last_error), "HIP error %d (%s) in %s", (int)e,
                hipGetErrorString(e), what);
  return RL4CO_ERR_HIP;
}

int record_arg_error(const char* what) {
  std::snprintf(g_last_error, sizeof(g_last_error), "%s", what);
  return RL4CO_ERR_ARG;
}

}  // namespace rl4co

extern "C" const char* rl4co_version(void) { return "rl4co_amd 0.1.0 (gfx950)"; }

extern "C" int rl4co_abi_version(void) { return RL4CO_ABI_VERSION; }

extern "C" const char* rl4co_last_error(void) { return g_last_error; }

// Deterministic-math probe: y[i] = f(x[i]) with f the fp32 exp / log / tanh of rl4co_math.h evaluated ON THE
// DEVICE (tests/test_math.py pins the device rounding to the host's bit for bit, and both to float64 libm).
namespace {
__global__ void __launch_bounds__(256) math_probe_kernel(int fn, const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = fn == 0 ? rl4co_expf(v) : (fn == 1 ? rl4co_logf(v) : rl4co_tanhf(v));
  }
}
}  // namespace

extern "C" int rl4co_math_probe_f32(int fn, const float* x, int64_t n, float* y, void* stream) {
  RL4CO_REQUIRE(x && y && n > 0 && fn >= 0 && fn <= 2);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(math_probe_kernel, dim3((int)blocks), dim3(256), 0, rl4co::as_stream(stream), fn, x, n, y);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

// Instance generation on the device (rl4co/envs/common/utils.py:34-62 get_sampler -> Uniform,
// tsp/generator.py:49-58, cvrp/generator.py:114-140): out[i] = low + (high - low) * u_i with u_i = k * 2^-24, k the top
// 24 bits of word (i & 3) of Philox4x32-10 block i / 4 keyed by `seed` (counter = (block lo, block hi, stream, tag)).
// mode 1 applies CVRP's demand map on top: (trunc(v) + 1) / capacity — integer demands min .. max over the capacity.
// One launch, 16-byte stores; the same words on the host: oracle_uniform_f32 (tests/test_gpu_data.py, bit for bit).
namespace {
__global__ void __launch_bounds__(256) uniform_kernel(float* __restrict__ out, int64_t n, float low, float high, uint64_t seed,
                                                      uint32_t stream_id, int mode, float capacity) {
  const int64_t nblk = (n + 3) >> 2;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (int64_t)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)b, (uint32_t)((uint64_t)b >> 32), stream_id, 0x52344347u};
    rl4co_philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u = (float)(c[i] >> 8) * 5.9604644775390625e-8f;  // k / 2^24 in [0, 1)
      float x = fmaf(high - low, u, low);
      if (mode == 1) x = (truncf(x) + 1.0f) / capacity;
      v[i] = x;
    }
    if (4 * b + 3 < n) {
      *reinterpret_cast<float4*>(out + 4 * b) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int i = 0; 4 * b + i < n; ++i) out[4 * b + i] = v[i];
    }
  }
}
}  // namespace

extern "C" int rl4co_uniform_f32(float* out, int64_t n, float low, float high, uint64_t seed, uint32_t stream_id, int mode,
                                 float capacity, void* stream) {
  RL4CO_REQUIRE(out && n > 0 && (mode == 0 || mode == 1) && (mode == 0 || capacity > 0.0f));
  RL4CO_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  int64_t blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(uniform_kernel, dim3((int)blocks), dim3(256), 0, rl4co::as_stream(stream), out, n, low, high, seed, stream_id,
                     mode, capacity);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
