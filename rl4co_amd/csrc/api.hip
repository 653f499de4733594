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
