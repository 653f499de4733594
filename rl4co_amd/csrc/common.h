// common.h — shared host/device helpers for the rl4co_amd HIP library.
#ifndef RL4CO_COMMON_H
#define RL4CO_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rl4co_amd.h"

namespace rl4co {

// Records the failing HIP call for rl4co_last_error(); returns RL4CO_ERR_HIP.
int record_hip_error(hipError_t e, const char* what);
int record_arg_error(const char* what);

#define RL4CO_HIP_TRY(expr)                                        \
  do {                                                             \
    hipError_t _e = (expr);                                        \
    if (_e != hipSuccess) return rl4co::record_hip_error(_e, #expr); \
  } while (0)

#define RL4CO_REQUIRE(cond)                                          \
  do {                                                               \
    if (!(cond)) return rl4co::record_arg_error("requirement failed: " #cond); \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Wave-wide (64-lane) butterfly helpers. On gfx950 a workgroup of 64 threads is
// exactly one wavefront, so these are the only cross-lane primitives the decode
// kernel needs.
__device__ inline float shfl_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ inline int shfl_xor_i(int v, int m) { return __shfl_xor(v, m, 64); }

}  // namespace rl4co

#endif
