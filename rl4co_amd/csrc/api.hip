// api.hip — library identification and error reporting for the C-ABI.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "common.h"

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
