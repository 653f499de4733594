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

// am_decode_ms.hip: multistart decode on the matrix cores (dispatched from rl4co_am_decode)
int launch_decode_ms(const rl4co_am_decode_args& a, hipStream_t stream);
int launch_decode_ms_f16(const rl4co_am_decode_args& a, hipStream_t stream);  // fp16 planes (csrc/elem16.h)

// am_teacher_mma.hip: teacher-forced backward on the matrix cores (dispatched from rl4co_am_teacher_backward)
int launch_teacher_mma(const rl4co_am_teacher_args& a, hipStream_t stream);
int launch_teacher_mma_f16(const rl4co_am_teacher_args& a, hipStream_t stream);  // fp16 planes
int teacher_mma_max_nodes();
int teacher_mma_max_steps();

// Workgroup barrier for LDS hand-offs only. __syncthreads() waits for EVERY outstanding memory
// operation (s_waitcnt vmcnt(0) lgkmcnt(0)) before s_barrier, which puts the L2 round trip of any
// in-flight global store / atomic / prefetch in front of each barrier; kernels that meet several
// times per step on LDS data only need their LDS traffic drained.
// LDS hand-over inside ONE wavefront (lanes exchange rows only their own wave wrote): DS operations of a wave
// execute in order, so a compiler/memory fence without an s_barrier is enough.
__device__ inline void lds_barrier_wave() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// bit `pos` of `bits` set ? x : -inf, in two VALU operations (v_bfe_i32 spreads the bit, v_bitop3_b32 selects) where the
// compiler's own lowering of the conditional takes three (and, compare, cndmask)
__device__ inline float keep_or_neg_inf(uint32_t bits, int pos, float x) {
  const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)bits, pos, 1);
  return __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(m, __builtin_bit_cast(uint32_t, x), 0xff800000u, 0xCA));  // m ? x : -inf
}

__device__ inline void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Wave-wide (64-lane) butterfly partner fetch WITHOUT the LDS crossbar: ds_bpermute (what
// __shfl_xor compiles to) costs an LDS round trip (~100+ cycles) per step and the decode kernels
// run dependent chains of them; DPP modifiers and the gfx950 permlane swaps are plain VALU ops.
//   S = 1, 2  : quad_perm
//   S = 4, 8  : row_half_mirror / row_mirror — equal to the xor-4 / xor-8 partner ONLY when the
//               value is already uniform inside each 4- / 8-lane group, which holds at that
//               point of an ascending butterfly (the way every caller uses them)
//   S = 16, 32: v_permlane16_swap / v_permlane32_swap (swap odd rows / the upper half of one
//               copy with even rows / the lower half of the other)
// The value returned is bit-identical to __shfl_xor(v, S, 64) under those conditions.
template <int CTRL>
__device__ inline int dpp_mov_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

template <int S>
__device__ inline int bfly_i(int v) {
  static_assert(S == 1 || S == 2 || S == 4 || S == 8 || S == 16 || S == 32, "butterfly step");
  if constexpr (S == 1) {
    return dpp_mov_i<0xB1>(v);  // quad_perm:[1,0,3,2]
  } else if constexpr (S == 2) {
    return dpp_mov_i<0x4E>(v);  // quad_perm:[2,3,0,1]
  } else if constexpr (S == 4) {
    return dpp_mov_i<0x141>(v);  // row_half_mirror
  } else if constexpr (S == 8) {
    return dpp_mov_i<0x140>(v);  // row_mirror
  } else if constexpr (S == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (threadIdx.x & 16) ? (int)r[0] : (int)r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
  }
}

template <int S>
__device__ inline float bfly_f(float v) {
  return __builtin_bit_cast(float, bfly_i<S>(__builtin_bit_cast(int, v)));
}

// v + partner over the butterfly steps LO, 2*LO, ..., < HI (ascending), i.e. the pairwise tree
template <int LO, int HI>
__device__ inline float bfly_sum(float v) {
  if constexpr (LO < HI) {
    v = v + bfly_f<LO>(v);
    return bfly_sum<LO * 2, HI>(v);
  } else {
    return v;
  }
}

template <int LO, int HI>
__device__ inline float bfly_max(float v) {
  if constexpr (LO < HI) {
    v = fmaxf(v, bfly_f<LO>(v));
    return bfly_max<LO * 2, HI>(v);
  } else {
    return v;
  }
}

// integer maximum over the whole wave
template <int S = 1>
__device__ inline int bfly_i_max(int v) {
  if constexpr (S < 64) {
    const int o = bfly_i<S>(v);
    return bfly_i_max<S * 2>(o > v ? o : v);
  } else {
    return v;
  }
}

// integer sum over the whole wave
template <int S = 1>
__device__ inline int bfly_i_sum(int v) {
  if constexpr (S < 64) {
    return bfly_i_sum<S * 2>(v + bfly_i<S>(v));
  } else {
    return v;
  }
}

// (maximum key, lowest index on ties) over the whole wave; idx == 0x7fffffff marks "empty"
template <int S = 1>
__device__ inline void bfly_argmax(float& best, int& bi) {
  if constexpr (S < 64) {
    const float ov = bfly_f<S>(best);
    const int oi = bfly_i<S>(bi);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) {
      best = ov;
      bi = oi;
    }
    bfly_argmax<S * 2>(best, bi);
  }
}

}  // namespace rl4co

#endif
