// elem16.h — the 16-bit element type of a training / attention translation unit.
//
// The training-encoder kernels (am_train_ops.hip, am_train_attn.hip) and the flash attention (am_attn_flash.hip) are
// written once against `elem_t` and compiled twice: as they stand for bfloat16 (torch.autocast(bfloat16), hidden
// implementation symbols rl4co_*_impl_bf16) and through the one-line wrappers *_f16.hip, which define RL4CO_ELEM_F16 and
// include the same source, for IEEE half (torch.autocast(float16) — the reference's default "16-mixed" precision,
// rl4co/utils/trainer.py:57 — rl4co_*_impl_f16). The C-ABI has ONE entry point per operation with the element type as its
// first argument (csrc/entry16.hip, r06). Storage, LDS layouts, transpose reads and MFMA shapes are identical; what differs is the MFMA
// opcode and the conversions: bf16 <-> fp32 is a 16-bit shift, half <-> fp32 the hardware's v_cvt (round to nearest
// even, overflow to infinity — what GradScaler's inf check expects of fp16 gradients).
#ifndef RL4CO_ELEM16_H
#define RL4CO_ELEM16_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef RL4CO_ELEM_F16
#define RL4CO_ELEM_F16 0
#endif

#if RL4CO_ELEM_F16
typedef _Float16 elem_t;
#define RL4CO_ENTRY(stem) __attribute__((visibility("hidden"))) stem##_impl_f16
#define RL4CO_IMPL(stem) stem##_impl_f16  // the name alone: a call from another translation unit of the same element type
#define RL4CO_CXX(stem) stem##_f16
#else
typedef __bf16 elem_t;
#define RL4CO_ENTRY(stem) __attribute__((visibility("hidden"))) stem##_impl_bf16
#define RL4CO_IMPL(stem) stem##_impl_bf16
#define RL4CO_CXX(stem) stem
#endif

namespace rl4co_e16 {

typedef elem_t e2 __attribute__((ext_vector_type(2)));
typedef elem_t e4 __attribute__((ext_vector_type(4)));
typedef elem_t e8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// two packed elements of a 32-bit word -> fp32 (low half first in memory)
__device__ inline float lo(uint32_t v) {
#if RL4CO_ELEM_F16
  return (float)__builtin_bit_cast(e2, v)[0];
#else
  return __uint_as_float(v << 16);
#endif
}
__device__ inline float hi(uint32_t v) {
#if RL4CO_ELEM_F16
  return (float)__builtin_bit_cast(e2, v)[1];
#else
  return __uint_as_float(v & 0xffff0000u);
#endif
}
__device__ inline uint32_t pack(float a, float b) {
  e2 v;
  v[0] = (elem_t)a;
  v[1] = (elem_t)b;
  return __builtin_bit_cast(uint32_t, v);
}

// four fp32 -> four packed elements as TWO pair conversions (v_cvt_pk_bf16_f32 a, b / v_cvt_pkrtz..: one instruction per
// pair). Built element by element — `v[i] = (elem_t)x[i]` with other arithmetic between the conversions — the compiler
// converts each value alone (v_cvt_pk x, 0) and merges the halves with v_perm_b32: three instructions per pair (r05 ISA:
// 592 single conversions + 296 v_perm in am_attn_flash alone).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline e4 cvt4(float a, float b, float c, float d) {
  const e2 lo2 = __builtin_convertvector(f2{a, b}, e2), hi2 = __builtin_convertvector(f2{c, d}, e2);
  const uint2 u = make_uint2(__builtin_bit_cast(uint32_t, lo2), __builtin_bit_cast(uint32_t, hi2));
  return __builtin_bit_cast(e4, u);
}

__device__ inline f16v mfma_32x32x16(const e8& a, const e8& b, const f16v& c) {
#if RL4CO_ELEM_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ inline f4 mfma_16x16x32(const e8& a, const e8& b, const f4& c) {
#if RL4CO_ELEM_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ inline f4 mfma_16x16x16(const e4& a, const e4& b, const f4& c) {
#if RL4CO_ELEM_F16
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
#endif
}

// four dims of a folded context-table row: fp32 (16 bytes at p), or (r06) the translation unit's 16-bit element type (8
// bytes at p) widened on load. BRANCH-FREE on purpose: one 16-byte load from the enclosing 16-byte line in either case
// (fp32 rows: p itself, it is 16-byte aligned), the halves picked by selects. Written as `if (half) load 8 else load 16`
// the compiler put a branch around each load and an s_waitcnt vmcnt(0) behind the 8-byte one — the row is requested one
// step block AHEAD of its use, so that wait exposed its whole L2 round trip and drained the context-row scatter issued
// before it as well (first r06 build: teacher backward 5.2 -> 9.8 ms). Table bases and row strides are multiples of 16 B.
__device__ inline float4 load_ctx4(const char* p, bool half_rows) {
  // (pointer arithmetic on p, not an integer round trip: the load must stay a GLOBAL load — a flat one counts on lgkmcnt too
  // and every LDS wait of the step block would wait for the prefetch)
  const uint32_t low = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)) & 15u;
  const uint4 raw = *reinterpret_cast<const uint4*>(p - low);
  const bool upper = (low & 8u) != 0;
  const uint32_t u0 = upper ? raw.z : raw.x, u1 = upper ? raw.w : raw.y;
  float4 v;
  v.x = half_rows ? lo(u0) : __uint_as_float(raw.x);
  v.y = half_rows ? hi(u0) : __uint_as_float(raw.y);
  v.z = half_rows ? lo(u1) : __uint_as_float(raw.z);
  v.w = half_rows ? hi(u1) : __uint_as_float(raw.w);
  return v;
}

}  // namespace rl4co_e16

#endif
