// entry16.hip — the C-ABI of the 16-bit training-encoder / attention kernels: ONE entry point per operation with the
// element type as its first argument (GENERATED from include/rl4co_amd.h at r06; until r05 the header carried every one
// of them twice). The implementations are the two builds of each translation unit (csrc/elem16.h: RL4CO_ENTRY), hidden
// symbols of the library.
#include <hip/hip_runtime.h>

#include "common.h"

extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_bwd_impl_bf16(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, void* dqkv, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_bwd_impl_f16(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, void* dqkv, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_bwd_wide_impl_bf16(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, void* dqkv, float* dq_partial, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_bwd_wide_impl_f16(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, void* dqkv, float* dq_partial, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_flash_impl_bf16(const void* qkv, int B, int N, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_flash_impl_f16(const void* qkv, int B, int N, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_flash_pre_impl_bf16(const void* qkv, const float* bound, int B, int N, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_flash_pre_impl_f16(const void* qkv, const float* bound, int B, int N, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_fwd_impl_bf16(const void* qkv, int B, int N, void* out, float* lse, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_attn_fwd_impl_f16(const void* qkv, int B, int N, void* out, float* lse, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_bnorm_apply_impl_bf16(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t M, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_bnorm_apply_impl_f16(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t M, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_bnorm_bwd_impl_bf16(const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma, int64_t M, float* sums, void* dy, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_bnorm_bwd_impl_f16(const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma, int64_t M, float* sums, void* dy, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_cross_attn_fwd_impl_bf16(const rl4co_cross_attn_args* args, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_cross_attn_fwd_impl_f16(const rl4co_cross_attn_args* args, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_cross_attn_bwd_impl_bf16(const rl4co_cross_attn_args* args, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_cross_attn_bwd_impl_f16(const rl4co_cross_attn_args* args, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_init_embed_impl_bf16(const float* feats, const float* w, const float* b, int64_t M, int F, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_init_embed_impl_f16(const float* feats, const float* w, const float* b, int64_t M, int F, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_init_embed_wgrad_impl_bf16(const void* dout, const float* feats, int64_t M, int F, float* partial, int* blocks_out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_init_embed_wgrad_impl_f16(const void* dout, const float* feats, int64_t M, int F, float* partial, int* blocks_out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_linear_impl_bf16(const void* a, const void* w, const float* bias, const void* mask, const void* residual, int64_t M, int N, int K, int relu, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_linear_impl_f16(const void* a, const void* w, const float* bias, const void* mask, const void* residual, int64_t M, int N, int K, int relu, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_bnorm_eval_impl_bf16(const void* x, const void* skip, const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t M, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_bnorm_eval_impl_f16(const void* x, const void* skip, const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t M, void* out, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_bnorm_stats_impl_bf16(const void* x, const void* s, int64_t M, void* y, float* sums, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_bnorm_stats_impl_f16(const void* x, const void* s, int64_t M, void* y, float* sums, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_inorm_bwd_impl_bf16(const void* dout, const void* y, const float* gamma, const float* mean, const float* rstd, int B, int N, void* dy, float* dgamma, float* dbeta, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_inorm_bwd_impl_f16(const void* dout, const void* y, const float* gamma, const float* mean, const float* rstd, int B, int N, void* dy, float* dgamma, float* dbeta, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_inorm_fwd_impl_bf16(const void* x, const void* s, const float* gamma, const float* beta, float eps, int B, int N, void* y, void* out, float* mean, float* rstd, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_inorm_fwd_impl_f16(const void* x, const void* s, const float* gamma, const float* beta, float eps, int B, int N, void* y, void* out, float* mean, float* rstd, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_lnorm_bwd_impl_bf16(const void* dout, const void* y, const float* stats, int B, int N, void* dy, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_lnorm_bwd_impl_f16(const void* dout, const void* y, const float* stats, int B, int N, void* dy, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_lnorm_fwd_impl_bf16(const void* x, const void* s, float eps, int B, int N, void* y, void* out, float* stats, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_skip_lnorm_fwd_impl_f16(const void* x, const void* s, float eps, int B, int N, void* y, void* out, float* stats, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_wgrad_impl_bf16(const void* dy, const void* x, int64_t M, int N, int K, int chunks, float* partial, float* partial_bias, int64_t chunk_stride, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl4co_wgrad_impl_f16(const void* dy, const void* x, int64_t M, int N, int K, int chunks, float* partial, float* partial_bias, int64_t chunk_stride, void* stream);

extern "C" int rl4co_attn_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, void* dqkv, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_attn_bwd_impl_bf16(qkv, out, dout, lse, B, N, dqkv, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_attn_bwd_impl_f16(qkv, out, dout, lse, B, N, dqkv, stream);
  return rl4co::record_arg_error("rl4co_attn_bwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_attn_bwd_wide(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, void* dqkv, float* dq_partial, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_attn_bwd_wide_impl_bf16(qkv, out, dout, lse, B, N, dqkv, dq_partial, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_attn_bwd_wide_impl_f16(qkv, out, dout, lse, B, N, dqkv, dq_partial, stream);
  return rl4co::record_arg_error("rl4co_attn_bwd_wide: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_cross_attn_fwd(int dtype, const rl4co_cross_attn_args* args, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_cross_attn_fwd_impl_bf16(args, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_cross_attn_fwd_impl_f16(args, stream);
  return rl4co::record_arg_error("rl4co_cross_attn_fwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_cross_attn_bwd(int dtype, const rl4co_cross_attn_args* args, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_cross_attn_bwd_impl_bf16(args, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_cross_attn_bwd_impl_f16(args, stream);
  return rl4co::record_arg_error("rl4co_cross_attn_bwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_attn_flash(int dtype, const void* qkv, int B, int N, void* out, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_attn_flash_impl_bf16(qkv, B, N, out, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_attn_flash_impl_f16(qkv, B, N, out, stream);
  return rl4co::record_arg_error("rl4co_attn_flash: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_attn_flash_pre(int dtype, const void* qkv, const float* bound, int B, int N, void* out, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_attn_flash_pre_impl_bf16(qkv, bound, B, N, out, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_attn_flash_pre_impl_f16(qkv, bound, B, N, out, stream);
  return rl4co::record_arg_error("rl4co_attn_flash_pre: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_attn_fwd(int dtype, const void* qkv, int B, int N, void* out, float* lse, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_attn_fwd_impl_bf16(qkv, B, N, out, lse, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_attn_fwd_impl_f16(qkv, B, N, out, lse, stream);
  return rl4co::record_arg_error("rl4co_attn_fwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_bnorm_apply(int dtype, const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t M, void* out, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_bnorm_apply_impl_bf16(y, mean, rstd, gamma, beta, M, out, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_bnorm_apply_impl_f16(y, mean, rstd, gamma, beta, M, out, stream);
  return rl4co::record_arg_error("rl4co_bnorm_apply: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_bnorm_bwd(int dtype, const void* dout, const void* y, const float* mean, const float* rstd, const float* gamma, int64_t M, float* sums, void* dy, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_bnorm_bwd_impl_bf16(dout, y, mean, rstd, gamma, M, sums, dy, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_bnorm_bwd_impl_f16(dout, y, mean, rstd, gamma, M, sums, dy, stream);
  return rl4co::record_arg_error("rl4co_bnorm_bwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_init_embed(int dtype, const float* feats, const float* w, const float* b, int64_t M, int F, void* out, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_init_embed_impl_bf16(feats, w, b, M, F, out, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_init_embed_impl_f16(feats, w, b, M, F, out, stream);
  return rl4co::record_arg_error("rl4co_init_embed: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_init_embed_wgrad(int dtype, const void* dout, const float* feats, int64_t M, int F, float* partial, int* blocks_out, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_init_embed_wgrad_impl_bf16(dout, feats, M, F, partial, blocks_out, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_init_embed_wgrad_impl_f16(dout, feats, M, F, partial, blocks_out, stream);
  return rl4co::record_arg_error("rl4co_init_embed_wgrad: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_linear(int dtype, const void* a, const void* w, const float* bias, const void* mask, const void* residual, int64_t M, int N, int K, int relu, void* out, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_linear_impl_bf16(a, w, bias, mask, residual, M, N, K, relu, out, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_linear_impl_f16(a, w, bias, mask, residual, M, N, K, relu, out, stream);
  return rl4co::record_arg_error("rl4co_linear: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_skip_bnorm_eval(int dtype, const void* x, const void* skip, const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t M, void* out, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_skip_bnorm_eval_impl_bf16(x, skip, mean, rstd, gamma, beta, M, out, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_skip_bnorm_eval_impl_f16(x, skip, mean, rstd, gamma, beta, M, out, stream);
  return rl4co::record_arg_error("rl4co_skip_bnorm_eval: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_skip_bnorm_stats(int dtype, const void* x, const void* s, int64_t M, void* y, float* sums, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_skip_bnorm_stats_impl_bf16(x, s, M, y, sums, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_skip_bnorm_stats_impl_f16(x, s, M, y, sums, stream);
  return rl4co::record_arg_error("rl4co_skip_bnorm_stats: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_skip_inorm_bwd(int dtype, const void* dout, const void* y, const float* gamma, const float* mean, const float* rstd, int B, int N, void* dy, float* dgamma, float* dbeta, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_skip_inorm_bwd_impl_bf16(dout, y, gamma, mean, rstd, B, N, dy, dgamma, dbeta, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_skip_inorm_bwd_impl_f16(dout, y, gamma, mean, rstd, B, N, dy, dgamma, dbeta, stream);
  return rl4co::record_arg_error("rl4co_skip_inorm_bwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_skip_inorm_fwd(int dtype, const void* x, const void* s, const float* gamma, const float* beta, float eps, int B, int N, void* y, void* out, float* mean, float* rstd, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_skip_inorm_fwd_impl_bf16(x, s, gamma, beta, eps, B, N, y, out, mean, rstd, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_skip_inorm_fwd_impl_f16(x, s, gamma, beta, eps, B, N, y, out, mean, rstd, stream);
  return rl4co::record_arg_error("rl4co_skip_inorm_fwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_skip_lnorm_bwd(int dtype, const void* dout, const void* y, const float* stats, int B, int N, void* dy, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_skip_lnorm_bwd_impl_bf16(dout, y, stats, B, N, dy, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_skip_lnorm_bwd_impl_f16(dout, y, stats, B, N, dy, stream);
  return rl4co::record_arg_error("rl4co_skip_lnorm_bwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_skip_lnorm_fwd(int dtype, const void* x, const void* s, float eps, int B, int N, void* y, void* out, float* stats, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_skip_lnorm_fwd_impl_bf16(x, s, eps, B, N, y, out, stats, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_skip_lnorm_fwd_impl_f16(x, s, eps, B, N, y, out, stats, stream);
  return rl4co::record_arg_error("rl4co_skip_lnorm_fwd: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
extern "C" int rl4co_wgrad(int dtype, const void* dy, const void* x, int64_t M, int N, int K, int chunks, float* partial, float* partial_bias, int64_t chunk_stride, void* stream) {
  if (dtype == RL4CO_DT_BF16) return rl4co_wgrad_impl_bf16(dy, x, M, N, K, chunks, partial, partial_bias, chunk_stride, stream);
  if (dtype == RL4CO_DT_F16) return rl4co_wgrad_impl_f16(dy, x, M, N, K, chunks, partial, partial_bias, chunk_stride, stream);
  return rl4co::record_arg_error("rl4co_wgrad: dtype must be RL4CO_DT_BF16 or RL4CO_DT_F16");
}
