/*
 * rl4co_math.h — deterministic fp32 exp / log / tanh shared by the HIP kernels
 * and by the specified-order C oracle (oracle/rollout_ref.c).
 *
 * Why not the vendor libm: glibc's expf/tanhf and ROCm's ocml versions differ in
 * the last ulp, and greedy decoding breaks ties at the tanh-saturation plateau
 * (utils/decoding.py:169-170: 10*tanh(x) == 10.0 for x >= ~9.01) by lowest
 * index — so host oracle and device kernel must round identically. Every
 * operation below is an explicit IEEE fp32 mul/add/fma/div in a fixed order;
 * both sides are compiled with -ffp-contract=off so nothing is re-fused.
 * Accuracy (tests/test_math.py, against float64 libm over the ranges the decode path
 * uses): exp <= 1.0 ulp, log <= 0.82 ulp, tanh <= 1.3 ulp measured (asserted at 1.25 /
 * 1.0 / 1.75); Cephes single-precision polynomials. The same test pins the device
 * evaluation to the host's bit for bit and the tanh == 1.0 knee to torch's.
 */
#ifndef RL4CO_MATH_H
#define RL4CO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RL4CO_HD __host__ __device__ static inline
#else
#define RL4CO_HD static inline
#endif

RL4CO_HD float rl4co_bits_to_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
RL4CO_HD uint32_t rl4co_float_to_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

/* 2^n for -126 <= n <= 127 */
RL4CO_HD float rl4co_pow2i(int n) { return rl4co_bits_to_float((uint32_t)(n + 127) << 23); }

RL4CO_HD float rl4co_expf(float x) {
  if (x != x) return x;
  if (x > 88.7228390f) return rl4co_bits_to_float(0x7f800000u); /* +inf */
  if (x < -103.9720840f) return 0.0f;                           /* also -inf */
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float y = fmaf(p, r * r, r) + 1.0f;
  int ni = (int)n;
  if (ni > 127) { /* 88.72 * log2e rounds to 128 at most */
    y = y * rl4co_pow2i(127);
    ni -= 127;
  } else if (ni < -126) { /* denormal results: two exact-power-of-two scalings */
    y = y * rl4co_pow2i(-100);
    ni += 100;
  }
  return y * rl4co_pow2i(ni);
}

RL4CO_HD float rl4co_logf(float x) {
  if (x != x) return x;
  if (x < 0.0f) return rl4co_bits_to_float(0x7fc00000u);
  if (x == 0.0f) return rl4co_bits_to_float(0xff800000u);
  uint32_t u = rl4co_float_to_bits(x);
  if (u == 0x7f800000u) return x;
  int e = 0;
  if (u < 0x00800000u) { /* denormal: scale by 2^23 */
    x = x * 8388608.0f;
    u = rl4co_float_to_bits(x);
    e = -23;
  }
  e += (int)(u >> 23) - 126;
  float m = rl4co_bits_to_float((u & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = (m + m) - 1.0f;
  } else {
    m = m - 1.0f;
  }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = fmaf(p, m, -1.1514610310e-1f);
  p = fmaf(p, m, 1.1676998740e-1f);
  p = fmaf(p, m, -1.2420140846e-1f);
  p = fmaf(p, m, 1.4249322787e-1f);
  p = fmaf(p, m, -1.6668057665e-1f);
  p = fmaf(p, m, 2.0000714765e-1f);
  p = fmaf(p, m, -2.4999993993e-1f);
  p = fmaf(p, m, 3.3333331174e-1f);
  float fe = (float)e;
  float y = (p * m) * z;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(z, -0.5f, y);
  float r = m + y;
  r = fmaf(fe, 0.693359375f, r);
  return r;
}

RL4CO_HD float rl4co_tanhf(float x) {
  if (x != x) return x;
  float z = fabsf(x);
  if (z > 44.0f) return x > 0.0f ? 1.0f : -1.0f;
  if (z >= 0.625f) {
    float s = rl4co_expf(z + z);
    float t = 1.0f - 2.0f / (s + 1.0f);
    return x < 0.0f ? -t : t;
  }
  float w = x * x;
  float p = -5.70498872745e-3f;
  p = fmaf(p, w, 2.06390887954e-2f);
  p = fmaf(p, w, -5.37397155531e-2f);
  p = fmaf(p, w, 1.33314422036e-1f);
  p = fmaf(p, w, -3.33332819422e-1f);
  return fmaf(p * w, x, x);
}

/* ---- Philox4x32-10 (Salmon et al. 2011), the in-kernel noise source for
 * throughput-mode sampling; keyed by (seed), counter = (offset+t, trajectory,
 * node/4, 0). Parity-mode sampling takes torch's Exp(1) draws as an input. ---- */
RL4CO_HD void rl4co_philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

/* Exp(1) draw for (step, trajectory, node): -log(u), u uniform on the open interval (0,1).
 * 23 random bits: (k + 0.5) * 2^-23 is exact in fp32 for k < 2^23 (a 24-bit significand), so
 * u <= 1 - 2^-24 < 1 and the draw is strictly positive — with 24 bits (k + 0.5) rounds up to 2^24
 * for the top k, u becomes 1.0, the noise 0 and a masked node's key 0/0 = NaN. */
RL4CO_HD float rl4co_exp1_noise(uint64_t seed, uint64_t step, uint32_t traj, uint32_t node) {
  uint32_t c[4] = {(uint32_t)step, traj, node >> 2, (uint32_t)(step >> 32) ^ 0x52344c43u};
  rl4co_philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  uint32_t w = c[node & 3];
  float u = ((float)(w >> 9) + 0.5f) * 1.1920928955078125e-7f; /* (k+0.5)/2^23 in (0,1) */
  return -rl4co_logf(u);
}

/* The four uniforms of nodes 4 g .. 4 g + 3 (one Philox block): u[i] is the value whose -log
 * rl4co_exp1_noise(seed, step, traj, 4 g + i) returns. */
RL4CO_HD void rl4co_uniform4(uint64_t seed, uint64_t step, uint32_t traj, uint32_t node_group, float u[4]) {
  uint32_t c[4] = {(uint32_t)step, traj, node_group, (uint32_t)(step >> 32) ^ 0x52344c43u};
  rl4co_philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int i = 0; i < 4; ++i) u[i] = ((float)(c[i] >> 9) + 0.5f) * 1.1920928955078125e-7f;
}

#endif /* RL4CO_MATH_H */
