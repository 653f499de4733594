#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out_c, float* out_tr) {
  __shared__ __align__(16) __bf16 lds[64 * 64];
  const int lane = threadIdx.x;
  // ---- MFMA layout probe: A[m][k] = m*16+k (as small ints exactly representable), B = identity-ish
  // C = A * B with B[k][n] = (k==n)  -> C[m][n] = A[m][n]
  {
    bf16x4 a, b;
    const int m = lane & 15, g = lane >> 4;
    for (int s = 0; s < 4; ++s) {
      a[s] = (__bf16)(float)(m * 16 + 4 * g + s);       // hypothesis: A[m = lane&15][k = 4g+s]
      b[s] = (__bf16)(((4 * g + s) == (lane & 15)) ? 1.0f : 0.0f);  // hypothesis: B[k = 4g+s][n = lane&15]
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out_c[lane * 4 + r] = c[r];
  }
  // ---- transpose read probe: lds[i] = i; every lane reads at address of its own choosing
  for (int i = lane; i < 64 * 64; i += 64) lds[i] = (__bf16)(float)(i % 256);
  __syncthreads();
  {
    // image: rows of 64 elements; group g = lane>>4 reads rows 4g..4g+3?? we pass per-lane address:
    // lane i of group supplies row (i>>2) + 4*g, col chunk 4*(i&3)
    const int i = lane & 15, g = lane >> 4;
    const __bf16* p = lds + ((i >> 2) + 4 * g) * 64 + 4 * (i & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    for (int r = 0; r < 4; ++r) {
      uint16_t bits = (uint16_t)v[r];
      out_tr[lane * 4 + r] = __uint_as_float((uint32_t)bits << 16);
    }
  }
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void timing(long long* out) {
  bf16x4 a = {1, 1, 1, 1}, b = {1, 1, 1, 1};
  bf16x8 a8 = {1, 1, 1, 1, 1, 1, 1, 1}, b8 = a8;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 1024; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c3, 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  f32x4 d0 = c0, d1 = c0, d2 = c0, d3 = c0;
  for (int i = 0; i < 1024; ++i) {
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, d1, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, d3, 0, 0, 0);
  }
  long long t2 = __builtin_readcyclecounter();
  f32x16 e0, e1;
  for (int i = 0; i < 16; ++i) { e0[i] = 0; e1[i] = 0; }
  for (int i = 0; i < 1024; ++i) {
    e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, e0, 0, 0, 0);
    e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, e1, 0, 0, 0);
  }
  long long t3 = __builtin_readcyclecounter();
  // VALU: dependent-free fma stream
  float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  for (int i = 0; i < 1024; ++i) {
    x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
    x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
  }
  long long t4 = __builtin_readcyclecounter();
  float y0 = threadIdx.x * 1e-3f, y1 = 0.1f, y2 = 0.2f, y3 = 0.3f;
  for (int i = 0; i < 1024; ++i) {
    y0 = __builtin_amdgcn_exp2f(y0) * 0.25f; y1 = __builtin_amdgcn_exp2f(y1) * 0.25f;
    y2 = __builtin_amdgcn_exp2f(y2) * 0.25f; y3 = __builtin_amdgcn_exp2f(y3) * 0.25f;
  }
  long long t5 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4;
  }
  if (c0[0] + c1[0] + c2[0] + c3[0] + d0[0] + d1[0] + d2[0] + d3[0] + e0[0] + e1[0] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 == 12345.0f) out[5] = 1;
}
int main() {
  long long* dtm; hipMalloc(&dtm, 64);
  for (int nw = 1; nw <= 2; ++nw) {
    timing<<<1, 64 * 4 * nw>>>(dtm);
    long long htm[8]; hipMemcpy(htm, dtm, 64, hipMemcpyDeviceToHost);
    printf("waves/SIMD=%d cycles (readcyclecounter units) per instr: mfma16x16x16 %.1f  mfma16x16x32 %.1f  mfma32x32x16 %.1f  v_fma %.2f  exp2+mul pair %.2f\n", nw,
           htm[0] / 4096.0, htm[1] / 4096.0, htm[2] / 2048.0, htm[3] / 8192.0, htm[4] / 4096.0);
  }
  float *dc, *dt; hipMalloc(&dc, 64*4*4); hipMalloc(&dt, 64*4*4);
  probe<<<1, 64>>>(dc, dt);
  float hc[256], ht[256];
  hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost); hipMemcpy(ht, dt, sizeof ht, hipMemcpyDeviceToHost);
  printf("C layout (lane: 4 regs) expected if C[m=4g+r][n=lane&15]: value = (4g+r)*16 + (lane&15)\n");
  int okc = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hc[l*4+r] != (float)((4*(l>>4)+r)*16 + (l&15))) okc = 0;
  printf("C_LAYOUT_OK=%d\n", okc);
  for (int l = 0; l < 64; l += 5) printf("lane %2d: %g %g %g %g\n", l, hc[l*4], hc[l*4+1], hc[l*4+2], hc[l*4+3]);
  printf("TR read: lane: 4 values (element index mod 256 of image rows of 64)\n");
  for (int l = 0; l < 64; ++l) printf("L%2d: %g %g %g %g\n", l, ht[l*4], ht[l*4+1], ht[l*4+2], ht[l*4+3]);
  return 0;
}
