// Probe of the fp6 (e2m3) path of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950: issue cadence against the fp8 form, operand
// element mapping, and the 32-wide conversion v_cvt_scalef32_pk32_fp6_f16 (rounding, subnormals, scale direction).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FMT>   // -1: f16 32x32x16; else cbsz = blgp = FMT (0 fp8, 1 bf8, 2 fp6, 3 bf6, 4 fp4)
__global__ __launch_bounds__(256, 1) void timing(float* out, int iters) {
  half8 a16, b16;
  v8i a8, b8;
  for (int e = 0; e < 8; ++e) { a16[e] = (_Float16)(0.001f * (threadIdx.x + e)); b16[e] = (_Float16)(0.002f * e); a8[e] = 0x12345678 * (threadIdx.x + e + 1); b8[e] = 0x9abcdef1 * (e + 3); }
  f32x16 acc = {0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (FMT < 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, b16, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc, FMT < 0 ? 0 : FMT, FMT < 0 ? 0 : FMT, 0, 127, 0, 127);
      asm volatile("" : "+v"(acc));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void layout(const float* A, const float* B, float* C, float scale, float* cvt_out) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  half32 av, bv;
  for (int j = 0; j < 32; ++j) { av[j] = (_Float16)A[r * 64 + 32 * h + j]; bv[j] = (_Float16)B[(32 * h + j) * 32 + r]; }
  const v6i a6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(av, scale);
  const v6i b6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(bv, 1.0f);
  v8i a = {a6[0], a6[1], a6[2], a6[3], a6[4], a6[5], 0, 0}, b = {b6[0], b6[1], b6[2], b6[3], b6[4], b6[5], 0, 0};
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, 127, 0, 127);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
  if (l == 0) {   // what the conversion does to single values: x -> fp6 -> f16
    half32 t;
    for (int j = 0; j < 32; ++j) t[j] = (_Float16)0;
    const float v[16] = {0.05f, 0.0624f, 0.07f, 0.125f, 0.19f, 0.3f, 0.9f, 1.0f, 1.06f, 1.07f, 1.9f, 3.3f, 7.4f, 9.f, -0.4f, -5.1f};
    for (int j = 0; j < 16; ++j) t[j] = (_Float16)v[j];
    const v6i w = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(t, scale);
    const half32 back = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(w, 1.0f);
    for (int j = 0; j < 16; ++j) { cvt_out[2 * j] = v[j]; cvt_out[2 * j + 1] = (float)back[j]; }
  }
}

static float fp6_round(float x, bool flush) {   // e2m3: normals 1 .. 7.5 (step 2^(e-3)), subnormals k / 8
  if (x == 0.f) return 0.f;
  float a = fabsf(x);
  int e; frexpf(a, &e); e -= 1;
  if (e < 0) e = 0;
  float q = ldexpf(1.f, e - 3);
  float r = nearbyintf(a / q) * q;
  if (r > 7.5f) r = 7.5f;
  if (flush && r < 1.0f) r = 0.f;
  return copysignf(r, x);
}

int main() {
  float* o; (void)hipMalloc(&o, 256 * 256 * 4);
  const int iters = 20000;
  float base = 0;
  for (int f = -1; f <= 4; ++f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      switch (f) {
        case -1: timing<-1><<<256, 256>>>(o, iters); break;
        case 0: timing<0><<<256, 256>>>(o, iters); break;
        case 1: timing<1><<<256, 256>>>(o, iters); break;
        case 2: timing<2><<<256, 256>>>(o, iters); break;
        case 3: timing<3><<<256, 256>>>(o, iters); break;
        case 4: timing<4><<<256, 256>>>(o, iters); break;
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    if (f < 0) base = ms;
    printf("format %d: %.2f ms, %.1f ns per instruction, x%.2f of the f16 32x32x16 (32 cycles) -> %.0f cycles\n", f, ms,
           ms * 1e6 / (iters * 8), ms / base, 32.0 * ms / base);
  }
  static float A[32 * 64], B[64 * 32], C[32 * 32], cv[32];
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
  for (auto& v : A) { v = rnd() * 7.f; if (fabsf(v) < 1.5f) v *= 0.3f; }
  for (auto& v : B) v = rnd() * 4.f;
  for (auto& v : A) v = (float)(_Float16)v;
  for (auto& v : B) v = (float)(_Float16)v;
  float *dA, *dB, *dC, *dcv;
  (void)hipMalloc(&dA, sizeof(A)); (void)hipMalloc(&dB, sizeof(B)); (void)hipMalloc(&dC, sizeof(C)); (void)hipMalloc(&dcv, sizeof(cv));
  (void)hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
  for (float scale : {1.0f, 2.0f}) {
    layout<<<1, 64>>>(dA, dB, dC, scale, dcv);
    (void)hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost); (void)hipMemcpy(cv, dcv, sizeof(cv), hipMemcpyDeviceToHost);
    for (int flush = 0; flush < 2; ++flush)
      for (int dir = 0; dir < 2; ++dir) {   // dir 0: conversion divides by scale; 1: multiplies
        double err = 0, mag = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
          double acc = 0;
          for (int kk = 0; kk < 64; ++kk) {
            const float as = dir ? A[i * 64 + kk] * scale : A[i * 64 + kk] / scale;
            acc += (double)fp6_round(as, flush) * (double)fp6_round(B[kk * 32 + j], flush);
          }
          err = fmax(err, fabs(acc - C[i * 32 + j])); mag = fmax(mag, fabs(acc));
        }
        printf("scale %.1f flush %d %s: max |C - ref| = %.3e (max |ref| %.3e)\n", scale, flush, dir ? "multiplies" : "divides", err, mag);
      }
    printf("conversion at scale %.1f:", scale);
    for (int j = 0; j < 16; ++j) printf("  %.4g->%.4g", cv[2 * j], cv[2 * j + 1]);
    printf("\n");
  }
  return 0;
}
