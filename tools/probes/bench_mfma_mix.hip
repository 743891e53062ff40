// Throughput of the proposed forward inner loop: per 64-deep K group 4 x v_mfma_f32_32x32x16_f16 + 2 x
// v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3) on ONE accumulator chain, against 12 x f16 MFMAs (hi*hi + hi*lo + lo*hi today).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, int sa, int sb) {
  half8 a16[4], b16[4];
  v8i a8[2], b8[2];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a16[i][e] = (_Float16)(0.001f * (threadIdx.x + i + e)); b16[i][e] = (_Float16)(0.002f * (e + i)); }
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) { a8[i][e] = 0x38383838 + threadIdx.x + e; b8[i][e] = 0x30303030 + i + e; }
  f32x16 acc = {0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[s], b16[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[(s + 1) & 3], b16[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[s], b16[(s + 1) & 3], acc, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[s], b16[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], b8[0], acc, 0, 0, 0, sa, 0, sb);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[1], b8[1], acc, 0, 0, 0, sa - 3, 0, sb - 11);
      }
      asm volatile("" : "+v"(acc));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* o; (void)hipMalloc(&o, 256 * 256 * 4);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) k<0><<<256, 256>>>(o, iters, 127, 127); else k<1><<<256, 256>>>(o, iters, 124, 127);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("mode %d (%s): %.2f ms for %d groups per wave -> %.1f ns per group\n", mode,
                           mode ? "4 f16 + 2 scaled fp8" : "12 f16", ms, iters * 4, ms * 1e6 / (iters * 4));
    }
  }
  return 0;
}
