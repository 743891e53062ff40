// What fp16 matrix rate does an MI355X SUSTAIN?  The nominal dense peak (2.5 PFLOP/s) is 1024 SIMDs x 2.4 GHz x 1024 FLOP per
// cycle; under the board's power cap a kernel that keeps every matrix pipe busy does not run at 2.4 GHz.  This probe issues
// nothing but v_mfma_f32_32x32x16_f16 (independent accumulator chains, operands rotating through registers filled with
// random values in [-1, 1] -- or with zeros, to show how much of the cost is data) for a few hundred milliseconds of wall
// time on all CUs and reports TFLOP/s from HIP events.  Variants: 1 or 2 waves per SIMD; the block-scaled fp8 64-deep form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>   // 0: f16 32x32x16, 1: fp8 scaled 32x32x64, 2: the forward's group (4 f16 + 2 fp8 per 64-deep k)
__global__ __launch_bounds__(512) void burn(const unsigned* seed, float* out, int iters) {
  half8 a[8], b[4];
  v8i a8[4], b8[2];
  unsigned s = seed[0] ? (threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u) : 0u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = seed[0] ? (_Float16)((int)(rnd() >> 16 & 0x3ff) / 512.f - 1.f) : (_Float16)0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = seed[0] ? (_Float16)((int)(rnd() >> 16 & 0x3ff) / 512.f - 1.f) : (_Float16)0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a8[i][e] = seed[0] ? (int)(rnd() & 0x7e7e7e7e) : 0;   // finite e4m3 bytes
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b8[i][e] = seed[0] ? (int)(rnd() & 0x7e7e7e7e) : 0;
  f32x16 c[4] = {{0}, {0}, {0}, {0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (KIND == 0) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + k) & 3], c[k], 0, 0, 0);
        else if (KIND == 1) c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[(u + k) & 3], b8[u & 1], c[k], 0, 0, 0, 120, 0, 120);
        else if (k < 2) {   // per (u, k < 2): 4 f16 on c[k] + 2 fp8 on c[k + 2] = one 64-deep group of the FAST arithmetic
          c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[0], c[k], 0, 0, 0);
          c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 1) & 7], b[1], c[k], 0, 0, 0);
          c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 2) & 7], b[2], c[k], 0, 0, 0);
          c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 3) & 7], b[3], c[k], 0, 0, 0);
          c[k + 2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[u & 3], b8[0], c[k + 2], 0, 0, 0, 120, 0, 120);
          c[k + 2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[(u + 1) & 3], b8[1], c[k + 2], 0, 0, 0, 120, 0, 120);
        }
      }
    }
  }
  float r = 0;
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) r += c[k][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float* out; unsigned* seed;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&seed, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int cus = 256;
  for (int kind = 0; kind < 3; ++kind)
    for (int waves = 1; waves <= 2; ++waves)
      for (int d = 1; d >= 0; --d) {
        const unsigned data = (unsigned)d;
        (void)hipMemcpy(seed, &data, 4, hipMemcpyHostToDevice);
        const int threads = 256 * waves;
        const int iters = (kind == 0 ? 600000 : kind == 1 ? 300000 : 150000) / waves;          // ~0.3 - 0.5 s per run
        // kind 2: per iteration 8 x 2 groups; a group = 64-deep 32 x 32 product = 131072 FLOP of useful (fp32-class) work
        const double flop = kind == 2 ? (double)cus * (threads / 64) * iters * 16.0 * 2.0 * 32 * 32 * 64
                                      : (double)cus * (threads / 64) * iters * 32.0 * (kind == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 64);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipEventRecord(e0);
          if (kind == 0) burn<0><<<cus, threads>>>(seed, out, iters); else if (kind == 1) burn<1><<<cus, threads>>>(seed, out, iters); else burn<2><<<cus, threads>>>(seed, out, iters);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double tf = flop / (ms * 1e-3) / 1e12;
        const double cyc_per_iter = kind == 0 ? 32.0 * 32 : kind == 1 ? 32.0 * 64 : 16.0 * 256;   // matrix cycles per loop iteration
        const double ghz = (double)iters * cyc_per_iter / (ms * 1e-3) / 1e9 * (waves == 2 ? 2.0 : 1.0);   // if the pipe never idles
        printf("%-28s %d wave(s)/SIMD, %s operands: %8.1f ms  %7.1f TFLOP/s = %.3f of the nominal %s peak  (pipe-busy clock >= %.2f GHz)\n",
               kind == 0 ? "v_mfma_f32_32x32x16_f16" : kind == 1 ? "v_mfma_scale_..x64 (fp8)" : "FAST group 4 f16 + 2 fp8", waves,
               data ? "random" : "zero", ms, tf, tf / (kind == 1 ? 5033.2 : 2516.6), kind == 1 ? "5.0 PF" : "2.5 PF (useful flops)", ghz);
      }
  return 0;
}
