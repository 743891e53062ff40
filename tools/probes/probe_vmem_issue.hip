// What one memory instruction costs the wave that issues it inside a stream of matrix instructions (gfx950): an LDS-DMA piece
// (global_load_lds_dwordx4, 1 KiB per wave) against the register-staged form of the same copy (global_load_dwordx4 into four
// VGPRs + ds_write_b128 one iteration later).  One op per 8 v_mfma_f32_32x32x16_f16 (what a data-gradient wave of
// csrc/bwd_pipe.hip issues), every CU busy, the source L2-resident (2 MB per XCD, re-read), 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o probe_vmem_issue probe_vmem_issue.hip && ./probe_vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048, OPS = 1;        // iterations of 8 matrix instructions, memory ops per iteration

template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(const char* src, long long* out, float seed) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f); }
  f32x16 acc = {0};
  asm volatile("" : "+v"(a), "+v"(b));
  const char* base = src + ((size_t)blockIdx.x * 8 + (wave & 7)) * 8192 + lane * 16;     // 8 KiB per wave, re-read
  const unsigned ldsw = (unsigned)(uintptr_t)lds + wave * 8192 + lane * 16;
  v4u held = {0, 0, 0, 0};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    const char* p = base + (it & 7) * 1024;
    const unsigned dst = ldsw + (it & 7) * 1024;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      if (m == 1) {
        if (KIND == 1) {
          const unsigned d = __builtin_amdgcn_readfirstlane(dst - lane * 16);
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(d) : "memory");
        } else if (KIND == 2 || KIND == 3) {
          if (KIND == 2) {      // the piece loaded one iteration ago goes to LDS, then the next load into the same registers
            asm volatile("s_waitcnt vmcnt(0)\n\tds_write_b128 %0, %1" :: "v"(dst), "v"(held) : "memory");
          }
          // ("+v": the destination stays reserved while the load is in flight -- with "=v" hipcc re-used it for the next address)
          asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(held) : "v"(p) : "memory");
        } else if (KIND == 4) {
          asm volatile("ds_write_b128 %0, %1" :: "v"(dst), "v"(held) : "memory");
        }
      }
      if (m == 7 && KIND == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      if (m == 7 && KIND == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + (wave & 7)] = t1 - t0;
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  if (s + (float)held[0] == 12345.678f) out[0] = 0;
}

template <int KIND, int THREADS>
static double run(const char* src, long long* out, const char* name) {
  hipFuncSetAttribute((const void*)k<KIND, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  std::vector<long long> h(256 * 8);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<KIND, THREADS>), dim3(256), dim3(THREADS), 65536, 0, src, out, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < THREADS / 64; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    best = std::min(best, sum / n / ITER);
  }
  printf("%-58s %2d waves/CU: %7.1f clocks per 8 matrix instructions\n", name, THREADS / 64, best);
  fflush(stdout);
  return best;
}

int main() {
  char* src; long long* out;
  hipMalloc(&src, 256 * 8 * 8192 + 65536); hipMemset(src, 0, 256 * 8 * 8192 + 65536);
  hipMalloc(&out, 256 * 8 * 8);
  const double n1 = run<0, 256>(src, out, "matrix instructions only");
  const double g1 = run<1, 256>(src, out, "+ 1 LDS-DMA piece (global_load_lds_dwordx4)");
  const double r1 = run<2, 256>(src, out, "+ 1 register-staged piece (global_load_dwordx4, ds_write_b128)");
  const double l1 = run<3, 256>(src, out, "+ 1 global_load_dwordx4 only");
  const double w1 = run<4, 256>(src, out, "+ 1 ds_write_b128 only");
  printf("   per op: LDS-DMA %.0f, register-staged %.0f (load %.0f + write %.0f)\n", g1 - n1, r1 - n1, l1 - n1, w1 - n1);
  const double n2 = run<0, 512>(src, out, "matrix instructions only");
  const double g2 = run<1, 512>(src, out, "+ 1 LDS-DMA piece");
  const double r2 = run<2, 512>(src, out, "+ 1 register-staged piece");
  const double l2 = run<3, 512>(src, out, "+ 1 global_load_dwordx4 only");
  const double w2 = run<4, 512>(src, out, "+ 1 ds_write_b128 only");
  printf("   per op (per wave; the SIMD's pipe is shared): LDS-DMA %.0f, register-staged %.0f (load %.0f + write %.0f)\n", g2 - n2, r2 - n2, l2 - n2, w2 - n2);
  return 0;
}
