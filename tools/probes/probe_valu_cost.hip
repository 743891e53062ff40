// Issue cost (cycles per wave-instruction, one wave per SIMD, independent operands) of the vector instructions the forward's
// epilogue is made of -- the kernel is bound by the ONE instruction stream of its wave, so each one's cost is what a "VALU diet"
// buys.  s_memtime around 16 x 32 back-to-back instances.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X
#define BENCH(NAME, ASM)                                                                                   \
  __global__ __launch_bounds__(256) void NAME(long long* out, float seed) {                                \
    float a = seed + threadIdx.x, b = seed * 0.5f, c = seed * 0.25f, d = 0.f, e = 0.f, s = 0x1p-17f;        \
    unsigned w = 0;                                                                                         \
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(w), "+s"(s));                       \
    const long long t0 = __builtin_amdgcn_s_memtime();                                                      \
    for (int i = 0; i < 16; ++i) { REP32(asm volatile(ASM : "+v"(d), "+v"(e), "+v"(w) : "v"(a), "v"(b), "v"(c), "s"(s));) } \
    asm volatile("s_nop 0" ::: "memory");                                                                   \
    const long long t1 = __builtin_amdgcn_s_memtime();                                                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                              \
    if (d + e + (float)w == 12345.6f) out[1] = 1;                                                           \
  }
BENCH(k_mul, "v_mul_f32 %0, %3, %4")
BENCH(k_fma, "v_fma_f32 %0, %3, %4, %5")
BENCH(k_sin, "v_sin_f32 %0, %3")
BENCH(k_cvt16, "v_cvt_pk_f16_f32 %0, %3, %4")
BENCH(k_cvt8, "v_cvt_pk_fp8_f32 %2, %3, %4")
BENCH(k_cvt8s, "v_cvt_scalef32_pk_fp8_f32 %2, %3, %4, %6")
BENCH(k_cvtbf8, "v_cvt_pk_bf8_f32 %2, %3, %4")
BENCH(k_fmamix, "v_fma_mix_f32 %0, %3, -1.0, %4 op_sel_hi:[1,0,0]")
BENCH(k_perm, "v_perm_b32 %2, %3, %4, %5")
BENCH(k_mov, "v_mov_b32 %0, %3")
BENCH(k_accw, "v_accvgpr_write_b32 a0, %3")
BENCH(k_fract, "v_fract_f32 %0, %3")
BENCH(k_nop, "s_nop 0")
typedef float f2 __attribute__((ext_vector_type(2)));
#define BENCH2(NAME, ASM)                                                                                  \
  __global__ __launch_bounds__(256) void NAME(long long* out, float seed) {                                \
    f2 a = {seed + threadIdx.x, seed}, b = {seed * 0.5f, 1.f}, d = {0.f, 0.f};                              \
    asm volatile("" : "+v"(a), "+v"(b), "+v"(d));                                                           \
    const long long t0 = __builtin_amdgcn_s_memtime();                                                      \
    for (int i = 0; i < 16; ++i) { REP32(asm volatile(ASM : "+v"(d) : "v"(a), "v"(b));) }                    \
    asm volatile("s_nop 0" ::: "memory");                                                                   \
    const long long t1 = __builtin_amdgcn_s_memtime();                                                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                              \
    if (d[0] + d[1] == 12345.6f) out[1] = 1;                                                                \
  }
BENCH2(k_pkmul, "v_pk_mul_f32 %0, %1, %2")
BENCH2(k_pkadd, "v_pk_add_f32 %0, %1, %2")
// independent destinations (no read-after-write between neighbours), 1 or 2 waves per SIMD
__global__ __launch_bounds__(512) void k_mul_indep(long long* out, float seed) {
  float a = seed + threadIdx.x, b = seed * 0.5f, d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
  asm volatile("" : "+v"(a), "+v"(b));
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i)
    asm volatile("v_mul_f32 %0, %8, %9\n v_mul_f32 %1, %8, %9\n v_mul_f32 %2, %8, %9\n v_mul_f32 %3, %8, %9\n"
                 "v_mul_f32 %4, %8, %9\n v_mul_f32 %5, %8, %9\n v_mul_f32 %6, %8, %9\n v_mul_f32 %7, %8, %9"
                 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) : "v"(a), "v"(b));
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 12345.6f) out[1] = 1;
}
__global__ __launch_bounds__(512) void k_mix_indep(long long* out, float seed) {   // VALU / SALU / LDS-free mix: v_mul, s_nop, v_mov alternating
  float a = seed + threadIdx.x, b = seed * 0.5f, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  asm volatile("" : "+v"(a), "+v"(b));
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i)
    asm volatile("v_mul_f32 %0, %4, %5\n s_nop 0\n v_mul_f32 %1, %4, %5\n s_nop 0\n v_mul_f32 %2, %4, %5\n s_nop 0\n v_mul_f32 %3, %4, %5\n s_nop 0"
                 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b));
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (d0 + d1 + d2 + d3 == 12345.6f) out[1] = 1;
}
int main() {
  long long* o; (void)hipMalloc(&o, 16); long long h[2];
#define RUN(NAME, LABEL) for (int r = 0; r < 2; ++r) { NAME<<<256, 256>>>(o, 1.5f); (void)hipDeviceSynchronize(); } \
  (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); printf("%-34s %.2f cycles per instruction\n", LABEL, (double)h[0] / 512.0 * 1.0);
  RUN(k_nop, "s_nop 0") RUN(k_mov, "v_mov_b32") RUN(k_mul, "v_mul_f32") RUN(k_fma, "v_fma_f32") RUN(k_pkmul, "v_pk_mul_f32") RUN(k_pkadd, "v_pk_add_f32")
  RUN(k_fmamix, "v_fma_mix_f32") RUN(k_fract, "v_fract_f32") RUN(k_sin, "v_sin_f32") RUN(k_cvt16, "v_cvt_pk_f16_f32") RUN(k_cvt8, "v_cvt_pk_fp8_f32")
  RUN(k_cvt8s, "v_cvt_scalef32_pk_fp8_f32") RUN(k_cvtbf8, "v_cvt_pk_bf8_f32") RUN(k_perm, "v_perm_b32") RUN(k_accw, "v_accvgpr_write_b32")
  for (int threads : {256, 512}) {
    for (int r = 0; r < 2; ++r) { k_mul_indep<<<256, threads>>>(o, 1.5f); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("8 independent v_mul_f32 per asm block, %d wave(s) per SIMD: %.2f ticks per instruction of one wave\n", threads / 256, (double)h[0] / 512.0);
    for (int r = 0; r < 2; ++r) { k_mix_indep<<<256, threads>>>(o, 1.5f); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("v_mul_f32 / s_nop 0 alternating,          %d wave(s) per SIMD: %.2f ticks per instruction of one wave\n", threads / 256, (double)h[0] / 512.0);
  }
  printf("(s_memtime ticks at a constant 100 MHz?  compare with s_nop 0 = 1 issue cycle)\n");
  return 0;
}
