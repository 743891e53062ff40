// Probe of v_mfma_i32_32x32x32_i8 on gfx950: operand lane / byte mapping, cycles per instruction against the block-scaled
// fp8 64-deep form and the fp16 16-deep form, and the float -> int8 pack paths (v_cvt_pknorm_i16_f32 + v_perm_b32 of the
// high bytes; v_cvt_pk_u8_f32).  ASSUMED mapping: lane l (r = l & 31, h = l >> 5) byte j (0..15) of A <-> A[row r][k = 16 h + j].
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef short s2 __attribute__((ext_vector_type(2)));

__global__ void layout(const signed char* A, const signed char* B, int* C) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v4i a, b;
  for (int d = 0; d < 4; ++d) {
    unsigned wa = 0, wb = 0;
    for (int j = 0; j < 4; ++j) {
      wa |= (unsigned)(unsigned char)A[r * 32 + 16 * h + 4 * d + j] << (8 * j);
      wb |= (unsigned)(unsigned char)B[(16 * h + 4 * d + j) * 32 + r] << (8 * j);
    }
    a[d] = (int)wa; b[d] = (int)wb;
  }
  i32x16 c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

// time N dependent-free back-to-back instructions of each kind (one wave per SIMD: 4 waves per block, 1 block per CU)
template <int KIND>
__global__ void timing(long long* out, int iters) {
  v4i a4 = {(int)threadIdx.x, 2, 3, 4}, b4 = {5, 6, 7, (int)threadIdx.x};
  v8i a8 = {1, 2, 3, 4, 5, 6, 7, (int)threadIdx.x}, b8 = {(int)threadIdx.x, 2, 3, 4, 5, 6, 7, 8};
  half8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(0.01f * (threadIdx.x + i)); bh[i] = (_Float16)(0.02f * i); }
  i32x16 ci[4] = {{0}, {0}, {0}, {0}};
  f32x16 cf[4] = {{0}, {0}, {0}, {0}};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0) ci[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, ci[u], 0, 0, 0);
      if (KIND == 1) cf[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, cf[u], 0, 0, 0, 127, 0, 127);
      if (KIND == 2) cf[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, cf[u], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0; float f = 0;
  for (int u = 0; u < 4; ++u) { s += ci[u][0]; f += cf[u][0]; }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = s + (long long)f; }
}

// the forward's per-group chain: 4 fp16 MFMAs on `acc`, then the two 64-deep fp8 MFMAs -- MODE 0: both on ONE correction
// accumulator (as render_fwd.hip does), MODE 1: on two accumulators, MODE 2: the fp8 pair interleaved between the fp16 ones
template <int MODE>
__global__ void chain(long long* out, int iters) {
  v8i a8 = {1, 2, 3, 4, 5, 6, 7, (int)threadIdx.x}, b8 = {(int)threadIdx.x, 2, 3, 4, 5, 6, 7, 8};
  half8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(0.01f * (threadIdx.x + i)); bh[i] = (_Float16)(0.02f * i); }
  f32x16 acc = {0}, c0 = {0}, c1 = {0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 127, 0, 127);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 127, 0, 116);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 127, 0, 127);
      if (MODE == 0) c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 127, 0, 116);
      else c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 0, 0, 0, 127, 0, 116);
    }
    asm volatile("" : "+v"(acc), "+v"(c0), "+v"(c1));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)(acc[0] + c0[0] + c1[0]); }
}

__global__ void packs(const float* x, unsigned* out) {
  // path 1: two v_cvt_pknorm_i16_f32 (snorm16 = round(x * 32767)) + v_perm_b32 picking the high byte of every half
  const float a = x[0], b = x[1], c = x[2], d = x[3];
  const s2 p0 = __builtin_amdgcn_cvt_pknorm_i16(a, b), p1 = __builtin_amdgcn_cvt_pknorm_i16(c, d);
  const unsigned w0 = __builtin_bit_cast(unsigned, p0), w1 = __builtin_bit_cast(unsigned, p1);
  out[0] = __builtin_amdgcn_perm(w1, w0, 0x07050301);      // bytes: w0.b1, w0.b3, w1.b1, w1.b3
  out[1] = w0; out[2] = w1;
}

int main() {
  static signed char A[32 * 32], B[32 * 32];
  static int C[32 * 32];
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 9) & 0xff) - 128; };
  for (auto& v : A) v = (signed char)rnd();
  for (auto& v : B) v = (signed char)rnd();
  signed char *dA, *dB; int* dC;
  hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dC, sizeof(C));
  hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
  layout<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
  long long bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    int acc = 0;
    for (int k = 0; k < 32; ++k) acc += (int)A[i * 32 + k] * (int)B[k * 32 + j];
    bad += acc != C[i * 32 + j];
  }
  printf("i8 32x32x32 layout (byte j of lane (r,h) <-> k = 16h + j): %lld mismatches of 1024\n", bad);
  long long* dT; hipMalloc(&dT, 16);
  long long T[2];
  const int iters = 2000;
  const char* names[3] = {"i32_32x32x32_i8", "scale_f32_32x32x64_f8f6f4 (fp8)", "f32_32x32x16_f16"};
  for (int kind = 0; kind < 3; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) timing<0><<<256, 256>>>(dT, iters);
      if (kind == 1) timing<1><<<256, 256>>>(dT, iters);
      if (kind == 2) timing<2><<<256, 256>>>(dT, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(T, dT, 16, hipMemcpyDeviceToHost);
    printf("%-34s %.1f cycles per instruction (s_memtime ticks)\n", names[kind], (double)T[0] / (iters * 4.0));
  }
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) chain<0><<<256, 256>>>(dT, iters);
      if (mode == 1) chain<1><<<256, 256>>>(dT, iters);
      if (mode == 2) chain<2><<<256, 256>>>(dT, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(T, dT, 16, hipMemcpyDeviceToHost);
    printf("group chain (4 f16 + 2 fp8-64), %s: %.1f cycles per group (matrix work 256)\n",
           mode == 0 ? "fp8 pair on ONE accumulator" : mode == 1 ? "fp8 pair on TWO accumulators" : "fp8 interleaved 2-1-2-1", (double)T[0] / iters);
  }
  float hx[4] = {0.5f, -0.5f, 0.999f, -0.01f}, *dx; unsigned hu[3], *du;
  hipMalloc(&dx, 16); hipMalloc(&du, 12);
  hipMemcpy(dx, hx, 16, hipMemcpyHostToDevice);
  packs<<<1, 1>>>(dx, du);
  hipMemcpy(hu, du, 12, hipMemcpyDeviceToHost);
  printf("pack: x = 0.5 -0.5 0.999 -0.01 -> bytes %d %d %d %d (snorm16 words %08x %08x)\n", (signed char)(hu[0] & 0xff),
         (signed char)((hu[0] >> 8) & 0xff), (signed char)((hu[0] >> 16) & 0xff), (signed char)(hu[0] >> 24), hu[1], hu[2]);
  return 0;
}
