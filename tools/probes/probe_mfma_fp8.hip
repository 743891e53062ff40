// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands) on gfx950: operand lane/byte mapping, scale semantics, and the
// f32 -> fp8 conversion builtins.  Prints the max error against a host evaluation under the ASSUMED mapping
//   lane l (r = l & 31, h = l >> 5) byte j (0..31) of A  <->  A[row r][k = 32 h + j];  B likewise with column r.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, float* C, int sa, int sb, float* cvt_out) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v8i a, b;
  for (int d = 0; d < 8; ++d) {
    int wa = 0, wb = 0;
    // bytes 4d .. 4d+3  <->  k = 32h + 4d + {0,1,2,3}
    wa = __builtin_amdgcn_cvt_pk_fp8_f32(A[r * 64 + 32 * h + 4 * d + 0], A[r * 64 + 32 * h + 4 * d + 1], wa, false);
    wa = __builtin_amdgcn_cvt_pk_fp8_f32(A[r * 64 + 32 * h + 4 * d + 2], A[r * 64 + 32 * h + 4 * d + 3], wa, true);
    wb = __builtin_amdgcn_cvt_pk_fp8_f32(B[(32 * h + 4 * d + 0) * 32 + r], B[(32 * h + 4 * d + 1) * 32 + r], wb, false);
    wb = __builtin_amdgcn_cvt_pk_fp8_f32(B[(32 * h + 4 * d + 2) * 32 + r], B[(32 * h + 4 * d + 3) * 32 + r], wb, true);
    a[d] = wa; b[d] = wb;
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  // C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
  if (l == 0) {   // conversion semantics: value of the fp8 byte, via the scale-free path  x -> fp8 -> f32
    const float t[4] = {0.3f, -1.7f, 300.f, 0.0019f};
    for (int i = 0; i < 4; ++i) {
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(t[i], 0.f, 0, false);
      cvt_out[i] = __builtin_amdgcn_cvt_f32_fp8(w, 0);
    }
  }
}

static float fp8_round(float x) {   // OCP e4m3fn, RNE, saturating at 448
  if (x == 0.f) return 0.f;
  float a = fabsf(x);
  int e; frexpf(a, &e); e -= 1;          // a = m * 2^e, m in [1, 2)
  if (e < -6) e = -6;
  float q = ldexpf(1.f, e - 3);
  float r = nearbyintf(a / q) * q;
  if (r > 448.f) r = 448.f;
  return copysignf(r, x);
}

int main() {
  static float A[32 * 64], B[64 * 32], C[32 * 32], ref[32 * 32], cv[4];
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
  for (auto& v : A) v = rnd() * 3.f;
  for (auto& v : B) v = rnd() * 0.4f;
  float *dA, *dB, *dC, *dcv;
  hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dC, sizeof(C)); hipMalloc(&dcv, sizeof(cv));
  hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
  const int cases[3][2] = {{127, 127}, {127 - 5, 127}, {127 + 2, 127 - 11}};
  for (auto& cs : cases) {
    k<<<1, 64>>>(dA, dB, dC, cs[0], cs[1], dcv);
    hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost); hipMemcpy(cv, dcv, sizeof(cv), hipMemcpyDeviceToHost);
    const double sc = ldexp(1.0, cs[0] - 127) * ldexp(1.0, cs[1] - 127);
    double err = 0, mag = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double acc = 0;
      for (int kk = 0; kk < 64; ++kk) acc += (double)fp8_round(A[i * 64 + kk]) * (double)fp8_round(B[kk * 32 + j]);
      acc *= sc;
      ref[i * 32 + j] = (float)acc;
      err = fmax(err, fabs(acc - C[i * 32 + j])); mag = fmax(mag, fabs(acc));
    }
    printf("scale_a %d scale_b %d: max |C - ref| = %.3e (max |ref| %.3e)\n", cs[0], cs[1], err, mag);
  }
  printf("cvt round trip: 0.3 -> %g, -1.7 -> %g, 300 -> %g, 0.0019 -> %g (host model: %g %g %g %g)\n", cv[0], cv[1], cv[2], cv[3],
         fp8_round(0.3f), fp8_round(-1.7f), fp8_round(300.f), fp8_round(0.0019f));
  return 0;
}
