// Semantics of v_cvt_scalef32_pk_fp8_f32 on gfx950: is the result fp8(src * scale) or fp8(src / scale)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(float* out) {
  const float a = 1.5e-4f, b = -3.0e-5f;
  v2s old = {0, 0};
  v2s r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, 2048.f, false);
  v2s r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, 1.f / 2048.f, false);
  int w1 = (unsigned short)r1[0], w2 = (unsigned short)r2[0];
  out[0] = __builtin_amdgcn_cvt_f32_fp8(w1, 0); out[1] = __builtin_amdgcn_cvt_f32_fp8(w1, 1);
  out[2] = __builtin_amdgcn_cvt_f32_fp8(w2, 0); out[3] = __builtin_amdgcn_cvt_f32_fp8(w2, 1);
  int w3 = __builtin_amdgcn_cvt_pk_fp8_f32(a * 2048.f, b * 2048.f, 0, false);
  out[4] = __builtin_amdgcn_cvt_f32_fp8(w3, 0); out[5] = __builtin_amdgcn_cvt_f32_fp8(w3, 1);
}
int main() {
  float* o; (void)hipMalloc(&o, 64); k<<<1, 1>>>(o);
  float h[6]; (void)hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
  printf("scale 2048: %g %g | scale 1/2048: %g %g | mul then cvt: %g %g  (inputs x 2048 = %g %g)\n", h[0], h[1], h[2], h[3], h[4], h[5],
         1.5e-4f * 2048, -3.0e-5f * 2048);
  return 0;
}
