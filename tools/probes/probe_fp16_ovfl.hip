// MODE.FP16_OVFL on gfx950: does an overflowing fp32 -> fp16 conversion (the compiler's v_cvt_pk_f16_f32 / v_cvt_f16_f32) clamp to
// +-65504 instead of producing infinity, and are true infinities / NaNs preserved?   hipcc --offload-arch=gfx950 -O3 probe_fp16_ovfl.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, float* out, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // hwreg(HW_REG_MODE, 23, 1)
  const int i = threadIdx.x;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const float a = in[2 * i], b = in[2 * i + 1];
  h2 h = {(_Float16)a, (_Float16)b};                   // v_cvt_pk_f16_f32
  asm volatile("" : "+v"(h));
  out[2 * i] = (float)h.x;
  out[2 * i + 1] = (float)h.y;
}
int main() {
  const float host[8] = {1.0f, 65504.f, 65520.f, 1e6f, -1e9f, INFINITY, -INFINITY, NAN};
  float *in, *out, res[8];
  hipMalloc(&in, sizeof host); hipMalloc(&out, sizeof host);
  hipMemcpy(in, host, sizeof host, hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, in, out, ovfl);
    hipMemcpy(res, out, sizeof res, hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d:", ovfl);
    for (int i = 0; i < 8; ++i) printf(" %g->%g", host[i], res[i]);
    printf("\n");
  }
  return 0;
}
