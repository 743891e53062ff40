#include <hip/hip_runtime.h>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* out) {
  float a = x[threadIdx.x * 2], b = x[threadIdx.x * 2 + 1];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 v = {a, b};
  half2v h = __builtin_convertvector(v, half2v);
  float r0, r1;
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(a));
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(b));
  out[threadIdx.x * 4 + 0] = r0; out[threadIdx.x * 4 + 1] = r1;
  out[threadIdx.x * 4 + 2] = a - (float)h[0]; out[threadIdx.x * 4 + 3] = b - (float)h[1];
}
int main() {
  float *x, *o; hipMalloc(&x, 64 * 2 * 4); hipMalloc(&o, 64 * 4 * 4);
  float hx[128]; for (int i = 0; i < 128; ++i) hx[i] = (float)(i * 0.013731 - 0.7) * (i % 3 == 0 ? 1e-3f : 1.f);
  hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
  k<<<1, 64>>>(x, o);
  float ho[256]; hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) { if (ho[i*4] != ho[i*4+2] || ho[i*4+1] != ho[i*4+3]) ++bad; }
  printf("mismatches %d  sample %g %g %g %g\n", bad, ho[4], ho[6], ho[5], ho[7]);
  return 0;
}
