// v_cvt_scalef32_pk_fp8_f32 against multiply + v_cvt_pk_fp8_f32: both word selections, preservation of the other word, a sweep
// of magnitudes at the scale 2^-17 (results x 2^17).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, int n) {
  for (int i = 0; i < n; ++i) {
    const float a = in[2 * i], b = in[2 * i + 1];
    const int old = 0x11223344;
    const v2s o = __builtin_bit_cast(v2s, old);
    out[4 * i + 0] = (unsigned)__builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, a, b, 0x1p-17f, false));
    out[4 * i + 1] = (unsigned)__builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, a, b, 0x1p-17f, true));
    out[4 * i + 2] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a * 0x1p17f, b * 0x1p17f, old, false);
    out[4 * i + 3] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a * 0x1p17f, b * 0x1p17f, old, true);
  }
}
int main() {
  const int n = 12;
  float h[2 * n]; unsigned r[4 * n];
  for (int i = 0; i < n; ++i) { h[2 * i] = ldexpf(1.37f, -12 - i); h[2 * i + 1] = -ldexpf(1.9f, -13 - i); }
  float* d; unsigned* o; (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 1>>>(d, o, n);
  (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("a=%.3e b=%.3e (x2^17: %8.4f %8.4f)  scaled lo %08x hi %08x | mul+cvt lo %08x hi %08x %s\n", h[2 * i], h[2 * i + 1], h[2 * i] * 131072.f,
           h[2 * i + 1] * 131072.f, r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], (r[4 * i] == r[4 * i + 2] && r[4 * i + 1] == r[4 * i + 3]) ? "" : "  <-- differ");
  return 0;
}
