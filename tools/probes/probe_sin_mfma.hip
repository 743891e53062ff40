// Hardware probe (not product code): accuracy of v_sin_f32 / v_cos_f32 / v_exp_f32 and
// the f16 MFMA 32x32x16 operand/accumulator maps + subnormal behaviour on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdint>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_sin(const float* x, float* s, float* c, float* e, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = __builtin_amdgcn_sinf(x[i]);   // sin(2*pi*x)
  c[i] = __builtin_amdgcn_cosf(x[i]);
  e[i] = __builtin_amdgcn_exp2f(x[i]);
}

// D = A(32x16) * B(16x32), one wave. A given as row-major [32][16] f16, B as [16][32] f16.
__global__ void k_mfma(const _Float16* A, const _Float16* B, float* D) {
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int g = 0; g < 16; ++g) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; D[row * 32 + r] = acc[g]; }
}

int main() {
  // ---- sin/cos/exp2 accuracy
  const int n = 1 << 22;
  std::vector<float> hx(n), hs(n), hc(n), he(n);
  for (int i = 0; i < n; ++i) hx[i] = -0.5f + (float)i / n;           // [-0.5,0.5) revolutions
  float *dx, *ds, *dc, *de;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&de, n * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  k_sin<<<n / 256, 256>>>(dx, ds, dc, de, n);
  hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(he.data(), de, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, ee = 0, ess = 0; const double PI2 = 6.283185307179586476925;
  double es_small = 0;
  for (int i = 0; i < n; ++i) {
    double rs = sin(PI2 * (double)hx[i]), rc = cos(PI2 * (double)hx[i]);
    es = fmax(es, fabs(hs[i] - rs)); ec = fmax(ec, fabs(hc[i] - rc));
    ess += (hs[i] - rs) * (hs[i] - rs);
    if (fabs(hx[i]) < 0.05) es_small = fmax(es_small, fabs(hs[i] - rs));
    double re = exp2((double)hx[i]); ee = fmax(ee, fabs(he[i] - re) / re);
  }
  printf("v_sin_f32 max abs err [-0.5,0.5) = %.3e rms=%.3e (|x|<0.05: %.3e)\n", es, sqrt(ess / n), es_small);
  printf("v_cos_f32 max abs err = %.3e\n", ec);
  printf("v_exp_f32 max rel err = %.3e\n", ee);
  // larger-range sin: x in [-40,40) revolutions
  for (int i = 0; i < n; ++i) hx[i] = -40.f + 80.f * (float)i / n;
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  k_sin<<<n / 256, 256>>>(dx, ds, dc, de, n);
  hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
  es = 0; for (int i = 0; i < n; ++i) es = fmax(es, fabs(hs[i] - sin(PI2 * (double)hx[i])));
  printf("v_sin_f32 max abs err [-40,40) rev = %.3e\n", es);

  // ---- MFMA map check with asymmetric integer data + subnormal probe
  std::vector<_Float16> A(32 * 16), B(16 * 32); std::vector<float> D(32 * 32), R(32 * 32);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (_Float16)((i * 3 + k * 5) % 7 - 3);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (_Float16)((k * 2 + j * 7) % 5 - 2);
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float r = 0; for (int k = 0; k < 16; ++k) r += (float)A[i * 16 + k] * (float)B[k * 32 + j];
    if (r != D[i * 32 + j]) ++bad;
  }
  printf("mfma_f32_32x32x16_f16 map mismatches: %d / 1024\n", bad);
  // subnormal: A = 2^-20 (fp16 subnormal), B = 2^10 -> product 2^-10 exactly if not flushed
  for (auto& v : A) v = (_Float16)0; for (auto& v : B) v = (_Float16)0;
  A[0] = (_Float16)9.5367431640625e-07f; B[0] = (_Float16)1024.f;
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  printf("f16 subnormal A (2^-20)*1024 -> %.6e (expect 9.765625e-04 if subnormals kept, host A=%g)\n", D[0], (float)A[0]);
  hipDeviceSynchronize();
  return 0;
}
